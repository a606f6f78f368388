"""Development probe: what a reduced-precision engine returns when its activations leave the IEEE-half range (two BatchNorms of the synthetic network scaled by 2e4): per-step maxima of the f32 / f16 / f32s engines.  On gfx950 the half conversions saturate: f16 / f32s stay FINITE and wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict
sd=synthetic_state_dict(0); hot=dict(sd)
for k in ("bn1.weight","bn1.bias","layer1.0.bn1.weight","layer1.0.bn1.bias"): hot[k]=sd[k]*2e4
img=torch.rand((2,256,512,3),generator=torch.Generator().manual_seed(2)).cuda()
for dt in ("f32","f16","f32s"):
    e=HourglassEngine(hot,dtype=dt,device="cuda:0")
    for k,(name,_) in enumerate(e.steps()[:6],start=1):
        t=e.forward_upto(img,k)
        print(dt,k,name,float(t.abs().max()), bool(torch.isfinite(t).all()))
    hm=e.forward(img); print(dt,"hm",float(hm.abs().max()), bool(torch.isfinite(hm).all()))
