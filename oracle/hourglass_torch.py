"""CPU oracle for the 2-D half of the hot path: stacked-hourglass forward in torch (fp32, NCHW).
TEST INFRASTRUCTURE ONLY -- the product package never imports this file.

PARITY UNPINNED for the network itself: the reference delegates the model to the un-vendored
`nely-df2d >= 0.14` (reference setup.py:31; call site df3d/core.py:177-185) and the trained
weights `sh8_deepfly.tar` (path only, reference df3d/config.py:30-32) are not in the checkout, so
no golden vector of the reference can exercise it here.  What the reference itself pins and this
file follows: 2 stacks (df3d/config.py:33), 19 output maps (df3d/config.py:36), 64x128 heat-maps
(df3d/config.py:18) from 256x512 inputs, hard arg-max + peak value (README.md:404).  The layer
structure is the published stacked-hourglass of Newell et al. in its pre-activation-bottleneck
PyTorch form that df2d uses: state_dict keys are kept compatible
(`conv1, bn1, layer{1,2,3}.0.*, hg.{s}.hg.{lvl}.{k}.0.*, res.{s}.0.*, fc.{s}.{0,1}, score.{s},
fc_.{s}, score_.{s}`) so trained weights can be loaded when a user has them.

The HIP engine is compared against this module layer by layer with seeded random parameters.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_CLASSES = 19
NUM_STACKS = 2
FEATS = 128  # bottleneck planes; trunk width = 2 * FEATS
DEPTH = 4


class PreActBottleneck(nn.Module):
    """x -> conv1x1(relu(bn1 x)) -> conv3x3(relu(bn2 .)) -> conv1x1(relu(bn3 .)) + skip(x)."""

    def __init__(self, cin, planes):
        super().__init__()
        cout = 2 * planes
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv1 = nn.Conv2d(cin, planes, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.bn3 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1)
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1)) if cin != cout else None

    def forward(self, x):
        t = self.conv1(F.relu(self.bn1(x)))
        t = self.conv2(F.relu(self.bn2(t)))
        t = self.conv3(F.relu(self.bn3(t)))
        return t + (x if self.downsample is None else self.downsample(x))


def _unit(cin, planes):
    return nn.Sequential(PreActBottleneck(cin, planes))


class Hourglass(nn.Module):
    def __init__(self, planes=FEATS, depth=DEPTH):
        super().__init__()
        self.depth = depth
        levels = []
        for lvl in range(depth):  # lvl 0 is the innermost (lowest-resolution) level
            n = 4 if lvl == 0 else 3
            levels.append(nn.ModuleList([_unit(2 * planes, planes) for _ in range(n)]))
        self.hg = nn.ModuleList(levels)

    def _level(self, n, x):
        blocks = self.hg[n - 1]
        up1 = blocks[0](x)
        low = blocks[1](F.max_pool2d(x, 2, stride=2))
        low = self._level(n - 1, low) if n > 1 else blocks[3](low)
        low = blocks[2](low)
        return up1 + F.interpolate(low, scale_factor=2, mode="nearest")

    def forward(self, x):
        return self._level(self.depth, x)


class HourglassNet(nn.Module):
    def __init__(self, num_stacks=NUM_STACKS, num_classes=NUM_CLASSES, feats=FEATS):
        super().__init__()
        self.num_stacks = num_stacks
        ch = 2 * feats
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _unit(64, 64)
        self.layer2 = _unit(128, feats)
        self.layer3 = _unit(ch, feats)
        self.hg = nn.ModuleList([Hourglass(feats) for _ in range(num_stacks)])
        self.res = nn.ModuleList([_unit(ch, feats) for _ in range(num_stacks)])
        self.fc = nn.ModuleList([nn.Sequential(nn.Conv2d(ch, ch, 1), nn.BatchNorm2d(ch)) for _ in range(num_stacks)])
        self.score = nn.ModuleList([nn.Conv2d(ch, num_classes, 1) for _ in range(num_stacks)])
        self.fc_ = nn.ModuleList([nn.Conv2d(ch, ch, 1) for _ in range(num_stacks - 1)])
        self.score_ = nn.ModuleList([nn.Conv2d(num_classes, ch, 1) for _ in range(num_stacks - 1)])

    def stem(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(self.layer1(x), 2, stride=2)
        return self.layer3(self.layer2(x))

    def forward(self, x, return_all=False):
        outs = []
        x = self.stem(x)
        for s in range(self.num_stacks):
            y = self.res[s](self.hg[s](x))
            y = F.relu(self.fc[s](y))
            score = self.score[s](y)
            outs.append(score)
            if s < self.num_stacks - 1:
                x = x + self.fc_[s](y) + self.score_[s](score)
        return outs if return_all else outs[-1]


def seeded_state_dict(seed=0, num_stacks=NUM_STACKS, gain=0.6):
    """Seeded synthetic parameters: He-normal conv weights, small biases,
    BN gamma~U[0.5,1.5], beta,mean~N(0,0.1), var~U[0.5,1.5].  `gain` (0.6) scales the He std so
    the final heat-maps of the 2-stack net stay O(10).  Returns {name: float32 tensor}."""
    net = HourglassNet(num_stacks=num_stacks)
    g = torch.Generator().manual_seed(seed)
    sd = net.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
            continue
        if v.ndim == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            # residual branches are damped so 30+ stacked residual adds stay O(1)
            scale = (2.0 / fan_in) ** 0.5 * (gain if not k.endswith("conv3.weight") else 0.5 * gain)
            out[k] = torch.randn(v.shape, generator=g) * scale
        elif k.endswith("running_var") or (k.endswith("weight") and v.ndim == 1):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        else:  # conv bias, bn bias, running_mean
            out[k] = torch.randn(v.shape, generator=g) * 0.1
    return out


def build(seed=0, num_stacks=NUM_STACKS):
    net = HourglassNet(num_stacks=num_stacks)
    net.load_state_dict(seeded_state_dict(seed, num_stacks))
    net.eval()
    return net


@torch.no_grad()
def forward_nhwc(net, images_nhwc):
    """images_nhwc: (N, 256, 512, 3) float32 -> heat-maps (N, 19, 64, 128) float32."""
    x = torch.as_tensor(images_nhwc, dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
    return net(x)


# ------------------------------------------------------------------------------------------------
# traced forward: the same arithmetic as HourglassNet.forward, recording every intermediate under the
# name of the engine plan step that produces it (deepfly3d_amd/csrc/hourglass.hip), NHWC float32.
# A convolution followed by BN(+ReLU) is recorded AFTER that BN/ReLU because the engine folds them.
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def forward_traced(net, images_nhwc):
    rec = {}

    def keep(name, t):
        rec[name] = t.permute(0, 2, 3, 1).contiguous()
        return t

    def block(name, seq, x):
        b = seq[0]
        t = keep(name + ".conv1", F.relu(b.bn2(b.conv1(F.relu(b.bn1(x))))))
        t = keep(name + ".conv2", F.relu(b.bn3(b.conv2(t))))
        skip = x if b.downsample is None else keep(name + ".downsample.0", b.downsample(x))
        return keep(name + ".conv3", b.conv3(t) + skip)

    def level(prefix, hg, n, x):
        lv = f"{prefix}.{n - 1}"
        blocks = hg.hg[n - 1]
        up1 = block(lv + ".0.0", blocks[0], x)
        low = keep(lv + ".pool", F.max_pool2d(x, 2, stride=2))
        low = block(lv + ".1.0", blocks[1], low)
        low = level(prefix, hg, n - 1, low) if n > 1 else block(lv + ".3.0", blocks[3], low)
        low = block(lv + ".2.0", blocks[2], low)
        return keep(lv + ".upadd", up1 + F.interpolate(low, scale_factor=2, mode="nearest"))

    x = torch.as_tensor(images_nhwc, dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
    x = keep("conv1", F.relu(net.bn1(net.conv1(x))))
    x = block("layer1.0", net.layer1, x)
    x = keep("maxpool", F.max_pool2d(x, 2, stride=2))
    x = block("layer2.0", net.layer2, x)
    x = block("layer3.0", net.layer3, x)
    for s in range(net.num_stacks):
        y = level(f"hg.{s}.hg", net.hg[s], net.hg[s].depth, x)
        y = block(f"res.{s}.0", net.res[s], y)
        y = keep(f"fc.{s}.0", F.relu(net.fc[s](y)))
        score = net.score[s](y)
        if s < net.num_stacks - 1:
            keep(f"score.{s}", score)
            t = keep(f"fc_.{s}", x + net.fc_[s](y))
            x = keep(f"score_.{s}", t + net.score_[s](score))
        else:
            rec[f"score.{s}"] = score.contiguous()  # the final heat-maps stay NCHW
    return rec
