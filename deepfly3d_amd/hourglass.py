"""Host side of the stacked-hourglass engine (a2): parameter packing + the forward call.

Mirrors what `df2d.inference` does around its model (call site reference df3d/core.py:177-185): take a
bearpaw-style `state_dict` (keys as in oracle/hourglass_torch.py / SURVEY.md App. B), run the network on
batches of views, hand back heat-maps.  All arithmetic happens in libdf3d_hip.so; torch is used only to own
device memory and the stream.  BatchNorm folding (pure parameter preprocessing, float64 numpy) happens here.
"""
import ctypes
import os

import numpy as np
import torch

from . import _native

BN_EPS = 1e-5


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().double().numpy()
    return np.asarray(v, dtype=np.float64)


def _bn_affine(sd, prefix):
    """Eval-mode BatchNorm as y = x * s + t."""
    g, b = _np(sd[prefix + ".weight"]), _np(sd[prefix + ".bias"])
    m, v = _np(sd[prefix + ".running_mean"]), _np(sd[prefix + ".running_var"])
    s = g / np.sqrt(v + BN_EPS)
    return s, b - m * s


def _following_bn(name):
    """Name of the BatchNorm that directly follows convolution `name` (and is folded into it), or None."""
    if name == "conv1":
        return "bn1"
    if name.endswith(".conv1"):
        return name[: -len("conv1")] + "bn2"
    if name.endswith(".conv2"):
        return name[: -len("conv2")] + "bn3"
    if name.startswith("fc.") and name.endswith(".0"):
        return name[:-1] + "1"
    return None


def _input_bn(name):
    return name[: -len("conv1")] + "bn1" if name.endswith(".conv1") and name != "conv1" else None


def pack_state_dict(engine_handle, state_dict):
    """Fill the engine's float32 parameter blob from a state_dict, following the manifest the C library exports."""
    lib = _native.load()
    n = lib.df3d_hg_num_params(engine_handle)
    blob = np.zeros(lib.df3d_hg_blob_floats(engine_handle), dtype=np.float32)
    d = _native.HGParam()
    cache = {}
    for i in range(n):
        _native.check(lib.df3d_hg_param_desc(engine_handle, i, ctypes.byref(d)), "df3d_hg_param_desc")
        name = d.name.decode()
        if name not in cache:
            w = _np(state_dict[name + ".weight"])  # (cout, cin, kh, kw)
            b = _np(state_dict[name + ".bias"])
            bn = _following_bn(name)
            if bn is not None:
                s, t = _bn_affine(state_dict, bn)
                w = w * s[:, None, None, None]
                b = b * s + t
            cache = {name: (w, b)}
        w, b = cache[name]
        view = blob[d.offset : d.offset + d.count]
        if d.kind == 0:
            cout, cin, kh, kw = w.shape
            if (cout, cin, kh * kw) != (d.cout, d.cin, d.taps):
                raise ValueError(f"{name}: state_dict shape {w.shape} does not match engine ({d.cout},{d.cin},{d.taps})")
            if d.taps == 49:  # stem: [148][64], k = ky*21 + kx*3 + c
                packed = np.zeros((d.count // 64, 64))  # rows 0..146 used (the slot is larger: see df3d_hip.h)
                packed[:147] = w.transpose(2, 3, 1, 0).reshape(147, 64)
            else:  # [tap][cout_pad][cin_pad]
                packed = np.zeros((d.taps, d.cout_pad, d.cin_pad))
                packed[:, :cout, :cin] = w.transpose(2, 3, 0, 1).reshape(d.taps, cout, cin)
                if d.kperm:  # K order the bf16 fused bottleneck expects (see include/df3d_hip.h: df3d_hg_param.kperm)
                    pos = np.arange(32)
                    q, hh, e = pos // 16, (pos // 8) % 2, pos % 8
                    src = 16 * q + 8 * (e // 4) + 4 * hh + (e % 4)
                    idx = (32 * np.arange(d.cin_pad // 32)[:, None] + src[None, :]).ravel()
                    packed = packed[:, :, idx]
            view[:] = packed.ravel().astype(np.float32)
        elif d.kind == 1:
            view[: d.cout] = b.astype(np.float32)
        else:
            s, t = _bn_affine(state_dict, _input_bn(name))
            view[: d.cin] = (s if d.kind == 2 else t).astype(np.float32)
    return blob


class HourglassEngine:
    """The device engine: `forward(images_nhwc) -> heat-maps (n, 19, H/4, W/4)` on the current torch stream."""

    def __init__(self, state_dict, dtype="f32", num_stacks=2, device=None, height=256, width=512, row_bytes=0, fuse=True, fuse_upadd=None, ring=None, l1=None,
                 chain_views=None, split1=None, w2d=None):
        _native.require_gpu()
        self.lib = _native.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = dtype
        code = {"f32": _native.DF3D_DTYPE_F32, "bf16": _native.DF3D_DTYPE_BF16, "f16": _native.DF3D_DTYPE_F16}[dtype]
        h = ctypes.c_void_p()
        _native.check(self.lib.df3d_hg_create(code, num_stacks, ctypes.byref(h)), "df3d_hg_create")
        self.h = h
        self.height, self.width = height, width
        _native.check(self.lib.df3d_hg_set_input(self.h, height, width), "df3d_hg_set_input")
        if not fuse or os.environ.get("DF3D_FUSE") == "0":   # (DF3D_FUSE=0: developer switch, every convolution as a launch of its own)
            _native.check(self.lib.df3d_hg_set_option(self.h, b"fuse", 0), "df3d_hg_set_option")
        if fuse_upadd is not None:  # default: the library's choice (on)
            # True / 1: added in the epilogue of the bottleneck that produces the up-path tensor (default); 2: folded into the input load of
            # the consuming bottleneck (round 2's form); False / 0: a pass of its own
            _native.check(self.lib.df3d_hg_set_option(self.h, b"fuse_upadd", int(fuse_upadd)), "df3d_hg_set_option")
        if ring is not None:  # default: the library's choice (LDS-DMA weight ring in the 256 -> 128 -> 128 -> 256 bottlenecks)
            _native.check(self.lib.df3d_hg_set_option(self.h, b"ring", 1 if ring else 0), "df3d_hg_set_option")
        if l1 is not None:  # default: on (bf16): layer1 with LDS-resident weights, writing only the pooled tensor its consumer reads
            _native.check(self.lib.df3d_hg_set_option(self.h, b"l1", 1 if l1 else 0), "df3d_hg_set_option")
        if row_bytes:
            _native.check(self.lib.df3d_hg_set_option(self.h, b"row_bytes", row_bytes), "df3d_hg_set_option")
        if split1 is None and os.environ.get("DF3D_SPLIT1"):
            split1 = int(os.environ["DF3D_SPLIT1"])
        if split1 is not None and dtype == "f32":  # fp32: conv1 of the identity-skip bottlenecks as a launch of its own (csrc/hg_c1_f32.h), bit-identical
            _native.check(self.lib.df3d_hg_set_option(self.h, b"split1", 1 if split1 else 0), "df3d_hg_set_option")
        if w2d is None and os.environ.get("DF3D_W2D"):
            w2d = int(os.environ["DF3D_W2D"])
        if w2d is not None and dtype != "f32":  # 16-bit: the 3x3's weights of the ring bottlenecks as direct per-wave fragment loads (csrc/hg_bt_ring.h), bit-identical
            _native.check(self.lib.df3d_hg_set_option(self.h, b"w2d", 1 if w2d else 0), "df3d_hg_set_option")
        if chain_views is None and os.environ.get("DF3D_CHAIN_VIEWS"):
            chain_views = int(os.environ["DF3D_CHAIN_VIEWS"])
        if chain_views is not None:  # chains of full-resolution steps in chunks of this many views (0 = whole batch per launch)
            _native.check(self.lib.df3d_hg_set_option(self.h, b"chain_views", int(chain_views)), "df3d_hg_set_option")
        blob = pack_state_dict(self.h, state_dict)
        self.blob = torch.from_numpy(blob).to(self.device)
        lowp_bytes = self.lib.df3d_hg_lowp_bytes(self.h)
        self.lowp = torch.empty(max(lowp_bytes, 16), dtype=torch.uint8, device=self.device) if lowp_bytes else None
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _native.check(
                self.lib.df3d_hg_set_weights(self.h, self.blob.data_ptr(), self.lowp.data_ptr() if lowp_bytes else None, stream),
                "df3d_hg_set_weights",
            )
        self._ws = None
        self.num_classes = 19

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.df3d_hg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _workspace(self, n):
        need = self.lib.df3d_hg_workspace_bytes(self.h, n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def work(self, n):
        f, b = ctypes.c_double(), ctypes.c_double()
        _native.check(self.lib.df3d_hg_work(self.h, n, ctypes.byref(f), ctypes.byref(b)))
        return f.value, b.value

    def steps(self):
        out = []
        buf = ctypes.create_string_buffer(96)
        hwc = (ctypes.c_int * 3)()
        for i in range(self.lib.df3d_hg_num_steps(self.h)):
            _native.check(self.lib.df3d_hg_step_desc(self.h, i, buf, 96, hwc))
            out.append((buf.value.decode(), tuple(hwc)))
        return out

    def _check_images(self, images):
        if not (images.is_cuda and images.dtype == torch.float32 and images.is_contiguous()):
            raise ValueError("images must be a contiguous float32 CUDA tensor [n, H, W, 3]")
        if tuple(images.shape[1:]) != (self.height, self.width, 3):
            raise ValueError(f"images must be [n, {self.height}, {self.width}, 3], got {tuple(images.shape)}")

    def forward(self, images, out=None):
        self._check_images(images)
        n = images.shape[0]
        if out is None:
            out = torch.empty((n, self.num_classes, self.height // 4, self.width // 4), dtype=torch.float32, device=self.device)
        ws = self._workspace(n)
        with torch.cuda.device(self.device):  # kernels launch on the current HIP device
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _native.check(
                self.lib.df3d_hg_forward(self.h, images.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                "df3d_hg_forward",
            )
        return out

    def forward_u8(self, frames_u8, flip=None, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), out=None, resize="bilinear"):
        """Heat-maps straight from camera frames: frames_u8 [n, H, W] or [n, H, W, C] uint8 cuda, flip [n] uint8 or None.  The stem
        samples the frames with df3d_preprocess_u8's arithmetic: bit for bit forward(preprocess(frames)), one kernel and one
        float image per batch fewer."""
        import ctypes

        if frames_u8.dim() == 3:
            frames_u8 = frames_u8.unsqueeze(-1)
        if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.dim() == 4):
            raise ValueError("frames must be a contiguous uint8 CUDA tensor [n, H, W(, C)]")
        if frames_u8.device != self.device:
            raise ValueError(f"frames live on {frames_u8.device}, the engine on {self.device}")
        n, fh, fw, fc = frames_u8.shape
        if fc not in (1, 3) or n < 1:
            raise ValueError("frames must have 1 or 3 channels and at least one view")
        if out is None:
            out = torch.empty((n, self.num_classes, self.height // 4, self.width // 4), dtype=torch.float32, device=self.device)
        fl = None
        if flip is not None:
            fl = flip.to(device=self.device, dtype=torch.uint8).contiguous()
        ws = self._workspace(n)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _native.check(
                self.lib.df3d_hg_forward_u8(self.h, frames_u8.data_ptr(), fl.data_ptr() if fl is not None else None, n, fh, fw, fc,
                                            (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std), _native.RESIZE_MODES[resize], out.data_ptr(), ws.data_ptr(),
                                            ws.numel(), stream),
                "df3d_hg_forward_u8",
            )
        return out

    def forward_upto(self, images, upto):
        """Layer-wise parity helper: output of plan step `upto - 1` as float32 NHWC (or NCHW for the final score)."""
        self._check_images(images)
        n = images.shape[0]
        name, (h, w, c) = self.steps()[upto - 1]
        last_nchw = name.startswith("score.") and upto == self.lib.df3d_hg_num_steps(self.h)
        shape = (n, c, h, w) if last_nchw else (n, h, w, c)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        ws = self._workspace(n)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _native.check(
                self.lib.df3d_hg_forward_upto(self.h, images.data_ptr(), n, upto, out.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                "df3d_hg_forward_upto",
            )
        return out
