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


class CheckpointMismatch(ValueError):
    """The state_dict is not the network the engine was built for (see `describe_state_dict`, `pack_state_dict`)."""


_IGNORED_SUFFIXES = ("num_batches_tracked",)


def describe_state_dict(state_dict):
    """What architecture a bearpaw-style hourglass state_dict describes, read off its key set and tensor shapes:
    {"num_stacks", "num_blocks" (bottlenecks per residual unit), "depth" (hourglass levels), "feats" (trunk width), "num_classes"}.
    Entries are None where the keys that would tell are absent.  (App. B's constants -- one block, depth 4, 256 features --
    are recall, not a reference fact; config.py:33,36 pin only the stacks and the 19 classes: a checkpoint is asked, not assumed.)"""
    import re

    def max_index(pattern):
        found = [int(m.group(1)) for k in state_dict for m in [re.match(pattern, k)] if m]
        return max(found) + 1 if found else None

    def shape(name):
        v = state_dict.get(name)
        return tuple(v.shape) if v is not None else None

    blocks = [max_index(r"layer[123]\.(\d+)\."), max_index(r"res\.\d+\.(\d+)\."), max_index(r"hg\.\d+\.hg\.\d+\.\d+\.(\d+)\.")]
    blocks = [b for b in blocks if b is not None]
    fc, sc = shape("fc.0.0.weight"), shape("score.0.weight")
    return {
        "num_stacks": max_index(r"hg\.(\d+)\."),
        "num_blocks": max(blocks) if blocks else None,
        "depth": max_index(r"hg\.\d+\.hg\.(\d+)\."),
        "feats": fc[0] // 2 if fc else None,
        "num_classes": sc[0] if sc else None,
    }


def _listing(keys, limit=16):
    """Keys grouped by module: `layer1.1.bn1.{bias,running_mean,...}`, at most `limit` modules."""
    groups = {}
    for k in sorted(keys):
        mod, _, field = k.rpartition(".")
        groups.setdefault(mod, []).append(field)
    items = [f"{m}.{f[0]}" if len(f) == 1 else f"{m}.{{{','.join(f)}}}" for m, f in groups.items()]
    return ", ".join(items[:limit]) + (f", ... ({len(keys)} keys in {len(items)} modules)" if len(items) > limit else "")


def pack_state_dict(engine_handle, state_dict, strict=True):
    """Fill the engine's float32 parameter blob from a state_dict, following the manifest the C library exports.

    strict (default): the state_dict must be EXACTLY the engine's network -- every key the manifest asks for present with the
    manifest's shape, and no parameter left over (only BatchNorm's `num_batches_tracked` counters are ignored).  A checkpoint with
    more stacks than the engine, with more than one bottleneck per residual unit (`layer1.1.*`, `hg.0.hg.3.0.1.*`: bearpaw's
    `num_blocks` > 1) or with another width would otherwise load "successfully" and produce wrong poses without a word; it raises
    CheckpointMismatch naming the missing and the unconsumed keys instead (reference: df3d/config.py:30-39 names the one
    checkpoint the reference loads; its architecture is not in the checkout)."""
    lib = _native.load()
    n = lib.df3d_hg_num_params(engine_handle)
    blob = np.zeros(lib.df3d_hg_blob_floats(engine_handle), dtype=np.float32)
    d = _native.HGParam()
    cache = {}
    consumed, missing, wrong = set(), set(), []

    def take(key):
        if key not in state_dict:
            missing.add(key)
            return None
        consumed.add(key)
        return _np(state_dict[key])

    def take_bn(prefix):
        parts = [take(prefix + sfx) for sfx in (".weight", ".bias", ".running_mean", ".running_var")]
        if any(x is None for x in parts):
            return None
        g, b, m, v = parts
        s = g / np.sqrt(v + BN_EPS)
        return s, b - m * s

    for i in range(n):
        _native.check(lib.df3d_hg_param_desc(engine_handle, i, ctypes.byref(d)), "df3d_hg_param_desc")
        name = d.name.decode()
        if name not in cache:
            w, b = take(name + ".weight"), take(name + ".bias")  # (cout, cin, kh, kw)
            bn = _following_bn(name)
            if bn is not None:
                st = take_bn(bn)
                if st is not None and w is not None and b is not None:
                    s, t = st
                    if s.shape[0] != w.shape[0]:
                        wrong.append(f"{bn}: {s.shape[0]} channels behind a convolution with {w.shape[0]} outputs")
                        w = None
                    else:
                        w = w * s[:, None, None, None]
                        b = b * s + t
                else:
                    w = None
            cache = {name: (w, b)}
        w, b = cache[name]
        view = blob[d.offset : d.offset + d.count]
        if d.kind == 0:
            if w is None:
                continue
            if w.ndim != 4 or (w.shape[0], w.shape[1], w.shape[2] * w.shape[3]) != (d.cout, d.cin, d.taps):
                wrong.append(f"{name}.weight: state_dict shape {tuple(w.shape)}, engine ({d.cout}, {d.cin}, {d.taps} taps)")
                continue
            cout, cin, kh, kw = w.shape
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
            if b is None or w is None:
                continue
            if b.shape != (d.cout,):
                wrong.append(f"{name}.bias: state_dict shape {tuple(b.shape)}, engine ({d.cout},)")
                continue
            view[: d.cout] = b.astype(np.float32)
        else:
            key = (_input_bn(name), "in")
            if key not in cache:
                cache[key] = take_bn(_input_bn(name))
            st = cache[key]
            if st is None:
                continue
            if st[0].shape != (d.cin,):
                wrong.append(f"{_input_bn(name)}: {st[0].shape[0]} channels, engine {d.cin}")
                continue
            view[: d.cin] = (st[0] if d.kind == 2 else st[1]).astype(np.float32)
    left = {k for k in state_dict if k not in consumed and not k.endswith(_IGNORED_SUFFIXES)} if strict else set()
    if strict or missing or wrong:   # (non-strict tolerates LEFT-OVER keys only: a parameter the engine needs is never optional)
        if missing or left or wrong:
            desc = describe_state_dict(state_dict)
            why = []
            if desc["num_blocks"] is not None and desc["num_blocks"] > 1:
                why.append(f"the checkpoint has {desc['num_blocks']} bottlenecks per residual unit (bearpaw num_blocks = {desc['num_blocks']}); "
                           "this engine implements num_blocks = 1 only")
            stacks = {int(k.split(".")[1]) for k in consumed if k.startswith("hg.")}
            if desc["num_stacks"] is not None and stacks and desc["num_stacks"] != max(stacks) + 1:
                why.append(f"the checkpoint has {desc['num_stacks']} stacks, the engine was built for {max(stacks) + 1} "
                           "(config['num_stacks'], reference df3d/config.py:33)")
            msg = "state_dict does not match the engine's network" + (": " + "; ".join(why) if why else "") + "."
            if wrong:
                msg += " Shapes: " + "; ".join(wrong[:8]) + ("; ..." if len(wrong) > 8 else "") + "."
            if missing:
                msg += f" Missing keys: {_listing(missing)}."
            if left:
                msg += f" Unconsumed keys: {_listing(left)}."
            msg += (f" (checkpoint describes {desc}).  If the reference is known to load this file non-strictly -- e.g. the first "
                    "config['num_stacks'] stacks of a network trained with more -- pass strict=False / set DF3D_CHECKPOINT_STRICT=0: missing keys still raise.")
            raise CheckpointMismatch(msg)
    return blob


class HourglassEngine:
    """The device engine: `forward(images_nhwc) -> heat-maps (n, 19, H/4, W/4)` on the current torch stream."""

    def __init__(self, state_dict, dtype="f32", num_stacks=2, device=None, height=256, width=512, row_bytes=0, fuse=True, fuse_upadd=None, ring=None, l1=None,
                 chain_views=None, split1=None, w2d=None, ring2=None, strict=None, no_reuse=False, wino=None):
        _native.require_gpu()
        self.lib = _native.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = dtype
        code = {"f32": _native.DF3D_DTYPE_F32, "bf16": _native.DF3D_DTYPE_BF16, "f16": _native.DF3D_DTYPE_F16, "f32s": _native.DF3D_DTYPE_F32S}[dtype]
        fp32_storage = dtype in ("f32", "f32s")   # f32s: the f32 engine's plan, buffers and kernels with split products (include/df3d_hip.h)
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
        if split1 is not None and fp32_storage:  # fp32: conv1 of the identity-skip bottlenecks as a launch of its own (csrc/hg_c1_f32.h), bit-identical
            _native.check(self.lib.df3d_hg_set_option(self.h, b"split1", int(split1)), "df3d_hg_set_option")   # (0, 1, or 8 + mask: development)
        if wino is None and os.environ.get("DF3D_WINO"):
            wino = int(os.environ["DF3D_WINO"])
        if wino is not None and dtype == "f32":  # exact fp32: the identity blocks' 3x3 as Winograd F(2x2, 3x3) (csrc/hg_bt_wino_f32.h); fp32 tolerance, not bit-identical to wino=0
            _native.check(self.lib.df3d_hg_set_option(self.h, b"wino", 1 if wino else 0), "df3d_hg_set_option")
        if w2d is None and os.environ.get("DF3D_W2D"):
            w2d = int(os.environ["DF3D_W2D"])
        if w2d is not None and not fp32_storage:  # 16-bit: the 3x3's weights of the ring bottlenecks as direct per-wave fragment loads (csrc/hg_bt_ring.h), bit-identical
            _native.check(self.lib.df3d_hg_set_option(self.h, b"w2d", 1 if w2d else 0), "df3d_hg_set_option")
        if ring2 is None and os.environ.get("DF3D_RING2"):
            ring2 = int(os.environ["DF3D_RING2"])
        if ring2 is not None and not fp32_storage:  # 16-bit: 1 (default) = round 4's ring bottleneck (csrc/hg_bt_ring.h MODE 2), 0 = round 3's; bit-identical
            _native.check(self.lib.df3d_hg_set_option(self.h, b"ring2", 1 if ring2 else 0), "df3d_hg_set_option")
        if chain_views is None and os.environ.get("DF3D_CHAIN_VIEWS"):
            chain_views = int(os.environ["DF3D_CHAIN_VIEWS"])
        if chain_views is not None:  # chains of full-resolution steps in chunks of this many views (0 = whole batch per launch)
            _native.check(self.lib.df3d_hg_set_option(self.h, b"chain_views", int(chain_views)), "df3d_hg_set_option")
        if no_reuse:  # tests: the alias-free workspace plan (every tensor keeps memory of its own)
            _native.check(self.lib.df3d_hg_set_option(self.h, b"no_reuse", 1), "df3d_hg_set_option")
        if strict is None:   # DF3D_CHECKPOINT_STRICT=0: the non-strict load (e.g. the first N stacks of a checkpoint trained with more)
            strict = os.environ.get("DF3D_CHECKPOINT_STRICT", "1") not in ("0", "false", "no")
        blob = pack_state_dict(self.h, state_dict, strict=strict)
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
        # overflow guard (f16 / f32s need every activation inside the IEEE-half range): planes with an infinity or a NaN, counted on the device by
        # the arg-max kernel of every batch (ops.heatmap_argmax(hm, nonfinite=engine.nonfinite_planes)); check_finite() reads it
        self.nonfinite_planes = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.df3d_hg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def check_finite(self, what="this recording"):
        """Raise if any heat-map plane the arg-max kernel has seen since the last check held an infinity or a NaN (one 4-byte read-back: call it
        once per recording, not per batch).  The reduced-precision engines overflow silently otherwise: inf / NaN heat-maps become a wrong
        points2d / heatmap_confidence in the pickle (the bar: reference tests/test_df3d.py:167-178)."""
        bad = int(self.nonfinite_planes.item())
        if bad:
            self.nonfinite_planes.zero_()
            hint = ("its activations left the IEEE-half range (|x| < 65 504) on these weights / images: rerun with dtype='f32' (df3d-cli --dtype f32)"
                    if self.dtype in ("f16", "f32s") else "rerun with dtype='f32' (df3d-cli --dtype f32)" if self.dtype == "bf16" else "the weights or the images hold non-finite values")
            raise _native.NativeLibraryError(f"{bad} heat-map plane(s) of {what} hold infinities or NaNs ({self.dtype} hourglass engine): {hint}; no result was written")

    # Reduced-precision engines against the exact one on a SAMPLE of the run's own input -- the guard that catches what the non-finite counter
    # cannot: on gfx950 the half conversions SATURATE (a value beyond 65 504 becomes 65 504, not an infinity) and a ReLU scrubs NaNs to zero, so an
    # f16 / f32s engine whose activations leave the half range returns FINITE, wrong heat-maps (measured: scripts/probe_overflow.py -- layer1
    # 3.3e4 where the exact engine has 1.5e8).  Allowed heat-map difference, as a fraction of the exact heat-maps' range (tests hold the engines
    # to tighter bars on synthetic weights: FP32_TOL / F16_TOL / BF16_TOL of tests/test_gpu_hourglass.py):
    CANARY_TOL = {"f32s": 1e-4, "f16": 2e-2, "bf16": 6e-2}

    def canary(self, exact, forward, what="the first batch"):
        """`forward(engine)` -> heat-maps of the SAME few views through `engine`; compares this engine with `exact` (an f32 HourglassEngine of the
        same weights).  Raises NativeLibraryError when they differ by more than CANARY_TOL[dtype] of the range or in a non-finite value."""
        if self.dtype == "f32":
            return 0.0
        got = forward(self).clone()
        ref = forward(exact)
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max()) / max(scale, 1e-30)
        if not (err <= self.CANARY_TOL[self.dtype]):   # (also true for a NaN)
            raise _native.NativeLibraryError(
                f"the {self.dtype} hourglass engine's heat-maps differ from the exact fp32 engine's by {err:.3g} of their range on {what} (allowed "
                f"{self.CANARY_TOL[self.dtype]:g}): its activations leave the range / precision of the format on these weights and images (half "
                f"conversions saturate at 65 504); rerun with dtype='f32' (df3d-cli --dtype f32); no result was written")
        return err

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
