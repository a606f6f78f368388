"""ORACLE (test infrastructure only): ctypes face of oracle/jpeg_oracle.c, the CPU restatement of libjpeg's
baseline luma decode ("islow" IDCT).  See the C file's header for what it restates and how it is pinned."""
import ctypes
import os

import numpy as np

from . import build_c

STATUS = {0: "ok", 1: "truncated", 2: "not a JPEG", 3: "unsupported", 4: "corrupt", 5: "shape mismatch"}
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = [p for p in build_c.build() if p.endswith("libjpeg_oracle.so")][0]
        _lib = ctypes.CDLL(path)
        _lib.jpeg_oracle_info.restype = ctypes.c_int
        _lib.jpeg_oracle_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t] + [ctypes.POINTER(ctypes.c_int)] * 3
        _lib.jpeg_oracle_decode_luma.restype = ctypes.c_int
        _lib.jpeg_oracle_decode_luma.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p]
    return _lib


def info(data):
    w, h, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib().jpeg_oracle_info(data, len(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(n))
    if rc:
        raise ValueError(f"jpeg oracle: {STATUS.get(rc, rc)}")
    return w.value, h.value, n.value


def decode_luma(data, with_coefficients=False):
    """bytes -> luma plane uint8 [H, W] (and optionally (coef int16 [bh, bw, 64], n_symbols))."""
    w, h, _ = info(data)
    out = np.zeros((h, w), np.uint8)
    stats = (ctypes.c_longlong * 3)()
    coef = np.zeros(((h + 15) // 8 + 2, (w + 15) // 8 + 2, 64), np.int16) if with_coefficients else None
    # the coefficient grid is MCU-padded; allocate generously, trim with the stats
    if coef is not None:
        flat = np.zeros(coef.size, np.int16)
        rc = lib().jpeg_oracle_decode_luma(data, len(data), w, h, out.ctypes.data, flat.ctypes.data, stats)
    else:
        rc = lib().jpeg_oracle_decode_luma(data, len(data), w, h, out.ctypes.data, None, stats)
    if rc:
        raise ValueError(f"jpeg oracle: {STATUS.get(rc, rc)}")
    if coef is not None:
        bw, bh = stats[1], stats[2]
        return out, flat[: bw * bh * 64].reshape(bh, bw, 64).copy(), stats[0]
    return out


def status(data, expect_w=0, expect_h=0):
    """Status code the decoder returns for `data` (0 = ok), without raising."""
    w, h, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib().jpeg_oracle_info(data, len(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(n))
    if rc:
        return rc
    out = np.zeros((h.value, w.value), np.uint8)
    return lib().jpeg_oracle_decode_luma(data, len(data), expect_w, expect_h, out.ctypes.data, None, None)
