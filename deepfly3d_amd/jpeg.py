"""Device JPEG front-end: the host only reads file bytes; header parsing, Huffman decoding and
the inverse DCT run in libdf3d_hip.so (csrc/jpeg.hip, C ABI `df3d_jpeg_decode_luma`).

Replaces the libjpeg decode inside df2d's DataLoader workers (call site reference df3d/core.py:177-185).  The
result is the JPEG's luma plane with libjpeg's default "islow" IDCT, bit-identical to Pillow / libjpeg-turbo for
grayscale and chroma-neutral files (the rig's monochrome cameras) and to libjpeg's own grayscale output
(`Image.draft("L", ...)`) for coloured ones.
"""
import numpy as np
import torch

from . import _native

STATUS = {0: "ok", 1: "truncated file", 2: "not a JPEG", 3: "unsupported JPEG (progressive / arithmetic / 12 bit / multi-scan)",
          4: "corrupt JPEG", 5: "image size differs from the expected one"}


class JpegDecodeError(ValueError):
    pass


def pack_files(blobs, pinned=True):
    """[bytes, ...] -> (uint8 tensor holding the files back to back, each starting on a 16-byte boundary, with 16
    spare bytes at the end; uint32 start offsets [n]; uint32 sizes [n]; total bytes)."""
    sizes = np.array([len(b) for b in blobs], dtype=np.int64)
    ends = np.cumsum((sizes + 15) // 16 * 16)
    starts = ends - (sizes + 15) // 16 * 16
    total = int(ends[-1]) if len(blobs) else 0
    if total >= 2**32 - 64:
        raise ValueError("more than 4 GiB of JPEG data in one batch")
    buf = torch.zeros(total + 16, dtype=torch.uint8)
    if pinned and torch.cuda.is_available():
        buf = buf.pin_memory()
    view = buf.numpy()
    for b, s in zip(blobs, starts):
        view[s : s + len(b)] = np.frombuffer(b, dtype=np.uint8)
    return buf, starts.astype(np.uint32), sizes.astype(np.uint32), total


def decode_luma(blobs, width, height, device=None, check=True, return_status=False):
    """[bytes, ...] of width x height baseline JPEGs -> uint8 CUDA tensor [n, height, width] (luma planes)."""
    _native.require_gpu()
    lib = _native.load()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    n = len(blobs)
    out = torch.empty((n, height, width), dtype=torch.uint8, device=dev)
    if n == 0:
        return (out, np.zeros(0, np.int32)) if return_status else out
    buf, starts, sizes, total = pack_files(blobs)
    files_dev = buf.to(dev, non_blocking=True)
    tab = torch.from_numpy(np.stack([starts, sizes]).view(np.int32)).to(dev, non_blocking=True)  # [2, n] uint32 bits
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    need = lib.df3d_jpeg_work_bytes(n, width, height, total)
    work = torch.empty((need,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _native.check(
        lib.df3d_jpeg_decode_luma(files_dev.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), n, total, width, height, out.data_ptr(),
                                  status.data_ptr(), work.data_ptr(), need, stream),
        "df3d_jpeg_decode_luma",
    )
    if check or return_status:
        st = status.cpu().numpy()
        if check and st.any():
            bad = int(np.flatnonzero(st)[0])
            raise JpegDecodeError(f"file {bad} of the batch: {STATUS.get(int(st[bad]), st[bad])}")
        if return_status:
            return out, st
    return out
