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


def decode_luma(blobs, width, height, device=None, check=True, return_status=False, sequential=False, return_path=False, stream_in_lds=True):
    """[bytes, ...] of width x height baseline JPEGs -> uint8 CUDA tensor [n, height, width] (luma planes)."""
    _native.require_gpu()
    lib = _native.load()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    n = len(blobs)
    out = torch.empty((n, height, width), dtype=torch.uint8, device=dev)
    if n == 0:
        extra = ((np.zeros(0, np.int32),) if return_status else ()) + ((np.zeros(0, np.int32),) if return_path else ())
        return (out, *extra) if extra else out
    buf, starts, sizes, total = pack_files(blobs)
    files_dev = buf.to(dev, non_blocking=True)
    tab = torch.from_numpy(np.stack([starts, sizes]).view(np.int32)).to(dev, non_blocking=True)  # [2, n] uint32 bits
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    path = torch.zeros((n,), dtype=torch.int32, device=dev) if return_path else None
    need = lib.df3d_jpeg_work_bytes(n, width, height, total)
    work = torch.empty((need,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):  # kernels launch on the current HIP device
        stream = torch.cuda.current_stream(dev).cuda_stream
        _native.check(
            lib.df3d_jpeg_decode_luma(files_dev.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), n, total, int(sizes.max()) if stream_in_lds else 0, width, height, out.data_ptr(),
                                      status.data_ptr(), path.data_ptr() if return_path else None, work.data_ptr(), need, 1 if sequential else 0, stream),
            "df3d_jpeg_decode_luma",
        )
    if check or return_status:
        st = status.cpu().numpy()
        if check and st.any():
            bad = int(np.flatnonzero(st)[0])
            raise JpegDecodeError(f"file {bad} of the batch: {STATUS.get(int(st[bad]), st[bad])}")
        if return_status:
            return (out, st, path.cpu().numpy()) if return_path else (out, st)
    return (out, path.cpu().numpy()) if return_path else out


class JpegFolderReader:
    """Streams JPEG files into the device decoder: a thread pool reads the files of batch k+1 straight into pinned
    staging memory (os.readinto releases the GIL) while the GPU decodes batch k.  Statuses are collected on the
    device and checked once, in `finish()`, so that no batch forces a host synchronisation.

        reader = JpegFolderReader(width, height, device)
        reader.prefetch(paths_0)
        for k in ...:
            luma = reader.decode_next(paths_k_plus_1_or_None)   # uint8 [n, H, W] on the device
        reader.finish()
    """

    SLOTS = 3

    def __init__(self, width, height, device=None, workers=None, pinned=True):
        import os
        from concurrent.futures import ThreadPoolExecutor

        _native.require_gpu()
        self.lib = _native.load()
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.width, self.height = int(width), int(height)
        self.pinned = bool(pinned)
        self.pool = ThreadPoolExecutor(max_workers=workers or max(2, min(16, (os.cpu_count() or 4) - 1)))
        self.slots = [dict(buf=None, event=None) for _ in range(self.SLOTS)]
        self.turn = 0
        self.pending = None
        self.work = None
        self.statuses = []  # (status tensor, paths)

    @staticmethod
    def _read_into(path, view):
        with open(path, "rb", buffering=0) as f:
            got = f.readinto(view)
        if got != len(view):
            raise IOError(f"{path}: short read ({got} of {len(view)} bytes)")

    def prefetch(self, paths):
        """Start reading `paths` into the next staging slot."""
        import os

        if self.pending is not None:
            raise RuntimeError("a batch is already being read")
        sizes = np.array([os.path.getsize(p) for p in paths], dtype=np.int64)
        padded = (sizes + 15) // 16 * 16
        ends = np.cumsum(padded)
        starts = ends - padded
        total = int(ends[-1]) if len(paths) else 0
        if total >= 2**32 - 64:
            raise ValueError("more than 4 GiB of JPEG data in one batch")
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % self.SLOTS
        if slot["event"] is not None:
            slot["event"].synchronize()  # the copy that last used this staging buffer has finished
        if slot["buf"] is None or slot["buf"].numel() < total + 16:
            slot["buf"] = torch.empty(int((total + 16) * 1.25), dtype=torch.uint8)
            if self.pinned:
                slot["buf"] = slot["buf"].pin_memory()
        view = memoryview(slot["buf"].numpy())
        futs = [self.pool.submit(self._read_into, p, view[s : s + n]) for p, s, n in zip(paths, starts, sizes)]
        self.pending = (slot, futs, starts.astype(np.uint32), sizes.astype(np.uint32), total, list(paths))

    def decode_next(self, next_paths=None):
        """Wait for the batch being read, launch its decode, start reading `next_paths`; returns luma [n, H, W]."""
        slot, futs, starts, sizes, total, paths = self.pending
        self.pending = None
        for f in futs:
            f.result()
        n = len(paths)
        out = torch.empty((n, self.height, self.width), dtype=torch.uint8, device=self.dev)
        if n:
            files_dev = slot["buf"][: total + 16].to(self.dev, non_blocking=True)
            tab = torch.from_numpy(np.stack([starts, sizes]).view(np.int32)).to(self.dev, non_blocking=True)
            slot["event"] = torch.cuda.Event()
            slot["event"].record(torch.cuda.current_stream(self.dev))
            status = torch.empty((n,), dtype=torch.int32, device=self.dev)
            need = self.lib.df3d_jpeg_work_bytes(n, self.width, self.height, total)
            if self.work is None or self.work.numel() < need:
                self.work = torch.empty((need,), dtype=torch.uint8, device=self.dev)
            with torch.cuda.device(self.dev):
                _native.check(
                    self.lib.df3d_jpeg_decode_luma(files_dev.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), n, total, int(sizes.max()), self.width,
                                                   self.height, out.data_ptr(), status.data_ptr(), None, self.work.data_ptr(), self.work.numel(), 0,
                                                   torch.cuda.current_stream(self.dev).cuda_stream),
                    "df3d_jpeg_decode_luma",
                )
            self.statuses.append((status, paths))
        if next_paths is not None:
            self.prefetch(next_paths)
        return out

    def finish(self):
        """Check every decode status (one host synchronisation) and release the reader threads."""
        try:
            for status, paths in self.statuses:
                st = status.cpu().numpy()
                if st.any():
                    bad = int(np.flatnonzero(st)[0])
                    raise JpegDecodeError(f"{paths[bad]}: {STATUS.get(int(st[bad]), st[bad])}")
        finally:
            self.statuses = []
            self.pool.shutdown(wait=False)
