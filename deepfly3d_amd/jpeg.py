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


def decode_luma(blobs, width, height, device=None, check=True, return_status=False, sequential=False, return_path=False):
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
            lib.df3d_jpeg_decode_luma(files_dev.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), n, total, int(sizes.max()), width, height, out.data_ptr(),
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


# pinned staging buffers outlive a reader: page-locking ~80 MB takes longer than decoding a whole batch, and df3d-cli
# (-r / -f) runs one reader per folder.  The pool is bounded by bytes: when a returned buffer does not fit, the smallest
# ones (the least likely to satisfy a later folder's size guess) are dropped first.
_PINNED_POOL = []
PINNED_POOL_MAX_BYTES = 384 << 20


def _pool_put(buf):
    _PINNED_POOL.append(buf)
    _PINNED_POOL.sort(key=lambda b: b.numel())
    while _PINNED_POOL and sum(b.numel() for b in _PINNED_POOL) > PINNED_POOL_MAX_BYTES:
        _PINNED_POOL.pop(0)


def clear_pinned_pool():
    """Release the page-locked staging buffers kept for the next reader (long-lived library users)."""
    del _PINNED_POOL[:]


class JpegFolderReader:
    """Streams JPEG files into the device decoder: the library's native reader threads (df3d_read_files, one call per
    batch, issued from a pool thread) put the files of batch k+1 straight into pinned staging memory while the GPU
    decodes batch k.  Every batch's decode statuses follow it to the host as one asynchronous copy of n int32 on the
    decode's stream; they are looked at when the NEXT batches are submitted (by then the copy has long finished: no
    stall), so a corrupt or unsupported file raises `JpegDecodeError` naming the file within two batches -- not after
    the whole folder has been inferred.  `finish()` checks what is still pending.

        reader = JpegFolderReader(width, height, device)
        for luma in reader.stream(list_of_path_batches):        # uint8 [n, H, W] on the device, decoded on a second stream
            ...                                                 # one batch ahead of the caller's work
        reader.finish()
    (`prefetch` / `decode_next` are the same steps one at a time, on the caller's stream.)
    """

    SLOTS = 3

    def __init__(self, width, height, device=None, workers=None, pinned=True, batch_capacity=None):
        import os
        from concurrent.futures import ThreadPoolExecutor

        _native.require_gpu()
        self.lib = _native.load()
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.width, self.height = int(width), int(height)
        self.pinned = bool(pinned)
        self.pool = ThreadPoolExecutor(max_workers=workers or self.SLOTS)  # one task per batch being read; the reading itself is native
        self.slots = [dict(buf=None, event=None) for _ in range(self.SLOTS)]
        self.turn = 0
        self.queue = []     # batches being read, oldest first (at most SLOTS - 1: one staging buffer may still feed a copy)
        self.work = None
        self.batch_capacity = batch_capacity  # largest batch `stream()` will see (None: the first batch is the largest)
        self.side = None    # second HIP stream of `stream()`: H2D copies + decode kernels under the caller's compute
        self._keep = None
        self.pending = []   # (event, pinned host statuses, n, paths) of decoded batches whose statuses have not been looked at
        self._status_host = []  # pinned int32 buffers, recycled

    READ_THREADS = 8  # native reader threads per batch (two batches may be in flight)

    def _pinned(self, nbytes):
        return torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=self.pinned)

    def _read_batch(self, slot, paths):
        """The files of one batch -> the slot's staging buffer through df3d_read_files (native threads open, size and read
        them; run from a pool thread, the interpreter lock is free meanwhile).  Returns (rc, starts, sizes, total bytes);
        rc = DF3D_ENOSPC: nothing was read, the buffer must hold total + 16 bytes (the caller, on the main thread, makes
        one: pinned allocations from a fresh thread cost hundreds of milliseconds)."""
        import ctypes
        import os

        n = len(paths)
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        starts, sizes = np.empty(n, np.uint32), np.empty(n, np.uint32)
        total = ctypes.c_size_t(0)
        buf = slot["buf"]
        rc = self.lib.df3d_read_files(arr, n, buf.data_ptr() if buf is not None else None, buf.numel() if buf is not None else 0, starts.ctypes.data,
                                      sizes.ctypes.data, ctypes.byref(total), self.READ_THREADS)
        if rc == _native.DF3D_EIO:
            msg = self.lib.df3d_last_error().decode()
            raise (FileNotFoundError if "No such file" in msg else IOError)(msg)
        if rc != _native.DF3D_ENOSPC:
            _native.check(rc, "df3d_read_files")
        return rc, starts, sizes, int(total.value)

    def prefetch(self, paths, sizes=None):
        """Start reading `paths` into the next staging slot (`sizes` is accepted for callers of round 1 and ignored: the native
        reader sizes the files as it opens them)."""
        import os

        if len(self.queue) >= self.SLOTS - 1:
            raise RuntimeError("too many batches are being read")
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % self.SLOTS
        if slot["event"] is not None:
            slot["event"].synchronize()  # the copy that last used this staging buffer has finished
        paths = list(paths)
        if paths:
            # room for the batch if its files are like the first one (camera frames are); a wrong guess costs one re-read
            guess = int(max(len(paths), self.batch_capacity or 0) * (os.stat(paths[0]).st_size + 16) * 1.25) + 4096
            if slot["buf"] is None and self.pinned:
                fit = [b for b in _PINNED_POOL if b.numel() >= guess]
                if fit:
                    slot["buf"] = min(fit, key=lambda b: b.numel())
                    _PINNED_POOL[:] = [b for b in _PINNED_POOL if b is not slot["buf"]]
            if slot["buf"] is None or slot["buf"].numel() < guess:
                slot["buf"] = self._pinned(guess)
        self.queue.append((slot, self.pool.submit(self._read_batch, slot, paths) if paths else None, paths))

    def _launch_decode(self, out, stream):
        """Wait for the batch being read and enqueue its H2D copy + decode on `stream` into `out[:n]`."""
        self.check_statuses()   # of the batches decoded before: fail within two batches of a bad file
        slot, fut, paths = self.queue.pop(0)
        n = len(paths)
        if n:
            rc, starts, sizes, total = fut.result()
            if rc == _native.DF3D_ENOSPC:  # the size guess was too small
                slot["buf"] = self._pinned((total + 16) * 1.25)
                rc, starts, sizes, total = self._read_batch(slot, paths)
                _native.check(rc, "df3d_read_files")
            if total >= 2**32 - 64:
                raise ValueError("more than 4 GiB of JPEG data in one batch")
        if n:
            with torch.cuda.device(self.dev), torch.cuda.stream(stream):
                files_dev = slot["buf"][: total + 16].to(self.dev, non_blocking=True)
                tab = torch.from_numpy(np.stack([starts, sizes]).view(np.int32)).to(self.dev, non_blocking=True)
                slot["event"] = torch.cuda.Event()
                slot["event"].record(stream)
                status = torch.empty((n,), dtype=torch.int32, device=self.dev)
                need = self.lib.df3d_jpeg_work_bytes(n, self.width, self.height, total)
                if self.work is None or self.work.numel() < need:
                    self.work = torch.empty((need,), dtype=torch.uint8, device=self.dev)
                _native.check(
                    self.lib.df3d_jpeg_decode_luma(files_dev.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), n, total, int(sizes.max()), self.width,
                                                   self.height, out.data_ptr(), status.data_ptr(), None, self.work.data_ptr(), self.work.numel(), 0,
                                                   stream.cuda_stream),
                    "df3d_jpeg_decode_luma",
                )
                self._keep = (files_dev, tab, status)  # alive until the next launch on this stream has been enqueued behind them
                host = self._status_host.pop() if self._status_host else None
                if host is None or host.numel() < n:
                    host = torch.empty(max(n, self.batch_capacity or 0), dtype=torch.int32, pin_memory=self.pinned)
                host[:n].copy_(status, non_blocking=True)
                done = torch.cuda.Event()
                done.record(stream)
            self.pending.append((done, host, n, paths))
        return n

    def check_statuses(self, wait=False):
        """Look at the decode statuses that have reached the host: every batch but the newest one is waited for (its decode
        was enqueued at least one batch ago), the newest only if `wait`.  Raises JpegDecodeError naming the first bad file."""
        while self.pending:
            done, host, n, paths = self.pending[0]
            if not (wait or len(self.pending) > 1 or done.query()):
                break
            done.synchronize()
            self.pending.pop(0)
            st = host[:n].numpy()
            bad = np.flatnonzero(st)
            if bad.size:
                b = int(bad[0])
                raise JpegDecodeError(f"{paths[b]}: {STATUS.get(int(st[b]), st[b])}")
            self._status_host.append(host)

    def decode_next(self, next_paths=None):
        """Wait for the batch being read, launch its decode, start reading `next_paths`; returns luma [n, H, W]."""
        n = len(self.queue[0][2])
        out = torch.empty((n, self.height, self.width), dtype=torch.uint8, device=self.dev)
        self._launch_decode(out, torch.cuda.current_stream(self.dev))
        if next_paths is not None:
            self.prefetch(next_paths)
        return out

    def stream(self, batches, sizes=None):
        """Generator over the decoded batches of `batches` (a list of path lists), pipelined three deep: while the caller's
        stream works on batch k (everything it enqueues between two `next()` calls), batch k + 1 is copied and decoded on a
        second HIP stream and the files of batch k + 2 are read by the thread pool.  Each yielded tensor (uint8 [n, H, W],
        one of two alternating device buffers) is valid on the caller's current stream until the next item is requested.  `sizes[k]`: byte sizes of batch k's files."""
        if not len(batches):
            return
        main = torch.cuda.current_stream(self.dev)
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.dev)
        side = self.side
        first = batches[0]
        cap = max(len(first), self.batch_capacity or 0)
        luma = [torch.empty((cap, self.height, self.width), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        for buf in luma:
            buf.record_stream(side)   # written on the side stream: if the consumer abandons the generator the allocator must not
                                      # hand the memory out again before that stream is done with it
        side.wait_stream(main)  # the buffers exist for the side stream
        decoded, consumed = [None, None], [None, None]
        size_of = (lambda k: None) if sizes is None else (lambda k: sizes[k])
        self.prefetch(first, size_of(0))
        if len(batches) > 1:
            self.prefetch(batches[1], size_of(1))   # both reads are under way before the first one is waited for
        n_cur = self._launch_decode(luma[0], side)
        decoded[0] = torch.cuda.Event()
        decoded[0].record(side)
        for k in range(len(batches)):
            main.wait_event(decoded[k & 1])
            yield luma[k & 1][:n_cur]
            consumed[k & 1] = torch.cuda.Event()
            consumed[k & 1].record(main)  # everything the caller enqueued on batch k
            if k + 1 < len(batches):
                nb = (k + 1) & 1
                if k + 2 < len(batches):
                    self.prefetch(batches[k + 2], size_of(k + 2))   # before waiting for batch k + 1's reads
                if consumed[nb] is not None:
                    side.wait_event(consumed[nb])  # batch k - 1 no longer reads this buffer
                n_cur = self._launch_decode(luma[nb], side)
                decoded[nb] = torch.cuda.Event()
                decoded[nb].record(side)
        main.wait_stream(side)

    def finish(self, check=True):
        """Check the decode statuses still pending (one host synchronisation) and release the reader threads.  `check=False`
        on an error path: release only (the exception under way is the one to report)."""
        try:
            if check:
                self.check_statuses(wait=True)
        finally:
            self.pending = []
            self.pool.shutdown(wait=False)
            for item in self.queue:  # an error path: file reads may still be writing into a staging buffer
                try:
                    if item[1] is not None:
                        item[1].result()
                except Exception:
                    pass
            self.queue = []
            if self.pinned:
                for slot in self.slots:
                    if slot["event"] is not None:
                        slot["event"].synchronize()  # the last copy out of this buffer has finished
                    if slot["buf"] is not None:
                        _pool_put(slot["buf"])
                    slot["buf"] = None
