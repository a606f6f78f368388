"""Device JPEG decode timing on one network batch worth of files (GPU box).

  python scripts/probe_jpeg.py [views=896] [reps=10]

Repeats the 14 committed reference images up to `views` files, checks the parallel Huffman paths against the
sequential kernel (bit-exact) and prints the time of one df3d_jpeg_decode_luma call (files already on the device):
plus the histogram of synchronisation passes.
"""
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfly3d_amd import _native, jpeg  # noqa: E402


def main():
    views = int(sys.argv[1]) if len(sys.argv) > 1 else 896
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(here, "tests/golden/images/*.jpg")))]
    blobs = [base[i % len(base)] for i in range(views)]
    lib = _native.load()
    dev = torch.device("cuda:0")
    buf, starts, sizes, total = jpeg.pack_files(blobs)
    files_dev = buf.to(dev)
    tab = torch.from_numpy(np.stack([starts, sizes]).view(np.int32)).to(dev)
    n, W, H = len(blobs), 960, 480
    out = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    path = torch.zeros((n,), dtype=torch.int32, device=dev)
    need = lib.df3d_jpeg_work_bytes(n, W, H, total)
    work = torch.empty((need,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(max_file, flags):
        _native.check(lib.df3d_jpeg_decode_luma(files_dev.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), n, total, max_file, W, H, out.data_ptr(), status.data_ptr(),
                                                path.data_ptr(), work.data_ptr(), need, flags, stream), "df3d_jpeg_decode_luma")

    run(0, 1)
    torch.cuda.synchronize()
    ref = out.clone()
    assert int(status.abs().sum()) == 0
    for name, mf in (("parallel Huffman", int(sizes.max())),):
        out.zero_()
        run(mf, 0)
        torch.cuda.synchronize()
        same = bool((out == ref).all())
        p = path.cpu().numpy()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(mf, 0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name:26s} {views} files {ms:7.3f} ms/call  {views / ms * 1e3:9.0f} views/s  identical to sequential: {same}  passes {np.bincount(np.maximum(p, 0)).tolist()}", flush=True)
        assert same


if __name__ == "__main__":
    main()
