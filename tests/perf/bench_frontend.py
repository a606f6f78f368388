"""Input front-end measurement: JPEG files -> frames/s, PCIe and file IO included.

  python tests/perf/bench_frontend.py [--frames 256] [--out profiles/r01_frontend.json]

Builds a folder of `frames` x 7 camera JPEGs by repeating the reference's 14 sample images (tests/golden/images),
then reports
  * decode only: device (read files + H2D + df3d_jpeg_decode_luma) vs libjpeg-turbo through Pillow on the host
    cores (thread pool, the reference's DataLoader-worker job) vs the single-thread C oracle
  * inference_folder end to end (files -> 2-D points), fp32 and bf16, synthetic weights
"""
import argparse
import glob
import io
import json
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepfly3d_amd import inference, jpeg  # noqa: E402


def host_cores():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except Exception:
        pass
    return os.cpu_count() or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    root = tempfile.mkdtemp(prefix="df3d_frontend_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    folder = os.path.join(root, "images")
    os.makedirs(folder)
    try:
        for c in range(7):
            for t in range(a.frames):
                shutil.copy(os.path.join(here, "tests/golden/images", f"camera_{c}_img_{t % 2}.jpg"), os.path.join(folder, f"camera_{c}_img_{t}.jpg"))
        paths = [os.path.join(folder, f"camera_{c}_img_{t}.jpg") for c in range(7) for t in range(a.frames)]
        nbytes = sum(os.path.getsize(p) for p in paths)
        res = {"frames": a.frames, "views": len(paths), "jpeg_megabytes": round(nbytes / 1e6, 1), "image": "960x480 baseline 4:2:0, ~70 KB",
               "host_cores": host_cores(), "device": torch.cuda.get_device_name(0)}

        # ---- decode only -----------------------------------------------------------------------------------
        chunks = [paths[i : i + 224] for i in range(0, len(paths), 224)]
        for rep in range(2):
            rd = jpeg.JpegFolderReader(960, 480)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rd.prefetch(chunks[0])
            for k in range(len(chunks)):
                luma = rd.decode_next(chunks[k + 1] if k + 1 < len(chunks) else None)
            rd.finish()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res["decode_device_views_per_s"] = round(len(paths) / dt)

        from PIL import Image

        def pil(p):
            with Image.open(p) as im:
                im.draft("L", im.size)
                return np.asarray(im)

        cores = host_cores()
        sample = paths[: min(len(paths), 64 * cores)]
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(pil, sample[:cores]))
            t0 = time.perf_counter()
            ref = list(ex.map(pil, sample))
            dt = time.perf_counter() - t0
        res["decode_libjpeg_turbo_views_per_s"] = round(len(sample) / dt)
        res["decode_libjpeg_turbo_threads"] = cores
        assert np.array_equal(luma[-1].cpu().numpy(), pil(paths[-1]))

        from oracle import jpeg as oj

        blobs = [open(p, "rb").read() for p in paths[:32]]
        oj.decode_luma(blobs[0])
        t0 = time.perf_counter()
        for b in blobs:
            oj.decode_luma(b)
        res["decode_c_oracle_1_thread_views_per_s"] = round(len(blobs) / (time.perf_counter() - t0))

        # ---- end to end ---------------------------------------------------------------------------------------
        os.environ["DF3D_SYNTHETIC_WEIGHTS"] = "0"
        for dt_name in ("f32", "f32s", "f16", "bf16"):
            inference.inference_folder(folder=folder, camera_ids_to_flip=[4, 5, 6], max_img_id=min(a.frames, 256) - 1, dtype=dt_name)  # warm-up: engine, full-size batch buffers
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pts, conf = inference.inference_folder(folder=folder, camera_ids_to_flip=[4, 5, 6], max_img_id=a.frames - 1, dtype=dt_name)
            dt = time.perf_counter() - t0
            res[f"inference_folder_{dt_name}_frames_per_s"] = round(a.frames / dt, 1)
            # the same work with the luma planes already resident on the device (no files, no PCIe, no JPEG decode): the rate the
            # file path is compared with
            eng = inference.get_engine(dtype=dt_name)
            nb = 896
            luma = torch.randint(0, 256, (nb, 480, 960), dtype=torch.uint8, device="cuda")
            flip = torch.zeros((nb,), dtype=torch.uint8, device="cuda")
            reps = max(1, (a.frames * 7) // nb)
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    inference.inference_frames(luma, flip, eng)
                torch.cuda.synchronize()
                dtr = time.perf_counter() - t0
            res[f"resident_{dt_name}_frames_per_s"] = round(reps * nb / 7 / dtr, 1)
            res[f"file_path_fraction_of_resident_{dt_name}"] = round(res[f"inference_folder_{dt_name}_frames_per_s"] / res[f"resident_{dt_name}_frames_per_s"], 3)
        print(json.dumps(res))
        if a.out:
            with open(a.out, "w") as f:
                json.dump(res, f, indent=1)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
