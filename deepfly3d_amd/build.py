"""Build libdf3d_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU).

    python -m deepfly3d_amd.build [--force] [--verbose]

The shared library lands next to this file so it travels with the source snapshot to the GPU box.
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
LIB_PATH = os.path.join(HERE, "libdf3d_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-ffp-contract=fast"]
# pose3d.hip and ba_lsmr.hip reproduce float64 scalar recurrences (One-Euro filter, LSMR rotations) exactly as the
# CPU reference arithmetic rounds them: no multiply-add fusion there
# render.hip (f4, the video frames): float64 pixel tests restated in numpy by oracle/render.py, compared bit for bit
FILE_FLAGS = {"pose3d.hip": ["-ffp-contract=off"], "ba_lsmr.hip": ["-ffp-contract=off"], "render.hip": ["-ffp-contract=off"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build libdf3d_hip.so)")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "df3d_hip.h"))
    return hdrs


def build(force=False, verbose=False):
    srcs = _sources()
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "stamp")
    hdr_digest = _digest(_deps())
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        extra = FILE_FLAGS.get(os.path.basename(src), [])
        dig = _digest([src]) + hdr_digest + " ".join(extra)
        dig_file = obj + ".sha"
        if not force and os.path.exists(obj) and os.path.exists(dig_file) and open(dig_file).read() == dig:
            return obj, False
        cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(dig_file, "w") as f:
            f.write(dig)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or force or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write("ok")
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
