#!/usr/bin/env python3
"""Benchmark of the DeepFly3D per-frame hot path on MI355X (BASELINE.json metric: frames/sec, 7-view 2D -> 3D).

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU `steps x frames_per_step` synthetic frames (default 8 x 128 = 1 024),
each 7 views of 256 x 512 x 3 float32, seeded and resident in HBM before the timed region; 2-stack hourglass in
fp32 with seeded synthetic weights (no checkpoints offline); fixed calib.pkl cameras.  One "step" = one batch of
`frames_per_step` frames through the whole path: hourglass -> arg-max/confidence -> 19->38 layout -> DLT.
N > 1: every rank owns its own frame range (weak scaling) and the per-frame results are gathered to rank 0 once
(RCCL, one packed `dist.gather`), inside the timed region.  Rank 0 prints ONE JSON line.

The line's headline is configs[1] (fp32).  At N = 1 the same process then times, under the same driver clock,
  config1_f32_split  configs[1] through the f32s engine: float32 tensors, weights and accumulation, the products as two-way IEEE-half splits on
                 the 16-bit matrix pipe (gfx950's exact-fp32 MFMA runs at 1/16 of the half-precision rate); NOT the headline -- its heat-maps differ
                 from the exact-fp32 engine's by the `max_rel_diff_vs_f32_engine` printed beside it (same 5e-5 test tolerance, tests/test_gpu_hourglass.py)
  config2_bf16   configs[2]: the same frames through the bf16 hourglass (all convolutions on MFMA, fp32 accumulate)
  config2_f16    the same kernels on IEEE half: the 16-bit engine whose heat-map confidences stay inside the reference's
                 own test tolerance (2e-3; bf16's are ~5e-3 off: tests/test_gpu_hourglass.py)
  config4_share  a short share of configs[4]: rank 0 of 8 of a 16 000-frame stream (2 000 frames, f16), a bundle adjustment
                 per 1 000-frame window beside the pipeline, the packed gather executed on a 1-rank RCCL group
each hourglass leg with its own roofline block.

    python bench.py --rank-share 8 --stream-frames 100000 [--ba-window 1000] [--force-collective]
runs ONE rank's full share of BASELINE configs[3] / configs[4] on the one GPU at hand (profiles/r03_rankshare_*.json).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --strong --stream-frames 100000 [--ba-window 1000] [--dtype f16]
is BASELINE's 8-GPU table in one shot: configs[3] (with --ba-window: configs[4]) STRONG-scaled -- the one stream sharded by frame over the N
ranks, one packed gather, Procrustes over the whole sequence on rank 0, all inside the timed region; `"scaling": "strong"`, per-rank frames
and rank 0's tail (gather / Procrustes ms) on the line.  N = 1 runs the whole stream on one GPU (the same pipeline as the default line).

Roofline fractions (per kernel and for the dominant one), spelled out because a fused kernel has more than one byte count:
  frac_mfma      FLOPs EXECUTED on the matrix pipe / time / dense MFMA peak of the dtype (the Winograd tail of the fp32 engine executes 16/36 of a
                 3x3's direct FLOPs: its direct-equivalent rate is printed beside the fraction, never priced against the peak)
  frac_hbm_min   (inputs read once + outputs written once, intermediates on chip) / time / 8 TB/s -- the least the launch can move
  frac_hbm_m1    bytes the fusion model M1 of SURVEY.md 8(d) charges (every convolution's input and output) / time / 8 TB/s
                 -- a convention: it exceeds what any kernel moves once convolutions are fused
  frac_hbm_pmc   HBM bytes rocprofv3's counters saw ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, profiles/traffic.json) / time / 8 TB/s
The whole step's HBM rate -- BASELINE's "hourglass HBM GB/s vs peak" -- is printed three ways in `config` (and in every hourglass leg):
hourglass_gbs_m1_end_to_end (the M1 convention), hourglass_gbs_pmc_end_to_end (bytes the counters saw: traffic.json x this run's launch counts,
when it covers the step) and hourglass_gbs_min_end_to_end (inputs once + outputs once per launch), each over this run's step time.
`bound` is the larger of frac_mfma and frac_hbm_pmc (frac_hbm_min when no counter pass covers the kernel); `frac` is that one.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md (dense).  f32s forms every float32 product from THREE half-precision products (two-way split of both
# operands, the lo x lo term dropped): its matrix-pipe roof in float32 FLOPs is the half-precision peak / 3
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "f32s": 2500.0 / 3}
PEAK_HBM_GBS = 8000.0
DTYPE_SHORT = {"f32": "fp32", "bf16": "bf16", "f16": "f16", "f32s": "fp32 with split-half products"}
DTYPE_WORDS = {"f32": "fp32", "bf16": "bf16 (all convolutions on MFMA, fp32 accumulate)", "f16": "IEEE-half f16 (all convolutions on MFMA, fp32 accumulate)",
               "f32s": "fp32 tensors and weights, every product as a two-way IEEE-half split of both operands on the 16-bit matrix pipe (fp32 accumulate)"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16", "f32s"], default="f32")
    ap.add_argument("--pool-frames", type=int, default=0, help="distinct frames resident in HBM (0 = steps*frames_per_step, capped at 1024)")
    ap.add_argument("--ba-window", type=int, default=0,
                    help="BASELINE configs[4]: run one bundle adjustment (HIP kernels + TRF/LSMR driver) per this many frames on "
                         "geometry-consistent synthetic detections, inside the timed region (0 = fixed calib.pkl, configs[1])")
    ap.add_argument("--rank-share", type=int, default=0,
                    help="run rank 0's share of a --stream-frames stream sharded over this many ranks (configs[3]/[4] on one GPU); overrides --steps")
    ap.add_argument("--stream-frames", type=int, default=100000)
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE configs[3] (with --ba-window: configs[4]) STRONG-scaled over the --gpus ranks: the ONE --stream-frames stream is sharded by "
                         "frame (rank r runs shard_range(stream, N, r, window)), one packed gather, Procrustes over the whole sequence on rank 0 -- all inside "
                         "the timed region; --steps is derived (the largest shard's batches)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: create a 1-rank process group (RCCL) and execute the packed gather anyway")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank control flow (no GPU, gloo): the pipeline is a stub that stamps every frame with its global index, "
                         "everything else -- shards, batches per rank, window bookkeeping, the packed gather -- is the real code; rank 0 checks the gathered sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-bf16-leg", "--no-legs", dest="no_legs", action="store_true",
                    help="skip the attached legs (configs[2] in bf16 and f16, the configs[4] share)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region rank 0 recomputes a strided sample of EVERY rank's frames on its own GPU (the peers' frame pools are "
                         "seeded by rank, so rank 0 can regenerate them) and requires the gathered records to be bit-identical; the line then carries "
                         "`verify`, `per_rank_frames_per_s`, `gather_ms`, `rccl_world`")
    ap.add_argument("--full", action="store_true",
                    help="print the whole record (per-kernel tables, legends, long workload texts: tens of KB) instead of the <= 4 KB line")
    ap.add_argument("--tables", default=None,
                    help="where the whole record goes beside the short line (default gpurun_out/bench_full_<dtype>.json; '-' = nowhere)")
    return ap.parse_args(argv)


def host_cores():
    """Usable host cores: CPU affinity, clipped by the cgroup CPU quota when the box is containerised."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def kernel_source_sha():
    """sha256 over WHAT THE GPU RUNS of the hourglass kernels -- the gfx950 code objects inside the library in use (its
    `.hip_fatbin` section: one clang offload bundle per .hip file; the bundles that hold `hgk::` kernels) -- not over the source
    text: a comment or a host-side edit leaves it unchanged, so `profiles/traffic.json` (which records the value it was
    collected with; there is no .git on the GPU box) only goes stale when a kernel really changed.  Falls back to the source
    text when the library cannot be parsed."""
    import struct

    try:
        from deepfly3d_amd import _native

        blob = open(_native.library_path(), "rb").read()
        shoff = struct.unpack_from("<Q", blob, 0x28)[0]
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
        secs = [struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize) for i in range(shnum)]
        stroff = secs[shstrndx][4]
        fat = None
        for name, _typ, _flags, _addr, off, size in secs:
            end = blob.index(b"\0", stroff + name)
            if blob[stroff + name:end] == b".hip_fatbin":
                fat = blob[off:off + size]
        if fat is None:
            raise ValueError("no .hip_fatbin section")
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = []
        pos = fat.find(magic)
        while pos >= 0:
            starts.append(pos)
            pos = fat.find(magic, pos + 1)
        bundles = [fat[a:b] for a, b in zip(starts, starts[1:] + [len(fat)])]
        mine = [x.rstrip(b"\0") for x in bundles if b"_ZN3hgk" in x]
        if not mine:
            raise ValueError("no hourglass code object in the library")
        h = hashlib.sha256()
        for x in sorted(mine):
            h.update(x)
        return h.hexdigest()[:16]
    except Exception:   # noqa: BLE001  (an unreadable library: hash the sources instead)
        h = hashlib.sha256()
        src = os.path.join(ROOT, "deepfly3d_amd", "csrc")
        for name in sorted(os.listdir(src)):
            if name.endswith((".hip", ".h")):
                h.update(name.encode())
                h.update(open(os.path.join(src, name), "rb").read())
        return "src:" + h.hexdigest()[:12]


def cpu_baseline(state_dict, frames_cpu, calib, target_seconds):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores, bounded sample."""
    from oracle import geometry as og
    from oracle import hourglass_torch as oh

    cores = host_cores()
    torch.set_num_threads(cores)
    net = oh.HourglassNet()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_dict.items()}, strict=False)
    net.eval()
    P = og.projection_matrices(calib["R"], calib["tvec"], calib["intr"])

    def run(frames):
        F = frames.shape[0]
        hm = oh.forward_nhwc(net, frames.reshape(F * 7, *frames.shape[2:])).numpy()
        pts, conf = og.heatmap_argmax(hm)
        pts = pts.reshape(F, 7, 19, 2).transpose(1, 0, 2, 3)
        p38 = og.relayout_19_to_38(pts, list(range(7)))
        return og.triangulate_dlt_batched(og.pixels_from_normalised(p38, [960, 480]), P)

    t0 = time.time()
    run(frames_cpu[:1])
    t1 = time.time() - t0
    n = int(max(1, min(frames_cpu.shape[0], round(target_seconds / max(t1, 1e-3)))))
    t0 = time.time()
    run(frames_cpu[:n])
    dt = time.time() - t0
    return {
        "value": n / dt,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} frames x 7 views (256x512 f32): torch-CPU fp32 hourglass + numpy arg-max/layout/DLT, {dt:.1f} s",
    }


def measure_roofline(engine, dtype, run_steps, nprof, default_size=True):
    """HIP events around every launch of each kernel class (same stream), over `nprof` steps; every kernel priced against
    both roofs with every byte model (module docstring), the dominant kernel's larger fraction is the line's `frac`."""
    import ctypes

    from deepfly3d_amd import _native

    lib = _native.load()
    _native.check(lib.df3d_hg_profile(engine.h, 1))
    run_steps(nprof)
    torch.cuda.synchronize()
    traffic, traffic_src = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f)
        traffic_src = (traffic.get("_meta") or {}).get("kernel_source_sha")
    per = []
    buf = ctypes.create_string_buffer(128)
    for k in range(lib.df3d_hg_profile_count(engine.h)):
        ms, fl, by, m1, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _native.check(lib.df3d_hg_profile_read(engine.h, k, buf, 128, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(m1), ctypes.byref(n)))
        if not n.value:
            continue
        name = buf.value.decode()
        ex = ctypes.c_double()
        _native.check(lib.df3d_hg_profile_executed_flops(engine.h, k, ctypes.byref(ex)))
        sec = 1e-3 * ms.value / n.value   # average launch
        pmc = traffic.get(name) if default_size else None   # (traffic.json holds bytes per launch AT THE DEFAULT STEP SIZE, 128 frames)
        # tflops = what the launch EXECUTED on the matrix pipe; tflops_direct = the direct-convolution FLOPs of its plan steps (the reference's
        # arithmetic; larger for the Winograd tail, which does a 3x3 with 16/36 of the multiplies) -- fractions of the roof are executed / peak
        row = {"kernel": name, "launches": n.value, "avg_us": 1e6 * sec, "total_ms": ms.value, "tflops": ex.value / n.value / sec / 1e12,
               "tflops_direct": fl.value / n.value / sec / 1e12, "flops_executed": ex.value, "flops_direct": fl.value,
               "bytes_min": by.value / n.value, "bytes_m1": m1.value / n.value, "bytes_pmc": pmc}
        row["frac_mfma"] = row["tflops"] / PEAK_TFLOPS[dtype]
        row["frac_hbm_min"] = row["bytes_min"] / sec / 1e9 / PEAK_HBM_GBS
        row["frac_hbm_m1"] = row["bytes_m1"] / sec / 1e9 / PEAK_HBM_GBS
        row["frac_hbm_pmc"] = pmc / sec / 1e9 / PEAK_HBM_GBS if pmc else None
        per.append(row)
    per.sort(key=lambda d: -d["total_ms"])
    _native.check(lib.df3d_hg_profile(engine.h, 0))
    # the whole hourglass step's HBM bytes as the counters saw them (every kernel of the step has a PMC figure, else None): BASELINE's
    # "hourglass HBM GB/s vs peak" as MOVED bytes, beside the M1 convention of config.hourglass_gbs_m1_end_to_end
    covered = sum(k["total_ms"] for k in per if k["bytes_pmc"]) / max(sum(k["total_ms"] for k in per), 1e-30)
    # (traffic.json holds bytes per launch AT THE DEFAULT STEP SIZE, 128 frames: no step figure for other sizes)
    step_pmc = sum(k["launches"] * k["bytes_pmc"] for k in per if k["bytes_pmc"]) / nprof if covered > 0.99 and default_size else None
    step_min = sum(k["launches"] * k["bytes_min"] for k in per) / nprof
    dom = per[0]
    hbm_frac = dom["frac_hbm_pmc"] if dom["frac_hbm_pmc"] is not None else dom["frac_hbm_min"]
    hbm_bytes = dom["bytes_pmc"] if dom["frac_hbm_pmc"] is not None else dom["bytes_min"]
    bound = "mfma" if dom["frac_mfma"] >= hbm_frac else "hbm"
    sec = dom["avg_us"] * 1e-6
    return {
        "bound": bound,
        "kernel": dom["kernel"],
        "achieved": dom["tflops"] if bound == "mfma" else hbm_bytes / sec / 1e9,
        "peak": PEAK_TFLOPS[dtype] if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": dom["frac_mfma"] if bound == "mfma" else hbm_frac,
        "frac_is": "frac_mfma" if bound == "mfma" else ("frac_hbm_pmc" if dom["frac_hbm_pmc"] is not None else "frac_hbm_min"),
        "fractions": {k: dom[k] for k in ("frac_mfma", "frac_hbm_min", "frac_hbm_m1", "frac_hbm_pmc")},
        "fractions_legend": "frac_mfma: executed FLOPs / dense MFMA peak; frac_hbm_min: inputs once + outputs once / 8 TB/s; frac_hbm_m1: bytes of the "
                            "fusion model M1 (SURVEY.md 8d: every convolution's input and output) / 8 TB/s, a convention above what fused kernels move; "
                            "frac_hbm_pmc: rocprofv3 FETCH/WRITE counters / 8 TB/s; bound = the larger of frac_mfma and frac_hbm_pmc",
        "traffic": dom["bytes_pmc"],  # HBM bytes per launch from rocprofv3 PMC passes ((2 x FETCH_SIZE + WRITE_SIZE) x 1024), profiles/traffic.json
        "traffic_kernel_source_sha": traffic_src,
        "traffic_is_current": (traffic_src == kernel_source_sha()) if traffic_src else None,
        "avg_launch_us": dom["avg_us"],
        "step_hbm_bytes_pmc": step_pmc,   # per hourglass step (all kernels): counters; None unless traffic.json covers > 99 % of the step's kernel time
        "step_hbm_bytes_min": step_min,   # ... and the least the step's launches can move (inputs once + outputs once per launch)
        "step_kernel_ms": sum(k["total_ms"] for k in per) / nprof,
        "executed_tflops": dom["tflops"],                      # the dominant kernel: FLOPs executed on the matrix pipe / time (= `achieved` when MFMA-bound) ...
        "direct_equivalent_tflops": dom["tflops_direct"],      # ... and the direct-convolution FLOPs of the same plan steps / time
        "step_flops_executed": sum(k["flops_executed"] for k in per) / nprof,
        "step_flops_direct": sum(k["flops_direct"] for k in per) / nprof,
        "kernels": per,
    }


POOL_CHUNK = 64   # frames per generator call (bounded temporary memory); part of the pool's definition: `pool_frames_of` replays it


def fill_pool(frames, seed_rank, dev):
    """The resident frame pool of rank `seed_rank`: seeded uniform values, drawn POOL_CHUNK frames at a time."""
    gen = torch.Generator(device=dev).manual_seed(seed_rank)
    for i in range(0, frames.shape[0], POOL_CHUNK):
        frames[i : i + POOL_CHUNK].uniform_(0.0, 1.0, generator=gen)


def pool_frames_of(seed_rank, pool, wanted, dev):
    """Frames `wanted` (sorted pool indices) of rank `seed_rank`'s pool, regenerated on `dev` by replaying that rank's generator calls
    chunk by chunk (0.7 GB of temporary memory, not the peer's 11 GB pool)."""
    gen = torch.Generator(device=dev).manual_seed(seed_rank)
    out = torch.empty((len(wanted), 7, 256, 512, 3), dtype=torch.float32, device=dev)
    tmp = torch.empty((POOL_CHUNK, 7, 256, 512, 3), dtype=torch.float32, device=dev)
    last = max(wanted) if wanted else -1
    for i in range(0, last + 1, POOL_CHUNK):
        n = min(POOL_CHUNK, pool - i)
        tmp[:n].uniform_(0.0, 1.0, generator=gen)
        for k, w in enumerate(wanted):
            if i <= w < i + n:
                out[k] = tmp[w - i]
    return out


def verify_sample_plan(ranges, per_rank):
    """[(rank, [local frame indices])]: first, last and evenly strided interior frames of every rank's non-empty range."""
    plan = []
    for r, (t0, t1) in enumerate(ranges):
        n = t1 - t0
        if n > 0:
            plan.append((r, sorted({int(round(x)) for x in np.linspace(0, n - 1, min(per_rank, n))})))
    return plan


class Job:
    """One workload on one engine: frames streamed from the resident pool through the pipeline in steps, optional bundle
    adjustment per window on a worker thread, optional packed gather -- timed as the contract asks."""

    def __init__(self, a, engine, frames, calib, dev, rank, world, total_frames, steps, ba_window, collective, force_collective, global_frames=None):
        from deepfly3d_amd.pipeline import FramePipeline

        # global_frames: strong scaling -- the length of the ONE sequence the ranks share (this rank holds total_frames of it, its
        # shard_range); None: weak scaling, every rank holds total_frames of a world x total_frames sequence
        self.global_frames = global_frames
        self.tail_ms = {}
        self.settled = None
        self.a, self.engine, self.frames, self.calib, self.dev = a, engine, frames, calib, dev
        self.rank, self.world, self.total_frames, self.steps, self.ba_window = rank, world, total_frames, steps, ba_window
        self.collective, self.force_collective = collective, force_collective
        self.fps_step = a.frames_per_step
        self.pool = frames.shape[0]
        self.align = ba_window if ba_window > 0 else 1
        self.pipe = FramePipeline(engine, calib["R"], calib["tvec"], calib["intr"])
        self.outs = self.pipe.allocate_outputs(total_frames)
        self.ba_px, self.ba_runs, self.ba_cams, self.ba_ms, self.ba_futures = None, [], [], [], []
        self.ba_pool = self.ba_stream = None
        self.timeline = {"enq": [], "ev": [], "ba": []} if os.environ.get("DF3D_BENCH_TIMELINE") else None
        if ba_window > 0:
            # geometry-consistent detections, one set per window: golden-like pose tiled + jitter, projected through
            # the adjusted cameras of the sample set, quantised to the heat-map grid (the random-weight network output is
            # meaningless for BA); prepared before the timed region like the frames
            from concurrent.futures import ThreadPoolExecutor

            from deepfly3d_amd.synthetic import synthetic_ba_window

            g3 = np.load(os.path.join(ROOT, "tests", "golden", "golden_3d.npz"))
            nwin = -(-total_frames // ba_window)
            self.ba_px = [synthetic_ba_window(g3["points3d_wo_procrustes"], g3["R"], g3["tvec"], g3["intr"], min(ba_window, total_frames - w * ba_window), rank, w)
                          for w in range(nwin)]
            # configs[4]: the re-calibration of a finished window runs on its own HIP stream from a worker thread (its inputs do
            # not depend on the frames still in flight), so its ~150 small kernels and ~40 host synchronisations slot in beside the
            # next batches' hourglass instead of draining the pipeline; every window is joined before the gather, inside the timed region
            self.ba_pool = ThreadPoolExecutor(max_workers=1)
            self.ba_stream = torch.cuda.Stream(device=dev)

    def recalibrate(self, window_px):
        from deepfly3d_amd.bundle_adjust import bundle_adjust

        t_ba = time.perf_counter()
        with torch.cuda.device(self.dev), torch.cuda.stream(self.ba_stream):
            Rn, tn, info = bundle_adjust(window_px, self.calib["R"], self.calib["tvec"], self.calib["intr"], device=self.dev, return_info=True,
                                         concurrent=True)   # beside the frame pipeline: the launch-based LSMR (no co-resident workgroups needed)
        self.ba_ms.append(round(1e3 * (time.perf_counter() - t_ba), 1))   # wall time of the solve (beside the pipeline when threaded)
        if self.timeline is not None:
            self.timeline["ba"].append((t_ba, time.perf_counter()))
        return np.concatenate([Rn.reshape(7, 9), tn.reshape(7, 3)], axis=1), info["nfev"]

    def step(self, i, record=True, solve=True):
        tl = self.timeline
        if tl is not None and record:
            tl["enq"].append(time.perf_counter())
        f0 = i * self.fps_step
        n = min(self.fps_step, self.total_frames - f0)
        lo = f0 % self.pool
        self.pipe.run_batch(self.frames[lo : lo + n], *self.outs, f0)
        if tl is not None and record:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            tl["ev"].append(ev)
        # a window closes with this batch (the last window of the share may be shorter)
        w = self.ba_window
        if solve and self.ba_px is not None and ((f0 + n) // w > f0 // w or (f0 + n == self.total_frames and self.total_frames % w)):
            closed = (f0 + n) // w - 1 if (f0 + n) // w > f0 // w else len(self.ba_px) - 1
            fut = self.ba_pool.submit(self.recalibrate, self.ba_px[closed])
            if record:
                self.ba_futures.append(fut)
            else:
                fut.result()

    def join_recalibrations(self):
        for fut in self.ba_futures:
            cams, nfev = fut.result()
            self.ba_cams.append(cams)
            self.ba_runs.append(nfev)
        del self.ba_futures[:]

    def gather(self):
        from deepfly3d_amd import distributed as dd

        if not self.collective:
            return None
        cams = None
        if self.ba_window > 0:
            cams = torch.from_numpy(np.stack(self.ba_cams) if self.ba_cams else np.zeros((0, 7, 12))).to(self.dev)
        num = self.global_frames if self.global_frames is not None else self.total_frames * self.world
        return dd.gather_results(*self.outs, num_frames=num, rank=self.rank, world_size=self.world, align=self.align, cameras=cams,
                                 force_collective=self.force_collective)

    def sequence_tail(self, gathered):
        """Strong scaling: what only rank 0 can do once the whole sequence is there -- Procrustes is sequence-global (medians over
        all frames, reference df3d/procrustes.py:123-135) -- on the device, inside the timed region."""
        from deepfly3d_amd.procrustes import procrustes_separate

        if self.global_frames is None or self.rank != 0:
            return None
        p3 = gathered[2] if gathered is not None else self.outs[2]
        return procrustes_separate(p3, device=self.dev, return_tensor=True)

    def settle(self, min_ms=300.0, agree=0.05, max_steps=64, max_ms=5000.0):
        """After the W warm-up steps: keep stepping (untimed) until the DEVICE is warm, not merely the code -- at least `min_ms` of GPU work
        has run and two consecutive steps agree within `agree`.  An idle MI355X sits at ~150 MHz and takes several 50-ms samples to reach
        its working clocks (profiles/r04_power_f16_samples.txt): a short run (--steps 3 of 32 frames = 30 ms of work) whose warm-up is a
        step COUNT measures that ramp, a third of the device's rate on a fresh box (GPUTEST_r04).  Bounded by `max_steps` / `max_ms`."""
        full = max(1, self.total_frames // self.fps_step)   # full batches only: a short last batch is not comparable
        total, prev, n = 0.0, None, 0
        torch.cuda.synchronize()
        while n < max_steps and total < max_ms:
            t0 = time.perf_counter()
            self.step(n % full, record=False, solve=False)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
            total += ms
            n += 1
            if total >= min_ms and prev is not None and abs(ms - prev) <= agree * max(ms, prev):
                break
            prev = ms
        self.settled = {"steps": n, "ms": round(total, 1), "last_step_ms": round(ms, 3)}

    def run(self, warmup):
        """W untimed steps, then exactly `steps` timed steps (+ joins + gather) between barrier + synchronize on both sides."""
        dist = torch.distributed
        for w in range(warmup if self.total_frames > 0 else 0):
            self.step(w % max(1, -(-self.total_frames // self.fps_step)), record=False, solve=False)
        if self.total_frames > 0:
            self.settle()
        if self.ba_px is not None:
            # the re-calibration has one-time costs of its own (allocator growth on its stream, the LSMR chunk's graph: ~0.7 s per
            # call until buffers and graph settle) and no window closes inside the W warm-up steps: warm it on its worker thread,
            # where its graph cache lives
            for _ in range(2):
                self.ba_pool.submit(self.recalibrate, self.ba_px[0]).result()
            del self.ba_ms[:]
        if self.collective and self.world == 1:
            # a multi-rank run creates its RCCL communicator in the barrier below; the 1-rank group of --force-collective would
            # create it inside the first gather, i.e. inside the timed region (0.9 s when nothing else hides it)
            dist.all_reduce(torch.zeros(1, device=self.dev))
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        t_start = time.perf_counter()
        tl = self.timeline
        if tl is not None:
            tl["ev0"] = torch.cuda.Event(enable_timing=True)
            tl["ev0"].record()
        ev_a = ev_b = None
        if self.a.verify and self.dev.type == "cuda":   # this rank's own batches on the GPU clock (no host synchronisation added)
            ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev_a.record()
        for i in range(self.steps):
            if i * self.fps_step < self.total_frames:   # (strong scaling: a shorter shard has fewer batches)
                self.step(i)
        if ev_b is not None:
            ev_b.record()
        marks = [time.perf_counter()]
        self.join_recalibrations()
        marks.append(time.perf_counter())
        gathered = self.gather()
        marks.append(time.perf_counter())
        self.aligned = self.sequence_tail(gathered)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        marks.append(time.perf_counter())
        self.gathered = gathered
        self.own_steps_ms = ev_a.elapsed_time(ev_b) if ev_b is not None else 1e3 * (marks[0] - t_start)
        self.tail_ms = {"steps_enqueued": 1e3 * (marks[0] - t_start), "recalibrations_joined": 1e3 * (marks[1] - marks[0]),
                        "gather": 1e3 * (marks[2] - marks[1]), "procrustes_and_drain": 1e3 * (marks[3] - marks[2])}
        if self.world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        if tl is not None:   # development aid: host enqueue time, GPU completion time of every step, the solves' intervals
            ev0 = tl["ev"][0]
            done = [ev0.elapsed_time(e) for e in tl["ev"]]
            enq = [1e3 * (x - t_start) for x in tl["enq"]]
            print("TIMELINE step: enqueue_ms gpu_done_ms(from step 0's end)", file=sys.stderr)
            for i in range(0, len(enq), max(1, len(enq) // 24)):
                print(f"  step {i:3d}: {enq[i]:8.1f} {done[i]:8.1f}", file=sys.stderr)
            print("  step 0 gpu done", round(tl["ev0"].elapsed_time(ev0), 1), "ms after the start; last step", round(done[-1], 1), "ms after step 0; elapsed", round(1e3 * elapsed, 1), file=sys.stderr)
            print("  host marks (ms): loop end, joined, gather returned, synchronised:", [round(1e3 * (m - t_start), 1) for m in marks], file=sys.stderr)
            print("  BA [start, end] ms:", [(round(1e3 * (a0 - t_start)), round(1e3 * (a1 - t_start))) for a0, a1 in tl["ba"][-len(self.ba_runs):]], file=sys.stderr)
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        gather_ok = None
        if self.collective and self.rank == 0 and self.world == 1:  # the 1-rank collective must hand back exactly what went in
            gather_ok = all(torch.equal(g, o) for g, o in zip(gathered[:3], self.outs))
        return elapsed, gather_ok

    def verify(self, ranges, per_rank=6):
        """--verify, after the timed region.  Every rank reports the rate of its own batches; rank 0 then recomputes a strided sample of every
        rank's frames on ITS device and compares with the gathered records bit for bit (tests/test_gpu_core.py:
        test_batch_composition_does_not_change_results is why this must hold: a frame's record does not depend on the batch it ran in,
        nor on the GPU it ran on).  Returns the line's fields on rank 0, None elsewhere."""
        dist = torch.distributed
        mine = torch.tensor([self.total_frames / max(self.own_steps_ms, 1e-9) * 1e3], dtype=torch.float64, device=self.dev)
        rates = [mine.clone() for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(rates, mine)
        if self.rank != 0:
            return None
        got = self.gathered[:3] if self.gathered is not None else self.outs
        plan = verify_sample_plan(ranges, per_rank)
        checked, bad = 0, []
        for r, local in plan:
            t0 = ranges[r][0]
            outs = self.recompute(r, local, t0)
            idx = torch.tensor([t0 + k for k in local], device=got[2].device)
            same = (torch.equal(outs[0].to(got[0].device), got[0][:, idx]) and torch.equal(outs[1].to(got[1].device), got[1][:, idx])
                    and torch.equal(outs[2].to(got[2].device), got[2][idx]))
            checked += len(local)
            if not same:
                bad.append(r)
        out = {"verify": {"frames_recomputed_on_rank0": checked, "ranks_sampled": len(plan), "bit_identical": not bad, "ranks_differing": bad},
               "per_rank_frames_per_s": [float(x.item()) for x in rates], "gather_ms": self.tail_ms.get("gather"),
               "rccl_world": dist.get_world_size() if dist.is_initialized() else 1}
        if bad:
            raise SystemExit(f"--verify: gathered records of rank(s) {bad} differ from rank 0's recomputation: {json.dumps(out)}")
        return out

    def recompute(self, r, local, t0):
        """Rank `r`'s local frames `local` through THIS rank's pipeline, as one batch."""
        outs = self.pipe.allocate_outputs(len(local))
        self.pipe.run_batch(self.frames_of_rank(r, local), *outs, 0)
        return outs

    def frames_of_rank(self, r, local):
        """The input frames rank `r` fed for its local frame indices `local` (frame f of a rank = its pool[f % pool])."""
        want = sorted({k % self.pool for k in local})
        got = pool_frames_of(r, self.pool, want, self.dev) if r != self.rank else self.frames[torch.tensor(want, device=self.dev)]
        pos = {w: i for i, w in enumerate(want)}
        return got[torch.tensor([pos[k % self.pool] for k in local], device=self.dev)]

    def roofline(self, dtype):
        return measure_roofline(self.engine, dtype, lambda n: [self.step(i, record=False, solve=False) for i in range(n)], min(self.steps, 4),
                                default_size=self.fps_step == 128)

    def close(self):
        if self.ba_pool is not None:
            self.ba_pool.shutdown(wait=True)


def ensure_one_rank_group(dev):
    """--force-collective at N = 1: a 1-rank RCCL process group, so that the packed gather really executes."""
    if not torch.distributed.is_initialized():
        if os.environ.get("MASTER_PORT"):
            port = int(os.environ["MASTER_PORT"])
        else:   # any free port: nothing else has to find this one-rank group
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)


def hourglass_leg(a, sd, dtype, frames, calib, dev, total_frames, config_words):
    """configs[2]-style leg at N = 1: the headline's frames through another engine, timed like the headline."""
    from deepfly3d_amd.hourglass import HourglassEngine

    eng = HourglassEngine(sd, dtype=dtype, device=dev)
    job = Job(a, eng, frames, calib, dev, 0, 1, total_frames, a.steps, 0, False, False)
    elapsed, _ = job.run(max(1, a.warmup))
    fl, by = eng.work(a.frames_per_step * 7)
    sec = elapsed / a.steps
    leg = {
        "workload": f"{config_words}: {total_frames} frames x 7 views of 256x512x3, 2-stack hourglass {DTYPE_WORDS[dtype]}, arg-max + 38-joint layout + fp64 DLT with fixed calib.pkl",
        "workload_short": f"{config_words.split(' on ')[0]}: {total_frames} frames x 7 views, hourglass {DTYPE_SHORT[dtype]}",
        "value": total_frames / elapsed, "unit": "frames/s", "steps": a.steps, "ms_per_step": 1e3 * sec, "dtype": dtype,
        "hourglass_tflops_end_to_end": fl / sec / 1e12,
        "hourglass_frac_mfma_end_to_end": fl / sec / 1e12 / PEAK_TFLOPS[dtype],
        "hourglass_gbs_m1_end_to_end": by / sec / 1e9,
        "hourglass_frac_hbm_m1_end_to_end": by / sec / 1e9 / PEAK_HBM_GBS,
    }
    if not a.no_roofline:
        leg["roofline"] = job.roofline(dtype)
        add_step_hbm(leg, leg["roofline"], sec)
        add_step_flops(leg, leg["roofline"], sec, dtype)
    if dtype == "f32s":   # the leg's price: what it differs by from the exact-fp32 engine, measured here on 2 frames of the run's own input
        x = frames[:2].reshape(14, 256, 512, 3).contiguous()
        exact = HourglassEngine(sd, dtype="f32", device=dev)
        ref = exact.forward(x).clone()
        got = eng.forward(x).clone()
        torch.cuda.synchronize()   # both engines have finished before either goes away
        leg["max_rel_diff_vs_f32_engine"] = float((got - ref).abs().max() / ref.abs().max())
        leg["argmax_cells_identical_to_f32_engine"] = bool(torch.equal(got.flatten(2).argmax(-1), ref.flatten(2).argmax(-1)))
        del exact
    del job, eng
    return leg


def add_step_flops(dst, roof, sec, dtype):
    """The hourglass step's matrix-pipe rate as EXECUTED FLOPs (per-kernel counts of the roofline pass) over this run's step time: the fraction of
    the roof; the direct-convolution rate (`hourglass_tflops_end_to_end`: the plan's FLOPs in the reference's arithmetic) stays beside it."""
    if roof.get("step_flops_executed"):
        dst["hourglass_tflops_executed_end_to_end"] = roof["step_flops_executed"] / sec / 1e12
        dst["hourglass_frac_mfma_end_to_end"] = roof["step_flops_executed"] / sec / 1e12 / PEAK_TFLOPS[dtype]


def add_step_hbm(dst, roof, sec):
    """BASELINE's second metric, "hourglass HBM GB/s vs peak", as bytes MOVED per step (rocprofv3 counters of profiles/traffic.json x this run's
    launch counts) over this run's step time -- beside the M1 convention, which charges fused-away traffic."""
    if roof.get("step_hbm_bytes_pmc"):
        dst["hourglass_gbs_pmc_end_to_end"] = roof["step_hbm_bytes_pmc"] / sec / 1e9
        dst["hourglass_frac_hbm_pmc_end_to_end"] = roof["step_hbm_bytes_pmc"] / sec / 1e9 / PEAK_HBM_GBS
    dst["hourglass_gbs_min_end_to_end"] = roof["step_hbm_bytes_min"] / sec / 1e9


def share_leg(a, sd, dtype, frames, calib, dev):
    """A short share of configs[4] in-process: rank 0 of 8 ranks of a 16 000-frame stream (2 000 frames), one bundle adjustment
    per 1 000-frame window beside the pipeline, the packed gather (frames + window cameras) executed on a 1-rank RCCL group."""
    from deepfly3d_amd import distributed as dd
    from deepfly3d_amd.hourglass import HourglassEngine

    ensure_one_rank_group(dev)
    t0, t1 = dd.shard_range(16000, 8, 0, 1000)
    total = t1 - t0
    steps = -(-total // a.frames_per_step)
    eng = HourglassEngine(sd, dtype=dtype, device=dev)
    job = Job(a, eng, frames, calib, dev, 0, 1, total, steps, 1000, True, True)
    elapsed, ok = job.run(max(1, a.warmup))
    out = {
        "workload": f"BASELINE configs[4], a short share on one GPU: rank 0 of 8 ranks of a 16000-frame 7-view stream = {total} frames through the resident pool, "
                    f"2-stack hourglass {DTYPE_WORDS[dtype]}, arg-max + 38-joint layout + fp64 DLT, bundle-adjustment re-calibration every 1000 frames, "
                    "packed gather (frame records + window cameras) executed on a 1-rank RCCL group; the full 13 000-frame share: profiles/r03_rankshare_*.json",
        "workload_short": f"BASELINE configs[4], rank 0 of 8 of a 16000-frame stream: {total} frames, hourglass {DTYPE_SHORT[dtype]}, BA every 1000 frames, gather on a 1-rank RCCL group",
        "value": total / elapsed, "unit": "frames/s", "dtype": dtype, "frames": total, "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
        "bundle_adjust_runs": len(job.ba_runs), "bundle_adjust_nfev": job.ba_runs, "bundle_adjust_wall_ms": job.ba_ms[-len(job.ba_runs):],
        "gather_roundtrip_exact": ok, "collective_backend": torch.distributed.get_backend(),
    }
    job.close()
    del job, eng
    return out


class _StubPipeline:
    """--dry-run: stands in for FramePipeline on the CPU.  Every frame's points3d[t, 0, 0] (and points2d / confidence) carry the frame's
    GLOBAL index, so rank 0 can check that the gather put every shard where it belongs."""

    def __init__(self, frame0):
        self.frame0 = frame0

    def allocate_outputs(self, T):
        return (torch.zeros((7, T, 38, 2), dtype=torch.float64), torch.zeros((7, T, 19), dtype=torch.float32), torch.zeros((T, 38, 3), dtype=torch.float64))

    def run_batch(self, frames, p2, cf, p3, t0):
        n = frames.shape[0]
        idx = torch.arange(self.frame0 + t0, self.frame0 + t0 + n, dtype=torch.float64)
        p3[t0:t0 + n] = idx[:, None, None]
        p2[:, t0:t0 + n] = idx[None, :, None, None]
        cf[:, t0:t0 + n] = (idx % 1024).to(torch.float32)[None, :, None]


def dry_run(a):
    """The N-rank control flow of `--strong` / the weak default on CPU tensors over gloo: what an 8-GPU box will execute around the kernels."""
    from deepfly3d_amd import distributed as dd

    rank, world, _ = dd.init_from_env(backend="gloo")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    fps_step = a.frames_per_step
    align = a.ba_window if a.ba_window > 0 else 1
    if a.strong:
        shards, a.steps = strong_plan(a.stream_frames, world, align, fps_step)
        t0, t1, _ = shards[rank]
        total, global_frames = t1 - t0, a.stream_frames
    else:
        shards = None
        total, global_frames, t0 = a.steps * fps_step, None, rank * a.steps * fps_step
    job = Job.__new__(Job)
    job.a, job.rank, job.world, job.total_frames, job.steps, job.ba_window = a, rank, world, total, a.steps, a.ba_window
    job.collective, job.force_collective, job.global_frames, job.align = world > 1, False, global_frames, align
    job.fps_step, job.pool, job.dev = fps_step, fps_step, torch.device("cpu")
    job.frames = torch.zeros((fps_step, 1))
    job.pipe = _StubPipeline(t0)
    job.outs = job.pipe.allocate_outputs(total)
    job.timeline = None
    nwin = -(-total // a.ba_window) if a.ba_window > 0 else 0
    job.ba_px = None
    job.ba_cams = [np.full((7, 12), float(t0 // a.ba_window + w)) for w in range(nwin)] if nwin else []   # window w of the stream, stamped
    job.ba_runs, job.ba_ms, job.ba_futures, job.ba_pool = [0] * nwin, [], [], None
    if world > 1:
        torch.distributed.barrier()
    t_start = time.perf_counter()
    for i in range(a.steps):
        if i * fps_step < total:
            job.step(i, record=False, solve=False)
    cams = torch.from_numpy(np.stack(job.ba_cams)) if nwin or a.ba_window else None
    if a.ba_window and cams is None:
        cams = torch.zeros((0, 7, 12), dtype=torch.float64)
    num = global_frames if global_frames is not None else total * world
    t_gather = time.perf_counter()
    got = dd.gather_results(*job.outs, num_frames=num, rank=rank, world_size=world, align=align, cameras=cams)
    job.tail_ms = {"gather": 1e3 * (time.perf_counter() - t_gather)}
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t_start
    verified = None
    if a.verify:   # the real run's Job.verify, with the stub recomputing a peer's frames (it stamps global frame indices)
        ranges = [(c, b) for c, b, _ in shards] if shards else [(r * total, (r + 1) * total) for r in range(world)]

        def recompute(r, local, first):
            outs = job.pipe.allocate_outputs(len(local))
            for k, f in enumerate(local):
                _StubPipeline(first + f - k).run_batch(job.frames[:1], *outs, k)
            return outs

        job.recompute, job.gathered, job.own_steps_ms = recompute, got, 1e3 * (t_gather - t_start)
        verified = job.verify(ranges)
    if rank == 0:
        p2, cf, p3 = got[:3]
        seq = torch.arange(num, dtype=torch.float64)
        ok = bool(torch.equal(p3[:, 0, 0], seq) and torch.equal(p2[3, :, 7, 1], seq) and torch.equal(cf[0, :, 0], (seq % 1024).to(torch.float32)))
        if a.ba_window:
            ok = ok and got[3].shape[0] == -(-num // a.ba_window) and bool(torch.equal(got[3][:, 0, 0], torch.arange(got[3].shape[0], dtype=torch.float64)))
        print(json.dumps({"metric": "frames/sec (7-view 2D->3D)", "dry_run": True, "value": num / elapsed, "unit": "frames/s (CPU stub: control flow only)", "n_gpus": world,
                          "steps": a.steps, "scaling": "strong" if a.strong else "weak", "frames_per_gpu": [b - c for c, b, _ in shards] if shards else total,
                          "collective_backend": torch.distributed.get_backend() if world > 1 else None, "gather_check": ok,
                          "windows_gathered": int(got[3].shape[0]) if a.ba_window else None, **(verified or {})}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _sig(x, digits=6):
    """Floats to `digits` significant digits, recursively: the line is read by people and by a driver with a size limit."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


_ROOF_KEEP = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_is", "fractions", "traffic", "traffic_is_current", "avg_launch_us",
              "step_hbm_bytes_pmc", "step_kernel_ms", "direct_equivalent_tflops")
_LEG_ROOF_KEEP = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us")


def short_line(full):
    """The ONE stdout line, <= 4 KB (tests/test_gpu_bench.py asserts the size): the headline, `config` with a one-sentence workload, `roofline`
    without the per-kernel table, every leg as its rate + its dominant kernel's roofline + its one-off fields, `cpu_baseline`.  Everything
    dropped here is in the side file the line names (`tables`)."""
    out = {}
    for k, v in full.items():
        if k == "roofline":
            out[k] = {q: v[q] for q in _ROOF_KEEP if q in v}
        elif k == "config":
            out[k] = {q: w for q, w in v.items() if w is not None and q not in ("hbm_m1_note", "hourglass_gbs_min_end_to_end")}
            out[k]["workload"] = out[k].pop("workload_short", v["workload"][:200])
        elif isinstance(v, dict) and "value" in v and k != "cpu_baseline":   # an attached leg
            leg = {q: w for q, w in v.items() if q not in ("roofline", "workload", "workload_short", "unit") and not q.startswith("hourglass_")}
            leg["workload"] = v.get("workload_short", v["workload"][:100])
            if "roofline" in v:
                leg["roofline"] = {q: v["roofline"][q] for q in _LEG_ROOF_KEEP}
            for q in ("hourglass_frac_mfma_end_to_end", "hourglass_frac_hbm_pmc_end_to_end"):
                if q in v:
                    leg[q] = v[q]
            out[k] = leg
        else:
            out[k] = v
    return _sig(out)


def emit(full, a):
    """Rank 0's output: the whole record to the side file, the short line (or, with --full, the whole record) as the one stdout line."""
    path = a.tables
    if path is None:
        path = os.path.join(ROOT, "gpurun_out", f"bench_full_{full['dtype']}.json")
    if path != "-":
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f)
                f.write("\n")
            full["tables"] = os.path.relpath(path, ROOT)
        except OSError as e:   # a read-only tree: the line still goes out
            full["tables"] = f"not written: {e!r}"
    print(json.dumps(full if a.full else short_line(full)), flush=True)


def strong_plan(stream_frames, world, align, frames_per_step):
    """Strong scaling: per rank (first frame, last frame + 1, batches) of the ONE stream, and the batches of the largest shard
    (the `steps` every rank loops over; a shorter shard skips its missing batches).  Pure: the CPU tests walk it for 8 ranks."""
    from deepfly3d_amd import distributed as dd

    ranges = dd.all_ranges(stream_frames, world, align)
    per_rank = [(t0, t1, -(-(t1 - t0) // frames_per_step)) for t0, t1 in ranges]
    return per_rank, max(b for _, _, b in per_rank)


def main(argv=None):
    a = parse(argv)
    if a.dry_run:
        return dry_run(a)
    from deepfly3d_amd import _native
    from deepfly3d_amd import distributed as dd
    from deepfly3d_amd.config import load_calibration
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.synthetic import synthetic_state_dict

    rank, world, local_rank = dd.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    _native.require_gpu()
    dev_index = local_rank % torch.cuda.device_count()  # (== local_rank on a node with one GPU per rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    if a.force_collective and world == 1:
        ensure_one_rank_group(dev)
    collective = dd.collective_needed(world, a.force_collective)

    fps_step = a.frames_per_step
    align = a.ba_window if a.ba_window > 0 else 1
    global_frames = shards = None
    if a.strong:
        if a.rank_share > 0:
            raise SystemExit("--strong shards the stream over the real ranks; --rank-share emulates one rank of a larger run: pick one")
        shards, a.steps = strong_plan(a.stream_frames, world, align, fps_step)
        t0, t1, _ = shards[rank]
        total_frames, global_frames = t1 - t0, a.stream_frames
    elif a.rank_share > 0:
        t0, t1 = dd.shard_range(a.stream_frames, a.rank_share, 0, align)
        total_frames = t1 - t0
        a.steps = -(-total_frames // fps_step)
    else:
        total_frames = a.steps * fps_step
    if a.ba_window > 0 and total_frames % a.ba_window and world > 1 and not a.strong:
        # every rank sees the same arguments: all of them stop here, in front of any collective
        raise SystemExit("per-GPU frames must be a multiple of --ba-window when N > 1")
    sd = synthetic_state_dict(0)
    engine = HourglassEngine(sd, dtype=a.dtype, device=dev)
    cal = load_calibration()
    calib = {k: np.stack([cal[c][k] for c in range(7)]) for k in ("R", "tvec", "intr", "distort")}

    pool = a.pool_frames or min(max(a.steps, 1) * fps_step, 1024)
    pool = max(fps_step, (pool // fps_step) * fps_step)
    frames = torch.empty((pool, 7, 256, 512, 3), dtype=torch.float32, device=dev)
    fill_pool(frames, rank, dev)

    job = Job(a, engine, frames, calib, dev, rank, world, total_frames, a.steps, a.ba_window, collective, a.force_collective, global_frames)
    elapsed, gather_ok = job.run(a.warmup)

    verified = None
    if a.verify:
        ranges = [(t0, t1) for t0, t1, _ in shards] if a.strong else [(r * total_frames, (r + 1) * total_frames) for r in range(world)]
        verified = job.verify(ranges)

    roof = None
    if not a.no_roofline and rank == 0:
        roof = job.roofline(a.dtype)

    # the attached legs (N = 1, plain configs[1] run only): same process, same driver clock
    legs = {}
    if a.dtype == "f32" and world == 1 and not a.no_legs and a.rank_share == 0 and a.ba_window == 0 and not a.strong:
        legs["config1_f32_split"] = hourglass_leg(a, sd, "f32s", frames, calib, dev, total_frames, "BASELINE configs[1] with split products")
        legs["config2_bf16"] = hourglass_leg(a, sd, "bf16", frames, calib, dev, total_frames, "BASELINE configs[2]")
        legs["config2_f16"] = hourglass_leg(a, sd, "f16", frames, calib, dev, total_frames,
                                            "BASELINE configs[2] on IEEE half (the 16-bit engine inside the reference's 2e-3 confidence tolerance)")
        try:
            legs["config4_share"] = share_leg(a, sd, "f16", frames, calib, dev)
        except Exception as e:  # reporting only: never hide the headline
            legs["config4_share"] = {"error": repr(e)}

    if rank == 0:
        ms_step = 1e3 * elapsed / a.steps
        fl, by = engine.work(fps_step * 7)
        per_step = total_frames / max(a.steps, 1) / fps_step  # < 1 when the last batch of a rank share is short
        words = DTYPE_WORDS[a.dtype]
        if a.strong:
            cfg = 4 if a.ba_window else 3
            workload = (f"BASELINE configs[{cfg}], strong-scaled: ONE {a.stream_frames}-frame 7-view stream sharded by frame over {world} rank(s) "
                        f"(rank r runs shard_range(stream, {world}, r, {align}): {[t1 - t0 for t0, t1, _ in shards]} frames) through a {pool}-frame resident pool each, "
                        f"2-stack hourglass {words}, arg-max + 38-joint layout + fp64 DLT"
                        + (f", bundle-adjustment re-calibration every {a.ba_window} frames" if a.ba_window else "")
                        + ", ONE packed gather to rank 0" + ("" if collective else " (N = 1: nothing to gather)") + ", Procrustes over the whole sequence on rank 0 -- all inside the timed region")
        elif a.rank_share:
            cfg = 4 if a.ba_window else 3
            workload = (f"BASELINE configs[{cfg}], ONE rank's share on one GPU: rank 0 of {a.rank_share} ranks of a {a.stream_frames}-frame 7-view stream = "
                        f"{total_frames} frames streamed through a {pool}-frame resident pool, 2-stack hourglass {words}, arg-max + 38-joint layout + fp64 DLT"
                        + (f", bundle-adjustment re-calibration every {a.ba_window} frames" if a.ba_window else "")
                        + (", packed gather executed on a 1-rank RCCL group" if collective else ""))
        elif a.ba_window:
            workload = (f"BASELINE configs[4] (per-GPU share): {total_frames} frames x 7 views of 256x512x3 per GPU, 2-stack hourglass {words}, "
                        f"arg-max + 38-joint layout + fp64 DLT, bundle-adjustment re-calibration every {a.ba_window} frames")
        else:
            workload = (f"BASELINE configs[{1 if a.dtype in ('f32', 'f32s') else 2}]: {total_frames} frames x 7 views of 256x512x3 per GPU, 2-stack hourglass {words}, "
                        "arg-max + 38-joint layout + fp64 DLT with fixed calib.pkl")
        cfg_no = (4 if a.ba_window else 3) if (a.strong or a.rank_share or a.ba_window) else (1 if a.dtype in ("f32", "f32s") else 2)
        short = (f"BASELINE configs[{cfg_no}]" + (", strong-scaled" if a.strong else f", rank 0 of {a.rank_share}" if a.rank_share else "")
                 + f": {a.stream_frames if a.strong else total_frames} frames x 7 views 256x512x3" + ("" if a.strong else " per GPU")
                 + f", 2-stack hourglass {DTYPE_SHORT[a.dtype]}, arg-max + 38-joint layout + fp64 DLT"
                 + (f", BA every {a.ba_window}" if a.ba_window else "") + (", one packed gather + Procrustes on rank 0" if a.strong else ""))
        sec = ms_step * 1e-3
        line = {
            "metric": "frames/sec (7-view 2D->3D)",
            "value": (a.stream_frames if a.strong else world * total_frames) / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "warmup_settle": job.settled,   # untimed steps after the W warm-up steps, until the clocks are up: >= 300 ms of GPU work and two steps within 5 %
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong" if a.strong else "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic (seeded frames resident in HBM, seeded hourglass weights, data/calib.pkl cameras)",
            "config": {
                "workload": workload,
                "workload_short": short,
                "frames_per_step": fps_step,
                "frames_per_gpu": total_frames if not a.strong else [t1 - t0 for t0, t1, _ in shards],
                "stream_frames": a.stream_frames if a.strong else None,
                "rank0_tail_ms": job.tail_ms if a.strong else None,   # host marks on rank 0: batches enqueued / recalibrations joined / gather / Procrustes + drain
                "parallelism": f"frame-sharded x{world}, one packed gather" + (" (executed: RCCL, 1 rank)" if collective and world == 1 else ""),
                "collective_executed": bool(collective),
                "collective_backend": torch.distributed.get_backend() if collective else None,
                "gather_roundtrip_exact": gather_ok,
                "bundle_adjust_every_frames": a.ba_window or None,
                "bundle_adjust_runs_rank0": len(job.ba_runs) or None,
                "bundle_adjust_nfev": job.ba_runs or None,
                "bundle_adjust_wall_ms": job.ba_ms[-len(job.ba_runs):] if job.ba_runs else None,
                "hourglass_tflops_end_to_end": per_step * fl / sec / 1e12,   # direct-convolution FLOPs of the plan (the reference's arithmetic) / step time
                "hourglass_frac_mfma_end_to_end": None,   # executed FLOPs / peak: filled in by the roofline pass (add_step_flops)
                "hourglass_gbs_m1_end_to_end": per_step * by / sec / 1e9,
                "hourglass_frac_hbm_m1_end_to_end": per_step * by / sec / 1e9 / PEAK_HBM_GBS,
                "hbm_m1_note": "hourglass activation bytes of the fusion model M1 (SURVEY.md 8d: a convention, above what the fused kernels move) / step time / 8 TB/s",
            },
        }
        if verified is not None:
            line.update(verified)
        if roof is not None:
            line["roofline"] = roof
            if per_step == 1.0:
                add_step_hbm(line["config"], roof, sec)
                add_step_flops(line["config"], roof, sec, a.dtype)
        line.update(legs)
        if not a.no_cpu_baseline and world == 1 and a.rank_share == 0 and not a.strong:
            try:
                line["cpu_baseline"] = cpu_baseline(sd, frames[:16].cpu(), calib, a.cpu_seconds)
            except Exception as e:  # the baseline is reporting only; never hide the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        emit(line, a)
    job.close()
    if torch.distributed.is_initialized():
        if world > 1:
            torch.distributed.barrier()   # rank 0 is still measuring the per-kernel table while the others are done: tear down together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
