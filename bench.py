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

The line's headline is configs[1] (fp32).  At N = 1 the same process then times configs[2] (bf16 hourglass, same
frames) and attaches it as `config2_bf16` with its own roofline, so both precisions are under the driver's clock.

    python bench.py --rank-share 8 --stream-frames 100000 [--ba-window 1000] [--force-collective]
runs ONE rank's share of BASELINE configs[3] / configs[4] on the one GPU at hand: rank 0's frame range of the
100 k-frame stream sharded over 8 ranks (aligned to the bundle-adjustment window), streamed through the resident
frame pool, one bundle adjustment per window interleaved, and -- with --force-collective -- the single packed gather
executed on a 1-rank RCCL process group, all inside the timed region.
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

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}  # /opt/skills/guides/MI355X_MICROARCH.md (dense)
PEAK_HBM_GBS = 8000.0
# which roof binds the hourglass per dtype (fp32: AI 55.6 FLOP/B > 19.7 balance -> FLOP-bound;
# bf16 on MFMA: 323.7 MB/view of activation traffic in the fusion model M1 -> HBM-bound)
BOUND = {"f32": "mfma", "bf16": "hbm", "f16": "hbm"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16"], default="f32")
    ap.add_argument("--pool-frames", type=int, default=0, help="distinct frames resident in HBM (0 = steps*frames_per_step, capped at 1024)")
    ap.add_argument("--ba-window", type=int, default=0,
                    help="BASELINE configs[4]: run one bundle adjustment (HIP kernels + TRF/LSMR driver) per this many frames on "
                         "geometry-consistent synthetic detections, inside the timed region (0 = fixed calib.pkl, configs[1])")
    ap.add_argument("--rank-share", type=int, default=0,
                    help="run rank 0's share of a --stream-frames stream sharded over this many ranks (configs[3]/[4] on one GPU); overrides --steps")
    ap.add_argument("--stream-frames", type=int, default=100000)
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: create a 1-rank process group (RCCL) and execute the packed gather anyway")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the attached configs[2] (bf16) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


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
    """sha256 over the HIP sources: profiles/traffic.json records the value it was collected with (no .git on the GPU box)."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "deepfly3d_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


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


def measure_roofline(engine, dtype, run_steps, nprof):
    """HIP events around every launch of each kernel class (same stream), over `nprof` steps; the dominant kernel
    priced against the roof SURVEY.md 8(d) assigns to this dtype."""
    import ctypes

    from deepfly3d_amd import _native

    lib = _native.load()
    _native.check(lib.df3d_hg_profile(engine.h, 1))
    run_steps(nprof)
    torch.cuda.synchronize()
    per = []
    buf = ctypes.create_string_buffer(128)
    for k in range(lib.df3d_hg_profile_count(engine.h)):
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _native.check(lib.df3d_hg_profile_read(engine.h, k, buf, 128, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n)))
        if n.value:
            per.append({"kernel": buf.value.decode(), "launches": n.value, "avg_us": 1e3 * ms.value / n.value, "total_ms": ms.value,
                        "tflops": fl.value / ms.value / 1e9, "gbs_algorithmic": by.value / ms.value / 1e6})
    per.sort(key=lambda d: -d["total_ms"])
    _native.check(lib.df3d_hg_profile(engine.h, 0))
    dom = per[0]
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        traffic = tj.get(dom["kernel"])
        traffic_src = (tj.get("_meta") or {}).get("kernel_source_sha")
    bound = BOUND[dtype]
    roof = {
        "bound": bound,
        "kernel": dom["kernel"],
        "achieved": dom["tflops"] if bound == "mfma" else dom["gbs_algorithmic"],
        "peak": PEAK_TFLOPS[dtype] if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "algorithmic_bytes_model": "M1 (SURVEY.md 8d): every convolution reads its input and writes its output once",
        "traffic": traffic,  # HBM bytes per launch from rocprofv3 PMC passes ((2 x FETCH_SIZE + WRITE_SIZE) x 1024), profiles/traffic.json
        "traffic_kernel_source_sha": traffic_src,
        "traffic_is_current": (traffic_src == kernel_source_sha()) if traffic_src else None,
        "avg_launch_us": dom["avg_us"],
        "mfma_tflops": dom["tflops"],
        "mfma_frac": dom["tflops"] / PEAK_TFLOPS[dtype],
        "hbm_gbs_algorithmic": dom["gbs_algorithmic"],
        "hbm_frac_algorithmic": dom["gbs_algorithmic"] / PEAK_HBM_GBS,
        "hbm_gbs_pmc": (traffic / (dom["avg_us"] * 1e-6) / 1e9) if traffic else None,
        "kernels": per,
    }
    roof["frac"] = roof["achieved"] / roof["peak"]
    return roof


def main():
    a = parse()
    from deepfly3d_amd import _native
    from deepfly3d_amd import distributed as dd
    from deepfly3d_amd.config import load_calibration
    from deepfly3d_amd.hourglass import HourglassEngine
    from deepfly3d_amd.pipeline import FramePipeline
    from deepfly3d_amd.synthetic import synthetic_state_dict

    rank, world, local_rank = dd.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    _native.require_gpu()
    dev_index = local_rank % torch.cuda.device_count()  # (== local_rank on a node with one GPU per rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    if a.force_collective and world == 1 and not torch.distributed.is_initialized():
        port = int(os.environ.get("MASTER_PORT", "29517"))
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    collective = dd.collective_needed(world, a.force_collective)

    fps_step = a.frames_per_step
    align = a.ba_window if a.ba_window > 0 else 1
    if a.rank_share > 0:
        t0, t1 = dd.shard_range(a.stream_frames, a.rank_share, 0, align)
        total_frames = t1 - t0
        a.steps = -(-total_frames // fps_step)
    else:
        total_frames = a.steps * fps_step
    sd = synthetic_state_dict(0)
    engine = HourglassEngine(sd, dtype=a.dtype, device=dev)
    cal = load_calibration()
    calib = {k: np.stack([cal[c][k] for c in range(7)]) for k in ("R", "tvec", "intr", "distort")}
    pipe = FramePipeline(engine, calib["R"], calib["tvec"], calib["intr"])

    pool = a.pool_frames or min(a.steps * fps_step, 1024)
    pool = max(fps_step, (pool // fps_step) * fps_step)
    gen = torch.Generator(device=dev).manual_seed(rank)
    frames = torch.empty((pool, 7, 256, 512, 3), dtype=torch.float32, device=dev)
    for i in range(0, pool, 64):  # bounded temporary memory
        frames[i : i + 64].uniform_(0.0, 1.0, generator=gen)
    outs = pipe.allocate_outputs(total_frames)

    ba_px, ba_runs, ba_cams = None, [], []
    if a.ba_window > 0:
        # geometry-consistent detections, one set per window: golden-like pose tiled + jitter, projected through
        # the adjusted cameras of the sample set, quantised to the heat-map grid (the random-weight network output is
        # meaningless for BA); prepared before the timed region like the frames
        from deepfly3d_amd.bundle_adjust import bundle_adjust
        from deepfly3d_amd.synthetic import synthetic_ba_window

        g3 = np.load(os.path.join(ROOT, "tests", "golden", "golden_3d.npz"))
        nwin = -(-total_frames // a.ba_window)
        ba_px = [synthetic_ba_window(g3["points3d_wo_procrustes"], g3["R"], g3["tvec"], g3["intr"], min(a.ba_window, total_frames - w * a.ba_window), rank, w)
                 for w in range(nwin)]

    # configs[4]: the re-calibration of a finished window runs on its own HIP stream from a worker thread (its inputs do not
    # depend on the frames still in flight), so its ~150 small kernels and ~40 host synchronisations slot in beside the next
    # batches' hourglass instead of draining the pipeline; every window is joined before the gather, inside the timed region
    ba_pool = ba_stream = None
    ba_futures = []
    if ba_px is not None:
        from concurrent.futures import ThreadPoolExecutor

        ba_pool = ThreadPoolExecutor(max_workers=1)
        ba_stream = torch.cuda.Stream(device=dev)

    ba_ms = []

    def recalibrate(window_px):
        t_ba = time.perf_counter()
        with torch.cuda.device(dev), torch.cuda.stream(ba_stream):
            Rn, tn, info = bundle_adjust(window_px, calib["R"], calib["tvec"], calib["intr"], device=dev, return_info=True)
        ba_ms.append(round(1e3 * (time.perf_counter() - t_ba), 1))   # wall time of the solve (beside the pipeline when threaded)
        if timeline is not None:
            timeline["ba"].append((t_ba, time.perf_counter()))
        return np.concatenate([Rn.reshape(7, 9), tn.reshape(7, 3)], axis=1), info["nfev"]

    timeline = {"enq": [], "ev": [], "ba": []} if os.environ.get("DF3D_BENCH_TIMELINE") else None

    def step(i, pipeline=pipe, record=True):
        if timeline is not None and record:
            timeline["enq"].append(time.perf_counter())
        f0 = i * fps_step
        n = min(fps_step, total_frames - f0)
        lo = f0 % pool
        pipeline.run_batch(frames[lo : lo + n], *outs, f0)
        if timeline is not None and record:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            timeline["ev"].append(ev)
        # a window closes with this batch (the last window of the share may be shorter)
        if ba_px is not None and ((f0 + n) // a.ba_window > f0 // a.ba_window or (f0 + n == total_frames and total_frames % a.ba_window)):
            closed = (f0 + n) // a.ba_window - 1 if (f0 + n) // a.ba_window > f0 // a.ba_window else len(ba_px) - 1
            fut = ba_pool.submit(recalibrate, ba_px[closed])
            if record:
                ba_futures.append(fut)
            else:
                fut.result()

    def join_recalibrations():
        for fut in ba_futures:
            cams, nfev = fut.result()
            ba_cams.append(cams)
            ba_runs.append(nfev)
        del ba_futures[:]

    def gather():
        if not collective:
            return
        cams = None
        if a.ba_window > 0:
            cams = torch.from_numpy(np.stack(ba_cams) if ba_cams else np.zeros((0, 7, 12))).to(dev)
        nf = total_frames * world
        if a.ba_window > 0 and total_frames % a.ba_window and world > 1:
            raise SystemExit("per-GPU frames must be a multiple of --ba-window when N > 1")
        return dd.gather_results(*outs, num_frames=nf, rank=rank, world_size=world, align=align, cameras=cams, force_collective=a.force_collective)

    for w in range(a.warmup):
        step(w % a.steps, record=False)
    if ba_px is not None:
        # the re-calibration has one-time costs of its own (allocator growth on its stream, the LSMR chunk's graph: ~0.7 s per
        # call until buffers and graph settle) and no window closes inside the W warm-up steps: warm it on its worker thread,
        # where its graph cache lives
        for _ in range(2):
            ba_pool.submit(recalibrate, ba_px[0]).result()
        del ba_ms[:]
    if collective and world == 1:
        # a multi-rank run creates its RCCL communicator in the barrier below; the 1-rank group of --force-collective would
        # create it inside the first gather, i.e. inside the timed region (0.9 s when nothing else hides it)
        torch.distributed.all_reduce(torch.zeros(1, device=dev))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t_start = time.perf_counter()
    if timeline is not None:
        timeline["ev0"] = torch.cuda.Event(enable_timing=True)
        timeline["ev0"].record()
    for i in range(a.steps):
        step(i)
    t_marks = [time.perf_counter()]
    join_recalibrations()
    t_marks.append(time.perf_counter())
    gathered = gather()
    t_marks.append(time.perf_counter())
    torch.cuda.synchronize()
    t_marks.append(time.perf_counter())
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t_start
    if timeline is not None:   # development aid: host enqueue time, GPU completion time of every step, the solves' intervals
        ev0 = timeline["ev"][0]
        done = [ev0.elapsed_time(e) for e in timeline["ev"]]
        enq = [1e3 * (x - t_start) for x in timeline["enq"]]
        print("TIMELINE step: enqueue_ms gpu_done_ms(from step 0's end)", file=sys.stderr)
        for i in range(0, len(enq), max(1, len(enq) // 24)):
            print(f"  step {i:3d}: {enq[i]:8.1f} {done[i]:8.1f}", file=sys.stderr)
        print("  step 0 gpu done", round(timeline["ev0"].elapsed_time(ev0), 1), "ms after the start; last step", round(done[-1], 1), "ms after step 0; elapsed", round(1e3 * elapsed, 1), file=sys.stderr)
        print("  host marks (ms): loop end, joined, gather returned, synchronised:", [round(1e3 * (m - t_start), 1) for m in t_marks], file=sys.stderr)
        print("  BA [start, end] ms:", [(round(1e3 * (a0 - t_start)), round(1e3 * (a1 - t_start))) for a0, a1 in timeline["ba"][-len(ba_runs):]], file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    gather_ok = None
    if collective and rank == 0 and world == 1:  # the 1-rank collective must hand back exactly what went in
        gather_ok = all(torch.equal(g, o) for g, o in zip(gathered[:3], outs))

    roof = None
    if not a.no_roofline and rank == 0:
        nprof = min(a.steps, 4)
        saved_ba, ba_px = ba_px, None  # kernel timing only
        roof = measure_roofline(engine, a.dtype, lambda n: [step(i, record=False) for i in range(n)], nprof)
        ba_px = saved_ba

    # configs[2] (bf16 hourglass, same frames and geometry) in the same process, under the same driver clock
    leg = None
    if a.dtype == "f32" and world == 1 and not a.no_bf16_leg and a.rank_share == 0 and a.ba_window == 0:
        e16 = HourglassEngine(sd, dtype="bf16", device=dev)
        p16 = FramePipeline(e16, calib["R"], calib["tvec"], calib["intr"])
        for w in range(max(1, a.warmup)):
            step(w % a.steps, pipeline=p16, record=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i, pipeline=p16, record=False)
        torch.cuda.synchronize()
        el16 = time.perf_counter() - t0
        fl16, by16 = e16.work(fps_step * 7)
        leg = {
            "workload": f"BASELINE configs[2]: {total_frames} frames x 7 views of 256x512x3, 2-stack hourglass bf16 (all convolutions on MFMA, fp32 accumulate), "
                        "arg-max + 38-joint layout + fp64 DLT with fixed calib.pkl",
            "value": total_frames / el16, "unit": "frames/s", "steps": a.steps, "ms_per_step": 1e3 * el16 / a.steps, "dtype": "bf16",
            "hourglass_tflops_end_to_end": fl16 / (el16 / a.steps) / 1e12,
            "hourglass_gbs_algorithmic_end_to_end": by16 / (el16 / a.steps) / 1e9,
            "hbm_frac": by16 / (el16 / a.steps) / 1e9 / PEAK_HBM_GBS,
        }
        if not a.no_roofline:
            leg["roofline"] = measure_roofline(e16, "bf16", lambda n: [step(i, pipeline=p16, record=False) for i in range(n)], min(a.steps, 4))
        del p16, e16

    if rank == 0:
        ms_step = 1e3 * elapsed / a.steps
        fl, by = engine.work(fps_step * 7)
        per_step = total_frames / a.steps / fps_step  # < 1 when the last batch of a rank share is short
        if a.rank_share:
            cfg = 4 if a.ba_window else 3
            workload = (f"BASELINE configs[{cfg}], ONE rank's share on one GPU: rank 0 of {a.rank_share} ranks of a {a.stream_frames}-frame 7-view stream = "
                        f"{total_frames} frames streamed through a {pool}-frame resident pool, 2-stack hourglass {a.dtype}, arg-max + 38-joint layout + fp64 DLT"
                        + (f", bundle-adjustment re-calibration every {a.ba_window} frames" if a.ba_window else "")
                        + (", packed gather executed on a 1-rank RCCL group" if collective else ""))
        elif a.ba_window:
            workload = (f"BASELINE configs[4] (per-GPU share): {total_frames} frames x 7 views of 256x512x3 per GPU, 2-stack hourglass {a.dtype}, "
                        f"arg-max + 38-joint layout + fp64 DLT, bundle-adjustment re-calibration every {a.ba_window} frames")
        else:
            workload = (f"BASELINE configs[{1 if a.dtype == 'f32' else 2}]: {total_frames} frames x 7 views of 256x512x3 per GPU, 2-stack hourglass {a.dtype}, "
                        "arg-max + 38-joint layout + fp64 DLT with fixed calib.pkl")
        line = {
            "metric": "frames/sec (7-view 2D->3D)",
            "value": world * total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic (seeded uniform frames resident in HBM, seeded synthetic hourglass weights, data/calib.pkl cameras)",
            "config": {
                "workload": workload,
                "frames_per_step": fps_step,
                "frames_per_gpu": total_frames,
                "parallelism": f"frame-sharded x{world}, one packed gather" + (" (executed: RCCL, 1 rank)" if collective and world == 1 else ""),
                "collective_executed": bool(collective),
                "collective_backend": torch.distributed.get_backend() if collective else None,
                "gather_roundtrip_exact": gather_ok,
                "bundle_adjust_every_frames": a.ba_window or None,
                "bundle_adjust_runs_rank0": len(ba_runs) or None,
                "bundle_adjust_nfev": ba_runs or None,
                "bundle_adjust_wall_ms": ba_ms[-len(ba_runs):] if ba_runs else None,
                "hourglass_tflops_end_to_end": per_step * fl / (ms_step * 1e-3) / 1e12,
                "hourglass_gbs_algorithmic_end_to_end": per_step * by / (ms_step * 1e-3) / 1e9,
                "hbm_frac": per_step * by / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "hbm_frac_note": "hourglass activation bytes of the fusion model M1 (SURVEY.md 8d) / step time / 8 TB/s",
            },
        }
        if roof is not None:
            line["roofline"] = roof
        if leg is not None:
            line["config2_bf16"] = leg
        if not a.no_cpu_baseline and world == 1 and a.rank_share == 0:
            try:
                line["cpu_baseline"] = cpu_baseline(sd, frames[:16].cpu(), calib, a.cpu_seconds)
            except Exception as e:  # the baseline is reporting only; never hide the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        if world > 1:
            torch.distributed.barrier()   # rank 0 is still measuring the per-kernel table while the others are done: tear down together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
