"""Everything profiles/<tag>_* needs from one collection run (gpurun_out/<tag>, gpurun_out/<tag>_counters), in one go:

    python scripts/summarize_round.py r03

* per dtype (f32, bf16, f16, f32s): kernel stats + FETCH/WRITE traffic (scripts/summarize_profile.py; rewrites profiles/traffic.json
  from scratch, stamped with the kernel-source hash) and the SQ counter table (scripts/summarize_counters.py)
* the bench lines of the same run (gpurun_out/<tag>/<dtype>_bench.log -> profiles/<tag>_<dtype>_bench.json) and, when present,
  the full rank-share lines (rankshare_*.log -> profiles/<tag>_rankshare_*.json)
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    traffic_only = "--traffic-only" in sys.argv[2:]   # collect_profiles.sh, ON the GPU box: profiles/traffic.json first, the kept bench lines after it
    src = os.path.join(ROOT, "gpurun_out", tag)
    cnt = os.path.join(ROOT, "gpurun_out", tag + "_counters")
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        os.remove(tj)
    for dt in ("f32", "bf16", "f16", "f32s"):
        if not os.path.isdir(os.path.join(src, f"{dt}_stats")):
            continue
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), f"{tag}_{dt}", os.path.join(src, f"{dt}_stats"),
                        os.path.join(src, f"{dt}_fetch"), os.path.join(src, f"{dt}_write")], check=True, stdout=subprocess.DEVNULL)
        groups = sorted(glob.glob(os.path.join(cnt, f"{dt}_g[0-9]")))
        if groups:
            subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_counters.py"), f"{tag}_{dt}", *groups], check=True, stdout=subprocess.DEVNULL)
    if traffic_only:
        return
    for log in sorted(glob.glob(os.path.join(src, "*_bench.log")) + glob.glob(os.path.join(src, "rankshare_*.log"))):
        rows = [ln for ln in open(log) if ln.startswith("{")]
        if not rows:
            print("no JSON line in", log)
            continue
        d = json.loads(rows[-1])
        full = os.path.join(src, os.path.basename(log)[: -len(".log")] + "_full.json")   # the whole record (per-kernel tables) beside the short line
        if os.path.exists(full):
            d = json.load(open(full))
        base = os.path.basename(log)[: -len(".log")]
        name = f"{tag}_{base}.json" if base.endswith("_bench") else f"{tag}_{base.split('_cfg')[0]}.json"
        with open(os.path.join(ROOT, "profiles", name), "w") as f:
            json.dump(d, f, indent=1)
        legs = {k: round(d[k]["value"], 1) for k in ("config1_f32_split", "config2_bf16", "config2_f16", "config4_share") if k in d and "value" in d[k]}
        print(f"{name}: {d['value']:.1f} frames/s {legs if legs else ''}")
    meta = json.load(open(tj))["_meta"]
    sys.path.insert(0, ROOT)
    from bench import kernel_source_sha

    print("traffic.json:", meta["kernel_source_sha"], "current sources:", kernel_source_sha())


if __name__ == "__main__":
    main()
