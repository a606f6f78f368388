"""What the two innermost hourglass levels (8 x 16 and 4 x 8 pixels per view) cost per step, from a rocprofv3 kernel trace:
    python scripts/inner_levels.py <dir with *kernel_trace.csv> [steps]
Groups the dispatches by (kernel, grid) and, per step, adds up the launches whose grid is at most one 8 x 16 tile per view (the
8 x 16 level's ring launches, the 4 x 8 level's per-convolution launches, their pools / upsample-adds), both as the sum of their
durations and as the wall-clock SPAN from the first such launch of a stack's inner part to the end of the last (gaps included):
the span is what one fused per-view kernel could at most replace; the per-tile latency chain it would keep is printed beside it."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name") or r.get("Name")
            g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            wgs = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 256)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, g // max(1, wgs)))
rows.sort()
hg = [r for r in rows if "hgk::" in r[2] or "hgk" in r[2]]
steps = max(1, sum(1 for r in hg if "stem_lp_kernel" in r[2] or "stem_kernel" in r[2]))   # one stem launch per forward (warm-up and settle steps included)
print(f"{len(rows)} dispatches, {len(hg)} of the hourglass, {steps} forwards in the trace")
groups = defaultdict(list)
for s, e, n, wg in hg:
    short = n.split("(")[0].replace("hgk::", "").replace("void ", "")
    groups[(short, wg)].append(e - s)
views = 896
print(f"{'kernel':70s} {'workgroups':>10s} {'launches':>8s} {'avg us':>9s} {'total ms / step':>15s}")
inner_keys = set()
for (n, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    per_step = sum(v) / steps * 1e-6
    mark = ""
    if ("conv_mfma" in n) or (wg == views and "bottleneck_ring" in n):   # one 8 x 16 tile per view, or the 4 x 8 level's per-convolution launches
        inner_keys.add((n, wg)); mark = "  <- inner level"
    print(f"{n[:70]:70s} {wg:10d} {len(v):8d} {sum(v) / len(v) * 1e-3:9.1f} {per_step:15.3f}{mark}")
inner = [(s, e, n) for s, e, n, wg in hg if (n.split('(')[0].replace('hgk::', '').replace('void ', ''), wg) in inner_keys]
tot = sum(e - s for s, e, _ in inner) / steps * 1e-6
# spans: consecutive inner launches separated by less than 100 us belong to one inner section
spans, cur = [], None
for s, e, n in inner:
    if cur is None or s - cur[1] > 100_000:
        if cur: spans.append(cur)
        cur = [s, e, 1]
    else:
        cur[1] = max(cur[1], e); cur[2] += 1
if cur: spans.append(cur)
span_ms = sum(b - a for a, b, _ in spans) / steps * 1e-6
print(f"inner levels per step: {len(inner) // steps} launches, sum of durations {tot:.3f} ms, wall-clock span of their sections {span_ms:.3f} ms "
      f"({len(spans) // steps} sections per step, gaps {span_ms - tot:.3f} ms)")
step_ms = (hg[-1][1] - hg[0][0]) / steps * 1e-6
print(f"hourglass span per step {step_ms:.2f} ms: the inner sections are {100 * span_ms / step_ms:.1f} % of it")
