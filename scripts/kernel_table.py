"""Print the per-kernel table of a bench line (bench.py's roofline block): python scripts/kernel_table.py <bench.log> [leg]"""
import json
import sys

lines = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(lines[-1])
if len(sys.argv) > 2:
    d = d[sys.argv[2]]
print(f"{d['value']:.1f} frames/s, {d['ms_per_step']:.2f} ms/step, dtype {d['dtype']}")
tot = sum(k["total_ms"] for k in d["roofline"]["kernels"])
for k in d["roofline"]["kernels"]:
    pmc = f"{k['frac_hbm_pmc']:.2f}" if k.get("frac_hbm_pmc") else " -  "
    print(f"  {k['kernel'][:72]:72s} x{k['launches']:3d} {k['avg_us']:8.1f} us  {100 * k['total_ms'] / tot:5.1f} %  {k['tflops']:7.1f} TF  mfma {k['frac_mfma']:.2f}  hbm_min {k['frac_hbm_min']:.2f}  pmc {pmc}")
