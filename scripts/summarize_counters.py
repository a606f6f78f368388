"""Per-kernel hardware-counter summary -> profiles/<tag>_counters.csv

    python scripts/summarize_counters.py <tag> <pmc_dir> [<pmc_dir> ...]

Every <pmc_dir> holds one rocprofv3 --pmc pass (*counter_collection.csv).  Per kernel instantiation: launches and the
per-launch average of every counter, plus
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)   (GRBM_GUI_ACTIVE is summed
                     over the 8 XCDs; MfmaUtil is rocprofv3's own derived metric, kept beside it)
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / average launch duration (from the same pass)
  lds_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac        = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(name):
    """rocprofv3 leaves names with a _Float16 template argument mangled (its demangler does not know the DF16_ code, nor does
    GNU c++filt): demangle with DF16_ spelled as the older half-precision code Dh, and name the type as the source does."""
    if not name.startswith("_Z"):
        return name
    import shutil
    import subprocess

    tool = shutil.which("c++filt")
    if not tool:
        return name
    out = subprocess.run([tool, name.replace("DF16_", "Dh")], capture_output=True, text=True).stdout.strip()
    if not out or out.startswith("_Z"):
        return name
    out = re.sub(r"<half\b", "<_Float16", out)
    return out if out.startswith("void ") else "void " + out


def short(name):
    name = demangle(name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:hgk|jpg)::(\w+(?:<[^(]*>)?)\(", name)
    return m.group(1) if m else name[:80]


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if "at::native" in k or "rocclr" in k:
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]] += 1
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
                    dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    dur[k][1] += 1
    names = sorted({c for k in agg for c in agg[k]})
    out = os.path.join(ROOT, "profiles", f"{tag}_counters.csv")
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches"] + [f"{c}_per_launch" for c in names] + ["mfma_busy_frac", "clock_ghz", "lds_conflict_frac", "wait_frac"])
        for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0.0)):
            a = {c: agg[k][c] / launches[k][c] for c in agg[k]}
            g = a.get("GRBM_GUI_ACTIVE")
            mf = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * g / 8.0) if g and "SQ_VALU_MFMA_BUSY_CYCLES" in a else ""
            clk = g / 8.0 / (dur[k][0] / dur[k][1]) if g and dur[k][1] else ""
            lc = a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else ""
            wf = a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else ""
            w.writerow([k, max(launches[k].values())] + [a.get(c, "") for c in names] + [mf, clk, lc, wf])
    print(open(out).read())


if __name__ == "__main__":
    main()
