"""Summarise rocprofv3 outputs into profiles/: python scripts/summarize_profile.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]

* copies <stats_dir>/*_kernel_stats.csv (top rows, kernel names shortened) to profiles/<tag>_kernel_stats.csv
* if PMC dirs are given: per-kernel HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes
  (FETCH_SIZE on gfx950 reports half of a wide coalesced read stream: MI355X_MICROARCH.md "HBM"; separate --pmc passes)
  -> profiles/<tag>_traffic.csv and profiles/traffic.json (read by bench.py for roofline.traffic)
"""
import collections, csv, glob, json, os, re, sys

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
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:25]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    if len(sys.argv) >= 5:
        # launches are counted PER PASS: bench.py's settle loop runs a variable number of untimed steps, so the FETCH and the WRITE pass need not see
        # the same number of launches (round 5 divided both sums by the FETCH pass's count: every WRITE_SIZE of profiles/r05_f16_traffic.csv is
        # 10/9 too large -- profiles/r06_ring_traffic.txt)
        agg = collections.defaultdict(lambda: {"n": 0, "nw": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
        for d in sys.argv[3:5]:
            for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
                a = agg[short(r["Kernel_Name"])]
                a[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "FETCH_SIZE":
                    a["n"] += 1
                elif r["Counter_Name"] == "WRITE_SIZE":
                    a["nw"] += 1
        traffic = {}
        tj = os.path.join(out_dir, "traffic.json")
        if os.path.exists(tj):
            traffic = json.load(open(tj))
        cls = collections.defaultdict(lambda: [0, 0.0])
        with open(os.path.join(out_dir, f"{tag}_traffic.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
            for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
                if not a["n"] or not a["nw"] or "at::native" in k:
                    continue
                b = (2 * a["FETCH_SIZE"] / a["n"] + a["WRITE_SIZE"] / a["nw"]) * 1024 * a["n"]   # (per launch x the FETCH pass's launches)
                w.writerow([k, a["n"], a["FETCH_SIZE"] / a["n"], a["WRITE_SIZE"] / a["nw"], b / a["n"]])
                m = re.match(r"hgk::(\w+(?:<[^(]*>)?)\(", k)
                if m:  # key = kernel instantiation exactly as bench.py's roofline.kernel names it
                    cls[m.group(1)][0] += a["n"]
                    cls[m.group(1)][1] += b
        for name, (n, b) in cls.items():
            traffic[name] = b / n
        # provenance: the kernel sources the counters were collected with (bench.py recomputes the same hash at run time
        # and reports `traffic_is_current`), plus the commit of this checkout when the summary is made
        import subprocess

        sys.path.insert(0, ROOT)
        from bench import kernel_source_sha

        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            head = None
        traffic["_meta"] = {"kernel_source_sha": kernel_source_sha(), "git_head_when_summarised": head, "tag": tag,
                            "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch, separate --pmc passes"}
        json.dump(traffic, open(tj, "w"), indent=1, sort_keys=True)
        print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
