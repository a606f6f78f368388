#!/bin/bash
# Same-box A/B of library variants on the 16-bit ring bottleneck: per-kernel time (bench.py's HIP-event table) AND HBM-side traffic
# (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes) per variant:
#   LIBS="scratch/variants/libdf3d_hip_A.so ..." [ENVS="DF3D_PK=0"] [DT=f16] [KERN=bottleneck_ring] bash scripts/ab_traffic.sh
# "default" (the in-tree library) always runs first; an entry of the form VAR=VALUE in LIBS runs the default library with that environment.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/ab
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
DT=${DT:-f16}
for lib in default ${LIBS:-}; do
  unset DF3D_LIB; envs=""
  tag=$(basename $lib .so | sed 's/libdf3d_hip_//')
  case $lib in
    default) ;;
XX
    *) export DF3D_LIB=$R/$lib;;
  esac
  env $envs python $R/bench.py --dtype $DT --steps 3 --warmup 1 --full --no-cpu-baseline --no-legs > $OUT/bench_$tag.log 2>&1
  rm -rf $OUT/f_$tag $OUT/w_$tag
  env $envs rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$tag -o pmc -- python $R/bench.py --dtype $DT --steps 1 --warmup 1 --full --no-cpu-baseline --no-roofline --no-legs > /dev/null 2>&1
  env $envs rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w_$tag -o pmc -- python $R/bench.py --dtype $DT --steps 1 --warmup 1 --full --no-cpu-baseline --no-roofline --no-legs > /dev/null 2>&1
  find $OUT/f_$tag $OUT/w_$tag -name "*kernel_trace.csv" -delete
  python - <<PY
import csv, glob, json, collections, re
tag, kern = "$tag", "${KERN:-bottleneck_ring}"
l = [x for x in open("$OUT/bench_$tag.log") if x.startswith("{")]
if not l:
    print(tag, "BENCH FAILED:", open("$OUT/bench_$tag.log").read()[-600:])
else:
    d = json.loads(l[-1])
    print("%-14s %.2f ms/step %.0f frames/s" % (tag, d["ms_per_step"], d["value"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for key, dirn in (("FETCH_SIZE", "f_"), ("WRITE_SIZE", "w_")):
        files = glob.glob("$OUT/" + dirn + tag + "/**/*counter_collection.csv", recursive=True)
        if not files: continue
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != key: continue
            name = r["Kernel_Name"]
            m = re.search(r"(bottleneck_ring_kernel|head_kernel|stem_lp_kernel|bottleneck_l1_kernel)I(\w+?)E", name)
            short = name[8:70]
            a = agg[short]
            if key == "FETCH_SIZE": a[0] += 1; a[1] += float(r["Counter_Value"])
            else: a[2] += float(r["Counter_Value"])
    times = {k["kernel"]: k for k in d["roofline"]["kernels"]}
    for k in d["roofline"]["kernels"]:
        if kern in k["kernel"]:
            print("    %-58s n=%-3d %8.1f us  min %.2f GB" % (k["kernel"], k["launches"], k["avg_us"], k["bytes_min"] / 1e9))
    for short, (n, f, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if "ring" in short and n:
            print("    pmc %-60s n=%-3d fetch(x2) %.3f GB  write %.3f GB  total %.3f GB" % (short, n, 2 * f * 1024 / n / 1e9, w * 1024 / n / 1e9, (2 * f + w) * 1024 / n / 1e9))
PY
done
