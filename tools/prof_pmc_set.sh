#!/bin/bash
# One PMC pass over the tracing kernel with a counter set given on the command line (kernel-trace only next to --pmc):
#   tools/prof_pmc_set.sh <outdir-under-gpurun_out> "<COUNTER ...>" [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; SET=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pass.log 2>&1 || echo "pass failed: $(tail -2 $OUT/pass.log | tr '\n' ' ')"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "Staged" not in k: continue
        agg["debug" if "true, true, " in k.split("Variant<")[1][12:] else "production"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"{c:36s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
