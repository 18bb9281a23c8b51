#!/bin/bash
# HBM traffic of the tracing kernel per launch: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md),
#   tools/prof_traffic.sh <outdir-under-gpurun_out> <workload>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; W=$2
mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --workload $W > $OUT/$c.log 2>&1 || echo "$c failed"
done
python3 - "$OUT" "$W" <<'PY'
import csv, glob, sys, collections, json
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        prod = ("Staged" in k or "traceKernel" in k) and ("false, false" in k)
        if prod: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
fetch = sum(agg["FETCH_SIZE"]) / max(len(agg["FETCH_SIZE"]), 1)   # KB
write = sum(agg["WRITE_SIZE"]) / max(len(agg["WRITE_SIZE"]), 1)
print(json.dumps({w: {"FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "launches": len(agg["FETCH_SIZE"])}}))
PY
