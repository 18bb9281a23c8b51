#!/bin/bash
# SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_VALU / SQ_BUSY_CYCLES of the tracing kernel for one library build:
#   tools/prof_pmc_mfma.sh <outdir-under-gpurun_out> [bench args...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pass -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pass.log 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "Staged" in r.get("Kernel_Name", "") and "Lb1ELb1ELb0ELb0" in r["Kernel_Name"] or ("Staged" in r.get("Kernel_Name", "") and "true, true, false, false" in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()): print(f"{c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
