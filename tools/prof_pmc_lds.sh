#!/bin/bash
# LDS counters of the tracing kernel for one library build: tools/prof_pmc_lds.sh <outdir-under-gpurun_out> [bench args...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/pass -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pass.log 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "Staged" in k and ("false, false" in k or "Lb0ELb0E" in k):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()): print(f"{c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
