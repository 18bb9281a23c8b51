#!/bin/bash
# Instruction-cache behaviour of the tracing kernel (the staged kernel is 80 - 100 KB of code, the instruction cache 64 KB per two CUs):
#   tools/prof_pmc_icache.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "Staged" not in k: continue
        m = re.search(r"Variant<([^>]*)>", k)
        a = [x.strip() for x in m.group(1).split(",")] if m else []
        agg["staged_debug" if len(a) > 2 and a[2] == "true" else "staged"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"{c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
