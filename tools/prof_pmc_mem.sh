#!/bin/bash
# PMC passes over the memory path of the tracing kernel (vector cache, texture addresser, L2, instruction latencies):
#   tools/prof_pmc_mem.sh <outdir-under-gpurun_out> [bench args...]
# One counter set per pass, kernel-trace only (MI355X_MICROARCH.md: separate --pmc passes). (A pass over the TA_* counters
# -- TA_TA_BUSY_sum, TA_ADDR_STALLED_BY_TC_CYCLES_sum ... -- never returned on this pool in round 3: left out.)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
  "TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/pass$i.log | tr '\n' ' ')"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "Staged" not in k and "traceKernel" not in k: continue
        m = re.search(r"Variant<([^>]*)>", k)
        a = [x.strip() for x in m.group(1).split(",")] if m else []
        dbg = len(a) > 2 and a[2] == "true"
        agg[("staged" if "Staged" in k else "trace") + ("_debug" if dbg else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            fh.write(f"{c:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/summary.txt").read())
PY
