#!/bin/bash
# One round's profile set on the GPU box (every rocprofv3 run under its own time limit; kernel-trace only next to --pmc):
#   tools/prof_round.sh <tag>      -> gpurun_out/<tag>/{bench.json, kernel_stats.csv, traffic.json, pmc_config2.txt, pmc_config3.txt}
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== bench under rocprofv3 --kernel-trace --stats"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --no-flow > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || echo "rocprof bench run failed"
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
echo "== bench alone (the driver's command)"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
echo "== HBM traffic"
for w in config2 config3 config5 bigmesh; do
  timeout -k 10 300 tools/prof_traffic.sh $TAG/traffic_$w $w > $OUT/traffic_$w.json 2> $OUT/traffic_$w.err || echo "traffic $w failed"
  tail -1 $OUT/traffic_$w.json
done
echo "== instruction mix / lane utilisation"
timeout -k 10 400 tools/prof_pmc_quick.sh $TAG/pmc_config2 > $OUT/pmc_config2.txt 2>&1 || echo "pmc config2 failed"
timeout -k 10 400 tools/prof_pmc_quick.sh $TAG/pmc_config3 --workload config3 > $OUT/pmc_config3.txt 2>&1 || echo "pmc config3 failed"
timeout -k 10 400 tools/prof_pmc_quick.sh $TAG/pmc_config5 --workload config5 > $OUT/pmc_config5.txt 2>&1 || echo "pmc config5 failed"
echo done
