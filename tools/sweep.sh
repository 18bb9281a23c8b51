#!/bin/bash
# batch-size sweep of config 2 for a set of env settings
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' ')
  for n in 1000000 4000000 10000000; do
    echo -n "[$spec] "; env $envs timeout -k 10 200 python tools/quick_bench.py $n config2 2>&1 | grep "config2_sphere " | cut -c1-150
  done
done
