#!/bin/bash
# A/B of library builds, configs 2 / 3 / 5 at N photons, one process per build and config, builds interleaved (ABAB):
#   tools/ab3.sh <n_photons> lib_dir ...
N=$1; shift
for rep in 1 2; do
for dir in "$@"; do
  for c in config2 config3 config5; do
    echo -n "$dir rep$rep: "
    env THEIA_AMD_LIB=$GRAFT_REPO_ROOT/theia_amd/$dir/libtheia_hip.so timeout -k 10 200 python tools/quick_bench.py $N $c 2>&1 | grep -o "config[0-9]_[a-z_]* *batch= *[0-9]* kernel= *[0-9.]* ms"
  done
done
done
