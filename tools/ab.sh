#!/bin/bash
# A/B of library builds on the GPU: tools/ab.sh <n_photons> lib_dir[:ENV=VAL,...] ...
N=$1; shift
for spec in "$@"; do
  dir=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  echo "=== $spec"
  env THEIA_AMD_LIB=$GRAFT_REPO_ROOT/theia_amd/$dir/libtheia_hip.so $envs timeout -k 10 200 python tools/quick_bench.py $N 2>&1 | grep -E "config2_sphere |config3_cube_torus_pol|config1_volume " 
done
