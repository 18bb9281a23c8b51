#!/bin/bash
# cold photon state: store / load cache policy A/B (builds: make EXTRA=-DTHX_COLD_NT_ST|-DTHX_COLD_NT_LD OUTDIR=../lib_<name>)
#   tools/cold_state_ab.sh lib lib_ntst lib_ntld lib_ntboth
mkdir -p gpurun_out/r04
for rep in 1 2; do for L in "$@"; do
  echo "== $L (run $rep)"
  THEIA_AMD_LIB=$GRAFT_REPO_ROOT/theia_amd/$L/libtheia_hip.so timeout -k 10 300 python tools/sweep.py 10000000 config3,config5,config2 2>&1 | grep "kernel ms"
done; done
for L in "$@"; do
  echo "== $L traffic config3"
  THEIA_AMD_LIB=$GRAFT_REPO_ROOT/theia_amd/$L/libtheia_hip.so bash tools/prof_traffic.sh r04/traffic_$L config3 2>&1 | tail -1
done
