#!/bin/bash
# ISA of one staged kernel variant (development aid): tools/isa_dump.sh <pol> <mis> <stack variant 0..3> [debug 0/1] [extra -D flags]
# writes /tmp/thx_<pol><mis>_v<variant>[d].s
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
POL=$1; MIS=$2; VAR=$3; DBG=${4:-0}; shift 4 || shift $#
SUF=""; if [ "$DBG" = 1 ]; then SUF=d; fi
OUT=/tmp/thx_${POL}${MIS}_v${VAR}${SUF}.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-vector-combine -mllvm -amdgpu-sdwa-peephole=0 \
  -DTHX_KIND=1 -DTHX_POL=$POL -DTHX_MIS=$MIS "$@" --cuda-device-only -S $ROOT/theia_amd/csrc/trace_variant.hip -o /tmp/thx_all.s 2>&1 | grep -E "error" || true
B=Lb0ELb0E; if [ "$DBG" = 1 ]; then B=Lb1ELb1E; fi
NAME="_ZN3thx17sceneStagedKernelINS_7VariantILb${POL}ELb${MIS}E${B}Li${VAR}EEEEEvNS_10LaunchArgsE"
L0=$(grep -n "^${NAME}:" /tmp/thx_all.s | cut -d: -f1)
L1=$(awk -v s=$L0 'NR>s && /^\.Lfunc_end/ {print NR; exit}' /tmp/thx_all.s)
sed -n "${L0},$((L1+80))p" /tmp/thx_all.s > $OUT
echo "$OUT: $(wc -l < $OUT) lines; $(grep -m1 next_free_vgpr $OUT | tr -d '\t'); $(grep -m1 private_segment_fixed_size $OUT | tr -d '\t')"
