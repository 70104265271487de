#!/bin/bash
# Alternate library builds (tools/ab/lib_<name>.so) over several processes.
#   bash tools/ab_libs.sh "chain4 chain3 kernels" 3 a b [c ...]
WS=${1:-chain4}; R=${2:-3}; shift 2; LIBS=${@:-a b}
for W in $WS; do
  for r in $(seq 1 $R); do
    for L in $LIBS; do
      VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so timeout 300 python tools/ab_pass_times.py $W 32 8 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
