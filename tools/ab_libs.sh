#!/bin/bash
# Alternate two library builds (tools/ab/lib_a.so, lib_b.so) over several processes.  bash tools/ab_libs.sh [chain4|chain3] [rounds]
W=${1:-chain4}; R=${2:-4}
for r in $(seq 1 $R); do
  for L in a b; do
    VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so timeout 300 python tools/ab_pass_times.py $W 32 8 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
