#!/bin/bash
# Round 3, GPU call K: rounds of loads in flight in the statistics kernels (A/B), bench JSON check.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
{
  for r in 1 2; do for L in default tsd4 tsd8 tsw4; do for F in 1 4 32 64 256; do
    if [ $L = default ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so; fi
    echo "=== $L frames $F"; timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | tail -1 | cut -c1-260
  done; done; done
  unset VRGDG_HIP_LIB
  echo "=== $(date) bench"; timeout 900 python bench.py --no-host-fed --no-cpu-baseline 2>$O/bench.err | tee $O/bench.json | cut -c1-300
} > $O/round.log 2>&1
cat $O/round.log
