#!/bin/bash
# Generic A/B over tools/ab/lib_<name>.so builds:  bash tools/gpu_ab_generic.sh "<workloads>" <frames> <reps> default a b ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
WS=$1; F=$2; R=$3; shift 3
{
for W in $WS; do for r in $(seq 1 $R); do for L in "$@"; do
  if [ $L = default ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so; fi
  timeout 300 python tools/ab_pass_times.py $W $F 6 2>&1 | grep -v amdgpu.ids | tail -1
done; done; done
} > gpurun_out/ab_generic.log 2>&1
cat gpurun_out/ab_generic.log
