#!/bin/bash
# Round 3, GPU call S: flat-march stencil with the wave index in an SGPR, A/B.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "stencil or sharpen or unsharp or sobel or laplacian" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for rep in 1 2; do for lib in default norfl; do
    echo "=== $(date) kernels $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py kernels 128 8 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py kernels 128 8 2>&1 | tail -1; fi
  done; done
} > $O/round.log 2>&1
cat $O/round.log
