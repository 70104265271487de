#!/bin/bash
# Round 3, GPU call AE: LDS-tile kernel with branch-free preloaded tile + halo pixels (light chains), A/B on the kernel table.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03ae; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "uint8 or u8 or tile or route or chain or stencil or adjust or variant" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for lib in default notp default notp; do
    echo "=== $(date) diag $lib"
    if [ $lib = default ]; then timeout 600 python tools/gpu_diag.py --frames 16 --iters 5 --out $O/diag_$lib.json 2>&1 | grep "diag\]" | grep -E "u8|tile|variant 1|uint8" | cut -c1-200
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 600 python tools/gpu_diag.py --frames 16 --iters 5 --out $O/diag_$lib.json 2>&1 | grep "diag\]" | grep -E "u8|tile|variant 1|uint8" | cut -c1-200; fi
  done
} > $O/round.log 2>&1
cat $O/round.log
