#!/bin/bash
# Round 3, GPU call Y: Lab-only pass 1 in wave form (no LDS exchange, no barriers) against the workgroup form.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "headline or produce or fused_chain or bench_geometry or colour_match or chain or shared or philox" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for rep in 1 2; do for lib in default pwave0 pw4 pw16; do
    echo "=== $(date) A/B $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1; fi
  done; done
} > $O/round.log 2>&1
cat $O/round.log
