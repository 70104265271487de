#!/bin/bash
# Round 3, GPU call R: pass 2 (apply march) without conditional blocks, rows requested ahead of the colour transfer.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "headline or apply or fused_chain or bench_geometry or colour_match or chain or staged" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for rep in 1 2; do for lib in default amgen amlate pipe0; do
    echo "=== $(date) A/B $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1; fi
  done; done
  for lib in default amgen; do
    echo "=== $(date) A/B fast $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py chain4fast 64 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py chain4fast 64 6 2>&1 | tail -1; fi
  done
} > $O/round.log 2>&1
cat $O/round.log
