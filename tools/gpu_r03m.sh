#!/bin/bash
# Round 3, GPU call M: division by the per-frame sigma (sweep + A/B), 512-entry Ziv log table (accuracy per interval, exhaustive powers, A/B).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
{
  echo "=== $(date) pytest (pow / division / colour match)"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "ziv or sigma or pow or colormatch or color_match or colour or device_math or headline" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  echo "=== $(date) ziv log accuracy per interval"; timeout 600 python tools/ziv_log_accuracy.py 2>&1 | tail -2; cp gpurun_out/ziv_log_accuracy.json $O/
  echo "=== $(date) div sigma sweep"; timeout 600 python tools/div_sigma_sweep.py --out $O/div_sigma_sweep.json 2>&1 | tail -1
  for lib in default nodivs bits7 default nodivs bits7; do
    echo "=== $(date) A/B $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1; fi
  done
} > $O/round.log 2>&1
cat $O/round.log
