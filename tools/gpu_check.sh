#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/check; mkdir -p $O
{
  echo "=== $(date) tests"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "pow or ziv" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
  echo "=== $(date) done"
} > $O/check.log 2>&1
cat $O/check.log | cut -c1-800
