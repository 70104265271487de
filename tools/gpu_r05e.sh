#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
exec < /dev/null
{
  echo "=== $(date) pytest"; timeout 600 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider 2>&1 | tail -45
  echo "=== $(date) diag lazy graph"; timeout 120 python tools/diag_lazy_graph.py --frames 16 --reps 14 2>&1 | grep "^\[diag\]"
  echo "=== $(date) done"
} > $O/run.log 2>&1
tail -80 $O/run.log | cut -c1-300
