#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
exec < /dev/null
{
  echo "=== $(date) pytest"; timeout 600 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider 2>&1 | tail -14
  echo "=== $(date) smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
  echo "=== $(date) done"
} > $O/run.log 2>&1
tail -30 $O/run.log | cut -c1-300
