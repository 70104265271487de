#!/bin/bash
TAG=${1:-r02b}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "=== $(date) cm parity probe"; timeout 900 python tools/probe_cm_parity.py $TAG 2>&1 | grep -v "^\[cm\] e2e.*fast\|amdgpu.ids" | cut -c1-420 | tail -80
  echo "=== $(date) pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | tail -40
  echo "=== $(date) done"
} > gpurun_out/round_${TAG}.log 2>&1
tail -130 gpurun_out/round_${TAG}.log
