#!/bin/bash
# One GPU iteration: full -m gpu suite, smoke, per-pass times of the headline chain, the default bench.   bash tools/gpu_round.sh <tag>
TAG=${1:-x}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/round_${TAG}.log
{
  echo "=== $(date) pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids\|vrgdg-amd"
  echo "=== $(date) pass times chain4 (32 frames)"; timeout 300 python tools/ab_pass_times.py chain4 32 6 2>&1 | tail -1
  echo "=== $(date) pass times chain4fast (32 frames)"; timeout 300 python tools/ab_pass_times.py chain4fast 32 6 2>&1 | tail -1
  echo "=== $(date) bench"; timeout 900 python bench.py 2>gpurun_out/bench_${TAG}.err | tee gpurun_out/bench_${TAG}.json | cut -c1-2500
  echo "=== $(date) done"
} > $O 2>&1
cat $O
