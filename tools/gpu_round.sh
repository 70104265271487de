#!/bin/bash
# One gpurun call: GPU parity tests, smoke, per-kernel diagnostics, bench, rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "=== $(date) build"; python -c "import __graft_entry__ as g; g.build(); print('build ok')"
  echo "=== $(date) pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -150
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke
  echo "=== $(date) diag"; timeout 900 python tools/gpu_diag.py --frames 16 --iters 5 --out gpurun_out/diag_${TAG}.json
  echo "=== $(date) bench"; timeout 900 python bench.py --steps 3 --warmup 1 | tee gpurun_out/bench_${TAG}.json
  echo "=== $(date) rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --frames 64 --no-cpu-baseline; cd $GRAFT_REPO_ROOT
  ls -R gpurun_out/prof_${TAG} | head -30
  echo "=== $(date) done"
} > gpurun_out/round_${TAG}.log 2>&1
tail -5 gpurun_out/round_${TAG}.log
