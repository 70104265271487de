#!/bin/bash
# Bench + kernel-trace stats in one call.  bash tools/gpu_bench.sh <tag>
TAG=${1:-b}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) bench chain4"; timeout 900 python bench.py --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}.json
  echo "=== $(date) bench chain3"; timeout 600 python bench.py --steps 4 --warmup 1 --workload chain3_4k --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}_chain3.json
  echo "=== $(date) bench chain4 video"; timeout 600 python bench.py --steps 3 --warmup 1 --dist video --no-cpu-baseline --frames 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}_video.json
  echo "=== $(date) rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1; cd $GRAFT_REPO_ROOT
  tail -2 gpurun_out/prof_${TAG}.log | cut -c1-600
  find gpurun_out/prof_${TAG} -name "*stats*" | head; 
  echo "=== $(date) done"
} > gpurun_out/benchrun_${TAG}.log 2>&1
tail -4 gpurun_out/benchrun_${TAG}.log | cut -c1-400
