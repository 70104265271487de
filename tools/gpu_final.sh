#!/bin/bash
# End-of-iteration evidence run: full GPU tests, smoke, bench (N=1 plain + N=1 under torchrun/RCCL), kernel-trace stats, PMC traffic.
TAG=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "=== $(date) pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) bench"; timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}.json
  echo "=== $(date) bench chain3"; timeout 600 python bench.py --workload chain3_4k --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}_chain3.json
  echo "=== $(date) bench 1080p grain+lut"; timeout 600 python bench.py --workload grain_lut_1080p --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}_1080p.json
  echo "=== $(date) bench colormatch"; timeout 600 python bench.py --workload colormatch_4k --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_${TAG}_cm.json
  echo "=== $(date) torchrun x1 (RCCL path)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --frames 32 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/bench_${TAG}_torchrun1.json
  echo "=== $(date) rocprof stats"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1; cd $GRAFT_REPO_ROOT
  head -6 gpurun_out/prof_${TAG}/trace_kernel_stats.csv | cut -c1-200
  echo "=== $(date) traffic"; bash tools/gpu_traffic.sh ${TAG} 2>&1 | tail -4
  echo "=== $(date) issue (VALU instr/px)"; bash tools/gpu_issue.sh ${TAG} 2>&1 | tail -3 | cut -c1-300
  echo "=== $(date) diag (kernels, VALU probe, host-fed)"; timeout 900 python tools/gpu_diag.py --frames 16 --iters 5 --valu --host --out gpurun_out/diag_${TAG}.json 2>&1 | grep -v amdgpu.ids | grep "diag\]" > gpurun_out/diag_${TAG}.log; tail -2 gpurun_out/diag_${TAG}.log | cut -c1-300
  echo "=== $(date) done"
} > gpurun_out/final_${TAG}.log 2>&1
tail -6 gpurun_out/final_${TAG}.log | cut -c1-300
