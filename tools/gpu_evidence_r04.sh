#!/bin/bash
# Evidence run of round 4: full GPU tests, smoke, benches (all BASELINE configs, both pixel distributions), rocprofv3 kernel stats, PMC traffic /
# instruction counts (with the transcendental and 64-bit integer classes), kernel table.   bash tools/gpu_evidence_r04.sh r04
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/ev_${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E " passed| failed| error|^FAILED" | tail -6
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) bench (default = driver's call)"; timeout 900 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  for W in chain3_4k grain_lut_1080p colormatch_4k; do
    echo "=== $(date) bench $W"; timeout 600 python bench.py --workload $W --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_$W.json | cut -c1-300
  done
  for W in chain4_4k chain3_4k grain_lut_1080p; do
    echo "=== $(date) bench $W video"; timeout 600 python bench.py --workload $W --dist video --no-cpu-baseline --no-host-fed 2>>$O/bench.err | tee $O/bench_${W}_video.json | cut -c1-300
  done
  echo "=== $(date) bench --gpus 2 on this 1-GPU box must refuse"; python bench.py --gpus 2 > $O/bench_gpus2.out 2>&1; echo "exit code $?" | tee -a $O/bench_gpus2.out; tail -2 $O/bench_gpus2.out
  echo "=== $(date) torchrun x1 (RCCL path, one rank)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --frames 32 --no-cpu-baseline --no-host-fed 2>>$O/bench.err | tail -1 | tee $O/bench_torchrun1.json | cut -c1-300
  for W in chain4_4k chain3_4k; do
    echo "=== $(date) rocprofv3 --kernel-trace --stats bench $W"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$W -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-fast-variant --no-live-traffic --no-host-fed --no-verify > $GRAFT_REPO_ROOT/$O/prof_$W.log 2>&1)
    head -8 $O/prof_$W/trace_kernel_stats.csv | cut -c1-220
  done
  echo "=== $(date) traffic"; bash tools/gpu_traffic.sh ${TAG} 2>&1 | tail -6
  echo "=== $(date) issue (VALU instr/px)"; bash tools/gpu_issue.sh ${TAG} 2>&1 | tail -16 | cut -c1-400
  echo "=== $(date) frames table"; timeout 600 python tools/frames_table.py 2>&1 | grep "frames\]" | cut -c1-330; cp gpurun_out/frames_table.json $O/frames_table.json 2>/dev/null
  echo "=== $(date) host-fed nodes (16 x 4K, median of 3 interleaved rounds)"; timeout 900 python tools/host_fed.py --frames 16 --out $O/host_fed_nodes.json 2>&1 | grep "host\]" | cut -c1-300
  echo "=== $(date) uint8 enhancer loop body"; timeout 300 python tools/bench_u8_enhancer.py --frames 8 --rounds 7 --json $O/u8_enhancer.json 2>&1 | grep "u8\]" | cut -c1-300
  echo "=== $(date) march: round 3's kernel against this tree's, interleaved"; timeout 600 python tools/ab_interleaved.py --libs r03=tools/ab/lib_r03.so,r04=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so --cases chain3,chain3_video,grain_sharpen,chain4 --frames 64 --rounds 7 --json $O/ab_r03_r04.json 2>&1 | grep "^\[ab\]" | cut -c1-900
  echo "=== $(date) diag (kernel table)"; timeout 900 python tools/gpu_diag.py --frames 16 --iters 5 --out $O/diag.json 2>&1 | grep "diag\]" > $O/diag.log; tail -3 $O/diag.log | cut -c1-300
  echo "=== $(date) done"
} > $O/evidence.log 2>&1
tail -60 $O/evidence.log | cut -c1-400
