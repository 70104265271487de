#!/bin/bash
# Evidence run of a round: full GPU tests, smoke, benches (all BASELINE configs, both pixel distributions), self-launch refusal,
# RCCL path with one rank, rocprofv3 kernel stats of the benches, PMC traffic / instruction counts / stall sets, kernel table.
#   bash tools/gpu_evidence.sh r02
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/ev_${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E " passed| failed| error" | tail -3
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) bench (default = driver's call)"; timeout 900 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  for W in chain3_4k grain_lut_1080p colormatch_4k; do
    echo "=== $(date) bench $W"; timeout 600 python bench.py --workload $W --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_$W.json | cut -c1-300
  done
  for W in chain4_4k chain3_4k grain_lut_1080p; do
    echo "=== $(date) bench $W video"; timeout 600 python bench.py --workload $W --dist video --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_${W}_video.json | cut -c1-300
  done
  echo "=== $(date) bench --gpus 2 on this 1-GPU box must refuse"; python bench.py --gpus 2 > $O/bench_gpus2.out 2>&1; echo "exit code $?" | tee -a $O/bench_gpus2.out; cat $O/bench_gpus2.out
  echo "=== $(date) torchrun x1 (RCCL path, one rank)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --frames 32 --no-cpu-baseline 2>>$O/bench.err | tail -1 | tee $O/bench_torchrun1.json | cut -c1-300
  for W in chain4_4k chain3_4k; do
    echo "=== $(date) rocprofv3 --kernel-trace --stats bench $W"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$W -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-fast-variant > $GRAFT_REPO_ROOT/$O/prof_$W.log 2>&1)
    head -8 $O/prof_$W/trace_kernel_stats.csv | cut -c1-220
  done
  echo "=== $(date) traffic"; bash tools/gpu_traffic.sh ${TAG} 2>&1 | tail -6
  echo "=== $(date) issue (VALU instr/px)"; bash tools/gpu_issue.sh ${TAG} 2>&1 | tail -4 | cut -c1-300
  echo "=== $(date) PMC stall sets"
  for K in chain3 chain4 chain4fast; do
    for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
               "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
               "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
      N=$(echo $SET | cut -d' ' -f1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/${K}_${N} -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py $K > $GRAFT_REPO_ROOT/$O/pmc_${K}_${N}.log 2>&1)
    done
  done
  python tools/summarize_pmc.py $O/pmc > $O/pmc_summary.txt 2>&1; grep -c "==" $O/pmc_summary.txt
  echo "=== $(date) diag (kernel table)"; timeout 900 python tools/gpu_diag.py --frames 16 --iters 5 --out $O/diag.json 2>&1 | grep "diag\]" > $O/diag.log; tail -2 $O/diag.log | cut -c1-300
  echo "=== $(date) done"
} > $O/evidence.log 2>&1
tail -12 $O/evidence.log | cut -c1-300
