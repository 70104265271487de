#!/bin/bash
# Round-2 first GPU call: parity tests, colour-match parity probe, VALU probe with effective clock, PMC stall evidence, benches.
TAG=${1:-r02a}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/round_${TAG}.log
{
  echo "=== $(date) pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | tail -80
  echo "=== $(date) cm parity probe"; timeout 900 python tools/probe_cm_parity.py $TAG 2>&1 | tail -120
  echo "=== $(date) valu probe (plain)"; timeout 600 python tools/probe_valu.py 20 2>&1 | tail -40
  cp gpurun_out/valu_long.json gpurun_out/valu_long_${TAG}.json
  echo "=== $(date) valu probe under rocprofv3 GRBM_GUI_ACTIVE"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/valu_pmc_${TAG} -o p -- python $GRAFT_REPO_ROOT/tools/probe_valu.py 20 > $GRAFT_REPO_ROOT/gpurun_out/valu_pmc_${TAG}.log 2>&1)
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5
  for W in chain4_4k chain3_4k; do
    echo "=== $(date) bench $W"; timeout 600 python bench.py --steps 3 --warmup 1 --workload $W --no-cpu-baseline 2>gpurun_out/bench_${TAG}_${W}.err | tee gpurun_out/bench_${TAG}_${W}.json
  done
  echo "=== $(date) bench chain4 fast"; VRGDG_CM_MATH=fast timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>>gpurun_out/bench_${TAG}_fast.err | tee gpurun_out/bench_${TAG}_chain4_fast.json
  echo "=== $(date) PMC stall sets"
  for K in chain3 chain4 chain4fast; do
    for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
               "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
               "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
      N=$(echo $SET | cut -d' ' -f1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}/${K}_${N} -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py $K > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${K}_${N}.log 2>&1)
    done
  done
  python tools/summarize_pmc.py gpurun_out/pmc_${TAG} gpurun_out/valu_pmc_${TAG} > gpurun_out/pmc_${TAG}_summary.txt 2>&1
  tail -100 gpurun_out/pmc_${TAG}_summary.txt
  echo "=== $(date) done"
} > $LOG 2>&1
tail -150 $LOG
