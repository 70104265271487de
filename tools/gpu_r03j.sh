#!/bin/bash
# Round 3, GPU call J: instruction-cache / issue counters of the stage kernel vs the stand-alone passes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
{
  for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
    N=$(echo $SET | cut -d' ' -f1)
    for S in 0 32; do
      (cd /tmp && VRGDG_CM_STAGE_FRAMES=$S timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${N}_s$S -o p -- python $GRAFT_REPO_ROOT/tools/ab_pass_times.py chain4 96 1 > $GRAFT_REPO_ROOT/$O/pmc_${N}_s$S.log 2>&1)
    done
  done
  python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r03j"
for d in sorted(glob.glob(O + "/pmc_*_s*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "vrg::" not in k: continue
            short = k.split("(")[0].replace("void vrg::", "")[:40]
            acc[short][r["Counter_Name"]] += float(r["Counter_Value"]); 
    print("==", os.path.basename(d))
    for k, v in acc.items():
        print("   ", k, {c: f"{x:.4g}" for c, x in v.items()})
PY
} > $O/round.log 2>&1
cat $O/round.log | cut -c1-400
