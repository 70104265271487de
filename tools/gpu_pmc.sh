#!/bin/bash
# PMC collection (own runs, kernel-trace only -- never combined with sys/hip traces).  bash tools/gpu_pmc.sh <tag>
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for K in lut chain3 chain3_v1 grainsharp; do
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
             "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum"; do
    N=$(echo $SET | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${K}_${N} -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py $K > $OUT/${K}_${N}.log 2>&1
  done
done
cd $OUT; python - <<'PY'
import csv, glob, os, collections
rows = collections.defaultdict(dict)
for f in glob.glob('*/**/*counter_collection.csv', recursive=True):
    k = f.split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name','')[:60]
        if 'vrg' not in name: continue
        key = (k.split('_')[0] if not k.startswith('chain3_v1') else 'chain3_v1', name)
        c = r['Counter_Name']; v = float(r['Counter_Value'])
        d = rows[key]
        d.setdefault(c, []).append(v)
with open('summary.txt','w') as out:
    for key, d in sorted(rows.items()):
        out.write(f"== {key}\n")
        for c, vs in sorted(d.items()):
            out.write(f"   {c:40s} avg {sum(vs)/len(vs):16.1f}  n={len(vs)}\n")
print(open('summary.txt').read()[:6000])
PY
