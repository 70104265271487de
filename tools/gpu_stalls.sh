#!/bin/bash
# Stall / activity counters of the headline chain's kernels (separate PMC passes, kernel-trace only).  bash tools/gpu_stalls.sh <tag>
TAG=${1:-s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/stalls_${TAG}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do      # (a TA_BUSY_avr / TA_*_STALLED / GRBM_GUI_ACTIVE set aborted inside rocprofv3 on this image: left out)
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py chain4 > $OUT/set$i.log 2>&1
done
cd $OUT; python - <<'PY'
import csv, glob, json, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('set*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if 'k_produce_lab<3, false, false>' in n or 'k_apply_march<20' in n or 'k_tstats' in n:
            rows[n[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in rows.items()}
for k, d in out.items():
    wc = d.get('SQ_WAVE_CYCLES')
    if wc:
        d['share_of_wave_cycles'] = {c: round(d[c] / wc, 4) for c in ('SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', ) if c in d}
json.dump(out, open('summary.json', 'w'), indent=1)
for k, d in out.items():
    print(k); [print('    ', c, v) for c, v in sorted(d.items(), key=lambda kv: str(kv[0]))]
PY
