#!/bin/bash
# Round 4, call A: fetch-pattern probes (timing with interleaved rounds + one PMC counter set per pass), MIOpen solver log of the
# depthwise 3x3 conv2d.   bash tools/gpu_r04a.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/probe_gather.py --frames 16 --rounds 7 --json $OUT/probe_gather_timing.json > $OUT/probe_gather_timing.log 2>&1
tail -30 $OUT/probe_gather_timing.log
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD" \
           "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  PROBE_PMC=1 timeout 180 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/probe_gather.py --frames 4 --rounds 1 > $OUT/pmc$i.log 2>&1
  echo "pmc set $i rc=$?"
done
cd $OUT; python - <<'PY'
import csv, glob, collections, json
PX = 4 * 2160 * 3840
rows = collections.OrderedDict()
for f in sorted(glob.glob('pmc*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        if 'k_dbg_lut_fetch' not in name: continue
        key = (int(r['Dispatch_Id']), name.split('(')[0][-40:])
        rows.setdefault(key, {})[r['Counter_Name']] = rows.get(key, {}).get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
out = []
for (d, n), c in sorted(rows.items()):
    out.append({"dispatch": d, "kernel": n, "per_px": {k: round(v / PX, 4) for k, v in c.items()}})
    print(d, n, {k: round(v / PX, 3) for k, v in c.items()})
json.dump(out, open('probe_gather_pmc.json', 'w'), indent=1)
PY
# ---- which MIOpen solver runs the reference's use_gpu laplacian / sobel (F.conv2d, groups = 3, 3x3, padding 1)
cd $GRAFT_REPO_ROOT
MIOPEN_ENABLE_LOGGING=1 MIOPEN_ENABLE_LOGGING_CMD=1 MIOPEN_LOG_LEVEL=6 timeout 300 python - > $OUT/miopen_conv_log.txt 2>&1 <<'PY'
import torch, torch.nn.functional as F
dev = torch.device("cuda", 0)
for shape in ((1, 3, 64, 64), (2, 3, 1080, 1920), (1, 3, 2160, 3840)):
    x = torch.rand(shape, device=dev)
    k = torch.tensor([[0., -1., 0.], [-1., 4., -1.], [0., -1., 0.]], device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    print("### conv2d", shape, flush=True)
    y = F.conv2d(x, k, padding=1, groups=3)
    torch.cuda.synchronize()
    print("### done", float(y.sum()), flush=True)
PY
grep -n -i "solver\|FindSol\|Chosen\|algo\|### \|MIOpenDriver\|kernel_name\|KernelName" $OUT/miopen_conv_log.txt | head -120 > $OUT/miopen_conv_solver_lines.txt
wc -l $OUT/miopen_conv_log.txt; head -c 6000 $OUT/miopen_conv_solver_lines.txt
