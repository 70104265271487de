#!/bin/bash
# Round 4, call B: the GPU suite on the tree with the uint8 enhancer kernel, its timing and traffic, L2 sector counters of the gather probes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
python tools/bench_u8_enhancer.py --frames 8 --rounds 7 --json $OUT/u8_enhancer.json > $OUT/u8_enhancer.log 2>&1; tail -6 $OUT/u8_enhancer.log
python tools/bench_u8_enhancer.py --frames 64 --rounds 5 --json $OUT/u8_enhancer_64.json > $OUT/u8_enhancer_64.log 2>&1; tail -3 $OUT/u8_enhancer_64.log
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU"; do
  i=$((i+1))
  PROBE_PMC=1 timeout 180 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/u8pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_u8_enhancer.py --frames 8 > $OUT/u8pmc$i.log 2>&1
  echo "u8 pmc $SET rc=$?"
done
i=0
for SET in "TCC_READ_SECTORS_sum TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum" "TCC_TAG_STALL_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  PROBE_PMC=1 timeout 180 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/secpmc$i -o p -- python $GRAFT_REPO_ROOT/tools/probe_gather.py --frames 4 --rounds 1 --modes 0,9,10,2,12 > $OUT/secpmc$i.log 2>&1
  echo "sector pmc set $i rc=$?"
done
cd $OUT; python - <<'PY'
import csv, glob, collections, json
def collect(pattern, match, PX):
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(pattern, recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r.get('Kernel_Name', '')
            if not any(m in name for m in match): continue
            key = (f.split('/')[0], int(r['Dispatch_Id']), name.split('(')[0][-44:])
            rows.setdefault(key, {})
            rows[key][r['Counter_Name']] = rows[key].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    out = []
    for (p, d, n), c in sorted(rows.items()):
        out.append({"pass": p, "dispatch": d, "kernel": n, "per_px": {k: round(v / PX, 4) for k, v in c.items()}})
        print(p, d, n, {k: round(v / PX, 4) for k, v in c.items()})
    return out
a = collect('u8pmc*/**/*counter_collection.csv', ('k_sharpen_grain', 'k_u8', 'k_f32', 'u8bgr', 'u8_to', 'to_u8'), 8 * 2160 * 3840)
b = collect('secpmc*/**/*counter_collection.csv', ('k_dbg_lut_fetch',), 4 * 2160 * 3840)
json.dump({"u8": a, "sectors": b}, open('pmc_summary.json', 'w'), indent=1)
PY
