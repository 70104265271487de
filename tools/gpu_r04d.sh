#!/bin/bash
# Round 4, call D: where the no-SLP build of the steady rows differs; memory operations at the row end (in-order memory counter); counters
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/diff_libs.py tools/ab/lib_r03.so tools/ab/lib_noslp.so 4 > $OUT/diff_r03_noslp.log 2>&1; tail -20 $OUT/diff_r03_noslp.log
python tools/ab_interleaved.py --libs r03=tools/ab/lib_r03.so,rot0io=tools/ab/lib_rot0io.so,endio=tools/ab/lib_endio.so,endionorot=tools/ab/lib_endionorot.so,endiow4=tools/ab/lib_endiow4.so \
   --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 7 --json $OUT/ab_march_endio.json > $OUT/ab_march_endio.log 2>&1
grep "^\[ab\]" $OUT/ab_march_endio.log | cut -c1-1500
tail -3 $OUT/ab_march_endio.log
cd /tmp; export TMPDIR=/tmp
for L in r03 endionorot; do
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_READ_SECTORS_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  VRGDG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$L.so timeout 180 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_${L}_$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_march.py 16 > $OUT/pmc_${L}_$i.log 2>&1
  echo "pmc $L set $i rc=$?"
done
done
cd $OUT; python - <<'PY'
import csv, glob, collections, json
PX = 16 * 2160 * 3840
rows = collections.OrderedDict()
for f in sorted(glob.glob('pmc_*/**/*counter_collection.csv', recursive=True)):
    lib = f.split('/')[0].split('_')[1]
    seen = collections.Counter()
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        if 'k_chain_march' not in name: continue
        per.setdefault(int(r['Dispatch_Id']), {})
        per[int(r['Dispatch_Id'])][r['Counter_Name']] = per[int(r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for n, (d, c) in enumerate(sorted(per.items())):
        key = (lib, 'uniform' if n < 2 else 'video', n % 2)
        rows.setdefault(key, {}).update(c)
out = []
for (lib, dist, rep), c in rows.items():
    if rep != 1: continue
    out.append({"lib": lib, "data": dist, "per_px": {k: round(v / PX, 4) for k, v in c.items()}})
    print(lib, dist, {k: round(v / PX, 4) for k, v in c.items()})
json.dump(out, open('pmc_march.json', 'w'), indent=1)
PY
