#!/bin/bash
# Round 4, call E: quad-cooperative LDS-DMA gathers in the steady rows (parity + A/B on uniform and video-like frames), the DPP-fold isolation
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
VRGDG_HIP_LIB=$PWD/tools/ab/lib_quad.so timeout 1500 python -m pytest tests -m gpu -x -q -k "chain or march or fused or bench_geometry or headline" > $OUT/pytest_quad.log 2>&1; echo "pytest quad rc=$?"; tail -3 $OUT/pytest_quad.log
python tools/ab_interleaved.py --libs r03=tools/ab/lib_r03.so,endio=tools/ab/lib_endio.so,quad=tools/ab/lib_quad.so,quadnorot=tools/ab/lib_quadnorot.so,noslpunfold=tools/ab/lib_noslpunfold.so,noslp2=tools/ab/lib_noslp2.so \
   --cases chain3,chain3_video --frames 64 --rounds 7 --json $OUT/ab_march_quad.json > $OUT/ab_march_quad.log 2>&1
grep "^\[ab\]" $OUT/ab_march_quad.log | cut -c1-1700
tail -3 $OUT/ab_march_quad.log
cd /tmp; export TMPDIR=/tmp
for L in quad; do
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_READ_SECTORS_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
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
