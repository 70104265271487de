#!/bin/bash
# Round 5: L1 / L2 counters of the shipped march (nt pixel loads / stores) next to round 4's, uniform and video-like pixels (tools/prof_march.py),
# one rocprofv3 --pmc pass per counter set and build; per pixel in $O/pmc_march_r04_r05.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp VRGDG_SELFCHECK=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
exec < /dev/null
for L in r05 r04m; do
  if [ $L = r05 ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$L.so; fi
  i=0
  for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_SECTORS_sum" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/${L}_$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_march.py 16 > /dev/null 2>&1 )
  done
done
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05h")
px = 16 * 2160 * 3840
res = {}
for d in sorted(glob.glob(os.path.join(O, "r0*_*"))):
    lib = os.path.basename(d).split("_")[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k_chain_march<3, true, 4>" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for c, vs in per.items():          # launches in order: uniform, uniform, video, video
            if len(vs) == 4:
                res.setdefault(lib, {}).setdefault("uniform", {})[c] = round(vs[1] / px, 4)
                res.setdefault(lib, {}).setdefault("video", {})[c] = round(vs[3] / px, 4)
json.dump(res, open(os.path.join(O, "pmc_march_r04_r05.json"), "w"), indent=1)
for lib, d in res.items():
    for dist, c in d.items():
        print("[pmc]", lib, dist, json.dumps(c))
PY
rm -rf $O/r05_* $O/r04m_*
