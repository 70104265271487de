#!/bin/bash
# Wave-state / LDS / L1 counters of the two gather-free forms of the march (tools/prof_skeleton.py); per pixel in gpurun_out/r05k/pmc_skeleton.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp VRGDG_SELFCHECK=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O
exec < /dev/null
i=0
for SET in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/p_$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_skeleton.py > /dev/null 2>&1 )
done
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05k")
px = 16 * 2160 * 3840
res = {}
for f in glob.glob(os.path.join(O, "p_*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "grain_unsharp <1,true,4>" if "k_chain_march<1, true, 4>" in n else ("grain_lut17lds_unsharp <3,true,12>" if "k_chain_march<3, true, 12>" in n else None)
        if key:
            per.setdefault((key, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (key, c), vs in per.items():
        res.setdefault(key, {})[c] = round(vs[-1] / px, 4)
json.dump(res, open(os.path.join(O, "pmc_skeleton.json"), "w"), indent=1)
for k, d in res.items():
    print("[pmc]", k, json.dumps(d))
PY
rm -rf $O/p_*
