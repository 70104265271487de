#!/bin/bash
# wave-state counters of chain 3 at forced occupancies (1 / 2 / 3 workgroups of four waves per CU): who issues, who waits
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp VRGDG_SELFCHECK=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
exec < /dev/null
for occ in 1 2 3; do
  export VRGDG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_r6_occ$occ.so
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/p_${occ}_$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_march.py 16 > /dev/null 2>&1 )
  done
done
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06k")
px = 16 * 2160 * 3840
res = {}
for occ in (1, 2, 3):
    for d in glob.glob(os.path.join(O, f"p_{occ}_*")):
        dur = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_chain_march<3, true, 4>" in r["Kernel_Name"]:
                    per.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 0)))
        for c, vs in per.items():
            # launches alternate uniform, uniform, video, video: the 2nd of each pair
            for name, idx in (("uniform", 1), ("video", 3)):
                if len(vs) > idx:
                    res.setdefault(f"{occ} workgroup(s) per CU, {name}", {})[c] = round(vs[idx][0] / px, 4)
                    res[f"{occ} workgroup(s) per CU, {name}"]["duration_ms"] = round(vs[idx][1] / 1e6, 3)
for k, d in sorted(res.items()):
    w = d.get("SQ_WAVE_CYCLES")
    if w:
        d["issuing_share_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
        d["waiting_at_waitcnt_share"] = round(d.get("SQ_WAIT_INST_ANY", 0) / w, 3)
        g = d.get("GRBM_GUI_ACTIVE")
        if g:
            d["resident_waves_per_simd"] = round(w * 4 / (g / 8 * 1024), 2)
            d["gpix_s_at_measured_clock"] = round(px / d["duration_ms"] / 1e6, 1) if d.get("duration_ms") else None
    print("[occ]", k, json.dumps(d))
json.dump(res, open(os.path.join(O, "pmc_march_occupancy.json"), "w"), indent=1)
PY
rm -rf $O/p_*
