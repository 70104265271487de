#!/bin/bash
# round 6, batch B: is the march VALU-issue bound (ballast), what do its instruction classes cost, what clock does the chip run under each kernel,
# and why does k_tstats_frame slow down from 256 to 512 frames
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp VRGDG_SELFCHECK=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $O
exec < /dev/null
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 600 python tools/ab_interleaved.py --libs base=$L,bal128=tools/ab/lib_r6_bal128.so,bal256=tools/ab/lib_r6_bal256.so,bal512=tools/ab/lib_r6_bal512.so \
   --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 5 --json $O/ab_ballast.json 2>&1 | grep -v amdgpu.ids > $O/ab_ballast.log
timeout 300 python tools/probe_valu_classes.py --json $O/valu_classes.json 2>&1 | grep -v amdgpu.ids > $O/valu_classes.log
timeout 300 python tools/sample_clocks.py --json $O/clocks.json 2>&1 | grep -v amdgpu.ids > $O/clocks.log
timeout 300 python tools/probe_tstats_sizes.py --json $O/tstats_sizes.json 2>&1 | grep -v amdgpu.ids > $O/tstats_sizes.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clk -o p -- python $GRAFT_REPO_ROOT/tools/prof_clock.py > $O/prof_clock.log 2>&1 )
python tools/summarize_pmc.py $O/clk > $O/clock_per_kernel.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clkp -o p -- python $GRAFT_REPO_ROOT/tools/probe_valu_classes.py --ms 5 --waves 2 > $O/prof_classes.log 2>&1 )
python - <<'PY' > $O/clock_probe_classes.txt 2>&1
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06b", "clkp")
dur = {}
for f in glob.glob(os.path.join(O, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur and "valu_rate" in r["Kernel_Name"]:
            n, d = dur[r["Dispatch_Id"]]
            if d > 2e6:
                print(n[:60], "dur_us", d / 1e3, "clock_GHz", round(float(r["Counter_Value"]) / 8 / d, 3))
PY
i=0
for SET in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUSY_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/ts_$i -o p -- python $GRAFT_REPO_ROOT/tools/probe_tstats_sizes.py --only 256,512 > $O/ts_$i.log 2>&1 )
done
python - <<'PY' > $O/tstats_pmc.txt 2>&1
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06b")
for d in sorted(glob.glob(os.path.join(O, "ts_*"))):
    if not os.path.isdir(d): continue
    grid = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid[r["Dispatch_Id"]] = (r.get("Grid_Size") or r.get("Grid_Size_X"), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_tstats_frame" in r["Kernel_Name"]:
                g = grid.get(r["Dispatch_Id"], ("?", 0))
                acc[(g[0], r["Counter_Name"])].append((float(r["Counter_Value"]), g[1]))
    for (g, c), vs in sorted(acc.items()):
        print(os.path.basename(d), "grid", g, c, "avg", sum(v for v, _ in vs) / len(vs), "dur_us", sum(t for _, t in vs) / len(vs) / 1e3, "n", len(vs))
PY
rm -rf $O/clk $O/clkp $O/ts_[0-9]
tail -5 $O/ab_ballast.log; cat $O/clock_per_kernel.txt | grep -A3 "==" | head -120; cat $O/clock_probe_classes.txt | head; cat $O/tstats_sizes.log; cat $O/tstats_pmc.txt | head -60; tail -12 $O/clocks.log; tail -50 $O/valu_classes.log
