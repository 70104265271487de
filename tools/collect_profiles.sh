#!/bin/bash
# Copy the summaries of an evidence run (tools/gpu_evidence.sh <tag>, merged back under gpurun_out/) into profiles/ under round names.
#   bash tools/collect_profiles.sh r02b r02
TAG=${1:?evidence tag}; R=${2:-r02}
E=gpurun_out/ev_${TAG}
cp $E/evidence.log profiles/${R}_evidence_run.log
cp $E/bench.json profiles/${R}_bench.json
for W in chain3_4k grain_lut_1080p colormatch_4k; do cp $E/bench_$W.json profiles/${R}_bench_$W.json; done
for W in chain4_4k chain3_4k grain_lut_1080p; do cp $E/bench_${W}_video.json profiles/${R}_bench_${W}_video.json; done
cp $E/bench_gpus2.out profiles/${R}_bench_gpus2_refused_on_1gpu_box.txt
cp $E/bench_torchrun1.json profiles/${R}_bench_torchrun1.json
cp $E/prof_chain4_4k/trace_kernel_stats.csv profiles/${R}_bench_chain4_rocprofv3_kernel_stats.csv
cp $E/prof_chain3_4k/trace_kernel_stats.csv profiles/${R}_bench_chain3_rocprofv3_kernel_stats.csv
cp gpurun_out/traffic_${TAG}/traffic.json profiles/${R}_pmc_traffic_fetch_write.json
cp gpurun_out/issue_${TAG}/summary.json profiles/${R}_pmc_valu_instr_per_px.json
[ -f $E/pmc_summary.txt ] && cp $E/pmc_summary.txt profiles/${R}_pmc_stall_summary_chain3_chain4_chain4fast.txt
cp $E/diag.json profiles/${R}_diag_kernels.json
cp gpurun_out/cm_test_measured.json profiles/${R}_cm_test_measured.json
for f in frames_table host_fed_nodes u8_enhancer ab_r03_r04; do [ -f $E/$f.json ] && cp $E/$f.json profiles/${R}_$f.json; done
ls -la profiles | grep ${R}_ | wc -l
