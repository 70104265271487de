#!/bin/bash
# Round 4, call F: full suite on the new default build (quad gathers in the steady rows, one-accumulator-per-lane statistics), ablations,
# the store-hazard confirmation, frames table
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
python tools/diff_libs.py tools/ab/lib_r03.so tools/ab/lib_noslp3.so 4 > $OUT/diff_r03_noslp3_with_store_padding.log 2>&1; head -3 $OUT/diff_r03_noslp3_with_store_padding.log
python tools/ab_interleaved.py --libs r03=tools/ab/lib_r03.so,new=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so,vgpr=tools/ab/lib_vgpr.so,abl1=tools/ab/lib_abl1.so,abl2=tools/ab/lib_abl2.so,abl3=tools/ab/lib_abl3.so \
   --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 7 --json $OUT/ab_march_final_and_ablations.json > $OUT/ab_march_final.log 2>&1
grep "^\[ab\]" $OUT/ab_march_final.log | cut -c1-1700
python tools/frames_table.py > $OUT/frames_table.log 2>&1; tail -12 $OUT/frames_table.log | cut -c1-400
cp gpurun_out/frames_table.json $OUT/ 2>/dev/null
