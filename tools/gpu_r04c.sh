#!/bin/bash
# Round 4, call C: the steady-row fast path of the march kernel -- parity tests of every fused-chain test with two builds, then the
# interleaved A/B of the variants
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "chain or march or fused or bench_geometry or headline" > $OUT/pytest_default.log 2>&1; echo "pytest default rc=$?"; tail -3 $OUT/pytest_default.log
VRGDG_HIP_LIB=$PWD/tools/ab/lib_noslp.so timeout 1500 python -m pytest tests -m gpu -x -q -k "chain or march or fused or bench_geometry or headline" > $OUT/pytest_noslp.log 2>&1; echo "pytest noslp rc=$?"; tail -3 $OUT/pytest_noslp.log
python tools/ab_interleaved.py --libs r03=tools/ab/lib_r03.so,slp=tools/ab/lib_slp.so,noslp=tools/ab/lib_noslp.so,norot=tools/ab/lib_norot.so,w2=tools/ab/lib_w2.so,nofast=tools/ab/lib_nofast.so,nofin=tools/ab/lib_nofin.so \
   --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 7 --json $OUT/ab_march_variants.json > $OUT/ab_march_variants.log 2>&1
grep "^\[ab\]" $OUT/ab_march_variants.log | cut -c1-1800
tail -3 $OUT/ab_march_variants.log
