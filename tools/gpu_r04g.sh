#!/bin/bash
# Round 4, call G: small-batch statistics -- one accumulator per lane, prefetch depths, against round 3's half-block form
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "statistics or stats or abi or colour_match" > $OUT/pytest_stats.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_stats.log
python tools/bench_stats.py --libs rows=tools/ab/lib_tsrows.so,d8=tools/ab/lib_tsd8.so,d16=tools/ab/lib_tsd16.so,d32=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so,d48=tools/ab/lib_tsd48.so --rounds 7 --json $OUT/bench_stats.json > $OUT/bench_stats.log 2>&1
grep "\[stats\]" $OUT/bench_stats.log
tail -3 $OUT/bench_stats.log
