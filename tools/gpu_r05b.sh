#!/bin/bash
# Round 5, second GPU call: changed paths (lazy download, cell-major twin, inference mode), A/B of the twin and of a march without SLP on
# 25^3 and 33^3 cubes, the statistics-overlap experiment with its kernel trace, host-fed rates.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
P=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
{
  echo "=== $(date) pytest subset"; timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "lazy or inference or adjacent or toolchain or lut or march or chain or pageable or surface" 2>&1 | tail -8
  echo "=== $(date) fuzz"; timeout 600 python tools/fuzz_march.py --cases 300 2>&1 | grep -v amdgpu.ids | tail -4
  echo "=== $(date) A/B 25^3"; timeout 600 python tools/ab_interleaved.py --libs new=$P,nocm=tools/ab/lib_nocm.so,noslp=tools/ab/lib_noslp.so --cases chain3,chain3_video,grain_lut --lut AMD_WarmFilm_25.cube --frames 64 --rounds 6 --json $O/ab_cellmajor_25.json 2>&1 | grep "^\[ab\]"
  echo "=== $(date) A/B 33^3"; timeout 600 python tools/ab_interleaved.py --libs new=$P,noslp=tools/ab/lib_noslp.so --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 6 --json $O/ab_noslp_33.json 2>&1 | grep "^\[ab\]"
  echo "=== $(date) overlap"; timeout 600 python tools/exp_tstats_overlap.py --frames 256 --rounds 5 --json $O/tstats_overlap.json 2>&1 | grep "^\[ovl\]"
  for S in "seq" "ovl 4"; do T=$(echo $S | tr ' ' '_'); ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$T -o t -- python $GRAFT_REPO_ROOT/tools/exp_tstats_overlap.py --frames 64 --once "$S" > /dev/null 2>&1 ); f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); echo "--- kernel stats, schedule $S (64 frames, 2 steps)"; grep "vrg" $f | cut -d, -f1-4 | cut -c1-160 | head -8; done
  echo "=== $(date) host fed"; timeout 600 python tools/host_fed.py --frames 16 --out $O/host_fed_nodes.json 2>&1 | grep "^\[host\]"
  echo "=== $(date) done"
} > $O/run.log 2>&1
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
tail -120 $O/run.log
