#!/bin/bash
# Round 3, GPU call A: full -m gpu suite, smoke, copy ceiling, per-pass times, default bench, frames-per-step table.   bash tools/gpu_r03a.sh <tag>
TAG=${1:-a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids\|vrgdg-amd"
  echo "=== $(date) copy ceiling"; timeout 600 python tools/copy_ceiling.py --out $O/copy_ceiling.json 2>&1 | grep "\[copy\]" | cut -c1-260
  echo "=== $(date) pass times chain4 (32 frames)"; timeout 300 python tools/ab_pass_times.py chain4 32 6 2>&1 | tail -1
  echo "=== $(date) pass times chain4 (256 frames)"; timeout 300 python tools/ab_pass_times.py chain4 256 4 2>&1 | tail -1
  echo "=== $(date) pass times chain3 (32 frames)"; timeout 300 python tools/ab_pass_times.py chain3 32 6 2>&1 | tail -1
  echo "=== $(date) kernels (16 frames)"; timeout 300 python tools/ab_pass_times.py kernels 16 6 2>&1 | tail -1
  for r in 1 2; do for L in default nodivt; do
    if [ $L = default ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so; fi
    echo "=== A/B $L chain4 32"; timeout 300 python tools/ab_pass_times.py chain4 32 6 2>&1 | tail -1
  done; done
  export VRGDG_HIP_LIB=$PWD/tools/ab/lib_gnt.so; echo "=== A/B grain nt kernels"; timeout 300 python tools/ab_pass_times.py kernels 16 6 2>&1 | tail -1; unset VRGDG_HIP_LIB
  echo "=== $(date) bench"; timeout 900 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-3000
  echo "=== $(date) frames table"; timeout 1200 python tools/frames_table.py --out $O/frames_table.json 2>&1 | grep "\[frames\]" | cut -c1-300
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
