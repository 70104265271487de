#!/bin/bash
# Round 3, GPU call E: full suite, Markstein statistics A/B, apply-march rows A/B, bench.   bash tools/gpu_r03e.sh <tag>
TAG=${1:-e}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids\|vrgdg-amd"
  for r in 1 2; do for L in default nomark rows120; do for F in 4 32 256; do
    if [ $L = default ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so; fi
    echo "=== A/B $L chain4 $F"; timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | tail -1
  done; done; done
  unset VRGDG_HIP_LIB
  echo "=== $(date) bench"; timeout 900 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  echo "=== $(date) frames table"; timeout 1200 python tools/frames_table.py --out $O/frames_table.json 2>&1 | grep "\[frames\]" | cut -c1-300
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
