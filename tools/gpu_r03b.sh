#!/bin/bash
# Round 3, GPU call B: pipelined pieces sweep, flat-march stencil, grain variants.   bash tools/gpu_r03b.sh <tag>
TAG=${1:-b}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest (new + stencil + chains)"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "pipelined or stencil or sharpen or laplacian or sobel or unsharp or fused_chain or bench_geometry or surface or enhancer or reentrant" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  for F in 32 256; do for P in 1 2 4 8 16; do
    echo "=== pieces $P frames $F"; VRGDG_CM_PIECES=$P timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | tail -1
  done; done
  echo "=== $(date) copy ceiling (flat stencil)"; timeout 600 python tools/copy_ceiling.py --out $O/copy_ceiling.json 2>&1 | grep "\[copy\]" | grep -i "stencil\|grain\|nt" | cut -c1-260
  echo "=== $(date) bench"; timeout 900 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-1200
  echo "=== $(date) frames table"; timeout 1200 python tools/frames_table.py --out $O/frames_table.json --frames 8,32,64,256 2>&1 | grep "\[frames\]" | cut -c1-300
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
