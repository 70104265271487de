#!/bin/bash
# Round 3, GPU call C: full suite, bench, frames table, host-fed nodes, rocprofv3 kernel stats of the bench.   bash tools/gpu_r03c.sh <tag>
TAG=${1:-c}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  echo "=== $(date) smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids\|vrgdg-amd"
  echo "=== $(date) pass times chain4 (1, 4, 8, 32 frames)"; for F in 4 8 32; do timeout 300 python tools/ab_pass_times.py chain4 $F 6 2>&1 | tail -1; done
  echo "=== $(date) bench"; timeout 900 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-600
  echo "=== $(date) frames table"; timeout 1200 python tools/frames_table.py --out $O/frames_table.json 2>&1 | grep "\[frames\]" | cut -c1-300
  echo "=== $(date) host fed"; timeout 600 python tools/host_fed.py --out $O/host_fed_nodes.json 2>&1 | grep "\[host\]" | cut -c1-250
  echo "=== $(date) host fed, 2 lanes on one GPU"; VRGDG_DEVICES=0,0 timeout 600 python tools/host_fed.py --out $O/host_fed_nodes_2lanes.json 2>&1 | grep "\[host\]" | cut -c1-250
  echo "=== $(date) rocprofv3 --kernel-trace --stats bench"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_chain4 -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fast-variant --no-live-traffic --no-host-fed > $GRAFT_REPO_ROOT/$O/prof_chain4.log 2>&1)
  head -12 $O/prof_chain4/trace_kernel_stats.csv | cut -c1-220
  echo "=== $(date) copy ceiling"; timeout 600 python tools/copy_ceiling.py --out $O/copy_ceiling.json 2>&1 | grep "\[copy\]" | cut -c1-260
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
