#!/bin/bash
# Round 3, GPU call W: the wave that synthesises a block's edge normals rotates over the four SIMDs, A/B; frames table.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "grain or headline or produce or fused_chain or seeded" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for rep in 1 2; do for lib in default norot; do
    echo "=== $(date) kernels $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py kernels 128 8 2>&1 | tail -1; timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py kernels 128 8 2>&1 | tail -1; VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1; fi
  done; done
  echo "=== $(date) frames table"; timeout 1200 python tools/frames_table.py --out $O/frames_table.json 2>&1 | grep "\[frames\]" | cut -c1-300
} > $O/round.log 2>&1
cat $O/round.log
