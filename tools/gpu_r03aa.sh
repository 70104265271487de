#!/bin/bash
# Round 3, GPU call AD: four-workgroup (split) statistics form, depth 2 branch-free vs depth 1 plain, at 48 / 64 / 96 frames.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03ad; mkdir -p $O
{
  for lib in default split1 default split1; do
    echo "=== $(date) frames $lib"
    if [ $lib = default ]; then timeout 600 python tools/frames_table.py --out $O/ft_$lib.json --frames 48,64,96 2>&1 | grep "\[frames\]" | cut -c1-230
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 600 python tools/frames_table.py --out $O/ft_$lib.json --frames 48,64,96 2>&1 | grep "\[frames\]" | cut -c1-230; fi
  done
} > $O/round.log 2>&1
cat $O/round.log
