#!/bin/bash
# Round 3, GPU call U: Lab pass with / without the pixel prefetch (both policies).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
{
  for rep in 1 2; do for lib in default nopf; do
    echo "=== $(date) bench colormatch_4k $lib"
    if [ $lib = default ]; then timeout 600 python bench.py --workload colormatch_4k --frames 256 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['passes_ms'], d['fast_variant']['value'], d['fast_variant']['ms_per_step'])"
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 600 python bench.py --workload colormatch_4k --frames 256 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['passes_ms'], d['fast_variant']['value'], d['fast_variant']['ms_per_step'])"; fi
  done; done
} > $O/round.log 2>&1
cat $O/round.log
