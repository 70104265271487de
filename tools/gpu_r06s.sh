#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06s; mkdir -p $O
exec < /dev/null
timeout 600 python tools/collect_bench_pmc.py $O/pmc_bench_kernels.json 2>&1 | grep "^\[pmc\]" | cut -c1-300
