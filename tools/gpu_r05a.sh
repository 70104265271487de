#!/bin/bash
# Round 5, first GPU call: the whole GPU suite (new: inference-mode nodes, toolchain self-check, bench legs), fetch-pattern probes on a
# 25^3 cube (record form against cell-major), the default bench line with its `configs` legs, the PMC summary bench.py falls back to.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25
  echo "=== $(date) probes 25^3"; timeout 300 python tools/probe_gather.py --lut AMD_WarmFilm_25.cube --modes 0,12,4,19,9 --rounds 5 --json $O/probe_gather_25.json 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) probes 33^3"; timeout 300 python tools/probe_gather.py --modes 0,12,4,19 --rounds 5 --json $O/probe_gather_33.json 2>&1 | grep -v amdgpu.ids
  echo "=== $(date) bench"; ( time timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; tail -c 600 $O/bench.err
  echo "=== $(date) pmc"; timeout 600 python tools/collect_bench_pmc.py $O/pmc_bench_kernels.json 2>&1 | grep -v amdgpu.ids | tail -12
  echo "=== $(date) done"
} > $O/run.log 2>&1
tail -60 $O/run.log
