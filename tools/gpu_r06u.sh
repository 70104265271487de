#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
exec < /dev/null
timeout 900 python tools/probe_numa.py --out gpurun_out/r06u/numa_host_fed.json 2>&1 | grep "^\[numa\]" | cut -c1-900
