#!/bin/bash
# round 6, batch C: the whole GPU suite after the deferred-graph / debug-library changes, the host-fed node rates, a kernel trace of the four-node graph
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O
exec < /dev/null
timeout 600 python -m pytest tests/test_deferred_graph.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_deferred.log
cat $O/pytest_deferred.log | tail -30
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -60 > $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 600 python tools/host_fed.py --frames 16 --out $O/host_fed_nodes.json 2>&1 | grep -v amdgpu.ids > $O/host_fed.log
cat $O/host_fed.log
