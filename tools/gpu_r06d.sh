#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O
exec < /dev/null
timeout 600 python -m pytest tests/test_deferred_graph.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > $O/pytest_deferred.log
tail -150 $O/pytest_deferred.log
