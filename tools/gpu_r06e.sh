#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O
exec < /dev/null
timeout 600 python -m pytest tests/test_deferred_graph.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_deferred.log
tail -5 $O/pytest_deferred.log
timeout 300 python tools/diag_lazy_graph.py --reps 14 2>&1 | grep diag > $O/diag_deferred.log
VRGDG_DEFER_GRAPH=0 timeout 300 python tools/diag_lazy_graph.py --reps 8 2>&1 | grep diag > $O/diag_lazy_only.log
cat $O/diag_deferred.log; cat $O/diag_lazy_only.log
for mode in 1 0; do
  ( cd /tmp && VRGDG_SELFCHECK=0 VRGDG_DEFER_GRAPH=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o g -- python $GRAFT_REPO_ROOT/tools/prof_graph.py > $O/prof_graph_$mode.log 2>&1 )
  f=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); cp "$f" $O/graph_defer${mode}_kernel_stats.csv 2>/dev/null
  grep "pending\|checksum" $O/prof_graph_$mode.log; cat $O/graph_defer${mode}_kernel_stats.csv | cut -c1-150
done
rm -rf $O/trace_1 $O/trace_0
