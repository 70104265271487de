#!/bin/bash
# Round 5, third GPU call (every step under its own timeout, stdin closed): the whole GPU suite on the tree with the lazy download and the
# any-geometry u8 kernel, host-fed rates, the kernel trace of the statistics-overlap experiment, the bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
exec < /dev/null
{
  echo "=== $(date) pytest"; timeout 480 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12
  echo "=== $(date) host fed"; timeout 150 python tools/host_fed.py --frames 16 --out $O/host_fed_nodes.json 2>&1 | grep "^\[host\]"
  echo "=== $(date) overlap trace"
  for S in "seq" "ovl 4"; do T=$(echo $S | tr ' ' '_'); ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$T -o t -- python $GRAFT_REPO_ROOT/tools/exp_tstats_overlap.py --frames 64 --once "$S" > /dev/null 2>&1 ); f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); echo "--- kernel stats, schedule $S (64 frames, 2 steps)"; [ -n "$f" ] && grep "vrg" "$f" | cut -d, -f1-4 | cut -c1-170 | head -8; [ -n "$f" ] && cp "$f" $O/overlap_${T}_kernel_stats.csv; done
  echo "=== $(date) bench"; ( time timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; tail -c 400 $O/bench.err
  echo "=== $(date) done"
} > $O/run.log 2>&1
rm -rf $O/prof_seq $O/prof_ovl_4
tail -60 $O/run.log | cut -c1-300
