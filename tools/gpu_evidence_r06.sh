#!/bin/bash
# Round 6, evidence run on the final tree (every step under its own timeout, stdin closed): the whole GPU suite, host-fed rates, the bench
# line, its PMC summary, rocprofv3 kernel statistics of the headline and of chain 3.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/${1:-r06g}; mkdir -p $O
exec < /dev/null
{
  echo "=== $(date) pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider 2>&1 | tail -40
  echo "=== $(date) smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  echo "=== $(date) host fed"; timeout 150 python tools/host_fed.py --frames 16 --out $O/host_fed_nodes.json 2>&1 | grep "^\[host\]" | cut -c1-260
  echo "=== $(date) bench"; ( time timeout 700 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; tail -c 300 $O/bench.err
  echo "=== $(date) pmc"; timeout 300 python tools/collect_bench_pmc.py $O/pmc_bench_kernels.json 2>&1 | grep "^\[pmc\]" | cut -c1-300
  for W in chain4_4k chain3_4k; do
    echo "=== $(date) rocprofv3 kernel stats $W"
    ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$W -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-host-fed --no-configs --no-fast-variant > $GRAFT_REPO_ROOT/$O/bench_prof_$W.json 2> /dev/null )
    f=$(find $O/prof_$W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${W}_kernel_stats.csv && head -4 "$f" | cut -c1-200
    rm -rf $O/prof_$W
  done
  echo "=== $(date) done"
} > $O/run.log 2>&1
tail -90 $O/run.log | cut -c1-330
