#!/bin/bash
# the graph leg measured 599 ms once (r06fin3) where it measures 49-52: reproduce?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06ae; export TMPDIR=/tmp
exec < /dev/null
for rep in 1 2 3; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-live-traffic --no-fast-variant > gpurun_out/r06ae/bench_$rep.json 2> gpurun_out/r06ae/err_$rep.log
  python - <<PY
import json
b=json.loads(open("gpurun_out/r06ae/bench_$rep.json").read().strip().splitlines()[-1])
g=b["configs"]["graph_four_nodes_device_resident"]
print($rep, b["value"], g.get("ms_per_graph"), g.get("Mpix_s"), g.get("clock_during_timed_steps",{}).get("socket_power_w_mean"), g.get("error"))
PY
done
