#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06ac; mkdir -p $O
exec < /dev/null
for rep in 1 2; do
for n in head prev new; do
  if [ $n = new ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_r6_$n.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline --no-host-fed --no-live-traffic --no-fast-variant > $O/bench_${n}_$rep.json 2> /dev/null
  python - <<PY
import json
b=json.loads(open("$O/bench_${n}_$rep.json").read().strip().splitlines()[-1])
print("$n", $rep, b["value"], b["ms_per_step"], b["roofline"]["passes_ms"], b["roofline"].get("clock_during_timed_steps",{}).get("sclk_mhz_median"), b["roofline"].get("clock_during_timed_steps",{}).get("socket_power_w_mean"))
PY
done; done
