#!/bin/bash
# which class of box is this?  the headline leg alone (40 s)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06ad; export TMPDIR=/tmp
exec < /dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline --no-host-fed --no-live-traffic --no-fast-variant > gpurun_out/r06ad/bench.json 2> /dev/null
python - <<'PY'
import json
b=json.loads(open("gpurun_out/r06ad/bench.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["passes_ms"], b["roofline"].get("clock_during_timed_steps",{}).get("sclk_mhz_median"), b["roofline"].get("clock_during_timed_steps",{}).get("socket_power_w_mean"))
PY
