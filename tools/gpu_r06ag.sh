#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06ag; export TMPDIR=/tmp
exec < /dev/null
timeout 900 python -m pytest tests/test_deferred_graph.py -m gpu -x -q > gpurun_out/r06ag/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r06ag/pytest.log | tail -2; grep -B3 -A30 "Error\|FAILED\|assert" gpurun_out/r06ag/pytest.log | head -70
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-fed --no-live-traffic --no-fast-variant --frames 64 > gpurun_out/r06ag/bench.json 2> gpurun_out/r06ag/err.log; python - <<'PY'
import json
b=json.loads(open("gpurun_out/r06ag/bench.json").read().strip().splitlines()[-1])
g=b["configs"]["graph_four_nodes_device_resident"]; print(b["value"], g.get("ms_per_graph"), g.get("vs_headline"), g.get("error"))
PY
