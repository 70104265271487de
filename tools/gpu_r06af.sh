#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06af; export TMPDIR=/tmp
exec < /dev/null
timeout 900 python -m pytest tests/test_deferred_graph.py tests/test_surface.py -m gpu -x -q > gpurun_out/r06af/pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r06af/pytest.log | tail -3; grep -B5 -A25 "Error\|FAILED" gpurun_out/r06af/pytest.log | head -60
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-live-traffic --no-fast-variant > gpurun_out/r06af/bench.json 2> gpurun_out/r06af/err.log
python - <<'PY'
import json
b=json.loads(open("gpurun_out/r06af/bench.json").read().strip().splitlines()[-1])
g=b["configs"]["graph_four_nodes_device_resident"]
print(b["value"], g.get("ms_per_graph"), g.get("Mpix_s"), g.get("vs_headline"), g.get("bit_identical_to_ops_fused_chain"), g.get("error"))
PY
