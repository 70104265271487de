#!/bin/bash
# Local wrapper: rebuild the library if any source changed (the .so travels with the snapshot), then run
# tools/gpu_quick.sh on the GPU box.   bash tools/gpu_go.sh <tag> [pytest -k expr] [diag --match]
set -e
cd "$(dirname "$0")/.."
python comfyui-vrgamedevgirl_amd/build_ext.py | tail -1
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-900} -- "bash tools/gpu_quick.sh '$1' '$2' '$3'" 2>&1 | tail -8
grep -n "passed\|failed\|^FAILED\|^E  *Failed" gpurun_out/quick_$1.log | cut -c1-300
