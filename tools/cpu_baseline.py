"""CPU baseline of the reference's own nodes on THIS box's host cores (the build container has the reference checkout):
per node and the chains of BASELINE.json's configs, warm-up 1, median of 3.  Writes profiles/r04_cpu_baseline_buildbox.json.

    python tools/cpu_baseline.py [frames_4k] [frames_1080p]
"""
import json, os, platform, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import reference_loader as RL
from oracle import restated as R

f4k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
f1080 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lut_cpu = R.parse_cube_file(os.path.join(ROOT, "comfyui-vrgamedevgirl_amd", "LUTS", "AMD_TealOrange_33.cube"))
out = {"host": platform.node(), "os_cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
       "reference_available": RL.reference_available(), "rows": []}
for label, (H, W, n) in {"4K": (2160, 3840, f4k), "1080p": (1080, 1920, f1080)}.items():
    for stages in (("grain", "lut", "colormatch", "sharpen"), ("grain", "lut", "sharpen"), ("grain", "lut"), ("colormatch",)):
        row = bench.cpu_baseline(stages, n, H, W, lut_cpu, per_node=(len(stages) == 4))
        row.update({"size": label, "chain": "+".join(stages)})
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_cpu_baseline_buildbox.json"), "w"), indent=1)
