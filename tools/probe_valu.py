"""VALU issue-rate probe, long runs: >= 20 ms per launch at 1 / 2 / 4 / 8 waves per SIMD, for rocprofv3 --pmc GRBM_GUI_ACTIVE
(effective clock = GRBM_GUI_ACTIVE / kernel duration).  Prints one JSON line per launch with HIP-event times.

    python tools/probe_valu.py [target_ms]
"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops

target_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
names = ["v_fma_f32", "v_mad_u64_u32", "v_log_f32", "v_pk_fma_f32", "v_xor_b32", "sqrt/sin/cos/rcp", "v_cmp+v_cndmask", "v_mul/fma_f64"]
rows = []
for wps in (1, 2, 4, 8):
    blocks = cus * wps
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    for mode, name in enumerate(names):
        iters = 2048
        def run(it):
            a, b = ops.HipEvent(), ops.HipEvent()
            a.record()
            _hip.check(_hip.lib().vrg_debug_valu_rate(_hip.ptr(out), blocks, it, mode, _hip.current_stream()), "valu")
            b.record()
            return a.elapsed_ms(b)
        run(256)
        t = run(iters)
        iters = max(256, int(iters * target_ms / max(t, 1e-3)))
        ts = sorted(run(iters) for _ in range(3))
        ms = ts[1]
        lane_instr = blocks * 256 * iters * 64
        row = {"instr": name, "mode": mode, "waves_per_simd": wps, "blocks": blocks, "iters": iters, "ms": round(ms, 3),
               "tera_lane_instr_s": round(lane_instr / ms / 1e9, 2),
               "cycles_per_wave_instr_at_2p4GHz": round(ms * 1e-3 * 2.4e9 / (iters * 64 * wps), 3)}
        rows.append(row)
        print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"device": torch.cuda.get_device_properties(0).name, "cus": cus, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "valu_long.json"), "w"), indent=1)
