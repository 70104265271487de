"""Issue cost of the instruction classes of the march's row step (vrg_debug_valu_rate modes 0..21) at 2 and 3 waves per SIMD, ~10 ms per launch.
One JSON line per (mode, waves per SIMD): lane-instructions per second and SIMD-cycles per wave-instruction at an ASSUMED clock (--ghz, default 2.4);
run it under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` for the real clock of each launch (tools/summarize_pmc.py: counter / 8 XCDs / duration).

    python tools/probe_valu_classes.py [--ms 10] [--ghz 2.4] [--json gpurun_out/valu_classes.json]
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops

ap = argparse.ArgumentParser()
ap.add_argument("--ms", type=float, default=10.0)
ap.add_argument("--ghz", type=float, default=2.4)
ap.add_argument("--json", default="")
ap.add_argument("--waves", default="2,3")
a = ap.parse_args()
NAMES = {0: "v_fma_f32", 1: "v_mad_u64_u32", 2: "v_log_f32", 3: "v_pk_fma_f32", 4: "v_xor_b32", 5: "sqrt/sin/cos/rcp", 6: "v_cmp+v_cndmask", 7: "v_mul/fma_f64",
         8: "v_add_f32", 9: "v_add_f32_dpp wave_shr", 10: "v_add_f32_dpp row_shr", 11: "v_mov_b32_dpp quad_perm", 12: "v_cndmask_b32", 13: "v_max_f32",
         14: "v_mul_f32", 15: "v_cvt_f32_u32", 16: "v_mul_hi_u32", 17: "v_mul_lo_u32", 18: "add -> s_nop 1 -> add_dpp(result)", 19: "v_sin_f32",
         20: "v_pk_mul/add_f32", 21: "v_mul -> v_add dependent"}
dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
rows = []
for wps in [int(v) for v in a.waves.split(",")]:
    blocks = cus * wps
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    for mode in sorted(NAMES):
        def run(it):
            e0, e1 = ops.HipEvent(), ops.HipEvent()
            e0.record()
            _hip.check(_hip.lib().vrg_debug_valu_rate(_hip.ptr(out), blocks, it, mode, _hip.current_stream()), "valu")
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_ms(e1)
        run(64)
        t = run(1024)
        iters = max(64, int(1024 * a.ms / max(t, 1e-3)))
        ms = sorted(run(iters) for _ in range(3))[1]
        row = {"mode": mode, "instr": NAMES[mode], "waves_per_simd": wps, "iters": iters, "ms": round(ms, 3),
               "tera_lane_instr_s": round(blocks * 256 * iters * 64 / ms / 1e9, 2),
               f"simd_cycles_per_wave_instr_at_{a.ghz}GHz": round(ms * 1e-3 * a.ghz * 1e9 / (iters * 64 * wps), 3)}
        rows.append(row)
        print(json.dumps(row), flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    json.dump({"device": torch.cuda.get_device_properties(0).name, "cus": cus, "rows": rows}, open(a.json, "w"), indent=1)
