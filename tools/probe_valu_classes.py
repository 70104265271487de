"""Issue cost of the instruction classes of the march's row step (vrg_debug_valu_rate modes 0..73) at 2 and 3 waves per SIMD, ~10 ms per launch.
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
         20: "v_pk_mul/add_f32", 21: "v_mul -> v_add dependent",
         30: "v_mov_b32", 31: "v_sub_f32", 32: "v_min_f32", 33: "v_med3_f32", 34: "v_fmamk_f32", 35: "v_fmac_f32", 36: "v_add_f32 clamp", 37: "v_cmp_u_f32", 38: "v_cmp_class_f32", 39: "v_cmp + 3 v_cndmask (per instruction)", 40: "v_cndmask_b32_e64 (SGPR mask)", 41: "v_rcp_f32", 42: "v_exp_f32", 43: "v_sqrt_f32", 44: "v_cos_f32", 45: "v_rndne_f32", 46: "v_floor_f32", 47: "v_fract_f32", 48: "v_ldexp_f32", 49: "v_frexp_mant_f32", 50: "v_cvt_i32_f32", 51: "v_cvt_u32_f32", 52: "v_mov_b32_dpp wave_shr", 53: "v_add_f32 + s_nop 1", 54: "v_readlane_b32", 55: "v_mul_f32 (SGPR operand)", 56: "v_mul_f32 (literal operand)", 57: "v_mul_f32 (inline constant)", 60: "v_add_u32", 61: "v_and_b32", 62: "v_lshlrev_b32", 63: "v_lshrrev_b32", 64: "v_or_b32", 65: "v_add3_u32", 66: "v_lshl_add_u32", 67: "v_bfe_u32", 68: "v_mad_u32_u24", 69: "v_addc_co_u32", 72: "v_pk_add_f32", 73: "v_pk_mul_f32"}
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
