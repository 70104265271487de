"""Shader clock and socket power WHILE a kernel of this library runs in a loop: a thread polls the amdgpu sysfs / hwmon files (freq1_input = sclk,
power1_average / power1_input) and, where present, `rocm-smi --showclocks --showpower --json`, every 100 ms during ~3 s of each workload.

    python tools/sample_clocks.py [--seconds 3] [--frames 64] [--json gpurun_out/clocks.json]
"""
import argparse, glob, json, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops, cube, VRGDG_IV_Adjustments as iv
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--json", default="")
a = ap.parse_args()


def sysfs_sources():
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input"):
                f = os.path.join(hw, name)
                if os.path.exists(f):
                    src[f"{os.path.basename(os.path.dirname(card))}.{name}"] = f
        f = os.path.join(card, "pp_dpm_sclk")
        if os.path.exists(f):
            src[f"{os.path.basename(os.path.dirname(card))}.pp_dpm_sclk"] = f
    return src


SRC = sysfs_sources()
HAVE_SMI = subprocess.run("which rocm-smi", shell=True, capture_output=True).returncode == 0


def poll(stop, rows):
    while not stop.is_set():
        r = {"t": time.time()}
        for k, f in SRC.items():
            try:
                txt = open(f).read().strip()
                if k.endswith("pp_dpm_sclk"):
                    cur = [l for l in txt.splitlines() if l.rstrip().endswith("*")]
                    r[k] = cur[0] if cur else txt[:80]
                else:
                    r[k] = int(txt)
            except Exception as e:
                r[k] = f"err {type(e).__name__}"
        rows.append(r)
        time.sleep(0.1)


def smi():
    if not HAVE_SMI:
        return None
    try:
        p = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(p.stdout) if p.stdout.strip().startswith("{") else p.stdout[-400:] + p.stderr[-200:]
    except Exception as e:
        return f"err {e}"


dev = torch.device("cuda", 0)
F, H, W = a.frames, 2160, 3840
x = torch.rand((F, H, W, 3), generator=torch.Generator(device=dev).manual_seed(3), device=dev)
xv = bench.make_frames(F, H, W, dev, 1234, "video")
out = torch.empty_like(x)
ws = torch.empty_like(x)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
gen = torch.Generator(device=dev)
ref_ms = ops.reference_stats(x[:1])
probe = torch.empty(512 * 256, dtype=torch.float32, device=dev)


def chain(src, **kw):
    gen.manual_seed(5)
    ops.fused_chain(src, ops.ChainSpec(**kw), generator=gen, out=out, lab_workspace=ws)


WORK = {
    "idle": lambda: time.sleep(0.05),
    "v_fma_f32 probe, 2 waves/SIMD": lambda: _hip.check(_hip.lib().vrg_debug_valu_rate(_hip.ptr(probe), 512, 40000, 0, _hip.current_stream()), "valu"),
    "copy (nt float4)": lambda: _hip.check(_hip.lib().vrg_debug_copy_f32(_hip.ptr(x), _hip.ptr(out), x.numel(), 1, _hip.current_stream()), "copy"),
    "chain3 uniform": lambda: chain(x, grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)),
    "chain3 video": lambda: chain(xv, grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)),
    "grain->unsharp": lambda: chain(x, grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False)),
    "grain": lambda: ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen),
    "unsharp": lambda: ops.stencil3x3(x, "unsharp", 0.5, False),
    "headline chain4 (device policy)": lambda: chain(x, grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), cm_chunk=1),
}
report = {"device": torch.cuda.get_device_name(0), "sources": sorted(SRC), "rocm_smi": HAVE_SMI, "workloads": {}}
for name, fn in WORK.items():
    fn(); torch.cuda.synchronize()
    rows, stop = [], threading.Event()
    th = threading.Thread(target=poll, args=(stop, rows)); th.start()
    t0 = time.time(); n = 0
    mid = None
    while time.time() - t0 < a.seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize(); n += 4
        if mid is None and time.time() - t0 > a.seconds / 2:
            mid = smi()           # (costs ~1 s of wall clock; the GPU idles meanwhile -- the sysfs rows are the continuous record)
    stop.set(); th.join()
    summ = {}
    for k in SRC:
        vals = [r[k] for r in rows if isinstance(r.get(k), int)]
        if vals:
            vals.sort()
            summ[k] = {"median": vals[len(vals) // 2], "min": vals[0], "max": vals[-1], "n": len(vals)}
        else:
            summ[k] = sorted({str(r.get(k)) for r in rows})[:6]
    report["workloads"][name] = {"calls": n, "sysfs": summ, "rocm_smi_mid_run": mid}
    print("[clk]", name, json.dumps(summ), flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    json.dump(report, open(a.json, "w"), indent=1)
