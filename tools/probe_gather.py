"""Fetch-pattern probes of the LUT record table (vrg_debug_lut_fetch), timed the way every round-4 A/B is timed:
every variant once per round, ROUNDS >= 5 interleaved rounds on one box, median and spread (max - min) per variant.

    python tools/probe_gather.py [--frames 16] [--rounds 7] [--modes 0,1,9,10,2,11,12,3,4] [--json out.json]
    PROBE_PMC=1 python tools/probe_gather.py --modes 0,12 --rounds 1     (under rocprofv3 --pmc: few launches, uniform and smooth)

mode 0: 6 x 16 B per lane (what the kernels do); 1: 3 x 16 B; 9: 1 x 16 B; 10: pieces 0 and 5 (same lines as mode 0, a third of the
lane-requests); 2: quad-cooperative into VGPRs; 11: the six pieces by LDS-DMA; 12: quad-cooperative LDS-DMA; 3: 64-byte records;
4: cell-major 128-byte records; 19: mode 12 over the cell-major table (round 5)."""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--inner", type=int, default=3)
ap.add_argument("--modes", default="0,1,9,10,2,11,12,3,4")
ap.add_argument("--json", default="")
ap.add_argument("--lut", default="AMD_TealOrange_33.cube", help="cube under the package's LUTS/ (AMD_WarmFilm_25.cube: the size 8 of the reference's 12 cubes have)")
a = ap.parse_args()
modes = [int(m) for m in a.modes.split(",")]
dev = torch.device("cuda", 0)
H, W = 2160, 3840
px = a.frames * H * W
g = torch.Generator(device=dev).manual_seed(3)
uniform = torch.rand((a.frames, H, W, 3), generator=g, device=dev)
yy = torch.linspace(0, 1, H, device=dev)[None, :, None, None]
xx = torch.linspace(0, 1, W, device=dev)[None, None, :, None]
ph = torch.arange(a.frames, device=dev, dtype=torch.float32)[:, None, None, None] * 0.37
ch = torch.arange(3, device=dev, dtype=torch.float32)[None, None, None, :]
smooth = (0.5 + 0.25 * torch.sin(6.0 * xx + ph + ch) + 0.2 * torch.cos(4.0 * yy - ph + 2 * ch) + 0.02 * torch.randn((a.frames, H, W, 3), generator=g, device=dev)).clamp_(0, 1).contiguous()
lut33 = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, a.lut)), dev)
n = lut33.size
nc = n - 1
# the record form [(N-1)][(N-1)][N][12] built here from the raw [b][g][r][ch] table (the library's own table is opaque: cubes up to 28^3 are
# stored cell-major since round 5): rec[b0][g0][r][ch * 4 + dg * 2 + db] = raw[b0 + db][g0 + dg][r][ch]
raw = lut33.nodes
recs = torch.stack([raw[db:db + nc, dg:dg + nc, :, ch] for ch in range(3) for dg in (0, 1) for db in (0, 1)], dim=-1).reshape(nc * nc, n, 12).contiguous()
rec_table = recs.reshape(-1).contiguous()
wide = torch.zeros((nc * nc * n * 16 + 16,), dtype=torch.float32, device=dev)
wide[:-16].view(-1, 16)[:, :12] = recs.view(-1, 12)
cellmajor = torch.zeros((nc * nc, nc, 32), dtype=torch.float32, device=dev)
cellmajor[:, :, :12] = recs[:, :-1]
cellmajor[:, :, 12:24] = recs[:, 1:]
cellmajor = cellmajor.reshape(-1).contiguous()
tables = {3: wide, 4: cellmajor, 19: cellmajor}
probe = torch.empty((px,), dtype=torch.float32, device=dev)


def launch(mode, src):
    t = tables.get(mode, rec_table)
    _hip.check(_hip.lib().vrg_debug_lut_fetch(_hip.ptr(src), _hip.ptr(probe), px, _hip.ptr(t), n, mode, _hip.current_stream()), "probe")


cases = [(m, name, src) for name, src in (("uniform", uniform), ("smooth", smooth)) for m in modes]
if os.environ.get("PROBE_PMC"):
    for m, name, src in cases:
        launch(m, src)
    torch.cuda.synchronize()
    sys.exit(0)
for m, name, src in cases:      # warm-up
    launch(m, src)
torch.cuda.synchronize()
times = {(m, name): [] for m, name, _ in cases}
for r in range(a.rounds):
    for m, name, src in cases:
        best = 1e9
        for _ in range(a.inner):
            e0, e1 = ops.HipEvent(), ops.HipEvent()
            e0.record(); launch(m, src); e1.record()
            best = min(best, e0.elapsed_ms(e1))
        times[(m, name)].append(best)
rows = []
for (m, name), ts in times.items():
    med = statistics.median(ts)
    rows.append({"mode": m, "data": name, "lut": a.lut, "frames": a.frames, "ms_median": round(med, 4), "ms_min": round(min(ts), 4), "ms_max": round(max(ts), 4),
                 "spread_pct": round(100.0 * (max(ts) - min(ts)) / med, 2), "gpix_s": round(px / med / 1e6, 1), "rounds": a.rounds})
    print("[probe]", rows[-1], flush=True)
if a.json:
    with open(a.json, "w") as fh:
        json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, fh, indent=1)
