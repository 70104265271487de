"""Mpixels/s of the headline chain against frames per step (bench.py --frames N, no CPU baseline / PMC legs): the small-batch regime of
real ComfyUI calls and of strong scaling.   python tools/frames_table.py [--out gpurun_out/frames_table.json] [--frames 4,8,16,32,64,128,256]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/frames_table.json")
ap.add_argument("--frames", default="4,8,16,32,64,128,256")
ap.add_argument("--workload", default="chain4_4k")
args = ap.parse_args()
rows = []
for F in [int(v) for v in args.frames.split(",")]:
    steps = max(5, min(40, 1280 // F))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", str(F), "--steps", str(steps), "--warmup", "3", "--workload", args.workload,
                        "--no-cpu-baseline", "--no-live-traffic", "--no-verify"], capture_output=True, text=True, timeout=900)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    row = {"frames_per_step": F, "steps": steps, "ms_per_step": line["ms_per_step"], "Mpix_s": line["value"], "passes_ms": line["roofline"]["passes_ms"],
           "sum_of_passes_ms": round(sum(line["roofline"]["passes_ms"].values()), 3), "reference_stats_ms_alone": line.get("reference_stats_ms_per_step"),
           "fast_variant_Mpix_s": (line.get("fast_variant") or {}).get("value")}
    rows.append(row)
    print("[frames]", row, flush=True)
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
with open(args.out, "w") as fh:
    json.dump({"workload": args.workload, "rows": rows}, fh, indent=1)
