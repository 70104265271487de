"""PCIe copy rates from / to page-locked host memory: one stream, two or four streams splitting the buffer (several SDMA engines?), both
directions at once, and a copy KERNEL reading the pinned host buffer over the bus instead of the DMA engines.
    python tools/probe_pcie.py [--mb 1024] [--json gpurun_out/pcie.json]"""
import argparse, json, os, sys, time
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=1024)
ap.add_argument("--json", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
n = a.mb << 20
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True); h_in.fill_(3)
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device=dev)
d_b = torch.empty(n, dtype=torch.uint8, device=dev); d_b.fill_(5)
streams = [torch.cuda.Stream(dev) for _ in range(8)]
rows = []


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def split_copy(dst, src, k, base=0):
    step = (n + k - 1) // k
    for i in range(k):
        with torch.cuda.stream(streams[base + i]):
            dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)


for k in (1, 2, 4):
    t = timed(lambda: split_copy(d_a, h_in, k)); rows.append({"what": f"host -> device, {k} stream(s)", "GB_s": round(n / t / 1e9, 2)})
    t = timed(lambda: split_copy(h_out, d_b, k)); rows.append({"what": f"device -> host, {k} stream(s)", "GB_s": round(n / t / 1e9, 2)})
for k in (1, 2):
    def duplex():
        split_copy(d_a, h_in, k, 0); split_copy(h_out, d_b, k, 4)
    t = timed(duplex); rows.append({"what": f"both directions at once, {k} stream(s) each", "GB_s_each_way": round(n / t / 1e9, 2)})
# chunked like the node pipeline: 64 MB pieces on one stream per direction
def chunked(mb):
    step = mb << 20
    for off in range(0, n, step):
        with torch.cuda.stream(streams[0]):
            d_a[off:off + step].copy_(h_in[off:off + step], non_blocking=True)
        with torch.cuda.stream(streams[4]):
            h_out[off:off + step].copy_(d_b[off:off + step], non_blocking=True)
for mb in (16, 64, 256):
    t = timed(lambda: chunked(mb)); rows.append({"what": f"both directions, {mb} MB pieces, one stream each", "GB_s_each_way": round(n / t / 1e9, 2)})
for r in rows:
    print(json.dumps(r), flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "MB": a.mb, "rows": rows}, open(a.json, "w"), indent=1)
