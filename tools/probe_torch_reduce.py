"""Ground truth for the device statistics: what torch-ROCm's `mean(dim=[2,3])` / `std(dim=[2,3])` return on the MI355X for
seeded CPU-generated Lab-like `[b,3,H,W]` tensors (the reductions of /root/reference/nodes.py:99-100,109-110).

    python tools/probe_torch_reduce.py collect gpurun_out/torch_reduce_truth.npz      # on the GPU box
    python tools/probe_torch_reduce.py check   gpurun_out/torch_reduce_truth.npz      # anywhere: emulator vs the saved bits

Inputs come from torch's CPU generator (identical on every machine with this torch build), so the emulator
(oracle/torch_device_reduce.py) can be developed against the saved outputs without a GPU.  Run `collect` under
`rocprofv3 --kernel-trace` as well: the trace carries grid / workgroup sizes and the template arguments of each reduce_kernel.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [
    # (b, H, W): video sizes, the three block shapes (b = 1, 2, >= 3)
    (1, 2160, 3840), (2, 2160, 3840), (3, 2160, 3840), (4, 1080, 1920), (1, 1080, 1920), (2, 1080, 1920), (8, 540, 960), (5, 540, 960),
    (1, 720, 1280), (1, 480, 640), (16, 64, 64), (4, 64, 64),
    # thumbnails, unaligned planes (H*W % 4 != 0), the vectorisation threshold (128 elements), tiny frames
    (1, 40, 40), (1, 72, 120), (3, 72, 120), (1, 5, 7), (2, 5, 7), (7, 5, 7), (1, 15, 15), (2, 15, 15), (5, 15, 15), (1, 11, 13), (5, 11, 13),
    (1, 8, 16), (1, 1, 127), (1, 1, 128), (1, 1, 129), (1, 1, 130), (3, 1, 131), (1, 31, 33), (4, 31, 33), (1, 1, 1), (2, 1, 1), (1, 1, 2), (1, 1, 3), (1, 2, 2),
    (1, 1, 5), (1, 3, 3), (1, 16, 16), (1, 23, 29), (2, 23, 29), (3, 23, 29), (1, 100, 100), (2, 100, 100), (3, 100, 100), (1, 64, 128), (1, 90, 91),
    (6, 90, 91), (1, 255, 257), (1, 256, 256), (1, 512, 512), (2, 511, 513), (1, 1023, 1025),
]


def make_input(b, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((b, 3, H, W), generator=g)
    scale = torch.tensor([100.0, 120.0, 120.0]).view(1, 3, 1, 1)
    off = torch.tensor([0.0, -60.0, -60.0]).view(1, 3, 1, 1)
    return (x * scale + off).contiguous()


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int32).numpy().copy()


def collect(path):
    dev = torch.device("cuda", 0)
    out = {}
    for i, (b, H, W) in enumerate(SHAPES):
        x = make_input(b, H, W, 1000 + i)
        xd = x.to(dev)
        m = xd.mean(dim=[2, 3], keepdim=True)
        s = xd.std(dim=[2, 3], keepdim=True)
        key = f"{b}x{H}x{W}"
        out[key + "_mean"] = bits(m).reshape(b, 3)
        out[key + "_std"] = bits(s).reshape(b, 3)
        out[key + "_insum"] = np.array([x.double().sum().item()])
        # the same reductions on a slice view whose planes start at odd offsets (what a non-fresh tensor would give)
        if H * W >= 8:
            big = torch.zeros(b * 3 * H * W + 3, device=dev)
            v = big[1:1 + b * 3 * H * W].view(b, 3, H, W)
            v.copy_(xd)
            out[key + "_mean_off1"] = bits(v.mean(dim=[2, 3], keepdim=True)).reshape(b, 3)
            out[key + "_std_off1"] = bits(v.std(dim=[2, 3], keepdim=True)).reshape(b, 3)
        torch.cuda.synchronize()
        print(key, "ok", flush=True)
    p = torch.cuda.get_device_properties(0)
    out["device_props"] = np.array([p.multi_processor_count, getattr(p, "max_threads_per_multi_processor", -1), getattr(p, "warp_size", -1)])
    np.savez_compressed(path, **out)
    print("saved", path, "props", out["device_props"])


def check(path):
    from oracle import torch_device_reduce as TR
    truth = np.load(path)
    bad = 0
    for i, (b, H, W) in enumerate(SHAPES):
        key = f"{b}x{H}x{W}"
        if key + "_mean" not in truth:
            continue
        x = make_input(b, H, W, 1000 + i)
        assert abs(x.double().sum().item() - float(truth[key + "_insum"][0])) == 0.0, "inputs differ from the collecting machine's"
        xn = x.numpy()
        for off, suffix in ((0, ""), (1, "_off1")):
            if key + "_mean" + suffix not in truth:
                continue
            m, s = TR.mean_std(xn, base_offset_elems=off)
            em = (m.view(np.int32) != truth[key + "_mean" + suffix]).sum()
            es = (s.view(np.int32) != truth[key + "_std" + suffix]).sum()
            # NaN std (single element): any NaN equals any NaN
            if es:
                tn = truth[key + "_std" + suffix].view(np.float32)
                es = int(((s.view(np.int32) != truth[key + "_std" + suffix]) & ~(np.isnan(tn) & np.isnan(s))).sum())
            status = "ok" if not (em or es) else f"MISMATCH mean {em} std {es}"
            if em or es:
                bad += 1
            print(f"{key + suffix:>22}  {status}")
    print("mismatching shapes:", bad)
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        sys.exit(1 if check(sys.argv[2]) else 0)
