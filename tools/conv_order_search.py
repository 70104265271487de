"""Offline search: in which order does the device conv2d (MIOpen, depthwise 3x3, zero padding) add its taps?
Input: gpurun_out/conv_capture.npz written by tools/probe_cm_parity.py on the GPU box (raw conv2d outputs of small frames).
Tries every binary summation tree over the non-zero taps (products by +-1, +-2, 4 are exact, so only the ORDER matters)."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "gpurun_out", "conv_capture.npz"))

def taps(x, H, W, weights):
    """list of fp32 arrays w*x shifted, zero padded, for the non-zero weights in raster order"""
    xp = np.zeros((x.shape[0], x.shape[1], x.shape[2] + 2, x.shape[3] + 2), np.float32)
    xp[:, :, 1:-1, 1:-1] = x
    out = []
    for kh in range(3):
        for kw in range(3):
            w = weights[kh][kw]
            if w != 0:
                out.append((f"{kh}{kw}", (np.float32(w) * xp[:, :, kh:kh + x.shape[2], kw:kw + x.shape[3]]).astype(np.float32)))
    return out

def trees(items):
    """all binary trees over the ordered list `items` (Catalan), as nested tuples"""
    if len(items) == 1:
        yield items[0]
        return
    for i in range(1, len(items)):
        for l in trees(items[:i]):
            for r in trees(items[i:]):
                yield (l, r)

def ev(t, vals):
    if isinstance(t, tuple):
        return (ev(t[0], vals) + ev(t[1], vals)).astype(np.float32)
    return vals[t]

def show(t, names):
    return f"({show(t[0], names)}+{show(t[1], names)})" if isinstance(t, tuple) else names[t]

K = {"lap": [[0, -1, 0], [-1, 4, -1], [0, -1, 0]], "gx": [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], "gy": [[-1, -2, -1], [0, 0, 0], [1, 2, 1]]}
for size in ("s", "m", "l"):
    x_full = z[f"{size}_x"]
    # the capture holds a 26x42 crop of the input and the 24x40 top-left crop of the output: rows/cols 0..23/0..39 need input
    # rows -1..24 -> the zero border on top/left, real pixels on the bottom/right of the crop
    for name, w in K.items():
        want = z[f"{size}_{name}"]
        H, W = want.shape[2], want.shape[3]
        xin = x_full
        xp = np.zeros((xin.shape[0], xin.shape[1], xin.shape[2] + 1, xin.shape[3] + 1), np.float32)
        xp[:, :, 1:, 1:] = xin                                  # zero row/col on top/left only
        tl = []
        for kh in range(3):
            for kw in range(3):
                if w[kh][kw] != 0:
                    tl.append((f"{w[kh][kw]:+d}*x[{kh - 1:+d},{kw - 1:+d}]", (np.float32(w[kh][kw]) * xp[:, :, kh:kh + H, kw:kw + W]).astype(np.float32)))
        if xin.shape[2] < H + 1 or xin.shape[3] < W + 1:        # small frame: the crop IS the frame, pad bottom/right too
            pad = np.zeros((xin.shape[0], xin.shape[1], H + 2, W + 2), np.float32)
            pad[:, :, 1:1 + xin.shape[2], 1:1 + xin.shape[3]] = xin
            tl = [(f"{w[kh][kw]:+d}*x[{kh - 1:+d},{kw - 1:+d}]", (np.float32(w[kh][kw]) * pad[:, :, kh:kh + H, kw:kw + W]).astype(np.float32))
                  for kh in range(3) for kw in range(3) if w[kh][kw] != 0]
        names = [t[0] for t in tl]
        vals = [t[1] for t in tl]
        n = len(vals)
        hits = []
        seen = set()
        for perm in itertools.permutations(range(n)):
            for t in trees(list(perm)):
                got = ev(t, vals)
                if np.array_equal(got, want):
                    s = show(t, names)
                    if s not in seen:
                        seen.add(s); hits.append(s)
        raster = vals[0].copy()
        for v in vals[1:]:
            raster = (raster + v).astype(np.float32)
        print(f"[{size}] {name}: raster order equal: {np.array_equal(raster, want)}; matching trees: {len(hits)}")
        for h in hits[:6]:
            print("      ", h)
