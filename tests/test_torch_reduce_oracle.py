"""CPU checks of oracle/torch_device_reduce.py -- the numpy restatement of torch-ROCm's mean / std reductions -- against ground
truth collected from torch itself on an MI355X (tests/golden/torch_reduce_truth.npz: result bits for seeded CPU-generated inputs,
tools/probe_torch_reduce.py collect) and against the launch geometry rocprofv3 recorded for the same calls
(tests/golden/torch_reduce_geometry.json).  The HIP kernels (csrc/vrg_torch_stats.hip) are compared with torch directly by the
-m gpu suite; this file pins the restatement that documents the algorithm, and the host-side geometry logic, without a GPU."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import torch_device_reduce as TR      # noqa: E402
import probe_torch_reduce as P                    # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
TRUTH = np.load(os.path.join(GOLDEN, "torch_reduce_truth.npz"))
CASES = [(i, s) for i, s in enumerate(P.SHAPES) if s[0] * s[1] * s[2] <= 600_000]      # the 1080p / 4K cases: tools/probe_torch_reduce.py check


def test_device_properties_of_the_collecting_gpu():
    assert list(TRUTH["device_props"]) == [TR.NUM_MP, 2048, TR.WARP]


@pytest.mark.parametrize("i,shape", CASES, ids=[f"{b}x{H}x{W}" for _, (b, H, W) in CASES])
def test_restatement_reproduces_the_device_bits(i, shape):
    b, H, W = shape
    key = f"{b}x{H}x{W}"
    x = P.make_input(b, H, W, 1000 + i)
    assert x.double().sum().item() == float(TRUTH[key + "_insum"][0])        # same inputs as on the collecting machine
    for off, suffix in ((0, ""), (1, "_off1")):
        if key + "_mean" + suffix not in TRUTH:
            continue
        m, s = TR.mean_std(x.numpy(), base_offset_elems=off)
        assert np.array_equal(m.view(np.int32), TRUTH[key + "_mean" + suffix]), "mean"
        ts = TRUTH[key + "_std" + suffix]
        same = (s.view(np.int32) == ts) | (np.isnan(s) & np.isnan(ts.view(np.float32)))
        assert same.all(), "std"


def test_geometry_is_the_one_the_profiler_saw():
    geo = json.load(open(os.path.join(GOLDEN, "torch_reduce_geometry.json")))["shapes"]
    assert len(geo) == 100
    for key, rec in geo.items():
        b, H, W = (int(v) for v in key.split("_")[0].split("x"))
        if H * W == 1:          # TensorIterator drops the size-1 reduced dimension: another (numerically irrelevant) geometry
            continue
        for name, vec in (("mean", 4), ("std", 2)):
            cfg = TR.ReduceConfig(3 * b, H * W, vec)
            assert [cfg.block_width, cfg.block_height] == rec[name]["block"], (key, name, cfg)
            outputs_per_block = 1 if cfg.split_warps else cfg.block_height
            assert rec[name]["grid_blocks"] == [-(-3 * b // outputs_per_block), 1], (key, name, cfg)      # never split across workgroups


def test_library_geometry_equals_the_restatement():
    """csrc/vrg_torch_stats.hip::ts_config (host code of the shipped library) against the restatement over a sweep of call shapes."""
    import ctypes as C
    from __graft_entry__ import load_package
    load_package()
    from comfyui_vrgamedevgirl_amd import _hip
    lib = _hip.load_library()
    out = (C.c_int32 * 4)()
    sizes = sorted(set([1, 2, 3, 5, 7, 35, 64, 127, 128, 129, 143, 225, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1600, 2047, 2048, 4096, 8191,
                        8192, 8193, 16384, 40000, 129600, 518400, 921600, 2073600, 8294400, 33177600]))
    for outputs in (3, 6, 9, 12, 15, 24, 48, 96, 300, 1500):
        for n in sizes:
            for vec in (2, 4):
                cfg = TR.ReduceConfig(outputs, n, vec)
                assert lib.vrg_debug_torch_reduce_config(outputs, n, vec, out) == 0
                assert list(out) == [cfg.block_width, cfg.block_height, int(cfg.split_warps), int(cfg.vectorize)], (outputs, n, vec, cfg, list(out))


def test_fma32_is_a_correctly_rounded_fused_multiply_add():
    rng = np.random.default_rng(5)
    a = rng.standard_normal(200_000).astype(np.float32) * np.float32(2.0) ** rng.integers(-30, 30, 200_000).astype(np.float32)
    b = rng.standard_normal(200_000).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64)) * (1 + rng.standard_normal(200_000) * 1e-7)).astype(np.float32)    # heavy cancellation
    got = TR.fma32(a, b, c)
    import fractions
    for j in range(0, 200_000, 997):
        exact = fractions.Fraction(float(a[j])) * fractions.Fraction(float(b[j])) + fractions.Fraction(float(c[j]))
        lo = np.float32(float(exact))                      # float(Fraction) rounds correctly to double; then to float: check both neighbours
        cands = [np.nextafter(lo, np.float32(-np.inf)), lo, np.nextafter(lo, np.float32(np.inf))]
        best = min(cands, key=lambda v: abs(fractions.Fraction(float(v)) - exact))
        ties = [v for v in cands if abs(fractions.Fraction(float(v)) - exact) == abs(fractions.Fraction(float(best)) - exact)]
        assert any(got[j] == v for v in ties), (j, got[j], cands)
