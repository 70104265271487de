"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the committed reference
fixtures, on a real MI355X.  Integer / index work and everything whose arithmetic is fully specified
(grain, LUT, stencils, fused chains without colour match) is BIT-EXACT.  Colour match: the element-wise path of the
default "device" policy is BIT-EQUAL to the device oracle (the restated kornia formulas evaluated by torch on this GPU);
what remains -- per-frame fp32 reductions whose value depends on the batch shape in the reference itself -- is held to
a budget stated in ulps (kornia itself is unpinned: the restatement is the oracle).

Noise: the reference draws grain with torch.randn on the device; the device oracle for "identical seeds" is
therefore torch.randn itself on this GPU -- our in-register Philox/Box-Muller must reproduce it bit for bit,
and leave the generator exactly where torch would have left it.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import restated as R
from oracle import truth64
from conftest import GOLDEN, PKG_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    from comfyui_vrgamedevgirl_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", torch.cuda.current_device())


def _npz(name):
    return np.load(os.path.join(GOLDEN, name))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def assert_bit_equal(got, want, what=""):
    got, want = got.detach().cpu(), want.detach().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not torch.equal(got, want):
        diff = (got.double() - want.double()).abs()
        bad = int((got != want).sum())
        idx = int(torch.nonzero((got != want).flatten())[0])
        pytest.fail(f"{what}: {bad}/{got.numel()} elements differ, max abs {diff.max().item():.3e}, first at flat {idx}: "
                    f"got {got.flatten()[idx].item()!r} want {want.flatten()[idx].item()!r}")


# ---------------------------------------------------------------------------------------- loading
def test_native_library_is_loaded(pkg):
    from comfyui_vrgamedevgirl_amd import _hip
    lib = _hip.lib()
    assert lib.vrg_abi_version() == _hip.ABI_VERSION == 8
    with open("/proc/self/maps") as fh:
        assert any("libvrgdg_hip.so" in line for line in fh), "HIP extension not mapped into the process"
    import ctypes as C
    cu, mt = C.c_int32(), C.c_int32()
    assert lib.vrg_device_info(C.byref(cu), C.byref(mt)) == 0
    props = torch.cuda.get_device_properties(0)
    assert cu.value == props.multi_processor_count and mt.value == props.max_threads_per_multi_processor


def test_constant_divisions_equal_ieee_division_for_every_fp32_input(pkg, dev):
    """The kernels divide by compile-time constants with multiply + 2 FMAs (3 issue slots instead of ~14).
    Swept over all 2^32 fp32 bit patterns on the device: identical to x / c wherever 1e-30 <= |x| <= 1e30, and
    for the unsharp divisor 9 (bit-exact stencil path) identical for every input."""
    from comfyui_vrgamedevgirl_amd import _hip
    counts = torch.zeros(18, dtype=torch.int64, device=dev)
    _hip.check(_hip.lib().vrg_selftest_divconst(_hip.ptr(counts), _hip.current_stream()), "selftest")
    c = counts.cpu().tolist()
    assert c[:9] == [0] * 9, c
    assert c[17] == 0, c            # c = 9: exact everywhere (+-0 compare equal, Inf handled)


def test_box_muller_radius_sqrt_equals_ieee_sqrt_for_every_philox_word(pkg, dev):
    from comfyui_vrgamedevgirl_amd import _hip
    counts = torch.zeros(1, dtype=torch.int64, device=dev)
    _hip.check(_hip.lib().vrg_selftest_bm_radius(_hip.ptr(counts), _hip.current_stream()), "vrg_selftest_bm_radius")
    assert int(counts.item()) == 0


def test_dpp_lane_shifts(pkg, dev):
    """The wave-march kernel takes 3x3 taps and misaligned normals from neighbouring lanes with DPP wave shifts."""
    from comfyui_vrgamedevgirl_amd import _hip
    out = torch.full((128,), -1.0, device=dev)
    _hip.check(_hip.lib().vrg_selftest_lanes(_hip.ptr(out), _hip.current_stream()), "selftest lanes")
    o = out.cpu().tolist()
    assert o[1:64] == [float(i - 1) for i in range(1, 64)], "lane_prev must read lane-1"
    assert o[64:127] == [float(i + 1) for i in range(0, 63)], "lane_next must read lane+1"


# ---------------------------------------------------------------------------------------- noise stream
NOISE_CASES = [  # frames, frame_elems, chunk_frames
    (1, 3 * 5 * 7, 1), (3, 3 * 5 * 7, 2), (2, 3 * 64 * 64, 1), (4, 3 * 256 * 256, 4), (8, 3 * 512 * 512, 4), (5, 3 * 270 * 480, 0),
    (2, 3 * 1080 * 1920, 1),
]


@pytest.mark.parametrize("frames,fe,chunk", NOISE_CASES)
def test_noise_stream_is_torch_randn(ops, dev, frames, fe, chunk):
    torch.manual_seed(1234)
    torch.randn(7, device=dev)                       # move the generator off its initial state
    gen = torch.cuda.default_generators[dev.index]
    state = gen.get_state()
    step = chunk if chunk > 0 else frames
    want = []
    for i in range(0, frames, step):
        want.append(torch.randn((min(step, frames - i)) * fe, device=dev))
    want = torch.cat(want)
    offset_after_torch = gen.get_offset()
    gen.set_state(state)
    main, tail, n_full = ops.plan_noise(frames, fe, chunk, dev)
    got = []
    if main is not None:
        got.append(ops.torch_stream_noise(n_full * main.chunk_frames, fe, main, dev).flatten())
    if tail is not None:
        got.append(ops.torch_stream_noise(tail.chunk_frames, fe, tail, dev).flatten())
    got = torch.cat(got)
    assert gen.get_offset() == offset_after_torch, "generator not advanced like torch.randn"
    assert_bit_equal(got, want, "noise stream")


def test_noise_stream_with_explicit_generator(ops, dev):
    g1 = torch.Generator(device=dev).manual_seed(99)
    g2 = torch.Generator(device=dev).manual_seed(99)
    fe = 3 * 100 * 60
    want = torch.cat([torch.randn(2 * fe, device=dev, generator=g1) for _ in range(3)])
    main, tail, n_full = ops.plan_noise(6, fe, 2, dev, g2)
    got = ops.torch_stream_noise(6, fe, main, dev).flatten()
    assert tail is None and g1.get_offset() == g2.get_offset()
    assert_bit_equal(got, want, "explicit generator")


# ---------------------------------------------------------------------------------------- grain
def test_grain_injected_matches_reference_fixtures(ops, dev):
    z = _npz("grain.npz")
    x = _t(z["x"]).to(dev)
    for tag in ("default", "strong_colour", "mono_all", "workflow_widgets"):
        I, s = float(z[f"{tag}.I"]), float(z[f"{tag}.s"])
        out = ops.film_grain_injected(x, _t(z[f"{tag}.noise"]).to(dev), I, s)
        assert_bit_equal(out, _t(z[f"{tag}.out"]), f"grain injected {tag}")


GRAIN_CASES = [((5, 12, 16, 3), 2), ((3, 5, 7, 3), 0), ((2, 256, 256, 3), 1), ((6, 270, 480, 3), 4), ((1, 1, 1, 3), 4),
               ((4, 360, 640, 3), 0)]


@pytest.mark.parametrize("shape,bs", GRAIN_CASES)
def test_grain_in_register_noise_matches_oracle_on_identical_seed(ops, dev, shape, bs):
    x = _rand(shape, 5, -0.1, 1.1)
    torch.manual_seed(42)
    got = ops.film_grain(x.to(dev), 0.04, 0.5, chunk_frames=bs)
    torch.manual_seed(42)
    want = R.fast_film_grain(x, 0.04, 0.5, bs, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    assert_bit_equal(got, want, f"grain {shape} bs={bs}")


def test_fast_film_grain_node_end_to_end(pkg, dev):
    node = pkg.NODE_CLASS_MAPPINGS["FastFilmGrain"]()
    x = _rand((7, 33, 47, 3), 8)
    keep = x.clone()
    torch.manual_seed(3)
    (out,) = node.apply_grain(x, 0.2, 0.3, 3)
    torch.manual_seed(3)
    want = R.fast_film_grain(x, 0.2, 0.3, 3, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    assert out.device.type == "cpu" and out.dtype == torch.float32 and torch.equal(x, keep)
    assert_bit_equal(out, want, "FastFilmGrain node")
    torch.manual_seed(3)
    (again,) = node.apply_grain(x.to(dev), 0.2, 0.3, 3)          # device-resident input, same stream
    assert_bit_equal(again, want, "FastFilmGrain node (GPU input)")


def test_seeded_per_frame_grain_is_batch_invariant_and_matches_oracle(pkg, ops, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_StandaloneVideoEnhancerNodes as enh
    frames = torch.full((4, 12, 16, 3), 0.5)
    st = {"sharpen_enabled": False, "grain_enabled": True, "grain_intensity": 0.04, "saturation_mix": 0.5, "seed": 42, "use_gpu": False}
    whole = enh._apply_effects_batch(frames, st, 100)
    split = torch.cat((enh._apply_effects_batch(frames[:2], st, 100), enh._apply_effects_batch(frames[2:], st, 102)))
    assert torch.equal(whole, split)                          # the reference's own test property

    def noise_fn(fseed, shape):
        g = torch.Generator(device=dev).manual_seed(fseed)
        return torch.randn(shape, generator=g, device=dev).cpu()

    x = _rand((3, 40, 56, 3), 12)
    st = dict(st, sharpen_enabled=True, sharpen_strength=0.7)
    got = enh._apply_effects_batch(x, st, 7)
    want = R.seeded_grain(R.unsharp(x, 0.7, False), 0.04, 0.5, 42, 7, noise_fn=noise_fn)
    assert_bit_equal(got, want, "effects batch")


@pytest.mark.parametrize("zero_border", [False, True], ids=["replicate", "zero"])
@pytest.mark.parametrize("shape", [(3, 37, 344, 3), (2, 64, 1024, 3), (2, 90, 500, 3), (2, 5, 4096, 3), (1, 1, 2048, 3), (2, 1080, 1920, 3),
                                   (1, 2160, 3840, 3)], ids=lambda s: "x".join(map(str, s)))
def test_fused_sharpen_then_seeded_grain_equals_the_two_kernels(pkg, ops, dev, shape, zero_border):
    """vrg_sharpen_grain_f32 (the enhancer's sharpen -> per-frame-seeded grain order in one pass, grain geometry leading) against the
    stencil kernel followed by the grain kernel -- each of which is held to the oracle elsewhere -- bit for bit: row ends inside a wave
    (W*3/4 not a multiple of 64), runs that end inside the frame's last vector row, one-row frames, both border rules, 1080p and 4K;
    and against the oracle itself for the small shapes."""
    from comfyui_vrgamedevgirl_amd import _hip, rng
    import ctypes as C
    x = _rand(shape, 31).to(dev)
    two = ops.film_grain_seeded_frames(ops.stencil3x3(x, "unsharp", 0.6, zero_border), 0.05, 0.4, 1234, 17)
    got = ops.sharpen_then_seeded_grain(x, 0.6, zero_border, 0.05, 0.4, 1234, 17)
    assert_bit_equal(got, two, "fused sharpen -> seeded grain")
    # the fused kernel itself ran (the op falls back to the two kernels for sizes the kernel refuses)
    out = torch.empty_like(x)
    d = ops.NoisePlan(1, rng.per_frame_seeded(x[0].numel(), 1234 + 17, dev)).desc()
    st = _hip.lib().vrg_sharpen_grain_f32(_hip.ptr(x), _hip.ptr(out), shape[0], shape[1], shape[2], 0.6, 1 if zero_border else 0,
                                          0.05, 0.4, float(np.float32(1.0 - 0.4)), C.byref(d), _hip.current_stream())
    assert st == _hip.VRG_OK
    assert_bit_equal(out, two, "vrg_sharpen_grain_f32")
    if x.numel() < 1 << 20:
        def noise_fn(fseed, shp):
            g = torch.Generator(device=dev).manual_seed(fseed)
            return torch.randn(shp, generator=g, device=dev).cpu()
        want = R.seeded_grain(R.unsharp(x.cpu(), 0.6, zero_border).contiguous(), 0.05, 0.4, 1234, 17, noise_fn=noise_fn)
        assert_bit_equal(got, want, "fused sharpen -> seeded grain vs oracle")
        half = shape[0] // 2
        if half:
            split = torch.cat((ops.sharpen_then_seeded_grain(x[:half], 0.6, zero_border, 0.05, 0.4, 1234, 17),
                               ops.sharpen_then_seeded_grain(x[half:], 0.6, zero_border, 0.05, 0.4, 1234, 17 + half)))
            assert torch.equal(split, got)                    # the reference's batch-invariance property (its own test, :39-61)


def test_fused_sharpen_grain_refuses_what_it_does_not_take(pkg, ops, dev):
    """Widths that are not a multiple of 4 or below 344 pixels, several frames per noise chunk, in-place: the entry point says so
    (VRG_ERR_UNSUPPORTED / VRG_ERR_BAD_ARG), and the operator still returns the reference's result through the two kernels."""
    from comfyui_vrgamedevgirl_amd import _hip, rng
    import ctypes as C
    lib = _hip.lib()
    for shp in ((2, 20, 56, 3), (2, 20, 346, 3)):
        x = _rand(shp, 5).to(dev)
        out = torch.empty_like(x)
        d = ops.NoisePlan(1, rng.per_frame_seeded(x[0].numel(), 3, dev)).desc()
        args = (shp[0], shp[1], shp[2], 0.5, 0, 0.04, 0.5, 0.5, C.byref(d), _hip.current_stream())
        assert lib.vrg_sharpen_grain_f32(_hip.ptr(x), _hip.ptr(out), *args) == _hip.VRG_ERR_UNSUPPORTED
        assert lib.vrg_sharpen_grain_f32(_hip.ptr(x), _hip.ptr(x), *args) == _hip.VRG_ERR_BAD_ARG
        got = ops.sharpen_then_seeded_grain(x, 0.5, False, 0.04, 0.5, 3, 0)
        assert_bit_equal(got, ops.film_grain_seeded_frames(ops.stencil3x3(x, "unsharp", 0.5, False), 0.04, 0.5, 3, 0), "fallback")
    x = _rand((2, 20, 512, 3), 5).to(dev)
    d = ops.NoisePlan(2, rng.per_frame_seeded(x.numel(), 3, dev)).desc()
    assert lib.vrg_sharpen_grain_f32(_hip.ptr(x), _hip.ptr(torch.empty_like(x)), 2, 20, 512, 0.5, 0, 0.04, 0.5, 0.5, C.byref(d),
                                     _hip.current_stream()) == _hip.VRG_ERR_UNSUPPORTED


def test_route_film_grain_tensor_seeded(pkg, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as lvt
    x = _rand((2, 24, 40, 3), 14)
    got = lvt._apply_film_grain_tensor(x, 7.0, -3.0, "cuda", 11)         # clamped to I=1, s=0
    g = torch.Generator(device=dev).manual_seed(11)
    noise = torch.randn(x.shape, generator=g, device=dev).cpu()
    assert_bit_equal(got, R.grain_apply(x, noise, 1.0, 0.0), "route grain")


# ---------------------------------------------------------------------------------------- LUT
@pytest.mark.parametrize("tag", ["a", "b"])
def test_lut_matches_reference_fixtures(ops, dev, tag):
    z = _npz("lut.npz")
    lut = ops.upload_lut({"lut": _t(z[f"{tag}.lut"]), "domain_min": _t(z[f"{tag}.dmin"]), "domain_max": _t(z[f"{tag}.dmax"])}, dev)
    img, img4 = _t(z[f"{tag}.img"]).to(dev), _t(z[f"{tag}.img4"]).to(dev)
    for s in (10.0, 3.3, 0.0, 25.0):
        assert_bit_equal(ops.lut3d(img, lut, s), _t(z[f"{tag}.out.s{s}"]), f"lut {tag} strength {s}")
    assert_bit_equal(ops.lut3d(img4, lut, 10.0), _t(z[f"{tag}.out4.s10.0"]), f"lut {tag} 4 channels")
    assert_bit_equal(ops.lut3d(img, lut, 6.5), _t(z[f"{tag}.route.s6.5"]), f"lut {tag} route")


def test_lut_node_and_helpers_on_shipped_cubes(pkg, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as lvt
    x = _rand((2, 31, 45, 3), 21, -0.2, 1.2)
    for name in ("AMD_TealOrange_33.cube", "AMD_WarmFilm_25.cube", "AMD_Identity_17.cube"):
        oracle_lut = R.parse_cube_file(os.path.join(iv.LUTS_DIR, name))
        for s in (10.0, 4.2):
            (out,) = iv.VRGDG_LUTS().apply_lut(x, name, "auto", s)
            assert out.device == x.device
            assert_bit_equal(out, R.apply_lut_with_strength(x, oracle_lut, s), f"{name} {s}")
        assert_bit_equal(lvt._apply_lut_tensor(x, name, 7.7, "cuda"), R.apply_lut_with_strength(x, oracle_lut, 7.7), name)
        direct = iv.VRGDG_LUTS._apply_cube_lut(x, oracle_lut["lut"], oracle_lut["domain_min"], oracle_lut["domain_max"])
        assert_bit_equal(direct, R.apply_cube_lut(x, oracle_lut["lut"], oracle_lut["domain_min"], oracle_lut["domain_max"]), name)
    with pytest.raises(ValueError):
        iv.VRGDG_LUTS._apply_cube_lut(x[..., :2], oracle_lut["lut"], oracle_lut["domain_min"], oracle_lut["domain_max"])


def test_lut_rgba_partial_strength_and_non_fp32_images(pkg, ops, dev):
    """apply_lut's strength blend runs over ALL channels (IV_Adjustments.py:355-359): alpha becomes a*(1-B) + a*B, which is
    not always a; and the reference grades non-fp32 images in fp32 and casts back (:293, :341-342)."""
    data, dlut = _lut_pair(ops, dev, "AMD_WarmFilm_25.cube")
    x = _rand((2, 17, 23, 4), 81, -0.05, 1.05)
    for s in (3.3, 7.0, 10.0, 0.0):
        assert_bit_equal(ops.lut3d(x.to(dev), dlut, s), R.apply_lut_with_strength(x, data, s), f"RGBA strength {s}")
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    with pytest.raises(ValueError):                                  # the reference evaluates float64 images in fp64: not offered
        iv.VRGDG_LUTS().apply_lut(_rand((1, 9, 11, 3), 82).to(torch.float64), "AMD_WarmFilm_25.cube", "auto", 10.0)
    xh = _rand((1, 9, 11, 3), 83).to(torch.float16)
    (out,) = iv.VRGDG_LUTS().apply_lut(xh, "AMD_WarmFilm_25.cube", "auto", 10.0)
    assert out.dtype == torch.float16
    assert_bit_equal(out, R.apply_lut_with_strength(xh, data, 10.0), "float16 image, full strength")


@pytest.mark.parametrize("n,dmin,dmax", [(32, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), (32, (-0.25, 0.0, 0.1), (1.5, 1.0, 0.9)), (25, (0.0, -0.1, 0.0), (1.0, 1.2, 2.0))])
def test_lut_photoshop_style_cubes_with_domain_lines(pkg, ops, dev, tmp_path, n, dmin, dmax):
    """The reference ships 25^3 / 32^3 cubes written by Photoshop with TITLE and DOMAIN_MIN / DOMAIN_MAX lines: a synthetic
    cube of those sizes through the parser, the record-table build and the kernels (stand-alone, fused, uint8), unit and
    non-unit domains (the IEEE division per axis)."""
    from comfyui_vrgamedevgirl_amd import cube
    g = torch.Generator().manual_seed(n * 7 + int(dmax[2] * 10))
    table = torch.rand((n, n, n, 3), generator=g)                       # [b][g][r][rgb]
    path = tmp_path / f"synthetic_{n}.cube"
    with open(path, "w") as fh:
        fh.write(f'TITLE "synthetic {n}"\n# comment\nLUT_3D_SIZE {n}\n')
        fh.write("DOMAIN_MIN " + " ".join(repr(float(v)) for v in dmin) + "\nDOMAIN_MAX " + " ".join(repr(float(v)) for v in dmax) + "\n\n")
        for b in range(n):
            for gg in range(n):
                for r in range(n):
                    v = table[b, gg, r]
                    fh.write(f"{v[0].item():.6f} {v[1].item():.6f} {v[2].item():.6f}\n")
    data = R.parse_cube_file(str(path))
    ours = cube.parse_cube_file(str(path))
    assert torch.equal(ours["lut"], data["lut"]) and torch.equal(ours["domain_min"], data["domain_min"]) and torch.equal(ours["domain_max"], data["domain_max"])
    dlut = ops.upload_lut(ours, dev)
    x = _rand((2, 61, 83, 3), n, -0.4, 1.7)
    x[0, 0, :6, :] = torch.tensor([0.0, 1.0, float(dmin[0]), float(dmax[1]), 0.5, 2.5]).view(6, 1)
    for s_ in (10.0, 4.2):
        assert_bit_equal(ops.lut3d(x.to(dev), dlut, s_), R.apply_lut_with_strength(x, data, s_), f"{n}^3 domain {dmin}..{dmax} strength {s_}")
    torch.manual_seed(3)
    fused = ops.fused_chain(x.to(dev), ops.ChainSpec(grain=(0.05, 0.5, 2), lut=(dlut, 10.0), sharpen=("unsharp", 0.6, False)))
    torch.manual_seed(3)
    o = R.fast_film_grain(x, 0.05, 0.5, 2, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    assert_bit_equal(fused, R.unsharp(R.apply_lut_with_strength(o, data, 10.0), 0.6, False), "fused chain on the synthetic cube")


def test_make_lut_node(pkg, dev, tmp_path, monkeypatch):
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    monkeypatch.setattr(iv, "LUTS_DIR", str(tmp_path))
    x = _rand((1, 20, 20, 3), 22)
    out, name, path = iv.VRGDG_MakeLUT().create_and_apply(x, "#0b1d51, #1f6aa5, #f3d27a", "palette", 9, "auto", 8.0)
    assert name == "0b1d51_1f6aa5_f3d27a_palette.cube" and os.path.isfile(path)
    from comfyui_vrgamedevgirl_amd import cube
    lut = {"lut": cube.build_palette_lut("#0b1d51, #1f6aa5, #f3d27a", 9), "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}
    assert_bit_equal(out, R.apply_lut_with_strength(x, lut, 8.0), "MakeLUT")


# ---------------------------------------------------------------------------------------- stencils
def test_stencils_match_reference_fixtures(ops, dev):
    z = _npz("stencil.npz")
    for tag in ("rand", "odd", "one", "row", "col", "c4", "const"):
        x = _t(z[f"{tag}.x"])
        xd = x.to(dev)
        for s in (0.5, 3.75):
            for zero in (False, True):
                assert_bit_equal(ops.stencil3x3(xd, "unsharp", s, zero), _t(z[f"{tag}.unsharp.{s}.{int(zero)}"]), f"unsharp {tag} {s} {zero}")
        for name, raster in (("laplacian", R.laplacian_zero_raster), ("sobel", R.sobel_zero_raster)):
            assert_bit_equal(ops.stencil3x3(xd, name, 0.8, False), _t(z[f"{tag}.{name}.0.8.0"]), f"{name} {tag} replicate")
            got = ops.stencil3x3(xd, name, 0.8, True)
            assert_bit_equal(got, raster(x, 0.8), f"{name} {tag} zero/raster")
            if x.shape[-1] == 3:   # the reference's own CPU conv2d output: laplacian bit-equal, sobel within one unit of the last place
                d = _unit_ulps(got, _t(z[f"{tag}.{name}.0.8.1"]))
                assert d <= (0.0 if name == "laplacian" else 1.0), (tag, name, d)
    # a video-sized frame: twelve rows of the reference's use_gpu=True outputs at 1080p (tests/golden/stencil_1080p_rows.npz)
    z = _npz("stencil_1080p_rows.npz")
    g = torch.Generator().manual_seed(int(z["seed"]))
    x = torch.rand(tuple(int(v) for v in z["shape"]), generator=g) * float(z["affine"][0]) + float(z["affine"][1])
    xd, rows = x.to(dev), z["rows"]
    assert np.array_equal(ops.stencil3x3(xd, "unsharp", 0.5, True).cpu().numpy()[0, rows], z["unsharp.0.5.1"])
    assert np.array_equal(ops.stencil3x3(xd, "laplacian", 0.8, True).cpu().numpy()[0, rows], z["laplacian.0.8.1"])
    sob = ops.stencil3x3(xd, "sobel", 0.8, True).cpu().numpy()[0, rows]
    d = np.abs(sob.astype(np.float64) - z["sobel.0.8.1"].astype(np.float64)) / 2.0 ** -23
    assert float(d.max()) <= 1.0 and int((d != 0).sum()) <= 69           # 45 of 69,120 elements, by one ulp(1.0)


def test_sharpen_nodes(pkg, dev):
    x = _rand((3, 37, 53, 3), 31, -0.1, 1.1)
    M = pkg.NODE_CLASS_MAPPINGS
    for key, fn, meth in (("FastUnsharpSharpen", R.unsharp, "apply_unsharp"), ("FastLaplacianSharpen", R.laplacian, "apply_laplacian"),
                          ("FastSobelSharpen", R.sobel, "apply_sobel")):
        (out,) = getattr(M[key](), meth)(x, 0.9, False)
        assert out.device.type == "cpu"
        assert_bit_equal(out, fn(x, 0.9, False), key)
    (out,) = M["FastUnsharpSharpen"]().apply_unsharp(x, 7.5, True)
    assert_bit_equal(out, R.unsharp(x, 7.5, True).contiguous(), "unsharp use_gpu")
    with pytest.raises(RuntimeError):
        M["FastLaplacianSharpen"]().apply_laplacian(_rand((1, 8, 8, 4), 1), 0.5, True)
    (out,) = M["FastUnsharpSharpen"]().apply_unsharp(_rand((1, 8, 8, 4), 1), 0.5, True)      # avg_pool path takes any C
    assert_bit_equal(out, R.unsharp(_rand((1, 8, 8, 4), 1), 0.5, True).contiguous(), "unsharp C=4")


# ---------------------------------------------------------------------------------------- colour match
# Two arithmetic policies (include/vrgdg_hip.h, enum vrg_cm_math):
#   "device" (default): held BIT-EQUAL to the device oracle -- oracle/restated.py (kornia's Lab formulas + nodes.py:91-124)
#       evaluated by torch on this GPU, which is what the reference executes here -- for the whole element-wise path: every
#       piece, the Lab image, and the apply pass given the same statistics.  The per-frame mean / std are torch fp32
#       reductions in the reference (their value depends on the batch shape, i.e. on the node's own batch_size widget);
#       ours are fp64-accumulated: compared in ulps against the fp64 statistics of the same Lab image.
#   "fast": table-driven powers; compared with both references in units of ulp(1.0) = 2^-23 of the [0,1] output.
# Budgets below are 2x the maxima measured on MI355X (profiles/r02_cm_parity.json); they replace round 1's 2e-5 (168 ulp).
ULP1 = 2.0 ** -23
CM_E2E_DEVICE_ULP = 32      # fp64-statistics variant vs the device oracle, end to end: statistics differences only (measured max 16.6; the chain: 21.1 against 2x this)
CM_FAST_VS_CPU_ULP = 64     # fast policy vs the reference on the CPU (Sleef powf, IEEE division) (measured max 26 on large frames)
CM_CROSS_REF_ULP = 64       # a policy against the OTHER reference (round 6: 192 -> 64 = 2x the measured maxima, profiles/r04_cm_test_measured.json:
                            # device policy vs the reference's CPU fixture 27.5, fast policy vs it 14.25, fast vs the device oracle 30.75, fast apply
                            # with CPU statistics 32.3, random sweep 48.4 per unit of stencil gain; the two references THEMSELVES differ by 30.75
                            # on a 270p frame -- north_star's "within 1 ulp" of "the reference" is met against the reference run on this GPU
                            # (0 ulp), not against its CPU run, by either policy)
MEASURED = {}


def _unit_ulps(got, want):
    return float((got.double().cpu() - want.double().cpu()).abs().max() / ULP1)


def _record(key, value):
    MEASURED[key] = max(MEASURED.get(key, 0.0), float(value))
    try:
        import json
        os.makedirs(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "cm_test_measured.json"), "w") as fh:
            json.dump(MEASURED, fh, indent=1, sort_keys=True)
    except OSError:
        pass


def _cm_image(shape, seed):
    x = _rand(shape, seed)
    x[0, 0, :8, :] = torch.tensor([0.0, 1.0, 0.04045, 0.040450003, 0.5, 1e-6, 0.0031308, 0.9999999]).view(8, 1)
    return x


def _dbg(pkg, x, op, y=0.0, triples=False):
    from comfyui_vrgamedevgirl_amd import _hip
    x = x.contiguous()
    out = torch.empty_like(x)
    n = x.numel() // (3 if triples else 1)
    _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(x), _hip.ptr(out), n, op, float(np.float32(y)), _hip.current_stream()), "vrg_debug_cm_math")
    return out


def _all_floats(lo, hi, dev):
    a, b = int(np.float32(lo).view(np.int32)), int(np.float32(hi).view(np.int32))
    return torch.arange(a, b + 1, dtype=torch.int32, device=dev).view(torch.float32)


def test_bench_two_rank_flow_on_one_gpu_with_gloo(dev):
    """bench.py's N > 1 path end to end on real hardware: self-launch under torch.distributed.run, two ranks, the reference
    frame's rows split and merged by the collective, frames sharded by absolute chunk index, max-over-ranks timing, one JSON
    line.  A 1-GPU box cannot host two RCCL ranks (duplicate device), so the ranks share cuda:0 and talk gloo
    (VRGDG_DIST_BACKEND): everything but the RCCL transport itself is the production code."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, VRGDG_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "8", "--steps", "1", "--warmup", "1",
                        "--cpu-frames", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dist_backend"] == "gloo" and len(line["per_rank_ms_per_step"]) == 2
    assert line["value"] > 0 and line["scaling"] == "weak" and line["config"]["frames_per_gpu"] == 8
    assert line["fast_variant"]["value"] > 0            # (which of the two is faster is not a property of two processes sharing one GPU)
    # N > 1: the second timed leg whose reference statistics cross the collective inside the steps, and rank 0's checks for any N
    leg = line["fp64_stats_leg"]
    assert leg["value"] > 0 and len(leg["per_rank_ms_per_step"]) == 2 and leg["reference_stats_allreduce_ms_per_step"] > 0 and leg["verified"] is True
    assert line["verified"] is True and "configs" not in line
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1        # rank 0 times it while rank 1 waits at the final barrier


def test_stats_allreduce_entry_point_over_rccl(pkg, ops, dev):
    """vrg_stats_allreduce with a real RCCL communicator (one rank: all a 1-GPU box can host; the N-rank algebra is the same
    code path and is covered on CPU by tests/test_sharding_gloo.py through the Python twin): equals sharding.allreduce_stats'
    arithmetic -- (n, n*mean) summed, mean_tot = sum / n_tot, M2 + n*(mean - mean_tot)^2 summed."""
    import ctypes as C
    from comfyui_vrgamedevgirl_amd import _hip
    import torch.distributed  # noqa: F401  (loads torch's bundled RCCL into the process)
    rccl = None
    for name in ("librccl.so", "librccl.so.1", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("no RCCL library to create a communicator with")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        x = _rand((3, 37, 29, 3), 77).to(dev)
        stats = ops.lab_stats(x)                                            # [3,3,3] fp64 triples
        want = stats.clone()
        n, mean, m2 = want[..., 0], want[..., 1], want[..., 2]
        mean_tot = (n * mean) / n
        delta = mean - mean_tot
        want = torch.stack([n, mean_tot, m2 + (n * delta) * delta], dim=-1)
        got = stats.clone()
        count = got.numel() // 3
        scratch = torch.empty(int(_hip.lib().vrg_stats_allreduce_scratch_bytes(count)) // 8, dtype=torch.float64, device=dev)
        _hip.check(_hip.lib().vrg_stats_allreduce(_hip.ptr(got), count, comm, _hip.ptr(scratch), _hip.current_stream()), "vrg_stats_allreduce")
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert _hip.lib().vrg_stats_allreduce(_hip.ptr(got), count, None, _hip.ptr(scratch), _hip.current_stream()) == 1      # null communicator
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_zero_border_stencils_against_the_device_conv2d(ops, dev):
    """use_gpu=True in the reference runs F.avg_pool2d / F.conv2d ON THE GPU (nodes.py:171, 248-257, 325-348).  avg_pool2d
    there is a plain raster sum: bit-equal.  conv2d is MIOpen: NOT a sum of the five (six) products in any order -- e.g. its
    corner output -1.7550163 for the two non-zero products -1 and -0.75501657 whose only fp32 sum is -1.7550166 (tools/
    conv_order_search.py finds no summation tree; the algorithm is Winograd-class) -- so it cannot be matched bit for bit;
    our raster-order sum is the correctly ordered fp32 sum (== torch's CPU conv2d on the bench box) and stays within
    4 ulp(1.0) of the device result (measured max 3.5 / 3.0)."""
    import torch.nn.functional as F
    x = _rand((2, 135, 240, 3), 141).to(dev)
    nchw = x.permute(0, 3, 1, 2).contiguous()
    blur = F.avg_pool2d(nchw, kernel_size=3, stride=1, padding=1)
    want = (nchw + 0.7 * (nchw - blur)).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    assert_bit_equal(ops.stencil3x3(x, "unsharp", 0.7, zero_border=True), want, "unsharp, zero border vs avg_pool2d on the device")
    kl = torch.tensor([[0, -1, 0], [-1, 4, -1], [0, -1, 0]], dtype=torch.float32, device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    want = (nchw + 0.7 * F.conv2d(nchw, kl, padding=1, groups=3)).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    d = _unit_ulps(ops.stencil3x3(x, "laplacian", 0.7, zero_border=True), want)
    _record("stencil.laplacian_zero_vs_device_conv2d", d)
    assert d <= 8, d
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32, device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    ky = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32, device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    gx, gy = F.conv2d(nchw, kx, padding=1, groups=3), F.conv2d(nchw, ky, padding=1, groups=3)
    want = (nchw + 0.7 * torch.sqrt(gx ** 2 + gy ** 2 + 1e-6)).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    d = _unit_ulps(ops.stencil3x3(x, "sobel", 0.7, zero_border=True), want)
    _record("stencil.sobel_zero_vs_device_conv2d", d)
    assert d <= 8, d


@pytest.mark.parametrize("y,lo,hi", [(2.4, 2.0 ** -12, 2.0), (1 / 2.4, 0.0031308, 4.0), (1 / 3.0, 0.008856, 4.0)])
def test_device_math_pow_is_torch_pow_for_every_input_of_the_domain(pkg, dev, y, lo, hi):
    """ocml powf as this hipcc links it == torch.pow on the device (ocml powf as libtorch_hip.so carries it), for EVERY fp32
    base the Lab transforms can feed it: 7.4e7 .. 1.1e8 inputs per exponent."""
    x = _all_floats(lo, hi, dev)
    want = torch.pow(x, y)
    assert torch.equal(_dbg(pkg, x, 0, y), want), "__ocml_pow_f32 linked by hipcc vs torch.pow"
    assert torch.equal(_dbg(pkg, x, 9, y), want), "dev_pow (ocml powf without its special-case scaffolding) vs torch.pow"
    # the flavour the Lab transforms instantiate for this exponent (selects dropped where a range argument proves them inert):
    # the whole domain again, then every 7th fp32 from 2^-20 up to FLT_MAX (1.8e8 bases), +Inf and NaN
    flavour = 10 if y > 1 else 11
    assert torch.equal(_dbg(pkg, x, flavour, y), want), "dev_pow flavour of the Lab transforms vs torch.pow"
    bits = torch.arange(int(np.float32(2.0 ** -20).view(np.uint32)), 0x7f800000 + 1, 7, dtype=torch.int64, device=dev).to(torch.int32)
    span = torch.cat([bits.view(torch.float32), torch.tensor([float("inf"), float("nan"), 3.4028235e38, 1.14e16, 1.15e16], device=dev)])
    a, b = _dbg(pkg, span, flavour, y), torch.pow(span, y)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    sp = torch.tensor([0.0, -0.0, -0.5, -1.0, 1.0, float("inf"), float("nan"), 1e-38, 1e-45, -1e-30, 3.0e38], device=dev)
    a, b = _dbg(pkg, sp, 0, y), torch.pow(sp, y)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    # dev_pow's contract is x > 0: normal, huge, subnormal, +Inf, NaN bases; results up to overflow and down to underflow
    sp = torch.tensor([1.0, float("inf"), float("nan"), 1e-38, 1e-45, 3.0e38, 1.17549435e-38, 2.0 ** -126, 2.0 ** 100, 1e-20, 0.99999994, 1.0000001], device=dev)
    for yy in (y, 40.0 * y, -y):
        a, b = _dbg(pkg, sp, 9, yy), torch.pow(sp, yy)
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (yy, a, b)
    g = torch.Generator(device=dev).manual_seed(9)
    wide = torch.exp(torch.rand(1 << 22, generator=g, device=dev) * 160.0 - 80.0)          # 1e-35 .. 1e35
    assert torch.equal(_dbg(pkg, wide, 9, y), torch.pow(wide, y))


@pytest.mark.parametrize("op,y,lo,hi", [(12, 2.4, 0.0625, 2.0), (13, 1 / 2.4, 0.0031308, 4.0), (14, 1 / 3.0, 0.008856, 4.0)])
def test_ziv_tested_power_is_torch_pow_for_every_input_of_its_domain(pkg, dev, op, y, lo, hi):
    """dev_pow_ziv as the Lab transforms call it (table log + ocml's own product / exp / final FMA, with a rounding test that sends
    a lane to the transcription of ocml powf whenever the cheaper logarithm could change a rounding): equal to torch.pow on the
    device for EVERY fp32 base of the call site's fast-path domain -- equality by enumeration there -- and, outside it, on every
    7th fp32 from 2^-20 to FLT_MAX and the specials, where every lane takes the transcription."""
    yf = float(np.float32(y))
    x = _all_floats(lo, hi, dev)
    assert torch.equal(_dbg(pkg, x, op, y), torch.pow(x, yf)), "dev_pow_ziv vs torch.pow inside the fast-path domain"
    slow = float(_dbg(pkg, x, 15, y).mean())
    _record(f"ziv.fallback_fraction_op{op}", slow)
    assert slow < 0.001, slow            # measured 0.02-0.03 % (0.05-0.15 % before the per-index half-width): the test is the cheap path, not the exception
    bits = torch.arange(int(np.float32(2.0 ** -20).view(np.uint32)), 0x7f800000 + 1, 7, dtype=torch.int64, device=dev).to(torch.int32)
    # (the callers clamp the base from below, and the flavours behind the test take x >= 2^-20, +Inf or NaN)
    span = torch.cat([bits.view(torch.float32), torch.tensor([float("inf"), float("nan"), 3.4028235e38, lo, hi], device=dev)])
    a, b = _dbg(pkg, span, op, y), torch.pow(span, yf)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


def test_ziv_interval_covers_the_distance_between_the_two_logarithms(pkg, dev):
    """The half-width of dev_pow_ziv's rounding test: per table index the table's fourth word A_j, and ziv_delta()'s constant relative to
    max(|e ln2|, |ln x|) -- both 1.25 x the exhaustively measured maxima of |ln x (ocml's epln) - ln x (table)| (tools/ziv_calibration.json).
    Re-measure them over every fp32 of [0.0031308, 4] on this device and hold the margin, index by index."""
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import importlib
    import re
    acc = importlib.import_module("ziv_log_accuracy").measure(0.0031308, 4.0, dev)
    for k in ("ocml_rel", "table_rel", "distance_rel_to_max_eln2_lnx", "distance_abs"):
        _record("ziv.log_" + k, acc[k])
    per = importlib.import_module("ziv_per_index").measure(dev)
    inc = open(os.path.join(PKG_DIR, "csrc", "vrg_ziv_log_table.inc")).read()
    words = re.findall(r"\{0x([0-9a-f]{8})u, 0x([0-9a-f]{8})u, 0x([0-9a-f]{8})u, 0x([0-9a-f]{8})u\}", inc)
    assert len(words) == 128
    rel_const = float(np.array([0x2e06f428], dtype=np.uint32).view(np.float32)[0])      # VRG_ZIV_REL_BITS
    A = np.array([int(w[3], 16) for w in words], dtype=np.uint32).view(np.float32).astype(np.float64) * rel_const       # the fourth word holds A_j / C
    measured = np.array(per["per_j_abs_max"])
    assert (measured > 0).all() and (measured * 1.2 <= A).all(), (measured * 1.2 / A).max()
    assert A.max() <= 2.0 ** -36.0 and np.median(A) <= 2.0 ** -38.5                      # what the per-index bound buys over the global 2^-35.7
    src = open(os.path.join(PKG_DIR, "csrc", "vrg_pixel_math.hpp")).read()
    assert "0x2e06f428u" in src
    assert per["global_rel"] * 1.2 <= rel_const, per["global_rel_log2"]
    assert abs(acc["distance_rel_to_max_eln2_lnx"] - per["global_rel"]) <= 1e-3 * per["global_rel"]      # the two tools agree
    assert acc["table_rel"] < acc["ocml_rel"] * 1.05               # the table tracks ocml's logarithm and is not less accurate than it


def test_device_math_divisions_are_torch_divisions(pkg, dev):
    """tensor / python scalar on the device is x * fl32(1 / c) with the reciprocal of the Python double (ATen
    BinaryDivTrueKernel) -- NOT the IEEE quotient and, for 1.055, not x * (1.0f / 1.055f) either; tensor / tensor is IEEE."""
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.cat([torch.rand(1 << 22, generator=g, device=dev) * 2 - 0.5, _all_floats(2.0 ** -10, 2.0 ** -9, dev),
                   torch.randn(1 << 20, generator=g, device=dev) * 100])
    xn = x.cpu().numpy()
    for c in (1.055, 12.92, 116.0, 500.0, 200.0, 7.787):
        assert torch.equal((x / c).cpu(), torch.from_numpy(xn * np.float32(1.0 / c))), c
    assert not torch.equal((x / 1.055).cpu(), torch.from_numpy(xn / np.float32(1.055)))
    for c in (0.95047, 1.08883, 0.40001):
        want = x / torch.full((1,), c, dtype=torch.float32, device=dev)
        assert torch.equal(want.cpu(), torch.from_numpy(xn / np.float32(c))), c
        assert torch.equal(_dbg(pkg, x, 2, c), want), c
    assert torch.equal(torch.pow(x, 3.0).cpu(), torch.from_numpy((xn * xn) * xn))        # pow(., 3.0) is (x*x)*x


def test_unscaled_sigma_division_is_the_division(pkg, dev):
    """(lab - mean) / std of the colour transfer (nodes.py:112) is an IEEE quotient on the device.  The apply kernels evaluate it as the
    backend's own division sequence WITHOUT its scalings and fix-up -- five FMAs around the frame's refined reciprocal -- wherever
    sigma_recip()'s per-frame and the per-pixel condition hold (vrg_pixel_math.hpp): equal to torch's quotient bit for bit on 2^27 triples
    of Lab-like values, on operands spread over the whole range the conditions admit (and beyond it: there the flag must be 0 or the
    values still equal), and at the edges."""
    g = torch.Generator(device=dev).manual_seed(77)

    def check(lab, mu, sd, min_ok):
        out = _dbg(pkg, torch.stack([lab, mu, sd], dim=-1), 20, triples=True)
        fast, ieee, ok = out[..., 0], out[..., 1], out[..., 2] == 1
        want = (lab - mu) / sd
        assert torch.equal(torch.isnan(ieee), torch.isnan(want)) and torch.equal(torch.nan_to_num(ieee, nan=-7.0).view(torch.int32), torch.nan_to_num(want, nan=-7.0).view(torch.int32))
        assert float(ok.float().mean()) >= min_ok, float(ok.float().mean())
        assert not torch.isnan(want[ok]).any()
        bad = fast[ok].view(torch.int32) != want[ok].view(torch.int32)
        assert not bool(bad.any()), (int(bad.sum()), lab[ok][bad][:4], mu[ok][bad][:4], sd[ok][bad][:4])

    n = 1 << 27
    check((torch.rand(n, generator=g, device=dev) - 0.4) * 250.0, (torch.rand(n, generator=g, device=dev) - 0.5) * 200.0,
          torch.exp(torch.rand(n, generator=g, device=dev) * 16.0 - 11.5), 0.999)                     # Lab-like; std + 1e-5 in 1e-5 .. 90
    n = 1 << 25

    def spread(lo, hi):      # +-2^U(lo, hi)
        m = torch.exp2(torch.rand(n, generator=g, device=dev) * (hi - lo) + lo)
        return torch.where(torch.rand(n, generator=g, device=dev) < 0.5, -m, m)
    check(spread(-70, 45), spread(-65, 42), spread(-42, 32).abs(), 0.5)                             # across and beyond the conditions
    mu = spread(-60, 40)
    k = torch.randint(-6, 7, (n,), generator=g, device=dev, dtype=torch.int32)
    near = (mu.view(torch.int32) + k).view(torch.float32)                                          # lab within a few ulp of the mean: d tiny or +0
    check(near, mu, spread(-40, 30).abs(), 0.9)
    edge = torch.tensor([[1.0, 1.0, 2.0 ** -40], [1.0, 2.0 ** -60, 2.0 ** 30], [-0.0, 2.0 ** -60, 1.0], [2.0 ** 39, -2.0 ** 38, 2.0 ** -40],
                         [2.0 ** 40, 1.0, 1.0], [float("inf"), 1.0, 1.0], [float("nan"), 1.0, 1.0], [1.0, 0.0, 1.0], [0.0, 0.0, 1.0], [-0.0, 0.0, 1.0],
                         [1e-45, 1.0, 1.0], [3.0, 3.0, 1e-5], [50.0, 50.000004, 1e-5], [1.0, 1.0, 2.0 ** -41], [1.0, 1.0, 2.0 ** 31]], device=dev)
    check(edge[:, 0].contiguous(), edge[:, 1].contiguous(), edge[:, 2].contiguous(), 0.3)


def test_lab_transforms_bit_equal_device_oracle(pkg, dev):
    x = _cm_image((3, 270, 480, 3), 11).to(dev)
    want_lab = R.kornia_rgb_to_lab(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    assert_bit_equal(_dbg(pkg, x, 5, triples=True), want_lab, "rgb_to_lab, device policy vs torch on the device")
    g = torch.Generator(device=dev).manual_seed(5)
    lab = want_lab.clone()
    lab[1:] = lab[1:] * (1.0 + 0.3 * torch.randn(lab[1:].shape, generator=g, device=dev))      # out of gamut: clamps, negative fz
    want_rgb = R.kornia_lab_to_rgb(lab.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    assert_bit_equal(_dbg(pkg, lab, 6, triples=True), want_rgb, "lab_to_rgb, device policy vs torch on the device")
    # the fast policy: distance to the same oracle, and to the reference on the CPU, in ulp(1.0) of the RGB output
    fast_rgb = _dbg(pkg, lab, 8, triples=True)
    cpu_rgb = R.kornia_lab_to_rgb(lab.permute(0, 3, 1, 2).cpu()).permute(0, 2, 3, 1).contiguous()
    _record("lab_to_rgb.fast_vs_device_oracle", _unit_ulps(fast_rgb, want_rgb))
    _record("lab_to_rgb.fast_vs_cpu_oracle", _unit_ulps(fast_rgb, cpu_rgb))
    _record("lab_to_rgb.cpu_oracle_vs_device_oracle", _unit_ulps(cpu_rgb, want_rgb))
    assert _unit_ulps(fast_rgb, want_rgb) <= 2 * 130 and _unit_ulps(fast_rgb, cpu_rgb) <= 2 * 130


def _same_with_nans(got, want, what):
    got, want = got.detach().cpu(), want.detach().cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what}: NaN positions differ"
    assert_bit_equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0), what)


def test_colour_match_special_values_like_the_device_oracle(pkg, ops, dev):
    """Out-of-range and non-finite pixels: negative, > 1, huge, +-Inf, NaN, subnormal.  kornia's where() evaluates both
    branches on every element and the clamps propagate NaN; the device policy must give what torch gives on the GPU."""
    vals = [0.0, -0.0, -0.3, -1e-30, 1e-40, 1.17549435e-38, 0.04045, 0.040450003, 0.5, 1.0, 1.5, 7.0, 3.0e4, 1e30, float("inf"), float("-inf"),
            float("nan")]
    n = len(vals)
    grid = torch.tensor(vals, dtype=torch.float32)
    x = torch.stack(torch.meshgrid(grid, grid, grid, indexing="ij"), dim=-1).reshape(1, n, n * n, 3).contiguous().to(dev)
    want = R.kornia_rgb_to_lab(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    _same_with_nans(_dbg(pkg, x, 5, triples=True), want, "rgb_to_lab on special values")
    lab = torch.stack(torch.meshgrid(torch.tensor([0.0, -20.0, 50.0, 100.0, 150.0, 1e6, float("inf"), float("nan")]),
                                     torch.tensor([0.0, -200.0, 120.0, 1e5, float("-inf"), float("nan")]),
                                     torch.tensor([0.0, -150.0, 90.0, 300.0, float("inf"), float("nan")]), indexing="ij"), dim=-1)
    lab = lab.reshape(1, 8, 36, 3).contiguous().to(dev)
    _same_with_nans(_dbg(pkg, lab, 6, triples=True), R.kornia_lab_to_rgb(lab.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous(),
                    "lab_to_rgb on special values")
    # whole colour match on finite out-of-range frames (element-wise path with the oracle's statistics), and a frame with a NaN
    # pixel: NaN statistics -> an all-NaN frame on both sides
    y = _rand((3, 40, 56, 3), 91, -0.5, 1.8)
    y[2, 3, 4, 1] = float("nan")
    yd, ref = y.to(dev), (_rand((1, 16, 16, 3), 92) * 0.7 + 0.1).to(dev)
    ims, _ = _stats_per_frame(yd, 1)
    rms, _ = _stats_per_frame(ref, 1)
    _same_with_nans(ops.colormatch_apply(yd, ims, rms, 0.6, cm_math="device"), R.color_match(yd, ref, 0.6, 1), "apply on out-of-range frames")
    got = ops.color_match(yd, ref, 0.6)
    assert bool(torch.isnan(got[2]).all()) and not bool(torch.isnan(got[:2]).any())


def test_lab_image_of_the_statistics_pass_bit_equal_device_oracle(ops, dev):
    """pass 1 stores the Lab image pass 2 reads: general kernel, and LUT -> Lab, against torch on the device"""
    data, dlut = _lut_pair(ops, dev, "AMD_WarmFilm_25.cube")
    x = _cm_image((2, 96, 128, 3), 12).to(dev)
    lab = torch.empty_like(x)
    ops.chain_stats(x, ops.ChainSpec(), lab_out=lab)
    assert_bit_equal(lab, R.kornia_rgb_to_lab(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous(), "Lab image")
    ops.chain_stats(x, ops.ChainSpec(lut=(dlut, 10.0)), lab_out=lab)
    y = ops.lut3d(x, dlut, 10.0)
    assert_bit_equal(lab, R.kornia_rgb_to_lab(y.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous(), "Lab image after the LUT")


def _stats_per_frame(x_dev, batch_size=1):
    """mean / std+1e-5 the way the reference forms them: torch reductions per batch_size chunk (their fp32 value depends on
    the chunk shape).  Returns [F,3,2] fp32 and the NCHW Lab image."""
    lab = R.kornia_rgb_to_lab(x_dev.permute(0, 3, 1, 2))
    ms = []
    for i in range(0, lab.shape[0], batch_size):
        mu, sd = R.lab_stats(lab[i:i + batch_size])
        ms.append(torch.stack([mu.flatten(1), sd.flatten(1)], dim=-1))
    return torch.cat(ms, dim=0).contiguous(), lab


@pytest.mark.parametrize("k,bs", [(1.0, 1), (0.35, 1), (0.8, 2)])
def test_colour_match_apply_bit_equal_device_oracle_with_injected_statistics(ops, dev, k, bs):
    """The reference's own device statistics fed to the HIP apply pass: what is left is the element-wise path
    ((lab - mu) / sigma * sigma_ref + mu_ref, blend, Lab -> RGB, clamp) -- bit-equal to the device oracle."""
    x = _cm_image((4, 135, 240, 3), 43).to(dev)
    ref = (_rand((1, 64, 80, 3), 44) * 0.7 + 0.1).to(dev)
    ims, _ = _stats_per_frame(x, bs)
    rms, _ = _stats_per_frame(ref, 1)
    want = R.color_match(x, ref, k, bs)                                  # device oracle end to end, same chunking
    assert_bit_equal(ops.colormatch_apply(x, ims, rms, k, cm_math="device"), want, "apply pass with the oracle's statistics")
    # and the other direction: OUR statistics fed to the oracle's element-wise path == our end-to-end result
    oms, orms = ops.finalize_stats(ops.lab_stats(x)), ops.finalize_stats(ops.lab_stats(ref))
    lab = R.kornia_rgb_to_lab(x.permute(0, 3, 1, 2))
    alt = R.color_match_apply(lab, oms[..., 0].view(4, 3, 1, 1), oms[..., 1].view(4, 3, 1, 1), orms[..., 0].view(1, 3, 1, 1),
                              orms[..., 1].view(1, 3, 1, 1), k).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    assert_bit_equal(ops.color_match(x, ref, k, cm_stats="fp64"), alt, "whole colour match (fp64 statistics) vs the oracle evaluated with them")
    # and with the device statistics (the default) the whole thing IS the device oracle
    assert_bit_equal(ops.color_match(x, ref, k, cm_chunk=bs), want, "whole colour match, device statistics")


def test_lab_statistics_against_fp64_and_the_device_reductions(ops, dev):
    x = _cm_image((3, 96, 128, 3), 41)
    x[2] = 0.25                                             # constant frame: sigma must be exactly 0
    xd = x.to(dev)
    stats = ops.lab_stats(xd)
    lab = R.kornia_rgb_to_lab(xd.permute(0, 3, 1, 2)).double()           # the (bit-equal) Lab image, then fp64 statistics
    n = 96 * 128
    mu64, var64 = lab.mean(dim=[2, 3]), lab.var(dim=[2, 3], unbiased=True)
    s = stats.cpu().numpy()
    assert np.array_equal(s[..., 0], np.full((3, 3), float(n)))
    assert np.max(np.abs(s[..., 1] - mu64.cpu().numpy()) / np.maximum(np.abs(mu64.cpu().numpy()), 1e-30)) < 1e-12
    assert np.all(s[2, :, 2] == 0.0)
    assert np.max(np.abs(s[:2, :, 2] / (n - 1) - var64.cpu().numpy()[:2]) / var64.cpu().numpy()[:2]) < 1e-11
    ms = ops.finalize_stats(stats)
    got_sd = np.sqrt(s[..., 2] / (n - 1))
    assert np.array_equal(ms.cpu().numpy()[..., 0], s[..., 1].astype(np.float32))
    assert np.array_equal(ms.cpu().numpy()[..., 1], got_sd.astype(np.float32) + np.float32(1e-5))
    # in ulps: ours (fp64 sums, one rounding) and torch's fp32 reductions on the device, both against the fp64 statistics
    t_ms, _ = _stats_per_frame(xd[:2], 1)
    sd64 = var64.sqrt() + float(np.float32(1e-5))

    def eps(a, b):
        return float(((a.double() - b).abs() / (b.abs() * ULP1)).max())
    ours = (eps(ms[:2, :, 0], mu64[:2]), eps(ms[:2, :, 1], sd64[:2]))
    theirs = (eps(t_ms[..., 0], mu64[:2]), eps(t_ms[..., 1], sd64[:2]))
    _record("stats.ours_vs_fp64.mean_eps", ours[0]); _record("stats.ours_vs_fp64.std_eps", ours[1])
    _record("stats.torch_device_vs_fp64.mean_eps", theirs[0]); _record("stats.torch_device_vs_fp64.std_eps", theirs[1])
    assert ours[0] <= 0.5 and ours[1] <= 1.0                 # mean: one rounding; std: sqrt rounded to fp32, then + 1e-5f in fp32
    # deterministic and independent of how many frames one call covers
    again = ops.lab_stats(xd[1:2]).cpu().numpy()
    assert np.array_equal(again[0], s[1])
    assert torch.equal(ops.lab_stats(xd, cm_math="fast")[..., 0], stats[..., 0])


def _torch_reductions(lab_nhwc, chunk):
    """(mean, std + 1e-5) as the reference takes them: torch reductions over dims [2,3] of the contiguous NCHW tensor of each call."""
    ms = []
    for i in range(0, lab_nhwc.shape[0], chunk):
        t = lab_nhwc[i:i + chunk].permute(0, 3, 1, 2).contiguous()
        ms.append(torch.stack([t.mean(dim=[2, 3]), t.std(dim=[2, 3]) + 1e-5], dim=-1))
    return torch.cat(ms, dim=0)


def _same_bits_or_nan(a, b):
    a, b = a.cpu(), b.cpu()
    return bool(((a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))).all())


@pytest.mark.parametrize("F,H,W,chunk", [
    (2, 2160, 3840, 1), (4, 1080, 1920, 2), (7, 540, 960, 3), (9, 270, 480, 4), (16, 64, 64, 16), (5, 72, 120, 2), (3, 40, 40, 1),
    # planes that start at unaligned addresses in the reference's tensor (H*W % 4 != 0), the vectorisation threshold, tiny frames
    (5, 15, 15, 2), (3, 15, 15, 1), (7, 11, 13, 3), (4, 5, 7, 2), (1, 5, 7, 1), (3, 1, 127, 1), (3, 1, 128, 3), (2, 1, 129, 2), (2, 1, 130, 1),
    (4, 31, 33, 4), (2, 255, 257, 1), (3, 1023, 1025, 3), (2, 90, 91, 2), (1, 1, 1, 1), (2, 1, 2, 1), (3, 2, 2, 2), (6, 23, 29, 5), (40, 16, 16, 40)])
def test_device_statistics_are_torch_reductions_bit_for_bit(ops, dev, F, H, W, chunk):
    """vrg_lab_stats_torch_f32 against `mean(dim=[2,3])` / `std(dim=[2,3]) + 1e-5` evaluated by torch on this GPU, per call of
    `chunk` frames (the geometry -- and with it the fp32 value -- changes with the call size and the frame size)."""
    g = torch.Generator().manual_seed(F * 1000 + H + W + chunk)
    lab = (torch.rand((F, H, W, 3), generator=g) * torch.tensor([100.0, 120.0, 120.0]) + torch.tensor([0.0, -60.0, -60.0])).to(dev)
    got = ops.lab_stats_device(lab, chunk)
    want = _torch_reductions(lab, chunk)
    assert _same_bits_or_nan(got, want), (got - want).abs().max()
    if F > chunk:       # explicit call sizes, in another split
        sizes = [1] + [chunk] * ((F - 1) // chunk) + ([(F - 1) % chunk] if (F - 1) % chunk else [])
        got2 = ops.lab_stats_device(lab, sizes)
        want2 = torch.cat([_torch_reductions(lab[:1], 1), _torch_reductions(lab[1:], chunk)], dim=0)
        assert _same_bits_or_nan(got2, want2)


def test_device_statistics_of_a_call_beyond_32bit_indexing(ops, dev):
    """A reduction call whose tensor has more than 2^29 elements (batch_size >= 22 at 4K): TensorIterator::with_32bit_indexing splits
    the OUTPUTS depth first until every piece is 32-bit indexable, and each piece is reduced with its own geometry and mean factor
    (ts_launch_planes).  22 x 4K frames = 66 planes -> 33 + 33."""
    F, H, W = 22, 2160, 3840
    g = torch.Generator(device=dev).manual_seed(77)
    lab = torch.empty((F, H, W, 3), device=dev)
    for i in range(F):
        lab[i] = torch.rand((H, W, 3), generator=g, device=dev) * 100 - 35
    got = ops.lab_stats_device(lab, F)
    want = _torch_reductions(lab, F)
    assert _same_bits_or_nan(got, want), (got - want).abs().max()
    # and a size where the split leaves planes of one frame in different pieces: 23 frames = 69 planes -> 34 + 35
    lab23 = torch.cat([lab, lab[:1] * 0.5], dim=0)
    del lab
    assert _same_bits_or_nan(ops.lab_stats_device(lab23, 23), _torch_reductions(lab23, 23))


def test_device_statistics_selfcheck_is_silent_on_this_torch_build(ops, dev):
    import warnings
    ops._TS_CHECKED.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ops.lab_stats_device(torch.rand((2, 8, 8, 3), device=dev), 1)
    assert dev.index in ops._TS_CHECKED or 0 in ops._TS_CHECKED


def test_device_statistics_numpy_restatement_equals_torch_on_this_gpu(dev):
    """oracle/torch_device_reduce.py (the CPU restatement the -m 'not gpu' suite checks against the committed ground truth) against
    torch on the device, on fresh data."""
    from oracle import torch_device_reduce as TR
    for seed, (b, H, W) in enumerate([(1, 135, 240), (2, 72, 120), (5, 33, 47), (3, 15, 15)]):
        x = _rand((b, 3, H, W), 900 + seed) * 100 - 30
        xd = x.to(dev)
        m, s = TR.mean_std(x.numpy())
        assert np.array_equal(m.view(np.int32), xd.mean(dim=[2, 3]).cpu().numpy().view(np.int32))
        assert np.array_equal(s.view(np.int32), xd.std(dim=[2, 3]).cpu().numpy().view(np.int32))


@pytest.mark.parametrize("F,H,W,bs,n_ref,k", [(5, 135, 240, 2, 1, 1.0), (6, 72, 120, 4, 1, 0.35), (4, 270, 480, 1, 1, 0.8), (3, 45, 51, 3, 3, 0.6),
                                            (7, 30, 50, 7, 1, 1.0), (2, 1080, 1920, 1, 1, 1.0), (3, 2160, 3840, 2, 1, 1.0)])
def test_colour_match_node_is_the_device_oracle_for_every_batch_size(pkg, dev, F, H, W, bs, n_ref, k):
    """The statistics depend on batch_size in the reference (one reduction call per chunk); so do ours: node == device oracle,
    bit for bit, for every chunking including ragged last chunks and odd frame sizes."""
    x, ref = _cm_image((F, H, W, 3), 71), _rand((n_ref, 37, 53, 3), 72) * 0.8 + 0.1
    (out,) = pkg.NODE_CLASS_MAPPINGS["ColorMatchToReference"]().match_color(x, ref, k, bs)
    assert_bit_equal(out, R.color_match(x.to(dev), ref.to(dev), k, bs), f"batch_size {bs}")


@pytest.mark.parametrize("shape,ref_shape,k,bs", [((4, 135, 240, 3), (1, 64, 80, 3), 1.0, 1), ((3, 270, 480, 3), (1, 300, 400, 3), 0.35, 1),
                                                  ((4, 96, 128, 3), (4, 50, 60, 3), 0.8, 4)])
def test_colour_match_end_to_end_ulp_budget_vs_device_oracle(pkg, ops, dev, shape, ref_shape, k, bs):
    """Node end to end (device policy) against the device oracle: BIT-EQUAL -- element-wise path and statistics (torch's own
    reductions per batch_size call, replayed).  The fp64-statistics variant keeps its measured distance (statistics only)."""
    x, ref = _cm_image(shape, 31), _rand(ref_shape, 32) * 0.7 + 0.1
    node = pkg.NODE_CLASS_MAPPINGS["ColorMatchToReference"]()
    (out,) = node.match_color(x, ref, k, bs)
    want = R.color_match(x.to(dev), ref.to(dev), k, bs).cpu()
    assert_bit_equal(out, want, "ColorMatchToReference vs the reference's formulas evaluated by torch on this GPU")
    if ref_shape[0] == 1:
        f64 = ops.color_match(x.to(dev), ref.to(dev), k, cm_stats="fp64").cpu()
        d = _unit_ulps(f64, want)
        _record(f"e2e.device_fp64stats_vs_device_oracle.{shape[1]}p", d)
        assert d <= CM_E2E_DEVICE_ULP, d
    cpu = R.color_match(x, ref, k, bs)
    _record(f"e2e.cpu_oracle_vs_device_oracle.{shape[1]}p", _unit_ulps(cpu, want))
    _record(f"e2e.device_vs_cpu_oracle.{shape[1]}p", _unit_ulps(out, cpu))
    fast = ops.color_match(x.to(dev), ref.to(dev), k, cm_math="fast").cpu() if ref_shape[0] == 1 else None
    if fast is not None:
        _record(f"e2e.fast_vs_cpu_oracle.{shape[1]}p", _unit_ulps(fast, cpu))
        _record(f"e2e.fast_vs_device_oracle.{shape[1]}p", _unit_ulps(fast, want))
        assert _unit_ulps(fast, cpu) <= CM_FAST_VS_CPU_ULP and _unit_ulps(fast, want) <= CM_CROSS_REF_ULP


def test_colour_match_node_against_fixtures_and_truth(pkg, dev):
    """The reference-generated CPU fixtures (tests/golden/colormatch.npz: the reference's match_color control flow run on
    the CPU with the restated kornia): both policies stay within the cross-reference band, no further from the fp64 truth
    than the reference's own fp32 result, and keep the control flow (chunking, reference batch, errors)."""
    z = _npz("colormatch.npz")
    x, ref1, ref4 = _t(z["x"]), _t(z["ref1"]), _t(z["ref4"])
    node = pkg.NODE_CLASS_MAPPINGS["ColorMatchToReference"]()
    tol = CM_CROSS_REF_ULP * ULP1
    for key, ref, k, bs in (("out.ref1.k1.bs1", ref1, 1.0, 1), ("out.ref1.k0.35.bs3", ref1, 0.35, 3), ("out.ref4.k0.8.bs4", ref4, 0.8, 4)):
        (out,) = node.match_color(x, ref, k, bs)
        truth = truth64.color_match64(x.numpy(), ref.numpy(), k)
        gold = z[key]
        err_ours = np.abs(out.numpy().astype(np.float64) - truth).reshape(4, -1).max(axis=1)
        err_ref = np.abs(gold.astype(np.float64) - truth).reshape(4, -1).max(axis=1)
        # bar: no further from the fp64 truth than the reference's own fp32 result, + tol -- on the
        # well-conditioned frames.  Frame 3 is constant (sigma = 0): there the reference's fp32 mean of N equal values is not that
        # value and (lab - mean) / 1e-5 amplifies the difference -- its output is chaotic (up to 0.6 off the truth on the CPU), and
        # differently so on every device.  The node reproduces what the reference does on THIS device, chaos included:
        assert np.all(err_ours[:3] <= err_ref[:3] + tol), (key, err_ours, err_ref)
        assert_bit_equal(out, R.color_match(x.to(dev), ref.to(dev), k, bs), key + " vs the device oracle (constant frame included)")
        d = float(np.abs(out.numpy()[:3] - gold[:3]).max() / ULP1)
        _record("fixtures.device_vs_cpu_fixture", d)
        assert d <= CM_CROSS_REF_ULP, (key, d)
        if ref.shape[0] == 1:          # the fp64 statistics return the exact limit on the constant frame
            from comfyui_vrgamedevgirl_amd import ops as _ops
            f64 = _ops.color_match(x.to(dev), ref.to(dev), k, cm_stats="fp64").cpu().numpy().astype(np.float64)
            assert np.abs(f64 - truth).reshape(4, -1).max(axis=1)[3] <= tol
    with pytest.raises(RuntimeError):
        node.match_color(x, ref4[:3], 1.0, 4)                # reference batch neither 1 nor the chunk size
    # a chunk of ONE frame against several references broadcasts the other way: n_ref output frames per input frame
    (out,) = node.match_color(x[:2], ref4[:3], 0.7, 1)
    want = R.color_match(x[:2].to(dev), ref4[:3].to(dev), 0.7, 1).cpu()
    assert out.shape == want.shape == (6,) + tuple(x.shape[1:])
    assert_bit_equal(out, want, "one-frame chunks broadcast against three references")
    (out,) = node.match_color(x, ref4[:3], 0.7, 3)           # chunks of 3 and 1: 3 + 3 frames
    want = R.color_match(x.to(dev), ref4[:3].to(dev), 0.7, 3).cpu()
    assert out.shape == want.shape == (6,) + tuple(x.shape[1:])
    assert_bit_equal(out, want, "chunks of 3 and 1 against three references (constant frame included)")
    from comfyui_vrgamedevgirl_amd import ops
    xd = x.to(dev)
    for mode in ("device", "fast"):
        a = ops.color_match(xd, ref1.to(dev), 0.35, cm_math=mode)              # Lab cached between the passes (default)
        b = ops.color_match(xd, ref1.to(dev), 0.35, cache_lab=False, cm_math=mode)
        assert torch.equal(a, b), mode
    fast = ops.color_match(xd, ref1.to(dev), 1.0, cm_math="fast").cpu().numpy()
    d = float(np.abs(fast[:3] - z["out.ref1.k1.bs1"][:3]).max() / ULP1)
    _record("fixtures.fast_vs_cpu_fixture", d)
    assert d <= CM_CROSS_REF_ULP


def test_colour_match_fast_policy_apply_with_cpu_oracle_statistics(ops, dev):
    """Fast policy, statistics injected from the reference on the CPU: the element-wise distance to the CPU reference
    (pow_pos vs Sleef powf; the divisions are IEEE on both sides)."""
    x = _rand((2, 40, 40, 3), 43)
    ref = _rand((1, 8, 8, 3), 44)
    mu, sd = R.lab_stats(R.kornia_rgb_to_lab(x.permute(0, 3, 1, 2)))
    rmu, rsd = R.lab_stats(R.kornia_rgb_to_lab(ref.permute(0, 3, 1, 2)))
    ims = torch.stack([mu.flatten(1), sd.flatten(1)], dim=-1).contiguous().to(dev)      # [F,3,2]
    rms = torch.stack([rmu.flatten(1), rsd.flatten(1)], dim=-1).contiguous().to(dev)
    got = ops.colormatch_apply(x.to(dev), ims, rms, 0.8, cm_math="fast").cpu()
    want = R.color_match(x, ref, 0.8, 2)
    d = _unit_ulps(got, want)
    _record("apply.fast_vs_cpu_oracle_injected_stats", d)
    assert d <= CM_CROSS_REF_ULP, d


# ---------------------------------------------------------------------------------------- fused chain
def _lut_pair(ops, dev, name="AMD_TealOrange_33.cube"):
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    data = R.parse_cube_file(os.path.join(iv.LUTS_DIR, name))
    return data, ops.upload_lut(data, dev)


CHAIN_CASES = [
    dict(grain=(0.04, 0.5, 4), lut=10.0, sharpen=("unsharp", 0.5, False)),
    dict(grain=(0.1, 0.2, 2), lut=6.0, sharpen=("unsharp", 2.0, True)),
    dict(grain=(0.04, 0.5, 0), lut=None, sharpen=None),
    dict(grain=None, lut=10.0, sharpen=("laplacian", 0.8, False)),
    dict(grain=(0.04, 0.5, 1), lut=10.0, sharpen=None),
    dict(grain=None, lut=None, sharpen=("sobel", 0.6, True)),
    dict(grain=(0.3, 1.0, 3), lut=None, sharpen=("unsharp", 1.0, False)),
]


CHAIN_SHAPES = [(5, 45, 70, 3), (4, 64, 128, 3), (2, 1, 1, 3), (1, 3, 1, 3), (1, 1, 5, 3), (3, 5, 7, 3), (2, 64, 61, 3), (2, 33, 62, 3),
                (1, 7, 123, 3), (1, 15, 17, 3), (1, 10, 33, 3), (4, 30, 200, 3)]


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("case", CHAIN_CASES)
@pytest.mark.parametrize("shape", CHAIN_SHAPES)
def test_fused_chain_equals_sequential_operators_and_oracle(ops, dev, case, shape, variant):
    data, dlut = _lut_pair(ops, dev)
    x = _rand(shape, 51, -0.05, 1.05)
    xd = x.to(dev)
    spec = ops.ChainSpec(grain=case["grain"], lut=(dlut, case["lut"]) if case["lut"] is not None else None, sharpen=case["sharpen"],
                         variant=variant)
    torch.manual_seed(77)
    fused = ops.fused_chain(xd, spec)
    torch.manual_seed(77)
    y = xd
    if case["grain"]:
        y = ops.film_grain(y, case["grain"][0], case["grain"][1], chunk_frames=case["grain"][2])
    if case["lut"] is not None:
        y = ops.lut3d(y, dlut, case["lut"])
    if case["sharpen"]:
        y = ops.stencil3x3(y, *case["sharpen"])
    assert_bit_equal(fused, y, "fused vs sequential kernels")
    # and against the CPU oracle with the very noise torch draws
    torch.manual_seed(77)
    o = x
    if case["grain"]:
        I, s, bs = case["grain"]
        o = R.fast_film_grain(o, I, s, bs, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    if case["lut"] is not None:
        o = R.apply_lut_with_strength(o, data, case["lut"])
    if case["sharpen"]:
        name, s, zero = case["sharpen"]
        if name == "unsharp":
            o = R.unsharp(o, s, zero).contiguous()
        elif zero:
            o = (R.laplacian_zero_raster if name == "laplacian" else R.sobel_zero_raster)(o, s)
        else:
            o = (R.laplacian if name == "laplacian" else R.sobel)(o, s, False)
    assert_bit_equal(fused, o, "fused vs oracle")


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("bs,shape", [(4, (3, 720, 1280, 3)), (0, (2, 600, 700, 3)), (1, (2, 540, 960, 3))])
def test_fused_chain_across_several_philox_groups(ops, dev, variant, bs, shape):
    """Chunks larger than 4*G elements: several Philox call indices, ragged quarter rows, sibling strips that wrap
    around row ends and cross frame boundaries."""
    data, dlut = _lut_pair(ops, dev, "AMD_WarmFilm_25.cube")
    x = _rand(shape, 52)
    spec = ops.ChainSpec(grain=(0.08, 0.4, bs), lut=(dlut, 10.0), sharpen=("unsharp", 0.7, False), variant=variant)
    torch.manual_seed(11)
    fused = ops.fused_chain(x.to(dev), spec)
    torch.manual_seed(11)
    o = R.fast_film_grain(x, 0.08, 0.4, bs, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    o = R.unsharp(R.apply_lut_with_strength(o, data, 10.0), 0.7, False)
    assert_bit_equal(fused, o, f"fused variant {variant} vs oracle, {shape} bs={bs}")
    # point-wise chain (no stencil) through the same kernels
    spec = ops.ChainSpec(grain=(0.08, 0.4, bs), lut=(dlut, 6.0), variant=variant)
    torch.manual_seed(12)
    fused = ops.fused_chain(x.to(dev), spec)
    torch.manual_seed(12)
    o = R.fast_film_grain(x, 0.08, 0.4, bs, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    assert_bit_equal(fused, R.apply_lut_with_strength(o, data, 6.0), f"point-wise variant {variant}")


@pytest.mark.parametrize("sharpen", [True, False])
def test_march_rows_with_nan_and_inf_take_the_pass_through_forms(ops, dev, sharpen):
    """Round 6: the steady rows of grain -> (unsharp) run a body WITHOUT the NaN / Inf pass-through selects when the wave's twelve input values
    of the row hold no NaN, and the full forms for a row that does -- and for the two steps whose stencil windows still see it.  Frames
    with NaN, +Inf and -Inf sprinkled over steady rows, priming rows and frame borders: the march (variant 2), the tile kernels (variant 1)
    and the stand-alone operators give the same BITS, NaN payloads included; clean frames next to them are untouched by the slow rows."""
    g = torch.Generator().manual_seed(91)
    x = torch.rand((4, 720, 1280, 3), generator=g)
    nan, inf = float("nan"), float("inf")
    marks = [(0, 5, 7, 0, nan), (0, 300, 64, 1, inf), (0, 300, 65, 2, -inf), (1, 400, 640, 0, nan), (1, 401, 640, 1, nan), (1, 719, 1279, 2, nan),
             (2, 0, 0, 0, -inf), (2, 360, 100, 1, inf), (2, 360, 101, 1, -inf), (2, 500, 1279, 0, nan), (3, 100, 0, 2, nan), (3, 100, 61, 0, inf)]
    for f, y, xx, c, v in marks:
        x[f, y, xx, c] = v
    # a whole clean frame region far from every mark exists in each frame: rows 150..250
    xd = x.to(dev)
    outs = {}
    for variant in (2, 1):
        spec = ops.ChainSpec(grain=(0.05, 0.4, 2), sharpen=("unsharp", 0.7, False) if sharpen else None, variant=variant)
        outs[variant] = ops.fused_chain(xd, spec, generator=torch.Generator(device=dev).manual_seed(17))
    y = ops.film_grain(xd, 0.05, 0.4, chunk_frames=2, generator=torch.Generator(device=dev).manual_seed(17))
    if sharpen:
        y = ops.stencil3x3(y, "unsharp", 0.7, False)
    bits = lambda t: t.view(torch.int32)
    assert torch.equal(bits(outs[2]), bits(outs[1])) and torch.equal(bits(outs[2]), bits(y))
    assert bool(torch.isnan(outs[2][0, 5, 7, 0])) and float(outs[2][0, 300, 64, 1]) == 1.0 and float(outs[2][0, 300, 65, 2]) == 0.0 or sharpen
    assert int(torch.isnan(outs[2]).sum()) >= 5
    # the CPU restatement agrees on a crop that holds a NaN, an Inf pair and clean rows (same noise: torch.randn on the device)
    torch.manual_seed(23)
    fused = ops.fused_chain(xd[:2], ops.ChainSpec(grain=(0.05, 0.4, 2), sharpen=("unsharp", 0.7, False) if sharpen else None))
    torch.manual_seed(23)
    o = R.fast_film_grain(x[:2], 0.05, 0.4, 2, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    if sharpen:
        o = R.unsharp(o, 0.7, False)
    got, want = fused.cpu(), o
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0))


@pytest.mark.parametrize("variant", [1, 2])
def test_fused_chain_with_colour_match(ops, dev, variant):
    data, dlut = _lut_pair(ops, dev, "AMD_WarmFilm_25.cube")
    x = _rand((4, 48, 80, 3), 61)
    ref = _rand((1, 30, 30, 3), 62)
    xd = x.to(dev)
    ref_ms = ops.reference_stats(ref.to(dev))
    spec = ops.ChainSpec(grain=(0.04, 0.5, 2), lut=(dlut, 10.0), colormatch=(ref_ms, 0.9), sharpen=("unsharp", 0.5, False), variant=variant)
    torch.manual_seed(5)
    fused = ops.fused_chain(xd, spec)
    torch.manual_seed(5)
    y = ops.film_grain(xd, 0.04, 0.5, chunk_frames=2)
    y = ops.lut3d(y, dlut, 10.0)
    y = ops.color_match(y, None, 0.9, ref_ms=ref_ms)
    y = ops.stencil3x3(y, "unsharp", 0.5, False)
    assert_bit_equal(fused, y, "fused 4-stage vs sequential kernels (device statistics: the same reductions over the same Lab image)")
    # with the fp64 statistics the fused pass 1 (shared-Philox kernel) and the stand-alone statistics kernel add the same fp64 terms
    # in a different order: the fp32 mean/std agree except when a sum sits on a rounding boundary (~1e-6 of cases)
    import dataclasses
    ref64 = ops.reference_stats(ref.to(dev), cm_stats="fp64")
    spec64 = dataclasses.replace(spec, colormatch=(ref64, 0.9), cm_stats="fp64")
    torch.manual_seed(5)
    fused64 = ops.fused_chain(xd, spec64)
    torch.manual_seed(5)
    y64 = ops.stencil3x3(ops.color_match(ops.lut3d(ops.film_grain(xd, 0.04, 0.5, chunk_frames=2), dlut, 10.0), None, 0.9, ref_ms=ref64, cm_stats="fp64"),
                         "unsharp", 0.5, False)
    assert (fused64 - y64).abs().max() <= 1e-6, "fused 4-stage vs sequential kernels, fp64 statistics"
    torch.manual_seed(5)
    assert_bit_equal(ops.fused_chain(xd, spec64, cache_lab=False), fused64, "Lab-caching vs recomputing two-pass forms (fp64 statistics)")
    torch.manual_seed(5)
    recompute = ops.fused_chain(xd, spec, cache_lab=False)          # 36 B/px form: grain/LUT/Lab evaluated in both passes
    assert_bit_equal(recompute, fused, "Lab-caching vs recomputing two-pass forms")
    ws = torch.empty_like(xd)
    torch.manual_seed(5)
    assert_bit_equal(ops.fused_chain(xd, spec, lab_workspace=ws), fused, "caller-supplied Lab workspace")
    torch.manual_seed(5)
    o = R.fast_film_grain(x, 0.04, 0.5, 2, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    o = R.apply_lut_with_strength(o, data, 10.0)
    o = R.color_match(o.to(dev), ref.to(dev), 0.9, 1).cpu()         # the colour match of the reference, evaluated on the device
    o = R.unsharp(o, 0.5, False)
    assert_bit_equal(fused, o, "grain -> LUT -> colour match -> unsharp vs the oracle (colour match evaluated by torch on the device)")
    d = _unit_ulps(fused64, o)
    _record("e2e.fused_chain4_fp64stats_vs_device_oracle", d)
    assert d <= 2 * CM_E2E_DEVICE_ULP, d                            # unsharp at 0.5 amplifies a difference by <= 1 + 2*0.5*(8/9)


@pytest.mark.parametrize("shape,bs", [((4, 48, 80, 3), 2), ((3, 96, 128, 3), 0), ((2, 540, 960, 3), 1), ((4, 270, 480, 3), 4)])
def test_shared_philox_statistics_pass(ops, dev, shape, bs):
    """Pass 1 for chains that start with grain (vrg_produce.hip: Philox shared over four element runs, normals
    through LDS) against the general per-element kernel (knob 0x200): the Lab images must be bit-identical
    (same normals, same arithmetic), the fp64 statistics equal to reduction-order rounding."""
    data, dlut = _lut_pair(ops, dev, "AMD_WarmFilm_25.cube")
    x = _rand(shape, 71).to(dev)
    lab_a, lab_b = torch.empty_like(x), torch.empty_like(x)
    torch.manual_seed(21)
    st_a = ops.chain_stats(x, ops.ChainSpec(grain=(0.06, 0.3, bs), lut=(dlut, 10.0)), lab_out=lab_a)
    torch.manual_seed(21)
    st_b = ops.chain_stats(x, ops.ChainSpec(grain=(0.06, 0.3, bs), lut=(dlut, 10.0), variant=0x200), lab_out=lab_b)
    assert torch.equal(lab_a, lab_b)
    assert torch.equal(st_a[..., 0], st_b[..., 0])
    assert ((st_a[..., 1] - st_b[..., 1]).abs() <= 1e-12 * (1 + st_b[..., 1].abs())).all()
    assert ((st_a[..., 2] - st_b[..., 2]).abs() <= 1e-11 * (1 + st_b[..., 2].abs())).all()
    # and both equal the statistics of the materialised grain -> LUT image
    torch.manual_seed(21)
    y = ops.lut3d(ops.film_grain(x, 0.06, 0.3, chunk_frames=bs), dlut, 10.0)
    st_c = ops.lab_stats(y)
    assert ((st_a[..., 1] - st_c[..., 1]).abs() <= 1e-12 * (1 + st_c[..., 1].abs())).all()
    # grain only (no LUT) goes through the same kernel
    torch.manual_seed(22)
    g_a = ops.chain_stats(x, ops.ChainSpec(grain=(0.06, 0.3, bs)), lab_out=lab_a)
    torch.manual_seed(22)
    g_b = ops.chain_stats(x, ops.ChainSpec(grain=(0.06, 0.3, bs), variant=0x200), lab_out=lab_b)
    assert torch.equal(lab_a, lab_b) and ((g_a - g_b).abs() <= 1e-11 * (1 + g_b.abs())).all()


# ---------------------------------------------------------------------------------------- full-size properties
def test_full_size_4k_properties(ops, dev):
    """At BASELINE.json's frame size the oracle is too slow; use size-independent properties instead."""
    data, dlut = _lut_pair(ops, dev)
    F, H, W = 4, 2160, 3840
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((F, H, W, 3), generator=g, device=dev)
    spec = ops.ChainSpec(grain=(0.04, 0.5, 2), lut=(dlut, 10.0), sharpen=("unsharp", 0.5, False))
    torch.manual_seed(9)
    fused = ops.fused_chain(x, spec)
    # (1) fused == stand-alone kernels back to back, bit for bit, at full size -- and both fused kernels agree
    torch.manual_seed(9)
    y = ops.stencil3x3(ops.lut3d(ops.film_grain(x, 0.04, 0.5, chunk_frames=2), dlut, 10.0), "unsharp", 0.5, False)
    assert torch.equal(fused, y)
    torch.manual_seed(9)
    for variant in (1, 2):
        torch.manual_seed(9)
        other = ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 2), lut=(dlut, 10.0), sharpen=("unsharp", 0.5, False), variant=variant))
        assert torch.equal(fused, other), variant
        del other
    # (2) the grain stream at full size is torch's
    torch.manual_seed(9)
    n = torch.cat([torch.randn((2, H, W, 3), device=dev) for _ in range(2)])
    torch.manual_seed(9)
    gr = ops.film_grain(x, 0.04, 0.5, chunk_frames=2)
    assert torch.equal(gr, ops.film_grain_injected(x, n, 0.04, 0.5))
    del n
    # (3) frames are independent units: processing a frame range alone == slicing the batch result
    torch.manual_seed(9)
    part = ops.fused_chain(x[2:4], ops.ChainSpec(lut=(dlut, 10.0), sharpen=("unsharp", 0.5, False)))
    whole = ops.fused_chain(x, ops.ChainSpec(lut=(dlut, 10.0), sharpen=("unsharp", 0.5, False)))
    assert torch.equal(part, whole[2:4])
    # (4) range and idempotent clamp; strength 0 sharpen is clamp(x)
    assert float(fused.min()) >= 0.0 and float(fused.max()) <= 1.0
    assert torch.equal(ops.stencil3x3(x, "unsharp", 0.0, False), x.clamp(0, 1))
    # (5) a one-row slab of the 4K frame against the CPU oracle (LUT + unsharp, rows 0..2 incl. the top border)
    cpu = x[0:1, 0:3].cpu()
    o = R.unsharp(R.apply_lut_with_strength(cpu, data, 10.0), 0.5, False)
    assert torch.equal(whole[0, 0:2].cpu(), o[0, 0:2])


# ---------------------------------------------------------------------------------------- 13-slider Adjust (8f-2)
def _adjust_meta():
    import json
    with open(os.path.join(GOLDEN, "adjust_cases.json")) as fh:
        return json.load(fh)


def _adjust_want(x, settings, fixture=None):
    """Bit-exact target: the reference's CPU result, except that the vignette distance uses the correctly rounded
    sqrt the device has (torch's CPU sqrt is 1 ulp off on ~0.5 % of inputs; oracle/restated.py adjust_tensor)."""
    norm = R.normalize_adjust_settings(settings)
    if norm["enabled"] and norm["vignette"] > 0.0:
        return R.adjust_tensor(x, settings, ieee_sqrt=True).contiguous()
    return _t(fixture) if fixture is not None else R.adjust_tensor(x, settings).contiguous()


def test_adjust_route_matches_reference_fixtures(pkg, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    z = _npz("adjust.npz")
    meta = _adjust_meta()
    for tag in meta["shapes"]:
        x = _t(z[f"{tag}.x"])
        for name, settings in meta["cases"].items():
            got = LVT._apply_adjust_tensor(x, settings, "cpu")
            assert got.is_cuda and got.dtype == torch.float32
            ref = _t(z[f"{tag}.{name}"])
            assert (got.cpu() - ref).abs().max().item() <= 1.2e-7, (tag, name)      # reference on CPU, as committed
            assert_bit_equal(got, _adjust_want(x, settings, z[f"{tag}.{name}"]), f"adjust {tag}.{name}")


ADJUST_BIG = [
    ((2, 97, 150, 3), {"clarity": 55, "sharpen": 25, "vignette": 80, "fade": 30, "exposure": 20}),
    ((1, 33, 64, 3), {"clarity": -40, "temperature": 60, "blacks": 50}),
    ((3, 64, 129, 3), {"sharpen": 100, "whites": -35, "tint": 44}),
    ((1, 32, 65, 3), {"saturation": -100, "contrast": 100, "vignette": 100}),
    ((2, 270, 480, 3), {"temperature": -12.5, "tint": 8, "saturation": 14, "exposure": -9, "contrast": 11, "highlights": -22,
                         "shadows": 17, "whites": 6, "blacks": -4, "sharpen": 33, "clarity": 21, "vignette": 28, "fade": 9}),
    ((1, 8, 300, 3), {"clarity": 100}),          # box shrinks to 7
    ((1, 300, 6, 3), {"clarity": 100, "sharpen": 50}),   # box shrinks to 5
]


@pytest.mark.parametrize("shape,settings", ADJUST_BIG)
def test_adjust_kernels_vs_oracle_across_tiles(ops, pkg, dev, shape, settings):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    x = _rand(shape, seed=sum(shape) + len(settings), lo=-0.1, hi=1.1)
    terms = ops.adjust_terms(LVT._normalize_adjust_settings(settings))
    xd = x.to(dev)
    got = ops.adjust(xd, terms)
    assert_bit_equal(got, _adjust_want(x, settings), f"adjust {shape} {settings}")
    # caller-provided output / workspace, and input left untouched
    out = torch.full_like(xd, -7.0)
    ws = torch.empty_like(xd)
    got2 = ops.adjust(xd, terms, out=out, workspace=ws)
    assert got2 is out and torch.equal(out, got) and torch.equal(xd.cpu(), x)


def test_adjust_1080p_all_sliders(ops, pkg, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    settings = ADJUST_BIG[4][1]
    x = _rand((1, 1080, 1920, 3), seed=77, lo=-0.05, hi=1.05)
    got = ops.adjust(x.to(dev), ops.adjust_terms(LVT._normalize_adjust_settings(settings)))
    assert_bit_equal(got, _adjust_want(x, settings), "adjust 1080p")


@pytest.mark.parametrize("shape", [(2, 64, 96, 3), (1, 270, 480, 3), (2, 7, 5, 3), (1, 33, 129, 3)])
def test_adjust_device_arithmetic_bit_equal_device_oracle(pkg, ops, dev, shape):
    """_apply_adjust_tensor(..., device="cuda") of the reference does its arithmetic on the GPU: `/ 0.45` and `/ 1.05` are
    multiplications by the fp32-rounded reciprocal there (ATen), avg_pool2d / linspace / sqrt are the device kernels.  Our
    kernels with device_math reproduce THAT bit for bit (the oracle evaluated by torch on this GPU); with device="cpu" they
    reproduce the CPU arithmetic (the committed fixtures, tests above)."""
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    x = _rand(shape, 301, -0.1, 1.1)
    cases = [{"highlights": 60, "shadows": -40, "whites": 30, "blacks": -25, "contrast": 15, "exposure": 12, "temperature": 20, "tint": -10},
             {"vignette": 55, "fade": 20, "saturation": 25},
             {"clarity": 45, "sharpen": 30, "vignette": 20, "contrast": 10, "highlights": -35},
             {"clarity": -60, "shadows": 80}]
    differs_from_cpu = 0
    for settings in cases:
        want = R.adjust_tensor(x.to(dev), settings)                      # torch ops on the device
        got = LVT._apply_adjust_tensor(x, settings, "cuda")
        assert_bit_equal(got, want, f"adjust device arithmetic {settings}")
        cpu_arith = LVT._apply_adjust_tensor(x, settings, "cpu")
        differs_from_cpu += int(not torch.equal(cpu_arith.cpu(), got.cpu()))
    if shape[1] >= 64:
        assert differs_from_cpu > 0, "the two arithmetics should differ somewhere on frames this large"


def test_adjust_argument_errors(ops, pkg, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    terms = ops.adjust_terms(LVT._normalize_adjust_settings({"clarity": 10, "sharpen": 10}))
    with pytest.raises(ValueError):
        ops.adjust(torch.zeros(1, 4, 4, 4, device=dev), terms)
    with pytest.raises(ValueError):
        ops.adjust(torch.zeros(1, 4, 4, 3, device=dev), terms, workspace=torch.zeros(1, 4, 5, 3, device=dev))
    with pytest.raises(RuntimeError):
        ops.adjust(torch.zeros(1, 4, 4, 3), terms)
    assert ops.adjust(torch.zeros(0, 4, 4, 3, device=dev), terms).shape == (0, 4, 4, 3)


def test_side_operands_are_validated(ops, pkg, dev):
    """Statistics rows and LUT record tables reach the kernels as raw pointers: wrong device / type / layout must raise."""
    x = _rand((2, 8, 8, 3), 5).to(dev)
    ms = ops.finalize_stats(ops.lab_stats(x))
    with pytest.raises(RuntimeError):
        ops.colormatch_apply(x, ms, ms[:1].cpu(), 1.0)
    with pytest.raises(ValueError):
        ops.colormatch_apply(x, ms[:1], ms[:1], 1.0)                     # statistics of one frame for two
    with pytest.raises(ValueError):
        ops.colormatch_apply(x, ms.double(), ms[:1], 1.0)
    with pytest.raises(ValueError):
        ops.fused_chain(x, ops.ChainSpec(colormatch=(ms.permute(0, 2, 1)[:1], 1.0)))      # [n, 2, 3] view
    lut = ops.upload_lut({"size": 5, "lut": _rand((5, 5, 5, 3), 9), "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}, dev)
    bad = ops.DeviceLut(lut.table.cpu(), lut.size, lut.domain_min, lut.domain_max)
    with pytest.raises(RuntimeError):
        ops.lut3d(x, bad, 10.0)
    with pytest.raises(RuntimeError):
        ops.fused_chain(x, ops.ChainSpec(lut=(bad, 10.0)))
    assert_bit_equal(ops.colormatch_apply(x, ms, ms[:1], 1.0), ops.color_match(x, None, 1.0, ref_ms=ms[:1], cache_lab=False, cm_stats="fp64"), "apply forms")


# ---------------------------------------------------------------------------------------- uint8 codec edge (8f-3)
def _frames_eq(got, want, what):
    got = np.stack([np.asarray(f) for f in got], axis=0) if isinstance(got, list) else np.asarray(got)
    want = np.stack(want, axis=0) if isinstance(want, list) else np.asarray(want)
    assert got.dtype == np.uint8 and got.shape == want.shape, (what, got.dtype, got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.nonzero(got != want)
        pytest.fail(f"{what}: {bad[0].size}/{got.size} bytes differ, first at {tuple(int(b[0]) for b in bad)}: "
                    f"got {got[bad][0]} want {want[bad][0]}")


class _ListWriter:
    def __init__(self):
        self.frames = []

    def write(self, frame):
        self.frames.append(np.array(frame, copy=True))


def test_u8_converters_match_reference_fixtures(pkg, ops, dev):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT, VRGDG_StandaloneVideoEnhancerNodes as SVE
    z = _npz("io_u8.npz")
    for mod in (LVT, SVE):
        for tag in ("rand", "ramp"):
            t = mod._frames_to_tensor(list(z[f"{tag}.frames"]))
            if mod is SVE:                       # the enhancer's loop keeps the decoded batch in uint8 and defers the / 255
                assert isinstance(t, SVE.DecodedFrames) and t.is_cuda and t.u8.dtype == torch.uint8 and len(t) == len(z[f"{tag}.frames"])
                _frames_eq(SVE._tensor_to_frames(t), z[f"{tag}.frames"], "deferred frames round trip")
                t = t.float_tensor()
            assert t.is_cuda and t.dtype == torch.float32
            assert_bit_equal(t, _t(z[f"{tag}.tensor"]), f"frames_to_tensor {tag}")
        for tag in ("tens", "edge"):
            _frames_eq(mod._tensor_to_frames(_t(z[f"{tag}.tensor"])), z[f"{tag}.frames"], f"tensor_to_frames {tag}")
    # NaN quantises to 0 (numpy's x86 cast), +-Inf saturate; CPU tensors are accepted like GPU ones
    odd = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.5, 1.0, 2.0]).reshape(1, 1, 2, 3)
    got = ops.f32_to_frames_u8(odd.to(dev)).cpu().numpy().reshape(-1)
    assert got.tolist() == [0, 255, 0, 255, 255, 127]          # B,G,R order within each pixel
    with pytest.raises(ValueError):
        ops.frames_u8_to_f32(torch.zeros(1, 2, 2, 3, device=dev))
    with pytest.raises(ValueError):
        LVT._stack_frames([np.zeros((2, 2, 3), dtype=np.float32)])


def test_u8_route_batches_match_reference_fixtures(pkg, dev, monkeypatch):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT, VRGDG_IV_Adjustments as iv
    z = _npz("io_u8.npz")
    frames = list(z["rand.frames"])
    monkeypatch.setattr(iv, "LUTS_DIR", GOLDEN)
    monkeypatch.setattr(iv.VRGDG_LUTS, "_LUT_CACHE", {})
    for s in (10.0, 4.5):
        w = _ListWriter()
        assert LVT._process_video_batch(frames, w, "synthetic_17.cube", s, "cuda") == 3
        _frames_eq(w.frames, z[f"batch.lut.s{s}"], f"_process_video_batch {s}")
    cases = _adjust_meta()["cases"]
    for name in ("all", "both", "fade_vig", "tone"):
        w = _ListWriter()
        assert LVT._process_adjust_batch(frames, w, cases[name], "cuda") == 3
        want = R.tensor_to_frames(_adjust_want(R.frames_to_tensor(frames), cases[name]))
        _frames_eq(w.frames, want, f"_process_adjust_batch {name}")
        diff = np.abs(np.stack(w.frames).astype(np.int16) - z[f"batch.adjust.{name}"].astype(np.int16))
        assert diff.max() <= 1 and (diff != 0).mean() < 0.01, name       # reference on CPU: torch.sqrt's last ulp
    # grain: one randn draw for the batch from a generator seeded per call, clamped sliders
    w = _ListWriter()
    LVT._process_film_grain_batch(frames, w, 0.2, 0.7, "cuda", seed=123)
    g = torch.Generator(device=dev).manual_seed(123)
    x = R.frames_to_tensor(frames)
    noise = torch.randn(x.shape, generator=g, device=dev).cpu()
    _frames_eq(w.frames, R.tensor_to_frames(R.grain_apply(x, noise, 0.2, 0.7)), "_process_film_grain_batch")


U8_CHAINS = [
    dict(grain=(0.06, 0.5, 2), lut=10.0, sharpen=("unsharp", 0.8, False)),
    dict(grain=None, lut=7.5, sharpen=("unsharp", 1.5, True)),
    dict(grain=(0.1, 1.0, 0), lut=None, sharpen=None),
    dict(grain=None, lut=10.0, sharpen=None),
    dict(grain=None, lut=None, sharpen=("laplacian", 0.4, False)),
    dict(grain=(0.05, 0.3, 1), lut=3.0, sharpen=None),
    dict(grain=None, lut=0.0, sharpen=None),                 # nothing to do: uint8 copy == convert -> convert
]


@pytest.mark.parametrize("shape", [(4, 45, 70, 3), (2, 64, 129, 3), (3, 1, 5, 3), (2, 270, 480, 3)])
@pytest.mark.parametrize("case", U8_CHAINS)
@pytest.mark.parametrize("variant", [0, 1])
def test_fused_chain_u8_equals_convert_chain_convert(ops, dev, case, shape, variant):
    data, dlut = _lut_pair(ops, dev)
    g = torch.Generator().manual_seed(sum(shape))
    frames = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    spec = ops.ChainSpec(grain=case["grain"], lut=(dlut, case["lut"]) if case["lut"] is not None else None, sharpen=case["sharpen"],
                         variant=variant)       # 0: the library's choice (grain alone: the shared-Philox grain kernel on uint8 frames)
    torch.manual_seed(5)
    got = ops.fused_chain(frames.to(dev), spec)
    assert got.dtype == torch.uint8 and got.shape == frames.shape
    torch.manual_seed(5)
    via_f32 = ops.f32_to_frames_u8(ops.fused_chain(ops.frames_u8_to_f32(frames.to(dev)), spec))
    _frames_eq(got.cpu().numpy(), via_f32.cpu().numpy(), "u8 chain vs convert -> fp32 chain -> convert")
    # and against the CPU oracle from the bytes up
    torch.manual_seed(5)
    o = R.frames_to_tensor(list(frames.numpy()))
    if case["grain"]:
        I, s, bs = case["grain"]
        o = R.fast_film_grain(o, I, s, bs, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    if case["lut"] is not None:
        o = R.apply_lut_with_strength(o, data, case["lut"])
    if case["sharpen"]:
        name, s, zero = case["sharpen"]
        o = R.unsharp(o, s, zero).contiguous() if name == "unsharp" else R.laplacian(o, s, False)
    _frames_eq(got.cpu().numpy(), R.tensor_to_frames(o), "u8 chain vs oracle")


@pytest.mark.parametrize("shape,settings", ADJUST_BIG[:5] + ADJUST_BIG[6:])
def test_adjust_u8_equals_convert_adjust_convert(ops, pkg, dev, shape, settings):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    g = torch.Generator().manual_seed(sum(shape) + 1)
    frames = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    terms = ops.adjust_terms(LVT._normalize_adjust_settings(settings))
    got = ops.adjust(frames.to(dev), terms)
    assert got.dtype == torch.uint8
    want = R.tensor_to_frames(_adjust_want(R.frames_to_tensor(list(frames.numpy())), settings))
    _frames_eq(got.cpu().numpy(), want, f"adjust u8 {shape}")


def test_u8_entry_points_reject_colour_match(ops, dev):
    ref_ms = torch.ones(1, 3, 2, device=dev)
    with pytest.raises(ValueError):
        ops.fused_chain(torch.zeros(1, 4, 4, 3, dtype=torch.uint8, device=dev), ops.ChainSpec(colormatch=(ref_ms, 1.0)))


def test_full_size_4k_properties_adjust_and_u8(ops, pkg, dev):
    """Size-independent properties of the widened rows at 4K (the CPU oracle needs minutes per frame here)."""
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    data, dlut = _lut_pair(ops, dev)
    H, W = 2160, 3840
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand((2, H, W, 3), generator=g, device=dev) * 1.1 - 0.05
    # (1) point stage + vignette commute with flips: torch.linspace is evaluated symmetrically from both ends
    t_pt = ops.adjust_terms(LVT._normalize_adjust_settings({"temperature": 15, "exposure": -20, "contrast": 30, "saturation": 12,
                                                              "highlights": 25, "blacks": -18, "fade": 12, "vignette": 70}))
    a = ops.adjust(x, t_pt)
    assert torch.equal(ops.adjust(x.flip(1).contiguous(), t_pt), a.flip(1))
    assert torch.equal(ops.adjust(x.flip(2).contiguous(), t_pt), a.flip(2))
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    # (2) disabled == clamp; frames are independent units
    off = ops.adjust_terms(LVT._normalize_adjust_settings({"enabled": False, "clarity": 50}))
    assert torch.equal(ops.adjust(x, off), x.clamp(0, 1))
    t_all = ops.adjust_terms(LVT._normalize_adjust_settings({"clarity": 45, "sharpen": 30, "vignette": 20, "contrast": 10}))
    both = ops.adjust(x, t_all)
    assert torch.equal(ops.adjust(x[1:2], t_all), both[1:2])
    # (3) a slab of the 4K frame against the CPU oracle: clarity + sharpen need 5 rows of context, vignette the frame size
    #     -> oracle on the top 24 rows with the kernels' own geometry is not separable; check the no-vignette form instead
    t_cs = ops.adjust_terms(LVT._normalize_adjust_settings({"clarity": 45, "sharpen": 30, "contrast": 10}))
    got = ops.adjust(x[:1], t_cs)
    slab = x[:1, :40].cpu()
    want = R.adjust_tensor(slab, {"clarity": 45, "sharpen": 30, "contrast": 10})
    assert torch.equal(got[0, :30].cpu(), want[0, :30])            # rows whose 9x9 + 3x3 windows lie inside the slab
    # (4) uint8 edge at full size: round trip is the identity, fused uint8 chain == convert -> fp32 chain -> convert
    u8 = torch.randint(0, 256, (2, H, W, 3), dtype=torch.uint8, device=dev)
    assert torch.equal(ops.f32_to_frames_u8(ops.frames_u8_to_f32(u8)), u8)
    spec = ops.ChainSpec(grain=(0.05, 0.5, 1), lut=(dlut, 8.0), sharpen=("unsharp", 0.6, False), variant=1)
    torch.manual_seed(3)
    direct = ops.fused_chain(u8, spec)
    torch.manual_seed(3)
    via = ops.f32_to_frames_u8(ops.fused_chain(ops.frames_u8_to_f32(u8), spec))
    assert torch.equal(direct, via)
    assert torch.equal(ops.adjust(u8, t_all), ops.f32_to_frames_u8(ops.adjust(ops.frames_u8_to_f32(u8), t_all)))


# ---------------------------------------------------------------------------------------- host-fed pipeline
def test_pipelined_host_staging_equals_sequential(pkg, dev, monkeypatch):
    """CPU tensors in / out: the three-stream pipeline (pieces of a few frames, page-locked result) returns exactly
    what the plain upload -> run -> download path returns, including the generator bookkeeping across pieces."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
    x = _rand((11, 48, 80, 3), 123)
    ref = _rand((1, 20, 30, 3), 124)
    frame_bytes = x[0].numel() * 4
    calls = {
        "grain bs=2": lambda: nodes.FastFilmGrain().apply_grain(x, 0.05, 0.5, 2)[0],
        "grain bs=0": lambda: nodes.FastFilmGrain().apply_grain(x, 0.05, 0.5, 0)[0],
        "unsharp": lambda: nodes.FastUnsharpSharpen().apply_unsharp(x, 0.7, False)[0],
        "colour match": lambda: nodes.ColorMatchToReference().match_color(x, ref, 0.8, 3)[0],
        "lut": lambda: iv.VRGDG_LUTS().apply_lut(x, "AMD_WarmFilm_25.cube", "auto", 7.0)[0],
    }
    for pipe_bytes in (frame_bytes * 3, frame_bytes // 2, 1 << 40):       # 3-frame pieces, 1-frame pieces, one piece
        monkeypatch.setattr(_devices, "PIPE_BYTES", pipe_bytes)
        for name, fn in calls.items():
            monkeypatch.setattr(nodes, "PIPELINED", True)
            torch.manual_seed(31)
            a = fn()
            after_a = torch.cuda.get_rng_state(dev)
            assert a.device.type == "cpu" and a.dtype == torch.float32 and a.shape == x.shape
            if name != "lut":
                monkeypatch.setattr(nodes, "PIPELINED", False)
                torch.manual_seed(31)
                b = fn()
                assert torch.equal(torch.cuda.get_rng_state(dev), after_a), name
            else:
                data = R.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_WarmFilm_25.cube"))
                b = R.apply_lut_with_strength(x, data, 7.0)
            assert torch.equal(a, b), (name, pipe_bytes)
    # results above the page-lock limit are staged through the ring instead
    monkeypatch.setattr(_devices, "PIN_LIMIT_BYTES", 0)
    monkeypatch.setattr(_devices, "PIPE_BYTES", frame_bytes * 2)
    monkeypatch.setattr(nodes, "PIPELINED", True)
    a = calls["unsharp"]()
    assert not a.is_pinned()
    monkeypatch.setattr(nodes, "PIPELINED", False)
    assert torch.equal(a, calls["unsharp"]())


def test_pageable_frames_through_the_page_locked_ring_equal_the_runtime_copy(pkg, dev, monkeypatch):
    """Pageable input frames reach the GPU through this pack's page-locked ring (_devices._UploadRing: vrg_host_copy with several host
    threads, then asynchronous uploads) or through the runtime's own pageable copy (VRGDG_PAGEABLE_UPLOAD=runtime), page-locked ones
    directly: the same results and generator state every way -- with ring slots smaller than a piece, larger than the whole batch, and
    of a size that divides neither frames nor pieces; uint8 frames (byte counts off every alignment) as well."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
    x = _rand((7, 45, 83, 3), 521)
    ref = _rand((1, 20, 30, 3), 522)
    frame_bytes = x[0].numel() * 4
    calls = {"grain": lambda t: nodes.FastFilmGrain().apply_grain(t, 0.05, 0.5, 2)[0],
             "colour match": lambda t: nodes.ColorMatchToReference().match_color(t, ref, 0.8, 1)[0],
             "unsharp": lambda t: nodes.FastUnsharpSharpen().apply_unsharp(t, 0.7, False)[0]}
    monkeypatch.setattr(_devices, "PIPE_BYTES", frame_bytes * 2)
    monkeypatch.setattr(_devices, "DEVICE_CACHE_BYTES", 0)
    want = {}
    for name, fn in calls.items():
        monkeypatch.setattr(_devices, "PAGEABLE_UPLOAD", "runtime")
        torch.manual_seed(77)
        want[name] = (fn(x), torch.cuda.get_rng_state(dev))
    monkeypatch.setattr(_devices, "PAGEABLE_UPLOAD", "ring")
    for chunk in (4099, frame_bytes, frame_bytes * 3 // 2 + 4, 64 << 20):
        monkeypatch.setattr(_devices, "STAGE_CHUNK_BYTES", chunk)
        _devices._STAGING.buffers.clear()
        for threads in (1, 3):
            monkeypatch.setattr(_devices, "STAGE_THREADS", threads)
            for name, fn in calls.items():
                torch.manual_seed(77)
                got = fn(x)
                assert torch.equal(got, want[name][0]), (name, chunk, threads)
                assert torch.equal(torch.cuda.get_rng_state(dev), want[name][1]), (name, chunk)
    torch.manual_seed(77)
    assert torch.equal(calls["grain"](x.pin_memory()), want["grain"][0])
    # bytes: the route the stand-alone enhancer feeds (frames as uint8) -- 45 * 83 * 3 bytes per frame, a multiple of nothing
    u8 = (x * 255.0).to(torch.uint8)
    double = lambda g, first: g.to(torch.int16).mul(2).clamp(max=255).to(torch.uint8)
    monkeypatch.setattr(_devices, "STAGE_CHUNK_BYTES", 10007)
    _devices._STAGING.buffers.clear()
    monkeypatch.setattr(_devices, "PIPE_BYTES", u8[0].numel() * 3)
    got = _devices.stream_frames(u8, double)
    assert got.dtype == torch.uint8 and torch.equal(got, (u8.to(torch.int16) * 2).clamp(max=255).to(torch.uint8))
    _devices._STAGING.buffers.clear()


def test_adjacent_nodes_skip_the_reupload_and_a_mutated_intermediate_is_uploaded_again(pkg, ops, dev, monkeypatch):
    """Two nodes of this pack one after the other in a graph: ComfyUI hands the second the very tensor the first returned, whose frames
    are still in HBM -- the second node reads them there instead of uploading them again (_devices._DEVICE_COPIES).  Same bits as the
    plain path, same generator state; an intermediate changed in place (torch's version counter) is uploaded like any other tensor; the
    device copy dies with the CPU tensor."""
    import gc
    from comfyui_vrgamedevgirl_amd import nodes, _devices
    monkeypatch.setattr(_devices, "LAZY_DOWNLOAD", False)          # eager downloads: the cache of device copies on its own (lazy: the test below)
    cache = _devices._DEVICE_COPIES
    cache.clear()
    x = _rand((6, 90, 160, 3), 77)
    gen = torch.cuda.default_generators[dev.index]
    torch.manual_seed(123)
    a = nodes.FastFilmGrain().apply_grain(x, 0.05, 0.4, 2)[0]
    off = gen.get_offset()
    assert a.device.type == "cpu" and id(a) in cache.entries
    h0, m0 = cache.hits, cache.misses
    b = nodes.FastUnsharpSharpen().apply_unsharp(a, 0.7, False)[0]              # the frames of `a` are taken from HBM
    assert cache.hits == h0 + 1 and gen.get_offset() == off
    plain = nodes.FastUnsharpSharpen().apply_unsharp(a.clone(), 0.7, False)[0]  # a copy is a tensor the cache has never seen: uploaded
    assert cache.misses >= m0 + 1 and torch.equal(b, plain)
    c = nodes.ColorMatchToReference().match_color(b, x[:1], 0.8, 2)[0]          # a third node on the second one's result: another hit
    assert cache.hits == h0 + 2
    assert torch.equal(c, nodes.ColorMatchToReference().match_color(b.clone(), x[:1], 0.8, 2)[0])
    a[0, 0, 0, 0] += 0.25                                                          # in-place change of the intermediate
    h1 = cache.hits
    d = nodes.FastUnsharpSharpen().apply_unsharp(a, 0.7, False)[0]
    assert cache.hits == h1 and id(a) not in cache.entries                      # stale copy dropped, frames uploaded
    assert torch.equal(d, nodes.FastUnsharpSharpen().apply_unsharp(a.clone(), 0.7, False)[0]) and not torch.equal(d, b)
    key = id(b)
    assert key in cache.entries
    del b
    gc.collect()
    assert key not in cache.entries                                             # the device copy goes with the CPU tensor
    old = _devices.DEVICE_CACHE_BYTES
    try:
        _devices.DEVICE_CACHE_BYTES = 0                                          # VRGDG_DEVICE_CACHE_GB=0: nothing is kept
        cache.clear()
        e = nodes.FastUnsharpSharpen().apply_unsharp(x, 0.7, False)[0]
        assert not cache.entries and torch.equal(nodes.FastUnsharpSharpen().apply_unsharp(e, 0.7, False)[0],
                                                 nodes.FastUnsharpSharpen().apply_unsharp(e.clone(), 0.7, False)[0])
    finally:
        _devices.DEVICE_CACHE_BYTES = old
        cache.clear()


def test_toolchain_selfcheck_passes_on_this_build_and_a_failed_march_falls_back(pkg, ops, dev, monkeypatch):
    """The first-use self-check of the two toolchain-coupled fast forms (ops.toolchain_selfcheck): dev_pow_ziv == torch.pow at the three
    call sites, march == tile kernels on steady rows -- both hold on the build the suite runs on; a march that failed it is replaced by
    the tile kernels for the automatic choice (same bits)."""
    res = ops.toolchain_selfcheck(dev, force=True)
    assert res == {"pow": True, "march": True}
    st = ops.toolchain_status(dev)
    assert st["pow_equals_ocml"] and st["march_equals_tile_kernels"]
    x = _rand((2, 512, 512, 3), 5).to(dev)
    lut = ops.upload_lut(R.parse_cube_file(os.path.join(PKG_DIR, "LUTS", "AMD_TealOrange_33.cube")), dev)
    spec = ops.ChainSpec(grain=(0.04, 0.5, 2), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False))
    want = ops.fused_chain(x, spec, generator=torch.Generator(device=dev).manual_seed(3))
    monkeypatch.setitem(ops._TOOLCHAIN, dev.index, {"pow": True, "march": False})
    assert ops._auto_variant(dev) == 1
    got = ops.fused_chain(x, spec, generator=torch.Generator(device=dev).manual_seed(3))
    assert torch.equal(got, want)


@pytest.mark.parametrize("lazy", [True, False])
def test_nodes_under_inference_mode_as_comfyui_runs_them(pkg, ops, dev, monkeypatch, lazy):
    """ComfyUI executes every node inside torch.inference_mode() (and the reference's _apply_effects_batch does the same): host-fed node
    calls, the hand-over of device frames between adjacent nodes included (lazy download / device-copy cache), give the bits of the plain
    path there (reading `_version` of an inference tensor raised after every kernel had run)."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
    monkeypatch.setattr(_devices, "LAZY_DOWNLOAD", lazy)
    cache = _devices._DEVICE_COPIES
    cache.clear()
    x = _rand((6, 90, 160, 3), 78)
    torch.manual_seed(321)
    want_a = nodes.FastFilmGrain().apply_grain(x, 0.05, 0.4, 2)[0].clone()
    want_b = nodes.FastUnsharpSharpen().apply_unsharp(want_a.clone(), 0.7, False)[0].clone()
    want_c = nodes.ColorMatchToReference().match_color(want_b.clone(), x[:1], 0.8, 2)[0].clone()
    cache.clear()
    e0 = cache.errors
    handed_over = lambda: cache.hits + _devices._LAZY.downloads_skipped
    with torch.inference_mode():
        xi = x.clone()
        torch.manual_seed(321)
        a = nodes.FastFilmGrain().apply_grain(xi, 0.05, 0.4, 2)[0]
        assert a.is_inference() and a.device.type == "cpu" and tuple(a.shape) == tuple(x.shape)
        assert (_devices.pending_of(a) is not None) if lazy else (id(a) in cache.entries)
        h0 = handed_over()
        b = nodes.FastUnsharpSharpen().apply_unsharp(a, 0.7, False)[0]
        assert handed_over() == h0 + 1
        c = nodes.ColorMatchToReference().match_color(b, xi[:1], 0.8, 2)[0]
        assert handed_over() == h0 + 2
        lut_names = [n for n in iv.VRGDG_LUTS.INPUT_TYPES()["required"]["lut_name"][0] if n.endswith(".cube")]
        d = iv.VRGDG_LUTS().apply_lut(c, lut_names[0], "auto", 7.5)[0]
        assert tuple(d.shape) == tuple(x.shape)
        if lazy:
            assert _devices.pending_of(a) is not None and _devices.pending_of(b) is not None      # consumed on the device: never downloaded
        a.numpy()[:] = 0.5                                                      # whole-tensor write through an alias: the copy is stale
        h1 = handed_over()
        z = nodes.FastUnsharpSharpen().apply_unsharp(a, 0.7, False)[0]
        assert handed_over() == h1
        assert torch.equal(torch.full_like(z, 0.5), z)                          # unsharp of a constant frame is the frame
        assert torch.equal(b, want_b) and torch.equal(c, want_c)                # (reading them downloads them)
    assert cache.errors == e0
    cache.clear()


def test_lazy_download_four_nodes_in_a_graph_cross_pcie_twice(pkg, ops, dev, monkeypatch):
    """grain -> LUT -> colour match -> unsharp as ComfyUI runs them: every node hands the next a CPU tensor.  With the lazy download
    (_devices.LazyFrames) the three intermediates stay in HBM -- 1 upload + 1 download instead of 1 + 4 -- and every result still has the
    eager path's bits whenever and however it is read: torch ops, numpy, iteration, slicing, pickling; shape questions do not download;
    a result nobody reads is downloaded by the timer and becomes an ordinary cached device copy; dropping a never-read result frees it."""
    import gc
    import pickle
    import time
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
    cache = _devices._DEVICE_COPIES
    x = _rand((6, 72, 128, 3), 79)
    ref = _rand((1, 30, 40, 3), 80)
    lut_name = "AMD_WarmFilm_25.cube"

    def graph(t):
        r = [nodes.FastFilmGrain().apply_grain(t, 0.05, 0.4, 2)[0]]
        r.append(iv.VRGDG_LUTS().apply_lut(r[-1], lut_name, "auto", 8.0)[0])
        r.append(nodes.ColorMatchToReference().match_color(r[-1], ref, 0.9, 3)[0])
        r.append(nodes.FastUnsharpSharpen().apply_unsharp(r[-1], 0.6, False)[0])
        return r

    monkeypatch.setattr(_devices, "LAZY_DOWNLOAD", False)
    cache.clear()
    torch.manual_seed(11)
    want = [w.clone() for w in graph(x)]
    state = torch.cuda.get_rng_state(dev)
    monkeypatch.setattr(_devices, "LAZY_DOWNLOAD", True)
    monkeypatch.setattr(_devices, "LAZY_SECONDS", 30.0)
    cache.clear()
    skipped0 = _devices._LAZY.downloads_skipped
    torch.manual_seed(11)
    got = graph(x)
    assert torch.equal(torch.cuda.get_rng_state(dev), state)
    assert _devices._LAZY.downloads_skipped == skipped0 + 3
    for g in got:
        assert isinstance(g, torch.Tensor) and g.device.type == "cpu" and g.dtype == torch.float32 and tuple(g.shape) == tuple(x.shape)
        assert g.is_contiguous() and g.numel() == x.numel() and len(g) == 6 and g.stride() == x.stride()
        assert _devices.pending_of(g) is not None                            # none of that downloaded anything
    # every way of reading a result downloads it first
    assert torch.equal(got[3], want[3]) and _devices.pending_of(got[3]) is None
    assert np.array_equal(got[2].numpy(), want[2].numpy()) and _devices.pending_of(got[2]) is None
    assert all(torch.equal(f, w) for f, w in zip(got[1], want[1])) and _devices.pending_of(got[1]) is None      # iteration (PreviewImage / SaveImage)
    assert torch.equal(pickle.loads(pickle.dumps(got[0])), want[0]) and _devices.pending_of(got[0]) is None
    # a downloaded result is an ordinary device copy: the next node still skips its upload, and an in-place change is seen
    h = cache.hits
    again = nodes.FastUnsharpSharpen().apply_unsharp(got[2], 0.6, False)[0]
    assert cache.hits == h + 1 and torch.equal(again, want[3])
    got[2].mul_(0.5)
    h = cache.hits
    changed = nodes.FastUnsharpSharpen().apply_unsharp(got[2], 0.6, False)[0]
    assert cache.hits == h and torch.equal(changed, nodes.FastUnsharpSharpen().apply_unsharp(want[2] * 0.5, 0.6, False)[0])
    # slices, .to(), torch.cat on a pending result
    torch.manual_seed(11)
    got = graph(x)
    assert torch.equal(got[3][2:4], want[3][2:4]) and torch.equal(torch.cat([got[2], got[2]])[6:], want[2])
    assert torch.equal(got[1].to(dev).cpu(), want[1]) and torch.equal(got[0].permute(0, 3, 1, 2), want[0].permute(0, 3, 1, 2))
    # a pending result as the REFERENCE image of a colour match: read where it is
    lz = nodes.FastUnsharpSharpen().apply_unsharp(x[:1], 0.4, False)[0]
    assert _devices.pending_of(lz) is not None
    with_lazy_ref = nodes.ColorMatchToReference().match_color(x, lz, 0.7, 2)[0]
    assert _devices.pending_of(lz) is not None
    plain_ref = nodes.FastUnsharpSharpen().apply_unsharp(x[:1].clone(), 0.4, False)[0].clone()
    assert torch.equal(with_lazy_ref, nodes.ColorMatchToReference().match_color(x, plain_ref, 0.7, 2)[0])
    # nobody reads it: the timer downloads it, after which it sits in the device-copy cache
    monkeypatch.setattr(_devices, "LAZY_SECONDS", 0.2)
    torch.manual_seed(11)
    lonely = nodes.FastFilmGrain().apply_grain(x, 0.05, 0.4, 2)[0]
    p = _devices.pending_of(lonely)
    assert p is not None
    deadline = time.time() + 10.0
    while not p.done and time.time() < deadline:
        time.sleep(0.05)
    assert p.done and id(lonely) in cache.entries and torch.equal(lonely, want[0])
    # dropped unread: the device pieces go with it
    monkeypatch.setattr(_devices, "LAZY_SECONDS", 30.0)
    torch.manual_seed(11)
    unread = nodes.FastFilmGrain().apply_grain(x, 0.05, 0.4, 2)[0]
    p = _devices.pending_of(unread)
    assert p.recipe is not None and p.nbytes == 0                              # (round 6: deferred -- nothing has run, no HBM is held yet)
    assert p.device_pieces() and p.recipe is None and p.nbytes > 0             # a consumer of this pack asked for the frames in HBM
    before = torch.cuda.memory_allocated(dev)
    del unread
    gc.collect()
    _devices._LAZY._sweep()
    assert p not in _devices._LAZY.pending
    del p
    gc.collect()
    assert torch.cuda.memory_allocated(dev) < before
    cache.clear()


def test_enhancer_loop_stays_on_gpu_between_uint8_edges(pkg, dev):
    """decode -> _frames_to_tensor -> _process_with_retry -> _tensor_to_frames (VRGDG_StandaloneVideoEnhancerNodes.py:405-420
    of the reference): frames cross PCIe as uint8, the fp32 tensor never leaves the GPU, result == oracle from the bytes up."""
    from comfyui_vrgamedevgirl_amd import VRGDG_StandaloneVideoEnhancerNodes as enh
    g = torch.Generator().manual_seed(77)
    frames = list(torch.randint(0, 256, (5, 36, 50, 3), generator=g, dtype=torch.uint8).numpy())
    st = {"sharpen_enabled": True, "sharpen_strength": 0.6, "grain_enabled": True, "grain_intensity": 0.05, "saturation_mix": 0.4,
          "seed": 9, "use_gpu": True}
    tensor = enh._frames_to_tensor(frames)
    enhanced, used = enh._process_with_retry(tensor, st, 30)
    assert tensor.is_cuda and enhanced.is_cuda and used == 5
    out = enh._tensor_to_frames(enhanced)

    def noise_fn(fseed, shape):
        gg = torch.Generator(device=dev).manual_seed(fseed)
        return torch.randn(shape, generator=gg, device=dev).cpu()

    want = R.seeded_grain(R.unsharp(R.frames_to_tensor(frames), 0.6, True).contiguous(), 0.05, 0.4, 9, 30, noise_fn=noise_fn)
    _frames_eq(out, R.tensor_to_frames(want), "enhancer loop")
    cpu_in = enh._apply_effects_batch(R.frames_to_tensor(frames), st, 30)         # CPU tensor in -> CPU tensor out, as in the reference
    assert cpu_in.device.type == "cpu" and torch.equal(cpu_in, want)
    # a batch the device refuses is halved like the reference's (:297-308): same bytes, smallest batch reported
    calls = {"n": 0}
    real = enh._apply_effects_batch

    def flaky(images, settings, frame_start=0):
        calls["n"] += 1
        if len(images) > 2:
            raise RuntimeError("HIP out of memory (simulated)")
        return real(images, settings, frame_start)

    enh._apply_effects_batch = flaky
    try:
        halves, used = enh._process_with_retry(enh._frames_to_tensor(frames), st, 30)
    finally:
        enh._apply_effects_batch = real
    assert used <= 2 and isinstance(halves, enh.DecodedFrames)
    _frames_eq(enh._tensor_to_frames(halves), out, "halved batches")


@pytest.mark.parametrize("zero_border", [False, True], ids=["replicate", "zero"])
@pytest.mark.parametrize("shape", [(3, 37, 344, 3), (2, 64, 1024, 3), (2, 90, 500, 3), (2, 5, 4096, 3), (1, 1, 2048, 3), (2, 1080, 1920, 3),
                                   (1, 2160, 3840, 3),
                                   # round 5, any width / alignment (k_sharpen_grain_u8_any): widths off the dword grid, rows shorter than a
                                   # block, frames whose byte count is no multiple of 4, down to 1 x 1
                                   (3, 480, 854, 3), (2, 768, 1366, 3), (2, 20, 56, 3), (2, 20, 346, 3), (3, 37, 343, 3), (2, 33, 342, 3),
                                   (5, 1, 1, 3), (4, 7, 5, 3), (3, 2, 1, 3), (2, 1, 9, 3), (1, 3, 2, 3), (3, 5, 7, 3), (2, 1, 2, 3), (7, 2, 2, 3),
                                   (2, 64, 85, 3)], ids=lambda s: "x".join(map(str, s)))
def test_u8_sharpen_then_seeded_grain_equals_the_converter_route(pkg, ops, dev, shape, zero_border):
    """vrg_sharpen_grain_u8 -- decoded B,G,R bytes in, / 255, unsharp, per-frame-seeded grain, * 255 clip truncate, bytes out: the
    enhancer's loop body (VRGDG_StandaloneVideoEnhancerNodes.py:417-421) in one kernel -- against converter -> fused fp32 kernel ->
    converter (each held to the reference's fixtures / the oracle elsewhere), byte for byte; against the oracle from the bytes up for the
    small shapes; and the reference's batch-invariance property (tests/test_standalone_video_enhancer.py:39-61) on bytes."""
    from comfyui_vrgamedevgirl_amd import _hip, rng
    import ctypes as C
    g = torch.Generator().manual_seed(sum(shape) + int(zero_border))
    frames = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    nb = min(8, shape[2])
    frames[0, 0, :nb] = torch.tensor([0, 255, 1, 254, 128, 127, 255, 0], dtype=torch.uint8)[:nb, None]      # saturated codes on a border
    x = frames.to(dev)
    route = ops.f32_to_frames_u8(ops.sharpen_then_seeded_grain(ops.frames_u8_to_f32(x), 0.6, zero_border, 0.05, 0.4, 1234, 17))
    got = ops.sharpen_then_seeded_grain(x, 0.6, zero_border, 0.05, 0.4, 1234, 17)
    assert got.dtype == torch.uint8
    _frames_eq(got.cpu().numpy(), route.cpu().numpy(), "uint8 sharpen -> seeded grain")
    out = torch.zeros_like(x)
    d = ops.NoisePlan(1, rng.per_frame_seeded(x[0].numel(), 1234 + 17, dev)).desc()
    st = _hip.lib().vrg_sharpen_grain_u8(_hip.ptr(x), _hip.ptr(out), shape[0], shape[1], shape[2], 0.6, 1 if zero_border else 0,
                                         0.05, 0.4, float(np.float32(1.0 - 0.4)), C.byref(d), _hip.current_stream())
    assert st == _hip.VRG_OK                                   # the kernel itself ran (the op falls back for what it refuses)
    _frames_eq(out.cpu().numpy(), route.cpu().numpy(), "vrg_sharpen_grain_u8")
    if x.numel() < 1 << 20:
        def noise_fn(fseed, shp):
            gg = torch.Generator(device=dev).manual_seed(fseed)
            return torch.randn(shp, generator=gg, device=dev).cpu()
        want = R.seeded_grain(R.unsharp(R.frames_to_tensor(list(frames.numpy())), 0.6, zero_border).contiguous(), 0.05, 0.4, 1234, 17,
                              noise_fn=noise_fn)
        _frames_eq(got.cpu().numpy(), np.stack(R.tensor_to_frames(want)), "uint8 sharpen -> seeded grain vs oracle")
        half = shape[0] // 2
        if half:
            split = torch.cat((ops.sharpen_then_seeded_grain(x[:half], 0.6, zero_border, 0.05, 0.4, 1234, 17),
                               ops.sharpen_then_seeded_grain(x[half:], 0.6, zero_border, 0.05, 0.4, 1234, 17 + half)))
            assert torch.equal(split, got)


def test_u8_sharpen_grain_refuses_what_it_does_not_take(pkg, ops, dev):
    """What is left of the refusals after round 5 (any width, height and alignment is taken now): several frames per noise chunk (the
    enhancer seeds every frame on its own), a batch of fewer than four bytes (one 1 x 1 frame), in-place -- the entry point says so and the
    operator returns the converter route's bytes; one effect off takes the reference's early returns."""
    from comfyui_vrgamedevgirl_amd import _hip, rng
    import ctypes as C
    lib = _hip.lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (2, 20, 512, 3), generator=g, dtype=torch.uint8).to(dev)
    d = ops.NoisePlan(2, rng.per_frame_seeded(x.numel(), 3, dev)).desc()
    assert lib.vrg_sharpen_grain_u8(_hip.ptr(x), _hip.ptr(torch.empty_like(x)), 2, 20, 512, 0.5, 0, 0.04, 0.5, 0.5, C.byref(d),
                                    _hip.current_stream()) == _hip.VRG_ERR_UNSUPPORTED
    d1 = ops.NoisePlan(1, rng.per_frame_seeded(x[0].numel(), 3, dev)).desc()
    assert lib.vrg_sharpen_grain_u8(_hip.ptr(x), _hip.ptr(x), 2, 20, 512, 0.5, 0, 0.04, 0.5, 0.5, C.byref(d1), _hip.current_stream()) == _hip.VRG_ERR_BAD_ARG
    one = torch.tensor([[[[7, 200, 90]]]], dtype=torch.uint8, device=dev)                  # 3 bytes: no dword to load
    d3 = ops.NoisePlan(1, rng.per_frame_seeded(3, 3, dev)).desc()
    assert lib.vrg_sharpen_grain_u8(_hip.ptr(one), _hip.ptr(torch.empty_like(one)), 1, 1, 1, 0.5, 0, 0.04, 0.5, 0.5, C.byref(d3),
                                    _hip.current_stream()) == _hip.VRG_ERR_UNSUPPORTED
    want = ops.f32_to_frames_u8(ops.sharpen_then_seeded_grain(ops.frames_u8_to_f32(one), 0.5, False, 0.04, 0.5, 3, 0))
    assert torch.equal(ops.sharpen_then_seeded_grain(one, 0.5, False, 0.04, 0.5, 3, 0), want)
    for strength, intensity in ((0.0, 0.05), (0.6, 0.0), (0.0, 0.0)):       # one effect (or both) off: the reference's early returns
        y = ops.frames_u8_to_f32(x)
        if strength > 0:
            y = ops.stencil3x3(y, "unsharp", strength, True)
        if intensity > 0:
            y = ops.film_grain_seeded_frames(y, intensity, 0.5, 3, 0)
        assert torch.equal(ops.sharpen_then_seeded_grain(x, strength, True, intensity, 0.5, 3, 0), ops.f32_to_frames_u8(y))


def test_u8_sharpen_grain_off_the_dword_grid(pkg, ops, dev):
    """Frames that do not start on a dword (a slice of a batch whose frames hold an odd number of bytes) and whose windows reach into
    neighbouring frames of a larger allocation: the bytes of the converter route, and nothing outside the frames is written."""
    g = torch.Generator().manual_seed(5)
    for shape in ((6, 5, 7, 3), (5, 3, 343, 3), (4, 9, 1, 3)):
        big = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).to(dev)
        x = big[1:-1]
        assert x.data_ptr() % 4 != 0 or shape[1] * shape[2] * 3 % 4 == 0
        for zero in (False, True):
            want = ops.f32_to_frames_u8(ops.sharpen_then_seeded_grain(ops.frames_u8_to_f32(x.clone()), 0.7, zero, 0.06, 0.3, 99, 4))
            got = ops.sharpen_then_seeded_grain(x, 0.7, zero, 0.06, 0.3, 99, 4)
            assert torch.equal(got, want), (shape, zero)
    # guard bytes around the output stay untouched
    x = torch.randint(0, 256, (3, 5, 7, 3), generator=g, dtype=torch.uint8).to(dev)
    from comfyui_vrgamedevgirl_amd import _hip, rng
    import ctypes as C
    fe = 5 * 7 * 3
    buf = torch.full((fe * 3 + 16,), 0xAB, dtype=torch.uint8, device=dev)
    d = ops.NoisePlan(1, rng.per_frame_seeded(fe, 7, dev)).desc()
    st = _hip.lib().vrg_sharpen_grain_u8(_hip.ptr(x), C.c_void_p(buf.data_ptr() + 5), 3, 5, 7, 0.5, 0, 0.04, 0.5, float(np.float32(0.5)), C.byref(d),
                                         _hip.current_stream())
    assert st == _hip.VRG_OK
    assert bool((buf[:5] == 0xAB).all()) and bool((buf[5 + 3 * fe:] == 0xAB).all())
    assert torch.equal(buf[5:5 + 3 * fe].view(3, 5, 7, 3), ops.sharpen_then_seeded_grain(x, 0.5, False, 0.04, 0.5, 7, 0))


# ---------------------------------------------------------------------------------------- opening colour match (8f-4)
def test_u8_channel_sums_are_exact(pkg, dev):
    from comfyui_vrgamedevgirl_amd import _hip
    for shape in ((3, 37, 53, 3), (1, 1, 1, 3), (2, 2160, 3840, 3)):
        g = torch.Generator().manual_seed(sum(shape))
        frames = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        x = frames.to(dev)
        sums = torch.full((shape[0], 3, 2), -1, dtype=torch.int64, device=dev)
        _hip.check(_hip.lib().vrg_u8_channel_sums(_hip.ptr(x), shape[0], shape[1], shape[2], _hip.ptr(sums), _hip.current_stream()), "sums")
        a = frames.numpy().reshape(shape[0], -1, 3).astype(np.uint64)
        want = np.stack([a.sum(axis=1), (a * a).sum(axis=1)], axis=-1).astype(np.int64)
        assert np.array_equal(sums.cpu().numpy(), want), shape


@pytest.mark.parametrize("n,dmin,dmax", [(17, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), (33, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)),
                                         (2, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), (9, (0.0, 0.0, 0.0), (2.0, 4.0, 0.5))])
def test_ffmpeg_style_lut3d_and_blend_equal_their_restatement(ops, dev, n, dmin, dmax):
    """vrg_lut3d_tetra_u8 vs oracle.restated.ffmpeg_lut3d_blend_u8 (ffmpeg's published lut3d / blend arithmetic restated;
    ffmpeg itself is absent: parity unpinned): every byte value on every axis, random frames, per-frame weights."""
    g = torch.Generator().manual_seed(100 + n)
    data = {"size": n, "lut": torch.rand((n, n, n, 3), generator=g) * 1.2 - 0.1, "domain_min": torch.tensor(dmin), "domain_max": torch.tensor(dmax)}
    lut = ops.upload_lut(data, dev)
    ramp = torch.arange(256, dtype=torch.uint8)
    axes = torch.stack([torch.stack([ramp, ramp.flip(0), ramp * 7], -1), torch.stack([ramp, ramp, ramp], -1),
                        torch.stack([ramp * 3, ramp, ramp.flip(0)], -1)], 0).reshape(3, 16, 16, 3)
    frames = torch.cat([axes, torch.randint(0, 256, (3, 16, 16, 3), generator=g, dtype=torch.uint8)], 0).contiguous()
    for weights in (None, [1.0, 0.85, 0.5, 0.3333333333333333, 0.0, 0.07]):
        got = ops.lut3d_ffmpeg_u8(frames.to(dev), lut, weights)
        _frames_eq(got.cpu().numpy(), R.ffmpeg_lut3d_blend_u8(frames.numpy(), data, weights), f"ffmpeg-style lut3d {n}^3 weights={weights}")
    assert ops.lut3d_ffmpeg_u8(frames[:0].to(dev), lut).shape == (0, 16, 16, 3)
    with pytest.raises(ValueError):
        ops.lut3d_ffmpeg_u8(frames.to(dev), lut, [1.0])


def test_opening_colour_match_statistics_and_frames(pkg, ops, dev):
    import json
    from comfyui_vrgamedevgirl_amd import VRGDG_WorkflowRunnerNodes as WR
    z = _npz("opening_match.npz")
    with open(os.path.join(GOLDEN, "opening_match.json")) as fh:
        meta = json.load(fh)
    for name, m in meta.items():
        ref_bgr = np.ascontiguousarray(z[f"{name}.ref"][..., ::-1])
        tgt_bgr = np.ascontiguousarray(z[f"{name}.tgt"][..., ::-1])
        rs, ts = WR._frame_channel_stats(ref_bgr), WR._frame_channel_stats(tgt_bgr)
        assert rs[0] == m["reference_mean"] and ts[0] == m["target_mean"], name          # PIL.ImageStat, bit for bit (doubles)
        assert [max(1.0, v) for v in rs[1]] == m["reference_std"] and [max(1.0, v) for v in ts[1]] == m["target_std"], name
        # the clip: 6 frames shaped like the target, fps 4 -> weights fade over the first frames
        g = np.random.default_rng(5)
        clip = [g.integers(0, 256, tgt_bgr.shape, dtype=np.uint8) for _ in range(6)]
        clip[0] = tgt_bgr
        out, info = WR._apply_scene_start_color_match_frames(clip, ref_bgr, fps=4.0, fade_seconds=m["fade_seconds"], strength=m["strength"])
        assert info["applied"] and info["scales"] == m["scales"] and info["offsets"] == m["offsets"]
        assert hashlib_sha256(info["cube_text"]) == m["cube_sha256"]
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "c.cube")
            with open(path, "w", encoding="utf-8", newline="\n") as fh:
                fh.write(info["cube_text"])
            lut = R.parse_cube_file(path)
        fade = max(0.05, min(30.0, m["fade_seconds"]))
        for i, frame in enumerate(clip):
            w = R.opening_match_weight(i, 4.0, m["strength"], fade)
            assert info["weights"][i] == w
            want = frame if w <= 0.0 else R.tensor_to_frames(R.apply_lut_with_strength(R.frames_to_tensor([frame]), lut, 10.0 * w))[0]
            _frames_eq(out[i], want, f"opening match {name} frame {i}")
        # the same clip with ffmpeg's filter arithmetic (restated): equal to its oracle, and -- the cube being affine per
        # channel below the clamp -- within ONE 8-bit step of the node arithmetic (ffmpeg truncates twice, the nodes once)
        out_ff, info_ff = WR._apply_scene_start_color_match_frames(clip, ref_bgr, fps=4.0, fade_seconds=m["fade_seconds"],
                                                                   strength=m["strength"], filter_arithmetic="ffmpeg")
        assert info_ff["weights"] == info["weights"] and info_ff["cube_text"] == info["cube_text"]
        want_ff = R.ffmpeg_lut3d_blend_u8(np.stack(clip, 0), lut, info["weights"])
        _frames_eq(np.stack(out_ff, 0), want_ff, f"opening match {name}, ffmpeg arithmetic")
        step = np.abs(np.stack(out_ff, 0).astype(np.int32) - np.stack(out, 0).astype(np.int32)).max()
        _record(f"opening_match.{name}.ffmpeg_vs_nodes_max_8bit_steps", int(step))
        assert step <= 2, step
    # the reference's `float(payload.get("strength", 0.85) or 0.85)`: 0 is falsy -> 0.85; only a negative value clamps to 0
    _, info = WR._apply_scene_start_color_match_frames(clip, ref_bgr, fps=4.0, strength=0.0)
    assert info["applied"] and info["weights"][0] == 0.85
    same, info = WR._apply_scene_start_color_match_frames(clip, ref_bgr, fps=4.0, strength=-1.0)
    assert info == {"applied": False, "reason": "strength is zero"}
    _frames_eq(same, clip, "strength <= 0 leaves the clip untouched")


def hashlib_sha256(text):
    import hashlib
    return hashlib.sha256(text.encode()).hexdigest()


# ---------------------------------------------------------------------------------------- randomized sweep
@pytest.mark.parametrize("seed", range(int(os.environ.get("VRG_SWEEP_SEEDS", "24"))))
def test_randomized_chain_and_adjust_sweep(ops, pkg, dev, seed):
    """Random shapes, stage subsets, strengths and slider sets (fixed seeds): fp32 and uint8 entry points against the
    CPU oracle, bit for bit.  Covers combinations the hand-written lists do not."""
    import random
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    rnd = random.Random(1000 + seed)
    F, H, W = rnd.randint(1, 5), rnd.randint(1, 90), rnd.randint(1, 150)
    name = rnd.choice(["AMD_TealOrange_33.cube", "AMD_WarmFilm_25.cube", "AMD_Identity_17.cube"])
    data, dlut = _lut_pair(ops, dev, name)
    grain = (round(rnd.uniform(0.001, 0.6), 3), round(rnd.uniform(0, 1), 2), rnd.choice([0, 1, 2, 3])) if rnd.random() < 0.7 else None
    lut_s = rnd.choice([10.0, round(rnd.uniform(0.1, 9.9), 1)]) if rnd.random() < 0.7 else None
    sharpen = (rnd.choice(["unsharp", "laplacian", "sobel"]), round(rnd.uniform(0.05, 2.0), 2), False) if rnd.random() < 0.7 else None
    if grain is None and lut_s is None and sharpen is None:
        lut_s = 10.0
    use_u8 = rnd.random() < 0.5
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(0, 256, (F, H, W, 3), generator=g, dtype=torch.uint8)
    x = R.frames_to_tensor(list(frames.numpy())) if use_u8 else torch.rand((F, H, W, 3), generator=g) * 1.1 - 0.05
    spec = ops.ChainSpec(grain=grain, lut=(dlut, lut_s) if lut_s is not None else None, sharpen=sharpen, variant=rnd.choice([0, 1, 2]))
    torch.manual_seed(seed)
    got = ops.fused_chain(frames.to(dev) if use_u8 else x.to(dev), spec)
    torch.manual_seed(seed)
    o = x
    if grain:
        o = R.fast_film_grain(o, grain[0], grain[1], grain[2], noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    if lut_s is not None:
        o = R.apply_lut_with_strength(o, data, lut_s)
    if sharpen:
        o = {"unsharp": R.unsharp, "laplacian": R.laplacian, "sobel": R.sobel}[sharpen[0]](o, sharpen[1], False)
        o = o.contiguous()
    if use_u8:
        _frames_eq(got.cpu().numpy(), R.tensor_to_frames(o), f"sweep {seed} u8 {spec}")
    else:
        assert_bit_equal(got, o, f"sweep {seed} {spec}")
    # Adjust with a random slider subset on the same frames
    keys = ["temperature", "tint", "saturation", "exposure", "contrast", "highlights", "shadows", "whites", "blacks", "sharpen",
            "clarity", "vignette", "fade"]
    settings = {k: round(rnd.uniform(-100, 100), 1) for k in rnd.sample(keys, rnd.randint(1, 8))}
    terms = ops.adjust_terms(LVT._normalize_adjust_settings(settings))
    if use_u8:
        want = R.tensor_to_frames(_adjust_want(x, settings))
        _frames_eq(ops.adjust(frames.to(dev), terms).cpu().numpy(), want, f"sweep {seed} adjust u8 {settings}")
    else:
        assert_bit_equal(ops.adjust(x.to(dev), terms), _adjust_want(x, settings), f"sweep {seed} adjust {settings}")


def test_nodes_are_reentrant_across_host_threads(pkg, dev):
    """ComfyUI's worker, the aiohttp routes and the enhancer's daemon thread may enter the kernels concurrently
    (threading): no shared mutable state besides the LUT cache and the staging ring."""
    import threading
    from comfyui_vrgamedevgirl_amd import nodes, VRGDG_IV_Adjustments as iv, VRGDG_LUTVideoTools as LVT
    inputs = [_rand((3, 40 + 7 * i, 64 + 5 * i, 3), 900 + i) for i in range(6)]

    def work(x):
        a = nodes.FastUnsharpSharpen().apply_unsharp(x, 0.8, False)[0]
        b = iv.VRGDG_LUTS().apply_lut(x, "AMD_WarmFilm_25.cube", "auto", 6.0)[0]
        c = LVT._apply_adjust_tensor(x, {"clarity": 30, "sharpen": 20, "vignette": 40}, "cuda").cpu()
        return a, b, c

    want = [work(x) for x in inputs]
    got = [None] * len(inputs)
    errors = []

    def run(i):
        try:
            for _ in range(3):
                got[i] = work(inputs[i])
        except Exception as exc:      # surfaced below
            errors.append(exc)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(inputs))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for w, g in zip(want, got):
        for a, b in zip(w, g):
            assert torch.equal(a, b)


def test_baseline_config_1_single_512_frame(pkg, ops, dev):
    """BASELINE.json configs[0]: Fast Film Grain on one 512x512 RGB frame, torch.manual_seed(0), I=0.04, s=0.5, bs=4.
    The reference's CPU noise (mt19937) cannot be reproduced in parallel, so the plumbing check injects it: the
    reference node's own arithmetic on that noise (oracle) == the HIP kernel on the same noise, bit for bit; and the node
    itself, on the device stream, == the oracle fed with torch's device noise."""
    from comfyui_vrgamedevgirl_amd import nodes
    torch.manual_seed(0)
    x = torch.rand(1, 512, 512, 3)
    noise = torch.randn(1, 512, 512, 3)
    assert_bit_equal(ops.film_grain_injected(x.to(dev), noise.to(dev), 0.04, 0.5), R.grain_apply(x, noise, 0.04, 0.5), "config 1, injected noise")
    torch.manual_seed(0)
    (got,) = nodes.FastFilmGrain().apply_grain(x, 0.04, 0.5, 4)
    torch.manual_seed(0)
    want = R.fast_film_grain(x, 0.04, 0.5, 4, noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    assert_bit_equal(got, want, "config 1, node on the device stream")


def test_baseline_scale_batch_crosses_32bit_element_counts(ops, dev):
    """BASELINE.json's per-GPU shard (256 x 4K frames = 6.4e9 elements, beyond 2^32): results over the whole batch equal
    results over frame ranges processed alone -- for the noise stream (absolute chunk index), the LUT gathers, the
    stencil, the statistics and the Adjust kernels.  Exercises the 64-bit frame addressing and the per-launch splitting."""
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 120 << 30:
        pytest.skip("needs ~100 GB of free HBM")
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    data, dlut = _lut_pair(ops, dev)
    F, H, W = 264, 2160, 3840                                   # 6.57e9 elements
    x = torch.empty((F, H, W, 3), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev).manual_seed(8)
    for i in range(0, F, 24):
        x[i:i + 24].copy_(torch.rand((min(24, F - i), H, W, 3), generator=g, device=dev))
    ref_ms = ops.finalize_stats(ops.lab_stats(x[:1]))
    spec = ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(dlut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False))
    gen = torch.Generator(device=dev).manual_seed(77)
    whole = ops.fused_chain(x, spec, generator=gen)
    probes = [(0, 4), (128, 132), (212, 216), (260, 264)]        # 212*H*W*3 > 2^32 elements
    fe = H * W * 3
    for a, b in probes:
        gen2 = torch.Generator(device=dev).manual_seed(77)
        stream = ops.rng.reserve(4 * fe, F // 4, dev, gen2)      # the same job-wide stream; this range starts at chunk a/4
        part = ops.fused_chain(x[a:b], spec, plans=(ops.NoisePlan(4, stream, chunk0=a // 4), None, 1))
        assert torch.equal(part, whole[a:b]), (a, b)
    del whole
    t = ops.adjust_terms(LVT._normalize_adjust_settings({"clarity": 30, "sharpen": 20, "vignette": 35, "exposure": 10}))
    whole = ops.adjust(x, t)
    for a, b in probes:
        assert torch.equal(ops.adjust(x[a:b], t), whole[a:b]), (a, b)


def test_grain_rng_chunks_beyond_32bit_byte_indexing(pkg, ops, dev):
    """batch_size = 0 (or >= 22 4K / >= 87 1080p frames per chunk): the reference's randn_like has more than 2^29 elements and
    ATen runs it as several kernels over 32-bit indexable sub-ranges, each with its own grid and generator offset
    (nodes.py:46-51 works there).  Same noise, same result, same generator state afterwards as torch itself."""
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 24 << 30:
        pytest.skip("needs ~20 GB of free HBM")
    F, H, W = 91, 1080, 1920                                     # 566,092,800 elements > 2^29: two leaves of 283,046,400
    fe = H * W * 3
    assert F * fe > 2 ** 29
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand((F, H, W, 3), generator=g, device=dev)
    torch.manual_seed(2024)
    got = ops.film_grain(x, 0.04, 0.5, chunk_frames=0)
    state_after = torch.cuda.default_generators[dev.index].get_offset()
    torch.manual_seed(2024)
    noise = torch.randn_like(x)                                   # the reference's draw (nodes.py:51)
    assert torch.cuda.default_generators[dev.index].get_offset() == state_after, "generator offset after an oversize chunk"
    want = ops.film_grain_injected(x, noise, 0.04, 0.5)
    assert torch.equal(got, want)
    del noise, want
    # the fused chain takes the same route for such chunks; and a ragged tail chunk below the limit after an oversize one
    torch.manual_seed(2024)
    fused = ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 0), sharpen=("unsharp", 0.5, False)))
    assert torch.equal(fused, ops.stencil3x3(got, "unsharp", 0.5, False))
    del fused, got
    torch.manual_seed(11)
    a = ops.film_grain(x, 0.1, 0.3, chunk_frames=88)              # chunks: 88 frames (oversize) + 3 frames
    torch.manual_seed(11)
    n1, n2 = torch.randn_like(x[:88]), torch.randn_like(x[88:])
    assert torch.equal(a[:88], ops.film_grain_injected(x[:88], n1, 0.1, 0.3)) and torch.equal(a[88:], ops.film_grain_injected(x[88:], n2, 0.1, 0.3))


@pytest.mark.parametrize("n", [2, 3, 5, 64])
def test_lut_extreme_sizes(ops, dev, n):
    """Smallest legal cubes (one or two cells per axis) and a large one (64^3: 9.4 MB of records, beyond one XCD's L2)."""
    g = torch.Generator().manual_seed(n)
    table = torch.rand((n, n, n, 3), generator=g)
    data = {"size": n, "lut": table, "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}
    x = torch.cat([_rand((2, 33, 47, 3), 5 + n, -0.2, 1.2).reshape(1, 1, -1, 3),
                   torch.tensor([0.0, 1.0, 0.5, 1.0 / max(n - 1, 1), 0.999999]).repeat(3, 1).t().reshape(1, 1, -1, 3)], dim=2)
    dlut = ops.upload_lut(data, dev)
    for s in (10.0, 3.7):
        assert_bit_equal(ops.lut3d(x.to(dev), dlut, s), R.apply_lut_with_strength(x, data, s), f"lut {n}^3 strength {s}")
    u8 = torch.randint(0, 256, (2, 19, 23, 3), generator=g, dtype=torch.uint8)
    got = ops.fused_chain(u8.to(dev), ops.ChainSpec(lut=(dlut, 10.0)))
    _frames_eq(got.cpu().numpy(), R.tensor_to_frames(R.apply_lut_with_strength(R.frames_to_tensor(list(u8.numpy())), data, 10.0)), f"u8 lut {n}^3")


@pytest.mark.parametrize("seed", range(int(os.environ.get("VRG_SWEEP_SEEDS", "24"))))
def test_randomized_colour_match_chain_sweep(ops, pkg, dev, seed):
    """Random shapes / reference batches / batch sizes / strengths for chains that contain colour match, against the oracle with its
    colour match evaluated on the device: BIT-EQUAL (element-wise path and torch-order statistics, on thumbnails of 6..80 px where
    planes are unaligned and the reductions take their small-frame geometries).  The fp64-statistics variant of the same chain keeps
    its band (an ulp of a mean moves every pixel of such a frame)."""
    import random
    rnd = random.Random(5000 + seed)
    H, W = rnd.randint(6, 80), rnd.randint(6, 120)
    n_ref = rnd.choice([1, 1, 2, 3])
    F = n_ref * rnd.randint(1, 3)
    data, dlut = _lut_pair(ops, dev, rnd.choice(["AMD_TealOrange_33.cube", "AMD_WarmFilm_25.cube"]))
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((F, H, W, 3), generator=g)
    ref = torch.rand((n_ref, rnd.randint(4, 40), rnd.randint(4, 40), 3), generator=g)
    k = round(rnd.uniform(0.05, 1.0), 2)
    grain = (round(rnd.uniform(0.01, 0.2), 3), round(rnd.uniform(0, 1), 2), n_ref) if rnd.random() < 0.6 else None
    lut_s = rnd.choice([10.0, 6.5]) if rnd.random() < 0.6 else None
    sharpen = ("unsharp", round(rnd.uniform(0.1, 1.5), 2), False) if rnd.random() < 0.6 else None
    ref_ms = ops.reference_stats(ref.to(dev))
    spec = ops.ChainSpec(grain=grain, lut=(dlut, lut_s) if lut_s is not None else None, colormatch=(ref_ms, k), sharpen=sharpen, cm_chunk=n_ref)
    torch.manual_seed(seed)
    got = ops.fused_chain(x.to(dev), spec)
    import dataclasses
    torch.manual_seed(seed)
    got64 = ops.fused_chain(x.to(dev), dataclasses.replace(spec, colormatch=(ops.reference_stats(ref.to(dev), cm_stats="fp64"), k), cm_stats="fp64"))
    torch.manual_seed(seed)
    o = x
    if grain:
        o = R.fast_film_grain(o, grain[0], grain[1], grain[2], noise_fn=lambda i, shp: torch.randn(shp, device=dev).cpu())
    if lut_s is not None:
        o = R.apply_lut_with_strength(o, data, lut_s)
    o = R.color_match(o.to(dev), ref.to(dev), k, n_ref).cpu().contiguous()
    if sharpen:
        o = R.unsharp(o, sharpen[1], False).contiguous()
    assert_bit_equal(got, o, f"seed {seed}: {F} frames {H}x{W}, {n_ref} reference(s), grain {grain}, lut {lut_s}, sharpen {sharpen}")
    err = _unit_ulps(got64, o)
    amp = 1.0 + (sharpen[1] * 2 if sharpen else 0.0)           # unsharp amplifies a difference by up to 1 + 2*strength*(8/9)
    _record("e2e.random_sweep_fp64stats_vs_device_oracle_over_amp", err / amp)
    assert err <= CM_CROSS_REF_ULP * amp, (seed, err, spec)


@pytest.mark.parametrize("n", [2, 5, 9, 17, 21, 22])
def test_lut_lds_resident_path(ops, dev, n):
    """Cubes whose node table fits the LDS (N <= 21) take the LDS-resident kernel once there are >= 65536 pixels;
    22^3 is the first size that stays on the global-memory gather kernel.  Both must equal the oracle bit for bit,
    for unit and non-unit domains and for blended strengths."""
    g = torch.Generator().manual_seed(100 + n)
    table = torch.rand((n, n, n, 3), generator=g)
    x = torch.cat([_rand((2, 200, 170, 3), 9 + n, -0.2, 1.3).reshape(1, 1, -1, 3),
                   torch.tensor([0.0, 1.0, 0.5, 1.0 / max(n - 1, 1), 0.999999]).repeat(3, 1).t().reshape(1, 1, -1, 3)], dim=2)
    assert x.shape[2] >= 65536
    for dmin, dmax in ((torch.zeros(3), torch.ones(3)), (torch.tensor([-0.25, 0.0, 0.125]), torch.tensor([1.5, 1.0, 0.875]))):
        data = {"size": n, "lut": table, "domain_min": dmin, "domain_max": dmax}
        dlut = ops.upload_lut(data, dev)
        for s in (10.0, 4.2):
            assert_bit_equal(ops.lut3d(x.to(dev), dlut, s), R.apply_lut_with_strength(x, data, s), f"lut {n}^3 strength {s}")


def test_lut_lds_resident_path_uint8_and_chain_entry(ops, dev):
    """The LDS-resident small-cube kernel behind the fused-chain entry points (LUT-only chains on fp32 and on uint8 frames:
    the route's _process_video_batch and the opening colour match use exactly this with 17^3 cubes)."""
    data, dlut = _lut_pair(ops, dev, "AMD_Identity_17.cube")
    g = torch.Generator().manual_seed(17)
    table = torch.rand((17, 17, 17, 3), generator=g)
    data2 = {"size": 17, "lut": table, "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}
    dlut2 = ops.upload_lut(data2, dev)
    u8 = torch.randint(0, 256, (3, 160, 200, 3), generator=g, dtype=torch.uint8)          # 96,000 pixels >= 65,536
    x = R.frames_to_tensor(list(u8.numpy()))
    for d, dl in ((data, dlut), (data2, dlut2)):
        for s in (10.0, 3.3):
            want = R.apply_lut_with_strength(x, d, s)
            assert_bit_equal(ops.fused_chain(x.to(dev), ops.ChainSpec(lut=(dl, s))), want, f"chain entry, LDS LUT, strength {s}")
            _frames_eq(ops.fused_chain(u8.to(dev), ops.ChainSpec(lut=(dl, s))).cpu().numpy(), R.tensor_to_frames(want), f"u8 LDS LUT {s}")


@pytest.mark.parametrize("lut_name", ["AMD_Identity_17.cube", None])
def test_march_with_lds_resident_lut(ops, dev, lut_name):
    """grain -> LUT -> sharpen with a cube of at most 21^3 on enough frames: the march kernel stages the node table in
    LDS (12-wave workgroups).  Must equal the LDS-tile kernel with the global record table (itself oracle-checked) and,
    on a slab, the CPU oracle."""
    if lut_name is None:
        g = torch.Generator().manual_seed(3)
        data = {"size": 21, "lut": torch.rand((21, 21, 21, 3), generator=g), "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}
        dlut = ops.upload_lut(data, dev)
    else:
        data, dlut = _lut_pair(ops, dev, lut_name)
    gd = torch.Generator(device=dev).manual_seed(12)
    x = torch.rand((8, 2160, 3840, 3), generator=gd, device=dev)
    for grain in ((0.05, 0.5, 4), None):
        for sharpen in (("unsharp", 0.6, False), None):
            if grain is None and sharpen is None:
                continue
            gen = torch.Generator(device=dev).manual_seed(5)
            a = ops.fused_chain(x, ops.ChainSpec(grain=grain, lut=(dlut, 8.5), sharpen=sharpen, variant=2), generator=gen)
            gen = torch.Generator(device=dev).manual_seed(5)
            b = ops.fused_chain(x, ops.ChainSpec(grain=grain, lut=(dlut, 8.5), sharpen=sharpen, variant=1), generator=gen)
            assert torch.equal(a, b), (grain, sharpen)
            del b
            gen = torch.Generator(device=dev).manual_seed(5)
            c = ops.fused_chain(x, ops.ChainSpec(grain=grain, lut=(dlut, 8.5), sharpen=sharpen), generator=gen)      # automatic choice
            assert torch.equal(a, c), (grain, sharpen, "auto")
            del c
    cpu = x[0:1, 0:4].cpu()
    want = R.unsharp(R.apply_lut_with_strength(cpu, data, 10.0), 0.5, False)
    got = ops.fused_chain(x, ops.ChainSpec(lut=(dlut, 10.0), sharpen=("unsharp", 0.5, False), variant=2))
    assert torch.equal(got[0, 0:3].cpu(), want[0, 0:3])


# ---------------------------------------------------------------------------------------- the benchmark's own geometry, held to the oracle
def _job_stream(ops, dev, fe, chunk, n_chunks, seed=42):
    """bench.py's job-wide noise stream: one generator, `n_chunks` chunks of `chunk` frames (absolute chunk index -> offset)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    return ops.rng.reserve(chunk * fe, n_chunks, dev, gen)


def test_headline_chain_at_bench_geometry_is_the_oracle(ops, dev):
    """BASELINE configs[4] as bench.py runs it -- 4K frames, grain chunk 4, AMD_TealOrange_33, colour match batch_size 1, unsharp 0.5,
    noise keyed by absolute chunk index, caller-supplied Lab workspace and output -- on 8 frames, against the oracle composition
    (oracle/chain_oracle.py: torch.randn on the device -> restated grain / LUT on the CPU -> restated colour match evaluated by torch on
    the device -> restated unsharp), BIT FOR BIT.  At 4K a chunk's Philox quarter G = 524,288 elements spans 11.4 rows and a chunk holds
    47.5 quarters per frame: the four sibling runs of k_produce_lab's workgroups cross frame boundaries inside every chunk, and the
    kernels that run are exactly the bench's: k_produce_lab<3,false,false> -> k_tstats_frame -> k_chain_tile<COLORMATCH|FROM_LAB>."""
    from oracle import chain_oracle as CO
    H, W, chunk, frames, first_chunk = 2160, 3840, 4, 8, 5             # chunks 5 and 6 of a larger job (a rank that is not rank 0)
    data, dlut = _lut_pair(ops, dev, "AMD_TealOrange_33.cube")
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.rand((frames, H, W, 3), generator=g, device=dev)
    ref = torch.rand((1, H, W, 3), generator=g, device=dev)
    stream = _job_stream(ops, dev, H * W * 3, chunk, 64)
    out, ws = torch.empty_like(x), torch.empty_like(x)
    ref_ms, ev = ops.reference_stats_async(ref)
    spec = ops.ChainSpec(grain=(0.04, 0.5, chunk), lut=(dlut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), cm_chunk=1,
                         cm_ref_event=ev)
    ops.fused_chain(x, spec, plans=(ops.NoisePlan(chunk, stream, chunk0=first_chunk), None, frames // chunk), out=out, lab_workspace=ws)
    want = CO.headline_chain(x.cpu(), dev, stages=("grain", "lut", "colormatch", "sharpen"), stream=stream, chunk0=first_chunk,
                             chunk_frames=chunk, lut_cpu=data, reference_dev=ref, cm_batch=1)
    assert_bit_equal(out, want, "4K x 8 frames, chunk 4: fused headline chain vs the oracle composition")


def test_grain_lut_1080p_at_bench_geometry_is_the_oracle(ops, dev):
    """BASELINE configs[1]: 1080p, grain (chunk 4) + 33^3 LUT, job-wide noise stream -- 8 frames against the oracle, bit for bit."""
    from oracle import chain_oracle as CO
    H, W, chunk, frames, first_chunk = 1080, 1920, 4, 8, 3
    data, dlut = _lut_pair(ops, dev, "AMD_TealOrange_33.cube")
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.rand((frames, H, W, 3), generator=g, device=dev)
    stream = _job_stream(ops, dev, H * W * 3, chunk, 32)
    out = torch.empty_like(x)
    ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, chunk), lut=(dlut, 10.0)),
                    plans=(ops.NoisePlan(chunk, stream, chunk0=first_chunk), None, frames // chunk), out=out)
    want = CO.headline_chain(x.cpu(), dev, stages=("grain", "lut"), stream=stream, chunk0=first_chunk, chunk_frames=chunk, lut_cpu=data)
    assert_bit_equal(out, want, "1080p x 8 frames, chunk 4: fused grain + LUT vs the oracle composition")


def test_chain3_4k_at_bench_geometry_is_the_oracle(ops, dev):
    """BASELINE configs[2]: 4K, grain (chunk 4) + LUT + unsharp through the wave-march kernel -- 4 frames against the oracle."""
    from oracle import chain_oracle as CO
    H, W, chunk, frames, first_chunk = 2160, 3840, 4, 4, 2
    data, dlut = _lut_pair(ops, dev, "AMD_TealOrange_33.cube")
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.rand((frames, H, W, 3), generator=g, device=dev)
    stream = _job_stream(ops, dev, H * W * 3, chunk, 16)
    out = torch.empty_like(x)
    ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, chunk), lut=(dlut, 10.0), sharpen=("unsharp", 0.5, False)),
                    plans=(ops.NoisePlan(chunk, stream, chunk0=first_chunk), None, frames // chunk), out=out)
    want = CO.headline_chain(x.cpu(), dev, stages=("grain", "lut", "sharpen"), stream=stream, chunk0=first_chunk, chunk_frames=chunk, lut_cpu=data)
    assert_bit_equal(out, want, "4K x 4 frames, chunk 4: fused grain + LUT + unsharp vs the oracle composition")


def test_colour_match_4k_at_bench_geometry_is_the_device_oracle(ops, dev):
    """BASELINE configs[3]: colour match alone on 4K frames (k_lab_partials Lab-only form -> k_tstats_frame -> k_chain_tile<FROM_LAB>,
    what bench.py --workload colormatch_4k runs), 3 frames, batch_size 1, against the restated reference evaluated by torch on the GPU."""
    H, W = 2160, 3840
    g = torch.Generator(device=dev).manual_seed(4321)
    x = torch.rand((3, H, W, 3), generator=g, device=dev)
    ref = torch.rand((1, H, W, 3), generator=g, device=dev)
    out, ws = torch.empty_like(x), torch.empty_like(x)
    ref_ms, ev = ops.reference_stats_async(ref)
    ops.fused_chain(x, ops.ChainSpec(colormatch=(ref_ms, 1.0), cm_chunk=1, cm_ref_event=ev), out=out, lab_workspace=ws)
    assert_bit_equal(out, R.color_match(x, ref, 1.0, 1), "4K colour match alone vs the device oracle")


def test_bench_verifies_its_own_output(dev):
    """bench.py --verify (on by default): after the timed steps the first and the last RNG chunk of `out` are re-derived with the
    stand-alone operators and with the oracle composition; the JSON line says so."""
    import json
    import subprocess
    from conftest import ROOT
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-live-traffic", "--no-fast-variant", "--digest", "--no-host-fed"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["verified"] is True, line["verify"]
    # the other single-GPU BASELINE configs ride along as legs of the same run, each checked against the oracle after its timed steps
    cfgs = line["configs"]
    for key in ("headline", "chain4_4k.video", "chain3_4k.uniform", "chain3_4k.video", "grain_lut_1080p.uniform", "grain_lut_1080p.video",
                "colormatch_4k.uniform", "chain3_4k.uniform.25cube", "chain3_4k.video.25cube", "chain3_4k.uniform.17cube"):
        leg = cfgs[key]
        assert "error" not in leg, (key, leg)
        assert leg["verified"] is True and leg["Mpix_s"] > 0 and leg["ms_per_step"] > 0 and 0 < leg["hbm_frac"] < 1, (key, leg)
        assert leg["dominant_kernel"] and leg["dominant_kernel_ms"] <= leg["ms_per_step"] * 1.05
    assert cfgs["colormatch_4k.uniform"]["frames"] == 16 and cfgs["grain_lut_1080p.uniform"]["height"] == 1080
    assert line["roofline"]["bound"] in ("hbm", "valu-issue") and "valu_busy_frac" in line["roofline"]
    assert line["verify"]["vs_standalone_operators"] and line["verify"]["vs_device_oracle"] and line["verify"]["frames_checked"] == [[0, 4], [4, 8]]
    assert len(line["output_sha256_per_rank_per_chunk"]) == 1 and len(line["output_sha256_per_rank_per_chunk"][0]) == 2


@pytest.mark.parametrize("cm_stats", ["device", "fp64"])
def test_output_bits_do_not_depend_on_the_number_of_ranks(dev, cm_stats):
    """GPU-count invariance, proven without a second GPU: bench.py --digest emits the SHA-256 of every RNG chunk of every rank's
    output.  Two ranks x 8 frames (sharing cuda:0 over gloo: everything but the RCCL transport is the production path -- noise keyed by
    absolute chunk index, reference statistics per rank / rows split + all-reduce) must produce, chunk for chunk, the digests of ONE
    rank x 16 frames -- with the device statistics (no exchange step) AND with the fp64 statistics merged by the collective."""
    import json
    import subprocess
    from conftest import ROOT

    def run(gpus, frames):
        env = dict(os.environ, VRGDG_DIST_BACKEND="gloo")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--frames", str(frames), "--steps", "1",
                            "--warmup", "0", "--no-cpu-baseline", "--no-live-traffic", "--no-fast-variant", "--no-verify", "--digest",
                            "--same-data", "--cm-stats", cm_stats, "--no-configs", "--no-fp64-leg"], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    two, one = run(2, 8), run(1, 16)
    assert two["n_gpus"] == 2 and len(two["output_sha256_per_rank_per_chunk"]) == 2
    flat_two = [d for rank in two["output_sha256_per_rank_per_chunk"] for d in rank]
    assert flat_two == one["output_sha256_per_rank_per_chunk"][0], "2 ranks x 8 frames vs 1 rank x 16 frames: output bits differ"


@pytest.mark.parametrize("cm_stats", ["device", "fp64"])
def test_bench_eight_rank_flow_on_one_gpu_with_gloo(dev, cm_stats):
    """The flow the driver launches on an 8-GPU node -- `bench.py --gpus 8` -> torch.distributed.run, one rank per GPU, barrier +
    max-over-ranks timing, the reference statistics per rank (device) / rows split eight ways + all-reduce (fp64) -- exercised with
    eight ranks sharing cuda:0 over gloo (8 x 3 buffers x 4 frames x 99.5 MB fit one GPU): every rank's chunk digests equal those of
    ONE rank x 32 frames, and the line carries the fields the scaling record needs.  What this cannot show: any N > 1 TIMING and the
    RCCL transport."""
    import json
    import subprocess
    from conftest import ROOT

    def run(gpus, frames):
        env = dict(os.environ, VRGDG_DIST_BACKEND="gloo")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--frames", str(frames), "--steps", "1",
                            "--warmup", "0", "--no-cpu-baseline", "--no-live-traffic", "--no-fast-variant", "--no-verify", "--digest",
                            "--same-data", "--cm-stats", cm_stats, "--no-configs", "--no-fp64-leg"], capture_output=True, text=True, env=env, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    eight, one = run(8, 4), run(1, 32)
    assert eight["n_gpus"] == 8 and eight["rccl_ranks"] == 8 and eight["scaling"] == "weak"
    assert len(eight["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in eight["per_rank_ms_per_step"])
    assert eight["ms_per_step"] >= max(eight["per_rank_ms_per_step"]) - 1e-6              # the line's time is the max over ranks
    assert "roofline" in eight and eight["roofline"]["bound"] in ("hbm", "valu-issue") and eight["config"]["frames_per_gpu"] == 4
    assert len(eight["output_sha256_per_rank_per_chunk"]) == 8
    flat = [d for rank in eight["output_sha256_per_rank_per_chunk"] for d in rank]
    assert flat == one["output_sha256_per_rank_per_chunk"][0], "8 ranks x 4 frames vs 1 rank x 32 frames: output bits differ"


def test_device_statistics_of_a_split_call_use_the_whole_calls_mean_factor(ops, dev):
    """ATen forms the mean factor float(outputs) / numel ONCE from the whole reduction call and hands it to every 32-bit sub-iterator;
    per piece it would differ by one ulp whenever outputs * H*W is not an fp32 number -- e.g. batch_size 87 at 1079 x 1919 (odd H and
    W: planes also start off the vector boundary)."""
    F, H, W = 87, 1079, 1919
    assert F * 3 * H * W > 2 ** 29
    g = torch.Generator(device=dev).manual_seed(5)
    lab = torch.empty((F, H, W, 3), device=dev)
    for i in range(0, F, 8):
        lab[i:i + 8] = torch.rand((min(8, F - i), H, W, 3), generator=g, device=dev) * 100 - 35
    o = F * 3
    assert np.float32(o) / np.float32(o * H * W) != np.float32(o // 2) / np.float32((o // 2) * H * W) or \
        np.float32(o) / np.float32(o * H * W) != np.float32(o - o // 2) / np.float32((o - o // 2) * H * W), "pick a size where the factors differ"
    got = ops.lab_stats_device(lab, F)
    want = _torch_reductions(lab, F)
    assert _same_bits_or_nan(got, want), (got - want).abs().max()


def test_node_path_over_several_gpu_lanes_equals_one_device(pkg, dev, monkeypatch):
    """VRGDG_DEVICES: a host-fed batch goes round-robin over several GPUs (ComfyUI runs its graph in ONE process: this, not torchrun,
    is how a node reaches the other GPUs of a node).  The 1-GPU test box exercises it with the device list [cuda:0, cuda:0, cuda:0] --
    three lanes with their own upload / compute / download streams: per-lane LUT tables and reference statistics, grain noise
    reserved from the primary generator in submission order -- and the results (and the generator afterwards) must equal the
    single-device path bit for bit."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
    x = _rand((13, 48, 80, 3), 223)
    ref = _rand((1, 20, 30, 3), 224)
    ref3 = _rand((3, 20, 30, 3), 225)
    frame_bytes = x[0].numel() * 4
    calls = {
        "grain bs=2": lambda: nodes.FastFilmGrain().apply_grain(x, 0.05, 0.5, 2)[0],
        "grain bs=0": lambda: nodes.FastFilmGrain().apply_grain(x, 0.05, 0.5, 0)[0],
        "sobel": lambda: nodes.FastSobelSharpen().apply_sobel(x, 0.7, False)[0],
        "colour match bs=3": lambda: nodes.ColorMatchToReference().match_color(x, ref, 0.8, 3)[0],
        "colour match, 3 references": lambda: nodes.ColorMatchToReference().match_color(x[:12], ref3, 1.0, 3)[0],
        "lut": lambda: iv.VRGDG_LUTS().apply_lut(x, "AMD_WarmFilm_25.cube", "auto", 7.0)[0],
    }
    monkeypatch.setattr(nodes, "PIPELINED", True)
    for pipe_bytes in (frame_bytes * 2, frame_bytes // 2):
        monkeypatch.setattr(_devices, "PIPE_BYTES", pipe_bytes)
        for name, fn in calls.items():
            monkeypatch.delenv("VRGDG_DEVICES", raising=False)
            torch.manual_seed(41)
            one = fn()
            after_one = torch.cuda.get_rng_state(dev)
            monkeypatch.setenv("VRGDG_DEVICES", "0,0,0")
            assert len(_devices.compute_devices()) == 3
            torch.manual_seed(41)
            many = fn()
            assert torch.equal(torch.cuda.get_rng_state(dev), after_one), name
            assert torch.equal(one, many), (name, pipe_bytes)
    monkeypatch.setenv("VRGDG_DEVICES", "all")
    assert [d.index for d in _devices.compute_devices()] == [dev.index] + [i for i in range(torch.cuda.device_count()) if i != dev.index]
    monkeypatch.setenv("VRGDG_DEVICES", "7,99")
    with pytest.raises(ValueError):
        _devices.compute_devices()


@pytest.mark.parametrize("F,H,W,b", [(1, 64, 3840, 1), (5, 8, 3840, 1), (6, 8, 3840, 2), (9, 8, 3840, 3), (32, 4, 3840, 1), (40, 4, 3840, 1), (12, 270, 480, 6),
                                     (2, 2160, 3840, 1), (1, 2160, 3840, 1), (2, 270, 482, 2), (1, 1, 2052, 1), (2, 31, 1028, 1), (2, 64, 3840, 2)])
def test_device_statistics_forms_agree_with_torch(pkg, ops, dev, F, H, W, b):
    """The whole-frame forms of the statistics replay -- one accumulator per lane (seven-wave workgroups owning 64 of torch's threads,
    <= 2 frames, scratch supplied, through the latency entry point only: the reference frame of a small step; steps that only some
    threads take, both block shapes), eight
    half-block workgroups per frame + finishing kernel (<= 32 frames, scratch supplied), four workgroups per frame (<= 64 frames), one
    workgroup per frame -- against torch's mean / std on the device, for every block shape ((256,2) / (128,4) / (64,8) = batch_size
    1 / 2 / >= 3) and ragged last calls."""
    import ctypes as C
    from comfyui_vrgamedevgirl_amd import _hip
    lab = (_rand((F, H, W, 3), 700 + F) * 130.0 - 45.0).to(dev)
    want = _torch_reductions(lab, b)
    lib = _hip.lib()
    for entry, with_scratch in (("vrg_lab_stats_torch_ws_f32", True), ("vrg_lab_stats_torch_lat_f32", True), ("vrg_lab_stats_torch_ws_f32", False),
                                ("vrg_lab_stats_torch_lat_f32", False)):
        got = torch.empty((F, 3, 2), device=dev)
        nbytes = int(lib.vrg_lab_stats_torch_scratch_bytes(F)) if with_scratch else 0
        scratch = torch.empty(max(nbytes // 4, 4), device=dev) if with_scratch else None
        _hip.check(getattr(lib, entry)(_hip.ptr(lab), F, H, W, b, _hip.ptr(got), ops._f32(1e-5), _hip.ptr(scratch) if with_scratch else None,
                                       nbytes, _hip.current_stream()), entry)
        assert _same_bits_or_nan(got, want), (entry, with_scratch, (got - want).abs().max())
    assert _same_bits_or_nan(ops.lab_stats_device(lab, b), want)
    assert _same_bits_or_nan(ops.lab_stats_device(lab, b, latency_form=True), want)


@pytest.mark.parametrize("F,b,scale", [(3, 1, 1e-33), (6, 2, 1e-36), (66, 1, 1e-33), (40, 1, 3e37), (4, 1, 1.0), (1, 1, 1e-33), (2, 2, 3e37), (2, 1, 1e-36)])
def test_device_statistics_markstein_fallback(pkg, ops, dev, F, b, scale):
    """The whole-frame statistics kernels divide by the running count with Markstein's sequence (reciprocal off the dependent chain),
    proven equal to the IEEE quotient for 2^-100 <= |delta| <= 2^100; a workgroup that meets another delta repeats its frame with the
    IEEE division.  Frames of tiny values (every delta below 2^-100), of huge ones (deltas beyond 2^100, m2 overflowing to Inf -> NaN
    like torch's) and a frame with a NaN pixel force that path: still torch's bits."""
    lab = ((_rand((F, 8, 3840, 3), 800 + F) - 0.3) * scale).to(dev)
    if scale == 1.0:
        lab[1, 3, 17, 2] = float("nan")
        lab[2, 0, 0, 0] = float("inf")
    if F <= 2 and scale < 1.0:
        lab[0, 2, 100, 1] = float("nan")               # the one-accumulator-per-lane form (<= 2 frames) meets a NaN as well
    want = _torch_reductions(lab, b)
    assert _same_bits_or_nan(ops.lab_stats_device(lab, b), want)
    assert _same_bits_or_nan(ops.lab_stats_device(lab, b, latency_form=True), want)
    import ctypes as C
    from comfyui_vrgamedevgirl_amd import _hip
    got = torch.empty((F, 3, 2), device=dev)          # and the forms that take no scratch buffer
    _hip.check(_hip.lib().vrg_lab_stats_torch_f32(_hip.ptr(lab), F, 8, 3840, b, _hip.ptr(got), ops._f32(1e-5), _hip.current_stream()), "stats")
    assert _same_bits_or_nan(got, want)


def test_first_use_of_the_per_process_caches_from_several_host_threads(pkg, ops, dev):
    """The small per-process caches (side streams, the device-statistics self-check, the HIP event pool, the LUT cache) are created on first
    use; several host threads hitting that first use at the same moment must neither fail nor change a result."""
    import threading
    from comfyui_vrgamedevgirl_amd import nodes, VRGDG_IV_Adjustments as iv
    x = _rand((5, 48, 64, 3), 931)
    ref = _rand((1, 20, 30, 3), 932)

    def work():
        a = nodes.ColorMatchToReference().match_color(x, ref, 0.9, 2)[0]
        b = iv.VRGDG_LUTS().apply_lut(x, "AMD_TealOrange_33.cube", "auto", 10.0)[0]
        e0, e1 = ops.HipEvent(), ops.HipEvent()
        e0.record(); e1.record()
        assert e0.elapsed_ms(e1) >= 0.0
        return a, b

    want = work()
    with ops._STATE_LOCK:
        ops._SIDE_STREAMS.clear(); ops._TS_CHECKED.clear(); ops.HipEvent._pool.clear()
    iv.VRGDG_LUTS._LUT_CACHE = {}
    got, errors = [None] * 6, []
    gate = threading.Barrier(6)

    def run(i):
        try:
            gate.wait()
            got[i] = work()
        except Exception as exc:
            errors.append(exc)
    threads = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for g in got:
        assert torch.equal(g[0], want[0]) and torch.equal(g[1], want[1])


def test_welford_division_by_the_count_is_the_ieee_quotient(pkg, dev):
    """delta / n as the whole-frame statistics kernels evaluate it (rn = 1.0f / n off the chain, q = delta * rn, e = fma(-n, q, delta),
    q + e * rn) equals the IEEE quotient for every fp32 significand of delta -- i.e. for every delta of the guarded range -- for the
    counts 1 .. 20,000 (a 4K frame reaches 8,101, an 8K frame 32,401; tools/welford_division_sweep.py sweeps all counts up to 2^20) and a
    sample of larger ones."""
    from comfyui_vrgamedevgirl_amd import _hip
    mis = torch.zeros(1, dtype=torch.int64, device=dev)
    for n0, cnt in ((1, 20000), (32000, 800), (65500, 100), (1048000, 576)):
        _hip.check(_hip.lib().vrg_selftest_welford_division(_hip.ptr(mis), n0, cnt, _hip.current_stream()), "vrg_selftest_welford_division")
    assert int(mis.item()) == 0
