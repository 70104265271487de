"""Kernel arithmetic checked without a GPU: csrc/vrg_pixel_math.hpp compiled for the host by g++ (test
scaffolding, see tests/host_math/host_math_check.cpp) against the oracle and the golden fixtures.

What this pins before any GPU minute is spent: operation order and rounding points of grain / LUT /
stencils (bit-exact), the Philox4x32-10 integer pipeline and torch's element->(subsequence, call, component)
mapping (bit-exact against oracle/philox.py), the Lab transforms (to libm-vs-torch pow tolerance)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from oracle import philox as PH
from oracle import restated as R
from conftest import GOLDEN, PKG_DIR, ROOT

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")

F32P = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    out = tmp_path_factory.mktemp("host_math") / "libhost_math.so"
    src = os.path.join(ROOT, "tests", "host_math", "host_math_check.cpp")
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-msse2", "-mfpmath=sse", "-mfma", "-fPIC", "-shared",
           "-I", os.path.join(PKG_DIR, "csrc"), src, "-o", str(out)]
    subprocess.run(cmd, check=True)
    lib = C.CDLL(str(out))
    lib.hm_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    lib.hm_randn.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64, F32P]
    lib.hm_grain.argtypes = [F32P, F32P, F32P, C.c_int64, C.c_float, C.c_float, C.c_float]
    lib.hm_lut.argtypes = [F32P, F32P, C.c_int64, F32P, C.c_int, F32P, F32P, C.c_int, C.c_float, C.c_float]
    lib.hm_stencil.argtypes = [F32P, F32P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    lib.hm_rgb_to_lab.argtypes = [F32P, F32P, C.c_int64]
    lib.hm_lab_to_rgb.argtypes = [F32P, F32P, C.c_int64]
    lib.hm_colormatch.argtypes = [F32P, F32P, C.c_int64, F32P, F32P, C.c_float, C.c_float]
    lib.hm_pow.argtypes = [F32P, F32P, C.c_int64, C.c_double]
    lib.hm_dev_pow.argtypes = [F32P, F32P, C.c_int64, C.c_float]
    lib.hm_rgb_to_lab_dev.argtypes = [F32P, F32P, C.c_int64]
    lib.hm_lab_to_rgb_dev.argtypes = [F32P, F32P, C.c_int64]
    lib.hm_exp_cores.argtypes = [F32P, F32P, F32P, C.c_int64]
    lib.hm_ziv.argtypes = [F32P, F32P, F32P, F32P, C.c_int64, C.c_float, C.c_uint32, C.c_uint32]
    lib.hm_div_sigma.argtypes = [F32P, F32P, F32P, F32P, C.c_int64]
    lib.hm_divc_mismatches.argtypes = [F32P, C.c_int64, C.c_int]
    lib.hm_divc_mismatches.restype = C.c_int64
    return lib


def f32(v):
    return float(np.float32(v))


def test_philox_known_answers(hm):
    # Random123 kat_vectors for philox4x32_10
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = PH.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want
        out = (C.c_uint32 * 4)()
        hm.hm_philox(key[0] | (key[1] << 32), ctr[2] | (ctr[3] << 32), ctr[0] | (ctr[1] << 32), out)
        assert tuple(out) == want


@pytest.mark.parametrize("numel,G,offset", [(1000, 256, 0), (5000, 512, 8), (3 * 7 * 5, 256, 4), (9000, 1024, 4096)])
def test_torch_element_mapping(hm, numel, G, offset):
    seed = 0x1234567887654321
    out = np.empty(numel, dtype=np.float32)
    hm.hm_randn(seed, offset, G, numel, out)
    want = PH.torch_stream_normals_f64(numel, seed, offset, G)
    # libm stand-ins for v_log/v_sin/v_cos: agreement to a few 1e-6 proves seed/offset/idx/call/component wiring
    assert np.max(np.abs(out.astype(np.float64) - want)) < 2e-5
    assert abs(out.mean()) < 0.2 and 0.8 < out.std() < 1.2


def test_grain_bit_exact(hm):
    z = np.load(os.path.join(GOLDEN, "grain.npz"))
    x = np.ascontiguousarray(z["x"])
    for tag in ("default", "strong_colour", "mono_all", "workflow_widgets"):
        I, s = float(z[f"{tag}.I"]), float(z[f"{tag}.s"])
        n = np.ascontiguousarray(z[f"{tag}.noise"])
        o = np.empty_like(x)
        hm.hm_grain(x, n, o, x.size // 3, f32(I), f32(s), f32(1.0 - s))
        assert np.array_equal(o, z[f"{tag}.out"]), tag


@pytest.mark.parametrize("tag", ["a", "b"])
def test_lut_bit_exact(hm, tag):
    z = np.load(os.path.join(GOLDEN, "lut.npz"))
    img = np.ascontiguousarray(z[f"{tag}.img"])
    table = np.ascontiguousarray(z[f"{tag}.lut"])
    dmin, dmax = np.ascontiguousarray(z[f"{tag}.dmin"]), np.ascontiguousarray(z[f"{tag}.dmax"])
    for s, key in ((10.0, "out.s10.0"), (25.0, "out.s25.0"), (3.3, "out.s3.3"), (6.5, "route.s6.5")):
        blend = max(0.0, min(10.0, s)) / 10.0
        mode = 1 if blend >= 1.0 else 2
        o = np.empty_like(img)
        hm.hm_lut(img, o, img.size // 3, table, table.shape[0], dmin, dmax, mode, f32(blend), f32(1.0 - blend))
        assert np.array_equal(o, z[f"{tag}.{key}"]), (tag, s)


def test_lut_random_vs_oracle(hm):
    g = torch.Generator().manual_seed(3)
    lut = R.parse_cube_file(os.path.join(GOLDEN, "synthetic_17.cube"))
    img = (torch.rand(1, 64, 64, 3, generator=g) * 1.3 - 0.15).contiguous()
    want = R.apply_cube_lut(img, lut["lut"], lut["domain_min"], lut["domain_max"]).numpy()
    o = np.empty_like(want)
    hm.hm_lut(img.numpy(), o, img.numel() // 3, lut["lut"].numpy(), 17, lut["domain_min"].numpy(), lut["domain_max"].numpy(),
              1, 1.0, 0.0)
    assert np.array_equal(o, want)


def test_stencils_bit_exact(hm):
    z = np.load(os.path.join(GOLDEN, "stencil.npz"))
    ops = {"unsharp": 0, "laplacian": 1, "sobel": 2}
    for tag in ("rand", "odd", "one", "row", "col", "c4", "const"):
        x = np.ascontiguousarray(z[f"{tag}.x"])
        F, H, W, Cn = x.shape
        for s in (0.5, 3.75):
            for zero in (0, 1):
                o = np.empty_like(x)
                hm.hm_stencil(x, o, F, H, W, Cn, 0, zero, f32(s))
                assert np.array_equal(o, z[f"{tag}.unsharp.{s}.{zero}"]), (tag, s, zero)
        for name in ("laplacian", "sobel"):
            o = np.empty_like(x)
            hm.hm_stencil(x, o, F, H, W, Cn, ops[name], 0, f32(0.8))
            assert np.array_equal(o, z[f"{tag}.{name}.0.8.0"]), (tag, name)
            # zero border: kernel order = explicit raster restatement (bit-exact) ~ conv2d (1 ulp)
            xt = torch.from_numpy(x)
            hm.hm_stencil(x, o, F, H, W, Cn, ops[name], 1, f32(0.8))
            want = (R.laplacian_zero_raster if name == "laplacian" else R.sobel_zero_raster)(xt, 0.8).numpy()
            assert np.array_equal(o, want), (tag, name, "zero")
            if Cn == 3:                # against the reference's CPU conv2d: laplacian bit-equal, sobel within one ulp(1.0)
                d = float(np.abs(o.astype(np.float64) - z[f"{tag}.{name}.0.8.1"].astype(np.float64)).max() / 2.0 ** -23)
                assert d <= (0.0 if name == "laplacian" else 1.0), (tag, name, d)


def test_lab_and_colormatch_close_to_oracle(hm):
    g = torch.Generator().manual_seed(11)
    rgb = torch.rand(1, 40, 40, 3, generator=g).contiguous()
    lab = np.empty((1600, 3), dtype=np.float32)
    hm.hm_rgb_to_lab(rgb.numpy(), lab, 1600)
    want = R.kornia_rgb_to_lab(rgb.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(-1, 3).numpy()
    assert np.max(np.abs(lab - want)) < 1.5e-4        # libm vs Sleef powf differ by an ulp; a,b = 500*(fx-fy), 200*(fy-fz) amplify it
    back = np.empty_like(lab)
    hm.hm_lab_to_rgb(np.ascontiguousarray(want), back, 1600)
    want_rgb = R.kornia_lab_to_rgb(torch.from_numpy(want).reshape(1, 40, 40, 3).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(-1, 3).numpy()
    assert np.max(np.abs(back - want_rgb)) < 2e-6
    # whole colour-match pixel function given the oracle's statistics
    ref = torch.rand(1, 8, 8, 3, generator=g)
    x_nchw = rgb.permute(0, 3, 1, 2)
    mu, sd = R.lab_stats(R.kornia_rgb_to_lab(x_nchw))
    rmu, rsd = R.lab_stats(R.kornia_rgb_to_lab(ref.permute(0, 3, 1, 2)))
    ims = torch.stack([mu.flatten(), sd.flatten()], dim=-1).contiguous().numpy()
    rms = torch.stack([rmu.flatten(), rsd.flatten()], dim=-1).contiguous().numpy()
    out = np.empty((1600, 3), dtype=np.float32)
    hm.hm_colormatch(rgb.numpy(), out, 1600, ims, rms, f32(0.8), f32(1.0 - 0.8))
    want = R.color_match(rgb, ref, 0.8, 1).reshape(-1, 3).numpy()
    assert np.max(np.abs(out - want)) < 5e-6


def test_pow_pos_is_faithful(hm):
    """pow_pos (fp32 double-word log2/exp2, one final rounding) against float64 pow, on the three
    (exponent, domain) pairs the Lab transforms use: never more than 1 ulp from the correctly rounded fp32 power,
    within 0.56 ulp of the true value."""
    rng = np.random.default_rng(0)
    for y32, lo, hi in ((np.float32(2.4), 0.0625, 4.0), (np.float32(1.0 / 3.0), 0.008856, 4.0), (np.float32(1.0 / 2.4), 0.0031308, 64.0)):
        x = np.exp(rng.uniform(np.log(lo), np.log(hi), size=2_000_000)).astype(np.float32)
        x = np.concatenate([x, np.array([lo, hi, 1.0, 0.5, 2.0, np.nextafter(np.float32(1), np.float32(0))], dtype=np.float32)])
        out = np.empty_like(x)
        hm.hm_pow(x, out, x.size, float(y32))
        want64 = np.power(x.astype(np.float64), np.float64(y32))
        want = want64.astype(np.float32)
        ulp = np.abs(out.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
        assert (ulp != 0).mean() < 0.06          # only values near a rounding boundary may round the other way
        ulp_size = np.abs(np.spacing(want).astype(np.float64))
        assert np.max(np.abs(out.astype(np.float64) - want64) / ulp_size) < 0.56
    bad = np.array([np.nan], dtype=np.float32)
    o = np.empty_like(bad)
    hm.hm_pow(bad, o, 1, 2.4)
    assert np.isnan(o[0])


def test_constant_division_by_fma_is_ieee_division(hm):
    """Sampled here (every 1009th fp32 bit pattern, both signs, 4.2M inputs per constant); the exhaustive
    2^32 sweep was run once with the same code (0 mismatches in [1e-30, 1e30]; c = 9: none for finite x)."""
    bits = np.arange(0, 2 ** 32, 1009, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    mid = x[(np.abs(x) >= 1e-30) & (np.abs(x) <= 1e30)]
    mid = np.ascontiguousarray(mid)
    for which in range(8):
        assert hm.hm_divc_mismatches(mid, mid.size, which) == 0, which
    allx = np.ascontiguousarray(x)
    assert hm.hm_divc_mismatches(allx, allx.size, 8) == 0          # /9 (unsharp): every input incl. denormals, Inf
    for which in (9, 10, 11):                                       # Adjust: guarded form, every input
        assert hm.hm_divc_mismatches(allx, allx.size, which) == 0, which


def _adjust_terms_array(pkg_ops, lvt, settings):
    d = pkg_ops.adjust_terms(lvt._normalize_adjust_settings(settings))
    return np.array([d.enabled, d.shift[0], d.shift[1], d.shift[2], d.exposure, d.contrast, d.saturation, d.highlights,
                     d.shadows, d.whites, d.blacks, d.has_clarity, d.clarity, d.has_sharpen, d.sharpen, d.has_fade,
                     d.fade_mul, d.fade_add, d.has_vignette, d.vignette], dtype=np.float32)


def test_adjust_bit_exact(hm, pkg):
    """Adjust arithmetic (csrc/vrg_adjust_math.hpp) on the host == the reference's fixtures, bit for bit:
    rounding order of the point stage, raster-order box sums, torch.linspace's two-sided evaluation."""
    import json
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT, ops
    hm.hm_adjust.argtypes = [F32P, F32P, C.c_int, C.c_int, C.c_int, F32P]
    z = np.load(os.path.join(GOLDEN, "adjust.npz"))
    with open(os.path.join(GOLDEN, "adjust_cases.json")) as fh:
        meta = json.load(fh)
    for tag in meta["shapes"]:
        x = np.ascontiguousarray(z[f"{tag}.x"])
        F, H, W, _ = x.shape
        for name, settings in meta["cases"].items():
            o = np.empty_like(x)
            hm.hm_adjust(x, o, F, H, W, _adjust_terms_array(ops, LVT, settings))
            want = z[f"{tag}.{name}"]
            if LVT._normalize_adjust_settings(settings)["vignette"] > 0.0 and settings.get("enabled", True) is not False:
                # torch's CPU sqrt is not correctly rounded (oracle/restated.py adjust_tensor): 1 ulp on a few pixels
                assert np.max(np.abs(o - want)) <= 1.2e-7, (tag, name)
                assert np.mean(o != want) < 0.03, (tag, name)
                want = R.adjust_tensor(torch.from_numpy(x), settings, ieee_sqrt=True).contiguous().numpy()
            assert np.array_equal(o, want), (tag, name, float(np.max(np.abs(o - want))))


@pytest.mark.parametrize("shape", [(1, 33, 70, 3), (2, 64, 9, 3), (1, 7, 130, 3)])
def test_adjust_vs_oracle_larger_frames(hm, pkg, shape):
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT, ops
    hm.hm_adjust.argtypes = [F32P, F32P, C.c_int, C.c_int, C.c_int, F32P]
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(*shape, generator=g) * 1.2 - 0.1
    for settings in ({"clarity": 55, "sharpen": 25, "vignette": 80, "fade": 30, "exposure": 20},
                     {"clarity": -40, "temperature": 60, "blacks": 50, "vignette": 100}):
        want = R.adjust_tensor(x, settings, ieee_sqrt=True).contiguous().numpy()
        o = np.empty_like(want)
        hm.hm_adjust(np.ascontiguousarray(x.numpy()), o, shape[0], shape[1], shape[2], _adjust_terms_array(ops, LVT, settings))
        assert np.array_equal(o, want), float(np.max(np.abs(o - want)))


def test_u8_codec_edge_conversions(hm):
    """unit_from_u8 == astype(float32) / 255.0 for all 256 codes; u8_from_unit == clip(x * 255, 0, 255).astype(uint8)
    on the fixture's grid values / neighbours and on 4M random floats; the round trip is the identity."""
    U8P = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    hm.hm_u8_to_unit.argtypes = [U8P, F32P, C.c_int64]
    hm.hm_unit_to_u8.argtypes = [F32P, U8P, C.c_int64]
    codes = np.arange(256, dtype=np.uint8)
    unit = np.empty(256, dtype=np.float32)
    hm.hm_u8_to_unit(codes, unit, 256)
    assert np.array_equal(unit, codes.astype(np.float32) / 255.0)
    back = np.empty(256, dtype=np.uint8)
    hm.hm_unit_to_u8(unit, back, 256)
    assert np.array_equal(back, codes)
    z = np.load(os.path.join(GOLDEN, "io_u8.npz"))
    rng = np.random.default_rng(5)
    for x in (np.ascontiguousarray(z["edge.tensor"]).ravel(), np.ascontiguousarray(z["tens.tensor"]).ravel(),
              (rng.random(1 << 22, dtype=np.float32) * 1.5 - 0.25), np.array([-0.0, 0.0, 1.0, 2.0, -3.0, 1e-45, 0.99999994], dtype=np.float32)):
        got = np.empty(x.size, dtype=np.uint8)
        hm.hm_unit_to_u8(x, got, x.size)
        assert np.array_equal(got, np.clip(x * np.float32(255.0), 0, 255).astype(np.uint8))


def test_cbrt_pow_is_correctly_rounded_on_its_domain(hm):
    """cbrt_pow(x) = x ** float32(1/3) on the Lab domain: with libm stand-ins for the hardware log2 / exp2 / rcp estimates
    (the Newton step squares their error away) the result equals the correctly rounded power; max error 0.5 ulp."""
    hm.hm_cbrt_pow.argtypes = [F32P, F32P, C.c_int64]
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0.008856, 1.2, 1 << 20), np.exp(rng.uniform(np.log(0.008856), np.log(1.2), 1 << 20)),
                        [0.008856, 1.0, 0.5, 0.125]]).astype(np.float32)
    got = np.empty_like(x)
    hm.hm_cbrt_pow(x, got, x.size)
    truth = np.power(x.astype(np.float64), np.float64(np.float32(1.0 / 3.0)))
    ulp = np.spacing(np.abs(truth).astype(np.float32)).astype(np.float64)
    err = np.abs(got.astype(np.float64) - truth) / ulp
    assert err.max() <= 0.5001, err.max()
    assert np.mean(got != truth.astype(np.float32)) < 1e-5


def test_device_policy_transcription_structure_on_the_host(hm):
    """dev_pow (ocml powf without its scaffolding) and the device-policy Lab transforms compiled for the host: with IEEE 1/x for
    v_rcp_f32 and libm expf for the backend's exp they are not bit-equal to the device, but any transcription slip (a constant,
    a dropped low-order term, a swapped operand) costs far more than the <= 2 ulp the double-word algorithm delivers."""
    rng = np.random.default_rng(3)
    for y, lo, hi in ((2.4, 0.0625, 4.0), (1 / 2.4, 0.0031308, 4.0), (1 / 3.0, 0.008856, 4.0), (2.4, 1e-30, 1e30)):
        x = np.exp(rng.uniform(np.log(lo), np.log(hi), 200000)).astype(np.float32)
        out = np.empty_like(x)
        hm.hm_dev_pow(x, out, x.size, f32(y))
        want = np.power(x.astype(np.float64), np.float64(np.float32(y)))
        ok = np.isfinite(want) & (want > 1e-37) & (want < 3e38)
        ulps = np.abs(out[ok].astype(np.float64) - want[ok]) / (np.abs(want[ok]) * 2.0 ** -23)
        assert ulps.max() < 2.0, (y, ulps.max())
    assert np.isinf(np.float32(_one(hm, np.inf, 2.4))) and np.isnan(_one(hm, np.nan, 2.4)) and _one(hm, 1.0, 2.4) == 1.0
    from oracle import restated as R
    import torch
    g = torch.Generator().manual_seed(5)
    rgb = torch.rand(4000, 3, generator=g)
    lab = np.empty((4000, 3), np.float32)
    hm.hm_rgb_to_lab_dev(np.ascontiguousarray(rgb.numpy()), lab, 4000)
    want = R.kornia_rgb_to_lab(rgb.t().reshape(1, 3, 1, 4000))[0, :, 0, :].t().numpy()
    assert np.abs(lab - want).max() < 3e-4                      # Lab units (L up to 100): a few ulp of the powers through the 500 / 200 gains
    back = np.empty((4000, 3), np.float32)
    hm.hm_lab_to_rgb_dev(np.ascontiguousarray(lab), back, 4000)
    assert np.abs(back - rgb.numpy()).max() < 2e-5              # round trip


def _one(hm, x, y):
    a = np.array([x], np.float32)
    o = np.empty(1, np.float32)
    hm.hm_dev_pow(a, o, 1, f32(y))
    return o[0]


def test_exp_core_with_integer_scaling_equals_the_plain_form(hm):
    """dev_exp_core_normal (round 6): rint(ph) as (ph + 1.5 * 2^23) - 1.5 * 2^23 and ldexp as an integer add to the exponent field.  On the
    host both forms call the same exp2f, so every difference would be the new form's own: bit-equal for every argument the Ziv route can
    hand it (|y ln x| <= 12 in its domains; checked to +-60, ties of the rounding included)."""
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-60.0, 60.0, 2_000_000), (np.arange(-80, 81) + 0.5) * np.log(2.0), np.arange(-80, 81) * np.log(2.0),
                        [0.0, -0.0, 1e-30, -1e-30, 12.0, -12.0]]).astype(np.float32)
    a, b = np.empty_like(x), np.empty_like(x)
    hm.hm_exp_cores(x, a, b, x.size)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("y,lo,hi", [(2.4, 0.0625, 2.0), (1 / 2.4, 0.0031308, 4.0), (1 / 3.0, 0.008856, 4.0)])
def test_ziv_route_on_the_host(hm, y, lo, hi):
    """The table logarithm of dev_pow_ziv (csrc/vrg_ziv_log_table.inc, T_hi on the 2^-21 grid), its product / exp / final FMA and the rounding
    test, against the transcription of ocml powf, both compiled for the host.  The half-widths are calibrated against the DEVICE's logarithm
    (its reciprocal is v_rcp_f32, here IEEE), so a passing lane may differ from the host transcription in rare arguments -- but a wrong table
    word, a dropped term or a broken test shows as thousands: <= 5 per million passing lanes differ (measured: 0 of 4 million per exponent),
    >= 99.9 % pass (measured 99.975-99.99 %), and no candidate is further than one ulp from the transcription."""
    rng = np.random.default_rng(12)
    lo_b, hi_b = int(np.float32(lo).view(np.uint32)), int(np.float32(hi).view(np.uint32))
    x = np.concatenate([np.exp(rng.uniform(np.log(lo), np.log(hi), 1_000_000)), rng.uniform(lo, min(hi, 1.2), 1_000_000), [lo, hi, 1.0, 0.99999994, 1.0000001]]).astype(np.float32)
    x = np.clip(x, np.float32(lo), np.float32(hi))
    cand, ok, ref = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    hm.hm_ziv(x, cand, ok, ref, x.size, f32(y), lo_b, hi_b)
    passed = ok == 1
    assert passed.mean() >= 0.999, passed.mean()
    differ = cand[passed].view(np.uint32) != ref[passed].view(np.uint32)
    assert differ.sum() <= 5e-6 * passed.sum(), (int(differ.sum()), int(passed.sum()))
    ulps = np.abs(cand.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    assert ulps.max() <= 1, ulps.max()
    # outside the domain the test must fail
    out = np.array([lo * 0.5, hi * 2.0, 1e-20, 1e20], np.float32)
    c2, o2, r2 = np.empty_like(out), np.empty_like(out), np.empty_like(out)
    hm.hm_ziv(out, c2, o2, r2, out.size, f32(y), lo_b, hi_b)
    assert not o2.any()


def test_unscaled_sigma_division_structure_on_the_host(hm):
    """div_sigma_unscaled (round 6) is the backend's division sequence minus its scalings and fix-up.  On the host the reciprocal is IEEE
    instead of v_rcp_f32, which only makes the Newton / Markstein steps start closer: the five FMAs must return the correctly rounded
    quotient for every ordinary operand pair (a wrong sign, a swapped operand or a missing step would not); bit-equality with the
    DEVICE's division is the GPU suite's test."""
    rng = np.random.default_rng(21)
    n = 2_000_000
    d = ((rng.random(n) - 0.4) * 250.0).astype(np.float32)
    sd = np.exp(rng.uniform(np.log(1e-5), np.log(90.0), n)).astype(np.float32)
    d[:5] = [0.0, 1.0, -1.0, 2.0 ** -80, 2.0 ** 39]
    a, b = np.empty_like(d), np.empty_like(d)
    hm.hm_div_sigma(d, sd, a, b, n)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), int((a.view(np.uint32) != b.view(np.uint32)).sum())
