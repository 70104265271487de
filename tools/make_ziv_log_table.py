"""Generate csrc/vrg_ziv_log_table.inc: the 128-entry table of dev_pow_ziv()'s double-word natural log (vrg_pixel_math.hpp).

    python tools/make_ziv_log_table.py

The argument is split like ocml's epln does, x = m * 2^e with m in [2/3, 4/3) -- from the bit pattern: d = bits(x) - 0x3f2aaaab,
e = d >> 23, m = bits((d & 0x7fffff) + 0x3f2aaaab) -- and the table is indexed by j = (d & 0x7fffff) >> 16: 128 intervals of
2^16 consecutive floats.  Entry j = {c_j, T_hi_j, T_lo_j, A_j / C}:
    c_j   a reciprocal of the interval with a 7-BIT significand, so that r = m * c_j - 1 is EXACT in fp32 (one FMA) for every m of
          the interval (checked here over all 2^16 of them) and |r| < 2^-6.6; c_j = 1 around m = 1 (then T = 0 and r = m - 1);
    T_j   = -ln(c_j) as an fp32 pair hi + lo, hi on the 2^-21 grid so that e ln2_head + T_hi is exact for the domains' exponents,
so that ln x = e ln2 + T_j + log1p(r).  Computed with Python's decimal at 60 digits.
"""
from __future__ import annotations

import os
import struct
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

getcontext().prec = 60
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "comfyui-vrgamedevgirl_amd", "csrc", "vrg_ziv_log_table.inc")
BASE = 0x3F2AAAAB


def hexf(x: float) -> str:
    return "0x%08xu" % struct.unpack("<I", struct.pack("<f", x))[0]


def f32(d) -> float:
    return float(np.float32(float(d)))


def exact_r(mbits: np.ndarray, c: float):
    """r = m * c - 1 as exact rationals' numerators over 2^60, and whether each is an fp32 number."""
    m = mbits.view(np.float32).astype(np.float64)
    cf = Fraction(c)
    scale = 1 << 40
    M = (m * (1 << 24)).astype(np.int64)                       # m = M / 2^24 exactly (m < 2)
    num = M * int(cf * (1 << 16)) - (1 << 40)                  # r * 2^40 exactly (c has <= 7 bits below 2^1: c * 2^16 is an integer)
    assert Fraction(int(cf * (1 << 16)), 1 << 16) == cf
    a = np.abs(num)
    nz = a != 0
    tz = np.zeros_like(a)
    t = a.copy()
    # trailing zeros
    for k in (32, 16, 8, 4, 2, 1):
        mask = nz & ((t & ((1 << k) - 1)) == 0)
        t = np.where(mask, t >> k, t)
        tz = np.where(mask, tz + k, tz)
    bl = np.zeros_like(a)
    tt = t.copy()
    for k in (32, 16, 8, 4, 2, 1):
        mask = tt >= (1 << k)
        tt = np.where(mask, tt >> k, tt)
        bl = np.where(mask, bl + k, bl)
    bl = bl + 1                                                 # bit length of the odd part
    ok = (~nz) | (bl <= 24)
    return num.astype(np.float64) / scale, bool(ok.all())


CAL = os.path.join(ROOT, "tools", "ziv_calibration.json")


def main():
    rows, worst = [], 0.0
    # Calibration against ocml's own logarithm (tools/ziv_per_index.py on the MI355X, every fp32 of [0.0031308, 4]): bias_j = the
    # midpoint of (ocml's double-word ln x - this table's) over the arguments with index j, folded into T_lo_j -- the table then tracks
    # OCML's logarithm, whose own error (2^-34.7) is a smooth function of m and almost constant over 1/128 of its range --, and A_j =
    # 1.25 x the largest distance that remains for index j: the half-width of dev_pow_ziv's rounding test (fourth word).
    import json
    cal = json.load(open(CAL)) if os.path.exists(CAL) else {"bias": [0.0] * 128, "A": [2.0 ** -35.7] * 128}
    for j in range(128):
        mbits = np.arange(BASE + (j << 16), BASE + ((j + 1) << 16), dtype=np.uint32)
        lo, hi = float(mbits[:1].view(np.float32)[0]), float(mbits[-1:].view(np.float32)[0])
        inv = 2.0 / (lo + hi)
        step = 2.0 ** (np.floor(np.log2(inv)) - 6)              # 7-bit significand grid
        best = None
        for k in (-1, 0, 1):
            c = (round(inv / step) + k) * step
            r, ok = exact_r(mbits, c)
            rmax = float(np.abs(r).max())
            if ok and (best is None or rmax < best[1]):
                best = (c, rmax)
        assert best is not None, j
        c, rmax = best
        worst = max(worst, rmax)
        T = -(Decimal(c).ln())
        # T_hi on the 2^-21 grid: e * ln2_head (a multiple of 2^-15, |.| <= 5.6 for the exponents e = -8 .. 2 of the fast-path domains
        # [0.0031308, 4]) + T_hi is then EXACT in fp32 (|sum| < 8: 3 + 21 bits), and ziv_log needs no error term for that sum.  T_lo
        # carries the rest (|T - T_hi| <= 2^-22: fp32 keeps it to 2^-46).  Outside those exponents the sum may round -- those arguments
        # are outside every domain and never take the fast path's result.
        th = float(int((T * (1 << 21)).to_integral_value(rounding="ROUND_HALF_EVEN"))) / float(1 << 21)
        assert f32(th) == th
        for e in range(-9, 4):
            exact = Fraction(e * 22713, 1 << 15) + Fraction(th)          # 0x3f317200 = 22713 * 2^-15
            got = np.float32(np.float32(e) * np.float32(0.693145751953125)) + np.float32(th)
            assert Fraction(float(got)) == exact, (j, e)
        # no bias around m = 1 (T = 0 there, and for |ln m| < 0.08 the distance between the two logarithms is RELATIVE to |ln x|: a
        # constant shift would widen the relative bound that dev_pow_ziv's test uses near x = 1)
        u0, u1 = cal.get("unbiased_indexes", [128, -1])
        bias = 0.0 if (c == 1.0 or u0 <= j <= u1) else cal["bias"][j]
        tl = f32(T - Decimal(th) + Decimal(bias))
        # fourth word: A_j / C with C = the relative bound's constant (vrg_pixel_math.hpp VRG_ZIV_REL_BITS), rounded UP: the kernel forms
        # C * min(|ln x| + 64 |e ln2|, A_j / C) with C folded into one loop-invariant factor
        Cc = float(np.array([0x2E06F428], dtype=np.uint32).view(np.float32)[0])
        a_over_c = np.float32(cal["A"][j] / Cc)
        if float(a_over_c) * Cc < cal["A"][j]:
            a_over_c = np.nextafter(a_over_c, np.float32(np.inf))
        rows.append((c, th, tl, float(a_over_c)))
        # ziv_log adds e ln2 + T_j and r - r^2/2 with the three-operation two-sum: needs a zero first operand or |first| >= |second|
        rj, _ = exact_r(mbits, c)
        s2max = float(np.abs(rj - 0.5 * rj * rj).max()) * (1.0 + 2.0 ** -20)
        for e in range(-126, 128):
            s1 = np.float32(np.float32(e) * np.float32(0.693145751953125)) + np.float32(th)      # 0x3f317200, as the kernel forms it
            assert s1 == 0.0 or abs(float(s1)) >= s2max, (j, e, float(s1), s2max)
    assert worst < 2.0 ** -6.5, worst
    lines = ["// generated by tools/make_ziv_log_table.py -- do not edit",
             "// {bits(c_j), bits(T_hi_j), bits(T_lo_j), bits(A_j / C)}: c_j a 7-bit reciprocal of interval j of m in [2/3, 4/3), T_j = -ln(c_j) + the",
             "// calibration bias towards ocml's logarithm, A_j = half-width of the rounding test for index j (tools/ziv_calibration.json), C = 0x2e06f428 (the relative bound);",
             "// max |m * c_j - 1| = %.6f (exact in fp32 for every m, checked by the generator)" % worst,
             "static constexpr unsigned VRG_ZIV_LOGT[128][4] = {"]
    for c, th, tl, aj in rows:
        lines.append("    {%s, %s, %s, %s}," % (hexf(c), hexf(th), hexf(tl), hexf(aj)))
    lines.append("};")
    with open(OUT, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("written", OUT, "max |r| = %.6f" % worst)


if __name__ == "__main__":
    main()
