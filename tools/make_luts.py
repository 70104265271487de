"""Generate the .cube files shipped in comfyui-vrgamedevgirl_amd/LUTS (procedural looks, our own data).

    python tools/make_luts.py

The reference pack ships third-party LUT assets; they are not copied here.  Users drop their own .cube files
into the LUTS folder exactly as with the reference.
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "comfyui-vrgamedevgirl_amd")


def _cube_module():
    spec = importlib.util.spec_from_file_location("_vrgdg_cube", os.path.join(PKG, "cube.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _grid(n):
    ax = np.linspace(0.0, 1.0, n, dtype=np.float64)
    b, g, r = np.meshgrid(ax, ax, ax, indexing="ij")
    return r, g, b


def identity(n):
    r, g, b = _grid(n)
    return np.stack([r, g, b], -1)


def teal_orange(n):
    r, g, b = _grid(n)
    luma = 0.2126 * r + 0.7152 * g + 0.0722 * b
    sc = lambda v: 0.5 + 0.5 * np.tanh(2.4 * (v - 0.5)) / np.tanh(1.2)       # gentle S-curve
    warm = np.clip((luma - 0.35) / 0.65, 0, 1)
    R = sc(r) + 0.06 * warm - 0.03 * (1 - warm)
    G = sc(g) + 0.01 * warm + 0.02 * (1 - warm)
    B = sc(b) - 0.07 * warm + 0.06 * (1 - warm)
    return np.clip(np.stack([R, G, B], -1), 0, 1)


def warm_film(n):
    r, g, b = _grid(n)
    R = 0.04 + 0.93 * r ** 0.92 + 0.02 * g
    G = 0.03 + 0.92 * g ** 0.98 + 0.015 * r
    B = 0.05 + 0.84 * b ** 1.06 + 0.02 * g * (1 - b)
    return np.clip(np.stack([R, G, B], -1), 0, 1)


def main():
    cube = _cube_module()
    out = os.path.join(PKG, "LUTS")
    os.makedirs(out, exist_ok=True)
    for name, fn, n in (("AMD_Identity_17.cube", identity, 17), ("AMD_TealOrange_33.cube", teal_orange, 33),
                        ("AMD_WarmFilm_25.cube", warm_film, 25)):
        path = os.path.join(out, name)
        cube.write_cube_file(torch.from_numpy(fn(n).astype(np.float32)), path)
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    sys.exit(main())
