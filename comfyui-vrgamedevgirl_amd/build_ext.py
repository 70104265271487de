"""Build libvrgdg_hip.so (hipcc, gfx950 only) in-tree.

    python comfyui-vrgamedevgirl_amd/build_ext.py [--force]

hipcc cross-compiles without a GPU.  The library has no torch dependency: it is a plain C-ABI shared
object (include/vrgdg_hip.h) that links against the HIP runtime by soname, so inside a torch process it
binds to the runtime torch already loaded.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libvrgdg_hip.so")
STAMP = LIB_PATH + ".stamp"

SOURCES = ("vrg_pointwise.hip", "vrg_stencil.hip", "vrg_chain.hip", "vrg_march.hip", "vrg_produce.hip", "vrg_adjust.hip",
           "vrg_api.hip")
HEADERS = ("vrg_common.hpp", "vrg_pixel_math.hpp", "vrg_chain_stages.hpp", "vrg_adjust_math.hpp", "vrg_pow_tables.inc")

# -ffp-contract=off : the reference performs one rounding per op; FMAs are written explicitly where
#                     torch's own device code has them (Box-Muller).
# IEEE divide/sqrt  : hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt) is kept.
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-result")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libvrgdg_hip.so for gfx950)")


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(INCLUDE, "vrgdg_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    want = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == want:
                return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, "-I", INCLUDE, "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[vrgdg-amd] building:", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(want)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
