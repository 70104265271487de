"""Build libvrgdg_hip.so -- the product: the C ABI of include/vrgdg_hip.h -- and libvrgdg_hip_debug.so -- the self-tests and probes of
include/vrgdg_hip_debug.h, for the test suite and the measurement tools only -- in-tree (hipcc, gfx950 only).

    python comfyui-vrgamedevgirl_amd/build_ext.py [--force]

hipcc cross-compiles without a GPU.  The libraries have no torch dependency: plain C-ABI shared objects that link against the HIP runtime
by soname, so inside a torch process they bind to the runtime torch already loaded.  Nothing the nodes call lives in the debug library
(round 6: csrc/vrg_probe.hip left the product).
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
DEBUG_LIB_PATH = os.path.join(PKG_DIR, "libvrgdg_hip_debug.so")
STAMP = LIB_PATH + ".stamp"

SOURCES = ("vrg_pointwise.hip", "vrg_stencil.hip", "vrg_chain.hip", "vrg_march.hip", "vrg_produce.hip", "vrg_apply_march.hip", "vrg_adjust.hip",
           "vrg_collective.hip", "vrg_lut_tetra.hip", "vrg_torch_stats.hip", "vrg_api.hip", "vrg_host.hip")
DEBUG_SOURCES = ("vrg_probe.hip",)
HEADERS = ("vrg_common.hpp", "vrg_pixel_math.hpp", "vrg_chain_stages.hpp", "vrg_adjust_math.hpp", "vrg_pow_tables.inc",
           "vrg_ziv_log_table.inc", "vrg_produce_body.hpp", "vrg_apply_body.hpp", "vrg_tstats_body.hpp", "vrg_lanes.hpp", "vrg_tstats_config.hpp")

# -ffp-contract=off : the reference performs one rounding per op; FMAs are written explicitly where
#                     torch's own device code has them (Box-Muller).
# IEEE divide/sqrt  : hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt) is kept.
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-result")


# Per-source extra flags.  -fno-slp-vectorize: plain -O3 packs adjacent fp32 adds / multiplies of the long double-word
# chains of the colour-match arithmetic into v_pk_*_f32 plus the v_mov_b32 that line their operands up; a packed op issues
# at 1.8x a plain one on gfx950 (profiles/r02_valu_issue_rate_long.json), so that is a loss: the device-exact colour-match
# passes run 11-14 % faster without it, grain -> LUT 3 % (A/B: profiles/r02_ab_slp_vectorize.log).  The wave-march kernel
# is the opposite (grain -> sharpen 26 % slower without SLP) and keeps the default.
# per translation unit, each an A/B on one box: -fno-slp-vectorize for the colour-match units (profiles/r02_ab_slp_vectorize.log); the
# "max-ilp" machine scheduler for the grain -> LUT -> sharpen march and the stand-alone stencils (chain 3 -3.4 %, sobel -6 %,
# profiles/r03_sched_strategy_ab.log; the colour-match passes and the grain kernels are 1-5 % slower with it and keep the default)
_MAX_ILP = ("-mllvm", "-amdgpu-sched-strategy=max-ilp")
EXTRA_FLAGS = {"vrg_chain.hip": ("-fno-slp-vectorize",), "vrg_produce.hip": ("-fno-slp-vectorize",), "vrg_apply_march.hip": ("-fno-slp-vectorize",),
               "vrg_march.hip": _MAX_ILP, "vrg_stencil.hip": _MAX_ILP,
               "vrg_pointwise.hip": ("-fno-slp-vectorize",)}       # grain / fused sharpen -> grain -1..-3 % (profiles/r03_sched_strategy_ab.log)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libvrgdg_hip.so for gfx950)")


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + DEBUG_SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    for name in ("vrgdg_hip.h", "vrgdg_hip_debug.h"):
        with open(os.path.join(INCLUDE, name), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _object_digest(name: str) -> str:
    h = hashlib.sha256()
    for dep in (name,) + HEADERS:
        with open(os.path.join(CSRC, dep), "rb") as fh:
            h.update(fh.read())
    for hname in ("vrgdg_hip.h", "vrgdg_hip_debug.h"):
        with open(os.path.join(INCLUDE, hname), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS + EXTRA_FLAGS.get(name, ())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every translation unit to an object (in parallel, cached by content digest under csrc/.obj) and link the two libraries."""
    want = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(DEBUG_LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == want:
                return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(CSRC, ".obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def compile_one(name: str) -> str:
        obj = os.path.join(obj_dir, name + ".o")
        stamp = obj + ".stamp"
        dig = _object_digest(name)
        if not force and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as fh:
                if fh.read().strip() == dig:
                    return obj
        cmd = [hipcc, *cflags, *EXTRA_FLAGS.get(name, ()), "-I", INCLUDE, "-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print("[vrgdg-amd] compiling:", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as fh:
            fh.write(dig)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES) + len(DEBUG_SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES + DEBUG_SOURCES))
    for lib, group in ((LIB_PATH, objs[:len(SOURCES)]), (DEBUG_LIB_PATH, objs[len(SOURCES):])):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + group + ["-ldl", "-pthread"]      # vrg_collective.hip: dlopen / dlsym (RCCL at run time)
        if verbose:
            print("[vrgdg-amd] linking:", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(want)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
