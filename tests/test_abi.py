"""The C-ABI library loads and exports every symbol include/vrgdg_hip.h declares; argument validation
works without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from conftest import PKG_DIR, ROOT


@pytest.fixture(scope="module")
def hip(pkg):
    from comfyui_vrgamedevgirl_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        from comfyui_vrgamedevgirl_amd import build_ext
        build_ext.build(verbose=False)
    return _hip


def _declared(name):
    header = open(os.path.join(ROOT, "include", name)).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return set(re.findall(r"\b(vrg_[a-z0-9_]+)\s*\(", header))


def _exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("vrg_")}


def test_header_symbols_are_exported(hip):
    """Two headers, two libraries: the drop-in boundary (vrgdg_hip.h <-> libvrgdg_hip.so) and the self-tests / probes of the test suite and
    the tools (vrgdg_hip_debug.h <-> libvrgdg_hip_debug.so).  Every prototype is exported by ITS library and bound; the product library
    exports nothing but the boundary -- no probe or self-test is linked into what the nodes load."""
    declared, debug = _declared("vrgdg_hip.h"), _declared("vrgdg_hip_debug.h")
    assert declared and debug, "no prototypes found"
    assert _exports(hip.LIB_PATH) == declared, "libvrgdg_hip.so exports differ from vrgdg_hip.h"
    assert _exports(hip.DEBUG_LIB_PATH) == debug, "libvrgdg_hip_debug.so exports differ from vrgdg_hip_debug.h"
    lib = hip.load_library()
    assert lib._debug is None                                   # loading the product does not open the debug library
    for sym in sorted(declared):
        assert hasattr(lib._cdll, sym), f"{sym} declared but not exported"
    assert lib._debug is None
    for sym in sorted(debug):
        assert hasattr(lib, sym), f"{sym} declared but not exported by the debug library"
    assert lib._debug is not None
    assert declared == set(hip.EXPORTED_SYMBOLS), "ctypes prototypes out of sync with vrgdg_hip.h"
    assert debug == set(hip.DEBUG_SYMBOLS), "ctypes prototypes out of sync with vrgdg_hip_debug.h"
    assert not [s for s in declared if s.startswith(("vrg_debug_", "vrg_selftest_"))]
    assert not (declared & debug)


def test_struct_layouts_match_the_header(hip):
    # vrg_noise_desc: 4*u64 + i64 + i32 + u32 = 48 bytes
    assert C.sizeof(hip.NoiseDesc) == 48
    assert hip.ChainDesc.noise.offset % 8 == 0 and hip.ChainDesc.lut.offset % 8 == 0
    assert hip.ChainDesc.img_ms.offset % 8 == 0


def test_versions_and_error_strings(hip):
    lib = hip.load_library()
    assert lib.vrg_abi_version() == 8 == hip.ABI_VERSION
    assert lib.vrg_error_string(0) == b"ok"
    assert b"argument" in lib.vrg_error_string(1)
    assert lib.vrg_lab_stats_scratch_bytes(3) == 3 * 128 * 6 * 8
    assert lib.vrg_lab_stats_scratch_bytes(-1) == 0
    # the torch-order statistics want a scratch buffer only for the batches their small-batch forms take (<= 32 frames): per-frame records
    # (sized for the larger of the two record layouts, so either entry point can use it); 0 = another form
    per = lib.vrg_lab_stats_torch_scratch_bytes(32) // 32
    assert per > 0 and per % 16 == 0 and lib.vrg_lab_stats_torch_scratch_bytes(32) == 32 * per
    assert lib.vrg_lab_stats_torch_scratch_bytes(3) == 3 * per and lib.vrg_lab_stats_torch_scratch_bytes(1) == per
    assert lib.vrg_lab_stats_torch_scratch_bytes(33) == 0 and lib.vrg_lab_stats_torch_scratch_bytes(0) == 0


def test_argument_validation_without_device(hip):
    lib = hip.load_library()
    null = C.c_void_p(0)
    one = C.c_void_p(16)   # never dereferenced: validation fails or sizes are zero
    nd = hip.NoiseDesc(seed0=1, chunk_frames=1, grid_threads=256)
    f3 = (C.c_float * 3)(0, 0, 0)
    g3 = (C.c_float * 3)(1, 1, 1)
    assert lib.vrg_grain_f32(null, one, 1, 4, 4, 0.1, 0.5, 0.5, C.byref(nd), null) == 1
    assert lib.vrg_grain_f32(one, one, 0, 4, 4, 0.1, 0.5, 0.5, C.byref(nd), null) == 0          # zero frames: no launch
    bad = hip.NoiseDesc(seed0=1, chunk_frames=1, grid_threads=100)
    assert lib.vrg_grain_f32(one, one, 1, 4, 4, 0.1, 0.5, 0.5, C.byref(bad), null) == 1
    nd2 = hip.NoiseDesc(seed0=1, chunk_frames=2, grid_threads=256)
    assert lib.vrg_grain_f32(one, one, 3, 4, 4, 0.1, 0.5, 0.5, C.byref(nd2), null) == 1         # ragged chunk
    # fused sharpen -> per-frame-seeded grain: in place / null / bad border are argument errors, sizes it does not take "unsupported"
    two = C.c_void_p(32)
    assert lib.vrg_sharpen_grain_f32(one, one, 1, 8, 512, 0.5, 0, 0.1, 0.5, 0.5, C.byref(nd), null) == 1       # in == out
    assert lib.vrg_sharpen_grain_f32(null, two, 1, 8, 512, 0.5, 0, 0.1, 0.5, 0.5, C.byref(nd), null) == 1
    assert lib.vrg_sharpen_grain_f32(one, two, 1, 8, 512, 0.5, 2, 0.1, 0.5, 0.5, C.byref(nd), null) == 1       # border
    assert lib.vrg_sharpen_grain_f32(one, two, 0, 8, 512, 0.5, 0, 0.1, 0.5, 0.5, C.byref(nd), null) == 0       # zero frames: no launch
    assert lib.vrg_sharpen_grain_f32(one, two, 1, 8, 56, 0.5, 0, 0.1, 0.5, 0.5, C.byref(nd), null) == 2        # width below 344
    assert lib.vrg_sharpen_grain_f32(one, two, 1, 8, 514, 0.5, 0, 0.1, 0.5, 0.5, C.byref(nd), null) == 2       # width % 4 != 0
    assert lib.vrg_sharpen_grain_f32(one, two, 2, 8, 512, 0.5, 0, 0.1, 0.5, 0.5, C.byref(nd2), null) == 2      # several frames per noise chunk
    assert lib.vrg_lut3d_f32(one, one, 0, 3, one, 17, f3, g3, 1, 1.0, 0.0, null) == 0
    assert lib.vrg_lut3d_f32(one, one, 4, 2, one, 17, f3, g3, 1, 1.0, 0.0, null) == 1            # C < 3
    assert lib.vrg_lut3d_f32(one, one, 4, 3, one, 17, f3, g3, 7, 1.0, 0.0, null) == 1            # bad blend mode
    assert lib.vrg_stencil3x3_f32(one, one, 0, 4, 4, 3, 0, 0, 0.5, null) == 0
    assert lib.vrg_stencil3x3_f32(one, one, 1, 4, 4, 3, 9, 0, 0.5, null) == 1
    assert lib.vrg_stencil3x3_f32(one, one, 1, 4, 4, 3, 0, 5, 0.5, null) == 1
    assert lib.vrg_colormatch_apply_f32(one, one, 1, 4, 4, null, one, 1, 1.0, 0.0, 0, null) == 1
    cd = hip.ChainDesc(stages=0)
    assert lib.vrg_fused_chain_f32(one, one, 1, 4, 4, C.byref(cd), null) == 1
    cd = hip.ChainDesc(stages=8, variant=99, stencil_op=0, border=0)
    assert lib.vrg_fused_chain_f32(one, one, 1, 4, 4, C.byref(cd), null) == 2                    # unknown variant
    assert lib.vrg_fused_chain_f32(one, one, 0, 4, 4, C.byref(cd), null) == 0


def test_host_copy_splits_over_threads_and_validates(hip):
    """vrg_host_copy: the staging copy of pageable frames (no device involved): every byte arrives for sizes around the part and page
    boundaries and any thread count; nothing beyond `bytes` is written; bad arguments are refused."""
    import torch
    lib = hip.load_library()
    src = torch.randint(0, 256, ((5 << 21) + 4099,), dtype=torch.uint8)
    for nbytes in (0, 1, 4095, 4096, (1 << 21) - 1, (1 << 21) + 1, (4 << 21) + 8191, src.numel()):
        for threads in (0, 1, 2, 5, 64, 1000):
            dst = torch.full((src.numel() + 64,), 7, dtype=torch.uint8)
            assert lib.vrg_host_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, threads) == 0
            assert torch.equal(dst[:nbytes], src[:nbytes]), (nbytes, threads)
            assert int(dst[nbytes:].min()) == 7 and int(dst[nbytes:].max()) == 7, (nbytes, threads)
    one = C.c_void_p(16)
    assert lib.vrg_host_copy(None, one, 8, 1) == 1 and lib.vrg_host_copy(one, None, 8, 1) == 1
    assert lib.vrg_host_copy(one, one, -1, 1) == 1 and lib.vrg_host_copy(one, one, 8, -2) == 1
    assert lib.vrg_host_copy(None, None, 0, 0) == 0
