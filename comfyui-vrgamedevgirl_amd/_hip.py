"""ctypes binding of libvrgdg_hip.so (the C ABI declared in include/vrgdg_hip.h).

There is NO CPU fallback: if the library is missing, or no HIP device is visible, every op raises.

The self-tests and probes of include/vrgdg_hip_debug.h live in a second library, libvrgdg_hip_debug.so, that nothing of the product loads:
it is opened the first time a test or a tool asks the handle for one of its names (`lib().vrg_debug_...`).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VRGDG_HIP_LIB") or os.path.join(PKG_DIR, "libvrgdg_hip.so")   # override: A/B builds (tools/build_variant.py, tools/ab_interleaved.py)
DEBUG_LIB_PATH = os.path.join(PKG_DIR, "libvrgdg_hip_debug.so")

VRG_OK = 0
VRG_ERR_BAD_ARG = 1
VRG_ERR_UNSUPPORTED = 2
BORDER_REPLICATE, BORDER_ZERO = 0, 1
STENCIL_UNSHARP, STENCIL_LAPLACIAN, STENCIL_SOBEL = 0, 1, 2
STAGE_GRAIN, STAGE_LUT, STAGE_COLORMATCH, STAGE_SHARPEN, STAGE_FROM_LAB = 1, 2, 4, 8, 16
CM_MATH_DEVICE, CM_MATH_FAST = 0, 1
ADJUST_DIV_IEEE, ADJUST_DIV_DEVICE = 0, 1
ABI_VERSION = 8


class NoiseDesc(C.Structure):
    """vrg_noise_desc"""
    _fields_ = [("seed0", C.c_uint64), ("seed_stride", C.c_uint64), ("offset0", C.c_uint64),
                ("offset_stride", C.c_uint64), ("chunk0", C.c_int64), ("chunk_frames", C.c_int32),
                ("grid_threads", C.c_uint32)]


class ChainDesc(C.Structure):
    """vrg_chain_desc"""
    _fields_ = [("stages", C.c_int32), ("variant", C.c_int32),
                ("intensity", C.c_float), ("sat", C.c_float), ("one_minus_sat", C.c_float),
                ("noise", NoiseDesc),
                ("lut", C.c_void_p), ("lut_size", C.c_int32),
                ("domain_min", C.c_float * 3), ("domain_max", C.c_float * 3),
                ("blend_mode", C.c_int32), ("blend", C.c_float), ("one_minus_blend", C.c_float),
                ("img_ms", C.c_void_p), ("ref_ms", C.c_void_p), ("ref_frames", C.c_int32),
                ("k", C.c_float), ("one_minus_k", C.c_float),
                ("stencil_op", C.c_int32), ("border", C.c_int32), ("strength", C.c_float),
                ("cm_math", C.c_int32)]


class AdjustDesc(C.Structure):
    """vrg_adjust_desc"""
    _fields_ = [("enabled", C.c_int32), ("shift", C.c_float * 3), ("exposure", C.c_float),
                ("contrast", C.c_float), ("saturation", C.c_float),
                ("highlights", C.c_float), ("shadows", C.c_float), ("whites", C.c_float), ("blacks", C.c_float),
                ("has_clarity", C.c_int32), ("clarity", C.c_float),
                ("has_sharpen", C.c_int32), ("sharpen", C.c_float),
                ("has_fade", C.c_int32), ("fade_mul", C.c_float), ("fade_add", C.c_float),
                ("has_vignette", C.c_int32), ("vignette", C.c_float),
                ("div_mode", C.c_int32)]


_F3 = C.c_float * 3
_P = C.c_void_p
_SIGNATURES = {
    "vrg_abi_version": (C.c_int, []),
    "vrg_error_string": (C.c_char_p, [C.c_int]),
    "vrg_device_info": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vrg_event_create": (C.c_int, [C.POINTER(_P)]),
    "vrg_event_record": (C.c_int, [_P, _P]),
    "vrg_event_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "vrg_event_destroy": (C.c_int, [_P]),
    "vrg_sharpen_grain_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_float,
                                        C.POINTER(NoiseDesc), _P]),
    "vrg_sharpen_grain_u8": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_float,
                                       C.POINTER(NoiseDesc), _P]),
    "vrg_grain_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                C.POINTER(NoiseDesc), _P]),
    "vrg_grain_injected_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, _P]),
    "vrg_lut_cells_floats": (C.c_int64, [C.c_int32]),
    "vrg_lut_prepare_f32": (C.c_int, [_P, C.c_int32, _P, _P]),
    "vrg_lut3d_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_int32, _F3, _F3, C.c_int32, C.c_float,
                                C.c_float, _P]),
    "vrg_stencil3x3_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_float, _P]),
    "vrg_adjust_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(AdjustDesc), _P]),
    "vrg_adjust_u8": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(AdjustDesc), _P]),
    "vrg_u8_channel_sums": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "vrg_u8bgr_to_f32rgb": (C.c_int, [_P, _P, C.c_int64, _P]),
    "vrg_f32rgb_to_u8bgr": (C.c_int, [_P, _P, C.c_int64, _P]),
    "vrg_lut3d_tetra_u8": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int32, _F3, _F3, _P, _P]),
    "vrg_fused_chain_u8": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(ChainDesc), _P]),
    "vrg_lab_stats_scratch_bytes": (C.c_int64, [C.c_int64]),
    "vrg_lab_stats_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P]),
    "vrg_lab_stats_finalize": (C.c_int, [_P, _P, C.c_int64, _P]),
    "vrg_lab_stats_torch_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_float, _P]),
    "vrg_lab_stats_torch_scratch_bytes": (C.c_int64, [C.c_int64]),
    "vrg_host_copy": (C.c_int, [_P, _P, C.c_int64, C.c_int32]),
    "vrg_lab_stats_torch_ws_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_float, _P, C.c_int64, _P]),
    "vrg_lab_stats_torch_lat_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_float, _P, C.c_int64, _P]),
    "vrg_stats_allreduce_scratch_bytes": (C.c_int64, [C.c_int64]),
    "vrg_stats_allreduce": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "vrg_colormatch_apply_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_float,
                                           C.c_float, C.c_int32, _P]),
    "vrg_fused_chain_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(ChainDesc), _P]),
    "vrg_chain_stats_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(ChainDesc), _P, _P, _P]),
    "vrg_chain_stats_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(ChainDesc)]),
    "vrg_chain_stats_lab_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(ChainDesc), _P, _P, _P]),
    "vrg_noise_f32": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(NoiseDesc), _P]),
    "vrg_selfcheck_pow_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
}

# include/vrgdg_hip_debug.h: self-tests and probes -- for the test suite and the measurement tools, not part of the drop-in boundary
_DEBUG_SIGNATURES = {
    "vrg_selftest_divconst": (C.c_int, [_P, _P]),
    "vrg_selftest_bm_radius": (C.c_int, [_P, _P]),
    "vrg_selftest_lanes": (C.c_int, [_P, _P]),
    "vrg_selftest_welford_division": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "vrg_debug_cm_math": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "vrg_debug_torch_reduce_config": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]),
    "vrg_debug_lut_fetch": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, _P]),
    "vrg_debug_copy_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "vrg_debug_valu_rate": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))
DEBUG_SYMBOLS = tuple(sorted(_DEBUG_SIGNATURES))

_lib = None
_lock = threading.Lock()


class Library:
    """Handle of the loaded product library.  Attribute access resolves the names of include/vrgdg_hip.h in libvrgdg_hip.so; a name of
    include/vrgdg_hip_debug.h -- and only such a name -- opens libvrgdg_hip_debug.so (once) and resolves there, so that tests and tools keep
    writing `lib().vrg_debug_...` while no node ever touches the debug library."""

    def __init__(self, cdll: C.CDLL, path: str, debug_path: str = DEBUG_LIB_PATH):
        self.__dict__["_cdll"] = cdll
        self.__dict__["_path"] = path
        self.__dict__["_debug_path"] = debug_path
        self.__dict__["_debug"] = None

    def debug(self) -> C.CDLL:
        if self._debug is None:
            with _lock:
                if self._debug is None:
                    if not os.path.exists(self._debug_path):
                        raise RuntimeError(f"libvrgdg_hip_debug.so not found at {self._debug_path} (self-tests / probes: build it with "
                                           f"`python {os.path.join(PKG_DIR, 'build_ext.py')}`)")
                    dbg = C.CDLL(self._debug_path)
                    for name, (res, args) in _DEBUG_SIGNATURES.items():
                        fn = getattr(dbg, name)   # AttributeError if the debug ABI drifted
                        fn.restype = res
                        fn.argtypes = args
                    self.__dict__["_debug"] = dbg
        return self._debug

    def __getattr__(self, name):
        if name in _DEBUG_SIGNATURES:
            return getattr(self.debug(), name)
        return getattr(self._cdll, name)


def load_library(path: str = LIB_PATH) -> Library:
    """dlopen the library and attach the prototypes (no device needed)."""
    import torch  # noqa: F401  -- FIRST: the HIP runtime must be the one torch loaded (same soname, shared streams/pointers)
    if not os.path.exists(path):
        raise RuntimeError(
            f"libvrgdg_hip.so not found at {path}: build it with `python {os.path.join(PKG_DIR, 'build_ext.py')}` "
            "(hipcc, gfx950). This package has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    if lib.vrg_abi_version() != ABI_VERSION:
        raise RuntimeError("libvrgdg_hip.so ABI version mismatch")
    return Library(lib, path)


def lib() -> Library:
    """The loaded library, for launching kernels: also requires a visible HIP device."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                import torch
                if not torch.cuda.is_available():
                    raise RuntimeError("comfyui-vrgamedevgirl_amd needs an AMD GPU (MI355X / gfx950) visible to "
                                       "PyTorch-ROCm; there is no CPU fallback")
                _lib = load_library()
    return _lib


def check(status: int, what: str):
    if status != VRG_OK:
        msg = lib().vrg_error_string(status).decode()
        if status == 1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg}")


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def current_stream() -> C.c_void_p:
    """torch's current stream of the CURRENT device.  ops.* make the tensors' device current first (ops._on_device), so
    that the stream, the kernels' hipGetDevice() and the pointers always belong to the same GPU."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
