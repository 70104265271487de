"""VRGDG_LUTS / VRGDG_MakeLUT nodes: 3D-LUT apply on the MI355X.

Keeps the names the rest of the reference pack imports from this module (``LUTS_DIR``, ``VRGDG_LUTS`` and its
``_load_lut`` / ``_apply_cube_lut`` class-level helpers -- VRGDG_LUTVideoTools.py:12, 173-178 there) and the
node surface of VRGDG_IV_Adjustments.py:140-434.  File handling lives in ``cube``; the trilinear kernel
in ``ops.lut3d``.
"""
from __future__ import annotations

import os

import torch

from . import cube, ops
from ._devices import Stage, compute_device, defer, stream_frames

LUTS_DIR = os.path.join(os.path.dirname(__file__), "LUTS")
SUPPORTED_LUT_EXTENSIONS = cube.SUPPORTED_LUT_EXTENSIONS
NAMED_COLORS = cube.NAMED_COLORS
_NO_LUTS = cube.NO_LUTS
_DEVICE_CHOICES = ["auto", "cuda", "cpu"]


def _list_lut_files():
    return cube.list_lut_files(LUTS_DIR)


def _strength_widget():
    return ("FLOAT", {"default": 10.0, "min": 0.0, "max": 10.0, "step": 0.1})


def _graded(image, lut_data, requested_device, strength):
    """Shared tail of both nodes: resolve device, run the kernel, hand the result back on image.device."""
    if isinstance(image, torch.Tensor) and image.dtype in (torch.float16, torch.bfloat16):
        # the reference grades half-precision images in fp32 and casts the RGB result back (IV_Adjustments.py:293, 341-342);
        # its partial-strength blend then runs in the image's dtype -- here the blend is fp32 as well, rounded once at the end.
        # (float64 images are promoted to an fp64 evaluation by the reference: not offered, ops.lut3d raises.)
        return _graded(image.to(torch.float32), lut_data, requested_device, strength).to(image.dtype)
    target = VRGDG_LUTS._resolve_device(requested_device, image)
    dev_lut = ops.upload_lut(lut_data, target)

    tables = {torch.device(target): dev_lut}

    def table_on(device):       # several GPUs (VRGDG_DEVICES): every device gets its own copy of the record table
        device = torch.device(device)
        if device not in tables:
            tables[device] = ops.upload_lut(lut_data, device)
        return tables[device]

    def on(device):
        lut_d = table_on(device)
        return lambda frames, _first, out=None: ops.lut3d(frames, lut_d, strength, out=out)

    fusable = image.dtype == torch.float32 and image.ndim == 4 and image.shape[0] > 0 and image.shape[-1] == 3
    stage = Stage("lut", on(target), 1, {"lut": dev_lut, "strength": strength},
                  for_device=lambda d: (on(d), {"lut": table_on(d), "strength": strength})) if fusable else None
    if image.device.type == "cpu" and image.dtype == torch.float32 and image.ndim == 4 and image.shape[0] > 0:
        # CPU tensor in, CPU tensor out: uploads, kernels and downloads overlapped (see _devices.stream_frames); recorded and fused with
        # the neighbouring nodes of this pack where the graph allows it (_devices.defer)
        return stream_frames(image, on(target), fn_for_device=on, stage=stage)
    if stage is not None and image.is_cuda and image.device == torch.device(target):
        res = defer(image, torch.device(target), stage, image.device)      # device frames in a device-resident graph: same deferral
        if res is not None:
            return res
    working = image.to(device=target)
    return ops.lut3d(working, dev_lut, strength).to(device=image.device)


class VRGDG_LUTS:
    CATEGORY = "VRGDG/IV Adjustments"
    RETURN_TYPES = ("IMAGE",)
    RETURN_NAMES = ("image",)
    FUNCTION = "apply_lut"

    _LUT_CACHE = {}

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "image": ("IMAGE",),
            "lut_name": (_list_lut_files(),),
            "device": (_DEVICE_CHOICES, {"default": "auto"}),
            "strength": _strength_widget(),
        }}

    @classmethod
    def IS_CHANGED(cls, image, lut_name, device, strength):
        if lut_name == _NO_LUTS:
            return f"missing|{device}|{strength}"
        state = cls._get_luts_folder_state()
        path = os.path.join(LUTS_DIR, lut_name)
        if not os.path.isfile(path):
            return f"{state}|missing|{lut_name}|{device}|{strength}"
        return f"{state}|{lut_name}|{os.path.getmtime(path)}|{device}|{strength}"

    @staticmethod
    def _resolve_device(requested_device, image):
        """'cuda' without a GPU is the reference's RuntimeError; every other choice computes on the MI355X
        (this implementation has no CPU path -- 'cpu' only decides where the *result* lives, which is
        ``image.device`` anyway)."""
        requested = str(requested_device or "auto").strip().lower()
        if requested == "cuda" and not torch.cuda.is_available():
            raise RuntimeError("VRGDG_LUTS: CUDA was selected, but CUDA is not available.")
        if image.device.type == "cuda":
            return image.device
        return compute_device()

    @staticmethod
    def _get_luts_folder_state():
        """The IS_CHANGED fingerprint of the LUTS folder (reference :186-201): "missing", "empty", or name:mtime:size of every cube."""
        if not os.path.isdir(LUTS_DIR):
            return "missing"

        def fingerprint(name):
            path = os.path.join(LUTS_DIR, name)
            try:
                return f"{name}:{os.path.getmtime(path)}:{os.path.getsize(path)}"
            except OSError:
                return f"{name}:missing"

        return "|".join(fingerprint(n) for n in _list_lut_files() if n != _NO_LUTS) or "empty"

    @classmethod
    def _load_lut(cls, lut_name):
        if lut_name == _NO_LUTS:
            raise ValueError("No LUT files were found in the LUTS folder.")
        path = os.path.join(LUTS_DIR, lut_name)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"LUT file not found: {path}")
        key = (path, os.path.getmtime(path), os.path.getsize(path))
        hit = cls._LUT_CACHE.get(key)
        if hit is None:
            hit = cls._parse_cube_file(path)
            cls._LUT_CACHE = {key: hit}   # one-entry cache, replaced atomically (nodes may run on several threads)
        return hit

    @staticmethod
    def _parse_cube_file(lut_path):
        return cube.parse_cube_file(lut_path)

    @classmethod
    def _apply_cube_lut(cls, image, lut_tensor, domain_min, domain_max):
        """Full-strength trilinear apply on tensors (the helper the pack's HTTP routes call directly)."""
        if image.ndim != 4 or image.shape[-1] < 3:
            raise ValueError("VRGDG_LUTS expects IMAGE input shaped like [batch, height, width, channels].")
        dev = image.device if image.device.type == "cuda" else compute_device()
        lut_data = {"lut": lut_tensor, "domain_min": domain_min, "domain_max": domain_max}
        out = ops.lut3d(image.to(device=dev, dtype=torch.float32), ops.upload_lut(lut_data, dev), 10.0)
        return out.to(device=image.device, dtype=image.dtype)

    def apply_lut(self, image, lut_name, device, strength):
        return (_graded(image, self._load_lut(lut_name), device, strength),)


class VRGDG_MakeLUT:
    CATEGORY = "VRGDG/IV Adjustments"
    RETURN_TYPES = ("IMAGE", "STRING", "STRING")
    RETURN_NAMES = ("image", "lut_name", "lut_path")
    FUNCTION = "create_and_apply"

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "image": ("IMAGE",),
            "colors": ("STRING", {"default": "#0b1d51, #1f6aa5, #f3d27a", "multiline": False}),
            "name_suffix": ("STRING", {"default": "palette", "multiline": False}),
            "lut_size": ("INT", {"default": 33, "min": 8, "max": 128, "step": 1}),
            "device": (_DEVICE_CHOICES, {"default": "auto"}),
            "strength": _strength_widget(),
        }}

    @classmethod
    def IS_CHANGED(cls, image, colors, name_suffix, lut_size, device, strength):
        return f"{colors}|{name_suffix}|{lut_size}|{device}|{strength}"

    def create_and_apply(self, image, colors, name_suffix, lut_size, device, strength):
        table = cube.build_palette_lut(colors, lut_size)
        color_slug = "_".join(cube.sanitize_filename_part(p) for p in str(colors).split(",") if p.strip())
        suffix_slug = cube.sanitize_filename_part(name_suffix)
        base = f"{color_slug}_{suffix_slug}" if suffix_slug else color_slug
        path = cube.next_available_lut_path(LUTS_DIR, base)
        cube.write_cube_file(table, path)
        lut_data = {"size": int(table.shape[0]), "lut": table,
                    "domain_min": torch.zeros(3, dtype=torch.float32), "domain_max": torch.ones(3, dtype=torch.float32)}
        return (_graded(image, lut_data, device, strength), os.path.basename(path), path)


NODE_CLASS_MAPPINGS = {"VRGDG_LUTS": VRGDG_LUTS, "VRGDG_MakeLUT": VRGDG_MakeLUT}
NODE_DISPLAY_NAME_MAPPINGS = {"VRGDG_LUTS": "VRGDG_LUTS", "VRGDG_MakeLUT": "VRGDG_MakeLUT"}
