"""Load the reference's own Python modules (TEST INFRASTRUCTURE ONLY).

Works only where ``/root/reference`` exists (the build container).  The GPU box
has no reference checkout: nothing under ``-m gpu`` tests, ``smoke()`` or
``bench.py`` may call this at run time; they use the committed fixtures in
``tests/golden/`` produced by ``oracle/make_golden.py``.

The reference's ``nodes.py`` imports ComfyUI / kornia / audio libraries at the
top (nodes.py:5-12) that are not installed; empty ``types.ModuleType`` stubs are
put in ``sys.modules`` for the duration of the import.  ``kornia.color``: the
INSTALLED library is used whenever it is importable (``installed_kornia()``) --
the fixtures of ``make_golden.py`` are then outputs of kornia itself and
``tests/test_oracle_golden.py::test_restated_lab_equals_installed_kornia`` pins
the restatement to it; where it is absent (this image: no index access) it is
stubbed with the restated Lab transforms of ``oracle.restated`` so that the
reference's *own* ``match_color`` control flow (statistics, blend, clamp,
permute) can still be executed.  ``kornia_source()`` says which one was used; it
is recorded in ``tests/golden/provenance.json``.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VRGDG_REFERENCE_ROOT", "/root/reference")

_STUB_NAMES = (
    "comfy", "comfy.model_management", "kornia", "kornia.color", "librosa",
    "torchaudio", "folder_paths", "av", "imageio", "requests",
)


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "nodes.py"))


def installed_kornia():
    """(kornia.color module, version string) of a REAL installed kornia, or None.  Never returns this file's stub."""
    import importlib
    for name in ("kornia.color", "kornia"):
        mod = sys.modules.get(name)
        if mod is not None and getattr(mod, "__vrgdg_stub__", False):
            return None                    # called while the stubs are installed and the stub is ours: kornia is absent
    try:
        kc = importlib.import_module("kornia.color")
        kornia = importlib.import_module("kornia")
    except Exception:
        return None
    if getattr(kc, "__vrgdg_stub__", False) or not hasattr(kc, "rgb_to_lab") or not hasattr(kc, "lab_to_rgb"):
        return None
    return kc, str(getattr(kornia, "__version__", "unknown"))


def kornia_source() -> str:
    """What stands behind kornia.color when the reference's modules are loaded here."""
    real = installed_kornia()
    return f"kornia {real[1]} (installed)" if real else "restated (oracle/restated.py: kornia is not installed)"


def _install_stubs():
    import torch
    from . import restated

    saved = {}
    for name in _STUB_NAMES:
        saved[name] = sys.modules.get(name)
    comfy = types.ModuleType("comfy")
    mm = types.ModuleType("comfy.model_management")
    mm.get_torch_device = lambda: torch.device("cpu")
    mm.intermediate_device = lambda: torch.device("cpu")
    comfy.model_management = mm
    sys.modules["comfy"] = comfy
    sys.modules["comfy.model_management"] = mm
    if installed_kornia() is None:
        kornia = types.ModuleType("kornia")
        kcolor = types.ModuleType("kornia.color")
        kornia.__vrgdg_stub__ = kcolor.__vrgdg_stub__ = True
        kcolor.rgb_to_lab = restated.kornia_rgb_to_lab
        kcolor.lab_to_rgb = restated.kornia_lab_to_rgb
        kornia.color = kcolor
        sys.modules["kornia"] = kornia
        sys.modules["kornia.color"] = kcolor
    # else: the reference's `import kornia.color` binds the installed library
    for name in ("librosa", "torchaudio", "folder_paths", "av", "imageio"):
        if saved[name] is None:
            sys.modules[name] = types.ModuleType(name)
    if saved["requests"] is None:
        try:
            import requests  # noqa: F401
        except Exception:
            sys.modules["requests"] = types.ModuleType("requests")
    return saved


def _restore_stubs(saved):
    for name, mod in saved.items():
        if mod is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = mod


def _load_file(modname: str, filename: str):
    path = os.path.join(REFERENCE_ROOT, filename)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_CACHE = {}


def load_nodes():
    """The reference's ``nodes.py`` (FastFilmGrain, ColorMatchToReference, Fast*Sharpen)."""
    if "nodes" not in _CACHE:
        saved = _install_stubs()
        try:
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                mod = _load_file("_vrgdg_reference_nodes", "nodes.py")
            # nodes.py imports numpy late (nodes.py:1325) as a module global; the
            # sharpen CPU paths rely on it.
            if not hasattr(mod, "np"):
                import numpy
                mod.np = numpy
            # keep the comfy / kornia stubs reachable from the module's globals
            _CACHE["nodes"] = mod
        finally:
            _restore_stubs(saved)
    return _CACHE["nodes"]


def load_iv_adjustments():
    """The reference's ``VRGDG_IV_Adjustments.py`` (needs only os, numpy, torch)."""
    if "iv" not in _CACHE:
        _CACHE["iv"] = _load_file("_vrgdg_reference_iv", "VRGDG_IV_Adjustments.py")
    return _CACHE["iv"]


class cv2_stub:
    """``with cv2_stub():`` -- the reference's frame converters import cv2 (absent here) for exactly one thing,
    ``cv2.cvtColor(frame, COLOR_BGR2RGB / COLOR_RGB2BGR)``, which for 3-channel uint8 frames is the channel
    reversal by definition.  Everything numeric in them (``astype(float32) / 255.0``, ``clip(x * 255, 0, 255)
    .astype(uint8)``) is numpy and runs as written."""

    def __enter__(self):
        import numpy as np
        self._saved = sys.modules.get("cv2")
        mod = types.ModuleType("cv2")
        mod.COLOR_BGR2RGB, mod.COLOR_RGB2BGR = 4, 4
        mod.cvtColor = lambda frame, code: np.ascontiguousarray(np.asarray(frame)[..., ::-1])
        sys.modules["cv2"] = mod
        return mod

    def __exit__(self, *exc):
        if self._saved is None:
            sys.modules.pop("cv2", None)
        else:
            sys.modules["cv2"] = self._saved
        return False


def _ast_extract(filename: str, names, namespace):
    """Same AST-exec pattern the reference's own tests use
    (tests/test_standalone_video_enhancer.py:20-36)."""
    path = os.path.join(REFERENCE_ROOT, filename)
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), namespace)
    return namespace


def load_lut_video_tools():
    """``_apply_film_grain_tensor``, ``_apply_lut_tensor`` (VRGDG_LUTVideoTools.py:172-185, 262-277)."""
    if "lvt" not in _CACHE:
        import torch
        iv = load_iv_adjustments()
        ns = {"torch": torch, "VRGDG_LUTS": iv.VRGDG_LUTS, "LUTS_DIR": iv.LUTS_DIR}
        _ast_extract(
            "VRGDG_LUTVideoTools.py",
            {"_apply_film_grain_tensor", "_apply_lut_tensor", "_normalize_adjust_settings",
             "_apply_adjust_tensor", "_frames_to_tensor", "_tensor_to_frames", "_process_video_batch",
             "_process_film_grain_batch", "_process_adjust_batch"},
            ns,
        )
        _CACHE["lvt"] = types.SimpleNamespace(**{k: v for k, v in ns.items()
                                                 if k.startswith(("_apply", "_normalize", "_frames", "_tensor", "_process"))})
    return _CACHE["lvt"]


def load_standalone_enhancer():
    """``_apply_unsharp``, ``_apply_seeded_grain``, ``_apply_effects_batch``
    (VRGDG_StandaloneVideoEnhancerNodes.py:233-294)."""
    if "sve" not in _CACHE:
        import torch
        import torch.nn.functional as F
        ns = {"torch": torch, "F": F}
        _ast_extract(
            "VRGDG_StandaloneVideoEnhancerNodes.py",
            {"_auto_batch_size", "_apply_unsharp", "_apply_seeded_grain", "_apply_effects_batch",
             "_process_with_retry", "_frames_to_tensor", "_tensor_to_frames"},
            ns,
        )
        _CACHE["sve"] = types.SimpleNamespace(**{k: v for k, v in ns.items() if k.startswith(("_a", "_p", "_f", "_t"))})
    return _CACHE["sve"]


def opening_color_match_reference(reference_rgb, target_rgb, workdir, strength=0.85, fade_seconds=1.0):
    """Run the numeric slice of the reference's ``_apply_scene_start_color_match``
    (VRGDG_WorkflowRunnerNodes.py:4382-4407: PIL.ImageStat of the two frames -> scales / offsets -> the 17^3 ``.cube``
    text and the ffmpeg blend-weight expression) on two HxWx3 uint8 RGB arrays.  The statements are taken from the
    reference's AST and executed as they stand; the ffmpeg calls around them (frame extraction, lut3d filter) are not run.
    Needs Pillow, which the reference imports for this function."""
    import numpy as np
    from PIL import Image, ImageStat
    path = os.path.join(REFERENCE_ROOT, "VRGDG_WorkflowRunnerNodes.py")
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_apply_scene_start_color_match")
    body = next(n for n in fn.body if isinstance(n, ast.Try)).body
    first = next(i for i, n in enumerate(body) if isinstance(n, ast.With) and "Image.open" in ast.unparse(n.items[0]))
    last = next(i for i, n in enumerate(body) if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]) == "weight")
    os.makedirs(workdir, exist_ok=True)
    ns = {"Image": Image, "ImageStat": ImageStat, "os": os, "strength": float(strength), "fade_seconds": float(fade_seconds),
          "reference_frame": os.path.join(workdir, "ref.png"), "target_frame": os.path.join(workdir, "tgt.png"),
          "cube_path": os.path.join(workdir, "match.cube")}
    Image.fromarray(np.ascontiguousarray(reference_rgb), "RGB").save(ns["reference_frame"])
    Image.fromarray(np.ascontiguousarray(target_rgb), "RGB").save(ns["target_frame"])
    exec(compile(ast.Module(body=body[first:last + 1], type_ignores=[]), path, "exec"), ns)
    with open(ns["cube_path"], "r", encoding="utf-8") as fh:
        cube_text = fh.read()
    return {k: ns[k] for k in ("reference_mean", "reference_std", "target_mean", "target_std", "scales", "offsets", "weight")} | {"cube_text": cube_text}
