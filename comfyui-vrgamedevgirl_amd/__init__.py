"""comfyui-vrgamedevgirl_amd -- MI355X-native drop-in for the per-pixel video post-processing nodes of
comfyui-vrgamedevgirl (Fast Film Grain, Color Match To Reference, Fast Unsharp / Laplacian / Sobel Sharpen,
VRGDG_LUTS, VRGDG_MakeLUT).

ComfyUI discovers this directory under ``custom_nodes/`` and reads NODE_CLASS_MAPPINGS /
NODE_DISPLAY_NAME_MAPPINGS (same keys as the reference pack, __init__.py:100-112, 172-178 there).  Importing
the package needs neither a GPU nor the built library; running a node needs both (no CPU fallback).
"""
from __future__ import annotations

from . import VRGDG_IV_Adjustments as _iv
from . import nodes as _nodes

__version__ = "0.1.0"

NODE_CLASS_MAPPINGS = {}
NODE_DISPLAY_NAME_MAPPINGS = {}
for _mod in (_nodes, _iv):
    NODE_CLASS_MAPPINGS.update(_mod.NODE_CLASS_MAPPINGS)
    NODE_DISPLAY_NAME_MAPPINGS.update(_mod.NODE_DISPLAY_NAME_MAPPINGS)

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
