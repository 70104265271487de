"""``.cube`` 3D-LUT files: parser, writer, one-entry cache, palette LUT builder (host side, numpy).

Behavioural contract = VRGDG_IV_Adjustments.py:25-137, 203-282 of the reference: the table is returned as
``[N, N, N, 3]`` fp32 indexed ``[blue, green, red, rgb]`` (red varies fastest in the file), with
``domain_min`` / ``domain_max`` ``[3]``.  The tables are tiny (<= 431 KB for 33^3) and parsed once; the hot
path is the HIP trilinear kernel, not this file.
"""
from __future__ import annotations

import os
import re

import numpy as np
import torch

SUPPORTED_LUT_EXTENSIONS = (".cube",)
NO_LUTS = "No LUT files found"

NAMED_COLORS = {
    "black": "#000000", "white": "#ffffff", "red": "#ff0000", "green": "#00ff00", "blue": "#0000ff",
    "yellow": "#ffff00", "cyan": "#00ffff", "magenta": "#ff00ff", "orange": "#ffa500", "purple": "#800080",
    "pink": "#ffc0cb", "teal": "#008080",
}


def list_lut_files(luts_dir: str):
    if not os.path.isdir(luts_dir):
        return [NO_LUTS]
    names = [n for n in os.listdir(luts_dir)
             if os.path.isfile(os.path.join(luts_dir, n)) and n.lower().endswith(SUPPORTED_LUT_EXTENSIONS)]
    names.sort(key=str.lower)
    return names or [NO_LUTS]


def _three_floats(tokens, what, path):
    if len(tokens) != 4:
        raise ValueError(f"Invalid {what} line in {path}")
    return np.array([float(tokens[1]), float(tokens[2]), float(tokens[3])], dtype=np.float32)


def parse_cube_file(lut_path: str) -> dict:
    size = None
    domain = {"DOMAIN_MIN": np.zeros(3, dtype=np.float32), "DOMAIN_MAX": np.ones(3, dtype=np.float32)}
    data = []
    with open(lut_path, "r", encoding="utf-8", errors="ignore") as handle:
        for raw in handle:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            head = line.upper()
            if head.startswith("TITLE "):
                continue
            if head.startswith("LUT_1D_SIZE"):
                raise ValueError(f"1D LUTs are not supported: {os.path.basename(lut_path)}")
            tokens = line.split()
            if head.startswith("LUT_3D_SIZE"):
                if len(tokens) != 2:
                    raise ValueError(f"Invalid LUT_3D_SIZE line in {lut_path}")
                size = int(tokens[1])
            elif head.startswith("DOMAIN_MIN"):
                domain["DOMAIN_MIN"] = _three_floats(tokens, "DOMAIN_MIN", lut_path)
            elif head.startswith("DOMAIN_MAX"):
                domain["DOMAIN_MAX"] = _three_floats(tokens, "DOMAIN_MAX", lut_path)
            elif len(tokens) == 3:
                data.append((float(tokens[0]), float(tokens[1]), float(tokens[2])))
            # anything else (unknown keyword, wrong token count) is ignored, like the reference
    if size is None:
        raise ValueError(f"Missing LUT_3D_SIZE in {lut_path}")
    expected = size * size * size * 3
    if len(data) * 3 != expected:
        raise ValueError(f"Invalid LUT data length in {lut_path}. Expected {expected} floats, got {len(data) * 3}.")
    table = np.asarray(data, dtype=np.float64).astype(np.float32).reshape(size, size, size, 3)
    return {"size": size, "lut": torch.from_numpy(table),
            "domain_min": torch.from_numpy(domain["DOMAIN_MIN"]), "domain_max": torch.from_numpy(domain["DOMAIN_MAX"])}


def write_cube_file(lut_tensor, lut_path: str):
    size = int(lut_tensor.shape[0])
    table = lut_tensor.detach().cpu().numpy().reshape(-1, 3)
    os.makedirs(os.path.dirname(lut_path), exist_ok=True)
    with open(lut_path, "w", encoding="utf-8") as handle:
        handle.write(f'TITLE "{os.path.basename(lut_path)}"\n')
        handle.write(f"LUT_3D_SIZE {size}\n")
        handle.write("DOMAIN_MIN 0.0 0.0 0.0\n")
        handle.write("DOMAIN_MAX 1.0 1.0 1.0\n")
        handle.writelines("%.6f %.6f %.6f\n" % (r, g, b) for r, g, b in table)


_NOT_ALNUM = re.compile(r"[\W_]+")            # runs of anything str.isalnum() rejects (\w is isalnum() or "_")
_HEX6 = re.compile(r"[0-9a-f]{6}")
_LUMA = (0.2126, 0.7152, 0.0722)               # the palette generator's luma weights (VRGDG_IV_Adjustments.py:98, 101)


def sanitize_filename_part(value) -> str:
    """Lower-case, every run of non-alphanumerics one underscore, none at the ends; "custom" for nothing (the slug rule of the files
    VRGDG_MakeLUT writes: VRGDG_IV_Adjustments.py:38-41, 395-402 -- a contract: existing graphs find their cubes by these names)."""
    words = [w for w in _NOT_ALNUM.split(str(value or "").strip().lower()) if w]
    return "_".join(words) if words else "custom"


def parse_hex_color(token) -> np.ndarray:
    """"#rgb", "#rrggbb" (the "#" optional) or one of NAMED_COLORS -> fp32 RGB in [0, 1] (VRGDG_IV_Adjustments.py:44-64; the error text is
    the reference's)."""
    text = str(token or "").strip().lower()
    text = NAMED_COLORS.get(text, text).removeprefix("#")
    if len(text) == 3:
        text = text[0] * 2 + text[1] * 2 + text[2] * 2
    if not _HEX6.fullmatch(text):
        raise ValueError(f"Invalid color '{text}'. Use hex like #ff8800 or a basic color name.")
    rgb24 = int(text, 16)
    return np.array([(rgb24 >> shift & 0xFF) / 255.0 for shift in (16, 8, 0)], dtype=np.float32)


def parse_color_list(colors_text) -> np.ndarray:
    tokens = [t for t in (piece.strip() for piece in str(colors_text or "").split(",")) if t]
    if not tokens:
        raise ValueError("Provide one or more colors separated by commas.")
    return np.stack([parse_hex_color(t) for t in tokens], axis=0)


def _luma(rgb: np.ndarray) -> np.ndarray:
    # three products, two sums, in this order and in the arrays' own precision: the generator's table must come out bit for bit
    return (_LUMA[0] * rgb[..., 0]) + (_LUMA[1] * rgb[..., 1]) + (_LUMA[2] * rgb[..., 2])


def build_palette_lut(colors_text, lut_size) -> torch.Tensor:
    """The cube VRGDG_MakeLUT generates from a palette (VRGDG_IV_Adjustments.py:67-123): every grid colour keeps its luma and 18 % of its
    chroma and takes the rest from the palette entry its luma points at.  Host-side, one-off, numpy in the reference's operation order --
    tests/test_oracle_golden.py::test_palette_lut_matches_reference_generator holds the table to the reference's bit for bit."""
    palette = parse_color_list(colors_text)
    ramp = np.linspace(0.0, 1.0, int(lut_size), dtype=np.float32)
    b, g, r = np.meshgrid(ramp, ramp, ramp, indexing="ij")                 # table index order [b][g][r]
    grid = np.stack([r, g, b], axis=-1)
    y = _luma(grid)
    if len(palette) == 1:
        tint = np.broadcast_to(palette[0], y.shape + (3,)).astype(np.float32)
    else:
        knots = np.linspace(0.0, 1.0, len(palette), dtype=np.float32)
        tint = np.stack([np.interp(y.ravel(), knots, palette[:, ch]) for ch in range(3)], axis=-1).reshape(y.shape + (3,)).astype(np.float32)
    tint = np.clip(tint * (y / np.maximum(_luma(tint), 1e-6))[..., None], 0.0, 1.0)      # the palette colour at the grid colour's luma
    table = np.clip((tint * 0.82) + ((tint + (grid - y[..., None])) * 0.18), 0.0, 1.0)
    return torch.from_numpy(table.astype(np.float32))


def next_available_lut_path(luts_dir: str, base_name: str) -> str:
    """<base>.cube, else <base>_2.cube, <base>_3.cube, ... -- the first that does not exist (VRGDG_IV_Adjustments.py:126-137)."""
    import itertools
    os.makedirs(luts_dir, exist_ok=True)
    names = itertools.chain([f"{base_name}.cube"], (f"{base_name}_{n}.cube" for n in itertools.count(2)))
    return next(path for path in (os.path.join(luts_dir, name) for name in names) if not os.path.exists(path))
