"""``.cube`` 3D-LUT files: parser, writer, one-entry cache, palette LUT builder (host side, numpy).

Behavioural contract = VRGDG_IV_Adjustments.py:25-137, 203-282 of the reference: the table is returned as
``[N, N, N, 3]`` fp32 indexed ``[blue, green, red, rgb]`` (red varies fastest in the file), with
``domain_min`` / ``domain_max`` ``[3]``.  The tables are tiny (<= 431 KB for 33^3) and parsed once; the hot
path is the HIP trilinear kernel, not this file.
"""
from __future__ import annotations

import os

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


def sanitize_filename_part(value) -> str:
    cleaned = "".join(ch if ch.isalnum() else "_" for ch in str(value or "").strip().lower())
    cleaned = "_".join(part for part in cleaned.split("_") if part)
    return cleaned or "custom"


def parse_hex_color(token) -> np.ndarray:
    token = str(token or "").strip().lower()
    token = NAMED_COLORS.get(token, token)
    if token.startswith("#"):
        token = token[1:]
    if len(token) == 3:
        token = "".join(ch * 2 for ch in token)
    if len(token) != 6 or any(ch not in "0123456789abcdef" for ch in token):
        raise ValueError(f"Invalid color '{token}'. Use hex like #ff8800 or a basic color name.")
    return np.array([int(token[i:i + 2], 16) / 255.0 for i in (0, 2, 4)], dtype=np.float32)


def parse_color_list(colors_text) -> np.ndarray:
    parts = [p.strip() for p in str(colors_text or "").split(",") if p.strip()]
    if not parts:
        raise ValueError("Provide one or more colors separated by commas.")
    return np.stack([parse_hex_color(p) for p in parts], axis=0)


def build_palette_lut(colors_text, lut_size) -> torch.Tensor:
    """Luma-indexed palette LUT (VRGDG_IV_Adjustments.py:75-123): host-side, one-off."""
    palette = parse_color_list(colors_text)
    axis = np.linspace(0.0, 1.0, int(lut_size), dtype=np.float32)
    blue, green, red = np.meshgrid(axis, axis, axis, indexing="ij")
    source = np.stack([red, green, blue], axis=-1)
    w = (0.2126, 0.7152, 0.0722)
    luma = (w[0] * source[..., 0]) + (w[1] * source[..., 1]) + (w[2] * source[..., 2])
    if palette.shape[0] == 1:
        target = np.empty(luma.shape + (3,), dtype=np.float32)
        target[...] = palette[0]
    else:
        stops = np.linspace(0.0, 1.0, palette.shape[0], dtype=np.float32)
        flat = luma.reshape(-1)
        target = np.stack([np.interp(flat, stops, palette[:, c]) for c in range(3)], axis=-1)
        target = target.reshape(luma.shape + (3,)).astype(np.float32)
    target_luma = (w[0] * target[..., 0]) + (w[1] * target[..., 1]) + (w[2] * target[..., 2])
    scale = luma / np.maximum(target_luma, 1e-6)
    target = np.clip(target * scale[..., None], 0.0, 1.0)
    chroma = source - luma[..., None]
    mixed = np.clip((target * 0.82) + ((target + chroma) * 0.18), 0.0, 1.0)
    return torch.from_numpy(mixed.astype(np.float32))


def next_available_lut_path(luts_dir: str, base_name: str) -> str:
    os.makedirs(luts_dir, exist_ok=True)
    candidate = os.path.join(luts_dir, f"{base_name}.cube")
    index = 2
    while os.path.exists(candidate):
        candidate = os.path.join(luts_dir, f"{base_name}_{index}.cube")
        index += 1
    return candidate
