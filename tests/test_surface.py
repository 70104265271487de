"""Drop-in boundary: the node surface must be the reference's (widget specs, names, order, categories)."""
import json
import os

import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def surface():
    with open(os.path.join(GOLDEN, "node_surface.json"), encoding="utf-8") as fh:
        return json.load(fh)


def _norm(v):
    if isinstance(v, (tuple, list)):
        return [_norm(i) for i in v]
    if isinstance(v, dict):
        return {k: _norm(i) for k, i in v.items()}
    return v


IMAGE_NODES = ("FastFilmGrain", "ColorMatchToReference", "FastUnsharpSharpen", "FastLaplacianSharpen", "FastSobelSharpen")


def test_mapping_keys(pkg, surface):
    assert set(surface) == set(pkg.NODE_CLASS_MAPPINGS)
    assert set(pkg.NODE_DISPLAY_NAME_MAPPINGS) == set(pkg.NODE_CLASS_MAPPINGS)
    for key, ref in surface.items():
        assert pkg.NODE_DISPLAY_NAME_MAPPINGS[key] == ref["display"], key


@pytest.mark.parametrize("key", IMAGE_NODES)
def test_image_node_surface(pkg, surface, key):
    cls, ref = pkg.NODE_CLASS_MAPPINGS[key], surface[key]
    got = _norm(cls.INPUT_TYPES())
    assert got == ref["INPUT_TYPES"]
    assert list(got["required"]) == list(ref["INPUT_TYPES"]["required"])          # widget order
    assert list(cls.RETURN_TYPES) == ref["RETURN_TYPES"]
    assert cls.FUNCTION == ref["FUNCTION"] and cls.CATEGORY == ref["CATEGORY"] and cls.DESCRIPTION == ref["DESCRIPTION"]
    assert callable(getattr(cls, cls.FUNCTION))
    import inspect
    params = list(inspect.signature(getattr(cls, cls.FUNCTION)).parameters)
    assert params == ["self"] + list(ref["INPUT_TYPES"]["required"])


def test_unsharp_strength_widget_supports_values_up_to_ten(pkg):
    # the one surface property the reference's own tests pin (tests/test_unsharp_strength.py:9-14)
    spec = pkg.NODE_CLASS_MAPPINGS["FastUnsharpSharpen"].INPUT_TYPES()["required"]["strength"][1]
    assert spec["max"] == 10.0
    for other in ("FastLaplacianSharpen", "FastSobelSharpen"):
        assert pkg.NODE_CLASS_MAPPINGS[other].INPUT_TYPES()["required"]["strength"][1]["max"] == 2.0


@pytest.mark.parametrize("key", ["VRGDG_LUTS", "VRGDG_MakeLUT"])
def test_lut_node_surface(pkg, surface, key):
    cls, ref = pkg.NODE_CLASS_MAPPINGS[key], surface[key]
    got = _norm(cls.INPUT_TYPES())
    want = ref["INPUT_TYPES"]
    if key == "VRGDG_LUTS":
        names = got["required"]["lut_name"][0]
        assert names == sorted(names, key=str.lower) and all(n.lower().endswith(".cube") for n in names)
        got["required"]["lut_name"] = ["<dynamic list of .cube files>"]
    assert got == want
    assert list(got["required"]) == list(want["required"])
    assert list(cls.RETURN_TYPES) == ref["RETURN_TYPES"] and list(cls.RETURN_NAMES) == ref["RETURN_NAMES"]
    assert cls.FUNCTION == ref["FUNCTION"] and cls.CATEGORY == ref["CATEGORY"]


def test_lut_module_keeps_the_names_the_routes_import(pkg):
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as lvt
    assert os.path.isdir(iv.LUTS_DIR)
    for name in ("_load_lut", "_apply_cube_lut", "_parse_cube_file", "_resolve_device", "_get_luts_folder_state", "IS_CHANGED"):
        assert hasattr(iv.VRGDG_LUTS, name)
    assert callable(lvt._apply_lut_tensor) and callable(lvt._apply_film_grain_tensor)
    s = iv.VRGDG_LUTS.IS_CHANGED(None, "AMD_Identity_17.cube", "auto", 10.0)
    assert "AMD_Identity_17.cube" in s and s.endswith("|auto|10.0")
    assert iv.VRGDG_LUTS.IS_CHANGED(None, "No LUT files found", "cpu", 1.0) == "missing|cpu|1.0"
    assert "|missing|nope.cube|" in iv.VRGDG_LUTS.IS_CHANGED(None, "nope.cube", "auto", 2.0)
    with pytest.raises(FileNotFoundError):
        iv.VRGDG_LUTS._load_lut("nope.cube")
    with pytest.raises(ValueError):
        iv.VRGDG_LUTS._load_lut("No LUT files found")


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.zeros(1, 4, 4, 3)
    with pytest.raises(RuntimeError):
        pkg.NODE_CLASS_MAPPINGS["FastFilmGrain"]().apply_grain(x, 0.04, 0.5, 4)
    with pytest.raises(RuntimeError):
        pkg.NODE_CLASS_MAPPINGS["FastUnsharpSharpen"]().apply_unsharp(x, 0.5, False)
    with pytest.raises(RuntimeError):
        pkg.NODE_CLASS_MAPPINGS["VRGDG_LUTS"]().apply_lut(x, "AMD_Identity_17.cube", "auto", 10.0)
    with pytest.raises(RuntimeError):
        pkg.NODE_CLASS_MAPPINGS["VRGDG_LUTS"]().apply_lut(x, "AMD_Identity_17.cube", "cuda", 10.0)


def test_product_never_imports_the_oracle():
    import re
    from conftest import PKG_DIR
    for fn in os.listdir(PKG_DIR):
        if fn.endswith(".py"):
            src = open(os.path.join(PKG_DIR, fn), encoding="utf-8").read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
