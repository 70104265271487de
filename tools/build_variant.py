"""Build an A/B variant of libvrgdg_hip.so with extra compiler flags / defines: tools/ab/lib_<name>.so
    python tools/build_variant.py b -fno-slp-vectorize
Used with VRGDG_HIP_LIB=tools/ab/lib_<name>.so (tools/ab_libs.sh)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-vrgamedevgirl_amd"))
import build_ext as be
name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "tools", "ab")
obj_dir = os.path.join(out_dir, "obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
cflags = [f for f in be.HIPCC_FLAGS if f != "-shared"] + extra
def one(src):
    obj = os.path.join(obj_dir, src + ".o")
    subprocess.run([be._hipcc(), *cflags, *be.EXTRA_FLAGS.get(src, ()), "-I", be.INCLUDE, "-c", os.path.join(be.CSRC, src), "-o", obj], check=True)
    return obj
with ThreadPoolExecutor(8) as pool:
    objs = list(pool.map(one, be.SOURCES))
lib = os.path.join(out_dir, f"lib_{name}.so")
subprocess.run([be._hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-ldl"], check=True)
print(lib)
