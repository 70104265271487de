"""Build an A/B variant of libvrgdg_hip.so: tools/ab/lib_<name>.so
    python tools/build_variant.py b -fno-slp-vectorize                               (every unit with the extra flags)
    python tools/build_variant.py c --unit vrg_march.hip -DVRG_MARCH_ROTATE=0        (only that unit rebuilt; the others linked from csrc/.obj)
    python tools/build_variant.py d --unit vrg_march.hip --source tools/ab/old.hip --no-unit-flags -mllvm -amdgpu-sched-strategy=max-ilp
    python tools/build_variant.py e --unit vrg_produce.hip,vrg_apply_march.hip --source self -DLAB_ZIV_NO_FALLBACK=1   (several units; "self" = the unit's own
                                                                  product source compiled as a lab variant, so that the LAB_ switches of the shared headers are let through)
--source: compile this file in the unit's place (it must sit in csrc/ or include its headers by absolute path: it is copied into csrc/ under a
temporary name); --no-unit-flags: drop build_ext.EXTRA_FLAGS of the unit (give the wanted ones explicitly).
Used with tools/ab_interleaved.py --libs name=tools/ab/lib_<name>.so,..."""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-vrgamedevgirl_amd"))
import build_ext as be
args = sys.argv[1:]
name = args.pop(0)
unit = source = None
unit_flags = True
if "--unit" in args:
    i = args.index("--unit"); unit = args[i + 1]; del args[i:i + 2]
if "--source" in args:
    i = args.index("--source"); source = args[i + 1]; del args[i:i + 2]
if "--no-unit-flags" in args:
    args.remove("--no-unit-flags"); unit_flags = False
extra = args
out_dir = os.path.join(ROOT, "tools", "ab")
obj_dir = os.path.join(out_dir, "obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
cflags = [f for f in be.HIPCC_FLAGS if f != "-shared"]
be.build(verbose=False)                      # the default objects exist and are current


def one(src):
    if unit is not None and src not in unit.split(","):
        return os.path.join(be.CSRC, ".obj", src + ".o")
    obj = os.path.join(obj_dir, src + ".o")
    path = os.path.join(be.CSRC, src)
    tmp = None
    if source is not None:
        tmp = os.path.join(be.CSRC, f"_variant_{name}_{src}")
        shutil.copyfile(path if source == "self" else (os.path.join(ROOT, source) if not os.path.isabs(source) else source), tmp)
        path = tmp
    try:
        flags = list(be.EXTRA_FLAGS.get(src, ())) if unit_flags else []
        if tmp:
            flags.append("-DVRG_LAB_VARIANT_SOURCE")        # a file that is not product source: vrg_common.hpp lets its -D switches through
        subprocess.run([be._hipcc(), *cflags, *flags, *extra, "-I", be.INCLUDE, "-x", "hip", "-c", path, "-o", obj], check=True)
    finally:
        if tmp:
            os.remove(tmp)
    return obj


with ThreadPoolExecutor(8) as pool:
    objs = list(pool.map(one, be.SOURCES))
lib = os.path.join(out_dir, f"lib_{name}.so")
subprocess.run([be._hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-ldl"], check=True)
print(lib)
