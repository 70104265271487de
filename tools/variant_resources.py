"""Resource usage (VGPRs, SGPRs, scratch, occupancy, LDS) of the kernels of ONE source file built the way tools/build_variant.py builds it.
    python tools/variant_resources.py tools/ab/r06/vrg_march_lab.hip [--filter march] -DLAB_WGW=1 ...
"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-vrgamedevgirl_amd"))
import build_ext as be
args = sys.argv[1:]
src = args.pop(0)
flt = ""
if "--filter" in args:
    i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
unit = "vrg_march.hip"
if "--unit" in args:
    i = args.index("--unit"); unit = args[i + 1]; del args[i:i + 2]
tmp = os.path.join(be.CSRC, f"_res_{os.getpid()}_{os.path.basename(src)}")
shutil.copyfile(src, tmp)
try:
    cflags = [f for f in be.HIPCC_FLAGS if f != "-shared"]
    r = subprocess.run([be._hipcc(), *cflags, *be.EXTRA_FLAGS.get(unit, ()), "-DVRG_LAB_VARIANT_SOURCE", *args, "-I", be.INCLUDE, "-Rpass-analysis=kernel-resource-usage",
                        "-x", "hip", "-c", tmp, "-o", "/dev/null"], capture_output=True, text=True)
finally:
    os.remove(tmp)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
def field(block, key):
    m = re.search(re.escape(key) + r": (\S+)", block)
    return m.group(1) if m else "?"
print(f"{'kernel':70s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS B':>7s}")
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    mangled = b.split("\n")[0].split(" ")[0].strip()
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void vrg::", "")
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} {field(b, 'VGPRs'):>5s} {field(b, 'SGPRs'):>5s} {field(b, 'ScratchSize [bytes/lane]'):>8s} {field(b, 'Occupancy [waves/SIMD]'):>4s} {field(b, 'LDS Size [bytes/block]'):>7s}")
