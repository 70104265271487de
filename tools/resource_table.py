"""Compile every translation unit with -Rpass-analysis=kernel-resource-usage and print one row per kernel
(VGPRs, SGPRs, scratch, occupancy in waves per SIMD, LDS).  No GPU needed.

    python tools/resource_table.py > profiles/rNN_kernel_resource_usage.txt
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comfyui-vrgamedevgirl_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "comfyui-vrgamedevgirl_amd"))
import build_ext as be

def field(block, key):
    m = re.search(re.escape(key) + r": (\S+)", block)
    return m.group(1) if m else "?"

rows = []
for src in be.SOURCES + be.DEBUG_SOURCES:
    cflags = [f for f in be.HIPCC_FLAGS if f != "-shared"]
    r = subprocess.run([be._hipcc(), *cflags, *be.EXTRA_FLAGS.get(src, ()), "-I", be.INCLUDE, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"],
                       capture_output=True, text=True)
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        mangled = b.split("\n")[0].split(" ")[0].strip()
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void vrg::", "")
        rows.append((src, name, field(b, "VGPRs"), field(b, "AGPRs"), field(b, "SGPRs"), field(b, "ScratchSize [bytes/lane]"),
                     field(b, "Occupancy [waves/SIMD]"), field(b, "LDS Size [bytes/block]")))
print(f"{'file':18s} {'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'waves/SIMD':>10s} {'LDS B':>7s}")
for r in rows:
    print(f"{r[0]:18s} {r[1][:78]:78s} {r[2]:>5s} {r[3]:>5s} {r[4]:>5s} {r[5]:>8s} {r[6]:>10s} {r[7]:>7s}")
