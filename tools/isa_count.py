"""Static instruction mix per kernel of one translation unit (no GPU needed): hipcc --cuda-device-only -S, then count v_* / s_* /
ds_* / global_* between each kernel label and its s_endpgm.
    python tools/isa_count.py vrg_produce.hip [-DFOO ...] [--dump out.s] [--filter k_produce_lab]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-vrgamedevgirl_amd"))
import build_ext as be

args = sys.argv[1:]
src = args.pop(0)
dump = filt = None
if "--dump" in args:
    i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
if "--filter" in args:
    i = args.index("--filter"); filt = args[i + 1]; del args[i:i + 2]
cflags = [f for f in be.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
out = dump or "/tmp/isa_count.s"
subprocess.run([be._hipcc(), *cflags, *be.EXTRA_FLAGS.get(src, ()), *args, "-I", be.INCLUDE, "--cuda-device-only", "-S",
                os.path.join(be.CSRC, src), "-o", out], check=True)
text = open(out).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void vrg::", "")
    if filt and filt not in name:
        continue
    body = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]
    n = lambda p: sum(1 for l in body if re.match(p, l))
    trans = n(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
    print(f"{name[:70]:70s} total {len(body):5d}  valu {n(r'v_'):5d} (trans {trans:3d}, cmp {n(r'v_cmp'):3d}, cndmask {n(r'v_cndmask'):3d}, mad_u64 {n(r'v_mad_u64'):3d}, pk {n(r'v_pk_'):3d})"
          f"  salu {n(r's_'):5d}  lds {n(r'ds_'):4d}  vmem {n(r'(global|buffer|flat|scratch)_'):4d}  waitcnt {n(r's_waitcnt'):4d} barrier {n(r's_barrier'):2d}")
