"""Issue-cycle budget of a kernel's hot loop from its ISA (no GPU needed) and the MEASURED cost of each instruction class (tools/probe_valu_classes.py
on the MI355X -> profiles/r06_valu_instruction_classes.json): compiles one translation unit to assembly the way build_ext does, finds the loops of a
kernel (a label with a backward branch to it), and prints for the chosen loop its opcode histogram, the class each opcode is priced as, and the sum.

    python tools/isa_cost.py vrg_march.hip --kernel "k_chain_march<1, true, 4>" [--loop 0] [--source tools/ab/r06/x.hip] [--costs profiles/...json] [--json out.json]

--loop N: the N-th largest innermost loop (default 0).  Costs are SIMD cycles per wave64 instruction at 2 waves per SIMD, converted from the
probe's assumed 2.4 GHz to real cycles with the clock each probe launch ran at (profiles/r06_clock_probe_classes_grbm.txt) where measured."""
import argparse, collections, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-vrgamedevgirl_amd"))
import build_ext as be

ap = argparse.ArgumentParser()
ap.add_argument("unit")
ap.add_argument("--kernel", required=True)
ap.add_argument("--loop", type=int, default=0)
ap.add_argument("--source", default="")
ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r06_valu_instruction_costs.json"))
ap.add_argument("--json", default="")
ap.add_argument("--list", action="store_true", help="list the loops of the kernel and stop")
ap.add_argument("--label", default="", help="price the lines from this label to the LAST backward branch to it (a loop whose body holds conditional blocks is not 'innermost' for --loop); "
                "blocks that are skipped at run time (the powers' transcriptions) are counted as if executed: an upper bound")
ap.add_argument("--exclude-after", default="", help="with --label: comma-separated substrings; a line range s_cbranch_execz X .. X: that holds one of them in any line is left out "
                "(0x3e76c4e1 = a coefficient of ocml's epln: the transcriptions of powf; v_div_fmas_f32: the IEEE divisions behind the unscaled / FMA forms)")
ap.add_argument("extra", nargs="*")
a, unknown = ap.parse_known_args()
extra = list(a.extra) + unknown
src = os.path.join(be.CSRC, a.unit)
tmp = None
if a.source:
    tmp = os.path.join(be.CSRC, f"_isa_{os.getpid()}_{os.path.basename(a.source)}")
    shutil.copyfile(a.source, tmp)
    src = tmp
    extra.append("-DVRG_LAB_VARIANT_SOURCE")
out = f"/tmp/isa_cost_{os.getpid()}.s"
try:
    cflags = [f for f in be.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([be._hipcc(), *cflags, *be.EXTRA_FLAGS.get(a.unit, ()), *extra, "-I", be.INCLUDE, "--cuda-device-only", "-S", "-x", "hip", src, "-o", out], check=True)
finally:
    if tmp:
        os.remove(tmp)
text = open(out).read()
os.remove(out)
body = None
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void vrg::", "")
    if name == a.kernel or (a.kernel in name and body is None):
        body, kname = m.group(2), name
        if name == a.kernel:
            break
if body is None:
    raise SystemExit(f"kernel {a.kernel!r} not found")
lines = []
for l in body.split("\n"):
    l = l.split(";")[0].split("//")[0].rstrip()
    if not l.strip() or l.strip().startswith("."):
        if re.match(r"^\.LBB\w+:", l.strip()):
            lines.append(l.strip())
        continue
    lines.append(l.strip())
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
loops = []
for j, l in enumerate(lines):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        i = labels.get(tgt)
        if i is not None and i < j:
            loops.append((i, j))
inner = [(i, j) for (i, j) in loops if not any((i2 > i or j2 < j) and i2 >= i and j2 <= j for (i2, j2) in loops if (i2, j2) != (i, j))]
inner.sort(key=lambda ij: ij[0] - ij[1])
if a.list or (not inner and not a.label):
    for k, (i, j) in enumerate(inner):
        print(f"loop {k}: {lines[i]} .. line {j}: {j - i} lines")
    outer = {}
    for (i, j) in loops:
        outer[i] = max(outer.get(i, 0), j)
    for i, j in sorted(outer.items(), key=lambda ij: ij[0] - ij[1])[:8]:
        print(f"label {lines[i]} .. last backward branch at line {j}: {j - i} lines   (--label)")
    raise SystemExit(0)
if a.label:
    i = labels[a.label.rstrip(":")]
    j = max(jj for (ii, jj) in loops if ii == i)
    body = lines[i:j + 1]
    if a.exclude_after:
        marks = a.exclude_after.split(",")

        def strip(seq):
            """drop the innermost `s_cbranch_execz X .. X:` ranges that hold a marker (the ranges around them stay)"""
            out, k = [], 0
            while k < len(seq):
                m = re.match(r"s_cbranch_execz\s+(\.LBB\w+)", seq[k])
                if m and (m.group(1) + ":") in seq[k:]:
                    end = k + seq[k:].index(m.group(1) + ":")
                    inner = strip(seq[k + 1:end])
                    out.append(seq[k])
                    if not any(any(mk in x for mk in marks) for x in inner):
                        out += inner
                    k = end
                    continue
                out.append(seq[k]); k += 1
            return out
        body = strip(body)
    ops = [l for l in body if not l.endswith(":")]
else:
    i, j = inner[a.loop]
    ops = [l for l in lines[i:j + 1] if not l.endswith(":")]
hist = collections.Counter()
for l in ops:
    op = l.split()[0]
    if op.startswith("v_") and ("dpp" in l.split(None, 1)[1] if len(l.split(None, 1)) > 1 else False) and not op.endswith("_dpp"):
        op += "_dpp"
    if op.startswith("v_") and " sdwa" in l or "_sel:" in l:
        op = op if op.endswith("_sdwa") else op + "_sdwa"
    hist[op] += 1
costs = json.load(open(a.costs)) if os.path.exists(a.costs) else {"classes": {}, "opcodes": {}, "default_valu": 4.0}


def price(op):
    """(class name, cycles) of one opcode"""
    oc = costs.get("opcodes", {})
    if op.endswith("_dpp"):                        # any DPP form issues at the half rate, whatever the base opcode
        return "dpp", costs["classes"].get("dpp", 4.0)
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    for key in (op, base):
        if key in oc:
            c = oc[key]
            return c, costs["classes"][c]
    for pat, c in costs.get("patterns", []):
        if re.match(pat, base):
            return c, costs["classes"][c]
    if op.startswith("v_"):
        return "valu_unmeasured", costs.get("default_valu", 4.0)
    return "other", 0.0


rows, total, by_class = [], 0.0, collections.Counter()
for op, n in hist.most_common():
    c, cyc = price(op)
    rows.append((op, n, c, cyc, n * cyc))
    total += n * cyc
    by_class[c] += n * cyc
valu = sum(n for op, n in hist.items() if op.startswith("v_"))
print(f"# {kname}: loop {a.loop} = {lines[i]} ({len(ops)} instructions, {valu} VALU)")
print(f"{'opcode':34s} {'n':>5s}  {'class':22s} {'cycles each':>11s} {'cycles':>9s}")
for op, n, c, cyc, t in rows:
    print(f"{op:34s} {n:5d}  {c:22s} {cyc:11.2f} {t:9.1f}")
print(f"{'VALU issue cycles per iteration':62s} {total:9.1f}")
for c, t in by_class.most_common():
    print(f"   {c:30s} {t:9.1f}  {100 * t / max(total, 1e-9):5.1f} %")
if a.json:
    json.dump({"kernel": kname, "loop_label": lines[i], "instructions": len(ops), "valu": valu, "issue_cycles": total, "by_class": dict(by_class),
               "opcodes": [{"op": op, "n": n, "class": c, "cycles_each": cyc} for op, n, c, cyc, _ in rows]}, open(a.json, "w"), indent=1)
