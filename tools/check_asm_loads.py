#!/usr/bin/env python3
"""Safety check for the inline-asm loads of attention_mfma.hip - GLOAD16_ASM (global, waited for by VM_WAIT) and the
transposing LDS reads vcol_frag_a / vcol_frag_asm (waited for by TR_WAIT): between an asm load and the explicit wait that
covers it (the next `s_waitcnt vmcnt(N)` / `s_waitcnt lgkmcnt(N)` inside an asm block; a counted LDS wait covers all but the newest N reads) no instruction may read or write the
load's destination registers - the compiler does not know the data is still in flight, so a copy or a spill there would
move garbage and free the register for something else.  Compiles the file to assembly and scans every kernel; exits
non-zero on a violation (also a CPU test: tests/test_lint.py).   python tools/check_asm_loads.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "efficientvlm_amd", "csrc", "attention_mfma.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "a.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src,
                    "-I" + os.path.join(ROOT, "include")], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


bad = total = 0
for f in re.split(r"\n(?=_Z[\w]+:)", txt):
    name = f.split(":", 1)[0]
    if not name.startswith("_Z"):
        continue
    lines = [l.strip() for l in f.splitlines()]
    in_asm, pending = False, []          # pending: [(line no, dest regs)]
    for i, l in enumerate(lines):
        if l.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if l.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        if in_asm and l.startswith("global_load_dwordx4"):
            dest = regs(l.split()[1].rstrip(","))
            pending.append((i, dest, "vm")); total += 1
            continue
        if in_asm and l.startswith(("ds_read_b64_tr_b16", "ds_read_b128")):
            dest = regs(l.split()[1].rstrip(","))
            pending.append((i, dest, "lgkm")); total += 1
            continue
        if in_asm and l.startswith("s_waitcnt vmcnt"):
            pending = [p for p in pending if p[2] != "vm"]
            continue
        m = re.match(r"s_waitcnt lgkmcnt\((\d+)\)", l) if in_asm else None
        if m:
            # a counted wait retires all but the newest N LDS reads: LDS operations complete in order, and whatever else sits
            # in the queue (the compiler's own LDS traffic, scalar loads) can only make the wait cover MORE of them
            n = int(m.group(1))
            lg = [p for p in pending if p[2] == "lgkm"]
            keep = set(id(p) for p in lg[len(lg) - n:]) if n else set()
            pending = [p for p in pending if p[2] != "lgkm" or id(p) in keep]
            continue
        if pending:
            used = set()
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", l):
                used |= regs(tok)
            for j, dest, _ in pending:
                if used & dest:
                    bad += 1
                    print(f"{name}: `{l}` touches v{sorted(used & dest)} of the asm load at +{j} before its wait")
print(f"{total} asm loads (global + transposing LDS) checked, {bad} violations")
sys.exit(1 if bad else 0)
