#!/usr/bin/env python3
"""Undefined-global check for the package (no linter in the image, no GPU in the build container: a NameError on a GPU-only
path would otherwise cost a gpurun round trip).  Every LOAD_GLOBAL of every code object must resolve in its module or in
builtins.    python tools/lint_names.py"""
import builtins, dis, importlib, os, pkgutil, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def walk(code):
    yield code
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            yield from walk(c)


def check(modname):
    mod = importlib.import_module(modname)
    src = getattr(mod, "__file__", None)
    if not src or not src.endswith(".py"):
        return []
    code = compile(open(src).read(), src, "exec")
    bad = []
    for c in walk(code):
        for ins in dis.get_instructions(c):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in mod.__dict__ and not hasattr(builtins, ins.argval):
                if c.co_name == "<module>" and ins.opname == "LOAD_NAME":
                    continue
                bad.append(f"{src}:{c.co_firstlineno} {c.co_name}: undefined name {ins.argval!r}")
    return bad


if __name__ == "__main__":
    import efficientvlm_amd
    mods = ["bench", "__graft_entry__"] + [m.name for m in pkgutil.walk_packages(efficientvlm_amd.__path__, "efficientvlm_amd.")
                                           if "libevlm" not in m.name]
    out = [b for m in mods for b in check(m)]
    print("\n".join(out) if out else f"ok: {len(mods)} modules")
    sys.exit(1 if out else 0)
