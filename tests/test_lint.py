"""No linter in the image and no GPU in the build container: every global name the package's code objects load must
resolve (a NameError on a GPU-only path would surface on the GPU box only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_global_names_in_the_package():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_names.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr


def test_inline_asm_loads_are_not_touched_before_their_explicit_waits():
    """attention_mfma.hip issues some loads as inline asm (transposing LDS reads next to an LDS-DMA in flight; teacher-map
    pieces ahead of one) and waits for them by hand: the generated code must not read, copy or spill their destination
    registers before that wait (tools/check_asm_loads.py compiles the file for gfx950 - no GPU needed - and scans it)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 violations" in r.stdout
