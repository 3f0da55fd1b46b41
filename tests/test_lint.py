"""No linter in the image and no GPU in the build container: every global name the package's code objects load must
resolve (a NameError on a GPU-only path would surface on the GPU box only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_global_names_in_the_package():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_names.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
