"""LayerNorm forward / backward kernel times on the step's shapes: python tools/ln_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientvlm_amd import ops

def timeit(f, n=100):
    """us per call, launches replayed from a hipGraph (host launch cost out of the picture)"""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3

for rows in (12608, 7680, 3840):
  for old in ("0", "1"):
    os.environ["EVLM_LN_FWD_3PASS"] = old
    x = torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(768, device="cuda"); b = torch.randn(768, device="cuda")
    with torch.no_grad():
        t = timeit(lambda: ops.layer_norm(x, w, b, 1e-5))
    mb = rows * 768 * 2 * 2 / 1e6
    print(f"ln_fwd rows={rows} three_pass={old}: {t:.2f} us  ({mb / t:.2f} TB/s of r+w, cache-resident input)")

# backward (dx + the deferred column sums of dgamma / dbeta as the training step runs them): forward + backward replayed
# from one graph, the forward's time subtracted
for rows in (12608, 7680, 3840):
    x = torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16).requires_grad_()
    w = torch.nn.Parameter(torch.randn(768, device="cuda")); b = torch.nn.Parameter(torch.randn(768, device="cuda"))
    gy = torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16)
    def fb():
        y = ops.layer_norm(x, w, b, 1e-5)
        torch.autograd.grad(y, (x, w, b), gy)
    def f():
        with torch.no_grad():
            ops.layer_norm(x, w, b, 1e-5)
    os.environ["EVLM_LN_FWD_3PASS"] = "0"
    tf, tfb = timeit(f, 50), timeit(fb, 50)
    mb = rows * 768 * 2 * 3 / 1e6
    print(f"ln_bwd rows={rows}: ~{tfb - tf:.2f} us (fwd+bwd {tfb:.2f}, fwd {tf:.2f})  ({mb / (tfb - tf):.2f} TB/s of dy + x read, dx written)")

# ... and the pre-LN block form (layer_norm_fork: the residual branch's gradient summed inside the backward kernel)
for rows in (12608, 7680, 3840):
    x = torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16).requires_grad_()
    w = torch.nn.Parameter(torch.randn(768, device="cuda")); b = torch.nn.Parameter(torch.randn(768, device="cuda"))
    gy = torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16)
    gx = torch.randn(rows, 768, device="cuda", dtype=torch.bfloat16)
    def fb():
        y, xa = ops.layer_norm_fork(x, w, b, 1e-5)
        torch.autograd.grad((y, xa), (x, w, b), (gy, gx))
    def f():
        with torch.no_grad():
            ops.layer_norm(x, w, b, 1e-5)
    tf, tfb = timeit(f, 50), timeit(fb, 50)
    mb = rows * 768 * 2 * 4 / 1e6
    print(f"ln_bwd + residual gradient rows={rows}: ~{tfb - tf:.2f} us (fwd+bwd {tfb:.2f}, fwd {tf:.2f})  ({mb / (tfb - tf):.2f} TB/s of dy + x + addend read, dx written)")
