"""dX = dY W with W read as stored (reduction-major Q operand, tr16 LDS reads) against the K-contiguous W^T copy:
   python tools/qt_bench.py       (launches replayed from a hipGraph)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientvlm_amd import ops, _lib as L


def timeit(f, n=40):
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


# (I rows, J = in features, K = out features): dX[I, J] = dY[I, K] W[K, J]
for I, J, K in [(12608, 768, 2304), (12608, 768, 3072), (12608, 3072, 768), (12608, 768, 768), (7680, 768, 3072),
                (7680, 768, 768), (7680, 3072, 768), (3840, 768, 3072)]:
    dY = torch.randn(I, K, device="cuda").bfloat16()
    W = (torch.randn(K, J, device="cuda") * 0.05).bfloat16()
    Wt = W.t().contiguous()
    out = torch.empty(I, J, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops._gemm(L.BF16, dY, Wt, out, I, J, K, K, K, J))
    k0 = L.load().evlm_gemm_last_kernel().decode()
    t1 = timeit(lambda: ops._gemm(L.BF16, dY, W, out, I, J, K, K, J, J, q_trans=1))
    k1 = L.load().evlm_gemm_last_kernel().decode()
    fl = 2.0 * I * J * K
    print(f"{I:6d} x {J:5d} x {K:5d}   W^T copy {t0:7.1f} us {fl / t0 / 1e6:7.1f} TF/s {k0:36s} as stored {t1:7.1f} us {fl / t1 / 1e6:7.1f} TF/s {k1}")
