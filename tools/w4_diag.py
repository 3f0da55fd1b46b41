import sys, os
sys.path.insert(0, os.getcwd())
import torch
from efficientvlm_amd import ops, _lib as L
dev = "cuda"
def run(I, J, K, reps=20):
    P = (torch.randn((I, K), device=dev) * 0.5).bfloat16(); Q = (torch.randn((J, K), device=dev) * 0.05).bfloat16()
    C = torch.empty((I, J), dtype=torch.bfloat16, device=dev)
    f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, K, K, J)
    for _ in range(5): f()
    torch.cuda.synchronize()
    err = ""
    if os.environ.get("EVLM_W4_CHECK"):
        ref = P.float() @ Q.float().t()
        err = f" maxerr {float((C.float() - ref).abs().max()):.4f} / {float(ref.abs().max()):.2f}"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"W4={os.environ.get('EVLM_W4','0')} DIAG={os.environ.get('EVLM_W4_DIAG','0')} I={I} J={J} K={K} {us:8.1f} us {2.0*I*J*K/us/1e6:8.1f} TF/s{err}", flush=True)
run(4096, 4096, 4096); run(4096, 4096, 8192); run(8192, 8192, 4096, 8)
