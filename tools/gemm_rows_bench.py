#!/usr/bin/env python3
"""evlm_gemm on the ViT products of the ITR-384 / VQA-480 steps (64 x 577 = 36 928 rows, 32 x 901 = 28 832 rows): time and TFLOP/s of
the routed kernel, the kernel's name, tiles and rounds of 256 CUs, and the vendor library's plain product beside it.  Routing
switches come from the environment (EVLM_PP192=0, EVLM_PP192_MULTI=0, EVLM_PP128=0): run once per variant.
    python tools/gemm_rows_bench.py [rows ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L
dev = "cuda"
lib = L.load()
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def bench(name, I, J, K, qt=0, **kw):
    dt = torch.bfloat16
    P = (torch.randn((I, K), device=dev) * 0.5).to(dt)
    Q = (torch.randn((K, J) if qt else (J, K), device=dev) * 0.5).to(dt)
    C = torch.empty((I, J), dtype=dt, device=dev)
    extra = {}
    if kw.get("bias"): extra["bias"] = torch.randn(J, device=dev)
    if kw.get("res"): extra["residual"] = torch.randn((I, J), device=dev).to(dt); extra["ldx"] = J
    if kw.get("act"): extra["act"] = kw["act"]; extra["preact"] = torch.empty((I, J), dtype=dt, device=dev); extra["ldx"] = J
    if kw.get("dact"): extra["dact"] = kw["dact"]; extra["aux"] = torch.randn((I, J), device=dev).to(dt); extra["ldx"] = J
    us = timeit(lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, P.stride(0), Q.stride(0), J, p_trans=0, q_trans=qt, **extra))
    kern = lib.evlm_gemm_last_kernel().decode()
    Bm = Q if qt else Q.t()
    lus = timeit(lambda: torch.matmul(P, Bm))
    t256 = -(-I // 256) * -(-J // 256)
    print(f"{name:30s} I={I:6d} J={J:5d} K={K:5d}  {us:8.1f} us {2.0*I*J*K/us/1e6:7.1f} TF/s  {kern:38s} 256-tiles {t256:5d} = {t256/256:5.2f} rounds   library {lus:8.1f} us {2.0*I*J*K/lus/1e6:7.1f} TF/s", flush=True)
rows = [int(a) for a in sys.argv[1:] if a.isdigit()] or [36928, 28832]
for M in rows:
    bench("qkv fwd (bias)", M, 2304, 768, bias=True)
    bench("out_proj fwd (bias+res)", M, 768, 768, bias=True, res=True)
    bench("fc1 fwd (bias+qgelu+preact)", M, 3072, 768, bias=True, act=L.ACT_QUICK_GELU)
    bench("fc2 fwd (bias+res)", M, 768, 3072, bias=True, res=True)
    bench("qkv dX", M, 768, 2304, qt=1)
    bench("fc2 dX (plain: gated elsewhere)", M, 3072, 768, qt=1)
    bench("fc1 dX", M, 768, 3072, qt=1)
    bench("out_proj dX", M, 768, 768, qt=1)
