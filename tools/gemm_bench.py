#!/usr/bin/env python3
"""Micro-benchmark of evlm_gemm on the shapes of the GD step (B=64): TFLOP/s per variant, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L

dev = "cuda"
LIB = "--lib" in sys.argv
def bench(name, I, J, K, pt, qt, reps=20, **kw):
    dt = torch.bfloat16
    P = (torch.randn((K, I) if pt else (I, K), device=dev) * 0.5).to(dt)
    Q = (torch.randn((K, J) if qt else (J, K), device=dev) * 0.5).to(dt)
    C = torch.empty((I, J), dtype=torch.float32 if kw.get("c_f32") else dt, device=dev)
    extra = {}
    if kw.get("bias"): extra["bias"] = torch.randn(J, device=dev)
    if kw.get("res"): extra["residual"] = torch.randn((I, J), device=dev).to(dt); extra["ldx"] = J
    if kw.get("act"): extra["act"] = kw["act"]; extra["preact"] = torch.empty((I, J), dtype=dt, device=dev); extra["ldx"] = J
    if kw.get("dact"): extra["dact"] = kw["dact"]; extra["aux"] = torch.randn((I, J), device=dev).to(dt); extra["ldx"] = J
    if kw.get("c_f32"): extra["c_f32"] = 1
    f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, P.stride(0), Q.stride(0), J, p_trans=pt, q_trans=qt, **extra)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    # measuring stick (never on the product path): the vendor library's plain product of the same operands through
    # torch.matmul (hipBLASLt / rocBLAS), no epilogue - bias / residual / activation would be extra launches there
    lib = ""
    if LIB and not kw.get("c_f32"):
        A = P.t() if pt else P; Bm = Q if qt else Q.t()
        g = lambda: torch.matmul(A, Bm)
        for _ in range(3): g()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): g()
        e1.record(); torch.cuda.synchronize()
        lus = e0.elapsed_time(e1) / reps * 1e3
        lib = f"   library plain product {lus:8.1f} us {2.0*I*J*K/lus/1e6:8.1f} TF/s"
    print(f"{name:34s} I={I:6d} J={J:6d} K={K:6d} pt={pt} qt={qt}  {us:9.1f} us  {2.0*I*J*K/us/1e6:8.1f} TF/s{lib}", flush=True)

M = 12608
bench("vit qkv fwd (bias)", M, 2304, 768, 0, 0, bias=True)
bench("vit out_proj fwd (bias+res)", M, 768, 768, 0, 0, bias=True, res=True)
bench("vit fc1 fwd (bias+qgelu+preact)", M, 3072, 768, 0, 0, bias=True, act=L.ACT_QUICK_GELU)
bench("vit fc2 fwd (bias+res)", M, 768, 3072, 0, 0, bias=True, res=True)
bench("plain NT 3072x768", M, 3072, 768, 0, 0)
bench("vit fc2 dX (dact)", M, 3072, 768, 0, 1, dact=L.ACT_QUICK_GELU)
bench("vit fc1 dX", M, 768, 3072, 0, 1)
bench("vit fc1 dW (f32 out)", 3072, 768, M, 1, 1, c_f32=True)
bench("vit fc2 dW (f32 out)", 768, 3072, M, 1, 1, c_f32=True)
bench("vit qkv dW (f32 out)", 2304, 768, M, 1, 1, c_f32=True)
bench("fusion out (bias+res) 7680", 7680, 768, 768, 0, 0, bias=True, res=True)
bench("text out (bias+res) 3840", 3840, 768, 768, 0, 0, bias=True, res=True)
bench("fusion ffn2 (bias+res) 7680", 7680, 768, 3072, 0, 0, bias=True, res=True)
bench("fusion ffn1 7680", 7680, 3072, 768, 0, 0, bias=True, act=L.ACT_GELU)
bench("text qkv fwd", 1920, 2304, 768, 0, 0, bias=True)
bench("text ffn1 fwd", 3840, 3072, 768, 0, 0, bias=True, act=L.ACT_GELU)
bench("mlm decoder fwd", 512, 30528, 768, 0, 0, bias=True)
bench("square 4096", 4096, 4096, 4096, 0, 0)
bench("square 8192", 8192, 8192, 8192, 0, 0, reps=5)
