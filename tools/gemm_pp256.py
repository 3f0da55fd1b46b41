#!/usr/bin/env python3
"""256x256 ping-pong GEMM: correctness against torch (same bf16 inputs, fp32 reference) and timing on the GD-step shapes.
Run twice to A/B against the 128x128 kernels:  EVLM_PP256_PCT=101 python tools/gemm_pp256.py   (disables the new kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from efficientvlm_amd import ops, _lib as L

dev = "cuda"
torch.manual_seed(0)

def run(name, I, J, K, reps=20, bias=False, res=False, act=0, dact=0, check=True):
    dt = torch.bfloat16
    P = (torch.randn((I, K), device=dev) * 0.5).to(dt)
    Q = (torch.randn((J, K), device=dev) * 0.05).to(dt)
    C = torch.empty((I, J), dtype=dt, device=dev)
    extra = {}
    b = r = h = aux = None
    if bias: b = torch.randn(J, device=dev); extra["bias"] = b
    if res: r = torch.randn((I, J), device=dev).to(dt); extra["residual"] = r; extra["ldx"] = J
    if act: h = torch.empty((I, J), dtype=dt, device=dev); extra["act"] = act; extra["preact"] = h; extra["ldx"] = J
    if dact: aux = torch.randn((I, J), device=dev).to(dt); extra["dact"] = dact; extra["aux"] = aux; extra["ldx"] = J
    f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, K, K, J, **extra)
    f(); torch.cuda.synchronize()
    err = ""
    if check:
        ref = P.float() @ Q.float().t()
        if bias: ref = ref + b
        pre = ref
        if act == L.ACT_GELU: ref = F.gelu(ref)
        if act == L.ACT_QUICK_GELU: ref = ref * torch.sigmoid(1.702 * ref)
        if dact == L.ACT_QUICK_GELU:
            x = aux.float(); s = torch.sigmoid(1.702 * x); ref = ref * (s + 1.702 * x * s * (1 - s))
        if res: ref = ref + r.float()
        d = (C.float() - ref).abs().max().item(); sc = ref.abs().max().item()
        err = f" maxerr {d:.4f} / {sc:.2f}"
        ok = d <= 1.2e-2 * sc
        if act:
            dh = (h.float() - pre).abs().max().item(); ok = ok and dh <= 1.2e-2 * pre.abs().max().item()
            err += f" preact {dh:.4f}"
        err += " OK" if ok else " FAIL"
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:30s} I={I:6d} J={J:6d} K={K:6d} {us:9.1f} us {2.0*I*J*K/us/1e6:8.1f} TF/s{err}", flush=True)

def run_dgrad(name, I, J, K, reps=20, dact=0):
    """dX[I,J] = dY[I,K] @ W[K,J]  (q_trans: W stored reduction-major)"""
    dt = torch.bfloat16
    P = (torch.randn((I, K), device=dev) * 0.5).to(dt)
    Q = (torch.randn((K, J), device=dev) * 0.05).to(dt)
    C = torch.empty((I, J), dtype=dt, device=dev)
    extra = {}
    aux = None
    if dact: aux = torch.randn((I, J), device=dev).to(dt); extra = dict(dact=dact, aux=aux, ldx=J)
    f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, K, J, J, q_trans=1, **extra)
    f(); torch.cuda.synchronize()
    ref = P.float() @ Q.float()
    if dact == L.ACT_QUICK_GELU:
        x = aux.float(); sg = torch.sigmoid(1.702 * x); ref = ref * (sg + 1.702 * x * sg * (1 - sg))
    if dact == L.ACT_GELU:
        x = aux.float(); ref = ref * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * 3.141592653589793) ** 0.5)
    d = (C.float() - ref).abs().max().item(); sc = ref.abs().max().item()
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:30s} I={I:6d} J={J:6d} K={K:6d} {us:9.1f} us {2.0*I*J*K/us/1e6:8.1f} TF/s maxerr {d:.4f} / {sc:.2f} {'OK' if d <= 1.2e-2 * sc else 'FAIL'}", flush=True)

def run_wgrad(name, I, J, K, reps=20, psum=False, accumulate=False):
    """dW[I,J] = dY[K,I]^T @ X[K,J]  (both reduction-major), f32 out; optional bias gradient and in-place accumulation"""
    dt = torch.bfloat16
    P = (torch.randn((K, I), device=dev) * 0.5).to(dt)
    Q = (torch.randn((K, J), device=dev) * 0.5).to(dt)
    C = torch.zeros((I, J), dtype=torch.float32, device=dev)
    ps = torch.zeros(I, dtype=torch.float32, device=dev) if psum else None
    if accumulate: C.fill_(0.25)
    f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, I, J, J, p_trans=1, q_trans=1, c_f32=1, psum=ps, accumulate=int(accumulate))
    f(); torch.cuda.synchronize()
    ref = P.float().t() @ Q.float()
    if accumulate: ref = ref + 0.25
    d = (C - ref).abs().max().item(); sc = ref.abs().max().item()
    msg = f"maxerr {d:.4f} / {sc:.2f} {'OK' if d <= 2e-3 * sc else 'FAIL'}"
    if psum:
        rp = P.float().sum(0); dp = (ps - rp).abs().max().item()
        msg += f" psum {dp:.4f} / {rp.abs().max().item():.1f} {'OK' if dp <= 2e-3 * rp.abs().max().item() + 1e-2 else 'FAIL'}"
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:30s} I={I:6d} J={J:6d} K={K:6d} {us:9.1f} us {2.0*I*J*K/us/1e6:8.1f} TF/s {msg}", flush=True)

M = 12608
run_dgrad("vit fc2 dX (dact)", M, 3072, 768, dact=L.ACT_QUICK_GELU)
run_dgrad("vit fc1 dX", M, 768, 3072)
run_dgrad("vit qkv dX", M, 768, 2304)
run_dgrad("vit out dX", M, 768, 768)
run_dgrad("text 4B fc2 dX", 7680, 3072, 768, dact=L.ACT_GELU)
run_dgrad("dgrad edge", 1000, 8 * 100, 320)
run_wgrad("vit fc1 dW + db", 3072, 768, M, psum=True, accumulate=True)
run_wgrad("vit fc2 dW + db", 768, 3072, M, psum=True, accumulate=True)
run_wgrad("vit qkv dW + db", 2304, 768, M, psum=True, accumulate=True)
run_wgrad("vit out dW + db", 768, 768, M, psum=True, accumulate=True)
run_wgrad("text out dW", 768, 768, 7680, psum=True)
run_wgrad("text out dW 2B", 768, 768, 3840, psum=True)
run_wgrad("wgrad edge", 520, 264, 1024, psum=True)
run("vit qkv fwd (bias)", M, 2304, 768, bias=True)
run("vit out_proj (bias+res)", M, 768, 768, bias=True, res=True)
run("vit fc1 teacher (bias+qgelu)", M, 3072, 768, bias=True)
run("vit fc1 student (+preact)", M, 3072, 768, bias=True, act=L.ACT_QUICK_GELU)
run("vit fc2 fwd (bias+res)", M, 768, 3072, bias=True, res=True)
run("dgrad-like dact", M, 3072, 768, dact=L.ACT_QUICK_GELU)
run("kv packed", M, 1536, 768, bias=True)
run("text 4B ffn1", 7680, 3072, 768, bias=True, act=L.ACT_GELU)
run("text 4B qkv", 7680, 2304, 768, bias=True)
run("text 4B out", 7680, 768, 768, bias=True, res=True)
run("text 4B ffn2", 7680, 768, 3072, bias=True, res=True)
run("text 2B ffn1", 3840, 3072, 768, bias=True, act=L.ACT_GELU)
run("edge rows/cols", 1000, 8 * 250, 192, bias=True, res=True)
run("K=128", 4096, 4096, 128)
run("K=192 (odd tile count)", 4096, 4096, 192)
run("square 4096", 4096, 4096, 4096)
run("square 8192", 8192, 8192, 8192, reps=5, check=False)
