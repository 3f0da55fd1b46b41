#!/usr/bin/env python3
"""The text-side products of the GD step (3 840 / 7 680 rows: 45-90 big tiles on 256 CUs) under the dispatcher's tuning
switches - run once per environment (the switches are read once per process):
  (default) | EVLM_PP128_MIN=64 (128 x 256 tiles for the 3 840-row shapes too) | EVLM_FORCE_MT=4 (128 x 128 tiles) |
  EVLM_PP128=0 (no 128 x 256 tiles).  Timed as 20 launches replayed from a hipGraph (no host launch cost)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L
dev = "cuda"
def bench(I, J, K, qt=0, reps=20, **kw):
    dt = torch.bfloat16
    P = (torch.randn((I, K), device=dev) * 0.5).to(dt)
    Q = (torch.randn((K, J) if qt else (J, K), device=dev) * 0.5).to(dt)
    C = torch.empty((I, J), dtype=dt, device=dev)
    extra = dict(bias=torch.randn(J, device=dev))
    if kw.get("res"): extra["residual"] = torch.randn((I, J), device=dev).to(dt); extra["ldx"] = J
    if kw.get("act"): extra["act"] = kw["act"]; extra["preact"] = torch.empty((I, J), dtype=dt, device=dev); extra["ldx"] = J
    f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, P.stride(0), Q.stride(0), J, p_trans=0, q_trans=qt, **extra)
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * reps) * 1e3
    return round(us, 1), round(2.0 * I * J * K / us / 1e6, 1), ops._lib().evlm_gemm_last_kernel().decode()
tag = {k: v for k, v in os.environ.items() if k.startswith("EVLM_")}
for name, I, J, K, kw in (("text out", 3840, 768, 768, dict(res=True)), ("text ffn2", 3840, 768, 3072, dict(res=True)),
                          ("text ffn1", 3840, 3072, 768, dict(act=L.ACT_GELU)), ("text qkv", 3840, 2304, 768, {}),
                          ("text dX qkv", 3840, 768, 2304, {}), ("fusion out", 7680, 768, 768, dict(res=True)),
                          ("fusion ffn2", 7680, 768, 3072, dict(res=True)), ("fusion ffn1", 7680, 3072, 768, dict(act=L.ACT_GELU)),
                          ("fusion dX qkv", 7680, 768, 2304, {})):
    us, tf, kn = bench(I, J, K, **kw)
    print(json.dumps(dict(env=tag, shape=name, I=I, J=J, K=K, us=us, tflops=tf, kernel=kn)), flush=True)
