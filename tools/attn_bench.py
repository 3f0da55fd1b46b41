#!/usr/bin/env python3
"""Isolated timings of the attention kernels on the GD step's shapes (one stream, events around 20 launches each):
forward with / without the map written, with the row lse (recomputing backward), with the fused map distillation; backward
from the stored map (EVLM_ATTN_STORE_P=1 form) and recomputing; ViT self-attention (B 64, 12 heads, 197 tokens), text
self-attention (256 x 30), cross-attention (256 text rows on 64 images, K/V shared through kv_index).
    python tools/attn_bench.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops

dev = "cuda"
tag = sys.argv[1] if len(sys.argv) > 1 else ""
torch.manual_seed(0)
H, dh = 12, 64
d = H * dh
bf = torch.bfloat16


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def report(name, us, mb):
    print(f"{tag:10s} {name:58s} {us:8.1f} us   {mb:7.1f} MB algorithmic = {mb / us if us else 0:5.2f} TB/s", flush=True)


def self_case(B, L, label):
    qkv = (torch.randn(B, L, 3 * d, device=dev) * 0.5).to(bf)
    Lp = (L + 7) // 8 * 8
    with torch.no_grad():
        _, Pt = ops.self_attention((torch.randn(B, L, 3 * d, device=dev) * 0.5).to(bf), H, dh, 0.125)
    gO = torch.randn(B, L, d, device=dev).to(bf)
    io = B * L * 3 * d * 2 + B * L * d * 2                      # q, k, v read + o written
    pm = B * H * L * Lp * 2                                      # one pass over a map
    with torch.no_grad():
        report(f"{label} fwd no-grad, map written", timeit(lambda: ops.self_attention(qkv, H, dh, 0.125, want_probs=True)), (io + pm) / 1e6)
        report(f"{label} fwd no-grad, no map", timeit(lambda: ops.self_attention(qkv, H, dh, 0.125, want_probs=False)), io / 1e6)
    for store in (True, False):
        ops.ATTN_STORE_P = store
        nm = "stored-map form" if store else "recomputing form"
        x = qkv.clone().requires_grad_(True)
        report(f"{label} fwd train ({nm}), no map wanted", timeit(lambda: ops.self_attention(x, H, dh, 0.125, want_probs=False)),
               (io + (pm if store else 0)) / 1e6)
        report(f"{label} fwd train ({nm}) + fused KD", timeit(lambda: ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=Pt, kd_weight=1.0)),
               (io + pm + (pm if store else 0)) / 1e6)

        def fb(kd):
            x.grad = None
            if kd:
                O, P, t = ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=Pt, kd_weight=1.0)
                ((O * gO).sum() + t).backward()
            else:
                O, P = ops.self_attention(x, H, dh, 0.125, want_probs=False)
                O.backward(gO)
        # backward alone = (fwd + bwd) - fwd, both timed the same way (autograd overhead included in both)
        for kd in (False, True):
            tf = timeit(lambda: ops.self_attention(x, H, dh, 0.125, want_probs=False, **(dict(kd_teacher=Pt, kd_weight=1.0) if kd else {})))
            tfb = timeit(lambda: fb(kd))
            bio = B * L * 3 * d * 2 * 2 + B * L * d * 2         # q, k, v read, dq, dk, dv written, dO read
            report(f"{label} fwd+bwd ({nm}{', fused KD' if kd else ''}) [bwd ~ {tfb - tf:6.1f} us]", tfb,
                   (io + bio + (2 * pm if store else 0) + (2 * pm if kd else 0)) / 1e6)
    ops.ATTN_STORE_P = False


def cross_case(B, Bkv, Lq, Lk, label):
    q = (torch.randn(B, Lq, d, device=dev) * 0.5).to(bf)
    kv = (torch.randn(Bkv, Lk, 2 * d, device=dev) * 0.5).to(bf)
    idx = (torch.arange(B, device=dev) % Bkv).to(torch.int32)
    gO = torch.randn(B, Lq, d, device=dev).to(bf)
    Lp = (Lk + 7) // 8 * 8
    io = B * Lq * d * 2 * 2 + Bkv * Lk * 2 * d * 2
    pm = B * H * Lq * Lp * 2
    with torch.no_grad():
        report(f"{label} fwd no-grad, map written", timeit(lambda: ops.cross_attention(q, kv, H, dh, 0.125, want_probs=True, kv_index=idx)), (io + pm) / 1e6)
        report(f"{label} fwd no-grad, no map", timeit(lambda: ops.cross_attention(q, kv, H, dh, 0.125, want_probs=False, kv_index=idx)), io / 1e6)
    for store in (True, False):
        ops.ATTN_STORE_P = store
        nm = "stored-map form" if store else "recomputing form"
        qq, kk = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)

        def fb():
            qq.grad = kk.grad = None
            O, _ = ops.cross_attention(qq, kk, H, dh, 0.125, want_probs=False, kv_index=idx)
            O.backward(gO)
        tf = timeit(lambda: ops.cross_attention(qq, kk, H, dh, 0.125, want_probs=False, kv_index=idx))
        tfb = timeit(fb)
        report(f"{label} fwd train ({nm})", tf, (io + (pm if store else 0)) / 1e6)
        report(f"{label} fwd+bwd ({nm}) [bwd ~ {tfb - tf:6.1f} us]", tfb, (2 * io + B * Lq * d * 2 + (4 * pm)) / 1e6)
    ops.ATTN_STORE_P = False


self_case(64, 197, "ViT 64x12x197")
self_case(256, 30, "text 256x12x30")
cross_case(256, 64, 30, 197, "cross 256 rows / 64 images")
