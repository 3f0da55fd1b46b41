#!/usr/bin/env python3
"""Cross-attention forward of one fusion layer at the GD-step shapes: the fused kernel (K/V projection + QK^T + softmax +
PV in one launch, evlm_xattn_fused_fwd) against the two-launch composite on the training path (packed K/V GEMM on the
256x256 kernel + MFMA attention sharing K/V through kv_index).  Reports time, executed FLOPs and the fraction of the
bf16 MFMA peak for both ("K/V-proj + QK^T + softmax + PV (+probs)"), with and without the map output."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L

dev, PEAK = "cuda", 2500.0
torch.manual_seed(0)

def bench(f, reps=30):
    """per-call time of `f` replayed from a hipGraph of `reps` calls (as on the training path: no host launch cost in the
    number - two eager Python-level ops per call are host-bound at these sizes)"""
    for _ in range(3): f()
    torch.cuda.synchronize()
    ops.reserve_tables()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    ops.flush_table_uploads()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3

def case(Bimg, rows_per_img, Lq=30, N=197, H=12, probs=False):
    d, dh = H * 64, 64
    Bq = Bimg * rows_per_img
    x = (torch.randn(Bimg, N, d, device=dev) * 0.5).bfloat16()
    q = (torch.randn(Bq, Lq, d, device=dev) * 0.5).bfloat16()
    Wk, Wv = [torch.nn.Parameter(torch.randn(d, d, device=dev) * 0.03, requires_grad=False) for _ in range(2)]
    bk, bv = [torch.nn.Parameter(torch.randn(d, device=dev) * 0.1, requires_grad=False) for _ in range(2)]
    idx = torch.arange(Bimg, device=dev).repeat(rows_per_img) if rows_per_img > 1 else None
    if idx is not None:
        idx = idx[torch.randperm(Bq, device=dev)]                  # hard negatives: arbitrary row -> image map
    scale = 0.125
    def composite():
        kv = ops.linear_packed(x, (Wk, Wv), (bk, bv))
        return ops.cross_attention(q, kv, H, dh, scale, want_probs=probs, kv_index=idx)
    def fused():
        return ops.cross_attention_fused(q, x, (Wk, Wv), (bk, bv), H, dh, scale, want_probs=probs, kv_index=idx)
    with torch.no_grad():
        (o1, p1), (o2, p2) = composite(), fused()
        err = float((o1.float() - o2.float()).abs().max() / o1.float().abs().max())
        perr = float((p1.float() - p2.float()).abs().max()) if probs else 0.0
        tc, tf = bench(composite), bench(fused)
        kvonly = bench(lambda: ops.linear_packed(x, (Wk, Wv), (bk, bv)))
        kv0 = ops.linear_packed(x, (Wk, Wv), (bk, bv))
        attn_only = bench(lambda: ops.cross_attention(q, kv0, H, dh, scale, want_probs=probs, kv_index=idx))
        # the training path (round 4): ONE K/V projection for the n fusion layers of the encoder (n = 3 student, 6 teacher),
        # each layer's attention reading its columns of the merged buffer - per layer: merged GEMM / n + attention
        mer = {}
        for n in (3, 6):
            Ws = [torch.nn.Parameter(torch.randn(d, d, device=dev) * 0.03, requires_grad=False) for _ in range(2 * n)]
            bs = [torch.nn.Parameter(torch.randn(d, device=dev) * 0.1, requires_grad=False) for _ in range(2 * n)]
            kvm, _ = ops.merged_kv(x, Ws, bs, n)
            tg = bench(lambda: ops.merged_kv(x, Ws, bs, n), reps=10)
            ta = bench(lambda: ops.cross_attention(q, kvm, H, dh, scale, want_probs=probs, kv_index=idx, kv_col=2 * d))
            def layers():
                kv_, _ = ops.merged_kv(x, Ws, bs, n)
                for i in range(n):
                    ops.cross_attention(q, kv_, H, dh, scale, want_probs=probs, kv_index=idx, kv_col=2 * d * i)
            tl = bench(layers, reps=6) / n
            mer[n] = (tg / n, ta, tl)
            del Ws, bs, kvm
    fl = 2.0 * Bimg * N * d * 2 * d + 4.0 * Bq * H * Lq * N * dh
    rec = dict(Bimg=Bimg, Bq=Bq, Lq=Lq, N=N, probs=probs, gflop=round(fl / 1e9, 2), composite_us=round(tc, 1),
               composite_kv_gemm_us=round(kvonly, 1), composite_attention_us=round(attn_only, 1), fused_us=round(tf, 1),
               composite_tflops=round(fl / tc / 1e6, 1), fused_tflops=round(fl / tf / 1e6, 1),
               composite_mfma_frac=round(fl / tc / 1e6 / PEAK, 4), fused_mfma_frac=round(fl / tf / 1e6 / PEAK, 4),
               max_rel_err_O=err, max_abs_err_P=perr)
    for n, (tg, ta, tl) in mer.items():
        rec[f"merged{n}_kv_gemm_us_per_layer"] = round(tg, 1)
        rec[f"merged{n}_attention_us"] = round(ta, 1)
        rec[f"merged{n}_composite_us_per_layer"] = round(tl, 1)
        rec[f"merged{n}_mfma_frac"] = round(fl / tl / 1e6 / PEAK, 4)
    print(json.dumps(rec), flush=True)

case(64, 4)                  # the batched fusion pass of a GD step: [pos ; neg ; neg ; mlm] text rows over 64 images
case(64, 4, probs=True)
case(64, 1)                  # one text row per image (retrieval scoring / a plain forward)
case(64, 3, probs=True)      # ITM pass of the ITR fine-tune at 224
case(256, 1)
