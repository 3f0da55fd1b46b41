#!/usr/bin/env python3
"""Attention forward on the long key sequences of the ITR / VQA steps (577 keys at 384 x 384, 901 at 480 x 480): the
streaming kernel (attn_fwd_stream_kernel: 128-key blocks through LDS, online softmax) against the whole-row kernels
(EVLM_ATTN_NO_STREAM=1), 20 launches replayed from a hipGraph.  ViT self-attention without a map (teacher layers whose map
nobody reads), with the row lse (student), with lse + fused map distillation; cross-attention of 3B text rows on B images."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops
dev = "cuda"; torch.manual_seed(0)
def bench(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); ops.reserve_tables()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    ops.flush_table_uploads(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3
H, dh, d = 12, 64, 768
for B, L in ((64, 577), (32, 901)):
    qkv = (torch.randn(B, L, 3 * d, device=dev) * 0.5).bfloat16()
    with torch.no_grad():
        Pt = ops.self_attention((torch.randn(B, L, 3 * d, device=dev) * 0.5).bfloat16(), H, dh, 0.125)[1]
    fl = 4.0 * B * H * L * L * dh
    out = dict(shape=f"ViT {B}x{H}x{L}", gflop=round(fl / 1e9, 1))
    for tag, env in (("stream", "0"), ("whole_row", "1")):
        os.environ["EVLM_ATTN_NO_STREAM"] = env
        with torch.no_grad():
            t0 = bench(lambda: ops.self_attention(qkv, H, dh, 0.125, want_probs=False))
        x = qkv.clone().requires_grad_(True)
        t1 = bench(lambda: ops.self_attention(x, H, dh, 0.125, want_probs=False))
        t2 = bench(lambda: ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=Pt, kd_weight=1.0))
        gO = torch.randn(B, L, d, device=dev).bfloat16()
        def fb(kd):
            if kd:
                O, _, k = ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=Pt, kd_weight=1.0)
                torch.autograd.grad((O.float() * gO.float()).sum() + k, x)
            else:
                O, _ = ops.self_attention(x, H, dh, 0.125, want_probs=False)
                torch.autograd.grad(O, x, gO)
        tb1, tb2 = bench(lambda: fb(False), 6), bench(lambda: fb(True), 6)
        with torch.no_grad():
            t3 = bench(lambda: ops.self_attention(qkv, H, dh, 0.125, want_probs=True))
        out[tag] = dict(no_grad_us=round(t0, 1), lse_us=round(t1, 1), lse_kd_us=round(t2, 1), map_written_us=round(t3, 1), fwd_bwd_us=round(tb1, 1), fwd_bwd_kd_us=round(tb2, 1),
                        no_grad_tflops=round(fl / t0 / 1e6, 1))
        if tag == "stream":
            # round 5 (ABI 8): the teacher keeps Q, K and its row lse instead of the map (ops.MapRecipe); the student's kernels
            # rebuild P_t.  teacher_recipe_us: the teacher's layer in that form (against map_written_us)
            tq = (torch.randn(B, L, 3 * d, device=dev) * 0.5).bfloat16()
            with torch.no_grad():
                rec = ops.self_attention_recipe(tq, H, dh, 0.125)[1]
                t4 = bench(lambda: ops.self_attention_recipe(tq, H, dh, 0.125))
            t5 = bench(lambda: ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=rec, kd_weight=1.0))
            def fbr():
                O, _, k = ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=rec, kd_weight=1.0)
                torch.autograd.grad((O.float() * gO.float()).sum() + k, x)
            t6 = bench(fbr, 6)
            out[tag].update(teacher_recipe_us=round(t4, 1), lse_kd_recipe_us=round(t5, 1), fwd_bwd_kd_recipe_us=round(t6, 1))
            del rec, tq
    print(json.dumps(out), flush=True)
    del Pt
for Bimg, rows, L in ((64, 3, 577), (32, 4, 901)):
    Bq = Bimg * rows
    q = (torch.randn(Bq, 30, d, device=dev) * 0.5).bfloat16()
    kv = (torch.randn(Bimg, L, 2 * d, device=dev) * 0.5).bfloat16()
    idx = torch.arange(Bimg, device=dev).repeat(rows)
    out = dict(shape=f"cross {Bq} rows x 30 on {Bimg} x {L}")
    for tag, env in (("stream", "0"), ("whole_row", "1")):
        os.environ["EVLM_ATTN_NO_STREAM"] = env
        with torch.no_grad():
            out[tag] = dict(no_map_us=round(bench(lambda: ops.cross_attention(q, kv, H, dh, 0.125, want_probs=False, kv_index=idx)), 1),
                            map_written_us=round(bench(lambda: ops.cross_attention(q, kv, H, dh, 0.125, want_probs=True, kv_index=idx)), 1))
    print(json.dumps(out), flush=True)
