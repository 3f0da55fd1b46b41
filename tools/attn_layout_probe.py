#!/usr/bin/env python3
"""Is the attention kernels' HBM rate bound by the LAYOUT of their operands?  In the packed activations a head's K / V / Q
rows are 128-byte pieces at a stride of the full row (1 536 .. 9 216 bytes); head-major buffers would make every (batch,
head) tile one contiguous range.  The existing kernels take row strides, so head-major operands can be EMULATED without a
new kernel: every (batch, head) becomes a "batch" of a one-head problem (same FLOPs, same bytes, contiguous tiles).
Timed from hipGraph replays (as tools/xattn_bench.py)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops

dev = "cuda"
torch.manual_seed(0)


def bench(f, reps=30):
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


H, dh, scale = 12, 64, 0.125
with torch.no_grad():
    # ---- cross-attention of the GD fusion pass: 256 text rows x 30 tokens on 64 images x 197 tokens ----
    Bimg, rows, Lq, N = 64, 4, 30, 197
    Bq = Bimg * rows
    idx = torch.arange(Bimg, device=dev).repeat(rows)[torch.randperm(Bq, device=dev)].to(torch.int32)
    q = (torch.randn(Bq, Lq, H * dh, device=dev) * 0.5).bfloat16()
    kv = (torch.randn(Bimg, N, 2 * H * dh, device=dev) * 0.5).bfloat16()
    qh = q.view(Bq, Lq, H, dh).permute(0, 2, 1, 3).reshape(Bq * H, Lq, dh).contiguous()
    kvh = torch.cat([kv[..., :H * dh].reshape(Bimg, N, H, dh), kv[..., H * dh:].reshape(Bimg, N, H, dh)], -1)   # [Bimg, N, H, 2 dh]
    kvh = kvh.permute(0, 2, 1, 3).reshape(Bimg * H, N, 2 * dh).contiguous()
    idxh = (idx.long()[:, None] * H + torch.arange(H, device=dev)[None, :]).reshape(-1).to(torch.int32)
    for env in ("1", "0"):
        os.environ["EVLM_ATTN_GROUP_PERSIST"] = env
        t_a = bench(lambda: ops.cross_attention(q, kv, H, dh, scale, want_probs=False, kv_index=idx))
        t_b = bench(lambda: ops.cross_attention(qh, kvh, 1, dh, scale, want_probs=False, kv_index=idxh))
        print(json.dumps({"case": "cross 256x30 on 64x197", "persistent": env == "1", "packed_us": round(t_a, 1),
                          "head_major_emulated_us": round(t_b, 1)}), flush=True)
    O1, _ = ops.cross_attention(q, kv, H, dh, scale, want_probs=False, kv_index=idx)
    O2, _ = ops.cross_attention(qh, kvh, 1, dh, scale, want_probs=False, kv_index=idxh)
    assert torch.equal(O1.view(Bq, Lq, H, dh).permute(0, 2, 1, 3).reshape(Bq * H, Lq, dh), O2)
    # ---- ViT self-attention: 64 x 197 tokens ----
    B, L = 64, 197
    qkv = (torch.randn(B, L, 3 * H * dh, device=dev) * 0.5).bfloat16()
    qkvh = qkv.view(B, L, 3, H, dh).permute(0, 3, 1, 2, 4).reshape(B * H, L, 3 * dh).contiguous()
    for want in (False, True):
        t_a = bench(lambda: ops.self_attention(qkv, H, dh, scale, want_probs=want))
        t_b = bench(lambda: ops.self_attention(qkvh, 1, dh, scale, want_probs=want))
        print(json.dumps({"case": "ViT self-attention 64x197", "map": want, "packed_us": round(t_a, 1),
                          "head_major_emulated_us": round(t_b, 1)}), flush=True)
# forward + backward (recomputing single-pass kernel)
gO = (torch.randn(B, L, H * dh, device=dev) * 0.5).bfloat16()
gOh = gO.view(B, L, H, dh).permute(0, 2, 1, 3).reshape(B * H, L, dh).contiguous()


def fb(x, g, h):
    x.grad = None
    O = ops.self_attention(x, h, dh, scale, want_probs=False)[0]
    O.backward(g)


xa, xb = qkv.clone().requires_grad_(True), qkvh.clone().requires_grad_(True)
t_a, t_b = bench(lambda: fb(xa, gO, H), reps=10), bench(lambda: fb(xb, gOh, 1), reps=10)
print(json.dumps({"case": "ViT self-attention fwd+bwd", "packed_us": round(t_a, 1), "head_major_emulated_us": round(t_b, 1)}), flush=True)
