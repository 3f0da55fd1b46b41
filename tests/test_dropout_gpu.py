"""Dropout with p > 0 (reference: efficient_models/eff_bert.py:180,214,242,346,372-379,456-460; the stock BERT config
trains with 0.1).  The reference's CUDA RNG stream cannot be reproduced, so parity is established by feeding the SAME
keep-masks to both sides: the HIP kernels regenerate their masks from (device {seed, step} word, call id, element index);
ops.dropout_mask materialises exactly that function for the CPU oracle (oracle.xvlm_oracle.DROPOUT_MASKS)."""
import math

import pytest
import torch

from helpers import close, load_det_weights, model_config
from oracle import schema, synth
from oracle import xvlm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hidden_dropout_kernel_is_its_own_backward_and_never_stores_a_mask(dtype):
    from efficientvlm_amd import ops
    ops.dropout_seed(1234)
    g = torch.Generator().manual_seed(1)
    for shape, p in (((7, 30, 768), 0.1), ((5, 13), 0.5), ((3, 8, 64), 0.25)):
        x = torch.randn(shape, generator=g).to(DEV, dtype).requires_grad_(True)
        r = torch.randn(shape, generator=g).to(DEV, dtype).requires_grad_(True)
        ops.DROPOUT_LOG = []
        y = ops.dropout(x, p, True, residual=r)
        (call, kind, shp, pp), = ops.DROPOUT_LOG
        ops.DROPOUT_LOG = None
        m = ops.dropout_mask(call, shape, p)
        assert set(torch.unique(m).tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
        keep = float((m > 0).float().mean())
        n = m.numel()
        assert abs(keep - (1 - p)) < 5 * math.sqrt(p * (1 - p) / n) + 1e-3, (keep, p)
        ref = (x.detach().float() * m + r.detach().float()).to(dtype)
        assert torch.equal(y.detach(), ref)
        gy = torch.randn(shape, generator=g).to(DEV, dtype)
        y.backward(gy)
        assert torch.equal(x.grad, (gy.float() * m).to(dtype)) and torch.equal(r.grad, gy)
        # another site (call id) and another step draw other masks; the same triple reproduces the mask
        y2 = ops.dropout(x.detach(), p, True)
        assert not torch.equal((y2 != 0), (m != 0) & (x.detach() != 0))
        assert torch.equal(ops.dropout_mask(call, shape, p), m)
        ops.dropout_tick()
        assert not torch.equal(ops.dropout_mask(call, shape, p), m)
    # identity outside training / at p = 0
    x = torch.randn(4, 8, device=DEV)
    assert ops.dropout(x, 0.1, False) is x and ops.dropout(x, 0.0, True) is x


@pytest.mark.parametrize("dtype,cross", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, False),
                                         (torch.bfloat16, True)])
def test_attention_probability_dropout_against_torch_with_the_same_mask(dtype, cross):
    """O = ((P .* M) V) * gate with the map P returned un-dropped; dQ / dK / dV (and the map's external gradient) through
    the regenerated mask"""
    from efficientvlm_amd import ops
    ops.dropout_seed(99)
    g = torch.Generator().manual_seed(3)
    B, H, dh, Lq, Lk, p = 3, 4, 16, 9, (21 if cross else 9), 0.2
    d = H * dh
    tol = 2e-5 if dtype == torch.float32 else 2.5e-2
    mask = torch.zeros(B, Lk)
    mask[1, Lk - 3:] = -10000.0
    gate = torch.rand(H, generator=g) + 0.5
    if cross:
        q = (torch.randn(B, Lq, d, generator=g) * 0.5).to(DEV, dtype).requires_grad_(True)
        kv = (torch.randn(B, Lk, 2 * d, generator=g) * 0.5).to(DEV, dtype).requires_grad_(True)
        ops.DROPOUT_LOG = []
        Oo, P = ops.cross_attention(q, kv, H, dh, 1.0 / math.sqrt(dh), mask=mask.to(DEV), gate=gate.to(DEV), dropout_p=p)
        qr, kr, vr = q.detach().float(), kv.detach().float()[..., :d], kv.detach().float()[..., d:]
    else:
        qkv = (torch.randn(B, Lq, 3 * d, generator=g) * 0.5).to(DEV, dtype).requires_grad_(True)
        ops.DROPOUT_LOG = []
        Oo, P = ops.self_attention(qkv, H, dh, 1.0 / math.sqrt(dh), mask=mask.to(DEV), gate=gate.to(DEV), dropout_p=p)
        x = qkv.detach().float()
        qr, kr, vr = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    (call, kind, shp, pp), = ops.DROPOUT_LOG
    ops.DROPOUT_LOG = None
    assert kind == "attention_probs" and shp == (B, H, Lq, Lk)
    M = ops.dropout_mask(call, (B, H, Lq, Lk), p)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qr, kr, vr))
    sp = lambda t, Ln: t.reshape(B, Ln, H, dh).transpose(1, 2)
    S = sp(qr, Lq) @ sp(kr, Lk).transpose(-1, -2) / math.sqrt(dh) + mask.to(DEV)[:, None, None, :]
    Pr = torch.softmax(S, -1)
    Or = ((Pr * M) @ sp(vr, Lk) * gate.to(DEV)[None, :, None, None]).transpose(1, 2).reshape(B, Lq, d)
    assert rel_err(P.float(), Pr) < tol and rel_err(Oo.float(), Or) < 2 * tol
    gO = torch.randn(B, Lq, d, generator=g).to(DEV, dtype)
    gP = (torch.randn(B, H, Lq, Lk, generator=g) * 0.1).to(DEV, dtype)
    ((Oo * gO).sum() + (P * gP).sum()).backward()
    ((Or * gO.float()).sum() + (Pr * gP.float()).sum()).backward()
    if cross:
        assert rel_err(q.grad.float(), qr.grad) < 4 * tol
        assert rel_err(kv.grad.float(), torch.cat([kr.grad, vr.grad], -1)) < 4 * tol
    else:
        assert rel_err(qkv.grad.float(), torch.cat([qr.grad, kr.grad, vr.grad], -1)) < 4 * tol


MFMA_CASES = [
    # name, B, Bkv, H, Lq, Lk, self-attention?, want the map?, causal?
    ("text_self_30", 5, 5, 3, 30, 30, True, False, False),
    ("text_self_30_map", 5, 5, 3, 30, 30, True, True, False),
    ("decoder_self_causal_12", 4, 4, 2, 12, 12, True, False, True),
    ("text_self_50", 3, 3, 2, 50, 50, True, True, False),
    ("cross_grouped_197", 8, 2, 3, 30, 197, False, False, False),
    ("cross_grouped_197_map", 8, 3, 2, 30, 197, False, True, False),
    ("cross_per_batch_197", 3, 3, 2, 30, 197, False, True, False),
    ("cross_stream_577", 6, 2, 2, 30, 577, False, False, False),
    ("cross_stream_577_map", 4, 2, 2, 30, 577, False, True, False),
    ("cross_stream_901", 3, 3, 1, 20, 901, False, False, False),
]


@pytest.mark.parametrize("case", MFMA_CASES, ids=[c[0] for c in MFMA_CASES])
def test_probability_dropout_inside_the_mfma_attention_kernels(case):
    """Round 6: the bf16 MFMA kernels (head dim 64) regenerate the keep-mask themselves - whole-row, grouped (shared K/V
    index), streaming and map-writing forwards; single-pass, kernels A + B and streaming backwards, with the recomputing
    (lse) form kept.  Against fp32 torch on the same bf16 operands with the SAME mask (ops.dropout_mask): the returned map is
    the un-dropped softmax (eff_bert.py:338-361), the context and every gradient go through P .* M, the gate gradient
    included; an external gradient on the map (the unfused distillation terms) rides along where the map is returned."""
    from efficientvlm_amd import ops
    name, B, Bkv, H, Lq, Lk, self_attn, want, causal = case
    ops.dropout_seed(4321)
    g = torch.Generator().manual_seed(11)
    dh, p = 64, 0.1
    d = H * dh
    scale = 1.0 / math.sqrt(dh)
    mask = torch.zeros(B, Lk)
    mask[1, Lk - 5:] = -10000.0
    gate = (torch.rand(H, generator=g) + 0.5).to(DEV).requires_grad_(True)
    idx = None
    if self_attn:
        qkv = (torch.randn(B, Lq, 3 * d, generator=g) * 0.7).to(DEV, torch.bfloat16).requires_grad_(True)
        ops.DROPOUT_LOG = []
        Oo, P = ops.self_attention(qkv, H, dh, scale, mask=mask.to(DEV), gate=gate, want_probs=want, causal=causal, dropout_p=p)
        x = qkv.detach().float()
        qr, kr, vr = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    else:
        q = (torch.randn(B, Lq, d, generator=g) * 0.7).to(DEV, torch.bfloat16).requires_grad_(True)
        kv = (torch.randn(Bkv, Lk, 2 * d, generator=g) * 0.7).to(DEV, torch.bfloat16).requires_grad_(True)
        if Bkv != B:
            idx = (torch.arange(B) % Bkv).to(DEV)
        ops.DROPOUT_LOG = []
        Oo, P = ops.cross_attention(q, kv, H, dh, scale, mask=mask.to(DEV), gate=gate, want_probs=want, kv_index=idx, dropout_p=p)
        qr = q.detach().float()
        kvf = kv.detach().float()
        kr, vr = kvf[..., :d], kvf[..., d:]
    (call, kind, shp, pp), = ops.DROPOUT_LOG
    ops.DROPOUT_LOG = None
    assert kind == "attention_probs" and shp == (B, H, Lq, Lk)
    M = ops.dropout_mask(call, (B, H, Lq, Lk), p)
    keep = float((M > 0).float().mean())
    assert abs(keep - (1 - p)) < 5 * math.sqrt(p * (1 - p) / M.numel()) + 1e-3, keep
    assert set(torch.unique(M).tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qr, kr, vr))
    gr = gate.detach().clone().requires_grad_(True)
    sp = lambda t, Ln: t.reshape(t.shape[0], Ln, H, dh).transpose(1, 2)
    K4, V4 = sp(kr, Lk), sp(vr, Lk)
    if idx is not None:
        K4, V4 = K4[idx], V4[idx]
    S = sp(qr, Lq) @ K4.transpose(-1, -2) * scale + mask.to(DEV)[:, None, None, :]
    if causal:
        S = S + torch.triu(torch.full((Lq, Lk), -10000.0, device=DEV), 1)
    Pr = torch.softmax(S, -1)
    Or = ((Pr * M) @ V4 * gr[None, :, None, None]).transpose(1, 2).reshape(B, Lq, d)
    tol = 2.5e-2
    assert rel_err(Oo.float(), Or) < tol, rel_err(Oo.float(), Or)
    gO = torch.randn(B, Lq, d, generator=g).to(DEV, torch.bfloat16)
    loss, lref = (Oo.float() * gO.float()).sum(), (Or * gO.float()).sum()
    if want:
        assert rel_err(P.float(), Pr) < tol
        gP = (torch.randn(B, H, Lq, Lk, generator=g) * 0.5).to(DEV, torch.bfloat16)
        loss, lref = loss + (P.float() * gP.float()).sum(), lref + (Pr * gP.float()).sum()
    else:
        assert P is None
    loss.backward()
    lref.backward()
    if self_attn:
        got, ref = qkv.grad.float(), torch.cat([qr.grad, kr.grad, vr.grad], -1)
        for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
            assert rel_err(got[..., sl], ref[..., sl]) < 2 * tol, (nm, rel_err(got[..., sl], ref[..., sl]))
    else:
        assert rel_err(q.grad.float(), qr.grad) < 2 * tol, rel_err(q.grad.float(), qr.grad)
        ref = torch.cat([kr.grad, vr.grad], -1)
        assert rel_err(kv.grad.float()[..., :d], ref[..., :d]) < 2 * tol, rel_err(kv.grad.float()[..., :d], ref[..., :d])
        assert rel_err(kv.grad.float()[..., d:], ref[..., d:]) < 2 * tol, rel_err(kv.grad.float()[..., d:], ref[..., d:])
    assert rel_err(gate.grad, gr.grad) < 2 * tol, rel_err(gate.grad, gr.grad)
    # dropout is really on: the p = 0 context differs
    with torch.no_grad():
        O0 = ((Pr @ V4) * gr[None, :, None, None]).transpose(1, 2).reshape(B, Lq, d)
    assert rel_err(Oo.float(), O0) > 0.05


def test_grouped_cross_attention_with_dropout_is_bit_identical_to_the_per_batch_kernel(monkeypatch):
    """the (K/V row, head)-grouped forward and the per-batch kernel regenerate the same mask for the same (batch, head,
    query, key) - whichever workgroup serves a text row - so their contexts, maps and row lse are bit-identical with p > 0"""
    from efficientvlm_amd import ops
    g = torch.Generator().manual_seed(5)
    B, Bkv, H, Lq, Lk, dh, p = 12, 3, 4, 30, 197, 64, 0.1
    d = H * dh
    q = (torch.randn(B, Lq, d, generator=g) * 0.7).to(DEV, torch.bfloat16)
    kv = (torch.randn(Bkv, Lk, 2 * d, generator=g) * 0.7).to(DEV, torch.bfloat16)
    idx = torch.tensor([0, 1, 2, 1, 0, 2, 2, 1, 0, 0, 1, 2]).to(DEV)
    mask = torch.zeros(B, Lk)
    mask[3, 150:] = -10000.0
    outs = []
    for no_group in ("0", "1"):
        monkeypatch.setenv("EVLM_ATTN_NO_GROUP", no_group)
        ops.dropout_seed(77)
        with torch.no_grad():
            outs.append(ops.cross_attention(q, kv, H, dh, 0.125, mask=mask.to(DEV), want_probs=True, kv_index=idx, dropout_p=p))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ops.dropout_seed(78)
    with torch.no_grad():
        other = ops.cross_attention(q, kv, H, dh, 0.125, mask=mask.to(DEV), want_probs=True, kv_index=idx, dropout_p=p)
    assert not torch.equal(other[0], outs[0][0]) and torch.equal(other[1], outs[0][1])      # another seed: another context, the same map


@pytest.mark.parametrize("dtype,M,K,N", [(torch.float32, 20, 64, 64), (torch.bfloat16, 3840, 768, 768), (torch.bfloat16, 7680, 768, 768),
                                         (torch.bfloat16, 7680, 3072, 768), (torch.bfloat16, 1920, 3072, 768), (torch.bfloat16, 90, 256, 128)])
def test_hidden_dropout_in_the_gemm_epilogue_and_the_layernorm_backward_equals_the_separate_passes(dtype, M, K, N):
    """Round 6 (ABI 9): LayerNorm(dropout(x W^T + b) + residual) with the keep-mask applied in the GEMM's residual epilogue
    (evlm_gemm_args.dropout_p; every kernel family the text-side products reach) and its backward's masked gradient written by
    the LayerNorm backward itself (evlm_layernorm_bwd_drop) - bit-identical, forward and every gradient, to the round-5 chain
    linear -> evlm_dropout(+ residual) -> LayerNorm with evlm_dropout on the gradient, and equal to fp32 torch given the mask"""
    from efficientvlm_amd import ops
    from efficientvlm_amd._lib import load
    g = torch.Generator().manual_seed(M + K)
    p = 0.1
    x0 = (torch.randn(M, K, generator=g) * 0.5).to(DEV, dtype)
    r0 = torch.randn(M, N, generator=g).to(DEV, dtype)
    W0 = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    b0 = (torch.randn(N, generator=g) * 0.1).to(DEV)
    gam0, bet0 = (torch.rand(N, generator=g) + 0.5).to(DEV), (torch.randn(N, generator=g) * 0.1).to(DEV)
    gy = torch.randn(M, N, generator=g).to(DEV, dtype)
    res = {}
    for fused in (True, False):
        ops.dropout_seed(31)
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        W, b, gam, bet = (t.clone().requires_grad_(True) for t in (W0, b0, gam0, bet0))
        ops.DROPOUT_LOG = []
        if fused:
            h = ops.linear(x, W, b, residual=r, dropout_p=p)
            kern = load().evlm_gemm_last_kernel().decode()
        else:
            h = ops.dropout(ops.linear(x, W, b), p, True, residual=r)
        (call, kind, shp, pp), = ops.DROPOUT_LOG
        ops.DROPOUT_LOG = None
        assert kind == "hidden" and tuple(shp) == (M, N)
        y = ops.layer_norm(h, gam, bet, 1e-12)
        y.backward(gy)
        res[fused] = [t.detach().clone() for t in (h, y, x.grad, r.grad, W.grad, b.grad, gam.grad, bet.grad)]
        mask = ops.dropout_mask(call, (M, N), p)
    names = ("h", "y", "dx", "dr", "dW", "db", "dgamma", "dbeta")
    for nm, a, c in zip(names, res[True], res[False]):
        if nm in ("dW", "db", "dgamma", "dbeta"):        # (split reductions / f32 atomics: the same operands in another order)
            assert rel_err(a.float(), c.float()) < 1e-5, (nm, kern, rel_err(a.float(), c.float()))
        else:
            assert torch.equal(a, c), (nm, kern, rel_err(a.float(), c.float()))
    # ... and against fp32 torch with the same mask
    xf, rf = x0.float().requires_grad_(True), r0.float().requires_grad_(True)
    Wf = (W0 if dtype == torch.float32 else W0.bfloat16().float()).clone().requires_grad_(True)
    hf = (xf @ Wf.t() + b0) * mask + rf
    yf = torch.nn.functional.layer_norm(hf, (N,), gam0, bet0, 1e-12)
    yf.backward(gy.float())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(res[True][0].float(), hf.detach()) < tol and rel_err(res[True][1].float(), yf.detach()) < tol
    assert rel_err(res[True][2].float(), xf.grad) < 2 * tol and rel_err(res[True][3].float(), rf.grad) < 2 * tol
    assert rel_err(res[True][4].float(), Wf.grad) < 2 * tol


@pytest.mark.parametrize("dtype,with_gate", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_ffn_block_with_hidden_dropout_in_the_second_products_epilogue(dtype, with_gate):
    """BertOutput inside ops.mlp: LayerNorm(dropout(gelu(x W1^T + b1) [.* z] W2^T + b2) + x) - fused (mask in the second
    product's epilogue, masked gradient from the LayerNorm backward) against the separate-pass chain, bit for bit"""
    from efficientvlm_amd import ops
    from efficientvlm_amd._lib import ACT_GELU, GATE_POST
    g = torch.Generator().manual_seed(3)
    M, d, Fh, p = (24, 64, 128, 0.25) if dtype == torch.float32 else (3840, 768, 3072, 0.1)
    x0 = (torch.randn(4, M // 4, d, generator=g) * 0.5).to(DEV, dtype)
    w1, w2 = (torch.randn(Fh, d, generator=g) * 0.05).to(DEV), (torch.randn(d, Fh, generator=g) * 0.05).to(DEV)
    b1, b2 = (torch.randn(Fh, generator=g) * 0.1).to(DEV), (torch.randn(d, generator=g) * 0.1).to(DEV)
    z0 = torch.rand(Fh, generator=g).to(DEV) if with_gate else None
    gam0, bet0 = (torch.rand(d, generator=g) + 0.5).to(DEV), (torch.randn(d, generator=g) * 0.1).to(DEV)
    gy = torch.randn(4, M // 4, d, generator=g).to(DEV, dtype)
    res = {}
    for fused in (True, False):
        ops.dropout_seed(5)
        x = x0.clone().requires_grad_(True)
        ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2, gam0, bet0)]
        z = z0.clone().requires_grad_(True) if with_gate else None
        if fused:
            h = ops.mlp(x, ps[0], ps[1], ps[2], ps[3], ACT_GELU, gate=z, gate_pos=GATE_POST, residual=x, dropout_p=p)
        else:
            h = ops.dropout(ops.mlp(x, ps[0], ps[1], ps[2], ps[3], ACT_GELU, gate=z, gate_pos=GATE_POST), p, True, residual=x)
        y = ops.layer_norm(h, ps[4], ps[5], 1e-12)
        y.backward(gy)
        res[fused] = [y.detach().clone(), x.grad.clone()] + [t.grad.clone() for t in ps] + ([z.grad.clone()] if with_gate else [])
    for i, (a, c) in enumerate(zip(res[True], res[False])):
        if i == 1 and dtype == torch.bfloat16:
            # (dx: the fused form adds the residual's gradient in the dX product's epilogue, the chain lets autograd add two
            # bf16 tensors - one rounding apart)
            assert rel_err(a.float(), c.float()) < 1e-2, rel_err(a.float(), c.float())
        elif i >= 1:                                     # (gradient sums: split reductions / f32 atomics, order-dependent)
            assert rel_err(a.float(), c.float()) < 1e-5, (i, rel_err(a.float(), c.float()))
        else:
            assert torch.equal(a, c), (i, rel_err(a.float(), c.float()))


def _models(p, seed=0):
    from efficientvlm_amd.models.model_pretrain import XVLM
    geom = synth.GEOMS["tiny"]
    s_cfg = O.model_cfg(geom, "s")
    student = XVLM(model_config(geom, "s", dropout=p))
    s_sd = load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"]), 400 + seed, geom["std"])
    return geom, s_cfg, student.to(DEV), s_sd


def _tie(sd):
    return {**sd, "text_encoder.cls.predictions.decoder.weight": sd["text_encoder.bert.embeddings.word_embeddings.weight"],
            "text_encoder.cls.predictions.decoder.bias": sd["text_encoder.cls.predictions.bias"]}


def test_pretrain_forward_backward_with_dropout_matches_the_oracle_given_the_same_masks():
    """X-VLM pre-training forward (text pass, ITM positive + hard-negative fusion passes, MLM pass) with the stock BERT
    dropout 0.1 in train mode, pass by pass as the reference issues it, fp32: every loss, the MLM logits and the parameter
    gradients against the oracle that multiplies by the SAME masks at the SAME sites (eff_bert.py:214,346,379,460)."""
    from efficientvlm_amd import ops
    from efficientvlm_amd.runtime import compute
    geom, s_cfg, student, s_sd = _models(0.1)
    student.train()
    student.batched_passes = False                 # the reference's pass order = the oracle's dropout-site order
    student.batched_itm = False                    # (positive and hard-negative fusion passes as two passes, xvlm.py:460-476)
    ops.dropout_seed(7)
    batch = synth.make_batch(geom, 4, seed=5, ragged=True)
    neg = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
    student.injected_neg_idx = neg
    gb = {k: v.to(DEV) for k, v in batch.items()}
    ops.DROPOUT_LOG = []
    with compute(torch.float32):
        S = student(gb["image"], gb["text_ids"], gb["text_atts"], text_ids_masked=gb["text_ids_masked"],
                    masked_pos=gb["masked_pos"], masked_ids=gb["masked_ids"], output_attentions=True, output_hidden_states=True)
        total = S["loss"]["loss_itc"] + S["loss"]["loss_itm"] + S["loss"]["loss_mlm"]
        total.backward()
    log, ops.DROPOUT_LOG = ops.DROPOUT_LOG, None
    n_layers_text, n_fusion = s_cfg["fusion_layer"], s_cfg["text_layers"] - s_cfg["fusion_layer"]
    # text pass: 1 + 3 per text layer; each fusion pass: 5 per fusion layer; MLM pass: 1 + 3 per text layer + 5 per fusion layer
    want = (1 + 3 * n_layers_text) + 2 * 5 * n_fusion + (1 + 3 * n_layers_text + 5 * n_fusion)
    assert len(log) == want, (len(log), want)
    masks = [ops.dropout_mask(c, shp, p).cpu() for c, kind, shp, p in log]
    leaves = {k: v.clone().requires_grad_(True) for k, v in s_sd.items()}
    O.DROPOUT_MASKS = iter(masks)
    try:
        oS = O.pretrain_forward(_tie(leaves), s_cfg, batch, neg)
        assert next(O.DROPOUT_MASKS, None) is None, "the oracle visited fewer dropout sites than the HIP path"
    finally:
        O.DROPOUT_MASKS = None
    ototal = oS["loss"]["loss_itc"] + oS["loss"]["loss_itm"] + oS["loss"]["loss_mlm"]
    ototal.backward()
    for k in S["loss"]:
        close(S["loss"][k], oS["loss"][k], 1e-4, 0, k)
    close(S["logits_dict"]["mlm_logits"].float(), oS["logits_dict"]["mlm_logits"], 1e-4, 1e-5, "mlm_logits")
    for k, tup in S["attention_dict"].items():
        for i, t in enumerate(tup):
            close(t.float(), oS["attention_dict"][k][i], 1e-4, 1e-6, f"{k}.{i}")
    n = 0
    for name, p in student.named_parameters():
        if p.grad is None or leaves[name].grad is None:
            continue
        ref = leaves[name].grad
        close(p.grad, ref, 0, 1e-3 * float(ref.norm()) + 2e-6, f"grad {name}")
        n += 1
    assert n > 100
    # and dropout really is active: the p = 0 losses differ
    with torch.no_grad():
        o0 = O.pretrain_forward(_tie(s_sd), s_cfg, batch, neg)
    assert abs(float(o0["loss"]["loss_mlm"]) - float(oS["loss"]["loss_mlm"])) > 1e-4


def test_eval_mode_and_frozen_teachers_build_from_the_stock_bert_config_and_ignore_dropout():
    """a model built from a config with dropout 0.1 (runtime.BertConfig's defaults are the stock ones) runs in eval mode
    exactly as the p = 0 model: dropout is an identity there (and nothing raises at construction any more)"""
    from efficientvlm_amd.runtime import compute
    geom, s_cfg, m1, _ = _models(0.1, seed=1)
    _, _, m0, _ = _models(0.0, seed=1)
    m1.eval(); m0.eval()
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, 3, seed=9).items()}
    neg = torch.tensor([1, 2, 0, 2, 0, 1])
    m1.injected_neg_idx = m0.injected_neg_idx = neg
    kw = dict(text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"],
              output_attentions=True, output_hidden_states=True)
    with torch.no_grad(), compute(torch.float32):
        a = m1(batch["image"], batch["text_ids"], batch["text_atts"], **kw)
        b = m0(batch["image"], batch["text_ids"], batch["text_atts"], **kw)
    for k in a["loss"]:
        assert torch.equal(a["loss"][k], b["loss"][k]), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_gd_trainer_with_stock_dropout_draws_new_masks_every_step(use_graph):
    """GDTrainer (bf16) with the stock BERT dropout 0.1 on the student: steps run, losses are finite, and with the
    learning rate at 0 two steps on the SAME batch differ (new masks per step - also across hipGraph replays, through the
    device-side step word) while the p = 0 student repeats its losses exactly"""
    from efficientvlm_amd.models.model_pretrain import XVLM
    from efficientvlm_amd.trainer import GDTrainer
    geom = synth.GEOMS["tiny"]
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, 4, seed=2).items()}
    outs = {}
    for p in (0.1, 0.0):
        torch.manual_seed(0)
        student = XVLM(model_config(geom, "s", dropout=p)).to(DEV)
        teacher = XVLM(model_config(geom, "t", dropout=0.1)).to(DEV)      # frozen / eval: dropout is an identity
        neg = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
        student.injected_neg_idx = teacher.injected_neg_idx = neg
        tr = GDTrainer(student, teacher, lr=0.0, dtype=torch.bfloat16, use_graph=use_graph)
        outs[p] = torch.stack([tr.step(batch).clone() for _ in range(3)]).cpu()
        assert torch.isfinite(outs[p]).all()
    # column 3 = the MLM loss (no atomically accumulated sums in it: bit-stable when nothing changes)
    assert float((outs[0.0][1, 3] - outs[0.0][2, 3]).abs()) <= 1e-6 * float(outs[0.0][1, 3].abs())
    assert float((outs[0.1][1, 3] - outs[0.1][2, 3]).abs()) > 1e-4 * float(outs[0.1][1, 3].abs())
    assert float((outs[0.1][0, 3] - outs[0.1][1, 3]).abs()) > 1e-4 * float(outs[0.1][1, 3].abs())
    assert abs(float(outs[0.1][1, 0]) - float(outs[0.0][1, 0])) < 0.2 * abs(float(outs[0.0][1, 0]))


def test_benchmarked_configuration_with_stock_dropout_matches_the_oracle_given_the_same_masks():
    """Round 6: the configuration bench.py times (bf16, batch 64, full geometry, joint hipGraph replay, teacher pipelined,
    batched text / fusion passes, merged K/V projection, grouped MFMA cross-attention) under the reference's stock
    training-mode dropout - student BERT hidden_dropout_prob = attention_probs_dropout_prob = 0.1 (eff_bert.py:180,214,346,
    372-381,456-462) - against the fp32 CPU oracle multiplying by the SAME masks: the masks of the REPLAYED step are
    regenerated from (device seed, the step word that replay drew, the call ids baked into the graph) and cut, per site, into
    the row ranges of the reference's four passes (text | fusion positive | fusion negatives | MLM).  Losses within 1e-3,
    every distillation term within 1e-2 / 5e-2, the whole gradient at cosine > 0.9999."""
    from efficientvlm_amd import ops
    from efficientvlm_amd.models.model_pretrain import XVLM
    from efficientvlm_amd.trainer import GDTrainer
    B, p = 64, 0.1
    geom = synth.GEOMS["full"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = XVLM(model_config(geom, "s", dropout=p)), XVLM(model_config(geom, "t"))
    load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"]), 1000 + 21, geom["std"])
    load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"]), 2000 + 21, geom["std"])
    student.to(DEV).train()
    teacher.to(DEV).eval()
    for q in teacher.parameters():
        q.requires_grad_(False)
    s_sd = schema.det_weights(schema.xvlm_schema(s_cfg, geom["max_pos"]), 1000 + 21, geom["std"])
    t_sd = schema.det_weights(schema.xvlm_schema(t_cfg, geom["max_pos"]), 2000 + 21, geom["std"])
    batch = synth.make_batch(geom, B, seed=77, ragged=True)
    g = torch.Generator().manual_seed(5)
    s_neg = torch.cat([(torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=g)) % B for _ in range(2)])
    t_neg = torch.cat([(torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=g)) % B for _ in range(2)])
    student.injected_neg_idx, teacher.injected_neg_idx = s_neg, t_neg
    student.keep_injected_neg = teacher.keep_injected_neg = True
    ops.dropout_seed(2024)
    tr = GDTrainer(student, teacher, lr=1e-4, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.bfloat16,
                   use_graph=True, pipeline_teacher=True)
    gb = {k: v.to(DEV) for k, v in batch.items()}
    ops.DROPOUT_LOG = []
    assert tr.step(gb) is None                    # primes the teacher pipeline
    out = tr.step(gb)                             # two eager warm-ups, the capture, then THIS step replayed from the joint graph
    torch.cuda.synchronize()
    log, ops.DROPOUT_LOG = ops.DROPOUT_LOG, None
    assert tr._joint, "the step did not run from a captured graph"
    got = [float(x) for x in out.tolist()]
    got_kd = {k: float(v) for k, v in tr.last_kd.items()}
    got_grad = {n: q.grad.detach().float().cpu().clone() for n, q in student.named_parameters()}

    # the sites of ONE student step: batched text pass [2B rows: clean | masked ids] = 1 + 3 per text layer, batched fusion
    # pass [4B rows: positive | text x negative image | negative text x image | masked] = 5 per fusion layer
    nt, nf = s_cfg["fusion_layer"], s_cfg["text_layers"] - s_cfg["fusion_layer"]
    per_step = (1 + 3 * nt) + 5 * nf
    assert len(log) % per_step == 0 and len(log) >= per_step, (len(log), per_step)
    sites = log[-per_step:]                       # (the capture pass: the call ids the replayed graph carries)
    assert [k for _, k, _, _ in sites[:4]] == ["hidden", "attention_probs", "hidden", "hidden"]
    assert all(shp[0] == 2 * B for _, _, shp, _ in sites[:1 + 3 * nt]) and all(shp[0] == 4 * B for _, _, shp, _ in sites[1 + 3 * nt:])
    st = ops.dropout_state(torch.device(DEV))
    now = int(st[1].item())
    st[1] = now - 1                               # the step word the replayed step drew its masks with (ticked on the device since)
    full = [ops.dropout_mask(c, shp, pp, kind=kind).cpu() for c, kind, shp, pp in sites]
    st[1] = now
    text_sites, fus_sites = full[:1 + 3 * nt], full[1 + 3 * nt:]
    order = ([m[:B] for m in text_sites] + [m[:B] for m in fus_sites] + [m[B:3 * B] for m in fus_sites]
             + [m[B:2 * B] for m in text_sites] + [m[3 * B:] for m in fus_sites])

    leaves = {k: v.clone().requires_grad_(True) for k, v in s_sd.items()}
    tie = lambda sd: {**sd, "text_encoder.cls.predictions.decoder.weight": sd["text_encoder.bert.embeddings.word_embeddings.weight"],
                      "text_encoder.cls.predictions.decoder.bias": sd["text_encoder.cls.predictions.bias"]}
    O.DROPOUT_MASKS = iter(order)
    try:
        oS = O.pretrain_forward(tie(leaves), s_cfg, batch, s_neg)
        assert next(O.DROPOUT_MASKS, None) is None, "the oracle visited fewer dropout sites than the HIP path"
    finally:
        O.DROPOUT_MASKS = None
    with torch.no_grad():
        oT = O.pretrain_forward(tie(t_sd), t_cfg, batch, t_neg)
    okd = O.kd_terms(oS, oT, 1.0)
    ototal, omix = O.gd_loss_mix(oS["loss"], okd)
    ototal.backward()
    want = [float(ototal), float(oS["loss"]["loss_itc"]), float(oS["loss"]["loss_itm"]), float(oS["loss"]["loss_mlm"]),
            float(omix["loss_kd"])]
    for name, a, b in zip(("total", "itc", "itm", "mlm", "kd"), got, want):
        assert abs(a - b) <= 1e-3 * abs(b), f"{name}: {a} vs oracle {b}"
    for k, v in got_kd.items():
        rt = 5e-2 if k.endswith("_logits") else 1e-2
        assert abs(v - float(okd[k])) <= rt * abs(float(okd[k])) + 1e-6, f"kd.{k}: {v} vs oracle {float(okd[k])}"
    stats, num, da, db = [], 0.0, 0.0, 0.0
    gmax = max(float(l.grad.norm()) for l in leaves.values() if l.grad is not None)
    for name, leaf in leaves.items():
        if leaf.grad is None or name not in got_grad or float(leaf.grad.norm()) < 1e-5 * gmax:
            continue
        a, b = got_grad[name].double().reshape(-1), leaf.grad.double().reshape(-1)
        stats.append((float((a - b).norm() / b.norm()), float((a * b).sum() / (a.norm() * b.norm())), name))
        num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
    assert len(stats) > 150
    worst = sorted(stats, reverse=True)[:5]
    assert num / math.sqrt(da * db) > 0.9999, (num / math.sqrt(da * db), worst)
    assert all(r < 0.25 and c > 0.96 for r, c, _ in stats), worst
    # and the dropout really ran: the p = 0 oracle losses differ
    with torch.no_grad():
        o0 = O.pretrain_forward(tie(s_sd), s_cfg, batch, s_neg)
    assert abs(float(o0["loss"]["loss_mlm"]) - want[3]) > 1e-4 * abs(want[3])
    tr.close()
