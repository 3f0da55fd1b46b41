"""Kernel-level parity: every HIP op (through the C ABI via efficientvlm_amd.ops) against a plain PyTorch fp32
reference of the same arithmetic.  fp32 path: <= 1e-4 relative (north_star tolerance); bf16 path: inputs are rounded
to bf16 first and the result must be within 2^-7 relative of the fp32 reference of those rounded inputs.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from efficientvlm_amd import ops as o
    return o


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 1.2e-2


def rnd(shape, dtype, g, scale=1.0):
    x = torch.randn(shape, generator=g) * scale
    return x.to(dtype).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pt,qt", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("I,J,K", [(200, 136, 96), (128, 256, 128), (77, 1002, 64), (300, 24, 200)])
def test_gemm_layouts(dtype, pt, qt, I, J, K):
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(I * 7 + J * 3 + K + pt * 2 + qt)
    p8 = lambda n: (n + 7) // 8 * 8
    # operands with padded leading dimensions (pads are zero)
    Pm = torch.zeros((K, p8(I)) if pt else (I, p8(K)), dtype=dtype, device=DEV)
    Qm = torch.zeros((K, p8(J)) if qt else (J, p8(K)), dtype=dtype, device=DEV)
    Pv = Pm[:, :I] if pt else Pm[:, :K]
    Qv = Qm[:, :J] if qt else Qm[:, :K]
    Pv.copy_(rnd(Pv.shape, dtype, g))
    Qv.copy_(rnd(Qv.shape, dtype, g))
    Cm = torch.zeros((I, p8(J)), dtype=dtype, device=DEV)
    o._gemm(L.dt(dtype), Pm, Qm, Cm, I, J, K, Pm.stride(0), Qm.stride(0), Cm.stride(0), p_trans=pt, q_trans=qt)
    A = Pv.float().t() if pt else Pv.float()
    Bm = Qv.float().t() if qt else Qv.float()
    ref = A @ Bm.t()
    assert rel_err(Cm[:, :J].float(), ref) < tol(dtype)
    assert float(Cm[:, J:].abs().sum()) == 0.0     # pad columns untouched


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    I, J, K = 150, 264, 128
    X, W = rnd((I, K), dtype, g), rnd((J, K), dtype, g, 0.2)
    bias, gate = rnd((J,), torch.float32, g), torch.rand(J, generator=g).to(DEV)
    R = rnd((I, J), dtype, g)
    lin = X.float() @ W.float().t() + bias
    for act, fn in ((L.ACT_GELU, F.gelu), (L.ACT_QUICK_GELU, lambda t: t * torch.sigmoid(1.702 * t))):
        for gp in (L.GATE_PRE, L.GATE_POST):
            Cm = torch.empty((I, J), dtype=dtype, device=DEV)
            H = torch.empty((I, J), dtype=dtype, device=DEV)
            o._gemm(L.dt(dtype), X, W, Cm, I, J, K, K, K, J, bias=bias, gate=gate, preact=H, residual=R, ldx=J, act=act, gate_pos=gp)
            ref = (fn(lin * gate) if gp == L.GATE_PRE else fn(lin) * gate) + R.float()
            assert rel_err(Cm.float(), ref) < tol(dtype), (act, gp)
            assert rel_err(H.float(), lin) < tol(dtype)
    # activation backward fused in the epilogue:  C = (X W^T) .* act'(aux)
    aux = rnd((I, J), dtype, g)
    for act in (L.ACT_GELU, L.ACT_QUICK_GELU):
        Cm = torch.empty((I, J), dtype=dtype, device=DEV)
        o._gemm(L.dt(dtype), X, W, Cm, I, J, K, K, K, J, aux=aux, ldx=J, dact=act)
        a32 = aux.float().requires_grad_(True)
        y = F.gelu(a32) if act == L.ACT_GELU else a32 * torch.sigmoid(1.702 * a32)
        (dact,) = torch.autograd.grad(y.sum(), a32)
        assert rel_err(Cm.float(), (X.float() @ W.float().t()) * dact) < tol(dtype)
    # f32 output from bf16 operands
    if dtype == torch.bfloat16:
        Cf = torch.empty((I, J), dtype=torch.float32, device=DEV)
        o._gemm(L.dt(dtype), X, W, Cf, I, J, K, K, K, J, c_f32=1)
        assert rel_err(Cf, X.float() @ W.float().t()) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_and_packed_autograd(dtype):
    o = ops()
    g = torch.Generator().manual_seed(11)
    B, Ls, K, N = 3, 37, 64, 72
    x = rnd((B, Ls, K), dtype, g).requires_grad_(True)
    ws = [torch.nn.Parameter(rnd((N, K), torch.float32, g, 0.2)) for _ in range(3)]
    bs = [torch.nn.Parameter(rnd((N,), torch.float32, g, 0.2)) for _ in range(3)]
    y = o.linear_packed(x, ws, bs)
    assert y.shape == (B, Ls, 3 * N)
    go = rnd(y.shape, dtype, g)
    y.backward(go)
    xr = x.detach().float().requires_grad_(True)
    wr = [w.detach().to(dtype).float().requires_grad_(True) for w in ws]
    br = [b.detach().clone().requires_grad_(True) for b in bs]
    yr = torch.cat([F.linear(xr, w, b) for w, b in zip(wr, br)], dim=-1)
    yr.backward(go.float())
    assert rel_err(y.float(), yr) < tol(dtype)
    assert rel_err(x.grad.float(), xr.grad) < tol(dtype)
    for w, w2, b, b2 in zip(ws, wr, bs, br):
        assert rel_err(w.grad, w2.grad) < tol(dtype)
        assert rel_err(b.grad, b2.grad) < tol(dtype)
    # CLS-slice input (strided rows), odd output width (padded ld), residual
    w = torch.nn.Parameter(rnd((10, K), torch.float32, g, 0.3))
    b = torch.nn.Parameter(rnd((10,), torch.float32, g))
    xs = x.detach()[:, 0, :]
    y2 = o.linear(xs, w, b)
    assert rel_err(y2.float(), F.linear(xs.float(), w.detach().to(dtype).float(), b.detach())) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act_name,gate_pos", [("quick_gelu", 0), ("gelu", 1)])
@pytest.mark.parametrize("with_gate", [False, True])
def test_mlp_block(dtype, act_name, gate_pos, with_gate):
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(13)
    M, K, Fh = 210, 64, 136
    x = rnd((2, M // 2, K), dtype, g).requires_grad_(True)
    w1 = torch.nn.Parameter(rnd((Fh, K), torch.float32, g, 0.3)); b1 = torch.nn.Parameter(rnd((Fh,), torch.float32, g, 0.3))
    w2 = torch.nn.Parameter(rnd((K, Fh), torch.float32, g, 0.3)); b2 = torch.nn.Parameter(rnd((K,), torch.float32, g, 0.3))
    gate = (torch.rand(1, 1, Fh, generator=g).to(DEV) * 1.2).clamp(0, 1).requires_grad_(True) if with_gate else None
    act = L.ACT_QUICK_GELU if act_name == "quick_gelu" else L.ACT_GELU
    y = o.mlp(x, w1, b1, w2, b2, act, gate=gate, gate_pos=gate_pos, residual=x)
    go = rnd(y.shape, dtype, g)
    y.backward(go)
    fn = (lambda t: t * torch.sigmoid(1.702 * t)) if act_name == "quick_gelu" else F.gelu
    xr = x.detach().float().requires_grad_(True)
    ps = [p.detach().to(dtype).float().requires_grad_(True) if p.dim() == 2 else p.detach().clone().requires_grad_(True)
          for p in (w1, b1, w2, b2)]
    gr = gate.detach().clone().requires_grad_(True) if with_gate else None
    h = F.linear(xr, ps[0], ps[1])
    if dtype == torch.bfloat16:   # the kernel rounds the hidden activation to bf16 between the two GEMMs
        pass
    if gate_pos == 0:
        a = fn(h * gr) if with_gate else fn(h)
    else:
        a = fn(h) * gr if with_gate else fn(h)
    yr = F.linear(a, ps[2], ps[3]) + xr
    yr.backward(go.float())
    t = tol(dtype) * (3 if dtype == torch.bfloat16 else 1)
    assert rel_err(y.float(), yr) < t
    assert rel_err(x.grad.float(), xr.grad) < t
    for p, pr in zip((w1, b1, w2, b2), ps):
        assert rel_err(p.grad, pr.grad) < t
    if with_gate:
        assert rel_err(gate.grad, gr.grad) < t


@pytest.mark.parametrize("M,K,Fh", [(1154, 768, 3072), (3 * 197, 384, 1536), (70, 128, 264)])
@pytest.mark.parametrize("act_name,gate_pos", [("quick_gelu", 0), ("gelu", 1)])
def test_gated_activation_backward_in_the_gemm_epilogue(M, K, Fh, act_name, gate_pos, monkeypatch):
    """Round 5: the backward of the L0-gated FFN activation (dH = (dY W2) act'(.) z and the gate gradient sum_rows(dA .
    d a/d z), eff_vit.py:213-224 / eff_bert.py intermediate gates) runs in the dX product's epilogue
    (evlm_gemm_args.dgate, gemm_bf16_pp192_kernel<*, true>) - against the separate evlm_gated_act_bwd pass it replaces
    (which rounds dA to bf16 first) and against plain fp32 autograd, at ragged row counts and a ragged column tile"""
    o = ops()
    from efficientvlm_amd import _lib as L
    act = L.ACT_QUICK_GELU if act_name == "quick_gelu" else L.ACT_GELU
    fn = (lambda t: t * torch.sigmoid(1.702 * t)) if act_name == "quick_gelu" else F.gelu
    g = torch.Generator().manual_seed(77)
    x0 = rnd((1, M, K), torch.bfloat16, g)
    w1 = torch.nn.Parameter(rnd((Fh, K), torch.float32, g, 0.05)); b1 = torch.nn.Parameter(rnd((Fh,), torch.float32, g, 0.1))
    w2 = torch.nn.Parameter(rnd((K, Fh), torch.float32, g, 0.05)); b2 = torch.nn.Parameter(rnd((K,), torch.float32, g, 0.1))
    gate0 = (torch.rand(1, 1, Fh, generator=g).to(DEV) * 1.3).clamp(0, 1)      # (some gates exactly 0 and 1, as hard-concrete gives)
    go = rnd((1, M, K), torch.bfloat16, g)

    def run(fold):
        monkeypatch.setattr(o, "_NO_GATED_DACT_FOLD", not fold)
        x, gate = x0.clone().requires_grad_(True), gate0.clone().requires_grad_(True)
        for p in (w1, b1, w2, b2): p.grad = None
        o.GEMM_PROFILE = prof = []
        try:
            o.mlp(x, w1, b1, w2, b2, act, gate=gate, gate_pos=gate_pos, residual=x).backward(go)
        finally:
            o.GEMM_PROFILE = None
        return x.grad.float(), gate.grad.clone(), w1.grad.clone(), b1.grad.clone(), prof

    folded, separate = run(True), run(False)
    assert any("gated dact" in str(r) for r in folded[4]), folded[4]
    assert not any("gated dact" in str(r) for r in separate[4])
    xr = x0.float().requires_grad_(True); gr = gate0.clone().requires_grad_(True)
    ps = [p.detach().bfloat16().float().requires_grad_(True) if p.dim() == 2 else p.detach().clone().requires_grad_(True) for p in (w1, b1, w2, b2)]
    h = F.linear(xr, ps[0], ps[1])
    a = fn(h * gr) if gate_pos == 0 else fn(h) * gr
    (F.linear(a, ps[2], ps[3]) + xr).backward(go.float())
    ref = (xr.grad, gr.grad, ps[0].grad, ps[1].grad)
    for name, f_, s_, r_ in zip(("dx", "dgate", "dW1", "db1"), folded, separate, ref):
        # the folded path skips one bf16 rounding (dA), so it may not be further from fp32 autograd than the separate pass
        ef, es = rel_err(f_.float(), r_), rel_err(s_.float(), r_)
        assert ef < 2e-2 and ef <= es * 1.25 + 1e-4, (name, ef, es)
        assert rel_err(f_.float(), s_.float()) < 2e-2, name


def test_a_batch_goes_into_its_static_buffers_in_one_launch():
    """ops.copy_few (evlm_copy_few: up to 8 copies per launch, the units passed by value - no device table to upload for
    sources that change every step): a training batch's six tensors, more than eight units, and pairs that do not qualify
    (odd byte counts, a host source, a strided destination) falling back to Tensor.copy_"""
    o = ops()
    g = torch.Generator().manual_seed(21)
    srcs = [torch.randn(64, 3, 32, 32, generator=g).to(DEV), torch.randint(0, 30000, (64, 30), generator=g).to(DEV),
            torch.randint(0, 2, (64, 30), generator=g).to(DEV), torch.randint(0, 30, (64, 8), generator=g).to(DEV),
            torch.randn(7, 3, generator=g).to(DEV),                       # 84 bytes: not a multiple of 16
            torch.randn(64, 16, generator=g),                              # a host tensor
            torch.randn(33, 8, generator=g).to(DEV)]
    srcs += [torch.randn(256, generator=g).to(DEV) for _ in range(9)]     # > 8 qualifying units: two launches
    dsts = [torch.empty_like(s_, device=DEV) for s_ in srcs]
    strided = torch.empty(33, 16, device=DEV)[:, ::2]
    dsts[6] = strided
    o.copy_few(list(zip(srcs, dsts)))
    torch.cuda.synchronize()
    for s_, d in zip(srcs, dsts):
        assert torch.equal(d.cpu(), s_.cpu())


def test_fusion_batch_selection_and_the_samplers_layout_vectors():
    """ops.select_batches (whole-sample x[sel] with a deterministic one-launch backward) against index_select + autograd's
    index_add, and the layout vectors evlm_sample_negatives writes beside its draws: sel4 = (r, r, drawn text, B + r) over
    the text pass's [text ; masked text] rows, img4 = (r, drawn image, r, r) - models/model_pretrain.py's fusion batch
    [pos ; text x negative image ; negative text x image ; masked text] (reference xvlm.py:436-458)"""
    o = ops()
    g = torch.Generator().manual_seed(9)
    B, Ln, d = 12, 30, 64
    x0 = torch.randn(2 * B, Ln, d, generator=g).to(DEV).bfloat16()
    sim = torch.randn(B, B, generator=g).to(DEV)
    o.dropout_seed(11)
    neg, sel4, img4 = o.sample_negatives(sim, torch.tensor(0.5, device=DEV), None, layout=True)
    ar = torch.arange(B, device=DEV)
    assert torch.equal(sel4, torch.cat([ar, ar, neg[B:], B + ar])) and torch.equal(img4.long(), torch.cat([ar, neg[:B], ar, ar]))
    assert bool((neg[:B] != ar).all()) and bool((neg[B:] != ar).all())
    x = x0.clone().requires_grad_(True)
    y = o.select_batches(x, sel4)
    assert torch.equal(y, x0[sel4])
    gy = torch.randn(y.shape, generator=g).to(DEV).bfloat16()
    y.backward(gy)
    ref = torch.zeros(2 * B, Ln, d, device=DEV).index_add_(0, sel4, gy.float())
    assert rel_err(x.grad.float(), ref) < 4e-3                      # (fp32 sums of up to 3 rows, rounded once)
    x.grad = None
    o.select_batches(x, sel4).backward(gy)
    x2 = x.grad.clone()
    x.grad = None
    o.select_batches(x, sel4).backward(gy)
    assert torch.equal(x.grad, x2)
    m = torch.randint(0, 2, (2 * B, Ln), generator=g).to(DEV)      # int64 masks: 240-byte samples
    assert torch.equal(o.select_batches(m, sel4), m[sel4])


@pytest.mark.parametrize("R,Cn,shape3", [(64, 30522, (8, 8)), (192, 2, None), (40, 1000, None)])
def test_losses_of_one_logits_tensor_sum_their_gradients_in_one_buffer(R, Cn, shape3, monkeypatch):
    """ops.join_grads: the hard-label CE (model_pretrain.py MLM / ITM heads) and the distillation KL (GeneralDistill.py:84-89)
    of the SAME logits write ONE padded gradient buffer (first backward writes, second accumulates in-kernel) - equal to
    autograd's sum of the two separate gradients, also over a retained graph run twice, with ignored rows, and with a third
    consumer that knows nothing of the join"""
    o = ops()
    g = torch.Generator().manual_seed(3)
    x0 = (torch.randn(R, Cn, generator=g) * 2).to(DEV).bfloat16()
    t = (torch.randn(R, Cn, generator=g) * 2).to(DEV).bfloat16()
    lab = torch.randint(0, Cn, (R,), generator=g).to(DEV)
    lab[::7] = -100
    w = (torch.randn(Cn, 16, generator=g) * 0.1).to(DEV).bfloat16()

    def run(join, third=False):
        monkeypatch.setattr(o, "_NO_GRAD_JOIN", not join)
        x = x0.clone().requires_grad_(True)
        y = x * 1.0                                                  # (a non-leaf producer, as the decoder product is)
        if shape3:
            y = y.view(*shape3, Cn)
        y = o.join_grads(y)
        assert (getattr(y, "_evlm_join", None) is not None) == join
        loss = o.cross_entropy(y, lab) * 0.6 + o.soft_cross_entropy(y, t.view(y.shape), 2.0) * 0.4
        if third:
            loss = loss + (y.float().reshape(R, Cn) @ w.float()).square().mean()
        loss.backward(retain_graph=True)
        g1 = x.grad.clone()
        x.grad = None
        loss.backward()
        return loss.detach(), g1, x.grad.clone()

    la, ga1, ga2 = run(True)
    lb, gb1, gb2 = run(False)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))       # (the KL forward sums its rows with f32 atomics: last-bit order noise)
    assert torch.equal(ga1, ga2) and torch.equal(gb1, gb2)
    # (the joined sum rounds to bf16 twice - first writer, accumulating writer; autograd.s sum of two bf16 tensors three times)
    xr = x0.float().requires_grad_(True)
    valid = lab != -100
    ref = F.cross_entropy(xr[valid], lab[valid]) * 0.6 + \
        F.kl_div(F.log_softmax(xr / 2.0, -1), F.softmax(t.float() / 2.0, -1), reduction="batchmean") * 0.4
    ref.backward()
    ea, eb = rel_err(ga1.float(), xr.grad), rel_err(gb1.float(), xr.grad)
    assert ea < 8e-3 and ea <= eb * 1.1 + 1e-4, (ea, eb)
    lc, gc1, gc2 = run(True, third=True)
    ld, gd1, gd2 = run(False, third=True)
    assert rel_err(gc1.float(), gd1.float()) < 8e-3 and torch.equal(gc1, gc2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Bt,E,grouped,packed", [(64, 256, False, False), (64, 256, True, False), (130, 256, True, True),
                                                 (7, 64, False, True), (512, 256, True, True)])
def test_itc_loss_in_one_launch_each_way(dtype, Bt, E, grouped, packed):
    """ops.itc_loss (evlm_itc_loss_fwd / _bwd) against plain fp32 autograd of efficient_models/xvlm.py:384-416:
    logits = I T^t / temp, (CE(logits, labels) + CE(logits^t, labels)) / 2 with identity labels or the idx-equality soft
    labels pos / pos.sum(1); its gradients wrt the features and the temperature; the un-scaled similarities it hands the
    hard-negative sampler; ragged batch sizes, the gathered [Bt, 2E] form, and run-to-run bit identity (fixed-order sums)"""
    o = ops()
    g = torch.Generator().manual_seed(5)
    I = F.normalize(torch.randn(Bt, E, generator=g), dim=-1).to(DEV).to(dtype)
    T = F.normalize(torch.randn(Bt, E, generator=g) + 0.5 * I.cpu().float(), dim=-1).to(DEV).to(dtype)
    temp = torch.nn.Parameter(torch.tensor(0.07, device=DEV))
    group = torch.randint(0, max(2, Bt // 3), (Bt,), generator=g).to(DEV) if grouped else None
    up = torch.tensor(0.7, device=DEV)

    def run():
        temp.grad = None
        if packed:
            both = torch.cat([I, T], 1).requires_grad_(True)
            loss, sim = o.itc_loss(both, None, temp, group)
            (loss * up).backward()
            return loss.detach(), sim, both.grad[:, :E], both.grad[:, E:], temp.grad.clone()
        a, b = I.clone().requires_grad_(True), T.clone().requires_grad_(True)
        loss, sim = o.itc_loss(a, b, temp, group)
        (loss * up).backward()
        return loss.detach(), sim, a.grad, b.grad, temp.grad.clone()

    loss, sim, dI, dT, dtemp = run()
    Ir, Tr = I.float().clone().requires_grad_(True), T.float().clone().requires_grad_(True)
    tr = temp.detach().clone().requires_grad_(True)
    logits = Ir @ Tr.t() / tr
    if grouped:
        pos = torch.eq(group.view(-1, 1), group.view(1, -1)).float()
        labels = pos / pos.sum(1, keepdim=True)
        ref = (-(F.log_softmax(logits, 1) * labels).sum(1).mean() - (F.log_softmax(logits.t(), 1) * labels).sum(1).mean()) / 2
    else:
        lab = torch.arange(Bt, device=DEV)
        ref = (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab)) / 2
    (ref * up).backward()
    assert abs(float(loss) - float(ref.detach())) < 2e-5 * max(1.0, abs(float(ref.detach())))
    assert rel_err(sim, (Ir @ Tr.t()).detach()) < 1e-5
    t = 1e-4 if dtype == torch.float32 else 6e-3          # (bf16: the gradients are rounded to bf16 on the way out)
    assert rel_err(dI.float(), Ir.grad) < t and rel_err(dT.float(), Tr.grad) < t
    assert abs(float(dtemp) - float(tr.grad)) < 1e-4 * max(1.0, abs(float(tr.grad)))
    again = run()
    assert torch.equal(again[0], loss) and torch.equal(again[2], dI) and torch.equal(again[3], dT) and torch.equal(again[4], dtemp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d,eps", [(64, 1e-12), (768, 1e-5), (1536, 1e-5)])
def test_layernorm(dtype, d, eps):
    o = ops()
    g = torch.Generator().manual_seed(17)
    x = (rnd((5, 41, d), dtype, g) * 2 + 0.5).requires_grad_(True)
    gm = torch.nn.Parameter(1 + 0.1 * rnd((d,), torch.float32, g)); bt = torch.nn.Parameter(rnd((d,), torch.float32, g))
    y = o.layer_norm(x, gm, bt, eps)
    go = rnd(y.shape, dtype, g)
    y.backward(go)
    xr = x.detach().float().requires_grad_(True)
    g2, b2 = gm.detach().clone().requires_grad_(True), bt.detach().clone().requires_grad_(True)
    yr = F.layer_norm(xr, (d,), g2, b2, eps)
    yr.backward(go.float())
    assert rel_err(y.float(), yr) < tol(dtype)
    assert rel_err(x.grad.float(), xr.grad) < tol(dtype) * 2
    assert rel_err(gm.grad, g2.grad) < tol(dtype) * 2
    assert rel_err(bt.grad, b2.grad) < tol(dtype) * 2


def _ref_attention(q, k, v, mask, gate, scale):
    s = q @ k.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask[:, None, None, :]
    p = torch.softmax(s, -1)
    o_ = p @ v
    if gate is not None:
        o_ = o_ * gate.view(1, -1, 1, 1)
    return o_, p


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows", [1024, 1027, 12608])
def test_layernorm_row_pair_kernel_is_bit_identical_to_the_one_row_kernels(dtype, rows, monkeypatch):
    """round 6: ln_fwd_pair768_kernel (d = 768: a wave owns two consecutive rows = three 16-byte chunks per lane, the next pair
    prefetched) against the three-pass kernel (EVLM_LN_FWD_3PASS=1, read per call): the same per-lane partial sums in the same
    order and the same reduction tree - outputs, saved mean / rstd and therefore the backward bit-identical; an odd row count
    exercises the half-filled last pair; the fused hidden-state distillation term agrees with the separate reduction"""
    from efficientvlm_amd import ops as o
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, 768, generator=g) * 2 + 0.3).to(DEV, dtype)
    w, b = (torch.rand(768, generator=g) + 0.5).to(DEV), torch.randn(768, generator=g).to(DEV)
    gy = torch.randn(rows, 768, generator=g).to(DEV, dtype)
    outs = []
    for three in ("0", "1"):
        monkeypatch.setenv("EVLM_LN_FWD_3PASS", three)
        xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = o.layer_norm(xx, ww, bb, 1e-12)
        y.backward(gy)
        outs.append((y.detach().clone(), xx.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    monkeypatch.setenv("EVLM_LN_FWD_3PASS", "0")
    t = (x.float() + 0.1 * torch.randn(rows, 768, generator=g).to(DEV)).to(dtype)
    slots = torch.zeros(o.hidden_kd_slots(), dtype=torch.float32, device=DEV)
    xx = x.clone().requires_grad_(True)
    y2, alias, kd = o.layer_norm_fork_kd(xx, w, b, 1e-12, t, slots, 0.5 / x.numel())
    assert torch.equal(y2.detach(), outs[0][0])
    ref = 0.5 * torch.nn.functional.mse_loss(x.float(), t.float())
    assert abs(float(kd.sum()) - float(ref)) <= 1e-4 * float(ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,L,dh", [(3, 12, 7, 64), (2, 4, 11, 16), (2, 12, 40, 64)])
def test_causal_self_attention(dtype, B, H, L, dh):
    """decoder self-attention (BertLMHeadModel, eff_bert.py:975-996): key-padding mask AND -10000 on keys after the query,
    forward and backward against the explicit [B, 1, L, L] additive mask in torch"""
    o = ops()
    g = torch.Generator().manual_seed(23 + L)
    d = H * dh
    qkv = rnd((B, L, 3 * d), dtype, g).requires_grad_(True)
    mask = torch.zeros(B, L)
    mask[0, L - 2:] = -10000.0
    mask = mask.to(DEV)
    scale = dh ** -0.5
    O, P = o.self_attention(qkv, H, dh, scale, mask=mask, causal=True)
    gO, gP = rnd(O.shape, dtype, g), rnd(P.shape, dtype, g, 0.1)
    (O.float() * gO.float()).sum().add((P.float() * gP.float()).sum()).backward()
    xr = qkv.detach().float().requires_grad_(True)
    sp = lambda t: t.view(B, L, H, dh).transpose(1, 2)
    q, k, v = sp(xr[..., :d]), sp(xr[..., d:2 * d]), sp(xr[..., 2 * d:])
    tri = torch.tril(torch.ones(L, L, device=DEV))
    full = (1.0 - tri[None, None] * (mask == 0).float()[:, None, None, :]) * -10000.0      # HF: (1 - causal*pad) * -10000
    Pr = torch.softmax(q @ k.transpose(-1, -2) * scale + full, -1)
    Or = (Pr @ v).transpose(1, 2).reshape(B, L, d)
    ((Or * gO.float()).sum() + (Pr * gP.float()).sum()).backward()
    t = tol(dtype)
    assert float(P.float().triu(1).abs().max()) == 0.0
    assert rel_err(P.float(), Pr) < t
    assert rel_err(O.float(), Or) < t * 2
    assert rel_err(qkv.grad.float(), xr.grad) < t * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,L,dh", [(2, 4, 5, 16), (2, 12, 197, 64), (3, 12, 30, 64), (1, 2, 70, 32), (1, 3, 577, 64),
                                      (2, 2, 901, 64)])          # 901 = 480x480 images: K and V take turns in LDS
def test_self_attention(dtype, B, H, L, dh):
    o = ops()
    g = torch.Generator().manual_seed(19 + L)
    d = H * dh
    qkv = rnd((B, L, 3 * d), dtype, g).requires_grad_(True)
    mask = torch.zeros(B, L)
    mask[0, L - 2:] = -10000.0
    mask = mask.to(DEV)
    gate = torch.rand(1, H, 1, 1, generator=g).to(DEV).requires_grad_(True)
    scale = dh ** -0.5
    O, P = o.self_attention(qkv, H, dh, scale, mask=mask, gate=gate)
    gO, gP = rnd(O.shape, dtype, g), rnd(P.shape, dtype, g, 0.1)
    (O.float() * gO.float()).sum().add((P.float() * gP.float()).sum()).backward()
    xr = qkv.detach().float().requires_grad_(True)
    gr = gate.detach().clone().requires_grad_(True)
    sp = lambda t: t.view(B, L, H, dh).transpose(1, 2)
    q, k, v = xr[..., :d], xr[..., d:2 * d], xr[..., 2 * d:]
    Or, Pr = _ref_attention(sp(q), sp(k), sp(v), mask, gr, scale)
    Or = Or.transpose(1, 2).reshape(B, L, d)
    ((Or * gO.float()).sum() + (Pr * gP.float()).sum()).backward()
    t = tol(dtype)
    assert rel_err(P.float(), Pr) < t
    assert rel_err(O.float(), Or) < t * 2
    assert rel_err(qkv.grad.float(), xr.grad) < t * 4
    assert rel_err(gate.grad, gr.grad) < t * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Lq,Lk,dh", [(2, 4, 8, 5, 16), (3, 12, 30, 197, 64), (2, 2, 30, 901, 64)])
def test_cross_attention(dtype, B, H, Lq, Lk, dh):
    o = ops()
    g = torch.Generator().manual_seed(23)
    d = H * dh
    q = rnd((B, Lq, d), dtype, g).requires_grad_(True)
    kv = rnd((B, Lk, 2 * d), dtype, g).requires_grad_(True)
    scale = 1.0 / math.sqrt(dh)
    O, P = o.cross_attention(q, kv, H, dh, scale)
    gO = rnd(O.shape, dtype, g)
    (O.float() * gO.float()).sum().backward()
    qr, kvr = q.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
    sp = lambda t, Ln: t.reshape(B, Ln, H, dh).transpose(1, 2)
    Or, Pr = _ref_attention(sp(qr, Lq), sp(kvr[..., :d], Lk), sp(kvr[..., d:], Lk), None, None, scale)
    Or = Or.transpose(1, 2).reshape(B, Lq, d)
    (Or * gO.float()).sum().backward()
    t = tol(dtype)
    assert rel_err(P.float(), Pr) < t
    assert rel_err(O.float(), Or) < t * 2
    assert rel_err(q.grad.float(), qr.grad) < t * 4
    assert rel_err(kv.grad.float(), kvr.grad) < t * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_losses(dtype):
    o = ops()
    g = torch.Generator().manual_seed(29)
    a = rnd((3, 7, 5, 11), dtype, g).requires_grad_(True)
    b = rnd((3, 7, 5, 11), dtype, g)
    l = o.mse(a, b, weight=11.0)
    l.backward()
    ar = a.detach().float().requires_grad_(True)
    lr = F.mse_loss(ar, b.float()) * 11.0
    lr.backward()
    assert rel_err(l, lr) < 1e-5 and rel_err(a.grad.float(), ar.grad) < tol(dtype)
    # hard-label CE with ignored rows, odd class count (padded gradient buffer)
    R, Cn = 24, 1003
    logits = rnd((4, 6, Cn), dtype, g, 2.0).requires_grad_(True)
    labels = torch.randint(0, Cn, (4, 6), generator=g)
    labels[0, 1] = -100; labels[3, 5] = -100
    labels = labels.to(DEV)
    l = o.cross_entropy(logits.view(-1, Cn), labels.view(-1))
    (l * 1.7).backward()
    lg = logits.detach().float().requires_grad_(True)
    lr = F.cross_entropy(lg.view(-1, Cn), labels.view(-1), ignore_index=-100)
    (lr * 1.7).backward()
    assert rel_err(l, lr) < 1e-5 and rel_err(logits.grad.float(), lg.grad) < tol(dtype)
    # KL (soft_cross_entropy) with temperature
    s = rnd((4, 6, Cn), dtype, g, 2.0).requires_grad_(True)
    t_ = rnd((4, 6, Cn), dtype, g, 2.0)
    l = o.soft_cross_entropy(s, t_, temperature=2.0)
    l.backward()
    sr = s.detach().float().requires_grad_(True)
    lr = F.kl_div(F.log_softmax(sr / 2.0, -1).view(-1, Cn), F.softmax(t_.float() / 2.0, -1).view(-1, Cn), reduction="batchmean")
    lr.backward()
    assert rel_err(l, lr) < 2e-5 and rel_err(s.grad.float(), sr.grad) < tol(dtype)
    # log_softmax
    x = rnd((9, 33), dtype, g, 3.0).requires_grad_(True)
    y = o.log_softmax(x)
    gy = rnd(y.shape, dtype, g)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    yr = F.log_softmax(xr, -1)
    yr.backward(gy.float())
    assert rel_err(y.float(), yr) < tol(dtype) and rel_err(x.grad.float(), xr.grad) < tol(dtype) * 2


def test_cross_entropy_with_an_out_of_range_label_is_loud_not_out_of_bounds():
    """a label outside [0, C) that is not the ignore index (corrupt masked_ids, vocabulary mismatch) must not index the
    logits row: the loss and that row's gradient are NaN (F.cross_entropy would trip a device assert), the other rows'
    gradients stay finite"""
    o = ops()
    g = torch.Generator().manual_seed(31)
    R, Cn = 6, 40
    for bad in (Cn, Cn + 100000, -7):
        logits = rnd((R, Cn), torch.float32, g, 2.0).requires_grad_(True)
        labels = torch.randint(0, Cn, (R,), generator=g)
        labels[2] = bad
        l = o.cross_entropy(logits, labels.to(DEV))
        l.backward()
        torch.cuda.synchronize()
        assert torch.isnan(l)
        assert torch.isnan(logits.grad[2]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embeddings_and_movement(dtype):
    o = ops()
    g = torch.Generator().manual_seed(31)
    V, Pm, d, B, Ls = 50, 20, 64, 3, 9
    word = torch.nn.Parameter(rnd((V, d), torch.float32, g)); pos = torch.nn.Parameter(rnd((Pm, d), torch.float32, g))
    typ = torch.nn.Parameter(rnd((2, d), torch.float32, g))
    ids = torch.randint(1, V, (B, Ls), generator=g); ids[1, 6:] = 0
    ids = ids.to(DEV)
    e = o.bert_embed(ids, word, pos, typ, 0, dtype)
    ge = rnd(e.shape, dtype, g)
    e.backward(ge)
    wr, pr, tr = (p.detach().clone().requires_grad_(True) for p in (word, pos, typ))
    er = F.embedding(ids, wr, padding_idx=0) + tr[0] + pr[:Ls][None]
    er.backward(ge.float())
    assert rel_err(e.float(), er) < tol(dtype)
    for p, p2 in ((word, wr), (pos, pr), (typ, tr)):
        assert rel_err(p.grad, p2.grad) < 1e-5
    # ViT patch embedding
    img = torch.randn(2, 3, 32, 32, generator=g).to(DEV)
    pw = torch.nn.Parameter(rnd((d, 3, 16, 16), torch.float32, g, 0.1)); cls = torch.nn.Parameter(rnd((d,), torch.float32, g))
    pe = torch.nn.Parameter(rnd((5, d), torch.float32, g))
    x = o.vit_embed(img, pw, cls, pe, 16, dtype)
    gx = rnd(x.shape, dtype, g)
    x.backward(gx)
    pw2, c2, p2 = (p.detach().clone().requires_grad_(True) for p in (pw, cls, pe))
    im = img.to(dtype).float()
    t = F.conv2d(im, pw2.to(dtype).float() if dtype == torch.bfloat16 else pw2, stride=16).flatten(2).transpose(1, 2)
    xr = torch.cat([c2.expand(2, 1, -1), t], 1) + p2[None]
    xr.backward(gx.float())
    assert rel_err(x.float(), xr) < tol(dtype)
    assert rel_err(pw.grad, pw2.grad) < tol(dtype) and rel_err(cls.grad, c2.grad) < 1e-5 and rel_err(pe.grad, p2.grad) < 1e-5
    # masked-position gather with duplicate positions
    h = rnd((B, Ls, d), dtype, g).requires_grad_(True)
    mp = torch.tensor([[1, 3, 0, 0], [2, 2, 5, 0], [8, 1, 4, 6]]).to(DEV)
    y = o.gather_rows(h, mp)
    gy = rnd(y.shape, dtype, g)
    y.backward(gy)
    hr = h.detach().float().requires_grad_(True)
    yr = torch.gather(hr, 1, mp.unsqueeze(2).expand(-1, -1, d))
    yr.backward(gy.float())
    assert torch.equal(y.float(), yr) and rel_err(h.grad.float(), hr.grad) < tol(dtype)
    # L2 normalise on a strided CLS slice, GELU
    z = rnd((B, Ls, d), dtype, g).requires_grad_(True)
    n = o.l2_normalize(z[:, 0, :])
    gn = rnd(n.shape, dtype, g)
    n.backward(gn)
    zr = z.detach().float().requires_grad_(True)
    nr = F.normalize(zr[:, 0, :], dim=-1)
    nr.backward(gn.float())
    assert rel_err(n.float(), nr) < tol(dtype) and rel_err(z.grad.float(), zr.grad) < tol(dtype) * 2
    u = rnd((7, 128), dtype, g, 2.0).requires_grad_(True)
    v = o.gelu(u)
    v.backward(torch.ones_like(v))
    ur = u.detach().float().requires_grad_(True)
    F.gelu(ur).sum().backward()
    assert rel_err(v.float(), F.gelu(ur)) < tol(dtype) and rel_err(u.grad.float(), ur.grad) < tol(dtype)
    # casts
    c = torch.randn(1001, generator=g).to(DEV)
    assert torch.equal(o.cast(c, torch.bfloat16), c.to(torch.bfloat16))


def test_l0_gates_match_reference_fixture(golden_dir):
    """train-mode z within fp32 rounding; eval-mode 0/1 masks BIT-EXACT against the reference's own output."""
    import os
    o = ops()
    fx = dict(np.load(os.path.join(golden_dir, "l0_full.npz")))
    names = {"vision_head": "vision_head_loga", "text_head": "text_head_loga", "cross_head": "cross_head_loga",
             "vision_intermediate": "vision_int_loga", "text_intermediate": "text_int_loga",
             "cross_intermediate": "cross_int_loga"}
    for t, pn in names.items():
        loga = torch.from_numpy(fx["in." + pn]).to(DEV).requires_grad_(True)
        eps = torch.from_numpy(fx["in.eps." + t]).to(DEV)
        z = o.l0_sample(loga, eps, 2.0 / 3.0)
        ref = torch.from_numpy(fx[f"train.z.{t}_z"]).reshape(z.shape)
        assert float((z.cpu() - ref).abs().max()) < 2e-6, t
        ze = o.l0_deterministic(loga, 2.0 / 3.0, 0.8)
        refe = fx[f"eval.z.{t}_z"].reshape(ze.shape)
        assert np.array_equal(ze.cpu().numpy(), refe), f"eval mask of {t} differs from the reference"


def test_fused_lagrangian_term_matches_the_reference_vectors_and_the_oracles_gradient(golden_dir):
    """evlm_l0_lagrangian_fwd / _bwd (round 5: the Lagrangian sparsity term of xvlm_l0_module.py in one launch each way)
    against tests/golden/l0_full.npz - (lagrangian, expected sparsity, target sparsity) captured from the REFERENCE's module
    at five points of the warm-up ramp, and its lambda gradients - and against the oracle's autograd for the gate
    log-alphas; with the step counter as a host number and as a device scalar (the captured pruning steps), and with
    gradients returned (plain tensors) as well as accumulated in place (parameters whose .grad is pre-allocated)"""
    import os
    from oracle import xvlm_oracle as Oo
    o = ops()
    fx = dict(np.load(os.path.join(golden_dir, "l0_full.npz")))
    names = ["vision_head_loga", "text_head_loga", "cross_head_loga", "vision_int_loga", "text_int_loga", "cross_int_loga"]
    consts = Oo.l0_constants(768, 3072, 12, 6, 3, 3)
    weights = [consts["params_per_head"]] * 3 + [consts["params_per_int"]] * 3
    xn = (0 - (-0.1)) / (1.1 - (-0.1))
    logit_c = (math.log(xn) - math.log(1 - xn)) * (2.0 / 3.0)
    mk = lambda: ([torch.from_numpy(fx["in." + n]).to(DEV).requires_grad_(True) for n in names],
                  torch.from_numpy(fx["in.lambda_1"]).to(DEV).requires_grad_(True),
                  torch.from_numpy(fx["in.lambda_2"]).to(DEV).requires_grad_(True))
    logas, l1, l2 = mk()
    for step, trip in zip(fx["lagrangian.steps"], fx["lagrangian.triples"]):
        for st in (int(step), torch.tensor(float(step), device=DEV)):
            lag, es, ts = o.l0_lagrangian(logas, weights, logit_c, 1e-6, consts["prunable"], 0.6, 0.0, 200, st, l1, l2)
            np.testing.assert_allclose([float(lag), float(es), float(ts)], trip, rtol=1e-5, atol=1e-7)
    # gradients at step 37: oracle autograd on the CPU (the same restatement the fixture pins)
    cl = {n: torch.from_numpy(fx["in." + n]).clone().requires_grad_(True) for n in names}
    c1 = torch.from_numpy(fx["in.lambda_1"]).clone().requires_grad_(True)
    c2 = torch.from_numpy(fx["in.lambda_2"]).clone().requires_grad_(True)
    lo, _, _ = Oo.l0_lagrangian(cl, c1, c2, consts, 37, target_sparsity=0.6, lagrangian_warmup=200)
    (lo * 1.7).backward()
    close_ = lambda a, b, what: np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=1e-9, err_msg=what)
    # (a) gradients returned
    logas, l1, l2 = mk()
    lag, _, _ = o.l0_lagrangian(logas, weights, logit_c, 1e-6, consts["prunable"], 0.6, 0.0, 200, 37, l1, l2)
    (lag * 1.7).backward()
    for p, n in zip(logas, names):
        close_(p.grad, cl[n].grad, n)
    close_(l1.grad, c1.grad, "lambda_1"); close_(l2.grad, c2.grad, "lambda_2")
    np.testing.assert_allclose(float(l1.grad) / 1.7, float(fx["grad.lambda_1"]), rtol=1e-5)
    np.testing.assert_allclose(float(l2.grad) / 1.7, float(fx["grad.lambda_2"]), rtol=1e-5)
    # (b) accumulated in place on top of what is already there (the trainers' flat gradient slabs)
    logas, l1, l2 = mk()
    for p in logas + [l1, l2]:
        p.grad = torch.zeros_like(p)
    o.WGRAD_INPLACE = True
    try:
        for _ in range(2):                       # two backward passes into the same buffers: twice the gradient
            lag, _, _ = o.l0_lagrangian(logas, weights, logit_c, 1e-6, consts["prunable"], 0.6, 0.0, 200,
                                        torch.tensor(37.0, device=DEV), l1, l2)
            (lag * 1.7).backward()
    finally:
        o.WGRAD_INPLACE = False
    for p, n in zip(logas, names):
        close_(p.grad * 0.5, cl[n].grad, n + " (in place)")
    close_(l1.grad * 0.5, c1.grad, "lambda_1 (in place)"); close_(l2.grad * 0.5, c2.grad, "lambda_2 (in place)")


@pytest.mark.parametrize("pt,qt", [(0, 0), (0, 1), (1, 1)])
def test_gemm_large_tiles_and_split_k(pt, qt):
    """shapes that select the 128x128 tile (>= 384 tiles), edge tiles in both dimensions, and the split-K weight-gradient
    path (f32 output, atomics)"""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(41 + pt + 2 * qt)
    dtype = torch.bfloat16
    for (I, J, K, c_f32) in [(2500, 2600, 128, 0), (2560, 2560, 192, 0), (320, 264, 1920, 1), (768, 768, 1216, 1), (2500, 2600, 128, 1)]:
        Pm = rnd((K, I) if pt else (I, K), dtype, g, 0.5)
        Qm = rnd((K, J) if qt else (J, K), dtype, g, 0.5)
        if pt and I % 8:
            continue
        Cm = torch.full((I, J), 7.0, dtype=torch.float32 if c_f32 else dtype, device=DEV)
        o._gemm(L.dt(dtype), Pm, Qm, Cm, I, J, K, Pm.stride(0), Qm.stride(0), J, p_trans=pt, q_trans=qt, c_f32=c_f32)
        A = Pm.float().t() if pt else Pm.float()
        Bm = Qm.float().t() if qt else Qm.float()
        ref = A @ Bm.t()
        assert rel_err(Cm.float(), ref) < (2e-5 if c_f32 else tol(dtype)), (I, J, K, c_f32)


@pytest.mark.parametrize("M", [256, 1802])      # 1802 = 2 x 901 tokens: not a multiple of the 64-row K tile (main part + padded tail)
def test_inplace_parameter_gradients_match_autograd(M):
    """WGRAD_INPLACE (the trainer's mode): kernels accumulate dW / db / dgamma / dbeta straight into param.grad"""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(43)
    dtype = torch.bfloat16
    K, N, Fh = 64, 72, 136
    x = rnd((2, M // 2, K), dtype, g)

    def make():
        gg = torch.Generator().manual_seed(7)
        mk = lambda *s: torch.nn.Parameter(rnd(s, torch.float32, gg, 0.3))
        return dict(ws=[mk(N, K) for _ in range(3)], bs=[mk(N) for _ in range(3)], w1=mk(Fh, 3 * N), b1=mk(Fh),
                    w2=mk(K, Fh), b2=mk(K), gm=mk(K), bt=mk(K))

    def run(P, inplace, slab=False):
        params = P["ws"] + P["bs"] + [P["w1"], P["b1"], P["w2"], P["b2"], P["gm"], P["bt"]]
        for p in params:
            p.grad = torch.full_like(p, 0.5) if inplace else None    # pre-existing content must be ADDED to
        if slab:      # q | k | v gradients back to back, as in the optimiser slabs: the packed dW is ONE product
            gw = torch.full((3 * N * K,), 0.5, device=DEV); gbv = torch.full((3 * N,), 0.5, device=DEV)
            for i in range(3):
                P["ws"][i].grad = gw[i * N * K:(i + 1) * N * K].view(N, K)
                P["bs"][i].grad = gbv[i * N:(i + 1) * N]
        xi = x.clone().requires_grad_(True)
        h = o.linear_packed(xi, P["ws"], P["bs"])
        y = o.mlp(h, P["w1"], P["b1"], P["w2"], P["b2"], L.ACT_GELU, gate_pos=L.GATE_POST)
        y = o.layer_norm(y, P["gm"], P["bt"], 1e-5)
        y2 = o.linear_packed(xi, P["ws"], P["bs"])                    # the same weights used twice
        o.WGRAD_INPLACE = inplace
        try:
            (y.float().square().mean() + y2.float().mean()).backward()
        finally:
            o.WGRAD_INPLACE = False
        return [p.grad.clone() - (0.5 if inplace else 0.0) for p in params], xi.grad

    ga, xa = run(make(), False)
    gb, xb = run(make(), True)
    gc, xc = run(make(), True, slab=True)
    assert rel_err(xb.float(), xa.float()) < 1e-6 and rel_err(xc.float(), xa.float()) < 1e-6
    for a, b, c in zip(ga, gb, gc):
        assert rel_err(b, a) < 3e-4      # the 0.5 pre-fill costs ~6e-8 absolute on gradients of ~1e-4
        assert rel_err(c, a) < 3e-4


@pytest.mark.parametrize("M,twice", [(2048, False), (1802, False), (256, False), (2048, True), (1802, True)])
def test_first_touch_assignment_of_weight_gradients(M, twice):
    """ops.WGRAD_ASSIGN on single layers, gradients poisoned with NaN beforehand: M = 2048 - the queued product ASSIGNS
    (no zero-fill, no read of C); 1802 = 28 x 64 + 10 - the queued main part is demoted by the tail's immediate product;
    256 - too short to be queued: zero-filled, then accumulated; `twice` - the same weights used by two products of one
    flush (atomics, no assignment).  An eligible weight nothing touches is zeroed by finish_assign.  Against autograd."""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(91)
    K, N, Fh = 64, 72, 136
    x = rnd((2, M // 2, K), torch.bfloat16, g)

    def make():
        gg = torch.Generator().manual_seed(7)
        mk = lambda *s: torch.nn.Parameter(rnd(s, torch.float32, gg, 0.3))
        return dict(ws=[mk(N, K) for _ in range(3)], bs=[mk(N) for _ in range(3)], w1=mk(Fh, 3 * N), b1=mk(Fh),
                    w2=mk(K, Fh), b2=mk(K), unused=mk(N, K))

    def run(P, assign):
        weights = P["ws"] + [P["w1"], P["w2"], P["unused"]]
        params = weights + P["bs"] + [P["b1"], P["b2"]]
        if assign:      # q | k | v gradients back to back, as in the optimiser slabs; weights poisoned, the rest zeroed
            gw = torch.full((3 * N * K,), float("nan"), device=DEV)
            for i in range(3):
                P["ws"][i].grad = gw[i * N * K:(i + 1) * N * K].view(N, K)
            for p in (P["w1"], P["w2"], P["unused"]):
                p.grad = torch.full_like(p, float("nan"))
            for p in P["bs"] + [P["b1"], P["b2"]]:
                p.grad = torch.zeros_like(p)
            state = {"skip": {p.grad.data_ptr(): p.grad for p in weights}, "done": set(), "pending": {}}
        xi = x.clone().requires_grad_(True)
        h = o.linear_packed(xi, P["ws"], P["bs"])
        y = o.mlp(h, P["w1"], P["b1"], P["w2"], P["b2"], L.ACT_GELU, gate_pos=L.GATE_POST)
        loss = y.float().square().mean()
        if twice:
            loss = loss + o.linear_packed(xi, P["ws"], P["bs"]).float().mean()
        if assign:
            o.WGRAD_INPLACE, o.WGRAD_DEFER, o.WGRAD_ASSIGN = True, [], state
        try:
            loss.backward()
            if assign:
                o.flush_wgrad()
                o.finish_assign()
        finally:
            o.WGRAD_INPLACE, o.WGRAD_DEFER, o.WGRAD_ASSIGN = False, None, None
        return [(p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in params], xi.grad

    ga, xa = run(make(), False)
    gb, xb = run(make(), True)
    assert rel_err(xb.float(), xa.float()) < 1e-6
    for a, b in zip(ga, gb):
        assert bool(torch.isfinite(b).all())
        assert rel_err(b, a) < 3e-4
    assert float(gb[5].abs().max()) == 0.0                # the weight no product touched


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cross_attention_shared_kv_index(dtype):
    """several query batches attend to the same K/V row (positive / hard-negative / MLM passes sharing an image):
    forward equals the materialised gather; dKV is the sum over the sharing batches"""
    o = ops()
    g = torch.Generator().manual_seed(47)
    B, Bkv, H, Lq, Lk, dh = 7, 3, 12, 30, 197, 64
    d = H * dh
    q = rnd((B, Lq, d), dtype, g).requires_grad_(True)
    kv = rnd((Bkv, Lk, 2 * d), dtype, g).requires_grad_(True)
    idx = torch.tensor([0, 2, 1, 1, 0, 2, 1]).to(DEV)
    scale = 1.0 / math.sqrt(dh)
    O, P = o.cross_attention(q, kv, H, dh, scale, kv_index=idx)
    gO, gP = rnd(O.shape, dtype, g), rnd(P.shape, dtype, g, 0.1)
    ((O.float() * gO.float()).sum() + (P.float() * gP.float()).sum()).backward()
    qr, kvr = q.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
    kvg = kvr[idx]
    sp = lambda t, Ln: t.reshape(B, Ln, H, dh).transpose(1, 2)
    Or, Pr = _ref_attention(sp(qr, Lq), sp(kvg[..., :d], Lk), sp(kvg[..., d:], Lk), None, None, scale)
    Or = Or.transpose(1, 2).reshape(B, Lq, d)
    ((Or * gO.float()).sum() + (Pr * gP.float()).sum()).backward()
    t = tol(dtype)
    assert rel_err(P.float(), Pr) < t and rel_err(O.float(), Or) < t * 2
    assert rel_err(q.grad.float(), qr.grad) < t * 4
    assert rel_err(kv.grad.float(), kvr.grad) < t * 4


@pytest.mark.parametrize("store_p", [False, True])
@pytest.mark.parametrize("B,H,L", [(3, 12, 197), (2, 4, 30), (2, 2, 577)])
def test_fused_attention_map_distillation_equals_the_separate_reduction(B, H, L, store_p, monkeypatch):
    """evlm_attn_fwd_args.kd_teacher: MSELoss(P, P_t) * P.shape[-1] (GeneralDistill.py:63-69) accumulated inside the attention
    forward kernel, and its gradient formed from P_t inside the backward kernel, against the separate path (the map read
    back by evlm_mse_fwd / evlm_mse_bwd and handed to the attention backward as dP_ext).  store_p: the round-2 form
    (backward from the stored bf16 map: the fused term is formed from the bf16-rounded probabilities, exactly the separate
    reduction's operands); otherwise - and by default for Lk <= 224 - the recomputing form, whose term is formed from the
    fp32 probabilities and is held to the fp32 softmax instead."""
    o = ops()
    monkeypatch.setattr(o, "ATTN_STORE_P", store_p)
    g = torch.Generator().manual_seed(61)
    dh, d = 64, H * 64
    qkv0 = rnd((B, L, 3 * d), torch.bfloat16, g, 0.7)
    with torch.no_grad():
        _, Pt = o.self_attention(rnd((B, L, 3 * d), torch.bfloat16, g, 0.7), H, dh, 0.125)       # a "teacher" map (padded rows)
    gO = rnd((B, L, d), torch.bfloat16, g)
    coef = 0.4
    a = qkv0.clone().requires_grad_(True)
    O1, P1, kd1 = o.self_attention(a, H, dh, 0.125, kd_teacher=Pt, kd_weight=float(L))
    ((O1 * gO).sum() + coef * kd1).backward()
    b = qkv0.clone().requires_grad_(True)
    O2, P2 = o.self_attention(b, H, dh, 0.125)
    kd2 = o.mse(P2, Pt, weight=float(L))
    ((O2 * gO).sum() + coef * kd2).backward()
    ref = torch.nn.functional.mse_loss(P2.detach().float(), Pt.float()) * L
    assert torch.equal(O1, O2) and torch.equal(P1, P2)
    assert rel_err(kd2, ref) < 1e-4
    recomputes = (not store_p) and L <= 224
    if recomputes:
        sp = lambda t: t.float().view(B, L, H, dh).transpose(1, 2)
        P32 = torch.softmax(sp(qkv0[..., :d]) @ sp(qkv0[..., d:2 * d]).transpose(-1, -2) * 0.125, -1)
        assert rel_err(kd1, torch.nn.functional.mse_loss(P32, Pt.float()) * L) < 1e-4
        assert rel_err(kd1, ref) < 2e-2                 # (the separate path squares differences of bf16-rounded maps)
    else:
        assert rel_err(kd1, ref) < 1e-4
    # the fused path keeps dP in fp32 where the separate one rounds it to bf16 on its way through HBM
    assert rel_err(a.grad.float(), b.grad.float()) < (1.2e-2 if recomputes else 6e-3)


@pytest.mark.parametrize("B,L,d", [(64, 197, 768), (3, 50, 512), (2, 577, 768), (5, 7, 1536)])
def test_fused_hidden_state_distillation_equals_layernorm_plus_the_separate_reduction(B, L, d):
    """evlm_layernorm_fwd_kd / evlm_layernorm_bwd_kd (round 5): the hidden-state distillation term of a pre-LN block's input
    (GeneralDistill.py:60-82: MSELoss(student state, teacher state)) formed inside the block's first LayerNorm - forward sum
    from the row in registers, backward gradient 2 c g (x - t) added to dx in the kernel - against the separate path
    (layer_norm_fork with a tap + evlm_mse): the LayerNorm output is bit-identical, the term agrees to f32 summation order,
    the input gradient to one bf16 rounding (the separate path rounds the term's gradient to bf16 on its way through HBM),
    and both are held to an fp32 torch reference"""
    o = ops()
    g = torch.Generator().manual_seed(17 + L)
    x0 = rnd((B, L, d), torch.bfloat16, g, 0.8)
    t = rnd((B, L, d), torch.bfloat16, g, 0.8)
    gam = torch.nn.Parameter((torch.rand(d, generator=g) + 0.5).to(DEV))
    bet = torch.nn.Parameter((torch.randn(d, generator=g) * 0.1).to(DEV))
    gy, gres = rnd((B, L, d), torch.bfloat16, g), rnd((B, L, d), torch.bfloat16, g)
    coef = 0.37

    def fused():
        x = x0.clone().requires_grad_(True)
        gam.grad = bet.grad = None
        slots = torch.zeros(o.hidden_kd_slots(), dtype=torch.float32, device=DEV)
        y, res, sl = o.layer_norm_fork_kd(x, gam, bet, 1e-5, t, slots, 1.0 / x.numel())
        term = sl.sum()
        ((y * gy).sum() + (res * gres).sum() + coef * term).backward()
        return y.detach(), term.detach(), x.grad, gam.grad.clone(), bet.grad.clone()

    def separate():
        x = x0.clone().requires_grad_(True)
        gam.grad = bet.grad = None
        y, res, tap = o.layer_norm_fork(x, gam, bet, 1e-5, tap=True)
        term = o.mse(tap, t)
        ((y * gy).sum() + (res * gres).sum() + coef * term).backward()
        return y.detach(), term.detach(), x.grad, gam.grad.clone(), bet.grad.clone()

    a, b = fused(), separate()
    assert torch.equal(a[0], b[0])
    ref = torch.nn.functional.mse_loss(x0.float(), t.float())
    assert rel_err(a[1], ref) < 1e-5 and rel_err(b[1], ref) < 1e-5
    xr = x0.float().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), gam.detach(), bet.detach(), 1e-5)
    ((yr * gy.float()).sum() + (xr * gres.float()).sum() + coef * torch.nn.functional.mse_loss(xr, t.float())).backward()
    l2 = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())
    assert l2(a[2].float(), xr.grad) < 4e-3 and l2(b[2].float(), xr.grad) < 4e-3
    assert l2(a[2].float(), b[2].float()) < 4e-3
    assert rel_err(a[3], b[3]) < 1e-5 and rel_err(a[4], b[4]) < 1e-5     # (same dy, same x: the column sums, up to atomics' order)
    # the distillation gradient alone (dy = 0, no residual gradient): exactly 2 c g (x - t), rounded once
    x = x0.clone().requires_grad_(True)
    slots = torch.zeros(o.hidden_kd_slots(), dtype=torch.float32, device=DEV)
    y, res, sl = o.layer_norm_fork_kd(x, gam, bet, 1e-5, t, slots, 1.0 / x.numel())
    ((y * 0).sum() + coef * sl.sum()).backward()
    want = (2 * coef / x0.numel()) * (x0.float() - t.float())
    assert l2(x.grad.float(), want) < 3e-3


@pytest.mark.parametrize("Bimg,rows,Lq,N,H", [(6, 4, 30, 197, 12), (5, 1, 30, 197, 12), (3, 3, 17, 100, 4), (2, 5, 40, 224, 2)])
def test_fused_cross_attention_forward_is_bit_identical_to_the_two_launch_path(Bimg, rows, Lq, N, H):
    """evlm_xattn_fused_fwd (K/V projection + QK^T + softmax + PV in one launch, K/V never in HBM; eff_bert.py:277-364 with
    encoder_hidden_states) against the training path's two launches (packed K/V GEMM + MFMA attention through kv_index):
    the projection runs the same K loop and the tiles are rounded to bf16 at the same point, so context AND probability
    map must agree BIT FOR BIT - with key masks, head gates, shared images in arbitrary order and ragged query tiles"""
    o = ops()
    g = torch.Generator().manual_seed(71)
    dh, d = 64, H * 64
    Bq = Bimg * rows
    x = rnd((Bimg, N, d), torch.bfloat16, g, 0.6)
    q = rnd((Bq, Lq, d), torch.bfloat16, g, 0.6)
    Wk, Wv = [torch.nn.Parameter(rnd((d, d), torch.float32, g, 0.04), requires_grad=False) for _ in range(2)]
    bk, bv = [torch.nn.Parameter(rnd((d,), torch.float32, g, 0.2), requires_grad=False) for _ in range(2)]
    idx = None
    if rows > 1:
        idx = torch.arange(Bimg).repeat(rows)[torch.randperm(Bq, generator=g)].to(DEV)
    mask = torch.zeros(Bq, N)
    mask[1, N - 7:] = -10000.0
    mask[Bq - 1, :3] = -10000.0
    gate = (torch.rand(H, generator=g) + 0.5).to(DEV)
    with torch.no_grad():
        assert o.xattn_fusable(q, x, (Wk, Wv), H, dh)
        kv = o.linear_packed(x, (Wk, Wv), (bk, bv))
        O1, P1 = o.cross_attention(q, kv, H, dh, 0.125, mask=mask.to(DEV), gate=gate, want_probs=True, kv_index=idx)
        O2, P2 = o.cross_attention_fused(q, x, (Wk, Wv), (bk, bv), H, dh, 0.125, mask=mask.to(DEV), gate=gate,
                                         want_probs=True, kv_index=idx)
        O3, P3 = o.cross_attention_fused(q, x, (Wk, Wv), (bk, bv), H, dh, 0.125, mask=mask.to(DEV), gate=gate,
                                         want_probs=False, kv_index=idx)
    assert torch.equal(O1, O2) and torch.equal(P1, P2) and torch.equal(O1, O3) and P3 is None
    # and against plain fp32 math (the bf16 tolerance of the attention tests)
    kvr = (x.float() @ torch.cat([Wk, Wv], 0).t() + torch.cat([bk, bv]))
    kvr = kvr if idx is None else kvr[idx]
    sp = lambda t, Ln: t.reshape(Bq, Ln, H, dh).transpose(1, 2)
    S = sp(q.float(), Lq) @ sp(kvr[..., :d], N).transpose(-1, -2) * 0.125 + mask.to(DEV)[:, None, None, :]
    Pr = torch.softmax(S, -1)
    Or = ((Pr @ sp(kvr[..., d:], N)) * gate[None, :, None, None]).transpose(1, 2).reshape(Bq, Lq, d)
    assert rel_err(P2.float(), Pr) < tol(torch.bfloat16) and rel_err(O2.float(), Or) < 2 * tol(torch.bfloat16)
    # a forward that needs gradients is never routed to the fused kernel
    assert not o.xattn_fusable(q.clone().requires_grad_(True), x, (Wk, Wv), H, dh)


def _pp256_case(o, L, g, I, J, K, qt=0, bias=False, res=False, act=0, dact=0, sk=False, half=False, x192=False):
    dtype = torch.bfloat16
    Pm = rnd((I, K), dtype, g, 0.5)
    Qm = rnd((K, J) if qt else (J, K), dtype, g, 0.1)
    Cm = torch.full((I, J), 7.0, dtype=dtype, device=DEV)
    kw = {}
    b = r = h = aux = None
    if bias:
        b = rnd((J,), torch.float32, g); kw["bias"] = b
    if res:
        r = rnd((I, J), dtype, g); kw.update(residual=r, ldx=J)
    if act:
        h = torch.full((I, J), 3.0, dtype=dtype, device=DEV); kw.update(act=act, preact=h, ldx=J)
    if dact:
        aux = rnd((I, J), dtype, g); kw.update(dact=dact, aux=aux, ldx=J)
    o._gemm(L.dt(dtype), Pm, Qm, Cm, I, J, K, Pm.stride(0), Qm.stride(0), J, q_trans=qt, **kw)
    # the case must have been served by the kernel it is meant to test (a routing change would otherwise test another one)
    served = L.load().evlm_gemm_last_kernel().decode()
    qts = "true" if qt else "false"
    want = (f"gemm_bf16_pp256_sk_kernel<{qts}>" if sk else f"gemm_bf16_pp128_kernel<{qts}>" if half else
            f"gemm_bf16_pp192_kernel<{qts}>" if x192 else f"gemm_bf16_pp256_kernel<false,{qts},0>")
    assert served == want, (served, want)
    ref = Pm.float() @ (Qm.float() if qt else Qm.float().t())
    if bias:
        ref = ref + b
    pre = ref
    if act == L.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    if act == L.ACT_QUICK_GELU:
        ref = ref * torch.sigmoid(1.702 * ref)
    if dact == L.ACT_QUICK_GELU:
        x = aux.float(); s = torch.sigmoid(1.702 * x); ref = ref * (s + 1.702 * x * s * (1 - s))
    if dact == L.ACT_GELU:
        x = aux.float()
        ref = ref * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * math.pi) ** 0.5)
    if res:
        ref = ref + r.float()
    assert rel_err(Cm.float(), ref) < tol(dtype), (I, J, K, qt, bias, res, act, dact)
    if act:
        assert rel_err(h.float(), pre) < tol(dtype), ("preact", I, J, K)


def test_gemm_pp256_persistent_tiles_and_epilogues():
    """shapes routed to the 256x256 ping-pong kernel: several tiles per workgroup (persistent loop + next-tile prefetch),
    ragged last tiles in both dimensions, odd and minimal K-tile counts, every fused epilogue it supports, and the
    reduction-major Q operand (data gradients, transposing LDS reads)"""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(47)
    _pp256_case(o, L, g, 6144 + 40, 4096 + 8, 192, bias=True)                    # 425 tiles: 2 per workgroup, ragged edges
    _pp256_case(o, L, g, 5120, 3072, 128, bias=True, res=True)                   # 240 tiles, minimal K (2 K tiles)
    _pp256_case(o, L, g, 6144, 2560, 320, bias=True, act=L.ACT_QUICK_GELU)       # pre-activation second output, odd K tiles
    _pp256_case(o, L, g, 6148, 4096, 256, bias=True, act=L.ACT_GELU)
    _pp256_case(o, L, g, 4096, 4096, 256, dact=L.ACT_QUICK_GELU)
    _pp256_case(o, L, g, 6144 + 8, 4096, 192, dact=L.ACT_GELU)                   # text-side FFN backward (erf-GELU), ragged rows
    _pp256_case(o, L, g, 6144, 4096 + 16, 192, qt=1, dact=L.ACT_GELU)
    _pp256_case(o, L, g, 6144 + 24, 4096, 192, qt=1)                             # dX = dY W, W reduction-major
    _pp256_case(o, L, g, 12608, 2304, 768)                                       # ViT QKV: 450 tiles, 1.76 rounds


def test_gemm_pp128_half_tiles_for_thinly_filled_launches():
    """gemm_bf16_pp128_kernel (128 x 256 tiles, two-phase K tile, three staging stages): the text-side products whose
    256 x 256 tiles would fill well under half of one round.  Step shapes, ragged rows / columns, every K-tile count
    modulo 3 (the stage ring), minimal K, each epilogue flavour, both Q layouts"""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(61)
    _pp256_case(o, L, g, 7680, 768, 768, bias=True, res=True, half=True)         # fusion output projection: 180 half tiles
    _pp256_case(o, L, g, 7680, 768, 3072, bias=True, res=True, half=True)        # fusion FC2: 48 K tiles
    _pp256_case(o, L, g, 7680, 768, 2304, qt=1, half=True)                       # dX of a QKV projection (W reduction-major)
    _pp256_case(o, L, g, 7680, 768, 3072, qt=1, dact=L.ACT_GELU, half=True)      # FFN backward through erf-GELU
    _pp256_case(o, L, g, 5120 + 40, 768 + 16, 128, bias=True, half=True)         # ragged both ways, 2 K tiles (164 half tiles)
    _pp256_case(o, L, g, 6000, 1024, 192, bias=True, act=L.ACT_QUICK_GELU, half=True)    # 3 K tiles, pre-activation output
    _pp256_case(o, L, g, 6000, 1024, 256, bias=True, act=L.ACT_GELU, half=True)  # 4 K tiles
    _pp256_case(o, L, g, 6000, 1024, 320, dact=L.ACT_QUICK_GELU, half=True)      # 5 K tiles
    _pp256_case(o, L, g, 12288, 512, 448, bias=True, res=True, half=True)        # 192 half tiles, 7 K tiles
    _pp256_case(o, L, g, 16384, 256, 384, bias=True, half=True)                  # one tile column, 6 K tiles


def test_gemm_pp192_tiles_for_part_filled_rounds():
    """gemm_bf16_pp192_kernel (192 x 256 tiles, three-phase K tile, 96 x 64 wave blocks): products whose 256 x 256 tiles
    fill 50-80 % of one round.  Step shapes, ragged rows / columns, odd / even / minimal K-tile counts, each epilogue
    flavour (incl. the pre-activation second output), both Q layouts"""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(67)
    _pp256_case(o, L, g, 12608, 768, 768, bias=True, res=True, x192=True)        # ViT out-projection: 150 -> 198 tiles
    _pp256_case(o, L, g, 12608, 768, 3072, bias=True, res=True, x192=True)       # ViT FC2
    _pp256_case(o, L, g, 12608, 768, 2304, qt=1, x192=True)                      # dX of the QKV projection
    _pp256_case(o, L, g, 12608, 768, 3072, qt=1, dact=L.ACT_QUICK_GELU, x192=True)
    _pp256_case(o, L, g, 3840, 3072, 768, bias=True, act=L.ACT_GELU, x192=True)  # text FC1: 180 -> 240 tiles, pre-activation
    _pp256_case(o, L, g, 3840, 2304, 768, bias=True, x192=True)
    _pp256_case(o, L, g, 12544, 768, 768, bias=True, x192=True)
    _pp256_case(o, L, g, 7680, 2304, 768, bias=True, x192=True)                  # 270 full tiles (1.05 rounds) -> 360: two rounds
    _pp256_case(o, L, g, 12608, 1536, 768, bias=True, res=True, x192=True)       # 300 -> 396
    _pp256_case(o, L, g, 9000 + 8, 1024 + 16, 128, bias=True, x192=True)         # ragged both ways, 2 K tiles
    _pp256_case(o, L, g, 9000, 1024, 192, bias=True, act=L.ACT_QUICK_GELU, x192=True)    # 3 K tiles
    _pp256_case(o, L, g, 9000, 1024, 320, dact=L.ACT_GELU, x192=True)            # 5 K tiles


def _stream_k_cases():
    """body of test_gemm_pp256_stream_k_partial_rounds (runs in a child process with EVLM_PP256_SK=1)"""
    from efficientvlm_amd import ops as o, _lib as L
    g = torch.Generator().manual_seed(59)
    for rep in range(2):
        _pp256_case(o, L, g, 7680, 768, 768, bias=True, res=True, sk=True)       # 4B-row fusion output projection (90 tiles)
        _pp256_case(o, L, g, 12608, 768, 768, bias=True, res=True, sk=True)      # ViT out-projection (150 tiles, ragged rows)
        _pp256_case(o, L, g, 12608, 768, 3072, bias=True, res=True, sk=True)     # ViT FC2: 48 K tiles per tile
        _pp256_case(o, L, g, 7680, 768, 3072, bias=True, res=True, sk=True)
        _pp256_case(o, L, g, 3840, 768, 3072, bias=True, res=True, sk=True)      # 45 tiles: up to 7 workgroups per tile
        _pp256_case(o, L, g, 3840, 3072, 768, bias=True, act=L.ACT_GELU, sk=True)
        _pp256_case(o, L, g, 12608, 768, 2304, qt=1, sk=True)                    # dX of the QKV projection
        _pp256_case(o, L, g, 7680, 768, 3072, qt=1, dact=L.ACT_GELU, sk=True)
        _pp256_case(o, L, g, 1024 + 8, 1024 + 16, 512, bias=True, sk=True)       # 25 tiles, ragged both ways, 8 K tiles
        _pp256_case(o, L, g, 4096, 1024, 640, bias=True, act=L.ACT_QUICK_GELU, sk=True)   # 64 tiles, 10 K tiles
    # determinism: same inputs, same bits (fixed summation order)
    Pm, Qm = rnd((7680, 3072), torch.bfloat16, g, 0.5), rnd((768, 3072), torch.bfloat16, g, 0.1)
    outs = []
    for _ in range(3):
        Cm = torch.empty((7680, 768), dtype=torch.bfloat16, device=DEV)
        o._gemm(L.BF16, Pm, Qm, Cm, 7680, 768, 3072, 3072, 3072, 768)
        outs.append(Cm)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # two streams at once: each has its own workspace
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    res = {}
    for st in (s1, s2):
        with torch.cuda.stream(st):
            for k in range(4):
                Cm = torch.empty((7680, 768), dtype=torch.bfloat16, device=DEV)
                o._gemm(L.BF16, Pm, Qm, Cm, 7680, 768, 3072, 3072, 3072, 768)
                res[(st, k)] = Cm
    torch.cuda.synchronize()
    assert all(torch.equal(v, outs[0]) for v in res.values())


def test_gemm_pp256_stream_k_partial_rounds():
    """opt-in (EVLM_PP256_SK=1) stream-K form of the 256x256 kernel: launches whose tiles fill only part of one round are
    cut along K as well (gemm_bf16_pp256_sk_kernel) - partial accumulators through the per-stream workspace, the tile's
    last workgroup adds them in its epilogue.  Every step shape that would take this path, ragged edges, each epilogue
    flavour, both Q layouts; run twice (the kernel must leave the workspace flags clean); bit-identical from run to run;
    two streams at once.  In a child process: the switch is read once per process."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    # (round 6: the kernel is out of the default library - `make -C efficientvlm_amd/csrc EXPERIMENTAL=1 LIB=...` builds it)
    exp = os.path.join(os.path.dirname(here), "tools", "_build", "libevlm_exp.so")
    if not os.path.exists(exp):
        pytest.skip("experimental library (stream-K form of the 256-row GEMM) not built")
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_ops_gpu as t; t._stream_k_cases(); print('SK_OK')" \
        % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, EVLM_PP256_SK="1", EVLM_LIB=exp), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "SK_OK" in r.stdout, r.stderr[-3000:]


def test_gemm_pp256_weight_gradient_variant_subprocess():
    """the f32 / split-K / bias-gradient form of the 256x256 kernel is off by default (EVLM_PP256_WGRAD): exercise it in a
    child process against torch"""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from efficientvlm_amd import ops, _lib as L
torch.manual_seed(0)
for (I, J, K, acc) in [(768, 768, 2048, True), (520, 264, 1024, False), (3072, 768, 1600, True), (256, 256, 128, False)]:
    P = (torch.randn((K, I), device="cuda") * 0.5).bfloat16(); Q = (torch.randn((K, J), device="cuda") * 0.5).bfloat16()
    C = torch.full((I, J), 0.25 if acc else 9.0, dtype=torch.float32, device="cuda")
    ps = torch.zeros(I, dtype=torch.float32, device="cuda")
    ops._gemm(L.BF16, P, Q, C, I, J, K, I, J, J, p_trans=1, q_trans=1, c_f32=1, psum=ps, accumulate=int(acc))
    ref = P.float().t() @ Q.float() + (0.25 if acc else 0.0)
    e = float((C - ref).norm() / ref.norm()); ep = float((ps - P.float().sum(0)).norm() / P.float().sum(0).norm())
    assert e < 2e-5 and ep < 2e-5, (I, J, K, acc, e, ep)
print("PP256_WGRAD_OK")
''' % repo
    env = dict(os.environ, EVLM_PP256_WGRAD="1", EVLM_PP256_PCT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PP256_WGRAD_OK" in r.stdout, r.stderr[-2000:]


def test_grouped_weight_gradients_match_per_layer_products():
    """evlm_wgrad_grouped (one persistent launch, every output tile owned by one workgroup) against torch, including ragged
    tiles, the bias-gradient row sums, accumulation into pre-filled C and two problems adding into the SAME C"""
    o = ops()
    from efficientvlm_amd import _lib as L
    g = torch.Generator().manual_seed(53)
    K = 1600
    shapes = [(768, 768), (2304, 768), (520, 264), (768, 3072), (256, 256)]
    probs, keep = [], []
    for (I, J) in shapes:
        dY = rnd((K, I), torch.bfloat16, g, 0.5); X = rnd((K, J), torch.bfloat16, g, 0.5)
        Cm = torch.full((I, J), 0.25, dtype=torch.float32, device=DEV); ps = torch.full((I,), 0.5, dtype=torch.float32, device=DEV)
        probs.append((dY, X, Cm, ps)); keep.append((dY, X))
    dY2 = rnd((K, 768), torch.bfloat16, g, 0.5); X2 = rnd((K, 768), torch.bfloat16, g, 0.5)

    def launch(plist):
        arr = (L.WgradProblem * len(plist))()
        for k, (dY, X, Cm, ps) in enumerate(plist):
            arr[k].P, arr[k].Q, arr[k].C = dY.data_ptr(), X.data_ptr(), Cm.data_ptr()
            arr[k].psum = ps.data_ptr() if ps is not None else None
            arr[k].I, arr[k].J, arr[k].ldp, arr[k].ldq, arr[k].ldc = dY.shape[1], X.shape[1], dY.shape[1], X.shape[1], X.shape[1]
        L.check(L.load().evlm_wgrad_grouped(arr, len(plist), K, L.stream()), "wgrad_grouped")

    launch(probs + [(dY2, X2, probs[0][2], None)])       # last problem adds into the first problem's C (atomic mode)
    for n, (dY, X, Cm, ps) in enumerate(probs):
        ref = dY.float().t() @ X.float() + 0.25
        if n == 0:
            ref = ref + dY2.float().t() @ X2.float()
        assert rel_err(Cm, ref) < 2e-5, ("C", n)
        assert rel_err(ps, dY.float().sum(0) + 0.5) < 2e-5, ("psum", n)
    for (_, _, Cm, ps) in probs:
        Cm.fill_(0.25); ps.fill_(0.5)
    launch(probs)                                          # single owner per C: load / add / store mode
    for n, (dY, X, Cm, ps) in enumerate(probs):
        assert rel_err(Cm, dY.float().t() @ X.float() + 0.25) < 2e-5, ("C rmw", n)


def test_grouped_transposes_and_weight_transposes_follow_the_optimiser():
    """evlm_transpose_grouped (ragged edge tiles included) and the W^T copies FlatAdamW keeps for dX = dY W"""
    o = ops()
    from efficientvlm_amd import _lib as L
    from efficientvlm_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(61)
    mats = [rnd((r, c), torch.bfloat16, g) for r, c in ((768, 768), (2304, 768), (72, 136), (8, 200))]
    outs = [torch.empty((m.shape[1], m.shape[0]), dtype=torch.bfloat16, device=DEV) for m in mats]
    rows, tiles = [], 0
    for m, t in zip(mats, outs):
        rows += [m.data_ptr(), t.data_ptr(), m.shape[0], m.shape[1], tiles]
        tiles += ((m.shape[0] + 63) // 64) * ((m.shape[1] + 63) // 64)
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    L.check(L.load().evlm_transpose_grouped(L.ptr(table), len(mats), tiles, L.stream()), "transpose")
    for m, t in zip(mats, outs):
        assert torch.equal(t, m.t().contiguous())

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v = (torch.nn.Linear(128, 192) for _ in range(3))
            self.fc = torch.nn.Linear(192, 128)
    net = Net().to(DEV)
    opt = FlatAdamW(net, lr=1e-2, weight_decay=0.0, lr_mult=1.0, max_grad_norm=0.0)
    packed = (net.q.weight, net.k.weight, net.v.weight)
    for step in range(2):
        wt = o.CACHE.get_t(packed)
        assert wt is not None and wt.shape == (128, 576)
        ref = torch.cat([p.detach() for p in packed], 0).to(torch.bfloat16).t().contiguous()
        assert torch.equal(wt, ref), step
        assert torch.equal(o.CACHE.get_t((net.fc.weight,)), net.fc.weight.detach().to(torch.bfloat16).t().contiguous())
        x = rnd((64, 128), torch.bfloat16, g).requires_grad_(True)
        opt.zero_grad()
        o.WGRAD_INPLACE = True
        try:
            y = o.linear(o.linear_packed(x, packed, (net.q.bias, net.k.bias, net.v.bias))[..., :192], net.fc.weight, net.fc.bias)
            y.float().square().mean().backward()
        finally:
            o.WGRAD_INPLACE = False
        W = torch.cat([p.detach() for p in packed], 0).to(torch.bfloat16).float()
        assert x.grad is not None and torch.isfinite(x.grad.float()).all()
        opt.set_schedule(1.0)
        opt.step()                       # parameters move -> the copies must follow


# ---- ITM hard-negative sampler (evlm_sample_negatives; reference efficient_models/xvlm.py:422-458) -------------------
def _neg_weights(sim, temp, group=None):
    """the reference's sampling weights, in float64: softmax(sim / temp) + 1e-5 with the positives zeroed"""
    s = sim.double().cpu().numpy() / temp
    B = s.shape[0]

    def w(mat):
        e = np.exp(mat - mat.max(1, keepdims=True))
        p = e / e.sum(1, keepdims=True) + 1e-5
        if group is None:
            p[np.arange(B), np.arange(B)] = 0
        else:
            g = np.asarray(group)
            p[g[:, None] == g[None, :]] = 0
        return p / p.sum(1, keepdims=True)
    return w(s.T), w(s)           # per text over images, per image over texts


def test_negative_sampler_draws_from_the_reference_distribution():
    ops = globals()['ops']()
    torch.manual_seed(3)
    B, temp, n = 6, 0.7, 6000
    sim = (torch.randn(B, B, device="cuda") * 1.5).contiguous()
    t = torch.tensor(temp, device="cuda")
    ops.dropout_seed(11)
    draws = []
    for _ in range(n):
        draws.append(ops.sample_negatives(sim, t))
        ops.dropout_tick()
    d = torch.stack(draws).cpu().numpy()                                   # [n, 2B]
    p_t2i, p_i2t = _neg_weights(sim, temp)
    for row in range(2 * B):
        p = p_t2i[row] if row < B else p_i2t[row - B]
        freq = np.bincount(d[:, row], minlength=B) / n
        assert freq[row % B] == 0                                          # the positive is never drawn
        assert np.all(np.abs(freq - p) < 4.5 * np.sqrt(p * (1 - p) / n) + 1e-3), (row, freq, p)


def test_negative_sampler_groups_determinism_and_long_rows():
    ops = globals()['ops']()
    torch.manual_seed(4)
    B = 300                                                                # several 64-wide chunks per row
    sim = torch.randn(B, B, device="cuda")
    group = torch.arange(B, device="cuda") // 3                            # triples of mutual positives
    t = torch.tensor(0.5, device="cuda")
    ops.dropout_seed(5)
    call0 = ops._DROP_CALL[0]
    a = ops.sample_negatives(sim, t, group)
    assert a.dtype == torch.int64 and a.shape == (2 * B,) and int(a.min()) >= 0 and int(a.max()) < B
    rows = torch.arange(B, device="cuda").repeat(2)
    assert not bool((group[a] == group[rows]).any())                       # never a member of the own group
    ops._DROP_CALL[0] = call0                                              # same (seed, step, call id) -> same draw
    assert torch.equal(ops.sample_negatives(sim, t, group), a)
    ops.dropout_tick()
    ops._DROP_CALL[0] = call0
    assert not torch.equal(ops.sample_negatives(sim, t, group), a)         # next step: new draws
    # one overwhelming candidate per row / column, placed beyond the first chunk
    sim2 = torch.zeros(B, B, device="cuda")
    tgt = (torch.arange(B, device="cuda") + 170) % B
    sim2[torch.arange(B), tgt] = 60.0                                      # image i -> text tgt[i]
    out = ops.sample_negatives(sim2, torch.tensor(1.0, device="cuda"))
    # (the +1e-5 floor leaves the other 298 candidates 0.3 % of the mass: ~2 of the 600 draws may land elsewhere)
    assert int((out[B:] != tgt).sum()) <= 8
    inv = torch.empty_like(tgt); inv[tgt] = torch.arange(B, device="cuda")
    assert int((out[:B] != inv).sum()) <= 8                                # text t -> the image whose target it is
    # a padded (strided) similarity buffer, as _matmul_nt returns it
    buf = torch.zeros(B, B + 4, device="cuda"); buf[:, :B] = sim2
    ops._DROP_CALL[0] -= 1                                                 # the same random numbers as for `out`
    assert torch.equal(ops.sample_negatives(buf[:, :B], torch.tensor(1.0, device="cuda")), out)


def test_model_forward_samples_its_negatives_with_the_device_sampler():
    """no injected indices: XVLM._sample_negatives -> evlm_sample_negatives; admissible, positive-free, fresh per step"""
    from helpers import load_fixture, batch_from_fixture, model_config
    from oracle import synth
    from efficientvlm_amd.models.model_pretrain import XVLM
    from efficientvlm_amd.runtime import compute
    ops = globals()['ops']()
    fx = load_fixture("gd_tiny.npz")
    geom = synth.GEOMS["tiny"]
    torch.manual_seed(0)
    model = XVLM(model_config(geom, "s")).to("cuda").eval()
    batch = batch_from_fixture(fx, "cuda")
    B = batch["image"].shape[0]
    ops.dropout_seed(9)
    seen = set()
    for _ in range(12):
        with torch.no_grad(), compute(torch.float32):
            model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                  masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], output_attentions=True,
                  output_hidden_states=True)
        neg = model.last_neg_idx.cpu()
        assert neg.shape == (2 * B,) and int(neg.min()) >= 0 and int(neg.max()) < B
        assert not bool((neg == torch.arange(B).repeat(2)).any())
        seen.add(tuple(neg.tolist()))
        ops.dropout_tick()
    assert B <= 2 or len(seen) > 1


@pytest.mark.parametrize("B,H,L", [(2, 3, 17), (2, 2, 30), (1, 2, 33), (2, 2, 64), (1, 3, 197), (2, 2, 208), (1, 2, 224)])
@pytest.mark.parametrize("kd", [False, True])
@pytest.mark.parametrize("store_p", [False, True])
def test_single_pass_attention_backward_equals_the_two_kernel_path(B, H, L, kd, store_p, monkeypatch):
    """attn_bwd_fused_kernel (dS kept in LDS, one launch) against kernels A + B (dS through HBM): same arithmetic in the
    same order - identical packed gradients; and both against the fp32 reference.  store_p False: the recomputing form of
    both (P rebuilt from Q, K and the saved row lse in fp32; kernel B reads the bf16 copy kernel A writes)."""
    o = ops()
    monkeypatch.setattr(o, "ATTN_STORE_P", store_p)
    dh, d = 64, H * 64
    g = torch.Generator().manual_seed(300 + L)
    base = rnd((B, L, 3 * d), torch.bfloat16, g)
    gate0 = torch.rand(1, H, 1, 1, generator=g).to(DEV)
    mask = torch.zeros(B, L); mask[0, L - 3:] = -10000.0
    mask = mask.to(DEV)
    scale = dh ** -0.5
    with torch.no_grad():
        _, Pt = o.self_attention(rnd((B, L, 3 * d), torch.bfloat16, g), H, dh, scale, mask=mask)
    gO = rnd((B, L, d), torch.bfloat16, g)
    gP = rnd((B, H, L, L), torch.bfloat16, g, 0.1)

    def run(split):
        monkeypatch.setenv("EVLM_ATTN_BWD_SPLIT", "1" if split else "0")
        qkv = base.clone().requires_grad_(True)
        gate = gate0.clone().requires_grad_(True)
        if kd:
            O, P, term = o.self_attention(qkv, H, dh, scale, mask=mask, gate=gate, kd_teacher=Pt, kd_weight=3.0)
            loss = (O.float() * gO.float()).sum() + term * 0.7
        else:
            O, P = o.self_attention(qkv, H, dh, scale, mask=mask, gate=gate)
            loss = (O.float() * gO.float()).sum() + (P.float() * gP.float()).sum()
        loss.backward()
        return qkv.grad.clone(), gate.grad.clone()

    (ga, gga), (gb, ggb) = run(False), run(True)
    assert torch.equal(ga, gb)
    assert rel_err(gga, ggb) < 1e-5                     # (atomics: order of the per-wave partial sums)
    # fp32 reference
    xr = base.float().requires_grad_(True)
    gr = gate0.clone().requires_grad_(True)
    sp = lambda t: t.view(B, L, H, dh).transpose(1, 2)
    Or, Pr = _ref_attention(sp(xr[..., :d]), sp(xr[..., d:2 * d]), sp(xr[..., 2 * d:]), mask, gr, scale)
    Or = Or.transpose(1, 2).reshape(B, L, d)
    if kd:
        lossr = (Or * gO.float()).sum() + 0.7 * 3.0 * (Pr - Pt.float()).pow(2).mean()
    else:
        lossr = (Or * gO.float()).sum() + (Pr * gP.float()).sum()
    lossr.backward()
    assert rel_err(ga.float(), xr.grad) < (5e-2 if store_p else 2e-2)
    assert rel_err(gga, gr.grad) < 5e-2


def _attn_problem(B, Bkv, H, Lq, Lk, seed, self_attn):
    dh, d = 64, H * 64
    g = torch.Generator().manual_seed(seed)
    if self_attn:
        x = rnd((B, Lq, 3 * d), torch.bfloat16, g)
        kv = None
    else:
        x = rnd((B, Lq, d), torch.bfloat16, g)
        kv = rnd((Bkv, Lk, 2 * d), torch.bfloat16, g)
    mask = torch.zeros(B, Lk); mask[0, Lk - 3:] = -10000.0
    idx = (torch.arange(B) * 5 % Bkv).to(DEV) if (not self_attn and Bkv != B) else None
    gO = rnd((B, Lq, d), torch.bfloat16, g)
    return x, kv, mask.to(DEV), idx, gO, dh, d


@pytest.mark.parametrize("case", ["vit197", "text30", "cross197", "cross_shared", "causal40", "vit577", "vit901", "cross577"])
def test_recomputing_attention_backward_needs_no_stored_map_and_tightens_the_gradients(case, monkeypatch):
    """The recomputing form (bf16, head dim 64, no dropout: Lk <= 224 always; 417..928 keys - the long-sequence kernel, ONE
    pass per key half with the row sums taken from dO . O - when the caller does not take the map): with want_probs=False NO [B, H, Lq, Lk]
    map exists in HBM (the forward returns None, the backward rebuilds P from Q, K and the saved row lse in fp32) and the
    gradients sit closer to the fp32 reference than those formed from the stored bf16 map (the round-2 form,
    ATTN_STORE_P) - the query / key gradient is the cancellation P .* (dP - delta)."""
    o = ops()
    B, Bkv, H, Lq, Lk, self_attn, causal = {"vit197": (3, 3, 12, 197, 197, True, False), "text30": (4, 4, 12, 30, 30, True, False),
                                            "cross197": (3, 3, 12, 30, 197, False, False),
                                            "cross_shared": (7, 3, 12, 30, 197, False, False),
                                            "causal40": (2, 2, 12, 40, 40, True, True),
                                            "vit577": (2, 2, 12, 577, 577, True, False), "vit901": (1, 1, 4, 901, 901, True, False),
                                            "cross577": (5, 2, 12, 30, 577, False, False)}[case]
    x0, kv0, mask, idx, gO, dh, d = _attn_problem(B, Bkv, H, Lq, Lk, 900 + Lq + Lk, self_attn)
    scale = dh ** -0.5

    def run(store_p, want):
        monkeypatch.setattr(o, "ATTN_STORE_P", store_p)
        x = x0.clone().requires_grad_(True)
        kv = kv0.clone().requires_grad_(True) if kv0 is not None else None
        if self_attn:
            O, P = o.self_attention(x, H, dh, scale, mask=mask, want_probs=want, causal=causal)
        else:
            O, P = o.cross_attention(x, kv, H, dh, scale, mask=mask, want_probs=want, kv_index=idx)
        (O.float() * gO.float()).sum().backward()
        return O.detach(), P, x.grad.float(), (kv.grad.float() if kv is not None else None)

    O_rc, P_rc, gx_rc, gkv_rc = run(False, False)
    assert P_rc is None                                   # nothing materialised
    O_rcp, P_rcp, gx_rcp, gkv_rcp = run(False, True)      # the map on request: same context ...
    O_st, P_st, gx_st, gkv_st = run(True, True)
    assert P_rcp is not None
    if Lk <= 224:                                         # ... and the same (recomputed) gradients;
        assert torch.equal(O_rc, O_rcp) and torch.equal(O_rc, O_st)
        assert torch.equal(gx_rc, gx_rcp)
    else:                                                 # on long key sequences a caller who takes the map keeps the stored-map
        assert torch.equal(gx_rcp, gx_st)                 # backward (an external dP may come back for it: two passes either way)
        # and the whole-row forward kernel that writes it; without the map the STREAMING kernel runs (online softmax over
        # 128-key blocks, 1 / l applied to the context): the same context at bf16 resolution, not the same bits
        assert torch.equal(O_rcp, O_st)
        assert rel_err(O_rc.float(), O_st.float()) < 1e-2
    assert torch.equal(P_rcp, P_st)
    # fp32 reference
    xr = x0.float().requires_grad_(True)
    sp = lambda t, Ln: t.reshape(t.shape[0], Ln, H, dh).transpose(1, 2)
    if self_attn:
        q, k, v = sp(xr[..., :d], Lq), sp(xr[..., d:2 * d], Lk), sp(xr[..., 2 * d:], Lk)
        kvr = None
    else:
        kvr = kv0.float().requires_grad_(True)
        kvg = kvr[idx] if idx is not None else kvr
        q, k, v = sp(xr, Lq), sp(kvg[..., :d], Lk), sp(kvg[..., d:], Lk)
    add = mask[:, None, None, :]
    if causal:
        tri = torch.tril(torch.ones(Lq, Lk, device=DEV))
        add = (1.0 - tri[None, None] * (mask == 0).float()[:, None, None, :]) * -10000.0
    Pr = torch.softmax(q @ k.transpose(-1, -2) * scale + add, -1)
    Or = (Pr @ v).transpose(1, 2).reshape(B, Lq, d)
    (Or * gO.float()).sum().backward()
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    e_rc, e_st = l2(gx_rc, xr.grad), l2(gx_st, xr.grad)
    assert rel_err(gx_rc, xr.grad) < 1.2e-2 and e_rc < 6e-3, (e_rc, e_st)
    if self_attn:                                          # the query / key part: P .* (dP - delta), where the bf16 map costs
        qk_rc, qk_st = l2(gx_rc[..., :2 * d], xr.grad[..., :2 * d]), l2(gx_st[..., :2 * d], xr.grad[..., :2 * d])
        assert qk_rc < 6e-3 and qk_rc <= qk_st * 1.02, (qk_rc, qk_st)
    if kvr is not None:
        k_rc, k_st = l2(gkv_rc, kvr.grad), l2(gkv_st, kvr.grad)
        assert rel_err(gkv_rc, kvr.grad) < 1.2e-2 and k_rc < 6e-3, (k_rc, k_st)


@pytest.mark.parametrize("B,H,L,with_kd", [(2, 12, 577, True), (1, 4, 901, True), (2, 4, 577, False), (2, 3, 450, True)])
def test_one_pass_long_sequence_backward_with_fused_distillation_and_gates(B, H, L, with_kd):
    """attn_bwd_dq_long_kernel in its ONE-PASS form (384 x 384 / 480 x 480 images: 577 / 901 keys; nobody takes the map):
    probabilities rebuilt from Q, K and the row lse in fp32, the row sums delta = sum_k P dP taken from dO . O (+ the
    distillation term's share, which the forward kernel leaves as kd_rowdot), the map-distillation dP formed in fp32 from
    the teacher map, head gates and their gradient collected in the same pass - against plain fp32 autograd of
    softmax / P V / gate / MSELoss(P, P_t) * L (eff_vit.py:144-195, GeneralDistill.py:63-69), with a key-padding mask"""
    o = ops()
    g = torch.Generator().manual_seed(300 + L)
    dh, d = 64, H * 64
    qkv0 = rnd((B, L, 3 * d), torch.bfloat16, g, 0.7)
    mask = torch.zeros(B, L)
    mask[0, L - 9:] = -10000.0
    mask = mask.to(DEV)
    gate0 = (torch.rand(H, generator=g) + 0.5)
    gate0[0] = 0.0                                        # a closed head: its context and dQ / dK / dV vanish, its gate gradient does not
    gate0 = gate0.to(DEV)
    with torch.no_grad():
        _, Pt = o.self_attention(rnd((B, L, 3 * d), torch.bfloat16, g, 0.7), H, dh, 0.125, mask=mask)     # a "teacher" map
    gO = rnd((B, L, d), torch.bfloat16, g)
    coef = 0.3
    x = qkv0.clone().requires_grad_(True)
    gate = gate0.clone().requires_grad_(True)
    if with_kd:
        O, P, kd = o.self_attention(x, H, dh, 0.125, mask=mask, gate=gate, want_probs=False, kd_teacher=Pt, kd_weight=float(L))
        ((O.float() * gO.float()).sum() + coef * kd).backward()
    else:
        O, P = o.self_attention(x, H, dh, 0.125, mask=mask, gate=gate, want_probs=False)
        (O.float() * gO.float()).sum().backward()
    assert P is None
    xr = qkv0.float().requires_grad_(True)
    gr = gate0.clone().requires_grad_(True)
    sp = lambda t: t.reshape(B, L, H, dh).transpose(1, 2)
    Pr = torch.softmax(sp(xr[..., :d]) @ sp(xr[..., d:2 * d]).transpose(-1, -2) * 0.125 + mask[:, None, None, :], -1)
    Or = ((Pr @ sp(xr[..., 2 * d:])) * gr[None, :, None, None]).transpose(1, 2).reshape(B, L, d)
    loss = (Or * gO.float()).sum()
    if with_kd:
        kdr = torch.nn.functional.mse_loss(Pr, Pt.float()) * L
        loss = loss + coef * kdr
        assert rel_err(kd, kdr) < 1e-4
    loss.backward()
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel_err(O.float(), Or) < 2 * tol(torch.bfloat16)
    assert l2(x.grad.float(), xr.grad) < 8e-3, l2(x.grad.float(), xr.grad)
    assert l2(x.grad.float()[..., :2 * d], xr.grad[..., :2 * d]) < 1e-2            # query / key part
    if not with_kd:                                                                 # the closed head (with the distillation
        assert float(x.grad.float()[..., :64].abs().max()) == 0.0                   # term its map still has a gradient)
    assert l2(gate.grad, gr.grad) < 5e-3, (gate.grad, gr.grad)


@pytest.mark.parametrize("B,H,L,tq", [(2, 12, 577, 0), (1, 4, 901, 0), (2, 3, 450, 1), (2, 12, 577, 2), (3, 2, 420, 0)])
def test_fused_map_distillation_from_a_teacher_recipe_equals_the_fp32_reference(B, H, L, tq, monkeypatch):
    """ABI 8 (round 5): the fused attention-map distillation of a long key sequence with the teacher's map REBUILT in the
    student's streaming kernels from the teacher's Q, K and row lse (ops.MapRecipe, evlm_attn_*_args.kd_tq / kd_tk /
    kd_tlse) instead of read from a [B, H, L, L] bf16 map: forward term, context and the q / k / v / gate gradients against
    plain fp32 autograd of softmax / P V / gate / MSELoss(P, P_t) * L with P_t = the TEACHER'S fp32 softmax (the recipe's
    probabilities are not bf16-rounded), and against the stored-map form of the same kernels at bf16 resolution.
    tq: EVLM_ATTN_STREAM_TQ (query tiles per wave of the forward kernel; 0 = the dispatcher's choice)."""
    import subprocess, sys
    if tq and os.environ.get("EVLM_ATTN_STREAM_TQ") != str(tq):
        # (the switch is read once per process)
        env = dict(os.environ, EVLM_ATTN_STREAM_TQ=str(tq))
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                            f"teacher_recipe and {B}-{H}-{L}-{tq}"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        return
    o = ops()
    g = torch.Generator().manual_seed(400 + L)
    dh, d = 64, H * 64
    qkv0 = rnd((B, L, 3 * d), torch.bfloat16, g, 0.7)
    tqkv = rnd((B, L, 3 * d), torch.bfloat16, g, 0.7)
    gate0 = (torch.rand(H, generator=g) + 0.5).to(DEV)
    gO = rnd((B, L, d), torch.bfloat16, g)
    coef = 0.3
    with torch.no_grad():
        assert o.map_recipe_supported(tqkv, H, dh)
        Ot, recipe = o.self_attention_recipe(tqkv, H, dh, 0.125)
        Ot2, Pt = o.self_attention(tqkv, H, dh, 0.125)                    # the same teacher, map written
    assert isinstance(recipe, o.MapRecipe) and tuple(recipe.shape) == (B, H, L, L) and rel_err(Ot.float(), Ot2.float()) < 1e-2

    def run(kd_teacher):
        x = qkv0.clone().requires_grad_(True)
        gate = gate0.clone().requires_grad_(True)
        O, P, kd = o.self_attention(x, H, dh, 0.125, gate=gate, want_probs=False, kd_teacher=kd_teacher, kd_weight=float(L))
        assert P is None
        ((O.float() * gO.float()).sum() + coef * kd).backward()
        return O.detach(), kd.detach(), x.grad, gate.grad

    O1, kd1, gx1, gg1 = run(recipe)
    O2, kd2, gx2, gg2 = run(Pt)
    sp = lambda t: t.reshape(B, L, H, dh).transpose(1, 2)
    with torch.no_grad():
        Ptr = torch.softmax(sp(tqkv.float()[..., :d]) @ sp(tqkv.float()[..., d:2 * d]).transpose(-1, -2) * 0.125, -1)
    xr = qkv0.float().requires_grad_(True)
    gr = gate0.clone().requires_grad_(True)
    Pr = torch.softmax(sp(xr[..., :d]) @ sp(xr[..., d:2 * d]).transpose(-1, -2) * 0.125, -1)
    Or = ((Pr @ sp(xr[..., 2 * d:])) * gr[None, :, None, None]).transpose(1, 2).reshape(B, L, d)
    kdr = torch.nn.functional.mse_loss(Pr, Ptr) * L
    ((Or * gO.float()).sum() + coef * kdr).backward()
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert torch.equal(O1, O2)                                   # (the context does not depend on the form of the term)
    assert rel_err(kd1, kdr) < 2e-4, (float(kd1), float(kdr))   # fp32 teacher probabilities: the reference's own operands
    assert rel_err(kd2, kdr) < 2e-2                              # (the stored map is bf16-rounded)
    assert l2(gx1.float(), xr.grad) < 8e-3 and l2(gx1.float()[..., :2 * d], xr.grad[..., :2 * d]) < 1e-2
    assert l2(gx1.float(), gx2.float()) < 1e-2
    assert l2(gg1, gr.grad) < 5e-3 and l2(gg1, gg2) < 5e-3


@pytest.mark.parametrize("B,Bkv,Lq,Lk,with_mask", [(7, 3, 30, 197, False), (256, 64, 30, 197, True), (9, 2, 17, 100, True),
                                                   (5, 4, 64, 224, False), (700, 90, 30, 197, False), (40, 3, 40, 150, True)])
def test_grouped_cross_attention_forward_is_bit_identical_to_the_per_batch_kernel(B, Bkv, Lq, Lk, with_mask, monkeypatch):
    """the grouped cross-attention forward kernels - attn_fwd_grouped_kernel (one workgroup per (K/V row, head) item: what
    ships) and attn_fwd_grouped_persist_kernel (round 5: one workgroup per CU walks the items with double-buffered K / V;
    EVLM_ATTN_GROUP_PERSIST=1 in a `make EXPERIMENTAL=1` library - in the default build the switch selects nothing and the
    "persistent" leg below repeats the per-item one) - against the per-batch kernel: same arithmetic per (batch, head, query) - identical
    context, map and row lse; a K/V row nobody attends to, uneven sharing, more items than CUs (1 080: several per
    workgroup) and waves with several tasks per item (40 batches on 3 rows) included; forward + backward through all"""
    o = ops()
    H, dh = 12, 64
    d = H * dh
    g = torch.Generator().manual_seed(77 + B)
    q0 = rnd((B, Lq, d), torch.bfloat16, g)
    kv0 = rnd((Bkv, Lk, 2 * d), torch.bfloat16, g)
    idx = torch.randint(0, max(1, Bkv - 1), (B,), generator=g).to(DEV)          # the last K/V row is never used
    mask = None
    if with_mask:
        mask = torch.zeros(B, Lk)
        mask[::3, Lk - 5:] = -10000.0
        mask = mask.to(DEV)
    gO = rnd((B, Lq, d), torch.bfloat16, g)

    def run(form, want):
        monkeypatch.setenv("EVLM_ATTN_NO_GROUP", "1" if form == "per-batch" else "0")
        monkeypatch.setenv("EVLM_ATTN_GROUP_PERSIST", "0" if form == "per-item" else "1")
        q, kv = q0.clone().requires_grad_(True), kv0.clone().requires_grad_(True)
        O, P = o.cross_attention(q, kv, H, dh, 0.125, mask=mask, want_probs=want, kv_index=idx)
        (O.float() * gO.float()).sum().backward()
        return O.detach(), (P.detach() if P is not None else None), q.grad, kv.grad

    for want in (True, False):
        b = run("per-batch", want)
        for form in ("persistent", "per-item"):
            a = run(form, want)
            assert torch.equal(a[0], b[0]), form
            assert (a[1] is None) == (b[1] is None) and (a[1] is None or torch.equal(a[1], b[1])), form
            assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), form     # (same lse -> same recomputed probabilities)
    # no-grad launches (no lse), with a closed head: its context is zero and its K / V are never staged
    gate = torch.ones(H, device=DEV)
    gate[1] = 0.0
    outs = []
    with torch.no_grad():
        for form in ("persistent", "per-item", "per-batch"):
            monkeypatch.setenv("EVLM_ATTN_NO_GROUP", "1" if form == "per-batch" else "0")
            monkeypatch.setenv("EVLM_ATTN_GROUP_PERSIST", "0" if form == "per-item" else "1")
            outs.append(o.cross_attention(q0, kv0, H, dh, 0.125, mask=mask, want_probs=True, kv_index=idx))
            outs.append(o.cross_attention(q0, kv0, H, dh, 0.125, mask=mask, gate=gate, want_probs=False, kv_index=idx))
    for k in (2, 4):
        assert torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1])
        assert torch.equal(outs[1][0], outs[k + 1][0])
    assert float(outs[1][0][..., 64:128].abs().max()) == 0.0 and float(outs[1][0][..., :64].abs().max()) > 0.0


@pytest.mark.parametrize("n,Bimg,rows,Lq,Lk,H", [(3, 5, 4, 30, 197, 12), (6, 3, 1, 17, 100, 4), (2, 4, 3, 30, 197, 12)])
def test_merged_kv_projection_equals_one_projection_per_layer(n, Bimg, rows, Lq, Lk, H):
    """ops.merged_kv (ONE K/V product of the image tokens for the n fusion layers of an encoder, every layer's attention
    reading / differentiating its columns of the merged buffers) against n separate packed projections
    (eff_bert.py:284-296 per layer): the forward is bit-identical (an output column's reduction does not depend on how many
    columns the product has), the weight / bias gradients agree to f32 rounding (dY^T X per column block); the gradient of the image
    tokens is ONE product over n * 2d columns instead of n products added in bf16 - compared at bf16 resolution.  A consumer
    that stays out of backward is an error, not a silently uninitialised column block."""
    o = ops()
    from efficientvlm_amd.runtime import compute
    dh, d, K = 64, H * 64, 768
    g = torch.Generator().manual_seed(500 + n)
    x0 = rnd((Bimg, Lk, K), torch.bfloat16, g)
    Ws = [torch.nn.Parameter((torch.randn(d, K, generator=g) * 0.03).to(DEV)) for _ in range(2 * n)]
    bs = [torch.nn.Parameter((torch.randn(d, generator=g) * 0.1).to(DEV)) for _ in range(2 * n)]
    qs = [rnd((Bimg * rows, Lq, d), torch.bfloat16, g) for _ in range(n)]
    gOs = [rnd((Bimg * rows, Lq, d), torch.bfloat16, g) for _ in range(n)]
    idx = (torch.arange(Bimg).repeat(rows)[torch.randperm(Bimg * rows, generator=g)]).to(DEV) if rows > 1 else None

    def run(merged, skip_last=False):
        for p_ in Ws + bs:
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        ql = [q.clone().requires_grad_(True) for q in qs]
        with compute(torch.bfloat16):
            if merged:
                kv, slot = o.merged_kv(x, Ws, bs, n)
                assert slot is not None and kv.shape[-1] == 2 * d * n
                outs = [o.cross_attention(ql[i], kv, H, dh, 0.125, want_probs=False, kv_index=idx, kv_col=2 * d * i, kv_grad=slot)[0]
                        for i in range(n)]
            else:
                outs = [o.cross_attention(ql[i], o.linear_packed(x, (Ws[2 * i], Ws[2 * i + 1]), (bs[2 * i], bs[2 * i + 1])), H, dh,
                                          0.125, want_probs=False, kv_index=idx)[0] for i in range(n)]
            use = outs[:-1] if skip_last else outs
            sum((O.float() * gO.float()).sum() for O, gO in zip(use, gOs)).backward()
        return ([O.detach() for O in outs], x.grad, [q.grad for q in ql[:len(use)]], [p_.grad.clone() for p_ in Ws + bs])

    a, b = run(True), run(False)
    for Oa, Ob in zip(a[0], b[0]):
        assert torch.equal(Oa, Ob)
    for qa, qb in zip(a[2], b[2]):
        assert torch.equal(qa, qb)
    for ga, gb in zip(a[3], b[3]):          # (f32 split reductions: the number of splits follows the tile count)
        assert rel_err(ga, gb) < 1e-5
    assert rel_err(a[1], b[1]) < 2e-2       # one product over n * 2d columns against n products added in bf16
    with pytest.raises(RuntimeError, match="merged K/V projection"):
        run(True, skip_last=True)


def test_additive_mask_is_one_gather_with_the_reference_bits():
    """(1 - m) * -10000 (eff_bert.py:953-1013) from a two-entry table: the same bits, -0.0 where the mask is 1"""
    o = ops()
    m = (torch.rand(7, 30) > 0.3).long().to(DEV)
    for view in (m[:, None, None, :], m[:, None, :, None].expand(7, 1, 30, 30), m.bool()[:, None, None, :], m.int()):
        got = o.additive_mask(view)
        want = (1.0 - view.to(torch.float32)) * -10000.0
        assert got.dtype == torch.float32 and got.shape == want.shape
        assert torch.equal(got.view(torch.int32), want.contiguous().view(torch.int32))
    f = torch.rand(3, 5, device=DEV)
    assert torch.equal(o.additive_mask(f), (1.0 - f) * -10000.0)


@pytest.mark.parametrize("case", ["vit577_kd", "vit901", "vit450_map", "cross577_shared", "cross901_map"])
def test_streaming_long_sequence_kernels_agree_with_the_whole_row_kernels(case, monkeypatch):
    """attn_fwd_stream_kernel / attn_fwd_stream_map_kernel / attn_bwd_dq_stream_kernel (+ kernel B rebuilding the map)
    against the whole-row kernels they replace on 225..928 keys (EVLM_ATTN_NO_STREAM=1): the same context, row lse,
    distillation term, map and gradients at bf16 resolution - online softmax over 128-key blocks with 1 / l applied to the
    context is a different rounding sequence, not different arithmetic (eff_vit.py:134-199, eff_bert.py:277-364).  Key
    padding masks, a closed head gate, shared K/V rows and a sequence that ends inside a block included."""
    o = ops()
    B, Bkv, H, Lq, Lk, self_attn, want, kd = {"vit577_kd": (2, 2, 12, 577, 577, True, False, True),
                                              "vit901": (1, 1, 4, 901, 901, True, False, False),
                                              "vit450_map": (2, 2, 3, 450, 450, True, True, False),
                                              "cross577_shared": (7, 3, 12, 30, 577, False, False, False),
                                              "cross901_map": (3, 3, 4, 30, 901, False, True, False)}[case]
    dh, d = 64, H * 64
    g = torch.Generator().manual_seed(4000 + Lq + Lk)
    x0 = rnd((B, Lq, 3 * d if self_attn else d), torch.bfloat16, g, 0.7)
    kv0 = None if self_attn else rnd((Bkv, Lk, 2 * d), torch.bfloat16, g, 0.7)
    idx = None if (self_attn or Bkv == B) else torch.randint(0, Bkv, (B,), generator=g).to(DEV)
    mask = torch.zeros(B, Lk)
    mask[0, Lk - 11:] = -10000.0
    mask = mask.to(DEV)
    gate0 = torch.rand(H, generator=g) + 0.5
    gate0[0] = 0.0
    gate0 = gate0.to(DEV)
    gO = rnd((B, Lq, d), torch.bfloat16, g)
    Pt = None
    if kd:
        with torch.no_grad():
            Pt = o.self_attention(rnd((B, Lq, 3 * d), torch.bfloat16, g, 0.7), H, dh, 0.125, mask=mask)[1]

    def run(no_stream):
        monkeypatch.setenv("EVLM_ATTN_NO_STREAM", "1" if no_stream else "0")
        x = x0.clone().requires_grad_(True)
        kv = kv0.clone().requires_grad_(True) if kv0 is not None else None
        gate = gate0.clone().requires_grad_(True)
        k_term = None
        if self_attn and kd:
            O, P, k_term = o.self_attention(x, H, dh, 0.125, mask=mask, gate=gate, want_probs=want, kd_teacher=Pt, kd_weight=float(Lk))
        elif self_attn:
            O, P = o.self_attention(x, H, dh, 0.125, mask=mask, gate=gate, want_probs=want)
        else:
            O, P = o.cross_attention(x, kv, H, dh, 0.125, mask=mask, gate=gate, want_probs=want, kv_index=idx)
        loss = (O.float() * gO.float()).sum()
        if k_term is not None:
            loss = loss + 0.3 * k_term
        if P is not None:
            loss = loss + (P.float() * 0.5).pow(2).sum()          # a gradient through the map too (stored-map backward)
        loss.backward()
        return (O.detach(), P.detach() if P is not None else None, k_term.detach() if k_term is not None else None,
                x.grad.float(), kv.grad.float() if kv is not None else None, gate.grad.clone())

    a, b = run(False), run(True)
    l2 = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    assert rel_err(a[0].float(), b[0].float()) < 1e-2 and l2(a[0].float(), b[0].float()) < 4e-3
    assert (a[1] is None) == (b[1] is None) == (not want)
    if want:
        assert rel_err(a[1].float(), b[1].float()) < 1e-2
        assert float(a[1][..., Lk:].float().abs().max() if a[1].shape[-1] > Lk else 0.0) == 0.0
    if kd:
        assert abs(float(a[2]) - float(b[2])) < 1e-4 * abs(float(b[2]))
    assert l2(a[3], b[3]) < 6e-3, l2(a[3], b[3])
    if a[4] is not None:
        assert l2(a[4], b[4]) < 6e-3, l2(a[4], b[4])
    assert l2(a[5], b[5]) < 6e-3, (a[5], b[5])


@pytest.mark.parametrize("B,H,L,kd", [(2, 12, 577, "map"), (1, 4, 901, "none"), (3, 2, 450, "none"), (2, 3, 641, "recipe"), (1, 12, 901, "recipe"),
                                     (2, 2, 620, "none"), (1, 2, 448, "map"), (1, 3, 928, "recipe")])       # (even chunk counts, a full last chunk, the longest row)
def test_round6_streaming_backward_kernels_are_bit_identical_to_the_forms_they_replace(B, H, L, kd, monkeypatch):
    """Round 6's long-sequence backward against round 4 / 5's (EVLM_ATTN_DQ_NO_BATCH=1, EVLM_ATTN_DKV_NO_STREAM=1):
    kernel A with every LDS read of a tile pair issued up front behind counted waits (attn_bwd_dq_stream_kernel<.., BATCH>), kernel
    B with its chunks by LDS-DMA into a double buffer and one barrier per chunk (attn_bwd_dkv_stream_kernel).  The same operands
    in the same k-slots and the same order of every sum: the input gradient bit for bit - without a distillation term, with a
    stored teacher map, with the teacher's recipe (Q, K, row lse); key padding mask, head gates, sequences that end inside a
    chunk / a key block."""
    o = ops()
    dh, d = 64, H * 64
    g = torch.Generator().manual_seed(5100 + L)
    x0 = rnd((B, L, 3 * d), torch.bfloat16, g, 0.7)
    mask = None
    if kd != "recipe":                                  # (the recipe form serves unmasked self-attention: the ViT)
        mask = torch.zeros(B, L)
        mask[0, L - 13:] = -10000.0
        mask = mask.to(DEV)
    gate0 = (torch.rand(H, generator=g) + 0.5).to(DEV)
    gO = rnd((B, L, d), torch.bfloat16, g)
    teacher = None
    with torch.no_grad():
        tx = rnd((B, L, 3 * d), torch.bfloat16, g, 0.7)
        if kd == "map":
            teacher = o.self_attention(tx, H, dh, 0.125, mask=mask)[1]
        elif kd == "recipe":
            teacher = o.self_attention_recipe(tx, H, dh, 0.125)[1]

    def run(old):
        monkeypatch.setenv("EVLM_ATTN_DKV_NO_STREAM", "1" if old else "0")
        monkeypatch.setenv("EVLM_ATTN_DQ_NO_BATCH", "1" if old else "0")
        x = x0.clone().requires_grad_(True)
        gate = gate0.clone().requires_grad_(True)
        if teacher is not None:
            O, _, k_term = o.self_attention(x, H, dh, 0.125, mask=mask, gate=gate, want_probs=False, kd_teacher=teacher, kd_weight=float(L))
            loss = (O.float() * gO.float()).sum() + 0.3 * k_term
        else:
            O, _ = o.self_attention(x, H, dh, 0.125, mask=mask, gate=gate, want_probs=False)
            loss = (O.float() * gO.float()).sum()
        loss.backward()
        return x.grad.clone(), gate.grad.clone()

    (xa, ga), (xb, gb) = run(False), run(True)
    assert torch.isfinite(xa.float()).all() and float(xa.float().abs().max()) > 0
    assert torch.equal(xa.view(torch.int16), xb.view(torch.int16))
    assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-5)        # (kernel A's gate gradient: f32 atomics, order not fixed)


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_kernel_b_rebuilding_the_map_on_short_rows_is_bit_identical_to_the_workspace_form(drop, monkeypatch):
    """EVLM_ATTN_KB_REBUILD=1 (round 6, opt-in): on <= 224 keys kernel B of the two-kernel cross-attention backward rebuilds the
    probabilities from Q, K and the row lse instead of reading the workspace kernel A writes - the same bf16 values, so the same
    gradients bit for bit; shared K/V rows (kv_index), key padding mask, head gates, with and without probability dropout."""
    o = ops()
    B, Bkv, H, Lq, Lk, dh = 12, 4, 12, 30, 197, 64
    d = H * dh
    g = torch.Generator().manual_seed(6100)
    x0 = rnd((B, Lq, d), torch.bfloat16, g, 0.7)
    kv0 = rnd((Bkv, Lk, 2 * d), torch.bfloat16, g, 0.7)
    idx = (torch.arange(B) % Bkv).to(DEV)
    mask = torch.zeros(B, Lk)
    mask[1, Lk - 9:] = -10000.0
    mask = mask.to(DEV)
    gate0 = (torch.rand(H, generator=g) + 0.5).to(DEV)
    gO = rnd((B, Lq, d), torch.bfloat16, g)

    def run(rebuild):
        monkeypatch.setenv("EVLM_ATTN_KB_REBUILD", "1" if rebuild else "0")
        x = x0.clone().requires_grad_(True)
        kv = kv0.clone().requires_grad_(True)
        gate = gate0.clone().requires_grad_(True)
        o.dropout_seed(77, DEV)
        O, _ = o.cross_attention(x, kv, H, dh, 0.125, mask=mask, gate=gate, want_probs=False, kv_index=idx, dropout_p=drop)
        (O.float() * gO.float()).sum().backward()
        return x.grad.clone(), kv.grad.clone()

    (xa, ka), (xb, kb) = run(True), run(False)
    assert torch.isfinite(ka.float()).all() and float(ka.float().abs().max()) > 0
    assert torch.equal(xa.view(torch.int16), xb.view(torch.int16))
    assert torch.equal(ka.view(torch.int16), kb.view(torch.int16))


def test_attention_lse_form_refuses_what_it_cannot_serve():
    """C ABI: the lse / recompute form exists for bf16, head dim 64, Lk <= 224 or 417..928; anything else answers with
    an error code - never a fault - and evlm_attention_lse_supported says so beforehand"""
    import ctypes as C
    from efficientvlm_amd import _lib as L
    lib = L.load()
    assert lib.evlm_attention_lse_supported(L.BF16, 64, 197, 0.0) == 1
    assert lib.evlm_attention_lse_supported(L.BF16, 64, 577, 0.0) == 1 and lib.evlm_attention_lse_supported(L.BF16, 64, 901, 0.0) == 1
    assert lib.evlm_attention_lse_supported(L.BF16, 64, 300, 0.0) == 0        # 225..416 keys: stored-map form only
    assert lib.evlm_attention_lse_supported(L.BF16, 64, 1000, 0.0) == 0
    assert lib.evlm_attention_lse_supported(L.F32, 64, 30, 0.0) == 0
    assert lib.evlm_attention_lse_supported(L.BF16, 32, 30, 0.0) == 0
    assert lib.evlm_attention_lse_supported(L.BF16, 64, 30, 0.1) == 1         # (round 6: the MFMA kernels regenerate the keep-mask)
    assert lib.evlm_attention_lse_supported(L.BF16, 64, 577, 0.1) == 1 and lib.evlm_attention_lse_supported(L.BF16, 64, 30, 1.0) == 0
    B, H, Lq, Lk, dh = 1, 2, 16, 300, 64
    d = H * dh
    q = torch.zeros(B, Lq, d, dtype=torch.bfloat16, device=DEV)
    kv = torch.zeros(B, Lk, 2 * d, dtype=torch.bfloat16, device=DEV)
    O = torch.empty_like(q)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    a = L.AttnFwdArgs(dtype=L.BF16, p_dtype=L.BF16, B=B, H=H, Lq=Lq, Lk=Lk, dh=dh, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, ldpr=304,
                      Q=L.ptr(q), K=L.ptr(kv), V=C.c_void_p(kv.data_ptr() + d * 2), scale=0.125, O=L.ptr(O), lse=L.ptr(lse))
    assert lib.evlm_attention_fwd(C.byref(a), L.stream()) != 0 and b"lse" in lib.evlm_last_error()
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    b = L.AttnBwdArgs(dtype=L.BF16, p_dtype=L.BF16, B=B, H=H, Lq=Lq, Lk=Lk, dh=dh, Bkv=B, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d,
                      lddq=d, lddk=2 * d, lddv=2 * d, ldpr=304, Q=L.ptr(q), K=L.ptr(kv), V=C.c_void_p(kv.data_ptr() + d * 2),
                      dO=L.ptr(O), scale=0.125, dQ=L.ptr(dq), dK=L.ptr(dkv), dV=C.c_void_p(dkv.data_ptr() + d * 2), lse=L.ptr(lse))
    assert lib.evlm_attention_bwd(C.byref(b), L.stream()) != 0
    b.lse = None                                          # neither P nor lse
    assert lib.evlm_attention_bwd(C.byref(b), L.stream()) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_mse_terms_equal_the_per_pair_reductions(dtype):
    """ops.mse_terms (evlm_mse_grouped: all pairs of all terms in one launch per direction) against ops.mse_sum per term:
    values and gradients, padded attention-map views, an odd-sized pair, an empty term, a term nobody differentiates"""
    o = ops()
    g = torch.Generator().manual_seed(77)
    mk = lambda *shape: rnd(shape, dtype, g)
    Pbuf_s, Pbuf_t = torch.zeros(4, 3, 30, 32, device=DEV, dtype=dtype), torch.zeros(4, 3, 30, 32, device=DEV, dtype=dtype)
    Pbuf_s[..., :30], Pbuf_t[..., :30] = mk(4, 3, 30, 30), mk(4, 3, 30, 30)
    shapes = [[(64, 197, 96)] * 3, [(8, 30, 96), (8, 30, 96)], [], [(5, 7, 3)], [(16, 33)]]
    weights = [[1.0, 1.0, 0.5], [2.0, 1.0], [], [30.0], [1.0]]

    def build():
        terms = []
        for shp, ws in zip(shapes, weights):
            terms.append(([(mk(*s).requires_grad_(True), mk(*s)) for s in shp], list(ws)))
        a = Pbuf_s.clone().requires_grad_(True)
        terms[1][0].append((a[..., :30], Pbuf_t[..., :30])); terms[1][1].append(30.0)       # padded map views
        return terms, a

    g.manual_seed(5); terms_a, pa = build()
    g.manual_seed(5); terms_b, pb = build()
    coef = [0.3, 1.0, 1.0, 2.0, 0.0]
    va = o.mse_terms(terms_a)
    vb = [o.mse_sum(p, w) if p else 0 for p, w in terms_b]
    assert va[2] == 0 and len(va) == 5
    la = sum(c * v for c, v in zip(coef[:4], va[:4]))             # term 4 takes no part in the loss
    lb = sum(c * v for c, v in zip(coef[:4], vb[:4]))
    la.backward(); lb.backward()
    t = 1e-5 if dtype == torch.float32 else 2e-3
    for x, y in zip(va, vb):
        if torch.is_tensor(x):
            assert rel_err(x, y) < t
    for (pa_, _), (pb_, _) in zip(terms_a, terms_b):
        for (xa, _), (xb, _) in zip(pa_, pb_):
            if not xa.is_leaf:
                continue                                          # (the padded views: checked through their base below)
            if xa.grad is None or xb.grad is None:
                assert xa.grad is None and xb.grad is None
            else:
                assert rel_err(xa.grad.float(), xb.grad.float()) < t
    assert rel_err(pa.grad.float(), pb.grad.float()) < t and float(pa.grad[..., 30:].abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["self197", "cross_shared", "self577"])
def test_closed_heads_are_skipped_in_a_no_grad_forward(kind):
    """BASELINE configs[4] "kernel skips masked heads": in a no-grad forward that wants no map, a head whose L0 gate is
    exactly 0 gets its zero context written without staging K / V or computing anything (attn_fwd_mfma_kernel,
    attn_fwd_grouped_kernel); open heads are untouched - bit-identical to the launch with those gates at 1 on the open heads
    and to plain fp32 math overall"""
    o = ops()
    g = torch.Generator().manual_seed(23)
    H, dh = 12, 64
    d = H * dh
    gate = torch.ones(H)
    gate[[1, 4, 5, 11]] = 0.0
    gate[2] = 0.5
    gate = gate.to(DEV)
    with torch.no_grad():
        if kind == "cross_shared":
            B, Bkv, Lq, Lk = 12, 3, 30, 197
            q = rnd((B, Lq, d), torch.bfloat16, g); kv = rnd((Bkv, Lk, 2 * d), torch.bfloat16, g)
            idx = (torch.arange(B) % Bkv).to(DEV)
            O, P = o.cross_attention(q, kv, H, dh, 0.125, gate=gate, want_probs=False, kv_index=idx)
            Oref, _ = o.cross_attention(q, kv, H, dh, 0.125, gate=torch.ones(H, device=DEV), want_probs=False, kv_index=idx)
        else:
            L = 197 if kind == "self197" else 577
            B = 3 if kind == "self197" else 2
            x = rnd((B, L, 3 * d), torch.bfloat16, g)
            O, P = o.self_attention(x, H, dh, 0.125, gate=gate, want_probs=False)
            Oref, _ = o.self_attention(x, H, dh, 0.125, gate=torch.ones(H, device=DEV), want_probs=False)
    assert P is None
    Oh, Rh = O.float().unflatten(-1, (H, dh)), Oref.float().unflatten(-1, (H, dh))
    for h in range(H):
        if float(gate[h]) == 0.0:
            assert float(Oh[..., h, :].abs().max()) == 0.0
        elif float(gate[h]) == 1.0:
            assert torch.equal(Oh[..., h, :], Rh[..., h, :])
        else:
            assert rel_err(Oh[..., h, :], Rh[..., h, :] * float(gate[h])) < 1e-2


def test_fused_map_distillation_term_of_a_forward_without_a_backward():
    """eff_vit.CLIPAttention with a teacher map and nothing to differentiate (a validation-loss pass under no_grad, or frozen
    inputs): no row lse exists without a backward, so the layer takes the stored-map form of the term instead of refusing
    the call - same term, same context as the training forward"""
    from efficientvlm_amd.efficient_models.eff_vit import CLIPAttention
    from efficientvlm_amd.runtime import compute
    torch.manual_seed(5)
    attn = CLIPAttention(768, 12, 0.0).to(DEV)
    x = (torch.randn(2, 197, 768, device=DEV) * 0.5).to(torch.bfloat16)      # (activations carry the compute dtype)
    with torch.no_grad(), compute(torch.bfloat16):
        _, Pt = attn(x * 0.9, output_attentions=True)
    with compute(torch.bfloat16):
        xg = x.clone().requires_grad_(True)
        out_g, _, kd_g = attn(xg, kd_teacher=Pt)
        with torch.no_grad():
            out_n, p_n, kd_n = attn(x, kd_teacher=Pt)
    assert p_n is None and torch.isfinite(kd_n).all()
    assert rel_err(out_n.float(), out_g.float()) < 1e-6
    assert rel_err(kd_n, kd_g.detach()) < 2e-2            # (stored-map form: the term of the bf16-rounded probabilities)


def test_one_wave_per_simd_gemm_kernel_passes_the_race_screen():
    """gemm_w4.hip (opt-in EVLM_W4=1: 4 waves x 128 x 128 or 96 x 128 of C, accumulators pinned in AGPRs, fragment reads and
    LDS-DMA inside the MFMA stream, persistent): forward and dX products (K-contiguous and reduction-major Q, edge tiles,
    short reductions) against fp32 torch products of the same bf16 inputs, repeated - a separate process, the switch is
    read once per process"""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (round 5: the kernel is out of the default library - `make -C efficientvlm_amd/csrc EXPERIMENTAL=1 LIB=...` builds it)
    exp = os.path.join(repo, "tools", "_build", "libevlm_exp.so")
    if not os.path.exists(exp):
        pytest.skip("experimental library (gemm_w4.hip) not built")
    env = dict(os.environ, EVLM_W4="1", EVLM_LIB=exp)
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "gemm_pp256_race_screen.py"), "2"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "race screen: CLEAN" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
