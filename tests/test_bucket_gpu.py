"""Bucket-padded batches of the fine-tune steps (round 6; reference: Eff_Retrieval.py:97 and Eff_VQA.py:97-98 tokenise with
padding='longest', dataset/vqa_dataset.py:101-116 hands a variable number of answers per question - nearly every batch of an
epoch has its own shape).  data.bucket_pad_itr / bucket_pad_vqa pad a batch on to one of a few shapes so that the captured
step (hipGraph) replays; the arithmetic must stay that of the 'longest'-padded batch: the distillation kernels skip the rows
of padded tokens / padded answer rows (ops.Ragged: real extents in DEVICE words, so one graph serves every real length of a
bucket) and the loss mixes rescale the padded denominators (kd_corr)."""
import math

import pytest
import torch

from helpers import load_det_weights, model_config
from oracle import schema, synth
from oracle import xvlm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ragged_distillation_terms_skip_what_lies_beyond_the_real_extents(dtype):
    """ops.mse_terms with ops.Ragged operands and ops.soft_cross_entropy(ragged=): hidden states [rows, tokens, d], attention
    maps [rows, H, tokens, keys] (row-padded buffers) and logits [rows, tokens, classes] whose padded region holds garbage -
    values and gradients equal those of the sliced tensors (times real / padded: the kernels keep the padded denominators),
    the gradient beyond the real extents is exactly zero, and the SAME launch parameters serve other extents (the words are
    read on the device)"""
    from efficientvlm_amd import ops
    g = torch.Generator().manual_seed(0)
    R, H, L, Lk, d, V = 6, 3, 16, 21, 64, 50
    ext = torch.tensor([11, 7, 4, 0], dtype=torch.int32, device=DEV)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV, dtype)
    hs, ht = mk(R, L, d).requires_grad_(True), mk(R, L, d)
    Lkp = (Lk + 7) // 8 * 8
    ms_full, mt_full = torch.zeros(R, H, L, Lkp, device=DEV, dtype=dtype), torch.zeros(R, H, L, Lkp, device=DEV, dtype=dtype)
    ms_full[..., :Lk], mt_full[..., :Lk] = mk(R, H, L, Lk), mk(R, H, L, Lk)
    ms = ms_full[..., :Lk].requires_grad_(True)
    mt = mt_full[..., :Lk]
    ls, lt = mk(R, L, V).requires_grad_(True), mk(R, L, V)
    hd = mk(R, L, d).requires_grad_(True)             # (decoder-side operands: two extents, rows and tokens)
    md_full = torch.zeros(R, H, L, Lkp, device=DEV, dtype=dtype)
    md_full[..., :Lk] = mk(R, H, L, Lk)
    md = md_full[..., :Lk].requires_grad_(True)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for trial in range(2):
        if trial == 1:
            ext.copy_(torch.tensor([5, 16, 6, 0], dtype=torch.int32))          # other extents, the same table contents
        e = ext.tolist()
        rq, rd = ops.Ragged(ext, inner=0), ops.Ragged(ext, inner=1, outer=2)
        for t in (hs, ms, ls, hd, md):
            t.grad = None
        vals = ops.mse_terms([([(hs, ht)], [1.0], rq), ([(ms, mt)], [float(Lk)], rq), ([(hd, ht)], [2.0], rd), ([(md, mt)], [1.0], rd)])
        kl = ops.soft_cross_entropy(ls, lt, 1.0, ragged=rd)
        (vals[0] + vals[1] + vals[2] + vals[3] + kl).backward()
        f32 = lambda t: t.detach().float().requires_grad_(True)
        hf, mf, lf, hdf, mdf = f32(hs), f32(ms), f32(ls), f32(hd), f32(md)
        mse = torch.nn.functional.mse_loss
        w0 = mse(hf[:, :e[0]], ht.float()[:, :e[0]]) * (e[0] / L)
        w1 = mse(mf[:, :, :e[0]], mt.float()[:, :, :e[0]]) * Lk * (e[0] / L)
        w2 = mse(hdf[:e[2], :e[1]], ht.float()[:e[2], :e[1]]) * 2.0 * (e[2] * e[1] / (R * L))
        w3 = mse(mdf[:e[2], :, :e[1]], mt.float()[:e[2], :, :e[1]]) * (e[2] * e[1] / (R * L))
        lsm = torch.log_softmax(lf[:e[2], :e[1]], -1)
        pt = torch.softmax(lt.float()[:e[2], :e[1]], -1)
        wk = (pt * (torch.log(pt) - lsm)).sum(-1).sum() / (R * L)
        for a, b in zip(list(vals) + [kl], (w0, w1, w2, w3, wk)):
            assert abs(float(a) - float(b)) <= tol * abs(float(b)) + 1e-7, (trial, float(a), float(b))
        (w0 + w1 + w2 + w3 + wk).backward()
        for got, ref in ((hs, hf), (ms, mf), (ls, lf), (hd, hdf), (md, mdf)):
            assert rel_err(got.grad.float(), ref.grad) < 2 * tol
        # beyond a term's extents the gradient is exactly zero
        zero = lambda t: t.numel() == 0 or float(t.abs().max()) == 0.0
        assert zero(hs.grad[:, e[0]:]) and zero(hd.grad[e[2]:]) and zero(hd.grad[:, e[1]:])
        assert zero(ls.grad[e[2]:]) and zero(ls.grad[:, e[1]:]) and zero(md.grad[e[2]:]) and zero(md.grad[:, :, e[1]:])


def _itr_pair(geom, seed):
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = EffXVLMforRetrieval(model_config(geom, "s")), TeacherITR(model_config(geom, "t"))
    load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), seed, geom["std"])
    load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False), seed + 1, geom["std"])
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
    student.l0_module.set_lagrangian_warmup_steps(10)
    return student.to(DEV), teacher.to(DEV), gen


def _fixed_negatives(monkeypatch):
    from efficientvlm_amd.efficient_models.xvlm import XVLMBase

    def fixed(self, image_feat, text_feat, idx):
        bs = image_feat.size(0)
        ar = torch.arange(bs, device=image_feat.device)
        return (ar + 1) % bs, (ar + 2) % bs
    monkeypatch.setattr(XVLMBase, "_sample_negatives", fixed)


def _eps(student, gen, types):
    return {t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6) for t in types}


@pytest.mark.parametrize("geom_name,dtype", [("tiny", torch.float32), ("full", torch.float32)])
def test_bucket_padded_itr_step_equals_the_longest_padded_step(geom_name, dtype, monkeypatch):
    """one Eff_Retrieval training step on a batch in the reference's padding (text padded to the batch's longest row, some rows
    shorter) against the same step on data.bucket_pad_itr(batch): every loss and distillation term and every parameter
    gradient agree to fp32 rounding (the padded tokens are masked keys; as queries their rows are skipped by the kernels)"""
    from efficientvlm_amd.data import bucket_pad_itr
    from efficientvlm_amd.trainer import ITRTrainer
    _fixed_negatives(monkeypatch)
    geom = dict(synth.GEOMS[geom_name])
    if geom_name == "full":
        geom.update(L=21, M=4)                    # (21 real tokens -> the 24-token bucket)
    B = 4
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, B, seed=3, ragged=True).items() if k in ("image", "text_ids", "text_atts")}
    idx = torch.arange(B, device=DEV)
    res = {}
    for padded in (False, True):
        student, teacher, gen = _itr_pair(geom, 61)
        student.l0_module.injected_eps = _eps(student, gen, O.L0_TYPES)
        tr = ITRTrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=dtype)
        b = bucket_pad_itr(batch) if padded else batch
        if padded:
            assert b["text_ids"].shape[1] in (16, 24) and b["text_ids"].shape[1] > batch["text_ids"].shape[1]
            assert b["extents"].tolist()[0] == batch["text_ids"].shape[1]
        out = tr.step(b, idx=idx)
        torch.cuda.synchronize()
        res[padded] = (out.cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in student.named_parameters() if p.grad is not None})
        tr.close()
    assert torch.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-7), (res[True][0], res[False][0])
    gmax = max(float(g.norm()) for g in res[False][1].values())
    n = 0
    for name, g0 in res[False][1].items():
        g1 = res[True][1][name]
        assert float((g1 - g0).norm()) <= 1e-5 * float(g0.norm()) + 1e-7 * gmax, (name, float((g1 - g0).norm()), float(g0.norm()))
        n += 1
    assert n > 100


def _vqa_models(geom, seed_s, seed_t):
    from efficientvlm_amd.efficient_models.model_generation import EffXVLMForVQA
    from efficientvlm_amd.models.model_generation import XVLMForVQA
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    cfg = lambda role, c: dict(model_config(geom, role), pad_token_id=0, num_dec_layers=c["text_layers"] - c["fusion_layer"])
    student, teacher = EffXVLMForVQA(cfg("s", s_cfg)), XVLMForVQA(cfg("t", t_cfg))
    load_det_weights(student, schema.vqa_schema(s_cfg, geom["max_pos"], l0=True), seed_s, geom["std"])
    load_det_weights(teacher, schema.vqa_schema(t_cfg, geom["max_pos"]), seed_t, geom["std"])
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
    student.l0_module.set_lagrangian_warmup_steps(10)
    return student.to(DEV), teacher.to(DEV), gen


def test_bucket_padded_vqa_step_equals_the_longest_padded_step():
    """one Eff_VQA training step on a batch as the reference collates it (questions and answers padded to their longest, sum k
    answer rows) against data.bucket_pad_vqa(batch) - question / answer tokens padded to their buckets, the answer rows to a
    multiple of the row block with weight 0: answer loss, every distillation term (decoder states, maps and logits included)
    and every parameter gradient agree to fp32 rounding"""
    from efficientvlm_amd.data import bucket_pad_vqa
    from efficientvlm_amd.trainer import VQATrainer
    geom = synth.GEOMS["tiny"]
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth.make_vqa_batch(geom, 4, seed=21).items()}
    res = {}
    for padded in (False, True):
        student, teacher, gen = _vqa_models(geom, 41, 42)
        student.l0_module.injected_eps = _eps(student, gen, O.L0_TYPES_VQA)
        tr = VQATrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32)
        b = bucket_pad_vqa(batch, row_block=8) if padded else batch
        if padded:
            assert b["answer_ids"].shape[0] % 8 == 0 and b["answer_ids"].shape[0] > batch["answer_ids"].shape[0]
            assert b["question_ids"].shape[1] == 16 and int(torch.as_tensor(b["k"]).sum()) == b["answer_ids"].shape[0]
        out = tr.step(b)
        torch.cuda.synchronize()
        res[padded] = (out.cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in student.named_parameters() if p.grad is not None})
        tr.close()
    assert torch.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-7), (res[True][0], res[False][0])
    gmax = max(float(g.norm()) for g in res[False][1].values())
    n = 0
    for name, g0 in res[False][1].items():
        g1 = res[True][1][name]
        assert float((g1 - g0).norm()) <= 1e-5 * float(g0.norm()) + 1e-7 * gmax, (name, float((g1 - g0).norm()), float(g0.norm()))
        n += 1
    assert n > 100


def test_one_captured_itr_step_serves_every_real_length_of_its_bucket(monkeypatch):
    """ITRTrainer(pipeline_teacher=True, capture_step=True) fed bucket-padded batches whose REAL text length changes from
    step to step (5 .. 8 tokens, all in the 16-token bucket): after the two warm-up / capture rounds every step replays from a
    hipGraph - one per teacher parity, not one per length - and the loss trajectory is the eager trainer's on the same
    batches in the reference's own padding"""
    from efficientvlm_amd.data import bucket_pad_itr
    from efficientvlm_amd.trainer import ITRTrainer
    _fixed_negatives(monkeypatch)
    base = synth.GEOMS["tiny"]
    B, lens = 4, [8, 6, 7, 5, 8, 6, 5, 7, 8]
    raw = []
    for i, L in enumerate(lens):
        geom = dict(base, L=L, M=2)
        raw.append({k: v.to(DEV) for k, v in synth.make_batch(geom, B, seed=70 + i, ragged=True).items()
                    if k in ("image", "text_ids", "text_atts")})
    idx = torch.arange(B, device=DEV)
    outs = {}
    for pipe in (False, True):
        student, teacher, gen = _itr_pair(base, 51)
        eps = [_eps(student, gen, O.L0_TYPES) for _ in lens]
        tr = ITRTrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=pipe,
                        use_graph=pipe, capture_step=pipe)
        feed = [bucket_pad_itr(b) for b in raw] if pipe else raw
        calls = feed + ([feed[0]] if pipe else [])
        seq, launches = [], []
        for c, b in enumerate(calls):
            i = c - 1 if pipe else c
            if i >= 0:
                student.l0_module.injected_eps = {t: e.clone() for t, e in eps[i].items()}
            o = tr.step(b, idx=idx)
            if o is not None:
                seq.append(o.clone())
                launches.append(tr.last_launch)
        torch.cuda.synchronize()
        outs[pipe] = torch.stack(seq).cpu()
        if pipe:
            assert len(tr._sgraphs) == 2, len(tr._sgraphs)          # one graph per teacher parity for ALL the lengths
            assert launches[4:] == ["hipGraph replay"] * (len(launches) - 4), launches
        tr.close()
    assert torch.allclose(outs[True], outs[False], rtol=5e-4, atol=1e-5), (outs[True], outs[False])
