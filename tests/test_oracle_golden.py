"""Pins oracle/xvlm_oracle.py (our CPU restatement) to vectors captured from the reference itself.

The fixtures under tests/golden/ were written by oracle/gen_golden.py, which imports the reference's
own modules (efficient_models.*, models.*, GeneralDistill.py helpers) in the build container.
fp32 throughout; tolerance 1e-5 relative on tensors / 1e-6 on the tiny-config scalars.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import schema, synth
from oracle import xvlm_oracle as O
from oracle.detinit import checksums

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = a.detach().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() if a.size else 0.0
    tol = atol + rtol * np.abs(b).max() if b.size else atol
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"


def weights_from_fixture(fx, tag, sch, seed, std):
    sd = schema.det_weights(sch, seed, std)
    ref_names = {k[len(tag) + 6:] for k in fx if k.startswith(tag + ".wchk.")}
    assert ref_names == set(sd), f"state-dict keys differ from the reference: {sorted(ref_names ^ set(sd))[:8]}"
    for n, (s, a) in checksums(sd).items():
        np.testing.assert_allclose([s, a], fx[f"{tag}.wchk.{n}"], rtol=1e-9, atol=1e-9, err_msg=n)
    return sd


def batch_from_fixture(fx):
    return {k[3:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("in.") and "." not in k[3:]}


def check_outputs(fx, tag, out, full=True, rtol=1e-5):
    for dn in ("hidden_dict", "attention_dict", "cross_attention_dict"):
        for k, tup in out[dn].items():
            if full:
                for i, t in enumerate(tup):
                    close(t, fx[f"{tag}.{k}.{i}"], rtol=rtol, what=f"{tag}.{k}.{i}")
                assert f"{tag}.{k}.{len(tup)}" not in fx
            else:
                chk = fx[f"{tag}.{k}.chk"]
                assert len(chk) == len(tup)
                for i, t in enumerate(tup):
                    got = [float(t.detach().double().sum()), float(t.detach().double().pow(2).sum().sqrt())]
                    np.testing.assert_allclose(got, chk[i], rtol=1e-4, atol=1e-4, err_msg=f"{tag}.{k}.{i}")
    for k, t in out["logits_dict"].items():
        if f"{tag}.{k}" in fx:
            close(t, fx[f"{tag}.{k}"], rtol=rtol, atol=1e-5, what=f"{tag}.{k}")
        else:
            got = [float(t.detach().double().sum()), float(t.detach().double().pow(2).sum().sqrt())]
            np.testing.assert_allclose(got, fx[f"{tag}.{k}.chk"], rtol=1e-4, atol=1e-3)
            close(t.reshape(-1, t.shape[-1])[:4, :64], fx[f"{tag}.{k}.head"], rtol=1e-4, atol=1e-5, what=k)
    for k, t in out["loss"].items():
        close(t, fx[f"{tag}.{k}"], rtol=rtol, what=f"{tag}.{k}")


def check_grads(fx, tag, sd, full, rtol):
    n_checked = 0
    for n, p in sd.items():
        key = f"{tag}.grad_chk.{n}"
        if key not in fx:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0 or n.endswith(("decoder.weight", "decoder.bias")), n
            continue
        g = p.grad
        assert g is not None, f"no grad for {n}"
        ref_l2 = fx[key][1]
        got_l2 = float(g.double().pow(2).sum().sqrt())
        assert abs(got_l2 - ref_l2) <= rtol * ref_l2 + 1e-6, f"grad L2 {n}: {got_l2} vs {ref_l2}"   # 1e-6 floor: k-bias grads are analytically 0
        if f"{tag}.grad.{n}" in fx:
            close(g, fx[f"{tag}.grad.{n}"], rtol=rtol, atol=rtol * ref_l2 + 1e-6, what=f"grad {n}")
        elif f"{tag}.grad_head.{n}" in fx:
            close(g.reshape(-1)[:64], fx[f"{tag}.grad_head.{n}"], rtol=rtol, atol=rtol * ref_l2 + 1e-6, what=f"grad {n}")
        n_checked += 1
    assert n_checked > 50


def tie(sd):
    """alias the tied MLM decoder tensors exactly like HF tie_weights does"""
    w = "text_encoder.bert.embeddings.word_embeddings.weight"
    if w in sd:
        sd["text_encoder.cls.predictions.decoder.weight"] = sd[w]
        sd["text_encoder.cls.predictions.decoder.bias"] = sd["text_encoder.cls.predictions.bias"]
    return sd


def leafify(sd):
    out, seen = {}, {}
    for k, v in sd.items():
        if torch.is_floating_point(v):
            if id(v) not in seen:
                seen[id(v)] = v.clone().requires_grad_(True)
            out[k] = seen[id(v)]
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("name,full", [("gd_tiny.npz", True), ("gd_full.npz", False),
                                       ("gd_region_tiny.npz", True), ("gd_region_full.npz", False)])
def test_gd_step_matches_reference(golden_dir, name, full):
    """general steps (GeneralDistill.py:264-387) and REGION steps (:158-262: idx_to_group_img / image_atts / bbox)"""
    fx = load(golden_dir, name)
    geom = synth.GEOMS[str(fx["meta.geom"])]
    seed = int(fx["meta.seed"])
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sd = weights_from_fixture(fx, "student", schema.xvlm_schema(s_cfg, geom["max_pos"]), 1000 + seed, geom["std"])
    t_sd = weights_from_fixture(fx, "teacher", schema.xvlm_schema(t_cfg, geom["max_pos"]), 2000 + seed, geom["std"])
    s_sd = leafify(tie(s_sd))
    t_sd = tie(t_sd)
    batch = batch_from_fixture(fx)
    # the synthetic batch generator itself is part of the contract
    if "in.idx_to_group_img" in fx:
        regen = synth.make_region_batch(geom, int(fx["meta.B"]), fx["in.text_ids"].shape[0], seed=seed, ragged=True)
        assert "student.loss_bbox" in fx and "student.bbox_hidden_states.0" in fx or not full
    else:
        regen = synth.make_batch(geom, int(fx["meta.B"]), seed=seed, ragged=True)
    for k, v in regen.items():
        assert torch.equal(v, batch[k]), k
    s_neg = torch.from_numpy(fx["in.student_neg_idx"])
    t_neg = torch.from_numpy(fx["in.teacher_neg_idx"])
    total, S, T, kd, mix = O.gd_step(s_sd, t_sd, s_cfg, t_cfg, batch, s_neg, t_neg)
    rt = 1e-5 if full else 1e-4
    check_outputs(fx, "student", S, full, rt)
    check_outputs(fx, "teacher", T, full, rt)
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], rtol=rt, what=f"kd.{k}")
    for k, v in mix.items():
        close(v, fx[f"mix.{k}"], rtol=rt, what=f"mix.{k}")
    close(total, fx["mix.total"], rtol=rt, what="total")
    total.backward()
    check_grads(fx, "student", s_sd, full, 2e-4 if full else 1e-3)


def test_itr_step_with_l0_matches_reference(golden_dir):
    fx = load(golden_dir, "itr_tiny.npz")
    geom = synth.GEOMS[str(fx["meta.geom"])]
    seed = int(fx["meta.seed"])
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sch = schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True)
    t_sch = schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False)
    s_sd = schema.det_weights(s_sch, 3000 + seed, geom["std"])
    for n in list(s_sd):
        if n.startswith("l0_module."):
            s_sd[n] = torch.from_numpy(fx["in.l0." + n[len("l0_module."):]]).clone()
    t_sd = weights_from_fixture(fx, "teacher", t_sch, 4000 + seed, geom["std"])
    s_sd = leafify(s_sd)
    batch = {k: torch.from_numpy(fx["in." + k]) for k in ("image", "text_ids", "text_atts")}
    idx = torch.from_numpy(fx["in.idx"])
    logas = {k[len("l0_module."):]: v for k, v in s_sd.items() if k.endswith("_loga")}
    eps = {t: torch.from_numpy(fx["in.eps." + t]) for t in O.L0_TYPES}
    zs = O.l0_forward(logas, True, eps)
    S = O.retrieval_forward(s_sd, s_cfg, batch, idx, torch.from_numpy(fx["in.student_neg_idx"]), zs)
    with torch.no_grad():
        T = O.retrieval_forward(t_sd, t_cfg, batch, idx, torch.from_numpy(fx["in.teacher_neg_idx"]))
    check_outputs(fx, "student", S)
    T["loss"] = {}
    check_outputs(fx, "teacher", T)
    kd = O.kd_terms(S, T, with_cross_attn=True)
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], what=f"kd.{k}")
    consts = O.l0_constants(geom["hidden"], geom["ffn"], geom["heads"], s_cfg["vit_layers"], s_cfg["fusion_layer"],
                            s_cfg["text_layers"] - s_cfg["fusion_layer"])
    lagr, es, ts = O.l0_lagrangian(logas, s_sd["l0_module.lambda_1"], s_sd["l0_module.lambda_2"], consts, 3,
                                   target_sparsity=0.25, lagrangian_warmup=10)
    close(lagr, fx["mix.lagrangian"], what="lagrangian")
    close(es, fx["mix.expected_sparsity"], what="expected_sparsity")
    assert abs(ts - float(fx["mix.target_sparsity"])) < 1e-12
    total, mix = O.itr_loss_mix(S["loss"], kd, lagr)
    for k, v in mix.items():
        close(v, fx[f"mix.{k}"], what=k)
    close(total, fx["mix.total"], what="total")
    total.backward()
    check_grads(fx, "student", s_sd, True, 2e-4)
    # eval mode: deterministic 0/1 masks, indices bit-exact
    with torch.no_grad():
        ze = O.l0_forward({k: v.detach() for k, v in logas.items()}, False)
        for k, v in ze.items():
            assert np.array_equal(v.numpy(), fx["eval.z." + k]), k
        E = O.retrieval_forward({k: v.detach() for k, v in s_sd.items()}, s_cfg, batch, idx,
                                torch.from_numpy(fx["eval.neg_idx"]), ze)
        close(E["loss"]["loss_itc"], fx["eval.loss_itc"], what="eval itc")
        close(E["loss"]["loss_itm"], fx["eval.loss_itm"], what="eval itm")


def test_l0_module_matches_reference(golden_dir):
    fx = load(golden_dir, "l0_full.npz")
    logas = {k[3:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in fx.items()
             if k.startswith("in.") and k.endswith("_loga")}
    lam1 = torch.from_numpy(fx["in.lambda_1"]).clone().requires_grad_(True)
    lam2 = torch.from_numpy(fx["in.lambda_2"]).clone().requires_grad_(True)
    eps = {t: torch.from_numpy(fx["in.eps." + t]) for t in O.L0_TYPES}
    zs = O.l0_forward(logas, True, eps)
    for k, v in zs.items():
        close(v, fx["train.z." + k], rtol=1e-6, atol=1e-7, what=k)
    consts = O.l0_constants(768, 3072, 12, 6, 3, 3)
    assert consts["prunable"] == int(fx["meta.prunable_model_size"]) == 92104704
    assert consts["params_per_head"] == int(fx["meta.params_per_head"]) == 196864
    assert consts["params_per_int"] == int(fx["meta.params_per_intermediate_dim"]) == 1537
    for step, trip in zip(fx["lagrangian.steps"], fx["lagrangian.triples"]):
        l, es, ts = O.l0_lagrangian(logas, lam1, lam2, consts, int(step), target_sparsity=0.6, lagrangian_warmup=200)
        np.testing.assert_allclose([float(l), float(es), float(ts)], trip, rtol=2e-6, atol=1e-7)
    tot = 0
    for i, (k, v) in enumerate(sorted(zs.items())):
        w = torch.linspace(0.5, 1.5, v.numel()).reshape(v.shape)
        tot = tot + (v * w).sum() * (i + 1)
    l, _, _ = O.l0_lagrangian(logas, lam1, lam2, consts, 37, target_sparsity=0.6, lagrangian_warmup=200)
    (tot + l).backward()
    for n, p in list(logas.items()) + [("lambda_1", lam1), ("lambda_2", lam2)]:
        close(p.grad, fx["grad." + n], rtol=1e-5, atol=1e-7, what="grad " + n)
    with torch.no_grad():
        ze = O.l0_forward(logas, False)
    for k, v in ze.items():
        ref = fx["eval.z." + k]
        assert np.array_equal(np.nonzero(v.numpy().reshape(-1) == 0)[0], np.nonzero(ref.reshape(-1) == 0)[0]), k
        assert np.array_equal(v.numpy(), ref)
    res = O.l0_model_size(ze, consts, 12, 3072)
    ref = json.loads(str(fx["eval.model_size_json"]))
    for k, v in ref.items():
        if isinstance(v, list):
            assert [int(x) for x in v] == [int(x) for x in res[k]], k
        else:
            assert abs(float(v) - float(res[k])) <= 1e-9 * max(1.0, abs(float(v))), k
    for n, p in logas.items():
        c = p.detach().clamp(min=np.log(1e-2), max=np.log(1e2))   # constrain_parameters, xvlm_l0_module.py:168-172
        close(c, fx["constrained." + n], rtol=0, atol=0, what=n)


def test_kd_helpers_match_reference(golden_dir):
    fx = load(golden_dir, "kd_helpers.npz")
    grab = lambda p, n: [torch.from_numpy(fx[f"{p}.{i}"]) for i in range(n)]
    s_h, t_h, s_a, t_a = grab("in.s_h", 7), grab("in.t_h", 13), grab("in.s_a", 6), grab("in.t_a", 12)
    ch, ca = O.get_cor_teacher(t_h, s_h), O.get_cor_teacher(t_a, s_a, True)
    assert [next(j for j, t in enumerate(t_h) if torch.equal(t, c)) for c in ch] == fx["out.cor_hidden_idx"].tolist() == [0, 2, 4, 6, 8, 10, 12]
    assert [next(j for j, t in enumerate(t_a) if torch.equal(t, c)) for c in ca] == fx["out.cor_attn_idx"].tolist() == [1, 3, 5, 7, 9, 11]
    close(O.get_kd_loss(s_h, ch), fx["out.hidden"], what="hidden")
    close(O.get_kd_loss(s_h, ch, is_img=True), fx["out.hidden_img"], what="hidden_img")
    close(O.get_kd_loss(s_a, ca, is_attn=True), fx["out.attn"], what="attn")
    close(O.soft_cross_entropy(torch.from_numpy(fx["in.s_l"]) / 2.0, torch.from_numpy(fx["in.t_l"]) / 2.0),
          fx["out.soft_ce"], what="soft_ce")


def test_vqa_step_with_l0_matches_reference(golden_dir):
    """Eff_VQA.py:95-176 training step: EffXVLMForVQA (gates of VQAL0Module, decoder gates included) vs XVLMForVQA teacher -
    causal answer decoder, weighted per-answer LM loss, every KD term, the Lagrangian, the loss mix and the gradients"""
    fx = load(golden_dir, "vqa_tiny.npz")
    geom = synth.GEOMS[str(fx["meta.geom"])]
    seed = int(fx["meta.seed"])
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sch = schema.vqa_schema(s_cfg, geom["max_pos"], l0=True)
    assert {k[len("student.wchk."):] for k in fx if k.startswith("student.wchk.")} == set(s_sch)      # checkpoint keys
    s_sd = schema.det_weights(s_sch, 5000 + seed, geom["std"])          # (the gate parameters are overwritten below)
    t_sd = weights_from_fixture(fx, "teacher", schema.vqa_schema(t_cfg, geom["max_pos"]), 6000 + seed, geom["std"])
    for k in list(s_sd):
        if k.startswith("l0_module."):
            s_sd[k] = torch.from_numpy(fx["in.l0." + k[len("l0_module."):]]).clone()

    def tie_dec(sd):
        sd["text_decoder.cls.predictions.decoder.weight"] = sd["text_decoder.bert.embeddings.word_embeddings.weight"]
        sd["text_decoder.cls.predictions.decoder.bias"] = sd["text_decoder.cls.predictions.bias"]
        return sd
    s_sd = leafify(tie_dec(s_sd))
    t_sd = tie_dec(t_sd)
    batch = batch_from_fixture(fx)
    regen = synth.make_vqa_batch(geom, int(fx["meta.B"]), seed=seed)
    for k, v in regen.items():
        assert torch.equal(v, batch[k]), k
    assert list(fx["meta.l0_types"]) == list(O.L0_TYPES_VQA)
    logas = {k[len("l0_module."):]: v for k, v in s_sd.items() if k.endswith("_loga")}
    eps = {t: torch.from_numpy(fx[f"in.eps.{t}"]) for t in O.L0_TYPES_VQA}
    zs = O.l0_forward(logas, True, eps)
    S = O.vqa_forward(s_sd, s_cfg, batch, zs)
    with torch.no_grad():
        T = O.vqa_forward(t_sd, t_cfg, batch)
    for tag, out in (("student", S), ("teacher", T)):
        for dn in ("hidden_dict", "attention_dict", "cross_attention_dict"):
            for k, tup in out[dn].items():
                for i, t in enumerate(tup):
                    close(t, fx[f"{tag}.{k}.{i}"], rtol=1e-5, what=f"{tag}.{k}.{i}")
                assert f"{tag}.{k}.{len(tup)}" not in fx
        close(out["logits_dict"]["logits"], fx[f"{tag}.logits"], rtol=1e-5, atol=1e-5, what="logits")
        close(out["loss"], fx[f"{tag}.loss"], rtol=1e-5, what="loss")
    kd = O.vqa_kd_terms(S, T)
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], rtol=1e-5, what=f"kd.{k}")
    nd = s_cfg["text_layers"] - s_cfg["fusion_layer"]
    consts = O.l0_constants(geom["hidden"], geom["ffn"], geom["heads"], s_cfg["vit_layers"], s_cfg["fusion_layer"], nd, nd)
    assert consts["prunable"] == int(fx["meta.prunable_model_size"])
    lagr, exp_s, tgt = O.l0_lagrangian(logas, s_sd["l0_module.lambda_1"], s_sd["l0_module.lambda_2"], consts, 3,
                                       target_sparsity=0.25, lagrangian_warmup=10)
    close(lagr, fx["mix.lagrangian"], rtol=1e-5, what="lagrangian")
    close(exp_s, fx["mix.expected_sparsity"], rtol=1e-6, what="expected sparsity")
    assert abs(tgt - float(fx["mix.target_sparsity"])) < 1e-9
    total, mix = O.vqa_loss_mix(S["loss"], kd, lagr)
    for k, v in mix.items():
        close(v, fx[f"mix.{k}"], rtol=1e-5, what=f"mix.{k}")
    close(total, fx["mix.total"], rtol=1e-5, what="total")
    total.backward()
    check_grads(fx, "student", s_sd, True, 2e-4)


def test_retrieval_rerank_loop_matches_the_reference_evaluation_function(golden_dir):
    """oracle/retrieval_eval_oracle.py against score matrices produced by the REFERENCE's own Eff_Retrieval.evaluation
    (ast-extracted and run on the reference's EffXVLMforRetrieval by oracle/gen_golden.py: tests/golden/rerank_tiny.npz) -
    one rank and both shards of a 2-rank run, the -100 fill of never-rescored pairs included"""
    from oracle import retrieval_eval_oracle as RO
    fx = load(golden_dir, "rerank_tiny.npz")
    geom = synth.GEOMS["tiny"]
    seed = int(fx["meta.seed"])
    cfg = O.model_cfg(geom, "s")
    sch = schema.xvlm_schema(cfg, geom["max_pos"], mlm=False, bbox=False, l0=True)
    sd = schema.det_weights(sch, 5000 + seed, geom["std"])
    for n in list(sd):
        if n.startswith("l0_module."):
            sd[n] = torch.from_numpy(fx["in.l0." + n[len("l0_module."):]]).clone()
    ref_names = {k[5:] for k in fx if k.startswith("wchk.")}
    assert ref_names == set(sd), sorted(ref_names ^ set(sd))[:8]
    for n, (s, a) in checksums(sd).items():
        if not n.startswith("l0_module."):
            np.testing.assert_allclose([s, a], fx[f"wchk.{n}"], rtol=1e-9, atol=1e-9, err_msg=n)
    images, ids, atts = (torch.from_numpy(fx["in." + k]) for k in ("image", "text_ids", "text_atts"))
    logas = {k[len("l0_module."):]: v for k, v in sd.items() if k.endswith("_loga")}
    zs = O.l0_forward(logas, False)
    for rank, world in ((0, 1), (0, 2), (1, 2)):
        i2t, t2i = RO.evaluation_scores(sd, cfg, images, ids, atts, int(fx["meta.k_test"]), zs=zs, rank=rank, world=world,
                                        text_bs=4)
        for got, name in ((i2t, "i2t"), (t2i, "t2i")):
            want = fx[f"out.r{rank}w{world}.{name}"]
            assert np.array_equal(got.numpy() == -100.0, want == -100.0), (rank, world, name)
            close(got, want, rtol=1e-5, atol=1e-6, what=f"r{rank}w{world}.{name}")
