"""Step-level parity on the GPU: the drop-in modules (HIP kernels underneath) against
  (1) the golden vectors captured from the reference itself (tests/golden/gd_tiny.npz, gd_full.npz, itr_tiny.npz), and
  (2) the CPU oracle on fresh seeded inputs.
fp32 compute: losses / logits within 1e-4 relative (north_star), gradients within 1e-3 of their L2 norm.
bf16 compute: losses within 3e-2 relative of the fp32 reference values (bf16 has 8 significand bits).
"""
import math

import numpy as np
import pytest
import torch

from helpers import grad_parity_stats, batch_from_fixture, close, load_det_weights, load_fixture, model_config
from oracle import schema, synth
from oracle import xvlm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_gd(geom, seed, fx=None):
    from efficientvlm_amd.models.model_pretrain import XVLM
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = XVLM(model_config(geom, "s")), XVLM(model_config(geom, "t"))
    load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"]), 1000 + seed, geom["std"], fx, "student")
    load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"]), 2000 + seed, geom["std"], fx, "teacher")
    student.to(DEV).train()
    teacher.to(DEV).eval()
    for p in teacher.parameters():
        p.requires_grad_(False)
    return student, teacher


def run_gd(student, teacher, fx, dtype):
    from efficientvlm_amd import distill
    from efficientvlm_amd.runtime import compute
    batch = batch_from_fixture(fx, DEV)
    student.injected_neg_idx = torch.from_numpy(fx["in.student_neg_idx"])
    teacher.injected_neg_idx = torch.from_numpy(fx["in.teacher_neg_idx"])
    with compute(dtype):
        total, S, T, kd, mix = distill.gd_forward(student, teacher, batch)
        total.backward()
    return total, S, T, kd, mix


def check_against_fixture(fx, tag, out, full, rtol):
    for dn in ("hidden_dict", "attention_dict", "cross_attention_dict"):
        for k, tup in out[dn].items():
            for i, t in enumerate(tup):
                if full:
                    close(t.float(), fx[f"{tag}.{k}.{i}"], rtol, 1e-6, f"{tag}.{k}.{i}")
                else:
                    got = [float(t.double().sum()), float(t.double().pow(2).sum().sqrt())]
                    np.testing.assert_allclose(got, fx[f"{tag}.{k}.chk"][i], rtol=10 * rtol, atol=1e-3, err_msg=f"{tag}.{k}.{i}")
    for k, t in out["logits_dict"].items():
        if f"{tag}.{k}" in fx:
            close(t.float(), fx[f"{tag}.{k}"], rtol, 1e-5, f"{tag}.{k}")
        else:
            close(t.float().reshape(-1, t.shape[-1])[:4, :64], fx[f"{tag}.{k}.head"], rtol, 1e-5, f"{tag}.{k}")
    for k, t in out["loss"].items():
        close(t, fx[f"{tag}.{k}"], rtol, 0, f"{tag}.{k}")


@pytest.mark.parametrize("name,full,batched", [("gd_tiny.npz", True, True), ("gd_full.npz", False, True),
                                               ("gd_region_tiny.npz", True, True), ("gd_region_tiny.npz", True, False),
                                               ("gd_region_full.npz", False, True)])
def test_gd_step_fp32_matches_reference_vectors(name, full, batched):
    """general steps and REGION steps (idx_to_group_img / image_atts / bbox + giou; GeneralDistill.py:158-262), the
    latter through both the batched and the pass-by-pass forward"""
    fx = load_fixture(name)
    geom = synth.GEOMS[str(fx["meta.geom"])]
    student, teacher = build_gd(geom, int(fx["meta.seed"]), fx)
    student.batched_passes = teacher.batched_passes = batched
    total, S, T, kd, mix = run_gd(student, teacher, fx, torch.float32)
    check_against_fixture(fx, "student", S, full, 1e-4)
    check_against_fixture(fx, "teacher", T, full, 1e-4)
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], 1e-4, 1e-7, f"kd.{k}")
    for k, v in mix.items():
        close(v, fx[f"mix.{k}"], 1e-4, 0, f"mix.{k}")
    close(total, fx["mix.total"], 1e-4, 0, "total")
    n = 0
    for nme, p in student.named_parameters():
        key = f"student.grad_chk.{nme}"
        if key not in fx:
            continue
        ref_l2 = float(fx[key][1])
        got = float(p.grad.double().pow(2).sum().sqrt())
        assert abs(got - ref_l2) <= 1e-3 * ref_l2 + 2e-6, f"grad L2 {nme}: {got} vs {ref_l2}"
        if f"student.grad.{nme}" in fx:
            close(p.grad, fx[f"student.grad.{nme}"], 0, 1e-3 * ref_l2 + 2e-6, f"grad {nme}")
        elif f"student.grad_head.{nme}" in fx:
            close(p.grad.reshape(-1)[:64], fx[f"student.grad_head.{nme}"], 0, 1e-3 * ref_l2 + 2e-6, f"grad {nme}")
        n += 1
    assert n > 100


def test_gd_step_bf16_tracks_fp32_reference():
    fx = load_fixture("gd_full.npz")
    geom = synth.GEOMS["full"]
    student, teacher = build_gd(geom, int(fx["meta.seed"]))
    total, S, T, kd, mix = run_gd(student, teacher, fx, torch.bfloat16)
    for k, t in S["loss"].items():
        close(t, fx[f"student.{k}"], 3e-2, 0, k)
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], 6e-2, 1e-5, f"kd.{k}")
    close(total, fx["mix.total"], 3e-2, 0, "total")
    # gradient direction: cosine with the fp32 reference gradient heads
    cos_n = cos_d1 = cos_d2 = 0.0
    for nme, p in student.named_parameters():
        key = f"student.grad_head.{nme}"
        if key in fx and p.grad is not None:
            a = p.grad.reshape(-1)[:64].double().cpu()
            b = torch.from_numpy(fx[key]).double()
            cos_n += float((a * b).sum()); cos_d1 += float((a * a).sum()); cos_d2 += float((b * b).sum())
    assert cos_n / (cos_d1 ** 0.5 * cos_d2 ** 0.5) > 0.98


def test_itr_step_with_l0_fp32_matches_reference_vectors():
    from efficientvlm_amd import distill
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    fx = load_fixture("itr_tiny.npz")
    geom = synth.GEOMS["tiny"]
    seed = int(fx["meta.seed"])
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = EffXVLMforRetrieval(model_config(geom, "s")), TeacherITR(model_config(geom, "t"))
    load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 3000 + seed,
                     geom["std"], fx, "student")
    load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False), 4000 + seed, geom["std"],
                     fx, "teacher")
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.from_numpy(fx["in.l0." + n]))
    student.l0_module.set_lagrangian_warmup_steps(10)
    student.to(DEV).train()
    teacher.to(DEV).eval()
    batch = {k: torch.from_numpy(fx["in." + k]).to(DEV) for k in ("image", "text_ids", "text_atts")}
    idx = torch.from_numpy(fx["in.idx"]).to(DEV)
    student.l0_module.injected_eps = {t: torch.from_numpy(fx["in.eps." + t]) for t in O.L0_TYPES}
    student.injected_neg_idx = torch.from_numpy(fx["in.student_neg_idx"])
    teacher.injected_neg_idx = torch.from_numpy(fx["in.teacher_neg_idx"])
    S = student(batch["image"], batch["text_ids"], batch["text_atts"], idx=idx, output_attentions=True, output_hidden_states=True)
    with torch.no_grad():
        T = teacher(batch["image"], batch["text_ids"], batch["text_atts"], idx=idx, output_attentions=True, output_hidden_states=True)
    check_against_fixture(fx, "student", S, True, 1e-4)
    T["loss"] = {}
    check_against_fixture(fx, "teacher", T, True, 1e-4)
    kd = distill.kd_terms(S, T, with_cross_attn=True)
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], 1e-4, 1e-7, f"kd.{k}")
    lagr, es, ts = student.l0_module.lagrangian_regularization(3)
    close(lagr, fx["mix.lagrangian"], 1e-5, 1e-8, "lagrangian")
    total, mix = distill.itr_loss_mix(S["loss"], kd, lagr)
    close(total, fx["mix.total"], 1e-4, 0, "total")
    total.backward()
    n = 0
    for nme, p in student.named_parameters():
        key = f"student.grad_chk.{nme}"
        if key not in fx:
            continue
        ref_l2 = float(fx[key][1])
        got = float(p.grad.double().pow(2).sum().sqrt())
        assert abs(got - ref_l2) <= 1e-3 * ref_l2 + 2e-6, f"grad L2 {nme}: {got} vs {ref_l2}"
        if f"student.grad.{nme}" in fx:
            close(p.grad, fx[f"student.grad.{nme}"], 0, 1e-3 * ref_l2 + 2e-6, f"grad {nme}")
        n += 1
    assert n > 100
    # eval mode: deterministic masks (bit-exact) and the two losses
    student.eval()
    student.injected_neg_idx = torch.from_numpy(fx["eval.neg_idx"])
    with torch.no_grad():
        zs = student.l0_module.forward(training=False)
        for k, v in zs.items():
            assert np.array_equal(v.cpu().numpy(), fx["eval.z." + k]), k
        itc_e, itm_e = student(batch["image"], batch["text_ids"], batch["text_atts"], idx=idx)
    close(itc_e, fx["eval.loss_itc"], 1e-4, 0, "eval itc")
    close(itm_e, fx["eval.loss_itm"], 1e-4, 0, "eval itm")
    # physically pruned model: heads / FFN units with a zero gate removed, no gates at run time -> the
    # REFERENCE's masked-dense eval losses
    from efficientvlm_amd import pruning
    n_before = sum(p.numel() for p in student.parameters())
    with torch.no_grad():
        pruning.update_params(student, zs)
        pruning.prune_model_with_z(zs, student, pad_to=8)
        student.injected_neg_idx = torch.from_numpy(fx["eval.neg_idx"])
        itc_p, itm_p = pruning.retrieval_eval_losses(student, batch["image"], batch["text_ids"], batch["text_atts"], idx=idx)
    assert sum(p.numel() for p in student.parameters()) < n_before
    heads_kept = [l.self_attn.num_heads for l in student.vision_encoder.encoder.layers]
    assert heads_kept == [int(fx["eval.z.vision_head_z"][i].sum()) for i in range(len(heads_kept))]
    close(itc_p, fx["eval.loss_itc"], 1e-4, 0, "pruned itc")
    close(itm_p, fx["eval.loss_itm"], 1e-4, 0, "pruned itm")


def test_gd_step_matches_oracle_on_fresh_inputs():
    """oracle as the checker on inputs the fixtures do not cover (different seed, batch and padding pattern)"""
    from efficientvlm_amd import distill
    from efficientvlm_amd.models.model_pretrain import XVLM
    from efficientvlm_amd.runtime import compute
    geom = synth.GEOMS["tiny"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = XVLM(model_config(geom, "s")), XVLM(model_config(geom, "t"))
    s_sd = load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"]), 77, geom["std"])
    t_sd = load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"]), 78, geom["std"])
    student.to(DEV).train(); teacher.to(DEV).eval()
    batch = synth.make_batch(geom, 5, seed=123, ragged=True)
    g = torch.Generator().manual_seed(9)
    s_neg = torch.stack([torch.randperm(4, generator=g)[0] for _ in range(10)])
    s_neg = torch.tensor([(i % 5 + 1 + int(s_neg[i])) % 5 for i in range(10)])
    t_neg = torch.tensor([(i % 5 + 2) % 5 for i in range(10)])
    student.injected_neg_idx, teacher.injected_neg_idx = s_neg, t_neg
    with compute(torch.float32):
        total, S, T, kd, mix = distill.gd_forward(student, teacher, {k: v.to(DEV) for k, v in batch.items()})
    tie = lambda sd: {**sd, "text_encoder.cls.predictions.decoder.weight": sd["text_encoder.bert.embeddings.word_embeddings.weight"],
                      "text_encoder.cls.predictions.decoder.bias": sd["text_encoder.cls.predictions.bias"]}
    with torch.no_grad():
        ototal, oS, oT, okd, omix = O.gd_step(tie(s_sd), tie(t_sd), s_cfg, t_cfg, batch, s_neg, t_neg)
    close(total, ototal, 1e-4, 0, "total")
    for k in kd:
        close(kd[k], okd[k], 1e-4, 1e-7, k)
    for k in S["loss"]:
        close(S["loss"][k], oS["loss"][k], 1e-4, 0, k)
    close(S["logits_dict"]["mlm_logits"].float(), oS["logits_dict"]["mlm_logits"], 1e-4, 1e-5, "mlm_logits")


def _tie(sd):
    return {**sd, "text_encoder.cls.predictions.decoder.weight": sd["text_encoder.bert.embeddings.word_embeddings.weight"],
            "text_encoder.cls.predictions.decoder.bias": sd["text_encoder.cls.predictions.bias"]}


@pytest.mark.parametrize("B", [64])
def test_benchmarked_configuration_matches_the_oracle(B, monkeypatch):
    """The configuration bench.py times - bf16, batch 64, full geometry, hipGraph replay, teacher pipelined one batch
    ahead, deferred grouped weight gradients, the 256 x 256 GEMM routing - held to the fp32 CPU oracle on the same
    weights, batch and hard negatives (GeneralDistill.py:286-376): step-0 losses, every KD term, and the gradients the
    optimiser consumes (read back from its flat slabs after the graph replay).  bf16 storage has 8 significand bits:
    losses within 1e-3 (measured 1e-6 ... 2e-5), every KD term within 1e-2 / the logit terms 5e-2 (measured <= 2.4e-4 /
    1.2e-2: since round 3 the attention backward rebuilds the probabilities in fp32 and the fused map term is formed from
    them), the whole gradient at cosine > 0.9999 with the oracle's, per-tensor bounds in the body."""
    from efficientvlm_amd import ops
    from efficientvlm_amd.trainer import GDTrainer
    geom = synth.GEOMS["full"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = build_gd(geom, 21)
    s_sd = schema.det_weights(schema.xvlm_schema(s_cfg, geom["max_pos"]), 1000 + 21, geom["std"])
    t_sd = schema.det_weights(schema.xvlm_schema(t_cfg, geom["max_pos"]), 2000 + 21, geom["std"])
    batch = synth.make_batch(geom, B, seed=77, ragged=True)
    g = torch.Generator().manual_seed(5)
    s_neg = torch.cat([(torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=g)) % B for _ in range(2)])
    t_neg = torch.cat([(torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=g)) % B for _ in range(2)])
    student.injected_neg_idx, teacher.injected_neg_idx = s_neg, t_neg
    student.keep_injected_neg = teacher.keep_injected_neg = True      # warm-up steps and captured graphs included
    tr = GDTrainer(student, teacher, lr=1e-4, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.bfloat16,
                   use_graph=True, pipeline_teacher=True)
    gb = {k: v.to(DEV) for k, v in batch.items()}
    fused_ln = []                                 # (round 5: the ViT hidden-state term is formed inside the LayerNorm kernels)
    orig_fork_kd = ops.layer_norm_fork_kd
    monkeypatch.setattr(ops, "layer_norm_fork_kd", lambda *a, **k: (fused_ln.append(1), orig_fork_kd(*a, **k))[1])
    assert tr.step(gb) is None                    # primes the teacher pipeline
    out = tr.step(gb)                             # student step on the first batch, replayed from the joint hipGraph
    torch.cuda.synchronize()
    assert tr._joint, "the step did not run from a captured graph"
    assert len(fused_ln) >= 6, "the image hidden-state distillation did not run fused (one LayerNorm per ViT layer)"
    got = [float(x) for x in out.tolist()]
    got_kd = {k: float(v) for k, v in tr.last_kd.items()}
    got_grad = {n: p.grad.detach().float().cpu().clone() for n, p in student.named_parameters()}

    leaves = {k: v.clone().requires_grad_(True) for k, v in s_sd.items()}
    ototal, oS, _, okd, omix = O.gd_step(_tie(leaves), _tie(t_sd), s_cfg, t_cfg, batch, s_neg, t_neg)
    ototal.backward()
    want = [float(ototal), float(oS["loss"]["loss_itc"]), float(oS["loss"]["loss_itm"]), float(oS["loss"]["loss_mlm"]),
            float(omix["loss_kd"])]
    for name, a, b in zip(("total", "itc", "itm", "mlm", "kd"), got, want):
        assert abs(a - b) <= 1e-3 * abs(b), f"{name}: {a} vs oracle {b}"
    assert set(got_kd) == {k for k, v in okd.items() if torch.is_tensor(v)}
    for k, v in got_kd.items():
        # (logit terms: KL divergences of nearly identical distributions formed from bf16 logits; the text-side map terms
        # still read bf16-stored maps, the image-map term is formed in-kernel from fp32 probabilities)
        rt = 5e-2 if k.endswith("_logits") else 1e-2
        assert abs(v - float(okd[k])) <= rt * abs(float(okd[k])) + 1e-6, f"kd.{k}: {v} vs oracle {float(okd[k])}"
    stats, num, da, db = [], 0.0, 0.0, 0.0
    gmax = max(float(l.grad.norm()) for l in leaves.values() if l.grad is not None)
    for name, leaf in leaves.items():
        # (key biases: softmax is invariant to them, their exact gradient is 0 up to fp32 rounding - nothing to compare)
        if leaf.grad is None or name not in got_grad or float(leaf.grad.norm()) < 1e-5 * gmax:
            continue
        a, b = got_grad[name].double().reshape(-1), leaf.grad.double().reshape(-1)
        stats.append((float((a - b).norm() / b.norm()), float((a * b).sum() / (a.norm() * b.norm())), name))
        num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
    assert len(stats) > 150
    worst = sorted(stats, reverse=True)[:5]
    import json, os
    if os.environ.get("EVLM_DUMP_GRAD_STATS"):
        with open(os.environ["EVLM_DUMP_GRAD_STATS"], "w") as f:
            json.dump({"global_cos": num / math.sqrt(da * db), "stats": sorted(stats, reverse=True),
                       "loss_rel_err": {n: abs(a - b) / abs(b) for n, a, b in zip(("total", "itc", "itm", "mlm", "kd"), got, want)},
                       "kd_rel_err": {k: abs(v - float(okd[k])) / (abs(float(okd[k])) + 1e-12) for k, v in got_kd.items()}}, f)
    # The whole gradient the optimiser sees, and every tensor individually.  Measured on MI355X (profiles/r03_grad_parity.json;
    # r02 in brackets, when the backward read bf16-stored probabilities): global cosine 0.999988 (0.99988), median relative L2
    # error 0.55 % (1.6 %), 90th percentile 1.1 % (14.8 %), worst tensor 17 % (36 %).  The noisiest tensors are now the two
    # ITC projection heads (13-17 %: their gradient is (softmax - labels) of logits that amplify the bf16 noise of the CLS
    # rows by 1 / temp = 14) and what sits behind all six ViT layers (class / position embeddings 11-13 %, layer-0 query /
    # key projections 8-9 %, every other query / key projection <= 5.3 %).
    rels = sorted(r for r, _, _ in stats)
    assert num / math.sqrt(da * db) > 0.9999, (num / math.sqrt(da * db), worst)
    assert all(r < 0.20 and c > 0.97 for r, c, _ in stats), worst      # (measured worst 17 %: a regression shows)
    qk = [r for r, _, n in stats if any(t in n for t in ("q_proj", "k_proj", ".query.", ".key."))]
    assert max(qk) < 0.12 and sorted(qk)[len(qk) // 2] < 0.02, sorted(qk)[-3:]
    assert rels[len(rels) // 2] < 0.01 and rels[int(0.9 * len(rels))] < 0.03, (rels[len(rels) // 2], rels[int(0.9 * len(rels))])

    # the routing this configuration is benchmarked with: the dominant kernels must be the ones that served this step
    ops.GEMM_PROFILE = []
    tr.opt.set_schedule(0.0)
    tr._step_eager(gb)
    torch.cuda.synchronize()
    recs, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    served = {}
    for rec in recs:
        served[rec[-1]] = served.get(rec[-1], 0) + 1
    # (the three tile flavours of the ping-pong GEMM - 256 / 192 / 128 rows, picked per shape - serve the large products)
    pp = {k: served.get(k, 0) for k in ("gemm_bf16_pp256_kernel<false,false,0>", "gemm_bf16_pp192_kernel<false>",
                                        "gemm_bf16_pp128_kernel<false>")}
    assert sum(pp.values()) >= 190 and all(v >= 30 for v in pp.values()), served
    assert served.get("gemm_bf16_pp256_grouped_kernel", 0) >= 2, served


@pytest.mark.gpu
def test_forward_phases_generator_equals_forward():
    """XVLM.forward_phases (the batched forward as a generator over its phases - what lets a trainer issue the pipelined
    teacher's image encoder and its text / fusion passes in two different hipGraph segments, and a trainer fork the teacher's
    branch at a phase of the student's forward) yields "vision_done", "text_done", the fusion layers, "fusion_done" and returns
    forward()'s dict, tensor for tensor"""
    from efficientvlm_amd import distill
    from efficientvlm_amd.runtime import compute
    geom = synth.GEOMS["tiny"]
    _, teacher = build_gd(geom, 5)
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, 4, seed=17).items()}
    teacher.injected_neg_idx = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
    teacher.keep_injected_neg = True
    with torch.no_grad(), compute(torch.float32):
        ref = teacher(batch["image"], batch["text_ids"], batch["text_atts"], **distill.model_kwargs(batch))
        gen = teacher.forward_phases(batch["image"], batch["text_ids"], batch["text_atts"], **distill.model_kwargs(batch))
        names = []
        try:
            while True:
                names.append(next(gen))
        except StopIteration as done:
            out = done.value
    # (round 6: the fusion pass is resumable LAYER BY LAYER - "fusion_layer_<i>" behind layer i of the teacher's 6 + 6 layers)
    assert [n for n in names if not n.startswith("fusion_layer_")] == ["vision_done", "text_done", "fusion_done"]
    assert [n for n in names if n.startswith("fusion_layer_")] == ["fusion_layer_%d" % i for i in range(6, 12)]
    assert names.index("fusion_layer_6") > names.index("text_done") and names.index("fusion_layer_11") < names.index("fusion_done")
    a, b = list(distill._tensors(ref)), list(distill._tensors(out))
    assert len(a) == len(b) and len(a) > 20
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)


_DP_SCRIPT = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth
from test_step_gpu import build_gd
from efficientvlm_amd import ops
from efficientvlm_amd.trainer import GDTrainer
dp = bool(os.environ.get("EVLM_FORCE_REDUCE"))
if dp:
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
geom = synth.GEOMS["tiny"]
student, teacher = build_gd(geom, 7)
tr = GDTrainer(student, teacher, lr=1e-3, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.float32,
               use_graph=False)
batch = {k: v.cuda() for k, v in synth.make_batch(geom, 4, seed=3).items()}
ops.dropout_seed(0)        # hard-negative draws (device Philox stream)
out = [tr.step(batch).tolist() for _ in range(3)]
assert tr.reducer.active == dp and (not dp or (len(tr._stages) == 4 and tr._sent == 4))
torch.cuda.synchronize()
if dp:
    dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_data_parallel_code_path_on_one_gpu_matches_plain_step():
    """The N>1 path (RCCL group, early-segment all-reduce launched from the backward hook on a side stream, eager
    launch) with world_size 1 must reproduce the plain trainer: a mean all-reduce over one rank is the identity."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(dp):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
        env.pop("EVLM_FORCE_REDUCE", None)
        if dp:
            env["EVLM_FORCE_REDUCE"] = "1"
        r = subprocess.run([sys.executable, "-c", _DP_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
        return np.array(json.loads(line[7:]))

    a, b = run(False), run(True)
    assert np.allclose(a, b, rtol=2e-4, atol=1e-5), (a, b)


_DP_SEG_SCRIPT = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth
from test_step_gpu import build_gd
from efficientvlm_amd.trainer import GDTrainer
dp = bool(os.environ.get("EVLM_FORCE_REDUCE"))
if dp:
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
geom = synth.GEOMS["tiny"]
student, teacher = build_gd(geom, 9)
neg = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
student.injected_neg_idx = teacher.injected_neg_idx = neg
student.keep_injected_neg = teacher.keep_injected_neg = True
tr = GDTrainer(student, teacher, lr=1e-3, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.float32,
               use_graph=True, pipeline_teacher=True)
batches = [{k: v.cuda() for k, v in synth.make_batch(geom, 4, seed=3 + i).items()} for i in range(3)]
out = []
for i in range(6):
    o = tr.step(batches[i % 3])
    if o is not None:
        out.append(o.tolist())
torch.cuda.synchronize()
if dp:
    assert tr._seg and not getattr(tr, "_segments_broken", False), "no segmented graphs"
    for sg in tr._seg.values():       # forward cut at the ITC gather, backward cut at the three gradient-stage hooks
        kinds = [s[0] for s in sg["segs"]]
        assert kinds == ["graph", "gather", "graph", "reduce", "graph", "reduce", "graph", "reduce", "graph", "reduce",
                         "graph"], kinds
    dist.destroy_process_group()
else:
    assert tr._joint
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("split", ["1", "0", "late"])
def test_segmented_graph_step_of_the_multi_gpu_path_matches_the_single_gpu_graph(split):
    """N > 1 code path with world_size 1 (RCCL group of one rank, the ITC all-gather forced through the collective): the
    student step replayed as hipGraph segments around the gather and the gradient all-reduce (trainer._student_segmented)
    must train exactly like the single-GPU joint graph - same losses over five optimiser steps on rotating batches.
    split: the pipelined teacher forward in two halves (image encoder in the first segment, text / fusion passes in the
    second: the forward suspended at its "vision_done" phase between two captures), whole behind the gather, or "late" - the
    image encoder behind the gather and the text pass / fusion pass / heads each beside one segment of the ViT backward (the
    forward suspended at three phases across four captures; an opt-in experiment, profiles/r05_exchange_overlap.md)"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(dp):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", EVLM_SEG_TEACHER_SPLIT=split)
        env.pop("EVLM_FORCE_REDUCE", None)
        if dp:
            env["EVLM_FORCE_REDUCE"] = "1"
        r = subprocess.run([sys.executable, "-c", _DP_SEG_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
        return np.array(json.loads(line[7:]))

    a, b = run(False), run(True)
    assert a.shape == b.shape and a.shape[0] == 5
    assert np.allclose(a, b, rtol=3e-4, atol=1e-5), (a, b)


_DP2_SCRIPT = r"""
import os, sys, json, hashlib, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth
from test_step_gpu import build_gd
from efficientvlm_amd.trainer import GDTrainer
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)                         # both ranks share the one GPU of the box: gloo moves the bytes
dist.init_process_group("gloo", rank=rank, world_size=2)
geom = synth.GEOMS["tiny"]
student, teacher = build_gd(geom, 9 + 100 * rank)           # DIFFERENT initial students: the constructor's broadcast must level them
neg = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
student.injected_neg_idx = teacher.injected_neg_idx = neg
student.keep_injected_neg = teacher.keep_injected_neg = True
tr = GDTrainer(student, teacher, lr=1e-3, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.float32,
               use_graph=True, pipeline_teacher=True)
assert tr.world == 2 and tr.reducer.active
batches = [{k: v.cuda() for k, v in synth.make_batch(geom, 4, seed=3 + i + 50 * rank).items()} for i in range(3)]
out = []
for i in range(6):
    o = tr.step(batches[i % 3])
    if o is not None:
        out.append(o.tolist())
torch.cuda.synchronize()
segmented = bool(tr._seg) and not getattr(tr, "_segments_broken", False)
flat = torch.cat([g["p"].detach().reshape(-1).float().cpu() for g in tr.opt.groups])
digest = hashlib.sha256(flat.numpy().tobytes()).hexdigest()
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"rank": rank, "out": out, "segmented": segmented, "digest": digest, "pnorm": float(flat.norm())}))
"""


def test_two_ranks_train_in_lockstep_through_the_segmented_multi_gpu_step():
    """TWO ranks (two processes on the one GPU of the box, gloo carrying the collectives - RCCL refuses two ranks on one
    device) through the shipping N > 1 path: parameter broadcast from rank 0 at construction (the students are built from
    different seeds), the step as hipGraph segments around a REAL two-rank ITC gather and staged gradient all-reduces, the
    pipelined teacher in two halves, rank agreement on the capture.  After five optimiser steps on rank-specific batches
    the parameter slabs of the two ranks are bit-identical (every rank applied the same averaged gradients to the same
    parameters), and each rank's loss trajectory equals the one it gets from the eager fallback path
    (EVLM_NO_SEGMENT_GRAPHS=1: hooks send the stages) - the two forms that must stay interchangeable rank by rank"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(port, eager):
        procs = []
        for rank in (0, 1):
            env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo")
            env.pop("EVLM_FORCE_REDUCE", None)
            if eager:
                env["EVLM_NO_SEGMENT_GRAPHS"] = "1"
            procs.append(subprocess.Popen([sys.executable, "-c", _DP2_SCRIPT], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        res = []
        for p in procs:
            try:
                so, se = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise AssertionError("a rank hung (collective sequences of the two ranks differ?)")
            assert p.returncode == 0, se[-3000:]
            res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
        return sorted(res, key=lambda r: r["rank"])

    seg = run(29571, eager=False)
    assert all(r["segmented"] for r in seg), "the segmented form was not used"
    assert seg[0]["digest"] == seg[1]["digest"], (seg[0]["pnorm"], seg[1]["pnorm"])
    assert len(seg[0]["out"]) == 5 and np.isfinite(np.array(seg[0]["out"])).all()
    eag = run(29573, eager=True)
    assert not any(r["segmented"] for r in eag)
    assert eag[0]["digest"] == eag[1]["digest"]
    for a, b in zip(seg, eag):
        assert np.allclose(np.array(a["out"]), np.array(b["out"]), rtol=3e-4, atol=1e-5), (a["out"], b["out"])


_DP2_ORACLE_SCRIPT = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth, schema
from oracle import xvlm_oracle as O
from helpers import load_det_weights, model_config
from test_dp_cpu import _gd_problem, _B_HALF
from efficientvlm_amd.models.model_pretrain import XVLM
from efficientvlm_amd.trainer import GDTrainer
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
geom = synth.GEOMS["tiny"]
_, s_cfg, t_cfg, s_sd, t_sd, batch, neg_local = _gd_problem()
student, teacher = XVLM(model_config(geom, "s")), XVLM(model_config(geom, "t"))
load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"]), 31, geom["std"])
load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"]), 32, geom["std"])
student.cuda().train(); teacher.cuda().eval()
student.injected_neg_idx = teacher.injected_neg_idx = neg_local[rank]
student.keep_injected_neg = teacher.keep_injected_neg = True
use_graph = bool(int(os.environ["EVLM_TEST_GRAPH"]))
wire = torch.bfloat16 if os.environ.get("EVLM_TEST_WIRE") == "bf16" else None
tr = GDTrainer(student, teacher, lr=0.0, weight_decay=0.0, max_grad_norm=0.0, dtype=torch.float32, use_graph=use_graph,
               pipeline_teacher=use_graph, grad_compress=wire)
lo, hi = rank * _B_HALF, (rank + 1) * _B_HALF
half = {k: v[lo:hi].cuda() for k, v in batch.items()}
out = tr.step(half)
if out is None:                                   # pipelined: the first call primed the teacher, the second trains on `half`
    out = tr.step(half)
torch.cuda.synchronize()
grads = {k: p.grad.detach().float().cpu() for k, p in student.named_parameters() if p.grad is not None}
torch.save({"grads": grads, "losses": out.tolist(),
            "segmented": bool(getattr(tr, "_seg", None)) and not getattr(tr, "_segments_broken", False)},
           os.environ["EVLM_TEST_OUT"] + f".{rank}")
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("use_graph,wire", [(False, "fp32"), (True, "fp32"), (True, "bf16")])
def test_two_rank_hip_gradients_equal_the_oracles_global_batch_gradient(use_graph, wire, tmp_path):
    """The data-parallel semantics of tests/test_dp_cpu.py (there: oracle forward + the product's gather and reducer) with
    the PRODUCT doing all of it on the GPU: two ranks (two processes on the one GPU, gloo), each running GDTrainer's N > 1
    step (eager with hook-driven stages / hipGraph segments with the pipelined teacher) on its half of a global batch -
    the rank-averaged gradient left in the slabs must equal the CPU oracle's single-process gradient on the WHOLE batch,
    with the reference's ITC term reaching the encoders scaled by 1 / world (its slice-only gather backward,
    efficient_models/xvlm.py:54-74); every rank reports the global ITC loss.  wire = bf16: the opt-in compressed exchange
    (divide by world in fp32, cast, sum, cast back) - the same check at the bound bf16 rounding allows"""
    import os, subprocess, sys
    from test_dp_cpu import _gd_problem, _tied, _B_HALF
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outp = str(tmp_path / "dp2")
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29581 + 2 * int(use_graph) + 4 * int(wire == "bf16")),
                   RANK=str(rank), WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo", EVLM_TEST_OUT=outp, EVLM_TEST_GRAPH=str(int(use_graph)),
                   EVLM_TEST_WIRE=wire)
        env.pop("EVLM_FORCE_REDUCE", None)
        procs.append(subprocess.Popen([sys.executable, "-c", _DP2_ORACLE_SCRIPT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung")
        assert p.returncode == 0, se[-3000:]
    res = [torch.load(outp + f".{r}") for r in (0, 1)]
    assert all(r["segmented"] == use_graph for r in res)
    # the oracle on the whole batch, ITC gradient into the features scaled by 1 / world (as in test_dp_cpu.py)
    Oo, s_cfg, t_cfg, s_sd, t_sd, batch, neg_local = _gd_problem()
    names = sorted(s_sd)
    leaves = {k: s_sd[k].clone().requires_grad_(True) for k in names}
    B = _B_HALF
    neg = torch.cat([neg_local[0][:B], neg_local[1][:B] + B, neg_local[0][B:], neg_local[1][B:] + B])
    S = Oo.pretrain_forward(_tied(leaves), s_cfg, batch, neg)
    with torch.no_grad():
        T = Oo.pretrain_forward(_tied(t_sd), t_cfg, batch, neg)
    loss = dict(S["loss"])
    i_feat, t_feat = S["features"]
    half_grad = lambda x: x * 0.5 + x.detach() * 0.5
    loss["loss_itc"] = Oo.contrastive_loss(half_grad(i_feat), half_grad(t_feat), leaves["temp"].clamp(0.001, 0.5))
    total, _ = Oo.gd_loss_mix(loss, Oo.kd_terms(S, T))
    total.backward()
    common = [k for k in names if k in res[0]["grads"] and leaves[k].grad is not None]
    covered = sum(leaves[k].numel() for k in common) / sum(leaves[k].numel() for k in names if leaves[k].grad is not None)
    assert covered > 0.99, covered
    ref = torch.cat([leaves[k].grad.reshape(-1) for k in common])
    for r in res:
        g = torch.cat([r["grads"][k].reshape(-1) for k in common])
        err = float((g - ref).norm() / ref.norm())
        assert err < (1e-4 if wire == "fp32" else 4e-3), err
        itc = float(S["loss"]["loss_itc"].detach())
        assert abs(r["losses"][1] - itc) < 1e-5 * max(1.0, abs(itc))
    assert all(torch.equal(res[0]["grads"][k], res[1]["grads"][k]) for k in common), "ranks disagree after the exchange"


_DP2_ITR_SCRIPT = r"""
import os, sys, json, hashlib, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth, schema
from oracle import xvlm_oracle as O
from helpers import load_det_weights, model_config
from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
from efficientvlm_amd.efficient_models.xvlm import XVLMBase
from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
from efficientvlm_amd.trainer import ITRTrainer
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
def fixed_negatives(self, image_feat, text_feat, idx):
    bs = image_feat.size(0)
    ar = torch.arange(bs, device=image_feat.device)
    return (ar + 1) % bs, (ar + 2) % bs
XVLMBase._sample_negatives = fixed_negatives
geom = synth.GEOMS["tiny"]
s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
B = 4
# EVLM_TEST_RECIPE=1: the reference's real recipe - stock BERT dropout 0.1 on the student (each rank its own mask stream) and
# batches in 'longest' padding whose real text length differs per rank and step, fed through data.bucket_pad_itr
RECIPE = bool(os.environ.get("EVLM_TEST_RECIPE"))
student, teacher = EffXVLMforRetrieval(model_config(geom, "s", dropout=0.1 if RECIPE else 0.0)), TeacherITR(model_config(geom, "t"))
load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 51 + 7 * rank, geom["std"])
load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False), 52, geom["std"])
gen = torch.Generator().manual_seed(8 + rank)                # rank-specific gate parameters too: the broadcast levels them
with torch.no_grad():
    for n, p in student.l0_module.named_parameters():
        p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
student.l0_module.set_lagrangian_warmup_steps(10)
student.cuda(); teacher.cuda()
tr = ITRTrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=True, use_graph=True)
assert tr.reducer.active and tr.reducer.world == 2
batches = [{k: v.cuda() for k, v in synth.make_batch(geom, B, seed=30 + i + 40 * rank, ragged=True).items()
            if k in ("image", "text_ids", "text_atts")} for i in range(3)]
if RECIPE:
    from efficientvlm_amd import ops
    from efficientvlm_amd.data import bucket_pad_itr
    ops.dropout_seed(100 + rank)
    batches = [bucket_pad_itr({k: v.cuda() for k, v in synth.make_batch(dict(geom, L=5 + (i + 2 * rank) % 4, M=2), B, seed=30 + i + 40 * rank,
                                                                         ragged=True).items() if k in ("image", "text_ids", "text_atts")})
               for i in range(3)]
    assert len({int(b["extents"][0]) for b in batches}) == 3 and all(b["text_ids"].shape[1] == 16 for b in batches)
idx = torch.arange(B).cuda() + B * rank
out, launches = [], []
for c in range(8):
    # (each rank draws its own gate noise, as each reference process does)
    student.l0_module.injected_eps = {t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6)
                                      for t in O.L0_TYPES}
    o = tr.step(batches[c % 3], idx=idx)
    if o is not None:
        out.append(o.tolist())
        launches.append(tr.last_launch)
torch.cuda.synchronize()
h = hashlib.sha256()
for k, v in sorted(student.state_dict().items()):
    h.update(v.detach().float().cpu().numpy().tobytes())
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"rank": rank, "out": out, "digest": h.hexdigest(), "launches": launches}))
"""


@pytest.mark.parametrize("recipe", [False, True], ids=["fixed shapes, p = 0", "bucket-padded ragged batches, dropout 0.1"])
def test_two_ranks_of_the_itr_pruning_step_stay_bit_identical(recipe):
    """the ITR pruning fine-tune (ITRTrainer: three optimisers, L0 gate parameters and Lagrange multipliers travelling in the
    LAST gradient stage, teacher prefetched through hipGraphs) on TWO ranks - two processes on the one GPU, gloo: students
    and gate parameters built differently per rank are levelled by the constructor's broadcast, every rank draws its own
    gate noise and batches; the student step replays as hipGraph SEGMENTS around its collectives (two gathers - ITC features,
    image ids - and the staged all-reduces: first step per prefetch parity eager, second captured, then replays); after seven
    optimiser steps every student tensor (gates and multipliers included) is bit-identical on the two ranks.  recipe (round 6):
    the same under the reference's real recipe - dropout 0.1 on the student with a mask stream per rank, batches whose real
    text length differs per rank and step, bucket-padded: one segment chain per prefetch parity serves them all"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK=str(rank), WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo")
        env.pop("EVLM_FORCE_REDUCE", None)
        env.pop("EVLM_TEST_RECIPE", None)
        if recipe:
            env["EVLM_TEST_RECIPE"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-c", _DP2_ITR_SCRIPT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (collective sequences of the two ranks differ?)")
        assert p.returncode == 0, se[-3000:]
        res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert len(res[0]["out"]) == 7 and np.isfinite(np.array(res[0]["out"])).all()
    assert res[0]["digest"] == res[1]["digest"]
    for r in res:
        assert r["launches"] == ["eager", "eager"] + ["hipGraph segments"] * 5, r["launches"]


_DP2_VQA_SCRIPT = r"""
import os, sys, json, hashlib, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth
from oracle import xvlm_oracle as O
from test_step_gpu import _vqa_models
from efficientvlm_amd.trainer import VQATrainer
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
geom = synth.GEOMS["tiny"]
student, teacher, *_ = _vqa_models(geom, 51 + 9 * rank, 52)
gen = torch.Generator().manual_seed(4 + rank)
with torch.no_grad():
    for n, p in student.l0_module.named_parameters():
        p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
student.l0_module.set_lagrangian_warmup_steps(10)
student.cuda(); teacher.cuda()
tr = VQATrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=True)
assert tr.reducer.active and tr.reducer.world == 2
out, launches = [], []
for c in range(12):
    # answer rows are part of a batch's shape (synth.make_vqa_batch: 7 / 8 / 9 rows for seed % 3 = 0 / 1 / 2).  Rank 0 sees one
    # batch kind throughout; rank 1 changes kind after three calls - so the two ranks reach their eager first steps and
    # their captures at DIFFERENT steps, and replay segments beside a peer that steps eagerly
    seed = 21 + 3 * c + 30 * rank + (1 if (rank == 1 and c >= 3) else 0)
    batch = {k: v.cuda() for k, v in synth.make_vqa_batch(geom, 4, seed=seed).items()}
    student.l0_module.injected_eps = {t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6)
                                      for t in O.L0_TYPES_VQA}
    o = tr.step(batch)
    if o is not None:
        out.append(o.tolist())
        launches.append(tr.last_launch)
torch.cuda.synchronize()
h = hashlib.sha256()
for k, v in sorted(student.state_dict().items()):
    h.update(v.detach().float().cpu().numpy().tobytes())
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"rank": rank, "out": out, "digest": h.hexdigest(), "launches": launches}))
"""


def test_two_ranks_of_the_vqa_pruning_step_stay_bit_identical():
    """the VQA pruning fine-tune (VQATrainer: causal answer decoder, VQAL0Module gates, three optimisers) on TWO ranks - two
    processes on the one GPU, gloo; differently built students levelled by the broadcast, rank-specific batches and gate
    noise, teacher prefetched, the student step replayed as hipGraph segments around the staged all-reduces.  The ranks see
    DIFFERENT batch kinds (answer rows), so one replays segments while the other still steps eagerly or captures - which is
    safe only because both forms issue the same collective sequence and no collective is held at capture time: after eleven
    optimiser steps every student tensor is bit-identical on the two ranks"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK=str(rank), WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo")
        env.pop("EVLM_FORCE_REDUCE", None)
        procs.append(subprocess.Popen([sys.executable, "-c", _DP2_VQA_SCRIPT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (collective sequences of the two ranks differ?)")
        assert p.returncode == 0, se[-3000:]
        res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert len(res[0]["out"]) == 11 and np.isfinite(np.array(res[0]["out"])).all()
    assert res[0]["digest"] == res[1]["digest"]
    by_rank = {r["rank"]: r["launches"] for r in res}
    assert by_rank[0] == ["eager", "eager"] + ["hipGraph segments"] * 9, by_rank[0]
    # rank 1: kind A for three calls (prime, eager, eager), then kind B: the waiting kind-A batch steps on a captured
    # pair, kind B's pairs start eagerly two steps after rank 0 had begun to replay
    assert by_rank[1][-1] == "hipGraph segments" and by_rank[1] != by_rank[0], by_rank[1]


_DP_SEQ_SCRIPT = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth
from test_step_gpu import build_gd
from efficientvlm_amd import runtime
from efficientvlm_amd.trainer import GDTrainer
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
geom = synth.GEOMS["tiny"]
student, teacher = build_gd(geom, 9)
tr = GDTrainer(student, teacher, dtype=torch.bfloat16, use_graph=True, pipeline_teacher=True)
batches = [{k: v.cuda() for k, v in synth.make_batch(geom, 4, seed=3 + i).items()} for i in range(3)]
seqs = []
inner, streams = tr._step_on_current, []
tr._step_on_current = lambda b, m: (streams.append(torch.cuda.current_stream() != torch.cuda.default_stream()), inner(b, m))[1]
for i in range(7):
    runtime.COLLECTIVES = []
    tr.step(batches[i % 3])                    # (called on the default stream, as a plain training loop does)
    seqs.append(runtime.COLLECTIVES)
torch.cuda.synchronize()
runtime.COLLECTIVES = None
segmented = bool(tr._seg) and not getattr(tr, "_segments_broken", False)
slab = sum(g.numel() for g in tr.opt.flat_grads)
dist.destroy_process_group()
print("RESULT " + json.dumps({"seqs": seqs, "segmented": segmented, "slab": slab, "off_default_stream": all(streams) and len(streams) == 7}))
"""


def test_segmented_and_eager_multi_gpu_steps_issue_the_same_collective_sequence():
    """N > 1 code path on one GPU (RCCL group of one rank): the student step replayed as hipGraph segments and the eager
    student step (what every rank falls back to when a capture fails anywhere) must issue the SAME collectives in the
    same order with the same sizes and wire dtype - ranks in different modes would otherwise hang each other - and the
    all-reduces of one step cover the gradient slabs exactly once"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(eager):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", EVLM_FORCE_REDUCE="1")
        env.pop("EVLM_NO_SEGMENT_GRAPHS", None)
        if eager:
            env["EVLM_NO_SEGMENT_GRAPHS"] = "1"
        r = subprocess.run([sys.executable, "-c", _DP_SEQ_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])

    seg, eag = run(False), run(True)
    assert seg["segmented"] and not eag["segmented"]
    # with a live reducer the step never runs on the legacy default stream: on this stack every process-group collective
    # makes THAT stream wait for the group's stream, which serialises the gradient exchange with backward
    # (profiles/r05_exchange_overlap.md)
    assert seg["off_default_stream"] and eag["off_default_stream"]
    assert seg["seqs"][0] == eag["seqs"][0]        # the priming call: two eager warm-up steps in both modes
    for a, b in zip(seg["seqs"][1:], eag["seqs"][1:]):
        assert a == b and len(a) >= 5, (a, b)
        assert a[0][0] == "all_gather" and all(k == "all_reduce" for k, _, _ in a[1:])
        assert all(d == "torch.float32" for _, _, d in a[1:])       # default wire: fp32
        assert sum(n for _, n, _ in a[1:]) == seg["slab"]


_DP_PRUNE_SEQ_SCRIPT = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth, schema
from oracle import xvlm_oracle as O
from helpers import load_det_weights, model_config
from efficientvlm_amd import runtime
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
geom = synth.GEOMS["tiny"]
kind = os.environ["EVLM_TEST_KIND"]
gen = torch.Generator().manual_seed(8)
if kind == "itr":
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    from efficientvlm_amd.trainer import ITRTrainer
    from efficientvlm_amd.efficient_models.xvlm import XVLMBase
    def fixed_negatives(self, image_feat, text_feat, idx):        # (the device draws differ from process to process)
        bs = image_feat.size(0)
        ar = torch.arange(bs, device=image_feat.device)
        return (ar + 1) % bs, (ar + 2) % bs
    XVLMBase._sample_negatives = fixed_negatives
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = EffXVLMforRetrieval(model_config(geom, "s")), TeacherITR(model_config(geom, "t"))
    load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 51, geom["std"])
    load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False), 52, geom["std"])
    types = O.L0_TYPES
else:
    from test_step_gpu import _vqa_models
    from efficientvlm_amd.trainer import VQATrainer
    student, teacher, *_ = _vqa_models(geom, 51, 52)
    types = O.L0_TYPES_VQA
student.l0_module.set_lagrangian_warmup_steps(10)
student.cuda(); teacher.cuda()
if kind == "itr":
    tr = ITRTrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=True)
    batches = [{k: v.cuda() for k, v in synth.make_batch(geom, 4, seed=30 + i, ragged=True).items()
                if k in ("image", "text_ids", "text_atts")} for i in range(3)]
    idx = torch.arange(4).cuda()
    step = lambda c: tr.step(batches[c % 3], idx=idx)
else:
    tr = VQATrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=True)
    batches = [{k: v.cuda() for k, v in synth.make_vqa_batch(geom, 4, seed=21 + 3 * i).items()} for i in range(3)]
    step = lambda c: tr.step(batches[c % 3])
assert tr.reducer.active
seqs, outs, launches = [], [], []
for c in range(8):
    student.l0_module.injected_eps = {t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6)
                                      for t in types}
    runtime.COLLECTIVES = []
    o = step(c)
    seqs.append(runtime.COLLECTIVES)
    if o is not None:
        outs.append(o.tolist()); launches.append(tr.last_launch)
torch.cuda.synchronize()
runtime.COLLECTIVES = None
slab = sum(g.numel() for g in tr.opt.flat_grads)
tr.close()
dist.destroy_process_group()
print("RESULT " + json.dumps({"seqs": seqs, "outs": outs, "launches": launches, "slab": slab, "stages": len(tr._stages)}))
"""


@pytest.mark.parametrize("kind", ["itr", "vqa"])
def test_segmented_and_eager_pruning_steps_issue_the_same_collective_sequence(kind):
    """ITRTrainer / VQATrainer on the N > 1 code path (one-rank RCCL group, EVLM_FORCE_REDUCE): the student step replayed as
    hipGraph segments and the eager student step must issue the SAME collectives - that equality is what lets a rank whose
    capture failed (or which meets a new batch kind) step eagerly beside peers that replay segments, with no agreement
    round - the all-reduces of a step cover the gradient slabs exactly once, and both forms produce the same losses"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(eager):
        env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", EVLM_FORCE_REDUCE="1",
                   EVLM_TEST_KIND=kind)
        env.pop("EVLM_NO_STEP_GRAPH", None)
        if eager:
            env["EVLM_NO_STEP_GRAPH"] = "1"
        r = subprocess.run([sys.executable, "-c", _DP_PRUNE_SEQ_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])

    seg, eag = run(False), run(True)
    assert seg["launches"] == ["eager", "eager"] + ["hipGraph segments"] * 5, seg["launches"]
    assert set(eag["launches"]) == {"eager"}
    assert seg["stages"] >= 3                                   # ViT layer groups leave from hooks inside backward
    n_gather = 2 if kind == "itr" else 0                       # ITC features + image ids (Eff_Retrieval.py passes idx)
    for a, b in zip(seg["seqs"][1:], eag["seqs"][1:]):
        assert a == b, (a, b)
        assert [k for k, _, _ in a[:n_gather]] == ["all_gather"] * n_gather
        assert all(k == "all_reduce" and d == "torch.float32" for k, _, d in a[n_gather:])
        assert sum(n for _, n, _ in a[n_gather:]) == seg["slab"]
    np.testing.assert_allclose(np.array(seg["outs"]), np.array(eag["outs"]), rtol=5e-4, atol=1e-5)


_DP_BRANCH_SCRIPT = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["EVLM_REPO"]); sys.path.insert(0, os.path.join(os.environ["EVLM_REPO"], "tests"))
import torch.distributed as dist
from oracle import synth
from test_step_gpu import build_gd
from efficientvlm_amd import ops, runtime
from efficientvlm_amd.trainer import GDTrainer
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
geom = synth.GEOMS["tiny"]
res = {}
for name, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    student, teacher = build_gd(geom, 11)
    neg = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
    student.injected_neg_idx = teacher.injected_neg_idx = neg
    student.keep_injected_neg = teacher.keep_injected_neg = True
    tr = GDTrainer(student, teacher, lr=0.0, dtype=dtype, use_graph=False)       # (lr 0: both steps see the same parameters)
    batch = {k: v.cuda() for k, v in synth.make_batch(geom, 4, seed=3).items()}
    slabs = []
    for world in (1, 2):
        tr.reducer.world = world          # what a two-rank group would make of the loss scale (the collective stays real)
        runtime.COLLECTIVES = []
        tr.step(batch)
        torch.cuda.synchronize()
        slabs.append(torch.cat([g.clone() for g in tr.opt.flat_grads]))
    res[name] = {"launch": tr.reducer.launch, "prescaled": tr.reducer.prescaled,
                 "ops": sorted({k for k, _, _ in runtime.COLLECTIVES}),
                 "half_err": float((slabs[1] * 2 - slabs[0]).norm() / slabs[0].norm()),
                 "nonzero": float(slabs[0].abs().sum()) > 0}
runtime.COLLECTIVES = None
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
"""


def test_rccl_branch_of_the_gradient_exchange_on_a_one_rank_group():
    """The shipping N > 1 branch of GradReducer, selected on a REAL RCCL group (one rank - all a one-GPU box offers): the
    grouped (coalesced) launch is probed at construction and used, the collective is a plain SUM (no ReduceOp.AVG, no
    division pass), and the mean's 1 / world is applied at the source: with the reducer told the world is 2, one eager
    step leaves HALF the gradient in the slabs (fp32 and bf16 compute; every backward kernel - the fused map-distillation
    terms inside the attention kernels, the deferred grouped weight gradients, the in-place embedding / LayerNorm / bias
    sums - is linear in the upstream scale, and a power-of-two scale is exact: what is left between the two runs is the
    order of the f32 atomics of the in-place sums, 1e-6)"""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EVLM_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", EVLM_FORCE_REDUCE="1")
    r = subprocess.run([sys.executable, "-c", _DP_BRANCH_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for name in ("f32", "bf16"):
        assert res[name]["launch"] == "coalesced" and res[name]["prescaled"], res
        assert res[name]["ops"] == ["all_gather", "all_reduce"], res
        assert res[name]["half_err"] < 2e-6 and res[name]["nonzero"], res


@pytest.mark.parametrize("geom_name,B", [("full", 64), ("tiny", 5)])
def test_first_touch_assignment_of_weight_gradients_equals_zero_fill_plus_accumulate(geom_name, B, monkeypatch):
    """ops.WGRAD_ASSIGN (the Linear weights' slab ranges are not zero-filled; the first grouped dY^T X product of a weight
    writes its tile, anything else zero-fills the range first, finish_assign zeroes what the step never touched) against
    the plain form (one zero-fill of every slab, every product accumulated): the same gradient slabs - with the slabs
    poisoned with NaN beforehand, so nothing stale can survive; B = 5 at the tiny geometry takes the tail path of
    reduction lengths that are not multiples of 64 (a queued assignment demoted by the tail's immediate product) and
    products too short to be queued"""
    from efficientvlm_amd import ops
    from efficientvlm_amd.trainer import GDTrainer
    geom = synth.GEOMS[geom_name]
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, B, seed=19).items()}
    slabs, skipped = [], []
    for off in ("1", ""):
        if off:
            monkeypatch.setenv("EVLM_NO_WGRAD_ASSIGN", off)
        else:
            monkeypatch.delenv("EVLM_NO_WGRAD_ASSIGN", raising=False)
        student, teacher = build_gd(geom, 5)
        neg = torch.tensor([(i + 1) % B for i in range(B)] + [(i + 2) % B for i in range(B)])
        student.injected_neg_idx = teacher.injected_neg_idx = neg        # (the same hard negatives in both runs)
        student.keep_injected_neg = teacher.keep_injected_neg = True
        tr = GDTrainer(student, teacher, dtype=torch.bfloat16, use_graph=False)
        for g in tr.opt.flat_grads:
            g.fill_(float("nan"))
        if off:
            tr.opt.zero_grad()        # (the plain form zero-fills in _forward_backward anyway; explicit for the NaN poison)
        # (measured BEFORE the step: weights whose first contribution turns out not to be an assigning product - most of
        # them at the tiny geometry - are moved to the step's grouped zero fill at the end of their first step)
        pre = 0 if off else sum(v.numel() for v in tr.opt.assign_state(student)["skip"].values())
        tr._forward_backward(batch)
        torch.cuda.synchronize()
        slabs.append({n: p.grad.detach().clone() for n, p in student.named_parameters()})
        skipped.append((pre, sum(g.numel() for g in tr.opt.flat_grads)))
        if not off:
            st = tr._assign
            assert not st.get("demote")        # settled: every range is either assigned to by its first product or in the fill list
            assert sum(v.numel() for v in st["skip"].values()) + sum(v.numel() for v in st["fill"]) == skipped[-1][1]
        del tr, student, teacher
    assert skipped[0][0] == 0 and skipped[1][0] > 0.6 * skipped[1][1]               # most of the slabs is never filled
    # The tensors under test: the weights whose ranges were not zero-filled.  (Two bf16 runs of the step differ by up to ~1 %
    # per tensor on their own - f32 atomics on the dX path, the MLM decoder's split-K product, flip bf16 roundings
    # downstream; a lost or doubled contribution would show as >= 5 % of one weight's gradient.)
    checked = 0
    for n, a in slabs[0].items():
        b = slabs[1][n]
        assert bool(torch.isfinite(b).all()), n
        if n.endswith(".weight") and a.dim() == 2 and a.numel() >= 4096 and "embed" not in n and float(a.norm()) > 0:
            assert float((a - b).norm()) <= 3e-2 * float(a.norm()), (n, float((a - b).norm() / a.norm()))
            checked += 1
    assert checked >= 60
    fa = torch.cat([v.reshape(-1) for v in slabs[0].values()])
    fb = torch.cat([v.reshape(-1) for v in slabs[1].values()])
    assert float((fa - fb).norm() / fa.norm()) < 1e-2


def test_deferred_grouped_weight_gradients_match_immediate_ones():
    """GDTrainer (bf16, B = 64, full geometry): gradients with the dW products queued and flushed as grouped launches
    against the same step with every dW launched in place (split-K kernels)"""
    from efficientvlm_amd import ops
    from efficientvlm_amd.trainer import GDTrainer
    geom = synth.GEOMS["full"]
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, 64, seed=11).items()}
    slabs = []
    for defer in (False, True):
        student, teacher = build_gd(geom, 5)
        tr = GDTrainer(student, teacher, dtype=torch.bfloat16, use_graph=False)
        tr.defer_wgrad = defer
        ops.dropout_seed(0)           # hard-negative draws (device Philox stream): the same in both runs
        tr._forward_backward(batch)
        torch.cuda.synchronize()
        slabs.append([g.clone() for g in tr.opt.flat_grads])
        del tr, student, teacher
    for a, b in zip(*slabs):
        assert float((a - b).norm() / a.norm()) < 3e-3


def test_itr_training_steps_match_oracle_forward_backward_plus_optimiser_restatement():
    """two Eff_Retrieval training steps (fp32, tiny geometry, same gate noise and hard negatives on both sides): the
    drop-in's ITRTrainer against oracle forward/backward + the restated optimisers (main AdamW incl. the gate parameters,
    +reg_lr gates, -reg_lr multipliers, constrain_parameters).  Losses of both steps; parameter UPDATES of the large
    tensors (elements whose gradient is numerically zero take Adam steps of arbitrary sign and are not compared)."""
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    from efficientvlm_amd.trainer import ITRTrainer
    from oracle import optim_oracle as OO
    geom = synth.GEOMS["tiny"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sch = schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True)
    t_sch = schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False)
    student, teacher = EffXVLMforRetrieval(model_config(geom, "s")), TeacherITR(model_config(geom, "t"))
    s_sd = load_det_weights(student, s_sch, 77, geom["std"])
    t_sd = load_det_weights(teacher, t_sch, 78, geom["std"])
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
            s_sd["l0_module." + n] = p.detach().clone()
    student.l0_module.set_lagrangian_warmup_steps(10)
    student.to(DEV); teacher.to(DEV)
    B = 4
    batch = synth.make_batch(geom, B, seed=9, ragged=True)
    idx = torch.tensor([0, 1, 1, 3])
    lr, wd, lr_mult, reg = 1e-3, 0.01, 2.0, 0.05
    tr = ITRTrainer(student, teacher, lr=lr, weight_decay=wd, lr_mult=lr_mult, reg_learning_rate=reg, dtype=torch.float32)
    # oracle side: leaf parameters, Adam state
    ref = {k: v.clone().float().requires_grad_(True) for k, v in s_sd.items() if torch.is_floating_point(v)}
    p0 = {k: v.detach().clone() for k, v in ref.items()}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
    state_l0 = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items() if k.startswith("l0_module.")}
    groups = OO.param_groups([(n, p) for n, p in student.named_parameters()], student.init_params, lr, wd, lr_mult)
    consts = O.l0_constants(geom["hidden"], geom["ffn"], geom["heads"], s_cfg["vit_layers"], s_cfg["fusion_layer"],
                            s_cfg["text_layers"] - s_cfg["fusion_layer"])
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    for step in range(2):
        eps = {t: torch.rand(ref["l0_module." + O.L0_PARAM[t]].shape, generator=gen).clamp(1e-6, 1 - 1e-6) for t in O.L0_TYPES}
        s_neg = torch.tensor([(i + 1 + step) % B for i in range(2 * B)])
        t_neg = torch.tensor([(i + 2) % B for i in range(2 * B)])
        student.l0_module.injected_eps = {t: eps[t].clone() for t in O.L0_TYPES}
        student.injected_neg_idx, teacher.injected_neg_idx = s_neg.clone(), t_neg.clone()
        got = tr.step(dev_batch, idx=idx.to(DEV)).cpu()
        # oracle step
        for v in ref.values():
            v.grad = None
        logas = {k[len("l0_module."):]: v for k, v in ref.items() if k.endswith("_loga")}
        zs = O.l0_forward(logas, True, eps)
        S = O.retrieval_forward(ref, s_cfg, batch, idx, s_neg, zs)
        with torch.no_grad():
            T = O.retrieval_forward(t_sd, t_cfg, batch, idx, t_neg)
        kd = O.kd_terms(S, T, with_cross_attn=True)
        lagr, _, _ = O.l0_lagrangian(logas, ref["l0_module.lambda_1"], ref["l0_module.lambda_2"], consts, step,
                                     target_sparsity=0.25, lagrangian_warmup=10)
        total, mix = O.itr_loss_mix(S["loss"], kd, lagr)
        total.backward()
        want = torch.stack([total.detach(), S["loss"]["loss_itc"].detach(), S["loss"]["loss_itm"].detach(),
                            mix["loss_kd"].detach(), lagr.detach().reshape(())])
        assert torch.allclose(got, want, rtol=2e-4, atol=1e-6), (step, got, want)
        with torch.no_grad():
            grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in ref.items()}
            for g in groups:
                for n in g["names"]:
                    OO.hf_adamw_step(ref[n], grads[n], state[n][0], state[n][1], step + 1, g["lr"], (0.9, 0.98), 1e-8,
                                     g["weight_decay"])
            for n in state_l0:
                rl = -reg if "lambda" in n else reg
                OO.hf_adamw_step(ref[n], grads[n], state_l0[n][0], state_l0[n][1], step + 1, rl, (0.9, 0.98), 1e-8, 0.0)
            for n in ref:
                if n.endswith("_loga"):
                    ref[n].clamp_(min=math.log(1e-2), max=math.log(1e2))
    mine = {n: p.detach().cpu() for n, p in student.named_parameters()}
    checked = 0
    for n in ("vision_encoder.encoder.layers.0.mlp.fc1.weight", "text_encoder.encoder.layer.4.crossattention.self.query.weight",
              "itm_head.0.weight", "l0_module.lambda_1", "l0_module.lambda_2", "l0_module.vision_int_loga"):
        if n not in mine:
            n = n.replace("text_encoder.", "text_encoder.bert.")
        d_ref, d_mine = ref[n].detach() - p0[n], mine[n] - p0[n]
        assert float(d_ref.abs().max()) > 0, n
        assert float((d_mine - d_ref).norm()) <= 0.02 * float(d_ref.norm()), (n, float((d_mine - d_ref).norm()), float(d_ref.norm()))
        checked += 1
    assert checked == 6


@pytest.mark.parametrize("with_gates", [False, True])
def test_retrieval_evaluation_rerank_matches_oracle(with_gates):
    """Eff_Retrieval.py:215-319 at tensor level (fp32, tiny geometry, 7 images x 11 texts, k_test = 4): score matrices of
    the batched, K/V-sharing implementation against the oracle's one-query-at-a-time restatement, for one rank and for
    the two shards of a 2-rank run (summed as the driver's all-reduce would)"""
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.retrieval_eval import evaluation_scores
    from oracle import retrieval_eval_oracle as RO
    geom = synth.GEOMS["tiny"]
    s_cfg = O.model_cfg(geom, "s")
    model = EffXVLMforRetrieval(model_config(geom, "s"))
    sd = load_det_weights(model, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 21, geom["std"])
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n, p in model.l0_module.named_parameters():
            if "lambda" not in n:
                p.copy_(torch.randn(p.shape, generator=gen) * 3.0)
            sd["l0_module." + n] = p.detach().clone()
    model.to(DEV)
    bi = synth.make_batch(geom, 7, seed=31, ragged=True)
    bt = synth.make_batch(geom, 11, seed=32, ragged=True)
    images, text_ids, text_atts = bi["image"], bt["text_ids"], bt["text_atts"]
    zs_o = None
    if with_gates:
        logas = {k[len("l0_module."):]: v for k, v in sd.items() if k.endswith("_loga")}
        zs_o = O.l0_forward(logas, False)
    want = [RO.evaluation_scores(sd, s_cfg, images, text_ids, text_atts, 4, zs=zs_o, rank=r, world=w)
            for r, w in ((0, 1), (0, 2), (1, 2))]
    kw = dict(k_test=4, zs="auto" if with_gates else None, query_bs=3, reduce=False)
    dimg, dids, datt = images.to(DEV), text_ids.to(DEV), text_atts.to(DEV)
    got = [evaluation_scores(model, dimg, dids, datt, rank=r, world=w, **kw) for r, w in ((0, 1), (0, 2), (1, 2))]
    for (wi, wt), (gi, gt) in zip(want, got):
        assert torch.equal(gi.cpu() == -100.0, wi == -100.0) and torch.equal(gt.cpu() == -100.0, wt == -100.0)
        assert torch.allclose(gi.cpu(), wi, rtol=2e-4, atol=2e-5) and torch.allclose(gt.cpu(), wt, rtol=2e-4, atol=2e-5)
    # the driver's SUM all-reduce of the two shards
    s_i, s_t = got[1][0] + got[2][0], got[1][1] + got[2][1]
    full_i, full_t = want[0]
    done = full_i != -100.0
    assert torch.allclose(s_i.cpu()[done], full_i[done] - 100.0, rtol=2e-4, atol=1e-4)
    assert torch.all(s_i.cpu()[~done] == -200.0) and torch.all(s_t.cpu()[full_t == -100.0] == -200.0)


def test_retrieval_evaluation_rerank_matches_the_reference_vectors():
    """retrieval_eval.evaluation_scores (batched queries, shared image K/V) against the score matrices of the REFERENCE's
    own Eff_Retrieval.evaluation (tests/golden/rerank_tiny.npz), fp32: same rescored pairs, scores within 2e-4"""
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.retrieval_eval import evaluation_scores
    fx = load_fixture("rerank_tiny.npz")
    geom = synth.GEOMS["tiny"]
    seed = int(fx["meta.seed"])
    cfg = O.model_cfg(geom, "s")
    model = EffXVLMforRetrieval(model_config(geom, "s"))
    load_det_weights(model, schema.xvlm_schema(cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 5000 + seed, geom["std"])
    with torch.no_grad():
        for n, p in model.l0_module.named_parameters():
            p.copy_(torch.from_numpy(fx["in.l0." + n]))
    model.to(DEV)
    images, ids, atts = (torch.from_numpy(fx["in." + k]).to(DEV) for k in ("image", "text_ids", "text_atts"))
    for rank, world in ((0, 1), (0, 2), (1, 2)):
        gi, gt = evaluation_scores(model, images, ids, atts, k_test=int(fx["meta.k_test"]), zs="auto", query_bs=3, reduce=False,
                                   rank=rank, world=world)
        for got, name in ((gi, "i2t"), (gt, "t2i")):
            want = torch.from_numpy(fx[f"out.r{rank}w{world}.{name}"])
            assert torch.equal(got.cpu() == -100.0, want == -100.0), (rank, world, name)
            assert torch.allclose(got.cpu(), want, rtol=2e-4, atol=2e-5), (rank, world, name)


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipelined_teacher_reproduces_the_unpipelined_training_trajectory(use_graph, monkeypatch):
    """GDTrainer(pipeline_teacher=True) runs the frozen teacher one batch ahead of the student.  With the hard negatives
    made a deterministic function of the features (shifted identity instead of the multinomial draw) the loss
    trajectory over four DISTINCT batches must equal the unpipelined trainer's, one call later - eagerly and through
    the two captured hipGraphs."""
    from efficientvlm_amd.efficient_models.xvlm import XVLMBase
    from efficientvlm_amd.trainer import GDTrainer

    def fixed_negatives(self, image_feat, text_feat, idx):
        bs = image_feat.size(0)
        ar = torch.arange(bs, device=image_feat.device)
        return (ar + 1) % bs, (ar + 2) % bs
    monkeypatch.setattr(XVLMBase, "_sample_negatives", fixed_negatives)
    geom = synth.GEOMS["tiny"]
    batches = [{k: v.to(DEV) for k, v in synth.make_batch(geom, 4, seed=60 + i).items()} for i in range(4)]
    outs = {}
    for pipe in (False, True):
        student, teacher = build_gd(geom, 9)
        tr = GDTrainer(student, teacher, lr=1e-3, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.float32,
                       use_graph=use_graph, pipeline_teacher=pipe)
        seq = []
        if pipe:
            assert tr.step(batches[0]) is None              # priming call
            for b in batches[1:] + [batches[0]]:
                seq.append(tr.step(b).clone())              # losses of the batch passed one call earlier
        else:
            for b in batches:
                seq.append(tr.step(b).clone())
        torch.cuda.synchronize()
        outs[pipe] = torch.stack(seq).cpu()
        del tr, student, teacher
    assert torch.allclose(outs[True], outs[False], rtol=5e-4, atol=1e-5), (outs[True], outs[False])
    assert float((outs[False][0] - outs[False][3]).abs().max()) > 1e-3      # the batches (and the training) do differ


def test_pipeline_state_of_batch_kinds_is_bounded(monkeypatch):
    """GDTrainer keeps static buffers and captured graphs per batch SHAPE; the state is bounded: the first MAX_BATCH_KINDS
    shapes get hipGraphs, later ones run the same kernels eagerly out of static buffers of which at most MAX_EAGER_KINDS
    stay alive (least recently used evicted - never the kind whose batch is still waiting for its student step).  Four
    batch sizes cycled through a trainer that may capture one kind and keep one eager kind must train exactly like one
    that captures them all."""
    from efficientvlm_amd.efficient_models.xvlm import XVLMBase
    from efficientvlm_amd.trainer import GDTrainer

    def fixed_negatives(self, image_feat, text_feat, idx):
        bs = image_feat.size(0)
        ar = torch.arange(bs, device=image_feat.device)
        return (ar + 1) % bs, (ar + 2) % bs
    monkeypatch.setattr(XVLMBase, "_sample_negatives", fixed_negatives)
    geom = synth.GEOMS["tiny"]
    batches = [{k: v.to(DEV) for k, v in synth.make_batch(geom, 3 + (i % 4), seed=70 + i).items()} for i in range(10)]
    outs = {}
    for cap, ecap in ((6, 4), (1, 1)):
        student, teacher = build_gd(geom, 9)
        tr = GDTrainer(student, teacher, lr=1e-3, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.float32,
                       use_graph=True, pipeline_teacher=True)
        tr.MAX_BATCH_KINDS, tr.MAX_EAGER_KINDS = cap, ecap
        seq = []
        for b in batches:
            o = tr.step(b)
            if o is not None:
                seq.append(o.clone())
            assert len(tr._pipes) <= cap + ecap + 1                 # (+1: the kind whose batch is waiting is never evicted)
            assert sum(1 for q in tr._pipes.values() if not q.get("eager")) <= cap
        torch.cuda.synchronize()
        outs[cap] = torch.stack(seq).cpu()
        del tr, student, teacher
    assert torch.allclose(outs[1], outs[6], rtol=5e-4, atol=1e-5), (outs[1], outs[6])


@pytest.mark.parametrize("use_graph", [False, True])
def test_region_steps_interleave_with_general_steps_in_reference_order(use_graph, monkeypatch):
    """The GD recipe draws a REGION step (idx_to_group_img / image_atts / bbox + giou losses) before a general step with
    probability 0.5 (GeneralDistill.py:157-262).  A mixed sequence G R G G R R G must give the same loss trajectory through
    the unpipelined trainer (one hipGraph per batch kind) and the pipelined one (which trains on the waiting general batch
    first, then the region batch, then re-primes) - i.e. the optimiser updates are applied in arrival order."""
    from efficientvlm_amd.efficient_models.xvlm import XVLMBase
    from efficientvlm_amd.trainer import GDTrainer

    def fixed_negatives(self, image_feat, text_feat, idx):
        bs = image_feat.size(0)
        ar = torch.arange(bs, device=image_feat.device)
        return (ar + 1) % bs, (ar + 2) % bs
    monkeypatch.setattr(XVLMBase, "_sample_negatives", fixed_negatives)
    geom = synth.GEOMS["tiny"]
    dev = lambda b: {k: v.to(DEV) for k, v in b.items()}
    G = [dev(synth.make_batch(geom, 4, seed=80 + i)) for i in range(4)]
    R = [dev(synth.make_region_batch(geom, 3, 6, seed=90 + i)) for i in range(3)]
    seq = [G[0], R[0], G[1], G[2], R[1], R[2], G[3]]
    outs = {}
    for pipe in (False, True):
        student, teacher = build_gd(geom, 9)
        tr = GDTrainer(student, teacher, lr=1e-3, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=torch.float32,
                       use_graph=use_graph, pipeline_teacher=pipe)
        got = []
        for b in seq:
            o = tr.step(b)
            if o is not None:
                got.append(o.clone())                       # (a graph replay returns its static output tensor)
        if pipe:
            assert len(got) == len(seq) - 1
            got.append(tr.step(G[0]).clone())               # flushes the losses of the last batch of `seq`
        torch.cuda.synchronize()
        outs[pipe] = torch.stack(got).cpu()
        w = torch.cat([p.detach().reshape(-1).float().cpu() for p in list(student.parameters())[:40]])
        outs[pipe, "w"] = w
        del tr, student, teacher
    assert torch.allclose(outs[True], outs[False], rtol=5e-4, atol=1e-5), (outs[True], outs[False])
    assert torch.isfinite(outs[False]).all()


def test_region_training_steps_match_oracle_plus_optimiser_restatement():
    """two REGION steps through GDTrainer (fp32, tiny geometry, same hard negatives on both sides) against oracle
    forward/backward (bbox branch included) + the restated clip + HF-AdamW: losses of both steps and the parameter
    UPDATES of large tensors, the bbox head (which only region steps train) among them"""
    from efficientvlm_amd.models.model_pretrain import XVLM
    from efficientvlm_amd.trainer import GDTrainer
    from oracle import optim_oracle as OO
    geom = synth.GEOMS["tiny"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    student, teacher = XVLM(model_config(geom, "s")), XVLM(model_config(geom, "t"))
    s_sd = load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"]), 31, geom["std"])
    t_sd = load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"]), 32, geom["std"])
    student.to(DEV); teacher.to(DEV)
    lr, wd, lr_mult = 1e-3, 0.01, 2.0
    tr = GDTrainer(student, teacher, lr=lr, weight_decay=wd, lr_mult=lr_mult, max_grad_norm=1.0, dtype=torch.float32,
                   use_graph=False)
    W, DW, DB = ("text_encoder.bert.embeddings.word_embeddings.weight", "text_encoder.cls.predictions.decoder.weight",
                 "text_encoder.cls.predictions.decoder.bias")
    ref = {k: v.clone().float().requires_grad_(True) for k, v in s_sd.items() if torch.is_floating_point(v) and k not in (DW, DB)}
    p0 = {k: v.detach().clone() for k, v in ref.items()}
    tie = lambda sd: {**sd, DW: sd[W], DB: sd["text_encoder.cls.predictions.bias"]}
    t_sd = tie(t_sd)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
    groups = OO.param_groups([(n, p) for n, p in student.named_parameters()], student.init_params, lr, wd, lr_mult)
    R = 6
    for step in range(2):
        batch = synth.make_region_batch(geom, 3, R, seed=70 + step)
        s_neg = torch.tensor([(i + 1 + step) % R for i in range(2 * R)])
        t_neg = torch.tensor([(i + 2) % R for i in range(2 * R)])
        student.injected_neg_idx, teacher.injected_neg_idx = s_neg.clone(), t_neg.clone()
        got = tr.step({k: v.to(DEV) for k, v in batch.items()}).cpu()
        for v in ref.values():
            v.grad = None
        total, S, T, kd, mix = O.gd_step(tie({**s_sd, **ref}), t_sd, s_cfg, t_cfg, batch, s_neg, t_neg)
        assert "loss_bbox" in S["loss"] and float(S["loss"]["loss_giou"]) > 0
        total.backward()
        want = torch.stack([total.detach(), S["loss"]["loss_itc"].detach(), S["loss"]["loss_itm"].detach(),
                            S["loss"]["loss_mlm"].detach(), mix["loss_kd"].detach()])
        assert torch.allclose(got, want, rtol=2e-4, atol=1e-6), (step, got, want)
        with torch.no_grad():
            names = [n for g in groups for n in g["names"]]
            grads = {n: (ref[n].grad if ref[n].grad is not None else torch.zeros_like(ref[n])) for n in names}
            OO.clip_grad_norm_(list(grads.values()), 1.0)
            for g in groups:
                for n in g["names"]:
                    OO.hf_adamw_step(ref[n], grads[n], state[n][0], state[n][1], step + 1, g["lr"], (0.9, 0.98), 1e-8,
                                     g["weight_decay"])
    mine = {n: p.detach().cpu() for n, p in student.named_parameters()}
    for n in ("vision_encoder.encoder.layers.5.mlp.fc1.weight", "vision_encoder.encoder.layers.0.self_attn.q_proj.weight",
              "text_encoder.bert.encoder.layer.4.crossattention.self.key.weight", "bbox_head.0.weight", "bbox_head.3.weight",
              "itm_head.0.weight", W):
        d_ref, d_mine = ref[n].detach() - p0[n], mine[n] - p0[n]
        assert float(d_ref.abs().max()) > 0, n
        assert float((d_mine - d_ref).norm()) <= 0.02 * float(d_ref.norm()), (n, float((d_mine - d_ref).norm()), float(d_ref.norm()))


def _vqa_models(geom, seed_s, seed_t, fx=None):
    from efficientvlm_amd.efficient_models.model_generation import EffXVLMForVQA
    from efficientvlm_amd.models.model_generation import XVLMForVQA
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    cfg = lambda role, c: dict(model_config(geom, role), pad_token_id=0, num_dec_layers=c["text_layers"] - c["fusion_layer"])
    student, teacher = EffXVLMForVQA(cfg("s", s_cfg)), XVLMForVQA(cfg("t", t_cfg))
    s_sd = load_det_weights(student, schema.vqa_schema(s_cfg, geom["max_pos"], l0=True), seed_s, geom["std"], fx, "student")
    t_sd = load_det_weights(teacher, schema.vqa_schema(t_cfg, geom["max_pos"]), seed_t, geom["std"], fx, "teacher")
    return student, teacher, s_sd, t_sd, s_cfg, t_cfg


def test_vqa_step_fp32_matches_reference_vectors():
    """Eff_VQA.py:95-176 step against tests/golden/vqa_tiny.npz (captured from EffXVLMForVQA / XVLMForVQA / VQAL0Module):
    every hidden state and attention map of the image encoder, question encoder and CAUSAL answer decoder, the logits,
    the weighted answer loss, each KD term, the Lagrangian, the loss mix, and the student gradients"""
    from types import SimpleNamespace as NS
    from efficientvlm_amd import distill
    from efficientvlm_amd.runtime import compute
    fx = load_fixture("vqa_tiny.npz")
    geom = synth.GEOMS[str(fx["meta.geom"])]
    seed = int(fx["meta.seed"])
    student, teacher, *_ = _vqa_models(geom, 5000 + seed, 6000 + seed, fx)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.from_numpy(fx["in.l0." + n]))
    student.l0_module.set_lagrangian_warmup_steps(10)
    student.to(DEV).train(); teacher.to(DEV).eval()
    assert list(student.l0_module.types) == list(fx["meta.l0_types"])
    assert int(student.l0_module.prunable_model_size) == int(fx["meta.prunable_model_size"])
    student.l0_module.injected_eps = {t: torch.from_numpy(fx[f"in.eps.{t}"]) for t in student.l0_module.types}
    b = {k[3:]: torch.from_numpy(v).to(DEV) for k, v in fx.items() if k.startswith("in.") and k.count(".") == 1}
    q, a = NS(input_ids=b["question_ids"], attention_mask=b["question_atts"]), NS(input_ids=b["answer_ids"], attention_mask=b["answer_atts"])
    kw = dict(train=True, k=b["k"].tolist(), weights=b["weights"], output_attentions=True, output_hidden_states=True)
    with compute(torch.float32):
        S = student(b["image"], q, a, **kw)
        with torch.no_grad():
            T = teacher(b["image"], q, a, **kw)
        kd = distill.vqa_kd_terms(S, T)
        lagr, exp_s, tgt = student.l0_module.lagrangian_regularization(3)
        total, mix = distill.vqa_loss_mix(S["loss"], kd, lagr)
        total.backward()
    for tag, out in (("student", S), ("teacher", T)):
        for dn in ("hidden_dict", "attention_dict", "cross_attention_dict"):
            for k, tup in out[dn].items():
                for i, t in enumerate(tup):
                    close(t.float(), fx[f"{tag}.{k}.{i}"], 1e-4, 1e-6, f"{tag}.{k}.{i}")
        close(out["logits_dict"]["logits"].float(), fx[f"{tag}.logits"], 1e-4, 1e-5, f"{tag}.logits")
        close(out["loss"], fx[f"{tag}.loss"], 1e-4, 0, f"{tag}.loss")
    for k, v in kd.items():
        close(v, fx[f"kd.{k}"], 1e-4, 1e-7, f"kd.{k}")
    for k, v in mix.items():
        close(v, fx[f"mix.{k}"], 1e-4, 0, f"mix.{k}")
    close(lagr, fx["mix.lagrangian"], 1e-4, 1e-7, "lagrangian")
    close(total, fx["mix.total"], 1e-4, 0, "total")
    n = 0
    for nme, p in student.named_parameters():
        key = f"student.grad_chk.{nme}"
        if key not in fx:
            continue
        ref_l2 = float(fx[key][1])
        got = float(p.grad.double().pow(2).sum().sqrt())
        assert abs(got - ref_l2) <= 1e-3 * ref_l2 + 2e-6, f"grad L2 {nme}: {got} vs {ref_l2}"
        if f"student.grad.{nme}" in fx:
            close(p.grad, fx[f"student.grad.{nme}"], 0, 1e-3 * ref_l2 + 2e-6, f"grad {nme}")
        n += 1
    assert n > 100


def test_vqa_training_steps_bf16_track_the_fp32_oracle():
    """VQATrainer (bf16 compute, three optimisers) for two steps on a fresh batch: step-0 losses within bf16 tolerance of
    the fp32 oracle, finite losses, decreasing answer loss is NOT asserted (two steps), the gate parameters move and stay
    inside constrain_parameters' interval"""
    from efficientvlm_amd.trainer import VQATrainer
    geom = synth.GEOMS["tiny"]
    student, teacher, s_sd, t_sd, s_cfg, t_cfg = _vqa_models(geom, 41, 42)
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
            s_sd["l0_module." + n] = p.detach().clone()
    student.l0_module.set_lagrangian_warmup_steps(10)
    student.to(DEV); teacher.to(DEV)
    batch = synth.make_vqa_batch(geom, 4, seed=21)
    tr = VQATrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.bfloat16)
    eps = {t: torch.rand(s_sd["l0_module." + O.L0_PARAM[t]].shape, generator=gen).clamp(1e-6, 1 - 1e-6) for t in O.L0_TYPES_VQA}
    student.l0_module.injected_eps = {t: e.clone() for t, e in eps.items()}
    loga0 = student.l0_module.decoder_int_loga.detach().clone()
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    got0 = tr.step(dev_batch).cpu()
    got1 = tr.step(dev_batch).cpu()
    tie = lambda sd: {**sd, "text_decoder.cls.predictions.decoder.weight": sd["text_decoder.bert.embeddings.word_embeddings.weight"],
                      "text_decoder.cls.predictions.decoder.bias": sd["text_decoder.cls.predictions.bias"]}
    with torch.no_grad():
        logas = {k[len("l0_module."):]: v for k, v in s_sd.items() if k.endswith("_loga")}
        S = O.vqa_forward(tie(s_sd), s_cfg, batch, O.l0_forward(logas, True, eps))
        T = O.vqa_forward(tie(t_sd), t_cfg, batch)
        kd = O.vqa_kd_terms(S, T)
        nd = s_cfg["text_layers"] - s_cfg["fusion_layer"]
        consts = O.l0_constants(geom["hidden"], geom["ffn"], geom["heads"], s_cfg["vit_layers"], s_cfg["fusion_layer"], nd, nd)
        lagr, _, _ = O.l0_lagrangian(logas, s_sd["l0_module.lambda_1"], s_sd["l0_module.lambda_2"], consts, 0,
                                     target_sparsity=0.25, lagrangian_warmup=10)
        total, mix = O.vqa_loss_mix(S["loss"], kd, lagr)
    want = torch.stack([total, S["loss"], mix["loss_kd"], lagr.reshape(())])
    assert torch.allclose(got0, want, rtol=4e-2, atol=2e-3), (got0, want)
    assert torch.isfinite(got1).all()
    la = student.l0_module.decoder_int_loga.detach()
    assert float((la - loga0.to(DEV)).abs().max()) > 0 and float(la.min()) >= math.log(1e-2) - 1e-6 and float(la.max()) <= math.log(1e2) + 1e-6


def test_vqa_captured_student_step_reproduces_the_eager_trajectory():
    """VQATrainer(pipeline_teacher=True, capture_step=True): the student step replays as a hipGraph per teacher-prefetch
    parity (first step per parity eager, second captured, then replays) - seven steps on three rotating batches must give
    the loss trajectory and the final gate parameters of the eager trainer: the Lagrangian ramp comes from the device-side
    step counter, the gate noise from the per-step draws staged into the static buffers, the three optimisers' bias
    corrections from their device-side schedules"""
    from efficientvlm_amd.trainer import VQATrainer
    geom = synth.GEOMS["tiny"]
    # (one batch kind: the number of answer rows is part of a batch's shape signature, and a new kind starts eagerly)
    b0 = {k: v.to(DEV) for k, v in synth.make_vqa_batch(geom, 4, seed=60).items()}
    batches = [b0, {**b0, "image": b0["image"] * 0.5}, {**b0, "image": b0["image"] + 0.25}]
    outs, logas = {}, {}
    for cap in (False, True):
        student, teacher, s_sd, t_sd, s_cfg, t_cfg = _vqa_models(geom, 41, 42)
        gen = torch.Generator().manual_seed(6)
        with torch.no_grad():
            for n, p in student.l0_module.named_parameters():
                p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
        student.l0_module.set_lagrangian_warmup_steps(5)
        student.to(DEV); teacher.to(DEV)
        eps = [{t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6)
                for t in O.L0_TYPES_VQA} for _ in range(7)]
        tr = VQATrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=True,
                        capture_step=cap)
        seq = []
        for c in range(8):
            if c >= 1:
                student.l0_module.injected_eps = {t: e.clone() for t, e in eps[c - 1].items()}
            o = tr.step(batches[c % 3])
            if o is not None:
                seq.append(o.clone())
        torch.cuda.synchronize()
        assert len(seq) == 7
        if cap:
            assert tr.last_launch == "hipGraph replay" and len(tr._sgraphs) >= 2
        outs[cap] = torch.stack(seq).cpu()
        logas[cap] = student.l0_module.decoder_int_loga.detach().cpu().clone()
    assert torch.allclose(outs[True], outs[False], rtol=5e-4, atol=1e-5), (outs[True], outs[False])
    assert torch.allclose(logas[True], logas[False], rtol=1e-4, atol=1e-6)


def test_vqa_captured_step_follows_the_eager_trainer_through_a_stop_prune_switch():
    """Eff_VQA.py:375-378 flips stop_prune at --stop_prune_epoch: from then on the student evaluates the deterministic gates
    (l0_module.forward(training=False)) and draws no gate noise.  The captured trainer keeps a gate-noise plan per step KIND
    (stop_prune is part of it): the first stop_prune step must not raise, the stop_prune replays must not consume host RNG
    draws the eager trainer never makes, and an injected draw is consumed by exactly one step - ten steps with the switch
    after the fifth give the eager trainer's losses, and both trainers leave the host generator in the same state"""
    from efficientvlm_amd.trainer import VQATrainer
    geom = synth.GEOMS["tiny"]
    b0 = {k: v.to(DEV) for k, v in synth.make_vqa_batch(geom, 4, seed=60).items()}
    batches = [b0, {**b0, "image": b0["image"] * 0.5}, {**b0, "image": b0["image"] + 0.25}]
    outs, rng, launches = {}, {}, {}
    for cap in (False, True):
        student, teacher, *_ = _vqa_models(geom, 41, 42)
        gen = torch.Generator().manual_seed(6)
        with torch.no_grad():
            for n, p in student.l0_module.named_parameters():
                p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
        student.l0_module.set_lagrangian_warmup_steps(5)
        student.to(DEV); teacher.to(DEV)
        tr = VQATrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=True,
                        capture_step=cap)
        torch.manual_seed(123)                    # the gate noise comes from the HOST generator in both trainers
        seq, ln = [], []
        for c in range(11):
            if c == 2:                            # one injected draw: consumed by this step alone, in both trainers
                student.l0_module.injected_eps = {t: torch.full(getattr(student.l0_module, O.L0_PARAM[t]).shape, 0.37)
                                                  for t in O.L0_TYPES_VQA}
            o = tr.step(batches[c % 3], stop_prune=(c >= 6))
            if o is not None:
                seq.append(o.clone()); ln.append(tr.last_launch)
        torch.cuda.synchronize()
        outs[cap], launches[cap] = torch.stack(seq).cpu(), ln
        rng[cap] = torch.rand(4)                  # (equal only if both trainers drew the same amount of noise)
        tr.close()
    assert launches[True][4:5] == ["hipGraph replay"] and launches[True][-1] == "hipGraph replay", launches[True]
    assert launches[True].count("eager") == 4                # two pairs per kind, one eager first step each
    assert torch.allclose(outs[True], outs[False], rtol=5e-4, atol=1e-5), (outs[True], outs[False])
    assert torch.equal(rng[True], rng[False])


def test_closing_and_dropping_trainers_while_another_one_replays():
    """trainer.close() destroys a trainer's hipGraphs, static buffers and pinned blocks with the device idle and cuts the
    model hooks that tie it into a reference cycle; a trainer dropped WITHOUT close() does the same from its finaliser.
    Five trainers (joint hipGraph each) are built, stepped and disposed of - three closed, two just dropped - between the
    replays of a sixth, whose loss trajectory must equal an undisturbed run of the same trainer."""
    import gc
    from efficientvlm_amd.trainer import GDTrainer
    geom = synth.GEOMS["tiny"]
    batches = [{k: v.to(DEV) for k, v in synth.make_batch(geom, 4, seed=3 + i).items()} for i in range(3)]

    def fresh(seed):
        student, teacher = build_gd(geom, seed)
        neg = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
        student.injected_neg_idx = teacher.injected_neg_idx = neg
        student.keep_injected_neg = teacher.keep_injected_neg = True
        return GDTrainer(student, teacher, dtype=torch.bfloat16, use_graph=True, pipeline_teacher=True)

    def trajectory(disturb):
        main = fresh(17)
        seq = []
        for c in range(13):
            o = main.step(batches[c % 3])
            if o is not None:
                seq.append(o.clone())
            if disturb and c in (2, 4, 6, 8, 10):
                other = fresh(30 + c)
                for i in range(4):
                    other.step(batches[i % 3])                 # captured and replayed
                if c in (2, 6, 10):
                    other.close()
                    assert other._closed and not other._joint and other.student.on_vision_grad is None
                    other.close()                              # idempotent
                del other                                      # (c = 4, 8: dropped un-closed; the finaliser closes it)
                if c == 8:
                    gc.collect()
        torch.cuda.synchronize()
        main.close()
        return torch.stack(seq).cpu()

    a, b = trajectory(False), trajectory(True)
    assert torch.isfinite(b).all()
    assert torch.allclose(a, b, rtol=2e-2, atol=1e-3), (a, b)      # (bf16 atomics: last bits differ from run to run)


@pytest.mark.parametrize("kind", ["itr384", "vqa480"])
def test_full_width_bf16_captured_pruning_steps_follow_the_eager_trajectory(kind):
    """The captured student step at FULL width in bf16 on the long image sequences (577 / 901 tokens: streaming attention
    kernels, ragged-tail weight gradients, the padded vocabulary head of the VQA decoder - none of which the tiny fp32
    capture tests reach): ten training steps through hipGraph replays against the same steps launched eagerly.  bf16
    atomics make the two runs differ in the last bits, so the trajectories are compared at 2 %; a replayed graph that reads
    or zero-fills the wrong memory (the round-4 graph memset defect showed up as a 2.4 % gap at step 5 and 9 % at step 9
    of exactly this comparison) does not pass, nor does a non-finite gradient."""
    from efficientvlm_amd.trainer import ITRTrainer, VQATrainer
    res = 384 if kind == "itr384" else 480
    geom = dict(synth.GEOMS["full"]); geom["image_res"] = res
    seqs, norms = {}, {}
    for cap in (False, True):
        torch.manual_seed(0)
        if kind == "itr384":
            from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
            from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
            student = EffXVLMforRetrieval(model_config(geom, "s", image_res=res)).to(DEV)
            teacher = TeacherITR(model_config(geom, "t", image_res=res)).to(DEV)
            tr = ITRTrainer(student, teacher, lr=3e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                            pipeline_teacher=True, capture_step=cap)
            batch = synth.make_batch(geom, 4, seed=5)
            batch["image"] = tint_images(batch["image"], 77)      # (samples with distinct features: see tint_images)
            batch = {k: v.to(DEV) for k, v in batch.items()}
            idx = torch.arange(4, device=DEV)
            # (the same hard negatives in both runs - every forward, captured ones included: the device draw's position in
            # its random stream differs between an eager and a replayed run, which left the ITM term comparable at 20 % only)
            student.injected_neg_idx = teacher.injected_neg_idx = torch.tensor([1, 2, 3, 0, 2, 3, 0, 1])
            student.keep_injected_neg = teacher.keep_injected_neg = True
            step = lambda: tr.step(batch, idx=idx)
        else:
            from efficientvlm_amd.efficient_models.model_generation import EffXVLMForVQA
            from efficientvlm_amd.models.model_generation import XVLMForVQA
            cfg = lambda role, nd: dict(model_config(geom, role, image_res=res), pad_token_id=0, num_dec_layers=nd)
            student = EffXVLMForVQA(cfg("s", 3)).to(DEV)
            teacher = XVLMForVQA(cfg("t", 6)).to(DEV)
            tr = VQATrainer(student, teacher, lr=5e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                            pipeline_teacher=True, capture_step=cap)
            batch = {k: v.to(DEV) for k, v in synth.make_vqa_batch(geom, 3, seed=5, La=8).items()}
            step = lambda: tr.step(batch)
        student.l0_module.set_lagrangian_warmup_steps(100)
        torch.manual_seed(1)                                  # the gate noise: the same host draws in both runs
        seq = []
        for _ in range(11):
            o = step()
            if o is not None:
                seq.append(o.clone())
        torch.cuda.synchronize()
        if cap:
            assert tr.last_launch == "hipGraph replay" and len(tr._sgraphs) == 2
        seqs[cap] = torch.stack(seq).float().cpu()
        norms[cap] = [float(g.float().norm()) for g in tr.opt.flat_grads]
        del tr, student, teacher
    assert len(seqs[True]) == 10 and torch.isfinite(seqs[True]).all() and all(math.isfinite(n) for n in norms[True])
    assert float(seqs[False][-1, 0]) < float(seqs[False][0, 0])                       # (the steps train)
    assert torch.allclose(seqs[True][:, 0], seqs[False][:, 0], rtol=2e-2), (seqs[True][:, 0], seqs[False][:, 0])
    # per term, the ITM term included (hard negatives injected: the same in both runs)
    tol = torch.full((seqs[True].shape[1],), 4e-2)
    assert bool(((seqs[True] - seqs[False]).abs() <= 2e-3 + tol * seqs[False].abs()).all()), (seqs[True], seqs[False])
    # the last step's gradient norms per optimiser group, within 10 % (round 4: a factor of two)
    for a, b in zip(norms[True], norms[False]):
        assert 0.9 * b <= a <= 1.1 * b, (norms[True], norms[False])


@pytest.mark.parametrize("use_graph", [False, True, "step"])
def test_itr_trainer_with_pipelined_teacher_reproduces_the_unpipelined_trajectory(use_graph, monkeypatch):
    """ITRTrainer(pipeline_teacher=True): the frozen teacher runs one batch ahead on a side stream (TeacherPrefetch; a
    hipGraph per parity whose outputs are consumed in place).  Same gate noise and (deterministic) hard negatives on both
    sides: the loss trajectory over three distinct batches equals the unpipelined trainer's, one call later."""
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.efficient_models.xvlm import XVLMBase
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    from efficientvlm_amd.trainer import ITRTrainer

    def fixed_negatives(self, image_feat, text_feat, idx):
        bs = image_feat.size(0)
        ar = torch.arange(bs, device=image_feat.device)
        return (ar + 1) % bs, (ar + 2) % bs
    monkeypatch.setattr(XVLMBase, "_sample_negatives", fixed_negatives)
    geom = synth.GEOMS["tiny"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    B = 4
    batches = [{k: v.to(DEV) for k, v in synth.make_batch(geom, B, seed=30 + i, ragged=True).items()
                if k in ("image", "text_ids", "text_atts")} for i in range(3)]
    idx = torch.arange(B, device=DEV)
    outs = {}
    for pipe in (False, True):
        student, teacher = EffXVLMforRetrieval(model_config(geom, "s")), TeacherITR(model_config(geom, "t"))
        load_det_weights(student, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 51, geom["std"])
        load_det_weights(teacher, schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False), 52, geom["std"])
        gen = torch.Generator().manual_seed(8)
        with torch.no_grad():
            for n, p in student.l0_module.named_parameters():
                p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
        student.l0_module.set_lagrangian_warmup_steps(10)
        student.to(DEV); teacher.to(DEV)
        eps = [{t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6)
                for t in O.L0_TYPES} for _ in range(3)]
        # "step": the student step itself replays as a hipGraph too (capture_step: first step per teacher parity eager, then
        # captured) - seven steps, so that both parities are captured AND replayed; the gate noise still comes from the
        # per-step injected draws, the Lagrangian ramp from the device-side step counter
        tr = ITRTrainer(student, teacher, lr=1e-3, reg_learning_rate=0.05, dtype=torch.float32, pipeline_teacher=pipe,
                        use_graph=bool(use_graph), capture_step=(use_graph == "step"))
        seq = []
        nsteps = 7 if use_graph == "step" else 3
        while len(eps) < nsteps:
            eps.append({t: torch.rand(getattr(student.l0_module, O.L0_PARAM[t]).shape, generator=gen).clamp(1e-6, 1 - 1e-6)
                        for t in O.L0_TYPES})
        calls = [batches[i % 3] for i in range(nsteps)] + ([batches[0]] if pipe else [])
        for c, b in enumerate(calls):
            i = c - 1 if pipe else c                        # index of the batch whose student step runs in this call
            if i >= 0:
                student.l0_module.injected_eps = {t: e.clone() for t, e in eps[i].items()}
            o = tr.step(b, idx=idx)
            assert (o is None) == (pipe and c == 0)
            if o is not None:
                seq.append(o.clone())
        torch.cuda.synchronize()
        outs[pipe] = torch.stack(seq).cpu()
        if pipe and use_graph == "step":
            assert tr.last_launch == "hipGraph replay" and len(tr._sgraphs) == 2
    assert torch.allclose(outs[True], outs[False], rtol=5e-4, atol=1e-5), (outs[True], outs[False])
    assert float((outs[False][0] - outs[False][2]).abs().max()) > 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# full-width steps of BASELINE configs[2] / [3] / [4] (per-GPU shards, long image sequences) against the oracle
# ---------------------------------------------------------------------------------------------------------------------
# worst single tensor of the ITR-384 / VQA-480 gradient checks (relative L2 against the oracle): the tensors that sit behind
# the ITC logits' 1 / temp = 14 (vision_proj, text_proj) and behind all six ViT layers (class / position embeddings) at
# batch 8 / 3, where nothing averages the bf16 noise of the residual stream
ITR_WORST_TENSOR = 0.25
VQA_WORST_TENSOR = 0.05      # (measured: every tensor <= 2.0 % at B = 3; round 4 admitted 25 %)


def tint_images(image, seed):
    """white-noise images + a per-image colour offset and brightness ramp.  With pure N(0, 1) pixels every image of a batch
    has the same statistics, so at random init the CLS features of all images are nearly identical (the 576 noise patches
    average out); the contrastive loss depends ONLY on the differences between the samples' features, which bf16 activations
    then resolve to ~20 % - any bf16 pipeline would.  Images that differ in colour / brightness, as real ones do, give the
    samples distinct features."""
    g = torch.Generator().manual_seed(seed)
    Bn, _, Hh, Ww = image.shape
    off = 1.5 * torch.randn(Bn, 3, 1, 1, generator=g)
    ramp = torch.linspace(-1.0, 1.0, Ww).view(1, 1, 1, Ww) * torch.randn(Bn, 3, 1, 1, generator=g)
    return image + off + ramp


# the tensors whose gradient comes through the ITC logits alone or sits behind all six ViT layers of the CLS row
ITC_PATH = ("vision_proj.", "text_proj.", "vision_encoder.class_embedding", "vision_encoder.pos_embed.")


@pytest.mark.parametrize("pipelined,B,scene", [(False, 8, "noise"), (True, 8, "noise"), (True, 8, "tinted"), (True, 64, "noise")])
def test_itr_384_step_full_width_bf16_tracks_the_fp32_oracle(pipelined, B, scene):
    """BASELINE configs[2] on one GPU at full width (`pipelined`: the teacher prefetched one call ahead, the image-map
    distillation then fused into the student's attention kernels and its maps never written): EffXVLMforRetrieval student + base teacher, 384 x 384 images = 577
    image tokens (MFMA attention with 26-tile / long-sequence kernels, ragged weight-gradient reductions 8 x 577), L0 gates
    sampled with injected noise, ITRTrainer step in bf16 - step-0 losses against the fp32 CPU oracle on the same weights,
    batch, gate noise and hard negatives (reference: Eff_Retrieval.py:75-213)"""
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    from efficientvlm_amd.trainer import ITRTrainer
    geom = dict(synth.GEOMS["full"], image_res=384)
    # (B = 64: the per-GPU batch configs[2] is quoted on - 36 928 image-token rows, multi-round GEMM launches, 768 attention
    # workgroups per layer; the oracle forward of that batch takes the host about a minute)
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sch = schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True)
    t_sch = schema.xvlm_schema(t_cfg, geom["max_pos"], mlm=False, bbox=False)
    student = EffXVLMforRetrieval(model_config(geom, "s", image_res=384))
    teacher = TeacherITR(model_config(geom, "t", image_res=384))
    s_sd = load_det_weights(student, s_sch, 91, geom["std"])
    t_sd = load_det_weights(teacher, t_sch, 92, geom["std"])
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
            s_sd["l0_module." + n] = p.detach().clone()
    student.l0_module.set_lagrangian_warmup_steps(10)
    student.to(DEV); teacher.to(DEV)
    batch = synth.make_batch(geom, B, seed=19, ragged=True, image_res=384)
    if scene == "tinted":
        batch["image"] = tint_images(batch["image"], 77)
    idx = torch.arange(B)
    idx[2] = idx[1]
    # gradient-level parity at EVERY batch (round 5: also at B = 64, the per-GPU shard configs[2] is quoted on - the oracle's
    # fp32 backward of that batch is a few minutes of host time; lr 0: the step leaves the parameters where the oracle has them)
    with_grads = True
    tr = ITRTrainer(student, teacher, lr=0.0, reg_learning_rate=0.0, dtype=torch.bfloat16, pipeline_teacher=pipelined)
    eps = {t: torch.rand(s_sd["l0_module." + O.L0_PARAM[t]].shape, generator=gen).clamp(1e-6, 1 - 1e-6) for t in O.L0_TYPES}
    s_neg = torch.tensor([(i + 3) % B for i in range(2 * B)])
    t_neg = torch.tensor([(i + 5) % B for i in range(2 * B)])
    student.l0_module.injected_eps = {t: e.clone() for t, e in eps.items()}
    student.injected_neg_idx, teacher.injected_neg_idx = s_neg.clone(), t_neg.clone()
    teacher.keep_injected_neg = True               # (the prefetched teacher runs a warm-up forward and two captures)
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    got = tr.step(dev_batch, idx=idx.to(DEV))
    if pipelined:                                  # the first call only starts the teacher; the image-map term of the second
        assert got is None                         # is formed inside the student's attention kernels (577 keys)
        got = tr.step(dev_batch, idx=idx.to(DEV))
    got = got.cpu()
    torch.cuda.synchronize()
    hip_grads = {n: p.grad.detach().float().cpu().clone() for n, p in student.named_parameters() if p.grad is not None}
    leaves = {k: (v.clone().float().requires_grad_(True) if (with_grads and torch.is_floating_point(v)) else v)
              for k, v in s_sd.items()}
    with torch.set_grad_enabled(with_grads):
        logas = {k[len("l0_module."):]: v for k, v in leaves.items() if k.endswith("_loga")}
        S = O.retrieval_forward(leaves, s_cfg, batch, idx, s_neg, O.l0_forward(logas, True, eps))
        with torch.no_grad():
            T = O.retrieval_forward(t_sd, t_cfg, batch, idx, t_neg)
        kd = O.kd_terms(S, T, with_cross_attn=True)
        consts = O.l0_constants(geom["hidden"], geom["ffn"], geom["heads"], s_cfg["vit_layers"], s_cfg["fusion_layer"],
                                s_cfg["text_layers"] - s_cfg["fusion_layer"])
        lagr, _, _ = O.l0_lagrangian(logas, leaves["l0_module.lambda_1"], leaves["l0_module.lambda_2"], consts, 0,
                                     target_sparsity=0.25, lagrangian_warmup=10)
        total, mix = O.itr_loss_mix(S["loss"], kd, lagr)
    want = torch.stack([total, S["loss"]["loss_itc"], S["loss"]["loss_itm"], mix["loss_kd"], lagr.reshape(())]).detach()
    assert torch.isfinite(got).all()
    assert torch.allclose(got, want, rtol=3e-2, atol=2e-3), (got, want)
    if with_grads:
        # the gradient the three optimisers consume, per tensor, against the oracle's (fp32 autograd on the CPU) - the bounds of
        # the GD test (test_benchmarked_configuration_matches_the_oracle).  Round 3 measured this configuration with a tool
        # only (worst tensor 29 %); the attention backward now rebuilds the probabilities of the 577-key rows in fp32 too.
        total.backward()
        st = grad_parity_stats(hip_grads, {k: v for k, v in leaves.items() if torch.is_tensor(v) and v.requires_grad})
        print("ITR-384 gradient parity:", scene, {k: v for k, v in st.items() if k != "stats"}, st["stats"][:8])
        assert st["global_cos"] > 0.9999 and st["median"] < 1e-2 and st["p90"] < 4e-2 and st["qk_median"] < 2e-2, st
        assert all(c > 0.95 for _, c, _ in st["stats"]), st["stats"][:8]
        if scene == "tinted":            # samples with distinct features: the bounds of the GD test hold for EVERY tensor
            # (measured: worst tensor 9.7 % - class embedding -, query / key projections <= 3.4 %; the scalar `temp`, whose
            # gradient is one cancelling sum over the ITC logits, 18 %)
            assert st["qk_max"] < 0.12 and st["max"] < ITR_WORST_TENSOR, st["stats"][:8]
            assert max(r for r, _, n in st["stats"] if n != "temp") < 0.12, st["stats"][:8]
        else:
            # white-noise images (eight near-identical samples): the tensors behind the ITC logits carry the bf16 noise of the
            # feature DIFFERENCES (see tint_images) - measured 20-29 %, the same with the stored-map and the recomputing
            # attention backward; everything else stays within the GD bounds
            itc = [r for r, _, n in st["stats"] if n.startswith(ITC_PATH)]
            rest = [r for r, _, n in st["stats"] if not n.startswith(ITC_PATH)]
            # (measured, round 5: 29.1 - 29.5 % / 12.6 - 13.4 % at B = 8 - round 4 admitted 32 % / 15 % -, 26.1 % / 9.0 % at B = 64)
            lim_itc, lim_rest = (0.31, 0.145) if B == 8 else (0.29, 0.11)
            assert max(itc) < lim_itc and max(rest) < lim_rest, st["stats"][:8]


@pytest.mark.parametrize("pipelined,B", [(False, 3), (True, 3), (True, 32)])
def test_vqa_480_step_full_width_bf16_tracks_the_fp32_oracle(pipelined, B):
    """BASELINE configs[3] on one GPU at full width: 480 x 480 images = 901 image tokens (the two-pass long-sequence dQ
    kernel, K / V taking turns in LDS), causal answer decoder, VQAL0Module gates; VQATrainer step in bf16 against the fp32
    CPU oracle (reference: Eff_VQA.py:74-200).  B = 32: the per-GPU batch configs[3] is quoted on."""
    from efficientvlm_amd.trainer import VQATrainer
    geom = dict(synth.GEOMS["full"], image_res=480)
    student, teacher, s_sd, t_sd, s_cfg, t_cfg = _vqa_models(geom, 51, 52)
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for n, p in student.l0_module.named_parameters():
            p.copy_(torch.full_like(p, 0.3) if "lambda" in n else torch.randn(p.shape, generator=gen) + 0.5)
            s_sd["l0_module." + n] = p.detach().clone()
    student.l0_module.set_lagrangian_warmup_steps(10)
    student.to(DEV); teacher.to(DEV)
    batch = synth.make_vqa_batch(geom, B, seed=23)
    batch["image"] = torch.randn(B, 3, 480, 480, generator=gen)
    with_grads = True                             # gradient-level parity at B = 3 AND at the quoted shard B = 32 (lr 0)
    tr = VQATrainer(student, teacher, lr=0.0, reg_learning_rate=0.0, dtype=torch.bfloat16, pipeline_teacher=pipelined)
    eps = {t: torch.rand(s_sd["l0_module." + O.L0_PARAM[t]].shape, generator=gen).clamp(1e-6, 1 - 1e-6) for t in O.L0_TYPES_VQA}
    student.l0_module.injected_eps = {t: e.clone() for t, e in eps.items()}
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    got = tr.step(dev_batch)
    if pipelined:                                 # the first call only starts the teacher; the image-map term of the second
        assert got is None                        # is formed inside the student's attention kernels (901 keys, no map in HBM)
        got = tr.step(dev_batch)
    got = got.cpu()
    torch.cuda.synchronize()
    hip_grads = {n: p.grad.detach().float().cpu().clone() for n, p in student.named_parameters() if p.grad is not None}
    tie = lambda sd: {**sd, "text_decoder.cls.predictions.decoder.weight": sd["text_decoder.bert.embeddings.word_embeddings.weight"],
                      "text_decoder.cls.predictions.decoder.bias": sd["text_decoder.cls.predictions.bias"]}
    leaves = {k: (v.clone().float().requires_grad_(True) if (with_grads and torch.is_floating_point(v)) else v)
              for k, v in s_sd.items()}
    with torch.set_grad_enabled(with_grads):
        logas = {k[len("l0_module."):]: v for k, v in leaves.items() if k.endswith("_loga")}
        S = O.vqa_forward(tie(leaves), s_cfg, batch, O.l0_forward(logas, True, eps))
        with torch.no_grad():
            T = O.vqa_forward(tie(t_sd), t_cfg, batch)
        kd = O.vqa_kd_terms(S, T)
        nd = s_cfg["text_layers"] - s_cfg["fusion_layer"]
        consts = O.l0_constants(geom["hidden"], geom["ffn"], geom["heads"], s_cfg["vit_layers"], s_cfg["fusion_layer"], nd, nd)
        lagr, _, _ = O.l0_lagrangian(logas, leaves["l0_module.lambda_1"], leaves["l0_module.lambda_2"], consts, 0,
                                     target_sparsity=0.25, lagrangian_warmup=10)
        total, mix = O.vqa_loss_mix(S["loss"], kd, lagr)
    want = torch.stack([total, S["loss"], mix["loss_kd"], lagr.reshape(())]).detach()
    assert torch.isfinite(got).all()
    assert torch.allclose(got, want, rtol=4e-2, atol=2e-3), (got, want)
    if with_grads:
        # per-tensor gradient parity against the oracle's fp32 autograd (bounds of the GD test; 901-key attention rows rebuilt
        # in fp32 by the one-pass long-sequence backward when the teacher is prefetched, stored bf16 maps otherwise)
        total.backward()
        st = grad_parity_stats(hip_grads, {k: v for k, v in leaves.items() if torch.is_tensor(v) and v.requires_grad})
        print("VQA-480 gradient parity:", {k: v for k, v in st.items() if k != "stats"}, st["stats"][:8])
        assert st["global_cos"] > 0.9999 and st["median"] < 1e-2 and st["p90"] < 3e-2, st
        assert st["qk_max"] < 0.12 and st["qk_median"] < 2e-2, st
        assert st["max"] < VQA_WORST_TENSOR and all(c > 0.95 for _, c, _ in st["stats"]), st["stats"][:8]


@pytest.mark.parametrize("keep", [0.25, 0.5, 0.75])
def test_full_width_pruned_model_matches_the_oracles_masked_dense_forward(keep):
    """BASELINE configs[4] at FULL width (768 hidden, 12 heads, 3072 FFN units, 224x224, 30 tokens): the physically pruned
    X-VLM-small (utils/xvlm_utils.py:37-145) run by the HIP kernels - bf16, the dtype the sweep is measured in, and fp32 -
    against the ORACLE's masked-dense eval forward (efficient_models/model_retrieval.py:76-93 with the 0/1 gates applied as
    multipliers) in fp32 on the CPU: ITC / ITM losses and the ITM logits.  `keep` of every gate vector survives."""
    from efficientvlm_amd import pruning
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.runtime import compute
    geom = synth.GEOMS["full"]
    s_cfg = O.model_cfg(geom, "s")
    B = 6
    batch = synth.make_batch(geom, B, seed=13, ragged=True)
    idx = torch.arange(B)
    neg = torch.tensor([(i + 1 + (i % 2)) % B for i in range(B)] + [(i + 2) % B for i in range(B)])
    gen = torch.Generator().manual_seed(int(keep * 1000))
    sch = schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True)
    want = {}
    for dtype in (torch.float32, torch.bfloat16):
        model = EffXVLMforRetrieval(model_config(geom, "s"))
        sd = load_det_weights(model, sch, 41, geom["std"])
        model.to(DEV).eval()
        if not want:                                   # gates + the oracle's masked-dense answer (once)
            logas = {k[len("l0_module."):]: v for k, v in sd.items() if k.endswith("_loga")}
            shapes = {t + "_z": shp for t, shp in O.l0_shapes(logas).items()}
            zs = {}
            for k, shp in shapes.items():
                n = shp[2] if k.endswith("head_z") else shp[-1]
                nk = max(1, int(round(n * keep)))
                z = torch.zeros(shp[0], n)
                for r in range(shp[0]):
                    z[r, torch.randperm(n, generator=gen)[:nk]] = 1.0
                zs[k] = z.view(shp)
            with torch.no_grad():
                S = O.retrieval_forward(sd, s_cfg, batch, idx, neg, zs)
            want = {"itc": S["loss"]["loss_itc"], "itm": S["loss"]["loss_itm"], "logits": S["logits_dict"]["itm_head_logits"]}
        dev_batch = {k: v.to(DEV) for k, v in batch.items()}
        zd = {k: v.to(DEV) for k, v in zs.items()}
        with torch.no_grad(), compute(dtype):
            n_before = sum(p.numel() for p in model.parameters())
            pruning.update_params(model, zd)
            pruning.prune_model_with_z(zd, model)
            assert sum(p.numel() for p in model.parameters()) < n_before * (0.55 + 0.5 * keep)
            model.injected_neg_idx = neg.clone()
            itc, itm, logits = pruning.retrieval_eval_losses(model, dev_batch["image"], dev_batch["text_ids"],
                                                             dev_batch["text_atts"], idx=idx.to(DEV), with_logits=True)
        rtol = 1e-4 if dtype == torch.float32 else 2e-2
        close(itc, want["itc"], rtol, 1e-6, f"itc keep {keep} {dtype}")
        close(itm, want["itm"], rtol, 1e-6, f"itm keep {keep} {dtype}")
        # ([3B, 2] logits of magnitude ~0.4 at random init, max-norm criterion of helpers.close: bf16 measured 2.6-2.9 %)
        close(logits.float(), want["logits"], rtol if dtype == torch.float32 else 4e-2, 1e-6, f"itm logits keep {keep} {dtype}")
        del model



@pytest.mark.parametrize("keep", [0.75, 0.5, 0.25])
def test_pruned_inference_equals_masked_dense_at_every_sparsity_of_the_sweep(keep):
    """BASELINE configs[4] (25 / 50 / 75 % retained heads + FFN units): the physically pruned model (utils/xvlm_utils.py:37-145:
    heads and FFN units with a zero gate removed) must score exactly what the masked-dense model scores with the same 0/1
    gates - retrieval losses of the eval forward, fp32, tiny geometry, the gates drawn so that `keep` of every gate
    vector survives"""
    from efficientvlm_amd import pruning
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.runtime import compute
    geom = synth.GEOMS["tiny"]
    s_cfg = O.model_cfg(geom, "s")
    model = EffXVLMforRetrieval(model_config(geom, "s"))
    load_det_weights(model, schema.xvlm_schema(s_cfg, geom["max_pos"], mlm=False, bbox=False, l0=True), 33, geom["std"])
    model.to(DEV).eval()
    gen = torch.Generator().manual_seed(int(keep * 100))
    zs = {}
    with torch.no_grad():
        ref = model.l0_module.forward(training=False)
        for k, v in ref.items():
            flat = v.reshape(v.shape[0], -1)
            n = flat.shape[1]
            nk = max(1, int(round(n * keep)))
            z = torch.zeros_like(flat)
            for r in range(flat.shape[0]):
                z[r, torch.randperm(n, generator=gen)[:nk].to(z.device)] = 1.0
            zs[k] = z.view_as(v)
    batch = {k: v.to(DEV) for k, v in synth.make_batch(geom, 5, seed=3, ragged=True).items()}
    idx = torch.arange(5, device=DEV)
    neg = torch.tensor([1, 2, 3, 4, 0, 2, 3, 4, 0, 1])
    with torch.no_grad(), compute(torch.float32):
        model.injected_neg_idx = neg.clone()
        dense = pruning.retrieval_eval_losses(model, batch["image"], batch["text_ids"], batch["text_atts"], idx=idx, zs=zs)
        n_before = sum(p.numel() for p in model.parameters())
        pruning.update_params(model, zs)
        pruning.prune_model_with_z(zs, model, pad_to=8)
        model.injected_neg_idx = neg.clone()
        pruned = pruning.retrieval_eval_losses(model, batch["image"], batch["text_ids"], batch["text_atts"], idx=idx)
    assert sum(p.numel() for p in model.parameters()) < n_before * (0.55 + 0.5 * keep)
    for a, b in zip(dense, pruned):
        close(b, a, 1e-4, 1e-6, f"keep {keep}")
    # round 6: the two encoders side by side (XVLMBase.get_pair_embeds: the text pass on a second stream) - the same bits as
    # the two calls in sequence, eagerly and replayed from a hipGraph
    side = torch.cuda.Stream()
    with torch.no_grad(), compute(torch.bfloat16):
        seq = model.get_pair_embeds(batch["image"], batch["text_ids"], batch["text_atts"])
        par = model.get_pair_embeds(batch["image"], batch["text_ids"], batch["text_atts"], side_stream=side)
        torch.cuda.synchronize()
        for a, b in zip(seq, par):
            assert torch.equal(a, b)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = model.get_pair_embeds(batch["image"], batch["text_ids"], batch["text_atts"], side_stream=side)
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(seq, cap):
            assert torch.equal(a, b)
