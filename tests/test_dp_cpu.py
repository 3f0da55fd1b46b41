"""Data-parallel host logic on CPU with gloo, world_size 2 (the N>1 path of bench.py / GDTrainer):
gradient slab reduction (mean, bucketed), the ITC all-gather with its slice-only backward (reference
efficient_models/xvlm.py:54-74), parameter broadcast, and rank-sharded synthetic batches."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _reduce_case(rank, world):
    from efficientvlm_amd.trainer import GradReducer
    torch.manual_seed(0)
    base = [torch.randn(1000), torch.randn(37), torch.randn(4096)]
    flats = [b.clone() * (rank + 1) for b in base]            # rank r holds (r+1) * base
    red = GradReducer(flats, bucket_bytes=1024)                # forces several buckets per slab
    assert red.world == world and len(red.buckets) > 3
    red.reduce()
    exp = [b * (sum(range(1, world + 1)) / world) for b in base]
    return all(torch.allclose(f, e, rtol=1e-6, atol=1e-6) for f, e in zip(flats, exp))


def test_grad_reducer_means_flat_slabs_over_ranks():
    assert run2(_reduce_case) == [True, True]


def _allgather_case(rank, world):
    from efficientvlm_amd.efficient_models.xvlm import allgather
    from oracle import xvlm_oracle as O
    g = torch.Generator().manual_seed(123)
    B, E = 3, 8
    img_all = torch.nn.functional.normalize(torch.randn(world * B, E, generator=g), dim=-1)
    txt_all = torch.nn.functional.normalize(torch.randn(world * B, E, generator=g), dim=-1)
    img = img_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    txt = txt_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    gi, gt = allgather(img), allgather(txt)
    ok = torch.equal(gi.detach(), img_all) and torch.equal(gt.detach(), txt_all)
    temp = torch.tensor(0.07)
    loss = O.contrastive_loss(gi, gt, temp)                    # oracle as the checker of the gathered ITC
    ref_i = img_all.clone().requires_grad_(True)
    ref_t = txt_all.clone().requires_grad_(True)
    ref = O.contrastive_loss(ref_i, ref_t, temp)
    ok = ok and torch.allclose(loss, ref)
    loss.backward()
    ref.backward()
    # the backward keeps ONLY this rank's slice of the gathered gradient (no cross-rank reduction)
    ok = ok and torch.allclose(img.grad, ref_i.grad[rank * B:(rank + 1) * B], atol=1e-7)
    ok = ok and torch.allclose(txt.grad, ref_t.grad[rank * B:(rank + 1) * B], atol=1e-7)
    # idx all-gather (soft ITC labels)
    idx = torch.arange(B).view(-1, 1) + rank * B
    ok = ok and torch.equal(allgather(idx).view(-1), torch.arange(world * B))
    return bool(ok)


def test_itc_allgather_forward_and_slice_only_backward():
    assert run2(_allgather_case) == [True, True]


def _shard_case(rank, world):
    from oracle import synth
    geom = synth.GEOMS["tiny"]
    b = synth.make_batch(geom, 4, seed=42 + rank)              # bench.py: weak scaling, seed 42 + rank
    t = b["image"].sum().reshape(1).clone()
    outs = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(outs, t)
    # broadcast of a parameter slab from rank 0 (GDTrainer start-up, apex_ddp_accelerator.py:75-77)
    p = torch.full((10,), float(rank + 1))
    dist.broadcast(p, 0)
    return (float(outs[0]) != float(outs[1])) and bool((p == 1.0).all())


def test_rank_sharded_batches_and_param_broadcast():
    assert run2(_shard_case) == [True, True]


def test_single_process_allgather_is_identity():
    from efficientvlm_amd.efficient_models.xvlm import allgather
    x = torch.randn(4, 3)
    assert allgather(x) is x


def _segment_case(rank, world):
    """early / late segment reduction (overlap path) gives the same result as one full reduce"""
    from efficientvlm_amd.trainer import GradReducer
    torch.manual_seed(1)
    base = torch.randn(5000)
    flat = [base.clone() * (rank + 1)]
    red = GradReducer(flat, bucket_bytes=4096)
    red.reduce_async([flat[0][:1234]])       # "early" part, launched from the backward hook
    red.reduce_async([flat[0][1234:]])       # "late" part, after backward
    red.finish()
    return bool(torch.allclose(flat[0], base * (sum(range(1, world + 1)) / world), rtol=1e-6, atol=1e-6))


def test_segmented_overlap_reduction_equals_full_reduce():
    assert run2(_segment_case) == [True, True]


# ---------------------------------------------------------------------------------------------------------------------
# 2 ranks == 1 rank on the same global batch (distributed tier): oracle forward of the student / teacher on
# each rank's half + the PRODUCT's ITC all-gather (slice-only backward) + the PRODUCT's GradReducer, against one process
# on the whole batch.
# ---------------------------------------------------------------------------------------------------------------------
_B_HALF = 2


def _gd_problem():
    from oracle import schema, synth
    from oracle import xvlm_oracle as O
    geom = synth.GEOMS["tiny"]
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sd = schema.det_weights(schema.xvlm_schema(s_cfg, geom["max_pos"]), 31, geom["std"])
    t_sd = schema.det_weights(schema.xvlm_schema(t_cfg, geom["max_pos"]), 32, geom["std"])
    batch = synth.make_batch(geom, 2 * _B_HALF, seed=17)       # every half: one row with M and one with M - 1 masked tokens
    # hard negatives are mined inside a rank's local batch (xvlm.py:422-458): per half, in local indices
    neg_local = [torch.tensor([1, 0, 1, 0]), torch.tensor([1, 0, 1, 0])]
    return O, s_cfg, t_cfg, s_sd, t_sd, batch, neg_local


def _tied(sd):
    return {**sd, "text_encoder.cls.predictions.decoder.weight": sd["text_encoder.bert.embeddings.word_embeddings.weight"],
            "text_encoder.cls.predictions.decoder.bias": sd["text_encoder.cls.predictions.bias"]}


def _two_rank_gd_case(rank, world):
    from efficientvlm_amd.efficient_models.xvlm import allgather
    from efficientvlm_amd.trainer import GradReducer
    torch.set_num_threads(2)
    O, s_cfg, t_cfg, s_sd, t_sd, batch, neg_local = _gd_problem()
    lo, hi = rank * _B_HALF, (rank + 1) * _B_HALF
    half = {k: v[lo:hi] for k, v in batch.items()}
    names = sorted(s_sd)
    leaves = {k: s_sd[k].clone().requires_grad_(True) for k in names}
    S = O.pretrain_forward(_tied(leaves), s_cfg, half, neg_local[rank])
    i_feat, t_feat = S["features"]
    temp = leaves["temp"].clamp(0.001, 0.5)
    # the reference gathers BOTH feature sets and computes the full similarity matrix on every rank (xvlm.py:395-416)
    S["loss"]["loss_itc"] = O.contrastive_loss(allgather(i_feat), allgather(t_feat), temp)
    with torch.no_grad():
        T = O.pretrain_forward(_tied(t_sd), t_cfg, half, neg_local[rank])
    total, _ = O.gd_loss_mix(S["loss"], O.kd_terms(S, T))
    total.backward()
    flat = torch.cat([(leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])).reshape(-1)
                      for k in names])
    GradReducer([flat], bucket_bytes=1 << 16).reduce()        # mean over ranks, several buckets
    return flat, float(S["loss"]["loss_itc"])


def test_two_rank_gradients_equal_one_rank_gradients_on_the_same_global_batch():
    """What data parallelism must preserve: with the global batch split over 2 ranks (ITM negatives mined per rank, as in
    the reference), the rank-averaged gradient equals the single-process gradient on the whole batch - except that the
    reference's ITC term reaches the encoders scaled by 1 / world: its all-gather backward keeps only the local slice of
    the gathered gradient and nothing sums the slices of the other ranks (efficient_models/xvlm.py:54-74; the
    data-parallel mean then divides by world).  The temperature, a direct input of the (global) ITC loss on every rank,
    keeps its full gradient.  Asserted: equality with the 1-rank gradient of 0.6 (itc* + itm + mlm) + 0.4 kd, where itc*
    is the global ITC loss with the gradient into the features scaled by 1 / world."""
    res = run2(_two_rank_gd_case)
    (g0, itc0), (g1, itc1) = res
    assert torch.equal(g0, g1), "ranks disagree after the reduction"
    O, s_cfg, t_cfg, s_sd, t_sd, batch, neg_local = _gd_problem()
    names = sorted(s_sd)
    leaves = {k: s_sd[k].clone().requires_grad_(True) for k in names}
    B = _B_HALF
    # the same negatives in global indices: [image negatives of rows 0..2B), [text negatives of rows 0..2B)
    neg = torch.cat([neg_local[0][:B], neg_local[1][:B] + B, neg_local[0][B:], neg_local[1][B:] + B])
    S = O.pretrain_forward(_tied(leaves), s_cfg, batch, neg)
    with torch.no_grad():
        T = O.pretrain_forward(_tied(t_sd), t_cfg, batch, neg)
    assert abs(float(S["loss"]["loss_itc"]) - itc0) < 1e-6 and abs(itc0 - itc1) < 1e-7     # every rank sees the global ITC loss
    loss = dict(S["loss"])
    i_feat, t_feat = S["features"]
    half_grad = lambda x: x * 0.5 + x.detach() * 0.5          # same value, gradient scaled by 1 / world
    loss["loss_itc"] = O.contrastive_loss(half_grad(i_feat), half_grad(t_feat), leaves["temp"].clamp(0.001, 0.5))
    total, _ = O.gd_loss_mix(loss, O.kd_terms(S, T))
    total.backward()
    ref = torch.cat([(leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])).reshape(-1)
                     for k in names])
    err = float((g0 - ref).norm() / ref.norm())
    assert err < 2e-5, err
    # and it is NOT the plain global-batch gradient: the ITC contribution to the encoders really is halved
    leaves2 = {k: s_sd[k].clone().requires_grad_(True) for k in names}
    S2 = O.pretrain_forward(_tied(leaves2), s_cfg, batch, neg)
    with torch.no_grad():
        T2 = O.pretrain_forward(_tied(t_sd), t_cfg, batch, neg)
    O.gd_loss_mix(S2["loss"], O.kd_terms(S2, T2))[0].backward()
    plain = torch.cat([(leaves2[k].grad if leaves2[k].grad is not None else torch.zeros_like(leaves2[k])).reshape(-1)
                       for k in names])
    assert float((g0 - plain).norm() / plain.norm()) > 1e-3


def _broadcast_case(rank, world):
    from efficientvlm_amd.optim import FlatAdamW
    from efficientvlm_amd.trainer import broadcast_parameters
    torch.manual_seed(100 + rank)                              # the drivers seed with args.seed + rank
    model = torch.nn.Sequential(torch.nn.Linear(12, 20), torch.nn.LayerNorm(20), torch.nn.Linear(20, 4))
    opt = FlatAdamW(model, lr=1e-3, weight_decay=0.01, lr_mult=2.0)
    before = torch.cat([g["p"] for g in opt.groups]).clone()
    broadcast_parameters(opt)
    after = torch.cat([g["p"] for g in opt.groups])
    views_ok = all(p.data_ptr() >= g["p"].data_ptr() for g in opt.groups for p in g["params"])
    return before, after.clone(), views_ok


def test_parameter_broadcast_aligns_differently_seeded_replicas():
    """trainer.broadcast_parameters (GD, ITR and VQA trainers at world > 1): replicas initialised with seed + rank leave
    with rank 0's parameters, still as views of the optimiser slabs"""
    (b0, a0, ok0), (b1, a1, ok1) = run2(_broadcast_case)
    assert not torch.equal(b0, b1) and torch.equal(a0, b0) and torch.equal(a1, b0) and ok0 and ok1


def _segments_case(rank, world):
    """GDTrainer at world 2 (CPU tensors, gloo): the gradient slabs are cut into STAGES in the order backward completes
    them - the text / fusion / head part (sent when backward enters the image encoder), two image-encoder layer groups
    (sent from hooks inside its backward) and the rest (sent after backward); together they must cover every gradient
    element exactly once, for every EVLM_DP_CUTS setting (a dropped hook point's ranges ride with the next stage)"""
    from efficientvlm_amd.models.model_pretrain import XVLM
    from efficientvlm_amd.trainer import GDTrainer
    from efficientvlm_amd.workload import GEOMS, model_config
    torch.manual_seed(5 + rank)
    geom = GEOMS["tiny"]
    res = {}
    for which, n_stages, hooks, vision in (("all", 4, [2, 4], True), ("vit", 3, [2, 4], False), ("vision", 2, [], True),
                                           ("none", 1, [], False)):
        os.environ["EVLM_DP_CUTS"] = which
        student, teacher = XVLM(model_config(geom, "s")), XVLM(model_config(geom, "t"))
        tr = GDTrainer(student, teacher, dtype=torch.float32, use_graph=False)
        ok = tr.reducer.active and len(tr._stages) == n_stages
        enc = student.vision_encoder.encoder
        ok = ok and sorted(enc.grad_hooks or {}) == hooks and (tr._vision_stage is not None) == vision
        for g in tr.opt.flat_grads:
            g.zero_()
        for seg in tr._stages:
            for v in seg:
                v.add_(1.0)
        # the slabs' padding words (segments are rounded up to 8 elements) are covered too, so == 1 everywhere
        ok = ok and all(bool((g == 1.0).all()) for g in tr.opt.flat_grads)
        if which == "all":
            names = {n: p for n, p in student.named_parameters()}
            p45 = names["vision_encoder.encoder.layers.5.mlp.fc1.weight"].grad
            p01 = names["vision_encoder.encoder.layers.0.mlp.fc1.weight"].grad
            ptx = names["text_encoder.bert.encoder.layer.0.output.dense.weight"].grad
            inside = lambda t, seg: any(v.data_ptr() <= t.data_ptr() < v.data_ptr() + v.numel() * 4 for v in seg)
            ok = ok and (inside(ptx, tr._stages[0]) and inside(p45, tr._stages[1]) and inside(p01, tr._stages[3])
                         and not inside(p01, tr._stages[1]))
            # replicas were built from different seeds: after construction they hold rank 0's parameters
            res["flat"] = torch.cat([g["p"] for g in tr.opt.groups]).clone()
        res[which] = bool(ok)
    os.environ.pop("EVLM_DP_CUTS", None)
    return res


def test_trainer_gradient_stages_partition_the_slabs_and_replicas_start_equal():
    r0, r1 = run2(_segments_case)
    for which in ("all", "vit", "vision", "none"):
        assert r0[which] and r1[which], which
    assert torch.equal(r0["flat"], r1["flat"])


def _itr_stage_case(rank, world):
    """ITRTrainer at world 2 (CPU, gloo): the stages partition the slabs, and the L0 gate parameters / multipliers - whose
    gradient receives a contribution from EVERY gated layer, the first ViT layer included - travel with the LAST stage"""
    from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
    from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
    from efficientvlm_amd.trainer import ITRTrainer
    from efficientvlm_amd.workload import GEOMS, model_config
    torch.manual_seed(9 + rank)
    geom = GEOMS["tiny"]
    student, teacher = EffXVLMforRetrieval(model_config(geom, "s")), TeacherITR(model_config(geom, "t"))
    tr = ITRTrainer(student, teacher, dtype=torch.float32)
    ok = tr.reducer.active and len(tr._stages) == 3 and sorted(student.vision_encoder.encoder.grad_hooks) == [2, 4]
    for g in tr.opt.flat_grads:
        g.zero_()
    for seg in tr._stages:
        for v in seg:
            v.add_(1.0)
    ok = ok and all(bool((g == 1.0).all()) for g in tr.opt.flat_grads)
    inside = lambda t, seg: any(v.data_ptr() <= t.data_ptr() < v.data_ptr() + v.numel() * 4 for v in seg)
    names = dict(student.named_parameters())
    for n, p in names.items():
        if n.startswith("l0_module."):
            ok = ok and inside(p.grad, tr._stages[-1]) and not inside(p.grad, tr._stages[0])
    ok = ok and inside(names["vision_encoder.encoder.layers.5.mlp.fc1.weight"].grad, tr._stages[0])
    ok = ok and inside(names["vision_encoder.encoder.layers.0.mlp.fc1.weight"].grad, tr._stages[-1])
    return bool(ok)


def test_itr_trainer_stages_keep_the_l0_parameters_for_the_last_exchange():
    assert run2(_itr_stage_case) == [True, True]


def _wire_case(rank, world):
    """the opt-in bf16 wire (GradReducer(compress=torch.bfloat16)) against the default fp32 wire on the same gradients:
    the mean's division happens in fp32 before the cast, so every rank's contribution carries ONE bf16 rounding and the
    two-rank sum one more"""
    from efficientvlm_amd.trainer import GradReducer
    g = torch.Generator().manual_seed(77)
    base = torch.randn(20000, generator=g) * torch.logspace(-6, 2, 20000)       # gradients span 8 decades
    mine = base * (1.0 + 0.37 * rank) + 0.01 * rank
    a, b = [mine.clone()], [mine.clone()]
    GradReducer(a, bucket_bytes=1 << 14).reduce()
    GradReducer(b, bucket_bytes=1 << 14, compress=torch.bfloat16).reduce()
    exact = (base * 2.37 + 0.01) / 2.0
    err32 = float((a[0] - exact).abs().max() / exact.abs().max())
    mag = (base.abs() + (base * 1.37 + 0.01).abs()) / 2.0                       # sum of the |contributions| (no cancellation)
    rel = (b[0] - a[0]).abs() / mag.clamp_min(1e-30)
    return err32, float(rel.max()), float((b[0] - a[0]).norm() / a[0].norm())


def test_bf16_wire_is_opt_in_and_bounded_against_the_fp32_wire():
    from efficientvlm_amd.trainer import GradReducer
    assert GradReducer([torch.zeros(4)]).compress is None            # default wire: fp32, as the reference's DDP
    for err32, rel_max, rel_l2 in run2(_wire_case):
        assert err32 < 1e-6
        # roundings to 8 significant bits (unit roundoff 2^-8; each contribution once, the sum once): <= 2 * 2^-8 of the
        # summed magnitudes per element, a few 1e-3 in norm
        assert rel_max < 2 * 2.0 ** -8 * 1.01 and rel_l2 < 4e-3, (rel_max, rel_l2)
