#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE, runs ONLY in the build container.

Imports the *reference* (swaggy-TN/EfficientVLM, read-only at /root/reference) with in-memory
third-party shims, runs its own modules on small seeded inputs and writes the
inputs + expected outputs as small ``.npz`` fixtures under ``tests/golden/``.  Nothing of the
reference (source, bytecode, pickles) is copied: fixtures hold arrays only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py            # writes tests/golden/*.npz

The GPU box never sees /root/reference; tests there consume the committed fixtures.

Weights are NOT stored.  Both sides regenerate them with ``oracle.detinit.det_state_dict`` (a
per-tensor seeded CPU generator keyed by the parameter name) and the fixture carries a
(sum, abs-sum) checksum for every tensor so drift in the generator is detected.
"""
import ast
import json
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("EVLM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle.detinit import det_state_dict, checksums  # noqa: E402
from oracle import synth  # noqa: E402


# --------------------------------------------------------------------------------------------
# shims (none of them touches reference files)
# --------------------------------------------------------------------------------------------
def install_shims():
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    import transformers.optimization as topt

    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer

    def find_pruneable_heads_and_indices(heads, n_heads, head_size, already_pruned_heads):
        mask = torch.ones(n_heads, head_size)
        heads = set(heads) - already_pruned_heads
        for head in heads:
            head = head - sum(1 if h < head else 0 for h in already_pruned_heads)
            mask[head] = 0
        mask = mask.view(-1).contiguous().eq(1)
        index = torch.arange(len(mask))[mask].long()
        return heads, index

    mu.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices
    if not hasattr(topt, "AdamW"):
        topt.AdamW = torch.optim.AdamW

    PTM = mu.PreTrainedModel

    def init_weights(self):  # transformers 4.12.5 semantics
        self.apply(self._init_weights)
        out = self.get_output_embeddings() if hasattr(self, "get_output_embeddings") else None
        if out is not None:
            out.weight = self.get_input_embeddings().weight

    PTM.init_weights = init_weights
    PTM.post_init = lambda self: None
    PTM.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n

    def invert_attention_mask(self, m):  # 4.12.5: (1-m) * -10000
        if m.dim() == 3:
            e = m[:, None, :, :]
        else:
            e = m[:, None, None, :]
        e = e.to(dtype=torch.float32)
        return (1.0 - e) * -10000.0

    PTM.invert_attention_mask = invert_attention_mask

    import transformers.file_utils as fu
    for name in ("add_code_sample_docstrings", "add_start_docstrings",
                 "add_start_docstrings_to_model_forward", "replace_return_docstrings"):
        if not hasattr(fu, name):
            setattr(fu, name, lambda *a, **k: (lambda f: f))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def box_area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    stub("torchvision")
    stub("torchvision.ops")
    stub("torchvision.ops.boxes", box_area=box_area)
    ident = lambda *a, **k: (a[0] if a else None)
    stub("timm")
    stub("timm.models")
    stub("timm.models.vision_transformer", _cfg=lambda **k: {}, PatchEmbed=object)
    stub("timm.models.registry", register_model=lambda f: f)
    stub("timm.models.layers", trunc_normal_=ident, DropPath=torch.nn.Identity, to_2tuple=lambda x: (x, x))

    # torch.load patch: model_pretrain.XVLM hard-codes load_*_params=True
    real_load = torch.load

    def fake_load(path, *a, **k):
        p = str(path)
        if p.endswith("pytorch_model.bin"):
            return {}
        if "clip-vit" in p or p.endswith("none"):
            n = fake_load.num_pos
            return {"vision_model.embeddings.position_embedding.weight": torch.zeros(n, fake_load.width)}
        return real_load(path, *a, **k)

    fake_load.num_pos = 197
    fake_load.width = 768
    torch.load = fake_load
    return fake_load


BERT_JSON = {"hidden_size": 768, "num_attention_heads": 12, "intermediate_size": 3072,
             "num_hidden_layers": 12, "hidden_act": "gelu", "hidden_dropout_prob": 0.0,
             "attention_probs_dropout_prob": 0.0, "layer_norm_eps": 1e-12,
             "max_position_embeddings": 512, "type_vocab_size": 2, "vocab_size": 30522,
             "pad_token_id": 0, "initializer_range": 0.02, "model_type": "bert"}


def write_configs(workdir, geom):
    """geom: dict from oracle.synth.GEOMS; returns (student_cfg, teacher_cfg) model dicts."""
    d = os.path.join(workdir, "data", "bert-base-uncased")
    os.makedirs(d, exist_ok=True)
    bj = dict(BERT_JSON)
    bj.update(hidden_size=geom["hidden"], num_attention_heads=geom["heads"],
              intermediate_size=geom["ffn"], vocab_size=geom["vocab"],
              max_position_embeddings=geom["max_pos"])
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(bj, f)
    cfgs = []
    for tag, nl, lad in (("small", geom["s_vit_layers"], 2), ("base", geom["t_vit_layers"], 4)):
        vj = {"ckpt": "none", "vision_width": geom["hidden"], "patch_size": 16, "hidden_act": "quick_gelu",
              "num_attention_heads": geom["heads"], "attention_dropout": 0.0,
              "intermediate_size": geom["ffn"], "num_hidden_layers": nl, "local_attn_depth": lad}
        p = os.path.join(workdir, f"vit_{tag}.json")
        with open(p, "w") as f:
            json.dump(vj, f)
        cfgs.append(p)
    base = {"use_clip_vit": True, "use_swin": False, "image_res": geom["image_res"], "patch_size": 16,
            "text_encoder": "data/bert-base-uncased", "embed_dim": geom["embed_dim"], "temp": 0.07,
            "accelerator": {"FP16_OPT_LEVEL": "O0"}, "sparsity": 0.25}
    s = dict(base, vision_config=cfgs[0], text_num_hidden_layers=geom["s_text_layers"])
    t = dict(base, vision_config=cfgs[1], text_num_hidden_layers=geom["t_text_layers"])
    return s, t


def load_gd_helpers():
    """ast-extract the three pure loss helpers from GeneralDistill.py (it cannot be imported)."""
    src = open(os.path.join(REF, "GeneralDistill.py")).read()
    tree = ast.parse(src)
    want = {"get_kd_loss", "soft_cross_entropy", "get_cor_teacher"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {"torch": torch, "KLDivLoss": torch.nn.KLDivLoss, "MSELoss": torch.nn.MSELoss}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<gd_helpers>", "exec"), ns)
    return ns["get_kd_loss"], ns["soft_cross_entropy"], ns["get_cor_teacher"]


class MultinomialRecorder:
    """records indices drawn by the per-row torch.multinomial(...).item() loops (xvlm.py:493-506)."""

    def __init__(self):
        self.real = torch.multinomial
        self.log = []

    def __enter__(self):
        def rec(w, n, *a, **k):
            r = self.real(w, n, *a, **k)
            self.log.append(int(r.reshape(-1)[0]))
            return r
        torch.multinomial = rec
        return self

    def __exit__(self, *a):
        torch.multinomial = self.real


def np_(t):
    a = t.detach().to(torch.float32).cpu().numpy() if t.is_floating_point() else t.detach().cpu().numpy()
    return a.copy()   # never alias live parameter memory (constrain_parameters clamps in place)


def tuple_to(d, key, tup):
    for i, t in enumerate(tup):
        d[f"{key}.{i}"] = np_(t)


def kd_terms(get_kd_loss, soft_ce, get_cor, S, T, temperature=1.0, with_cross_attn=False):
    """the per-term KD scalars exactly as GeneralDistill.py:300-366 / Eff_Retrieval.py:113-163 build them."""
    mse = torch.nn.MSELoss()
    sh, th = S["hidden_dict"], T["hidden_dict"]
    sa, ta = S["attention_dict"], T["attention_dict"]
    out = {}

    def pair(name, hkey, akey, is_img=False):
        t_h = get_cor(th[hkey], sh[hkey])
        t_a = get_cor(ta[akey], sa[akey], is_attn=True)
        out[name + "_hidden"] = get_kd_loss(sh[hkey], t_h, False, mse, "cpu", is_img=is_img)
        out[name + "_attn"] = get_kd_loss(sa[akey], t_a, True, mse, "cpu")

    pair("text", "text_hidden_states", "text_attentions")
    pair("image", "image_hidden_states", "image_attentions", is_img=True)
    pair("itm_pos", "itm_pos_hidden_states", "itm_pos_attentions")
    pair("itm_neg", "itm_neg_hidden_states", "itm_neg_attentions")
    if "mlm_hidden_states" in sh:
        pair("mlm", "mlm_hidden_states", "mlm_attentions")
        out["mlm_logits"] = soft_ce(S["logits_dict"]["mlm_logits"] / temperature,
                                    T["logits_dict"]["mlm_logits"] / temperature)
    out["itm_logits"] = soft_ce(S["logits_dict"]["itm_head_logits"] / temperature,
                                T["logits_dict"]["itm_head_logits"] / temperature)
    if with_cross_attn:
        sc, tc = S["cross_attention_dict"], T["cross_attention_dict"]
        for nm in ("itm_pos", "itm_neg"):
            k = nm + "_cross_attentions"
            out[nm + "_cross"] = get_kd_loss(sc[k], get_cor(tc[k], sc[k], is_attn=True), True, mse, "cpu")
    return out


def gd_total(loss, kd):
    """GeneralDistill.py:369-376"""
    loss_small = loss["loss_itc"] + loss["loss_itm"] + loss["loss_mlm"]
    loss_text_kd = kd["text_attn"] + kd["text_hidden"]
    loss_img_kd = kd["image_attn"] + 0.1 * kd["image_hidden"]
    loss_cross_kd = (kd["itm_neg_attn"] + kd["itm_neg_hidden"] + kd["itm_pos_attn"] + kd["itm_pos_hidden"]
                     + kd["mlm_attn"] + kd["mlm_hidden"])
    loss_kd = kd["itm_logits"] + kd["mlm_logits"] + loss_text_kd + loss_img_kd + loss_cross_kd
    return loss_small * 0.6 + loss_kd * 0.4, dict(loss_small=loss_small, loss_text_kd=loss_text_kd,
                                                  loss_img_kd=loss_img_kd, loss_cross_kd=loss_cross_kd,
                                                  loss_kd=loss_kd)


def itr_total(loss, kd, lagr):
    """Eff_Retrieval.py:165-178"""
    loss_text_kd = kd["text_hidden"] + kd["text_attn"]
    loss_img_kd = 0.2 * kd["image_hidden"] + kd["image_attn"]
    loss_cross_kd = (kd["itm_neg_hidden"] + kd["itm_pos_hidden"] + kd["itm_pos_attn"] + kd["itm_pos_cross"]
                     + kd["itm_neg_attn"] + kd["itm_neg_cross"]) * 0.5
    loss_kd = kd["itm_logits"] + (loss_text_kd + loss_img_kd + loss_cross_kd) * 0.33
    loss_small = loss["loss_itc"] + loss["loss_itm"]
    return (loss_kd + loss_small) * 0.5 + lagr, dict(loss_text_kd=loss_text_kd, loss_img_kd=loss_img_kd,
                                                     loss_cross_kd=loss_cross_kd, loss_kd=loss_kd)


def dump_outputs(fx, tag, out, full):
    for dn in ("hidden_dict", "attention_dict", "cross_attention_dict"):
        for k, tup in out[dn].items():
            if full:
                tuple_to(fx, f"{tag}.{k}", tup)
            else:  # checksums only (sum, L2)
                fx[f"{tag}.{k}.chk"] = np.array([[float(t.double().sum()), float(t.double().pow(2).sum().sqrt())]
                                                 for t in tup])
    for k, t in out["logits_dict"].items():
        if full or t.numel() < 4096:
            fx[f"{tag}.{k}"] = np_(t)
        else:
            fx[f"{tag}.{k}.chk"] = np.array([float(t.double().sum()), float(t.double().pow(2).sum().sqrt())])
            fx[f"{tag}.{k}.head"] = np_(t.reshape(-1, t.shape[-1])[:4, :64])
    if "loss" in out:
        for k, t in out["loss"].items():
            fx[f"{tag}.{k}"] = np_(t)


def grads_to(fx, tag, model, full):
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad
        fx[f"{tag}.grad_chk.{n}"] = np.array([float(g.double().sum()), float(g.double().pow(2).sum().sqrt())])
        if full and g.numel() <= 4096:
            fx[f"{tag}.grad.{n}"] = np_(g)
        elif g.numel() > 0:
            flat = g.reshape(-1)
            fx[f"{tag}.grad_head.{n}"] = np_(flat[:64])


# --------------------------------------------------------------------------------------------
def gen_gd(geom_name, B, seed, full, region_rows=0):
    """one GeneralDistill general step: student fwd, teacher fwd (no_grad), KD losses, backward.
    region_rows > 0: a REGION step instead (GeneralDistill.py:158-262) - B images expanded to region_rows (text, region)
    rows, ret_bbox_loss=True, bbox + giou in the task loss."""
    geom = synth.GEOMS[geom_name]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, tcfg = write_configs(work, geom)
    FAKE.num_pos = (geom["image_res"] // 16) ** 2 + 1
    FAKE.width = geom["hidden"]
    from models.model_pretrain import XVLM
    torch.manual_seed(seed)
    student = XVLM(scfg)
    teacher = XVLM(tcfg)
    student.load_state_dict(det_state_dict(student.state_dict(), seed=1000 + seed, std=geom["std"]), strict=True)
    teacher.load_state_dict(det_state_dict(teacher.state_dict(), seed=2000 + seed, std=geom["std"]), strict=True)
    if region_rows:
        batch = synth.make_region_batch(geom, B, region_rows, seed=seed, ragged=True)
        extra = dict(image_atts=batch["image_atts"], idx_to_group_img=batch["idx_to_group_img"],
                     target_bbox=batch["target_bbox"], is_image=batch["is_image"], ret_bbox_loss=True)
    else:
        batch = synth.make_batch(geom, B, seed=seed, ragged=True)
        extra = {}
    student.train()
    teacher.eval()
    get_kd_loss, soft_ce, get_cor = load_gd_helpers()

    torch.manual_seed(seed + 7)
    with MultinomialRecorder() as rec_s:
        S = student(batch["image"], batch["text_ids"], batch["text_atts"],
                    text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"],
                    masked_ids=batch["masked_ids"], output_attentions=True, output_hidden_states=True, **extra)
    with torch.no_grad(), MultinomialRecorder() as rec_t:
        T = teacher(batch["image"], batch["text_ids"], batch["text_atts"],
                    text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"],
                    masked_ids=batch["masked_ids"], output_attentions=True, output_hidden_states=True, **extra)
    kd = kd_terms(get_kd_loss, soft_ce, get_cor, S, T)
    total, mix = gd_total(S["loss"], kd)
    if region_rows:                                   # GeneralDistill.py:257-260
        mix["loss_small"] = mix["loss_small"] + S["loss"]["loss_bbox"] + S["loss"]["loss_giou"]
        total = 0.6 * mix["loss_small"] + 0.4 * mix["loss_kd"]
    total.backward()

    fx = {"meta.geom": np.array(geom_name), "meta.B": np.array(B), "meta.seed": np.array(seed)}
    for k, v in batch.items():
        fx[f"in.{k}"] = np_(v)
    fx["in.student_neg_idx"] = np.array(rec_s.log, dtype=np.int64)   # first B: image negs, next B: text negs
    fx["in.teacher_neg_idx"] = np.array(rec_t.log, dtype=np.int64)
    for tag, m in (("student", student), ("teacher", teacher)):
        for n, (a, b) in checksums(m.state_dict()).items():
            fx[f"{tag}.wchk.{n}"] = np.array([a, b])
    dump_outputs(fx, "student", S, full)
    dump_outputs(fx, "teacher", T, full)
    for k, v in kd.items():
        fx[f"kd.{k}"] = np_(v)
    for k, v in mix.items():
        fx[f"mix.{k}"] = np_(v)
    fx["mix.total"] = np_(total)
    grads_to(fx, "student", student, full)
    return fx


def gen_itr(geom_name, B, seed):
    """one Eff_Retrieval step with L0 masks (efficient_models.*) against the models.* teacher."""
    geom = synth.GEOMS[geom_name]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, tcfg = write_configs(work, geom)
    from efficient_models.model_retrieval import EffXVLMforRetrieval
    from models.model_retrieval import XVLM as TeacherITR
    torch.manual_seed(seed)
    student = EffXVLMforRetrieval(scfg)
    teacher = TeacherITR(tcfg)
    student.load_state_dict(det_state_dict(student.state_dict(), seed=3000 + seed, std=geom["std"]), strict=True)
    teacher.load_state_dict(det_state_dict(teacher.state_dict(), seed=4000 + seed, std=geom["std"]), strict=True)
    # make the gates interesting: head loga ~ N(0.5,1), int loga ~ N(0,1); lambdas non-zero
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 99)
        l0 = student.l0_module
        for nm in ("vision_head_loga", "text_head_loga", "cross_head_loga"):
            getattr(l0, nm).copy_(torch.randn(getattr(l0, nm).shape, generator=g) + 0.5)
        for nm in ("vision_int_loga", "text_int_loga", "cross_int_loga"):
            getattr(l0, nm).copy_(torch.randn(getattr(l0, nm).shape, generator=g))
        l0.lambda_1.fill_(0.3)
        l0.lambda_2.fill_(-0.2)
    l0.set_lagrangian_warmup_steps(10)
    batch = synth.make_batch(geom, B, seed=seed, ragged=True)
    idx = torch.tensor([0, 1, 1, 3][:B], dtype=torch.long)  # a duplicated image id (soft ITC labels)
    student.train()
    teacher.eval()
    get_kd_loss, soft_ce, get_cor = load_gd_helpers()

    eps_log = []
    real_get_eps = l0.get_eps

    def rec_eps(size):
        e = real_get_eps(size)
        eps_log.append(e.clone())
        return e
    l0.get_eps = rec_eps

    torch.manual_seed(seed + 7)
    with MultinomialRecorder() as rec_s:
        S = student(batch["image"], batch["text_ids"], batch["text_atts"], idx=idx,
                    output_attentions=True, output_hidden_states=True)
    with torch.no_grad(), MultinomialRecorder() as rec_t:
        T = teacher(batch["image"], batch["text_ids"], batch["text_atts"], idx=idx,
                    output_attentions=True, output_hidden_states=True)
    kd = kd_terms(get_kd_loss, soft_ce, get_cor, S, T, with_cross_attn=True)
    lagr, exp_s, tgt_s = l0.lagrangian_regularization(3)
    total, mix = itr_total(S["loss"], kd, lagr)
    total.backward()

    fx = {"meta.geom": np.array(geom_name), "meta.B": np.array(B), "meta.seed": np.array(seed)}
    for k in ("image", "text_ids", "text_atts"):
        fx[f"in.{k}"] = np_(batch[k])
    fx["in.idx"] = idx.numpy()
    fx["in.student_neg_idx"] = np.array(rec_s.log, dtype=np.int64)
    fx["in.teacher_neg_idx"] = np.array(rec_t.log, dtype=np.int64)
    for t, e in zip(l0.types, eps_log):
        fx[f"in.eps.{t}"] = np_(e)
    for n, p in l0.named_parameters():
        fx[f"in.l0.{n}"] = np_(p)
    for tag, m in (("student", student), ("teacher", teacher)):
        for n, (a, b) in checksums(m.state_dict()).items():
            fx[f"{tag}.wchk.{n}"] = np.array([a, b])
    dump_outputs(fx, "student", S, True)
    dump_outputs(fx, "teacher", T, True)
    for k, v in kd.items():
        fx[f"kd.{k}"] = np_(v)
    for k, v in mix.items():
        fx[f"mix.{k}"] = np_(v)
    fx["mix.total"] = np_(total)
    fx["mix.lagrangian"] = np_(lagr)
    fx["mix.expected_sparsity"] = np_(exp_s)
    fx["mix.target_sparsity"] = np.array(float(tgt_s))
    grads_to(fx, "student", student, True)
    # eval-mode (deterministic masks) forward: loss pair
    student.eval()
    with torch.no_grad(), MultinomialRecorder() as rec_e:
        torch.manual_seed(seed + 11)
        itc_e, itm_e = student(batch["image"], batch["text_ids"], batch["text_atts"], idx=idx)
        zs = l0.forward(training=False)
    fx["eval.neg_idx"] = np.array(rec_e.log, dtype=np.int64)
    fx["eval.loss_itc"] = np_(itc_e)
    fx["eval.loss_itm"] = np_(itm_e)
    for k, v in zs.items():
        fx[f"eval.z.{k}"] = np_(v)
    return fx


def load_reference_evaluation():
    """ast-extract Eff_Retrieval.evaluation (the driver cannot be imported: ruamel / apex / dataset at module level)"""
    src = open(os.path.join(REF, "Eff_Retrieval.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "evaluation")
    fn.decorator_list = []                                  # (@torch.no_grad(): applied by the caller below)
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    return compile(mod, "<Eff_Retrieval.evaluation>", "exec")


def gen_rerank(seed):
    """the retrieval evaluation / rerank loop of Eff_Retrieval.py:215-319, run by the REFERENCE's own function on the
    reference's EffXVLMforRetrieval (tiny geometry, deterministic gates): score matrices for one rank and for the two
    shards of a 2-rank run"""
    import datetime, time
    geom = synth.GEOMS["tiny"]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, _ = write_configs(work, geom)
    from efficient_models.model_retrieval import EffXVLMforRetrieval
    torch.manual_seed(seed)
    model = EffXVLMforRetrieval(scfg)
    model.load_state_dict(det_state_dict(model.state_dict(), seed=5000 + seed, std=geom["std"]), strict=True)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for n, p in model.l0_module.named_parameters():
            if "lambda" not in n:
                p.copy_(torch.randn(p.shape, generator=g) * 3.0)
    n_img, n_txt, k_test = 7, 11, 4
    bi = synth.make_batch(geom, n_img, seed=seed + 31, ragged=True)
    bt = synth.make_batch(geom, n_txt, seed=seed + 32, ragged=True)
    images, text_ids, text_atts = bi["image"], bt["text_ids"], bt["text_atts"]

    class Enc:
        def __init__(self, ids, atts):
            self.input_ids, self.attention_mask = ids, atts

        def to(self, device):
            return self

    def tokenizer(text, **kw):                             # "texts" are row indices into the pre-tokenised tensors
        rows = torch.tensor(list(text), dtype=torch.long)
        return Enc(text_ids[rows], text_atts[rows])

    class DS:
        text = list(range(n_txt))
        image = list(range(n_img))

    class Loader:
        dataset = DS()

        def __iter__(self):
            for i in range(0, n_img, 3):
                yield images[i:i + 3], torch.arange(i, min(n_img, i + 3))

    class Meter:
        def __init__(self, delimiter="  "):
            pass

        def log_every(self, it, freq, header=None):
            return it

    code = load_reference_evaluation()
    fx = {"meta.seed": np.array(seed), "meta.k_test": np.array(k_test), "in.image": np_(images), "in.text_ids": np_(text_ids),
          "in.text_atts": np_(text_atts)}
    for n, p in model.l0_module.named_parameters():
        fx[f"in.l0.{n}"] = np_(p)
    for n, (a, b) in checksums(model.state_dict()).items():
        fx[f"wchk.{n}"] = np.array([a, b])
    for rank, world in ((0, 1), (0, 2), (1, 2)):
        rw = types.SimpleNamespace(MetricLogger=Meter, get_world_size=lambda w=world: w, get_rank=lambda r=rank: r)
        ns = dict(torch=torch, utils=rw, time=time, datetime=datetime, dist=None, args=types.SimpleNamespace(distributed=False),
                  print=lambda *a, **k: None)
        exec(code, ns)
        with torch.no_grad():
            i2t, t2i, _ = ns["evaluation"](model, Loader(), tokenizer, "cpu",
                                           {"batch_size_test_text": 4, "max_tokens": geom["L"], "k_test": k_test})
        fx[f"out.r{rank}w{world}.i2t"] = np.asarray(i2t)
        fx[f"out.r{rank}w{world}.t2i"] = np.asarray(t2i)
    return fx


def gen_l0(seed):
    """XVLML0Module standalone at FULL size (heads 12, ffn 3072; 6/3/3 layers)."""
    geom = synth.GEOMS["full"]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, _ = write_configs(work, geom)
    from efficient_models.xvlm_l0_module import XVLML0Module
    torch.manual_seed(seed)
    l0 = XVLML0Module(scfg, target_sparsity=0.6, lagrangian_warmup=200)
    g = torch.Generator().manual_seed(seed)
    fx = {}
    with torch.no_grad():
        for nm in ("vision_head_loga", "text_head_loga", "cross_head_loga"):
            p = getattr(l0, nm)
            v = torch.randn(p.shape, generator=g) * 2.0
            v[0, 1] = v[0, 0]                     # exact ties -> top-k tie-break order is pinned
            v[-1, 3] = v[-1, 7]
            p.copy_(v)
        for nm in ("vision_int_loga", "text_int_loga", "cross_int_loga"):
            p = getattr(l0, nm)
            v = torch.randn(p.shape, generator=g) * 2.0
            v[0, 5:9] = v[0, 4]
            v[1, 100] = v[1, 2000]
            p.copy_(v)
        l0.lambda_1.fill_(0.7)
        l0.lambda_2.fill_(1.3)
    for n, p in l0.named_parameters():
        fx[f"in.{n}"] = np_(p)
    eps_log = []
    real = l0.get_eps

    def rec_eps(size):
        e = real(size)
        eps_log.append(e.clone())
        return e
    l0.get_eps = rec_eps
    zs = l0.forward(training=True)
    for t, e in zip(l0.types, eps_log):
        fx[f"in.eps.{t}"] = np_(e)
    for k, v in zs.items():
        fx[f"train.z.{k}"] = np_(v)
    # gradient of a fixed functional of z wrt loga (pins the hard-concrete backward incl. the clamp)
    tot = 0
    for i, (k, v) in enumerate(sorted(zs.items())):
        w = torch.linspace(0.5, 1.5, v.numel()).reshape(v.shape)
        tot = tot + (v * w).sum() * (i + 1)
    lag = []
    for step in (0, 37, 200, 500):
        l, es, ts = l0.lagrangian_regularization(step)
        lag.append([float(l), float(es), float(ts)])
    fx["lagrangian.steps"] = np.array([0, 37, 200, 500])
    fx["lagrangian.triples"] = np.array(lag, dtype=np.float64)
    l, _, _ = l0.lagrangian_regularization(37)
    (tot + l).backward()
    for n, p in l0.named_parameters():
        fx[f"grad.{n}"] = np_(p.grad)
    ze = l0.forward(training=False)
    for k, v in ze.items():
        fx[f"eval.z.{k}"] = np_(v)
    res = l0.calculate_model_size(ze)
    fx["eval.model_size_json"] = np.array(json.dumps(res, default=lambda o: o.tolist() if hasattr(o, "tolist") else float(o)))
    fx["meta.prunable_model_size"] = np.array(l0.prunable_model_size)
    fx["meta.params_per_head"] = np.array(l0.params_per_head)
    fx["meta.params_per_intermediate_dim"] = np.array(l0.params_per_intermediate_dim)
    l0.constrain_parameters()
    for n, p in l0.named_parameters():
        fx[f"constrained.{n}"] = np_(p)
    return fx


def gen_kd_helpers(seed):
    """the three loss helpers on random lists of the §3.1 lengths."""
    get_kd_loss, soft_ce, get_cor = load_gd_helpers()
    g = torch.Generator().manual_seed(seed)
    fx = {}
    mse = torch.nn.MSELoss()
    s_h = [torch.randn(2, 5, 8, generator=g) for _ in range(7)]
    t_h = [torch.randn(2, 5, 8, generator=g) for _ in range(13)]
    s_a = [torch.softmax(torch.randn(2, 3, 5, 5, generator=g), -1) for _ in range(6)]
    t_a = [torch.softmax(torch.randn(2, 3, 5, 5, generator=g), -1) for _ in range(12)]
    s_l = torch.randn(2, 4, 50, generator=g)
    t_l = torch.randn(2, 4, 50, generator=g)
    tuple_to(fx, "in.s_h", s_h); tuple_to(fx, "in.t_h", t_h)
    tuple_to(fx, "in.s_a", s_a); tuple_to(fx, "in.t_a", t_a)
    fx["in.s_l"], fx["in.t_l"] = np_(s_l), np_(t_l)
    ch, ca = get_cor(t_h, s_h), get_cor(t_a, s_a, is_attn=True)
    fx["out.hidden"] = np_(get_kd_loss(s_h, ch, False, mse, "cpu"))
    fx["out.hidden_img"] = np_(get_kd_loss(s_h, ch, False, mse, "cpu", is_img=True))
    fx["out.attn"] = np_(get_kd_loss(s_a, ca, True, mse, "cpu"))
    fx["out.soft_ce"] = np_(soft_ce(s_l / 2.0, t_l / 2.0))
    fx["out.cor_hidden_idx"] = np.array([next(j for j, t in enumerate(t_h) if t is not None and torch.equal(t, c)) for c in ch])
    fx["out.cor_attn_idx"] = np.array([next(j for j, t in enumerate(t_a) if torch.equal(t, c)) for c in ca])
    return fx


def gen_optim(seed):
    """parameter grouping of the reference's optimisers (optim.py:4-21 create_L0_optimizer, :23-69 create_optimizer) on
    the tiny student models: which parameter NAME lands in which group with which lr / weight decay.  (The arithmetic of
    the AdamW steps is transformers 4.12.5's, which this container does not have: the class is shimmed with
    torch.optim.AdamW only to let the reference code build its groups.)"""
    import utils
    import optim as ref_optim
    geom = synth.GEOMS["tiny"]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, tcfg = write_configs(work, geom)
    FAKE.num_pos = (geom["image_res"] // 16) ** 2 + 1
    FAKE.width = geom["hidden"]
    from efficient_models.model_retrieval import EffXVLMforRetrieval
    from models.model_pretrain import XVLM as PretrainXVLM
    out = {}
    args = utils.AttrDict(dict(lr=1e-4, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1))
    for tag, model in (("itr_student", EffXVLMforRetrieval(scfg)), ("gd_student", PretrainXVLM(scfg))):
        names = {id(p): n for n, p in model.named_parameters()}
        opt = ref_optim.create_optimizer(args, model)
        out[tag] = {"init_params": sorted(getattr(model, "init_params", []) or []),
                    "groups": [{"lr": g["lr"], "weight_decay": g["weight_decay"],
                                "names": [names[id(p)] for p in g["params"]]} for g in opt.param_groups]}
        if hasattr(model, "l0_module"):
            l0n = {id(p): n for n, p in model.l0_module.named_parameters()}
            o1, o2 = ref_optim.create_L0_optimizer(args, model.l0_module)
            out[tag + ".l0"] = [{"lr": g["lr"], "weight_decay": g["weight_decay"], "betas": list(g["betas"]), "eps": g["eps"],
                                 "names": [l0n[id(p)] for p in g["params"]]} for g in o1.param_groups]
            out[tag + ".lagrangian"] = [{"lr": g["lr"], "weight_decay": g["weight_decay"], "betas": list(g["betas"]),
                                         "eps": g["eps"], "names": [l0n[id(p)] for p in g["params"]]}
                                        for g in o2.param_groups]
    main = ref_optim.create_optimizer(args, PretrainXVLM(scfg)).param_groups[0]
    out["adamw_defaults"] = {"betas": list(main["betas"]), "eps": main["eps"]}
    return out


def load_vqa_helpers():
    """ast-extract get_kd_loss / soft_cross_entropy / get_cor_teacher from Eff_VQA.py (it cannot be imported)"""
    src = open(os.path.join(REF, "Eff_VQA.py")).read()
    tree = ast.parse(src)
    want = {"get_kd_loss", "soft_cross_entropy", "get_cor_teacher"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {"torch": torch, "KLDivLoss": torch.nn.KLDivLoss, "MSELoss": torch.nn.MSELoss}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<vqa_helpers>", "exec"), ns)
    return ns["get_kd_loss"], ns["soft_cross_entropy"], ns["get_cor_teacher"]


def gen_vqa(geom_name, B, seed):
    """one Eff_VQA.py training step (:95-176): EffXVLMForVQA student with VQAL0Module gates (decoder gates included)
    against the XVLMForVQA teacher - weighted answer LM loss, text / cross / image / decoder hidden + attention KD, logit
    KD, Lagrangian - and the backward."""
    geom = synth.GEOMS[geom_name]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, tcfg = write_configs(work, geom)
    nd_s = geom["s_text_layers"] - geom["s_text_layers"] // 2
    nd_t = geom["t_text_layers"] - geom["t_text_layers"] // 2
    scfg.update(pad_token_id=0, num_dec_layers=nd_s)
    tcfg.update(pad_token_id=0, num_dec_layers=nd_t)
    sys.modules["dataset"] = types.ModuleType("dataset")          # only build_tokenizer is imported (inference path)
    sys.modules["dataset"].build_tokenizer = lambda *a, **k: None
    from efficient_models.model_generation import EffXVLMForVQA
    from models.model_generation import XVLMForVQA
    torch.manual_seed(seed)
    student = EffXVLMForVQA(scfg)
    teacher = XVLMForVQA(tcfg)
    student.load_state_dict(det_state_dict(student.state_dict(), seed=5000 + seed, std=geom["std"]), strict=True)
    teacher.load_state_dict(det_state_dict(teacher.state_dict(), seed=6000 + seed, std=geom["std"]), strict=True)
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 99)
        l0 = student.l0_module
        for nm in ("vision_head_loga", "text_head_loga", "cross_head_loga", "decoder_head_loga"):
            getattr(l0, nm).copy_(torch.randn(getattr(l0, nm).shape, generator=g) + 0.5)
        for nm in ("vision_int_loga", "text_int_loga", "cross_int_loga", "decoder_int_loga"):
            getattr(l0, nm).copy_(torch.randn(getattr(l0, nm).shape, generator=g))
        l0.lambda_1.fill_(0.3)
        l0.lambda_2.fill_(-0.2)
    l0.set_lagrangian_warmup_steps(10)
    batch = synth.make_vqa_batch(geom, B, seed=seed)
    NS = types.SimpleNamespace
    question = NS(input_ids=batch["question_ids"], attention_mask=batch["question_atts"])
    answer = NS(input_ids=batch["answer_ids"], attention_mask=batch["answer_atts"])
    k = batch["k"].tolist()
    student.train()
    teacher.eval()
    get_kd_loss, soft_ce, get_cor = load_vqa_helpers()
    eps_log = []
    real_get_eps = l0.get_eps

    def rec_eps(size):
        e = real_get_eps(size)
        eps_log.append(e.clone())
        return e
    l0.get_eps = rec_eps
    torch.manual_seed(seed + 7)
    S = student(batch["image"], question, answer, train=True, k=k, weights=batch["weights"], output_attentions=True,
                output_hidden_states=True, stop_prune=False)
    with torch.no_grad():
        T = teacher(batch["image"], question, answer, train=True, k=k, weights=batch["weights"], output_attentions=True,
                    output_hidden_states=True)
    mse = torch.nn.MSELoss()
    sh, th, sa, ta = S["hidden_dict"], T["hidden_dict"], S["attention_dict"], T["attention_dict"]
    sc, tc = S["cross_attention_dict"], T["cross_attention_dict"]
    kd = {}
    s_text_h, t_text_h = sh["text_hidden_states"], get_cor(th["text_hidden_states"], sh["text_hidden_states"])
    s_text_a = sa["text_attentions"]
    t_text_a = get_cor(ta["text_attentions"], s_text_a, is_attn=True)
    t_cross_a = get_cor(tc["cross_attentions"], sc["cross_attentions"], is_attn=True)
    # Eff_VQA.py:118-137 hard-codes the split of a (3 text + 3 fusion)-layer student: [:4] / [4:] states, [:3] / [3:] maps
    kd["text_hidden"] = get_kd_loss(s_text_h[:4], t_text_h[:4], False, mse, "cpu")
    kd["text_attn"] = get_kd_loss(s_text_a[:3], t_text_a[:3], True, mse, "cpu")
    kd["cross_hidden"] = get_kd_loss(s_text_h[4:], t_text_h[4:], False, mse, "cpu")
    kd["cross_self_attn"] = get_kd_loss(s_text_a[3:], t_text_a[3:], True, mse, "cpu")
    kd["cross_attn"] = get_kd_loss(sc["cross_attentions"], t_cross_a, True, mse, "cpu")
    kd["image_hidden"] = get_kd_loss(sh["image_hidden_states"], get_cor(th["image_hidden_states"], sh["image_hidden_states"]),
                                     False, mse, "cpu", is_img=True)
    kd["image_attn"] = get_kd_loss(sa["image_attentions"], get_cor(ta["image_attentions"], sa["image_attentions"], is_attn=True),
                                   True, mse, "cpu")
    kd["decoder_hidden"] = get_kd_loss(sh["decoder_hidden_states"],
                                       get_cor(th["decoder_hidden_states"], sh["decoder_hidden_states"]), False, mse, "cpu",
                                       is_img=True)
    kd["decoder_attn"] = get_kd_loss(sa["decoder_attentions"],
                                     get_cor(ta["decoder_attentions"], sa["decoder_attentions"], is_attn=True), True, mse, "cpu")
    kd["decoder_cross"] = get_kd_loss(sc["decoder_cross_attentions"],
                                      get_cor(tc["decoder_cross_attentions"], sc["decoder_cross_attentions"], is_attn=True),
                                      True, mse, "cpu")
    kd["logits"] = soft_ce(S["logits_dict"]["logits"] / 1.0, T["logits_dict"]["logits"] / 1.0)
    # Eff_VQA.py:165-176
    loss_text_kd = kd["text_attn"] + kd["text_hidden"]
    loss_img_kd = kd["image_attn"] + kd["image_hidden"] * 0.2
    loss_cross_kd = (kd["cross_hidden"] + kd["cross_self_attn"] + kd["cross_attn"]) * 0.5
    loss_decoder_kd = kd["decoder_attn"] + kd["decoder_hidden"] + kd["decoder_cross"]
    loss_kd = kd["logits"] + loss_text_kd + loss_img_kd + loss_cross_kd + loss_decoder_kd
    lagr, exp_s, tgt_s = l0.lagrangian_regularization(3)
    total = loss_kd * 0.4 + S["loss"] * 0.6 + lagr
    total.backward()

    fx = {"meta.geom": np.array(geom_name), "meta.B": np.array(B), "meta.seed": np.array(seed)}
    for kk, v in batch.items():
        fx[f"in.{kk}"] = np_(v)
    for t, e in zip(l0.types, eps_log):
        fx[f"in.eps.{t}"] = np_(e)
    fx["meta.l0_types"] = np.array(list(l0.types))
    for n, p in l0.named_parameters():
        fx[f"in.l0.{n}"] = np_(p)
    for tag, m in (("student", student), ("teacher", teacher)):
        for n, (a, b) in checksums(m.state_dict()).items():
            fx[f"{tag}.wchk.{n}"] = np.array([a, b])
    for tag, out in (("student", S), ("teacher", T)):
        for dn in ("hidden_dict", "attention_dict", "cross_attention_dict"):
            for kk, tup in out[dn].items():
                tuple_to(fx, f"{tag}.{kk}", tup)
        fx[f"{tag}.logits"] = np_(out["logits_dict"]["logits"])
        fx[f"{tag}.loss"] = np_(out["loss"])
    for kk, v in kd.items():
        fx[f"kd.{kk}"] = np_(v)
    for kk, v in dict(loss_text_kd=loss_text_kd, loss_img_kd=loss_img_kd, loss_cross_kd=loss_cross_kd,
                      loss_decoder_kd=loss_decoder_kd, loss_kd=loss_kd, total=total, lagrangian=lagr,
                      expected_sparsity=exp_s).items():
        fx[f"mix.{kk}"] = np_(v)
    fx["mix.target_sparsity"] = np.array(float(tgt_s))
    fx["meta.prunable_model_size"] = np.array(int(l0.prunable_model_size))
    grads_to(fx, "student", student, True)
    return fx


def gen_ckpt_remap(seed):
    """checkpoint load / remap contract (efficient_models/xvlm.py:183-208 load_pretrained, models/vit.py:222-247
    interpolate_pos_embed): a small synthetic checkpoint through the reference's own loader"""
    import efficient_models.xvlm as rx
    g = torch.Generator().manual_seed(seed)
    ck = {"vision_encoder.position_ids": torch.arange(5)[None],
          "vision_encoder.pos_embed.weight": torch.randn(5, 8, generator=g),            # 2x2 patches + cls
          "vision_encoder.class_embedding": torch.randn(8, generator=g),
          "text_encoder.bert.embeddings.word_embeddings.weight": torch.randn(11, 8, generator=g),
          "text_encoder.bert.encoder.layer.0.attention.self.query.weight": torch.randn(8, 8, generator=g),
          "text_encoder.cls.predictions.bias": torch.randn(11, generator=g),
          "temp": torch.tensor(0.07)}
    fx = {"in." + k: np_(v) for k, v in ck.items()}
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    path = os.path.join(work, "ckpt.th")
    real_load = torch.load.__wrapped__ if hasattr(torch.load, "__wrapped__") else None
    torch.save({"model": ck}, path)
    cfg = {"image_res": 48, "patch_size": 16, "use_clip_vit": True}                     # 3x3 patches: 5 -> 10 tokens
    out = rx.load_pretrained(path, cfg, is_eval=False, load_text=True)
    for k, v in out.items():
        fx["out." + k] = np_(v)
    out_eval = rx.load_pretrained(path, cfg, is_eval=True, load_text=True)
    fx["eval_keys"] = np.array(sorted(out_eval.keys()))
    for n, res in ((9, "same"), (36, "up6")):
        fx[f"interp.{res}"] = np_(rx.interpolate_pos_embed(torch.from_numpy(fx["in.vision_encoder.pos_embed.weight"])[None]
                                                          if n != 9 else out["vision_encoder.pos_embed.weight"][None],
                                                          num_patches=n, num_extra_tokens=1))
    return fx


def gen_vqa_remap(seed):
    """EffXVLMForVQA.load_pretrained (efficient_models/model_generation.py:57-96, non-eval): a GD-style pre-training checkpoint
    (text encoder under `text_encoder.bert.*`) loaded into the VQA student - its text layers feed the question encoder,
    its fusion layers are MOVED into the answer decoder (layer indices re-based).  Records, per parameter of the VQA model,
    the (sum, abs-sum) checksum after loading and which keys kept their construction-time value."""
    geom = synth.GEOMS["tiny"]
    work = tempfile.mkdtemp(prefix="evlm_oracle_")
    os.chdir(work)
    scfg, _ = write_configs(work, geom)
    scfg.update(pad_token_id=0, num_dec_layers=geom["s_text_layers"] - geom["s_text_layers"] // 2)
    sys.modules["dataset"] = types.ModuleType("dataset")
    sys.modules["dataset"].build_tokenizer = lambda *a, **k: None
    from efficient_models.model_generation import EffXVLMForVQA
    from models.model_pretrain import XVLM
    FAKE.num_pos = (geom["image_res"] // 16) ** 2 + 1
    FAKE.width = geom["hidden"]
    torch.manual_seed(seed)
    pre = XVLM(scfg)
    ck = det_state_dict(pre.state_dict(), seed=7000 + seed, std=geom["std"])
    path = os.path.join(work, "pretrain.th")
    torch.save({"model": ck}, path)
    vqa = EffXVLMForVQA(scfg)
    vqa.load_state_dict(det_state_dict(vqa.state_dict(), seed=8000 + seed, std=geom["std"]), strict=True)
    before = {k: v.clone() for k, v in vqa.state_dict().items()}
    vqa.load_pretrained(path, scfg, is_eval=False)
    fx = {"meta.seed": np.array(seed)}
    for n, (a, b) in checksums(vqa.state_dict()).items():
        fx[f"after.{n}"] = np.array([a, b])
    fx["untouched"] = np.array(sorted(k for k, v in vqa.state_dict().items()
                                      if torch.is_floating_point(v) and torch.equal(v, before[k])))
    fx["ckpt_keys"] = np.array(sorted(ck.keys()))
    return fx


def save(name, fx):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, name)
    np.savez_compressed(p, **fx)
    print(f"wrote {p}  ({os.path.getsize(p) / 1024:.1f} KiB, {len(fx)} arrays)")


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is only present in the build container"
    sys.path.insert(0, REF)
    FAKE = install_shims()
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29917")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["kd", "l0", "gd_tiny", "itr_tiny", "gd_full", "optim", "ckpt", "gd_region_tiny",
                              "gd_region_full", "vqa_tiny", "vqa_remap"]
    if "ckpt" in which:
        save("ckpt_remap.npz", gen_ckpt_remap(13))
    if "optim" in which:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "optim_groups.json"), "w") as f:
            json.dump(gen_optim(7), f, indent=1, sort_keys=True)
        print("wrote", os.path.join(OUT, "optim_groups.json"))
    if "kd" in which:
        save("kd_helpers.npz", gen_kd_helpers(5))
    if "l0" in which:
        save("l0_full.npz", gen_l0(11))
    if "gd_tiny" in which:
        save("gd_tiny.npz", gen_gd("tiny", B=3, seed=3, full=True))
    if "itr_tiny" in which:
        save("itr_tiny.npz", gen_itr("tiny", B=4, seed=4))
    if "rerank_tiny" in which:
        save("rerank_tiny.npz", gen_rerank(seed=6))
    if "gd_full" in which:
        save("gd_full.npz", gen_gd("full", B=2, seed=2, full=False))
    if "vqa_remap" in which:
        save("vqa_remap.npz", gen_vqa_remap(17))
    if "vqa_tiny" in which:
        save("vqa_tiny.npz", gen_vqa("tiny", B=3, seed=12))
    if "gd_region_tiny" in which:
        save("gd_region_tiny.npz", gen_gd("tiny", B=3, seed=6, full=True, region_rows=6))
    if "gd_region_full" in which:
        save("gd_region_full.npz", gen_gd("full", B=2, seed=8, full=False, region_rows=4))
    dist.destroy_process_group()
