"""Distillation losses and the general-distillation (GD) step — the loss helpers every reference driver
duplicates (GeneralDistill.py:60-104) plus the loss mixes of GeneralDistill.py:369-376, Eff_Retrieval.py:165-178 and
Eff_VQA.py:165-176,
with the same names and argument meaning.  MSE and the dual-softmax KL run as HIP reductions (evlm_mse_*, evlm_kl_*);
each term is a device scalar, so a whole step issues no host synchronisation.
"""
import os

import torch

from . import ops


def get_cor_teacher(teacher_reps, student_reps, is_attn=False):
    """GeneralDistill.py:91-104: teacher layer -> student layer map (every k-th state / last map of each block)"""
    teacher_reps = [t.detach() if t is not None else None for t in teacher_reps]    # (None: a map the teacher skipped)
    nt, ns = len(teacher_reps), len(student_reps)
    if is_attn:
        assert nt % ns == 0
        k = nt // ns
        return [teacher_reps[i * k + k - 1] for i in range(ns)]
    assert (nt - 1) % (ns - 1) == 0
    k = (nt - 1) // (ns - 1)
    return [teacher_reps[i * k] for i in range(ns)]


def _kd_pairs(student_reps, teacher_reps, is_attn=False, is_img=False):
    """the (student, teacher) pairs and weights of one get_kd_loss call (GeneralDistill.py:60-82)"""
    pairs, weights = [], []
    for layer, (s, t) in enumerate(zip(student_reps, teacher_reps)):
        if is_attn:
            pairs.append((s, t)); weights.append(float(s.shape[-1]))
        elif is_img and layer == 6:
            continue
        else:
            pairs.append((s, t)); weights.append(1.0)
    return pairs, weights


def get_kd_loss(student_reps=None, teacher_reps=None, is_attn=False, loss=None, device="cuda", is_img=False):
    """GeneralDistill.py:60-82.  `loss`/`device` are accepted for signature compatibility; the MSE is the HIP kernel.
    The reference's torch.where(att <= -1e2, 0, att) is a no-op on probabilities and is not issued."""
    pairs, weights = _kd_pairs(student_reps, teacher_reps, is_attn, is_img)
    return ops.mse_sum(pairs, weights) if pairs else 0


def soft_cross_entropy(predicts, targets, temperature=1.0, ragged=None):
    """GeneralDistill.py:84-89; callers that pre-divide by T (as the reference does) pass temperature=1."""
    return ops.soft_cross_entropy(predicts, targets, temperature, ragged=ragged)


def batch_ragged(batch):
    """ops.Ragged views of a bucket-padded batch's real extents (data.bucket_pad_itr / bucket_pad_vqa: batch['extents'] =
    device int32 [text or question tokens, answer tokens, answer rows, 0]) -> (text / question side, decoder side), or
    (None, None) for a batch in the reference's own 'longest' padding"""
    ext = batch.get("extents") if isinstance(batch, dict) else None
    if ext is None:
        return None, None
    return ops.Ragged(ext, inner=0), ops.Ragged(ext, inner=1, outer=2)


def kd_terms(S, T, temperature=1.0, with_cross_attn=False, fused=None, side=None, ragged=None):
    """the per-pair KD scalars of GeneralDistill.py:300-366 (+ cross-attention maps, Eff_Retrieval.py:141-159).
    fused: {term name: scalar} of terms the attention kernels already produced (fuse_image_map_kd) - same arithmetic,
    no separate pass over the maps.  ragged (ops.Ragged): the text of the batch is bucket-padded - the text-side terms skip
    the token rows beyond the real length in their kernels (their means keep the padded denominators: *_loss_mix rescales)."""
    sh, th, sa, ta = S["hidden_dict"], T["hidden_dict"], S["attention_dict"], T["attention_dict"]
    out = {}
    fused = fused or {}

    names, terms = [], []            # every MSE term of the step goes into ONE grouped launch (ops.mse_terms)
    # student operands: where the batched forward reports its lists as row ranges of un-split tensors, take those
    rows = S.get("batched") or {}
    s_list = lambda d, key: rows[key] if (key in rows and all(x is not None for x in rows[key])) else d[key]

    def pair(name, hkey, akey, is_img=False):
        rag = None if is_img else ragged           # (image tokens are never padded)
        if name + "_hidden" in fused:
            out[name + "_hidden"] = fused[name + "_hidden"]
        else:
            names.append(name + "_hidden")
            terms.append(_kd_pairs(s_list(sh, hkey), get_cor_teacher(th[hkey], sh[hkey]), is_img=is_img) + (rag,))
        if name + "_attn" in fused:
            out[name + "_attn"] = fused[name + "_attn"]
        else:
            names.append(name + "_attn")
            terms.append(_kd_pairs(s_list(sa, akey), get_cor_teacher(ta[akey], sa[akey], True), is_attn=True) + (rag,))

    pair("text", "text_hidden_states", "text_attentions")
    pair("image", "image_hidden_states", "image_attentions", is_img=True)
    pair("itm_pos", "itm_pos_hidden_states", "itm_pos_attentions")
    pair("itm_neg", "itm_neg_hidden_states", "itm_neg_attentions")
    if "mlm_hidden_states" in sh:
        pair("mlm", "mlm_hidden_states", "mlm_attentions")
        out["mlm_logits"] = soft_cross_entropy(S["logits_dict"]["mlm_logits"], T["logits_dict"]["mlm_logits"], temperature)
    out["itm_logits"] = soft_cross_entropy(S["logits_dict"]["itm_head_logits"], T["logits_dict"]["itm_head_logits"], temperature)
    if with_cross_attn:
        sc, tc = S["cross_attention_dict"], T["cross_attention_dict"]
        for nm in ("itm_pos", "itm_neg"):
            k = nm + "_cross_attentions"
            names.append(nm + "_cross")
            terms.append(_kd_pairs(s_list(sc, k), get_cor_teacher(tc[k], sc[k], True), is_attn=True) + (ragged,))
    if side is not None:
        # side = (stream, event recorded behind the student's fusion pass): the grouped MSE forward - and, since autograd
        # runs a node on its forward's stream, its backward - runs beside the task heads (MLM decoder product, the CEs)
        # instead of after / before them: ~0.2 ms of HBM-bound reductions each way off the student's critical path
        stream, fork = side
        cur = torch.cuda.current_stream()
        stream.wait_event(fork)
        with torch.cuda.stream(stream):
            values = ops.mse_terms(terms)
        cur.wait_stream(stream)
        if not torch.cuda.is_current_stream_capturing():
            for v in values:
                if torch.is_tensor(v):
                    v.record_stream(cur)
    else:
        values = ops.mse_terms(terms)
    for name, value in zip(names, values):
        out[name] = value
    return out


_MIX = {}


def _mix_weights(names, rows, device):
    """[len(rows), len(names)] coefficient matrix of a loss mix (cached per device): row r = weights of output r"""
    key = (tuple(names), tuple(tuple(sorted(r.items())) for r in rows), str(device))
    M = _MIX.get(key)
    if M is None:
        M = torch.tensor([[r.get(n, 0.0) for n in names] for r in rows], dtype=torch.float32, device=device)
        _MIX[key] = M
    return M


def gd_loss_mix(loss, kd):
    """GeneralDistill.py:369-376 (general step) / :252-260 (region step: + bbox + giou in the task term).  The ~17 device
    scalars are stacked once and every output is one weighted sum of them (5 small launches instead of ~20 scalar adds /
    multiplies on the step's critical path; the same linear combination, summed in one pass)."""
    task = ["loss_itc", "loss_itm", "loss_mlm"] + (["loss_bbox", "loss_giou"] if "loss_bbox" in loss else [])
    text, img = {"text_attn": 1.0, "text_hidden": 1.0}, {"image_attn": 1.0, "image_hidden": 0.1}
    cross = {k: 1.0 for k in ("itm_neg_attn", "itm_neg_hidden", "itm_pos_attn", "itm_pos_hidden", "mlm_attn", "mlm_hidden")}
    logit = {"itm_logits": 1.0, "mlm_logits": 1.0}
    names = task + list(logit) + list(text) + list(img) + list(cross)
    vals = [loss[n] if n in loss else kd[n] for n in names]
    dev = next(t.device for t in vals if torch.is_tensor(t))
    v = torch.stack([t.float().reshape(()) if torch.is_tensor(t) else torch.full((), float(t), device=dev) for t in vals])
    small = {n: 1.0 for n in task}
    kd_all = {**logit, **text, **img, **cross}
    total_w = {**{n: 0.6 for n in task}, **{n: 0.4 * w for n, w in kd_all.items()}}
    M = _mix_weights(names, [total_w, small, text, img, cross, kd_all], v.device)
    total = (v * M[0]).sum()
    with torch.no_grad():
        rep = (M[1:] * v.detach()).sum(1)
    return total, dict(loss_small=rep[0], loss_text_kd=rep[1], loss_img_kd=rep[2], loss_cross_kd=rep[3], loss_kd=rep[4])


def itr_loss_mix(loss, kd, lagrangian, kd_corr=None):
    """Eff_Retrieval.py:165-178.  kd_corr (device f32 [2], a bucket-padded batch: data.bucket_pad_itr): every text-side term
    is a mean over [.., text tokens, ..] whose kernels summed the REAL token rows but divided by the padded count (and took
    the padded key count as the map terms' weight) - all of them are off by the one factor kd_corr[0] = padded / real length"""
    loss_text_kd = kd["text_hidden"] + kd["text_attn"]
    loss_img_kd = 0.2 * kd["image_hidden"] + kd["image_attn"]
    loss_cross_kd = (kd["itm_neg_hidden"] + kd["itm_pos_hidden"] + kd["itm_pos_attn"] + kd["itm_pos_cross"]
                     + kd["itm_neg_attn"] + kd["itm_neg_cross"]) * 0.5
    if kd_corr is not None:
        loss_text_kd, loss_cross_kd = loss_text_kd * kd_corr[0], loss_cross_kd * kd_corr[0]
    loss_kd = kd["itm_logits"] + (loss_text_kd + loss_img_kd + loss_cross_kd) * 0.33
    loss_small = loss["loss_itc"] + loss["loss_itm"]
    return (loss_kd + loss_small) * 0.5 + lagrangian, dict(loss_text_kd=loss_text_kd, loss_img_kd=loss_img_kd,
                                                           loss_cross_kd=loss_cross_kd, loss_kd=loss_kd)


def vqa_kd_terms(S, T, temperature=1.0, fused=None, ragged=(None, None)):
    """Eff_VQA.py:113-163.  The split of the question encoder's lists at state 4 / map 3 is hard-coded there for the
    (3 text + 3 fusion)-layer student; the decoder-hidden term passes is_img=True (skip of list index 6: a no-op on the
    student's 4 decoder states).  Round 6: every MSE term of the step in ONE grouped launch each way (ops.mse_terms; they
    were ten launch chains); ragged = (question side, decoder side) ops.Ragged of a bucket-padded batch (batch_ragged)."""
    sh, th, sa, ta = S["hidden_dict"], T["hidden_dict"], S["attention_dict"], T["attention_dict"]
    sc, tc = S["cross_attention_dict"], T["cross_attention_dict"]
    s_h, s_a = sh["text_hidden_states"], sa["text_attentions"]
    t_h, t_a = get_cor_teacher(th["text_hidden_states"], s_h), get_cor_teacher(ta["text_attentions"], s_a, True)
    cor = lambda d_t, d_s, key, attn: get_cor_teacher(d_t[key], d_s[key], attn)
    rq, rd = ragged
    fused = fused or {}
    spec = [("text_hidden", _kd_pairs(s_h[:4], t_h[:4]), rq), ("text_attn", _kd_pairs(s_a[:3], t_a[:3], is_attn=True), rq),
            ("cross_hidden", _kd_pairs(s_h[4:], t_h[4:]), rq), ("cross_self_attn", _kd_pairs(s_a[3:], t_a[3:], is_attn=True), rq),
            ("cross_attn", _kd_pairs(sc["cross_attentions"], cor(tc, sc, "cross_attentions", True), is_attn=True), rq)]
    if "image_hidden" not in fused:
        spec.append(("image_hidden", _kd_pairs(sh["image_hidden_states"], cor(th, sh, "image_hidden_states", False), is_img=True), None))
    if "image_attn" not in fused:
        spec.append(("image_attn", _kd_pairs(sa["image_attentions"], cor(ta, sa, "image_attentions", True), is_attn=True), None))
    spec += [("decoder_hidden", _kd_pairs(sh["decoder_hidden_states"], cor(th, sh, "decoder_hidden_states", False), is_img=True), rd),
             ("decoder_attn", _kd_pairs(sa["decoder_attentions"], cor(ta, sa, "decoder_attentions", True), is_attn=True), rd),
             ("decoder_cross", _kd_pairs(sc["decoder_cross_attentions"], cor(tc, sc, "decoder_cross_attentions", True), is_attn=True), rd)]
    values = ops.mse_terms([pw + (rag,) for _, pw, rag in spec])
    out = {name: v for (name, _, _), v in zip(spec, values)}
    for k in ("image_hidden", "image_attn"):
        if k in fused:
            out[k] = fused[k]
    out["logits"] = soft_cross_entropy(S["logits_dict"]["logits"], T["logits_dict"]["logits"], temperature, ragged=rd)
    return out


def vqa_loss_mix(loss_small, kd, lagrangian, kd_corr=None):
    """Eff_VQA.py:165-176.  kd_corr (device f32 [2], a bucket-padded batch: data.bucket_pad_vqa): the question-side terms
    are rescaled by kd_corr[0] = padded / real question length, the decoder-side terms and the logit term by kd_corr[1] =
    (padded rows x padded answer tokens) / (real rows x real tokens) - see itr_loss_mix"""
    loss_text_kd = kd["text_attn"] + kd["text_hidden"]
    loss_img_kd = kd["image_attn"] + kd["image_hidden"] * 0.2
    loss_cross_kd = (kd["cross_hidden"] + kd["cross_self_attn"] + kd["cross_attn"]) * 0.5
    loss_decoder_kd = kd["decoder_attn"] + kd["decoder_hidden"] + kd["decoder_cross"]
    logits_kd = kd["logits"]
    if kd_corr is not None:
        loss_text_kd, loss_cross_kd = loss_text_kd * kd_corr[0], loss_cross_kd * kd_corr[0]
        loss_decoder_kd, logits_kd = loss_decoder_kd * kd_corr[1], logits_kd * kd_corr[1]
    loss_kd = logits_kd + loss_text_kd + loss_img_kd + loss_cross_kd + loss_decoder_kd
    return loss_kd * 0.4 + loss_small * 0.6 + lagrangian, dict(loss_text_kd=loss_text_kd, loss_img_kd=loss_img_kd,
                                                              loss_cross_kd=loss_cross_kd, loss_decoder_kd=loss_decoder_kd,
                                                              loss_kd=loss_kd)


_SIDE = {}


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from _tensors(v)


def kd_teacher_slots(T, S):
    """[(dict name, key, index)] of the teacher tensors the GD KD terms read (kd_terms + get_cor_teacher), given the
    student's output structure - what a pipelined trainer has to keep of a teacher forward"""
    used = []
    for hkey, akey in (("text_hidden_states", "text_attentions"), ("image_hidden_states", "image_attentions"),
                       ("itm_pos_hidden_states", "itm_pos_attentions"), ("itm_neg_hidden_states", "itm_neg_attentions"),
                       ("mlm_hidden_states", "mlm_attentions")):
        if hkey not in S["hidden_dict"]:
            continue
        nt, ns = len(T["hidden_dict"][hkey]), len(S["hidden_dict"][hkey])
        k = (nt - 1) // (ns - 1)
        used += [("hidden_dict", hkey, i * k) for i in range(ns)]
        nt, ns = len(T["attention_dict"][akey]), len(S["attention_dict"][akey])
        k = nt // ns
        used += [("attention_dict", akey, i * k + k - 1) for i in range(ns)]
    used += [("logits_dict", key, None) for key in T["logits_dict"]]
    return used


def student_and_teacher(student_call, teacher_call, ref_tensor, overlap):
    """(student outputs, teacher outputs under no_grad); with `overlap` the teacher forward is issued on a second HIP
    stream (joined before returning) so that it shares the chip with the student forward"""
    if overlap and ref_tensor.is_cuda:
        cur = torch.cuda.current_stream()
        dev = ref_tensor.device
        side = _SIDE.get(dev)
        if side is None:
            side = _SIDE[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            T = teacher_call()
        S = student_call()
        cur.wait_stream(side)
        if not torch.cuda.is_current_stream_capturing():
            for t in _tensors(T):          # allocated on the side stream, consumed on this one
                t.record_stream(cur)
        return S, T
    S = student_call()
    with torch.no_grad():
        T = teacher_call()
    return S, T


def model_kwargs(batch):
    """keyword arguments of XVLM.forward for a general batch or - when the batch carries `idx_to_group_img` - a REGION
    batch (GeneralDistill.py:176-183: image_atts, idx_to_group_img, target_bbox, is_image, ret_bbox_loss=True)"""
    kw = dict(text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"],
              output_attentions=True, output_hidden_states=True)
    if "idx_to_group_img" in batch:
        kw.update(image_atts=batch["image_atts"], idx_to_group_img=batch["idx_to_group_img"],
                  target_bbox=batch["target_bbox"], is_image=batch.get("is_image"), ret_bbox_loss=True)
    return kw


def fuse_image_map_kd(student, teacher_out, batch):
    """Arm the student's image encoder for the FUSED attention-map distillation: when the teacher's maps of this batch
    already exist (pipelined trainer), each student ViT layer's attention kernel compares its probabilities - still in
    registers - with the corresponding teacher map (get_cor_teacher: map i*k + k-1) and returns the layer's term; the
    backward forms dP from the teacher map in-kernel.  Returns the encoder (call collect_fused_kd after the forward) or
    None when the problem does not qualify (fp32 parity path, region batches, missing maps)."""
    from .runtime import compute_dtype
    enc = getattr(getattr(student, "vision_encoder", None), "encoder", None)
    if enc is None or not hasattr(enc, "kd_teacher_maps") or teacher_out is None or "idx_to_group_img" in batch:
        return None
    if os.environ.get("EVLM_NO_FUSED_KD"):          # (A/B switch)
        return None
    if compute_dtype() != torch.bfloat16:
        return None
    maps = teacher_out["attention_dict"].get("image_attentions")
    ns = len(enc.layers)
    if maps is None or len(maps) % ns != 0:
        return None
    cor = get_cor_teacher(maps, [None] * ns, is_attn=True)
    attn = enc.layers[0].self_attn
    if all(isinstance(m, ops.MapRecipe) for m in cor):
        pass                                   # (the teacher kept Q, K and row lse instead of its maps: rebuilt in-kernel)
    elif any(m is None or isinstance(m, ops.MapRecipe) or ops._padded_base(m) is None for m in cor):
        return None
    elif not ops.attention_kd_fusable(cor[0], attn.num_heads, attn.head_dim, cor[0].shape[-1]):
        return None
    enc.kd_teacher_maps = cor
    # ... and the hidden-state term of the same encoder inside each layer's first LayerNorm (round 5): student state i - the
    # input of layer i - against teacher state i * k (get_cor_teacher); the last pair (index 6 of a 6-layer student) is the
    # one get_kd_loss(is_img=True) skips (GeneralDistill.py:71-72), so every term pair is a layer input
    enc.kd_teacher_states = None
    states = teacher_out["hidden_dict"].get("image_hidden_states")
    if (states is not None and not os.environ.get("EVLM_NO_FUSED_HIDDEN_KD") and ns == 6 and (len(states) - 1) % ns == 0
            and hasattr(enc, "kd_teacher_states")):
        k = (len(states) - 1) // ns
        cor_h = [states[i * k] for i in range(ns)]
        if all(torch.is_tensor(t) and t.is_contiguous() and t.dtype == torch.bfloat16 and t.numel() > 0 for t in cor_h):
            enc.kd_teacher_states = [t.detach() for t in cor_h]
    return enc


def collect_fused_kd(enc):
    """{'image_attn': sum of the per-layer terms} produced during the forward the encoder was armed for"""
    terms, enc.kd_teacher_maps = enc.kd_fused, None
    enc.kd_fused = None
    hidden, armed = getattr(enc, "kd_hidden_fused", None), getattr(enc, "kd_teacher_states", None) is not None
    enc.kd_hidden_fused = enc.kd_teacher_states = None
    if not terms or any(t is None for t in terms) or len(terms) != len(enc.layers):
        raise RuntimeError("fused attention-map distillation: a ViT layer did not report its term")
    out = {"image_attn": torch.stack(terms).sum()}
    if armed and hidden is not None:            # (None: the forward did not qualify - no grad, no hidden states asked for)
        if any(t is None for t in hidden) or len(hidden) != len(enc.layers):
            raise RuntimeError("fused hidden-state distillation: a ViT layer did not report its term")
        out["image_hidden"] = torch.stack(hidden).sum()
    return out


def student_forward_fused_kd(student, call, teacher_out, batch):
    """student forward (`call()`) with the image-map distillation fused into its attention kernels when the teacher's
    outputs of this batch exist already (pipelined trainers); returns (student outputs, {'image_attn': term} or {})"""
    enc = fuse_image_map_kd(student, teacher_out, batch) if teacher_out is not None else None
    try:
        S = call()
    except BaseException:
        if enc is not None:                    # disarm the encoder; the forward's own exception is the one to report
            enc.kd_teacher_maps = enc.kd_fused = enc.kd_teacher_states = enc.kd_hidden_fused = None
        raise
    return S, (collect_fused_kd(enc) if enc is not None else {})


def gd_forward(student, teacher, batch, temperature=1.0, overlap_teacher=False, teacher_out=None):
    """student forward (autograd on), teacher forward (no_grad), every KD term and the GD loss mix
    (GeneralDistill.py:289-376; region step :158-262).  batch: dict(image, text_ids, text_atts, text_ids_masked,
    masked_pos, masked_ids [, idx_to_group_img, image_atts, target_bbox, is_image: a region batch]).

    overlap_teacher: the two forwards are independent (each draws its own hard negatives, as in the reference), so the
    teacher runs on a second HIP stream: its HBM-bound kernels (LayerNorm, attention maps) and partly-filled GEMM launches
    share the chip with the student's instead of queueing behind them."""
    kw = model_kwargs(batch)
    call = lambda m: m(batch["image"], batch["text_ids"], batch["text_atts"], **kw)
    fused = {}
    if teacher_out is not None:            # teacher outputs of THIS batch computed earlier (trainer: teacher pipelining)
        enc = fuse_image_map_kd(student, teacher_out, batch)
        try:
            S, T = call(student), teacher_out
        except BaseException:
            if enc is not None:                # disarm the encoder; the forward's own exception is the one to report
                enc.kd_teacher_maps = enc.kd_fused = enc.kd_teacher_states = enc.kd_hidden_fused = None
            raise
        if enc is not None:
            fused = collect_fused_kd(enc)      # (validates the terms: only after a forward that completed)
    else:
        S, T = student_and_teacher(lambda: call(student), lambda: call(teacher), batch["image"], overlap_teacher)
    fork = getattr(student, "kd_fork", None)
    # (the side stream starts at the student's fork event - it sees NOTHING issued after it: only teacher tensors that were
    # complete before this step began may be read there, i.e. the pipelined trainer's; a teacher forward of this step joins
    # the main stream after that event)
    side = (student.text_stream, fork) if (teacher_out is not None and fork is not None
                                           and getattr(student, "text_stream", None) is not None) else None
    kd = kd_terms(S, T, temperature, fused=fused, side=side)
    student.kd_fork = None
    total, mix = gd_loss_mix(S["loss"], kd)
    return total, S, T, kd, mix
