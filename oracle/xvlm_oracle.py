"""CPU restatement (plain PyTorch, fp32) of the EfficientVLM distillation hot path.

TEST INFRASTRUCTURE — the checker, never the product.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this file.  It is written from SURVEY.md §8a, function by
function, against the reference tree (paths below are relative to the reference checkout) and is
pinned by the golden fixtures in tests/golden/ that oracle/gen_golden.py captured from the
reference itself (tests/test_oracle_golden.py).

Everything is a pure function of a *state dict* (parameter name -> tensor, the reference's own
checkpoint keys) so it shares no code with the nn.Module boundary it checks.
"""
import math

import torch
import torch.nn.functional as F

LIMIT_A, LIMIT_B, EPSILON = -0.1, 1.1, 1e-6   # efficient_models/xvlm_l0_module.py:16


# ---------------------------------------------------------------------------------------------
# configuration helpers
# ---------------------------------------------------------------------------------------------
def model_cfg(geom, role):
    """shape config of the student ('s') or teacher ('t') of oracle.synth.GEOMS[...]"""
    vit_layers = geom[f"{role}_vit_layers"]
    text_layers = geom[f"{role}_text_layers"]
    return dict(hidden=geom["hidden"], heads=geom["heads"], ffn=geom["ffn"], vocab=geom["vocab"],
                image_res=geom["image_res"], patch=16, embed_dim=geom["embed_dim"],
                vit_layers=vit_layers, local_attn_depth=2 if vit_layers == 6 else 4,
                text_layers=text_layers, fusion_layer=text_layers // 2,
                bert_eps=1e-12, vit_eps=1e-5)


def quick_gelu(x):
    # transformers ACT2FN["quick_gelu"]: x * sigmoid(1.702 x)  (configs/config_clipvit*.json hidden_act)
    return x * torch.sigmoid(1.702 * x)


# ---------------------------------------------------------------------------------------------
# CLIP ViT  (efficient_models/eff_vit.py == models/clip_vit.py with z=None)
# ---------------------------------------------------------------------------------------------
def vit_attention(sd, p, x, heads, mask=None, head_z=None, head_layer_z=None):
    """CLIPAttention.forward, eff_vit.py:123-204.  x [B,N,d] -> (out [B,N,d], probs [B,H,N,N])"""
    B, N, d = x.shape
    dh = d // heads
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * dh ** -0.5        # :134
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    sh = lambda t: t.view(B, N, heads, dh).transpose(1, 2)
    s = sh(q) @ sh(k).transpose(-1, -2)                                                  # :144
    if mask is not None:
        s = s + mask                                                                     # :163-164
    probs = torch.softmax(s, dim=-1)                                                     # :167
    o = probs @ sh(v)                                                                    # :181 (dropout p=0)
    if head_z is not None:
        o = o * head_z                                                                   # :194-195
    o = o.transpose(1, 2).reshape(B, N, d)
    o = F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])                  # :199
    if head_layer_z is not None:
        o = o * head_layer_z                                                             # :201-202
    return o, probs


def vit_mlp(sd, p, x, mlp_z=None):
    """CLIPMLP.forward, eff_vit.py:214-220 — gate BEFORE quick-GELU"""
    h = F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])
    if mlp_z is not None:
        h = h * mlp_z
    return F.linear(quick_gelu(h), sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def vit_layer(sd, p, x, cfg, mask=None, head_z=None, head_layer_z=None, mlp_z=None):
    """CLIPEncoderLayer.forward, eff_vit.py:231-273 (pre-LN residual block)"""
    h = F.layer_norm(x, (x.shape[-1],), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg["vit_eps"])
    a, probs = vit_attention(sd, p + "self_attn.", h, cfg["heads"], mask, head_z, head_layer_z)
    x = x + a
    h = F.layer_norm(x, (x.shape[-1],), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg["vit_eps"])
    x = x + vit_mlp(sd, p + "mlp.", h, mlp_z)
    return x, probs


def vit_forward(sd, p, image, cfg, head_z=None, head_layer_z=None, mlp_z=None,
                idx_to_group_img=None, image_atts=None):
    """CLIPVisionTransformer.forward (eff_vit.py:432-474) + CLIPEncoder.forward (:290-383).

    returns (out, hidden_states tuple(L+1), attentions tuple(L) [, out_fullatts])."""
    B = image.shape[0]
    d = cfg["hidden"]
    pe = F.conv2d(image, sd[p + "patch_embed.weight"], None, stride=cfg["patch"])        # :444
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd[p + "class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, pe], dim=1) + sd[p + "pos_embed.weight"][None]                   # :447-449
    x = F.layer_norm(x, (d,), sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], cfg["vit_eps"])
    L, lad = cfg["vit_layers"], cfg["local_attn_depth"]
    do_gather = idx_to_group_img is not None
    blk = None
    if do_gather and image_atts is not None:                                             # :325-333
        full = torch.ones(x.shape[:2], dtype=x.dtype)
        blk = torch.cat([image_atts.to(x.dtype), full], dim=0)[:, None, None, :]
        blk = (1.0 - blk) * -10000.0
        blk = blk.expand(-1, -1, blk.size(-1), -1)
    hs, atts = [], []
    for i in range(L):
        hs.append(x)                                                                     # :351-352
        lp = f"{p}encoder.layers.{i}."
        hz = head_z[i] if head_z is not None else None
        hlz = head_layer_z[i] if head_layer_z is not None else None
        mz = mlp_z[i] if mlp_z is not None else None
        if lad > 0 and i >= L - lad:
            if do_gather:                                                                # :354-357
                do_gather = False
                x = torch.cat([x[idx_to_group_img], x], dim=0)
            x, pr = vit_layer(sd, lp, x, cfg, blk, hz, hlz, mz)
        else:
            x, pr = vit_layer(sd, lp, x, cfg, None, hz, hlz, mz)
        atts.append(pr)
    hs.append(x)                                                                         # :377-378
    out = F.layer_norm(x, (d,), sd[p + "post_layernorm.weight"], sd[p + "post_layernorm.bias"], cfg["vit_eps"])
    if idx_to_group_img is not None:
        bs = len(idx_to_group_img)
        return out[:bs], tuple(hs), tuple(atts), out[bs:]
    return out, tuple(hs), tuple(atts)


# ---------------------------------------------------------------------------------------------
# BERT text / fusion encoder  (efficient_models/eff_bert.py == models/xbert.py with z=None)
# ---------------------------------------------------------------------------------------------
# Dropout (eff_bert.py:214,346,379,460).  None = the p = 0 configuration of the fixtures.  Tests of the p > 0 path set this
# to an iterator of keep / (1 - p) masks, one per dropout site in the order the reference module tree visits them
# (embeddings; per layer: self-attention probabilities, self-output, [cross-attention probabilities, cross-output],
# FFN output): the reference's CUDA RNG stream cannot be reproduced, the SAME mask on both sides can.
DROPOUT_MASKS = None


def _drop(x):
    if DROPOUT_MASKS is None:
        return x
    m = next(DROPOUT_MASKS)
    assert m.numel() == x.numel(), f"dropout site order mismatch: mask {tuple(m.shape)} vs {tuple(x.shape)}"
    return x * m.view(x.shape)


def bert_embeddings(sd, p, ids, eps):
    """BertEmbeddings.forward, eff_bert.py:188-215 (token_type 0, absolute positions; dropout :214 through _drop)"""
    L = ids.shape[1]
    # nn.Embedding(..., padding_idx=pad_token_id=0), eff_bert.py:171: the pad row gets no lookup gradient
    e = F.embedding(ids, sd[p + "word_embeddings.weight"], padding_idx=0) + sd[p + "token_type_embeddings.weight"][0]
    e = e + sd[p + "position_embeddings.weight"][:L][None]
    return _drop(F.layer_norm(e, (e.shape[-1],), sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"], eps))


def bert_attention(sd, p, x, mask, heads, eps, enc=None, enc_mask=None, head_z=None):
    """BertAttention = BertSelfAttention (eff_bert.py:266-364) + BertSelfOutput (:374-381).

    probs are returned BEFORE dropout (:338-361); context *= head_z (:354-355); head_layer_z is dead
    plumbing."""
    B, Lq, d = x.shape
    dh = d // heads
    src = x if enc is None else enc
    m = mask if enc is None else enc_mask
    q = F.linear(x, sd[p + "self.query.weight"], sd[p + "self.query.bias"])
    k = F.linear(src, sd[p + "self.key.weight"], sd[p + "self.key.bias"])
    v = F.linear(src, sd[p + "self.value.weight"], sd[p + "self.value.bias"])
    sh = lambda t: t.view(t.shape[0], t.shape[1], heads, dh).permute(0, 2, 1, 3)
    s = sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(dh)                                  # :317,:330-331
    if m is not None:
        s = s + m                                                                        # :335
    probs = torch.softmax(s, dim=-1)
    ctx = _drop(probs) @ sh(v)                                                           # :346-352 (probs returned un-dropped)
    if head_z is not None:
        ctx = ctx * head_z
    ctx = ctx.permute(0, 2, 1, 3).reshape(B, Lq, d)
    o = _drop(F.linear(ctx, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]))   # :375-379
    o = F.layer_norm(o + x, (d,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return o, probs


def bert_layer(sd, p, x, mask, cfg, has_cross, enc=None, enc_mask=None, head_z=None, mlp_z=None):
    """BertLayer.forward + feed_forward_chunk, eff_bert.py:480-560 — gate AFTER erf-GELU (:553-557)"""
    if has_cross and head_z is not None:
        head_z, cross_head_z = head_z                                                    # :493-497
    else:
        cross_head_z = None
    eps = cfg["bert_eps"]
    a, probs = bert_attention(sd, p + "attention.", x, mask, cfg["heads"], eps, head_z=head_z)
    cprobs = None
    if has_cross:
        a, cprobs = bert_attention(sd, p + "crossattention.", a, mask, cfg["heads"], eps,
                                   enc=enc, enc_mask=enc_mask, head_z=cross_head_z)
    h = F.gelu(F.linear(a, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    if mlp_z is not None:
        h = h * mlp_z
    o = _drop(F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]))   # :459-460
    o = F.layer_norm(o + a, (o.shape[-1],), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return o, probs, cprobs


def bert_encoder(sd, p, x, mask, cfg, mode, enc=None, enc_mask=None, head_z=None, mlp_z=None):
    """BertEncoder.forward, eff_bert.py:570-694 (gate indexing :612-620 reproduced verbatim,
    including the offset-free multi_modal quirk of SURVEY.md §3.3)."""
    F_, NL = cfg["fusion_layer"], cfg["text_layers"]
    lo, hi = {"text": (0, F_), "fusion": (F_, NL), "multi_modal": (0, NL)}[mode]
    hs, atts, catts = [], [], []
    for i in range(lo, hi):
        hs.append(x)
        if i >= F_ and head_z is not None:
            first = (i - F_) * 2
            cur_hz = (head_z[first], head_z[first + 1])
            cur_mz = mlp_z[i - F_]
        elif head_z is not None:
            cur_hz, cur_mz = head_z[i], mlp_z[i]
        else:
            cur_hz, cur_mz = None, None
        x, pr, cpr = bert_layer(sd, f"{p}layer.{i}.", x, mask, cfg, i >= F_, enc, enc_mask, cur_hz, cur_mz)
        atts.append(pr)
        if cpr is not None:
            catts.append(cpr)
    hs.append(x)
    return x, tuple(hs), tuple(atts), tuple(catts)


def ext_mask(m):
    """get_extended_attention_mask (eff_bert.py:953-1013) / HF-4.12.5 invert_attention_mask:
    additive (1-m)*-10000 broadcast as [B,1,1,L]"""
    return (1.0 - m[:, None, None, :].to(torch.float32)) * -10000.0


def causal_ext_mask(m):
    """get_extended_attention_mask, is_decoder branch (eff_bert.py:975-1012): (1 - causal * padding) * -10000, [B,1,L,L]"""
    L = m.shape[1]
    ids = torch.arange(L)
    causal = (ids[None, None, :] <= ids[None, :, None]).to(torch.float32)                # key <= query
    return (1.0 - causal[:, None, :, :] * m[:, None, None, :].to(torch.float32)) * -10000.0


def bert_model(sd, p, cfg, input_ids=None, attention_mask=None, encoder_embeds=None,
               encoder_hidden_states=None, encoder_attention_mask=None, mode="multi_modal",
               head_z=None, mlp_z=None, is_decoder=False):
    """BertModel.forward, eff_bert.py:1015-1162"""
    x = bert_embeddings(sd, p + "embeddings.", input_ids, cfg["bert_eps"]) if encoder_embeds is None else encoder_embeds
    mask = causal_ext_mask(attention_mask) if is_decoder else ext_mask(attention_mask)
    enc_mask = None
    if encoder_hidden_states is not None:
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2])
        enc_mask = ext_mask(encoder_attention_mask)
    return bert_encoder(sd, p + "encoder.", x, mask, cfg, mode, encoder_hidden_states, enc_mask, head_z, mlp_z)


def mlm_head(sd, p, x, eps):
    """BertLMPredictionHead, eff_bert.py:712-746 (decoder weight tied to word embeddings)"""
    h = F.gelu(F.linear(x, sd[p + "transform.dense.weight"], sd[p + "transform.dense.bias"]))
    h = F.layer_norm(h, (h.shape[-1],), sd[p + "transform.LayerNorm.weight"], sd[p + "transform.LayerNorm.bias"], eps)
    return F.linear(h, sd[p + "decoder.weight"], sd[p + "bias"])


def bert_mlm(sd, p, cfg, ids_masked, atts, image_embeds, image_atts, masked_pos, labels, head_z=None, mlp_z=None):
    """BertForMaskedLM.forward, eff_bert.py:1634-1714: multi_modal encode, gather masked_pos, CE(-100)"""
    x, hs, at, cat = bert_model(sd, p + "bert.", cfg, input_ids=ids_masked, attention_mask=atts,
                                encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                                mode="multi_modal", head_z=head_z, mlp_z=mlp_z)
    g = torch.gather(x, 1, masked_pos.unsqueeze(2).expand(-1, -1, x.size(-1)))           # :1631-1632
    logits = mlm_head(sd, p + "cls.predictions.", g, cfg["bert_eps"])
    loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1), ignore_index=-100)
    return loss, logits, hs, at, cat


# ---------------------------------------------------------------------------------------------
# X-VLM base  (efficient_models/xvlm.py:211-569 == models/xvlm.py:280-612)
# ---------------------------------------------------------------------------------------------
def build_mlp_fwd(sd, p, x):
    """build_mlp, xvlm.py:77-83: Linear - LayerNorm(1e-5) - GELU - Linear"""
    h = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    h = F.gelu(F.layer_norm(h, (h.shape[-1],), sd[p + "1.weight"], sd[p + "1.bias"], 1e-5))
    return F.linear(h, sd[p + "3.weight"], sd[p + "3.bias"])


def get_features(sd, image_embeds, text_embeds):
    """XVLMBase.get_features, xvlm.py:375-382"""
    i = F.normalize(F.linear(image_embeds[:, 0, :], sd["vision_proj.weight"], sd["vision_proj.bias"]), dim=-1)
    t = F.normalize(F.linear(text_embeds[:, 0, :], sd["text_proj.weight"], sd["text_proj.bias"]), dim=-1)
    return i, t


def contrastive_loss(image_feat_all, text_feat_all, temp, idx_all=None):
    """XVLMBase.get_contrastive_loss, xvlm.py:384-416, on the ALREADY all-gathered features."""
    logits = image_feat_all @ text_feat_all.t() / temp
    n = logits.shape[0]
    if idx_all is None:
        labels = torch.arange(n)
        return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2
    idx_all = idx_all.view(-1, 1)
    pos = torch.eq(idx_all, idx_all.t()).float()
    labels = pos / pos.sum(1, keepdim=True)
    l1 = -torch.sum(F.log_softmax(logits, dim=1) * labels, dim=1).mean()
    l2 = -torch.sum(F.log_softmax(logits.t(), dim=1) * labels, dim=1).mean()
    return (l1 + l2) / 2


def negative_weights(image_feat, text_feat, temp, idx=None):
    """the no-grad sampling weights of get_matching_loss, xvlm.py:422-438"""
    with torch.no_grad():
        w_i2t = F.softmax(image_feat @ text_feat.t() / temp, dim=1) + 1e-5
        w_t2i = F.softmax(text_feat @ image_feat.t() / temp, dim=1) + 1e-5
        if idx is None:
            w_i2t.fill_diagonal_(0)
            w_t2i.fill_diagonal_(0)
        else:
            m = torch.eq(idx.view(-1, 1), idx.view(1, -1))
            w_i2t.masked_fill_(m, 0)
            w_t2i.masked_fill_(m, 0)
    return w_i2t, w_t2i


def matching_loss(sd, cfg, image_embeds, image_atts, text_embeds, text_atts, neg_idx, head_z=None, mlp_z=None):
    """XVLMBase.get_matching_loss, xvlm.py:418-490 with the hard-negative indices INJECTED
    (neg_idx = [B image negatives drawn from weights_t2i rows, then B text negatives from weights_i2t])."""
    bs = image_embeds.shape[0]
    img_neg, txt_neg = neg_idx[:bs], neg_idx[bs:]
    te_all = torch.cat([text_embeds, text_embeds[txt_neg]], dim=0)
    ta_all = torch.cat([text_atts, text_atts[txt_neg]], dim=0)
    ie_all = torch.cat([image_embeds[img_neg], image_embeds], dim=0)
    ia_all = torch.cat([image_atts[img_neg], image_atts], dim=0)
    enc_p = _bert_prefix(sd)
    pos = bert_model(sd, enc_p, cfg, encoder_embeds=text_embeds, attention_mask=text_atts,
                     encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts, mode="fusion",
                     head_z=head_z, mlp_z=mlp_z)
    neg = bert_model(sd, enc_p, cfg, encoder_embeds=te_all, attention_mask=ta_all,
                     encoder_hidden_states=ie_all, encoder_attention_mask=ia_all, mode="fusion",
                     head_z=head_z, mlp_z=mlp_z)
    out = build_mlp_fwd(sd, "itm_head.", torch.cat([pos[0][:, 0, :], neg[0][:, 0, :]], dim=0))
    labels = torch.cat([torch.ones(bs, dtype=torch.long), torch.zeros(2 * bs, dtype=torch.long)])
    return dict(loss=F.cross_entropy(out, labels), pos_hidden_states=pos[1], neg_hidden_states=neg[1],
                pos_attentions=pos[2], neg_attentions=neg[2], pos_cross_attentions=pos[3],
                neg_cross_attentions=neg[3], logits=out)


def _bert_prefix(sd):
    # BertForMaskedLM nests the encoder under ".bert." (xvlm.py:306); fine-tune models use BertModel directly
    return "text_encoder.bert." if any(k.startswith("text_encoder.bert.") for k in sd) else "text_encoder."


def box_cxcywh_to_xyxy(x):
    """models/box_ops.py:8-12"""
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def giou_rowwise(b1, b2):
    """diagonal of models/box_ops.py:41-56 generalized_box_iou(b1, b2) (the only part xvlm.py:559 reads), xyxy boxes"""
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, 2:], b2[:, 2:]) - torch.max(b1[:, :2], b2[:, :2])).clamp(min=0)      # :26-29
    inter = wh[:, 0] * wh[:, 1]
    union = area1 + area2 - inter
    iou = inter / union
    wh_c = (torch.max(b1[:, 2:], b2[:, 2:]) - torch.min(b1[:, :2], b2[:, :2])).clamp(min=0)    # :50-53
    area_c = wh_c[:, 0] * wh_c[:, 1]
    return iou - (area_c - union) / area_c


def bbox_loss(output_coord, target_bbox, is_image=None):
    """XVLMBase.get_bbox_loss, xvlm.py:544-569: L1 + (1 - GIoU), rows with is_image=1 masked out of both sums"""
    l1 = (output_coord - target_bbox).abs()
    b1, b2 = box_cxcywh_to_xyxy(output_coord), box_cxcywh_to_xyxy(target_bbox)
    if (b1[:, 2:] < b1[:, :2]).any() or (b2[:, 2:] < b2[:, :2]).any():                         # :553-556
        giou = torch.zeros(output_coord.size(0))
    else:
        giou = 1 - giou_rowwise(b1, b2)
    if is_image is None:
        n = target_bbox.size(0)
    else:
        n = torch.sum(1 - is_image)
        l1 = l1 * (1 - is_image.view(-1, 1))
        giou = giou * (1 - is_image)
    return l1.sum() / n, giou.sum() / n


def pretrain_forward(sd, cfg, batch, neg_idx):
    """models/model_pretrain.py:11-82 XVLM.forward.  A batch that carries `idx_to_group_img` is a REGION batch
    (ret_bbox_loss=True, GeneralDistill.py:176-178): the ViT splits into per-region masked copies for its last
    `local_attn_depth` layers, ITC / ITM / MLM run on the region embeddings with the region patch masks, and the bbox
    head regresses the box from a fusion pass over the FULL-attention image embeddings."""
    region = "idx_to_group_img" in batch
    if region:
        image_embeds, image_hs, image_at, full = vit_forward(
            sd, "vision_encoder.", batch["image"], cfg, idx_to_group_img=batch["idx_to_group_img"],
            image_atts=batch["image_atts"])
        image_atts = batch["image_atts"]
        image_embeds_fullatts = full[batch["idx_to_group_img"]]                          # xvlm.py:294-297
    else:
        image_embeds, image_hs, image_at = vit_forward(sd, "vision_encoder.", batch["image"], cfg)
        image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long)
    bp = _bert_prefix(sd)
    text_embeds, text_hs, text_at, _ = bert_model(sd, bp, cfg, input_ids=batch["text_ids"],
                                                  attention_mask=batch["text_atts"], mode="text")
    temp = sd["temp"].clamp(0.001, 0.5)                                                  # :35-36
    i_feat, t_feat = get_features(sd, image_embeds, text_embeds)
    loss_itc = contrastive_loss(i_feat, t_feat, temp)
    itm = matching_loss(sd, cfg, image_embeds, image_atts, text_embeds, batch["text_atts"], neg_idx)
    mlm = bert_mlm(sd, "text_encoder.", cfg, batch["text_ids_masked"], batch["text_atts"], image_embeds,
                   image_atts, batch["masked_pos"], batch["masked_ids"])
    loss = {"loss_itc": loss_itc, "loss_itm": itm["loss"], "loss_mlm": mlm[0]}
    extra_h, extra_a, extra_c = {}, {}, {}
    if region:                                                                           # model_pretrain.py:62-74
        bb = bert_model(sd, bp, cfg, encoder_embeds=text_embeds, attention_mask=batch["text_atts"],
                        encoder_hidden_states=image_embeds_fullatts,
                        encoder_attention_mask=torch.ones(image_embeds_fullatts.shape[:2]), mode="fusion")
        coord = torch.sigmoid(build_mlp_fwd(sd, "bbox_head.", bb[0][:, 0, :]))           # xvlm.py:540
        loss["loss_bbox"], loss["loss_giou"] = bbox_loss(coord, batch["target_bbox"], batch.get("is_image"))
        extra_h, extra_a, extra_c = ({"bbox_hidden_states": bb[1]}, {"bbox_attentions": bb[2]},
                                     {"bbox_cross_attentions": bb[3]})
    return {
        "loss": loss, "output_coord": coord if region else None,
        "hidden_dict": {"image_hidden_states": image_hs, "text_hidden_states": text_hs,
                        "itm_pos_hidden_states": itm["pos_hidden_states"],
                        "itm_neg_hidden_states": itm["neg_hidden_states"], "mlm_hidden_states": mlm[2], **extra_h},
        "attention_dict": {"image_attentions": image_at, "text_attentions": text_at,
                           "itm_pos_attentions": itm["pos_attentions"],
                           "itm_neg_attentions": itm["neg_attentions"], "mlm_attentions": mlm[3], **extra_a},
        "cross_attention_dict": {"itm_pos_cross_attentions": itm["pos_cross_attentions"],
                                 "itm_neg_cross_attentions": itm["neg_cross_attentions"],
                                 "mlm_cross_attentions": mlm[4], **extra_c},
        "logits_dict": {"itm_head_logits": itm["logits"], "mlm_logits": mlm[1]},
        "features": (i_feat, t_feat),
    }


def retrieval_forward(sd, cfg, batch, idx, neg_idx, zs=None):
    """efficient_models/model_retrieval.py:25-93 (zs given) / models/model_retrieval.py:20-67 (zs None)."""
    z = zs or {}
    image_embeds, image_hs, image_at = vit_forward(sd, "vision_encoder.", batch["image"], cfg,
                                                   head_z=z.get("vision_head_z"), mlp_z=z.get("vision_intermediate_z"))
    image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long)
    bp = _bert_prefix(sd)
    text_embeds, text_hs, text_at, _ = bert_model(sd, bp, cfg, input_ids=batch["text_ids"],
                                                  attention_mask=batch["text_atts"], mode="text",
                                                  head_z=z.get("text_head_z"), mlp_z=z.get("text_intermediate_z"))
    i_feat, t_feat = get_features(sd, image_embeds, text_embeds)
    loss_itc = contrastive_loss(i_feat, t_feat, sd["temp"], idx)
    itm = matching_loss(sd, cfg, image_embeds, image_atts, text_embeds, batch["text_atts"], neg_idx,
                        head_z=z.get("cross_head_z"), mlp_z=z.get("cross_intermediate_z"))
    return {
        "loss": {"loss_itc": loss_itc, "loss_itm": itm["loss"]},
        "hidden_dict": {"image_hidden_states": image_hs, "text_hidden_states": text_hs,
                        "itm_pos_hidden_states": itm["pos_hidden_states"],
                        "itm_neg_hidden_states": itm["neg_hidden_states"]},
        "attention_dict": {"image_attentions": image_at, "text_attentions": text_at,
                           "itm_pos_attentions": itm["pos_attentions"], "itm_neg_attentions": itm["neg_attentions"]},
        "cross_attention_dict": {"itm_pos_cross_attentions": itm["pos_cross_attentions"],
                                 "itm_neg_cross_attentions": itm["neg_cross_attentions"]},
        "logits_dict": {"itm_head_logits": itm["logits"]},
        "features": (i_feat, t_feat),
    }


# ---------------------------------------------------------------------------------------------
# KD losses  (GeneralDistill.py:60-104, loss mixes :369-376 and Eff_Retrieval.py:165-178)
# ---------------------------------------------------------------------------------------------
def get_cor_teacher(teacher_reps, student_reps, is_attn=False):
    """GeneralDistill.py:91-104"""
    t = [r.detach() for r in teacher_reps]
    nt, ns = len(t), len(student_reps)
    if is_attn:
        assert nt % ns == 0
        k = nt // ns
        return [t[i * k + k - 1] for i in range(ns)]
    assert (nt - 1) % (ns - 1) == 0
    k = (nt - 1) // (ns - 1)
    return [t[i * k] for i in range(ns)]


def get_kd_loss(student_reps, teacher_reps, is_attn=False, is_img=False):
    """GeneralDistill.py:60-82: sum over layers of MSE(mean); attention maps x last-dim size; the
    image branch skips list index 6.  (The torch.where(att <= -1e2, 0, att) is kept for parity.)"""
    total = 0
    for layer, (s, t) in enumerate(zip(student_reps, teacher_reps)):
        if is_attn:
            s = torch.where(s <= -1e2, torch.zeros_like(s), s)
            t = torch.where(t <= -1e2, torch.zeros_like(t), t)
            total = total + F.mse_loss(s, t) * s.shape[-1]
        elif is_img and layer == 6:
            continue
        else:
            total = total + F.mse_loss(s, t)
    return total


def soft_cross_entropy(predicts, targets):
    """GeneralDistill.py:84-89: KLDiv(log_softmax(s), softmax(t), 'batchmean') over flattened rows"""
    C = predicts.shape[-1]
    return F.kl_div(F.log_softmax(predicts, dim=-1).view(-1, C), F.softmax(targets, dim=-1).view(-1, C),
                    reduction="batchmean")


def kd_terms(S, T, temperature=1.0, with_cross_attn=False):
    sh, th, sa, ta = S["hidden_dict"], T["hidden_dict"], S["attention_dict"], T["attention_dict"]
    out = {}

    def pair(name, hkey, akey, is_img=False):
        out[name + "_hidden"] = get_kd_loss(sh[hkey], get_cor_teacher(th[hkey], sh[hkey]), is_img=is_img)
        out[name + "_attn"] = get_kd_loss(sa[akey], get_cor_teacher(ta[akey], sa[akey], True), is_attn=True)

    pair("text", "text_hidden_states", "text_attentions")
    pair("image", "image_hidden_states", "image_attentions", is_img=True)
    pair("itm_pos", "itm_pos_hidden_states", "itm_pos_attentions")
    pair("itm_neg", "itm_neg_hidden_states", "itm_neg_attentions")
    if "mlm_hidden_states" in sh:
        pair("mlm", "mlm_hidden_states", "mlm_attentions")
        out["mlm_logits"] = soft_cross_entropy(S["logits_dict"]["mlm_logits"] / temperature,
                                               T["logits_dict"]["mlm_logits"] / temperature)
    out["itm_logits"] = soft_cross_entropy(S["logits_dict"]["itm_head_logits"] / temperature,
                                           T["logits_dict"]["itm_head_logits"] / temperature)
    if with_cross_attn:
        sc, tc = S["cross_attention_dict"], T["cross_attention_dict"]
        for nm in ("itm_pos", "itm_neg"):
            k = nm + "_cross_attentions"
            out[nm + "_cross"] = get_kd_loss(sc[k], get_cor_teacher(tc[k], sc[k], True), is_attn=True)
    return out


def gd_loss_mix(loss, kd):
    """GeneralDistill.py:369-376 (general batch) / :252-260 (region batch: + bbox + giou in the task term)"""
    loss_small = loss["loss_itc"] + loss["loss_itm"] + loss["loss_mlm"]
    if "loss_bbox" in loss:
        loss_small = loss_small + loss["loss_bbox"] + loss["loss_giou"]
    loss_text_kd = kd["text_attn"] + kd["text_hidden"]
    loss_img_kd = kd["image_attn"] + 0.1 * kd["image_hidden"]
    loss_cross_kd = (kd["itm_neg_attn"] + kd["itm_neg_hidden"] + kd["itm_pos_attn"] + kd["itm_pos_hidden"]
                     + kd["mlm_attn"] + kd["mlm_hidden"])
    loss_kd = kd["itm_logits"] + kd["mlm_logits"] + loss_text_kd + loss_img_kd + loss_cross_kd
    return loss_small * 0.6 + loss_kd * 0.4, dict(loss_small=loss_small, loss_text_kd=loss_text_kd,
                                                  loss_img_kd=loss_img_kd, loss_cross_kd=loss_cross_kd,
                                                  loss_kd=loss_kd)


def itr_loss_mix(loss, kd, lagrangian):
    """Eff_Retrieval.py:165-178"""
    loss_text_kd = kd["text_hidden"] + kd["text_attn"]
    loss_img_kd = 0.2 * kd["image_hidden"] + kd["image_attn"]
    loss_cross_kd = (kd["itm_neg_hidden"] + kd["itm_pos_hidden"] + kd["itm_pos_attn"] + kd["itm_pos_cross"]
                     + kd["itm_neg_attn"] + kd["itm_neg_cross"]) * 0.5
    loss_kd = kd["itm_logits"] + (loss_text_kd + loss_img_kd + loss_cross_kd) * 0.33
    loss_small = loss["loss_itc"] + loss["loss_itm"]
    return (loss_kd + loss_small) * 0.5 + lagrangian, dict(loss_text_kd=loss_text_kd, loss_img_kd=loss_img_kd,
                                                           loss_cross_kd=loss_cross_kd, loss_kd=loss_kd)


def gd_step(s_sd, t_sd, s_cfg, t_cfg, batch, s_neg, t_neg, temperature=1.0):
    """one GeneralDistill.py general step (forward of both models + every loss); the caller backprops."""
    S = pretrain_forward(s_sd, s_cfg, batch, s_neg)
    with torch.no_grad():
        T = pretrain_forward(t_sd, t_cfg, batch, t_neg)
    kd = kd_terms(S, T, temperature)
    total, mix = gd_loss_mix(S["loss"], kd)
    return total, S, T, kd, mix


# ---------------------------------------------------------------------------------------------
# VQA fine-tune  (efficient_models/model_generation.py:97-187, models/model_generation.py:301-377, Eff_VQA.py:95-176)
# ---------------------------------------------------------------------------------------------
def lm_head_decoder(sd, p, cfg, ids, atts, enc, enc_atts, labels, head_z=None, mlp_z=None):
    """BertLMHeadModel.forward, eff_bert.py:1332-1443 with reduction='none', label_smoothing 0: causal decoder over the
    answers, every layer cross-attending to `enc`; next-token CE summed per sequence (-100 labels contribute 0)."""
    x, hs, at, cat = bert_model(sd, p + "bert.", cfg, input_ids=ids, attention_mask=atts, encoder_hidden_states=enc,
                                encoder_attention_mask=enc_atts, mode="multi_modal", head_z=head_z, mlp_z=mlp_z,
                                is_decoder=True)
    logits = mlm_head(sd, p + "cls.predictions.", x, cfg["bert_eps"])
    shifted, lab = logits[:, :-1, :], labels[:, 1:]                                      # :1419-1421
    ce = F.cross_entropy(shifted.reshape(-1, shifted.shape[-1]), lab.reshape(-1), reduction="none", ignore_index=-100)
    return ce.view(logits.shape[0], -1).sum(1), logits, hs, at, cat


def vqa_forward(sd, cfg, batch, zs=None):
    """EffXVLMForVQA.forward train branch (zs given; model_generation.py:97-187) / XVLMForVQA.forward (zs None;
    models/model_generation.py:301-377).  batch: image, question_ids/atts [B, Lq], answer_ids/atts [sum k, La], k [B],
    weights [sum k]; pad_token_id 0."""
    z = lambda key: zs[key] if zs is not None else None
    image_embeds, image_hs, image_at = vit_forward(sd, "vision_encoder.", batch["image"], cfg,
                                                   head_z=z("vision_head_z"), mlp_z=z("vision_intermediate_z"))
    image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long)
    enc_hz = torch.cat((zs["text_head_z"], zs["cross_head_z"]), dim=0) if zs is not None else None       # :124
    enc_mz = torch.cat((zs["text_intermediate_z"], zs["cross_intermediate_z"]), dim=0) if zs is not None else None
    q_last, q_hs, q_at, q_cat = bert_model(sd, "text_encoder.", cfg, input_ids=batch["question_ids"],
                                           attention_mask=batch["question_atts"], encoder_hidden_states=image_embeds,
                                           encoder_attention_mask=image_atts, mode="multi_modal", head_z=enc_hz, mlp_z=enc_mz)
    rep = torch.repeat_interleave(torch.arange(q_last.shape[0]), batch["k"])              # :126-131
    targets = batch["answer_ids"].masked_fill(batch["answer_ids"] == 0, -100)             # :113
    nd = cfg["text_layers"] - cfg["fusion_layer"]
    dec_cfg = dict(cfg, text_layers=nd, fusion_layer=0)                                   # :36-40
    per_seq, logits, d_hs, d_at, d_cat = lm_head_decoder(sd, "text_decoder.", dec_cfg, batch["answer_ids"],
                                                         batch["answer_atts"], q_last[rep], batch["question_atts"][rep],
                                                         targets, z("decoder_head_z"), z("decoder_intermediate_z"))
    loss = (batch["weights"] * per_seq).sum() / batch["image"].shape[0]                   # :166-167
    return {"loss": loss,
            "hidden_dict": {"image_hidden_states": image_hs, "text_hidden_states": q_hs, "decoder_hidden_states": d_hs},
            "attention_dict": {"image_attentions": image_at, "text_attentions": q_at, "decoder_attentions": d_at},
            "cross_attention_dict": {"cross_attentions": q_cat, "decoder_cross_attentions": d_cat},
            "logits_dict": {"logits": logits}}


def vqa_kd_terms(S, T, temperature=1.0):
    """Eff_VQA.py:113-163.  The split of the question encoder's lists at state 4 / map 3 is HARD-CODED there for the
    (3 text + 3 fusion)-layer student; the decoder-hidden term passes is_img=True (skip of list index 6: a no-op on the
    student's 4 states)."""
    sh, th, sa, ta = S["hidden_dict"], T["hidden_dict"], S["attention_dict"], T["attention_dict"]
    sc, tc = S["cross_attention_dict"], T["cross_attention_dict"]
    s_h, s_a = sh["text_hidden_states"], sa["text_attentions"]
    t_h, t_a = get_cor_teacher(th["text_hidden_states"], s_h), get_cor_teacher(ta["text_attentions"], s_a, True)
    kd = {"text_hidden": get_kd_loss(s_h[:4], t_h[:4]), "text_attn": get_kd_loss(s_a[:3], t_a[:3], is_attn=True),
          "cross_hidden": get_kd_loss(s_h[4:], t_h[4:]), "cross_self_attn": get_kd_loss(s_a[3:], t_a[3:], is_attn=True),
          "cross_attn": get_kd_loss(sc["cross_attentions"], get_cor_teacher(tc["cross_attentions"], sc["cross_attentions"], True),
                                    is_attn=True)}
    kd["image_hidden"] = get_kd_loss(sh["image_hidden_states"], get_cor_teacher(th["image_hidden_states"], sh["image_hidden_states"]),
                                     is_img=True)
    kd["image_attn"] = get_kd_loss(sa["image_attentions"], get_cor_teacher(ta["image_attentions"], sa["image_attentions"], True),
                                   is_attn=True)
    kd["decoder_hidden"] = get_kd_loss(sh["decoder_hidden_states"],
                                       get_cor_teacher(th["decoder_hidden_states"], sh["decoder_hidden_states"]), is_img=True)
    kd["decoder_attn"] = get_kd_loss(sa["decoder_attentions"],
                                     get_cor_teacher(ta["decoder_attentions"], sa["decoder_attentions"], True), is_attn=True)
    kd["decoder_cross"] = get_kd_loss(sc["decoder_cross_attentions"],
                                      get_cor_teacher(tc["decoder_cross_attentions"], sc["decoder_cross_attentions"], True),
                                      is_attn=True)
    kd["logits"] = soft_cross_entropy(S["logits_dict"]["logits"] / temperature, T["logits_dict"]["logits"] / temperature)
    return kd


def vqa_loss_mix(loss_small, kd, lagrangian):
    """Eff_VQA.py:165-176"""
    loss_text_kd = kd["text_attn"] + kd["text_hidden"]
    loss_img_kd = kd["image_attn"] + kd["image_hidden"] * 0.2
    loss_cross_kd = (kd["cross_hidden"] + kd["cross_self_attn"] + kd["cross_attn"]) * 0.5
    loss_decoder_kd = kd["decoder_attn"] + kd["decoder_hidden"] + kd["decoder_cross"]
    loss_kd = kd["logits"] + loss_text_kd + loss_img_kd + loss_cross_kd + loss_decoder_kd
    return loss_kd * 0.4 + loss_small * 0.6 + lagrangian, dict(loss_text_kd=loss_text_kd, loss_img_kd=loss_img_kd,
                                                              loss_cross_kd=loss_cross_kd, loss_decoder_kd=loss_decoder_kd,
                                                              loss_kd=loss_kd)


# ---------------------------------------------------------------------------------------------
# hard-concrete L0 gates  (efficient_models/xvlm_l0_module.py)
# ---------------------------------------------------------------------------------------------
L0_TYPES = ("vision_head", "text_head", "cross_head", "vision_intermediate", "text_intermediate", "cross_intermediate")
L0_PARAM = {"vision_head": "vision_head_loga", "text_head": "text_head_loga", "cross_head": "cross_head_loga",
            "vision_intermediate": "vision_int_loga", "text_intermediate": "text_int_loga",
            "cross_intermediate": "cross_int_loga"}


# efficient_models/generation_l0_module.py (VQAL0Module): the same module + gates for the answer decoder, whose every
# layer has self- AND cross-attention (decoder_head_loga [2 * layers, heads], decoder_int_loga [layers, ffn])
L0_TYPES_VQA = ("vision_head", "text_head", "cross_head", "decoder_head", "vision_intermediate", "text_intermediate",
                "cross_intermediate", "decoder_intermediate")
L0_PARAM.update({"decoder_head": "decoder_head_loga", "decoder_intermediate": "decoder_int_loga"})


def l0_constants(hidden, ffn, heads, n_vit, n_text, n_cross, n_dec=0):
    """xvlm_l0_module.py:46-56,116-166 (generation_l0_module.py:47-57,119-173 with n_dec decoder layers)"""
    per_head_layer = hidden * hidden * 4 + hidden * 4
    per_head = per_head_layer // heads
    per_mlp_layer = hidden * ffn * 2 + hidden + hidden * 4
    per_int = per_mlp_layer // ffn
    prunable = (per_head * heads * (n_vit + n_text + 2 * n_cross + 2 * n_dec)
                + per_mlp_layer * (n_vit + n_text + n_cross + n_dec))
    return dict(params_per_head=per_head, params_per_int=per_int, prunable=prunable, hidden=hidden)


def cdf_qz0(loga, temperature=2. / 3.):
    """cdf_qz(0, loga), xvlm_l0_module.py:174-178"""
    xn = (0 - LIMIT_A) / (LIMIT_B - LIMIT_A)
    logits = math.log(xn) - math.log(1 - xn)
    return torch.sigmoid(logits * temperature - loga).clamp(min=EPSILON, max=1 - EPSILON)


def l0_sample_z(loga, eps, temperature=2. / 3.):
    """_sample_z + quantile_concrete, xvlm_l0_module.py:180-182,246-250"""
    y = torch.sigmoid((torch.log(eps) - torch.log(1 - eps) + loga) / temperature)
    return F.hardtanh(y * (LIMIT_B - LIMIT_A) + LIMIT_A, min_val=0, max_val=1)


def l0_deterministic_z(loga_row, temperature=2. / 3., magical_number=0.8):
    """_deterministic_z for ONE layer row, xvlm_l0_module.py:253-271"""
    size = loga_row.numel()
    expected_nonzeros = torch.sum(1 - cdf_qz0(loga_row, temperature))
    num_zeros = round(size - expected_nonzeros.item())
    soft = torch.sigmoid(loga_row / temperature * magical_number)
    if num_zeros > 0:
        _, ind = torch.topk(soft, k=num_zeros, largest=False)
        soft = torch.ones_like(soft)
        soft[ind] = 0.
        return soft
    return torch.ones_like(soft)


def _l0_types(logas):
    return L0_TYPES_VQA if "decoder_head_loga" in logas else L0_TYPES


def l0_shapes(logas):
    s = {}
    for t in _l0_types(logas):
        n, m = logas[L0_PARAM[t]].shape
        s[t] = [n, 1, m, 1, 1] if t.endswith("head") else [n, 1, 1, m]
    return s


def l0_forward(logas, training, eps=None, temperature=2. / 3., magical_number=0.8):
    """XVLML0Module.forward, xvlm_l0_module.py:321-341.  eps: dict type -> uniform draws (train)."""
    shapes = l0_shapes(logas)
    zs = {}
    for t in _l0_types(logas):
        la = logas[L0_PARAM[t]]
        if training:
            zs[t + "_z"] = l0_sample_z(la, eps[t], temperature).reshape(shapes[t])
        else:
            rows = [l0_deterministic_z(la[i], temperature, magical_number).reshape(shapes[t][1:]) for i in range(la.shape[0])]
            zs[t + "_z"] = torch.stack(rows)
    return zs


def l0_lagrangian(logas, lambda_1, lambda_2, consts, pruned_steps, target_sparsity, start_sparsity=0.0,
                  lagrangian_warmup=0, temperature=2. / 3.):
    """lagrangian_regularization, xvlm_l0_module.py:198-237"""
    n = 0
    for t in _l0_types(logas):
        per = consts["params_per_head"] if t.endswith("head") else consts["params_per_int"]
        n = n + torch.sum(1 - cdf_qz0(logas[L0_PARAM[t]], temperature)) * per
    expected_sparsity = 1 - n / consts["prunable"]
    tgt = target_sparsity
    if lagrangian_warmup > 0:
        tgt = (target_sparsity - start_sparsity) * min(1, pruned_steps / lagrangian_warmup) + start_sparsity
    loss = lambda_1 * (expected_sparsity - tgt) + lambda_2 * (expected_sparsity - tgt) ** 2
    return loss, expected_sparsity, tgt


def l0_model_size(zs, consts, heads, ffn):
    """calculate_model_size, xvlm_l0_module.py:284-319"""
    cnt = {}
    for t in L0_TYPES:
        z = zs[t + "_z"]
        m = heads if t.endswith("head") else ffn
        cnt[t] = (z.reshape(-1, m) > 0).sum(-1).tolist()
    head_n = sum(cnt["cross_head"]) + sum(cnt["text_head"]) + sum(cnt["vision_head"])
    int_n = sum(cnt["vision_intermediate"]) + sum(cnt["text_intermediate"]) + sum(cnt["cross_intermediate"])
    remaining = head_n * consts["params_per_head"] + int_n * 2 * consts["hidden"]
    pruned = consts["prunable"] - remaining
    return {"vision_intermediate_dims": cnt["vision_intermediate"], "text_intermediate_dims": cnt["text_intermediate"],
            "cross_intermediate_dims": cnt["cross_intermediate"], "vision_head_nums": cnt["vision_head"],
            "text_head_nums": cnt["text_head"], "cross_head_nums": cnt["cross_head"],
            "pruned_params": pruned, "remaining_params": remaining,
            "pruned_model_sparsity": pruned / consts["prunable"]}
