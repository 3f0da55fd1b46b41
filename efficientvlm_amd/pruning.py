"""Physically pruned models: turn the deterministic 0/1 gates of the L0 module into removed attention
heads and narrowed FFN GEMMs - the counterparts of the reference's utils/xvlm_utils.py:37-145 (`update_params`,
`prune_model_with_z`) and :228-244 (`prune_intermediate_layers`, `prune_vision_intermediate_layers`), same names and
argument meaning.

MI355X-specific: the kept FFN width is rounded UP to a multiple of `pad_to` (64 = the K tile of the bf16 GEMMs) with
zero rows in the up-projection and zero columns in the down-projection.  GELU / quick-GELU map 0 to 0, so the padded
model computes exactly the function of the reference's unpadded one while every GEMM stays on the MFMA fast path
(`pad_to=8` is the minimum the kernels accept: bf16 rows must be 16-byte multiples).

A pruned model carries no gates: call it with head_z / mlp_z = None (e.g. `retrieval_eval_losses` below).
"""
import torch
from torch import nn

from .efficient_models.eff_vit import prune_linear_layer


def _z(zs, key, layer):
    return zs[key][layer].detach().cpu().reshape(-1).clone()


def update_params(model, zs, cross_layers=3):
    """utils/xvlm_utils.py:37-84: fold the gate VALUES into the weights they multiply (value projection rows per head,
    down-projection columns per FFN unit), so that a model run without gates reproduces the gated one."""
    text, vision = model.text_encoder, model.vision_encoder
    core = text.bert if hasattr(text, "bert") else text
    vision_layers, text_layers = len(vision.encoder.layers), core.config.fusion_layer

    def scale_value(att, hz):
        hz = torch.repeat_interleave(hz, att.weight.shape[0] // hz.numel()).to(att.weight.device)   # (reference: 64)
        with torch.no_grad():
            att.weight.mul_(hz[:, None])
            att.bias.mul_(hz)

    def scale_down(lin, iz):
        with torch.no_grad():
            lin.weight.mul_(iz.to(lin.weight.device)[None, :])

    with torch.no_grad():
        for layer in range(text_layers):
            if "text_intermediate_z" in zs:
                scale_down(core.encoder.layer[layer].output.dense, _z(zs, "text_intermediate_z", layer))
            if "text_head_z" in zs:
                scale_value(core.encoder.layer[layer].attention.self.value, _z(zs, "text_head_z", layer))
        for layer in range(vision_layers):
            if "vision_intermediate_z" in zs:
                scale_down(vision.encoder.layers[layer].mlp.fc2, _z(zs, "vision_intermediate_z", layer))
            if "vision_head_z" in zs:
                scale_value(vision.encoder.layers[layer].self_attn.v_proj, _z(zs, "vision_head_z", layer))
        for layer in range(cross_layers):
            blk = core.encoder.layer[text_layers + layer]
            if "cross_intermediate_z" in zs:
                scale_down(blk.output.dense, _z(zs, "cross_intermediate_z", layer))
            if "cross_head_z" in zs:          # (self, cross) gates interleaved per fusion layer
                scale_value(blk.attention.self.value, _z(zs, "cross_head_z", 2 * layer))
                scale_value(blk.crossattention.self.value, _z(zs, "cross_head_z", 2 * layer + 1))


def _pad_ffn(up, down, pad_to):
    """append zero units so that the FFN width is a multiple of pad_to (act(0) = 0: the function is unchanged)"""
    n = up.weight.shape[0]
    m = (n + pad_to - 1) // pad_to * pad_to
    if m == n:
        return up, down
    dev, dt = up.weight.device, up.weight.dtype
    up2 = nn.Linear(up.weight.shape[1], m).to(device=dev, dtype=dt)
    down2 = nn.Linear(m, down.weight.shape[0]).to(device=dev, dtype=dt)
    with torch.no_grad():
        up2.weight.zero_(); up2.bias.zero_(); down2.weight.zero_()
        up2.weight[:n].copy_(up.weight); up2.bias[:n].copy_(up.bias)
        down2.weight[:, :n].copy_(down.weight); down2.bias.copy_(down.bias)
    return up2, down2


def prune_intermediate_layers(bert, keep_dims, device, pad_to=64):
    """utils/xvlm_utils.py:228-235 (BERT FFN: intermediate.dense rows / output.dense columns)"""
    for layer, keep in keep_dims.items():
        blk = bert.encoder.layer[layer]
        if len(keep) == 0:
            raise NotImplementedError("an FFN with no kept unit: the reference sets the modules to None and its forward "
                                      "cannot run such a layer either")
        idx = torch.as_tensor(keep, dtype=torch.long, device=device)
        up = prune_linear_layer(blk.intermediate.dense, idx, dim=0)
        down = prune_linear_layer(blk.output.dense, idx, dim=1)
        blk.intermediate.dense, blk.output.dense = _pad_ffn(up, down, pad_to)


def prune_vision_intermediate_layers(vision_encoder, keep_dims, device, pad_to=64):
    """utils/xvlm_utils.py:237-244 (CLIP MLP: fc1 rows / fc2 columns)"""
    for layer, keep in keep_dims.items():
        mlp = vision_encoder.encoder.layers[layer].mlp
        if len(keep) == 0:
            raise NotImplementedError("an MLP with no kept unit cannot be run by the reference forward either")
        idx = torch.as_tensor(keep, dtype=torch.long, device=device)
        up = prune_linear_layer(mlp.fc1, idx, dim=0)
        down = prune_linear_layer(mlp.fc2, idx, dim=1)
        mlp.fc1, mlp.fc2 = _pad_ffn(up, down, pad_to)


def prune_model_with_z(zs, model, cross_layers=3, pad_to=64, verbose=False):
    """utils/xvlm_utils.py:87-145: heads whose gate is 0 are removed from q/k/v (rows) and the output projection
    (columns); FFN units whose gate is 0 are removed from the up- (rows) and down-projection (columns)."""
    if zs is None:
        return None, None
    vision, text = model.vision_encoder, model.text_encoder
    core = text.bert if hasattr(text, "bert") else text
    device = next(core.parameters()).device

    def zero_heads(key):
        out = {}
        for layer in range(len(zs[key])):
            idx = torch.where(_z(zs, key, layer) == 0)[0].tolist()
            if len(idx) == zs[key][layer].numel():
                raise NotImplementedError("an attention block with no kept head cannot be run by the reference forward")
            out[layer] = idx
            if verbose:
                print(f"{key} layer {layer}: heads {idx} pruned")
        return out

    if "vision_head_z" in zs:
        vision.prune_heads(zero_heads("vision_head_z"))
    if "text_head_z" in zs:
        core.prune_heads(zero_heads("text_head_z"))
    if "cross_head_z" in zs:
        core.prune_heads(zero_heads("cross_head_z"), is_cross="cross")
    if "vision_intermediate_z" in zs:
        keep = {l: _z(zs, "vision_intermediate_z", l).nonzero().reshape(-1).tolist() for l in range(len(zs["vision_intermediate_z"]))}
        prune_vision_intermediate_layers(vision, keep, device, pad_to)
    if "text_intermediate_z" in zs and "cross_intermediate_z" in zs:
        allz = torch.cat((zs["text_intermediate_z"].detach().cpu(), zs["cross_intermediate_z"].detach().cpu()), dim=0)
        keep = {l: allz[l].reshape(-1).nonzero().reshape(-1).tolist() for l in range(len(allz))}
        prune_intermediate_layers(core, keep, device, pad_to)
    from . import ops
    ops.CACHE.invalidate()
    return model, zs


def retrieval_eval_losses(model, image, text_ids, text_atts, idx=None, zs=None, with_logits=False):
    """(loss_itc, loss_itm) of the eval branch of efficient_models/model_retrieval.py:76-93: without `zs` for a (pruned,
    gate-free) model; with `zs` (a dict of 0/1 gate tensors as l0_module.forward(training=False) returns) the masked-dense
    form the pruned model must reproduce.  with_logits: additionally the ITM head's [3B, 2] logits (logits_dict[
    "itm_head_logits"] of the training branch, :60-74)"""
    z = zs or {}
    image_embeds, image_atts = model.get_vision_embeds(image, head_z=z.get("vision_head_z"),
                                                       mlp_z=z.get("vision_intermediate_z"))[:2]
    text_embeds = model.get_text_embeds(text_ids, text_atts, head_z=z.get("text_head_z"), mlp_z=z.get("text_intermediate_z"))
    image_feat, text_feat = model.get_features(image_embeds, text_embeds)
    loss_itc = model.get_contrastive_loss(image_feat, text_feat, idx=idx)
    if with_logits:
        itm = model.get_matching_loss(image_embeds, image_atts, image_feat, text_embeds, text_atts, text_feat, idx=idx,
                                      output_attentions=False, output_hidden_states=True,
                                      head_z=z.get("cross_head_z"), mlp_z=z.get("cross_intermediate_z"))
        return loss_itc, itm["loss"], itm["logits"]
    loss_itm = model.get_matching_loss(image_embeds, image_atts, image_feat, text_embeds, text_atts, text_feat, idx=idx,
                                       head_z=z.get("cross_head_z"), mlp_z=z.get("cross_intermediate_z"))
    return loss_itc, loss_itm
