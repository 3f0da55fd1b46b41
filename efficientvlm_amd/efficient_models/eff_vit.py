"""CLIP-ViT image encoder with L0 gate hooks — drop-in for the reference's efficient_models/eff_vit.py
(and, with z=None, models/clip_vit.py).  Same class names, constructor arguments, forward signatures, return
tuples and state-dict keys; the arithmetic runs in the gfx950 kernels of libevlm_hip.so:

    patch conv        -> im2row + MFMA GEMM + token assembly           (eff_vit.py:444-449)
    LN1 / LN2 / pre / post LayerNorm -> evlm_layernorm                 (:250,:263,:452,:468)
    q,k,v projections -> ONE packed [3d,d] GEMM                         (:134-136)
    bmm-softmax-bmm   -> evlm_attention (probabilities are an output)   (:144-195, head gate :194-195)
    out_proj + residual -> GEMM with residual epilogue                  (:199,:261)
    fc1 * mlp_z, quick_gelu, fc2 + residual -> two GEMMs with fused epilogues (:214-220,:266)

nn.Linear / nn.LayerNorm / nn.Conv2d / nn.Embedding objects are kept only as PARAMETER CONTAINERS so the
checkpoint keys are the reference's; their ATen forward is never called.
"""
from typing import Optional

import torch
from torch import nn

from .. import ops
from .._lib import ACT_GELU, ACT_QUICK_GELU, GATE_PRE
from ..runtime import compute_dtype

_ACT = {"quick_gelu": ACT_QUICK_GELU, "gelu": ACT_GELU}


def find_pruneable_heads_and_indices(heads, n_heads, head_size, already_pruned_heads):
    """transformers 4.12.5 modeling_utils helper (removed upstream): kept rows after dropping `heads`."""
    mask = torch.ones(n_heads, head_size)
    heads = set(heads) - already_pruned_heads
    for head in heads:
        head = head - sum(1 if h < head else 0 for h in already_pruned_heads)
        mask[head] = 0
    mask = mask.view(-1).contiguous().eq(1)
    index = torch.arange(len(mask))[mask].long()
    return heads, index


def prune_linear_layer(layer, index, dim=0):
    """transformers prune_linear_layer: keep `index` rows (dim=0) / columns (dim=1)"""
    index = index.to(layer.weight.device)
    W = layer.weight.index_select(dim, index).clone().detach()
    b = None
    if layer.bias is not None:
        b = layer.bias.clone().detach() if dim == 1 else layer.bias[index].clone().detach()
    new = nn.Linear(W.shape[1], W.shape[0], bias=layer.bias is not None).to(layer.weight.device)
    new.weight.requires_grad = False
    new.weight.copy_(W.contiguous())
    new.weight.requires_grad = True
    if b is not None:
        new.bias.requires_grad = False
        new.bias.copy_(b.contiguous())
        new.bias.requires_grad = True
    return new


class CLIPAttention(nn.Module):
    """eff_vit.py:82-204"""

    def __init__(self, hidden_size, num_attention_heads, attention_dropout):
        super().__init__()
        self.embed_dim = hidden_size
        self.num_heads = num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        assert self.head_dim * self.num_heads == self.embed_dim, \
            f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.embed_dim} and `num_heads`: {self.num_heads})."
        self.scale = self.head_dim ** -0.5
        self.dropout = attention_dropout
        if attention_dropout:
            raise NotImplementedError("attention dropout > 0 is not used by any reference config (config_clipvit*.json: 0.0)")
        self.k_proj = nn.Linear(hidden_size, self.embed_dim)
        self.v_proj = nn.Linear(hidden_size, self.embed_dim)
        self.q_proj = nn.Linear(hidden_size, self.embed_dim)
        self.out_proj = nn.Linear(hidden_size, self.embed_dim)
        self.pruned_heads = set()

    def prune_heads(self, heads):
        """eff_vit.py:105-121"""
        if len(heads) == 0:
            return
        heads, index = find_pruneable_heads_and_indices(heads, self.num_heads, self.head_dim, self.pruned_heads)
        self.q_proj = prune_linear_layer(self.q_proj, index)
        self.k_proj = prune_linear_layer(self.k_proj, index)
        self.v_proj = prune_linear_layer(self.v_proj, index)
        self.out_proj = prune_linear_layer(self.out_proj, index, dim=1)
        self.num_heads = self.num_heads - len(heads)
        self.embed_dim = self.head_dim * self.num_heads
        self.pruned_heads = self.pruned_heads.union(heads)

    def forward(self, hidden_states, attention_mask=None, causal_attention_mask=None, output_attentions=False,
                head_z=None, head_layer_z=None, residual=None, kd_teacher=None, kd_word=None, p_out=None, recipe=False):
        """hidden_states [B,N,C] -> (attn_output [B,N,C], probs [B,H,N,N] | None).

        `residual` (extension): added in the out_proj GEMM epilogue; the layer passes the block input.
        `kd_teacher` (extension): the frozen teacher's map of the corresponding layer; the attention kernel then also
        returns MSELoss(probs, kd_teacher) * probs.shape[-1] (GeneralDistill.py:63-69) as a third output, computed while
        the probabilities are in registers."""
        bsz, tgt_len, _ = hidden_states.shape
        if causal_attention_mask is not None:
            raise NotImplementedError("causal masks are never passed on the vision path (eff_vit.py:254)")
        mask2d = None
        if attention_mask is not None:
            if attention_mask.size() != (bsz, 1, tgt_len, tgt_len):
                raise ValueError(f"Attention mask should be of size {(bsz, 1, tgt_len, tgt_len)}, but is {attention_mask.size()}")
            mask2d = attention_mask[:, 0, 0, :]       # the reference mask is a key mask expanded over query rows (:339)
        if head_layer_z is not None:
            raise NotImplementedError("head_layer_z is never produced by the reference L0 modules (SURVEY.md §3.4)")
        qkv = ops.linear_packed(hidden_states, (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight),
                                (self.q_proj.bias, self.k_proj.bias, self.v_proj.bias))
        if kd_teacher is not None:
            # (the map itself is materialised only when the caller wants it: the kernels compare the probabilities with the
            # teacher's in registers, and the backward rebuilds them from Q and K)
            out, probs, kd = ops.self_attention(qkv, self.num_heads, self.head_dim, self.scale, mask=mask2d, gate=head_z,
                                                # (no lse exists without a backward: a no_grad forward with a teacher map -
                                                # a validation-loss pass - takes the stored-map form of the term)
                                                want_probs=bool(output_attentions) or not (
                                                    torch.is_grad_enabled() and qkv.requires_grad
                                                    and ops.attention_recomputes(qkv, self.head_dim, tgt_len)),
                                                kd_teacher=kd_teacher,
                                                kd_weight=float(tgt_len) if kd_word is None else ops.KdSlot(kd_word, tgt_len))
            out = ops.linear(out, self.out_proj.weight, self.out_proj.bias, residual=residual)
            return out, (probs if output_attentions else None), kd
        if recipe and output_attentions and p_out is None and ops.map_recipe_supported(qkv, self.num_heads, self.head_dim, mask2d):
            # (`recipe`, extension: a frozen teacher whose kept maps are only read by fused distillation terms on long key
            # sequences hands out ops.MapRecipe - its QKV buffer + row lse - in the map's slot; no map is written)
            out, probs = ops.self_attention_recipe(qkv, self.num_heads, self.head_dim, self.scale, gate=head_z)
        else:
            out, probs = ops.self_attention(qkv, self.num_heads, self.head_dim, self.scale, mask=mask2d, gate=head_z,
                                            want_probs=bool(output_attentions), p_out=p_out if output_attentions else None)
        out = ops.linear(out, self.out_proj.weight, self.out_proj.bias, residual=residual)
        return out, (probs if output_attentions else None)


class CLIPMLP(nn.Module):
    """eff_vit.py:207-220 — the gate multiplies fc1's output BEFORE the activation"""

    def __init__(self, hidden_act, hidden_size, intermediate_size):
        super().__init__()
        self.hidden_act = hidden_act
        self.act_code = _ACT[hidden_act]
        self.fc1 = nn.Linear(hidden_size, intermediate_size)
        self.fc2 = nn.Linear(intermediate_size, hidden_size)

    def forward(self, hidden_states, mlp_z=None, residual=None):
        return ops.mlp(hidden_states, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.act_code,
                       gate=mlp_z, gate_pos=GATE_PRE, residual=residual)


class CLIPEncoderLayer(nn.Module):
    """eff_vit.py:223-273"""
    input_tap = None       # (set by forward: alias of the block input for the hidden-state distillation, see there)

    def __init__(self, hidden_size, hidden_act, num_attention_heads, attention_dropout, intermediate_size):
        super().__init__()
        self.self_attn = CLIPAttention(hidden_size, num_attention_heads, attention_dropout)
        self.layer_norm1 = nn.LayerNorm(hidden_size)
        self.mlp = CLIPMLP(hidden_act, hidden_size, intermediate_size)
        self.layer_norm2 = nn.LayerNorm(hidden_size)

    def forward(self, hidden_states, attention_mask: None, output_attentions: Optional[bool] = False, head_z=None,
                head_layer_z=None, mlp_z=None, kd_teacher=None, kd_word=None, p_out=None, kd_state=None, recipe=False):
        # (layer_norm_fork: the residual branch gets an ALIAS of the block input, so that the LayerNorm backward kernel sums
        # the two gradients of the input itself - evlm_layernorm_bwd_add - instead of autograd adding them element-wise)
        # (`tap`: a third alias of the block input, for the hidden-state distillation that reads it - CLIPEncoder reports it
        # in `encoder_states`, so that the input's two extra gradients reach the LayerNorm backward kernel as addends)
        # (`kd_state` = (teacher state, zeroed slots, weight / numel): the hidden-state distillation term of the block input is
        # formed INSIDE this LayerNorm's kernels - forward sum and backward gradient - and nothing else reads the input)
        self.kd_hidden_term = None
        if kd_state is not None and torch.is_grad_enabled() and hidden_states.requires_grad:
            h, residual, self.kd_hidden_term = ops.layer_norm_fork_kd(hidden_states, self.layer_norm1.weight,
                                                                      self.layer_norm1.bias, self.layer_norm1.eps, *kd_state)
            self.input_tap = None
        else:
            h, residual, self.input_tap = ops.layer_norm_fork(hidden_states, self.layer_norm1.weight, self.layer_norm1.bias,
                                                              self.layer_norm1.eps, tap=True)
        attn_out = self.self_attn(hidden_states=h, attention_mask=attention_mask,
                                  causal_attention_mask=None, output_attentions=output_attentions,
                                  head_z=head_z, head_layer_z=head_layer_z, residual=residual, kd_teacher=kd_teacher,
                                  kd_word=kd_word, p_out=p_out, recipe=recipe)
        hidden_states, attn_weights = attn_out[0], attn_out[1]
        self.kd_term = attn_out[2] if kd_teacher is not None else None
        h, residual = ops.layer_norm_fork(hidden_states, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps)
        hidden_states = self.mlp(h, mlp_z=mlp_z, residual=residual)
        outputs = (hidden_states,)
        if output_attentions:
            outputs += (attn_weights,)
        return outputs


class CLIPEncoder(nn.Module):
    """eff_vit.py:276-383"""

    def __init__(self, hidden_size, hidden_act, num_attention_heads, attention_dropout, intermediate_size,
                 num_hidden_layers, local_attn_depth):
        super().__init__()
        self.depth = num_hidden_layers
        self.local_attn_depth = local_attn_depth
        # extension (None = the reference's behaviour): the set of layers whose attention map is materialised when
        # output_attentions is set; the others return None in their slot (a frozen teacher whose other maps nobody reads)
        self.attn_keep = None
        # extension: per-layer teacher maps for the FUSED attention-map distillation (set by distill.gd_forward when the
        # teacher's outputs of this batch already exist - the pipelined trainer); the per-layer terms land in kd_fused
        self.kd_teacher_maps = None
        self.kd_fused = None
        # extension: per-layer teacher STATES (the input of student layer i against get_cor_teacher's state) for the fused
        # hidden-state distillation - formed inside each layer's first LayerNorm; the per-layer slot vectors land in
        # kd_hidden_fused (distill.fuse_image_map_kd / collect_fused_kd)
        self.kd_teacher_states = None
        self.kd_hidden_fused = None
        # extension (False = maps as tensors): a FROZEN TEACHER whose kept image maps are read by nothing but the student's
        # fused distillation kernels returns ops.MapRecipe objects in their slots on long key sequences (trainer.ITRTrainer /
        # VQATrainer set it with the prefetched teacher): QKV buffer + row lse instead of a [B, H, L, L] map
        self.attn_recipe = False
        # extension (False = the reference's behaviour of returning every map): a layer whose map distillation ran fused in
        # its attention kernel does not materialise the map (None in its slot) - nobody else reads a student's ViT maps in
        # the GD recipe (trainer.GDTrainer sets this on the student)
        self.kd_drop_maps = False
        # extension: {layer index: callback} - a tensor hook on the INPUT of that layer: it fires when backward has finished
        # with layers >= index (their parameter gradients are complete): the data-parallel trainer reduces them right then
        self.grad_hooks = None
        # extension: {layer index: caller-owned padded map buffer} - the layer's attention kernel writes its probability map
        # there (a pipelined frozen teacher parks its maps in persistent buffers: written in place, never copied)
        self.attn_out = None
        self.layers = nn.ModuleList([CLIPEncoderLayer(hidden_size, hidden_act, num_attention_heads, attention_dropout,
                                                      intermediate_size) for _ in range(num_hidden_layers)])

    def forward(self, inputs_embeds, idx_to_group_img=None, image_atts=None, output_attentions=None,
                output_hidden_states=None, head_z=None, head_layer_z=None, mlp_z=None):
        # (the L0 gates as per-layer rows that carry their gradient slot: ops.GateGradSlot)
        head_z, mlp_z = ops.gate_rows(head_z), ops.gate_rows(mlp_z)
        do_gather = idx_to_group_img is not None
        if do_gather and (image_atts is not None):                                     # :325-333
            full_atts = torch.ones(inputs_embeds.shape[:2], dtype=torch.float32, device=inputs_embeds.device)
            blk = torch.cat([image_atts.to(torch.float32), full_atts], dim=0).unsqueeze(1).unsqueeze(2)
            blk = (1.0 - blk) * -10000.0
            image_atts_blk = blk.expand(-1, -1, blk.size(-1), -1)
        else:
            image_atts_blk = None
        encoder_states = () if output_hidden_states else None
        all_attentions = () if output_attentions else None
        hidden_states = inputs_embeds
        kd_maps = self.kd_teacher_maps if (self.kd_teacher_maps is not None and not do_gather and output_attentions) else None
        self.kd_fused = [] if kd_maps is not None else None
        kd_words = torch.zeros(len(self.layers), dtype=torch.float32, device=inputs_embeds.device) if kd_maps is not None else None
        kd_states = self.kd_teacher_states if (self.kd_teacher_states is not None and not do_gather
                                               and output_hidden_states and torch.is_grad_enabled()
                                               and inputs_embeds.requires_grad) else None
        self.kd_hidden_fused = [] if kd_states is not None else None
        if kd_states is not None:
            nslot = ops.hidden_kd_slots()
            kd_hslots = ops.zeros_small((len(self.layers), nslot), torch.float32, inputs_embeds.device)
        for idx, encoder_layer in enumerate(self.layers):
            states_slot = len(encoder_states) if output_hidden_states else None
            if output_hidden_states:
                encoder_states = encoder_states + (hidden_states,)
            want_map = (bool(output_attentions) and (self.attn_keep is None or idx in self.attn_keep)
                        and not (kd_maps is not None and self.kd_drop_maps))
            kw = dict(output_attentions=want_map,
                      head_z=head_z[idx] if head_z is not None else None,
                      head_layer_z=head_layer_z[idx] if head_layer_z is not None else None,
                      mlp_z=mlp_z[idx] if mlp_z is not None else None)
            kdkw = dict(kd_teacher=kd_maps[idx], kd_word=kd_words[idx]) if kd_maps is not None else {}
            if self.attn_out and idx in self.attn_out and want_map and not do_gather and image_atts_blk is None:
                kdkw["p_out"] = self.attn_out[idx]
            if self.attn_recipe and want_map and not do_gather and image_atts_blk is None and not torch.is_grad_enabled():
                kdkw["recipe"] = True
            if kd_states is not None and kd_states[idx] is not None:
                # MSELoss (mean over the elements) with weight 1 (GeneralDistill.py:78-80)
                kdkw["kd_state"] = (kd_states[idx], kd_hslots[idx], 1.0 / float(hidden_states.numel()))
            if self.grad_hooks and idx in self.grad_hooks and hidden_states.requires_grad:
                cb = self.grad_hooks[idx]
                hidden_states.register_hook(lambda grad, cb=cb: (cb(), grad)[1])
            if (self.local_attn_depth > 0) and (idx >= self.depth - self.local_attn_depth):
                if do_gather:                                                           # :354-357
                    do_gather = False
                    hidden_states_bs = torch.index_select(hidden_states, 0, idx_to_group_img.view(-1))
                    hidden_states = torch.cat([hidden_states_bs, hidden_states], dim=0)
                gathered = hidden_states is not (encoder_states[states_slot] if states_slot is not None else hidden_states)
                layer_outputs = encoder_layer(hidden_states, attention_mask=image_atts_blk, **kw, **kdkw)
            else:
                gathered = False
                layer_outputs = encoder_layer(hidden_states, attention_mask=None, **kw, **kdkw)
            if states_slot is not None and not gathered and encoder_layer.input_tap is not None:
                # (same values: the alias the layer's first LayerNorm handed out - see CLIPEncoderLayer.forward)
                encoder_states = encoder_states[:states_slot] + (encoder_layer.input_tap,) + encoder_states[states_slot + 1:]
            encoder_layer.input_tap = None
            if kd_maps is not None:
                self.kd_fused.append(encoder_layer.kd_term)
            if kd_states is not None:
                self.kd_hidden_fused.append(encoder_layer.kd_hidden_term)
                encoder_layer.kd_hidden_term = None
            hidden_states = layer_outputs[0]
            if output_attentions:
                all_attentions = all_attentions + ((layer_outputs[1] if want_map else None),)
        if output_hidden_states:
            encoder_states = encoder_states + (hidden_states,)
        return (hidden_states, encoder_states, all_attentions)


class CLIPVisionTransformer(nn.Module):
    """eff_vit.py:386-474"""

    def __init__(self, image_size, patch_size, hidden_size, hidden_act, num_attention_heads, attention_dropout,
                 intermediate_size, num_hidden_layers, local_attn_depth=0):
        super().__init__()
        self.image_size = image_size
        self.patch_size = patch_size
        self.num_patch_embed = (self.image_size // self.patch_size) ** 2
        self.patch_embed = nn.Conv2d(in_channels=3, out_channels=hidden_size, kernel_size=self.patch_size,
                                     stride=self.patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(hidden_size))
        self.num_pos_embed = self.num_patch_embed + 1
        self.pos_embed = nn.Embedding(self.num_pos_embed, hidden_size)
        self.register_buffer("position_ids", torch.arange(self.num_pos_embed).expand((1, -1)))
        self.pre_layrnorm = nn.LayerNorm(hidden_size)
        self.encoder = CLIPEncoder(hidden_size, hidden_act, num_attention_heads, attention_dropout, intermediate_size,
                                   num_hidden_layers, local_attn_depth=local_attn_depth)
        self.post_layernorm = nn.LayerNorm(hidden_size)

    def prune_heads(self, heads_to_prune):
        """eff_vit.py:424-430"""
        for layer, heads in heads_to_prune.items():
            self.encoder.layers[layer].self_attn.prune_heads(heads)

    def forward(self, x, idx_to_group_img=None, image_atts=None, output_attentions=None, output_hidden_states=None,
                head_z=None, head_layer_z=None, mlp_z=None):
        hidden_states = ops.vit_embed(x, self.patch_embed.weight, self.class_embedding, self.pos_embed.weight,
                                      self.patch_size, compute_dtype())
        hidden_states = ops.layer_norm(hidden_states, self.pre_layrnorm.weight, self.pre_layrnorm.bias, self.pre_layrnorm.eps)
        encoder_outputs = self.encoder(inputs_embeds=hidden_states, idx_to_group_img=idx_to_group_img,
                                       image_atts=image_atts, output_attentions=output_attentions,
                                       output_hidden_states=output_hidden_states, head_z=head_z,
                                       head_layer_z=head_layer_z, mlp_z=mlp_z)
        outputs = ops.layer_norm(encoder_outputs[0], self.post_layernorm.weight, self.post_layernorm.bias,
                                 self.post_layernorm.eps)
        if idx_to_group_img is not None:
            bs = len(idx_to_group_img)
            outputs, outputs_fullatts = torch.split(outputs, [bs, outputs.size(0) - bs])
            return (outputs, encoder_outputs[1], encoder_outputs[2], outputs_fullatts)
        return (outputs, encoder_outputs[1], encoder_outputs[2])
