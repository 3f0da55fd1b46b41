"""BERT text / cross-modal encoder with L0 gate hooks — drop-in for the reference's efficient_models/eff_bert.py
(and, with z=None, models/xbert.py): same class names, constructor arguments, forward keywords, output objects and
state-dict keys.  Only the classes on the distillation path are provided (SURVEY.md §2 row 3:
BertFor{PreTraining,NSP,SequenceClassification,...} and load_tf_weights_in_bert are unused HF boilerplate).

Arithmetic (all in libevlm_hip.so):
    embeddings gather-add            -> evlm_bert_embed + evlm_layernorm (eps 1e-12)      (eff_bert.py:188-215)
    query/key/value                  -> ONE packed GEMM ([3d,d] self, [2d,d] key/value over image tokens for cross)
    matmul /sqrt(d) +mask softmax matmul *= head_z -> evlm_attention (probabilities returned BEFORE dropout, :338-361)
    BertSelfOutput dense + input     -> GEMM with residual epilogue, then LayerNorm        (:374-381)
    BertIntermediate gelu * mlp_z, BertOutput dense + input -> two GEMMs with fused epilogues, then LayerNorm (:445-462,:553-557)
    MLM head                         -> gather rows, dense+GELU epilogue, LayerNorm, tied-decoder GEMM, fused CE (:1691-1702)
    causal LM head (VQA decoder)     -> causal flag of evlm_attention, tied-decoder GEMM, evlm_ce_weighted (:1332-1443)

Dropout (hidden_dropout_prob / attention_probs_dropout_prob, eff_bert.py:180,214,242,346,372-379,456-460): in training
mode with p > 0 the hidden-state sites apply the keep-mask inside the producing GEMM's residual epilogue (round 6:
evlm_gemm_args.dropout_p; their gradients leave the following LayerNorm's backward already masked, evlm_layernorm_bwd_drop;
the embedding site and EVLM_NO_FUSED_DROPOUT=1 run evlm_dropout) and the attention kernels - the bf16 MFMA ones included -
drop the probabilities that form the context; masks are counter-based (Philox, keyed by a device
{seed, step} word and a per-site call id) and regenerated in backward - no mask tensor exists.  eval() models and p = 0
take the fully fused p = 0 path.  The reference's CUDA RNG stream is not reproducible: parity tests feed
the SAME masks to the oracle (ops.dropout_mask).
"""
import math
import os

import torch
from torch import nn

from .. import ops
from .._lib import ACT_GELU, GATE_POST
from ..runtime import BertConfig, ModelOutput, compute_dtype
from .eff_vit import find_pruneable_heads_and_indices, prune_linear_layer

__all__ = ["BertConfig", "BertModel", "BertForMaskedLM", "BertEmbeddings", "BertSelfAttention", "BertSelfOutput",
           "BertAttention", "BertIntermediate", "BertOutput", "BertLayer", "BertEncoder", "BertLMPredictionHead",
           "BertOnlyMLMHead", "BertPredictionHeadTransform", "BertPreTrainedModel", "MaskedLMOutput", "BertLMHeadModel",
           "CausalLMOutputWithCrossAttentions", "CausalMask"]


# the no-grad cross-attention forward through the fused kernel (evlm_xattn_fused_fwd); False = always the two-launch form
# (packed K/V GEMM + attention kernel).  Set from the measured comparison in profiles/r02_xattn_fused.md.
FUSED_CROSS_ATTENTION = False
# one K/V projection of the image tokens for ALL fusion layers of an encoder (round 4); EVLM_NO_MERGED_KV=1: one per layer
MERGED_CROSS_KV = not os.environ.get("EVLM_NO_MERGED_KV")
# hidden-state dropout inside the producing GEMM's residual epilogue (round 6); EVLM_NO_FUSED_DROPOUT=1: the separate
# evlm_dropout pass per site (A/B switch, and what tests compare the fused form with)
FUSED_HIDDEN_DROPOUT = not os.environ.get("EVLM_NO_FUSED_DROPOUT")


def _p(config, name):
    p = float(getattr(config, name, 0.0) or 0.0)
    if not 0.0 <= p < 1.0:
        raise ValueError(f"{name} = {p} must lie in [0, 1)")
    return p


def _run(gen):
    """drive a *_gen generator to its end and return its value"""
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


class BertEmbeddings(nn.Module):
    """eff_bert.py:168-215"""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.hidden_dropout_prob = _p(config, "hidden_dropout_prob")
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.position_embedding_type = getattr(config, "position_embedding_type", "absolute")
        if self.position_embedding_type != "absolute":
            raise NotImplementedError("only absolute position embeddings are used by the reference configs")
        self.config = config

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0):
        if inputs_embeds is not None or position_ids is not None or past_key_values_length:
            raise NotImplementedError("inputs_embeds / custom position_ids / cached decoding are off the distillation path")
        if token_type_ids is not None and bool((token_type_ids != 0).any()):
            raise NotImplementedError("token_type_ids != 0 never occur on the distillation path (eff_bert.py:201-202)")
        L = input_ids.shape[1]
        e = ops.bert_embed(input_ids, self.word_embeddings.weight, self.position_embeddings.weight[:L],
                           self.token_type_embeddings.weight, self.config.pad_token_id, compute_dtype())
        e = ops.layer_norm(e, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        return ops.dropout(e, self.hidden_dropout_prob, self.training)                  # eff_bert.py:214


class BertSelfAttention(nn.Module):
    """eff_bert.py:218-364 (parameters + head bookkeeping; the math is issued by BertAttention.forward)"""

    def __init__(self, config, is_cross_attention):
        super().__init__()
        self.config = config
        if config.hidden_size % config.num_attention_heads != 0 and not hasattr(config, "embedding_size"):
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.fp16 = getattr(config, "fp16", False)   # pre- vs post-scaling of Q: same maths
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        kv_in = config.encoder_width if is_cross_attention else config.hidden_size
        self.key = nn.Linear(kv_in, self.all_head_size)
        self.value = nn.Linear(kv_in, self.all_head_size)
        self.attention_probs_dropout_prob = _p(config, "attention_probs_dropout_prob")
        self.is_cross_attention = is_cross_attention
        self.input_alias = None        # set by forward: alias of its input for the residual of BertSelfOutput (see forward)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, head_z=None,
                encoder_batch_index=None, encoder_kv=None):
        """returns (context [B,L,all_head], probs | None).  masks are additive [B,1,1,Lk] (key masks).

        encoder_batch_index (extension): LongTensor [B] mapping each text batch row to its row of encoder_hidden_states,
        so the K/V projection of an image shared by several text rows is computed (and differentiated) once."""
        if head_mask is not None or past_key_value is not None:
            raise NotImplementedError("head_mask / past_key_value are off the distillation path")
        H, dh = self.num_attention_heads, self.attention_head_size
        scale = 1.0 / math.sqrt(dh)
        drop = self.attention_probs_dropout_prob if self.training else 0.0          # eff_bert.py:346 (probs returned un-dropped)
        # (linear_fork: the BertSelfOutput residual gets an ALIAS of the block input, whose gradient the projection's dX GEMM
        # adds in its epilogue - one use of the input for autograd, no element-wise gradient add per block)
        if encoder_hidden_states is not None:
            q, self.input_alias = ops.linear_fork(hidden_states, (self.query.weight,), (self.query.bias,))
            if (FUSED_CROSS_ATTENTION and drop == 0.0 and
                    ops.xattn_fusable(q, encoder_hidden_states, (self.key.weight, self.value.weight), H, dh)):
                # no-grad forward (frozen teacher, inference): K/V projection + attention in ONE launch, K/V never in HBM
                ctx, probs = ops.cross_attention_fused(q, encoder_hidden_states, (self.key.weight, self.value.weight),
                                                       (self.key.bias, self.value.bias), H, dh, scale,
                                                       mask=_key_mask(encoder_attention_mask), gate=head_z,
                                                       want_probs=bool(output_attentions), kv_index=encoder_batch_index)
                return ((ctx, probs) if output_attentions else (ctx,)) + (None,)
            if encoder_kv is not None:                          # (BertEncoder projected K/V of every fusion layer at once)
                kv, col, slot = encoder_kv
            else:
                kv, col, slot = ops.linear_packed(encoder_hidden_states, (self.key.weight, self.value.weight),
                                                  (self.key.bias, self.value.bias)), None, None
            ctx, probs = ops.cross_attention(q, kv, H, dh, scale, mask=_key_mask(encoder_attention_mask), gate=head_z,
                                             want_probs=bool(output_attentions), kv_index=encoder_batch_index, dropout_p=drop,
                                             kv_col=col, kv_grad=slot)
        else:
            qkv, self.input_alias = ops.linear_fork(hidden_states, (self.query.weight, self.key.weight, self.value.weight),
                                                    (self.query.bias, self.key.bias, self.value.bias))
            ctx, probs = ops.self_attention(qkv, H, dh, scale, mask=_key_mask(attention_mask), gate=head_z,
                                            want_probs=bool(output_attentions), causal=isinstance(attention_mask, CausalMask),
                                            dropout_p=drop)
        outputs = (ctx, probs) if output_attentions else (ctx,)
        return outputs + (None,)


class CausalMask:
    """The decoder's additive [B,1,L,L] mask (1 - causal * padding) * -10000 of get_extended_attention_mask's is_decoder
    branch (eff_bert.py:975-1012) in FACTORED form: `key` is the additive key-padding part [B,1,1,L]; the causal part
    (keys after the query) is applied inside the attention kernel (evlm_attn_fwd_args.causal), so the [B,1,L,L] tensor
    is never materialised."""

    def __init__(self, key):
        self.key = key


def _key_mask(m):
    """additive [B,1,1,Lk] (or [B,Lk]) -> [B,Lk] fp32 for the attention kernel"""
    if m is None:
        return None
    if isinstance(m, CausalMask):
        m = m.key
    if m.dim() == 4:
        if m.shape[1] != 1 or m.shape[2] != 1:
            raise NotImplementedError("arbitrary per-query masks are not on the path (padding and causal masks are)")
        m = m[:, 0, 0, :]
    return m


class BertSelfOutput(nn.Module):
    """eff_bert.py:367-381"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.hidden_dropout_prob = _p(config, "hidden_dropout_prob")

    def forward(self, hidden_states, input_tensor, head_layer_z=None):
        if head_layer_z is not None:
            raise NotImplementedError("head_layer_z is dead plumbing in the reference (BertEncoder never forwards it)")
        drop = self.hidden_dropout_prob if self.training else 0.0
        if drop > 0.0 and not FUSED_HIDDEN_DROPOUT:                  # LayerNorm(dropout(dense(h)) + input), eff_bert.py:374-381
            h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
            h = ops.dropout(h, drop, True, residual=input_tensor)
        else:                         # the residual - and, round 6, the keep-mask ahead of it - ride on the GEMM epilogue
            h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, residual=input_tensor, dropout_p=drop)
        return ops.layer_norm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)


class BertAttention(nn.Module):
    """eff_bert.py:384-433"""

    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = BertSelfAttention(config, is_cross_attention)
        self.output = BertSelfOutput(config)
        self.pruned_heads = set()

    def prune_heads(self, heads):
        if len(heads) == 0:
            return
        heads, index = find_pruneable_heads_and_indices(heads, self.self.num_attention_heads,
                                                        self.self.attention_head_size, self.pruned_heads)
        self.self.query = prune_linear_layer(self.self.query, index)
        self.self.key = prune_linear_layer(self.self.key, index)
        self.self.value = prune_linear_layer(self.self.value, index)
        self.output.dense = prune_linear_layer(self.output.dense, index, dim=1)
        self.self.num_attention_heads = self.self.num_attention_heads - len(heads)
        self.self.all_head_size = self.self.attention_head_size * self.self.num_attention_heads
        self.pruned_heads = self.pruned_heads.union(heads)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, head_z=None, head_layer_z=None,
                encoder_batch_index=None, encoder_kv=None):
        self_outputs = self.self(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                                 past_key_value, output_attentions, head_z=head_z, encoder_batch_index=encoder_batch_index,
                                 **({} if encoder_kv is None else {"encoder_kv": encoder_kv}))
        residual, self.self.input_alias = self.self.input_alias, None
        attention_output = self.output(self_outputs[0], residual if residual is not None else hidden_states,
                                       head_layer_z=head_layer_z)
        return (attention_output,) + self_outputs[1:]


class BertIntermediate(nn.Module):
    """eff_bert.py:436-448 (parameter container; fused into BertLayer.feed_forward_chunk)"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise NotImplementedError("BERT hidden_act other than erf-GELU is not used by the reference configs")

    def forward(self, hidden_states):
        return ops.linear(hidden_states, self.dense.weight, self.dense.bias, act=ACT_GELU)


class BertOutput(nn.Module):
    """eff_bert.py:451-462"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.hidden_dropout_prob = _p(config, "hidden_dropout_prob")

    def forward(self, hidden_states, input_tensor):
        drop = self.hidden_dropout_prob if self.training else 0.0
        if drop > 0.0 and not FUSED_HIDDEN_DROPOUT:                  # eff_bert.py:458-462
            h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
            h = ops.dropout(h, drop, True, residual=input_tensor)
        else:
            h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, residual=input_tensor, dropout_p=drop)
        return ops.layer_norm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)


class BertLayer(nn.Module):
    """eff_bert.py:465-560"""

    def __init__(self, config, layer_num):
        super().__init__()
        self.config = config
        self.attention = BertAttention(config)
        self.has_cross_attention = (layer_num >= config.fusion_layer)
        if self.has_cross_attention:
            self.layer_num = layer_num
            self.crossattention = BertAttention(config, is_cross_attention=True)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, head_z=None,
                head_layer_z=None, mlp_z=None, encoder_batch_index=None, keep_self_map=True, keep_cross_map=True,
                encoder_kv=None):
        """keep_self_map / keep_cross_map (extension, default = reference behaviour): with output_attentions set, a map
        that is not kept is not materialised and its slot in the returned tuple holds None"""
        if self.has_cross_attention and head_z is not None:
            assert isinstance(head_z, tuple)
            head_z, cross_head_z = head_z
        else:
            cross_head_z = None
        want_self = bool(output_attentions) and keep_self_map
        self_attention_outputs = self.attention(hidden_states, attention_mask, head_mask,
                                                output_attentions=want_self, head_z=head_z, head_layer_z=head_layer_z)
        attention_output = self_attention_outputs[0]
        outputs = self_attention_outputs[1:-1] if want_self else ((None,) if output_attentions else ())
        if self.has_cross_attention:
            assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
            if type(encoder_hidden_states) == list:                               # eff_bert.py:517-527
                k = (self.layer_num - self.config.fusion_layer) % len(encoder_hidden_states)
                enc, enc_mask = encoder_hidden_states[k], encoder_attention_mask[k]
            else:
                enc, enc_mask = encoder_hidden_states, encoder_attention_mask
            want_cross = bool(output_attentions) and keep_cross_map
            cross_attention_outputs = self.crossattention(attention_output, attention_mask, head_mask, enc, enc_mask,
                                                          output_attentions=want_cross, head_z=cross_head_z,
                                                          encoder_batch_index=encoder_batch_index,
                                                          encoder_kv=None if type(encoder_hidden_states) == list else encoder_kv)
            attention_output = cross_attention_outputs[0]
            outputs = outputs + (cross_attention_outputs[1:-1] if want_cross else ((None,) if output_attentions else ()))
        self.mlp_z = mlp_z
        layer_output = self.feed_forward_chunk(attention_output)
        return (layer_output,) + outputs + (None,)

    def feed_forward_chunk(self, attention_output):
        """eff_bert.py:552-560: gelu(dense(x)) * mlp_z -> dense -> +x -> LayerNorm (the gate comes AFTER the activation)"""
        o = self.output
        drop = o.hidden_dropout_prob if self.training else 0.0
        if drop > 0.0 and not FUSED_HIDDEN_DROPOUT:                  # BertOutput: LayerNorm(dropout(dense(h)) + input), :458-462
            h = ops.mlp(attention_output, self.intermediate.dense.weight, self.intermediate.dense.bias, o.dense.weight,
                        o.dense.bias, ACT_GELU, gate=self.mlp_z, gate_pos=GATE_POST, residual=None)
            h = ops.dropout(h, drop, True, residual=attention_output)
        else:                         # (round 6: the keep-mask in the second product's residual epilogue)
            h = ops.mlp(attention_output, self.intermediate.dense.weight, self.intermediate.dense.bias, o.dense.weight,
                        o.dense.bias, ACT_GELU, gate=self.mlp_z, gate_pos=GATE_POST, residual=attention_output, dropout_p=drop)
        return ops.layer_norm(h, o.LayerNorm.weight, o.LayerNorm.bias, o.LayerNorm.eps)


class BertEncoder(nn.Module):
    """eff_bert.py:563-694"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])
        self.fusion_layer = self.config.fusion_layer
        # extension (None = the reference's behaviour): absolute layer indices whose self- / cross-attention maps are
        # materialised when output_attentions is set (see BertLayer.forward)
        self.attn_keep = None
        self.cross_keep = None

    def _merged_cross_kv(self, layers, enc, enc_index):
        """{layer index: (merged K/V buffer, first column, gradient slot)} - the cross-attention K / V of every fusion layer
        in `layers` from ONE product over the image tokens (N = n * 2 * all_head_size instead of n products of N = 2 *
        all_head_size: eff_bert.py:284-296 per layer).  Empty when the layers cannot share a buffer (fp32 parity path,
        physically pruned projections of different widths, a list of encoder states)."""
        xl = [i for i in layers if self.layer[i].has_cross_attention]
        if (not MERGED_CROSS_KV or len(xl) < 2 or enc is None or type(enc) == list or not enc.is_cuda
                or enc.dtype != torch.bfloat16 or enc.dim() != 3 or enc.shape[1] > 928):
            return {}
        sas = [self.layer[i].crossattention.self for i in xl]
        d = sas[0].all_head_size
        if any(sa.all_head_size != d or sa.attention_head_size != 64 or sa.key.bias is None or sa.value.bias is None
               for sa in sas) or d % 8 != 0:
            return {}
        if enc_index is None and FUSED_CROSS_ATTENTION:
            return {}
        ws = [w for sa in sas for w in (sa.key.weight, sa.value.weight)]
        bs = [b for sa in sas for b in (sa.key.bias, sa.value.bias)]
        kv, slot = ops.merged_kv(enc, ws, bs, len(xl))
        return {i: (kv, 2 * d * n, slot) for n, i in enumerate(xl)}

    def forward(self, *args, **kwargs):
        return _run(self.forward_gen(*args, **kwargs))

    def forward_gen(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                    encoder_attention_mask=None, past_key_values=None, use_cache=None, output_attentions=False,
                    output_hidden_states=False, return_dict=True, mode="multi_modal", head_z=None, head_layer_z=None, mlp_z=None,
                    encoder_batch_index=None):
        """forward() as a GENERATOR (extension, round 6): yields ("layer", i) behind every layer and returns forward()'s value
        (StopIteration.value).  A trainer resumes it layer by layer where it wants the next piece issued - the pipelined
        teacher's fusion pass beside successive hipGraph segments of the multi-GPU student step (trainer._capture_segments,
        `late` placement); forward() drives it to the end."""
        all_hidden_states = () if output_hidden_states else None
        all_self_attentions = () if output_attentions else None
        all_cross_attentions = () if output_attentions else None
        if mode == "text":
            start_layer, output_layer = 0, self.fusion_layer
        elif mode == "fusion":
            start_layer, output_layer = self.fusion_layer, self.config.num_hidden_layers
        elif mode == "multi_modal":
            start_layer, output_layer = 0, self.config.num_hidden_layers
        else:
            raise ValueError(f"mode {mode} is not supported")
        merged = self._merged_cross_kv(range(start_layer, output_layer), encoder_hidden_states, encoder_batch_index)
        # (the L0 gates as per-layer rows that carry their gradient slot: ops.GateGradSlot; indexing below is unchanged)
        head_z, mlp_z = ops.gate_rows(head_z), ops.gate_rows(mlp_z)
        for i in range(start_layer, output_layer):
            layer_module = self.layer[i]
            if output_hidden_states:
                all_hidden_states = all_hidden_states + (hidden_states,)
            # gate indexing reproduced verbatim, including the offset-free multi_modal quirk (eff_bert.py:612-620)
            if i >= self.fusion_layer and head_z is not None:
                first = (i - self.fusion_layer) * 2
                cur_head_z = (head_z[first], head_z[first + 1])
                cur_mlp_z = mlp_z[i - self.fusion_layer]
            elif head_z is not None:
                cur_head_z = head_z[i]
                cur_mlp_z = mlp_z[i]
            else:
                cur_mlp_z, cur_head_z = None, None
            layer_outputs = layer_module(hidden_states, attention_mask, None, encoder_hidden_states, encoder_attention_mask,
                                         None, output_attentions, head_z=cur_head_z if head_z is not None else None,
                                         mlp_z=cur_mlp_z if mlp_z is not None else None,
                                         encoder_batch_index=encoder_batch_index,
                                         keep_self_map=self.attn_keep is None or i in self.attn_keep,
                                         keep_cross_map=self.cross_keep is None or i in self.cross_keep,
                                         encoder_kv=merged.get(i))
            hidden_states = layer_outputs[0]
            if output_attentions:
                all_self_attentions = all_self_attentions + (layer_outputs[1],)
                if len(layer_outputs) > 3:
                    all_cross_attentions = all_cross_attentions + (layer_outputs[2],)
            yield ("layer", i)
        if output_hidden_states:
            all_hidden_states = all_hidden_states + (hidden_states,)
        if not return_dict:
            return tuple(v for v in [hidden_states, None, all_hidden_states, all_self_attentions, all_cross_attentions]
                         if v is not None)
        return ModelOutput(last_hidden_state=hidden_states, past_key_values=None, hidden_states=all_hidden_states,
                           attentions=all_self_attentions, cross_attentions=all_cross_attentions)


class BertPredictionHeadTransform(nn.Module):
    """eff_bert.py:712-726"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, act=ACT_GELU)
        return ops.layer_norm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)


class BertLMPredictionHead(nn.Module):
    """eff_bert.py:729-746: decoder.weight is tied to the word embeddings, decoder.bias to `bias`"""

    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        return ops.linear(h, self.decoder.weight, self.decoder.bias)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertPreTrainedModel(nn.Module):
    """the slice of transformers.PreTrainedModel (4.12.5) the path uses: config, _init_weights, init_weights + tying"""
    base_model_prefix = "bert"

    def __init__(self, config):
        super().__init__()
        self.config = BertConfig.from_any(config)

    def _init_weights(self, module):
        """eff_bert.py:792-802"""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def get_input_embeddings(self):
        return None

    def get_output_embeddings(self):
        return None

    def init_weights(self):
        self.apply(self._init_weights)
        self.tie_weights()

    def tie_weights(self):
        out, inp = self.get_output_embeddings(), self.get_input_embeddings()
        if out is not None and inp is not None:
            out.weight = inp.weight

    @property
    def dtype(self):
        return torch.float32


class BertModel(BertPreTrainedModel):
    """eff_bert.py:896-1162"""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__(config)
        config = self.config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = None
        if add_pooling_layer:
            raise NotImplementedError("the pooler is never built on the path (xvlm.py:180 add_pooling_layer=False)")
        self.init_weights()

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def set_input_embeddings(self, value):
        self.embeddings.word_embeddings = value

    def prune_heads(self, heads_to_prune, is_cross=None):
        """eff_bert.py:925-949"""
        if is_cross == "cross":
            for layer, heads in heads_to_prune.items():
                remainder, quotient = layer % 2, layer // 2
                if remainder == 0:
                    self.encoder.layer[3 + quotient].attention.prune_heads(heads)
                else:
                    self.encoder.layer[3 + quotient].crossattention.prune_heads(heads)
        elif is_cross == "decoder":
            for layer, heads in heads_to_prune.items():
                remainder, quotient = layer % 2, layer // 2
                if remainder == 0:
                    self.encoder.layer[quotient].attention.prune_heads(heads)
                else:
                    self.encoder.layer[quotient].crossattention.prune_heads(heads)
        else:
            for layer, heads in heads_to_prune.items():
                self.encoder.layer[layer].attention.prune_heads(heads)

    def get_extended_attention_mask(self, attention_mask, input_shape, device, is_decoder):
        """eff_bert.py:953-1013: (1 - m) * -10000 broadcast to [B,1,1,L]"""
        if attention_mask.dim() == 3:
            ext = attention_mask[:, None, :, :]
        elif attention_mask.dim() == 2:
            if is_decoder:                      # causal AND padding: see CausalMask
                if attention_mask.shape[1] != input_shape[1]:
                    raise NotImplementedError("prefix (past_key_values) masks are off the training path")
                return CausalMask(ops.additive_mask(attention_mask[:, None, None, :]))
            ext = attention_mask[:, None, None, :]
        else:
            raise ValueError("Wrong shape for input_ids (shape {}) or attention_mask (shape {})".format(
                input_shape, attention_mask.shape))
        return ops.additive_mask(ext)

    def invert_attention_mask(self, encoder_attention_mask):
        """transformers 4.12.5 ModuleUtilsMixin.invert_attention_mask: (1 - m) * -10000 (fp32)"""
        if encoder_attention_mask.dim() == 3:
            ext = encoder_attention_mask[:, None, :, :]
        else:
            ext = encoder_attention_mask[:, None, None, :]
        return ops.additive_mask(ext)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, is_decoder=False, mode="multi_modal", head_z=None, head_layer_z=None, mlp_z=None,
                encoder_batch_index=None):
        return _run(self.forward_gen(input_ids, attention_mask, token_type_ids, position_ids, head_mask, inputs_embeds, encoder_embeds,
                                     encoder_hidden_states, encoder_attention_mask, past_key_values, use_cache, output_attentions,
                                     output_hidden_states, return_dict, is_decoder, mode, head_z, head_layer_z, mlp_z,
                                     encoder_batch_index))

    def forward_gen(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                    inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                    past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                    return_dict=None, is_decoder=False, mode="multi_modal", head_z=None, head_layer_z=None, mlp_z=None,
                    encoder_batch_index=None):
        """forward() as a generator over its encoder's layers (BertEncoder.forward_gen); returns forward()'s value"""
        output_attentions = output_attentions if output_attentions is not None else self.config.output_attentions
        output_hidden_states = output_hidden_states if output_hidden_states is not None else self.config.output_hidden_states
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        elif input_ids is not None:
            input_shape, device = input_ids.size(), input_ids.device
        elif inputs_embeds is not None:
            input_shape, device = inputs_embeds.size()[:-1], inputs_embeds.device
        elif encoder_embeds is not None:
            input_shape, device = encoder_embeds.size()[:-1], encoder_embeds.device
        else:
            raise ValueError("You have to specify either input_ids or inputs_embeds or encoder_embeds")
        batch_size, seq_length = input_shape
        if past_key_values is not None or head_mask is not None:
            raise NotImplementedError("past_key_values / head_mask are off the distillation path")
        if attention_mask is None:
            attention_mask = torch.ones((batch_size, seq_length), device=device)
        extended_attention_mask = self.get_extended_attention_mask(attention_mask, input_shape, device, is_decoder)
        if encoder_hidden_states is not None:
            if type(encoder_hidden_states) == list:
                enc_shape = encoder_hidden_states[0].size()[:2]
            else:
                enc_shape = encoder_hidden_states.size()[:2]
            if type(encoder_attention_mask) == list:
                encoder_extended_attention_mask = [self.invert_attention_mask(m) for m in encoder_attention_mask]
            elif encoder_attention_mask is None:
                # (the reference builds an all-ones mask here: an additive bias of 0.0 on every key - the attention kernels'
                # no-mask form adds the same 0.0, so nothing is built)
                encoder_extended_attention_mask = None
            else:
                encoder_extended_attention_mask = self.invert_attention_mask(encoder_attention_mask)
        else:
            encoder_extended_attention_mask = None
        if encoder_embeds is None:
            embedding_output = self.embeddings(input_ids=input_ids, position_ids=position_ids,
                                               token_type_ids=token_type_ids, inputs_embeds=inputs_embeds)
        else:
            embedding_output = encoder_embeds
        encoder_outputs = yield from self.encoder.forward_gen(
            embedding_output, attention_mask=extended_attention_mask, head_mask=None,
            encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=encoder_extended_attention_mask,
            output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=return_dict, mode=mode,
            head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z, encoder_batch_index=encoder_batch_index)
        sequence_output = encoder_outputs[0]
        if not return_dict:
            return (sequence_output, None) + tuple(encoder_outputs[1:])
        return ModelOutput(last_hidden_state=sequence_output, pooler_output=None,
                           past_key_values=encoder_outputs.past_key_values, hidden_states=encoder_outputs.hidden_states,
                           attentions=encoder_outputs.attentions, cross_attentions=encoder_outputs.cross_attentions)


class MaskedLMOutput(ModelOutput):
    """eff_bert.py:1601-1608"""


class BertForMaskedLM(BertPreTrainedModel):
    """eff_bert.py:1610-1728"""

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertModel(self.config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(self.config)
        self.init_weights()

    def get_input_embeddings(self):
        return self.bert.embeddings.word_embeddings

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def set_output_embeddings(self, new_embeddings):
        self.cls.predictions.decoder = new_embeddings

    def gather_seq_out_by_pos(self, seq, pos):
        return ops.gather_rows(seq, pos)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False,
                mode="multi_modal", return_logits=False, masked_pos=None, head_z=None, head_layer_z=None, mlp_z=None):
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        outputs = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                            position_ids=position_ids, head_mask=head_mask, inputs_embeds=inputs_embeds,
                            encoder_embeds=encoder_embeds, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, output_attentions=output_attentions,
                            output_hidden_states=output_hidden_states, return_dict=return_dict, is_decoder=is_decoder,
                            mode=mode, head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)
        sequence_output = outputs[0]
        if masked_pos is not None:
            sequence_output = self.gather_seq_out_by_pos(sequence_output, masked_pos)
        prediction_scores = self.cls(sequence_output)
        if return_logits:
            return prediction_scores
        masked_lm_loss = None
        if labels is not None:
            masked_lm_loss = ops.cross_entropy(prediction_scores.reshape(-1, self.config.vocab_size), labels.reshape(-1))
        if not return_dict:
            output = (prediction_scores,) + tuple(outputs[2:])
            return ((masked_lm_loss,) + output) if masked_lm_loss is not None else output
        return MaskedLMOutput(loss=masked_lm_loss, logits=prediction_scores, hidden_states=outputs.hidden_states,
                              attentions=outputs.attentions, cross_attentions=outputs.cross_attentions)


class CausalLMOutputWithCrossAttentions(ModelOutput):
    """transformers.modeling_outputs.CausalLMOutputWithCrossAttentions (fields: loss, logits, past_key_values, hidden_states,
    attentions, cross_attentions)"""


class BertLMHeadModel(BertPreTrainedModel):
    """eff_bert.py:1308-1443: BERT with a causal language-modelling head - the VQA answer decoder (every layer has self-
    and cross-attention when config.fusion_layer == 0).  The next-token loss with reduction='none' is returned per sequence
    (sum over its non-ignored positions), as the reference does."""

    def __init__(self, config, label_smoothing=0.0):
        super().__init__(config)
        self.bert = BertModel(self.config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(self.config)
        if label_smoothing:
            raise NotImplementedError("label smoothing is 0 in every reference config (model_generation.py:41)")
        self.label_smoothing = label_smoothing
        self.init_weights()

    def get_input_embeddings(self):
        return self.bert.embeddings.word_embeddings

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def set_output_embeddings(self, new_embeddings):
        self.cls.predictions.decoder = new_embeddings

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                is_decoder=True, reduction="mean", mode="multi_modal", return_logits=False, head_z=None, mlp_z=None,
                encoder_batch_index=None, sequence_weights=None):
        """sequence_weights (extension, with reduction='none'): per-sequence weights w; the returned loss is then the
        scalar sum_r w[r] * (summed next-token CE of sequence r), computed by one fused kernel pass over the logits
        instead of materialising the shifted copy and the per-token loss vector."""
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        outputs = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                            head_mask=head_mask, inputs_embeds=inputs_embeds, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, past_key_values=past_key_values,
                            use_cache=False if labels is not None else use_cache, output_attentions=output_attentions,
                            output_hidden_states=output_hidden_states, return_dict=return_dict, is_decoder=is_decoder,
                            mode=mode, head_z=head_z, mlp_z=mlp_z, encoder_batch_index=encoder_batch_index)
        sequence_output = outputs[0]
        prediction_scores = self.cls(sequence_output)
        if return_logits:
            return prediction_scores[:, :-1, :].contiguous()
        lm_loss = None
        if labels is not None:
            # next-token prediction (eff_bert.py:1419-1421) WITHOUT the shifted copy of the logits: position t is scored
            # against label t+1, the last position against ignore_index
            B, La = labels.shape
            nxt = torch.cat([labels[:, 1:], labels.new_full((B, 1), -100)], dim=1)
            flat = prediction_scores.reshape(B * La, self.config.vocab_size)
            if reduction == "none":
                w = sequence_weights if sequence_weights is not None else None
                if w is None:
                    raise NotImplementedError("reduction='none' is served through sequence_weights (the VQA loss is "
                                              "sum(weights * per-sequence loss), model_generation.py:166)")
                lm_loss = ops.cross_entropy_weighted_sum(flat, nxt.reshape(-1), w.reshape(B, 1).expand(B, La).reshape(-1))
            elif reduction == "mean":
                lm_loss = ops.cross_entropy(flat, nxt.reshape(-1))
            else:
                raise NotImplementedError(reduction)
        if not return_dict:
            output = (prediction_scores,) + tuple(outputs[2:])
            return ((lm_loss,) + output) if lm_loss is not None else output
        return CausalLMOutputWithCrossAttentions(loss=lm_loss, logits=prediction_scores, past_key_values=outputs.past_key_values,
                                                 hidden_states=outputs.hidden_states, attentions=outputs.attentions,
                                                 cross_attentions=outputs.cross_attentions)
