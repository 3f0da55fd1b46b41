"""X-VLM base model — drop-in for the reference's efficient_models/xvlm.py (and models/xvlm.py, which is the same
code without the z keyword arguments): XVLMBase, AllGather/allgather, build_mlp, build_vision_encoder,
build_text_encoder, load_pretrained, with the reference's method names, keyword names, return structures and
state-dict keys.

Kernel use: vision/text encoders are the HIP-backed eff_vit / eff_bert; projection heads, ITM head, L2
normalisation, similarity matrices and the ITC / ITM / MLM cross-entropies are HIP ops.  What stays in PyTorch is
plumbing only: the all-gather collective (RCCL), index_select/cat for the hard-negative batches, label tensors, and
scalar glue (division by `temp`).  The hard-negative draw itself is one evlm_sample_negatives launch.
"""
import os
import weakref

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..runtime import BertConfig, log_collective, read_json
from .eff_bert import BertForMaskedLM, BertModel
from .eff_vit import CLIPVisionTransformer


def load_params_change_prefix(state_dict: dict, prefix: str, new_prefix: str):
    """efficient_models/xvlm.py:24-36"""
    if prefix == new_prefix:
        return state_dict
    out = {}
    for k, v in state_dict.items():
        if k.startswith(prefix):
            k = k.replace(prefix, new_prefix)
        out[k] = v
    return out


def load_params_choose_layers(prefix: str, state_dict: dict, mapper: dict):
    """efficient_models/xvlm.py:38-51: keep layers {1,3,..,11} of a 12-layer checkpoint as layers {0..5}"""
    for k in list(state_dict.keys()):
        if k.startswith(prefix):
            new_k = None
            for i in mapper.keys():
                if k.startswith(f"{prefix}.{i}."):
                    new_k = k.replace(f"{prefix}.{i}.", f"{prefix}.{mapper[i]}.")
                    break
            if new_k:
                state_dict[new_k] = state_dict[k]
            del state_dict[k]
    return state_dict


# set by trainer.GDTrainer while it captures / replays a multi-GPU step as hipGraph segments: called INSTEAD of
# dist.all_gather(output_list, tensor) - it ends the running capture, issues the collective eagerly, starts the next one
GATHER_HOOK = None
_NO_BATCH_SELECT = bool(os.environ.get("EVLM_NO_BATCH_SELECT"))  # (A/B switch: the fusion batch built with cat / index_select)
_NO_FUSED_ITC = bool(os.environ.get("EVLM_NO_FUSED_ITC"))      # (A/B switch: the ITC loss as ~35 torch / HIP launches)


class AllGather(torch.autograd.Function):
    """efficient_models/xvlm.py:54-74: all_gather forward; backward keeps ONLY the local slice (no reduction)."""

    @staticmethod
    def forward(ctx, tensor, rank, world_size):
        output = [torch.empty_like(tensor) for _ in range(world_size)]
        src = tensor.contiguous()
        if GATHER_HOOK is not None:          # a trainer that captures the step in hipGraph SEGMENTS around its collectives
            GATHER_HOOK(output, src)
        else:
            log_collective("all_gather", src)
            dist.all_gather(output, src)
        ctx.rank = rank
        ctx.batch_size = tensor.shape[0]
        return torch.cat(output, 0)

    @staticmethod
    def backward(ctx, grad_output):
        return (grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None, None)


def allgather(tensor, rank=None, world_size=None):
    """single-process runs (no process group) degrade to identity, which is what world_size == 1 computes"""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    if world_size == 1 and not os.environ.get("EVLM_FORCE_REDUCE"):     # (forced: exercise the collective path on one GPU)
        return tensor
    return AllGather.apply(tensor, rank, world_size)


def build_mlp(input_dim, output_dim):
    """efficient_models/xvlm.py:77-83 (parameter container; evaluated by mlp_head_forward)"""
    return nn.Sequential(nn.Linear(input_dim, input_dim * 2), nn.LayerNorm(input_dim * 2), nn.GELU(),
                         nn.Linear(input_dim * 2, output_dim))


def mlp_head_forward(head, x):
    """Linear - LayerNorm - GELU - Linear through the HIP ops"""
    h = ops.linear(x, head[0].weight, head[0].bias)
    h = ops.layer_norm(h, head[1].weight, head[1].bias, head[1].eps)
    h = ops.gelu(h)
    return ops.linear(h, head[3].weight, head[3].bias)


def interpolate_pos_embed(pos_embed_checkpoint, num_patches, num_extra_tokens=1):
    """models/vit.py:222-247: bicubic resize of the patch position embeddings (checkpoint-load utility)"""
    embedding_size = pos_embed_checkpoint.shape[-1]
    orig_size = int((pos_embed_checkpoint.shape[-2] - num_extra_tokens) ** 0.5)
    new_size = int(num_patches ** 0.5)
    if orig_size != new_size:
        extra_tokens = pos_embed_checkpoint[:, :num_extra_tokens]
        pos_tokens = pos_embed_checkpoint[:, num_extra_tokens:]
        pos_tokens = pos_tokens.reshape(-1, orig_size, orig_size, embedding_size).permute(0, 3, 1, 2)
        pos_tokens = F.interpolate(pos_tokens, size=(new_size, new_size), mode="bicubic", align_corners=False)
        pos_tokens = pos_tokens.permute(0, 2, 3, 1).flatten(1, 2)
        return torch.cat((extra_tokens, pos_tokens), dim=1)
    return pos_embed_checkpoint


def build_vision_encoder(config, load_params=False):
    """efficient_models/xvlm.py:86-139 (CLIP-ViT branch; Swin/DeiT are used by no shipped config)"""
    num_patches = (config["image_res"] // config["patch_size"]) ** 2
    if not config.get("use_clip_vit", True):
        raise NotImplementedError("only use_clip_vit=True is on the distillation path (SURVEY.md §2 row 6)")
    vision_config = read_json(config["vision_config"])
    assert config["patch_size"] == vision_config["patch_size"]
    vision_width = vision_config["vision_width"]
    vision_encoder = CLIPVisionTransformer(image_size=config["image_res"], patch_size=vision_config["patch_size"],
                                           hidden_size=vision_config["vision_width"], hidden_act=vision_config["hidden_act"],
                                           num_attention_heads=vision_config["num_attention_heads"],
                                           attention_dropout=vision_config["attention_dropout"],
                                           intermediate_size=vision_config["intermediate_size"],
                                           num_hidden_layers=vision_config["num_hidden_layers"],
                                           local_attn_depth=vision_config["local_attn_depth"])
    if load_params:
        state_dict_orig = torch.load(vision_config["ckpt"], map_location="cpu")
        state_dict = {}
        for k, v in state_dict_orig.items():
            if k.startswith("vision_model."):
                k = k[13:]
                if k.startswith("embeddings."):
                    k = k[11:]
                    k = k.replace("patch_embedding.weight", "patch_embed.weight")
                    k = k.replace("position_embedding.weight", "pos_embed.weight")
                if k != "position_ids":
                    state_dict[k] = v
        pos = interpolate_pos_embed(state_dict["pos_embed.weight"].unsqueeze(dim=0), num_patches=num_patches, num_extra_tokens=1)
        state_dict["pos_embed.weight"] = pos.squeeze(dim=0)
        assert vision_config["num_hidden_layers"] in [6, 12], "param initialization not implemented"
        if vision_config["num_hidden_layers"] == 6:
            load_params_choose_layers("encoder.layers", state_dict, {1: 0, 3: 1, 5: 2, 7: 3, 9: 4, 11: 5})
        msg = vision_encoder.load_state_dict(state_dict, strict=False)
        print("### Load ViT: ", flush=True)
        print("missing_keys: ", msg.missing_keys)
        print("unexpected_keys: ", msg.unexpected_keys)
    return vision_encoder, vision_width


def build_text_encoder(config, vision_width, load_text_params=False, use_mlm_loss=False, config_text=None):
    """efficient_models/xvlm.py:142-180"""
    init_params = []
    if config_text is None:
        tc = config["text_encoder"]
        config_text = BertConfig.from_any(tc) if isinstance(tc, dict) else BertConfig.from_json_file(os.path.join(tc, "config.json"))
    else:
        config_text = BertConfig.from_any(config_text)
    config_text.num_hidden_layers = config["text_num_hidden_layers"] if "text_num_hidden_layers" in config else 12
    assert config_text.num_hidden_layers in [6, 12], "param initialization not implemented"
    config_text.fusion_layer = config_text.num_hidden_layers // 2
    config_text.encoder_width = vision_width
    if use_mlm_loss:
        if ("accelerator" in config.keys()) and (config["accelerator"]["FP16_OPT_LEVEL"] != "O0"):
            config_text.fp16 = True
        text_encoder = BertForMaskedLM(config=config_text)
        if load_text_params:
            path = os.path.join(config["text_encoder"], "pytorch_model.bin")
            print("### Initializing text encoder from ", path)
            state_dict = torch.load(path, map_location="cpu")
            if "roberta-base" in config["text_encoder"]:
                state_dict = load_params_change_prefix(state_dict, "roberta.", new_prefix="bert.")
            elif "bert-base-uncased" not in config["text_encoder"]:
                raise NotImplementedError
            if config_text.num_hidden_layers == 6:
                load_params_choose_layers("bert.encoder.layer", state_dict, {1: 0, 3: 1, 5: 2, 7: 3, 9: 4, 11: 5})
            msg = text_encoder.load_state_dict(state_dict, strict=False)
            print("missing_keys: ", msg.missing_keys, flush=True)
            print("unexpected_keys: ", msg.unexpected_keys, flush=True)
            init_params += [f"text_encoder.{k}" for k in msg.missing_keys]
    else:
        assert load_text_params is False
        text_encoder = BertModel(config=config_text, add_pooling_layer=False)
    return text_encoder, init_params


def load_pretrained(ckpt_rpath, config, is_eval=False, load_text=False):
    """efficient_models/xvlm.py:183-208 (CLIP-ViT branch)"""
    checkpoint = torch.load(ckpt_rpath, map_location="cpu")
    state_dict = checkpoint["model"] if "model" in checkpoint.keys() else checkpoint
    if is_eval:
        return state_dict
    num_patches = (config["image_res"] // config["patch_size"]) ** 2
    print("### Loading pretrained vision encoder", flush=True)
    state_dict.pop("vision_encoder.position_ids", None)
    pos = interpolate_pos_embed(state_dict["vision_encoder.pos_embed.weight"].unsqueeze(dim=0), num_patches=num_patches,
                                num_extra_tokens=1)
    state_dict["vision_encoder.pos_embed.weight"] = pos.squeeze(dim=0)
    if load_text:
        print("### Loading pretrained text encoder", flush=True)
        for key in list(state_dict.keys()):
            if key.startswith("text_encoder.") and "bert." in key:
                state_dict[key.replace("bert.", "")] = state_dict.pop(key)
    return state_dict


class XVLMBase(nn.Module):
    """efficient_models/xvlm.py:211-569"""

    def __init__(self, config=None, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                 use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False, config_text=None):
        super().__init__()
        self.init_params = []
        self.vision_encoder, vision_width = build_vision_encoder(config, load_params=load_vision_params)
        self.text_encoder, init_params = build_text_encoder(config, vision_width=vision_width,
                                                            load_text_params=load_text_params, use_mlm_loss=use_mlm_loss,
                                                            config_text=config_text)
        self.init_params.extend(init_params)
        self.num_text_layers = self.text_encoder.config.fusion_layer
        self.num_cross_layers = self.text_encoder.config.num_hidden_layers - self.num_text_layers
        self.vision_width = vision_width
        self.text_width = self.text_encoder.config.hidden_size
        if use_contrastive_loss:
            self.embed_dim = config["embed_dim"]
            self.vision_proj = nn.Linear(self.vision_width, self.embed_dim)
            self.text_proj = nn.Linear(self.text_width, self.embed_dim)
            self.init_params.extend(["vision_proj." + n for n, _ in self.vision_proj.named_parameters()])
            self.init_params.extend(["text_proj." + n for n, _ in self.text_proj.named_parameters()])
            self.temp = nn.Parameter(torch.ones([]) * config["temp"])
            self.init_params.extend(["temp"])
        if use_matching_loss:
            self.itm_head = build_mlp(input_dim=self.text_width, output_dim=2)
            self.init_params.extend(["itm_head." + n for n, _ in self.itm_head.named_parameters()])
        if use_bbox_loss:
            self.bbox_head = build_mlp(input_dim=self.text_width, output_dim=4)
            self.init_params.extend(["bbox_head." + n for n, _ in self.bbox_head.named_parameters()])
        named_parameters = set(n for n, _ in self.named_parameters())
        for n in set(self.init_params):
            if n not in named_parameters:
                print(f"warning: {n} not in named_parameters")
                self.init_params.remove(n)
        # test / replay hook: when set to a LongTensor [2B] (B image negatives then B text negatives) the
        # hard-negative indices are taken from it instead of torch.multinomial (the reference's RNG stream cannot be
        # reproduced on another device, SURVEY.md §7 "RNG parity").  Cleared after one use.
        self.injected_neg_idx = None
        self.last_neg_idx = None

    def load_pretrained(self, ckpt_rpath, config, is_eval=False):
        state_dict = load_pretrained(ckpt_rpath, config, is_eval=is_eval, load_text=True)
        msg = self.load_state_dict(state_dict, strict=False)
        print("load checkpoint from %s" % ckpt_rpath)
        print("missing_keys: ", [p for p in msg.missing_keys if "vision_encoder" not in p])
        print("unexpected_keys: ", msg.unexpected_keys)

    # ---- encoders -----------------------------------------------------------------------------
    def get_vision_embeds(self, image, image_atts=None, idx_to_group_img=None, output_attentions=None,
                          output_hidden_states=None, head_z=None, head_layer_z=None, mlp_z=None):
        """efficient_models/xvlm.py:262-301"""
        if idx_to_group_img is None:
            if not output_attentions and self.get_vision_embeds_returns_pair:
                image_embeds = self.vision_encoder(image, head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)[0]
                image_atts = ops.const_ones(image_embeds.size()[:-1], torch.long, image.device)
                return image_embeds, image_atts
            image_embeds, image_hidden_states, image_all_attentions = self.vision_encoder(
                image, output_attentions=output_attentions, output_hidden_states=output_hidden_states, head_z=head_z,
                head_layer_z=head_layer_z, mlp_z=mlp_z)
            image_atts = ops.const_ones(image_embeds.size()[:-1], torch.long, image.device)
            return image_embeds, image_atts, image_hidden_states, image_all_attentions
        if image_atts is None:
            image_embeds_fullatts = self.vision_encoder(image, head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)[0]
            image_embeds_fullatts = torch.index_select(image_embeds_fullatts, 0, idx_to_group_img.view(-1))
            image_atts = ops.const_ones(image_embeds_fullatts.size()[:-1], torch.long, image.device)
            return image_embeds_fullatts, image_atts
        assert image_atts.size(0) == idx_to_group_img.size(0)
        image_embeds, image_hidden_states, image_all_attentions, image_embeds_fullatts = self.vision_encoder(
            image, idx_to_group_img=idx_to_group_img, image_atts=image_atts, output_attentions=output_attentions,
            output_hidden_states=output_hidden_states, head_z=head_z, head_layer_z=head_layer_z)
        image_embeds_fullatts = torch.index_select(image_embeds_fullatts, 0, idx_to_group_img.view(-1))
        return image_embeds, image_atts, image_embeds_fullatts, image_hidden_states, image_all_attentions

    # efficient_models.XVLMBase returns (embeds, atts) when output_attentions is falsy; models.XVLMBase always returns
    # the 4-tuple.  models/xvlm.py flips this switch.
    get_vision_embeds_returns_pair = True

    def _text_core(self):
        return self.text_encoder.bert if hasattr(self.text_encoder, "bert") else self.text_encoder

    def get_text_embeds(self, text_ids, text_atts, output_attentions=None, output_hidden_states=None, head_z=None,
                        head_layer_z=None, mlp_z=None):
        """efficient_models/xvlm.py:303-313"""
        assert output_hidden_states == output_attentions
        outputs = self._text_core()(text_ids, attention_mask=text_atts, return_dict=True, mode="text",
                                    output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                                    head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)
        if output_attentions:
            return outputs.last_hidden_state, outputs.hidden_states, outputs.attentions
        return outputs.last_hidden_state

    def get_pair_embeds(self, image, text_ids, text_atts, vision_kw=None, text_kw=None, side_stream=None):
        """extension (inference): get_vision_embeds and get_text_embeds of one batch SIDE BY SIDE - the text encoder's ~45 small
        launches (30-token rows) on `side_stream`, forked from and joined back into the current stream, under the image
        encoder's GEMMs.  The two passes share nothing until the ITC features / the fusion layers (efficient_models/xvlm.py:
        262-313), so the results are those of the two calls in sequence, bit for bit; captured into a hipGraph the fork
        becomes two branches of the graph.  Without a stream (or on CPU tensors) it IS the two calls in sequence.
        Returns (image_embeds, image_atts, text_embeds)."""
        vision_kw, text_kw = vision_kw or {}, text_kw or {}
        if side_stream is None or not image.is_cuda:
            ie, ia = self.get_vision_embeds(image, **vision_kw)[:2]
            return ie, ia, self.get_text_embeds(text_ids, text_atts, **text_kw)
        cur = torch.cuda.current_stream()
        side_stream.wait_stream(cur)
        with torch.cuda.stream(side_stream):
            te = self.get_text_embeds(text_ids, text_atts, **text_kw)
        ie, ia = self.get_vision_embeds(image, **vision_kw)[:2]
        cur.wait_stream(side_stream)
        te.record_stream(cur)
        return ie, ia, te

    def get_cross_embeds(self, image_embeds, image_atts, text_ids=None, text_embeds=None, text_atts=None,
                         output_hidden_states=None, output_attentions=None, head_z=None, head_layer_z=None, mlp_z=None):
        """efficient_models/xvlm.py:315-373"""
        assert text_atts is not None
        assert output_attentions == output_hidden_states
        encoder = self._text_core()
        kw = dict(attention_mask=text_atts, encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                  output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=True,
                  head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)
        if text_embeds is not None:
            outputs = encoder(encoder_embeds=text_embeds, mode="fusion", **kw)
        elif text_ids is not None:
            outputs = encoder(text_ids, **kw)
        else:
            raise ValueError
        if not output_attentions:
            return outputs.last_hidden_state
        return outputs.last_hidden_state, outputs.hidden_states, outputs.attentions, outputs.cross_attentions

    def get_features(self, image_embeds=None, text_embeds=None):
        """efficient_models/xvlm.py:375-382"""
        # (measured, round 3: running these two 64-row heads and the normalisation in exact fp32 moves the heads' gradient
        # error against the fp32 oracle from 18 to 13-17 % only - the noise comes from the bf16 CLS rows upstream, amplified
        # by 1 / temp in the ITC logits - and costs 0.15 ms per step on the exact-fp32 GEMM kernel: not kept)
        def img():
            return ops.l2_normalize(ops.linear(image_embeds[:, 0, :], self.vision_proj.weight, self.vision_proj.bias))

        def txt():
            return ops.l2_normalize(ops.linear(text_embeds[:, 0, :], self.text_proj.weight, self.text_proj.bias))
        if image_embeds is None:
            return txt()
        if text_embeds is None:
            return img()
        return img(), txt()

    # ---- losses ---------------------------------------------------------------------------------
    def get_contrastive_loss(self, image_feat, text_feat, idx=None):
        """efficient_models/xvlm.py:384-416"""
        assert image_feat.size(-1) == self.embed_dim
        assert text_feat.size(-1) == self.embed_dim
        # similarity logits are formed in exact fp32 whatever the compute dtype ([B,256] x [B,256]: negligible cost)
        # ONE all-gather of [B, 2E] for both feature sets (latency-bound message: SURVEY.md 2.2); the slice-only backward of
        # the reference's two gathers (xvlm.py:54-74) is unchanged - it acts row-wise
        gathered = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("EVLM_FORCE_REDUCE"))
        world = dist.get_world_size() if gathered else 1
        # (the one-launch form holds a row block of the gathered coefficients in LDS: gathered batches of up to 4 096 rows)
        if (image_feat.is_cuda and not _NO_FUSED_ITC and image_feat.dtype == text_feat.dtype and self.embed_dim <= 256
                and self.embed_dim % 8 == 0 and image_feat.shape[0] * world <= 4096):
            # round 5: logits, both cross-entropies (soft labels when idx is given) and the whole backward in ONE launch each
            # way (ops.itc_loss) - it was ~35 launches of 4-27 us on the student's critical path
            both = allgather(torch.cat([image_feat, text_feat], dim=1)) if gathered else None
            group = None
            if idx is not None:
                idx = idx.view(-1, 1)
                assert idx.size(0) == image_feat.size(0)
                group = allgather(idx).view(-1)
            if gathered:
                loss, sim = ops.itc_loss(both, None, self.temp, group)
                B, r0 = image_feat.shape[0], dist.get_rank() * image_feat.shape[0]
                sim = sim[r0:r0 + B, r0:r0 + B]
            else:
                loss, sim = ops.itc_loss(image_feat, text_feat, self.temp, group)
            # this rank's block is what the ITM hard-negative sampler draws from.  WEAK references to the features: a strong
            # one would keep this forward's autograd graph - and the AccumulateGrad nodes of its parameters, with the stream
            # they were created on - alive into the next step, which a later hipGraph capture of the backward cannot join
            self._itc_sim = (weakref.ref(image_feat), weakref.ref(text_feat), sim.detach())
            return loss
        both = allgather(torch.cat([image_feat, text_feat], dim=1))
        image_feat_all = ops.cast(both[:, :self.embed_dim], torch.float32)
        text_feat_all = ops.cast(both[:, self.embed_dim:], torch.float32)
        logits = _matmul_nt(image_feat_all, text_feat_all).float() / self.temp
        logits_t = _matmul_nt(text_feat_all, image_feat_all).float() / self.temp
        bsz = image_feat_all.shape[0]
        if idx is None:
            labels = ops.const_tensor("arange", bsz, image_feat.device)
            loss_i2t = ops.cross_entropy(logits, labels)
            loss_t2i = ops.cross_entropy(logits_t, labels)
        else:
            idx = idx.view(-1, 1)
            assert idx.size(0) == image_feat.size(0)
            idx_all = allgather(idx)
            pos_idx = torch.eq(idx_all, idx_all.t()).float()
            labels = pos_idx / pos_idx.sum(1, keepdim=True)
            loss_i2t = -torch.sum(ops.log_softmax(logits) * labels, dim=1).mean()
            loss_t2i = -torch.sum(ops.log_softmax(logits_t) * labels, dim=1).mean()
        return (loss_i2t + loss_t2i) / 2

    @torch.no_grad()
    def _sample_negatives(self, image_feat, text_feat, idx):
        """efficient_models/xvlm.py:422-458, batched: one device-side sampling launch for both directions, no host syncs.
        A caller that set self._want_neg_layout gets, from the same launch, the batched fusion pass's row / image index
        vectors (ops.sample_negatives) in self._neg_layout = (all 2B draws, sel4, img4) - None when the draws were injected."""
        bs = image_feat.size(0)
        cached, self._itc_sim = getattr(self, "_itc_sim", None), None
        layout, self._want_neg_layout, self._neg_layout = getattr(self, "_want_neg_layout", False), False, None
        if self.injected_neg_idx is not None:
            neg = self.injected_neg_idx.to(image_feat.device).long()
            # consumed by ONE forward unless keep_injected_neg is set (parity tests of trainers that run warm-up steps and
            # capture graphs: every forward, captured ones included, then uses the same - device-resident - indices)
            self.injected_neg_idx = neg if getattr(self, "keep_injected_neg", False) else None
            assert neg.numel() == 2 * bs
            return neg[:bs], neg[bs:]
        if cached is not None and cached[0]() is image_feat and cached[1]() is text_feat:
            sim_i2t = cached[2]                  # (formed by the ITC loss kernel of this forward, same features)
        else:
            sim_i2t = _matmul_nt(image_feat.float(), text_feat.float())               # (the HIP fp32 GEMM, not a vendor one)
        # softmax(sim / temp) + 1e-5, positives zeroed, one categorical draw per row and per column: ONE launch
        if layout:
            self._neg_layout = ops.sample_negatives(sim_i2t, self.temp, None if idx is None else idx.view(-1), layout=True)
            neg = self._neg_layout[0]
        else:
            neg = ops.sample_negatives(sim_i2t, self.temp, None if idx is None else idx.view(-1))
        return neg[:bs], neg[bs:]

    def get_matching_loss(self, image_embeds, image_atts, image_feat, text_embeds, text_atts, text_feat, idx=None,
                          output_attentions=None, output_hidden_states=None, head_z=None, head_layer_z=None, mlp_z=None):
        """efficient_models/xvlm.py:418-490"""
        bs = image_embeds.size(0)
        batched = bool(output_hidden_states and self.batched_itm)
        self._neg_layout, self._want_neg_layout = None, batched and image_embeds.is_cuda
        img_neg, txt_neg = self._sample_negatives(image_feat, text_feat, idx)      # (tests replace this method: no new arguments)
        lay, self._want_neg_layout = self._neg_layout, False
        self.last_neg_idx = lay[0] if lay is not None else torch.cat([img_neg, txt_neg])
        if batched:
            return self._matching_loss_batched(image_embeds, image_atts, text_embeds, text_atts, img_neg, txt_neg,
                                               head_z, head_layer_z, mlp_z)
        image_embeds_neg = torch.index_select(image_embeds, 0, img_neg)
        image_atts_neg = torch.index_select(image_atts, 0, img_neg)
        text_embeds_neg = torch.index_select(text_embeds, 0, txt_neg)
        text_atts_neg = torch.index_select(text_atts, 0, txt_neg)
        text_embeds_all = torch.cat([text_embeds, text_embeds_neg], dim=0)
        text_atts_all = torch.cat([text_atts, text_atts_neg], dim=0)
        image_embeds_all = torch.cat([image_embeds_neg, image_embeds], dim=0)
        image_atts_all = torch.cat([image_atts_neg, image_atts], dim=0)
        zkw = dict(head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)
        if output_hidden_states:
            pos_last, pos_hs, pos_att, pos_catt = self.get_cross_embeds(
                image_embeds, image_atts, text_embeds=text_embeds, text_atts=text_atts,
                output_attentions=output_attentions, output_hidden_states=output_hidden_states, **zkw)
            neg_last, neg_hs, neg_att, neg_catt = self.get_cross_embeds(
                image_embeds_all, image_atts_all, text_embeds=text_embeds_all, text_atts=text_atts_all,
                output_attentions=output_attentions, output_hidden_states=output_hidden_states, **zkw)
        else:
            pos_last = self.get_cross_embeds(image_embeds, image_atts, text_embeds=text_embeds, text_atts=text_atts, **zkw)
            neg_last = self.get_cross_embeds(image_embeds_all, image_atts_all, text_embeds=text_embeds_all,
                                             text_atts=text_atts_all, **zkw)
        cls_rows = torch.cat([pos_last[:, 0, :], neg_last[:, 0, :]], dim=0)
        output = ops.join_grads(mlp_head_forward(self.itm_head, cls_rows))      # (this CE + the distillation KL: one gradient buffer)
        dev = image_embeds.device
        itm_labels = torch.cat([torch.ones(bs, dtype=torch.long, device=dev), torch.zeros(2 * bs, dtype=torch.long, device=dev)], dim=0)
        matching_loss = ops.cross_entropy(output, itm_labels)
        if not output_hidden_states:
            return matching_loss
        return {"loss": matching_loss, "pos_hidden_states": pos_hs, "neg_hidden_states": neg_hs, "pos_attentions": pos_att,
                "neg_attentions": neg_att, "pos_cross_attentions": pos_catt, "neg_cross_attentions": neg_catt, "logits": output}

    # One fusion pass over [positive (B) ; negative (2B)] rows instead of the reference's two (xvlm.py:460-476): row-wise
    # identical arithmetic; the hard-negative image copies are not materialised - their cross-attention K / V come from
    # the B distinct images through the attention kernels' batch index (projected, and differentiated, once per image).
    batched_itm = True

    def _matching_loss_batched(self, image_embeds, image_atts, text_embeds, text_atts, img_neg, txt_neg, head_z,
                               head_layer_z, mlp_z):
        bs, dev = image_embeds.size(0), image_embeds.device
        lay = getattr(self, "_neg_layout", None)
        ones = image_atts is ops.const_ones(image_atts.shape, image_atts.dtype, dev)       # (get_vision_embeds' all-ones mask)
        if (lay is not None and ones and not _NO_BATCH_SELECT and text_embeds.is_contiguous() and text_atts.is_contiguous()
                and (text_embeds[0].numel() * text_embeds.element_size()) % 16 == 0 and text_embeds[0].numel() % 8 == 0
                and (text_atts[0].numel() * text_atts.element_size()) % 16 == 0):
            # round 5: the sampling launch wrote this layout's row / image indices (the first 3B entries of its 4-block
            # vectors: pos | (text, image_neg) | (text_neg, image)); one selection launch each for embeddings and masks
            sel3, img3 = lay[1][:3 * bs], lay[2][:3 * bs]
            txt_all, atts_all = ops.select_batches(text_embeds, sel3), ops.select_batches(text_atts, sel3)
            enc_mask = ops.const_ones((3 * bs,) + tuple(image_atts.shape[1:]), image_atts.dtype, dev)
        else:
            ar = ops.const_tensor("arange", bs, dev)
            txt_all = torch.cat([text_embeds, text_embeds, torch.index_select(text_embeds, 0, txt_neg)], 0)
            atts_all = torch.cat([text_atts, text_atts, torch.index_select(text_atts, 0, txt_neg)], 0)
            img_index = torch.cat([ar, img_neg, ar], 0)            # pos | (text, image_neg) | (text_neg, image)
            enc_mask, img3 = torch.index_select(image_atts, 0, img_index), img_index.to(torch.int32)
        f = self._text_core()(encoder_embeds=txt_all, attention_mask=atts_all, encoder_hidden_states=image_embeds,
                              encoder_attention_mask=enc_mask,
                              encoder_batch_index=img3, return_dict=True, mode="fusion", output_attentions=True,
                              output_hidden_states=True, head_z=head_z, head_layer_z=head_layer_z, mlp_z=mlp_z)
        two = lambda tup: tuple(zip(*[torch.split(x, [bs, 2 * bs], 0) if x is not None else (None, None) for x in tup]))
        (pos_hs, neg_hs), (pos_att, neg_att), (pos_catt, neg_catt) = two(f.hidden_states), two(f.attentions), \
            two(f.cross_attentions)
        output = ops.join_grads(mlp_head_forward(self.itm_head, f.last_hidden_state[:, 0, :]))
        itm_labels = ops.const_tensor("itm_labels", bs, dev)
        # extension: the same lists as row ranges of the UN-split pass outputs (ops.RowSlice) - the distillation losses take
        # these, so each batched tensor receives ONE gradient buffer instead of autograd concatenating a positive and a
        # negative piece (the cross-attention maps of a 384 x 384 step are 80 MB each: six 54-us cats per step)
        rs = lambda tup, r0, r1: [ops.RowSlice(x, r0, r1) if x is not None else None for x in tup]
        batched = {"itm_pos_hidden_states": rs(f.hidden_states, 0, bs), "itm_neg_hidden_states": rs(f.hidden_states, bs, 3 * bs),
                   "itm_pos_attentions": rs(f.attentions, 0, bs), "itm_neg_attentions": rs(f.attentions, bs, 3 * bs),
                   "itm_pos_cross_attentions": rs(f.cross_attentions, 0, bs),
                   "itm_neg_cross_attentions": rs(f.cross_attentions, bs, 3 * bs)}
        return {"loss": ops.cross_entropy(output, itm_labels), "pos_hidden_states": pos_hs, "neg_hidden_states": neg_hs,
                "pos_attentions": pos_att, "neg_attentions": neg_att, "pos_cross_attentions": pos_catt,
                "neg_cross_attentions": neg_catt, "logits": output, "batched": batched}

    def get_mlm_loss(self, text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids,
                     output_attentions=None, output_hidden_states=None, head_z=None, head_layer_z=None, mlp_z=None):
        """efficient_models/xvlm.py:492-518"""
        assert output_hidden_states == output_attentions
        outputs = self.text_encoder(text_ids_masked, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                                    encoder_attention_mask=image_atts, return_dict=True, labels=masked_ids,
                                    masked_pos=masked_pos, output_attentions=output_attentions,
                                    output_hidden_states=output_hidden_states, head_z=head_z, head_layer_z=head_layer_z,
                                    mlp_z=mlp_z)
        if not output_attentions:
            return outputs.loss
        return outputs.loss, outputs.logits, outputs.hidden_states, outputs.attentions, outputs.cross_attentions

    def predict_bbox(self, image_embeds, text_embeds, text_atts, output_attentions=None, output_hidden_states=None,
                     head_z=None, head_layer_z=None, mlp_z=None):
        """efficient_models/xvlm.py:520-542: fusion pass over the FULL-attention image embeddings, bbox head on [CLS],
        sigmoid -> (cx, cy, w, h).  Returns (coord,) or (coord, hidden_states, attentions, cross_attentions)."""
        assert image_embeds.size(0) == text_embeds.size(0)
        ones = torch.ones(image_embeds.shape[:2], device=image_embeds.device)
        outputs = self.get_cross_embeds(image_embeds, ones, text_embeds=text_embeds, text_atts=text_atts,
                                        output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                                        head_z=head_z, mlp_z=mlp_z)
        last = outputs[0] if output_attentions else outputs
        output_coord = self.bbox_coord(last[:, 0, :])
        return ((output_coord,) + tuple(outputs[1:])) if output_attentions else (output_coord,)

    def bbox_coord(self, cls_rows):
        """bbox_head (HIP GEMMs / LayerNorm / GELU) + sigmoid in fp32 (<= 128 x 4 values)"""
        return torch.sigmoid(ops.cast(mlp_head_forward(self.bbox_head, cls_rows), torch.float32))

    def get_bbox_loss(self, output_coord, target_bbox, is_image=None):
        """efficient_models/xvlm.py:544-569: L1 + (1 - GIoU), whole-image rows (is_image = 1) masked out of both sums.
        The reference's degenerate-box early-out (:553-556, a host-side `if`) is a device-side select here, so the step
        stays free of host synchronisation."""
        from ..models import box_ops
        output_coord = output_coord.float()
        target_bbox = target_bbox.to(output_coord.dtype)
        loss_bbox = (output_coord - target_bbox).abs()
        boxes1 = box_ops.box_cxcywh_to_xyxy(output_coord)
        boxes2 = box_ops.box_cxcywh_to_xyxy(target_bbox)
        degenerate = (boxes1[:, 2:] < boxes1[:, :2]).any() | (boxes2[:, 2:] < boxes2[:, :2]).any()
        loss_giou = 1 - box_ops.generalized_box_iou_rowwise(boxes1, boxes2)
        loss_giou = torch.where(degenerate, torch.zeros_like(loss_giou), loss_giou)
        if is_image is None:
            num_boxes = target_bbox.size(0)
        else:
            keep = (1 - is_image).to(output_coord.dtype)
            num_boxes = keep.sum()
            loss_bbox = loss_bbox * keep.view(-1, 1)
            loss_giou = loss_giou * keep
        return loss_bbox.sum() / num_boxes, loss_giou.sum() / num_boxes


def _matmul_nt(a, b):
    """a [m,k] @ b[n,k]^T with gradients to both, through the HIP GEMM (ITC similarity matrices)"""
    return _MatmulNT.apply(a, b)


class _MatmulNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        from .._lib import dt
        ac, bc = a.contiguous(), b.contiguous()
        m, k = ac.shape
        n = bc.shape[0]
        ldc = (n + 7) // 8 * 8
        cbuf = torch.empty((m, ldc), dtype=ac.dtype, device=a.device) if ldc == n else ops.zeros_small((m, ldc), ac.dtype, a.device)
        ops._gemm(dt(ac), ac, bc, cbuf, m, n, k, k, k, ldc)
        ctx.save_for_backward(ac, bc)
        return cbuf[:, :n]

    @staticmethod
    def backward(ctx, dc):
        from .._lib import dt
        ac, bc = ctx.saved_tensors
        m, k = ac.shape
        n = bc.shape[0]
        ldd = (n + 7) // 8 * 8
        if ldd == n and dc.is_contiguous() and dc.dtype == ac.dtype:
            d = dc
        else:
            d = ops.zeros_small((m, ldd), ac.dtype, ac.device)
            d[:, :n].copy_(dc)
        da = torch.empty_like(ac)
        db = torch.empty_like(bc)
        ops._gemm(dt(ac), d, bc, da, m, k, n, ldd, k, k, p_trans=0, q_trans=1)       # dA = dC B
        ops._gemm(dt(ac), d, ac, db, n, k, m, ldd, k, k, p_trans=1, q_trans=1)       # dB = dC^T A
        return da, db
