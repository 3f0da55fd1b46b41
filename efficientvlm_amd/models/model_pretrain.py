"""GD / pre-training model — drop-in for models/model_pretrain.py:XVLM (reference :5-82), the object
GeneralDistill.py builds for both student and teacher."""
import os

import torch

from .xvlm import XVLMBase


from .. import ops
from ..efficient_models.xvlm import mlp_head_forward

_NO_BATCH_SELECT = bool(os.environ.get("EVLM_NO_BATCH_SELECT"))      # (A/B switch: the fusion batch built with cat / index_select)


class XVLM(XVLMBase):
    def __init__(self, config):
        # the reference hard-codes load_vision_params=load_text_params=True (model_pretrain.py:7-8); a config may set
        # 'load_params': False to build from random init without the CLIP / BERT checkpoint files (bench, tests).
        load = bool(config.get("load_params", True))
        super().__init__(config, load_vision_params=load, load_text_params=load, use_contrastive_loss=True,
                         use_matching_loss=True, use_mlm_loss=True, use_bbox_loss=True, config_text=None)

    # One batched pass instead of the reference's four: the text layers run once on [text ; masked text]
    # (2B rows) and the fusion layers once on [ITM-pos ; ITM-neg (2B) ; MLM] (4B rows); the cross-attention K/V projection
    # of each image is computed once per fusion layer and shared through a batch index instead of being recomputed for
    # the gathered hard-negative copies.  Row-wise identical arithmetic, 3-4x larger GEMMs, ~1/3 of the launches.
    batched_passes = True
    on_vision_grad = None      # optional callback: fired (tensor hook) when backward has produced d(loss)/d(image_embeds)
    # optional callback(name) at fixed points of the batched forward - "vision_done" (image encoder finished), "text_done"
    # (text layers finished, ITC / fusion passes next): a trainer forks side-stream work (the pipelined teacher) there
    phase_hook = None
    text_stream = None         # optional torch.cuda.Stream: see _forward_batched
    kd_fork = None             # set by _forward_batched (with text_stream): event behind the fusion pass
    # extension (False = the reference's behaviour): a frozen TEACHER's task losses are never read by the distillation
    # loss (GeneralDistill.py:300-376 uses its hidden states, attention maps and logits only); with this set the batched
    # forward skips ITC / ITM / MLM cross-entropies - and with them the ITC feature all-gather, the teacher forward's only
    # collective, so that forward can be captured into a hipGraph on multi-GPU runs too.  `loss` is then an empty dict.
    skip_task_losses = False

    def forward(self, image, text_ids, text_atts, text_ids_masked=None, masked_pos=None, masked_ids=None, image_atts=None,
                idx_to_group_img=None, target_bbox=None, is_image=None, ret_bbox_loss=False, output_attentions=None,
                output_hidden_states=None):
        assert output_attentions == output_hidden_states
        region = (image_atts, idx_to_group_img, target_bbox, is_image) if ret_bbox_loss else None
        if self.batched_passes and output_attentions:
            return self._forward_batched(image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, region)
        if ret_bbox_loss:                                                                 # model_pretrain.py:16-18
            assert output_attentions, "the reference's region branch unpacks the 5-tuple of the output_attentions path"
            image_embeds, image_atts, image_embeds_fullatts, image_hidden_states, image_attentions = self.get_vision_embeds(
                image, image_atts=image_atts, idx_to_group_img=idx_to_group_img, output_attentions=output_attentions,
                output_hidden_states=output_hidden_states)
        else:
            out = self.get_vision_embeds(image, output_attentions=output_attentions, output_hidden_states=output_hidden_states)
            image_embeds, image_atts, image_hidden_states, image_attentions = out
        t = self.get_text_embeds(text_ids, text_atts, output_attentions=output_attentions, output_hidden_states=output_hidden_states)
        text_embeds, text_hidden_states, text_attentions = t if output_attentions else (t, None, None)
        hidden_dict = {"image_hidden_states": image_hidden_states, "text_hidden_states": text_hidden_states}
        attention_dict = {"image_attentions": image_attentions, "text_attentions": text_attentions}
        cross_attention_dict, logits_dict = {}, {}
        with torch.no_grad():
            self.temp.clamp_(0.001, 0.5)
        image_feat, text_feat = self.get_features(image_embeds, text_embeds)
        loss_itc = self.get_contrastive_loss(image_feat, text_feat)
        itm = self.get_matching_loss(image_embeds, image_atts, image_feat, text_embeds, text_atts, text_feat,
                                     output_attentions=output_attentions, output_hidden_states=output_hidden_states)
        mlm = self.get_mlm_loss(text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids,
                                output_attentions=output_attentions, output_hidden_states=output_hidden_states)
        if not output_attentions:
            return {"loss": {"loss_itc": loss_itc, "loss_itm": itm, "loss_mlm": mlm}}
        hidden_dict["itm_pos_hidden_states"] = itm["pos_hidden_states"]
        hidden_dict["itm_neg_hidden_states"] = itm["neg_hidden_states"]
        attention_dict["itm_pos_attentions"] = itm["pos_attentions"]
        attention_dict["itm_neg_attentions"] = itm["neg_attentions"]
        cross_attention_dict["itm_pos_cross_attentions"] = itm["pos_cross_attentions"]
        cross_attention_dict["itm_neg_cross_attentions"] = itm["neg_cross_attentions"]
        logits_dict["itm_head_logits"] = itm["logits"]
        hidden_dict["mlm_hidden_states"] = mlm[2]
        attention_dict["mlm_attentions"] = mlm[3]
        logits_dict["mlm_logits"] = mlm[1]
        cross_attention_dict["mlm_cross_attentions"] = mlm[4]
        loss = {"loss_itc": loss_itc, "loss_itm": itm["loss"], "loss_mlm": mlm[0]}
        if ret_bbox_loss:                                                                 # model_pretrain.py:62-74
            bbox_output = self.predict_bbox(image_embeds_fullatts, text_embeds, text_atts,
                                            output_attentions=output_attentions, output_hidden_states=output_hidden_states)
            loss["loss_bbox"], loss["loss_giou"] = self.get_bbox_loss(bbox_output[0], target_bbox, is_image=is_image)
            hidden_dict["bbox_hidden_states"], attention_dict["bbox_attentions"] = bbox_output[1], bbox_output[2]
            cross_attention_dict["bbox_cross_attentions"] = bbox_output[3]
            self.last_output_coord = bbox_output[0].detach()     # (never keep the autograd graph alive)
        return {"loss": loss, "hidden_dict": hidden_dict, "attention_dict": attention_dict,
                "cross_attention_dict": cross_attention_dict, "logits_dict": logits_dict}

    def _forward_batched(self, image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, region=None):
        """the batched forward run to completion, `phase_hook` called at its phase points"""
        gen = self._forward_batched_gen(image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, region)
        try:
            while True:
                name = next(gen)
                if self.phase_hook is not None:
                    self.phase_hook(name)
        except StopIteration as done:
            return done.value

    def forward_phases(self, image, text_ids, text_atts, text_ids_masked=None, masked_pos=None, masked_ids=None,
                       image_atts=None, idx_to_group_img=None, target_bbox=None, is_image=None, ret_bbox_loss=False,
                       output_attentions=True, output_hidden_states=True):
        """extension: forward() as a GENERATOR over the phases of the batched forward - it yields "vision_done" (image
        encoder finished) and "text_done" (text layers finished) and returns forward()'s dict (StopIteration.value).  A
        trainer resumes it where it wants the rest issued: the pipelined teacher's image encoder in one hipGraph segment of
        the multi-GPU step, its text / fusion passes in the next.  Each resume runs under the CALLER's grad mode, autocast
        state and current stream."""
        assert self.batched_passes and output_attentions and output_hidden_states
        region = (image_atts, idx_to_group_img, target_bbox, is_image) if ret_bbox_loss else None
        return self._forward_batched_gen(image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, region)

    def _forward_batched_gen(self, image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, region=None):
        """same outputs as the pass-by-pass forward above (reference model_pretrain.py:11-82), batched as described at
        `batched_passes`.  region = (image_atts [R, N], idx_to_group_img [R], target_bbox [R, 4], is_image [R] | None)
        for a REGION batch: the image encoder yields R region-masked embeddings + the full-attention embeddings of the
        n_img images, ITC / ITM / MLM use the former with the region patch masks, and the bbox pass is a fifth block of
        fusion rows that cross-attends to the full-attention embeddings through the same batch index."""
        B = text_ids.shape[0]
        dev = image.device
        core = self._text_core()

        atts2 = torch.cat([text_atts, text_atts], 0)

        def text_pass():      # text layers 0..F-1 on [text_ids ; text_ids_masked]
            return core(torch.cat([text_ids, text_ids_masked], 0), attention_mask=atts2,
                        return_dict=True, mode="text", output_attentions=True, output_hidden_states=True)
        # extension: the text pass is independent of the image encoder until the ITC features - with `text_stream` set
        # (single-GPU trainers) it is issued on that stream BESIDE the image encoder: its ~25 small FORWARD launches share
        # the chip with the ViT's large products.  (Autograd runs a node on its forward's stream, so the backward of the
        # text pass is issued on that stream too - but its nodes were created before the ViT's and the engine runs later
        # nodes first, so they reach the device behind the whole ViT backward: no overlap there.  The trainer joins the
        # stream explicitly before it reads the gradients: GDTrainer._join_text_stream.)
        side = self.text_stream if (self.text_stream is not None and image.is_cuda) else None
        t = None
        if side is not None:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                t = text_pass()
        if region is None:
            image_embeds, image_atts, image_hidden_states, image_attentions = self.get_vision_embeds(
                image, output_attentions=True, output_hidden_states=True)
            enc_states, enc_atts = image_embeds, image_atts
        else:
            image_atts, idx_to_group_img, target_bbox, is_image = region
            assert image_atts.size(0) == idx_to_group_img.size(0) == B
            image_embeds, image_hidden_states, image_attentions, full = self.vision_encoder(
                image, idx_to_group_img=idx_to_group_img, image_atts=image_atts, output_attentions=True,
                output_hidden_states=True)
            # cross-attention sources: rows 0..R-1 the region embeddings, rows R.. the full-attention ones
            enc_states = torch.cat([image_embeds, full], 0)
            enc_atts = torch.cat([image_atts, torch.ones(full.shape[:2], dtype=image_atts.dtype, device=dev)], 0)
        if self.on_vision_grad is not None and image_embeds.requires_grad:
            cb = self.on_vision_grad
            image_embeds.register_hook(lambda grad: (cb(), grad)[1])
        yield "vision_done"
        if side is not None:
            cur.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                for x in list(t.hidden_states) + [a for a in t.attentions if a is not None] + [t.last_hidden_state]:
                    x.record_stream(cur)          # allocated on the side stream, consumed on this one
        else:
            t = text_pass()
        # torch.split, not slices: its backward is ONE concatenation per tensor (a slice's is zero-fill + copy + add)
        halves = lambda tup: tuple(zip(*[torch.split(x, [B, B], 0) if x is not None else (None, None) for x in tup]))
        text_hidden_states, mlm_text_hidden = halves(t.hidden_states)
        if t.hidden_states[-1] is t.last_hidden_state:
            text_embeds, mlm_text = text_hidden_states[-1], mlm_text_hidden[-1]
        else:
            text_embeds, mlm_text = torch.split(t.last_hidden_state, [B, B], 0)
        text_attentions, mlm_text_attentions = halves(t.attentions)
        yield "text_done"
        with torch.no_grad():
            self.temp.clamp_(0.001, 0.5)
        image_feat, text_feat = self.get_features(image_embeds, text_embeds)
        skip = self.skip_task_losses and not torch.is_grad_enabled()
        loss_itc = None if skip else self.get_contrastive_loss(image_feat, text_feat)
        fast = (region is None and image.is_cuda and not _NO_BATCH_SELECT and t.last_hidden_state.is_contiguous()
                and t.last_hidden_state.shape[0] == 2 * B and (t.last_hidden_state[0].numel() * t.last_hidden_state.element_size()) % 16 == 0
                and t.last_hidden_state[0].numel() % 8 == 0 and (atts2[0].numel() * atts2.element_size()) % 16 == 0)
        self._neg_layout, self._want_neg_layout = None, fast
        img_neg, txt_neg = self._sample_negatives(image_feat, text_feat, None)     # (tests replace this method: no new arguments)
        lay, self._want_neg_layout = self._neg_layout, False
        self.last_neg_idx = lay[0] if lay is not None else torch.cat([img_neg, txt_neg])
        # fusion layers on [pos (B) ; neg (2B: text|text_neg x img_neg|img) ; mlm (B)]
        sizes = [B, 2 * B, B]
        if lay is not None:
            # round 5: the sampling launch wrote the row / image indices of this layout itself; ONE selection launch builds the
            # embeddings (and one the masks) from the text pass's un-split [text ; masked text] output - it was two index_select,
            # five cat and a cast, and in backward a zero-fill, an atomic index_add and three adds over [B, L, d]
            _, sel4, img4 = lay
            txt_all = ops.select_batches(t.last_hidden_state, sel4)
            atts_all = ops.select_batches(atts2, sel4)
            img_index = None
        else:
            ar = ops.const_tensor("arange", B, dev)
            txt_all = torch.cat([text_embeds, text_embeds, torch.index_select(text_embeds, 0, txt_neg), mlm_text], 0)
            atts_all = torch.cat([text_atts, text_atts, torch.index_select(text_atts, 0, txt_neg), text_atts], 0)
            img_index = torch.cat([ar, img_neg, ar, ar], 0)
            if region is not None:                      # + bbox rows: the text again, attending to the un-masked image
                txt_all = torch.cat([txt_all, text_embeds], 0)
                atts_all = torch.cat([atts_all, text_atts], 0)
                img_index = torch.cat([img_index, B + idx_to_group_img.view(-1)], 0)
                sizes.append(B)
            img4 = img_index.to(torch.int32)             # (cast once here, not in every cross-attention)
        fkw = dict(encoder_embeds=txt_all, attention_mask=atts_all, encoder_hidden_states=enc_states,
                   # (a general batch attends to every image token - get_vision_embeds' all-ones mask: no mask is built)
                   encoder_attention_mask=None if region is None else torch.index_select(enc_atts, 0, img_index),
                   encoder_batch_index=img4,
                   return_dict=True, mode="fusion", output_attentions=True, output_hidden_states=True)
        if hasattr(core, "forward_gen"):
            # (round 6) the fusion pass LAYER BY LAYER: "fusion_layer_<i>" behind layer i - a trainer that spreads the frozen
            # teacher over several hipGraph segments resumes it a few layers at a time (trainer._capture_segments, `late`)
            gen = core.forward_gen(**fkw)
            try:
                while True:
                    ph = next(gen)
                    yield "fusion_layer_%d" % ph[1]
            except StopIteration as done:
                f = done.value
        else:
            f = core(**fkw)
        yield "fusion_done"
        # (the hidden-state / attention-map distillation terms depend on nothing past this point: a trainer that runs them
        # on the side stream - distill.kd_terms - forks from HERE, beside the task heads below)
        self.kd_fork = None
        if side is not None and not os.environ.get("EVLM_NO_KD_STREAM"):
            self.kd_fork = torch.cuda.Event()
            self.kd_fork.record(torch.cuda.current_stream())
        thirds = lambda tup: tuple(zip(*[torch.split(x, sizes, 0) if x is not None else (None,) * len(sizes)
                                         for x in tup]))                                       # pos | neg | mlm [| bbox]
        f_hid, f_att, f_cross = thirds(f.hidden_states), thirds(f.attentions), thirds(f.cross_attentions)
        last = f.last_hidden_state
        # (join_grads: the hard-label CE here and the distillation KL of distill.kd_terms sum their gradients into ONE buffer)
        itm_logits = ops.join_grads(mlp_head_forward(self.itm_head, last[:3 * B, 0, :]))
        itm_labels = ops.const_tensor("itm_labels", B, dev)
        loss_itm = None if skip else ops.cross_entropy(itm_logits, itm_labels)
        # MLM head on the masked positions of the last quarter
        enc = self.text_encoder
        mlm_last = f_hid[2][-1] if f.hidden_states[-1] is last else last[3 * B:4 * B]
        mlm_seq = enc.gather_seq_out_by_pos(mlm_last, masked_pos)
        mlm_logits = ops.join_grads(enc.cls(mlm_seq))
        loss_mlm = None if skip else ops.cross_entropy(mlm_logits, masked_ids.reshape(-1))
        nF = len(t.attentions)
        hidden_dict = {"image_hidden_states": image_hidden_states, "text_hidden_states": text_hidden_states,
                       "itm_pos_hidden_states": f_hid[0], "itm_neg_hidden_states": f_hid[1],
                       "mlm_hidden_states": tuple(mlm_text_hidden[:nF]) + f_hid[2]}
        attention_dict = {"image_attentions": image_attentions, "text_attentions": text_attentions,
                          "itm_pos_attentions": f_att[0], "itm_neg_attentions": f_att[1],
                          "mlm_attentions": tuple(mlm_text_attentions) + f_att[2]}
        cross_attention_dict = {"itm_pos_cross_attentions": f_cross[0], "itm_neg_cross_attentions": f_cross[1],
                                "mlm_cross_attentions": f_cross[2]}
        logits_dict = {"itm_head_logits": itm_logits, "mlm_logits": mlm_logits}
        loss = {} if skip else {"loss_itc": loss_itc, "loss_itm": loss_itm, "loss_mlm": loss_mlm}
        # extension: the same lists as row ranges of the UN-split batched tensors (keys of hidden_dict / attention_dict):
        # the distillation losses take these, so their gradients reach each batched tensor as one buffer instead of
        # autograd concatenating per-chunk gradients (distill.kd_terms, ops.RowSlice)
        rs = lambda tup, r0, r1: [ops.RowSlice(x, r0, r1) if x is not None else None for x in tup]
        batched = {"text_hidden_states": rs(t.hidden_states, 0, B), "text_attentions": rs(t.attentions, 0, B),
                   "itm_pos_hidden_states": rs(f.hidden_states, 0, B), "itm_pos_attentions": rs(f.attentions, 0, B),
                   "itm_neg_hidden_states": rs(f.hidden_states, B, 3 * B), "itm_neg_attentions": rs(f.attentions, B, 3 * B),
                   "mlm_hidden_states": rs(t.hidden_states[:nF], B, 2 * B) + rs(f.hidden_states, 3 * B, 4 * B),
                   "mlm_attentions": rs(t.attentions, B, 2 * B) + rs(f.attentions, 3 * B, 4 * B)}
        if region is not None:                                                            # model_pretrain.py:62-74
            coord = self.bbox_coord(last[4 * B:, 0, :])
            if not skip:
                loss["loss_bbox"], loss["loss_giou"] = self.get_bbox_loss(coord, target_bbox, is_image=is_image)
            hidden_dict["bbox_hidden_states"], attention_dict["bbox_attentions"] = f_hid[3], f_att[3]
            cross_attention_dict["bbox_cross_attentions"] = f_cross[3]
            self.last_output_coord = coord.detach()
        return {"loss": loss, "hidden_dict": hidden_dict, "attention_dict": attention_dict,
                "cross_attention_dict": cross_attention_dict, "logits_dict": logits_dict, "batched": batched}
