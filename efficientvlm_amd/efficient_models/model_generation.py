"""VQA student with L0 gates — drop-in for efficient_models/model_generation.py:EffXVLMForVQA (reference :22-187, the
`train=True` branch that Eff_VQA.py:95-99 drives): image encoder, question encoder (text + fusion layers in one
multi_modal pass), causal answer decoder (BertLMHeadModel whose every layer cross-attends to the question states), the
per-answer weighted LM loss.  Answer ranking at inference (`train=False`, rank_answer) is evaluation code outside the
training path and raises.

MI355X notes: the k[b] candidate answers of question b attend to ONE copy of its encoder states (K/V projected once per
question, shared through the attention kernels' batch index) instead of the reference's k-fold repeated tensor; the
shifted-logits copy and the per-token loss vector are never materialised (evlm_ce_weighted over the full logits)."""
import copy

import torch

from .. import ops
from .eff_bert import BertLMHeadModel
from .generation_l0_module import VQAL0Module
from .xvlm import XVLMBase, load_pretrained


def _decoder_config(config, config_enc):
    """model_generation.py:36-40: decoder = as many layers as there are fusion layers, ALL with cross-attention over
    hidden_size-wide encoder states"""
    config_dec = copy.deepcopy(config_enc)
    config_dec.encoder_width = config_enc.hidden_size
    config_dec.fusion_layer = 0
    config_dec.num_hidden_layers = config["num_dec_layers"]
    return config_dec


def remap_vqa_checkpoint(state_dict, num_text_layers, same_width):
    """model_generation.py:57-91 (non-eval load): `bert.` prefixes dropped for the encoder; the text encoder's FUSION layers
    (index >= num_text_layers) and its non-layer tensors are MOVED (not copied) to `text_decoder.*`, layer indices
    re-based at 0; cross-attention key / value weights are left out when the widths differ."""
    for key in list(state_dict.keys()):
        if "bert." in key:
            state_dict[key.replace("bert.", "")] = state_dict[key]
        if "text_encoder." in key:
            if "layer." in key:
                parts = key.split(".")
                layer_num = int(parts[4])
                if layer_num < num_text_layers:
                    del state_dict[key]
                    continue
                if (not same_width) and ("crossattention.self.key" in key or "crossattention.self.value" in key):
                    del state_dict[key]
                    continue
                parts[4] = str(layer_num - num_text_layers)
                moved = ".".join(parts)
            else:
                moved = key
            state_dict[moved.replace("text_encoder", "text_decoder")] = state_dict[key]
            del state_dict[key]
    return state_dict


class _VQABase(XVLMBase):
    """the structure and train-branch forward shared by the student (gates) and the teacher (no gates)"""

    def _build(self, config):
        assert isinstance(config["pad_token_id"], int)
        self.pad_token_id = config["pad_token_id"]
        config_enc = self.text_encoder.config
        self.num_text_layers = config_enc.fusion_layer
        self.num_cross_layers = config_enc.num_hidden_layers - config_enc.fusion_layer
        assert config["num_dec_layers"] == self.num_cross_layers, "initialization not implemented"
        self.cross_encoder_width = config_enc.encoder_width          # i.e. vision_width
        self.dec_encoder_width = config_enc.hidden_size
        self.text_decoder = BertLMHeadModel(config=_decoder_config(config, config_enc))
        if self.dec_encoder_width != self.cross_encoder_width:
            self.init_params = ["text_decoder." + n for n, _ in self.text_decoder.named_parameters()
                                if ("crossattention.self.key" in n) or ("crossattention.self.value" in n)]
        else:
            self.init_params = []

    def load_pretrained(self, ckpt_rpath, config, is_eval=False):
        if is_eval:
            state_dict = load_pretrained(ckpt_rpath, config, is_eval=True)
        else:
            state_dict = load_pretrained(ckpt_rpath, config, load_text=False)
            print("### Loading pretrained text encoder", flush=True)
            remap_vqa_checkpoint(state_dict, self.num_text_layers, self.dec_encoder_width == self.cross_encoder_width)
        msg = self.load_state_dict(state_dict, strict=False)
        print("load checkpoint from %s" % ckpt_rpath)
        print("missing_keys: ", [p for p in msg.missing_keys if "vision_encoder" not in p])
        print("unexpected_keys: ", msg.unexpected_keys)

    def _train_forward(self, image, quesiton, answer, k, weights, zs, output_attentions, output_hidden_states):
        z = (lambda name: zs[name]) if zs is not None else (lambda name: None)
        if output_attentions:
            image_embeds, image_hidden_states, image_attentions = self.vision_encoder(
                image, output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                head_z=z("vision_head_z"), mlp_z=z("vision_intermediate_z"))
        else:
            image_embeds = self.vision_encoder(image, head_z=z("vision_head_z"), mlp_z=z("vision_intermediate_z"))[0]
        image_atts = ops.const_ones(image_embeds.size()[:-1], torch.long, image.device)
        # k: number of answers per question; weights: weight of each answer          (model_generation.py:110-113)
        answer_targets = answer.input_ids.masked_fill(answer.input_ids == self.pad_token_id, -100)
        enc_hz = torch.cat((zs["text_head_z"], zs["cross_head_z"]), dim=0) if zs is not None else None      # :123-124
        enc_mz = torch.cat((zs["text_intermediate_z"], zs["cross_intermediate_z"]), dim=0) if zs is not None else None
        question_output = self.text_encoder(quesiton.input_ids, attention_mask=quesiton.attention_mask,
                                            encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                                            return_dict=True, output_attentions=output_attentions,
                                            output_hidden_states=output_hidden_states, head_z=enc_hz, mlp_z=enc_mz)
        n_ans = answer.input_ids.shape[0]
        kt = torch.as_tensor(k, device=image.device, dtype=torch.long)
        rep = torch.repeat_interleave(torch.arange(kt.numel(), device=image.device), kt, output_size=n_ans)   # :126-131
        question_atts = torch.index_select(quesiton.attention_mask, 0, rep)
        answer_output = self.text_decoder(answer.input_ids, attention_mask=answer.attention_mask,
                                          encoder_hidden_states=question_output.last_hidden_state,
                                          encoder_attention_mask=question_atts, encoder_batch_index=rep,
                                          labels=answer_targets, return_dict=True, reduction="none",
                                          sequence_weights=weights, output_attentions=output_attentions,
                                          output_hidden_states=output_hidden_states, head_z=z("decoder_head_z"),
                                          mlp_z=z("decoder_intermediate_z"))
        loss = answer_output.loss / image.size(0)                 # sum(weights * per-answer loss) / bs          (:166-167)
        if not output_attentions:
            return loss
        return {"loss": loss,
                "hidden_dict": {"image_hidden_states": image_hidden_states, "text_hidden_states": question_output.hidden_states,
                                "decoder_hidden_states": answer_output.hidden_states},
                "attention_dict": {"image_attentions": image_attentions, "text_attentions": question_output.attentions,
                                   "decoder_attentions": answer_output.attentions},
                "cross_attention_dict": {"cross_attentions": question_output.cross_attentions,
                                         "decoder_cross_attentions": answer_output.cross_attentions},
                "logits_dict": {"logits": answer_output.logits}}


class EffXVLMForVQA(_VQABase):
    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False, config_text=None)
        self._build(config)
        self.l0_module = VQAL0Module(config, target_sparsity=config["sparsity"])

    def forward(self, image, quesiton, answer=None, k=None, weights=None, train=True, output_attentions=None,
                output_hidden_states=None, stop_prune=False):
        if not train:
            raise NotImplementedError("answer ranking (rank_answer, model_generation.py:189-300) is evaluation code outside "
                                      "the distillation training path")
        if not stop_prune:
            zs = self.l0_module.forward(training=True)
        else:
            with torch.no_grad():
                zs = self.l0_module.forward(training=False)
        return self._train_forward(image, quesiton, answer, k, weights, zs, output_attentions, output_hidden_states)
