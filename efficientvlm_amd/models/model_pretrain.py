"""GD / pre-training model — drop-in for models/model_pretrain.py:XVLM (reference :5-82), the object
GeneralDistill.py builds for both student and teacher."""
import torch

from .xvlm import XVLMBase


class XVLM(XVLMBase):
    def __init__(self, config):
        # the reference hard-codes load_vision_params=load_text_params=True (model_pretrain.py:7-8); a config may set
        # 'load_params': False to build from random init without the CLIP / BERT checkpoint files (bench, tests).
        load = bool(config.get("load_params", True))
        super().__init__(config, load_vision_params=load, load_text_params=load, use_contrastive_loss=True,
                         use_matching_loss=True, use_mlm_loss=True, use_bbox_loss=True, config_text=None)

    def forward(self, image, text_ids, text_atts, text_ids_masked=None, masked_pos=None, masked_ids=None, image_atts=None,
                idx_to_group_img=None, target_bbox=None, is_image=None, ret_bbox_loss=False, output_attentions=None,
                output_hidden_states=None):
        assert output_attentions == output_hidden_states
        if ret_bbox_loss:
            raise NotImplementedError("region (bbox) batches are outside the benchmarked general-distillation path")
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
        return {"loss": loss, "hidden_dict": hidden_dict, "attention_dict": attention_dict,
                "cross_attention_dict": cross_attention_dict, "logits_dict": logits_dict}
