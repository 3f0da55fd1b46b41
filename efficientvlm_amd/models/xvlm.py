"""models/xvlm.py of the reference (teacher / GD models): XVLMBase without z arguments.  Differences from the
efficient_models variant that callers can observe: get_vision_embeds ALWAYS returns the 4-tuple
(models/xvlm.py:331-336)."""
from ..efficient_models.xvlm import (AllGather, XVLMBase as _EffXVLMBase, allgather, build_mlp,  # noqa: F401
                                     build_text_encoder, build_vision_encoder, interpolate_pos_embed,
                                     load_params_change_prefix, load_params_choose_layers, load_pretrained)


class XVLMBase(_EffXVLMBase):
    get_vision_embeds_returns_pair = False
