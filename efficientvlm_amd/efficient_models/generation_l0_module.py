"""Hard-concrete L0 gates of the VQA model - drop-in for efficient_models/generation_l0_module.py:VQAL0Module: the gates of
XVLML0Module plus `decoder_head_loga` [2 * decoder layers, heads] (self- and cross-attention of every answer-decoder layer,
interleaved) and `decoder_int_loga` [decoder layers, ffn]; decoder layers = fusion layers (generation_l0_module.py:47)."""
from .xvlm_l0_module import XVLML0Module


class VQAL0Module(XVLML0Module):
    with_decoder = True
