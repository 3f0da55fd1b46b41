"""efficientvlm_amd — MI355X-native distillation-training hot path of EfficientVLM.

Drop-in module surface (same class names / signatures / state-dict keys as the reference):
    efficientvlm_amd.efficient_models.{eff_vit, eff_bert, xvlm, xvlm_l0_module, model_retrieval}
    efficientvlm_amd.models.{clip_vit, xbert, xvlm, model_pretrain, model_retrieval}
All arithmetic runs in hand-written gfx950 HIP kernels (efficientvlm_amd/csrc) behind the C ABI of
include/evlm_hip.h; there is no CPU / ATen fallback.
"""
__version__ = "0.1.0"
