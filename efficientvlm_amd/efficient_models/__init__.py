from .xvlm import XVLMBase, build_mlp, load_pretrained  # noqa: F401
