from .xvlm import XVLMBase, build_mlp, load_pretrained  # noqa: F401  (models/__init__.py:1-3 of the reference)
