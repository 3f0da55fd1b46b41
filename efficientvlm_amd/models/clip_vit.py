"""models/clip_vit.py of the reference == efficient_models/eff_vit.py with every z hook removed:
one implementation serves both import paths."""
from ..efficient_models.eff_vit import (CLIPAttention, CLIPEncoder, CLIPEncoderLayer, CLIPMLP,  # noqa: F401
                                        CLIPVisionTransformer)
