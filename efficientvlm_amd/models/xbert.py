"""models/xbert.py of the reference == efficient_models/eff_bert.py without the z hooks."""
from ..efficient_models.eff_bert import *  # noqa: F401,F403
from ..efficient_models.eff_bert import BertConfig, BertForMaskedLM, BertModel  # noqa: F401
