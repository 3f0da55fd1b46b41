"""Small host-side runtime shared by the drop-in modules: compute-dtype switch, config + output containers."""
import contextlib
import json
import os
from collections import OrderedDict

import torch

_STATE = {"dtype": torch.float32}


def set_compute_dtype(dtype):
    """torch.float32 = exact-fp32 parity path; torch.bfloat16 = fast path (bf16 storage, fp32 accumulate)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _STATE["dtype"] = dtype


def compute_dtype():
    return _STATE["dtype"]


@contextlib.contextmanager
def compute(dtype):
    old = _STATE["dtype"]
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _STATE["dtype"] = old


COLLECTIVES = None      # tests: set to a list -> every collective of a step is appended as (kind, numel, dtype)


def log_collective(kind, tensor):
    if COLLECTIVES is not None:
        COLLECTIVES.append((kind, int(tensor.numel()), str(tensor.dtype)))


def read_json(rpath):
    """utils/__init__.py read_json (plain local file; the reference's HDFS branch is out of scope)."""
    if isinstance(rpath, dict):
        return dict(rpath)
    with open(rpath, "r") as f:
        return json.load(f)


class AttrDict(dict):
    """utils/__init__.py:317-320"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


class BertConfig:
    """the fields of transformers.BertConfig the hot path reads (efficient_models/xvlm.py:147-160, eff_bert.py)."""

    _DEFAULTS = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                     initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0,
                     position_embedding_type="absolute", chunk_size_feed_forward=0, output_attentions=False,
                     output_hidden_states=False, use_return_dict=True, use_cache=False, is_decoder=False,
                     fusion_layer=6, encoder_width=768, fp16=False)

    def __init__(self, **kw):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path):
        return cls(**read_json(path))

    @classmethod
    def from_any(cls, cfg):
        """accept our BertConfig, a dict, or any object with the same attributes (e.g. HF BertConfig)"""
        if isinstance(cfg, cls):
            return cfg
        if isinstance(cfg, dict):
            return cls(**cfg)
        out = cls()
        for k in list(cls._DEFAULTS) + ["fusion_layer", "encoder_width", "fp16"]:
            if hasattr(cfg, k):
                setattr(out, k, getattr(cfg, k))
        return out

    def to_dict(self):
        return {k: getattr(self, k) for k in self.__dict__}


class ModelOutput(OrderedDict):
    """transformers ModelOutput semantics the callers rely on (xvlm.py:311,347,518): attribute access, and integer
    indexing over the non-None fields in declaration order."""

    def __init__(self, **fields):
        super().__init__()
        for k, v in fields.items():
            super().__setitem__(k, v)

    def __getattr__(self, name):
        try:
            return super().__getitem__(name)
        except KeyError:
            raise AttributeError(name)

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)

    def __getitem__(self, k):
        if isinstance(k, str):
            return super().__getitem__(k)
        return self.to_tuple()[k]
