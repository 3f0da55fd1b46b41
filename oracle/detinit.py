"""Deterministic, name-keyed weight initialisation shared by the golden generator and the tests.

TEST INFRASTRUCTURE.  Weights are never stored in fixtures: both the reference run
(oracle/gen_golden.py) and the parity tests regenerate them from (seed, parameter name) with a
CPU ``torch.Generator`` and compare per-tensor checksums recorded in the fixture.
"""
import zlib

import torch

# tied parameters share one storage in the reference (eff_bert.py:735-741 + HF tie_weights):
# generate them from ONE canonical name so both aliases get identical values.
_TIED = (
    ("cls.predictions.decoder.weight", "bert.embeddings.word_embeddings.weight"),
    ("cls.predictions.decoder.bias", "cls.predictions.bias"),
)


def canonical(name: str) -> str:
    for alias, target in _TIED:
        if name.endswith(alias):
            return name[: -len(alias)] + target
    return name


def _is_norm_weight(name: str) -> bool:
    n = name.lower()
    return n.endswith(".weight") and ("layernorm" in n or "layer_norm" in n or "layrnorm" in n
                                      or n.endswith("_head.1.weight"))


def det_tensor(name: str, shape, seed: int, std: float = 0.02) -> torch.Tensor:
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(canonical(name).encode())) % (2 ** 63))
    x = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if name.endswith("temp"):
        return torch.full(tuple(shape), 0.07)
    if name.endswith("lambda_1") or name.endswith("lambda_2"):
        return torch.zeros(tuple(shape))
    if name.endswith("_loga"):
        return x  # overwritten by callers that care
    if _is_norm_weight(name):
        return 1.0 + 0.1 * x
    return std * x


def det_state_dict(sd, seed: int, std: float = 0.02):
    """returns a new state dict with every floating tensor regenerated deterministically."""
    out = {}
    for k, v in sd.items():
        if torch.is_floating_point(v):
            out[k] = det_tensor(k, v.shape, seed, std).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def checksums(sd):
    return {k: (float(v.double().sum()), float(v.double().abs().sum()))
            for k, v in sd.items() if torch.is_floating_point(v)}
