"""shared helpers for the parity tests: model configs of oracle.synth.GEOMS for the drop-in modules, weight
loading from the deterministic generator, fixture access."""
import os

import numpy as np
import torch

from oracle import schema, synth
from oracle import xvlm_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


from efficientvlm_amd.workload import model_config  # noqa: E402,F401  (one definition: the package's)


def load_det_weights(model, sch, seed, std, fx=None, tag=None):
    """load deterministic weights; strict=True proves the module exposes exactly the reference's state-dict keys"""
    sd = schema.det_weights(sch, seed, std)
    full = dict(sd)
    for k, v in model.state_dict().items():
        if k not in full:
            assert not torch.is_floating_point(v), f"unexpected floating key {k}"
            full[k] = v
    model.load_state_dict(full, strict=True)
    if fx is not None:
        ref_names = {k[len(tag) + 6:] for k in fx if k.startswith(tag + ".wchk.")}
        mine = {k for k, v in model.state_dict().items() if torch.is_floating_point(v)}
        assert ref_names == mine, sorted(ref_names ^ mine)[:6]
    return sd


def close(a, b, rtol, atol=0.0, what=""):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() if a.size else 0.0
    tol = atol + rtol * (np.abs(b).max() if b.size else 0.0)
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"


def batch_from_fixture(fx, device):
    keys = ("image", "text_ids", "text_atts", "text_ids_masked", "masked_pos", "masked_ids",
            "idx_to_group_img", "image_atts", "target_bbox", "is_image")           # the last four: region batches
    return {k: torch.from_numpy(fx["in." + k]).to(device) for k in keys if "in." + k in fx}
