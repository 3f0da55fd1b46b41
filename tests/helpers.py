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
    """MAX-NORM criterion: max|a - b| <= atol + rtol * max|b|.  "1e-4 relative" in the tests that use it means relative to
    the LARGEST reference element - right for maps, logits and losses (scalars: the plain relative error); elements far
    below the maximum may be off by more than rtol in their own scale.  Gradients are held to relative L2 norms instead."""
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


def grad_parity_stats(hip_grads, leaves, floor=1e-5):
    """per-tensor comparison of the HIP path's gradients (name -> tensor) with the oracle's (name -> leaf with .grad):
    relative L2 error and cosine of every tensor whose oracle gradient norm exceeds `floor` x the largest one, plus the
    global cosine.  Returns dict(global_cos, median, p90, max, qk_median, qk_max, stats=[(rel, cos, name)] worst first)."""
    import math
    gmax = max(float(l.grad.norm()) for l in leaves.values() if l.grad is not None)
    stats, num, da, db = [], 0.0, 0.0, 0.0
    for name, leaf in leaves.items():
        if leaf.grad is None or name not in hip_grads or float(leaf.grad.norm()) < floor * gmax:
            continue
        a, b = hip_grads[name].double().reshape(-1).cpu(), leaf.grad.double().reshape(-1)
        stats.append((float((a - b).norm() / b.norm()), float((a * b).sum() / (a.norm() * b.norm())), name))
        num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
    rels = sorted(r for r, _, _ in stats)
    qk = sorted(r for r, _, n in stats if any(t in n for t in ("q_proj", "k_proj", ".query.", ".key.")))
    return {"global_cos": num / math.sqrt(da * db), "median": rels[len(rels) // 2], "p90": rels[int(0.9 * len(rels))],
            "max": rels[-1], "qk_median": qk[len(qk) // 2], "qk_max": qk[-1], "n": len(stats),
            "stats": sorted(stats, reverse=True)}
