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


def model_config(geom, role, image_res=None, sparsity=0.25):
    """the config dict the reference constructors take (Pretrain_XVLM_small_4m.yaml keys), with the json files inlined"""
    vit_layers, text_layers = geom[f"{role}_vit_layers"], geom[f"{role}_text_layers"]
    vision = {"ckpt": "none", "vision_width": geom["hidden"], "patch_size": 16, "hidden_act": "quick_gelu",
              "num_attention_heads": geom["heads"], "attention_dropout": 0.0, "intermediate_size": geom["ffn"],
              "num_hidden_layers": vit_layers, "local_attn_depth": 2 if vit_layers == 6 else 4}
    bert = {"hidden_size": geom["hidden"], "num_attention_heads": geom["heads"], "intermediate_size": geom["ffn"],
            "num_hidden_layers": 12, "hidden_act": "gelu", "hidden_dropout_prob": 0.0,
            "attention_probs_dropout_prob": 0.0, "layer_norm_eps": 1e-12, "max_position_embeddings": geom["max_pos"],
            "type_vocab_size": 2, "vocab_size": geom["vocab"], "pad_token_id": 0, "initializer_range": 0.02}
    return {"use_clip_vit": True, "use_swin": False, "vision_config": vision, "image_res": image_res or geom["image_res"],
            "patch_size": 16, "text_encoder": bert, "text_num_hidden_layers": text_layers, "embed_dim": geom["embed_dim"],
            "temp": 0.07, "accelerator": {"FP16_OPT_LEVEL": "O0"}, "sparsity": sparsity, "load_params": False}


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
