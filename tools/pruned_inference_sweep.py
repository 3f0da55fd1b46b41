#!/usr/bin/env python3
"""BASELINE.json configs[4]: forward throughput of the physically pruned X-VLM-small (eff_vit + eff_bert) at 100 / 75 / 50 /
25 % retained heads + FFN units, B = 64, 224x224, 30 tokens, bf16, random init, synthetic data.  One "pair" = image encoder
+ text encoder + ITC features + the 3 fusion layers on the (image, text) pair + ITM head (the retrieval scoring path of
Eff_Retrieval.py:216-332).  Prints one JSON line per sparsity.
    pruned_inference_sweep.py masked      the MASKED-DENSE form instead (the un-pruned model with the 0 / 1 gates as multipliers,
                                          efficient_models/model_retrieval.py:76-93): the attention kernels skip closed heads
                                          (EVLM_ATTN_NO_HEAD_SKIP=1 turns that off for an A/B), the GEMMs stay full size"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import model_config
from oracle import synth
from efficientvlm_amd import pruning
from efficientvlm_amd.runtime import compute
from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
from efficientvlm_amd.efficient_models.xvlm import mlp_head_forward

geom = synth.GEOMS["full"]; dev = torch.device("cuda"); B = 64
batch = {k: v.to(dev) for k, v in synth.make_batch(geom, B, seed=1).items()}


def masks(keep, g):
    zs = {}
    def pick(n_layers, width, shape):
        z = torch.zeros(n_layers, width)
        k = max(1, int(round(width * keep)))
        for l in range(n_layers):
            z[l, torch.randperm(width, generator=g)[:k]] = 1
        return z.reshape(shape)
    zs["vision_head_z"] = pick(6, 12, (6, 1, 12, 1, 1)); zs["text_head_z"] = pick(3, 12, (3, 1, 12, 1, 1))
    zs["cross_head_z"] = pick(6, 12, (6, 1, 12, 1, 1))
    zs["vision_intermediate_z"] = pick(6, 3072, (6, 1, 1, 3072)); zs["text_intermediate_z"] = pick(3, 3072, (3, 1, 1, 3072))
    zs["cross_intermediate_z"] = pick(3, 3072, (3, 1, 1, 3072))
    return zs


def score(model, z=None):
    z = z or {}
    # round 6: the two encoders side by side (XVLMBase.get_pair_embeds: the text pass on a second stream under the image
    # encoder's GEMMs; EVLM_NO_PAIR_STREAM=1: in sequence, the round-5 form)
    image_embeds, image_atts, text_embeds = model.get_pair_embeds(
        batch["image"], batch["text_ids"], batch["text_atts"],
        vision_kw=dict(head_z=z.get("vision_head_z"), mlp_z=z.get("vision_intermediate_z")),
        text_kw=dict(head_z=z.get("text_head_z"), mlp_z=z.get("text_intermediate_z")), side_stream=SIDE)
    image_feat, text_feat = model.get_features(image_embeds, text_embeds)
    cross = model.get_cross_embeds(image_embeds, image_atts, text_embeds=text_embeds, text_atts=batch["text_atts"],
                                   head_z=z.get("cross_head_z"), mlp_z=z.get("cross_intermediate_z"))
    return image_feat, text_feat, mlp_head_forward(model.itm_head, cross[:, 0, :])


MASKED = len(sys.argv) > 1 and sys.argv[1] == "masked"
SIDE = None if os.environ.get("EVLM_NO_PAIR_STREAM") else torch.cuda.Stream()


KEEPS = [float(v) for v in sys.argv[sys.argv.index("--keep") + 1].split(",")] if "--keep" in sys.argv else [1.0, 0.75, 0.5, 0.25]
for keep in KEEPS:
    torch.manual_seed(0)
    model = EffXVLMforRetrieval(model_config(geom, "s")).to(dev).eval()
    n0 = sum(p.numel() for n, p in model.named_parameters() if not n.startswith("l0_module"))
    zd = None
    if keep < 1.0:
        zs = masks(keep, torch.Generator().manual_seed(3))
        if MASKED:
            zd = {k: v.to(dev) for k, v in zs.items()}
        else:
            with torch.no_grad():
                pruning.update_params(model, zs); pruning.prune_model_with_z(zs, model)
    n1 = sum(p.numel() for n, p in model.named_parameters() if not n.startswith("l0_module"))
    with torch.no_grad(), compute(torch.bfloat16):
        for _ in range(3): score(model, zd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): score(model, zd)
        torch.cuda.synchronize(); dt_eager = (time.perf_counter() - t0) / 20
        graph = torch.cuda.CUDAGraph()                 # the forward is ~250 small launches: replay it as one hipGraph
        with torch.cuda.graph(graph):
            out = score(model, zd)
        for _ in range(3): graph.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): graph.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(json.dumps({"retained": keep, "params_M": round(n1 / 1e6, 1), "params_dense_M": round(n0 / 1e6, 1),
                      "ms_per_batch": round(dt * 1e3, 3), "pairs_per_s": round(B / dt, 1), "launch": "hipGraph replay", "encoders": "side by side" if SIDE is not None else "in sequence",
                      "form": ("masked dense" + ("" if os.environ.get("EVLM_ATTN_NO_HEAD_SKIP") else ", closed heads skipped in-kernel"))
                              if MASKED else "physically pruned",
                      "ms_per_batch_eager": round(dt_eager * 1e3, 3)}), flush=True)
