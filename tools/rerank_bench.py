#!/usr/bin/env python3
"""Throughput of the retrieval rerank loop (efficientvlm_amd.retrieval_eval.evaluation_scores) on synthetic data:
X-VLM-small with deterministic L0 gates, bf16, 224x224, 30 tokens, k_test = 128."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import model_config
from oracle import synth
from efficientvlm_amd.runtime import compute
from efficientvlm_amd.retrieval_eval import evaluation_scores
from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
geom = synth.GEOMS["full"]; dev = torch.device("cuda")
torch.manual_seed(0)
model = EffXVLMforRetrieval(model_config(geom, "s")).to(dev).eval()
n_img, n_txt, k = 256, 1280, 128
bi = synth.make_batch(geom, n_img, seed=1); bt = synth.make_batch(geom, n_txt, seed=2)
images, ids, atts = bi["image"].to(dev), bt["text_ids"].to(dev), bt["text_atts"].to(dev)
for qb in (1, 8, 16):
    with compute(torch.bfloat16):
        evaluation_scores(model, images[:32], ids[:160], atts[:160], k_test=16, zs=None, query_bs=qb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i2t, t2i = evaluation_scores(model, images, ids, atts, k_test=k, zs=None, query_bs=qb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"query_bs": qb, "images": n_img, "texts": n_txt, "k_test": k, "seconds": round(dt, 3),
                      "rescored_pairs_per_s": round((n_img + n_txt) * k / dt, 1)}), flush=True)
