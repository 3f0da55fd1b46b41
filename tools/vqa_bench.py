#!/usr/bin/env python3
"""BASELINE.json configs[3] on one GPU: the VQA pruning fine-tune step (Eff_VQA.py:74-200 = trainer.VQATrainer: X-VLM-small
student with VQAL0Module gates fwd+bwd, X-VLM-base teacher fwd, weighted answer LM loss + hidden / attention / logit KD,
Lagrangian, three optimisers), per-GPU batch 32 (256 over 8 GPUs), 480x480 images (901 tokens), 30-token questions, about
4 candidate answers of <= 8 tokens per question, bf16, synthetic data, random init, hipGraph replay of the student step (EVLM_NO_STEP_GRAPH=1: eager; EVLM_FORCE_REDUCE=1: the N > 1 path as graph segments on a one-rank RCCL group).
usage: vqa_bench.py [image_res] [batch]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import model_config
from oracle import synth
from efficientvlm_amd.trainer import VQATrainer
from efficientvlm_amd.efficient_models.model_generation import EffXVLMForVQA
from efficientvlm_amd.models.model_generation import XVLMForVQA
res = int(sys.argv[1]) if len(sys.argv) > 1 else 480
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
geom = dict(synth.GEOMS["full"]); geom["image_res"] = res
dev = torch.device("cuda")
if os.environ.get("EVLM_FORCE_REDUCE"):       # the N > 1 code path (collectives, gradient stages, graph segments) on a one-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(0)
cfg = lambda role, nd: dict(model_config(geom, role, image_res=res), pad_token_id=0, num_dec_layers=nd)
student = EffXVLMForVQA(cfg("s", 3)).to(dev)
teacher = XVLMForVQA(cfg("t", 6)).to(dev)
student.l0_module.set_lagrangian_warmup_steps(100)
pipe = not os.environ.get("EVLM_NO_PIPELINE")
tr = VQATrainer(student, teacher, lr=5e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                pipeline_teacher=pipe, capture_step=not os.environ.get("EVLM_NO_STEP_GRAPH"))
batch = synth.make_vqa_batch(geom, B, seed=5, La=8)
batch["k"] = torch.full((B,), 4, dtype=torch.long)                 # 4 answers per question
n = 4 * B
reps = (n + batch["answer_ids"].shape[0] - 1) // batch["answer_ids"].shape[0]
for key in ("answer_ids", "answer_atts", "weights"):
    batch[key] = torch.cat([batch[key]] * reps, 0)[:n]
batch = {k: v.to(dev) for k, v in batch.items()}
for _ in range(6): out = tr.step(batch)      # (prime, one eager step per parity, one capture per parity)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for _ in range(K): out = tr.step(batch)
host = (time.perf_counter() - t0) / K      # host time per step (before the device has caught up)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": "VQA pruning fine-tune step", "teacher_pipelined": pipe, "launch": tr.last_launch, "host_ms_per_step": round(host * 1e3, 2), "image_res": res, "batch": B, "answers": n,
                  "ms_per_step": round(dt * 1e3, 2), "questions_per_s": round(B / dt, 1),
                  "losses[total,answer,kd,lagrangian]": [round(float(x), 4) for x in out.tolist()]}))
