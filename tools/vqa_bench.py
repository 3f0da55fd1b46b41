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
res = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 480
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 32
geom = dict(synth.GEOMS["full"]); geom["image_res"] = res
dev = torch.device("cuda")
if os.environ.get("EVLM_FORCE_REDUCE"):       # the N > 1 code path (collectives, gradient stages, graph segments) on a one-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(0)
# --dropout P: the student BERT's (question encoder + answer decoder) hidden / attention-probability dropout; default 0
DROP = float(sys.argv[sys.argv.index("--dropout") + 1]) if "--dropout" in sys.argv else 0.0
cfg = lambda role, nd: dict(model_config(geom, role, image_res=res, dropout=DROP if role == "s" else 0.0), pad_token_id=0, num_dec_layers=nd)
student = EffXVLMForVQA(cfg("s", 3)).to(dev)
teacher = XVLMForVQA(cfg("t", 6)).to(dev)
student.l0_module.set_lagrangian_warmup_steps(100)
pipe = not os.environ.get("EVLM_NO_PIPELINE")
tr = VQATrainer(student, teacher, lr=5e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                pipeline_teacher=pipe, capture_step=not os.environ.get("EVLM_NO_STEP_GRAPH"))
if "--ragged" in sys.argv:
    # Round 6: batches as an epoch of the reference delivers them - questions padded to the batch's longest (8..40 tokens),
    # 1..10 candidate answers per question (dataset/vqa_dataset.py:101-116), answers padded to their longest (3..8 tokens) -
    # fed through data.bucket_pad_vqa: a few shapes, so the captured step replays.
    import random
    from efficientvlm_amd.data import bucket_pad_vqa
    steps = int(sys.argv[sys.argv.index("--ragged") + 1]) if len(sys.argv) > sys.argv.index("--ragged") + 1 else 200
    rnd = random.Random(0)
    images = torch.randn(B, 3, res, res, generator=torch.Generator().manual_seed(1)).to(dev)
    def ragged_batch():
        Lq, La = rnd.randint(8, 40), rnd.randint(3, 8)
        g = dict(geom); g["L"] = Lq; g["M"] = 2
        b = synth.make_vqa_batch(g, B, seed=rnd.randint(0, 1 << 20), La=max(La, 3))
        k = torch.tensor([rnd.randint(1, 10) for _ in range(B)], dtype=torch.long)
        n = int(k.sum())
        reps = (n + b["answer_ids"].shape[0] - 1) // b["answer_ids"].shape[0]
        for key in ("answer_ids", "answer_atts", "weights"):
            b[key] = torch.cat([b[key]] * reps, 0)[:n]
        b["k"] = k
        b = {kk: v.to(dev) for kk, v in b.items() if kk != "image"}
        b["image"] = images
        return bucket_pad_vqa(b)
    # (the batches are synthesised BEFORE the clock starts - 48 of them, fed in a random order: building one on the host takes
    # longer than the step it feeds)
    feed = [ragged_batch() for _ in range(48)]
    launches, host_s, shapes, t_all = [], 0.0, set(), None
    for s in range(steps + 48):
        if s == 48:
            torch.cuda.synchronize(); t_all = time.perf_counter(); launches, host_s = [], 0.0
        b = feed[s] if s < 48 else feed[rnd.randrange(48)]
        shapes.add((int(b["question_ids"].shape[1]), int(b["answer_ids"].shape[1]), int(b["answer_ids"].shape[0])))
        t0 = time.perf_counter()
        out = tr.step(b)
        host_s += time.perf_counter() - t0
        if out is not None:
            launches.append(tr.last_launch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t_all) / steps
    # host ISSUE time per step, measured with the device idle at every call (in the loop above the host runs ahead until the
    # launch queue pushes back: its time per call then converges to the device's)
    host_s = 0.0
    for s in range(40):
        b = feed[rnd.randrange(48)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = tr.step(b)
        host_s += time.perf_counter() - t0
        launches.append(tr.last_launch)
    host_s *= steps / 40.0
    torch.cuda.synchronize()
    rep = sum(1 for l in launches if l.startswith("hipGraph"))
    print(json.dumps({"workload": "VQA pruning fine-tune step, ragged epoch (questions 8..40 tokens, 1..10 answers each, bucket-padded)",
                      "dropout": DROP, "launch": tr.last_launch, "image_res": res, "batch": B, "steps": steps, "padded_shapes[q_len,a_len,rows]": sorted(shapes),
                      "replayed_from_hipgraph": rep, "replay_frac": round(rep / len(launches), 4),
                      "host_ms_per_step": round(host_s / steps * 1e3, 2), "ms_per_step": round(dt * 1e3, 2),
                      "questions_per_s": round(B / dt, 1), "captured_pairs": len(tr._sgraphs),
                      "losses[total,answer,kd,lagrangian]": [round(float(x), 4) for x in out.tolist()]}))
    sys.exit(0)
batch = synth.make_vqa_batch(geom, B, seed=5, La=8)
batch["k"] = torch.full((B,), 4, dtype=torch.long)                 # 4 answers per question
n = 4 * B
reps = (n + batch["answer_ids"].shape[0] - 1) // batch["answer_ids"].shape[0]
for key in ("answer_ids", "answer_atts", "weights"):
    batch[key] = torch.cat([batch[key]] * reps, 0)[:n]
batch = {k: v.to(dev) for k, v in batch.items()}
for _ in range(6): out = tr.step(batch)      # (prime, one eager step per parity, one capture per parity)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 8
for _ in range(K): out = tr.step(batch)
host = (time.perf_counter() - t0) / K      # host time per step (before the device has caught up)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": "VQA pruning fine-tune step", "dropout": DROP, "teacher_pipelined": pipe, "launch": tr.last_launch, "host_ms_per_step": round(host * 1e3, 2), "image_res": res, "batch": B, "answers": n,
                  "ms_per_step": round(dt * 1e3, 2), "questions_per_s": round(B / dt, 1),
                  "losses[total,answer,kd,lagrangian]": [round(float(x), 4) for x in out.tolist()]}))
