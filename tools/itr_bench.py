#!/usr/bin/env python3
"""BASELINE.json configs[2] on one GPU: the ITR pruning fine-tune step (Eff_Retrieval.py:train body = trainer.ITRTrainer:
X-VLM-small student with L0 gates fwd+bwd, X-VLM-base teacher fwd, ITC + ITM + KD incl. cross-attention maps, Lagrangian,
three optimisers), B = 64, 384x384 images (577 tokens), 30 text tokens, bf16, synthetic data, random init, hipGraph replay of the student step (EVLM_NO_STEP_GRAPH=1: eager; EVLM_FORCE_REDUCE=1: the N > 1 path as graph segments on a one-rank RCCL group)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import model_config
from oracle import synth
from efficientvlm_amd.trainer import ITRTrainer
from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
res = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 384
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 64
geom = dict(synth.GEOMS["full"]); geom["image_res"] = res
dev = torch.device("cuda")
if os.environ.get("EVLM_FORCE_REDUCE"):       # the N > 1 code path (collectives, gradient stages, graph segments) on a one-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(0)
# --dropout P: the student BERT's hidden / attention-probability dropout (the stock recipe trains with 0.1; default 0 as BASELINE quotes)
DROP = float(sys.argv[sys.argv.index("--dropout") + 1]) if "--dropout" in sys.argv else 0.0
student = EffXVLMforRetrieval(model_config(geom, "s", image_res=res, dropout=DROP)).to(dev)
teacher = TeacherITR(model_config(geom, "t", image_res=res)).to(dev)
student.l0_module.set_lagrangian_warmup_steps(100)
pipe = not os.environ.get("EVLM_NO_PIPELINE")
tr = ITRTrainer(student, teacher, lr=3e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                pipeline_teacher=pipe, capture_step=not os.environ.get("EVLM_NO_STEP_GRAPH"))
batch = {k: v.to(dev) for k, v in synth.make_batch(geom, B, seed=5).items()}
idx = torch.arange(B, device=dev)
if "--ragged" in sys.argv:
    # Round 6: batches as an epoch of the reference delivers them - every batch padded to ITS longest caption (Eff_Retrieval.py:97
    # padding='longest'; here a random real length of 8..40 tokens per batch) - fed through data.bucket_pad_itr: a few shapes,
    # so the captured step replays.  Reports how many steps replayed from a hipGraph and the mean host time per step.
    import random
    from efficientvlm_amd.data import bucket_pad_itr
    steps = int(sys.argv[sys.argv.index("--ragged") + 1]) if len(sys.argv) > sys.argv.index("--ragged") + 1 else 200
    rnd = random.Random(0)
    pool = {}
    def ragged_batch():
        L = rnd.randint(8, 40)
        if L not in pool:                      # (one synthetic batch per real length: the data itself is not the subject)
            g = dict(geom); g["L"] = L; g["M"] = 2
            pool[L] = {k: v.to(dev) for k, v in synth.make_batch(g, B, seed=100 + L, ragged=True).items() if k in ("image", "text_ids", "text_atts")}
        return L, bucket_pad_itr(pool[L])
    launches, host_s, shapes, t_all = [], 0.0, set(), None
    for s in range(steps + 32):
        if s == 32:
            torch.cuda.synchronize(); t_all = time.perf_counter(); launches, host_s = [], 0.0
        L, b = ragged_batch()
        shapes.add(int(b["text_ids"].shape[1]))
        t0 = time.perf_counter()
        out = tr.step(b, idx=idx)
        host_s += time.perf_counter() - t0
        if out is not None:
            launches.append(tr.last_launch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t_all) / steps
    # host ISSUE time per step, measured with the device idle at every call (in the loop above the host runs ahead until the
    # launch queue pushes back: its time per call then converges to the device's)
    host_s = 0.0
    for s in range(40):
        L, b = ragged_batch()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = tr.step(b, idx=idx)
        host_s += time.perf_counter() - t0
        launches.append(tr.last_launch)
    host_s *= steps / 40.0
    torch.cuda.synchronize()
    rep = sum(1 for l in launches if l.startswith("hipGraph"))
    print(json.dumps({"workload": "ITR pruning fine-tune step, ragged epoch (real text length 8..40 per batch, bucket-padded)",
                      "dropout": DROP, "launch": tr.last_launch, "image_res": res, "batch": B, "steps": steps, "padded_text_lengths": sorted(shapes),
                      "replayed_from_hipgraph": rep, "replay_frac": round(rep / len(launches), 4),
                      "host_ms_per_step": round(host_s / steps * 1e3, 2), "ms_per_step": round(dt * 1e3, 2),
                      "pairs_per_s": round(B / dt, 1), "captured_pairs": len(tr._sgraphs),
                      "losses[total,itc,itm,kd,lagrangian]": [round(float(x), 4) for x in out.tolist()]}))
    sys.exit(0)
for _ in range(6): out = tr.step(batch, idx=idx)      # (prime, one eager step per parity, one capture per parity)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 10
for _ in range(K): out = tr.step(batch, idx=idx)
host = (time.perf_counter() - t0) / K      # host time per step (before the device has caught up)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": "ITR pruning fine-tune step", "dropout": DROP, "teacher_pipelined": pipe, "launch": tr.last_launch, "host_ms_per_step": round(host * 1e3, 2), "image_res": res, "batch": B, "ms_per_step": round(dt * 1e3, 2),
                  "pairs_per_s": round(B / dt, 1), "losses[total,itc,itm,kd,lagrangian]": [round(float(x), 4) for x in out.tolist()]}))
