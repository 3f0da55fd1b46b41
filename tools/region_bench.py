#!/usr/bin/env python3
"""The GD recipe WITH region steps on one GPU (GeneralDistill.py:157-262, configs/Pretrain_XVLM_small_4m.yaml regions:
iter_perc 0.5, batch_size 128, max_images 48): region step = 48 images expanded to 128 (text, region) rows, ViT region
split for the last 2 (student) / 4 (teacher) layers, ITC/ITM/MLM on the region embeddings, bbox fusion pass + L1/GIoU;
general step = the bench.py workload (B = 64).  bf16, synthetic data, random init, hipGraph replay.  Reports the region
step alone and the alternating recipe (R G R G ...: what iter_perc = 0.5 averages to) unpipelined and pipelined."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import synth
import bench
from efficientvlm_amd.trainer import GDTrainer
geom = synth.GEOMS["full"]; dev = torch.device("cuda")
N_IMG, R, B, K = 48, 128, 64, 10
todev = lambda b: {k: v.to(dev) for k, v in b.items()}
Rb = [todev(synth.make_region_batch(geom, N_IMG, R, seed=300 + i, ragged=False)) for i in range(2)]
Gb = [todev(synth.make_batch(geom, B, seed=400 + i)) for i in range(2)]


def run(pipe, seq, label):
    s, t = bench.build(geom, dev, 1234)
    tr = GDTrainer(s, t, dtype=torch.bfloat16, use_graph=True, pipeline_teacher=pipe)
    for b in seq * 2:
        out = tr.step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        for b in seq:
            o = tr.step(b)
            out = o if o is not None else out
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    rows = sum(b["text_ids"].shape[0] for b in seq)
    print(json.dumps({"workload": label, "pipelined_teacher": pipe, "ms_per_cycle": round(dt * 1e3, 2),
                      "rows_per_cycle": rows, "pairs_per_s": round(rows / dt, 1),
                      "last[total,itc,itm,mlm,kd]": [round(float(x), 4) for x in out.tolist()]}), flush=True)
    del tr, s, t
    torch.cuda.empty_cache()


run(False, [Rb[0], Rb[1]], "2 region steps (48 images -> 128 rows each)")
run(False, [Rb[0], Gb[0], Rb[1], Gb[1]], "recipe cycle R G R G (region 128 rows, general 64 pairs)")
run(True, [Rb[0], Gb[0], Rb[1], Gb[1]], "recipe cycle R G R G (region 128 rows, general 64 pairs)")
