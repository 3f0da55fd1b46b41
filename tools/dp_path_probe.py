#!/usr/bin/env python3
"""Where the N > 1 code path of the GD step spends its time on ONE GPU (RCCL group of one rank): eager student step vs
hipGraph segments, gradient exchange on / off.  EVLM_FORCE_REDUCE=1 python tools/dp_path_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EVLM_FORCE_REDUCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
import torch, torch.distributed as dist
import bench
from efficientvlm_amd.workload import GEOMS, make_batch
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
geom = GEOMS["full"]; dev = torch.device("cuda", 0)
batches = [{k: v.to(dev) for k, v in make_batch(geom, 64, seed=42 + 1000 * i).items()} for i in range(4)]

def run(tag, env=None, patch=None, steps=10):
    for k, v in (env or {}).items():
        os.environ[k] = v
    s, t = bench.build(geom, dev, 1234)
    tr = bench.make_trainer(s, t, torch.bfloat16, True, True)
    if patch: patch(tr)
    it = 0
    for _ in range(6):
        tr.step(batches[it % 4]); it += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(batches[it % 4]); it += 1
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{tag:50s} {el / steps * 1e3:7.2f} ms/step   (host issue {host / steps * 1e3:6.2f} ms)", flush=True)
    for k in (env or {}):
        os.environ.pop(k, None)
    del tr, s, t

def no_reduce(tr):
    tr.reducer.reduce_async = lambda tensors: None
    tr.reducer.finish = lambda: None
def fp32_wire(tr):
    tr.reducer.compress = None

for rep in range(2):
    run("segments, fp32 wire", patch=fp32_wire)
    run("segments, bf16 wire")
    run("segments, no gradient exchange", patch=no_reduce)
dist.destroy_process_group()
