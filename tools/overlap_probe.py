#!/usr/bin/env python3
"""How much of the frozen teacher's forward the joint hipGraph hides: the benchmarked step (B = 64, teacher pipelined one
batch ahead, forked onto the side stream), the same graph with the teacher branch empty (student step alone) and with the
student branch empty (teacher forward alone).  J against S + T: what the two branches gain from sharing the chip.

    python tools/overlap_probe.py [--steps 20] [--reps 2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from efficientvlm_amd.workload import GEOMS, make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--only", default="", help="J, S or T: that run alone (for a kernel trace)")
ap.add_argument("--dp", action="store_true", help="the N > 1 code path (hipGraph segments, one-rank RCCL group) instead of the joint graph")
args = ap.parse_args()
torch.cuda.set_device(0)
if args.dp:
    import torch.distributed as dist
    os.environ["EVLM_FORCE_REDUCE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29563")
    dist.init_process_group("nccl", rank=0, world_size=1)
geom = GEOMS["full"]; dev = torch.device("cuda", 0)
batches = [{k: v.to(dev) for k, v in make_batch(geom, 64, seed=42 + 1000 * i).items()} for i in range(4)]


def run(tag, patch=None):
    s, t = bench.build(geom, dev, 1234)
    tr = bench.make_trainer(s, t, torch.bfloat16, True, True)
    if patch: patch(tr)
    it = 0
    for _ in range(7):
        tr.step(batches[it % 4]); it += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(batches[it % 4]); it += 1
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{tag:52s} {el / args.steps * 1e3:7.2f} ms/step", flush=True)
    del tr, s, t
    return el / args.steps * 1e3


def no_teacher(tr):
    tr._teacher_eager = lambda pipe, k: None
    tr._teacher_begin = lambda pipe, k: None
    tr._teacher_finish = lambda st, pipe, k: None


def no_student(tr):
    z = torch.zeros(5, device=dev)
    tr._student_eager = lambda pipe, k: z


if args.only:
    run(*{"J": ("J: joint graph",), "S": ("S: student step alone", no_teacher), "T": ("T: teacher forward alone", no_student)}[args.only])
    sys.exit(0)
for rep in range(args.reps):
    j = run("J: joint graph (student step || teacher forward)")
    s = run("S: student step alone (teacher branch empty)", no_teacher)
    t = run("T: teacher forward alone (student branch empty)", no_student)
    print(f"   S + T = {s + t:.2f} ms; J = {j:.2f} ms: sharing the chip hides {s + t - j:.2f} ms "
          f"({(s + t - j) / t * 100:.0f} % of the teacher forward)", flush=True)
