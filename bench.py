#!/usr/bin/env python3
"""Headline benchmark: image-text pairs/s of ONE general-distillation step (X-VLM-small student forward+backward,
X-VLM-base teacher forward under no_grad, every KD loss, gradient reduction, global-norm clip + AdamW) on synthetic
224x224 images + 30-token captions, batch 64 per GPU, bf16 compute (BASELINE.json configs[1]).

    python bench.py [--gpus N --steps K --warmup W]             # N=1 directly
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                  # N>1: one rank per GPU over RCCL

The frozen teacher is pipelined one batch ahead of the student (GDTrainer(pipeline_teacher=True)): every
timed step runs one teacher forward (on the next batch), one student forward + backward and one optimiser step; four
distinct synthetic batches are fed round-robin and a priming call precedes the warm-up.  --no-pipeline runs both models on
the same batch inside each step.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     - the dominant kernel (the bf16 MFMA GEMM kernel with the largest share of the step): algorithmic FLOPs per
                 launch / its average launch duration, timed live with HIP events on the launch stream during one extra
                 instrumented step;
  cpu_baseline - oracle/ (the CPU restatement of the reference) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FLOPS_PER_PAIR = 161.4e9         # 3 x student fwd (32.41 GF) + teacher fwd (64.20 GF)


def build(geom, dev, seed):
    from helpers import model_config
    from efficientvlm_amd.models.model_pretrain import XVLM
    torch.manual_seed(seed)
    student = XVLM(model_config(geom, "s")).to(dev)
    teacher = XVLM(model_config(geom, "t")).to(dev)
    return student, teacher


def _cpu_threads():
    """threads the CPU baseline may use: the cores this process is actually allowed on, capped at 32 (an oversubscribed
    torch thread pool on a quota-limited container is slower than a few threads)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_baseline_child(B=2, budget_s=25.0):
    """oracle (kind='port') GD step, fp32, on the host cores; runs in a CHILD process (never touches the GPU) so the
    parent can bound it with a timeout.  Prints one JSON object."""
    from oracle import schema, synth
    from oracle import xvlm_oracle as O
    geom = synth.GEOMS["full"]
    nthreads = _cpu_threads()
    torch.set_num_threads(nthreads)
    s_cfg, t_cfg = O.model_cfg(geom, "s"), O.model_cfg(geom, "t")
    s_sd = schema.det_weights(schema.xvlm_schema(s_cfg, geom["max_pos"]), 1, geom["std"])
    t_sd = schema.det_weights(schema.xvlm_schema(t_cfg, geom["max_pos"]), 2, geom["std"])
    for sd in (s_sd, t_sd):
        sd["text_encoder.cls.predictions.decoder.weight"] = sd["text_encoder.bert.embeddings.word_embeddings.weight"]
        sd["text_encoder.cls.predictions.decoder.bias"] = sd["text_encoder.cls.predictions.bias"]
    leaves = {}
    for k, v in s_sd.items():
        leaves.setdefault(id(v), v.clone().requires_grad_(True))
    s_sd = {k: leaves[id(v)] for k, v in s_sd.items()}
    batch = synth.make_batch(geom, B, seed=42)
    neg = torch.tensor([(i + 1) % B for i in range(2 * B)])
    times = []
    t_start = time.time()
    for it in range(8):
        t0 = time.time()
        total, *_ = O.gd_step(s_sd, t_sd, s_cfg, t_cfg, batch, neg, neg)
        total.backward()
        for p in leaves.values():
            p.grad = None
        dt = time.time() - t0
        if it > 0 or dt > budget_s / 2:
            times.append(dt)
        if time.time() - t_start > budget_s and times:
            break
    t = sorted(times)[len(times) // 2]
    print(json.dumps({"value": round(B / t, 3), "unit": "pairs/s", "cores": nthreads, "kind": "port",
                      "sample": f"{len(times)} timed GD steps of batch {B} (224x224, 30 tokens), fp32, "
                                f"oracle/xvlm_oracle.py on {nthreads} host threads, median step {t:.2f} s"}), flush=True)


def cpu_baseline(timeout_s=150):
    """run the CPU baseline in a child process with a hard time limit"""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], capture_output=True,
                           text=True, timeout=timeout_s, env=env, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "pairs/s", "cores": _cpu_threads(), "kind": "port",
                "sample": "child failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "pairs/s", "cores": _cpu_threads(), "kind": "port",
                "sample": f"one batch-2 oracle GD step did not finish in {timeout_s} s on this host"}


def roofline_leg(trainer, batch):
    """one extra eager step with every GEMM launch bracketed by HIP events on its launch stream; launches are attributed to
    the kernel that served them (evlm_gemm_last_kernel).  `traffic` = HBM bytes per launch of the dominant kernel from the
    rocprofv3 PMC passes of this same command (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    KiB -> bytes), null when that summary is absent."""
    from efficientvlm_amd import ops
    from efficientvlm_amd._lib import BF16
    ops.GEMM_PROFILE = []
    trainer.opt.set_schedule(0.0)
    overlap, trainer.overlap_teacher = trainer.overlap_teacher, False     # one stream: a launch is timed with the chip to itself
    trainer._step_eager(batch)
    trainer.overlap_teacher = overlap
    torch.cuda.synchronize()
    recs, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    groups = {}
    for dtype, pt, qt, I, J, K, e0, e1, kern in recs:
        if dtype != BF16:
            continue
        g = groups.setdefault(kern, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += 2.0 * I * J * K
        g[2] += e0.elapsed_time(e1) * 1e-3
    dom = max(groups, key=lambda k: groups[k][2])
    n, fl, tm = groups[dom]
    allfl, alltm = sum(g[1] for g in groups.values()), sum(g[2] for g in groups.values())
    ach = fl / tm / 1e12
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pmc = json.load(f)
        ent = next((v for k, v in pmc.get("kernels", {}).items() if dom.split("<")[0] in k and
                    dom.split("<")[1].replace(" ", "") in k.replace(" ", "")), None)
        if ent:
            traffic = ent["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError, IndexError):
        pass
    return {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "launches": n,
            "avg_launch_us": round(tm / n * 1e6, 2), "flop_per_launch": round(fl / n / 1e9, 3),
            "all_gemm_kernels": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                     "time_ms": round(v[2] * 1e3, 3)} for k, v in groups.items()},
            "all_gemm_tflops": round(allfl / alltm / 1e12, 1), "gemm_time_ms_per_step": round(alltm * 1e3, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE.json configs[1]: 64)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="teacher forward inside the same step as its student step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child()
        return

    # stdout carries exactly ONE line (the JSON result): anything libraries write to fd 1 meanwhile - RCCL prints a
    # version / hostname block there when its first communicator is created - is sent to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dp = bool(os.environ.get("EVLM_FORCE_REDUCE"))      # exercise the N>1 code path (collectives) on one GPU
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" IS RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", local_rank)

    from oracle import synth
    from efficientvlm_amd.trainer import GDTrainer
    geom = synth.GEOMS["full"]
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    student, teacher = build(geom, dev, seed=1234)
    pipelined = not args.no_pipeline
    trainer = GDTrainer(student, teacher, lr=1e-4, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=dtype,
                        use_graph=not args.no_graph, pipeline_teacher=pipelined)
    B = args.batch
    # weak scaling: B per GPU.  FOUR distinct synthetic batches, resident in HBM, fed round-robin: with the teacher
    # pipelined one batch ahead of the student, every step runs the teacher on a batch the student has not seen yet
    batches = [{k: v.to(dev) for k, v in synth.make_batch(geom, B, seed=42 + rank + 1000 * i).items()} for i in range(4)]
    batch = batches[0]
    it = 0
    if pipelined:
        trainer.step(batches[0])          # primes the pipeline (teacher outputs of the first batch); not a step
        it = 1

    for _ in range(args.warmup):
        out = trainer.step(batches[it % 4]); it += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(batches[it % 4]); it += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = [float(x) for x in out.tolist()]

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / elapsed
        res = {"metric": "image-text pairs/sec/node (GD step, X-VLM-small student + base teacher)",
               "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "GeneralDistill general step: X-VLM-small (6+3+3) student fwd+bwd, X-VLM-base (12+6+6) "
                                      "teacher fwd, ITC+ITM+MLM + hidden/attention/logit KD, grad all-reduce, clip 1.0, AdamW",
                          "image": "224x224", "text_len": geom["L"], "masked": geom["M"], "batch_per_gpu": B,
                          "global_batch": B * world, "parallelism": f"dp{world}",
                          "launch": "eager" if (args.no_graph or world > 1 or force_dp) else "hipGraph replay",
                          "teacher_pipelined": pipelined, "distinct_batches": 4,
                          "init": "random (reference init), no checkpoints"},
               "step_model_tflops": round(value * FLOPS_PER_PAIR / 1e12, 1),
               "step_mfma_frac": round(value * FLOPS_PER_PAIR / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
               "last_losses": {"total": losses[0], "itc": losses[1], "itm": losses[2], "mlm": losses[3], "kd": losses[4]}}
        if not args.no_roofline and dtype == torch.bfloat16:
            res["roofline"] = roofline_leg(trainer, batch)
        if world == 1 and pipelined and not args.no_roofline:
            # the same step WITHOUT teacher pipelining (both models on the same batch inside each step), for reference
            del trainer
            s2, t2 = build(geom, dev, seed=1234)
            tr2 = GDTrainer(s2, t2, lr=1e-4, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, dtype=dtype,
                            use_graph=not args.no_graph, pipeline_teacher=False)
            for i in range(3):
                tr2.step(batches[i % 4])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n2 = min(args.steps, 10)
            for i in range(n2):
                tr2.step(batches[i % 4])
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            res["unpipelined"] = {"value": round(B * n2 / e2, 2), "unit": "pairs/s", "ms_per_step": round(e2 / n2 * 1e3, 3),
                                  "steps": n2}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
