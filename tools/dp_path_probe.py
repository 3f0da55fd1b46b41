#!/usr/bin/env python3
"""Where the N > 1 code path of the GD step spends its time on ONE GPU (RCCL group of one rank, ITC gather forced through
the collective): hipGraph segments with the backward cut at the gradient-stage hooks (EVLM_DP_CUTS) against the uncut form,
the teacher forked inside a segment or replayed as its own graph (EVLM_SEG_TEACHER), the eager student step, and the
single-GPU joint graph.

A one-rank all-reduce moves nothing, so the exchange is SIMULATED for the overlap question: `--wire-gbps G` makes every
all-reduce additionally hold its stream for bytes / G (a one-thread spin kernel: the timeline of a ring all-reduce at that
algorithm bandwidth, without its CU / HBM contention).  "exchange hidden" = (segments + simulated wire) - (segments, no
exchange).

    EVLM_FORCE_REDUCE=1 python tools/dp_path_probe.py [--wire-gbps 170] [--steps 12] [--only tag,tag]"""
import argparse, contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EVLM_FORCE_REDUCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
import torch, torch.distributed as dist
import bench
from efficientvlm_amd.workload import GEOMS, make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--wire-gbps", type=float, default=170.0, help="simulated all-reduce algorithm bandwidth (8-GPU ring "
                "over 7 xGMI links at ~300 GB/s bus bandwidth: ~170 GB/s)")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--only", default="")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
geom = GEOMS["full"]; dev = torch.device("cuda", 0)
batches = [{k: v.to(dev) for k, v in make_batch(geom, 64, seed=42 + 1000 * i).items()} for i in range(4)]

# spin-kernel calibration (cycles per microsecond of torch.cuda._sleep)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
CYC_PER_US = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
_real_all_reduce = dist.all_reduce
SIM = {"on": False, "bytes": 0}


def all_reduce_sim(t, *a, **kw):
    h = _real_all_reduce(t, *a, **kw)
    if SIM["on"]:
        nb = t.numel() * t.element_size()
        SIM["bytes"] += nb
        torch.cuda._sleep(int(nb / (args.wire_gbps * 1e3) * CYC_PER_US))     # bytes / (GB/s) = ns * 1e-... -> us
    return h


dist.all_reduce = all_reduce_sim


def run(tag, env=None, patch=None, sim=False):
    if args.only and tag.split(":")[0] not in args.only.split(","):
        return
    saved = {}
    for k, v in (env or {}).items():
        saved[k] = os.environ.get(k)
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    s, t = bench.build(geom, dev, 1234)
    tr = bench.make_trainer(s, t, torch.bfloat16, True, True)
    if patch: patch(tr)
    if os.environ.get("EVLM_PROBE_SEGMENT_TIMES") and tr.reducer.active:
        time_segments(tr)
    SIM["on"], SIM["bytes"] = sim is True, 0
    it = 0
    # EVLM_PROBE_STREAM=1: the training loop on a pool stream instead of the (legacy) default stream
    loop_stream = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("EVLM_PROBE_STREAM") else contextlib.nullcontext()
    with loop_stream:
        for _ in range(7):
            tr.step(batches[it % 4]); it += 1
        torch.cuda.synchronize(); SIM["bytes"] = 0; t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(batches[it % 4]); it += 1
        host = time.perf_counter() - t0
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    wire = f"  simulated wire {SIM['bytes'] / args.steps / 1e6:6.1f} MB = {SIM['bytes'] / args.steps / (args.wire_gbps * 1e6):5.2f} ms/step" if sim else ""
    mode = "segments" if (tr._seg and not getattr(tr, "_segments_broken", False)) else ("joint graph" if tr._joint else "eager")
    print(f"{tag:58s} {el / args.steps * 1e3:7.2f} ms/step   (host issue {host / args.steps * 1e3:6.2f} ms) [{mode}]{wire}", flush=True)
    SIM["on"] = False
    if getattr(tr, "_seg_times", None):
        report_segments(tr, tag.split(":")[0])
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    del tr, s, t


def sleep_only(tr):
    """the exchange as NOTHING BUT the simulated wire on the reducer's stream (no process-group call at all): separates what
    the collectives' own stream bookkeeping costs from what a long kernel beside the segments costs"""
    red = tr.reducer

    def reduce_async(tensors):
        red.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(red.stream):
            for t in red._buckets(tensors):
                nb = t.numel() * t.element_size()
                SIM["bytes"] += nb
                torch.cuda._sleep(int(nb / (args.wire_gbps * 1e3) * CYC_PER_US))

    def finish():
        torch.cuda.current_stream().wait_stream(red.stream)
    red.reduce_async, red.finish = reduce_async, finish


def sleep_fresh(tr):
    """... and on a stream created just now, ordered behind the segment by an event"""
    red = tr.reducer
    red.stream = torch.cuda.Stream()
    sleep_only(tr)


def time_segments(tr):
    """HIP events on the step's stream around every item of the segment chain (printed for the last step)"""
    def replay(segs, last_reduce):
        evs = [torch.cuda.Event(enable_timing=True)]
        evs[0].record()
        names = []
        for kind, item in segs:
            if kind == "graph":
                item.replay()
            elif kind == "gather":
                dist.all_gather(item[0], item[1])
            else:
                tr.reducer.reduce_async(item)
                if item is last_reduce:
                    tr.reducer.finish()
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); names.append(kind)
        tr._seg_times = (names, evs)
    tr._replay_segments = replay


def report_segments(tr, tag):
    names, evs = tr._seg_times
    torch.cuda.synchronize()
    print(f"#   {tag}: " + "  ".join(f"{n} {evs[i].elapsed_time(evs[i + 1]):.2f}" for i, n in enumerate(names)), flush=True)


_NCCL_ALIAS = []


def find_nccl_stream():
    """the process group launches its kernels on an internal stream it takes from torch's pool of 32 - the pool
    torch.cuda.Stream() cycles through.  Find OUR handle of that stream: the one whose pending sleep delays a collective."""
    if _NCCL_ALIAS:
        return _NCCL_ALIAS[0]
    x = torch.ones(4, device=dev)
    w = _real_all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize()        # (the group's stream exists now)
    for _ in range(40):
        s_ = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(s_):
            torch.cuda._sleep(int(15e3 * CYC_PER_US))
        t0 = time.perf_counter()
        w = _real_all_reduce(x, async_op=True); w.wait()
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        if dt > 8e-3:
            _NCCL_ALIAS.append(s_)
            print(f"# the process group's stream found among torch's pool streams (a collective behind a 15 ms sleep on it took {dt * 1e3:.1f} ms)", flush=True)
            return s_
    raise RuntimeError("the process group's stream is not one of torch's pool streams")


def sim_on_nccl(from_main):
    """the faithful form: the spin kernels sit ON the process group's own stream, in front of each stage's (one-rank, empty)
    all-reduce - where the RCCL kernels of a real run execute.  from_main: the collectives are issued from the step's
    stream (no reducer stream in the chain) instead of from the reducer's."""
    def patch(tr):
        red, ns = tr.reducer, find_nccl_stream()
        if from_main:
            red.stream = None
        real = red.reduce_async

        def reduce_async(tensors):
            src = red.stream if red.stream is not None else torch.cuda.current_stream()
            if red.stream is not None:
                red.stream.wait_stream(torch.cuda.current_stream())
            ns.wait_stream(src)
            with torch.cuda.stream(ns):
                for t in red._buckets(tensors):
                    nb = t.numel() * t.element_size()
                    SIM["bytes"] += nb
                    torch.cuda._sleep(int(nb / (args.wire_gbps * 1e3) * CYC_PER_US))
            real(tensors)
        red.reduce_async = reduce_async
    return patch


def no_reduce(tr):
    tr.reducer.reduce_async = lambda tensors: None
    tr.reducer.finish = lambda: None


print(f"# spin calibration: {CYC_PER_US:.0f} cycles/us; simulated wire {args.wire_gbps:.0f} GB/s; "
      f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}", flush=True)
for rep in range(args.reps):
    run("joint: single-GPU joint graph (no process-group path)", env={"EVLM_FORCE_REDUCE": None})
    run("cuts_all: segments, cuts=all, teacher forked in two halves, fp32 wire", env={"EVLM_DP_CUTS": "all"})
    run("cuts_all_sim: same + simulated wire", env={"EVLM_DP_CUTS": "all"}, sim=True)
    run("nosplit: segments, cuts=all, whole teacher forward behind the gather", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER_SPLIT": "0"})
    run("nosplit_sim: same + simulated wire", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER_SPLIT": "0"}, sim=True)
    run("cuts_all_noex: same, no gradient exchange", env={"EVLM_DP_CUTS": "all"}, patch=no_reduce)
    # `late` placement (round 6: the teacher's fusion pass resumed layer by layer beside the ViT-backward segments)
    late = {"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER_SPLIT": "late"}
    run("late: segments, teacher image encoder behind the gather, text / fusion layers beside the ViT backward", env=late)
    run("late_sim: same + simulated wire", env=late, sim=True)
    run("late_noex: same, no gradient exchange", env=late, patch=no_reduce)
    run("late5: round 5's late plan (text pass | whole fusion pass | heads)", env=dict(late, EVLM_SEG_LATE_PLAN="text_done,fusion_done"))
    run("cuts_all_sleep: the simulated wire alone (no process-group calls)", env={"EVLM_DP_CUTS": "all"}, patch=sleep_only, sim=True)
    run("cuts_all_sleepfresh: same on a fresh stream", env={"EVLM_DP_CUTS": "all"}, patch=sleep_fresh, sim=True)
    run("cuts_all_nccl: simulated wire ON the process group's stream", env={"EVLM_DP_CUTS": "all"}, patch=sim_on_nccl(False), sim="nccl")
    run("cuts_all_nccl_main: same, collectives issued from the step's stream", env={"EVLM_DP_CUTS": "all"}, patch=sim_on_nccl(True), sim="nccl")
    run("tgraph_noex: teacher as its own graph on the side stream, no exchange", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"}, patch=no_reduce)
    run("tgraph_nccl: same + simulated wire on the group's stream", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"}, patch=sim_on_nccl(False), sim="nccl")

    def serial_teacher(inner):
        def patch(tr):
            tr._side = torch.cuda.current_stream()         # the teacher graph replays IN FRONT of the segments, same stream
            inner(tr)
        return patch
    run("tserial_noex: teacher graph serial on the step's stream, no exchange", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"}, patch=serial_teacher(no_reduce))
    run("tserial_nccl: same + simulated wire on the group's stream", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"}, patch=serial_teacher(sim_on_nccl(False)), sim="nccl")
    run("tserial_nccl_main: same, collectives issued from the step's stream", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"}, patch=serial_teacher(sim_on_nccl(True)), sim="nccl")
    for vc in ("4", "3", "2"):
        run(f"cuts_v{vc}: segments, ONE ViT cut at layer {vc}", env={"EVLM_DP_VIT_CUTS": vc})
        run(f"cuts_v{vc}_sim: same + simulated wire", env={"EVLM_DP_VIT_CUTS": vc}, sim=True)
    run("cuts_421: segments, ViT cuts at 4,2,1", env={"EVLM_DP_VIT_CUTS": "4,2,1"})
    run("cuts_421_sim: same + simulated wire", env={"EVLM_DP_VIT_CUTS": "4,2,1"}, sim=True)
    for fk in ("start", "vision_done", "text_done"):
        for jn in ("end", "vision"):
            run(f"joint_{fk}_{jn}: joint graph, teacher fork={fk} join={jn}",
                env={"EVLM_FORCE_REDUCE": None, "EVLM_TEACHER_FORK": fk, "EVLM_TEACHER_JOIN": jn})
    run("cuts_vit: segments, cuts=vit (first send after ViT 5,4)", env={"EVLM_DP_CUTS": "vit"})
    run("cuts_vit_sim: same + simulated wire", env={"EVLM_DP_CUTS": "vit"}, sim=True)
    run("cuts_none: segments, no cut in backward (round-2 form)", env={"EVLM_DP_CUTS": "none"})
    run("cuts_none_sim: same + simulated wire", env={"EVLM_DP_CUTS": "none"}, sim=True)
    run("tgraph: segments, cuts=all, teacher as its own graph", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"})
    run("tgraph_sim: same + simulated wire", env={"EVLM_DP_CUTS": "all", "EVLM_SEG_TEACHER": "graph"}, sim=True)
    run("bf16: segments, cuts=all, bf16 wire (opt-in) + simulated wire", env={"EVLM_DP_CUTS": "all", "EVLM_BF16_WIRE": "1"}, sim=True)
    run("eager: eager student step (fallback), hooks send the stages", env={"EVLM_NO_SEGMENT_GRAPHS": "1"})
dist.destroy_process_group()
