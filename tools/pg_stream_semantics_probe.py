#!/usr/bin/env python3
"""Which stream does a process-group collective make WAIT, on this torch / RCCL stack?  One-rank RCCL group; the group's
own stream (found among torch's pool streams) is kept busy by a 10 ms spin kernel, a collective is issued from stream R
(a side stream) in several forms, and HIP events tell when the DEFAULT stream and R got past the call.
    python tools/pg_stream_semantics_probe.py"""
import os, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
CYC = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
x = torch.ones(1 << 20, device=dev); y = torch.ones(1 << 20, device=dev)
# everything below runs on pool streams: the legacy default stream has implicit-synchronisation rules of its own
M = torch.cuda.Stream()
with torch.cuda.stream(M):
    w = dist.all_reduce(x, async_op=True); w.wait()
torch.cuda.synchronize()
ns = None
for _ in range(40):
    s_ = torch.cuda.Stream()
    if s_.cuda_stream == M.cuda_stream:
        continue
    torch.cuda.synchronize()
    with torch.cuda.stream(s_):
        torch.cuda._sleep(int(15e3 * CYC))
    t0 = time.perf_counter()
    with torch.cuda.stream(M):
        w = dist.all_reduce(x, async_op=True); w.wait()
    M.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dt > 8e-3:
        ns = s_; break
print("the group's stream", "FOUND among torch's pool streams" if ns is not None else "NOT among torch's pool streams", flush=True)
assert ns is not None
R = torch.cuda.Stream()
while R.cuda_stream in (ns.cuda_stream, M.cuda_stream):
    R = torch.cuda.Stream()


def case(name, issue, main, issuer):
    torch.cuda.synchronize()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record(main)
    with torch.cuda.stream(ns):
        torch.cuda._sleep(int(10e3 * CYC))                  # the group's stream is busy for 10 ms
    if issuer is not main:
        issuer.wait_stream(main)
    with torch.cuda.stream(issuer):
        h = issue()
        c.record(issuer)
    b.record(main)
    torch.cuda.synchronize()
    print(f"{name:58s}: caller's stream past the call after {a.elapsed_time(b):6.2f} ms, issuing stream after {a.elapsed_time(c):6.2f} ms", flush=True)
    return h


def coalesced(async_ops):
    with dist._coalescing_manager(async_ops=async_ops) as cm:
        dist.all_reduce(x); dist.all_reduce(y)
    return cm


D = torch.cuda.default_stream()
for label, main in (("caller on a pool stream M", M), ("caller on the DEFAULT stream", D)):
    print("---", label, flush=True)
    with torch.cuda.stream(main):
        case("all_reduce(async_op=True) from side stream R", lambda: dist.all_reduce(x, async_op=True), main, R)
        case("all_reduce(async_op=False) from R", lambda: dist.all_reduce(x), main, R)
        case("coalescing manager (async_ops=True) from R", lambda: coalesced(True), main, R)
        case("all_gather (sync) from R", lambda: dist.all_gather([torch.empty_like(x)], x), main, R)
        case("all_reduce(async_op=True) from the caller's stream", lambda: dist.all_reduce(x, async_op=True), main, main)
        case("coalescing manager (async_ops=True) from the caller's stream", lambda: coalesced(True), main, main)
dist.destroy_process_group()
