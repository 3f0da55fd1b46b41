#!/usr/bin/env python3
"""Does work on stream B run BESIDE a long kernel on stream A?  (the gradient exchange of the N > 1 step is a long-running
kernel on its own stream beside hipGraph segments: profiles/r05_exchange_overlap.md)
   B's work: eager kernels / a hipGraph replay of one chain / a hipGraph replay with a forked branch inside."""
import torch, time
dev = "cuda"
x = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
w = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
A, B, C = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
CYC = 20_000_000 / (e0.elapsed_time(e1) * 1e3)


def chain(n=40):
    y = x
    for _ in range(n):
        y = torch.mm(y, w)
    return y


def forked(n=20):
    C.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(C):
        a = chain(n)
    b = chain(n)
    torch.cuda.current_stream().wait_stream(C)
    return a, b


def make_graph(body):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(B):
        body(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=B):
            body()
    return g


def timed(run_b, sleep_ms):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if sleep_ms:
        with torch.cuda.stream(A):
            torch.cuda._sleep(int(sleep_ms * 1e3 * CYC))
    with torch.cuda.stream(B):
        run_b()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


g1, g2 = make_graph(chain), make_graph(forked)
forms = {"eager chain": chain, "graph, one chain": g1.replay, "graph, forked branch inside": g2.replay,
         "graph x3 back to back": lambda: (g1.replay(), g1.replay(), g1.replay())}
for name, f in forms.items():
    for _ in range(2):
        timed(f, 0)
    alone = min(timed(f, 0) for _ in range(3))
    both = min(timed(f, 3.0) for _ in range(3))
    print(f"{name:32s}: alone {alone:6.2f} ms, beside a 3.0 ms kernel on another stream {both:6.2f} ms "
          f"-> {'OVERLAPPED' if both < alone + 1.0 else 'SERIALISED' if both > alone + 2.5 else 'partly'}")
# ... and when stream A first WAITS for stream B (as the reducer's stream waits for the segment that produced the gradients)
for name, f in forms.items():
    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(B):
            f()
        A.wait_stream(B)
        with torch.cuda.stream(A):
            torch.cuda._sleep(int(3.0 * 1e3 * CYC))
        with torch.cuda.stream(B):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    run(); t = min(run() for _ in range(3))
    alone = min(timed(f, 0) for _ in range(3))
    print(f"{name:32s}: [B work][A waits for B, 3.0 ms kernel on A][B work again] = {t:6.2f} ms (2 x alone = {2 * alone:6.2f})")
# ... and with B's work issued on the DEFAULT (null) stream, as a training loop that never sets a stream does
def timed0(run_b, sleep_ms):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if sleep_ms:
        with torch.cuda.stream(A):
            torch.cuda._sleep(int(sleep_ms * 1e3 * CYC))
    run_b()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for name, f in forms.items():
    for _ in range(2):
        timed0(f, 0)
    alone = min(timed0(f, 0) for _ in range(3))
    both = min(timed0(f, 3.0) for _ in range(3))
    print(f"default stream, {name:32s}: alone {alone:6.2f} ms, beside a 3.0 ms kernel on another stream {both:6.2f} ms "
          f"-> {'OVERLAPPED' if both < alone + 1.0 else 'SERIALISED' if both > alone + 2.5 else 'partly'}")
# ... and when the long kernel's stream first waits for the default stream (reduce_async: stream.wait_stream(current))
for name, f in forms.items():
    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        A.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(A):
            torch.cuda._sleep(int(3.0 * 1e3 * CYC))
        f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    run(); t = min(run() for _ in range(3))
    alone = min(timed0(f, 0) for _ in range(3))
    print(f"default stream, {name:32s}: [work][A waits, 3.0 ms kernel on A][work again] = {t:6.2f} ms (2 x alone = {2 * alone:6.2f})")
