#!/usr/bin/env python3
"""Burst against sustained rate of one GEMM shape: the same launch repeated for ~0.5 s, TFLOP/s per chunk of launches
(HIP events).  tools/gemm_bench.py times 20 back-to-back launches out of an idle chip (boost clock); a training step keeps
the matrix pipes busy for good and the chip settles at its power-limited clock.

    python tools/gemm_sustained.py [I J K] [--chunks 12] [--chunk 400]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L

a = [int(x) for x in sys.argv[1:] if x.isdigit()]
I, J, K = (a + [12608, 768, 3072])[:3] if len(a) >= 3 else (12608, 768, 3072)
chunks = int(sys.argv[sys.argv.index("--chunks") + 1]) if "--chunks" in sys.argv else 12
chunk = int(sys.argv[sys.argv.index("--chunk") + 1]) if "--chunk" in sys.argv else 400
dev = "cuda"; dt = torch.bfloat16
P = (torch.randn(I, K, device=dev) * 0.5).to(dt); Q = (torch.randn(J, K, device=dev) * 0.5).to(dt)
C = torch.empty(I, J, dtype=dt, device=dev)
f = lambda: ops._gemm(L.BF16, P, Q, C, I, J, K, K, K, J, p_trans=0, q_trans=0)
for _ in range(3): f()
torch.cuda.synchronize()
import time; time.sleep(0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
ev[0].record()
for c in range(chunks):
    for _ in range(chunk): f()
    ev[c + 1].record()
torch.cuda.synchronize()
t = 0.0
print(f"# I={I} J={J} K={K}: {chunk} launches per chunk")
for c in range(chunks):
    ms = ev[c].elapsed_time(ev[c + 1]); t += ms
    print(f"t = {t:7.1f} ms   {ms / chunk * 1e3:7.1f} us/launch   {2.0 * I * J * K * chunk / ms / 1e9:7.1f} TF/s")
