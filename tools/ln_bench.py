import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import _lib as L
lib = L.load(); dev = "cuda"
for rows, d in [(12608, 768), (7680, 768), (3840, 768), (1920, 768)]:
    x = torch.randn(rows, d, device=dev).bfloat16(); dy = torch.randn(rows, d, device=dev).bfloat16(); dx = torch.empty_like(x)
    g = torch.ones(d, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
    dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
    ws = torch.empty(lib.evlm_layernorm_bwd_blocks(rows) * 2 * d, device=dev)
    f = lambda: L.check(lib.evlm_layernorm_bwd(L.BF16, L.ptr(dy), L.ptr(x), L.ptr(g), L.ptr(mean), L.ptr(rstd), rows, d, L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(ws), L.stream()), "ln")
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"rows={rows} d={d}: {us:.1f} us  {rows*d*2*3/us/1e6:.2f} TB/s")
