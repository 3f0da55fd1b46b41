import sys, os, torch
sys.path.insert(0, os.getcwd())
from efficientvlm_amd import _lib as L
lib = L.load()
x = torch.randn(95_000_000, device="cuda"); out = torch.zeros(1, device="cuda"); ws = torch.zeros(2050, device="cuda")
for w in (None, ws):
    for _ in range(5): lib.evlm_sumsq(L.ptr(x), x.numel(), L.ptr(out), L.ptr(w), L.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.evlm_sumsq(L.ptr(x), x.numel(), L.ptr(out), L.ptr(w), L.stream())
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("EVLM_SUMSQ_BLOCKS", "default"), "workspace" if w is not None else "atomics", round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us")
out.zero_(); lib.evlm_sumsq(L.ptr(x), x.numel(), L.ptr(out), L.ptr(ws), L.stream()); a = out.item()
out.zero_(); lib.evlm_sumsq(L.ptr(x), x.numel(), L.ptr(out), L.ptr(ws), L.stream()); b = out.item()
print("deterministic:", a == b, a, float((x.double() ** 2).sum()))
