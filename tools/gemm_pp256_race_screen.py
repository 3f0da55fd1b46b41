#!/usr/bin/env python3
"""Race screen for the 256x256 ping-pong GEMM schedules: many repetitions over several shapes (forward, dX, grouped dW),
every result compared with an fp32 torch product of the same bf16 inputs.  A sync-structure edit must pass this."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L
dev = "cuda"; torch.manual_seed(1)
bad = 0
def check(C, ref, what):
    global bad
    e = float((C.float() - ref).abs().max()); s = float(ref.abs().max())
    if not (e <= 1.5e-2 * s):
        bad += 1; print("MISMATCH", what, e, s, flush=True)
shapes = [(4096, 4096, 128), (4096 + 40, 2304, 768), (12608, 768, 3072), (7680, 3072, 768), (2560, 2560, 192), (12608, 2304, 768)]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for (I, J, K) in shapes:
        P = (torch.randn((I, K), device=dev) * 0.5).bfloat16(); Q = (torch.randn((J, K), device=dev) * 0.05).bfloat16()
        C = torch.empty((I, J), dtype=torch.bfloat16, device=dev)
        ops._gemm(L.BF16, P, Q, C, I, J, K, K, K, J)
        check(C, P.float() @ Q.float().t(), ("NN", I, J, K, rep))
        Qt = Q.t().contiguous()
        ops._gemm(L.BF16, P, Qt, C, I, J, K, K, J, J, q_trans=1)
        check(C, P.float() @ Qt.float(), ("NT", I, J, K, rep))
    # grouped weight gradients
    Kr = 1600 + 64 * (rep % 5)
    probs = []
    for (I, J) in [(768, 768), (2304, 768), (768, 3072), (520, 264)]:
        dY = (torch.randn((Kr, I), device=dev) * 0.5).bfloat16(); X = (torch.randn((Kr, J), device=dev) * 0.5).bfloat16()
        probs.append((dY, X, torch.zeros((I, J), device=dev), torch.zeros(I, device=dev)))
    arr = (L.WgradProblem * len(probs))()
    for k, (dY, X, Cm, ps) in enumerate(probs):
        arr[k].P, arr[k].Q, arr[k].C, arr[k].psum = dY.data_ptr(), X.data_ptr(), Cm.data_ptr(), ps.data_ptr()
        arr[k].I, arr[k].J, arr[k].ldp, arr[k].ldq, arr[k].ldc = dY.shape[1], X.shape[1], dY.shape[1], X.shape[1], X.shape[1]
    L.check(L.load().evlm_wgrad_grouped(arr, len(probs), Kr, L.stream()), "wgrad_grouped")
    for n, (dY, X, Cm, ps) in enumerate(probs):
        ref = dY.float().t() @ X.float()
        e = float((Cm - ref).norm() / ref.norm())
        if not (e < 2e-5): bad += 1; print("MISMATCH grouped", n, rep, e, flush=True)
torch.cuda.synchronize()
print("race screen:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
