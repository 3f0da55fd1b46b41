#!/usr/bin/env python3
"""Forward + backward of one long-sequence ViT self-attention layer (64 x 12 x 577, or `B L` on the command line), launched
eagerly N times - the thing to put under rocprofv3 (kernel trace or one --pmc pass) when working on the streaming kernels:
     rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d out -o p -- python3 tools/attn_long_probe.py 64 577 5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 577
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5
KD = "--kd" in sys.argv           # fused map distillation against the teacher's recipe (Q, K, row lse): the ITR / VQA flavour
H, dh, d = 12, 64, 768
torch.manual_seed(0)
x = (torch.randn(B, L, 3 * d, device="cuda") * 0.5).bfloat16().requires_grad_(True)
gO = torch.randn(B, L, d, device="cuda").bfloat16()
rec = None
if KD:
    with torch.no_grad():
        rec = ops.self_attention_recipe((torch.randn(B, L, 3 * d, device="cuda") * 0.5).bfloat16(), H, dh, 0.125)[1]
for _ in range(N):
    if KD:
        O, _, k = ops.self_attention(x, H, dh, 0.125, want_probs=False, kd_teacher=rec, kd_weight=1.0)
        torch.autograd.grad((O.float() * gO.float()).sum() + k, x)
    else:
        O, _ = ops.self_attention(x, H, dh, 0.125, want_probs=False)
        torch.autograd.grad(O, x, gO)
torch.cuda.synchronize()
print("done")
