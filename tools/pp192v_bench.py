#!/usr/bin/env python3
"""HISTORICAL (profiles/r03_pp192v_staging.md): A/B of the register-staged 192 x 256 GEMM (gemm_bf16_pp192v_kernel, in the
tree at commit 9e6a81c only: EVLM_PP192V=1, EVLM_PP192V_DIAG=0/1/2) against the shipping LDS-DMA form on the GD step's
shapes; one child process per variant (the switches are read once).  On the current tree every variant runs the shipping
kernel."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("ViT FC2", 12608, 768, 3072, 0), ("ViT out-proj", 12608, 768, 768, 0), ("QKV dX", 12608, 768, 2304, 1),
          ("FC1 dX", 12608, 768, 3072, 1), ("text FC1 3840", 3840, 3072, 768, 0), ("K=8192 probe", 12608, 768, 8192, 0)]


def child():
    import torch
    from efficientvlm_amd import ops, _lib as L
    dev = "cuda"
    torch.manual_seed(0)
    tag = f"V={os.environ.get('EVLM_PP192V', '0')} DIAG={os.environ.get('EVLM_PP192V_DIAG', '0')}"
    for name, I, J, K, qt in SHAPES:
        dt = torch.bfloat16
        P = (torch.randn((I, K), device=dev) * 0.5).to(dt)
        Q = (torch.randn((K, J) if qt else (J, K), device=dev) * 0.05).to(dt)
        Cm = torch.empty((I, J), dtype=dt, device=dev)
        f = lambda: ops._gemm(L.BF16, P, Q, Cm, I, J, K, K, J if qt else K, J, q_trans=qt)
        f(); torch.cuda.synchronize()
        kern = L.load().evlm_gemm_last_kernel().decode()
        ref = P.float() @ (Q.float() if qt else Q.float().t())
        err = float((Cm.float() - ref).abs().max() / ref.abs().max())
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"{tag:12s} {name:16s} I={I:6d} J={J:5d} K={K:5d} qt={qt} {us:8.1f} us {2.0 * I * J * K / us / 1e6:7.1f} TF/s  "
              f"relerr {err:.2e} {'OK' if err < 1.2e-2 else 'WRONG (expected for DIAG 1/2)'}  [{kern}]", flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for rep in range(2):
            for v, d in (("0", "0"), ("1", "0"), ("1", "1"), ("1", "2")):
                env = dict(os.environ, EVLM_PP192V=v, EVLM_PP192V_DIAG=d)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
