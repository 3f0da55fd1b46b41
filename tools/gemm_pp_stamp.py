"""In-kernel stamps of the 256x256 ping-pong GEMM (library built with -DPP_STAMP; s_memtime ticks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from efficientvlm_amd import ops, _lib as L
dev = "cuda"
def run(I, J, K, res=False):
    dt = torch.bfloat16
    P = (torch.randn((I, K), device=dev) * 0.5).to(dt); Q = (torch.randn((J, K), device=dev) * 0.05).to(dt)
    C = torch.empty((I, J), dtype=dt, device=dev)
    b = torch.randn(J, device=dev)
    extra = {}
    if res: extra = dict(residual=torch.randn((I, J), device=dev).to(dt), ldx=J)
    st = torch.zeros(256 * 4 * 6, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        st.zero_()
        e0.record()
        ops._gemm(L.BF16, P, Q, C, I, J, K, K, K, J, bias=b, psum=st, **extra)
        e1.record()
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 4, 6).astype(np.float64)
    t00 = s[:, 0, 5][s[:, 0, 5] > 0].min()
    print(f"I={I} J={J} K={K} res={res}: wall {e0.elapsed_time(e1)*1e3:.1f} us; span {s[:,:,4].max()-t00:.0f} ticks")
    for n in range(4):
        v = s[:, n, :]; m = v[:, 0] > 0
        if not m.any(): break
        v = v[m]
        print(f"   tile#{n} ({m.sum():3d} wgs): start@{np.median(v[:,0]-t00):7.0f} kloop {np.median(v[:,1]-v[:,0]):6.0f}  next-issue {np.median(v[:,2]-v[:,1]):5.0f}"
              f"  epilogue-issue {np.median(v[:,3]-v[:,2]):6.0f}  drain {np.median(v[:,4]-v[:,3]):6.0f}   end@ med {np.median(v[:,4]-t00):7.0f} max {np.max(v[:,4]-t00):7.0f}")
def run_tt(I, J, K, acc=True):
    dt = torch.bfloat16
    P = (torch.randn((K, I), device=dev) * 0.5).to(dt); Q = (torch.randn((K, J), device=dev) * 0.5).to(dt)
    C = torch.zeros((I, J), dtype=torch.float32, device=dev)
    st = torch.zeros(256 * 4 * 6, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        st.zero_()
        e0.record()
        ops._gemm(L.BF16, P, Q, C, I, J, K, I, J, J, p_trans=1, q_trans=1, c_f32=1, psum=st, accumulate=int(acc))
        e1.record()
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 4, 6).astype(np.float64)
    print(f"TT I={I} J={J} K={K}: wall {e0.elapsed_time(e1)*1e3:.1f} us")
    for n in range(4):
        v = s[:, n, :]; m = v[:, 0] > 0
        if not m.any(): break
        v = v[m]
        print(f"   item#{n} ({m.sum():3d} wgs): kloop {np.median(v[:,1]-v[:,0]):6.0f}  next-issue {np.median(v[:,2]-v[:,1]):5.0f}"
              f"  epilogue-issue {np.median(v[:,3]-v[:,2]):6.0f}  drain {np.median(v[:,4]-v[:,3]):6.0f}")
pass
for shp in [(12608, 2304, 768, False), (12608, 768, 768, True), (12608, 3072, 768, False), (12608, 768, 3072, True), (7680, 768, 768, True), (7680, 3072, 768, False)]:
    run(shp[0], shp[1], shp[2], res=shp[3])
