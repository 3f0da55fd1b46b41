import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L
dev="cuda"
def t(I,J,K,pt=0,qt=0,reps=20):
    dt=torch.bfloat16
    P=(torch.randn((K,I) if pt else (I,K),device=dev)*0.5).to(dt); Q=(torch.randn((K,J) if qt else (J,K),device=dev)*0.5).to(dt)
    C=torch.empty((I,J),dtype=dt,device=dev)
    f=lambda: ops._gemm(L.BF16,P,Q,C,I,J,K,P.stride(0),Q.stride(0),J,p_trans=pt,q_trans=qt)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/reps*1e3
    print(f"I={I} J={J} K={K}: {us:8.1f} us {2.0*I*J*K/us/1e6:7.1f} TF/s",flush=True)
for K in (64,128,256,768,1536,3072,6144):
    t(12608,3072,K)
for I in (1280,2560,5120,12800):
    t(I,3072,768)
t(12800,768,768); t(12800,2304,768)
