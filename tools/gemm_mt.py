import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvlm_amd import ops, _lib as L
dev="cuda"
def t(I,J,K,pt=0,qt=0,reps=20,**kw):
    dt=torch.bfloat16
    P=(torch.randn((K,I) if pt else (I,K),device=dev)*0.5).to(dt); Q=(torch.randn((K,J) if qt else (J,K),device=dev)*0.5).to(dt)
    C=torch.empty((I,J),dtype=dt,device=dev)
    ex={}
    if kw.get("bias"): ex["bias"]=torch.randn(J,device=dev)
    if kw.get("res"): ex["residual"]=torch.randn((I,J),device=dev).to(dt); ex["ldx"]=J
    f=lambda: ops._gemm(L.BF16,P,Q,C,I,J,K,P.stride(0),Q.stride(0),J,p_trans=pt,q_trans=qt,**ex)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/reps*1e3
    print(f"I={I:6d} J={J:5d} K={K:5d} pt={pt} qt={qt}: {us:8.1f} us {2.0*I*J*K/us/1e6:7.1f} TF/s",flush=True)
for (I,J,K) in [(12608,768,768),(12608,2304,768),(12608,3072,768),(12608,768,3072),(12608,1536,768),(7680,768,768),(7680,2304,768),(7680,3072,768),(7680,768,3072),(3840,768,768),(3840,2304,768),(3840,3072,768),(3840,768,3072)]:
    t(I,J,K,bias=True)
for (I,J,K) in [(12608,768,768),(12608,768,3072),(12608,3072,768),(7680,768,768),(7680,768,3072),(7680,3072,768)]:
    t(I,J,K,0,1)
