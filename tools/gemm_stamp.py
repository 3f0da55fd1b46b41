import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from efficientvlm_amd import ops, _lib as L
dev="cuda"
def run(I,J,K,pt=0,qt=0):
    dt=torch.bfloat16
    P=(torch.randn((K,I) if pt else (I,K),device=dev)*0.5).to(dt); Q=(torch.randn((K,J) if qt else (J,K),device=dev)*0.5).to(dt)
    C=torch.empty((I,J),dtype=dt,device=dev)
    ntile=((I+127)//128)*((J+127)//128)
    st=torch.zeros(ntile*4+16,dtype=torch.int64,device=dev)
    for _ in range(3):
        ops._gemm(L.BF16,P,Q,C,I,J,K,P.stride(0),Q.stride(0),J,p_trans=pt,q_trans=qt,psum=st)
    torch.cuda.synchronize()
    s=st.cpu().numpy()[:ntile*4].reshape(ntile,4).astype(np.float64)
    pro=(s[:,1]-s[:,0]); main=(s[:,2]-s[:,1]); epi=(s[:,3]-s[:,2]); tot=(s[:,3]-s[:,0])
    span=(s[:,3].max()-s[:,0].min())
    print(f"I={I} J={J} K={K} pt={pt} qt={qt}: tiles={ntile} cycles(100MHz ticks? raw): prologue med={np.median(pro):.0f} main med={np.median(main):.0f} epilogue med={np.median(epi):.0f} total med={np.median(tot):.0f}  span={span:.0f}")
    order=np.argsort(s[:,0]); 
    print("   first starts:", (s[order[:5],0]-s[:,0].min()).astype(int), " 512th start:", int(s[order[min(512,ntile-1)],0]-s[:,0].min()), "last start:", int(s[order[-1],0]-s[:,0].min()))
os.environ["EVLM_FORCE_MT"]="4"
run(12608,3072,768); run(12608,768,3072); run(12608,2304,768); run(4096,4096,4096)
