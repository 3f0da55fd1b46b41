import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import torch
from oracle import synth
from efficientvlm_amd import ops
from efficientvlm_amd.trainer import GDTrainer
import bench
geom=synth.GEOMS["full"]; dev=torch.device("cuda")
s,t=bench.build(geom,dev,1234)
tr=GDTrainer(s,t,dtype=torch.bfloat16,use_graph=False)
batch={k:v.to(dev) for k,v in synth.make_batch(geom,64,seed=42).items()}
for _ in range(2): tr.step(batch)
torch.cuda.synchronize()
ops.GEMM_PROFILE=[]
tr.opt.set_schedule(0.0); tr._step_eager(batch); torch.cuda.synchronize()
recs,ops.GEMM_PROFILE=ops.GEMM_PROFILE,None
agg={}
for dt,pt,qt,I,J,K,e0,e1,kn in recs:
    k=(dt,pt,qt,I,J,K,kn); a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=e0.elapsed_time(e1)*1e-3
tot=sum(v[1] for v in agg.values())
print(f"total gemm time {tot*1e3:.2f} ms, {len(recs)} launches")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    dt,pt,qt,I,J,K,kn=k
    fl=2.0*I*J*K*v[0]
    print(f"dt={dt} pt={pt} qt={qt} I={I:6d} J={J:6d} K={K:6d} n={v[0]:3d} time={v[1]*1e3:7.3f} ms ({v[1]/tot*100:4.1f}%) {fl/v[1]/1e12:7.1f} TF/s {kn}")
