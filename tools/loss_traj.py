import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import torch
from oracle import synth
from efficientvlm_amd.trainer import GDTrainer
import bench
geom=synth.GEOMS["full"]; dev=torch.device("cuda")
B=int(sys.argv[2]) if len(sys.argv)>2 else 16
for mode in sys.argv[1].split(","):
    s,t=bench.build(geom,dev,1234)
    tr=GDTrainer(s,t,dtype=torch.bfloat16,use_graph=(mode=="graph"))
    batch={k:v.to(dev) for k,v in synth.make_batch(geom,B,seed=42).items()}
    outs=[]
    for i in range(8):
        outs.append([round(float(x),4) for x in tr.step(batch).tolist()])
    print(mode, "gnorm", float(tr.opt.grad_norm()))
    for o in outs: print("   ", o)
