"""Where do the small copies / fills of one eager GD step come from?  (python-level call-site counting)"""
import sys, os, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import synth
from efficientvlm_amd.trainer import GDTrainer
import bench
geom = synth.GEOMS["full"]; dev = torch.device("cuda")
s, t = bench.build(geom, dev, 1234)
tr = GDTrainer(s, t, dtype=torch.bfloat16, use_graph=False)
batch = {k: v.to(dev) for k, v in synth.make_batch(geom, 64, seed=42).items()}
for _ in range(2): tr.step(batch)
torch.cuda.synchronize()
agg = collections.Counter()
ON = [False]
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "efficientvlm_amd" in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.line[:70]}"
    return "?"
def wrap(obj, name, label, pred=None):
    orig = getattr(obj, name)
    def w(*a, **k):
        if ON[0] and (pred is None or pred(*a, **k)):
            agg[(label, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)
wrap(torch, "zeros", "zeros"); wrap(torch, "zeros_like", "zeros_like"); wrap(torch, "cat", "cat"); wrap(torch, "ones", "ones")
wrap(torch.Tensor, "copy_", "copy_"); wrap(torch.Tensor, "clone", "clone"); wrap(torch.Tensor, "zero_", "zero_")
wrap(torch.Tensor, "contiguous", "contiguous(copy)", lambda self, *a, **k: not self.is_contiguous())
wrap(torch.Tensor, "to", "to(convert)", lambda self, *a, **k: any(isinstance(x, torch.dtype) and x != self.dtype for x in list(a) + list(k.values())))
wrap(torch.Tensor, "float", "float()", lambda self: self.dtype != torch.float32)
wrap(torch.Tensor, "reshape", "reshape(copy)", lambda self, *a: not self.is_contiguous())
ON[0] = True
tr.step(batch); torch.cuda.synchronize()
ON[0] = False
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:5d}  {name:18s} {where}")
