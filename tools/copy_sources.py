"""Which host call sites issue the small ATen device ops (copies / fills / adds / cats) of one eager GD step.
   python tools/copy_sources.py            (run on the GPU box; prints per (op, call site) counts and device time)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from efficientvlm_amd.workload import GEOMS, make_batch  # noqa: E402

dev = torch.device("cuda", 0)
geom = GEOMS["full"]
student, teacher = bench.build(geom, dev, seed=1)
tr = bench.make_trainer(student, teacher, torch.bfloat16, False, False)
batches = [{k: v.to(dev) for k, v in make_batch(geom, 64, seed=40 + i).items()} for i in range(2)]
for i in range(3):
    tr.step(batches[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr.step(batches[1])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    dt = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    if dt <= 0:
        continue
    site = "(autograd engine)" if not ev.stack else "?"
    mine = [fr.split("efficientvlm_amd/")[-1] for fr in ev.stack or [] if "efficientvlm_amd/" in fr]
    if mine:
        site = " < ".join(mine[:3])
    a = agg[(ev.name, site)]
    a[0] += 1
    a[1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"ATen device time in one eager step: {tot / 1e3:.3f} ms over {sum(v[0] for _, v in rows)} top-level ops")
for (name, site), (n, t) in rows[:90]:
    print(f"{t:9.1f} us  {n:4d}x  {name:28s} {site}")
