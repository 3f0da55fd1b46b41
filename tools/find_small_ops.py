"""Where do the small ATen fills / copies / adds of one eager GD step come from?
TorchDispatchMode: every ATen call is attributed to the autograd node that is executing (backward) or to the python
call site inside efficientvlm_amd (forward)."""
import sys, os, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from oracle import synth
from efficientvlm_amd.trainer import GDTrainer
import bench
geom = synth.GEOMS["full"]; dev = torch.device("cuda")
KIND = next((a for a in sys.argv[1:] if a in ("gd", "itr", "vqa")), "gd")       # which step: GD (default), ITR-384, VQA-480
if KIND == "gd":
    # --dropout P: the student BERT's stock training-mode dropout (the ATen list must not grow: no index_select of K / V, no
    # separate mask passes - the masks are regenerated inside the attention kernels and the GEMM epilogues)
    DROP = float(sys.argv[sys.argv.index("--dropout") + 1]) if "--dropout" in sys.argv else 0.0
    s, t = bench.build(geom, dev, 1234, dropout=DROP)
    # (teacher pipelined one batch ahead, as bench.py runs it: the fused distillation paths are armed; launched eagerly)
    tr = GDTrainer(s, t, dtype=torch.bfloat16, use_graph=False, pipeline_teacher=True)
    batch = {k: v.to(dev) for k, v in synth.make_batch(geom, 64, seed=42).items()}
    step = lambda: tr.step(batch)
else:
    from helpers import model_config
    from efficientvlm_amd.trainer import ITRTrainer, VQATrainer
    res = 384 if KIND == "itr" else 480
    geom = dict(geom); geom["image_res"] = res
    torch.manual_seed(0)
    if KIND == "itr":
        from efficientvlm_amd.efficient_models.model_retrieval import EffXVLMforRetrieval
        from efficientvlm_amd.models.model_retrieval import XVLM as TeacherITR
        s = EffXVLMforRetrieval(model_config(geom, "s", image_res=res)).to(dev)
        t = TeacherITR(model_config(geom, "t", image_res=res)).to(dev)
        tr = ITRTrainer(s, t, lr=3e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                        pipeline_teacher=True, capture_step=False)
        batch = {k: v.to(dev) for k, v in synth.make_batch(geom, 16, seed=5).items()}
        idx = torch.arange(16, device=dev)
        step = lambda: tr.step(batch, idx=idx)
    else:
        from efficientvlm_amd.efficient_models.model_generation import EffXVLMForVQA
        from efficientvlm_amd.models.model_generation import XVLMForVQA
        cfg = lambda role, nd: dict(model_config(geom, role, image_res=res), pad_token_id=0, num_dec_layers=nd)
        s = EffXVLMForVQA(cfg("s", 3)).to(dev); t = XVLMForVQA(cfg("t", 6)).to(dev)
        tr = VQATrainer(s, t, lr=5e-5, weight_decay=0.01, lr_mult=2, reg_learning_rate=0.1, dtype=torch.bfloat16,
                        pipeline_teacher=True, capture_step=False)
        batch = {k: v.to(dev) for k, v in synth.make_vqa_batch(geom, 8, seed=5, La=8).items()}
        step = lambda: tr.step(batch)
    s.l0_module.set_lagrangian_warmup_steps(100)
for _ in range(3): step()
torch.cuda.synchronize()
agg = collections.Counter()
WANT = None
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if True:
            node = torch._C._current_autograd_node()
            where = None
            if node is not None:
                where = "bwd:" + node.name()
            for f in reversed(traceback.extract_stack()[:-1]):
                if "efficientvlm_amd" in f.filename:
                    where = (where + " @ " if where else "fwd @ ") + f"{os.path.basename(f.filename)}:{f.lineno}"
                    break
            shp = None
            for a in args:
                if isinstance(a, torch.Tensor):
                    shp = (tuple(a.shape), str(a.dtype).replace("torch.", "")); break
            agg[(name, where or "?", shp)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    step()
torch.cuda.synchronize()
# metadata-only ops launch nothing: listed only with --all
NOLAUNCH = ("view", "detach", "slice", "select", "empty", "as_strided", "split", "record_stream", "t.", "transpose", "unsqueeze",
            "squeeze", "expand", "permute", "alias", "_unsafe_view", "reshape", "unbind", "unflatten", "is_", "size", "stride",
            "_local_scalar", "chunk", "narrow", "new_empty", "lift_fresh", "resize_", "set_", "_reshape_alias", "unfold")
rows = [(k, n) for k, n in agg.items() if "--all" in sys.argv or not k[0].startswith(NOLAUNCH)]
print(f"# {sum(n for _, n in rows)} launching ATen calls in one eager step")
for (name, where, shp), n in sorted(rows, key=lambda kv: -kv[1])[:200]:
    print(f"{n:5d}  {name:22s} {where:60s} {shp}")
