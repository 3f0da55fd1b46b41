"""Optimiser for the distillation step — the reference's optim.py:create_optimizer parameter grouping
(decay / no-decay x lr / lr*lr_mult for `model.init_params`) with HF-AdamW semantics (betas (0.9, 0.98), eps 1e-8, Adam
update then decoupled decay) and the accelerator's global-norm clipping (apex_ddp_accelerator.py:99-102), executed by
two HIP kernels per group over FLAT fp32 buffers (evlm_sumsq, evlm_adamw_step).

MI355X-first layout: every trainable parameter and its gradient are views into one contiguous fp32 slab per group, so
 - the optimiser is 2 launches per group instead of ~400 tensor-wise launches,
 - data-parallel gradient reduction is a handful of large RCCL all-reduces over contiguous memory (no bucket copies),
 - zeroing gradients is one memset.
"""
import math

import torch

from . import _lib as L
from . import ops

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight")   # optim.py:35-43 (substring match)


_QKV_RANK = {"q_proj": 0, "k_proj": 1, "v_proj": 2, "query": 0, "key": 1, "value": 2}


def _pack_order(named):
    """forward (registration) order, except that (1) the q / k / v projections of one attention module are placed in the
    order the fused QKV / KV GEMMs pack them (CLIP registers k, v, q) and (2) the cross-attention key / value projections of
    ALL fusion layers of one encoder follow each other (k_i, v_i, k_i+1, v_i+1, ...: BertEncoder projects the image tokens
    for every fusion layer in one product), so that each packed operand is ONE contiguous slab range and needs no gather
    copy."""
    first = {}
    def slot(n):
        parts = n.split(".")
        if len(parts) >= 2 and parts[-2] in _QKV_RANK:
            if (len(parts) >= 6 and parts[-4:-2] == ["crossattention", "self"] and parts[-6] == "layer"
                    and parts[-5].isdigit() and parts[-2] in ("key", "value")):
                return (".".join(parts[:-5]) + ".crossattention.kv", parts[-1]), (int(parts[-5]), _QKV_RANK[parts[-2]])
            return (".".join(parts[:-2]), parts[-1]), (0, _QKV_RANK[parts[-2]])
        return None, None
    for idx, (n, _) in enumerate(named):
        k, _r = slot(n)
        if k is not None:
            first.setdefault(k, idx)
    def key(item):
        idx, (n, _) = item
        k, r = slot(n)
        if k is not None:
            return (first[k],) + r
        return (idx, 0, 0)
    return [it for _, it in sorted(enumerate(named), key=key)]


def _seg(p):
    """slab words a parameter occupies: its elements rounded up to 8 (32-byte fp32 / 16-byte bf16 alignment).  A tall matrix
    whose row count is not a multiple of 64 (the 30 522-row vocabulary matrix) is followed by ZERO rows up to the next
    multiple: the vocabulary head's dX = dY W reads W out of the bf16 mirror in place as a [ceil64(rows), K] operand
    (ops._Linear.backward) - the rows behind the matrix meet dY's zero padding columns, and 0 x (a neighbouring parameter
    gone non-finite) would be NaN in every row of dX.  Zero gradient, zero moments: AdamW leaves the pad at zero."""
    k = p.numel()
    if p.dim() == 2 and p.shape[0] >= 4096 and p.shape[0] % 64:
        k = (p.shape[0] + 63) // 64 * 64 * p.shape[1]
    return (k + 7) // 8 * 8


class FlatAdamW:
    def __init__(self, model, lr=1e-4, weight_decay=0.01, lr_mult=1.0, betas=(0.9, 0.98), eps=1e-8, max_grad_norm=1.0,
                 late_prefix="vision_encoder."):
        self.betas, self.eps, self.max_grad_norm = betas, eps, max_grad_norm
        large = set(getattr(model, "init_params", []) or [])
        groups = [dict(weight_decay=weight_decay, lr=lr, params=[], names=[]),
                  dict(weight_decay=0.0, lr=lr, params=[], names=[]),
                  dict(weight_decay=weight_decay, lr=lr * lr_mult, params=[], names=[]),
                  dict(weight_decay=0.0, lr=lr * lr_mult, params=[], names=[])]
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        for n, p in _pack_order(named):       # forward order; q/k/v of one attention block adjacent in packing order
            nd = any(s in n for s in NO_DECAY)
            gi = (3 if n in large else 1) if nd else (2 if n in large else 0)
            groups[gi]["params"].append(p)
            groups[gi]["names"].append(n)
        self.groups = [g for g in groups if g["params"]]
        self.step_count = 0
        self._assign = None
        self._scheduled, self._last_mult = False, 1.0      # has set_schedule() run since the last step()?
        dev = named[0][1].device
        self._gn = torch.zeros(4, dtype=torch.float32, device=dev)       # (16 bytes: a unit of the grouped zero fill)
        self.gnorm_sq = self._gn[:1]
        self._gnorm_zeroed = False
        # fixed-order sum of squares (include/evlm_hip.h: evlm_sumsq): data-parallel replicas clip by bit-identical factors
        self.sumsq_ws = torch.zeros(2050, dtype=torch.float32, device=dev) if torch.device(dev).type == "cuda" else None
        self.hyper = torch.ones(3, dtype=torch.float32, device=dev)       # {lr multiplier, bias_c1, bias_c2}
        lowp = dev.type == "cuda"
        for g in self.groups:
            n = sum(_seg(p) for p in g["params"])          # 32-byte (fp32) / 16-byte (bf16) aligned segments
            g["p"] = torch.zeros(n, dtype=torch.float32, device=dev)
            g["g"] = torch.zeros(n, dtype=torch.float32, device=dev)
            g["m"] = torch.zeros(n, dtype=torch.float32, device=dev)
            g["v"] = torch.zeros(n, dtype=torch.float32, device=dev)
            # bf16 mirror of the parameter slab, refreshed by the AdamW kernel itself: the compute-dtype weight copies the
            # GEMMs read are views into it (ops.CACHE), so no per-tensor cast launches remain in the step
            g["pb"] = torch.zeros(n, dtype=torch.bfloat16, device=dev) if lowp else None
            off = 0
            late = []                  # [lo, hi) slab ranges of the parameters whose gradients arrive LAST in backward
            for p, nme in zip(g["params"], g["names"]):
                k = p.numel()
                seg = _seg(p)
                if nme.startswith(late_prefix):
                    if late and late[-1][1] == off:
                        late[-1][1] = off + seg
                    else:
                        late.append([off, off + seg])
                g["p"][off:off + k].copy_(p.data.reshape(-1))
                p.data = g["p"][off:off + k].view(p.shape)
                p.grad = g["g"][off:off + k].view(p.shape)
                if lowp:
                    ops.CACHE.register_slab(p, g["p"], g["pb"], off, seg)
                off += seg
            g["late"] = late
            if lowp:
                ops.CACHE.refresh_slab(g["p"], g["pb"])
        ops.CACHE.invalidate()

    @property
    def flat_grads(self):
        return [g["g"] for g in self.groups]

    def grad_segments(self):
        """(early, late) lists of slab views: `late` = the image encoder (its gradients are produced last in backward),
        `early` = everything else (text / fusion encoder, heads): complete once backward reaches the image encoder."""
        early, late = [], []
        for g in self.groups:
            pos, n = 0, g["g"].numel()
            for lo, hi in g["late"]:
                if lo > pos:
                    early.append(g["g"][pos:lo])
                late.append(g["g"][lo:hi])
                pos = hi
            if pos < n:
                early.append(g["g"][pos:n])
        return early, late

    def grad_ranges(self, pred):
        """gradient-slab views covering exactly the parameters whose name satisfies `pred` (adjacent members merged): the
        data-parallel reducer sends a layer group's gradients the moment backward has finished with it"""
        out = []
        for g in self.groups:
            off, cur = 0, None
            for p, nme in zip(g["params"], g["names"]):
                seg = _seg(p)
                if pred(nme):
                    if cur is not None and cur[1] == off:
                        cur[1] = off + seg
                    else:
                        cur = [off, off + seg]
                        out.append((g, cur))
                off += seg
        return [g["g"][lo:hi] for g, (lo, hi) in out]

    def zero_grad(self, skip_assigned=False):
        """skip_assigned: leave out the ranges of assign_state() - the caller promises that this step's backward runs with
        ops.WGRAD_ASSIGN = that state (every such range is then written by its first contribution or zero-filled by
        ops.finish_assign)"""
        # ONE fill launch for all ranges (+ the norm accumulator step() sums into): they were 7 launches of a GD step
        ranges = ([g["g"] for g in self.groups] if (not skip_assigned or self._assign is None) else list(self._assign["fill"]))
        if self.gnorm_sq.is_cuda:
            ops._keep_table(ops.zero_grouped(ranges + [self._gn]))
            self._gnorm_zeroed = True
        else:
            for r in ranges:
                r.zero_()

    def assign_state(self, model):
        """state for ops.WGRAD_ASSIGN: the weights of the model's nn.Linear modules (not tied to an embedding) - the
        parameters whose gradient is one dY^T X product per step - and the slab ranges a step still has to zero-fill
        (everything else: biases, LayerNorms, embeddings, scalars, the L0 parameters, padding words)"""
        if self._assign is not None:
            return self._assign
        tied = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Embedding)}
        lin = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Linear) and id(m.weight) not in tied}
        skip, fill = {}, []
        for g in self.groups:
            off, pos = 0, 0
            for p in g["params"]:
                k = p.numel()
                seg = _seg(p)
                if id(p) in lin and p.dim() == 2 and k >= 4096 and k == seg:
                    if off > pos:
                        fill.append(g["g"][pos:off])
                    skip[g["g"].data_ptr() + off * 4] = g["g"][off:off + k]
                    pos = off + seg
                off += seg
            if g["g"].numel() > pos:
                fill.append(g["g"][pos:])
        self._assign = {"skip": skip, "done": set(), "pending": {}, "fill": fill}
        return self._assign

    def set_schedule(self, lr_mult=1.0):
        """host-side per-step scalars -> device (call OUTSIDE a captured graph, before replay)"""
        self._scheduled, self._last_mult = True, lr_mult
        self.step_count += 1
        b1, b2 = self.betas
        # a FRESH pinned block per step: the host may run several (graph-replayed) steps ahead of the device, so a
        # single staging buffer would be overwritten before its asynchronous upload has executed.  torch's pinned-memory
        # cache recycles a block only after the copy that read it has completed.
        host = torch.tensor([lr_mult, 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count], dtype=torch.float32)
        if self.hyper.is_cuda:
            host = host.pin_memory()
        self.hyper.copy_(host, non_blocking=True)

    def step(self):
        """clip by global norm + AdamW; pure device work (capturable).  Gradients must live in the flat slabs.
        A reference-style loop (optimizer.step() with no scheduler call, optim.py:67 + GeneralDistill.py:386) never calls
        set_schedule(): the step then advances the count itself with the last lr multiplier, so Adam's bias corrections
        are never silently left at 1.  Under hipGraph capture the scalars cannot be staged from here: that is an error."""
        lib = L.load()
        if not self._scheduled:
            if self.hyper.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdamW.step() captured into a hipGraph without set_schedule(): call "
                                   "set_schedule(lr_mult) before every replay (it stages the step's lr multiplier and "
                                   "bias corrections on the device)")
            self.set_schedule(self._last_mult)
        self._scheduled = False
        for g in self.groups:               # autograd may have re-pointed .grad if a view was replaced: re-anchor
            off = 0
            for p in g["params"]:
                k = p.numel()
                if p.grad is not None and p.grad.data_ptr() != g["g"].data_ptr() + off * 4:
                    g["g"][off:off + k].copy_(p.grad.reshape(-1))
                    p.grad = g["g"][off:off + k].view(p.shape)
                off += _seg(p)
        if not self._gnorm_zeroed:          # (zero_grad() of this step cleared it together with the gradient ranges)
            self.gnorm_sq.zero_()
        self._gnorm_zeroed = False
        for g in self.groups:
            L.check(lib.evlm_sumsq(L.ptr(g["g"]), g["g"].numel(), L.ptr(self.gnorm_sq), L.ptr(self.sumsq_ws), L.stream()), "sumsq")
        b1, b2 = self.betas
        for g in self.groups:
            L.check(lib.evlm_adamw_step(L.ptr(g["p"]), L.ptr(g["g"]), L.ptr(g["m"]), L.ptr(g["v"]), g["p"].numel(),
                                        g["lr"], b1, b2, self.eps, g["weight_decay"], 1.0, 1.0, L.ptr(self.gnorm_sq),
                                        float(self.max_grad_norm or 0.0), L.ptr(g["pb"]), L.ptr(self.hyper), L.stream()), "adamw")
        ops.CACHE.invalidate()
        # W^T copies of THIS optimiser's bf16 mirror (one grouped launch; no-op in fp32 runs)
        ops.CACHE.refresh_transposed([g["pb"] for g in self.groups if g.get("pb") is not None])

    def grad_norm(self):
        return self.gnorm_sq.sqrt()


class TensorAdamW:
    """HF-AdamW over a handful of small tensors that stay where they are (one evlm_adamw_step launch per tensor, own
    moment buffers) - the reference's l0_optimizer / lagrangian_optimizer (optim.py:4-21): the gate parameters and the
    Lagrange multipliers are ALSO members of the main optimiser's groups (they are parameters of the model), so these
    optimisers must not re-home them.  lr may be negative (gradient ascent on lambda_1 / lambda_2)."""

    def __init__(self, named_params, lr, weight_decay=0.0, betas=(0.9, 0.98), eps=1e-8):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.betas, self.eps = betas, eps
        self.param_groups = [dict(params=self.params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)]
        self.state = [dict(m=torch.zeros_like(p, dtype=torch.float32), v=torch.zeros_like(p, dtype=torch.float32))
                      for p in self.params]
        self.step_count = 0
        self.hyper, self._scheduled = None, False

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def set_schedule(self):
        """stage this step's bias corrections on the DEVICE (call outside a captured graph, before its replay): a captured
        step() then reads them from there instead of baking the host values of its capture pass into the graph"""
        self.step_count += 1
        b1, b2 = self.betas
        host = torch.tensor([1.0, 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count], dtype=torch.float32)
        if self.hyper is None:
            self.hyper = torch.ones(3, dtype=torch.float32, device=self.params[0].device)
        if self.hyper.is_cuda:
            host = host.pin_memory()               # (a fresh pinned block per step: FlatAdamW.set_schedule)
        self.hyper.copy_(host, non_blocking=True)
        self._scheduled = True

    def step(self):
        lib = L.load()
        b1, b2 = self.betas
        g = self.param_groups[0]
        staged = self._scheduled
        if not staged:
            if self.params and self.params[0].is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("TensorAdamW.step() captured into a hipGraph without set_schedule()")
            self.step_count += 1
        self._scheduled = False
        c1, c2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
        for p, st in zip(self.params, self.state):
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("TensorAdamW: parameters must be contiguous fp32 CUDA tensors (no CPU fallback)")
            L.check(lib.evlm_adamw_step(L.ptr(p.data), L.ptr(p.grad), L.ptr(st["m"]), L.ptr(st["v"]), p.numel(),
                                        float(g["lr"]), b1, b2, self.eps, float(g["weight_decay"]), c1, c2, None, 0.0,
                                        None, L.ptr(self.hyper) if staged else None, L.stream()), "adamw")


def create_L0_optimizer(args, l0_module):
    """optim.py:4-21 signature: (l0_optimizer over the gate log-alphas, lr = +reg_learning_rate;
    lagrangian_optimizer over lambda_1 / lambda_2, lr = -reg_learning_rate)"""
    get = (lambda k, d=None: args.get(k, d)) if isinstance(args, dict) else (lambda k, d=None: getattr(args, k, d))
    reg = get("reg_learning_rate")
    named = list(l0_module.named_parameters())
    return (TensorAdamW([(n, p) for n, p in named if "lambda" not in n], lr=reg),
            TensorAdamW([(n, p) for n, p in named if "lambda" in n], lr=-reg))


def create_optimizer(args, model, max_grad_norm=1.0):
    """optim.py:23-69 signature (args.lr, args.weight_decay, args.lr_mult)"""
    get = (lambda k, d=None: args.get(k, d)) if isinstance(args, dict) else (lambda k, d=None: getattr(args, k, d))
    return FlatAdamW(model, lr=get("lr"), weight_decay=get("weight_decay"), lr_mult=get("lr_mult", 1),
                     max_grad_norm=max_grad_norm)


def linear_schedule(step, num_warmup_steps, num_training_steps):
    """scheduler.py:14-22 (LambdaLR factor)"""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
