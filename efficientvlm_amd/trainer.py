"""GD training step runtime: student + frozen teacher, KD loss mix, backward, data-parallel gradient reduction,
global-norm clip + AdamW — the body of GeneralDistill.py:train (reference :286-387) for a general or a region batch; and
the pruning fine-tune steps of Eff_Retrieval.py / Eff_VQA.py (ITRTrainer, VQATrainer).

MI355X-first execution
  * one process per GPU; the teacher is a plain replica on every rank and never communicates;
  * the whole step is device-resident (no .item() syncs: the reference's 2B multinomial().item() calls and 11 meter
    reads per step are gone), so it is captured into hipGraphs and replayed (~900 kernel launches);
  * the frozen teacher runs one batch ahead of the student on a second stream inside the same graph (pipeline_teacher);
  * gradients live in the optimiser's flat fp32 slabs, so data parallelism is a few large RCCL all-reduces over
    contiguous memory on a side stream (xGMI is point-to-point: few, large messages), issued in STAGES as backward
    completes them (install_grad_stages): text / fusion / heads when backward enters the image encoder, the ViT layer
    groups from hooks inside its backward, the rest after it;
  * with N > 1 the step replays as hipGraph SEGMENTS cut at its collectives (RCCL is not capturable on this stack): the
    ITC gather in the forward, the stage boundaries in the backward - the all-reduce of a stage overlaps the next
    segment; a capture that fails on any rank switches every rank to the eager step (same collective sequence).
"""
import os
import contextlib
import gc
import time

import torch
import torch.distributed as dist

from . import distill, ops
from .optim import FlatAdamW
from .runtime import compute, log_collective


def dist_ready():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


_QUIESCE_S = float(os.environ.get("EVLM_CAPTURE_QUIESCE_MS", "300")) / 1e3


def _quiesce_collectives():
    """before a hipGraph capture in a process with an RCCL group: no collective of an EARLIER step may still be on the
    process group's watchdog list.  The watchdog thread polls the end event of every unfinished Work (every 100 ms); on
    this stack such a hipEventQuery, when it lands while a stream of this process captures, can fail with
    hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing stream") and the watchdog
    then terminates the process (seen in 2 of 3 runs of the ITR segment capture after round 5's teacher-recipe change
    moved the captures' timing; profiles/r05_capture_watchdog.md).  Captures happen a bounded number of times per
    trainer, so draining the device and giving the watchdog three polls to retire the finished Works costs ~0.3 s each."""
    if _QUIESCE_S <= 0 or not torch.cuda.is_available() or not dist.is_available() or not dist.is_initialized():
        return
    # (any RCCL group of the process has a watchdog: the default group or a sub-group a reducer was given)
    groups = [None] + [g for g in _RCCL_GROUPS if g is not None]
    try:
        if not any(dist.get_backend(g) == "nccl" for g in groups):
            return
    except (RuntimeError, ValueError):
        pass
    torch.cuda.synchronize()
    time.sleep(_QUIESCE_S)


_RCCL_GROUPS = []      # process groups handed to a GradReducer (registered there): _quiesce_collectives looks at them too


@contextlib.contextmanager
def no_gc_during_capture():
    """around a hipGraph capture: the cyclic garbage collector must not run inside it.  A collection that fires during a
    capture (any allocation can trigger one) may finalise objects of EARLIER work that were only reachable through cycles -
    a trainer and its hooks, its captured graphs, pinned staging blocks - and their destructors (hipGraphExecDestroy,
    hipHostFree, frees into another graph's pool) are not permitted while a stream captures: the process aborts
    (seen once in ~5 runs of the GPU suite, inside TeacherPrefetch's capture after many trainers had come and gone).
    Everything collectable is collected before the capture starts."""
    was = gc.isenabled()
    gc.collect()
    _quiesce_collectives()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class GradReducer:
    """mean all-reduce of flat gradient slabs in <= bucket_bytes pieces on a dedicated stream.

    ONE arithmetic on every backend (RCCL on GPU, gloo in the CPU / two-ranks-on-one-GPU tests): the collective is always
    a SUM.  The mean's 1 / world is applied
      * by the CALLER at the source when `prescaled` (the trainers: `scale_loss(total).backward()` - backward then produces
        gradient / world, no pass over the slabs at all; world sizes are powers of two, so the scaling is exact), or
      * by one in-place multiply of the ranges before they travel otherwise (the generic `reduce()` of the tests).
    No ReduceOp.AVG: RCCL's averaging collective was a branch only a real multi-GPU run could reach.
    The wire carries fp32 by default - what the reference's DDP wrappers reduce (apex_ddp_accelerator.py:87,
    Eff_Retrieval.py:449).  `compress` = torch.bfloat16 is an OPT-IN that halves the wire bytes (the slabs stay fp32:
    cast -> sum all-reduce -> cast back); its error against the fp32 wire is bounded in tests/test_dp_cpu.py.

    Overlap with backward: the slab ranges of a layer group are handed to `reduce_async` the moment backward has
    finished with the group (GDTrainer: tensor hooks in the eager step, cuts between hipGraph segments in the captured
    one); the collectives run on the side stream under the rest of backward, `finish()` joins them before the optimiser.
    Few, large messages: xGMI is point-to-point, rings are per-link bound.  The ranges of one stage go out as ONE grouped
    launch where the backend can coalesce device tensors (`launch` = "coalesced"); that private torch API is PROBED once at
    construction and the ranks agree on the outcome (MIN), so a refusal anywhere puts every rank on plain per-range async
    all-reduces (`launch` = "per-range") - never a mixed collective sequence."""

    def __init__(self, flat_grads, bucket_bytes=64 << 20, compress=None, group=None, force=False, prescaled=False):
        self.flat = list(flat_grads)
        self.group = group
        if group is not None and group not in _RCCL_GROUPS:
            _RCCL_GROUPS.append(group)
        self.compress = compress
        self.bucket_bytes = bucket_bytes
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        self.active = self.world > 1 or (force and ready)
        self.prescaled = prescaled
        self.buckets = self._buckets(self.flat)
        # (a high-priority stream was measured - profiles/r05_exchange_overlap.md: it gets a hardware queue of its own, and a
        # long kernel on it then starves every normal-priority queue: 24.7 against 17.9 ms in the simulated-wire probe)
        self.stream = torch.cuda.Stream() if (self.flat and self.flat[0].is_cuda) else None
        self._pending = []
        self.coalesce = self.active and self._probe_coalescing()
        self.launch = "coalesced" if self.coalesce else "per-range"

    def scale_loss(self, total):
        """the mean's 1 / world at the source: backward of the returned scalar leaves gradient / world in the slabs"""
        if not (self.active and self.prescaled) or self.world == 1:
            return total
        return total * (1.0 / self.world)

    def _probe_coalescing(self):
        """can this process group coalesce all-reduces of device (or host) tensors?  Tried once on two one-element tensors;
        every rank then takes the MINIMUM of the outcomes (a plain all-reduce - the collective every backend has)."""
        ok = hasattr(dist, "_coalescing_manager") and not os.environ.get("EVLM_NO_COALESCE")
        dev = self.flat[0].device if self.flat else torch.device("cpu")
        if ok:
            try:
                # (gloo coalesces host tensors only: device slabs over gloo - the two-ranks-on-one-GPU tests - go one by one)
                if dev.type == "cuda" and dist.get_backend(self.group) != "nccl":
                    raise RuntimeError("backend cannot coalesce device tensors")
                a, b = torch.ones(1, device=dev), torch.ones(1, device=dev)
                with dist._coalescing_manager(self.group, async_ops=True) as cm:
                    dist.all_reduce(a, op=dist.ReduceOp.SUM, group=self.group)
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                cm.wait()
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                ok = bool(a.item() == self.world and b.item() == self.world)
            except Exception:                      # any refusal: AttributeError / NotImplementedError / RuntimeError ...
                ok = False
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() > 0.5)

    def _buckets(self, tensors):
        out = []
        for t in tensors:
            n, per = t.numel(), max(1, self.bucket_bytes // t.element_size())
            for o in range(0, n, per):
                out.append(t[o:min(n, o + per)])
        return out

    def reduce_async(self, tensors):
        """enqueue the mean all-reduce of `tensors` behind everything already enqueued on the current stream"""
        if not self.active:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self.stream)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            buckets = self._buckets(tensors)
            if not self.prescaled and self.world > 1:
                for b in buckets:
                    b.mul_(1.0 / self.world)                  # the mean's division, in fp32, before anything travels
            wire = buckets if self.compress is None else [b.to(self.compress) for b in buckets]
            op = dist.ReduceOp.SUM
            for w in wire:
                log_collective("all_reduce", w)
            if self.coalesce and len(wire) > 1:
                # ONE grouped launch for the ranges of a stage (the bulk of a stage is one range; the no-decay groups'
                # biases / LayerNorm ranges are latency-bound messages of their own otherwise)
                with dist._coalescing_manager(self.group, async_ops=True) as cm:
                    for w in wire:
                        dist.all_reduce(w, op=op, group=self.group)
                self._pending.append(cm)
            else:
                for w in wire:
                    self._pending.append(dist.all_reduce(w, op=op, group=self.group, async_op=True))
            if self.compress is not None:
                for h in self._pending:                       # (the cast-back reads the reduced wire buffers)
                    h.wait()
                self._pending = []
                for b, w in zip(buckets, wire):
                    b.copy_(w)

    def finish(self):
        if not self.active:
            return
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for h in self._pending:
                    h.wait()
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for h in self._pending:
                h.wait()
        self._pending = []

    def reduce(self):
        self.reduce_async(self.flat)
        self.finish()


def broadcast_parameters(opt, src=0):
    """rank `src`'s student parameters to every rank, slab by slab (apex_ddp_accelerator.py:75-77 broadcasts every
    state-dict tensor at set-up; torch DDP - Eff_Retrieval.py:449, Eff_VQA.py:327 - does it at construction).  The drivers
    seed with args.seed + rank and the L0 log-alphas are drawn with normal_(), so without this the replicas start - and,
    with only gradients averaged, stay - different.  The bf16 mirrors the GEMMs read are re-derived."""
    if not dist_ready():
        return
    for g in opt.groups:
        dist.broadcast(g["p"], src)
        if g.get("pb") is not None:
            ops.CACHE.refresh_slab(g["p"], g["pb"])


def install_grad_stages(owner, opt, student, which="all"):
    """Gradient exchange overlapped with backward.  Returns (stages, vision stage index | None): `stages` = slab ranges in
    the order backward completes them; stage i < last is sent from a hook the moment it is complete (owner._stage_done(i)),
    the last one after backward:
      "vision": text / fusion encoder + heads, complete when backward enters the image encoder (tensor hook on the ViT
                output, fired by the model through owner._on_vision_grad; the autograd engine runs every text-side node -
                created later - before any ViT node);
      ("vit", b): ViT layers >= b, hook on the input of layer b (6 layers: {4,5} + post-norm, then {2,3});
      last: layers {0,1} + embeddings.
    which = all | vit | vision | none drops hook points (their ranges ride with the next stage).  The eager step sends from
    the hooks; GDTrainer's captured step (hipGraph segments) CUTS its capture at the same points, so both issue the same
    collective sequence.  EVLM_DP_VIT_CUTS (e.g. "4,2,1") overrides the layer cuts."""
    # (the L0 gate parameters and multipliers belong to no encoder: every gated layer - the first ViT layer included -
    # contributes to their gradient, which is therefore complete only when backward is: they travel with the LAST stage)
    is_l0 = lambda nme: nme.startswith("l0_module.")
    early = opt.grad_ranges(lambda nme: not nme.startswith("vision_encoder.") and not is_l0(nme))
    late = opt.grad_ranges(lambda nme: nme.startswith("vision_encoder."))
    l0_ranges = opt.grad_ranges(is_l0)
    points = [("vision", early)]
    enc = getattr(getattr(student, "vision_encoder", None), "encoder", None)
    vit_hooks = enc is not None and hasattr(enc, "grad_hooks") and len(enc.layers) >= 4
    if vit_hooks:
        n = len(enc.layers)
        cuts = [n - n // 3, n - 2 * (n // 3)]              # 6 layers: hooks at 4 and 2
        if os.environ.get("EVLM_DP_VIT_CUTS"):             # e.g. "4,2,1": a smaller last (exposed) stage
            cuts = sorted({int(c) for c in os.environ["EVLM_DP_VIT_CUTS"].split(",") if 0 < int(c) < n}, reverse=True)
        layer_of = lambda nme: int(nme.split("encoder.layers.")[1].split(".")[0]) if "encoder.layers." in nme else None
        in_vit = lambda nme: nme.startswith("vision_encoder.")
        hi = n
        for b in cuts:
            pred = (lambda lo_, hi_: lambda nme: in_vit(nme) and (
                (layer_of(nme) is not None and lo_ <= layer_of(nme) < hi_) or
                (hi_ == n and layer_of(nme) is None and "post_layernorm" in nme)))(b, hi)
            points.append((("vit", b), opt.grad_ranges(pred)))
            hi = b
        rest = (lambda hi_: lambda nme: in_vit(nme) and not (
            (layer_of(nme) is not None and layer_of(nme) >= hi_) or (layer_of(nme) is None and "post_layernorm" in nme)))(hi)
        points.append((None, opt.grad_ranges(rest) + l0_ranges))
    else:
        points.append((None, late + l0_ranges))
    active = {"all": lambda k: True, "vit": lambda k: k != "vision", "vision": lambda k: k == "vision",
              "none": lambda k: False}[which]
    stages, hooks, carry = [], [], []
    for key, ranges in points:
        carry = carry + list(ranges)
        if key is None or active(key):
            stages.append(carry)
            hooks.append(key)
            carry = []
    vision_stage = None
    if vit_hooks:
        enc.grad_hooks = {}
    for i, key in enumerate(hooks[:-1]):
        if key == "vision":
            vision_stage = i
        else:
            enc.grad_hooks[key[1]] = (lambda i_: lambda: owner._stage_done(i_))(i)
    return stages, vision_stage


class _StagedExchange:
    """the sending side of install_grad_stages (shared by the three trainers); needs self.reducer, self._stages, self._sent and
    self._cut (None outside a segmented capture)"""
    _cut = None

    def _stage_done(self, i):
        """tensor hook: backward has finished with gradient stage i - send it under the rest of backward"""
        if self._sent == i:
            self._send(self._stages[i])
            self._sent = i + 1

    def _send(self, ranges):
        ops.flush_wgrad()                 # the queued weight gradients of the stage must be in the slabs first
        ops.assign_settle(ranges)         # (... and what the step never writes must be zero before it travels)
        if self._cut is not None:         # capture pass of the segmented step: the graph segment ends here
            self._cut(ranges)
        else:
            self.reducer.reduce_async(ranges)

    def _reduce_rest(self):
        """what backward has not sent from its hooks (the last stage; everything when no hook fired)"""
        for i in range(self._sent, len(self._stages)):
            self._send(self._stages[i])
        self._sent = len(self._stages)
        if self._cut is None:
            self.reducer.finish()

    def _wants_pool_stream(self, on_gpu):
        """live reducer over RCCL: the step must not run on the legacy default stream (GDTrainer.step).  Not for gloo - the
        two-ranks-on-one-GPU tests and dry runs: its device-tensor path blocks the host per collective and measured 4 x
        slower with the step on a pool stream (6.5 against 1.5 s per step of the bench.py dry run)"""
        if not (on_gpu and self.reducer.active) or os.environ.get("EVLM_NO_LOOP_STREAM"):
            return False
        try:
            return dist.get_backend(self.reducer.group) == "nccl"
        except Exception:
            return False

    def _off_default_stream(self, fn):
        """run fn() on a pool stream of this trainer when the caller is on the legacy default stream (joined both ways)"""
        cur = torch.cuda.current_stream()
        if cur != torch.cuda.default_stream():
            return fn()
        if getattr(self, "_loop", None) is None:
            self._loop = torch.cuda.Stream()
        self._loop.wait_stream(cur)
        with torch.cuda.stream(self._loop):
            out = fn()
        cur.wait_stream(self._loop)
        return out

    def _replay_segments(self, segs, last_reduce):
        """a step captured as hipGraph segments around its collectives: graphs replay on the current stream, the ITC gather
        is issued between two of them, every gradient stage's all-reduce goes to the reducer's stream beside the NEXT
        segment - except the last one, which the optimiser segment waits for"""
        for kind, item in segs:
            if kind == "graph":
                item.replay()
            elif kind == "gather":
                log_collective("all_gather", item[1])
                dist.all_gather(item[0], item[1])
            else:
                self.reducer.reduce_async(item)
                if item is last_reduce:
                    self.reducer.finish()

    # ---- lifetime --------------------------------------------------------------------------------------------------
    # A trainer owns hipGraphs (hipGraphExec objects, their private memory pools), pinned staging blocks and side streams,
    # and sits in reference cycles (model hooks point back at it), so without close() all of that dies whenever Python's
    # cyclic collector happens to run - possibly while ANOTHER trainer captures or replays, which this stack answers with
    # a fault inside hipGraphLaunch / an abort inside the capture (DESIGN.md "runtime landmines").  close() releases them at
    # a moment of the caller's choosing, with the device idle.
    _closed = False
    _GRAPH_ATTRS = ("_joint", "_seg", "_graphs", "_pipes", "_sgraphs", "_seen", "_eps")

    def close(self):
        """destroy this trainer's hipGraphs, static buffers and pinned blocks with the device idle, and cut the hooks that
        tie the models to it.  Idempotent; the trainer must not step afterwards."""
        if self._closed:
            return
        self._closed = True
        cuda = torch.cuda.is_available()
        if cuda:
            torch.cuda.synchronize()
        for model in (getattr(self, "student", None),):
            if model is None:
                continue
            if getattr(model, "on_vision_grad", None) is not None:
                model.on_vision_grad = None
            if hasattr(model, "phase_hook"):
                model.phase_hook = None
            enc = getattr(getattr(model, "vision_encoder", None), "encoder", None)
            if enc is not None and getattr(enc, "grad_hooks", None):
                enc.grad_hooks = {}
        # (the ITR / VQA trainers ask their frozen teacher for attention-map RECIPES instead of maps: a teacher that outlives
        # its trainer hands out maps again)
        tenc = getattr(getattr(getattr(self, "teacher", None), "vision_encoder", None), "encoder", None)
        if tenc is not None and getattr(tenc, "attn_recipe", False):
            tenc.attn_recipe = False
        pre = getattr(self, "prefetch", None)
        if pre is not None:
            pre.close()
        for name in self._GRAPH_ATTRS:
            v = getattr(self, name, None)
            if isinstance(v, (dict, set)):
                v.clear()
        for name in ("graph", "static", "out", "_pending", "_last_ST", "_tpool", "_spool", "_seg_pool", "_step_pool"):
            if hasattr(self, name):
                setattr(self, name, None)
        self.last_kd = {}
        gc.collect()                       # the graphs die HERE (device idle), not at a later collection
        if cuda:
            torch.cuda.synchronize()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):                     # last resort only (a trainer dropped without close()): never raise from here
        try:
            if not self._closed and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                self.close()
        except Exception:
            pass


def teacher_map_filter(student, teacher, with_cross):
    """The KD terms read every k-th attention map of the (deeper) teacher (get_cor_teacher: map i*k + k-1 for student map i;
    GD reads no cross-attention map, the ITR fine-tune does): tell the frozen teacher's encoders to materialise only those.
    The other layers then run without writing their [B, H, Lq, Lk] probabilities to HBM (60 MB per ViT layer at B = 64)."""
    sv, tv = getattr(student.vision_encoder, "encoder", None), getattr(teacher.vision_encoder, "encoder", None)
    if sv is not None and tv is not None and hasattr(tv, "attn_keep"):
        ns, nt = len(sv.layers), len(tv.layers)
        if nt % ns == 0:
            k = nt // ns
            tv.attn_keep = {i * k + k - 1 for i in range(ns)}
    sc = student.text_encoder.bert if hasattr(student.text_encoder, "bert") else student.text_encoder
    tc = teacher.text_encoder.bert if hasattr(teacher.text_encoder, "bert") else teacher.text_encoder
    se, te = sc.encoder, tc.encoder
    if hasattr(te, "attn_keep"):
        fs, ft = se.fusion_layer, te.fusion_layer
        ls, lt = len(se.layer), len(te.layer)
        if ft % fs == 0 and (lt - ft) % (ls - fs) == 0:
            k1, k2 = ft // fs, (lt - ft) // (ls - fs)
            keep = {i * k1 + k1 - 1 for i in range(fs)} | {ft + i * k2 + k2 - 1 for i in range(ls - fs)}
            te.attn_keep = keep
            te.cross_keep = ({l for l in keep if l >= ft} if with_cross else set())
    # the answer decoder of the VQA models (every layer cross-attends to the question states: Eff_VQA.py:150-160 reads the
    # decoder's self- and cross-attention maps through get_cor_teacher too)
    sd, td = getattr(student, "text_decoder", None), getattr(teacher, "text_decoder", None)
    if sd is not None and td is not None and hasattr(td.bert.encoder, "attn_keep"):
        ls, lt = len(sd.bert.encoder.layer), len(td.bert.encoder.layer)
        if lt % ls == 0:
            k = lt // ls
            td.bert.encoder.attn_keep = {i * k + k - 1 for i in range(ls)}
            td.bert.encoder.cross_keep = set(td.bert.encoder.attn_keep) if with_cross else set()


class GDTrainer(_StagedExchange):
    def __init__(self, student, teacher, lr=1e-4, weight_decay=0.01, lr_mult=2.0, max_grad_norm=1.0, temperature=1.0,
                 dtype=torch.bfloat16, use_graph=True, grad_compress=None, pipeline_teacher=False):
        """pipeline_teacher: the frozen teacher's forward for batch i+1 runs (second stream) WHILE the student trains on
        batch i - exact, since the teacher never changes; step(batch) then returns the losses of the batch passed to the
        PREVIOUS call (None on the first call, which only primes the pipeline).  Every step still executes one teacher
        forward, one student forward + backward and one optimiser step."""
        self.student, self.teacher = student, teacher
        self.pipeline_teacher = pipeline_teacher
        self._keep_ST, self._last_ST = False, None
        self.last_kd = {}
        self.dtype, self.temperature = dtype, temperature
        for p in teacher.parameters():
            p.requires_grad_(False)
        teacher.eval()
        student.train()
        self.opt = FlatAdamW(student, lr=lr, weight_decay=weight_decay, lr_mult=lr_mult, max_grad_norm=max_grad_norm)
        force = bool(os.environ.get("EVLM_FORCE_REDUCE"))
        if grad_compress is None and dtype == torch.bfloat16 and os.environ.get("EVLM_BF16_WIRE"):
            grad_compress = torch.bfloat16           # opt-in: bf16 on the wire (default: fp32, as the reference's DDP)
        self.reducer = GradReducer(self.opt.flat_grads, compress=grad_compress, force=force, prescaled=True)
        self.world = self.reducer.world
        # gradient exchange overlapped with backward: install_grad_stages (stages sent from hooks / capture cut there)
        self._stages, self._sent, self._cut = [list(self.opt.flat_grads)], 0, None
        self._vision_stage, self._join_at_vision = None, None
        self._seg, self._seg_pool, self._cap_stream = {}, None, None
        if self.reducer.active and hasattr(student, "on_vision_grad"):
            self._stages, self._vision_stage = install_grad_stages(self, self.opt, student,
                                                                   os.environ.get("EVLM_DP_CUTS", "all"))
        if hasattr(student, "on_vision_grad"):
            student.on_vision_grad = self._on_vision_grad
        # single GPU: the student's text pass beside its image encoder (multi-GPU keeps them in sequence: the "vision"
        # gradient stage relies on every text-side gradient being issued before backward enters the image encoder)
        if (hasattr(student, "text_stream") and not self.reducer.active and not os.environ.get("EVLM_NO_TEXT_STREAM")
                and next(student.parameters()).is_cuda):
            student.text_stream = torch.cuda.Stream()
        self.use_graph = use_graph
        self.wgrad_inplace = True
        if not os.environ.get("EVLM_TEACHER_ALL_MAPS"):
            teacher_map_filter(student, teacher, with_cross=False)
        if not os.environ.get("EVLM_STUDENT_ALL_MAPS"):
            # maps of the STUDENT that nothing in the GD recipe reads are not materialised (None in their output slots): the
            # cross-attention maps (GeneralDistill.py:300-366 distils self-attention maps only) and the ViT maps whose
            # distillation term was formed inside the attention kernel; their backward rebuilds the probabilities
            enc = getattr(getattr(student, "vision_encoder", None), "encoder", None)
            if enc is not None and hasattr(enc, "kd_drop_maps"):
                enc.kd_drop_maps = True
            tenc = getattr(getattr(student.text_encoder, "bert", student.text_encoder), "encoder", None)
            if tenc is not None and hasattr(tenc, "cross_keep"):
                tenc.cross_keep = set()
        if hasattr(teacher, "skip_task_losses"):
            teacher.skip_task_losses = True       # nobody reads a frozen teacher's ITC / ITM / MLM losses (nor gathers for them)
        self.defer_wgrad = self.dtype == torch.bfloat16 and not os.environ.get("EVLM_NO_DEFER_WGRAD")
        self.overlap_teacher = not os.environ.get("EVLM_NO_OVERLAP_TEACHER")     # teacher forward on a second stream
        self.graph = None
        self.static = None
        self.out = None
        self._graphs = {}
        self._pipes, self._pending = {}, None
        self._side = torch.cuda.Stream() if next(student.parameters()).is_cuda else None
        self._tpool = self._spool = None
        self._joint = {}
        if self.world > 1:
            broadcast_parameters(self.opt)

    # ---- the step body (pure device work) ----------------------------------------------------------
    def _forward_backward(self, batch, teacher_out=None):
        # (first-touch assignment of the Linear weights' gradients: their slab ranges are not zero-filled, ops.WGRAD_ASSIGN)
        self._assign = (self.opt.assign_state(self.student)
                        if (self.wgrad_inplace and self.defer_wgrad and not os.environ.get("EVLM_NO_WGRAD_ASSIGN")) else None)
        self.opt.zero_grad(skip_assigned=self._assign is not None)
        ops.begin_step(batch["image"].device)
        try:
            return self._forward_backward_body(batch, teacher_out)
        finally:
            ops.end_step()

    def _forward_backward_body(self, batch, teacher_out=None):
        with compute(self.dtype):
            total, S, T, kd, mix = distill.gd_forward(self.student, self.teacher, batch, self.temperature,
                                                      overlap_teacher=self.overlap_teacher, teacher_out=teacher_out)
            if self._keep_ST:
                self._last_ST = (S, T)
            # the individual KD terms of this step (device scalars; under capture: static tensors of that graph)
            self.last_kd = {k: v.detach() for k, v in kd.items() if torch.is_tensor(v)}
            ops.WGRAD_INPLACE = self.wgrad_inplace      # kernels sum parameter gradients straight into the flat slabs
            ops.WGRAD_DEFER = [] if (self.wgrad_inplace and self.defer_wgrad) else None   # ... dW products grouped per K
            ops.WGRAD_ASSIGN = self._assign
            try:
                # (multi-GPU: the mean's 1 / world enters here, the all-reduces are plain sums)
                hook = getattr(self.student, "phase_hook", None)
                if hook is not None:
                    hook("forward_done")
                self.reducer.scale_loss(total).backward()
                self._join_text_stream()
                ops.flush_wgrad()
                ops.finish_assign()
            finally:
                ops.WGRAD_INPLACE = False
                ops.WGRAD_DEFER = None
                ops.WGRAD_ASSIGN = None
                if self._assign is not None:     # (non-empty only when backward raised: the next step starts clean)
                    self._assign["done"].clear()
                    self._assign["pending"].clear()
                ops.LN_DEFER.clear()             # (empty after a flush; stale only when backward raised)
        if ops.DROPOUT_USED:              # p > 0 configurations: next step (next graph replay) draws new masks
            ops.dropout_tick(total.device)
        return torch.stack([total.detach().float(), S["loss"]["loss_itc"].detach().float(),
                            S["loss"]["loss_itm"].detach().float(), S["loss"]["loss_mlm"].detach().float(),
                            mix["loss_kd"].detach().float()])

    def _join_text_stream(self):
        """single GPU: the student's text pass ran on student.text_stream, so autograd ran its backward there too (the
        in-place word-embedding gradient, the queued dY / X of the deferred weight gradients, the MSE backward).  Join that
        stream explicitly before anything reads the slabs - not left to autograd's AccumulateGrad / end-of-backward stream
        syncs, which only order it as long as some gradient of that pass is accumulated out of place.  Valid inside a
        hipGraph capture (an event wait between two streams of the capture)."""
        ts = getattr(self.student, "text_stream", None)
        if ts is not None and not os.environ.get("EVLM_NO_TEXT_JOIN"):
            torch.cuda.current_stream().wait_stream(ts)

    def _on_vision_grad(self):
        """tensor hook on the ViT output (backward is about to enter the image encoder): a side-stream teacher forward that
        was asked to finish here joins, then the text / fusion / head gradient stage goes out"""
        if self._join_at_vision is not None:
            self._join_at_vision()
            self._join_at_vision = None
        if self._vision_stage is not None:
            self._stage_done(self._vision_stage)

    def _step_eager(self, batch, teacher_out=None):
        self._sent = 0
        out = self._forward_backward(batch, teacher_out)
        if self.reducer.active:
            self._reduce_rest()
        self.opt.step()
        return out

    # ---- teacher pipelining ------------------------------------------------------------------------
    # The frozen teacher's forward for the batch handed to step() runs on a SIDE stream while the main stream trains the
    # student on the batch of the previous call (whose teacher outputs are waiting in persistent buffers).  Per batch KIND
    # (general / region, i.e. per set of input shapes) there are two parities of static buffers - inputs and the teacher
    # tensors the KD terms read.  Single GPU with graphs: ONE hipGraph per (waiting batch, new batch) combination holds
    # both halves (teacher forked onto the side stream, joined at the end), captured lazily into one memory pool.
    # Multi-GPU: the student step stays eager (RCCL is not capturable here) and only the teacher replays a graph.  Kinds
    # may alternate freely; optimiser updates are applied in arrival order, one call late.
    MAX_BATCH_KINDS = 6          # batch shapes whose steps are captured into hipGraphs (first come); further kinds run eagerly
    MAX_EAGER_KINDS = 4          # ... with at most this many of them keeping their static buffers (least recently used evicted)
    _tick = 0

    def _evict_pipe(self):
        """drop the least recently used EAGER batch kind (never the one whose batch is still waiting for its student
        step).  Kinds with captured graphs are never dropped: destroying graphs that share a memory pool with live ones
        trips an internal assertion of the caching allocator on this stack."""
        waiting = self._pending[0] if self._pending is not None else None
        victims = sorted((p.get("tick", 0), sig) for sig, p in self._pipes.items() if p.get("eager") and p is not waiting)
        if victims:
            torch.cuda.synchronize()
            del self._pipes[victims[0][1]]

    def _pipe_create(self, batch, eager=False):
        """state of one batch kind: static buffers x2, two eager warm-up steps (lr 0, optimiser state restored), the
        persistent teacher-output buffers x2 (only the tensors the KD terms read; attention maps keep their padded rows),
        the graphs"""
        B = [{k: v.clone() for k, v in batch.items()} for _ in range(2)]
        state = [(g["m"].clone(), g["v"].clone()) for g in self.opt.groups], self.opt.step_count
        cur = torch.cuda.current_stream()
        self._keep_ST = True
        warm = torch.cuda.Stream()                         # warm-up on a side stream, as torch.cuda.graph asks for
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):
            for _ in range(2):
                self.opt.set_schedule(0.0)
                self._step_eager(B[0])
        cur.wait_stream(warm)
        self._keep_ST = False
        torch.cuda.synchronize()
        for g, (m, v) in zip(self.opt.groups, state[0]):       # the warm-up steps (lr 0) leave no trace
            g["m"].copy_(m); g["v"].copy_(v)
        self.opt.step_count = state[1]
        S, T = self._last_ST
        slots = distill.kd_teacher_slots(T, S)
        dummy = torch.zeros(0, device=batch["image"].device)

        def persist():
            out = {"loss": {}, "hidden_dict": {k: [dummy] * len(v) for k, v in T["hidden_dict"].items()},
                   "attention_dict": {k: [dummy] * len(v) for k, v in T["attention_dict"].items()},
                   "cross_attention_dict": {}, "logits_dict": {}}
            for d, key, i in slots:
                t = T[d][key] if i is None else T[d][key][i]
                base = ops._padded_base(t)
                buf = torch.empty_like(base)[..., :t.shape[-1]] if base is not None else torch.empty_like(t.contiguous())
                if i is None:
                    out[d][key] = buf
                else:
                    out[d][key][i] = buf
            return out
        pipe = dict(B=B, T=[persist(), persist()], slots=slots, par=0, tgraphs=None, eager=eager)
        self._last_ST = None
        del S, T
        side = self._side
        if self.use_graph and not eager and self.reducer.active and not os.environ.get("EVLM_NO_TEACHER_GRAPH"):
            # the frozen teacher's forward holds no collective (skip_task_losses): capturable on multi-GPU runs too
            side.wait_stream(cur)
            ops.reserve_tables()
            tg = []
            try:
                for k in (0, 1):
                    g = torch.cuda.CUDAGraph()
                    # (thread_local: on multi-GPU runs the RCCL watchdog thread may query events while this captures)
                    with no_gc_during_capture(), torch.cuda.graph(g, pool=self._tpool, stream=side,
                                                                  capture_error_mode="thread_local"):
                        self._teacher_eager(pipe, k)
                    ops.flush_table_uploads()
                    tg.append(g)
                    self._tpool = g.pool()
                pipe["tgraphs"] = tg
            except RuntimeError as e:        # same kernels, launched one by one: slower on the host, never wrong
                if not self.reducer.active:
                    raise
                import sys
                print(f"[efficientvlm_amd] teacher hipGraph capture failed on the multi-GPU path ({e}); "
                      "the teacher forward stays eager", file=sys.stderr)
                pipe["tgraphs"] = None
                torch.cuda.synchronize()
            cur.wait_stream(side)
        torch.cuda.synchronize()
        return pipe

    def _teacher_eager(self, pipe, k):
        """teacher forward on batch buffer k -> persistent outputs k (current stream).  The image encoder's attention maps
        (most of the bytes: 60 MB per kept layer) are WRITTEN into the persistent buffers by the attention kernels
        themselves; everything else is parked by ONE grouped copy launch."""
        self._teacher_finish(self._teacher_begin(pipe, k), pipe, k)

    def _teacher_begin(self, pipe, k):
        """first half of _teacher_eager: the teacher's IMAGE ENCODER on batch buffer k (current stream).  Returns the state
        _teacher_finish resumes - possibly on another stream state / in another hipGraph segment: the forward is suspended
        at its "vision_done" phase (models/model_pretrain.py:forward_phases)."""
        b = pipe["B"][k]
        enc = getattr(getattr(self.teacher, "vision_encoder", None), "encoder", None)
        region = "idx_to_group_img" in b
        if enc is not None and hasattr(enc, "attn_out") and not region:
            maps = pipe["T"][k]["attention_dict"].get("image_attentions", [])
            enc.attn_out = {i: ops._padded_base(m) if ops._padded_base(m) is not None else m
                            for i, m in enumerate(maps) if torch.is_tensor(m) and m.numel() > 0}
        st = {"enc": enc, "gen": None, "T": None}
        try:
            with torch.no_grad(), compute(self.dtype):
                if hasattr(self.teacher, "forward_phases") and getattr(self.teacher, "batched_passes", False):
                    st["gen"] = self.teacher.forward_phases(b["image"], b["text_ids"], b["text_atts"], **distill.model_kwargs(b))
                    while next(st["gen"]) != "vision_done":
                        pass
                else:                                  # a teacher without phases: everything here
                    st["T"] = self.teacher(b["image"], b["text_ids"], b["text_atts"], **distill.model_kwargs(b))
        finally:
            if enc is not None and hasattr(enc, "attn_out"):
                enc.attn_out = None
        return st

    def _teacher_advance(self, st, until):
        """resume the suspended teacher forward of _teacher_begin up to its phase `until` ("text_done", "fusion_done")"""
        if st["gen"] is None:
            return
        with torch.no_grad(), compute(self.dtype):
            try:
                while next(st["gen"]) != until:
                    pass
            except StopIteration as done:          # (a forward without that phase: it is complete)
                st["T"], st["gen"] = done.value, None

    def _teacher_finish(self, st, pipe, k):
        """second half: the teacher's text / fusion passes, then its outputs parked in the persistent buffers k"""
        T = st["T"]
        if T is None:
            with torch.no_grad(), compute(self.dtype):
                try:
                    while True:
                        next(st["gen"])
                except StopIteration as done:
                    T = done.value
        st["gen"] = None
        pairs = []
        for d, key, i in pipe["slots"]:
            src = T[d][key] if i is None else T[d][key][i]
            dst = pipe["T"][k][d][key] if i is None else pipe["T"][k][d][key][i]
            sb, db = ops._padded_base(src), ops._padded_base(dst)
            if sb is not None and db is not None:
                src, dst = sb, db
            if src.data_ptr() == dst.data_ptr():
                continue                                   # written in place
            if not src.is_contiguous():
                src = src.contiguous()
            nb = src.numel() * src.element_size()
            if nb % 16 or src.data_ptr() % 16 or dst.data_ptr() % 16 or not dst.is_contiguous():
                dst.copy_(src)
            else:
                pairs.append((src, dst))
        if pairs:
            pipe.setdefault("copy_tables", []).append(ops.copy_grouped(pairs))      # (kept alive: captured launches read it)
            if len(pipe["copy_tables"]) > 16 and not torch.cuda.is_current_stream_capturing():
                del pipe["copy_tables"][:-16]

    def _student_eager(self, pipe, k):
        """student forward + backward on (batch k, teacher outputs k), gradient reduction, optimiser step"""
        return self._step_eager(pipe["B"][k], pipe["T"][k])

    def _step_pipelined(self, batch, lr_mult):
        """teacher forward of `batch` on the side stream || student step on the batch of the previous call; returns the
        losses of THAT batch (None on the first call, which only starts the pipeline)"""
        sig = tuple(sorted((k, tuple(v.shape)) for k, v in batch.items()))
        pipe = self._pipes.get(sig)
        if pipe is None:
            # bounded state (a short last batch, odd region sizes ...): the first MAX_BATCH_KINDS shapes get graphs, later
            # ones run the same kernels eagerly out of static buffers that are recycled
            eager = sum(1 for q in self._pipes.values() if not q.get("eager")) >= self.MAX_BATCH_KINDS
            if eager and sum(1 for q in self._pipes.values() if q.get("eager")) >= self.MAX_EAGER_KINDS:
                self._evict_pipe()
            pipe = self._pipes[sig] = self._pipe_create(batch, eager)
        self._tick += 1
        pipe["tick"] = self._tick
        cur, side = torch.cuda.current_stream(), self._side
        cur.wait_stream(side)                     # the waiting batch's teacher outputs are complete
        p = pipe["par"] = 1 - pipe["par"]
        ops.copy_few([(v, pipe["B"][p][name]) for name, v in batch.items()])      # (one launch: they were six copies per step)
        out = None
        done = False                              # both halves of this call issued (joint graph / graph segments)?
        graphs_ok = self.use_graph and not pipe.get("eager") and (self._pending is None or not self._pending[0].get("eager"))
        if graphs_ok and not self.reducer.active and self._pending is not None:
            # single GPU: ONE hipGraph per (waiting batch, new batch) combination holds both halves - the teacher forward
            # of the new batch forked onto the side stream, the student step of the waiting batch on the capture stream,
            # joined at the end.  (Two separately launched graphs - one per stream - ran 1.5 % faster but aborted with a
            # hardware exception in ~15 % of runs on this stack; eager launches beside a graph do not.)
            pp, pk = self._pending
            self.opt.set_schedule(lr_mult)
            key = (id(pipe), p, id(pp), pk)
            jg = self._joint.get(key)
            if jg is None:
                ops.CACHE.invalidate()                             # capture the casts of the trainable weights too
                ops.reserve_tables()
                g = torch.cuda.CUDAGraph()
                with no_gc_during_capture(), torch.cuda.graph(g, pool=self._spool):
                    res = self._joint_body(pipe, p, pp, pk)
                ops.flush_table_uploads()
                self._spool = g.pool()
                jg = self._joint[key] = (g, res, self.last_kd)
            jg[0].replay()
            out, self.last_kd = jg[1], jg[2]
            done = True
        elif (graphs_ok and self.reducer.active and self._pending is not None
              and not os.environ.get("EVLM_NO_SEGMENT_GRAPHS") and not getattr(self, "_segments_broken", False)):
            # multi-GPU: hipGraph segments around the collectives; the new batch's teacher forward is forked onto the side
            # stream INSIDE a segment (as in the single-GPU joint graph), or - EVLM_SEG_TEACHER=graph - replays as its
            # own graph on the side stream beside the segments
            pp, pk = self._pending
            self.opt.set_schedule(lr_mult)
            own_graph = os.environ.get("EVLM_SEG_TEACHER", "fork") == "graph" and pipe["tgraphs"] is not None
            if own_graph:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    pipe["tgraphs"][p].replay()
            out = self._student_segmented(None if own_graph else pipe, p, pp, pk)
            done = out is not None
            if not done and own_graph:            # (the fallback below must not run the teacher a second time)
                out = self._student_eager(pp, pk)
                done = True
        if not done:
            side.wait_stream(cur)                 # inputs copied; every earlier reader of this parity's buffers is done
            with torch.cuda.stream(side):
                if pipe["tgraphs"] is not None:
                    pipe["tgraphs"][p].replay()
                else:
                    self._teacher_eager(pipe, p)
            if self._pending is not None:
                pp, pk = self._pending
                if not self.opt._scheduled:       # (a segmented attempt that fell back has staged this step already)
                    self.opt.set_schedule(lr_mult)
                out = self._student_eager(pp, pk)
        self._pending = (pipe, p)
        return out

    # ---- multi-GPU: the student step as hipGraph SEGMENTS around its collectives --------------------------------------
    # RCCL collectives cannot be captured on this stack, and an eager step costs the host ~1 100 launches (more time than
    # the GPU needs).  So the step is captured as a chain of graphs with the collectives issued eagerly between them:
    #   [forward up to the ITC feature gather] all_gather [rest of the forward, backward of heads / fusion / text layers]
    #   all-reduce(stage 0) [ViT layers 5,4 backward] all-reduce(stage 1) [layers 3,2] all-reduce(stage 2) [layers 1,0,
    #   embeddings, last grouped weight gradients] all-reduce(stage 3) [clip + AdamW]
    # Every all-reduce runs on the reducer's stream while the NEXT graph segment replays on the main stream: the exchange
    # is overlapped with backward exactly as in the eager step, and both forms issue the same collective sequence.  The
    # capture is cut inside the model's forward by efficient_models.xvlm.GATHER_HOOK (which the ITC all-gather calls
    # instead of dist.all_gather) and inside backward by the gradient-stage hooks (_stage_done -> _send -> _cut): the
    # capture pass runs autograd single-threaded, so a hook ends / begins captures on the thread that started them.
    def _capture_segments(self, tpipe, tp, pipe, k):
        """segments of: student step on (pipe, k); with `tpipe` the teacher forward of (tpipe, tp) is forked onto the side
        stream in the segment behind the ITC gather and joins where that segment ends"""
        from .efficient_models import xvlm as X
        if self._cap_stream is None:
            self._cap_stream, self._seg_pool = torch.cuda.Stream(), torch.cuda.graph_pool_handle()
        cur, cs, side = torch.cuda.current_stream(), self._cap_stream, self._side
        ops.reserve_tables()
        torch.cuda.synchronize()
        segs, state = [], {"g": None, "forked": tpipe is None, "joined": tpipe is None, "half": None,
                           "late_left": self._late_plan()}
        # the teacher in two halves (default): its image encoder beside the student's forward in the FIRST segment, its text /
        # fusion passes in the segment behind the ITC gather - each half about as long as the student work it shares the
        # chip with (one fork behind the gather leaves the first segment without a partner and the teacher outlasts the
        # second).  EVLM_SEG_TEACHER_SPLIT=0: the whole forward behind the gather.
        mode = os.environ.get("EVLM_SEG_TEACHER_SPLIT", "1")
        # "late" (round 5 experiment): the pairing the single-GPU joint graph gets for free - the teacher's image encoder
        # (chip-filling products) beside the student's PART-FILLED phase behind the gather, its text / fusion passes
        # (part-filled) beside the student's ViT backward (chip-filling) in the segment behind the first gradient stage
        late = tpipe is not None and mode == "late"
        split = tpipe is not None and not late and mode not in ("", "0")

        def fork_teacher():
            side.wait_stream(cs)
            with torch.cuda.stream(side):
                if state["half"] is not None:
                    self._teacher_finish(state["half"], tpipe, tp)
                    state["half"] = None
                else:
                    self._teacher_eager(tpipe, tp)
            state["forked"] = True

        def fork_teacher_piece():
            """late mode: the next piece of the teacher's forward behind its image encoder - text pass, fusion pass, heads +
            parking of the outputs - beside ONE segment of the student's ViT backward (joins where that segment ends)"""
            side.wait_stream(cs)
            with torch.cuda.stream(side):
                if state["half"] is None:            # (no gather cut the forward: the whole teacher here)
                    self._teacher_eager(tpipe, tp)
                    state["late_left"] = []
                else:
                    ph = state["late_left"].pop(0)
                    if ph is None:
                        self._teacher_finish(state["half"], tpipe, tp)
                        state["half"] = None
                    else:
                        self._teacher_advance(state["half"], ph)
            state["vision_open"] = True
            if not state["late_left"]:
                state["forked"] = state["joined"] = True

        def fork_teacher_vision():
            side.wait_stream(cs)
            with torch.cuda.stream(side):
                state["half"] = self._teacher_begin(tpipe, tp)
            state["vision_open"] = True              # (joins where this segment ends: end())

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            # (thread_local: the RCCL watchdog thread queries events while this thread captures)
            state["g"].capture_begin(pool=self._seg_pool, capture_error_mode="thread_local")

        def end():
            if state.get("vision_open"):
                cs.wait_stream(side)                 # the teacher's first half joins before its segment ends
                state["vision_open"] = False
            if state["forked"] and not state["joined"]:
                cs.wait_stream(side)                 # the teacher branch joins before its segment ends
                state["joined"] = True
            state["g"].capture_end()
            segs.append(("graph", state["g"]))
            state["g"] = None

        def gather(out_list, src):
            # (no collective is issued during the capture pass - nothing executes in it anyway: the watchdog thread of the
            # process group must not find an event of in-flight RCCL work tied to a capturing stream)
            end()
            segs.append(("gather", (out_list, src)))
            begin()
            if late:
                fork_teacher_vision()
            elif not state["forked"]:
                fork_teacher()

        def cut(ranges):
            if not late and not state["forked"]:     # (no gather cut the forward: single-rank group without the forced gather)
                fork_teacher()
            end()
            segs.append(("reduce", ranges))
            begin()
            if late and not state["forked"]:         # the teacher's next piece beside this segment of the ViT backward
                fork_teacher_piece()

        ops.CACHE.invalidate()
        cs.wait_stream(cur)
        self._cut = cut
        try:
            with no_gc_during_capture(), torch.cuda.stream(cs), torch.autograd.set_multithreading_enabled(False):
                X.GATHER_HOOK = gather
                try:
                    begin()
                    self._sent = 0
                    if split:
                        fork_teacher_vision()
                    out = self._forward_backward(pipe["B"][k], pipe["T"][k])
                    self._reduce_rest()                  # the remaining stages: one cut each
                    while late and not state["forked"]:  # (fewer cuts than teacher pieces: the rest beside the optimiser)
                        fork_teacher_piece()
                    self.opt.step()                      # clip + AdamW: the segment behind the last all-reduce
                    end()
                finally:
                    X.GATHER_HOOK = None
                    if state["g"] is not None:           # an exception inside a capture: close it before re-raising
                        try:
                            state["g"].capture_end()
                        except RuntimeError:
                            pass
                kd = self.last_kd
        finally:
            self._cut = None
        cur.wait_stream(cs)
        torch.cuda.synchronize()
        ops.flush_table_uploads()
        last_reduce = [item for kind, item in segs if kind == "reduce"][-1]
        if os.environ.get("EVLM_DEBUG_SEGMENTS"):
            import sys
            print("[efficientvlm_amd] segments:", [kind if kind != "reduce" else f"reduce({sum(r.numel() for r in item) * 4 / 1e6:.0f} MB)"
                                                   for kind, item in segs], file=sys.stderr)
        return dict(segs=segs, out=out, kd=kd, last_reduce=last_reduce)

    def _late_plan(self):
        """`late` placement: the phases up to which the teacher's forward (suspended behind its image encoder) is resumed beside
        the successive segments of the student's backward - one entry per gradient-stage cut, None = the rest.  Round 6: the
        fusion pass resumes LAYER BY LAYER (BertEncoder.forward_gen), so its 2.6 ms spread over the three ViT-backward segments
        (~1.1 ms each) and the optimiser segment instead of sitting whole beside one of them: [text pass + first fusion layer]
        [2 layers] [2 layers] [last layer + heads + parking].  EVLM_SEG_LATE_PLAN="text_done,fusion_done" restores round 5's."""
        env = os.environ.get("EVLM_SEG_LATE_PLAN")
        if env:
            return [p for p in env.split(",") if p] + [None]
        cfg = getattr(getattr(self.teacher, "text_encoder", None), "config", None)
        fl, n = getattr(cfg, "fusion_layer", None), getattr(cfg, "num_hidden_layers", None)
        if fl is None or n is None or n - fl < 4:
            return ["text_done", "fusion_done", None]
        per = (n - fl) / 3.0                                   # fusion layers per ViT-backward segment, the first shares with the text pass
        marks = [fl + max(0, int(round(per * 0.5)) - 1), fl + int(round(per * 1.5)) - 1, fl + int(round(per * 2.5)) - 1]
        return ["fusion_layer_%d" % m for m in marks] + [None]

    def _ranks_agree(self, ok):
        """did the capture succeed on EVERY rank?  (MIN all-reduce of the local flag: a rank that fell back to the eager
        step alone would issue a different collective sequence than its peers replaying segments - a hang)"""
        if not dist_ready():
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=self.opt.gnorm_sq.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def _student_segmented(self, tpipe, tp, pipe, k):
        """replay (capture on first use) the segmented student step; returns None when the capture failed on ANY rank -
        every rank then runs the eager step from here on"""
        key = (id(tpipe), tp, id(pipe), k)
        sg = self._seg.get(key)
        if sg is None:
            scheduled = self.opt._scheduled
            err = None
            try:
                sg = self._capture_segments(tpipe, tp, pipe, k)
            except Exception as e:                # a capture this stack refuses (RuntimeError) - or ANY other failure of
                err, sg = e, None                 # the capture pass: every rank must still reach the agreement round
                try:                              # below, or its peers wait in that all-reduce until the RCCL timeout
                    torch.cuda.synchronize()
                except RuntimeError:
                    pass
            self.opt._scheduled = scheduled              # the captured optimiser step consumed the flag, not the schedule
            agreed = self._ranks_agree(sg is not None)
            if err is not None and not isinstance(err, RuntimeError):
                raise err                         # a bug, not a refusal: surfaced - after the peers have been told
            if not agreed:
                import sys
                print(f"[efficientvlm_amd] hipGraph segments of the multi-GPU student step failed "
                      f"({err if err is not None else 'on another rank'}); the student step stays eager on every rank",
                      file=sys.stderr)
                self._segments_broken = True
                self._seg.clear()
                return None
            self._seg[key] = sg
        self._replay_segments(sg["segs"], sg["last_reduce"])
        self.opt._scheduled = False
        self.last_kd = sg["kd"]
        return sg["out"]

    def _joint_body(self, pipe, p, pp, pk):
        """single GPU: student step on (pp, pk) with the teacher forward of (pipe, p) forked onto the side stream.
        EVLM_TEACHER_FORK = start | vision_done | text_done (default) | fusion_done | forward_done: where the fork sits in the
        student's forward;
        EVLM_TEACHER_JOIN = end | vision: join at the end of the step, or when backward enters the image encoder (the ViT
        backward - large, chip-filling GEMMs - then runs alone; the teacher shares the chip with the text-side work)."""
        cur, side = torch.cuda.current_stream(), self._side
        # (round 4, 40-step A/Bs on one box: fork at "start" 14.79 / 14.80 / 14.78 ms, at "text_done" - the student's image and
        # text encoders are through, its small fusion / head kernels begin - 14.65 / 14.67 / 14.66, "vision_done" 16.6,
        # "fusion_done" 15.4, "forward_done" 15.5)
        fork_at = os.environ.get("EVLM_TEACHER_FORK", "text_done")
        join_at = os.environ.get("EVLM_TEACHER_JOIN", "end")
        state = {"forked": False}

        def fork():
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._teacher_eager(pipe, p)
            state["forked"] = True

        def phase(name):
            if name == fork_at and not state["forked"]:
                fork()

        if fork_at == "start":
            fork()
        self.student.phase_hook = phase
        if join_at == "vision":
            self._join_at_vision = lambda: cur.wait_stream(side) if state["forked"] else None
        try:
            out = self._student_eager(pp, pk)
        finally:
            self.student.phase_hook = None
            self._join_at_vision = None
        if not state["forked"]:
            fork()
        cur.wait_stream(side)
        return out

    def step(self, batch, lr_mult=1.0):
        """one GD step on a general or a region batch (the latter carries idx_to_group_img / image_atts / target_bbox /
        is_image); returns a device tensor [total, itc, itm, mlm, kd] (no host sync)."""
        if self._wants_pool_stream(batch["image"].is_cuda):
            # N > 1: NEVER on the legacy default stream.  On this stack a process-group collective - issued from any stream,
            # async or not - makes the DEFAULT stream wait for the group's own stream: a training loop that runs there
            # stalls at every gradient stage until its all-reduce has finished (measured with a simulated wire: 18.4 ms
            # per step on the default stream, 16.0 on a pool stream, 15.3 without any exchange;
            # profiles/r05_exchange_overlap.md).  The pruning trainers' captured steps already run on a stream of their own.
            return self._off_default_stream(lambda: self._step_on_current(batch, lr_mult))
        return self._step_on_current(batch, lr_mult)

    def _step_on_current(self, batch, lr_mult):
        if self.pipeline_teacher:
            return self._step_pipelined(batch, lr_mult)
        return self._step_unpipelined(batch, lr_mult)

    def _step_unpipelined(self, batch, lr_mult):
        if not self.use_graph or self.reducer.active:
            # multi-GPU: the step runs eagerly - RCCL collectives cannot be captured into a hipGraph on this stack
            # (a capture probe core-dumped); with pipeline_teacher the teacher half still replays a graph
            self.opt.set_schedule(lr_mult)
            return self._step_eager(batch)
        sig = tuple(sorted((k, tuple(v.shape)) for k, v in batch.items()))     # one hipGraph per batch kind / shape
        if sig not in self._graphs:
            self._graphs[sig] = self._capture(batch)
        graph, static, out, self.last_kd = self._graphs[sig]
        self.graph, self.static, self.out = graph, static, out
        self.opt.set_schedule(lr_mult)
        for k, v in batch.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return out

    def _capture(self, batch):
        static = {k: v.clone() for k, v in batch.items()}
        state = [(g["m"].clone(), g["v"].clone()) for g in self.opt.groups], self.opt.step_count
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                             # warm-up on a side stream (allocator + weight caches);
                self.opt.set_schedule(0.0)                 # lr multiplier 0: parameters are left untouched
                self._step_eager(static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.CACHE.invalidate()                             # capture the casts of the trainable weights too
        ops.reserve_tables()
        self.opt.set_schedule(0.0)                         # (the captured optimiser step reads the staged scalars at replay)
        graph = torch.cuda.CUDAGraph()
        pool = next(iter(self._graphs.values()))[0].pool() if self._graphs else None   # the kinds never run concurrently
        with no_gc_during_capture(), torch.cuda.graph(graph, pool=pool):
            out = self._step_eager(static)
        ops.flush_table_uploads()
        # the warm-up advanced Adam's step count and moments: restore them so replay k is optimiser step k
        for g, (m, v) in zip(self.opt.groups, state[0]):
            g["m"].copy_(m)
            g["v"].copy_(v)
        self.opt.step_count = state[1]
        return graph, static, out, self.last_kd


class TeacherPrefetch:
    """The frozen teacher's forward, one batch ahead of its consumer, on a side stream.  submit(batch) copies the batch
    into a static buffer, starts the teacher on it and returns (static batch, teacher outputs) of the PREVIOUS submit
    (None the first time); the caller trains the student on that pair while the new teacher forward shares the chip.
    With graphs on, each parity of each batch shape replays its own hipGraph from its OWN memory pool: a graph's output
    tensors then stay valid until that same graph is replayed two submits later - exactly as long as they are needed -
    so nothing is copied."""

    # batch shapes whose teacher forward is captured (first come: each holds two private pools).  Round 6: a bucket-padded
    # epoch (data.bucket_pad_itr / _vqa) has a handful of shapes - 4 text lengths, or ~3 question lengths x ~2 answer-row
    # counts - and ALL of them should replay; EVLM_MAX_GRAPH_KINDS overrides
    MAX_GRAPH_KINDS = int(os.environ.get("EVLM_MAX_GRAPH_KINDS", "12"))
    MAX_EAGER_KINDS = 4         # ... further shapes run the same kernels eagerly; least recently used static buffers evicted

    def __init__(self, run_teacher, use_graph=True):
        self.run_teacher = run_teacher          # fn(batch dict) -> teacher outputs (any nest of tensors)
        self.use_graph = use_graph
        self.side = torch.cuda.Stream()
        self.states, self.pending, self.last_key = {}, None, None
        self._tick = 0

    def close(self):
        """drop the graphs / static buffers (caller: trainer.close(), device idle)"""
        self.states.clear()
        self.pending = None

    def _create(self, batch):
        cur = torch.cuda.current_stream()
        st = dict(B=[{k: v.clone() for k, v in batch.items()} for _ in range(2)], T=[None, None], graphs=None, par=0)
        graphed = sum(1 for q in self.states.values() if q["graphs"] is not None)
        if self.use_graph and graphed >= self.MAX_GRAPH_KINDS:
            # bounded state: real VQA batches differ in their number of answer rows from batch to batch
            eager = sorted((q.get("tick", 0), sig) for sig, q in self.states.items()
                           if q["graphs"] is None and (self.pending is None or q is not self.pending[0]))
            if len(eager) >= self.MAX_EAGER_KINDS:
                torch.cuda.synchronize()
                del self.states[eager[0][1]]
        elif self.use_graph:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.run_teacher(st["B"][0])                   # warm-up: allocator, cached weight casts
            torch.cuda.synchronize()
            ops.reserve_tables()
            st["graphs"] = []
            for k in (0, 1):
                g = torch.cuda.CUDAGraph()
                with no_gc_during_capture(), torch.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):    # (no shared pool: see the class docstring)
                    st["T"][k] = self.run_teacher(st["B"][k])
                st["graphs"].append(g)
            torch.cuda.synchronize()
            ops.flush_table_uploads()
        return st

    def submit(self, batch):
        batch = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        sig = tuple(sorted((k, tuple(v.shape)) for k, v in batch.items()))
        st = self.states.get(sig)
        if st is None:
            st = self.states[sig] = self._create(batch)
        self._tick += 1
        st["tick"] = self._tick
        cur, side = torch.cuda.current_stream(), self.side
        cur.wait_stream(side)                     # the waiting batch's teacher outputs are complete
        p = st["par"] = 1 - st["par"]
        for name, v in batch.items():
            st["B"][p][name].copy_(v, non_blocking=True)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if st["graphs"] is not None:
                st["graphs"][p].replay()
            else:
                st["T"][p] = self.run_teacher(st["B"][p])
                for t in distill._tensors(st["T"][p]):     # allocated on the side stream, consumed on the main one
                    t.record_stream(cur)
        if os.environ.get("EVLM_PREFETCH_SERIAL"):    # (A/B switch: the teacher's forward runs to its end BEFORE the student step)
            cur.wait_stream(side)
        prev, self.pending = self.pending, (st, p)
        # (identity of the static buffers the returned pair lives in - a captured student step is keyed by it - or None when
        # the teacher ran eagerly into fresh tensors)
        self.last_key = None if (prev is None or prev[0]["graphs"] is None) else (id(prev[0]), prev[1])
        return None if prev is None else (prev[0]["B"][prev[1]], prev[0]["T"][prev[1]])


class _CapturedStep:
    """The pruning fine-tune step replayed from hipGraphs, one (chain) per static (batch, teacher outputs) pair of the teacher
    prefetch (`capture_step=True`, the default; needs the prefetched teacher with graphs).  What made the step host-dependent
    enters through device memory that is refilled before every replay: the Lagrangian warm-up counter (`_pruned_dev`; the
    reference ramps the target sparsity with a Python step count, Eff_Retrieval.py:113), the gate noise (`l0.static_eps`:
    drawn on the HOST generator in the order and shapes one eager forward draws them - the same stream of numbers as the
    eager trainer, and as the reference's CPU draws) and the three optimisers' schedules (FlatAdamW / TensorAdamW
    `.set_schedule`).  The first step on a pair runs eagerly (it records the gate-noise plan of its KIND - for the VQA
    trainer `stop_prune` is part of the kind: a stop_prune step evaluates the deterministic gates and draws nothing - and
    warms allocator and weight caches), the second is captured, later ones replay.

    One GPU: ONE graph per pair.  N > 1 (a live reducer): a chain of graph SEGMENTS cut at the step's collectives, exactly
    as GDTrainer._capture_segments does - the ITC feature (and image-id) gathers of the retrieval forward through
    xvlm.GATHER_HOOK, the gradient stages through _StagedExchange._cut from the hooks inside backward, [three optimisers +
    constrain_parameters] behind the last all-reduce - with the collectives issued eagerly between the replays
    (torch DDP's bucketed all-reduce under backward: Eff_Retrieval.py:449, Eff_VQA.py:327).  The eager step and the chain
    issue the SAME collective sequence (tests/test_step_gpu.py), so a rank whose capture fails simply keeps stepping
    eagerly: no agreement round is held here - VQA batch kinds (answer rows) differ from rank to rank, so ranks reach
    their captures at different steps and a collective at that moment would pair with a peer's gradient all-reduce."""
    capture_step = False
    # static pairs whose step is captured (first come); pairs beyond that step eagerly.  (kinds x 2 teacher parities [x 2
    # stop_prune]; the graphs share one memory pool.)  EVLM_MAX_STEP_GRAPHS overrides
    MAX_STEP_GRAPHS = int(os.environ.get("EVLM_MAX_STEP_GRAPHS", "32"))

    def _cap_init(self):
        dev = next(self.student.parameters()).device
        self._sgraphs, self._seen, self._eps = {}, set(), {}
        self._pruned_dev = torch.zeros((), dtype=torch.float32, device=dev)
        self._cap_stream, self._step_pool = None, None
        self._capture_failed = None
        self.last_launch = "eager"

    def _stage_eps(self, kind):
        """this step's gate noise -> the static device buffers of its kind; an injected draw (tests) is consumed by ONE step,
        as XVLML0Module.forward(training=True) consumes it; a kind without draws (stop_prune) touches no generator"""
        plan, static = self._eps[kind]
        if not plan:
            return
        l0 = self.student.l0_module
        inj = l0.injected_eps
        for typ, shape in plan:
            src = inj[typ] if (inj is not None and typ in inj) else l0.get_eps(shape)
            src = src.to(torch.float32)
            if not src.is_cuda:
                src = src.contiguous().pin_memory()
            static[typ].copy_(src, non_blocking=True)
        l0.injected_eps = None

    def _set_scheduled(self, flag):
        self.opt._scheduled = self.l0_opt._scheduled = self.lagrangian_opt._scheduled = flag

    # The Lagrangian sparsity term depends on nothing but the gate parameters, so it could run on a side stream beside the
    # student forward (EVLM_L0_STREAM=1; one GPU only: a branch may not straddle a segment cut).  MEASURED SLOWER and off: as
    # ~90 small launches on a third stream of the captured step it cost +2.0 ms per ITR-384 step and +2.4 ms per VQA-480 step
    # (profiles/r05_pruning_residue.md) - the graph's cross-stream hand-overs outweigh what the branch hides.  The term is
    # now one kernel launch each way instead (ops.l0_lagrangian).
    def _lagrangian_begin(self, pruned_steps):
        l0 = self.student.l0_module
        if self.reducer.active or not os.environ.get("EVLM_L0_STREAM") or not self._pruned_dev.is_cuda:
            return None
        if getattr(self, "_l0_stream", None) is None:
            self._l0_stream = torch.cuda.Stream()
        cur, ls = torch.cuda.current_stream(), self._l0_stream
        ls.wait_stream(cur)
        with torch.cuda.stream(ls):
            lag = l0.lagrangian_regularization(pruned_steps)[0]
        if not torch.cuda.is_current_stream_capturing():
            lag.record_stream(cur)
        return lag

    def _lagrangian_end(self, lag, pruned_steps):
        if lag is None:
            return self.student.l0_module.lagrangian_regularization(pruned_steps)[0]
        torch.cuda.current_stream().wait_stream(self._l0_stream)
        return lag

    def _capture_one(self, body, cs):
        """single GPU: the whole step as one graph"""
        g = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(g, pool=self._step_pool, stream=cs, capture_error_mode="thread_local"):
            out = body(self._pruned_dev, True)
        self._step_pool = g.pool()
        return dict(segs=[("graph", g)], out=out, last_reduce=None)

    def _capture_chain(self, body, cs):
        """N > 1: the step as graph segments around its collectives (no collective is issued during the capture pass)"""
        from .efficient_models import xvlm as X
        if self._step_pool is None:
            self._step_pool = torch.cuda.graph_pool_handle()
        cur = torch.cuda.current_stream()
        segs, state = [], {"g": None}

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            # (thread_local: the process group's watchdog thread queries events while this thread captures)
            state["g"].capture_begin(pool=self._step_pool, capture_error_mode="thread_local")

        def end():
            state["g"].capture_end()
            segs.append(("graph", state["g"]))
            state["g"] = None

        def gather(out_list, src):
            end()
            segs.append(("gather", (out_list, src)))
            begin()

        def cut(ranges):
            end()
            segs.append(("reduce", ranges))
            begin()

        cs.wait_stream(cur)
        self._cut = cut
        try:
            # (autograd single-threaded: the gradient-stage hooks end / begin captures on the thread that started them)
            with no_gc_during_capture(), torch.cuda.stream(cs), torch.autograd.set_multithreading_enabled(False):
                X.GATHER_HOOK = gather
                try:
                    begin()
                    out = body(self._pruned_dev, True)       # (its _reduce_rest() cuts once per remaining stage)
                    end()
                finally:
                    X.GATHER_HOOK = None
                    if state["g"] is not None:               # an exception inside a capture: close it before re-raising
                        try:
                            state["g"].capture_end()
                        except RuntimeError:
                            pass
        finally:
            self._cut = None
        cur.wait_stream(cs)
        torch.cuda.synchronize()
        return dict(segs=segs, out=out, last_reduce=[item for kind, item in segs if kind == "reduce"][-1])

    def _run_step(self, key, body, lr_mult):
        """body(pruned_steps, staged) -> device loss stack"""
        l0 = self.student.l0_module
        if not self.capture_step or key is None:
            self.last_launch = "eager"
            if self._wants_pool_stream(torch.cuda.is_available()):
                return self._off_default_stream(lambda: body(self.global_step, False))      # (see GDTrainer.step)
            return body(self.global_step, False)
        # Every step of a capturing trainer - the eager ones too - runs on ONE dedicated stream: autograd remembers the stream
        # an AccumulateGrad node was created on and synchronises a later backward with it, which inside a capture would pull
        # a non-capturing stream into the graph (hipGraphInstantiate then faults on this stack)
        cur = torch.cuda.current_stream()
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
        cs = self._cap_stream
        kind = tuple(key[2:])
        ent = self._sgraphs.get(key)
        first = key not in self._seen
        if first or (ent is None and (len(self._sgraphs) >= self.MAX_STEP_GRAPHS or self._capture_failed)):
            # first step on this pair: eager, recording the gate-noise plan of its kind; also every step of a pair beyond the
            # bound on captured pairs (real VQA batches vary in their answer rows: each shape would hold a step's
            # activations in its graph's pool for good) and of a trainer whose capture this stack refused
            self._seen.add(key)
            l0.eps_trace = []
            cs.wait_stream(cur)
            try:
                with torch.cuda.stream(cs):
                    out = body(self.global_step, False)
            finally:
                plan, l0.eps_trace = l0.eps_trace, None
            cur.wait_stream(cs)
            if kind not in self._eps:
                dev = self._pruned_dev.device
                self._eps[kind] = (plan, {t: torch.empty(shape, dtype=torch.float32, device=dev) for t, shape in plan})
            elif plan != self._eps[kind][0]:
                raise RuntimeError("captured pruning step: the gate-noise draws of this batch kind differ from the recorded plan")
            self.last_launch = "eager"
            return out
        self._pruned_dev.fill_(float(self.global_step))
        self._stage_eps(kind)
        self.opt.set_schedule(lr_mult)
        self.l0_opt.set_schedule()
        self.lagrangian_opt.set_schedule()
        static = self._eps[kind][1]
        if ent is None:
            ops.CACHE.invalidate()                    # capture the casts of the trainable weights too
            ops.reserve_tables()
            torch.cuda.synchronize()
            l0.static_eps = static
            err = None
            try:
                ent = self._capture_chain(body, cs) if self.reducer.active else self._capture_one(body, cs)
            except RuntimeError as e:                 # a capture this stack refuses
                if not self.reducer.active:
                    raise
                err = e
                try:
                    torch.cuda.synchronize()
                except RuntimeError:
                    pass
            finally:
                l0.static_eps = None
            # (the capture pass consumed the optimisers' "staged" flags, not the staged values: the replay reads them)
            if err is not None:
                import sys
                print(f"[efficientvlm_amd] hipGraph segments of the multi-GPU pruning step failed ({err}); this rank keeps "
                      "stepping eagerly (same collective sequence)", file=sys.stderr)
                self._capture_failed = str(err)
                # this step: the captured body executed eagerly (schedules and noise are staged already)
                self._set_scheduled(True)
                l0.static_eps = static
                cs.wait_stream(cur)
                try:
                    with torch.cuda.stream(cs):
                        out = body(self._pruned_dev, True)
                finally:
                    l0.static_eps = None
                cur.wait_stream(cs)
                self.last_launch = "eager"
                return out
            ops.flush_table_uploads()
            self._sgraphs[key] = ent
        # EVLM_REPLAY_PRIORITY=1: the graph replays on a HIGH-priority stream (its kernels are the step's critical path, the
        # teacher's graph of the next batch runs on the prefetch's side stream): 44.7 -> 44.3 ms on the ITR step, two A/B
        # pairs.  Opt-in: queue priority on this stack is close to strict - the GD step as two graphs with the student on
        # a high-priority stream SERIALISED the teacher behind it (22.1 ms against 16.0 without priority and 14.8 for the
        # joint graph); a priority on the joint graph's capture stream changes nothing.
        if not os.environ.get("EVLM_REPLAY_PRIORITY") or self.reducer.active:
            self._replay_segments(ent["segs"], ent["last_reduce"])
        else:
            if getattr(self, "_hp_stream", None) is None:
                self._hp_stream = torch.cuda.Stream(priority=-1)
            hp = self._hp_stream
            hp.wait_stream(cur)
            with torch.cuda.stream(hp):
                self._replay_segments(ent["segs"], ent["last_reduce"])
            cur.wait_stream(hp)
        self._set_scheduled(False)
        self.last_launch = "hipGraph segments" if self.reducer.active else "hipGraph replay"
        return ent["out"]


class ITRTrainer(_StagedExchange, _CapturedStep):
    """Pruning fine-tune step of Eff_Retrieval.py:75-213 (image-text retrieval with hard-concrete L0 gates): student with
    gates forward + backward, teacher forward, ITC + ITM + hidden / attention / cross-attention / logit KD, the Lagrangian
    sparsity term, THREE optimisers (main AdamW over every student parameter - the gate parameters included, as in the
    reference -, +reg_lr on the gate log-alphas, -reg_lr = ascent on lambda_1 / lambda_2; optim.py:4-69), no gradient
    clipping (the reference calls optimizer.step() directly here), then constrain_parameters().  With the teacher prefetched the step
    replays from hipGraphs (_CapturedStep: one graph on one GPU, segments around the collectives with N > 1)."""

    def __init__(self, student, teacher, lr=3e-5, weight_decay=0.01, lr_mult=2.0, reg_learning_rate=0.1,
                 dtype=torch.float32, temperature=1.0, pipeline_teacher=False, use_graph=True, capture_step=True):
        """pipeline_teacher: as in GDTrainer - the frozen teacher runs one batch ahead (TeacherPrefetch: hipGraph on a side
        stream) and step() returns the losses of the batch of the PREVIOUS call (None on the first).
        capture_step (default on; effective with pipeline_teacher and use_graph): the student step replays from hipGraphs
        too - one graph on one GPU, a chain of segments around the collectives with N > 1 (_CapturedStep);
        EVLM_NO_STEP_GRAPH=1 turns it off."""
        from .optim import create_L0_optimizer
        self.student, self.teacher = student, teacher
        self.dtype, self.temperature = dtype, temperature
        for p in teacher.parameters():
            p.requires_grad_(False)
        teacher.eval()
        student.train()
        self.opt = FlatAdamW(student, lr=lr, weight_decay=weight_decay, lr_mult=lr_mult, max_grad_norm=0.0)
        # (EVLM_FORCE_REDUCE=1: the N > 1 code path - collectives, gradient stages, graph segments - on a one-rank group)
        self.reducer = GradReducer(self.opt.flat_grads, prescaled=True, force=bool(os.environ.get("EVLM_FORCE_REDUCE")))
        self._stages, self._sent = [list(self.opt.flat_grads)], 0
        if self.reducer.active:
            broadcast_parameters(self.opt)      # gates and multipliers included: they are members of the main groups
            # the exchange overlaps backward as torch DDP's buckets do in the reference (Eff_Retrieval.py:449, Eff_VQA.py:327):
            # text side + ViT layers {4,5} go out when backward reaches ViT layer 3, {2,3} at layer 1, the rest after backward
            self._stages, _ = install_grad_stages(self, self.opt, student, os.environ.get("EVLM_DP_CUTS", "vit").replace("all", "vit"))
        self.l0_opt, self.lagrangian_opt = create_L0_optimizer({"reg_learning_rate": reg_learning_rate}, student.l0_module)
        self.defer_wgrad = dtype == torch.bfloat16 and not os.environ.get("EVLM_NO_DEFER_WGRAD")
        self.overlap_teacher = not os.environ.get("EVLM_NO_OVERLAP_TEACHER")
        if not os.environ.get("EVLM_TEACHER_ALL_MAPS"):
            teacher_map_filter(student, teacher, with_cross=True)
        self.global_step = 0
        self.prefetch = TeacherPrefetch(self._teacher_forward, use_graph) if pipeline_teacher else None
        self.capture_step = bool(capture_step and pipeline_teacher and use_graph and next(student.parameters()).is_cuda
                                 and not os.environ.get("EVLM_NO_STEP_GRAPH"))
        self._cap_init()
        enc = getattr(getattr(student, "vision_encoder", None), "encoder", None)
        if enc is not None and hasattr(enc, "kd_drop_maps") and not os.environ.get("EVLM_STUDENT_ALL_MAPS"):
            enc.kd_drop_maps = True        # a ViT map whose distillation term was formed in-kernel is not materialised
        # the prefetched teacher's kept image maps have ONE reader - the student's fused distillation kernels (bf16) - so on
        # long key sequences it keeps its QKV buffers + row lse instead (ops.MapRecipe: 115 MB per layer at 577 tokens
        # instead of a 517 MB map written once and read twice); EVLM_NO_KD_RECIPE=1: maps as before
        tenc = getattr(getattr(teacher, "vision_encoder", None), "encoder", None)
        if (tenc is not None and hasattr(tenc, "attn_recipe") and pipeline_teacher and dtype == torch.bfloat16
                and not os.environ.get("EVLM_NO_FUSED_KD") and not os.environ.get("EVLM_TEACHER_ALL_MAPS")):
            tenc.attn_recipe = True

    def _teacher_forward(self, b):
        with torch.no_grad(), compute(self.dtype):
            return self.teacher(b["image"], b["text_ids"], b["text_atts"], idx=b.get("idx"), output_attentions=True,
                                output_hidden_states=True)

    def step(self, batch, idx=None, lr_mult=1.0):
        """batch: dict(image, text_ids, text_atts); idx: image ids for the soft ITC labels.  Returns a device tensor
        [total, itc, itm, kd, lagrangian]."""
        T_ready, key = None, None
        if self.prefetch is not None:
            prev = self.prefetch.submit(dict(batch, idx=idx) if idx is not None else dict(batch))
            if prev is None:
                return None
            batch, T_ready = prev
            idx = batch.get("idx")
            key = self.prefetch.last_key
        out = self._run_step(key, lambda pruned, staged: self._body(batch, idx, T_ready, pruned, lr_mult, staged), lr_mult)
        self.global_step += 1
        return out

    def _body(self, batch, idx, T_ready, pruned_steps, lr_mult, staged):
        """the device work of one step (capturable when T_ready / batch are static and `staged`: schedules set outside)"""
        ops.begin_step(batch["image"].device)    # ONE fill for the step's loss words and small zeroed buffers (as the GD step)
        try:
            return self._body_in_arena(batch, idx, T_ready, pruned_steps, lr_mult, staged)
        finally:
            ops.end_step()

    def _body_in_arena(self, batch, idx, T_ready, pruned_steps, lr_mult, staged):
        self.opt.zero_grad()
        l0 = self.student.l0_module
        with compute(self.dtype):
            lag = self._lagrangian_begin(pruned_steps)
            kw = dict(idx=idx, output_attentions=True, output_hidden_states=True)
            fused = {}
            if T_ready is not None:      # the teacher's maps exist: the image-map term is formed inside the attention kernels
                S, fused = distill.student_forward_fused_kd(
                    self.student, lambda: self.student(batch["image"], batch["text_ids"], batch["text_atts"], **kw),
                    T_ready, batch)
                T = T_ready
            else:
                S, T = distill.student_and_teacher(
                    lambda: self.student(batch["image"], batch["text_ids"], batch["text_atts"], **kw),
                    lambda: self.teacher(batch["image"], batch["text_ids"], batch["text_atts"], **kw),
                    batch["image"], self.overlap_teacher)
            # (a bucket-padded batch - data.bucket_pad_itr - carries its real text length in device words: the text-side terms
            # skip the padded token rows and are rescaled to the 'longest'-padded denominators)
            kd = distill.kd_terms(S, T, self.temperature, with_cross_attn=True, fused=fused, ragged=distill.batch_ragged(batch)[0])
            lagrangian = self._lagrangian_end(lag, pruned_steps)
            total, mix = distill.itr_loss_mix(S["loss"], kd, lagrangian, kd_corr=batch.get("kd_corr"))
            ops.WGRAD_INPLACE = True
            ops.WGRAD_DEFER = [] if self.defer_wgrad else None
            self._sent = 0
            try:
                self.reducer.scale_loss(total).backward()   # (multi-GPU: the gradient stages leave from hooks inside it)
                ops.flush_wgrad()
            finally:
                ops.WGRAD_INPLACE = False
                ops.WGRAD_DEFER = None
                ops.LN_DEFER.clear()             # (empty after a flush; stale only when backward raised)
        if ops.DROPOUT_USED:              # hard-negative draws (device Philox stream): a new step word per step / replay
            ops.dropout_tick(total.device)
        if self.reducer.active:
            self._reduce_rest()
        if not staged:
            self.opt.set_schedule(lr_mult)
        self.opt.step()
        self.l0_opt.step()
        self.lagrangian_opt.step()
        l0.constrain_parameters()
        return torch.stack([total.detach().float(), S["loss"]["loss_itc"].detach().float(),
                            S["loss"]["loss_itm"].detach().float(), mix["loss_kd"].detach().float(),
                            lagrangian.detach().float().reshape(())])


class VQATrainer(_StagedExchange, _CapturedStep):
    """Pruning fine-tune step of Eff_VQA.py:74-200 (visual question answering with hard-concrete L0 gates on the image
    encoder, question encoder AND answer decoder): student forward + backward, teacher forward, the weighted answer LM loss,
    text / fusion / image / decoder hidden + attention KD, logit KD, the Lagrangian, THREE optimisers as in ITRTrainer, no
    gradient clipping, constrain_parameters().  With the teacher prefetched the student step replays from hipGraphs per
    (batch shape, teacher-prefetch parity, stop_prune) - _CapturedStep; a new number of answer rows is a new batch kind and
    starts with an eager step (bounded: MAX_STEP_GRAPHS pairs are captured, later kinds step eagerly)."""

    def __init__(self, student, teacher, lr=5e-5, weight_decay=0.01, lr_mult=2.0, reg_learning_rate=0.1,
                 dtype=torch.float32, temperature=1.0, pipeline_teacher=False, use_graph=True, capture_step=True):
        from .optim import create_L0_optimizer
        self.student, self.teacher = student, teacher
        self.dtype, self.temperature = dtype, temperature
        for p in teacher.parameters():
            p.requires_grad_(False)
        teacher.eval()
        student.train()
        self.opt = FlatAdamW(student, lr=lr, weight_decay=weight_decay, lr_mult=lr_mult, max_grad_norm=0.0)
        # (EVLM_FORCE_REDUCE=1: the N > 1 code path - collectives, gradient stages, graph segments - on a one-rank group)
        self.reducer = GradReducer(self.opt.flat_grads, prescaled=True, force=bool(os.environ.get("EVLM_FORCE_REDUCE")))
        self._stages, self._sent = [list(self.opt.flat_grads)], 0
        if self.reducer.active:
            broadcast_parameters(self.opt)      # gates and multipliers included: they are members of the main groups
            # the exchange overlaps backward as torch DDP's buckets do in the reference (Eff_Retrieval.py:449, Eff_VQA.py:327):
            # text side + ViT layers {4,5} go out when backward reaches ViT layer 3, {2,3} at layer 1, the rest after backward
            self._stages, _ = install_grad_stages(self, self.opt, student, os.environ.get("EVLM_DP_CUTS", "vit").replace("all", "vit"))
        self.l0_opt, self.lagrangian_opt = create_L0_optimizer({"reg_learning_rate": reg_learning_rate}, student.l0_module)
        self.defer_wgrad = dtype == torch.bfloat16 and not os.environ.get("EVLM_NO_DEFER_WGRAD")
        self.overlap_teacher = not os.environ.get("EVLM_NO_OVERLAP_TEACHER")
        if not os.environ.get("EVLM_TEACHER_ALL_MAPS"):     # (as the ITR trainer: only the maps Eff_VQA.py:113-163 reads)
            teacher_map_filter(student, teacher, with_cross=True)
        self.global_step = 0
        self.prefetch = TeacherPrefetch(self._teacher_forward, use_graph) if pipeline_teacher else None
        self.capture_step = bool(capture_step and pipeline_teacher and use_graph and next(student.parameters()).is_cuda
                                 and not os.environ.get("EVLM_NO_STEP_GRAPH"))
        self._cap_init()
        enc = getattr(getattr(student, "vision_encoder", None), "encoder", None)
        if enc is not None and hasattr(enc, "kd_drop_maps") and not os.environ.get("EVLM_STUDENT_ALL_MAPS"):
            enc.kd_drop_maps = True        # a ViT map whose distillation term was formed in-kernel is not materialised
        # the prefetched teacher's kept image maps have ONE reader - the student's fused distillation kernels (bf16) - so on
        # long key sequences it keeps its QKV buffers + row lse instead (ops.MapRecipe: 115 MB per layer at 577 tokens
        # instead of a 517 MB map written once and read twice); EVLM_NO_KD_RECIPE=1: maps as before
        tenc = getattr(getattr(teacher, "vision_encoder", None), "encoder", None)
        if (tenc is not None and hasattr(tenc, "attn_recipe") and pipeline_teacher and dtype == torch.bfloat16
                and not os.environ.get("EVLM_NO_FUSED_KD") and not os.environ.get("EVLM_TEACHER_ALL_MAPS")):
            tenc.attn_recipe = True

    def _teacher_forward(self, b):
        from types import SimpleNamespace as NS
        with torch.no_grad(), compute(self.dtype):
            return self.teacher(b["image"], NS(input_ids=b["question_ids"], attention_mask=b["question_atts"]),
                                NS(input_ids=b["answer_ids"], attention_mask=b["answer_atts"]), train=True, k=b["k"],
                                weights=b["weights"], output_attentions=True, output_hidden_states=True)

    def step(self, batch, lr_mult=1.0, stop_prune=False):
        """batch: dict(image [B], question_ids / question_atts [B, Lq], answer_ids / answer_atts [sum k, La], k [B] (tensor
        or list), weights [sum k]).  Returns a device tensor [total, answer loss, kd, lagrangian]."""
        T_ready, key = None, None
        if self.prefetch is not None:
            batch = dict(batch, k=torch.as_tensor(batch["k"], device=batch["image"].device))
            prev = self.prefetch.submit(batch)
            if prev is None:
                return None
            batch, T_ready = prev
            key = None if self.prefetch.last_key is None else self.prefetch.last_key + (bool(stop_prune),)
        out = self._run_step(key, lambda pruned, staged: self._body(batch, T_ready, pruned, lr_mult, stop_prune, staged), lr_mult)
        self.global_step += 1
        return out

    def _body(self, batch, T_ready, pruned_steps, lr_mult, stop_prune, staged):
        """the device work of one step (capturable when T_ready / batch are static and `staged`: schedules set outside)"""
        ops.begin_step(batch["image"].device)
        try:
            return self._body_in_arena(batch, T_ready, pruned_steps, lr_mult, stop_prune, staged)
        finally:
            ops.end_step()

    def _body_in_arena(self, batch, T_ready, pruned_steps, lr_mult, stop_prune, staged):
        from types import SimpleNamespace as NS
        self.opt.zero_grad()
        l0 = self.student.l0_module
        question = NS(input_ids=batch["question_ids"], attention_mask=batch["question_atts"])
        answer = NS(input_ids=batch["answer_ids"], attention_mask=batch["answer_atts"])
        kw = dict(train=True, k=batch["k"], weights=batch["weights"], output_attentions=True, output_hidden_states=True)
        with compute(self.dtype):
            lag = self._lagrangian_begin(pruned_steps)
            fused = {}
            if T_ready is not None:
                S, fused = distill.student_forward_fused_kd(
                    self.student, lambda: self.student(batch["image"], question, answer, stop_prune=stop_prune, **kw),
                    T_ready, batch)
                T = T_ready
            else:
                S, T = distill.student_and_teacher(
                    lambda: self.student(batch["image"], question, answer, stop_prune=stop_prune, **kw),
                    lambda: self.teacher(batch["image"], question, answer, **kw), batch["image"], self.overlap_teacher)
            kd = distill.vqa_kd_terms(S, T, self.temperature, fused=fused, ragged=distill.batch_ragged(batch))
            lagrangian = self._lagrangian_end(lag, pruned_steps)
            total, mix = distill.vqa_loss_mix(S["loss"], kd, lagrangian, kd_corr=batch.get("kd_corr"))
            ops.WGRAD_INPLACE = True
            ops.WGRAD_DEFER = [] if self.defer_wgrad else None
            self._sent = 0
            try:
                self.reducer.scale_loss(total).backward()   # (multi-GPU: the gradient stages leave from hooks inside it)
                ops.flush_wgrad()
            finally:
                ops.WGRAD_INPLACE = False
                ops.WGRAD_DEFER = None
                ops.LN_DEFER.clear()             # (empty after a flush; stale only when backward raised)
        if ops.DROPOUT_USED:              # hard-negative draws (device Philox stream): a new step word per step / replay
            ops.dropout_tick(total.device)
        if self.reducer.active:
            self._reduce_rest()
        if not staged:
            self.opt.set_schedule(lr_mult)
        self.opt.step()
        self.l0_opt.step()
        self.lagrangian_opt.step()
        l0.constrain_parameters()
        return torch.stack([total.detach().float(), S["loss"].detach().float(), mix["loss_kd"].detach().float(),
                            lagrangian.detach().float().reshape(())])
