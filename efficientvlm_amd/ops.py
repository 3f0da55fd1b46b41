"""Autograd-aware host wrappers over the C ABI of libevlm_hip.so.

Every function here launches hand-written gfx950 kernels on the current HIP stream through ctypes
(`_lib`); PyTorch only supplies device memory, streams and the autograd tape.  Nothing falls back to
ATen math: a CPU tensor or a missing library raises.

Compute dtype: activations are either torch.float32 (exact-fp32 parity path, fp32-input MFMA) or
torch.bfloat16 (fast path: bf16 storage, fp32 accumulate).  Parameters stay fp32 (checkpoint ABI); the
bf16 path reads packed bf16 compute copies from `WeightCache`.
"""
import ctypes as C
import os
import math

import torch

from . import _lib as L

_EMPTY = {}
_NO_FORK = bool(__import__("os").environ.get("EVLM_NO_FORK"))      # measurement aid: residual-gradient adds left to autograd


def _lib():
    return L.load()


# ---------------------------------------------------------------------------------------------------
# weight cache: packed compute copies of fp32 master parameters
# ---------------------------------------------------------------------------------------------------
class WeightCache:
    """compute-dtype copies of (tuples of) fp32 parameters, re-cast when a parameter's version changes.

    A tuple of [n_i, k] weights is packed row-wise into one [sum n_i, k] matrix (fused QKV / KV
    projections); biases likewise into one fp32 vector."""

    def __init__(self):
        self._store = {}
        self._slab = {}         # id(param) -> [fp32 slab, bf16 slab, offset, numel, version, weakref]
        self._t_units = {}      # (bf16 view pointer, rows, cols) -> (bf16 slab view [N, K], transposed copy [K, N])
        self._t_tables = {}     # owner (slab pointers | None) -> (device table, tiles, units, unit keys) of a grouped transpose
        self._t_keep = []
        self.epoch = 0          # bump to force a re-cast of trainable weights (e.g. before graph capture)

    def invalidate(self):
        self.epoch += 1

    # ---- transposed bf16 copies of slab-backed weights (W^T for dX = dY W) --------------------------------------------
    def get_t(self, params):
        """[K, sum N_i] bf16 = transpose of the packed bf16 weight of `params`, or None when they are not adjacent members
        of an optimiser slab.  Created (and filled) on first use, afterwards kept current by refresh_transposed(), which
        the optimiser calls after every parameter update."""
        if not self._slab or not params[0].is_cuda or NO_WT:
            return None
        src = self._from_slab(params, torch.bfloat16)
        if src is None or src.dim() != 2 or src.shape[0] % 8 or src.shape[1] % 8:
            return None
        key = (src.data_ptr(), src.shape[0], src.shape[1])
        u = self._t_units.get(key)
        if u is not None and u[2]() is not params[0]:          # the slab memory was recycled by another model
            u = None
        if u is None:
            import weakref
            dst = torch.empty((src.shape[1], src.shape[0]), dtype=torch.bfloat16, device=src.device)
            u = self._t_units[key] = (src, dst, weakref.ref(params[0]))
            self._t_keep.append(self._transpose([u])[0])
        return u[1]

    def _transpose(self, units):
        rows, tiles = [], 0
        for src, dst, _ in units:
            rows += [src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], tiles]
            tiles += ((src.shape[0] + 63) // 64) * ((src.shape[1] + 63) // 64)
        table = _upload_table(rows, units[0][0].device)       # (capture-safe: a table may have to be rebuilt inside one)
        L.check(_lib().evlm_transpose_grouped(L.ptr(table), len(units), tiles, L.stream()), "transpose_grouped")
        return table, tiles

    def refresh_transposed(self, slabs=None):
        """re-derive the W^T copies from the bf16 mirror: ONE grouped launch (capturable: the table is a device tensor).
        slabs (the bf16 parameter slabs of ONE optimiser): only the copies of weights inside them - what a trainer's captured
        step may bake into its graph.  (One table over every model alive in the process would leave a graph transposing
        the weights of OTHER models into their copies - freed memory once those models are gone.)"""
        if not self._t_units:
            return
        dead = [k for k, u in self._t_units.items() if u[2]() is None]           # models that are gone
        if dead:
            for k in dead:
                del self._t_units[k]
            self._t_tables = {o: e for o, e in self._t_tables.items() if not (set(e[3]) & set(dead))}
        units = list(self._t_units.items())
        okey = None
        if slabs is not None:
            spans = [(s.data_ptr(), s.data_ptr() + s.numel() * s.element_size()) for s in slabs if s is not None]
            units = [(k, u) for k, u in units if any(lo <= u[0].data_ptr() < hi for lo, hi in spans)]
            okey = tuple(sorted(lo for lo, _ in spans))
        if not units:
            return
        ukeys = tuple(k for k, _ in units)
        ent = self._t_tables.get(okey)
        if ent is None or ent[3] != ukeys:
            table, tiles = self._transpose([u for _, u in units])
            self._t_tables[okey] = (table, tiles, len(units), ukeys)
            self._t_keep.append(table)                # a captured graph may hold an older table: tables are never freed
            return
        table, tiles, n, _ = ent
        L.check(_lib().evlm_transpose_grouped(L.ptr(table), n, tiles, L.stream()), "transpose_grouped")

    # ---- optimiser-owned parameter slabs (optim.FlatAdamW): the bf16 mirror is kept current by the AdamW kernel ----
    def register_slab(self, p, slab32, slab16, off, seg=None):
        """seg: slab words reserved for the parameter (>= numel; what lies between is ZERO and stays zero: optim._seg)"""
        import weakref
        self._slab[id(p)] = [slab32, slab16, off, p.numel(), p._version, weakref.ref(p), seg or p.numel()]

    def zero_padded_rows(self, W, rows):
        """may a [rows, K] operand be read in place at W (a [N, K] bf16 view, rows > N)?  Only when W is the mirror of a
        slab-backed parameter whose segment reserves - and keeps at zero - the words behind the matrix"""
        for e in self._slab.values():
            if e[1] is not None and W.data_ptr() == e[1].data_ptr() + e[2] * 2 and e[5]() is not None:
                return W.dim() == 2 and W.is_contiguous() and W.numel() == e[3] and rows * W.shape[1] <= e[6]
        return False

    def refresh_slab(self, slab32, slab16):
        """re-cast a whole slab (construction; or a parameter was modified outside the optimiser)"""
        L.check(_lib().evlm_cast(L.F32, L.ptr(slab32), L.BF16, L.ptr(slab16), slab32.numel(), L.stream()), "cast")
        self.refresh_transposed([slab16])
        for ent in self._slab.values():
            if ent[0] is slab32:
                pr = ent[5]()
                if pr is not None:
                    ent[4] = pr._version

    def _from_slab(self, params, dtype):
        ents = []
        for p in params:
            e = self._slab.get(id(p))
            if e is None or e[5]() is not p or p.data_ptr() != e[0].data_ptr() + e[2] * 4:
                return None
            ents.append(e)
        if dtype not in (torch.float32, torch.bfloat16) or any(e[0] is not ents[0][0] for e in ents):
            return None
        if dtype == torch.bfloat16:
            for p, e in zip(params, ents):
                if p._version != e[4]:
                    self.refresh_slab(e[0], e[1])
        for a, b in zip(ents[:-1], ents[1:]):            # packed operand: the members must be back to back
            if a[2] + a[3] != b[2]:
                return None
        if any(p.shape[1:] != params[0].shape[1:] for p in params):
            return None
        slab = ents[0][1] if dtype == torch.bfloat16 else ents[0][0]
        n = sum(e[3] for e in ents)
        rows = sum(p.shape[0] for p in params)
        return slab[ents[0][2]:ents[0][2] + n].view((rows,) + tuple(params[0].shape[1:]))

    def get(self, params, dtype):
        if self._slab and params[0].is_cuda:
            v = self._from_slab(params, dtype)
            if v is not None:
                return v
        key = (tuple(id(p) for p in params), dtype)
        vers = tuple(p._version for p in params)
        grad = any(p.requires_grad for p in params)
        ent = self._store.get(key)
        # id() values are recycled once a parameter dies: an entry is only valid for the very objects it was built from
        if ent is not None and not all(r() is p for r, p in zip(ent[4], params)):
            ent = None
        if ent is not None and ent[1] == vers and (not grad or ent[2] == self.epoch) and ent[3] == params[0].data_ptr():
            return ent[0]
        if len(params) == 1 and params[0].dtype == dtype and params[0].is_contiguous():
            buf = params[0].detach()
        else:
            rows = sum(p.shape[0] for p in params)
            shape = (rows,) + tuple(params[0].shape[1:])
            buf = ent[0] if (ent is not None and ent[0].shape == shape and ent[3] == params[0].data_ptr()) \
                else torch.empty(shape, dtype=dtype, device=params[0].device)
            lib, r0 = _lib(), 0
            for p in params:
                pd = p.detach()
                if not pd.is_contiguous():
                    pd = pd.contiguous()
                dst = buf[r0:r0 + p.shape[0]]
                L.check(lib.evlm_cast(L.dt(pd), L.ptr(pd), L.dt(dst), L.ptr(dst), pd.numel(), L.stream()), "cast")
                r0 += p.shape[0]
        import weakref
        self._store[key] = (buf, vers, self.epoch, params[0].data_ptr(), tuple(weakref.ref(p) for p in params))
        return buf


CACHE = WeightCache()
# set to a list to time every GEMM launch with HIP events on the launch stream (bench.py roofline leg)
GEMM_PROFILE = None
# set to [0.0] to count the FLOPs of the attention cores that are launched (QK^T + PV forward, 5 products backward)
ATTN_FLOPS = None
# A/B switch: no bf16 W^T copies - dX = dY W reads W reduction-major through the transposing LDS reads (round 4: those reads
# no longer drain the staging pipeline, see gemm_pp256_core.h)
NO_WT = bool(os.environ.get("EVLM_NO_WT"))
ATTN_STORE_P = bool(os.environ.get("EVLM_ATTN_STORE_P"))   # A/B switch: attention backward from the stored bf16 map (round-2 form)
# long key sequences (417..928) recompute too when nobody wants the map (round 4: one pass); "0": stored map (round 3)
ATTN_RC_LONG = os.environ.get("EVLM_ATTN_RC_LONG", "1") not in ("", "0")


def _as2d(x):
    """view x [..., K] as rows x K with a uniform row stride; returns (tensor, rows, K, ld)."""
    K = x.shape[-1]
    if x.stride(-1) != 1:
        x = x.contiguous()
    if x.dim() == 2:
        if x.stride(0) % 8 != 0 and x.shape[0] > 1:
            x = x.contiguous()
        return x, x.shape[0], K, (x.stride(0) if x.shape[0] > 1 else K)
    if not x.is_contiguous():
        # collapsible leading dims?  (e.g. x[:, 0, :] of a contiguous [B, L, d])
        x2 = x.reshape(-1, K) if _collapsible(x) else x.contiguous().view(-1, K)
        return _as2d(x2)
    return x, x.numel() // K, K, K


def _collapsible(x):
    try:
        v = x.view(-1, x.shape[-1])
        return v.data_ptr() == x.data_ptr()
    except RuntimeError:
        return False


SK_WORKSPACE_BYTES = 4096 + 256 * 262144        # EVLM_GEMM_SK_WORKSPACE_BYTES (include/evlm_hip.h)
_SK_WS = {}
_SK_ON = os.environ.get("EVLM_PP256_SK", "0") not in ("", "0")     # stream-K GEMM launches: opt-in (gemm_pp256.hip)


def _sk_workspace(dev):
    """the stream-K workspace of the CURRENT stream of `dev` (flags + partial-tile slots, evlm_gemm_args.sk_workspace):
    launches of one stream run one after the other and may share it; launches of different streams may overlap and may
    not.  Zero on first use; the kernels leave the flags zero.  (Inside a hipGraph capture a new stream's workspace comes
    from the graph's pool, its zero-fill replayed with the graph.)"""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _SK_WS.get(key)
    if ws is None:
        ws = _SK_WS[key] = torch.zeros(SK_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
    return ws


def _gemm(dtype, P, Q, Cm, I, J, K, ldp, ldq, ldc, p_trans=0, q_trans=0, bias=None, gate=None, preact=None,
          aux=None, residual=None, ldx=0, alpha=1.0, act=L.ACT_NONE, gate_pos=L.GATE_PRE, dact=L.ACT_NONE, c_f32=0,
          accumulate=0, p_off=0, psum=None, dgate=None, drop=None):
    """drop: DropSlot of a hidden-dropout site - C = (..) .* keep / (1 - p) + residual in the epilogue (evlm_gemm_args.dropout_p)"""
    Pp = L.ptr(P) if not p_off else C.c_void_p(P.data_ptr() + p_off * P.element_size())
    a = L.GemmArgs(dtype=dtype, c_f32=c_f32, p_trans=p_trans, q_trans=q_trans, I=I, J=J, K=K, ldp=ldp, ldq=ldq,
                   ldc=ldc, ldx=ldx, P=Pp, Q=L.ptr(Q), C=L.ptr(Cm), bias=L.ptr(bias), gate=L.ptr(gate),
                   preact=L.ptr(preact), aux=L.ptr(aux), residual=L.ptr(residual), alpha=alpha, act=act,
                   gate_pos=gate_pos, dact=dact, accumulate=accumulate, psum=L.ptr(psum), dgate=L.ptr(dgate))
    if drop is not None:
        a.dropout_p, a.rng_state, a.call_id = drop.p, L.ptr(drop.state), drop.call
    if _SK_ON and dgate is None and dtype == L.BF16 and not c_f32 and not p_trans and K >= 512 and I * J >= 16 * 65536:
        a.sk_workspace = L.ptr(_sk_workspace(P.device))     # (shapes stream-K can apply to: gemm_pp256.hip)
    if GEMM_PROFILE is None:
        L.check(_lib().evlm_gemm(C.byref(a), L.stream()), "gemm")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(_lib().evlm_gemm(C.byref(a), L.stream()), "gemm")
    e1.record()
    GEMM_PROFILE.append((dtype, p_trans, q_trans, I, J, K, e0, e1, _lib().evlm_gemm_last_kernel().decode()))


def _colsum(x2d, I, J, ld, out=None, x_off=0):
    """out[j] += sum_i x[i, x_off + j]; a fresh zero vector when `out` is None"""
    if out is None:
        out = zeros_small(J, torch.float32, x2d.device)
    xp = L.ptr(x2d) if not x_off else C.c_void_p(x2d.data_ptr() + x_off * x2d.element_size())
    L.check(_lib().evlm_colsum(L.dt(x2d), xp, I, J, ld, L.ptr(out), L.stream()), "colsum")
    return out


# When True (set by the trainer around backward), parameter gradients are ACCUMULATED straight into `param.grad`
# (the optimiser's flat fp32 slabs) by the kernels themselves - f32 atomics from the dW GEMM, the bias column sums, the
# LayerNorm and embedding backward kernels - and the autograd Functions return None for those inputs.  This removes the
# ~500 per-step `grad += dW` element-wise launches and the dW temporaries.  Off by default (plain autograd semantics).
WGRAD_INPLACE = False


# Deferred weight gradients (trainer mode, with WGRAD_INPLACE): set to a list and the in-place dW products of Linear /
# MLP backward are queued instead of launched; flush_wgrad() then runs every queued product of one reduction length as
# ONE grouped launch of the 256x256 kernel, each output tile owned by one workgroup - no split-K and none of its f32
# atomics.  The queue keeps dY and X alive until the flush (MI355X: 288 GB of HBM make that free).
WGRAD_DEFER = None
WGRAD_DEFER_MIN_K = 1024


LN_DEFER = []          # (workspace, blocks, d, dgamma, dbeta) of LayerNorm backwards whose column sums wait for flush_wgrad


def _flush_ln():
    """the queued LayerNorm-backward workspaces -> dgamma / dbeta, ONE launch (evlm_layernorm_bwd_reduce_grouped)"""
    global LN_DEFER
    q, LN_DEFER = LN_DEFER, []
    if not q:
        return
    rows = []
    for ws, nblk, d, dg, db in q:
        rows += [ws.data_ptr(), nblk, d, dg.data_ptr(), db.data_ptr()]
    table = _upload_table(rows, q[0][0].device)
    L.check(_lib().evlm_layernorm_bwd_reduce_grouped(L.ptr(table), len(q), max(r[2] for r in q), L.stream()), "ln_reduce_grouped")


# First-touch ASSIGNMENT of weight gradients (set by a trainer together with WGRAD_DEFER): {"skip": {grad data_ptr: grad view}
# of the Linear weights whose slab ranges the step's zero-fill leaves out, "done": set of data_ptrs initialised this step}.
# The first grouped product of such a weight WRITES its tile (C = dY^T X: no fill before, no read of C in the kernel -
# ~370 MB less of each per GD step); any other first contribution zero-fills the range itself, and finish_assign() zeroes
# what the step never touched (heads a batch kind does not use).
WGRAD_ASSIGN = None


def _assign_prepare(params, will_assign):
    """before a contribution to `params`' gradients: returns True when it may ASSIGN (every member still untouched and
    eligible).  Otherwise members that are eligible and untouched are zero-filled here, and a queued-but-unflushed
    assigning product of a member is demoted to an accumulating one (its range zero-filled now): an assignment launched
    at the flush would wipe out what this contribution is about to add."""
    st = WGRAD_ASSIGN
    if st is None:
        return False
    keys = [p.grad.data_ptr() if p.grad is not None else None for p in params]
    fresh = [k is not None and k in st["skip"] and k not in st["done"] for k in keys]
    if will_assign and all(fresh):
        st["done"].update(keys)
        return True
    for k, f in zip(keys, fresh):
        if f:
            st["skip"][k].zero_()
            st["done"].add(k)
            # a weight whose first contribution is not an assigning product (the small heads: their dW is not one of the
            # grouped products) would be zero-filled HERE, by a launch of its own, every step: from the next step on it
            # belongs to the ranges the step's ONE grouped fill zeroes (finish_assign moves it)
            st.setdefault("demote", set()).add(k)
        elif k is not None and k in st["pending"]:
            rec = st["pending"].pop(k)
            if rec[10]:
                rec[10] = False
                for kk in rec[11]:
                    st["skip"][kk].zero_()
                    st["pending"].pop(kk, None)
    return False


def assign_settle(ranges):
    """before gradient ranges leave for a data-parallel exchange in the middle of backward: eligible gradients inside them
    that nothing has written yet (a head this batch kind does not use) are zero-filled NOW - finish_assign() at the end of
    backward would race with the exchange"""
    st = WGRAD_ASSIGN
    if st is None:
        return
    spans = [(r.data_ptr(), r.data_ptr() + r.numel() * r.element_size()) for r in ranges]
    for k, g in st["skip"].items():
        if k not in st["done"] and any(lo <= k < hi for lo, hi in spans):
            g.zero_()
            st["done"].add(k)


def finish_assign():
    """end of a step's backward: zero the eligible gradients nothing has written (a head this batch kind does not use)"""
    st = WGRAD_ASSIGN
    if st is None:
        return
    _keep_table(zero_grouped([g for k, g in st["skip"].items() if k not in st["done"]]))
    st["done"].clear()
    st["pending"].clear()
    if st.get("demote") and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        fill = st.get("fill")                    # (the optimiser's list of ranges its zero_grad() fills; absent in bare states)
        for k in st["demote"] if fill is not None else ():
            g = st["skip"].pop(k, None)
            if g is not None:
                fill.append(g)
        st["demote"].clear()


def flush_wgrad():
    """launch the queued weight-gradient products and LayerNorm column-sum reductions (no-op when nothing is queued)"""
    global WGRAD_DEFER
    _flush_ln()
    q = WGRAD_DEFER
    if not q:
        return
    WGRAD_DEFER = [] if q is not None else None
    if WGRAD_ASSIGN is not None:
        WGRAD_ASSIGN["pending"].clear()          # (the queued assignments are launched below, in queue order per K)
    by_k = {}
    for rec in q:
        by_k.setdefault(rec[0], []).append(rec)
    lib = _lib()
    for K, recs in by_k.items():
        # large problems first: the persistent workgroups take items in order
        recs.sort(key=lambda r: -(r[5] * r[6]))
        arr = (L.WgradProblem * len(recs))()
        ptrs = [rec[4].data_ptr() for rec in recs]
        for rec in recs:                       # two contributions to one C in ONE launch run on atomics: no assignment then
            if rec[10] and ptrs.count(rec[4].data_ptr()) > 1:
                rec[10] = False
                for kk in rec[11]:
                    WGRAD_ASSIGN["skip"][kk].zero_()
        for k, (_, d2, p_off, x2, cgrad, I, J, ldd, ldp, ps, assign, _keys) in enumerate(recs):
            arr[k].P = d2.data_ptr() + p_off * d2.element_size()
            arr[k].Q = x2.data_ptr()
            arr[k].C = cgrad.data_ptr()
            arr[k].psum = ps.data_ptr() if ps is not None else None
            arr[k].I, arr[k].J, arr[k].ldp, arr[k].ldq, arr[k].ldc = I, J, ldd, ldp, J
            arr[k].assign = int(bool(assign))
        if GEMM_PROFILE is None:
            L.check(lib.evlm_wgrad_grouped(arr, len(recs), K, L.stream()), "wgrad_grouped")
            continue
        # profiled as ONE record: the launch computes sum_n 2 I_n J_n K flops, reported as a product with I = sum of
        # I_n J_n / J_0 so that 2 I J K is exact
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.evlm_wgrad_grouped(arr, len(recs), K, L.stream()), "wgrad_grouped")
        e1.record()
        ij = sum(r[5] * r[6] for r in recs)
        GEMM_PROFILE.append((L.BF16, 1, 1, ij, 1, K, e0, e1, "gemm_bf16_pp256_grouped_kernel"))


def _inplace(p):
    g = getattr(p, "grad", None)
    return WGRAD_INPLACE and g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape


def _wgrad(dtype, d2, ldd, x2, ldp, M, K, params, rows, biases=None):
    """dW_i = dY[:, rows_i]^T X for every packed weight (and, when `biases` is given, db_i = column sums of dY riding
    on the same GEMM).  Returns (weight grads, bias grads); entries are None where accumulated in place."""
    gw, gb, r0 = [], [], 0
    if dtype == L.BF16 and M % 64 != 0 and M > 64 and all(_inplace(w) for w in params):
        # a reduction length that is not a multiple of the 64-row K tile (B x 901 image tokens at 480x480, odd batches):
        # the first floor64(M) rows take the MFMA path (in place, grouped), the <= 63 remaining rows are zero-padded to
        # one K tile and accumulated by a second, tiny product - instead of the whole product falling to the generic kernel
        M0 = M - M % 64
        gw, gb = _wgrad(dtype, d2, ldd, x2, ldp, M0, K, params, rows, biases)
        N = sum(rows)
        # (leading dimensions padded to 16 bytes: N may be ragged - the 30 522-wide vocabulary head of the VQA decoder)
        # The two 64-row operands are PERSISTENT per (stream, width, tail length): zero-filled once - the rows behind the
        # tail are never written - and refilled by ONE grouped copy, instead of two fills + two copies per product (the
        # VQA step at 480 x 480 has ~76 such products: 32 x 901 rows = 450.5 K tiles)
        nt = M - M0
        sid = torch.cuda.current_stream(d2.device).cuda_stream if d2.is_cuda else 0
        td = _scratch(("wtail_d", sid, _pad8(N), nt), (64, _pad8(N)), d2.dtype, d2.device, zero=True)
        tx = _scratch(("wtail_x", sid, _pad8(K), nt), (64, _pad8(K)), x2.dtype, x2.device, zero=True)
        tail = lambda t, ld, w: torch.as_strided(t, (nt, w), (ld, 1), t.storage_offset() + M0 * ld)
        sd, sx = tail(d2, ldd, N), tail(x2, ldp, K)
        if (d2.is_cuda and ldd == N and ldp == K and N % 8 == 0 and K % 8 == 0 and sd.data_ptr() % 16 == 0
                and sx.data_ptr() % 16 == 0 and (nt * N * d2.element_size()) % 16 == 0 and (nt * K * x2.element_size()) % 16 == 0):
            copy_grouped([(sd, td[:nt]), (sx, tx[:nt])])
        else:
            td[:nt, :N].copy_(sd)
            tx[:nt, :K].copy_(sx)
        _, gb_t = _wgrad(dtype, td, td.stride(0), tx, tx.stride(0), 64, K, params, rows, biases)
        gb = [a if b is None else (b if a is None else a + b) for a, b in zip(gb, gb_t)]
        return gw, gb
    fast = dtype == L.BF16 and M % 64 == 0
    has_b = biases is not None and len(biases) and biases[0] is not None
    if not (fast and all(_inplace(w) for w in params)):
        _assign_prepare(params, False)            # (autograd will ADD these into .grad: eligible ranges zero-filled first)
        N = sum(rows)
        dW = torch.empty((N, K), dtype=torch.float32, device=x2.device)
        db = zeros_small(N, torch.float32, x2.device) if (has_b and fast) else None
        _gemm(dtype, d2, x2, dW, N, K, M, ldd, ldp, K, p_trans=1, q_trans=1, c_f32=1, psum=db)
        for i, r in enumerate(rows):
            gw.append(dW[r0:r0 + r])
            if has_b:
                if db is not None:
                    gb.append(db[r0:r0 + r])
                else:
                    gb.append(_colsum(d2, M, r, ldd, x_off=r0))
            r0 += r
        return gw, (gb if has_b else [None] * len(rows))
    # packed members whose gradient views are back to back in the optimiser slab (q | k | v) take ONE product: a third
    # of the launches and of the split-K atomics
    def adjacent(t0, t1):
        return t0.data_ptr() + t0.numel() * t0.element_size() == t1.data_ptr()
    i = 0
    while i < len(params):
        j = i + 1
        bi_in = has_b and biases[i] is not None and _inplace(biases[i])
        while j < len(params) and adjacent(params[j - 1].grad, params[j].grad) and (
                not has_b or (bi_in and biases[j] is not None and _inplace(biases[j]) and adjacent(biases[j - 1].grad, biases[j].grad))):
            j += 1
        r = sum(rows[i:j])
        b = biases[i] if has_b else None
        if b is not None and _inplace(b):
            ps, res = b.grad, None
        elif b is not None:
            ps = res = zeros_small(r, torch.float32, x2.device)
        else:
            ps = res = None
        if (WGRAD_DEFER is not None and res is None and M >= WGRAD_DEFER_MIN_K and r % 8 == 0 and K % 8 == 0
                and (r0 * d2.element_size()) % 16 == 0 and M * max(ldd, ldp) < (1 << 31)):
            assign = _assign_prepare(params[i:j], True)
            keys = [w.grad.data_ptr() for w in params[i:j]] if assign else []
            rec = [M, d2, r0, x2, params[i].grad, r, K, ldd, ldp, ps, assign, keys]
            for kk in keys:
                WGRAD_ASSIGN["pending"][kk] = rec
            WGRAD_DEFER.append(rec)
        else:
            _assign_prepare(params[i:j], False)
            _gemm(dtype, d2, x2, params[i].grad, r, K, M, ldd, ldp, K, p_trans=1, q_trans=1, c_f32=1, accumulate=1, p_off=r0, psum=ps)
        for k in range(i, j):
            gw.append(None)
            gb.append(res if k == i else None)
        if res is not None and j - i > 1:       # (only reachable when has_b is False for the merged members)
            raise AssertionError("merged members must have in-place bias gradients")
        r0 += r
        i = j
    return gw, gb


def _bgrad(d2, M, ldd, params, rows):
    out, r0 = [], 0
    for b, r in zip(params, rows):
        if _inplace(b):
            _colsum(d2, M, r, ldd, out=b.grad, x_off=r0)
            out.append(None)
        else:
            out.append(_colsum(d2, M, r, ldd, x_off=r0))
        r0 += r
    return out


def _pad8(n):
    return (n + 7) // 8 * 8


# Loss kernels accumulate into f32 device words that must start at zero.  A trainer calls begin_step() once per step: ONE
# fill of a small block, the ~30 loss scalars of a step are then views of it (instead of ~30 one-element fill launches).
_ZERO_BLOCK = [None, 0]
_ZERO_ARENA = [None, 0]        # bytes: the small zero-initialised work buffers of a step (zeros_small)
ZERO_ARENA_BYTES = 2 << 20
ZERO_SMALL_MAX = 256 << 10


def begin_step(device):
    """fresh zeroed block of loss accumulators for this step (capturable: a plain allocation + fill) and the arena the
    step's small zero-initialised buffers are carved from - ONE fill launch instead of one per buffer"""
    arena = torch.zeros(1024 + ZERO_ARENA_BYTES, dtype=torch.uint8, device=device)
    _ZERO_BLOCK[0] = _alias(arena, 0, (256,), torch.float32)
    _ZERO_BLOCK[1] = 0
    _ZERO_ARENA[0], _ZERO_ARENA[1] = arena, 1024


def end_step():
    _ZERO_BLOCK[0] = None
    _ZERO_ARENA[0] = None


def zero_scalar(device):
    blk, used = _ZERO_BLOCK
    if blk is None or used >= blk.numel() or blk.device != torch.device(device):
        return torch.zeros((), dtype=torch.float32, device=device)
    _ZERO_BLOCK[1] = used + 1
    return blk[used]


def _alias(arena, byte_off, shape, dtype):
    """a tensor OVER a byte range of `arena`'s storage that autograd does not know as a view of it (own version counter, no
    base): kernels and in-place ATen ops on one such tensor say nothing about the others"""
    t = torch.empty(0, dtype=dtype, device=arena.device)
    return t.set_(arena.untyped_storage(), byte_off // t.element_size(), tuple(shape))


def zeros_small(shape, dtype, device):
    """torch.zeros(shape) for a buffer a kernel is about to write into / accumulate into: inside a trainer's step small ones
    are views of the step's pre-zeroed arena (no launch); anything else is a plain torch.zeros"""
    arena, used = _ZERO_ARENA
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    n = 1
    for k in shape:
        n *= k
    nb = n * torch.empty((), dtype=dtype).element_size()
    if (arena is None or nb == 0 or nb > ZERO_SMALL_MAX or arena.device != torch.device(device)
            or used + nb > arena.numel()):
        return torch.zeros(shape, dtype=dtype, device=device)
    _ZERO_ARENA[1] = (used + nb + 255) // 256 * 256
    return _alias(arena, used, shape, dtype)


_CONST = {}


def const_tensor(kind, n, device):
    """small index / label tensors that depend on a batch size only - built once per (kind, n, device) instead of two or three
    launches per forward: "arange" = 0..n-1 (int64), "itm_labels" = n ones then 2n zeros (int64: positives, hard negatives), "mask_table" = (-10000, -0) for additive_mask.
    Read-only by contract.  (Not cached when first asked for inside a hipGraph capture: the tensor would live in that
    graph's pool.)"""
    key = (kind, int(n), str(device))
    t = _CONST.get(key)
    if t is None:
        if kind == "arange":
            t = torch.arange(n, device=device)
        elif kind == "mask_table":        # additive_mask: (1 - m) * -10000 for m = 0, 1 (-0.0 for m = 1, as the product gives)
            t = torch.tensor([-10000.0, -0.0], dtype=torch.float32, device=device)
        elif kind == "itm_labels":
            t = torch.cat([torch.ones(n, dtype=torch.long, device=device), torch.zeros(2 * n, dtype=torch.long, device=device)])
        else:
            raise ValueError(kind)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _CONST[key] = t
    return t


def const_ones(shape, dtype, device):
    """torch.ones(shape) for a READ-ONLY all-ones tensor (the image encoder's attention mask of every forward): built once
    per (shape, dtype, device), no fill launch per forward.  Never write to it."""
    key = ("ones", tuple(shape), dtype, str(device))
    t = _CONST.get(key)
    if t is None:
        t = torch.ones(tuple(shape), dtype=dtype, device=device)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):      # (a capture's allocations live in its graph's pool)
            _CONST[key] = t
    return t


def additive_mask(mask):
    """(1.0 - mask.float()) * -10000.0 (eff_bert.py:953-1013, transformers' invert_attention_mask): ONE gather launch for a
    0/1 integer mask on the GPU instead of cast + rsub + mul; the same bits (-0.0 where the mask is 1)"""
    if mask.is_cuda and mask.dtype in (torch.int64, torch.int32, torch.uint8, torch.bool):
        return const_tensor("mask_table", 2, mask.device)[mask.long()]
    return (1.0 - mask.to(dtype=torch.float32)) * -10000.0


_SCRATCH = {}


def _scratch(key, shape, dtype, device, zero=False):
    """persistent work buffer (static address: safe under hipGraph capture); zero-filled once when `zero`"""
    k = (key, dtype, str(device))
    t = _SCRATCH.get(k)
    if t is None:
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        _SCRATCH[k] = t
    return t


# ---------------------------------------------------------------------------------------------------
# Linear (optionally several weights packed along the output dim) with fused epilogue
# ---------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual);  W = rows of `weights` packed (fused QKV / KV projections)."""

    @staticmethod
    def forward(ctx, x, residual, act, out_f32, nw, *wb):
        # (nw may be (nw, DropSlot): hidden dropout between the product and the residual - y = drop(x W^T + b) + residual)
        nw, drop = nw if isinstance(nw, tuple) else (nw, None)
        weights, biases = wb[:nw], wb[nw:]
        L.require_cuda(x, *weights)
        x2, M, K, ldp = _as2d(x)
        dtype = L.dt(x2)
        W = CACHE.get(weights, x2.dtype)
        b = CACHE.get(biases, torch.float32) if biases and biases[0] is not None else None
        N = W.shape[0]
        ydt = torch.float32 if out_f32 else x2.dtype
        ldc = _pad8(N)
        if ldc == N:
            ybuf = torch.empty((M, ldc), dtype=ydt, device=x.device)
        elif M * ldc * (4 if out_f32 else x2.element_size()) <= ZERO_SMALL_MAX:
            ybuf = zeros_small((M, ldc), ydt, x.device)
        else:
            # (the vocabulary head: [512, 30 528] for 30 522 classes - only the padding columns need the zeros, not 31 MB)
            ybuf = torch.empty((M, ldc), dtype=ydt, device=x.device)
            ybuf[:, N:].zero_()
        need_grad = any(ctx.needs_input_grad)
        preact = None
        if act != L.ACT_NONE and need_grad:
            preact = torch.empty((M, ldc), dtype=x2.dtype, device=x.device)
        r2 = None
        if residual is not None:
            r2, rm, rn, ldr = _as2d(residual)
            if ldr != ldc or r2.dtype != x2.dtype:
                raise RuntimeError("residual must be contiguous [M, N] of the activation dtype")
        if drop is not None and (r2 is None or act != L.ACT_NONE or out_f32 or ldc != N or N % 8 != 0):
            raise RuntimeError("linear(dropout=): needs a residual, no activation, an output of the activation dtype and N % 8 == 0")
        _gemm(dtype, x2, W, ybuf, M, N, K, ldp, K, ldc, bias=b, preact=preact, residual=r2, ldx=ldc, act=act,
              c_f32=1 if out_f32 else 0, drop=drop)
        ctx.drop_slot = drop
        ctx.save_for_backward(x2, W, preact)
        ctx.params = (weights, biases)
        ctx.meta = (M, N, K, ldp, ldc, act, nw, tuple(w.shape[0] for w in weights), biases and biases[0] is not None,
                    residual is not None, x.shape, out_f32)
        y = ybuf[:, :N] if ldc != N else ybuf
        return y.unflatten(0, x.shape[:-1]) if x.dim() > 2 else y

    @staticmethod
    def backward(ctx, dy):
        return _linear_backward(ctx, dy, None)


def _linear_backward(ctx, dy, dx_add):
    """backward of _Linear; dx_add (optional, [.., K]): a gradient that reaches x along another branch (the residual that
    bypasses this projection) - it rides on the dX GEMM's residual epilogue instead of a separate element-wise add"""
    if True:
        x2, W, preact = ctx.saved_tensors
        M, N, K, ldp, ldc, act, nw, rows, has_bias, has_res, xshape, out_f32 = ctx.meta
        dtype = L.dt(x2)
        lib = _lib()
        d2 = dy.reshape(M, N)
        if getattr(ctx, "drop_slot", None) is not None:      # the product's gradient is dy .* M (dy itself goes to the residual)
            d2 = ctx.drop_slot.masked_grad(dy).reshape(M, N)
        if d2.dtype != x2.dtype:          # f32-output heads only: bring the gradient to the compute dtype
            d2 = cast(d2.contiguous(), x2.dtype)
        if d2.stride(1) != 1 or d2.stride(0) % 8 != 0 or d2.stride(0) < _pad8(N):
            buf = zeros_small((M, _pad8(N)), x2.dtype, x2.device)
            buf[:, :N].copy_(d2)
            d2 = buf
        ldd = d2.stride(0)
        dres = dy if has_res else None
        if act != L.ACT_NONE:
            dh = torch.empty_like(preact) if ldd == ldc else torch.zeros_like(preact)
            if ldd != ldc:
                raise RuntimeError("internal: activation backward needs matching leading dimensions")
            L.check(lib.evlm_gated_act_bwd(dtype, L.ptr(d2), L.ptr(preact), None, M, _pad8(N), ldc, act, L.GATE_POST,
                                           L.ptr(dh), None, L.stream()), "act_bwd")
            d2, ldd = dh, ldc
        dx = None
        if ctx.needs_input_grad[0]:
            n64 = (N + 63) // 64 * 64
            if dtype == L.BF16 and N % 64 != 0 and N >= 4096 and M <= 2048 and ldd >= n64 and act == L.ACT_NONE:
                # skinny product with a vocabulary-sized ragged reduction (MLM decoder): zero-pad W's rows to a multiple
                # of 64 (dY's padding columns are zero already) so the split-K f32 path applies, then cast once
                # When W is the bf16 mirror of a slab-backed parameter the slab layout keeps n64 - N ZERO rows behind the
                # matrix (optim._seg: zero gradient and moments, so AdamW leaves them at zero) and the product reads the slab
                # in place - no 47 MB copy of the vocabulary matrix per step.  Anything else (a weight outside a slab) is
                # copied into a zero-padded scratch: the rows meet dY's zero columns, but 0 x non-finite is NaN
                Wp = None
                if CACHE.zero_padded_rows(W, n64):
                    Wp = torch.as_strided(W, (n64, K), (K, 1))
                if Wp is None:
                    Wp = _scratch(("wpad", n64, K), (n64, K), x2.dtype, x2.device, zero=True)
                    Wp[:N].copy_(W)
                d32 = torch.empty((M, K), dtype=torch.float32, device=x2.device)      # (a split reduction zero-fills its output itself)
                _gemm(dtype, d2, Wp, d32, M, K, n64, ldd, K, K, p_trans=0, q_trans=1, c_f32=1)
                dxb = cast(d32, x2.dtype)
            else:
                dxb = torch.empty((M, K), dtype=x2.dtype, device=x2.device)
                Wt = CACHE.get_t(ctx.params[0]) if (dtype == L.BF16 and N % 64 == 0) else None
                rkw = {}
                if dx_add is not None:
                    r2 = dx_add.reshape(M, K)
                    if not r2.is_contiguous() or r2.dtype != x2.dtype:
                        r2 = r2.to(x2.dtype).contiguous()
                    rkw, dx_add = dict(residual=r2, ldx=K), None
                if Wt is not None:      # dX = dY (W^T)^T: both operands K-contiguous, like the forward product
                    _gemm(dtype, d2, Wt, dxb, M, K, N, ldd, N, K, p_trans=0, q_trans=0, **rkw)
                else:
                    _gemm(dtype, d2, W, dxb, M, K, N, ldd, K, K, p_trans=0, q_trans=1, **rkw)     # dX = dY W
            dx = dxb.view(xshape)
            if dx_add is not None:          # (the padded-vocabulary branch above has no epilogue to ride on)
                dx = dx + dx_add.view(xshape)
        weights, biases = ctx.params
        gw, gb = _wgrad(dtype, d2, ldd, x2, ldp, M, K, weights, rows, biases if has_bias else None)   # dW = dY^T X, db
        return (dx, dres, None, None, None, *gw, *gb)


class _LinearFork(torch.autograd.Function):
    """(linear(x), alias of x): the alias feeds the residual branch that bypasses this projection (post-LN BERT blocks:
    a = LayerNorm(x + dense(attention(x)))), so autograd sees one use of x and the gradient of the alias is added in the
    dX GEMM's epilogue (no element-wise add over [rows, d] per block)."""

    @staticmethod
    def forward(ctx, x, residual, act, out_f32, nw, *wb):
        y = _Linear.forward(ctx, x, residual, act, out_f32, nw, *wb)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dxa):
        if dy is None:
            return (dxa,) + (None,) * (4 + 2 * ctx.meta[6])
        return _linear_backward(ctx, dy, dxa)


def linear(x, weight, bias=None, act=L.ACT_NONE, residual=None, out_f32=False, dropout_p=0.0):
    """dropout_p > 0 (training): y = dropout(x W^T + b) + residual with the keep-mask applied in the GEMM's residual epilogue
    (BertSelfOutput / BertOutput, eff_bert.py:372-381,456-462); the result carries its DropSlot so that the LayerNorm that
    consumes it writes the masked gradient beside its own (layer_norm)"""
    if dropout_p and dropout_p > 0.0:
        slot = DropSlot(dropout_p, x.device, tuple(x.shape[:-1]) + (weight.shape[0],))
        y = _Linear.apply(x, residual, act, out_f32, (1, slot), weight, bias)
        y._evlm_drop_slot = slot
        return y
    return _Linear.apply(x, residual, act, out_f32, 1, weight, bias)


def linear_packed(x, weights, biases):
    """one GEMM for several Linear layers sharing the input: returns [.., sum N_i]"""
    return _Linear.apply(x, None, L.ACT_NONE, False, len(weights), *weights, *biases)


def linear_fork(x, weights, biases):
    """(packed linear(x), alias of x for a residual branch that bypasses it): see _LinearFork"""
    weights, biases = tuple(weights), tuple(biases)
    if not (torch.is_grad_enabled() and x.requires_grad) or _NO_FORK:
        return _Linear.apply(x, None, L.ACT_NONE, False, len(weights), *weights, *biases), x
    return _LinearFork.apply(x, None, L.ACT_NONE, False, len(weights), *weights, *biases)


# ---------------------------------------------------------------------------------------------------
# MLP block: y = act_gate(x W1^T + b1) W2^T + b2 + residual     (CLIPMLP / BertIntermediate+BertOutput.dense)
# ---------------------------------------------------------------------------------------------------
class _MLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gate, residual, act, gate_pos, drop=None):
        L.require_cuda(x, w1, w2)
        x2, M, K, ldp = _as2d(x)
        dtype = L.dt(x2)
        W1, W2 = CACHE.get((w1,), x2.dtype), CACHE.get((w2,), x2.dtype)
        Fh, N = W1.shape[0], W2.shape[0]
        need_grad = any(ctx.needs_input_grad)
        h = torch.empty((M, Fh), dtype=x2.dtype, device=x.device) if need_grad else None
        a = torch.empty((M, Fh), dtype=x2.dtype, device=x.device)
        g32 = None
        if gate is not None:
            g32 = gate.detach().reshape(-1).to(torch.float32).contiguous()
        _gemm(dtype, x2, W1, a, M, Fh, K, ldp, K, Fh, bias=b1.detach(), gate=g32, preact=h, ldx=Fh, act=act, gate_pos=gate_pos)
        y = torch.empty((M, N), dtype=x2.dtype, device=x.device)
        r2 = None
        if residual is not None:
            r2 = residual.reshape(M, N)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        if drop is not None and (r2 is None or N % 8 != 0):
            raise RuntimeError("mlp(dropout=): needs the residual and N % 8 == 0")
        _gemm(dtype, a, W2, y, M, N, Fh, Fh, Fh, N, bias=b2.detach(), residual=r2, ldx=N, drop=drop)
        ctx.drop_slot = drop
        ctx.save_for_backward(x2, W1, W2, h, a, g32)
        ctx.params = (w1, b1, w2, b2)
        # residual IS the input (post-LN BERT FFN: LayerNorm(x + FFN(x))): the backward then adds dy in the dX GEMM's
        # epilogue and returns ONE gradient for x, instead of autograd summing the two with an element-wise add
        ctx.res_is_x = residual is x and N == K and not _NO_FORK
        ctx.meta = (M, K, Fh, N, ldp, act, gate_pos, x.shape, gate.shape if gate is not None else None,
                    residual is not None)
        ctx.gate_slot = getattr(gate, "_evlm_gate_slot", None) if gate is not None else None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W1, W2, h, a, g32 = ctx.saved_tensors
        M, K, Fh, N, ldp, act, gate_pos, xshape, gshape, has_res = ctx.meta
        dtype = L.dt(x2)
        lib = _lib()
        d2 = dy.reshape(M, N)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        d_res = d2                                 # what the residual branch receives: dy itself
        if ctx.drop_slot is not None:              # ... while the second product's gradient is dy .* M
            d2 = ctx.drop_slot.masked_grad(dy).reshape(M, N)
            if not d2.is_contiguous():
                d2 = d2.contiguous()
        dev = x2.device
        dh = torch.empty((M, Fh), dtype=x2.dtype, device=dev)
        dgate = None
        bf = dtype == L.BF16
        # W^T copies (kept current by the optimiser): the input-gradient products then read K-contiguous operands
        W2t = CACHE.get_t((ctx.params[2],)) if (bf and N % 64 == 0) else None
        W1t = CACHE.get_t((ctx.params[0],)) if (bf and Fh % 64 == 0) else None
        q2 = dict(p_trans=0, q_trans=0) if W2t is not None else dict(p_trans=0, q_trans=1)
        Q2, ld2 = (W2t, N) if W2t is not None else (W2, Fh)
        if g32 is None:
            # dH = (dY W2) .* act'(h), fused into the GEMM epilogue
            _gemm(dtype, d2, Q2, dh, M, Fh, N, N, ld2, Fh, aux=h, ldx=Fh, dact=act, **q2)
        else:
            gslot = ctx.gate_slot if (ctx.gate_slot is not None and ctx.gate_slot[0].row(ctx.gate_slot[1]).numel() == Fh) else None
            dg = gslot[0].row(gslot[1]) if gslot is not None else torch.zeros(Fh, dtype=torch.float32, device=dev)
            if bf and N % 64 == 0 and N >= 128 and Fh % 8 == 0 and not _NO_GATED_DACT_FOLD:
                # round 5: dH = (dY W2) act'(.) z AND the gate gradient in the dX product's epilogue (evlm_gemm_args.dgate) -
                # no dA round trip through HBM, no second pass over [rows, ffn]
                _gemm(dtype, d2, Q2, dh, M, Fh, N, N, ld2, Fh, aux=h, ldx=Fh, dact=act, gate=g32, gate_pos=gate_pos, dgate=dg, **q2)
            else:
                da = torch.empty((M, Fh), dtype=x2.dtype, device=dev)
                _gemm(dtype, d2, Q2, da, M, Fh, N, N, ld2, Fh, **q2)
                L.check(lib.evlm_gated_act_bwd(dtype, L.ptr(da), L.ptr(h), L.ptr(g32), M, Fh, Fh, act, gate_pos, L.ptr(dh),
                                               L.ptr(dg), L.stream()), "gated_act_bwd")
            dgate = gslot[0].take(gslot[1], gshape) if gslot is not None else dg.view(gshape)
        w1, b1, w2, b2 = ctx.params
        (dW2,), (db2,) = _wgrad(dtype, d2, N, a, Fh, M, Fh, (w2,), (N,), (b2,))
        dx = None
        if ctx.needs_input_grad[0]:
            dxb = torch.empty((M, K), dtype=x2.dtype, device=dev)
            rkw = dict(residual=d_res, ldx=K) if ctx.res_is_x else {}
            if W1t is not None:
                _gemm(dtype, dh, W1t, dxb, M, K, Fh, Fh, Fh, K, p_trans=0, q_trans=0, **rkw)
            else:
                _gemm(dtype, dh, W1, dxb, M, K, Fh, Fh, K, K, p_trans=0, q_trans=1, **rkw)
            dx = dxb.view(xshape)
        (dW1,), (db1,) = _wgrad(dtype, dh, Fh, x2, ldp, M, K, (w1,), (Fh,), (b1,))
        dres = None if (ctx.res_is_x and dx is not None) else (dy if has_res else None)
        return dx, dW1, db1, dW2, db2, dgate, dres, None, None, None


def mlp(x, w1, b1, w2, b2, act, gate=None, gate_pos=L.GATE_PRE, residual=None, dropout_p=0.0):
    """dropout_p > 0 (training): y = dropout(act_gate(x W1^T + b1) W2^T + b2) + residual, the mask in the second product's
    residual epilogue (BertOutput, eff_bert.py:456-462); see linear()"""
    if dropout_p and dropout_p > 0.0:
        slot = DropSlot(dropout_p, x.device, tuple(x.shape[:-1]) + (w2.shape[0],))
        y = _MLP.apply(x, w1, b1, w2, b2, gate, residual, act, gate_pos, slot)
        y._evlm_drop_slot = slot
        return y
    return _MLP.apply(x, w1, b1, w2, b2, gate, residual, act, gate_pos)


# ---------------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, drop_slot=None):
        L.require_cuda(x, gamma)
        ctx.drop_slot = drop_slot
        d = x.shape[-1]
        xc = x if x.is_contiguous() else x.contiguous()
        rows = xc.numel() // d
        y = torch.empty_like(xc)
        need = any(ctx.needs_input_grad)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need else None
        L.check(_lib().evlm_layernorm_fwd(L.dt(xc), L.ptr(xc), L.ptr(gamma.detach()), L.ptr(beta.detach()), eps, rows, d,
                                          L.ptr(y), L.ptr(mean), L.ptr(rstd), L.stream()), "layernorm_fwd")
        ctx.save_for_backward(xc, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, gamma, mean, rstd = ctx.saved_tensors
        d = xc.shape[-1]
        rows = xc.numel() // d
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(xc)
        pg, pb = ctx.params
        inplace = _inplace(pg) and _inplace(pb)
        dg = pg.grad if inplace else torch.zeros(d, dtype=torch.float32, device=xc.device)
        db = pb.grad if inplace else torch.zeros(d, dtype=torch.float32, device=xc.device)
        nblk = _lib().evlm_layernorm_bwd_blocks(rows)
        ws = torch.empty(nblk * 2 * d, dtype=torch.float32, device=xc.device)
        defer = inplace and WGRAD_DEFER is not None          # column sums reduced with the grouped weight gradients
        if defer:
            LN_DEFER.append((ws, nblk, d, dg, db))
        slot = ctx.drop_slot
        if slot is not None and slot.shape == (rows, d):
            # x = dropout(dense(h)) + input: dx goes to `input`, dx .* M to the product - both written by this kernel
            dxm = torch.empty_like(xc)
            L.check(_lib().evlm_layernorm_bwd_drop(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(gamma.detach()), L.ptr(mean), L.ptr(rstd),
                                                   rows, d, L.ptr(dx), L.ptr(dxm), slot.p, L.ptr(slot.state), slot.call,
                                                   None if defer else L.ptr(dg), None if defer else L.ptr(db), L.ptr(ws),
                                                   L.stream()), "layernorm_bwd_drop")
            slot.hand(dx, dxm)
        else:
            L.check(_lib().evlm_layernorm_bwd(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(gamma.detach()), L.ptr(mean), L.ptr(rstd),
                                              rows, d, L.ptr(dx), None if defer else L.ptr(dg), None if defer else L.ptr(db),
                                              L.ptr(ws), L.stream()), "layernorm_bwd")
        return (dx, None, None, None, None) if inplace else (dx, dg, db, None, None)


class _LayerNormFork(torch.autograd.Function):
    """(LayerNorm(x), x [, x]) - the extra outputs are aliases of x: one for the residual branch that bypasses the norm
    (pre-LN blocks: h = x + f(LN(x))) and, with `tap`, one for a distillation term that reads x itself (the hidden-state KD).
    Autograd then sees ONE use of x, and the backward kernel sums the incoming gradients itself (evlm_layernorm_bwd_add, up
    to two addends) instead of autograd issuing an element-wise add over [rows, d] per extra use."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, tap=False):
        y = _LayerNorm.forward(ctx, x, gamma, beta, eps)
        ctx.set_materialize_grads(False)        # an alias nobody differentiates through must not cost a zero tensor
        return (y, x.view_as(x), x.view_as(x)) if tap else (y, x.view_as(x))

    @staticmethod
    def backward(ctx, dy, dres=None, dtap=None):
        xc, gamma, mean, rstd = ctx.saved_tensors
        adds = [g for g in (dres, dtap) if g is not None]
        if dy is None:
            out = None if not adds else (adds[0] if len(adds) == 1 else adds[0] + adds[1])
            return out, None, None, None, None
        d = xc.shape[-1]
        rows = xc.numel() // d
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(xc)
        pg, pb = ctx.params
        inplace = _inplace(pg) and _inplace(pb)
        dg = pg.grad if inplace else torch.zeros(d, dtype=torch.float32, device=xc.device)
        db = pb.grad if inplace else torch.zeros(d, dtype=torch.float32, device=xc.device)
        nblk = _lib().evlm_layernorm_bwd_blocks(rows)
        ws = torch.empty(nblk * 2 * d, dtype=torch.float32, device=xc.device)
        defer = inplace and WGRAD_DEFER is not None          # column sums reduced with the grouped weight gradients
        if defer:
            LN_DEFER.append((ws, nblk, d, dg, db))
        pdg, pdb = (None, None) if defer else (L.ptr(dg), L.ptr(db))
        if adds:
            rcs = [g if (g.is_contiguous() and g.dtype == xc.dtype) else g.to(xc.dtype).contiguous() for g in adds]
            L.check(_lib().evlm_layernorm_bwd_add(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(rcs[0]),
                                                  L.ptr(rcs[1]) if len(rcs) > 1 else None, L.ptr(gamma.detach()), L.ptr(mean),
                                                  L.ptr(rstd), rows, d, L.ptr(dx), pdg, pdb, L.ptr(ws), L.stream()),
                    "layernorm_bwd_add")
        else:
            L.check(_lib().evlm_layernorm_bwd(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(gamma.detach()), L.ptr(mean), L.ptr(rstd),
                                              rows, d, L.ptr(dx), pdg, pdb, L.ptr(ws), L.stream()), "layernorm_bwd")
        return (dx.view(ctx.xshape), None, None, None, None) if inplace else (dx.view(ctx.xshape), dg, db, None, None)


class _LayerNormForkKD(torch.autograd.Function):
    """_LayerNormFork with the hidden-state distillation term of x fused in (evlm_layernorm_fwd_kd / _bwd_kd): returns
    (LayerNorm(x), alias of x for the residual branch, kd_slots).  kd_slots [evlm_layernorm_fwd_kd_slots()] f32, zeroed by
    the caller, receives kd_coef * sum (x - kd_teacher)^2 spread over its cache lines (sum it for the term: weight *
    MSELoss(x, kd_teacher) with kd_coef = weight / numel); its incoming gradient is the term's upstream scalar, and the
    backward kernel adds 2 kd_coef * g * (x - kd_teacher) to dx itself - no squared-difference pass over [rows, d] in either
    direction, no gradient buffer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, kd):
        # (kd = (teacher state, slots, coefficient) travels as ONE non-tensor argument: the slots are written by the kernel
        # and returned as a fresh output, like _Attention's KdSlot word - not an autograd input modified in place)
        kd_teacher, kd_slots, kd_coef = kd
        L.require_cuda(x, gamma, kd_teacher)
        d = x.shape[-1]
        xc = x if x.is_contiguous() else x.contiguous()
        if (not kd_teacher.is_contiguous() or kd_teacher.shape != xc.shape or kd_teacher.dtype != xc.dtype
                or kd_slots.dtype != torch.float32 or kd_slots.numel() < _lib().evlm_layernorm_fwd_kd_slots()):
            raise RuntimeError("fused hidden-state distillation: the teacher state must be a contiguous tensor of the "
                               "student state's shape and dtype, kd_slots a zeroed f32 vector of evlm_layernorm_fwd_kd_slots()")
        rows = xc.numel() // d
        y = torch.empty_like(xc)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(_lib().evlm_layernorm_fwd_kd(L.dt(xc), L.ptr(xc), L.ptr(gamma.detach()), L.ptr(beta.detach()), eps, rows, d,
                                             L.ptr(y), L.ptr(mean), L.ptr(rstd), L.ptr(kd_teacher), L.ptr(kd_slots),
                                             float(kd_coef), L.stream()), "layernorm_fwd_kd")
        ctx.save_for_backward(xc, gamma, mean, rstd, kd_teacher)
        ctx.params = (gamma, beta)
        ctx.xshape = x.shape
        ctx.kd_coef = float(kd_coef)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x), kd_slots

    @staticmethod
    def backward(ctx, dy, dres=None, dkd=None):
        xc, gamma, mean, rstd, kd_t = ctx.saved_tensors
        d = xc.shape[-1]
        rows = xc.numel() // d
        if dy is None:
            raise RuntimeError("fused hidden-state distillation: the LayerNorm output took no part in the loss")
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(xc)
        pg, pb = ctx.params
        inplace = _inplace(pg) and _inplace(pb)
        dg = pg.grad if inplace else torch.zeros(d, dtype=torch.float32, device=xc.device)
        db = pb.grad if inplace else torch.zeros(d, dtype=torch.float32, device=xc.device)
        nblk = _lib().evlm_layernorm_bwd_blocks(rows)
        ws = torch.empty(nblk * 2 * d, dtype=torch.float32, device=xc.device)
        defer = inplace and WGRAD_DEFER is not None
        if defer:
            LN_DEFER.append((ws, nblk, d, dg, db))
        pdg, pdb = (None, None) if defer else (L.ptr(dg), L.ptr(db))
        rc = None
        if dres is not None:
            rc = dres if (dres.is_contiguous() and dres.dtype == xc.dtype) else dres.to(xc.dtype).contiguous()
        if dkd is not None:
            # (every slot's gradient is the term's upstream scalar - the slots are only ever summed: its first word serves)
            g = dkd.reshape(-1)[:1]
            g = g if g.dtype == torch.float32 else g.float()
            L.check(_lib().evlm_layernorm_bwd_kd(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(rc), None, L.ptr(gamma.detach()),
                                                 L.ptr(mean), L.ptr(rstd), rows, d, L.ptr(dx), pdg, pdb, L.ptr(ws),
                                                 L.ptr(kd_t), L.ptr(g), 2.0 * ctx.kd_coef, L.stream()), "layernorm_bwd_kd")
        elif rc is not None:
            L.check(_lib().evlm_layernorm_bwd_add(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(rc), None, L.ptr(gamma.detach()),
                                                  L.ptr(mean), L.ptr(rstd), rows, d, L.ptr(dx), pdg, pdb, L.ptr(ws),
                                                  L.stream()), "layernorm_bwd_add")
        else:
            L.check(_lib().evlm_layernorm_bwd(L.dt(xc), L.ptr(dyc), L.ptr(xc), L.ptr(gamma.detach()), L.ptr(mean), L.ptr(rstd),
                                              rows, d, L.ptr(dx), pdg, pdb, L.ptr(ws), L.stream()), "layernorm_bwd")
        return (dx.view(ctx.xshape), None, None, None, None) if inplace else (dx.view(ctx.xshape), dg, db, None, None)


def layer_norm_fork_kd(x, gamma, beta, eps, kd_teacher, kd_slots, kd_coef):
    """(LayerNorm(x), alias of x for the residual branch, kd_slots): see _LayerNormForkKD"""
    return _LayerNormForkKD.apply(x, gamma, beta, eps, (kd_teacher, kd_slots, kd_coef))


def hidden_kd_slots():
    return _lib().evlm_layernorm_fwd_kd_slots()


def layer_norm(x, gamma, beta, eps):
    slot = getattr(x, "_evlm_drop_slot", None)
    if slot is not None and torch.is_grad_enabled() and x.requires_grad:
        return _LayerNorm.apply(x, gamma, beta, eps, slot)
    return _LayerNorm.apply(x, gamma, beta, eps, None)


def layer_norm_fork(x, gamma, beta, eps, tap=False):
    """(LayerNorm(x), alias of x for the residual branch [, alias of x for a distillation term]): see _LayerNormFork"""
    if not (torch.is_grad_enabled() and x.requires_grad) or _NO_FORK:
        y = _LayerNorm.apply(x, gamma, beta, eps, None)
        return (y, x, x) if tap else (y, x)
    return _LayerNormFork.apply(x, gamma, beta, eps, tap)


# ---------------------------------------------------------------------------------------------------
# attention core (probabilities are an output)
# ---------------------------------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    """qbuf: [B, Lq, *] packed buffer holding Q at column q_off; kvbuf: [Bkv, Lk, *] holding K at k_off and V at
    v_off (qbuf is kvbuf for self-attention).  Returns (O [B,Lq,H*dh], P [B,H,Lq,Lk])."""

    @staticmethod
    def forward(ctx, qbuf, kvbuf, mask, gate, H, dh, q_off, k_off, v_off, scale, want_probs, kv_index=None, causal=False,
                dropout_p=0.0, kd_teacher=None, kd_weight=1.0, p_out=None, kv_grad=None):
        L.require_cuda(qbuf, kvbuf)
        assert qbuf.is_contiguous() and kvbuf.is_contiguous()
        B, Lq, ldq = qbuf.shape
        Bkv, Lk, ldk = kvbuf.shape
        dev, tdt = qbuf.device, qbuf.dtype
        O = torch.empty((B, Lq, H * dh), dtype=tdt, device=dev)
        need = any(ctx.needs_input_grad)
        Lkp = _pad8(Lk)          # probability rows are padded to 16 bytes; the kernels zero the padding
        # Recomputing form (bf16 MFMA kernels, Lk <= 224; round 6: with or without probability dropout - the kernels
        # regenerate the keep-mask from (rng state, call id, row, key)): the forward keeps the per-row log2-sum-exp instead
        # of the map, the backward rebuilds P from Q and K in fp32 - no [B, H, Lq, Lk] bf16 map is written or read back
        # unless a caller wants it, and the q / k gradients are formed from fp32 probabilities, as the reference's
        # autocast softmax does.  EVLM_ATTN_STORE_P=1: the round-2 form (backward from the stored bf16 map).
        # Long key sequences (417..928: 384 x 384 / 480 x 480 images) take that form when NOBODY wants the map (so no dP can
        # come back for it): the backward then needs one pass over the keys - delta from dO . O (+ the fused distillation
        # term's row sums, kd_rowdot) - and the student's 518 MB-per-layer maps of the ITR / VQA steps are never written.
        # A caller that asks for the map keeps the stored-map form there.  EVLM_ATTN_RC_LONG=0: the round-3 behaviour.
        rc = bool(need and not ATTN_STORE_P
                  and _lib().evlm_attention_lse_supported(L.dt(tdt), dh, Lk, float(dropout_p or 0.0))
                  and not (dropout_p and causal and Lk > 224)
                  and (Lk <= 224 or (not want_probs and ATTN_RC_LONG)))
        lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev) if rc else None
        Pbuf = torch.empty((B, H, Lq, Lkp), dtype=tdt, device=dev) if (want_probs or (need and not rc)) else None
        if p_out is not None and Pbuf is not None:         # caller-owned (persistent) buffer for the map: no copy later
            if tuple(p_out.shape) != (B, H, Lq, Lkp) or p_out.dtype != tdt or not p_out.is_contiguous():
                raise RuntimeError("p_out must be a contiguous [B, H, Lq, pad8(Lk)] buffer of the activation dtype")
            Pbuf = p_out
        P = (Pbuf[..., :Lk] if Lkp != Lk else Pbuf) if Pbuf is not None else None
        m32 = mask.detach().to(torch.float32).contiguous() if mask is not None else None
        g32 = gate.detach().reshape(-1).to(torch.float32).contiguous() if gate is not None else None
        es = qbuf.element_size()
        a = L.AttnFwdArgs(dtype=L.dt(tdt), p_dtype=L.dt(tdt), B=B, H=H, Lq=Lq, Lk=Lk, dh=dh, ldq=ldq, ldk=ldk, ldv=ldk,
                          ldo=H * dh, ldpr=Lkp, Q=C.c_void_p(qbuf.data_ptr() + q_off * es),
                          K=C.c_void_p(kvbuf.data_ptr() + k_off * es), V=C.c_void_p(kvbuf.data_ptr() + v_off * es),
                          kv_index=L.ptr(kv_index), mask=L.ptr(m32), head_gate=L.ptr(g32), scale=scale, O=L.ptr(O),
                          P=L.ptr(Pbuf), causal=int(bool(causal)), lse=L.ptr(lse), Bkv=Bkv if kv_index is not None else 0)
        kd = kd_base = rkd = None
        recipe = kd_teacher if isinstance(kd_teacher, MapRecipe) else None
        if recipe is not None:                    # the teacher's map is REBUILT in the kernel from its Q, K and row lse (ABI 8)
            tq, tlse = recipe.qkv, recipe.lse
            if (not rc or Lk <= 224 or Pbuf is not None or m32 is not None or kv_index is not None or causal
                    or tuple(tq.shape) != (B, Lk, 3 * H * dh) or tq.dtype != torch.bfloat16 or not tq.is_contiguous()
                    or recipe.H != H or recipe.dh != dh or abs(recipe.scale - scale) > 1e-12 or qbuf is not kvbuf):
                raise RuntimeError("fused attention-map distillation from a MapRecipe needs a self-attention on 225..928 keys in "
                                   "the recomputing form (bf16, no map output, no mask) and a teacher of the same head geometry")
            if isinstance(kd_weight, KdSlot):
                kd, kd_weight = kd_weight.word, kd_weight.weight
            else:
                kd = zero_scalar(dev)
            a.kd_tq, a.kd_tk, a.kd_tld, a.kd_tlse = L.ptr(tq), C.c_void_p(tq.data_ptr() + H * dh * es), 3 * H * dh, L.ptr(tlse)
            a.kd_loss, a.kd_weight = L.ptr(kd), float(kd_weight)
            rkd = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
            a.kd_rowdot = L.ptr(rkd)
        elif kd_teacher is not None:              # fused map distillation: the teacher's (padded) map, read once in-kernel
            kd_base = _padded_base(kd_teacher, Lkp) if kd_teacher.shape[-1] != Lkp else kd_teacher
            if (kd_base is None or not kd_base.is_contiguous() or tuple(kd_base.shape) != (B, H, Lq, Lkp)
                    or kd_base.dtype != tdt or tdt != torch.bfloat16 or (Pbuf is None and lse is None)):
                raise RuntimeError("fused attention-map distillation needs the teacher map as a [B, H, Lq, Lk] view of a "
                                   "row-padded contiguous bf16 buffer (what the attention kernels return)")
            if isinstance(kd_weight, KdSlot):     # caller-provided zeroed f32 word (one fill for all layers of an encoder)
                kd, kd_weight = kd_weight.word, kd_weight.weight
            else:
                kd = zero_scalar(dev)
            a.kd_teacher, a.kd_loss, a.kd_weight = L.ptr(kd_base), L.ptr(kd), float(kd_weight)
            if rc and Lk > 224:                   # the distillation term's share of the backward's row sums
                rkd = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
                a.kd_rowdot = L.ptr(rkd)
        drop = None
        if dropout_p and dropout_p > 0.0:
            drop = (float(dropout_p), dropout_state(dev), _next_drop_call("attention_probs", (B, H, Lq, Lk), dropout_p))
            a.dropout_p, a.rng_state, a.call_id = drop[0], L.ptr(drop[1]), drop[2]
        L.check(_lib().evlm_attention_fwd(C.byref(a), L.stream()), "attention_fwd")
        if ATTN_FLOPS is not None:
            ATTN_FLOPS[0] += 4.0 * B * H * Lq * Lk * dh
        ctx.set_materialize_grads(False)        # an unused probability map must not cost a zero tensor in backward
        ctx.save_for_backward(qbuf, kvbuf, Pbuf, g32, kv_index, kd_base, lse, m32 if rc else None,
                              O if (rc and Lk > 224) else None, rkd,
                              recipe.qkv if recipe is not None else None, recipe.lse if recipe is not None else None)
        ctx.meta = (H, dh, q_off, k_off, v_off, scale, qbuf is kvbuf or qbuf.data_ptr() == kvbuf.data_ptr(),
                    gate.shape if gate is not None else None)
        ctx.causal = int(bool(causal))
        ctx.gate_slot = getattr(gate, "_evlm_gate_slot", None) if gate is not None else None
        ctx.drop = drop
        ctx.kd_weight = float(kd_weight) if kd_teacher is not None else 0.0
        ctx.kv_grad = kv_grad
        return O, P, kd

    @staticmethod
    def backward(ctx, dO, dP, dkd):
        qbuf, kvbuf, P, g32, kv_index, kd_base, lse, m32, O_fwd, rkd, t_qkv, t_lse = ctx.saved_tensors
        H, dh, q_off, k_off, v_off, scale, self_attn, gshape = ctx.meta
        B, Lq, ldq = qbuf.shape
        Bkv, Lk, ldk = kvbuf.shape
        dev, tdt = qbuf.device, qbuf.dtype
        es = qbuf.element_size()
        if dO is None:
            dO = torch.zeros((B, Lq, H * dh), dtype=tdt, device=dev)
        dOc = dO if dO.is_contiguous() else dO.contiguous()
        Lkp = _pad8(Lk)
        dPc = None
        if dP is not None:
            dPc = _padded_base(dP, Lkp)          # the KD loss hands back a view of a padded buffer: no copy
            if dPc is None:
                dPc = torch.zeros((B, H, Lq, Lkp), dtype=tdt, device=dev)
                dPc[..., :Lk].copy_(dP)
        d = H * dh
        slot = ctx.kv_grad
        assert ldq == (3 * d if self_attn else d) and (ldk == (3 * d if self_attn else 2 * d) or slot is not None), \
            "packed buffers must be exact"
        dqbuf = torch.empty_like(qbuf)                       # the kernels write every element of the packed grads
        if slot is not None:
            # K/V columns of a projection merged over several layers (KVGradSlot): every consumer writes its own columns of
            # ONE gradient buffer; the first to run hands the buffer to autograd, the others hand back nothing - no
            # [Bkv, Lk, n * 2d] additions, no zero fill
            first = slot.buf is None
            if first:
                slot.buf = torch.empty_like(kvbuf)
            dkvbuf = slot.buf
            slot.count += 1
        else:
            dkvbuf = dqbuf if self_attn else torch.empty_like(kvbuf)
        # dS workspace of the two-kernel path; the single-pass kernel (self-attention problems that fit one workgroup,
        # attention_mfma.hip) keeps dS in LDS and takes none
        single_pass = (tdt == torch.bfloat16 and dh == 64 and kv_index is None and Lq <= 224 and Lk <= 224
                       and (ctx.drop is None or (lse is not None and Lq <= 64 and Lk <= 64))
                       and os.environ.get("EVLM_ATTN_BWD_SPLIT", "0") in ("", "0"))
        dS = None if single_pass else torch.empty((B, H, Lq, Lkp), dtype=tdt, device=dev)
        # (two-kernel recomputing path: the second kernel reads the map the first one rebuilds)
        # ... except on the streaming one-pass path (attention_mfma.hip:launch_bwd_dq_stream - the same conditions): there
        # kernel B rebuilds the probabilities itself and a [B, H, Lq, Lkp] workspace (518 MB per layer at ITR-384, reserved
        # for the life of a captured step's pool) would never be touched
        env_on = lambda n: os.environ.get(n, "0") not in ("", "0")
        streams = (lse is not None and Lk > 224 and Lk <= 928 and not ctx.causal and O_fwd is not None and dPc is None
                   and ((kd_base is None and t_qkv is None) or dkd is None or rkd is not None) and tdt == torch.bfloat16 and dh == 64
                   and not env_on("EVLM_ATTN_NO_STREAM") and not env_on("EVLM_ATTN_STREAM_PWS"))
        # round 6 (EVLM_ATTN_KB_REBUILD=1, A/B): on <= 224 keys, too, kernel B may rebuild the probabilities from Q, K and the row lse
        # instead of reading a workspace kernel A writes (the cross-attention backward of the GD step: 2 x 37 MB per layer)
        kb_rebuild = (env_on("EVLM_ATTN_KB_REBUILD") and lse is not None and not single_pass and Lk <= 224 and P is None
                      and tdt == torch.bfloat16 and dh == 64)
        P_ws = (torch.empty((B, H, Lq, Lkp), dtype=tdt, device=dev)
                if (lse is not None and not single_pass and not streams and not kb_rebuild) else None)
        gslot = ctx.gate_slot if (g32 is not None and ctx.gate_slot is not None
                                  and ctx.gate_slot[0].row(ctx.gate_slot[1]).numel() == H) else None
        dgate = (gslot[0].row(gslot[1]) if gslot is not None
                 else (torch.zeros(H, dtype=torch.float32, device=dev) if g32 is not None else None))
        a = L.AttnBwdArgs(dtype=L.dt(tdt), p_dtype=L.dt(tdt), B=B, H=H, Lq=Lq, Lk=Lk, dh=dh, Bkv=Bkv, ldq=ldq, ldk=ldk, ldv=ldk,
                          ldo=H * dh, lddq=ldq, lddk=ldk, lddv=ldk, ldpr=Lkp,
                          Q=C.c_void_p(qbuf.data_ptr() + q_off * es), K=C.c_void_p(kvbuf.data_ptr() + k_off * es),
                          V=C.c_void_p(kvbuf.data_ptr() + v_off * es), P=L.ptr(P), dO=L.ptr(dOc), dP_ext=L.ptr(dPc),
                          kv_index=L.ptr(kv_index), head_gate=L.ptr(g32), scale=scale, dS=L.ptr(dS),
                          dQ=C.c_void_p(dqbuf.data_ptr() + q_off * es), dK=C.c_void_p(dkvbuf.data_ptr() + k_off * es),
                          dV=C.c_void_p(dkvbuf.data_ptr() + v_off * es), dgate=L.ptr(dgate),
                          lse=L.ptr(lse), mask=L.ptr(m32), causal=ctx.causal if lse is not None else 0, P_ws=L.ptr(P_ws),
                          O=L.ptr(O_fwd), kd_rowdot=L.ptr(rkd))
        if t_qkv is not None and dkd is not None:         # ... from the teacher's Q, K and row lse (MapRecipe: ABI 8)
            gk = dkd.to(torch.float32).contiguous()
            a.kd_tq, a.kd_tk, a.kd_tld, a.kd_tlse = (L.ptr(t_qkv), C.c_void_p(t_qkv.data_ptr() + H * dh * es), 3 * H * dh,
                                                     L.ptr(t_lse))
            a.kd_gout, a.kd_weight = L.ptr(gk), ctx.kd_weight
        if kd_base is not None and dkd is not None:       # dP of the fused distillation term is formed in-kernel from P_t
            gk = dkd.to(torch.float32).contiguous()
            a.kd_teacher, a.kd_gout, a.kd_weight = L.ptr(kd_base), L.ptr(gk), ctx.kd_weight
        if ctx.drop is not None:          # the keep-mask is regenerated from the same (state, call id), never stored
            a.dropout_p, a.rng_state, a.call_id = ctx.drop[0], L.ptr(ctx.drop[1]), ctx.drop[2]
        L.check(_lib().evlm_attention_bwd(C.byref(a), L.stream()), "attention_bwd")
        if ATTN_FLOPS is not None:
            ATTN_FLOPS[0] += (10.0 if lse is not None else 8.0) * B * H * Lq * Lk * dh     # dP, dV, dQ, dK (+ S when P is recomputed)
        dg = (gslot[0].take(gslot[1], gshape) if gslot is not None else (dgate.view(gshape) if dgate is not None else None))
        if self_attn:
            return (dqbuf, None, None, dg) + (None,) * 14
        if slot is not None and not first:
            return (dqbuf, None, None, dg) + (None,) * 14
        return (dqbuf, dkvbuf, None, dg) + (None,) * 14


def self_attention(qkv, H, dh, scale, mask=None, gate=None, want_probs=True, causal=False, dropout_p=0.0,
                   kd_teacher=None, kd_weight=1.0, p_out=None):
    """qkv: [B, L, 3*H*dh] packed (q | k | v); causal: additionally -10000 on keys after the query (decoder self-attention:
    the backward works from the saved probabilities, so only the forward kernel knows about masks); dropout_p: dropout of
    the probabilities that form the context (the returned map stays un-dropped, eff_bert.py:338-361)"""
    d = H * dh
    O, P, kd = _Attention.apply(qkv, qkv, mask, gate, H, dh, 0, d, 2 * d, scale, want_probs, None, causal, dropout_p,
                                kd_teacher, kd_weight, p_out)
    return (O, P) if kd_teacher is None else (O, P, kd)


class MapRecipe(tuple):
    """What a frozen teacher keeps of one self-attention layer INSTEAD of its probability map when the map's only reader is a
    fused distillation term on a long key sequence (ABI 8: evlm_attn_*_args.kd_tq / kd_tk / kd_tlse): its packed QKV buffer
    [B, L, 3 H dh] (bf16) and its row lse [B, H, L] (f32, log2 domain).  The student's streaming attention kernels rebuild
    P_t = 2^(scale q_t.k_t log2e - lse_t) per key tile - 113 MB + 1.8 MB per ViT layer at 577 tokens instead of a 517 MB map
    written once and read twice.  A tuple of tensors, so the trainers' nest walkers (distill._tensors) see its members."""

    def __new__(cls, qkv, lse, H, dh, scale):
        self = super().__new__(cls, (qkv, lse))
        self.H, self.dh, self.scale = int(H), int(dh), float(scale)
        return self

    qkv = property(lambda self: self[0])
    lse = property(lambda self: self[1])

    @property
    def shape(self):                       # (shape of the map it stands for)
        B, Lk = self[0].shape[0], self[0].shape[1]
        return (B, self.H, Lk, Lk)

    def detach(self):
        return self


def map_recipe_supported(qkv, H, dh, mask=None, causal=False):
    """can a no-grad self-attention on this packed buffer hand out a MapRecipe (streaming kernels: 225..928 keys, bf16,
    head dim 64, no mask)?"""
    return bool(qkv.is_cuda and qkv.dtype == torch.bfloat16 and dh == 64 and 224 < qkv.shape[1] <= 928 and mask is None
                and not causal and not torch.is_grad_enabled() and not os.environ.get("EVLM_NO_KD_RECIPE")
                and os.environ.get("EVLM_ATTN_NO_STREAM", "0") in ("", "0") and ATTN_RC_LONG and not ATTN_STORE_P
                # (the student's side needs the recomputing form of this length: 417..928 keys)
                and _lib().evlm_attention_lse_supported(L.dt(qkv.dtype), dh, qkv.shape[1], 0.0))


def self_attention_recipe(qkv, H, dh, scale, gate=None):
    """no-grad self-attention that keeps (context, MapRecipe) - no probability map is written (see MapRecipe)"""
    L.require_cuda(qkv)
    assert qkv.is_contiguous() and not torch.is_grad_enabled()
    B, Lq, ld = qkv.shape
    d = H * dh
    dev, es = qkv.device, qkv.element_size()
    O = torch.empty((B, Lq, d), dtype=qkv.dtype, device=dev)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
    g32 = gate.detach().reshape(-1).to(torch.float32).contiguous() if gate is not None else None
    a = L.AttnFwdArgs(dtype=L.dt(qkv.dtype), p_dtype=L.dt(qkv.dtype), B=B, H=H, Lq=Lq, Lk=Lq, dh=dh, ldq=ld, ldk=ld, ldv=ld,
                      ldo=d, ldpr=_pad8(Lq), Q=C.c_void_p(qkv.data_ptr()), K=C.c_void_p(qkv.data_ptr() + d * es),
                      V=C.c_void_p(qkv.data_ptr() + 2 * d * es), head_gate=L.ptr(g32), scale=scale, O=L.ptr(O), lse=L.ptr(lse))
    L.check(_lib().evlm_attention_fwd(C.byref(a), L.stream()), "attention_fwd")
    if ATTN_FLOPS is not None:
        ATTN_FLOPS[0] += 4.0 * B * H * Lq * Lq * dh
    return O, MapRecipe(qkv, lse, H, dh, scale)


class KdSlot:
    """weight of a fused map-distillation term + the (already zeroed) device f32 word its kernel accumulates into; a plain
    object, so autograd does not see the word as an input of the attention Function"""

    def __init__(self, word, weight):
        self.word, self.weight = word, float(weight)


def attention_recomputes(x, dh, Lk, dropout_p=0.0):
    """will the attention backward of this problem rebuild the probabilities from Q and K (bf16 MFMA kernels, lse saved by
    the forward) instead of reading a stored map - provided, on long key sequences, that the caller does not ask for the map?"""
    return bool(x.is_cuda and not ATTN_STORE_P and (Lk <= 224 or ATTN_RC_LONG)
                and _lib().evlm_attention_lse_supported(L.dt(x.dtype), dh, Lk, float(dropout_p or 0.0)))


def attention_kd_fusable(x, H, dh, Lk):
    """can the attention-map distillation of this problem run inside the attention kernels (bf16 MFMA path)?"""
    return x.is_cuda and x.dtype == torch.bfloat16 and dh == 64 and Lk <= 928


class GateGradSlot:
    """gradient buffer of ONE L0 gate tensor z [n, ...] (a row per gated layer: xvlm_l0_module.py zs['*_z']): zeroed, f32,
    a slice of the step's zero arena.  Every consumer of row i - an attention or an FFN backward kernel - ACCUMULATES its
    gate gradient into row i itself; the first consumer of a row hands the row to autograd, later ones hand back nothing
    (the KVGradSlot protocol), and _GateFanout.backward returns the whole buffer (a row whose only consumers know nothing of
    the slot is added the ordinary way; a row with BOTH kinds of consumer raises - its alias would be counted twice).  Before: one zero-filled vector per
    consumer, one zero-filled [n, ...] tensor + one add per row in autograd's SelectBackward - ~70 launches of a few
    microseconds each on the critical path of a pruning step."""

    def __init__(self, z):
        self.buf = zeros_small(tuple(z.shape), torch.float32, z.device)
        self.handed = set()

    def row(self, i):
        return self.buf[i].reshape(-1)

    def take(self, i, shape):
        if i in self.handed:
            return None
        self.handed.add(i)
        return self.buf[i].view(shape)


class _GateFanout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, slot):
        ctx.slot = slot
        return tuple(z[i] for i in range(z.shape[0]))

    @staticmethod
    def backward(ctx, *grads):
        slot = ctx.slot
        buf = slot.buf
        for i, g in enumerate(grads):          # a row whose consumer did not go through the slot (an op that knows nothing
            if g is not None and g.data_ptr() != buf[i].data_ptr():      # of it): its gradient is added the ordinary way
                if i in slot.handed:
                    # a slot-aware consumer handed autograd an ALIAS of buf[i]; summed with another consumer's gradient it
                    # comes back as a fresh tensor that already contains buf[i] - adding it would count the row twice
                    raise RuntimeError(f"gate row {i} has both a slot-aware consumer (attention / FFN kernels) and one that "
                                       "returns its own gradient: read the gate through ops.gate_rows consumers only, or "
                                       "set EVLM_NO_GATE_SLOTS=1")
                buf[i].add_(g.reshape(buf[i].shape).to(buf.dtype))
        slot.handed = set()
        return buf, None


def gate_rows(z):
    """the rows of an L0 gate tensor z [n, ...] as a tuple the encoders index per layer (head_z[i], mlp_z[i]): each row
    carries its GateGradSlot, so the kernels that consume it accumulate its gradient in place (see GateGradSlot).
    Anything else - no gradient wanted, not a float32 CUDA tensor, EVLM_NO_GATE_SLOTS=1 - is returned unchanged."""
    if (not torch.is_tensor(z) or not z.is_cuda or z.dtype != torch.float32 or z.dim() < 1
            or not (torch.is_grad_enabled() and z.requires_grad) or _NO_GATE_SLOTS):
        return z
    slot = GateGradSlot(z)
    rows = _GateFanout.apply(z, slot)
    for i, r in enumerate(rows):
        r._evlm_gate_slot = (slot, i)
    return rows


_NO_GATE_SLOTS = bool(os.environ.get("EVLM_NO_GATE_SLOTS"))
# evlm_gemm_args.dgate (the gated activation backward inside the dX product's epilogue) is OPT-IN, EVLM_GATED_DACT_FOLD=1:
# same-box A/B x3 (profiles/r05_pruning_residue.md) ITR-384 36.39 vs 36.43 ms, VQA-480 32.01 vs 31.95 ms - the 192-row
# kernel that hosts the flavour costs on these products what the saved pass over [rows, ffn] was worth
_NO_GATED_DACT_FOLD = not os.environ.get("EVLM_GATED_DACT_FOLD")


class KVGradSlot:
    """gradient buffer shared by the `n` attention calls that read their K / V out of ONE merged projection (see
    merged_kv): allocated by the first backward that runs, complete when all `n` have written their columns"""

    def __init__(self, n):
        self.buf, self.expected, self.count = None, int(n), 0


class _KVJoin(torch.autograd.Function):
    """identity between a merged K/V projection and its consumers: checks, when the gradient passes, that every consumer
    has written its columns of the shared buffer (a fusion layer whose output went unused would leave its columns
    uninitialised)"""

    @staticmethod
    def forward(ctx, kv, slot):
        ctx.slot = slot
        return kv.view_as(kv)

    @staticmethod
    def backward(ctx, g):
        slot = ctx.slot
        buf, count = slot.buf, slot.count
        slot.buf, slot.count = None, 0
        if g is not None and (buf is None or g.data_ptr() != buf.data_ptr() or count != slot.expected):
            raise RuntimeError(f"merged K/V projection: {count} of {slot.expected} consumers produced a gradient "
                               "(every layer reading the merged buffer must take part in backward)")
        return g, None


def merged_kv(x, weights, biases, n):
    """K/V of `n` cross-attention layers that read the same tokens in ONE product: x [Bkv, Lk, K] -> ([Bkv, Lk, n * 2d]
    holding k_0 | v_0 | k_1 | v_1 ..., KVGradSlot | None).  eff_bert.py:284-296 issues one Linear pair per layer; the
    arithmetic per output column is the same."""
    kv = linear_packed(x, tuple(weights), tuple(biases))
    if not (torch.is_grad_enabled() and kv.requires_grad):
        return kv, None
    slot = KVGradSlot(n)
    return _KVJoin.apply(kv, slot), slot


def cross_attention(q, kv, H, dh, scale, mask=None, gate=None, want_probs=True, kv_index=None, dropout_p=0.0,
                    kv_col=None, kv_grad=None):
    """q: [B, Lq, H*dh]; kv: [Bkv, Lk, 2*H*dh] packed (k | v).  kv_index (int32 [B]) maps each query batch to its K/V
    row, so image tokens shared by several text batches are projected (and their gradient reduced) once.
    kv_col / kv_grad: kv is a merged buffer (merged_kv) and this call's k | v start at column kv_col."""
    d = H * dh
    k_off = 0 if kv_col is None else int(kv_col)
    if kv_index is not None:
        if q.dtype == torch.bfloat16 and dh == 64 and kv.shape[1] <= 928:
            kv_index = kv_index.to(torch.int32).contiguous()
        else:                                   # exact-fp32 / generic path: materialise the gather (autograd scatters back)
            if kv_col is not None:
                raise RuntimeError("merged K/V buffers need the in-kernel K/V index (bf16, dh = 64)")
            kv = torch.index_select(kv, 0, kv_index.long())
            kv_index = None
    return _Attention.apply(q, kv, mask, gate, H, dh, 0, k_off, k_off + d, scale, want_probs, kv_index, False, dropout_p,
                            None, 1.0, None, kv_grad)[:2]


_PIN_CHUNKS, _PIN_USED = [], 0            # pinned int64 chunks (every one stays alive: pending uploads read them)
_PIN_CHUNK_WORDS = 1 << 17                # 1 MiB each: ~700 tables of 45 copies
_DEV_CHUNKS = {}                          # device index -> [int64 device chunk (outside any graph pool), words used]
_PENDING_UPLOADS = []                     # (device slice, pinned slice) of tables built during a capture


def reserve_tables(min_words=1 << 14):
    """make sure the table pools (pinned host words + a device arena on the current device) have `min_words` free int64
    words - call OUTSIDE a hipGraph capture, before one starts (neither pinning nor a persistent allocation is capturable; a
    capture consumes a few KiB of tables and the words are never reclaimed: the captured launches keep reading them).  Grows
    a pool by another chunk when the current one is short.  Also sends up tables an earlier capture left pending."""
    global _PIN_USED
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("reserve_tables() inside a hipGraph capture")
    flush_table_uploads()
    if not _PIN_CHUNKS or _PIN_USED + min_words > _PIN_CHUNKS[-1].numel():
        _PIN_CHUNKS.append(torch.empty(max(_PIN_CHUNK_WORDS, min_words), dtype=torch.int64).pin_memory())
        _PIN_USED = 0
    d = torch.cuda.current_device()
    ent = _DEV_CHUNKS.get(d)
    if ent is None or ent[1] + min_words > ent[0].numel():
        ent = _DEV_CHUNKS.setdefault(d, [None, 0])
        ent[0] = torch.empty(max(_PIN_CHUNK_WORDS, min_words), dtype=torch.int64, device=torch.device("cuda", d))
        ent[1] = 0            # (the chunk it replaces stays alive through the slices handed out of it)


def flush_table_uploads():
    """send up the tables built during a hipGraph capture - ONCE, eagerly, after the capture has ended and before the graph
    is first replayed (every trainer does so right behind its captures; reserve_tables() and the next eager table do it
    too).  A captured launch reads its table out of the persistent device arena: there is no memcpy node per table in the
    graph (the GD step's graphs held ~55 of them, each a 6 us copy kernel on some launch's critical path)."""
    global _PENDING_UPLOADS
    if not _PENDING_UPLOADS or torch.cuda.is_current_stream_capturing():
        return
    q, _PENDING_UPLOADS = _PENDING_UPLOADS, []
    devs = set()
    for dst, src in q:
        dst.copy_(src, non_blocking=True)
        devs.add(dst.device)
    for d in devs:
        torch.cuda.synchronize(d)         # (the graph may be replayed on any stream)


def _upload_table(rows, dev):
    """flat list of int64 -> device tensor.  Inside a hipGraph capture the table is a slice of the persistent device arena
    (reserve_tables), filled once after the capture (flush_table_uploads) out of a pinned slice that is never rewritten:
    its contents - addresses inside the graph's pool, sizes - are the same at every replay."""
    global _PIN_USED
    if torch.cuda.is_current_stream_capturing():
        ent = _DEV_CHUNKS.get(torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device())
        if (not _PIN_CHUNKS or _PIN_USED + len(rows) > _PIN_CHUNKS[-1].numel() or ent is None
                or ent[1] + len(rows) > ent[0].numel()):
            raise RuntimeError("table pools missing / exhausted (ops.reserve_tables() before the capture starts)")
        host = _PIN_CHUNKS[-1][_PIN_USED:_PIN_USED + len(rows)]
        _PIN_USED += len(rows)
        host.copy_(torch.tensor(rows, dtype=torch.int64))
        if os.environ.get("EVLM_TABLE_MEMCPY_NODES"):        # A/B switch: the round-2 form, one captured memcpy per table
            table = torch.empty(len(rows), dtype=torch.int64, device=dev)
            table.copy_(host, non_blocking=True)
            return table
        table = ent[0][ent[1]:ent[1] + len(rows)]
        ent[1] += len(rows)
        _PENDING_UPLOADS.append((table, host))
        return table
    reserve_tables()
    return torch.tensor(rows, dtype=torch.int64).to(dev)


def copy_grouped(pairs):
    """[(src, dst)] contiguous same-size tensors -> ONE copy launch (evlm_copy_grouped); returns the device table (keep it
    alive while a captured graph may replay the launch)"""
    rows, blocks = [], 0
    for src, dst in pairs:
        nb = src.numel() * src.element_size()
        if (nb != dst.numel() * dst.element_size() or nb % 16 or src.data_ptr() % 16 or dst.data_ptr() % 16
                or not src.is_contiguous() or not dst.is_contiguous()):
            raise RuntimeError("copy_grouped: contiguous, 16-byte aligned tensors of equal byte size (multiple of 16)")
        rows += [src.data_ptr(), dst.data_ptr(), nb, blocks]
        blocks += (nb + 65535) // 65536
    dev = pairs[0][0].device
    table = _upload_table(rows, dev)
    L.check(_lib().evlm_copy_grouped(L.ptr(table), len(pairs), blocks, L.stream()), "copy_grouped")
    return table


def copy_few(pairs):
    """[(src, dst)] -> as few launches as possible WITHOUT a device table (evlm_copy_few: up to 8 units per launch, passed
    by value): for copies whose source addresses change from step to step - a training batch going into its static buffers.
    Pairs that do not qualify (size not a multiple of 16 bytes, unaligned, not contiguous, dtypes differ) use Tensor.copy_."""
    good = []
    for src, dst in pairs:
        nb = src.numel() * src.element_size()
        if (src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and nb and nb % 16 == 0 and src.is_contiguous() and dst.is_contiguous()
                and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0 and nb == dst.numel() * dst.element_size()):
            good.append((src, dst, nb))
        else:
            dst.copy_(src, non_blocking=True)
    for i in range(0, len(good), 8):
        part = good[i:i + 8]
        n = len(part)
        srcs = (C.c_void_p * n)(*[p_[0].data_ptr() for p_ in part])
        dsts = (C.c_void_p * n)(*[p_[1].data_ptr() for p_ in part])
        nbs = (C.c_int64 * n)(*[p_[2] for p_ in part])
        L.check(_lib().evlm_copy_few(srcs, dsts, nbs, n, L.stream()), "copy_few")


def zero_grouped(tensors):
    """zero-fill every tensor of `tensors` in ONE launch (evlm_copy_grouped units without a source) - a step's gradient
    ranges were one fill launch each.  Tensors that do not qualify (not contiguous, not 16-byte sized / aligned) are filled
    one by one.  Returns the device table (keep it alive while a captured graph may replay the launch) or None."""
    rows, blocks, n = [], 0, 0
    for t in tensors:
        nb = t.numel() * t.element_size()
        if nb == 0:
            continue
        if not t.is_cuda or nb % 16 or t.data_ptr() % 16 or not t.is_contiguous():
            t.zero_()
            continue
        rows += [0, t.data_ptr(), nb, blocks]
        blocks += (nb + 65535) // 65536
        n += 1
        dev = t.device
    if n == 0:
        return None
    if n == 1:                                   # (one range: a plain fill, no table)
        for t in tensors:
            if t.data_ptr() == rows[1]:
                t.zero_()
                return None
    key = (dev, tuple(rows))
    table = _ZERO_CACHE.get(key)                 # (a step zeroes the same ranges every time: the table is uploaded once)
    if table is None:
        table = _upload_table(rows, dev)
        # (never evicted: a later capture may have recorded a launch that reads a cached table; tables are a few hundred bytes)
        if not torch.cuda.is_current_stream_capturing() and len(_ZERO_CACHE) < 256:
            _ZERO_CACHE[key] = table
    L.check(_lib().evlm_copy_grouped(L.ptr(table), n, blocks, L.stream()), "zero_grouped")
    return table


_ZERO_CACHE = {}
_ZERO_TABLES = []         # tables of captured / in-flight zero_grouped launches (bounded: the oldest go once 64 are held)


def _keep_table(t):
    if t is not None:
        _ZERO_TABLES.append(t)
        if len(_ZERO_TABLES) > 64 and not torch.cuda.is_current_stream_capturing():
            del _ZERO_TABLES[:-64]


def xattn_fusable(q, x_img, weights, H, dh):
    """does the fused cross-attention forward (evlm_xattn_fused_fwd: K/V projection + attention in one launch, nothing
    kept for a backward) apply?  no-grad forwards only - the frozen teacher, inference"""
    if torch.is_grad_enabled() and (q.requires_grad or x_img.requires_grad or any(w.requires_grad for w in weights)):
        return False
    d = H * dh
    return (q.is_cuda and q.dtype == torch.bfloat16 and x_img.dtype == torch.bfloat16 and dh == 64 and H % 2 == 0
            and x_img.shape[-1] == d and weights[0].shape == (d, d) and x_img.shape[1] <= 224 and d >= 128)


def cross_attention_fused(q, x_img, weights, biases, H, dh, scale, mask=None, gate=None, want_probs=False, kv_index=None):
    """q [Bq, Lq, d] (projected queries), x_img [Bimg, N, d] image tokens, weights = (Wk, Wv), biases = (bk, bv).
    Returns (O [Bq, Lq, d], P [Bq, H, Lq, N] | None) like cross_attention(q, linear_packed(x_img, ...), ...) - same
    arithmetic (the K/V tiles are rounded to bf16 exactly where the two-launch path rounds them), no K/V in HBM."""
    L.require_cuda(q, x_img)
    Bq, Lq, d = q.shape
    Bimg, N, _ = x_img.shape
    qc = q if q.is_contiguous() else q.contiguous()
    xc = x_img if x_img.is_contiguous() else x_img.contiguous()
    W = CACHE.get(tuple(weights), torch.bfloat16)
    b = CACHE.get(tuple(biases), torch.float32) if biases and biases[0] is not None else None
    O = torch.empty((Bq, Lq, d), dtype=torch.bfloat16, device=q.device)
    Lkp = _pad8(N)
    Pbuf = torch.empty((Bq, H, Lq, Lkp), dtype=torch.bfloat16, device=q.device) if want_probs else None
    m32 = mask.detach().to(torch.float32).contiguous() if mask is not None else None
    g32 = gate.detach().reshape(-1).to(torch.float32).contiguous() if gate is not None else None
    idx = kv_index.to(torch.int32).contiguous() if kv_index is not None else None
    if idx is None and Bq != Bimg:
        raise RuntimeError("cross_attention_fused: without kv_index every query batch needs its own image")
    a = L.XAttnFusedArgs(dtype=L.BF16, Bimg=Bimg, Bq=Bq, N=N, Lq=Lq, d=d, H=H, dh=dh, ldx=d, ldq=d, ldo=d, ldpr=Lkp,
                         X=L.ptr(xc), Wkv=L.ptr(W), bias_kv=L.ptr(b), Q=L.ptr(qc), kv_index=L.ptr(idx), mask=L.ptr(m32),
                         head_gate=L.ptr(g32), scale=scale, O=L.ptr(O), P=L.ptr(Pbuf))
    L.check(_lib().evlm_xattn_fused_fwd(C.byref(a), L.stream()), "xattn_fused_fwd")
    if GEMM_PROFILE is not None or ATTN_FLOPS is not None:
        if ATTN_FLOPS is not None:
            ATTN_FLOPS[0] += 4.0 * Bq * H * Lq * N * dh + 2.0 * Bimg * N * d * 2 * d
    P = (Pbuf[..., :N] if Lkp != N else Pbuf) if Pbuf is not None else None
    return O, P


# ---------------------------------------------------------------------------------------------------
# dropout (counter-based masks: regenerated in backward, never stored)
# ---------------------------------------------------------------------------------------------------
_DROP_STATE = {}
_DROP_CALL = [0]
_DROP_KINDS = {}          # call id -> kind of the sites logged while DROPOUT_LOG is set (dropout_mask reads it)
DROPOUT_LOG = None        # tests: set to a list to collect (call id, kind, shape, p) of every dropout site of a forward
DROPOUT_USED = False      # a trainer bumps the device-side step word once per step while this is set


def dropout_state(device):
    """device int64[2] {seed, step} every dropout site of this device draws from (evlm_dropout, include/evlm_hip.h)"""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    st = _DROP_STATE.get(key)
    if st is None:
        st = _DROP_STATE[key] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
    return st


def dropout_seed(seed, device="cuda"):
    """re-seed the device random stream of `device` (dropout masks, hard-negative draws): step word and call-site counter
    back to 0, so two runs issuing the same calls draw the same numbers"""
    _DROP_CALL[0] = 0
    st = dropout_state(torch.device(device))
    st.copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))


def dropout_tick(device="cuda"):
    """advance the step word ON THE DEVICE (capturable): masks of a replayed hipGraph change from replay to replay"""
    st = dropout_state(torch.device(device))
    st[1:].add_(1)


def _next_drop_call(kind, shape, p):
    global DROPOUT_USED
    DROPOUT_USED = True
    _DROP_CALL[0] = (_DROP_CALL[0] + 1) & 0xFFFFFFFF
    if DROPOUT_LOG is not None:
        DROPOUT_LOG.append((_DROP_CALL[0], kind, tuple(shape), float(p)))
        _DROP_KINDS[_DROP_CALL[0]] = kind
    return _DROP_CALL[0]


def sample_negatives(sim, temp, group=None, layout=False):
    """ITM hard negatives in one launch (evlm_sample_negatives; reference efficient_models/xvlm.py:422-458): sim f32 [B, B]
    image x text similarities, temp the (device) temperature, group optional int64 [B] positive-group ids.  Returns int64
    [2B]: for every text an image index, then for every image a text index.  Draws from the device Philox stream of
    dropout_state() - fresh on every step (and every graph replay) once the trainer ticks the step word."""
    global DROPOUT_USED
    L.require_cuda(sim)
    B = sim.shape[0]
    assert sim.dim() == 2 and sim.shape[1] == B and sim.dtype == torch.float32 and sim.stride(1) == 1
    t = temp.detach().reshape(-1)[:1].to(torch.float32) if torch.is_tensor(temp) else \
        torch.full((1,), float(temp), dtype=torch.float32, device=sim.device)
    g = None if group is None else group.reshape(-1).to(torch.int64).contiguous()
    DROPOUT_USED = True
    _DROP_CALL[0] = (_DROP_CALL[0] + 1) & 0xFFFFFFFF
    out = torch.empty(2 * B, dtype=torch.int64, device=sim.device)
    sel4 = torch.empty(4 * B, dtype=torch.int64, device=sim.device) if layout else None
    img4 = torch.empty(4 * B, dtype=torch.int32, device=sim.device) if layout else None
    L.check(_lib().evlm_sample_negatives(L.ptr(sim), B, sim.stride(0), L.ptr(t), L.ptr(g) if g is not None else None,
                                         L.ptr(dropout_state(sim.device)), _DROP_CALL[0], L.ptr(out),
                                         L.ptr(sel4) if layout else None, L.ptr(img4) if layout else None, L.stream()),
            "sample_negatives")
    # layout: + (sel4, img4), the batched fusion pass's row / image indices written by the same launch (evlm_hip.h)
    return (out, sel4, img4) if layout else out


class _SelectBatches(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sel):
        L.require_cuda(x, sel)
        xc = x if x.is_contiguous() else x.contiguous()
        n, rows = sel.numel(), xc.shape[0]
        out = torch.empty((n,) + tuple(xc.shape[1:]), dtype=xc.dtype, device=xc.device)
        rb = (xc.numel() // rows) * xc.element_size()
        L.check(_lib().evlm_select_batches_fwd(L.ptr(xc), L.ptr(sel), n, rb, L.ptr(out), L.stream()), "select_batches_fwd")
        ctx.save_for_backward(sel)
        ctx.meta = (n, rows, xc.numel() // rows, tuple(xc.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        (sel,) = ctx.saved_tensors
        n, rows, w, shape = ctx.meta
        dc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty(shape, dtype=dy.dtype, device=dy.device)
        L.check(_lib().evlm_select_batches_bwd(L.dt(dc), L.ptr(dc), L.ptr(sel), n, rows, w, L.ptr(dx), L.stream()), "select_batches_bwd")
        return dx, None


def select_batches(x, sel):
    """x[sel] over whole samples (dim 0) in one launch, with a deterministic one-launch backward (evlm_select_batches_*):
    sel int64 [n], x contiguous with samples of a multiple of 16 bytes (and, to be differentiable, of 8 elements)"""
    assert sel.dtype == torch.int64 and sel.is_contiguous()
    return _SelectBatches.apply(x, sel)


class _ITCLoss(torch.autograd.Function):
    """evlm_itc_loss_fwd / _bwd.  a: image features [Bt, E], b: text features [Bt, E] - or a = the gathered [Bt, 2E] buffer
    [image | text] and b = None (its gradient then comes back as one [Bt, 2E] tensor for the gather's slice-only backward)"""

    @staticmethod
    def forward(ctx, a, b, temp, group):
        L.require_cuda(a)
        packed = b is None
        if a.stride(-1) != 1 or (a.stride(0) * a.element_size()) % 16 or a.data_ptr() % 16:
            a = a.contiguous()
        if not packed and (b.stride(-1) != 1 or (b.stride(0) * b.element_size()) % 16 or b.data_ptr() % 16 or b.dtype != a.dtype):
            b = b.to(a.dtype).contiguous()
        Bt = a.shape[0]
        E = a.shape[1] // 2 if packed else a.shape[1]
        I, ldi = a, a.stride(0)
        Tp, ldt = (C.c_void_p(a.data_ptr() + E * a.element_size()), ldi) if packed else (L.ptr(b), b.stride(0))
        t = temp.detach().reshape(-1)[:1]
        assert t.dtype == torch.float32
        g = None if group is None else group.reshape(-1).to(torch.int64).contiguous()
        lds = _pad8(Bt)
        sim = torch.empty((Bt, lds), dtype=torch.float32, device=a.device)
        stats = zeros_small((4 * Bt + 8,), torch.float32, a.device)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        L.check(_lib().evlm_itc_loss_fwd(L.dt(a), L.ptr(I), ldi, Tp, ldt, Bt, E, L.ptr(t), L.ptr(g) if g is not None else None,
                                         L.ptr(sim), lds, L.ptr(stats), L.ptr(loss), L.stream()), "itc_loss_fwd")
        ctx.save_for_backward(a, b, t, g, sim, stats)
        ctx.dims = (packed, Bt, E, lds, tuple(temp.shape))
        sim_v = sim[:, :Bt]
        ctx.mark_non_differentiable(sim_v)
        ctx.set_materialize_grads(False)          # (no zero tensor for the similarities' "gradient")
        return loss, sim_v

    @staticmethod
    def backward(ctx, dloss, _dsim):
        if dloss is None:
            return None, None, None, None
        a, b, t, g, sim, stats = ctx.saved_tensors
        packed, Bt, E, lds, tshape = ctx.dims
        dl = dloss.reshape(-1)[:1]
        if dl.dtype != torch.float32:
            dl = dl.float()
        ldi = a.stride(0)
        Tp, ldt = (C.c_void_p(a.data_ptr() + E * a.element_size()), ldi) if packed else (L.ptr(b), b.stride(0))
        if packed:
            da = torch.empty((Bt, 2 * E), dtype=a.dtype, device=a.device)
            dIp, lddi, dTp, lddt, db = L.ptr(da), 2 * E, C.c_void_p(da.data_ptr() + E * da.element_size()), 2 * E, None
        else:
            da, db = torch.empty((Bt, E), dtype=a.dtype, device=a.device), torch.empty((Bt, E), dtype=a.dtype, device=a.device)
            dIp, lddi, dTp, lddt = L.ptr(da), E, L.ptr(db), E
        dtemp = torch.empty(1, dtype=torch.float32, device=a.device)
        L.check(_lib().evlm_itc_loss_bwd(L.dt(a), L.ptr(a), ldi, Tp, ldt, Bt, E, L.ptr(t), L.ptr(g) if g is not None else None,
                                         L.ptr(sim), lds, L.ptr(stats), L.ptr(dl), dIp, lddi, dTp, lddt, L.ptr(dtemp),
                                         L.stream()), "itc_loss_bwd")
        return da, db, dtemp.view(tshape), None


def itc_loss(image_feat, text_feat, temp, group=None):
    """the ITC loss of efficient_models/xvlm.py:384-416 over the (gathered) batch in one launch each way: returns (loss, sim)
    - sim f32 [Bt, Bt] = I T^t un-scaled, which the ITM hard-negative sampler reads instead of forming it again.
    text_feat None: image_feat is the gathered [Bt, 2E] buffer [image | text]."""
    return _ITCLoss.apply(image_feat, text_feat, temp, group)


def dropout_mask(call_id, shape, p, device="cuda", kind=None):
    """keep / (1 - p) of dropout site `call_id` as an f32 tensor of `shape` (what the kernels regenerate on the fly; tests
    hand it to the CPU oracle).  A site's elements are indexed with its rows padded to a multiple of 8 columns (csrc/common.h:
    one Philox call per 8 consecutive columns): attention-probability sites ([B, H, Lq, Lk], any Lk) are generated padded and
    sliced; hidden-state sites are addressed flat (their rows are multiples of 8 wherever a kernel other than evlm_dropout
    regenerates them).  kind: "attention_probs" | "hidden"; default = what DROPOUT_LOG recorded for the call id."""
    kind = kind or _DROP_KINDS.get(call_id, "hidden")
    shape = tuple(shape)
    pshape = shape[:-1] + ((shape[-1] + 7) // 8 * 8,) if kind == "attention_probs" else shape
    n = 1
    for v in pshape:
        n *= v
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.check(_lib().evlm_dropout_mask(n, float(p), L.ptr(dropout_state(torch.device(device))), call_id, L.ptr(out), L.stream()),
            "dropout_mask")
    out = out.view(*pshape)
    return out[..., :shape[-1]].contiguous() if pshape != shape else out


class DropSlot:
    """One hidden-dropout site fused into a GEMM's residual epilogue: y = (x W^T + b) .* M + residual (linear / mlp with
    dropout_p).  Holds what regenerates M = keep / (1 - p) - the device {seed, step} word and the site's call id - and carries
    the site's MASKED gradient from the LayerNorm that consumes y to the product's backward: evlm_layernorm_bwd_drop writes
    dx (for the residual) and dx .* M (for the product) in one pass (`hand`), _linear_backward / _MLP.backward pick the latter
    up when the gradient they receive IS that dx (`masked_grad`); any other gradient (y consumed by something else) is masked
    by one evlm_dropout call.  A plain object: autograd does not see it."""

    def __init__(self, p, device, shape):
        self.p = float(p)
        self.state = dropout_state(device)
        shape = tuple(int(v) for v in shape)
        rows = 1
        for v in shape[:-1]:
            rows *= v
        self.shape = (rows, shape[-1])             # the [rows, N] matrix the kernels index (flat: N % 8 == 0)
        self.call = _next_drop_call("hidden", shape, p)
        self._dx_ptr, self._masked = None, None

    def hand(self, dx, masked):
        self._dx_ptr, self._masked = dx.data_ptr(), masked

    def masked_grad(self, dy):
        ptr_, m = self._dx_ptr, self._masked
        self._dx_ptr, self._masked = None, None
        if m is not None and dy.data_ptr() == ptr_ and dy.numel() == m.numel() and dy.dtype == m.dtype:
            return m.view(dy.shape)
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        out = torch.empty_like(dyc)
        L.check(_lib().evlm_dropout(L.dt(dyc), L.ptr(dyc), None, dyc.numel(), self.p, L.ptr(self.state), self.call, L.ptr(out),
                                    L.stream()), "dropout")
        return out


class _Dropout(torch.autograd.Function):
    """y = x .* keep / (1 - p) (+ residual); backward is the same kernel on dy"""

    @staticmethod
    def forward(ctx, x, residual, p):
        L.require_cuda(x)
        xc = x if x.is_contiguous() else x.contiguous()
        rc = None
        if residual is not None:
            rc = residual if residual.is_contiguous() else residual.contiguous()
            if rc.shape != xc.shape or rc.dtype != xc.dtype:
                raise RuntimeError("dropout: residual must match the input")
        st = dropout_state(x.device)
        call = _next_drop_call("hidden", xc.shape, p)
        y = torch.empty_like(xc)
        L.check(_lib().evlm_dropout(L.dt(xc), L.ptr(xc), L.ptr(rc), xc.numel(), float(p), L.ptr(st), call, L.ptr(y), L.stream()),
                "dropout")
        ctx.meta = (float(p), st, call, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, st, call, has_res = ctx.meta
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dyc)
        L.check(_lib().evlm_dropout(L.dt(dyc), L.ptr(dyc), None, dyc.numel(), p, L.ptr(st), call, L.ptr(dx), L.stream()), "dropout")
        return dx, (dy if has_res else None), None


def dropout(x, p, training=True, residual=None):
    """nn.Dropout(p)(x) [+ residual]: identity (plus the residual) when p == 0 or not training"""
    if not training or not p or p <= 0.0:
        if residual is not None:
            raise RuntimeError("dropout(p = 0) with a residual: fuse the residual into the producing GEMM instead")
        return x
    return _Dropout.apply(x, residual, float(p))


# ---------------------------------------------------------------------------------------------------
# losses (device scalars)
# ---------------------------------------------------------------------------------------------------
def _padded_base(x, ldp=None):
    """if x is a [..., :n] view of a contiguous [..., ld] buffer (attention maps with 16-byte padded rows), return that
    buffer (a view, no copy); otherwise None."""
    if x.dim() < 2 or x.stride(-1) != 1 or x.is_contiguous():
        return None
    ld = x.stride(-2)
    if ld < x.shape[-1] or (ldp is not None and ld != ldp):
        return None
    shape = tuple(x.shape[:-1]) + (ld,)
    strides, acc = [], 1
    for s_ in reversed(shape):
        strides.append(acc)
        acc *= s_
    strides = tuple(reversed(strides))
    if tuple(x.stride()[:-1]) != strides[:-1]:
        return None
    return x.as_strided(shape, strides, x.storage_offset())


class _MSE(torch.autograd.Function):
    """weight * mean((a-b)^2); gradient flows to `a` only (b is the detached teacher).  Attention maps arrive as views
    of row-padded buffers whose padding is zero in both operands: the reduction then runs over the padded buffers
    (same sum) and is normalised by the true element count."""

    @staticmethod
    def forward(ctx, a, b, weight):
        L.require_cuda(a, b)
        n_true = a.numel()
        pa, pb = _padded_base(a), _padded_base(b)
        if pa is not None and pb is not None and pa.shape == pb.shape:
            ac, bc, padded = pa, pb, True
        else:
            ac = a if a.is_contiguous() else a.contiguous()
            bc = b if b.is_contiguous() else b.contiguous()
            padded = False
        w = weight * (ac.numel() / n_true)        # kernels divide by the element count they sweep
        out = zero_scalar(a.device)
        L.check(_lib().evlm_mse_fwd(L.dt(ac), L.ptr(ac), L.dt(bc), L.ptr(bc), ac.numel(), w, L.ptr(out), L.stream()), "mse_fwd")
        ctx.save_for_backward(ac, bc)
        ctx.meta = (w, padded, a.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        ac, bc = ctx.saved_tensors
        w, padded, shape = ctx.meta
        gc = g.to(torch.float32).contiguous()
        ga = torch.empty_like(ac)
        L.check(_lib().evlm_mse_bwd(L.dt(ac), L.ptr(ac), L.dt(bc), L.ptr(bc), ac.numel(), w, L.ptr(gc), L.ptr(ga),
                                    L.stream()), "mse_bwd")
        return (ga[..., :shape[-1]] if padded else ga.view(shape)), None, None


def mse(a, b, weight=1.0):
    return _MSE.apply(a, b.detach(), float(weight))


class _MSESum(torch.autograd.Function):
    """sum_i weight_i * mean((a_i - b_i)^2) in ONE scalar: the kernels accumulate into the same fp32 cell, so a KD term
    over L layers costs L reductions and nothing else (no per-layer zero-fill, no chain of scalar adds)."""

    @staticmethod
    def forward(ctx, weights, n, *tensors):
        out = zero_scalar(tensors[0].device)
        saved, meta = [], []
        for i in range(n):
            a, b = tensors[i], tensors[n + i]
            L.require_cuda(a, b)
            n_true = a.numel()
            pa, pb = _padded_base(a), _padded_base(b)
            if pa is not None and pb is not None and pa.shape == pb.shape:
                ac, bc, padded = pa, pb, True
            else:
                ac = a if a.is_contiguous() else a.contiguous()
                bc = b if b.is_contiguous() else b.contiguous()
                padded = False
            w = weights[i] * (ac.numel() / n_true)
            L.check(_lib().evlm_mse_fwd(L.dt(ac), L.ptr(ac), L.dt(bc), L.ptr(bc), ac.numel(), w, L.ptr(out), L.stream()),
                    "mse_fwd")
            saved += [ac, bc]
            meta.append((w, padded, a.shape))
        ctx.save_for_backward(*saved)
        ctx.meta = meta
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        gc = g.to(torch.float32).contiguous()
        grads = []
        for i, (w, padded, shape) in enumerate(ctx.meta):
            ac, bc = ctx.saved_tensors[2 * i], ctx.saved_tensors[2 * i + 1]
            if not ctx.needs_input_grad[2 + i]:
                grads.append(None)
                continue
            ga = torch.empty_like(ac)
            L.check(_lib().evlm_mse_bwd(L.dt(ac), L.ptr(ac), L.dt(bc), L.ptr(bc), ac.numel(), w, L.ptr(gc), L.ptr(ga),
                                        L.stream()), "mse_bwd")
            grads.append(ga[..., :shape[-1]] if padded else ga.view(shape))
        return (None, None) + tuple(grads) + (None,) * ctx.n


def mse_sum(pairs, weights=None):
    """sum over (a, b) pairs of weight * mse(a, b); gradient to the a's only"""
    n = len(pairs)
    ws = [1.0] * n if weights is None else [float(w) for w in weights]
    return _MSESum.apply(ws, n, *[a for a, _ in pairs], *[b.detach() for _, b in pairs])


def _f32_bits(x):
    import struct
    return struct.unpack("<i", struct.pack("<f", float(x)))[0] & 0xFFFFFFFF


class RowSlice:
    """rows [r0, r1) of the leading dimension of `full`, as a distillation operand (mse_terms): the batched forward keeps
    its hidden states / attention maps un-split, and every term names its rows - the backward then writes each term's
    gradient straight into ONE full-size (row-padded) buffer instead of autograd concatenating per-chunk gradients."""

    def __init__(self, full, r0, r1):
        self.full, self.r0, self.r1 = full, int(r0), int(r1)

    @property
    def shape(self):
        return (self.r1 - self.r0,) + tuple(self.full.shape[1:])

    @property
    def dtype(self):
        return self.full.dtype

    def tensor(self):
        return self.full[self.r0:self.r1]


class Ragged:
    """Real extents of a bucket-padded distillation operand (mse_terms): `ext` = device int32 words refilled per batch,
    `inner` = index of the word holding the real length of the operand's second-to-last dimension (the token axis of
    [.., L, d] states and of [.., H, L, Lk] maps), `outer` = index of the word holding the real length of its FIRST dimension
    (answer rows), either may be None.  The kernels skip what lies beyond (evlm_mse_grouped, ragged units); the mean's
    denominator stays the padded one - distill.ragged_correction rescales the term."""

    def __init__(self, ext, inner=None, outer=None):
        self.ext, self.inner, self.outer = ext, inner, outer


def _rag_words(rag, abuf):
    """words 8..11 of a grouped-MSE unit for operand buffer `abuf` ([outer.., inner items, unit])"""
    if rag is None or (rag.inner is None and rag.outer is None):
        return [0, 0, 0, 0]
    unit, items = int(abuf.shape[-1]), int(abuf.shape[-2])
    S = unit * items
    mult = 1
    for v in abuf.shape[1:-2]:
        mult *= int(v)
    if unit % 8 or rag.ext.dtype != torch.int32 or not rag.ext.is_cuda:
        raise RuntimeError("ragged distillation operand: rows must be multiples of 8 elements, extents device int32 words")
    si = 0xFF if rag.inner is None else int(rag.inner)
    so = 0xFF if rag.outer is None else int(rag.outer)
    return [S, unit, rag.ext.data_ptr(), si | (so << 8) | (mult << 16)]


def _mse_operand(x):
    """(buffer, padded?) the kernels sweep for x: the row-padded base of a map view, or x made contiguous"""
    base = _padded_base(x)
    if base is not None:
        return base, True
    return (x if x.is_contiguous() else x.contiguous()), False


class _MSETerms(torch.autograd.Function):
    """T distillation terms, term t = sum over its pairs of weight * mean((a - b)^2), in ONE launch per direction
    (evlm_mse_grouped).  spec: per pair (term, weight, index of its `a` tensor, r0, r1, Ragged | None); tensors: the distinct
    a's, then one b per pair.  Returns T scalars; each a receives ONE gradient buffer (rows no term covers are zero)."""

    @staticmethod
    def forward(ctx, spec, n_terms, n_a, *tensors):
        dev = tensors[0].device
        outs = [zero_scalar(dev) for _ in range(n_terms)]
        A = [_mse_operand(t) for t in tensors[:n_a]]
        saved, meta, rows, blocks = [a for a, _ in A], [], [], 0
        for i, (term, weight, ai, r0, r1, rag) in enumerate(spec):
            abuf, padded = A[ai]
            full = tensors[ai]
            b = tensors[n_a + i]
            L.require_cuda(abuf, b)
            bbuf, bpad = _mse_operand(b)
            if bpad != padded or tuple(bbuf.shape[1:]) != tuple(abuf.shape[1:]) or bbuf.shape[0] != r1 - r0:
                raise RuntimeError("mse_terms: operands of a pair must have the same (row-padded) layout")
            per_row = abuf[0].numel()
            ne, n_true = (r1 - r0) * per_row, (r1 - r0) * full[0].numel()
            w = weight * (ne / n_true)
            nb = max(1, min(256, (ne // 8 + 1023) // 1024))      # (each block ends in one atomic on the term word)
            rows += [abuf.data_ptr() + r0 * per_row * abuf.element_size(), bbuf.data_ptr(), ne, blocks, nb,
                     outs[term].data_ptr(), 0, _f32_bits(w / ne)] + _rag_words(rag, abuf)
            blocks += nb
            saved.append(bbuf)
            meta.append((term, w, ai, r0, r1, rag))
        table = _upload_table(rows, dev)
        L.check(_lib().evlm_mse_grouped(L.dt(saved[0].dtype), 0, L.ptr(table), len(spec), blocks, L.stream()), "mse_grouped")
        ctx.save_for_backward(*saved)
        ctx.meta, ctx.n_a = meta, n_a
        ctx.a_info = [(padded, tuple(t.shape)) for (_, padded), t in zip(A, tensors[:n_a])]
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n_a = ctx.n_a
        abufs, bbufs = ctx.saved_tensors[:n_a], ctx.saved_tensors[n_a:]
        live = [[] for _ in range(n_a)]                       # per a: the pairs whose term received a gradient
        for i, (term, w, ai, r0, r1, _rag) in enumerate(ctx.meta):
            if gs[term] is not None and ctx.needs_input_grad[3 + ai]:
                live[ai].append(i)
        grads, rows, blocks, units, keep = [None] * n_a, [], 0, 0, []
        for ai in range(n_a):
            if not live[ai]:
                continue
            abuf = abufs[ai]
            cover = sorted((ctx.meta[i][3], ctx.meta[i][4]) for i in live[ai])
            end, full_cover = 0, True
            for r0, r1 in cover:
                if r0 < end:
                    raise RuntimeError("mse_terms: terms overlap on rows of one operand")
                full_cover &= r0 == end
                end = r1
            ga = torch.empty_like(abuf) if (full_cover and end == abuf.shape[0]) else torch.zeros_like(abuf)
            per_row = abuf[0].numel()
            for i in live[ai]:
                term, w, _, r0, r1, rag = ctx.meta[i]
                g = gs[term]
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(torch.float32).contiguous()
                keep.append(g)
                ne = (r1 - r0) * per_row
                off = r0 * per_row * abuf.element_size()
                nb = max(1, min(2048, (ne // 8 + 255) // 256))
                rows += [abuf.data_ptr() + off, bbufs[i].data_ptr(), ne, blocks, nb, g.data_ptr(), ga.data_ptr() + off,
                         _f32_bits(2.0 * w / ne)] + _rag_words(rag, abuf)
                blocks += nb
                units += 1
            padded, shape = ctx.a_info[ai]
            grads[ai] = ga[..., :shape[-1]] if padded else ga.view(shape)
        if units:
            table = _upload_table(rows, abufs[0].device)
            L.check(_lib().evlm_mse_grouped(L.dt(abufs[0].dtype), 1, L.ptr(table), units, blocks, L.stream()), "mse_grouped")
        return (None, None, None) + tuple(grads) + (None,) * len(bbufs)


def mse_terms(terms):
    """terms: [(pairs, weights [, Ragged])] with pairs = [(a, b)]; returns the list of scalars  sum_i weights_i * mse(a_i, b_i),
    one per term (0 for an empty term).  An `a` may be a RowSlice of a larger tensor; a term's Ragged names the real extents
    of its (bucket-padded) operands.  All pairs of all terms run in one launch forward and one backward when they share a
    dtype (bf16 or f32); gradient flows to the a's only."""
    terms = [tuple(t) + (None,) * (3 - len(t)) for t in terms]
    flat = [(t, a, b.detach(), 1.0 if weights is None else float(weights[i]), rag)
            for t, (pairs, weights, rag) in enumerate(terms) for i, (a, b) in enumerate(pairs)]
    if not flat:
        return [0 for _ in terms]
    dts = {a.dtype for _, a, _, _, _ in flat} | {b.dtype for _, _, b, _, _ in flat}
    if len(dts) != 1 or next(iter(dts)) not in (torch.bfloat16, torch.float32):
        if any(rag is not None for _, _, _, _, rag in flat):
            raise RuntimeError("ragged distillation terms need operands of one dtype (bf16 or f32)")
        plain = lambda a: a.tensor() if isinstance(a, RowSlice) else a
        return [mse_sum([(plain(a), b) for a, b in pairs], weights) if pairs else 0 for pairs, weights, _ in terms]
    fulls, index, spec, Bt = [], {}, [], []
    for t, a, b, w, rag in flat:
        full, r0, r1 = (a.full, a.r0, a.r1) if isinstance(a, RowSlice) else (a, 0, a.shape[0])
        ai = index.setdefault(id(full), len(fulls))
        if ai == len(fulls):
            fulls.append(full)
        spec.append((t, w, ai, r0, r1, rag))
        Bt.append(b)
    outs = _MSETerms.apply(spec, len(terms), len(fulls), *fulls, *Bt)
    used = {t for t, _, _, _, _ in flat}
    return [outs[t] if t in used else 0 for t in range(len(terms))]


class GradJoin:
    """shared gradient buffer of ONE logits tensor that several loss ops read (join_grads): every loss backward that knows
    of it writes / adds its gradient into one padded buffer and hands autograd nothing; the join point's own backward -
    which autograd runs after ALL consumers, joined or not - passes the buffer (plus whatever other consumers produced) on"""

    def __init__(self, stream):
        self.buf, self.stream = None, stream


_NO_GRAD_JOIN = bool(os.environ.get("EVLM_NO_GRAD_JOIN"))      # (A/B switch: one gradient tensor per loss, autograd adds them)


class _JoinPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        slot = ctx.slot
        buf, slot.buf = slot.buf, None            # (a second backward over a retained graph starts over)
        if buf is None:
            return g, None
        d = buf[0]
        return (d if g is None else d + g), None


def join_grads(x):
    """an alias of `x` (logits that feed several loss ops - the hard-label CE and the distillation KL of the MLM / ITM heads)
    whose loss backward kernels sum into ONE padded buffer instead of each producing a tensor for autograd to add (an
    element-wise pass over [rows, classes]) and the producer's dX product re-padding the sum.  Pass the RETURNED object to
    every loss op; consumers that know nothing of the join still work (their gradients are added at the join point)."""
    if not (x.is_cuda and torch.is_grad_enabled() and x.requires_grad) or _NO_GRAD_JOIN:
        return x
    slot = GradJoin(torch.cuda.current_stream(x.device))
    y = _JoinPoint.apply(x, slot)
    y._evlm_join = slot
    return y


def _join_target(ctx, R, ldd, dtype):
    """the joined buffer [R, ldd] a loss backward should ADD into (None: none yet / not joinable) and whether this backward
    takes part in the join at all - only on the stream the logits were produced on: the join point's backward runs there,
    behind every kernel this stream was given before it"""
    j = ctx.join
    if j is None or torch.cuda.current_stream() != j.stream:
        return None, None
    if j.buf is not None and (j.buf[1].shape != (R, ldd) or j.buf[1].dtype != dtype):
        return None, None
    return (j.buf[1] if j.buf is not None else None), j


def _rows2d(x):
    Cn = x.shape[-1]
    x2 = x.reshape(-1, Cn) if x.dim() != 2 else x
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    return x2, x2.shape[0], Cn, x2.stride(0) if x2.shape[0] > 1 else Cn


class _CE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        L.require_cuda(logits, labels)
        x2, R, Cn, ld = _rows2d(logits)
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        out = zero_scalar(logits.device)
        lse = torch.empty(2 * R, dtype=torch.float32, device=logits.device)
        valid = torch.empty(1, dtype=torch.int32, device=logits.device)      # (written by the finish kernel)
        L.check(_lib().evlm_ce_fwd(L.dt(x2), L.ptr(x2), R, Cn, ld, L.ptr(lab), ignore_index, 1.0, L.ptr(lse), L.ptr(valid),
                                   L.ptr(out), L.stream()), "ce_fwd")
        ctx.save_for_backward(x2, lab, lse, valid)
        ctx.meta = (R, Cn, ld, ignore_index, logits.shape)
        ctx.join = getattr(logits, "_evlm_join", None)
        return out

    @staticmethod
    def backward(ctx, g):
        x2, lab, lse, valid = ctx.saved_tensors
        R, Cn, ld, ignore_index, shape = ctx.meta
        gc = g.to(torch.float32).contiguous()
        ldd = _pad8(Cn)
        acc, join = _join_target(ctx, R, ldd, x2.dtype)
        dl = acc if acc is not None else torch.empty((R, ldd), dtype=x2.dtype, device=x2.device)      # (padding columns: zeroed by the kernel)
        L.check(_lib().evlm_ce_bwd(L.dt(x2), L.ptr(x2), R, Cn, ld, L.ptr(lab), ignore_index, 1.0, L.ptr(lse), L.ptr(valid),
                                   L.ptr(gc), L.ptr(dl), ldd, int(acc is not None), L.stream()), "ce_bwd")
        if acc is not None:
            return None, None, None
        d = dl[:, :Cn] if ldd != Cn else dl
        if len(shape) > 2:
            d = d.unflatten(0, shape[:-1])
        if join is not None:                   # first joined writer: the join point hands (view, buffer) on
            join.buf = (d, dl)
            return None, None, None
        return d, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    return _CE.apply(logits, labels, ignore_index)


class _CEWeighted(torch.autograd.Function):
    """sum_r w[r] * CE(logits[r], labels[r]) over the rows with labels != ignore_index (no normalisation)"""

    @staticmethod
    def forward(ctx, logits, labels, row_weight, ignore_index):
        L.require_cuda(logits, labels, row_weight)
        x2, R, Cn, ld = _rows2d(logits)
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        rw = row_weight.detach().reshape(-1).to(torch.float32).contiguous()
        assert lab.numel() == R and rw.numel() == R
        out = zero_scalar(logits.device)
        lse = torch.empty(2 * R, dtype=torch.float32, device=logits.device)
        valid = torch.empty(1, dtype=torch.int32, device=logits.device)      # (written by the finish kernel)
        L.check(_lib().evlm_ce_weighted_fwd(L.dt(x2), L.ptr(x2), R, Cn, ld, L.ptr(lab), ignore_index, 1.0, L.ptr(rw),
                                            L.ptr(lse), L.ptr(valid), L.ptr(out), L.stream()), "ce_weighted_fwd")
        ctx.save_for_backward(x2, lab, lse, rw)
        ctx.meta = (R, Cn, ld, ignore_index, logits.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x2, lab, lse, rw = ctx.saved_tensors
        R, Cn, ld, ignore_index, shape = ctx.meta
        gc = g.to(torch.float32).contiguous()
        ldd = _pad8(Cn)
        dl = torch.empty((R, ldd), dtype=x2.dtype, device=x2.device)      # (padding columns: zeroed by the kernel)
        L.check(_lib().evlm_ce_weighted_bwd(L.dt(x2), L.ptr(x2), R, Cn, ld, L.ptr(lab), ignore_index, 1.0, L.ptr(rw),
                                            L.ptr(lse), L.ptr(gc), L.ptr(dl), ldd, 0, L.stream()), "ce_weighted_bwd")
        d = dl[:, :Cn] if ldd != Cn else dl
        if len(shape) > 2:
            d = d.unflatten(0, shape[:-1])
        return d, None, None, None


def cross_entropy_weighted_sum(logits, labels, row_weight, ignore_index=-100):
    """sum over rows of row_weight * CE (rows with ignore_index skipped): the VQA decoder's weighted answer loss"""
    return _CEWeighted.apply(logits, labels, row_weight, ignore_index)


class _KL(torch.autograd.Function):
    """soft_cross_entropy(predicts, targets): KLDiv(log_softmax(s*inv_t), softmax(t*inv_t), batchmean)"""

    @staticmethod
    def forward(ctx, s, t, inv_t, rag=None):
        L.require_cuda(s, t)
        s2, R, Cn, lds = _rows2d(s)
        t2, Rt, Ct, ldt = _rows2d(t)
        assert (R, Cn) == (Rt, Ct)
        out = zero_scalar(s.device)
        ls = torch.empty(R, dtype=torch.float32, device=s.device)
        lt = torch.empty(R, dtype=torch.float32, device=s.device)
        ctx.rag = None
        if rag is not None and (rag.inner is not None or rag.outer is not None):
            # ragged rows (bucket-padded batches): s is [outer, inner, C]; rows beyond the real extents take no part
            if s.dim() != 3:
                raise RuntimeError("ragged soft_cross_entropy: logits must be [rows, tokens, classes]")
            ctx.rag = (rag.ext, int(s.shape[1]), 0xFF if rag.inner is None else int(rag.inner),
                       0xFF if rag.outer is None else int(rag.outer))
            L.check(_lib().evlm_kl_fwd_rows(L.dt(s2), L.ptr(s2), lds, L.dt(t2), L.ptr(t2), ldt, R, Cn, inv_t, 1.0, L.ptr(ls),
                                            L.ptr(lt), L.ptr(out), L.ptr(ctx.rag[0]), ctx.rag[1], ctx.rag[2], ctx.rag[3],
                                            L.stream()), "kl_fwd_rows")
        else:
            L.check(_lib().evlm_kl_fwd(L.dt(s2), L.ptr(s2), lds, L.dt(t2), L.ptr(t2), ldt, R, Cn, inv_t, 1.0, L.ptr(ls), L.ptr(lt),
                                       L.ptr(out), L.stream()), "kl_fwd")
        ctx.save_for_backward(s2, t2, ls, lt)
        ctx.meta = (R, Cn, lds, ldt, inv_t, s.shape)
        ctx.join = getattr(s, "_evlm_join", None)
        return out

    @staticmethod
    def backward(ctx, g):
        s2, t2, ls, lt = ctx.saved_tensors
        R, Cn, lds, ldt, inv_t, shape = ctx.meta
        gc = g.to(torch.float32).contiguous()
        ldd = _pad8(Cn)
        acc, join = _join_target(ctx, R, ldd, s2.dtype)
        ds = acc if acc is not None else torch.empty((R, ldd), dtype=s2.dtype, device=s2.device)      # (padding columns: zeroed by the kernel)
        if ctx.rag is not None:
            L.check(_lib().evlm_kl_bwd_rows(L.dt(s2), L.ptr(s2), lds, L.dt(t2), L.ptr(t2), ldt, R, Cn, inv_t, 1.0, L.ptr(ls),
                                            L.ptr(lt), L.ptr(gc), L.ptr(ds), ldd, int(acc is not None), L.ptr(ctx.rag[0]),
                                            ctx.rag[1], ctx.rag[2], ctx.rag[3], L.stream()), "kl_bwd_rows")
        else:
            L.check(_lib().evlm_kl_bwd(L.dt(s2), L.ptr(s2), lds, L.dt(t2), L.ptr(t2), ldt, R, Cn, inv_t, 1.0, L.ptr(ls), L.ptr(lt),
                                       L.ptr(gc), L.ptr(ds), ldd, int(acc is not None), L.stream()), "kl_bwd")
        if acc is not None:
            return None, None, None, None
        d = ds[:, :Cn] if ldd != Cn else ds
        if len(shape) > 2:
            d = d.unflatten(0, shape[:-1])
        if join is not None:
            join.buf = (d, ds)
            return None, None, None, None
        return d, None, None, None


def soft_cross_entropy(predicts, targets, temperature=1.0, ragged=None):
    """ragged (ops.Ragged): the logits are [rows, tokens, classes] of a bucket-padded batch - rows / tokens beyond the real
    extents take no part (the mean keeps the padded count: the caller rescales)"""
    return _KL.apply(predicts, targets.detach(), 1.0 / float(temperature), ragged)


class _LogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L.require_cuda(x)
        xc = x if x.is_contiguous() else x.contiguous()
        Cn = xc.shape[-1]
        R = xc.numel() // Cn
        y = torch.empty_like(xc)
        L.check(_lib().evlm_log_softmax_fwd(L.dt(xc), L.ptr(xc), R, Cn, Cn, L.ptr(y), Cn, L.stream()), "log_softmax_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        Cn = y.shape[-1]
        R = y.numel() // Cn
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(y)
        L.check(_lib().evlm_log_softmax_bwd(L.dt(y), L.ptr(y), L.ptr(dyc), R, Cn, Cn, L.ptr(dx), L.stream()), "log_softmax_bwd")
        return dx


def log_softmax(x):
    return _LogSoftmax.apply(x)


# ---------------------------------------------------------------------------------------------------
# embeddings / gathers / activations
# ---------------------------------------------------------------------------------------------------
class _BertEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, word, pos, typ, pad_id, dtype):
        L.require_cuda(ids, word)
        B, Ln = ids.shape
        d = word.shape[1]
        idc = ids.to(torch.int64).contiguous()
        out = torch.empty((B, Ln, d), dtype=dtype, device=word.device)
        L.check(_lib().evlm_bert_embed_fwd(L.dt(dtype), L.ptr(idc), B, Ln, d, L.ptr(word.detach()), L.ptr(pos.detach()),
                                           L.ptr(typ.detach()), L.ptr(out), L.stream()), "bert_embed_fwd")
        ctx.save_for_backward(idc)
        ctx.params = (word, typ)
        ctx.meta = (word.shape, pos.shape, typ.shape, pad_id)
        return out

    @staticmethod
    def backward(ctx, de):
        (idc,) = ctx.saved_tensors
        wshape, pshape, tshape, pad_id = ctx.meta
        B, Ln = idc.shape
        d = wshape[1]
        dec = de if de.is_contiguous() else de.contiguous()
        word, typ = ctx.params
        wi, ti = _inplace(word), _inplace(typ)
        dword = word.grad if wi else torch.zeros(wshape, dtype=torch.float32, device=de.device)   # 94 MB at full size
        dpos = zeros_small(pshape, torch.float32, de.device)
        dtyp = typ.grad if ti else torch.zeros(tshape, dtype=torch.float32, device=de.device)
        L.check(_lib().evlm_bert_embed_bwd(L.dt(dec), L.ptr(idc), B, Ln, d, L.ptr(dec), pad_id, L.ptr(dword), L.ptr(dpos),
                                           L.ptr(dtyp), L.stream()), "bert_embed_bwd")
        return None, (None if wi else dword), dpos, (None if ti else dtyp), None, None


def bert_embed(ids, word, pos, typ, pad_id, dtype):
    return _BertEmbed.apply(ids, word, pos, typ, pad_id, dtype)


class _VitEmbed(torch.autograd.Function):
    """conv patch-embed (as im2row + GEMM) + class token + position embeddings"""

    @staticmethod
    def forward(ctx, image, patch_w, cls, pos, patch, dtype):
        L.require_cuda(image, patch_w)
        B, Cn, R, _ = image.shape
        d = patch_w.shape[0]
        G = R // patch
        Tn, K = G * G, Cn * patch * patch
        lib = _lib()
        img = image.to(torch.float32).contiguous()
        patches = torch.empty((B * Tn, K), dtype=dtype, device=image.device)
        L.check(lib.evlm_im2row(L.dt(dtype), L.ptr(img), B, Cn, R, patch, L.ptr(patches), L.stream()), "im2row")
        W = CACHE.get((patch_w,), dtype).view(d, K)
        tok = torch.empty((B * Tn, d), dtype=dtype, device=image.device)
        _gemm(L.dt(dtype), patches, W, tok, B * Tn, d, K, K, K, d)
        x = torch.empty((B, Tn + 1, d), dtype=dtype, device=image.device)
        L.check(lib.evlm_vit_embed_fwd(L.dt(dtype), L.ptr(tok), L.ptr(cls.detach()), L.ptr(pos.detach()), B, Tn, d, L.ptr(x),
                                       L.stream()), "vit_embed_fwd")
        ctx.save_for_backward(patches)
        ctx.params = (patch_w, cls, pos)
        ctx.meta = (B, Tn, d, K, patch_w.shape, pos.shape)
        return x

    @staticmethod
    def backward(ctx, dx):
        (patches,) = ctx.saved_tensors
        B, Tn, d, K, wshape, pshape = ctx.meta
        lib = _lib()
        dxc = dx if dx.is_contiguous() else dx.contiguous()
        patch_w, cls, pos = ctx.params
        dtok = torch.empty((B * Tn, d), dtype=dxc.dtype, device=dx.device)
        ci, pi = _inplace(cls), _inplace(pos)
        dcls = cls.grad if ci else torch.zeros(d, dtype=torch.float32, device=dx.device)
        dpos = pos.grad if pi else torch.zeros(pshape, dtype=torch.float32, device=dx.device)
        L.check(lib.evlm_vit_embed_bwd(L.dt(dxc), L.ptr(dxc), B, Tn, d, L.ptr(dtok), L.ptr(dcls), L.ptr(dpos), L.stream()),
                "vit_embed_bwd")
        if _inplace(patch_w) and L.dt(dxc) == L.BF16 and (B * Tn) % 64 == 0:
            _gemm(L.BF16, dtok, patches, patch_w.grad, d, K, B * Tn, d, K, K, p_trans=1, q_trans=1, c_f32=1, accumulate=1)
            dW = None
        else:
            dWb = torch.empty((d, K), dtype=torch.float32, device=dx.device)
            _gemm(L.dt(dxc), dtok, patches, dWb, d, K, B * Tn, d, K, K, p_trans=1, q_trans=1, c_f32=1)
            dW = dWb.view(wshape)
        return None, dW, (None if ci else dcls), (None if pi else dpos), None, None


def vit_embed(image, patch_w, cls, pos, patch, dtype):
    return _VitEmbed.apply(image, patch_w, cls, pos, patch, dtype)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos):
        L.require_cuda(x, pos)
        B, Ln, d = x.shape
        M = pos.shape[1]
        xc = x if x.is_contiguous() else x.contiguous()
        pc = pos.to(torch.int64).contiguous()
        out = torch.empty((B, M, d), dtype=x.dtype, device=x.device)
        L.check(_lib().evlm_gather_rows_fwd(L.dt(xc), L.ptr(xc), L.ptr(pc), B, Ln, M, d, L.ptr(out), L.stream()), "gather_fwd")
        ctx.save_for_backward(pc)
        ctx.meta = (B, Ln, M, d)
        return out

    @staticmethod
    def backward(ctx, dout):
        (pc,) = ctx.saved_tensors
        B, Ln, M, d = ctx.meta
        dc = dout if dout.is_contiguous() else dout.contiguous()
        dx = torch.empty((B, Ln, d), dtype=dout.dtype, device=dout.device)
        L.check(_lib().evlm_gather_rows_bwd(L.dt(dc), L.ptr(dc), L.ptr(pc), B, Ln, M, d, L.ptr(dx), L.stream()), "gather_bwd")
        return dx, None


def gather_rows(x, pos):
    return _GatherRows.apply(x, pos)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        L.require_cuda(x)
        xc = x if x.is_contiguous() else x.contiguous()
        y = torch.empty_like(xc)
        L.check(_lib().evlm_act_fwd(L.dt(xc), L.ptr(xc), xc.numel(), act, L.ptr(y), L.stream()), "act_fwd")
        ctx.save_for_backward(xc)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        d = xc.shape[-1]
        rows = xc.numel() // d
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(xc)
        L.check(_lib().evlm_gated_act_bwd(L.dt(xc), L.ptr(dyc), L.ptr(xc), None, rows, d, d, ctx.act, L.GATE_POST, L.ptr(dx),
                                          None, L.stream()), "act_bwd")
        return dx, None


def gelu(x):
    return _Act.apply(x, L.ACT_GELU)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        L.require_cuda(x)
        x2, rows, d, ld = _as2d(x)
        y = torch.empty((rows, d), dtype=x2.dtype, device=x.device)
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(_lib().evlm_l2norm_fwd(L.dt(x2), L.ptr(x2), rows, d, ld, eps, L.ptr(y), L.ptr(inv), L.stream()), "l2norm_fwd")
        ctx.save_for_backward(y, inv)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        rows, d = y.shape
        dyc = dy.reshape(rows, d)
        if not dyc.is_contiguous():
            dyc = dyc.contiguous()
        dx = torch.empty_like(y)
        L.check(_lib().evlm_l2norm_bwd(L.dt(y), L.ptr(y), L.ptr(dyc), L.ptr(inv), rows, d, L.ptr(dx), L.stream()), "l2norm_bwd")
        return dx.view(ctx.shape), None


def l2_normalize(x, eps=1e-12):
    return _L2Norm.apply(x, eps)


def cast(x, dtype):
    """differentiable dtype cast through the HIP cast kernel"""
    if x.dtype == dtype:
        return x
    return _Cast.apply(x, dtype)


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        L.require_cuda(x)
        xc = x if x.is_contiguous() else x.contiguous()
        y = torch.empty(xc.shape, dtype=dtype, device=x.device)
        L.check(_lib().evlm_cast(L.dt(xc), L.ptr(xc), L.dt(dtype), L.ptr(y), xc.numel(), L.stream()), "cast")
        ctx.src = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty(dyc.shape, dtype=ctx.src, device=dy.device)
        L.check(_lib().evlm_cast(L.dt(dyc), L.ptr(dyc), L.dt(ctx.src), L.ptr(dx), dyc.numel(), L.stream()), "cast")
        return dx, None


# ---------------------------------------------------------------------------------------------------
# L0 gates
# ---------------------------------------------------------------------------------------------------
class _L0Sample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loga, eps, temperature):
        L.require_cuda(loga, eps)
        la = loga.detach().contiguous()
        ep = eps.to(torch.float32).contiguous()
        z = torch.empty_like(la)
        L.check(_lib().evlm_l0_sample_fwd(L.ptr(la), L.ptr(ep), la.numel(), temperature, L.ptr(z), L.stream()), "l0_sample_fwd")
        ctx.save_for_backward(la, ep)
        ctx.t = temperature
        return z

    @staticmethod
    def backward(ctx, dz):
        la, ep = ctx.saved_tensors
        dzc = dz.to(torch.float32).contiguous()
        dl = torch.empty_like(la)
        L.check(_lib().evlm_l0_sample_bwd(L.ptr(la), L.ptr(ep), L.ptr(dzc), la.numel(), ctx.t, L.ptr(dl), L.stream()), "l0_sample_bwd")
        return dl, None, None


def l0_sample(loga, eps, temperature):
    return _L0Sample.apply(loga, eps, float(temperature))


_L0_TABLES = {}


class _L0Lagrangian(torch.autograd.Function):
    """(lagrangian, expected sparsity, target sparsity) of xvlm_l0_module.py:lagrangian_regularization in ONE launch
    (evlm_l0_lagrangian_fwd) - and one more for every gradient (gate log-alphas, lambda_1, lambda_2), accumulated in place
    where the parameters' gradients live in the optimiser's slabs."""

    @staticmethod
    def forward(ctx, cfg, lambda_1, lambda_2, steps, *logas):
        weights, logit_c, eps, prunable, target_sp, start_sp, warmup = cfg
        L.require_cuda(lambda_1, *logas)
        dev = logas[0].device
        if any(la.dtype != torch.float32 or not la.is_contiguous() for la in logas):
            raise RuntimeError("l0_lagrangian: the gate log-alphas must be contiguous float32 tensors")
        key = ("fwd",) + tuple((la.data_ptr(), la.numel(), float(w)) for la, w in zip(logas, weights))
        table = _L0_TABLES.get(key)
        if table is None:
            rows = []
            for la, w in zip(logas, weights):
                rows += [la.data_ptr(), la.numel(), _f32_bits(w)]
            if torch.cuda.is_current_stream_capturing():
                table = _upload_table(rows, dev)              # (lives in the capture's table arena; not cached)
            else:
                table = _L0_TABLES[key] = torch.tensor(rows, dtype=torch.int64).to(dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        steps_dev, steps_host = (steps.to(torch.float32).reshape(-1), 0.0) if torch.is_tensor(steps) else (None, float(steps))
        L.check(_lib().evlm_l0_lagrangian_fwd(L.ptr(table), len(logas), float(logit_c), float(eps), float(prunable),
                                              float(target_sp), float(start_sp), float(warmup), L.ptr(steps_dev), steps_host,
                                              L.ptr(lambda_1.detach()), L.ptr(lambda_2.detach()), L.ptr(out), L.stream()),
                "l0_lagrangian_fwd")
        ctx.save_for_backward(out, table)
        ctx.params = (lambda_1, lambda_2) + tuple(logas)
        ctx.cfg = (float(logit_c), float(eps), float(prunable))
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g, _ges=None, _gts=None):
        out, table = ctx.saved_tensors
        lambda_1, lambda_2 = ctx.params[:2]
        logas = ctx.params[2:]
        n = len(logas) + 4
        if g is None:
            return (None,) * n
        dev = out.device
        gs = g.to(torch.float32).reshape(-1)
        inplace = all(_inplace(p) for p in ctx.params)
        if inplace:
            grads = [p.grad for p in ctx.params]
        else:
            grads = [torch.zeros_like(p, dtype=torch.float32) for p in ctx.params]
        key = ("bwd",) + tuple(t.data_ptr() for t in grads[2:])
        gtable = _L0_TABLES.get(key) if inplace else None
        if gtable is None:
            rows = [t.data_ptr() for t in grads[2:]]
            if torch.cuda.is_current_stream_capturing():
                gtable = _upload_table(rows, dev)
            else:
                gtable = torch.tensor(rows, dtype=torch.int64).to(dev)
                if inplace:
                    _L0_TABLES[key] = gtable
        logit_c, eps, prunable = ctx.cfg
        L.check(_lib().evlm_l0_lagrangian_bwd(L.ptr(table), L.ptr(gtable), len(logas), max(la.numel() for la in logas), logit_c,
                                              eps, prunable, L.ptr(out), L.ptr(lambda_1.detach()), L.ptr(lambda_2.detach()),
                                              L.ptr(gs), L.ptr(grads[0]), L.ptr(grads[1]), L.stream()), "l0_lagrangian_bwd")
        if inplace:
            return (None,) * n
        return (None, grads[0], grads[1], None) + tuple(grads[2:])


def l0_lagrangian(logas, weights, logit_c, eps, prunable, target_sp, start_sp, warmup, steps, lambda_1, lambda_2):
    cfg = (tuple(float(w) for w in weights), logit_c, eps, prunable, target_sp, start_sp, warmup)
    return _L0Lagrangian.apply(cfg, lambda_1, lambda_2, steps, *logas)


def l0_deterministic(loga, temperature, magical_number):
    L.require_cuda(loga)
    la = loga.detach().to(torch.float32).contiguous()
    rows, size = la.shape
    z = torch.empty_like(la)
    L.check(_lib().evlm_l0_deterministic(L.ptr(la), rows, size, float(temperature), float(magical_number), L.ptr(z), L.stream()),
            "l0_deterministic")
    return z


def default_scale(dh):
    return 1.0 / math.sqrt(dh)
