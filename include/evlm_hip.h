/*
 * evlm_hip.h — C ABI of libevlm_hip.so: the MI355X (gfx950) kernels behind the EfficientVLM
 * distillation hot path.
 *
 * The reference (swaggy-TN/EfficientVLM) has no native boundary: its hot path issues stock ATen ops
 * from Python nn.Modules.  This header is the boundary a maintainer would bind to
 * replace those op sequences; every entry point cites the reference lines whose arithmetic it replaces.
 * Paths are relative to the reference checkout.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never synchronises,
 *     never allocates; workspaces are caller-provided;
 *   - dtype: EVLM_F32 (exact fp32, fp32-input MFMA) or EVLM_BF16 (bf16 storage, fp32 accumulate);
 *   - returns 0 on success, non-zero on error; evlm_last_error() gives the message (thread-local).
 *   - row-major tensors; `ld*` = leading dimension in ELEMENTS.  For 16-byte vector access every
 *     ld must be a multiple of 8 (bf16) / 4 (f32) and padding columns inside ld must hold finite
 *     values (callers keep them zero).
 */
#ifndef EVLM_HIP_H
#define EVLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { EVLM_F32 = 0, EVLM_BF16 = 1 };
enum { EVLM_ACT_NONE = 0, EVLM_ACT_GELU = 1, EVLM_ACT_QUICK_GELU = 2 };
enum { EVLM_GATE_PRE_ACT = 0, EVLM_GATE_POST_ACT = 1 };

const char* evlm_last_error(void);
int evlm_abi_version(void);
/* name of the kernel that served the calling thread's last evlm_gemm (profiling aid; static string) */
const char* evlm_gemm_last_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   C[i,j] = epi( alpha * sum_k P(i,k) * Q(j,k) )      i<I, j<J
 *   P is stored [I][K] (p_trans=0, K contiguous) or [K][I] (p_trans=1); Q likewise with J.
 *   epi(v):  v += bias[j];  preact[i,j] = v (optional store);
 *            gate_pos==PRE : v *= gate[j];  v = act(v);   gate_pos==POST: v = act(v) * gate[j]
 *            dact != 0     : v *= act'(aux[i,j])           (backward of the activation, gate==NULL)
 *            v += residual[i,j]
 *   C is dtype or f32 (c_f32=1); preact/aux/residual have dtype `dtype` and leading dimension ldx.
 * Replaces: every nn.Linear on the path — CLIPAttention q/k/v/out_proj (efficient_models/eff_vit.py:134-199),
 *   CLIPMLP fc1 (*mlp_z) quick_gelu fc2 (eff_vit.py:214-220), BertSelfAttention query/key/value
 *   (eff_bert.py:277-296), BertSelfOutput.dense (:375), BertIntermediate gelu (:445-447) * mlp_z (:553-557),
 *   BertOutput.dense (:459), the tied MLM decoder (:744), patch-embed conv as im2row GEMM (eff_vit.py:444),
 *   build_mlp / vision_proj / text_proj (efficient_models/xvlm.py:77-83,230-231) — and their autograd
 *   backward products dX = dY*W (p_trans=0,q_trans=1) and dW = dY^T*X (p_trans=1,q_trans=1).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype;            /* element type of P, Q, preact, aux, residual */
  int c_f32;            /* 1: C is f32 regardless of dtype */
  int p_trans, q_trans;
  int I, J, K;
  int ldp, ldq, ldc, ldx;
  const void* P;
  const void* Q;
  void* C;
  const float* bias;    /* [J] f32 or NULL */
  const float* gate;    /* [J] f32 or NULL */
  void* preact;         /* [I,J] or NULL */
  const void* aux;      /* [I,J] or NULL (with dact) */
  const void* residual; /* [I,J] or NULL */
  float alpha;
  int act;              /* EVLM_ACT_* */
  int gate_pos;         /* EVLM_GATE_* */
  int dact;             /* EVLM_ACT_* : multiply by act'(aux) */
  int accumulate;       /* 1: C += result (f32 atomics; bare f32-output bf16 GEMMs only) — weight gradients summed
                           straight into the optimiser's gradient slab */
  float* psum;          /* optional [I] f32: psum[i] += sum_k P(i,k)  (bias gradient = column sums of dY, produced by the
                           dW GEMM itself from the dY tile it already holds: one extra MFMA against a ones fragment).
                           bf16, K % 64 == 0 only; ACCUMULATED with f32 atomics */
  void* sk_workspace;   /* optional, EVLM_GEMM_SK_WORKSPACE_BYTES, 16-byte aligned, ZERO when first handed over and owned by
                           the calls of ONE stream at a time: lets bf16 products whose 256x256 tiles fill only part of
                           the chip be cut along K as well (stream-K: partial accumulators + flags live here; the
                           kernel leaves the flags zero again).  NULL: never cut */
  float* dgate;         /* ABI 8, optional [J] f32, ACCUMULATED - the backward of an L0-GATED activation folded into the dX product
                           (CLIP MLP fc1 * mlp_z before quick_gelu, eff_vit.py:214-220; BERT gelu(dense) * mlp_z, eff_bert.py:
                           552-557): with gate [J], aux = the pre-activation rows h [I, J] and dact, the epilogue turns the
                           product dA = P Q^T into  dH = dA act'(h z) z  (gate_pos PRE)  /  dA act'(h) z  (POST)  and adds the
                           gate gradient  sum_rows dA act'(h z) h  /  sum_rows dA act(h)  to dgate.  bf16, no bias / residual /
                           preact, a product the 256-column ping-pong kernels take (K % 64 == 0, K >= 128, J, ldc, ldx % 8 == 0,
                           P not transposed); replaces a second pass over [I, J] (evlm_gated_act_bwd) */
  /* ABI 9 - hidden-state dropout in the residual epilogue (BertSelfOutput / BertOutput: LayerNorm(dropout(dense(h)) + input),
     eff_bert.py:372-381,456-462):  C = (alpha P Q^T + bias) .* keep / (1 - p) + residual,  the keep-mask of site `call_id`
     regenerated from rng_state exactly as evlm_dropout generates it for the [I, J] result (element index i * J + j; ONE Philox
     call per 8 consecutive columns).  Needs residual, J % 8 == 0, a bf16 / f32 output of `dtype` and no act / gate / dact.
     Replaces the separate evlm_dropout pass over [I, J] (one launch, a read and a write of the tensor per site). */
  float dropout_p;
  const int64_t* rng_state;
  uint32_t call_id;
} evlm_gemm_args;
#define EVLM_GEMM_SK_WORKSPACE_BYTES (4096 + 256 * 262144)

int evlm_gemm(const evlm_gemm_args* args, void* stream);

/* Grouped weight gradients: C_n += P_n^T Q_n for n problems that share the reduction length K (bf16 operands, f32 C),
 * ONE persistent launch per <= 40 problems with every 256x256 output tile owned by one workgroup (no split-K atomics).
 * P_n = dY stored [K][I] (row stride ldp), Q_n = X stored [K][J] (ldq), C_n [I][J] (ldc) is ACCUMULATED into;
 * psum_n (optional [I] f32) += column sums of dY (the bias gradient).  Replaces the weight branch of every nn.Linear
 * backward of a layer group (autograd's per-layer grad_weight = grad_output^T @ input): the caller defers them and
 * flushes once per reduction length (ops.flush_wgrad), which MI355X's HBM capacity makes free. */
typedef struct {
  const void* P; const void* Q; void* C; float* psum;
  int I, J, ldp, ldq, ldc;
  int assign;               /* 1: C = P^T Q - this step's FIRST contribution to C, which then needs no zero-fill before and
                               no read in the kernel (ignored - accumulated with atomics - when another problem of the same
                               call writes the same C);  0: C += P^T Q */
} evlm_wgrad_problem;
int evlm_wgrad_grouped(const evlm_wgrad_problem* problems, int n, int K, void* stream);

/* out[j] (+)= sum_i X[i,j]   (f32 out; bias gradients = column sums of dY).  out must be zeroed by the caller
 * unless it should accumulate.  Replaces the bias branch of Linear backward. */
int evlm_colsum(int dtype, const void* X, int I, int J, int ldx, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (two-pass mean/variance in f32, like ATen).
 * Replaces nn.LayerNorm: eff_vit.py layer_norm1/2, pre_layrnorm, post_layernorm (eps 1e-5);
 *   eff_bert.py BertEmbeddings.LayerNorm (:212), BertSelfOutput (:380), BertOutput (:461),
 *   BertPredictionHeadTransform (:725) (eps 1e-12); build_mlp LayerNorm (xvlm.py:80).
 * mean/rstd: [rows] f32, saved for backward (may be NULL in no-grad mode).
 * ---------------------------------------------------------------------------------------------- */
int evlm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, float eps,
                       int rows, int d, void* y, float* mean, float* rstd, void* stream);
/* dx = LN backward; dgamma/dbeta [d] f32 are ACCUMULATED (caller zeroes).  partials: optional f32 workspace of
 * evlm_layernorm_bwd_blocks(rows) * 2 * d floats (uninitialised is fine): the per-workgroup column sums go through it
 * with plain stores + a small second kernel instead of every workgroup's atomics landing on the same 2*d floats. */
int evlm_layernorm_bwd_blocks(int rows);
int evlm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, int rows, int d, void* dx, float* dgamma, float* dbeta, float* partials,
                       void* stream);
/* the same with  dx += addend  ([rows, d], dtype): the gradient arriving at x along the residual branch that bypasses this
 * LayerNorm (CLIPEncoderLayer: hidden = residual + f(layer_norm(residual)), eff_vit.py:250-266) is summed in the kernel,
 * replacing the element-wise add autograd would issue for the two uses of x.  addend2 (or NULL): a second such gradient -
 * of a distillation term that reads x itself (the hidden-state KD "tap", GeneralDistill.py:60-82). */
int evlm_layernorm_bwd_add(int dtype, const void* dy, const void* x, const void* addend, const void* addend2,
                           const float* gamma, const float* mean, const float* rstd, int rows, int d, void* dx,
                           float* dgamma, float* dbeta, float* partials, void* stream);
/* ABI 9 - the backward of  y = LayerNorm(dropout(dense(h)) + input)  (BertSelfOutput / BertOutput, eff_bert.py:372-381,
 * 456-462): dx = the LayerNorm's input gradient (what `input` receives) AND dx_dropped = dx .* keep / (1 - p) (what dense(h)
 * receives), the keep-mask of hidden-dropout site (rng_state, call_id) regenerated over [rows, d] exactly as the forward's
 * evlm_gemm_args.dropout_p / evlm_dropout generated it.  One pass instead of evlm_layernorm_bwd + evlm_dropout on dx. */
int evlm_layernorm_bwd_drop(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, int rows, int d, void* dx, void* dx_dropped, float dropout_p,
                            const int64_t* rng_state, uint32_t call_id, float* dgamma, float* dbeta, float* partials,
                            void* stream);
/* LayerNorm with the HIDDEN-STATE DISTILLATION TERM of its input fused in (ABI 7; GeneralDistill.py:60-82 get_kd_loss on
 * image_hidden_states: MSELoss(student state, teacher state), the state being the input of a pre-LN ViT block,
 * eff_vit.py:250).  Forward: kd_slots[evlm_layernorm_fwd_kd_slots() floats, zeroed by the caller] receive
 * kd_coef * sum (x - kd_teacher)^2 spread over 32 cache lines (the caller sums the slots; kd_coef = weight / numel).
 * Backward: dx += kd_k * kd_gout[0] * (x - kd_teacher) with kd_k = 2 * weight / numel and kd_gout the device scalar
 * gradient of the term; addend / addend2 as in evlm_layernorm_bwd_add (either may be NULL). */
int evlm_layernorm_fwd_kd_slots(void);
int evlm_layernorm_fwd_kd(int dtype, const void* x, const float* gamma, const float* beta, float eps, int rows, int d,
                          void* y, float* mean, float* rstd, const void* kd_teacher, float* kd_slots, float kd_coef,
                          void* stream);
int evlm_layernorm_bwd_kd(int dtype, const void* dy, const void* x, const void* addend, const void* addend2,
                          const float* gamma, const float* mean, const float* rstd, int rows, int d, void* dx,
                          float* dgamma, float* dbeta, float* partials, const void* kd_teacher, const float* kd_gout,
                          float kd_k, void* stream);
/* dgamma == dbeta == NULL (workspace given): the per-block column sums stay in `partials`; reduce the workspaces of many
 * LayerNorms at once with  table: device int64 [n][5] = {partials, blocks (evlm_layernorm_bwd_blocks(rows)), d, dgamma,
 * dbeta}  (accumulated), d_max = largest d in the table. */
int evlm_layernorm_bwd_reduce_grouped(const int64_t* table, int n_units, int d_max, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-head attention core with the probability map as an OUTPUT (the KD losses consume it).
 *   S = scale * Q K^T + mask[b, k] ;  P = softmax(S) ;  O = (P V) * head_gate[h]
 *   Q: [B, Lq, H, dh] with row stride ldq (so packed QKV buffers work), K/V: [Bkv, Lk, H, dh] (ldk, ldv);
 *   kv_index (int32 [B] or NULL) maps query batch b to its K/V batch row (hard-negative reuse);
 *   mask: additive f32 [B, Lk] or NULL (the reference's (1-m)*-10000, eff_bert.py:1012, eff_vit.py:339);
 *   P: [B, H, Lq, Lk] of p_dtype (EVLM_F32 or EVLM_BF16) with ROW STRIDE ldpr >= Lk (a multiple of 8; the kernels
 *   write zeros into the padding columns, so KD reductions may run over the padded buffer);  O: [B, Lq, H*dh] (ldo).
 * Replaces CLIPAttention bmm-softmax-bmm (eff_vit.py:144-195) and BertSelfAttention matmul /sqrt(d)
 *   +mask softmax matmul *= head_z (eff_bert.py:317-355), self- and cross-attention alike.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype, p_dtype;
  int B, H, Lq, Lk, dh;
  int ldq, ldk, ldv, ldo;       /* row strides in elements */
  int ldpr;                     /* row stride of P */
  const void* Q; const void* K; const void* V;
  const int32_t* kv_index;
  const float* mask;
  const float* head_gate;       /* [H] f32 or NULL */
  float scale;
  void* O; void* P;
  int causal;                   /* != 0: additionally add -10000 where key > query - the decoder's causal mask
                                   (get_extended_attention_mask is_decoder branch, eff_bert.py:975-996); needs Lq == Lk */
  float dropout_p;              /* attention_probs_dropout_prob (eff_bert.py:242,346): O = ((P .* keep / (1-p)) V) * gate, while
                                   the map written to P stays the un-dropped softmax (:338-361).  0 = off.  ABI 9: served by the
                                   bf16 MFMA kernels too (whole-row, shared-K/V grouped, streaming, map-writing), which regenerate
                                   the mask per (batch, head, query, 8 consecutive keys) in registers */
  const int64_t* rng_state;     /* device int64[2] {seed, step} (see evlm_dropout); required when dropout_p > 0 */
  uint32_t call_id;             /* identifies this dropout site; the backward call passes the same triple */
  /* fused attention-map distillation (GeneralDistill.py:63-69: MSELoss(student_att, teacher_att) * att.shape[-1]): with
     kd_teacher = the teacher's map [B, H, Lq, ldpr] (bf16, padding zero - the frozen teacher ran a batch ahead), the kernel
     adds  kd_weight * sum((P - P_t)^2) / (B H Lq Lk)  to *kd_loss while P is still in registers, instead of a separate
     reduction re-reading both maps from HBM.  bf16 MFMA path only (NULL elsewhere). */
  const void* kd_teacher;
  float* kd_loss;
  float kd_weight;
  /* optional [B, H, Lq] f32: per-row log2-sum-exp of the scaled, masked scores,  lse = m + log2(sum_k 2^(s_k - m))  with
     s_k = (scale * q.k + mask) * log2(e).  With it the backward RECOMPUTES the probabilities from Q and K in fp32
     (evlm_attn_bwd_args.lse) and P may be NULL: no [B, H, Lq, Lk] map is written or read back unless a caller wants it
     (the reference keeps softmax in fp32 under Apex O1; a bf16-stored P costs 13-36 % of the q / k gradients' norm).
     bf16 MFMA path, Lk <= 224 or 417 <= Lk <= 928 (ABI 9: with or without dropout - the kernels regenerate the keep-mask):
     evlm_attention_lse_supported(). */
  float* lse;
  int Bkv;                      /* with kv_index: number of K/V batch rows (0 = unknown).  When given, problems in which
                                   several short query batches share a K/V row run one workgroup per (K/V row, head) that
                                   stages K and V once for all of them */
  /* optional [B, H, Lq] f32, with lse AND kd_teacher: kd_rowdot[row] = sum_k P (P - P_t), the distillation term's share of
     the backward's row sum  delta = sum_k P dP  (evlm_attn_bwd_args.kd_rowdot: one-pass long-sequence backward) */
  float* kd_rowdot;
  /* ABI 8 - the fused map distillation against a teacher map that is NOT in memory (long key sequences: 225..928 keys,
     self-attention, no mask; kd_teacher NULL): the frozen teacher kept its projected queries / keys - kd_tq / kd_tk, bf16
     [B, L, H, dh] views of its packed QKV buffer with row stride kd_tld - and its row lse kd_tlse [B, H, L] (what its own
     forward call wrote through `lse`) instead of a [B, H, L, L] map; the kernel rebuilds P_t = 2^(scale q_t.k_t log2e - lse_t)
     per key tile in fp32 and forms the same term (and kd_rowdot).  113 MB + 1.8 MB per ViT layer at 577 tokens instead of
     a 517 MB map written once and read twice (Eff_Retrieval.py:141-147, Eff_VQA.py:140-146 distil these maps). */
  const void* kd_tq; const void* kd_tk; int kd_tld; const float* kd_tlse;
} evlm_attn_fwd_args;
int evlm_attention_fwd(const evlm_attn_fwd_args* a, void* stream);

/* backward: given dO and (optionally) an external gradient dP_ext on the probability map (from the
 * attention-map KD loss), produce dQ, dK, dV (+ dgate[H] accumulated, f32).  dS ([B,H,Lq,Lk], dtype) is a
 * caller-provided workspace; it may be NULL for bf16 self-attention problems (no kv_index, head dim 64; with dropout: the lse form
 * and Lq, Lk <= 64) with
 * Lq, Lk <= 224, which run in one pass with dS kept in LDS (any other problem then fails with an error, never a fault).  With kv_index (several query batches sharing one K/V row: the image tokens of the
 * positive, hard-negative and MLM fusion passes) dK/dV are [Bkv,Lk,H,dh]: the bf16 MFMA path sums the sharing query
 * batches inside one workgroup per K/V row (deterministic); the generic path ACCUMULATES with f32 atomics and then
 * expects dK/dV to be caller-zeroed f32 buffers. */
typedef struct {
  int dtype, p_dtype;
  int B, H, Lq, Lk, dh;
  int Bkv;                      /* K/V batch rows (== B without kv_index) */
  int ldq, ldk, ldv, ldo;
  int lddq, lddk, lddv;
  int ldpr;                     /* row stride of P, dP_ext and dS */
  const void* Q; const void* K; const void* V; const void* P;
  const void* dO; const void* dP_ext;
  const int32_t* kv_index;
  const float* head_gate;
  float scale;
  void* dS;
  void* dQ; void* dK; void* dV;
  float* dgate;
  float dropout_p;              /* as in the forward call: the keep-mask is regenerated, never stored */
  const int64_t* rng_state;
  uint32_t call_id;
  /* backward of the fused map distillation: dP += (*kd_gout) * kd_weight * 2 (P - P_t) / (B H Lq Lk), formed in
     registers from P_t - no dP_ext tensor is written or read for it (dP_ext may still carry other gradients) */
  const void* kd_teacher;
  const float* kd_gout;         /* device f32 word: dL/d(kd term) */
  float kd_weight;
  /* recomputing backward (set when the forward call wrote `lse`): P = 2^(s - lse) is rebuilt in fp32 registers from Q, K,
     the same additive `mask` ([B, Lk] f32 or NULL) and `causal` flag as in the forward call.  P may then be NULL; problems
     that take the two-kernel path (kv_index, Lq > 224) additionally need `P_ws` ([B, H, Lq, ldpr] of dtype, uninitialised)
     for the second kernel unless P is given. */
  const float* lse;
  const float* mask;
  int causal;
  void* P_ws;
  /* optional, recomputing form on 417 <= Lk <= 928 keys (384 x 384 / 480 x 480 images): O = the forward call's output
     ([B, Lq, H*dh], ldo) and - with kd_teacher - kd_rowdot = the forward call's kd_rowdot.  Without a dP_ext the row sums
     delta = sum_k P dP = dO . O + kd_weight' * kd_rowdot  are then taken from them and the kernel makes ONE pass over the
     keys instead of two (the flash-attention identity, extended by the fused distillation term).  Ignored elsewhere. */
  const void* O;
  const float* kd_rowdot;
  /* ABI 8: the teacher's map rebuilt in the kernel, as in the forward call (one-pass streaming kernel only) */
  const void* kd_tq; const void* kd_tk; int kd_tld; const float* kd_tlse;
} evlm_attn_bwd_args;
int evlm_attention_bwd(const evlm_attn_bwd_args* a, void* stream);
/* 1 when evlm_attention_fwd / _bwd serve (dtype, dh, Lk, dropout_p) through the lse / recompute form, else 0 */
int evlm_attention_lse_supported(int dtype, int dh, int Lk, float dropout_p);

/* many device-to-device copies in ONE launch.  table: DEVICE int64 [n_units][4] = {source pointer, destination pointer, bytes,
 * index of the unit's first workgroup}; bytes a multiple of 16, pointers 16-byte aligned, one workgroup per 64 KiB;
 * total_blocks = sum over units of ceil(bytes / 65536).  Plumbing of the teacher pipeline (the frozen teacher's outputs,
 * computed one batch ahead, are parked in persistent buffers: GeneralDistill.py:295-298 consumes them in the same step). */
int evlm_copy_grouped(const int64_t* table, int n_units, int total_blocks, void* stream);
/* ... up to 8 such copies whose addresses change from call to call (a step's input batch going into static buffers): the units
 * are passed by value with the launch - no device table.  src / dst / nbytes: HOST arrays of n entries (n <= 8), each unit a
 * multiple of 16 bytes and 16-byte aligned (ABI 8). */
int evlm_copy_few(const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused cross-attention FORWARD: K/V projection of the image tokens + Q K^T + softmax + P V (+ the map) in one launch.
 *   K = X Wk^T + bk, V = X Wv^T + bv  (Wkv: [2d, d] rows Wk then Wv, bias_kv [2d] likewise - the packed layout the
 *   two-launch path uses);  S = scale * Q K^T + mask[bq, k];  P = softmax(S);  O = (P V) * head_gate[h].
 *   X [Bimg, N, ldx] (N <= 224 tokens: 224 x 224 images), Q [Bq, Lq, ldq] ALREADY projected queries, kv_index (int32 [Bq] or
 *   NULL: identity, Bq == Bimg) maps each query batch to its image, mask additive f32 [Bq, N] or NULL,
 *   O [Bq, Lq, ldo], P [Bq, H, Lq, ldpr] or NULL.  bf16, head dim 64, even H, d = 64 H.
 * One workgroup per (image, head pair): the image tokens are read once per workgroup, K and V never reach HBM - which
 * is also why there is no backward for it: this entry point serves forwards that keep nothing (the frozen teacher's
 * fusion layers, inference).  Replaces BertSelfAttention.forward with encoder_hidden_states
 * (efficient_models/eff_bert.py:277-364: key / value Linear :284-287, matmul :317, /sqrt(d) :330, + mask :335, softmax :338,
 * matmul :352, *= head_z :354-355) for all text rows that attend to one image.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype;                    /* EVLM_BF16 */
  int Bimg, Bq, N, Lq, d, H, dh;
  int ldx, ldq, ldo, ldpr;
  const void* X; const void* Wkv; const float* bias_kv;
  const void* Q;
  const int32_t* kv_index;
  const float* mask;
  const float* head_gate;
  float scale;
  void* O; void* P;
} evlm_xattn_fused_args;
int evlm_xattn_fused_fwd(const evlm_xattn_fused_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dropout of hidden states:  y = x .* keep / (1 - p)  (+ residual),  keep ~ Bernoulli(1 - p) per element.
 * Replaces nn.Dropout(hidden_dropout_prob) in BertEmbeddings (eff_bert.py:180,214), BertSelfOutput (:372,:379) and
 * BertOutput (:456,:460); with `residual` it also forms the "+ input_tensor" the LayerNorm that follows consumes.
 * The keep-mask is a pure function of (rng_state = device int64[2] {seed, step}, call_id, element index) - Philox4x32-10,
 * counter {index / 8, call_id, step}, key seed, element (index % 8) = one 16-bit lane of the 128-bit output, kept iff the
 * lane >= round(p 2^16) (ABI 9; csrc/common.h) - so the backward pass is THE SAME CALL on dy (residual NULL) and no mask
 * tensor exists in HBM; a captured hipGraph draws new masks on every replay once `step` has been bumped on the device.
 * evlm_dropout_mask writes keep / (1 - p) as f32 for the flat indices 0 .. n-1 (tests feed it to the CPU oracle).  The
 * attention kernels index their mask by ((b*H + h)*Lq + q) * Lk8 + k with Lk8 = 8 ceil(Lk / 8) (rows padded to 8 keys: a lane
 * of the MFMA kernels owns 8 consecutive keys of a query per key-tile pair = one Philox call); the GEMM residual epilogue
 * (evlm_gemm_args.dropout_p) and evlm_layernorm_bwd_drop index [rows, d] results flat, as evlm_dropout does.
 * ---------------------------------------------------------------------------------------------- */
int evlm_dropout(int dtype, const void* x, const void* residual, int64_t n, float p, const int64_t* rng_state,
                 uint32_t call_id, void* y, void* stream);
int evlm_dropout_mask(int64_t n, float p, const int64_t* rng_state, uint32_t call_id, float* mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Distillation / task losses.  Scalars are f32 DEVICE words; forward ACCUMULATES weight*term into
 * *loss so one word can collect a whole loss mix without host syncs.  `gout` is a device f32 word
 * holding dL/d(term-sum) for the backward kernels.
 * ---------------------------------------------------------------------------------------------- */
/* *loss += weight * sum((a-b)^2) / n      — MSELoss() of get_kd_loss (GeneralDistill.py:60-82); the
 * attention-map variant passes weight = last-dim size (":69 * student_att.shape[-1]"). */
int evlm_mse_fwd(int dtype_a, const void* a, int dtype_b, const void* b, int64_t n, float weight,
                 float* loss, void* stream);
/* grad_a = (*gout) * weight * 2 (a-b) / n */
int evlm_mse_bwd(int dtype_a, const void* a, int dtype_b, const void* b, int64_t n, float weight,
                 const float* gout, void* grad_a, void* stream);
/* Every (a, b) pair of a distillation step in ONE launch per direction (a GD step has ~40, most of a few MB).
 * table: device int64 [n_units][12] = {a, b, n, first block of the unit, blocks of the unit,
 *   loss word (forward: += coef * sum (a-b)^2) | gout word (backward), grad_a (backward: = coef * gout * (a-b)),
 *   coef as f32 bits (forward: weight / n, backward: 2 weight / n), S, unit, ext, slots};  a and b share `dtype`, are
 * contiguous and 16-byte aligned; first blocks ascend from 0 and total_blocks is their sum.  backward != 0 selects the
 * gradient kernel.
 * ABI 9 - S != 0 marks a RAGGED unit, for bucket-padded batches (the reference pads a batch to its longest text,
 * Eff_Retrieval.py:97 / Eff_VQA.py:97-98, and feeds a variable number of answer rows, dataset/vqa_dataset.py:101-116; a
 * captured step wants a few fixed shapes): the operands are [outer][inner items][unit elements] with S elements per outer
 * block (S, unit multiples of 8) and only  outer < ext[outer slot] * mult,  inner item < ext[inner slot]  take part - ext
 * (word 10) = device int32 words holding the batch's REAL extents, slots (word 11) = inner slot | outer slot << 8 | mult <<
 * 16, slot 0xFF = "all".  The rest adds nothing to the sum and gets a zero gradient; coef keeps the padded element count
 * (the caller rescales the term by padded / real). */
int evlm_mse_grouped(int dtype, int backward, const int64_t* table, int n_units, int total_blocks, void* stream);

/* hard-label cross entropy, mean over rows with label != ignore_index (F.cross_entropy: MLM loss
 * eff_bert.py:1697-1699, ITM xvlm.py:484, ITC :399-400).  logits [R,C] (ld), labels int64.
 * lse [2R] f32 workspace (lse, then per-row losses) saved for backward; *loss += weight * mean;
 * valid_count: device int32 word (written).  The backward kernels write every column 0 .. ldd-1 of a gradient row (the
 * padding columns C .. ldd-1 as zeros): dlogits needs no initialisation.  accumulate != 0 (ABI 8; also evlm_kl_bwd): dlogits
 * already holds another loss's gradient of the same logits (same R, ldd) and this one is ADDED to it - the hard-label and the
 * distillation loss of the MLM / ITM logits then hand their producer ONE padded buffer (no element-wise add, no re-padding). */
int evlm_ce_fwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                float weight, float* lse, int32_t* valid_count, float* loss, void* stream);
int evlm_ce_bwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                float weight, const float* lse, const int32_t* valid_count, const float* gout,
                void* dlogits, int ldd, int accumulate, void* stream);
/* weighted-SUM form: *loss += weight * sum_r row_weight[r] * ce[r] (ignored rows contribute 0, no normalisation) - the
 * per-answer weighted next-token loss of the VQA decoder: BertLMHeadModel reduction='none' summed per sequence
 * (eff_bert.py:1419-1431) times `weights`, summed (efficient_models/model_generation.py:166-167).  row_weight [R] f32. */
int evlm_ce_weighted_fwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                         float weight, const float* row_weight, float* lse, int32_t* valid_count, float* loss, void* stream);
int evlm_ce_weighted_bwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                         float weight, const float* row_weight, const float* lse, const float* gout,
                         void* dlogits, int ldd, int accumulate, void* stream);

/* soft_cross_entropy (GeneralDistill.py:84-89): KLDiv(log_softmax(s*inv_t), softmax(t*inv_t), batchmean)
 * over R rows of C classes.  lse_s/lse_t [R] f32 saved for backward.  d s = (p_s - p_t) * inv_t / R; columns C .. ldds-1
 * of ds are written as zeros (no initialisation needed). */
int evlm_kl_fwd(int dtype_s, const void* s, int lds, int dtype_t, const void* t, int ldt, int R, int C,
                float inv_t, float weight, float* lse_s, float* lse_t, float* loss, void* stream);
int evlm_kl_bwd(int dtype_s, const void* s, int lds, int dtype_t, const void* t, int ldt, int R, int C,
                float inv_t, float weight, const float* lse_s, const float* lse_t, const float* gout,
                void* ds, int ldds, int accumulate, void* stream);
/* ABI 9 - the same over RAGGED rows (bucket-padded VQA batches, Eff_VQA.py:97-98 / dataset/vqa_dataset.py:101-116: logits
 * [answer rows][answer tokens][vocabulary] padded to a few shapes): row r = (o, i), i = r % row_inner, takes part iff
 * i < ext[inner_slot] and o < ext[outer_slot] (ext: device int32 words with the batch's real extents; slot 0xFF = all); the
 * other rows add nothing and get a zero gradient.  The mean keeps the padded row count R (the caller rescales the term). */
int evlm_kl_fwd_rows(int dtype_s, const void* s, int lds, int dtype_t, const void* t, int ldt, int R, int C,
                     float inv_t, float weight, float* lse_s, float* lse_t, float* loss, const int32_t* ext, int row_inner,
                     int inner_slot, int outer_slot, void* stream);
int evlm_kl_bwd_rows(int dtype_s, const void* s, int lds, int dtype_t, const void* t, int ldt, int R, int C,
                     float inv_t, float weight, const float* lse_s, const float* lse_t, const float* gout,
                     void* ds, int ldds, int accumulate, const int32_t* ext, int row_inner, int inner_slot, int outer_slot,
                     void* stream);

/* row-wise log_softmax (for the soft-label ITC branch, xvlm.py:411-414) and its backward */
int evlm_log_softmax_fwd(int dtype, const void* x, int R, int C, int ld, void* y, int ldy, void* stream);
int evlm_log_softmax_bwd(int dtype, const void* y, const void* dy, int R, int C, int ld, void* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embeddings and data movement
 * ---------------------------------------------------------------------------------------------- */
/* e[b,l,:] = word[ids[b,l]] + type0 + pos[l]   (BertEmbeddings before LayerNorm, eff_bert.py:203-211) */
int evlm_bert_embed_fwd(int dtype, const int64_t* ids, int B, int L, int d, const float* word,
                        const float* pos, const float* type0, void* out, void* stream);
/* scatter-add of de into dword (skipping pad id, nn.Embedding padding_idx eff_bert.py:171), dpos, dtype0 (f32, accumulated) */
int evlm_bert_embed_bwd(int dtype, const int64_t* ids, int B, int L, int d, const void* de, int pad_id,
                        float* dword, float* dpos, float* dtype0, void* stream);
/* patches[b*G*G + gy*G + gx, c*p*p + py*p + px] = image[b,c,gy*p+py,gx*p+px]  (Conv2d k=s=p as a GEMM, eff_vit.py:444) */
int evlm_im2row(int dtype, const float* image, int B, int C, int R, int p, void* patches, void* stream);
/* x[b,0,:] = cls + pos[0];  x[b,1+t,:] = tok[b,t,:] + pos[1+t]   (eff_vit.py:447-449) */
int evlm_vit_embed_fwd(int dtype, const void* tok, const float* cls, const float* pos, int B, int T, int d,
                       void* x, void* stream);
/* dtok = dx[:,1:,:]; dcls += sum_b dx[b,0]; dpos += sum_b dx[b]  (f32 accumulated) */
int evlm_vit_embed_bwd(int dtype, const void* dx, int B, int T, int d, void* dtok, float* dcls, float* dpos,
                       void* stream);
/* out[b,m,:] = x[b,pos[b,m],:]   (gather_seq_out_by_pos, eff_bert.py:1631-1632) and its scatter-add backward */
int evlm_gather_rows_fwd(int dtype, const void* x, const int64_t* pos, int B, int L, int M, int d, void* out,
                         void* stream);
int evlm_gather_rows_bwd(int dtype, const void* dout, const int64_t* pos, int B, int L, int M, int d, void* dx,
                         void* stream);
/* y = x / max(||x||_2, eps) per row (F.normalize(dim=-1) of get_features, efficient_models/xvlm.py:375-382);
 * x rows have stride ldx (the CLS-token slice), y is dense [rows,d]; inv_norm [rows] f32 saved for backward. */
int evlm_l2norm_fwd(int dtype, const void* x, int rows, int d, int ldx, float eps, void* y, float* inv_norm, void* stream);
int evlm_l2norm_bwd(int dtype, const void* y, const void* dy, const float* inv_norm, int rows, int d, void* dx, void* stream);
/* Grouped bf16 transposes in one launch: unit u = row-major [R][C] (R, C multiples of 8, 16-byte aligned pointers)
 * copied to [C][R].  `table` is a DEVICE array of 5 int64 per unit {src, dst, R, C, first tile index}, tiles are 64 x 64,
 * total_tiles = sum of ceil(R/64)*ceil(C/64).  Keeps W^T copies of the trainable nn.Linear weights current so that the
 * input gradient of every Linear (grad_input = grad_output @ weight, torch autograd) reads a K-contiguous operand. */
int evlm_transpose_grouped(const int64_t* table, int n_units, int total_tiles, void* stream);
/* dst = cast(src) between f32 and bf16; n elements.  (master f32 weights -> bf16 compute copies, grads back) */
int evlm_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
/* y[i,j] = a[i,j] * dact(h[i,j]) * gate-aware backward of the gated activations (used when gates are present):
 *   PRE  (ViT,  eff_vit.py:216-218):  a = act(h*z):  dh = da*act'(h z)*z ;  dz[j] += sum_i da*act'(h z)*h
 *   POST (BERT, eff_bert.py:553-557): a = act(h)*z:  dh = da*act'(h)*z   ;  dz[j] += sum_i da*act(h)      */
/* y = act(x), n elements (the GELU of build_mlp, xvlm.py:81, which follows a LayerNorm and so cannot ride a GEMM epilogue) */
int evlm_act_fwd(int dtype, const void* x, int64_t n, int act, void* y, void* stream);
/* gate == NULL (then dgate == NULL): plain activation backward dh = da * act'(h). */
int evlm_gated_act_bwd(int dtype, const void* da, const void* h, const float* gate, int I, int J, int ld,
                       int act, int gate_pos, void* dh, float* dgate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hard-concrete L0 gates (efficient_models/xvlm_l0_module.py)
 * ---------------------------------------------------------------------------------------------- */
/* z = hardtanh(sigmoid((log u - log(1-u) + loga)/T)*1.2 - 0.1, 0, 1)   (:180-182,246-250); all f32 */
int evlm_l0_sample_fwd(const float* loga, const float* eps, int64_t n, float temperature, float* z, void* stream);
int evlm_l0_sample_bwd(const float* loga, const float* eps, const float* dz, int64_t n, float temperature,
                       float* dloga, void* stream);
/* The Lagrangian sparsity term in one launch each way (ABI 7; xvlm_l0_module.py get_num_parameters_and_constraint :196-213 +
 * lagrangian_regularization :215-235; Eff_Retrieval.py:160-163, Eff_VQA.py:160-163 add it to the loss):
 *   expected size n = sum_t w_t * sum_i (1 - clamp(sigmoid(logit_c - loga_t[i]), eps, 1 - eps)),  es = 1 - n / prunable,
 *   ts = warmup > 0 ? (target_sp - start_sp) * min(1, steps / warmup) + start_sp : target_sp   (steps = *steps_dev if given),
 *   out[0] = lambda1 (es - ts) + lambda2 (es - ts)^2,  out[1] = es,  out[2] = ts.
 * table: device int64 [ntypes][3] = {loga pointer, element count, bit pattern of the f32 weight w_t (parameters_per_dim)}.
 * Backward ACCUMULATES into gtable[t] (device int64 [ntypes] of f32 gradient buffers) and dlambda1 / dlambda2 (may be NULL);
 * gout = device scalar gradient of out[0]; n_max = largest element count in the table. */
int evlm_l0_lagrangian_fwd(const int64_t* table, int ntypes, float logit_c, float eps, float prunable, float target_sp,
                           float start_sp, float warmup, const float* steps_dev, float steps_host, const float* lambda1,
                           const float* lambda2, float* out, void* stream);
int evlm_l0_lagrangian_bwd(const int64_t* table, const int64_t* gtable, int ntypes, int64_t n_max, float logit_c, float eps,
                           float prunable, const float* out_fwd, const float* lambda1, const float* lambda2, const float* gout,
                           float* dlambda1, float* dlambda2, void* stream);
/* eval masks (:253-271): per row of `size` gates, k = round(size - sum(1-cdf_qz(0))) smallest
 * sigmoid(loga/T*magic) are set to 0, the rest 1.  Selection order = ascending (value, index). */
int evlm_l0_deterministic(const float* loga, int rows, int size, float temperature, float magical_number,
                          float* z, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ITC loss (efficient_models/xvlm.py:384-416; ABI 8), one launch each way.  I / T: [Bt, E] image / text features of the
 * GATHERED batch (dtype bf16 or f32, leading dimensions ldi / ldt - the two may be column ranges of one gathered buffer;
 * E <= 256, a multiple of 8, rows 16-byte aligned), temp: device f32 word, group: optional int64 [Bt] (idx: equal entries
 * are positives of each other; NULL: the diagonal).
 *   fwd: sim f32 [Bt, lds] = I T^t (un-scaled: evlm_sample_negatives reads this rank's block of it),
 *        loss[0] = (CE(sim / temp, labels) + CE(sim^t / temp, labels)) / 2, labels = pos / pos.sum(1);
 *        stats f32 [4 Bt + 8] (row / column log-sum-exps, work words) - its word [4 Bt] must be ZERO on entry and is zero
 *        again on exit (arrival counter of the fixed-order final sum).
 *   bwd: dI, dT ([Bt, E], dtype, written) = dloss[0] * d loss / d I, d T; dtemp[0] (written) = dloss[0] * d loss / d temp.
 * ---------------------------------------------------------------------------------------------- */
int evlm_itc_loss_fwd(int dtype, const void* I, int ldi, const void* T, int ldt, int Bt, int E, const float* temp,
                      const int64_t* group, float* sim, int lds, float* stats, float* loss, void* stream);
int evlm_itc_loss_bwd(int dtype, const void* I, int ldi, const void* T, int ldt, int Bt, int E, const float* temp,
                      const int64_t* group, const float* sim, int lds, float* stats, const float* dloss,
                      void* dI, int lddi, void* dT, int lddt, float* dtemp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ITM hard negatives (efficient_models/xvlm.py:422-458 - there 2B host-synchronising torch.multinomial calls).
 * sim: f32 [B, ld] image x text similarities (un-scaled), temp: device f32 word, group: optional int64 [B] ids whose
 * equal entries are positives of each other (NULL: only the diagonal).  out int64 [2B]: out[t] = image drawn for text t
 * from softmax_i(sim[i,t]/temp) + 1e-5 with positives zeroed, out[B+i] = text drawn for image i likewise.  The draw is
 * the inverse CDF at a Philox(rng_state {seed, step}, call_id, row) uniform: one launch, no host sync, replayable.
 * ---------------------------------------------------------------------------------------------- */
int evlm_sample_negatives(const float* sim, int B, int ld, const float* temp, const int64_t* group,
                          const int64_t* rng_state, uint32_t call_id, int64_t* out, int64_t* sel4, int32_t* img4, void* stream);
/* sel4 / img4 (ABI 8, both or neither): the batched fusion pass's layout [pos B ; text x negative image B ; negative text x
 * image B ; masked text B] written by the same launch - sel4 int64 [4B] = rows of the text pass's [text ; masked text] output
 * (r, r, out[B + r], B + r), img4 int32 [4B] = the image each fusion row attends to (r, out[r], r, r).
 *
 * out[k] = x[sel[k]] over whole samples of row_bytes bytes (a multiple of 16; any dtype), n <= 65535 - and its deterministic
 * backward dx[r] = sum_{k : sel[k] == r} dy[k] in ascending k (zero rows where nothing selects r); row_elems % 8 == 0. */
int evlm_select_batches_fwd(const void* x, const int64_t* sel, int n, int64_t row_bytes, void* out, void* stream);
int evlm_select_batches_bwd(int dtype, const void* dy, const int64_t* sel, int n, int rows, int64_t row_elems, void* dx,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser-side helpers (next-tier row §8f-1; kept minimal here)
 * ---------------------------------------------------------------------------------------------- */
/* *out += sum(x^2) over n f32 values (global grad norm, apex_ddp_accelerator.py:99-102).  workspace (ABI 5):
 * EVLM_SUMSQ_WORKSPACE_FLOATS f32 words of device memory, zero before the first use (the kernel leaves word 0 zero), not
 * shared by launches that may overlap - the sum is then formed in a FIXED order: bit-identical from launch to launch and
 * from rank to rank, which data-parallel replicas need to stay bit-identical (their clip factors derive from it).  NULL:
 * one f32 atomic per block, order-dependent in the last bits. */
#define EVLM_SUMSQ_WORKSPACE_FLOATS 2050
int evlm_sumsq(const float* x, int64_t n, float* out, float* workspace, void* stream);
/* HF-AdamW step (optim.py:67, transformers AdamW: Adam update then p -= lr*wd*p), with the gradient
 * pre-scaled by min(1, max_norm/ (sqrt(*gnorm_sq)+1e-6)) (clip_grad_norm_, apex_ddp_accelerator.py:99-102); optionally
 * refreshes a bf16 copy.  hyper (device f32[3] or NULL) = {lr multiplier, bias_c1, bias_c2}: when given it overrides the
 * host bias corrections and scales lr, so a captured hipGraph can be replayed with per-step schedules. */
int evlm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float bias_c1, float bias_c2, const float* gnorm_sq,
                    float max_norm, void* p_bf16, const float* hyper, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVLM_HIP_H */
