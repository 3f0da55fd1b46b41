// Distillation / task loss reductions for gfx950 — all HBM-bound streaming kernels (16-byte loads,
// wave shuffles + one LDS step per block, one atomic per block).  Scalars stay on the device: the forward
// kernels accumulate weight*term into a device word, the backward kernels read dL/d(term) from a device word.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// MSE   (get_kd_loss, GeneralDistill.py:60-82)
// ---------------------------------------------------------------------------------------------
// Four 16-byte loads of each operand in flight per thread, and at most 2 workgroups per CU: the reduction ends in ONE f32
// atomic per workgroup on the same word, and those serialise at the memory side (~12 ns each): with the 2 048 workgroups
// this kernel first launched, their tail alone cost ~25 us (1.39 TB/s on the 25 MB average problem of a GD step).
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void mse_fwd_kernel(const TA* __restrict__ a, const TB* __restrict__ b, int64_t n,
                                                      float coef, float* __restrict__ loss) {
  __shared__ float red[16];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t nv = n >> 3, stride = (int64_t)gridDim.x * blockDim.x;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; c + 3 * stride < nv; c += 4 * stride) {
    float x0[8], y0[8], x1[8], y1[8], x2[8], y2[8], x3[8], y3[8];
    load8<TA>(a + c * 8, x0); load8<TB>(b + c * 8, y0);
    load8<TA>(a + (c + stride) * 8, x1); load8<TB>(b + (c + stride) * 8, y1);
    load8<TA>(a + (c + 2 * stride) * 8, x2); load8<TB>(b + (c + 2 * stride) * 8, y2);
    load8<TA>(a + (c + 3 * stride) * 8, x3); load8<TB>(b + (c + 3 * stride) * 8, y3);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d0 = x0[e] - y0[e], d1 = x1[e] - y1[e], d2 = x2[e] - y2[e], d3 = x3[e] - y3[e];
      s0 = fmaf(d0, d0, s0); s1 = fmaf(d1, d1, s1); s2 = fmaf(d2, d2, s2); s3 = fmaf(d3, d3, s3);
    }
  }
  for (; c < nv; c += stride) {
    float x[8], y[8];
    load8<TA>(a + c * 8, x);
    load8<TB>(b + c * 8, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = x[e] - y[e]; s0 = fmaf(d, d, s0); }
  }
  float s = (s0 + s1) + (s2 + s3);
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x) { const float d = to_f(a[i]) - to_f(b[i]); s = fmaf(d, d, s); }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(loss, s * coef);
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const TA* __restrict__ a, const TB* __restrict__ b, int64_t n,
                                                      float coef, const float* __restrict__ gout, TA* __restrict__ ga) {
  const float c2 = coef * gout[0];
  const int64_t nv = n >> 3;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nv; c += (int64_t)gridDim.x * blockDim.x) {
    float x[8], y[8];
    load8<TA>(a + c * 8, x);
    load8<TB>(b + c * 8, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = c2 * (x[e] - y[e]);
    store8<TA>(ga + c * 8, x);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x) ga[i] = from_f<TA>(c2 * (to_f(a[i]) - to_f(b[i])));
}

#define DISPATCH2(da, db, NAME, ...)                                                         \
  if (da == EVLM_F32 && db == EVLM_F32) { typedef float TA; typedef float TB; __VA_ARGS__ }  \
  else if (da == EVLM_BF16 && db == EVLM_BF16) { typedef bf16 TA; typedef bf16 TB; __VA_ARGS__ } \
  else if (da == EVLM_BF16 && db == EVLM_F32) { typedef bf16 TA; typedef float TB; __VA_ARGS__ } \
  else if (da == EVLM_F32 && db == EVLM_BF16) { typedef float TA; typedef bf16 TB; __VA_ARGS__ } \
  else return evlm_set_error("%s: bad dtypes", NAME);

extern "C" int evlm_mse_fwd(int dtype_a, const void* a, int dtype_b, const void* b, int64_t n, float weight,
                            float* loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(a && b && loss && n > 0, "evlm_mse_fwd: bad args");
  const int grid = imin(512, (n / 8 + 1023) / 1024 + 1);
  const float coef = weight / (float)n;
  DISPATCH2(dtype_a, dtype_b, "evlm_mse_fwd",
    hipLaunchKernelGGL((mse_fwd_kernel<TA, TB>), dim3(grid), dim3(256), 0, stream, (const TA*)a, (const TB*)b, n, coef, loss);)
  EVLM_LAUNCH_CHECK("evlm_mse_fwd");
  return 0;
}
extern "C" int evlm_mse_bwd(int dtype_a, const void* a, int dtype_b, const void* b, int64_t n, float weight,
                            const float* gout, void* grad_a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(a && b && gout && grad_a && n > 0, "evlm_mse_bwd: bad args");
  const int grid = imin(2048, (n / 8 + 255) / 256 + 1);
  const float coef = 2.0f * weight / (float)n;
  DISPATCH2(dtype_a, dtype_b, "evlm_mse_bwd",
    hipLaunchKernelGGL((mse_bwd_kernel<TA, TB>), dim3(grid), dim3(256), 0, stream, (const TA*)a, (const TB*)b, n, coef, gout, (TA*)grad_a);)
  EVLM_LAUNCH_CHECK("evlm_mse_bwd");
  return 0;
}

// ---- grouped MSE: every (student, teacher) pair of a distillation step in ONE launch per direction --------------------
// A GD step holds ~40 pairs (hidden states and attention maps of GeneralDistill.py:300-366), most of them a few MB: 40
// forward + 40 backward launches of 5-8 us each.  table: int64 [n][12] =
//   {a, b, n (elements), first block of the unit, blocks of the unit, loss word (fwd) / gout word (bwd), grad_a (bwd),
//    coef as f32 bits, S, unit, ext, slots}   with coef = weight / n (fwd), 2 weight / n (bwd); S = 0: a plain unit, else
//   a RAGGED one (below).
// A block finds its unit with one vector load + ballot per 64 units (first blocks ascend), then runs the single-pair loop
// on its share of the unit.
#define MSE_UNIT 12       // int64 words per unit
__device__ __forceinline__ int grouped_unit(const int64_t* __restrict__ table, int n_units) {
  const int lane = threadIdx.x & 63;
  int u = -1;
  for (int u0 = 0; u0 < n_units; u0 += 64) {
    const bool le = u0 + lane < n_units && table[MSE_UNIT * (int64_t)(u0 + lane) + 3] <= (int64_t)blockIdx.x;
    const unsigned long long m = __ballot(le);
    if (!m) break;
    u = u0 + __popcll(m) - 1;
  }
  return __builtin_amdgcn_readfirstlane(u);
}
// ABI 9 - RAGGED units (bucket-padded batches: text padded to a few lengths, answer rows to a few counts, so that a captured
// step replays whatever the batch's real extents are).  The operand is [outer][inner items][unit elements] with S = elements
// per outer block; only outer < ext[outer slot] * mult and inner item < ext[inner slot] take part, ext being DEVICE int32
// words refilled per batch (the captured launch never learns the real extents).  What lies beyond - the rows of padded
// tokens / padded answer rows, which hold arbitrary values - adds nothing to the sum and receives a zero gradient.  S and
// unit are multiples of 8, so a 16-byte chunk is valid or not as a whole.  (The mean's denominator stays the padded element
// count baked into coef; the caller rescales the term by padded / real - distill.ragged_correction.)
struct MseRag { int64_t S, lim_in, lim_out; };
__device__ __forceinline__ MseRag mse_rag(const int64_t* e) {
  MseRag r;
  r.S = e[8];
  if (r.S) {
    const int32_t* ext = reinterpret_cast<const int32_t*>(e[10]);
    const int si = (int)(e[11] & 0xFF), so = (int)((e[11] >> 8) & 0xFF);
    const int64_t mult = e[11] >> 16;
    r.lim_in = si == 0xFF ? r.S : (int64_t)ext[si] * e[9];
    r.lim_out = so == 0xFF ? (int64_t)1 << 62 : (int64_t)ext[so] * mult;
  }
  return r;
}
__device__ __forceinline__ bool mse_valid(const MseRag& r, int64_t off) {
  const int64_t o = off / r.S;
  return o < r.lim_out && off - o * r.S < r.lim_in;
}
template <typename T>
__global__ __launch_bounds__(256) void mse_grouped_fwd_kernel(const int64_t* __restrict__ table, int n_units) {
  __shared__ float red[16];
  const int64_t* e = table + MSE_UNIT * (int64_t)grouped_unit(table, n_units);
  const T* a = reinterpret_cast<const T*>(e[0]);
  const T* b = reinterpret_cast<const T*>(e[1]);
  const int64_t n = e[2], nv = n >> 3;
  const int blk = (int)(blockIdx.x - e[3]);
  const int64_t stride = e[4] * 256;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t c = blk * 256 + threadIdx.x;
  const MseRag rg = mse_rag(e);
  if (rg.S) {                                   // (unit-uniform) ragged unit: chunk by chunk, the invalid ones skipped
    for (; c < nv; c += stride) {
      if (!mse_valid(rg, c * 8)) continue;
      float x[8], y[8];
      load8<T>(a + c * 8, x);
      load8<T>(b + c * 8, y);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = x[k] - y[k]; s0 = fmaf(d, d, s0); }
    }
  }
  for (; c + 3 * stride < nv; c += 4 * stride) {
    float x0[8], y0[8], x1[8], y1[8], x2[8], y2[8], x3[8], y3[8];
    load8<T>(a + c * 8, x0); load8<T>(b + c * 8, y0);
    load8<T>(a + (c + stride) * 8, x1); load8<T>(b + (c + stride) * 8, y1);
    load8<T>(a + (c + 2 * stride) * 8, x2); load8<T>(b + (c + 2 * stride) * 8, y2);
    load8<T>(a + (c + 3 * stride) * 8, x3); load8<T>(b + (c + 3 * stride) * 8, y3);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d0 = x0[k] - y0[k], d1 = x1[k] - y1[k], d2 = x2[k] - y2[k], d3 = x3[k] - y3[k];
      s0 = fmaf(d0, d0, s0); s1 = fmaf(d1, d1, s1); s2 = fmaf(d2, d2, s2); s3 = fmaf(d3, d3, s3);
    }
  }
  for (; c < nv; c += stride) {
    float x[8], y[8];
    load8<T>(a + c * 8, x);
    load8<T>(b + c * 8, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float d = x[k] - y[k]; s0 = fmaf(d, d, s0); }
  }
  float s = (s0 + s1) + (s2 + s3);
  if (blk == 0 && !rg.S)
    for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += 256) { const float d = to_f(a[i]) - to_f(b[i]); s = fmaf(d, d, s); }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(reinterpret_cast<float*>(e[5]), s * __int_as_float((int)e[7]));
}
template <typename T>
__global__ __launch_bounds__(256) void mse_grouped_bwd_kernel(const int64_t* __restrict__ table, int n_units) {
  const int64_t* e = table + MSE_UNIT * (int64_t)grouped_unit(table, n_units);
  const T* a = reinterpret_cast<const T*>(e[0]);
  const T* b = reinterpret_cast<const T*>(e[1]);
  T* ga = reinterpret_cast<T*>(e[6]);
  const int64_t n = e[2], nv = n >> 3;
  const int blk = (int)(blockIdx.x - e[3]);
  const float c2 = __int_as_float((int)e[7]) * reinterpret_cast<const float*>(e[5])[0];
  const MseRag rg = mse_rag(e);
  for (int64_t c = blk * 256 + threadIdx.x; c < nv; c += e[4] * 256) {
    float x[8], y[8];
    if (rg.S && !mse_valid(rg, c * 8)) {        // beyond the real extents: a zero gradient (nothing is read)
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = 0.f;
      store8<T>(ga + c * 8, x);
      continue;
    }
    load8<T>(a + c * 8, x);
    load8<T>(b + c * 8, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = c2 * (x[k] - y[k]);
    store8<T>(ga + c * 8, x);
  }
  if (blk == 0 && !rg.S)
    for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += 256) ga[i] = from_f<T>(c2 * (to_f(a[i]) - to_f(b[i])));
}
extern "C" int evlm_mse_grouped(int dtype, int backward, const int64_t* table, int n_units, int total_blocks, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(table && n_units > 0 && total_blocks > 0, "evlm_mse_grouped: bad args");
  if (dtype == EVLM_BF16) {
    if (backward) hipLaunchKernelGGL(mse_grouped_bwd_kernel<bf16>, dim3(total_blocks), dim3(256), 0, stream, table, n_units);
    else hipLaunchKernelGGL(mse_grouped_fwd_kernel<bf16>, dim3(total_blocks), dim3(256), 0, stream, table, n_units);
  } else if (dtype == EVLM_F32) {
    if (backward) hipLaunchKernelGGL(mse_grouped_bwd_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, table, n_units);
    else hipLaunchKernelGGL(mse_grouped_fwd_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, table, n_units);
  } else return evlm_set_error("evlm_mse_grouped: bad dtype");
  EVLM_LAUNCH_CHECK("evlm_mse_grouped");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// row-wise log-sum-exp helper: one 256-thread block per row
// ---------------------------------------------------------------------------------------------
// one sweep over the row: 16-byte loads where the row is 16-byte aligned, running (max, sum exp) per thread merged at the
// end (online softmax), so a vocabulary-sized row is read once instead of twice
template <typename T>
__device__ __forceinline__ float row_lse(const T* __restrict__ x, int C, float mul, float* red) {
  float m = -INFINITY, s = 0.f;
  const bool vec = ((uintptr_t)x & 15) == 0;
  const int nv = vec ? (C >> 3) : 0;
  for (int c = threadIdx.x; c < nv; c += blockDim.x) {
    float v[8];
    load8<T>(x + c * 8, v);
    float mx = v[0] * mul;
#pragma unroll
    for (int e = 1; e < 8; ++e) mx = fmaxf(mx, v[e] * mul);
    if (mx > m) { s *= __expf(m - mx); m = mx; }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf(v[e] * mul - m);
  }
  for (int c = nv * 8 + threadIdx.x; c < C; c += blockDim.x) {
    const float v = to_f(x[c]) * mul;
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  const float M = block_max(m, red);
  s = block_sum(m == -INFINITY ? 0.f : s * __expf(m - M), red);
  return M + __logf(s);
}

// hard-label CE:  rowloss[r] = lse - logit[label]   (0 for ignored rows).  A label outside [0, C) that is not the ignore
// index (a corrupt masked_ids entry, a vocabulary-size mismatch) never indexes the row: its loss - and with it the step's
// total and, in ce_bwd_kernel, its gradient row - becomes NaN, where F.cross_entropy would trip a device assert.
template <typename T>
__global__ __launch_bounds__(256) void ce_row_kernel(const T* __restrict__ logits, int C, int ld,
                                                     const int64_t* __restrict__ labels, int ignore_index,
                                                     float* __restrict__ lse_out) {
  __shared__ float red[16];
  const int r = blockIdx.x, R = gridDim.x;
  const T* x = logits + (size_t)r * ld;
  const float lse = row_lse<T>(x, C, 1.0f, red);
  if (threadIdx.x == 0) {
    const int64_t lb = labels[r];
    lse_out[r] = lse;
    const bool bad = lb != ignore_index && (lb < 0 || lb >= C);
    lse_out[R + r] = (lb == ignore_index) ? 0.f : (bad ? __builtin_nanf("") : lse - to_f(x[bad ? 0 : lb]));
  }
}
// roww == nullptr: mean over the non-ignored rows; else the WEIGHTED SUM  sum_r roww[r] * rowloss[r]  (no normalisation)
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ rowloss, const int64_t* __restrict__ labels,
                                                        int R, int ignore_index, float weight, const float* __restrict__ roww,
                                                        int32_t* __restrict__ valid, float* __restrict__ loss) {
  __shared__ float red[16];
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    s += roww ? rowloss[r] * roww[r] : rowloss[r];
    n += (labels[r] != ignore_index) ? 1.f : 0.f;
  }
  s = block_sum(s, red);
  n = block_sum(n, red);
  if (threadIdx.x == 0) {
    valid[0] = (int32_t)n;
    atomicAdd(loss, roww ? weight * s : weight * s / n);   // mean form: all rows ignored -> NaN, as F.cross_entropy
  }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, int C, int ld,
                                                     const int64_t* __restrict__ labels, int ignore_index, float weight,
                                                     const float* __restrict__ lse, const int32_t* __restrict__ valid,
                                                     const float* __restrict__ gout, const float* __restrict__ roww,
                                                     T* __restrict__ dl, int ldd, int acc) {
  const int r = blockIdx.x;
  const int64_t lb = labels[r];
  const T* x = logits + (size_t)r * ld;
  T* d = dl + (size_t)r * ldd;
  // (the padding columns C .. ldd-1 of a gradient row are written here too - zeros: the caller allocates, never fills)
  // acc: dl already holds another loss's gradient of the same logits (ops.join_grads) - this one is added to it
  if (lb == ignore_index) {
    if (!acc) for (int c = threadIdx.x; c < ldd; c += blockDim.x) d[c] = from_f<T>(0.f);
    return;
  }
  float g = roww ? gout[0] * weight * roww[r] : gout[0] * weight / (float)valid[0];
  if (lb < 0 || lb >= C) g = __builtin_nanf("");          // out-of-range label: a loud gradient row (see ce_row_kernel)
  const float l = lse[r];
  const bool vec = (((uintptr_t)x | (uintptr_t)d) & 15) == 0;
  const int nv = vec ? (C >> 3) : 0;
  for (int c = threadIdx.x; c < nv; c += blockDim.x) {     // 16-byte pieces (the 30 522-wide MLM rows: 2-byte stores before)
    float v[8];
    load8<T>(x + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g * (__expf(v[e] - l) - ((c * 8 + e) == lb ? 1.f : 0.f));
    if (acc) {
      float o[8];
      load8<T>(d + c * 8, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += o[e];
    }
    store8<T>(d + c * 8, v);
  }
  for (int c = nv * 8 + threadIdx.x; c < (acc ? C : ldd); c += blockDim.x) {
    const float p = c < C ? __expf(to_f(x[c]) - l) : 0.f;
    d[c] = from_f<T>((c < C ? g * (p - (c == lb ? 1.f : 0.f)) : 0.f) + (acc ? to_f(d[c]) : 0.f));
  }
}

extern "C" int evlm_ce_fwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                           float weight, float* lse, int32_t* valid_count, float* loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(logits && labels && lse && valid_count && loss && R > 0 && C > 0, "evlm_ce_fwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_ce_fwd",
    hipLaunchKernelGGL((ce_row_kernel<T>), dim3(R), dim3(256), 0, stream, (const T*)logits, C, ld, labels, ignore_index, lse);)
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)(lse + R), labels, R, ignore_index, weight, (const float*)nullptr, valid_count, loss);
  EVLM_LAUNCH_CHECK("evlm_ce_fwd");
  return 0;
}
extern "C" int evlm_ce_weighted_fwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                                    float weight, const float* row_weight, float* lse, int32_t* valid_count, float* loss,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(logits && labels && row_weight && lse && valid_count && loss && R > 0 && C > 0, "evlm_ce_weighted_fwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_ce_weighted_fwd",
    hipLaunchKernelGGL((ce_row_kernel<T>), dim3(R), dim3(256), 0, stream, (const T*)logits, C, ld, labels, ignore_index, lse);)
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)(lse + R), labels, R, ignore_index, weight, row_weight, valid_count, loss);
  EVLM_LAUNCH_CHECK("evlm_ce_weighted_fwd");
  return 0;
}
extern "C" int evlm_ce_weighted_bwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                                    float weight, const float* row_weight, const float* lse, const float* gout,
                                    void* dlogits, int ldd, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(logits && labels && row_weight && lse && gout && dlogits, "evlm_ce_weighted_bwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_ce_weighted_bwd",
    hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3(R), dim3(256), 0, stream, (const T*)logits, C, ld, labels, ignore_index, weight, lse, (const int32_t*)nullptr, gout, row_weight, (T*)dlogits, ldd, accumulate);)
  EVLM_LAUNCH_CHECK("evlm_ce_weighted_bwd");
  return 0;
}
extern "C" int evlm_ce_bwd(int dtype, const void* logits, int R, int C, int ld, const int64_t* labels, int ignore_index,
                           float weight, const float* lse, const int32_t* valid_count, const float* gout,
                           void* dlogits, int ldd, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(logits && labels && lse && valid_count && gout && dlogits, "evlm_ce_bwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_ce_bwd",
    hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3(R), dim3(256), 0, stream, (const T*)logits, C, ld, labels, ignore_index, weight, lse, valid_count, gout, (const float*)nullptr, (T*)dlogits, ldd, accumulate);)
  EVLM_LAUNCH_CHECK("evlm_ce_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// KL(log_softmax(s*it) || softmax(t*it)), batchmean   (soft_cross_entropy, GeneralDistill.py:84-89)
// ---------------------------------------------------------------------------------------------
// ABI 9 - ragged rows (bucket-padded VQA batches: logits [answer rows][answer tokens][vocabulary]): row r = (o, i) with
// i = r % row_inner takes part iff i < ext[inner slot] and o < ext[outer slot]; the others add nothing / get a zero gradient
struct KlRows { const int32_t* ext; int inner, slots; };
__device__ __forceinline__ bool kl_row_valid(const KlRows& rr, int r) {
  if (!rr.ext) return true;
  const int si = rr.slots & 0xFF, so = (rr.slots >> 8) & 0xFF;
  const int o = r / rr.inner, i = r - o * rr.inner;
  return (si == 0xFF || i < rr.ext[si]) && (so == 0xFF || o < rr.ext[so]);
}
template <typename TS, typename TT>
__global__ __launch_bounds__(256) void kl_fwd_kernel(const TS* __restrict__ s, int lds_, const TT* __restrict__ t, int ldt,
                                                     int C, float it, float coef, float* __restrict__ lse_s,
                                                     float* __restrict__ lse_t, float* __restrict__ loss, KlRows rr) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  if (!kl_row_valid(rr, r)) {                    // (block-uniform)
    if (threadIdx.x == 0) { lse_s[r] = 0.f; lse_t[r] = 0.f; }
    return;
  }
  const TS* sr = s + (size_t)r * lds_;
  const TT* tr = t + (size_t)r * ldt;
  const float ls = row_lse<TS>(sr, C, it, red);
  const float lt = row_lse<TT>(tr, C, it, red);
  float acc = 0.f;
  const bool vec = (((uintptr_t)sr | (uintptr_t)tr) & 15) == 0;
  const int nv = vec ? (C >> 3) : 0;
  for (int c = threadIdx.x; c < nv; c += blockDim.x) {
    float sv[8], tv[8];
    load8<TS>(sr + c * 8, sv);
    load8<TT>(tr + c * 8, tv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float lpt = tv[e] * it - lt, lps = sv[e] * it - ls;
      acc += __expf(lpt) * (lpt - lps);
    }
  }
  for (int c = nv * 8 + threadIdx.x; c < C; c += blockDim.x) {
    const float lpt = to_f(tr[c]) * it - lt, lps = to_f(sr[c]) * it - ls;
    acc += __expf(lpt) * (lpt - lps);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    lse_s[r] = ls; lse_t[r] = lt;
    atomicAdd(loss, acc * coef);
  }
}
template <typename TS, typename TT>
__global__ __launch_bounds__(256) void kl_bwd_kernel(const TS* __restrict__ s, int lds_, const TT* __restrict__ t, int ldt,
                                                     int C, float it, float coef, const float* __restrict__ lse_s,
                                                     const float* __restrict__ lse_t, const float* __restrict__ gout,
                                                     TS* __restrict__ ds, int ldds, int acc, KlRows rr) {
  const int r = blockIdx.x;
  const TS* sr = s + (size_t)r * lds_;
  const TT* tr = t + (size_t)r * ldt;
  TS* dr = ds + (size_t)r * ldds;
  if (!kl_row_valid(rr, r)) {                    // a row beyond the real extents: zero gradient (left alone when accumulating)
    if (!acc)
      for (int c = threadIdx.x; c < ldds; c += blockDim.x) dr[c] = from_f<TS>(0.f);
    return;
  }
  const float g = gout[0] * coef * it, ls = lse_s[r], lt = lse_t[r];
  const bool vec = (((uintptr_t)sr | (uintptr_t)tr | (uintptr_t)dr) & 15) == 0;
  const int nv = vec ? (C >> 3) : 0;
  for (int c = threadIdx.x; c < nv; c += blockDim.x) {
    float sv[8], tv[8];
    load8<TS>(sr + c * 8, sv);
    load8<TT>(tr + c * 8, tv);
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[e] = g * (__expf(sv[e] * it - ls) - __expf(tv[e] * it - lt));
    if (acc) {                                                       // (added to another loss's gradient of the same logits)
      float o[8];
      load8<TS>(dr + c * 8, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) sv[e] += o[e];
    }
    store8<TS>(dr + c * 8, sv);
  }
  for (int c = nv * 8 + threadIdx.x; c < (acc ? C : ldds); c += blockDim.x)     // (+ the padding columns: zeros, never filled by the caller)
    dr[c] = from_f<TS>((c < C ? g * (__expf(to_f(sr[c]) * it - ls) - __expf(to_f(tr[c]) * it - lt)) : 0.f) + (acc ? to_f(dr[c]) : 0.f));
}
static int kl_fwd_impl(int dtype_s, const void* s, int lds_, int dtype_t, const void* t, int ldt, int R, int C,
                       float inv_t, float weight, float* lse_s, float* lse_t, float* loss, KlRows rr, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(s && t && lse_s && lse_t && loss && R > 0 && C > 0, "evlm_kl_fwd: bad args");
  const float coef = weight / (float)R;
  DISPATCH2(dtype_s, dtype_t, "evlm_kl_fwd",
    hipLaunchKernelGGL((kl_fwd_kernel<TA, TB>), dim3(R), dim3(256), 0, stream, (const TA*)s, lds_, (const TB*)t, ldt, C, inv_t, coef, lse_s, lse_t, loss, rr);)
  EVLM_LAUNCH_CHECK("evlm_kl_fwd");
  return 0;
}
extern "C" int evlm_kl_fwd(int dtype_s, const void* s, int lds_, int dtype_t, const void* t, int ldt, int R, int C,
                           float inv_t, float weight, float* lse_s, float* lse_t, float* loss, void* stream_) {
  return kl_fwd_impl(dtype_s, s, lds_, dtype_t, t, ldt, R, C, inv_t, weight, lse_s, lse_t, loss, KlRows{nullptr, 1, 0xFFFF}, stream_);
}
extern "C" int evlm_kl_fwd_rows(int dtype_s, const void* s, int lds_, int dtype_t, const void* t, int ldt, int R, int C,
                                float inv_t, float weight, float* lse_s, float* lse_t, float* loss, const int32_t* ext,
                                int row_inner, int inner_slot, int outer_slot, void* stream_) {
  EVLM_REQUIRE(ext && row_inner > 0 && R % row_inner == 0, "evlm_kl_fwd_rows: bad row extents");
  return kl_fwd_impl(dtype_s, s, lds_, dtype_t, t, ldt, R, C, inv_t, weight, lse_s, lse_t, loss,
                     KlRows{ext, row_inner, (inner_slot & 0xFF) | ((outer_slot & 0xFF) << 8)}, stream_);
}
static int kl_bwd_impl(int dtype_s, const void* s, int lds_, int dtype_t, const void* t, int ldt, int R, int C,
                       float inv_t, float weight, const float* lse_s, const float* lse_t, const float* gout,
                       void* ds, int ldds, int accumulate, KlRows rr, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(s && t && lse_s && lse_t && gout && ds, "evlm_kl_bwd: bad args");
  const float coef = weight / (float)R;
  DISPATCH2(dtype_s, dtype_t, "evlm_kl_bwd",
    hipLaunchKernelGGL((kl_bwd_kernel<TA, TB>), dim3(R), dim3(256), 0, stream, (const TA*)s, lds_, (const TB*)t, ldt, C, inv_t, coef, lse_s, lse_t, gout, (TA*)ds, ldds, accumulate, rr);)
  EVLM_LAUNCH_CHECK("evlm_kl_bwd");
  return 0;
}
extern "C" int evlm_kl_bwd(int dtype_s, const void* s, int lds_, int dtype_t, const void* t, int ldt, int R, int C,
                           float inv_t, float weight, const float* lse_s, const float* lse_t, const float* gout,
                           void* ds, int ldds, int accumulate, void* stream_) {
  return kl_bwd_impl(dtype_s, s, lds_, dtype_t, t, ldt, R, C, inv_t, weight, lse_s, lse_t, gout, ds, ldds, accumulate,
                     KlRows{nullptr, 1, 0xFFFF}, stream_);
}
extern "C" int evlm_kl_bwd_rows(int dtype_s, const void* s, int lds_, int dtype_t, const void* t, int ldt, int R, int C,
                                float inv_t, float weight, const float* lse_s, const float* lse_t, const float* gout,
                                void* ds, int ldds, int accumulate, const int32_t* ext, int row_inner, int inner_slot,
                                int outer_slot, void* stream_) {
  EVLM_REQUIRE(ext && row_inner > 0 && R % row_inner == 0, "evlm_kl_bwd_rows: bad row extents");
  return kl_bwd_impl(dtype_s, s, lds_, dtype_t, t, ldt, R, C, inv_t, weight, lse_s, lse_t, gout, ds, ldds, accumulate,
                     KlRows{ext, row_inner, (inner_slot & 0xFF) | ((outer_slot & 0xFF) << 8)}, stream_);
}

// ---------------------------------------------------------------------------------------------
// log_softmax rows
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void lsm_fwd_kernel(const T* __restrict__ x, int C, int ld, T* __restrict__ y, int ldy) {
  __shared__ float red[16];
  const T* xr = x + (size_t)blockIdx.x * ld;
  T* yr = y + (size_t)blockIdx.x * ldy;
  const float l = row_lse<T>(xr, C, 1.0f, red);
  for (int c = threadIdx.x; c < C; c += blockDim.x) yr[c] = from_f<T>(to_f(xr[c]) - l);
}
template <typename T>
__global__ __launch_bounds__(256) void lsm_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, int C, int ld,
                                                      T* __restrict__ dx) {
  __shared__ float red[16];
  const T* yr = y + (size_t)blockIdx.x * ld;
  const T* dr = dy + (size_t)blockIdx.x * ld;
  T* xr = dx + (size_t)blockIdx.x * ld;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) s += to_f(dr[c]);
  s = block_sum(s, red);
  for (int c = threadIdx.x; c < C; c += blockDim.x) xr[c] = from_f<T>(to_f(dr[c]) - __expf(to_f(yr[c])) * s);
}
extern "C" int evlm_log_softmax_fwd(int dtype, const void* x, int R, int C, int ld, void* y, int ldy, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && y && R > 0 && C > 0, "evlm_log_softmax_fwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_log_softmax_fwd",
    hipLaunchKernelGGL((lsm_fwd_kernel<T>), dim3(R), dim3(256), 0, stream, (const T*)x, C, ld, (T*)y, ldy);)
  EVLM_LAUNCH_CHECK("evlm_log_softmax_fwd");
  return 0;
}
extern "C" int evlm_log_softmax_bwd(int dtype, const void* y, const void* dy, int R, int C, int ld, void* dx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(y && dy && dx && R > 0 && C > 0, "evlm_log_softmax_bwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_log_softmax_bwd",
    hipLaunchKernelGGL((lsm_bwd_kernel<T>), dim3(R), dim3(256), 0, stream, (const T*)y, (const T*)dy, C, ld, (T*)dx);)
  EVLM_LAUNCH_CHECK("evlm_log_softmax_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// optimiser-side helpers
// ---------------------------------------------------------------------------------------------
// DETERMINISTIC with a workspace (ws[0]: arrival counter, kept zero between launches; ws[2 + b]: partial sum of block b):
// the last block to arrive sums the partials in a fixed order and adds the result to *out - the same bits on every rank of
// a data-parallel run, whose replicas stay bit-identical only if their clip factors are (f32 atomics in arrival order
// differ in the last bits from launch to launch).  ws == NULL: one f32 atomic per block (order-dependent rounding).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out,
                                                    float* __restrict__ ws) {
  __shared__ float red[16];
  __shared__ int last;
  float s = 0.f;
  const int64_t nv = n >> 2;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nv; c += (int64_t)gridDim.x * blockDim.x) {
    float v[4];
    Vec4<float>::load(x + c * 4, v);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += blockDim.x) s += x[i] * x[i];
  s = block_sum(s, red);
  if (!ws) {
    if (threadIdx.x == 0) atomicAdd(out, s);
    return;
  }
  if (threadIdx.x == 0) {
    // no fence (a device-scope release writes the L2 back: ~25 us over a thousand blocks): the partial goes out as a
    // RETURNING device-scope atomic - performed at the coherence point before its value comes back - and only then the
    // arrival count moves; the last arriver reads the partials with device-scope loads
    const float old = __hip_atomic_exchange(ws + 2 + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" :: "v"(old) : "memory");
    last = __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(ws), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  float t = 0.f;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x)      // device-scope loads (past this CU's caches), all in flight
    t += __hip_atomic_load(ws + 2 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = block_sum(t, red);
  if (threadIdx.x == 0) {
    *out += t;                                                // single writer; launches of a stream run in order
    atomicExch(reinterpret_cast<unsigned int*>(ws), 0u);
  }
}
extern "C" int evlm_sumsq(const float* x, int64_t n, float* out, float* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && out && n > 0, "evlm_sumsq: bad args");
  EVLM_REQUIRE(((uintptr_t)x) % 16 == 0, "evlm_sumsq: x must be 16-byte aligned");
  static const int cap = getenv("EVLM_SUMSQ_BLOCKS") ? imax(1, atoi(getenv("EVLM_SUMSQ_BLOCKS"))) : 1024;    // (tuning aid)
  const int grid = imin(imin(cap, EVLM_SUMSQ_WORKSPACE_FLOATS - 2), (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, stream, x, n, out, workspace);
  EVLM_LAUNCH_CHECK("evlm_sumsq");
  return 0;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, const float* __restrict__ gnorm_sq,
                                                    float max_norm, bf16* __restrict__ pb, const float* __restrict__ hyper) {
  if (hyper) { lr *= hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }   // per-step scalars from device memory (graph replay)
  float clip = 1.0f;
  if (gnorm_sq && max_norm > 0.f) clip = fminf(1.0f, max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f));
  const float step = lr * sqrtf(bc2) / bc1;   // transformers.AdamW: step_size = lr * sqrt(bc2) / bc1
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float pi = p[i] - step * mi / (sqrtf(vi) + eps);
    pi -= lr * wd * pi;                                               // decoupled decay AFTER the Adam update
    p[i] = pi;
    if (pb) pb[i] = (bf16)pi;
  }
}
extern "C" int evlm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float bias_c1, float bias_c2, const float* gnorm_sq,
                               float max_norm, void* p_bf16, const float* hyper, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(p && g && m && v && n > 0, "evlm_adamw_step: bad args");
  const int grid = imin(4096, (n + 255) / 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2, gnorm_sq, max_norm, (bf16*)p_bf16, hyper);
  EVLM_LAUNCH_CHECK("evlm_adamw_step");
  return 0;
}

// ---- ITC loss (reference efficient_models/xvlm.py:384-416) in one launch each way ---------------------------------------
// logits = I T^t / temp over the GATHERED batch, loss = (CE(logits, labels) + CE(logits^t, labels)) / 2 with labels the
// identity (idx None) or pos / pos.sum(1) for pos[i,j] = (idx_i == idx_j).  Written with torch ops that is ~35 launches
// of 4-7 us on [Bt, Bt] and [Bt, 256] tensors (two exact-fp32 GEMMs at 27 us each, two divisions by the temperature and
// their six-launch backward, two cross-entropies, the gradient products) - a launch-latency chain on the student's
// critical path.  Here one wave per batch row i forms BOTH its row of the logits (image i against every text) and its
// column (text i against every image) from the features in exact fp32, their log-sum-exps and label terms; the last
// workgroup to arrive sums the row terms in a fixed order.  The backward's wave i rebuilds d logits[i, :] and
// d logits[:, i] from the stored similarities and log-sum-exps and accumulates dI_i, dT_i and its share of d temp.
// stats f32 [4 Bt + 8]: lse_row | lse_col | row term (fwd) / row share of d temp (bwd) | positives per row | arrival
// counter (kept zero between launches) .
#define ITC_EMAX 256
template <typename T>
__global__ __launch_bounds__(256) void itc_fwd_kernel(const T* __restrict__ I, int ldi, const T* __restrict__ Tx, int ldt,
                                                      int Bt, int E, const float* __restrict__ temp,
                                                      const int64_t* __restrict__ group, float* __restrict__ sim, int lds,
                                                      float* __restrict__ stats, float* __restrict__ loss) {
  __shared__ __attribute__((aligned(16))) float feat[4][2][ITC_EMAX];
  __shared__ float red[16];
  __shared__ int last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 4 + w;
  const float it = 1.0f / temp[0];
  if (i < Bt) {
    for (int e = lane; e < E; e += 64) {
      feat[w][0][e] = to_f(I[(int64_t)i * ldi + e]);
      feat[w][1][e] = to_f(Tx[(int64_t)i * ldt + e]);
    }
  }
  __syncthreads();
  if (i < Bt) {
    const int64_t gi = group ? group[i] : 0;
    float mr = -INFINITY, zr = 0.f, mc = -INFINITY, zc = 0.f, lr = 0.f, lc = 0.f, cnt = 0.f;
    // (measured alternative, round 5: lanes along the feature dimension - coalesced row reads, one wave reduction per dot
    // product - 29.3 us against 18.5 us for this lane-per-column form at Bt = 64: twelve dependent shuffles per column)
    for (int c = 0; c < Bt; c += 64) {
      const int j = c + lane;
      if (j >= Bt) continue;
      const T* tj = Tx + (int64_t)j * ldt;
      const T* ij = I + (int64_t)j * ldi;
      float sr = 0.f, sc = 0.f;
      for (int e = 0; e < E; e += 8) {
        float a[8], b[8];
        load8<T>(tj + e, a);
        load8<T>(ij + e, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sr = fmaf(feat[w][0][e + k], a[k], sr);
          sc = fmaf(feat[w][1][e + k], b[k], sc);
        }
      }
      sim[(int64_t)i * lds + j] = sr;
      const float xr = sr * it, xc = sc * it;
      if (xr > mr) { zr = zr * __expf(mr - xr) + 1.f; mr = xr; } else zr += __expf(xr - mr);
      if (xc > mc) { zc = zc * __expf(mc - xc) + 1.f; mc = xc; } else zc += __expf(xc - mc);
      if (group ? group[j] == gi : j == i) { lr += xr; lc += xc; cnt += 1.f; }
    }
    const float Mr = wave_max(mr), Mc = wave_max(mc);
    zr = wave_sum(zr * __expf(mr - Mr));
    zc = wave_sum(zc * __expf(mc - Mc));
    lr = wave_sum(lr); lc = wave_sum(lc); cnt = wave_sum(cnt);
    const float lse_r = Mr + logf(zr), lse_c = Mc + logf(zc);
    if (lane == 0) {
      stats[i] = lse_r;
      stats[Bt + i] = lse_c;
      stats[3 * Bt + i] = cnt;
      const float term = (lse_r - lr / cnt) + (lse_c - lc / cnt);
      const float old = __hip_atomic_exchange(stats + 2 * Bt + i, term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(old) : "memory");
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(stats + 4 * Bt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  float t = 0.f;
  for (int b = threadIdx.x; b < Bt; b += blockDim.x) t += __hip_atomic_load(stats + 2 * Bt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = block_sum(t, red);
  if (threadIdx.x == 0) {
    loss[0] = t / (2.0f * (float)Bt);
    atomicExch(reinterpret_cast<unsigned int*>(stats + 4 * Bt), 0u);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void itc_bwd_kernel(const T* __restrict__ I, int ldi, const T* __restrict__ Tx, int ldt,
                                                      int Bt, int E, const float* __restrict__ temp,
                                                      const int64_t* __restrict__ group, const float* __restrict__ sim, int lds,
                                                      float* __restrict__ stats, const float* __restrict__ dloss,
                                                      T* __restrict__ dI, int lddi, T* __restrict__ dT, int lddt,
                                                      float* __restrict__ dtemp) {
  extern __shared__ float coef[];                 // [4 waves][2][Bt]: d logits[i, :] and d logits[:, i]
  __shared__ float red[16];
  __shared__ int last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 4 + w;
  const float it = 1.0f / temp[0], g = dloss[0] / (2.0f * (float)Bt);
  float* ca = coef + (size_t)w * 2 * Bt;
  float* cb = ca + Bt;
  if (i < Bt) {
    const int64_t gi = group ? group[i] : 0;
    const float lse_ri = stats[i], lse_ci = stats[Bt + i], ici = 1.0f / stats[3 * Bt + i];
    float dt = 0.f;
    for (int c = 0; c < Bt; c += 64) {
      const int j = c + lane;
      if (j >= Bt) continue;
      const float sij = sim[(int64_t)i * lds + j], sji = sim[(int64_t)j * lds + i];
      const bool pos = group ? group[j] == gi : j == i;
      const float Lij = pos ? ici : 0.f, Lji = pos ? 1.0f / stats[3 * Bt + j] : 0.f;
      // d loss / d logits[i, j]: row i's softmax over j (image-to-text) + column j's softmax over i (text-to-image)
      const float a = g * ((__expf(sij * it - lse_ri) - Lij) + (__expf(sij * it - stats[Bt + j]) - Lji));
      const float b = g * ((__expf(sji * it - stats[j]) - Lji) + (__expf(sji * it - lse_ci) - Lij));
      ca[j] = a; cb[j] = b;
      dt = fmaf(a, sij, dt);
    }
    dt = wave_sum(dt);
    if (lane == 0) {
      const float old = __hip_atomic_exchange(stats + 2 * Bt + i, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(old) : "memory");
    }
    // (the coefficients were written and are read by this wave only: LDS operations of one wave complete in order)
    __builtin_amdgcn_wave_barrier();
    const int e = lane * 4;
    if (e < E) {
      float aI[4] = {0.f, 0.f, 0.f, 0.f}, aT[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int j = 0; j < Bt; ++j) {
        float t[4], x[4];
        Vec4<T>::load(Tx + (int64_t)j * ldt + e, t);
        Vec4<T>::load(I + (int64_t)j * ldi + e, x);
        const float a = ca[j], b = cb[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) { aI[k] = fmaf(a, t[k], aI[k]); aT[k] = fmaf(b, x[k], aT[k]); }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { aI[k] *= it; aT[k] *= it; }
      Vec4<T>::store(dI + (int64_t)i * lddi + e, aI);
      Vec4<T>::store(dT + (int64_t)i * lddt + e, aT);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(stats + 4 * Bt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  float t = 0.f;
  for (int b = threadIdx.x; b < Bt; b += blockDim.x) t += __hip_atomic_load(stats + 2 * Bt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = block_sum(t, red);
  if (threadIdx.x == 0) {
    dtemp[0] = -t * it * it;                      // logits = sim / temp: d/d temp = -sim / temp^2
    atomicExch(reinterpret_cast<unsigned int*>(stats + 4 * Bt), 0u);
  }
}
static bool itc_args_ok(int dtype, const void* I, int ldi, const void* Tx, int ldt, int Bt, int E) {
  const int al = dtype == EVLM_BF16 ? 8 : 4;          // 16-byte rows
  return (dtype == EVLM_BF16 || dtype == EVLM_F32) && I && Tx && Bt > 0 && E > 0 && E <= ITC_EMAX && E % 8 == 0 && ldi >= E &&
         ldt >= E && ldi % al == 0 && ldt % al == 0 && ((uintptr_t)I) % 16 == 0 && ((uintptr_t)Tx) % 16 == 0;
}
extern "C" int evlm_itc_loss_fwd(int dtype, const void* I, int ldi, const void* Tx, int ldt, int Bt, int E, const float* temp,
                                 const int64_t* group, float* sim, int lds, float* stats, float* loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(itc_args_ok(dtype, I, ldi, Tx, ldt, Bt, E) && temp && sim && lds >= Bt && stats && loss,
               "evlm_itc_loss_fwd: bad args (bf16 / f32 features, E <= 256 and a multiple of 8, 16-byte aligned rows)");
  const dim3 grid(ceil_div(Bt, 4)), block(256);
  if (dtype == EVLM_BF16)
    hipLaunchKernelGGL(itc_fwd_kernel<bf16>, grid, block, 0, stream, (const bf16*)I, ldi, (const bf16*)Tx, ldt, Bt, E, temp, group, sim, lds, stats, loss);
  else
    hipLaunchKernelGGL(itc_fwd_kernel<float>, grid, block, 0, stream, (const float*)I, ldi, (const float*)Tx, ldt, Bt, E, temp, group, sim, lds, stats, loss);
  EVLM_LAUNCH_CHECK("evlm_itc_loss_fwd");
  return 0;
}
extern "C" int evlm_itc_loss_bwd(int dtype, const void* I, int ldi, const void* Tx, int ldt, int Bt, int E, const float* temp,
                                 const int64_t* group, const float* sim, int lds, float* stats, const float* dloss,
                                 void* dI, int lddi, void* dT, int lddt, float* dtemp, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(itc_args_ok(dtype, I, ldi, Tx, ldt, Bt, E) && temp && sim && lds >= Bt && stats && dloss && dI && dT && dtemp &&
               lddi >= E && lddt >= E && lddi % 4 == 0 && lddt % 4 == 0 && Bt <= 4096,
               "evlm_itc_loss_bwd: bad args (as evlm_itc_loss_fwd; gradient rows 4-element aligned, Bt <= 4096)");
  const dim3 grid(ceil_div(Bt, 4)), block(256);
  const size_t lds_bytes = (size_t)4 * 2 * Bt * sizeof(float);
  if (lds_bytes > 64 * 1024) {       // gathered batches beyond 2 048 rows (64 ranks x 64): up to 128 of the 160 KiB
    (void)hipFuncSetAttribute((const void*)itc_bwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    (void)hipFuncSetAttribute((const void*)itc_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  }
  if (dtype == EVLM_BF16)
    hipLaunchKernelGGL(itc_bwd_kernel<bf16>, grid, block, lds_bytes, stream, (const bf16*)I, ldi, (const bf16*)Tx, ldt, Bt, E, temp, group, sim, lds, stats, dloss, (bf16*)dI, lddi, (bf16*)dT, lddt, dtemp);
  else
    hipLaunchKernelGGL(itc_bwd_kernel<float>, grid, block, lds_bytes, stream, (const float*)I, ldi, (const float*)Tx, ldt, Bt, E, temp, group, sim, lds, stats, dloss, (float*)dI, lddi, (float*)dT, lddt, dtemp);
  EVLM_LAUNCH_CHECK("evlm_itc_loss_bwd");
  return 0;
}

// ---- ITM hard-negative sampling (reference efficient_models/xvlm.py:422-458) ----------------------------------------
// The reference draws, per text, one image from  softmax_i(sim[i,t] / temp) + 1e-5  with the positives zeroed, and per
// image one text likewise - 2B host-synchronising torch.multinomial(...).item() calls.  Here: ONE launch, one wave per
// draw.  Row r < B is the draw for text r (over column r of sim), row B + r the draw for image r (over row r of sim).
// The draw is the inverse CDF of those weights at u = Philox(seed, step, call_id, row) in (0,1): a categorical sample
// with exactly the reference's probabilities (not torch's stream of random numbers - DESIGN.md §2, known deviations).
__global__ __launch_bounds__(256) void sample_neg_kernel(const float* __restrict__ sim, int B, int ld,
                                                         const float* __restrict__ temp, const int64_t* __restrict__ group,
                                                         const int64_t* __restrict__ rng_state, uint32_t call,
                                                         int64_t* __restrict__ out, int64_t* __restrict__ sel4,
                                                         int32_t* __restrict__ img4) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * B) return;
  const bool t2i = row < B;
  const int r = t2i ? row : row - B;
  const int64_t stride = t2i ? ld : 1;
  const float* s = t2i ? sim + r : sim + (int64_t)r * ld;
  const float it = 1.0f / temp[0];
  const int64_t gr = group ? group[r] : 0;
  float m = -INFINITY;
  for (int j = lane; j < B; j += 64) m = fmaxf(m, s[j * stride] * it);
  m = wave_max(m);
  float z = 0.f;
  for (int j = lane; j < B; j += 64) z += __expf(s[j * stride] * it - m);
  z = wave_sum(z);
  const float iz = 1.0f / z;
  auto weight = [&](int j) -> float {
    if (j >= B) return 0.f;
    const bool pos = group ? group[j] == gr : j == r;
    return pos ? 0.f : __expf(s[j * stride] * it - m) * iz + 1e-5f;
  };
  float W = 0.f;
  for (int j = lane; j < B; j += 64) W += weight(j);
  W = wave_sum(W);
  const DropRng rng = drop_rng(rng_state, call, 0.f);
  uint32_t o[4];
  philox4(rng, (uint64_t)row, o);
  const float u = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f) * W;
  float acc = 0.f;
  int pick = -1, last = 0;
  for (int c = 0; c < B && pick < 0; c += 64) {
    const float w = weight(c + lane);
    float inc = w;                                            // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float v = __shfl_up(inc, d, 64);
      if (lane >= d) inc += v;
    }
    const unsigned long long hit = __ballot(acc + inc > u && w > 0.f);
    const unsigned long long any = __ballot(w > 0.f);
    if (any) last = c + 63 - __clzll(any);
    if (hit) pick = c + __ffsll(hit) - 1;
    acc += __shfl(inc, 63, 64);
  }
  if (pick < 0) pick = last;                                  // u rounded onto the total: the last admissible index
  if (lane == 0) {
    out[row] = pick;
    // the fusion pass's batch layout [pos B ; neg 2B (text x negative image | negative text x image) ; masked text B]:
    // sel4 = rows of the text pass's [text ; masked text] output, img4 = the image every fusion row attends to
    if (sel4 && img4) {
      if (t2i) {                       // row r: the image drawn for text r
        sel4[r] = r; sel4[B + r] = r; sel4[3 * B + r] = B + r;
        img4[r] = r; img4[B + r] = pick; img4[2 * B + r] = r; img4[3 * B + r] = r;
      } else {                         // row B + r: the text drawn for image r
        sel4[2 * B + r] = pick;
      }
    }
  }
}
extern "C" int evlm_sample_negatives(const float* sim, int B, int ld, const float* temp, const int64_t* group,
                                     const int64_t* rng_state, uint32_t call_id, int64_t* out, int64_t* sel4, int32_t* img4,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(sim && temp && rng_state && out && B > 0 && ld >= B && !sel4 == !img4, "evlm_sample_negatives: bad args");
  hipLaunchKernelGGL(sample_neg_kernel, dim3(ceil_div(2 * B, 4)), dim3(256), 0, stream, sim, B, ld, temp, group, rng_state,
                     call_id, out, sel4, img4);
  EVLM_LAUNCH_CHECK("evlm_sample_negatives");
  return 0;
}
