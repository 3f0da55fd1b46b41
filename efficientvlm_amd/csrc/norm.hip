// LayerNorm forward / backward for gfx950.  HBM-bound: one 64-lane wave per row, 16-byte vector loads,
// two-pass mean / variance in fp32 (matches ATen's numerics; BERT uses eps = 1e-12 so the variance must be exact).
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int rows, int d,
                                                     T* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (size_t)row * d;
  const int nchunk = d >> 3;
  float s = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    float v[8];
    load8<T>(xr + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
  }
  for (int i = (nchunk << 3) + lane; i < d; i += 64) s += to_f(xr[i]);
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    float v[8];
    load8<T>(xr + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = v[e] - mean; q += t * t; }
  }
  for (int i = (nchunk << 3) + lane; i < d; i += 64) { const float t = to_f(xr[i]) - mean; q += t * t; }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  T* yr = y + (size_t)row * d;
  for (int c = lane; c < nchunk; c += 64) {
    float v[8], gm[8], bt[8];
    load8<T>(xr + c * 8, v);
    load8<float>(gamma + c * 8, gm);
    load8<float>(beta + c * 8, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
    store8<T>(yr + c * 8, v);
  }
  for (int i = (nchunk << 3) + lane; i < d; i += 64) yr[i] = from_f<T>((to_f(xr[i]) - mean) * rstd * gamma[i] + beta[i]);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// d % 8 == 0 and d <= 512 * DCH: the row is read ONCE and kept in registers (chunk c of lane l: l + 64 c), gamma / beta
// chunks are loaded once per wave, and a wave walks rows  first, first + stride, ...  - same sums in the same order as
// ln_fwd_kernel (bit-identical output), one pass over memory instead of three dependent ones (-10 % on [12608, 768]).
// KD (round 5): the hidden-state distillation term of a block INPUT (GeneralDistill.py:60-82: MSE(student state, teacher
// state)) rides on this kernel - the student's row is in registers, the teacher's (resident: the pipelined teacher ran a
// batch ahead) is read once: kd_slots[(block & 31) * 32] += kd_coef * sum (x - t)^2, one atomic per workgroup, 32 slots
// on 32 different cache lines (the caller sums them).  evlm_mse_grouped no longer reads the 6 x 2 x 19 MB of the ViT states.
#define LN_KD_SLOTS 32
template <typename T, int DCH, bool KD = false>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int rows, int d,
                                                         T* __restrict__ y, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, const T* __restrict__ kd_t = nullptr,
                                                         float* __restrict__ kd_slots = nullptr, float kd_coef = 0.f) {
  const int lane = threadIdx.x & 63, nchunk = d >> 3;
  float sq = 0.f;
  float gm[DCH][8], bt[DCH][8];
#pragma unroll
  for (int k = 0; k < DCH; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) { load8<float>(gamma + c * 8, gm[k]); load8<float>(beta + c * 8, bt[k]); }
  }
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const T* xr = x + (size_t)row * d;
    float v[DCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < DCH; ++k) {
      const int c = lane + 64 * k;
      if (c < nchunk) {
        load8<T>(xr + c * 8, v[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[k][e];
        if (KD) {
          float tv[8];
          load8<T>(kd_t + (size_t)row * d + c * 8, tv);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float df = v[k][e] - tv[e]; sq = fmaf(df, df, sq); }
        }
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < DCH; ++k)
      if (lane + 64 * k < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = v[k][e] - mean; q += t * t; }
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    T* yr = y + (size_t)row * d;
#pragma unroll
    for (int k = 0; k < DCH; ++k) {
      const int c = lane + 64 * k;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * rstd * gm[k][e] + bt[k][e];
        store8<T>(yr + c * 8, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
  if (KD) {
    __shared__ float kdsum[4];
    sq = wave_sum(sq);
    if (lane == 0) kdsum[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0)
      atomicAdd(kd_slots + (blockIdx.x & (LN_KD_SLOTS - 1)) * 32, ((kdsum[0] + kdsum[1]) + (kdsum[2] + kdsum[3])) * kd_coef);
  }
}

// Round 6: d = 768 (the hidden size of both models: 82 launches per GD step, 80 per ITR-384 step).  The kernel above gives a
// row of 96 16-byte chunks to 64 lanes - a second chunk for half of them -, walks its rows one after the other with nothing
// in flight while it reduces, and measured 3.6 TB/s on [12 608, 768] / 2.8 TB/s on [36 928, 768].  Here a wave owns a PAIR
// of consecutive rows = 192 contiguous chunks = exactly three per lane, and the NEXT pair's loads are issued before the
// current pair is touched (raw 16-byte registers, converted where used).  Row 0 of a pair sits in the lanes the kernel above
// would give it (chunks l, l + 64); row 1 sits in that assignment ROTATED by 32 lanes (l ^ 32), which the xor-butterfly of
// wave_sum does not see: per-lane partial sums in the same order, the same tree - bit-identical outputs, mean / rstd included.
template <typename T> struct Raw8;
template <typename T, bool KD>
__global__ __launch_bounds__(256) void ln_fwd_pair768_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int rows,
                                                             T* __restrict__ y, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, const T* __restrict__ kd_t,
                                                             float* __restrict__ kd_slots, float kd_coef) {
  constexpr int D = 768;                        // 96 16-byte chunks of bf16 per row
  const int lane = threadIdx.x & 63;
  const bool up = lane >= 32;                    // chunk slot 1 (pair chunk lane + 64) belongs to row 1 for the upper lanes
  // column chunk of this lane's three pair chunks: slot 0 -> row 0 chunk lane; slot 1 -> row 0 chunk lane + 64 | row 1 chunk
  // lane - 32; slot 2 -> row 1 chunk lane + 32
  const int cc[3] = {lane, up ? lane - 32 : lane + 64, lane + 32};
  // gamma / beta sit in LDS (6 KiB), read where used: held in registers - 48 of them - the kernel ran 3 waves per SIMD
  __shared__ __attribute__((aligned(16))) float gs[D], bs[D];
  for (int i = threadIdx.x; i < D; i += 256) { gs[i] = gamma[i]; bs[i] = beta[i]; }
  __syncthreads();
  const int npairs = (rows + 1) >> 1, stride = gridDim.x * 4;
  int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  Raw8<T> xn[3], tn[KD ? 3 : 1];
  auto issue = [&](int p) {
    // (an odd row count: the last pair's second row reads the first one again and is not stored)
    const size_t r0 = (size_t)2 * p, r1 = min(2 * p + 1, rows - 1);
    const size_t o[3] = {r0 * D + lane * 8, (up ? r1 : r0) * D + cc[1] * 8, r1 * D + cc[2] * 8};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      xn[j].load(x + o[j]);
      if (KD) tn[KD ? j : 0].load(kd_t + o[j]);
    }
  };
  float sq = 0.f;
  if (pair < npairs) issue(pair);
  for (; pair < npairs; pair += stride) {
    Raw8<T> xc[3], tc[KD ? 3 : 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) { xc[j] = xn[j]; if (KD) tc[KD ? j : 0] = tn[KD ? j : 0]; }
    if (pair + stride < npairs) issue(pair + stride);
    const bool has1 = 2 * pair + 1 < rows;
    // (the row stays as LOADED - 12 raw registers - and is converted in each of the three passes: 24 floats held across the
    // reductions, plus the next pair in flight, cost a wave per SIMD)
    // row sums in the order of ln_fwd_reg_kernel: a lane's first chunk (the lower-numbered one), then its second
    float s0 = 0.f, s1 = 0.f;
    {
      float v[8];
      xc[0].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s0 += v[e];
      xc[1].get(v);
      if (!up) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s0 += v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += v[e];
      }
      xc[2].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s1 += v[e];
    }
    if (KD) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float v[8], tv[8];
        xc[j].get(v);
        tc[KD ? j : 0].get(tv);
        if (j == 0 || has1 || (j == 1 && !up)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float df = v[e] - tv[e]; sq = fmaf(df, df, sq); }
        }
      }
    }
    const float mean0 = wave_sum(s0) / (float)D, mean1 = wave_sum(s1) / (float)D;
#pragma unroll
    for (int j = 0; j < 3; ++j) xc[j].opaque();
    // (t * t rounded, then added: the one-row kernels' loops compile to a multiply and an add, and the outputs are to be theirs
    // bit for bit - with two interleaved chains the compiler otherwise contracts SOME of these into fmas)
    float q0 = 0.f, q1 = 0.f;
    {
      float v[8];
      xc[0].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[e] - mean0; q0 += mul_rn(t, t); }
      xc[1].get(v);
      if (!up) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = v[e] - mean0; q0 += mul_rn(t, t); }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = v[e] - mean1; q1 += mul_rn(t, t); }
      }
      xc[2].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[e] - mean1; q1 += mul_rn(t, t); }
    }
    const float rstd0 = rsqrtf(wave_sum(q0) / (float)D + eps), rstd1 = rsqrtf(wave_sum(q1) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < 3; ++j) xc[j].opaque();
    const size_t r0 = (size_t)2 * pair, r1 = r0 + 1;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const bool second = j == 2 || (j == 1 && up);
      if (second && !has1) continue;
      const float mu = second ? mean1 : mean0, rs = second ? rstd1 : rstd0;
      float v[8], o[8], gm[8], bt[8];
      xc[j].get(v);
      load8<float>(gs + cc[j] * 8, gm);
      load8<float>(bs + cc[j] * 8, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu) * rs * gm[e] + bt[e];
      store8<T>(y + (second ? r1 : r0) * D + cc[j] * 8, o);
    }
    if (lane == 0) {
      if (mean_out) { mean_out[r0] = mean0; if (has1) mean_out[r1] = mean1; }
      if (rstd_out) { rstd_out[r0] = rstd0; if (has1) rstd_out[r1] = rstd1; }
    }
  }
  if (KD) {
    __shared__ float kdsum[4];
    sq = wave_sum(sq);
    if (lane == 0) kdsum[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0)
      atomicAdd(kd_slots + (blockIdx.x & (LN_KD_SLOTS - 1)) * 32, ((kdsum[0] + kdsum[1]) + (kdsum[2] + kdsum[3])) * kd_coef);
  }
}
// grid of the pair kernel: every wave makes the same number of trips (>= 2, so that the prefetch has something to hide),
// all workgroups resident
static int ln_pair_blocks(int rows) {
  const int npairs = (rows + 1) / 2;
  const int trips = imax(2, ceil_div(npairs, 4 * 1280));       // (86 VGPRs: 5 workgroups of 4 waves per CU)
  return imax(1, ceil_div(npairs, 4 * trips));
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma += sum_rows dy*xhat;  dbeta += sum_rows dy.
// One wave per row, the row (x, dy) is read ONCE and kept in registers (DCH 16-byte chunks per lane: d <= 512*DCH).
// Each block walks rows blockIdx.x*4 + w, += gridDim.x*4 and keeps per-lane partial dgamma/dbeta for the columns it
// owns; the 4 waves are reduced through LDS, then ONE f32 atomic per column per block.
template <typename T, int DCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ addend,
                                                     const T* __restrict__ addend2,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int rows, int d, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ partials, const T* __restrict__ kd_t = nullptr,
                                                     const float* __restrict__ kd_g = nullptr, float kd_k = 0.f) {
  extern __shared__ float sred[];   // [4 waves][2][d]
  // fused hidden-state distillation (see ln_fwd_reg_kernel): dx += kd_k * (*kd_g) * (x - t), kd_k = 2 w / n, *kd_g = the
  // upstream gradient of the term - what evlm_mse_grouped's backward wrote into an addend buffer before
  const float kdk = kd_t ? kd_k * kd_g[0] : 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  float pg[DCH][8], pb[DCH][8], gm[DCH][8];
#pragma unroll
  for (int u = 0; u < DCH; ++u) {
    const int c = lane + 64 * u;
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[u][e] = 0.f; pb[u][e] = 0.f; gm[u][e] = 0.f; }
    if (c < nchunk) load8<float>(gamma + c * 8, gm[u]);
  }
  // TWO rows per wave per trip: both rows' loads are in flight before either is reduced (one wave owns only a few rows,
  // so without this every trip is a bare HBM round trip)
  for (int row0 = (blockIdx.x * 4 + wave) * 2; row0 < rows; row0 += gridDim.x * 8) {
    const int nr = min(2, rows - row0);
    float xh[2][DCH][8], gd[2][DCH][8];
    float mu[2], rs[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = min(row0 + r, rows - 1);
      mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        const int c = lane + 64 * u;
        if (c < nchunk) {
          load8<T>(x + (size_t)row * d + c * 8, xh[r][u]);
          load8<T>(dy + (size_t)row * d + c * 8, gd[r][u]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r >= nr) break;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        const int c = lane + 64 * u;
        if (c < nchunk) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xv = (xh[r][u][e] - mu[r]) * rs[r], dv = gd[r][u][e];
            xh[r][u][e] = xv;
            gd[r][u][e] = gm[u][e] * dv;
            s1 += gd[r][u][e]; s2 += gd[r][u][e] * xv;
            pg[u][e] += dv * xv; pb[u][e] += dv;
          }
        }
      }
      s1 = wave_sum(s1) / (float)d;
      s2 = wave_sum(s2) / (float)d;
      T* dxr = dx + (size_t)(row0 + r) * d;
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        const int c = lane + 64 * u;
        if (c < nchunk) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs[r] * (gd[r][u][e] - s1 - xh[r][u][e] * s2);
          if (addend) {               // the gradient that reaches x past this LayerNorm (residual branch): summed here
            float ad[8];
            load8<T>(addend + (size_t)(row0 + r) * d + c * 8, ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += ad[e];
          }
          if (addend2) {              // ... and a second one: the distillation term that reads x itself (the "tap")
            float ad[8];
            load8<T>(addend2 + (size_t)(row0 + r) * d + c * 8, ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += ad[e];
          }
          if (kd_t) {                 // ... or that term's gradient formed here: x = xhat / rstd + mean (registers), t read once
            float tv[8];
            load8<T>(kd_t + (size_t)(row0 + r) * d + c * 8, tv);
            const float irs = 1.0f / rs[r];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(kdk, fmaf(xh[r][u][e], irs, mu[r]) - tv[e], o[e]);
          }
          store8<T>(dxr + c * 8, o);
        }
      }
    }
  }
  float* mine = sred + wave * 2 * d;
#pragma unroll
  for (int u = 0; u < DCH; ++u) {
    const int c = lane + 64 * u;
    if (c < nchunk) {
      store8<float>(mine + c * 8, pg[u]);
      store8<float>(mine + d + c * 8, pb[u]);
    }
  }
  __syncthreads();
  // every block adding into the same 2*d floats is the contended-atomic regime (an order of magnitude below the plain
  // store rate): with a workspace the block's column sums leave as plain stores and ln_bwd_reduce_kernel finishes them
  for (int i = threadIdx.x; i < 2 * d; i += 256) {
    const float v = sred[i] + sred[2 * d + i] + sred[4 * d + i] + sred[6 * d + i];
    if (partials) partials[(size_t)blockIdx.x * 2 * d + i] = v;
    else atomicAdd((i < d ? dgamma + i : dbeta + (i - d)), v);
  }
}

// Round 5: the same arithmetic, software-pipelined.  The kernel above spends each trip in phases - loads, reduction, the
// addend's loads, stores - with two waves per SIMD (177 VGPRs at d = 768) and nothing in flight while both reduce: 25 us
// for 56 MB in the step (2.2 TB/s).  Here a wave owns ONE row per trip, keeps the row as LOADED (16-byte raw registers,
// converted where used - twice - instead of 8 floats per chunk held across the reduction) and issues the NEXT row's x /
// dy / addend loads before it touches the current one; gamma is read from LDS.  ~1/3 fewer registers, every load of a row
// in flight behind the previous row's arithmetic.  EXTRA: the rare second addend / fused distillation teacher row, loaded
// at the top of the row's trip.
template <> struct Raw8<bf16> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void get(float o[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
  }
  // (the compiler otherwise keeps the converted floats of the first pass alive across the row reduction for the second)
  __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(v)); }
};
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void get(float o[8]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  }
  __device__ __forceinline__ void opaque() {}
};
// DROP (round 6, ABI 9): a SECOND output dxm = dx .* keep / (1 - p) of hidden-dropout site `call` - the gradient of
// dense(h) in LayerNorm(dropout(dense(h)) + input) (eff_bert.py:372-381,456-462) - written beside dx (= the gradient of the
// input branch) from the same registers: a lane's 16-byte chunk is 8 consecutive columns = one Philox call.  Replaces the
// evlm_dropout pass over dy in the backward of every such site.
struct LnDrop { void* dxm; float p; const int64_t* rng; uint32_t call; };
template <typename T, int DCH, bool ADD, bool EXTRA, bool DROP = false>
__global__ __launch_bounds__(256) void ln_bwd_pipe_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ addend,
                                                          const T* __restrict__ addend2,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int rows, int d, T* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ partials, const T* __restrict__ kd_t,
                                                          const float* __restrict__ kd_g, float kd_k, LnDrop drp = LnDrop{nullptr, 0.f, nullptr, 0}) {
  extern __shared__ __attribute__((aligned(16))) float sred[];   // [4 waves][2][d] column sums, then [d] gamma
  float* gs = sred + 8 * (size_t)d;
  for (int i = threadIdx.x; i < d; i += 256) gs[i] = gamma[i];
  const float kdk = (EXTRA && kd_t) ? kd_k * kd_g[0] : 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  DropRng rng;
  if (DROP) rng = drop_rng(drp.rng, drp.call, drp.p);
  float pg[DCH][8], pb[DCH][8];
#pragma unroll
  for (int u = 0; u < DCH; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[u][e] = 0.f; pb[u][e] = 0.f; }
  __syncthreads();
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  Raw8<T> xn[DCH], dn[DCH], an[DCH];
  float mun = 0.f, rsn = 0.f;
  auto issue = [&](int r) {
#pragma unroll
    for (int u = 0; u < DCH; ++u) {
      const int c = lane + 64 * u;
      if (c < nchunk) {
        xn[u].load(x + (size_t)r * d + c * 8);
        dn[u].load(dy + (size_t)r * d + c * 8);
        if (ADD) an[u].load(addend + (size_t)r * d + c * 8);
      }
    }
    mun = mean[r]; rsn = rstd[r];
  };
  if (row < rows) issue(row);
  for (; row < rows; row += stride) {
    Raw8<T> xc[DCH], dc[DCH], ac[DCH], a2[DCH], kt[DCH];
#pragma unroll
    for (int u = 0; u < DCH; ++u) { xc[u] = xn[u]; dc[u] = dn[u]; if (ADD) ac[u] = an[u]; }
    const float mu = mun, rs = rsn;
    if (row + stride < rows) issue(row + stride);
    if (EXTRA) {
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        const int c = lane + 64 * u;
        if (c < nchunk) {
          if (addend2) a2[u].load(addend2 + (size_t)row * d + c * 8);
          if (kd_t) kt[u].load(kd_t + (size_t)row * d + c * 8);
        }
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < DCH; ++u) {
      const int c = lane + 64 * u;
      if (c < nchunk) {
        float xv[8], dv[8], gm[8];
        xc[u].get(xv); dc[u].get(dv);
        load8<float>(gs + c * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu) * rs, gd = gm[e] * dv[e];
          s1 += gd; s2 += gd * xh;
          pg[u][e] += dv[e] * xh; pb[u][e] += dv[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int u = 0; u < DCH; ++u) { xc[u].opaque(); dc[u].opaque(); }
    T* dxr = dx + (size_t)row * d;
    const float irs = 1.0f / rs;
#pragma unroll
    for (int u = 0; u < DCH; ++u) {
      const int c = lane + 64 * u;
      if (c < nchunk) {
        float xv[8], dv[8], gm[8], o[8];
        xc[u].get(xv); dc[u].get(dv);
        load8<float>(gs + c * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu) * rs;
          o[e] = rs * (gm[e] * dv[e] - s1 - xh * s2);
        }
        if (ADD) {                    // the gradient that reaches x past this LayerNorm (residual branch): summed here
          float ad[8];
          ac[u].get(ad);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += ad[e];
        }
        if (EXTRA) {
          if (addend2) {              // ... and a second one: the distillation term that reads x itself (the "tap")
            float ad[8];
            a2[u].get(ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += ad[e];
          }
          if (kd_t) {                 // ... or that term's gradient formed here (same expression as ln_bwd_kernel: x rebuilt from xhat)
            float tv[8];
            kt[u].get(tv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(kdk, fmaf((xv[e] - mu) * rs, irs, mu) - tv[e], o[e]);
          }
        }
        store8<T>(dxr + c * 8, o);
        if (DROP) {                   // (the mask meets the value AS STORED: what a separate evlm_dropout pass over dx would read)
          float f[8], om[8];
          drop_factor8(rng, ((uint64_t)row * d + c * 8) >> 3, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) om[e] = mul_rn(to_f(from_f<T>(o[e])), f[e]);
          store8<T>(reinterpret_cast<T*>(drp.dxm) + (size_t)row * d + c * 8, om);
        }
      }
    }
  }
  float* mine = sred + wave * 2 * d;
#pragma unroll
  for (int u = 0; u < DCH; ++u) {
    const int c = lane + 64 * u;
    if (c < nchunk) {
      store8<float>(mine + c * 8, pg[u]);
      store8<float>(mine + d + c * 8, pb[u]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 256) {
    const float v = sred[i] + sred[2 * d + i] + sred[4 * d + i] + sred[6 * d + i];
    if (partials) partials[(size_t)blockIdx.x * 2 * d + i] = v;
    else atomicAdd((i < d ? dgamma + i : dbeta + (i - d)), v);
  }
}

// column sums of partials [nblk][2*d] -> dgamma / dbeta (accumulated).  grid (ceil(2d/256), LN_RED_SLICES): each thread sums
// its slice of the block rows for one column, then one atomic per column per slice.
#define LN_RED_SLICES 16
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partials, int nblk, int d,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * d) return;
  const int per = (nblk + LN_RED_SLICES - 1) / LN_RED_SLICES;
  const int r0 = blockIdx.y * per, r1 = min(nblk, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += partials[(size_t)r * 2 * d + i]; a1 += partials[(size_t)(r + 1) * 2 * d + i];
    a2 += partials[(size_t)(r + 2) * 2 * d + i]; a3 += partials[(size_t)(r + 3) * 2 * d + i];
  }
  for (; r < r1; ++r) a0 += partials[(size_t)r * 2 * d + i];
  if (r1 > r0) atomicAdd((i < d ? dgamma + i : dbeta + (i - d)), (a0 + a1) + (a2 + a3));
}

// the same reduction for MANY LayerNorms in one launch (blockIdx.z = unit): a backward pass leaves ~30 workspaces behind,
// each worth a 6 us launch.  table: int64 [n][5] = {partials, nblk, d, dgamma, dbeta}
__global__ __launch_bounds__(256) void ln_bwd_reduce_grouped_kernel(const int64_t* __restrict__ table) {
  const int64_t* e = table + 5 * (int64_t)blockIdx.z;
  const float* partials = reinterpret_cast<const float*>(e[0]);
  const int nblk = (int)e[1], d = (int)e[2];
  float* dgamma = reinterpret_cast<float*>(e[3]);
  float* dbeta = reinterpret_cast<float*>(e[4]);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * d) return;
  const int per = (nblk + LN_RED_SLICES - 1) / LN_RED_SLICES;
  const int r0 = blockIdx.y * per, r1 = min(nblk, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += partials[(size_t)r * 2 * d + i]; a1 += partials[(size_t)(r + 1) * 2 * d + i];
    a2 += partials[(size_t)(r + 2) * 2 * d + i]; a3 += partials[(size_t)(r + 3) * 2 * d + i];
  }
  for (; r < r1; ++r) a0 += partials[(size_t)r * 2 * d + i];
  if (r1 > r0) atomicAdd((i < d ? dgamma + i : dbeta + (i - d)), (a0 + a1) + (a2 + a3));
}
extern "C" int evlm_layernorm_bwd_reduce_grouped(const int64_t* table, int n_units, int d_max, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(table && n_units > 0 && d_max > 0, "evlm_layernorm_bwd_reduce_grouped: bad args");
  hipLaunchKernelGGL(ln_bwd_reduce_grouped_kernel, dim3(ceil_div(2 * d_max, 256), LN_RED_SLICES, n_units), dim3(256), 0, stream,
                     table);
  EVLM_LAUNCH_CHECK("evlm_layernorm_bwd_reduce_grouped");
  return 0;
}

extern "C" int evlm_layernorm_fwd_kd_slots(void) { return LN_KD_SLOTS * 32; }

extern "C" int evlm_layernorm_fwd_kd(int dtype, const void* x, const float* gamma, const float* beta, float eps,
                                     int rows, int d, void* y, float* mean, float* rstd, const void* kd_teacher,
                                     float* kd_slots, float kd_coef, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && y && gamma && beta && rows > 0 && d > 0 && kd_teacher && kd_slots, "evlm_layernorm_fwd_kd: bad args");
  EVLM_REQUIRE(d % 8 == 0 && d <= 2048, "evlm_layernorm_fwd_kd: d=%d unsupported (multiple of 8, <= 2048)", d);
  dim3 block(256), rgrid(rows >= 4096 ? ceil_div(rows, 12) : ceil_div(rows, 4));
  static const bool pair_kd = !getenv("EVLM_LN_FWD_NO_PAIR");        // (A/B switch)
  if (pair_kd && d == 768 && rows >= 1024) {
    EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_fwd_kd",
      hipLaunchKernelGGL((ln_fwd_pair768_kernel<T, true>), dim3(ln_pair_blocks(rows)), block, 0, stream, (const T*)x, gamma, beta, eps,
                         rows, (T*)y, mean, rstd, (const T*)kd_teacher, kd_slots, kd_coef);)
    EVLM_LAUNCH_CHECK("evlm_layernorm_fwd_kd");
    return 0;
  }
#define LN_FWD_KD(DCH_) hipLaunchKernelGGL((ln_fwd_reg_kernel<T, DCH_, true>), rgrid, block, 0, stream, (const T*)x, gamma, beta, eps, rows, d, (T*)y, mean, rstd, (const T*)kd_teacher, kd_slots, kd_coef)
  EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_fwd_kd",
    if (d <= 512) LN_FWD_KD(1); else if (d <= 1024) LN_FWD_KD(2); else if (d <= 1536) LN_FWD_KD(3); else LN_FWD_KD(4);)
#undef LN_FWD_KD
  EVLM_LAUNCH_CHECK("evlm_layernorm_fwd_kd");
  return 0;
}

extern "C" int evlm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, float eps,
                                  int rows, int d, void* y, float* mean, float* rstd, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && y && gamma && beta && rows > 0 && d > 0, "evlm_layernorm_fwd: bad args");
  EVLM_REQUIRE(d % 8 == 0, "evlm_layernorm_fwd: d=%d must be a multiple of 8", d);
  dim3 grid(ceil_div(rows, 4)), block(256);
  const char* env = getenv("EVLM_LN_FWD_3PASS");      // (debug / A-B switch)
  static const bool pair_on = !getenv("EVLM_LN_FWD_NO_PAIR");        // (A/B switch)
  if (pair_on && d == 768 && rows >= 1024 && !(env && atoi(env))) {      // round 6: row pairs, next pair prefetched
    EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_fwd",
      hipLaunchKernelGGL((ln_fwd_pair768_kernel<T, false>), dim3(ln_pair_blocks(rows)), block, 0, stream, (const T*)x, gamma, beta, eps,
                         rows, (T*)y, mean, rstd, (const T*)nullptr, (float*)nullptr, 0.f);)
    EVLM_LAUNCH_CHECK("evlm_layernorm_fwd");
    return 0;
  }
  if (d <= 2048 && !(env && atoi(env))) {     // the row fits the registers of one wave: single pass, 3 rows per wave
    dim3 rgrid(rows >= 4096 ? ceil_div(rows, 12) : ceil_div(rows, 4));
#define LN_FWD(DCH_) hipLaunchKernelGGL((ln_fwd_reg_kernel<T, DCH_>), rgrid, block, 0, stream, (const T*)x, gamma, beta, eps, rows, d, (T*)y, mean, rstd)
    EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_fwd",
      if (d <= 512) LN_FWD(1); else if (d <= 1024) LN_FWD(2); else if (d <= 1536) LN_FWD(3); else LN_FWD(4);)
#undef LN_FWD
    EVLM_LAUNCH_CHECK("evlm_layernorm_fwd");
    return 0;
  }
  EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_fwd",
    hipLaunchKernelGGL((ln_fwd_kernel<T>), grid, block, 0, stream, (const T*)x, gamma, beta, eps, rows, d, (T*)y, mean, rstd);)
  EVLM_LAUNCH_CHECK("evlm_layernorm_fwd");
  return 0;
}

// one wave per row per trip, 3 workgroups of 4 waves resident per CU (<= 160 VGPRs): the trip count of a launch is what 768
// resident workgroups need, and the grid is then cut so that EVERY wave makes that many trips - 12 608 rows on 768
// workgroups left 320 waves a fifth row while 2 752 waited (a 20 % tail), on 631 workgroups every wave takes five
static int ln_bwd_blocks(int rows) {
  const int trips = imax(2, ceil_div(rows, 4 * 768));      // (>= 2: a workgroup's 2 d column sums leave through a workspace)
  return imax(1, ceil_div(rows, 4 * trips));
}

extern "C" int evlm_layernorm_bwd_blocks(int rows) { return ln_bwd_blocks(rows); }

static int layernorm_bwd_impl(int dtype, const void* dy, const void* x, const void* addend, const void* addend2, const float* gamma, const float* mean,
                                  const float* rstd, int rows, int d, void* dx, float* dgamma, float* dbeta,
                                  float* partials, void* stream_, const void* kd_t = nullptr, const float* kd_g = nullptr,
                                  float kd_k = 0.f) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(dy && x && gamma && mean && rstd && dx && rows > 0, "evlm_layernorm_bwd: bad args");
  // dgamma == dbeta == NULL with a workspace: the column sums stay in the workspace, the caller reduces them later
  // (evlm_layernorm_bwd_reduce_grouped: one launch for all the LayerNorms of a backward pass)
  EVLM_REQUIRE((dgamma && dbeta) || (partials && !dgamma && !dbeta), "evlm_layernorm_bwd: dgamma / dbeta missing");
  EVLM_REQUIRE(d % 8 == 0 && d <= 2048, "evlm_layernorm_bwd: d=%d unsupported (multiple of 8, <= 2048)", d);
  const int nblk = ln_bwd_blocks(rows);                // row pairs per wave
  dim3 grid(nblk), block(256);
  const size_t lds = 8 * (size_t)d * sizeof(float);
  static const bool pipe = !getenv("EVLM_LN_BWD_PHASED");       // (A/B switch: the phased two-rows-per-trip kernel)
  if (pipe && d <= 1024) {
    // the pipelined kernel (one row per wave per trip, next row's loads in flight): addend2 / kd_t only in its EXTRA flavour
    const size_t lds_p = 9 * (size_t)d * sizeof(float);
    const bool extra = addend2 || kd_t;
#define LN_BWDP(DCH_, ADD_, EX_) hipLaunchKernelGGL((ln_bwd_pipe_kernel<T, DCH_, ADD_, EX_>), grid, block, lds_p, stream, (const T*)dy, (const T*)x, (const T*)addend, (const T*)addend2, gamma, mean, rstd, rows, d, (T*)dx, dgamma, dbeta, partials, (const T*)kd_t, kd_g, kd_k)
#define LN_BWDP_D(DCH_) do { if (extra) { if (addend) LN_BWDP(DCH_, true, true); else LN_BWDP(DCH_, false, true); } \
                             else if (addend) LN_BWDP(DCH_, true, false); else LN_BWDP(DCH_, false, false); } while (0)
    EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_bwd",
      if (d <= 512) LN_BWDP_D(1); else LN_BWDP_D(2);)
#undef LN_BWDP_D
#undef LN_BWDP
  } else {
#define LN_BWD(DCH_) hipLaunchKernelGGL((ln_bwd_kernel<T, DCH_>), grid, block, lds, stream, (const T*)dy, (const T*)x, (const T*)addend, (const T*)addend2, gamma, mean, rstd, rows, d, (T*)dx, dgamma, dbeta, partials, (const T*)kd_t, kd_g, kd_k)
  EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_bwd",
    if (d <= 512) LN_BWD(1); else if (d <= 1024) LN_BWD(2); else if (d <= 1536) LN_BWD(3); else LN_BWD(4);)
#undef LN_BWD
  }
  if (partials && dgamma)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(2 * d, 256), LN_RED_SLICES), dim3(256), 0, stream, partials, nblk, d,
                       dgamma, dbeta);
  EVLM_LAUNCH_CHECK("evlm_layernorm_bwd");
  return 0;
}

extern "C" int evlm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, int rows, int d, void* dx, float* dgamma, float* dbeta,
                                  float* partials, void* stream) {
  return layernorm_bwd_impl(dtype, dy, x, nullptr, nullptr, gamma, mean, rstd, rows, d, dx, dgamma, dbeta, partials, stream);
}
// dx = LayerNorm backward AND dx_dropped = dx .* keep / (1 - p) of hidden-dropout site (rng_state, call_id) over [rows, d]
// (flat element index row * d + col, as evlm_dropout / evlm_gemm_args.dropout_p): see ln_bwd_pipe_kernel DROP
template <typename T>
__global__ __launch_bounds__(256) void ln_drop_copy_kernel(const T* __restrict__ x, int64_t n8, float p, const int64_t* __restrict__ state,
                                                           uint32_t call, T* __restrict__ y) {
  const DropRng r = drop_rng(state, call, p);
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n8; c += (int64_t)gridDim.x * blockDim.x) {
    float v[8], f[8];
    load8<T>(x + c * 8, v);
    drop_factor8(r, (uint64_t)c, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = mul_rn(v[e], f[e]);
    store8<T>(y + c * 8, v);
  }
}
extern "C" int evlm_layernorm_bwd_drop(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, int rows, int d, void* dx, void* dx_dropped, float dropout_p,
                                       const int64_t* rng_state, uint32_t call_id, float* dgamma, float* dbeta, float* partials,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(dy && x && gamma && mean && rstd && dx && dx_dropped && rng_state && rows > 0, "evlm_layernorm_bwd_drop: bad args");
  EVLM_REQUIRE(dropout_p > 0.f && dropout_p < 1.f, "evlm_layernorm_bwd_drop: dropout_p = %f outside (0, 1)", (double)dropout_p);
  EVLM_REQUIRE((dgamma && dbeta) || (partials && !dgamma && !dbeta), "evlm_layernorm_bwd_drop: dgamma / dbeta missing");
  EVLM_REQUIRE(d % 8 == 0 && d <= 2048, "evlm_layernorm_bwd_drop: d=%d unsupported (multiple of 8, <= 2048)", d);
  static const bool pipe = !getenv("EVLM_LN_BWD_PHASED");
  if (!pipe || d > 1024) {             // (outside the pipelined kernel's range: the plain backward, then one masked copy)
    if (int e = layernorm_bwd_impl(dtype, dy, x, nullptr, nullptr, gamma, mean, rstd, rows, d, dx, dgamma, dbeta, partials, stream_)) return e;
    const int64_t n8 = (int64_t)rows * d / 8;
    EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_bwd_drop",
      hipLaunchKernelGGL((ln_drop_copy_kernel<T>), dim3(imin(ceil_div(n8, 256), 2048)), dim3(256), 0, stream, (const T*)dx, n8,
                         dropout_p, rng_state, call_id, (T*)dx_dropped);)
    EVLM_LAUNCH_CHECK("evlm_layernorm_bwd_drop");
    return 0;
  }
  const int nblk = ln_bwd_blocks(rows);
  dim3 grid(nblk), block(256);
  const size_t lds_p = 9 * (size_t)d * sizeof(float);
  const LnDrop drp{dx_dropped, dropout_p, rng_state, call_id};
#define LN_BWDD(DCH_) hipLaunchKernelGGL((ln_bwd_pipe_kernel<T, DCH_, false, false, true>), grid, block, lds_p, stream, (const T*)dy, (const T*)x, (const T*)nullptr, (const T*)nullptr, gamma, mean, rstd, rows, d, (T*)dx, dgamma, dbeta, partials, (const T*)nullptr, (const float*)nullptr, 0.f, drp)
  EVLM_DISPATCH_DTYPE(dtype, "evlm_layernorm_bwd_drop", if (d <= 512) LN_BWDD(1); else LN_BWDD(2);)
#undef LN_BWDD
  if (partials && dgamma)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(2 * d, 256), LN_RED_SLICES), dim3(256), 0, stream, partials, nblk, d,
                       dgamma, dbeta);
  EVLM_LAUNCH_CHECK("evlm_layernorm_bwd_drop");
  return 0;
}
// dx = LayerNorm backward + addend: the gradient that reaches x along the residual branch past this LayerNorm (pre-LN
// blocks: h = x + f(LN(x))) is summed inside the kernel instead of by a separate element-wise add over [rows, d]
extern "C" int evlm_layernorm_bwd_add(int dtype, const void* dy, const void* x, const void* addend, const void* addend2,
                                      const float* gamma, const float* mean, const float* rstd, int rows, int d, void* dx,
                                      float* dgamma, float* dbeta, float* partials, void* stream) {
  EVLM_REQUIRE(addend, "evlm_layernorm_bwd_add: null addend");
  return layernorm_bwd_impl(dtype, dy, x, addend, addend2, gamma, mean, rstd, rows, d, dx, dgamma, dbeta, partials, stream);
}
// ... and with the gradient of the fused hidden-state distillation term of x (evlm_layernorm_fwd_kd) formed in the kernel:
// dx += kd_k * kd_gout[0] * (x - kd_teacher);  addend / addend2 may be NULL
extern "C" int evlm_layernorm_bwd_kd(int dtype, const void* dy, const void* x, const void* addend, const void* addend2,
                                     const float* gamma, const float* mean, const float* rstd, int rows, int d, void* dx,
                                     float* dgamma, float* dbeta, float* partials, const void* kd_teacher,
                                     const float* kd_gout, float kd_k, void* stream) {
  EVLM_REQUIRE(kd_teacher && kd_gout, "evlm_layernorm_bwd_kd: null kd_teacher / kd_gout");
  return layernorm_bwd_impl(dtype, dy, x, addend, addend2, gamma, mean, rstd, rows, d, dx, dgamma, dbeta, partials, stream,
                            kd_teacher, kd_gout, kd_k);
}
