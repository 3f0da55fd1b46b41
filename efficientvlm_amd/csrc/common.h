// Shared device/host helpers for the gfx950 kernels of libevlm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/evlm_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// ---- error plumbing ------------------------------------------------------------------------
int evlm_set_error(const char* fmt, ...);
#define EVLM_REQUIRE(cond, ...)                      \
  do {                                               \
    if (!(cond)) return evlm_set_error(__VA_ARGS__); \
  } while (0)
#define EVLM_LAUNCH_CHECK(name)                                                    \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) return evlm_set_error("%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- scalar conversions --------------------------------------------------------------------
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

// ---- 4-wide vector access (4 consecutive elements) -------------------------------------------
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  __device__ __forceinline__ static void load(const float* p, float v[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ __forceinline__ static void store(float* p, const float v[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct Vec4<bf16> {
  __device__ __forceinline__ static void load(const bf16* p, float v[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
  }
  __device__ __forceinline__ static void store(bf16* p, const float v[4]) {
    bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = t;
  }
};

// 8 consecutive elements -> 8 floats (16-byte load for bf16, 2x16-byte for f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  Vec4<float>::load(p, v);
  Vec4<float>::load(p + 4, v + 4);
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float v[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  Vec4<float>::store(p, v);
  Vec4<float>::store(p + 4, v + 4);
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float v[8]) {
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = t;
}

// ---- activations (fp32 math) ---------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoidf_(1.702f * x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
  const float s = sigmoidf_(1.702f * x);
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
// bf16-path variants: hardware reciprocal (1 ulp) and the Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7, far below a
// bf16 ulp) instead of the IEEE division and erff() - the activation runs in GEMM epilogues, where a 256x256 tile applies
// it to 128 values per lane with nothing else to hide behind
__device__ __forceinline__ float quick_gelu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad_fast(float x) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
// returns erf(x / sqrt 2) and exp(-x^2 / 2) (shared by GELU and its derivative)
__device__ __forceinline__ float erf_half_fast(float x, float& e) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  e = __expf(-ax * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  return copysignf(1.0f - poly * e, x);
}
__device__ __forceinline__ float gelu_erf_fast(float x) { float e; return 0.5f * x * (1.0f + erf_half_fast(x, e)); }
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  float e;
  const float cdf = 0.5f * (1.0f + erf_half_fast(x, e));
  return cdf + x * 0.3989422804014327f * e;
}
__device__ __forceinline__ float act_apply_fast(int act, float x) {
  return act == EVLM_ACT_GELU ? gelu_erf_fast(x) : (act == EVLM_ACT_QUICK_GELU ? quick_gelu_fast(x) : x);
}
__device__ __forceinline__ float act_grad_fast(int act, float x) {
  return act == EVLM_ACT_GELU ? gelu_erf_grad_fast(x) : (act == EVLM_ACT_QUICK_GELU ? quick_gelu_grad_fast(x) : 1.0f);
}
__device__ __forceinline__ float act_apply(int act, float x) {
  return act == EVLM_ACT_GELU ? gelu_erf(x) : (act == EVLM_ACT_QUICK_GELU ? quick_gelu(x) : x);
}
__device__ __forceinline__ float act_grad(int act, float x) {
  return act == EVLM_ACT_GELU ? gelu_erf_grad(x) : (act == EVLM_ACT_QUICK_GELU ? quick_gelu_grad(x) : 1.0f);
}

// ---- counter-based dropout masks (Philox4x32-10) ---------------------------------------------
// A dropout site is identified by (seed, step, call id).  A site is a matrix [rows][cols]; with its columns padded to a
// multiple of 8 (cols8 = 8 ceil(cols / 8)) element (row, col) has the index idx = row * cols8 + col, and it keeps its value
// iff the 16-bit lane (idx & 7) of Philox(counter = {idx >> 3 lo, idx >> 3 hi, call, step}, key = seed) - lane e = bits
// 16 (e & 1) .. +15 of output word e >> 1 - is >= round(p * 2^16).  ONE Philox call serves 8 consecutive columns of a row:
// the 16-byte piece a lane of the attention kernels owns per key-tile pair (8 consecutive keys of one query), the 16-byte
// piece a lane of the GEMM epilogue stores, the 8 elements a thread of the element-wise kernels handles.  (Round 6: before,
// a 32-bit word per element of the FLAT index - one Philox call per 4 flat elements, which in the attention layouts was
// one call per element.)  A kept element is multiplied by exactly 1 / (1 - p), as the reference's nn.Dropout does; the
// keep probability itself is 1 - round(p 2^16) / 2^16 (p = 0.1: 0.899994).  The mask is a pure function of those numbers,
// so the backward kernels REGENERATE it: no mask tensor ever exists in HBM.  rng_state is a DEVICE int64[2] = {seed, step}:
// a captured hipGraph draws fresh masks on every replay once the host (or a one-word kernel) has bumped `step`; the call
// id is a host constant of each call site.  Hidden-state sites have cols % 8 == 0 or are addressed flat (evlm_dropout).
struct DropRng { uint32_t k0, k1, call, step; uint32_t thresh; float scale; };
__device__ __forceinline__ DropRng drop_rng(const int64_t* __restrict__ state, uint32_t call, float p) {
  DropRng r;
  const uint64_t seed = (uint64_t)state[0], step = (uint64_t)state[1];
  r.k0 = (uint32_t)seed; r.k1 = (uint32_t)(seed >> 32); r.call = call; r.step = (uint32_t)step;
  const float t = rintf(p * 65536.0f);
  r.thresh = t >= 65535.0f ? 65535u : (uint32_t)t;
  r.scale = 1.0f / (1.0f - p);
  return r;
}
// (scalars, not arrays, all the way: a select between two elements of a private array - the half-call forms below - is turned
// into a dynamically indexed load, and the array into scratch memory)
__device__ __forceinline__ void philox4s(const DropRng& r, uint64_t block, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
  uint32_t c0 = (uint32_t)block, c1 = (uint32_t)(block >> 32), c2 = r.call, c3 = r.step, k0 = r.k0, k1 = r.k1;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;      // (v_mad_u64_u32: hi and lo at once)
    const uint32_t h0 = (uint32_t)(m0 >> 32), l0 = (uint32_t)m0, h1 = (uint32_t)(m1 >> 32), l1 = (uint32_t)m1;
    c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
__device__ __forceinline__ void philox4(const DropRng& r, uint64_t block, uint32_t (&o)[4]) {
  philox4s(r, block, o[0], o[1], o[2], o[3]);
}
// x .* m rounded on its own (HIP's __fmul_rn is a plain `x * y`, which the compiler contracts with a following add into one
// fma): the hidden-dropout sites compute round(x m) + residual wherever they run - evlm_dropout, the GEMM residual epilogues,
// the LayerNorm backward's masked output - so that the fused and the separate forms agree bit for bit
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
// keep / (1 - p) of the 8 consecutive elements idx8 * 8 .. + 7 (idx8 = padded index >> 3): ONE Philox call
__device__ __forceinline__ void drop_factor8(const DropRng& r, uint64_t idx8, float (&f)[8]) {
  uint32_t o0, o1, o2, o3;
  philox4s(r, idx8, o0, o1, o2, o3);
  f[0] = (o0 & 0xFFFFu) >= r.thresh ? r.scale : 0.f; f[1] = (o0 >> 16) >= r.thresh ? r.scale : 0.f;
  f[2] = (o1 & 0xFFFFu) >= r.thresh ? r.scale : 0.f; f[3] = (o1 >> 16) >= r.thresh ? r.scale : 0.f;
  f[4] = (o2 & 0xFFFFu) >= r.thresh ? r.scale : 0.f; f[5] = (o2 >> 16) >= r.thresh ? r.scale : 0.f;
  f[6] = (o3 & 0xFFFFu) >= r.thresh ? r.scale : 0.f; f[7] = (o3 >> 16) >= r.thresh ? r.scale : 0.f;
}
// ... of elements 4 hi .. 4 hi + 3 of that group (lanes that own 4 consecutive columns: half of a call's factors)
__device__ __forceinline__ void drop_factor4(const DropRng& r, uint64_t idx8, bool hi, float (&f)[4]) {
  uint32_t o0, o1, o2, o3;
  philox4s(r, idx8, o0, o1, o2, o3);
  const uint32_t a = hi ? o2 : o0, b = hi ? o3 : o1;
  f[0] = (a & 0xFFFFu) >= r.thresh ? r.scale : 0.f; f[1] = (a >> 16) >= r.thresh ? r.scale : 0.f;
  f[2] = (b & 0xFFFFu) >= r.thresh ? r.scale : 0.f; f[3] = (b >> 16) >= r.thresh ? r.scale : 0.f;
}
// ... of ONE element (shape-generic kernels whose threads do not own 8 consecutive columns)
__device__ __forceinline__ float drop_factor(const DropRng& r, uint64_t idx) {
  uint32_t o0, o1, o2, o3;
  philox4s(r, idx >> 3, o0, o1, o2, o3);
  const uint32_t w = (idx & 4) ? ((idx & 2) ? o3 : o2) : ((idx & 2) ? o1 : o0);
  return ((idx & 1) ? (w >> 16) : (w & 0xFFFFu)) >= r.thresh ? r.scale : 0.f;
}

// ---- wave / block reductions (64-lane waves) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` holds >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int imin(int64_t a, int64_t b) { return (int)(a < b ? a : b); }
static inline int imax(int64_t a, int64_t b) { return (int)(a > b ? a : b); }

// run `body<T>` for the runtime dtype
#define EVLM_DISPATCH_DTYPE(dt, NAME, ...)                        \
  if ((dt) == EVLM_F32) { typedef float T; __VA_ARGS__ }          \
  else if ((dt) == EVLM_BF16) { typedef bf16 T; __VA_ARGS__ }     \
  else return evlm_set_error("%s: bad dtype %d", NAME, (int)(dt));
