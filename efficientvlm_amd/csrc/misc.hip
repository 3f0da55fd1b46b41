// Embeddings, data movement, gate kernels and the error plumbing of libevlm_hip.so (gfx950).
// Everything here is HBM-bound: 16-byte vector accesses, consecutive lanes on consecutive addresses.
#include <stdarg.h>
#include "common.h"

// ---- error plumbing --------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int evlm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}
extern "C" const char* evlm_last_error(void) { return g_err; }
extern "C" int evlm_abi_version(void) { return 9; }

// ---- BERT embeddings -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const int64_t* __restrict__ ids, int L, int d,
                                                             const float* __restrict__ word, const float* __restrict__ pos,
                                                             const float* __restrict__ type0, T* __restrict__ out, int rows) {
  const int per = d >> 3;
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < (int64_t)rows * per; id += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(id / per), c = (int)(id - (int64_t)row * per);
    const int l = row % L;
    float w[8], p[8], t[8];
    load8<float>(word + (size_t)ids[row] * d + c * 8, w);
    load8<float>(pos + (size_t)l * d + c * 8, p);
    load8<float>(type0 + c * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = (w[e] + t[e]) + p[e];   // same association order as eff_bert.py:207-211
    store8<T>(out + (size_t)row * d + c * 8, w);
  }
}
// block (l, c): position l, sequences [c*EMB_BC, ...): dpos[l] += sum_b de[b,l]; dtype0 += that; dword[ids[b,l]] += de[b,l]
// (f32 atomics; the batch is cut into chunks so that L x B/EMB_BC workgroups share the sweep instead of L)
#define EMB_BC 8
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const int64_t* __restrict__ ids, int B, int L, int d,
                                                             const T* __restrict__ de, int pad_id, float* __restrict__ dword,
                                                             float* __restrict__ dpos, float* __restrict__ dtype0) {
  const int l = blockIdx.x;
  const int b0 = blockIdx.y * EMB_BC, b1 = min(B, b0 + EMB_BC);
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    float acc = 0.f;
    for (int b = b0; b < b1; ++b) {
      const float g = to_f(de[((size_t)b * L + l) * d + j]);
      acc += g;
      const int64_t w = ids[(size_t)b * L + l];
      if (w != pad_id) atomicAdd(dword + (size_t)w * d + j, g);
    }
    atomicAdd(dpos + (size_t)l * d + j, acc);
    atomicAdd(dtype0 + j, acc);
  }
}
extern "C" int evlm_bert_embed_fwd(int dtype, const int64_t* ids, int B, int L, int d, const float* word,
                                   const float* pos, const float* type0, void* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(ids && word && pos && type0 && out && B > 0 && L > 0 && d % 8 == 0, "evlm_bert_embed_fwd: bad args");
  const int rows = B * L;
  const int grid = imin(2048, ceil_div((int64_t)rows * (d / 8), 256));
  EVLM_DISPATCH_DTYPE(dtype, "evlm_bert_embed_fwd",
    hipLaunchKernelGGL((bert_embed_fwd_kernel<T>), dim3(grid), dim3(256), 0, stream, ids, L, d, word, pos, type0, (T*)out, rows);)
  EVLM_LAUNCH_CHECK("evlm_bert_embed_fwd");
  return 0;
}
extern "C" int evlm_bert_embed_bwd(int dtype, const int64_t* ids, int B, int L, int d, const void* de, int pad_id,
                                   float* dword, float* dpos, float* dtype0, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(ids && de && dword && dpos && dtype0, "evlm_bert_embed_bwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_bert_embed_bwd",
    hipLaunchKernelGGL((bert_embed_bwd_kernel<T>), dim3(L, ceil_div(B, EMB_BC)), dim3(256), 0, stream, ids, B, L, d, (const T*)de, pad_id, dword, dpos, dtype0);)
  EVLM_LAUNCH_CHECK("evlm_bert_embed_bwd");
  return 0;
}

// ---- ViT patch embedding as a GEMM: im2row + token assembly ------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2row_kernel(const float* __restrict__ img, int B, int C, int R, int p,
                                                     T* __restrict__ out) {
  const int G = R / p, K = C * p * p, per = K >> 3, pc = p >> 3;
  const int64_t total = (int64_t)B * G * G * per;
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = id / per;
    const int c8 = (int)(id - row * per);
    const int b = (int)(row / (G * G)), g = (int)(row - (int64_t)b * G * G), gy = g / G, gx = g - gy * G;
    const int c = c8 / (p * pc), rem = c8 - c * p * pc, py = rem / pc, px0 = (rem - py * pc) * 8;
    float v[8];
    load8<float>(img + (((size_t)b * C + c) * R + gy * p + py) * R + gx * p + px0, v);
    store8<T>(out + row * K + c8 * 8, v);
  }
}
extern "C" int evlm_im2row(int dtype, const float* image, int B, int C, int R, int p, void* patches, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(image && patches && B > 0 && p % 8 == 0 && R % p == 0, "evlm_im2row: bad args (patch %d must be a multiple of 8)", p);
  const int G = R / p;
  const int64_t total = (int64_t)B * G * G * (C * p * p / 8);
  const int grid = imin(4096, (total + 255) / 256);
  EVLM_DISPATCH_DTYPE(dtype, "evlm_im2row",
    hipLaunchKernelGGL((im2row_kernel<T>), dim3(grid), dim3(256), 0, stream, image, B, C, R, p, (T*)patches);)
  EVLM_LAUNCH_CHECK("evlm_im2row");
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void vit_embed_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, int B, int Tn, int d,
                                                            T* __restrict__ x) {
  const int per = d >> 3, N = Tn + 1;
  const int64_t total = (int64_t)B * N * per;
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = id / per;
    const int c = (int)(id - row * per);
    const int b = (int)(row / N), n = (int)(row - (int64_t)b * N);
    float v[8], ps[8];
    if (n == 0) load8<float>(cls + c * 8, v);
    else load8<T>(tok + ((size_t)b * Tn + n - 1) * d + c * 8, v);
    load8<float>(pos + (size_t)n * d + c * 8, ps);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += ps[e];
    store8<T>(x + row * d + c * 8, v);
  }
}
// block (n, q): token position n, a quarter of the batch; a thread owns 8 channels (16-byte loads / stores):
// dpos[n] += sum_b dx[b,n]; n==0 -> dcls too; n>=1 -> dtok copy.  (One 2-byte access per thread and a serial loop over the
// whole batch - the first version - took 70 us on [64, 197, 768].)
template <typename T>
__global__ __launch_bounds__(128) void vit_embed_bwd_kernel(const T* __restrict__ dx, int B, int Tn, int d,
                                                            T* __restrict__ dtok, float* __restrict__ dcls,
                                                            float* __restrict__ dpos) {
  const int n = blockIdx.x, N = Tn + 1, nq = gridDim.y;
  const int b0 = (int)(((int64_t)B * blockIdx.y) / nq), b1 = (int)(((int64_t)B * (blockIdx.y + 1)) / nq);
  for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = b0; b < b1; ++b) {
      float v[8];
      load8<T>(dx + ((size_t)b * N + n) * d + c * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
      if (n > 0) store8<T>(dtok + ((size_t)b * Tn + n - 1) * d + c * 8, v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      atomicAdd(dpos + (size_t)n * d + c * 8 + e, acc[e]);
      if (n == 0) atomicAdd(dcls + c * 8 + e, acc[e]);
    }
  }
}
extern "C" int evlm_vit_embed_fwd(int dtype, const void* tok, const float* cls, const float* pos, int B, int T_, int d,
                                  void* x, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(tok && cls && pos && x && d % 8 == 0, "evlm_vit_embed_fwd: bad args");
  const int64_t total = (int64_t)B * (T_ + 1) * (d / 8);
  const int grid = imin(4096, (total + 255) / 256);
  EVLM_DISPATCH_DTYPE(dtype, "evlm_vit_embed_fwd",
    hipLaunchKernelGGL((vit_embed_fwd_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)tok, cls, pos, B, T_, d, (T*)x);)
  EVLM_LAUNCH_CHECK("evlm_vit_embed_fwd");
  return 0;
}
extern "C" int evlm_vit_embed_bwd(int dtype, const void* dx, int B, int T_, int d, void* dtok, float* dcls, float* dpos,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(dx && dtok && dcls && dpos && d % 8 == 0, "evlm_vit_embed_bwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_vit_embed_bwd",
    hipLaunchKernelGGL((vit_embed_bwd_kernel<T>), dim3(T_ + 1, B >= 8 ? 4 : 1), dim3(128), 0, stream, (const T*)dx, B, T_, d, (T*)dtok, dcls, dpos);)
  EVLM_LAUNCH_CHECK("evlm_vit_embed_bwd");
  return 0;
}

// ---- masked-position gather (MLM head input) ----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_fwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ pos,
                                                              int L, int M, int d, T* __restrict__ out, int rows) {
  const int per = d >> 3;
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < (int64_t)rows * per; id += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(id / per), c = (int)(id - (int64_t)row * per), b = row / M;
    float v[8];
    load8<T>(x + ((size_t)b * L + pos[row]) * d + c * 8, v);
    store8<T>(out + (size_t)row * d + c * 8, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ pos,
                                                              int L, int M, int d, T* __restrict__ dx, int rows) {
  const int per = d >> 3;
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < (int64_t)rows * per; id += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(id / per), c = (int)(id - (int64_t)row * per), b = row / L, l = row - b * L;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < M; ++m)
      if (pos[(size_t)b * M + m] == l) {
        float v[8];
        load8<T>(dout + ((size_t)b * M + m) * d + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    store8<T>(dx + (size_t)row * d + c * 8, acc);
  }
}
extern "C" int evlm_gather_rows_fwd(int dtype, const void* x, const int64_t* pos, int B, int L, int M, int d, void* out,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && pos && out && d % 8 == 0, "evlm_gather_rows_fwd: bad args");
  const int rows = B * M;
  const int grid = imin(2048, ceil_div((int64_t)rows * (d / 8), 256));
  EVLM_DISPATCH_DTYPE(dtype, "evlm_gather_rows_fwd",
    hipLaunchKernelGGL((gather_rows_fwd_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)x, pos, L, M, d, (T*)out, rows);)
  EVLM_LAUNCH_CHECK("evlm_gather_rows_fwd");
  return 0;
}
extern "C" int evlm_gather_rows_bwd(int dtype, const void* dout, const int64_t* pos, int B, int L, int M, int d, void* dx,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(dout && pos && dx && d % 8 == 0, "evlm_gather_rows_bwd: bad args");
  const int rows = B * L;
  const int grid = imin(2048, ceil_div((int64_t)rows * (d / 8), 256));
  EVLM_DISPATCH_DTYPE(dtype, "evlm_gather_rows_bwd",
    hipLaunchKernelGGL((gather_rows_bwd_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)dout, pos, L, M, d, (T*)dx, rows);)
  EVLM_LAUNCH_CHECK("evlm_gather_rows_bwd");
  return 0;
}

// ---- whole-sample selection (the fusion pass's batch: [text ; text ; hard-negative text ; masked text]) ---------------
// out[k] = x[sel[k]] for samples of `row_bytes` bytes (a multiple of 16): ONE launch where the reference's
// cat([t, t, index_select(t, neg), m]) (models/model_pretrain.py:179-184 of this package; reference xvlm.py:436-458) is two,
// and a DETERMINISTIC backward dx[r] = sum over {k : sel[k] == r} of dy[k] in ascending k, one launch, where autograd runs
// a zero-fill, an atomic index_add and three element-wise adds.
__global__ __launch_bounds__(256) void select_batches_fwd_kernel(const uint4* __restrict__ x, const int64_t* __restrict__ sel,
                                                                 int64_t w16, uint4* __restrict__ out) {
  const int k = blockIdx.y;
  const uint4* src = x + sel[k] * w16;
  uint4* dst = out + (int64_t)k * w16;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < w16; c += (int64_t)gridDim.x * blockDim.x) dst[c] = src[c];
}
template <typename T>
__global__ __launch_bounds__(256) void select_batches_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ sel,
                                                                 int n, int64_t w8, T* __restrict__ dx) {
  const int64_t r = blockIdx.y;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < w8; c += (int64_t)gridDim.x * blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < n; ++k)
      if (sel[k] == r) {
        float v[8];
        load8<T>(dy + ((int64_t)k * w8 + c) * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    store8<T>(dx + (r * w8 + c) * 8, acc);
  }
}
extern "C" int evlm_select_batches_fwd(const void* x, const int64_t* sel, int n, int64_t row_bytes, void* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && sel && out && n > 0 && n <= 65535 && row_bytes > 0 && row_bytes % 16 == 0 && ((uintptr_t)x) % 16 == 0 &&
               ((uintptr_t)out) % 16 == 0, "evlm_select_batches_fwd: bad args (samples of a multiple of 16 bytes, 16-byte aligned)");
  const int64_t w16 = row_bytes / 16;
  hipLaunchKernelGGL(select_batches_fwd_kernel, dim3(imin(64, ceil_div(w16, 256)), n), dim3(256), 0, stream, (const uint4*)x, sel, w16, (uint4*)out);
  EVLM_LAUNCH_CHECK("evlm_select_batches_fwd");
  return 0;
}
extern "C" int evlm_select_batches_bwd(int dtype, const void* dy, const int64_t* sel, int n, int rows, int64_t row_elems, void* dx,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(dy && sel && dx && n > 0 && rows > 0 && rows <= 65535 && row_elems > 0 && row_elems % 8 == 0 &&
               ((uintptr_t)dy) % 16 == 0 && ((uintptr_t)dx) % 16 == 0, "evlm_select_batches_bwd: bad args");
  const int64_t w8 = row_elems / 8;
  EVLM_DISPATCH_DTYPE(dtype, "evlm_select_batches_bwd",
    hipLaunchKernelGGL((select_batches_bwd_kernel<T>), dim3(imin(64, ceil_div(w8, 256)), rows), dim3(256), 0, stream, (const T*)dy, sel, n, w8, (T*)dx);)
  EVLM_LAUNCH_CHECK("evlm_select_batches_bwd");
  return 0;
}

// ---- dtype cast ---------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nv; c += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8<TS>(s + c * 8, v);
    store8<TD>(d + c * 8, v);
  }
  if (blockIdx.x == 0) for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x) d[i] = from_f<TD>(to_f(s[i]));
}
// ---------------------------------------------------------------------------------------------
// grouped bf16 transposes: unit u is a row-major [R_u][C_u] matrix copied to [C_u][R_u].  One launch covers every unit
// (64 x 64 tiles through LDS, both sides move as 16-byte pieces); the descriptor table lives in device memory:
//   table[5u .. 5u+4] = { src pointer, dst pointer, R, C, index of the unit's first tile }
// Used for the W^T copies of the trainable weights (dX = dY W then reads a K-contiguous operand like the forward).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_grouped_kernel(const int64_t* __restrict__ table, int n_units) {
  __shared__ bf16 tile[64][72];                            // 72: rows stay 16-byte aligned, column reads spread over banks
  int u = 0;
  while (u + 1 < n_units && (int64_t)blockIdx.x >= table[5 * (u + 1) + 4]) ++u;      // (block-uniform scan)
  const bf16* src = reinterpret_cast<const bf16*>(table[5 * u]);
  bf16* dst = reinterpret_cast<bf16*>(table[5 * u + 1]);
  const int R = (int)table[5 * u + 2], C = (int)table[5 * u + 3];
  const int t = blockIdx.x - (int)table[5 * u + 4], tc = (C + 63) >> 6;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  for (int id = threadIdx.x; id < 64 * 8; id += 256) {     // 64 rows x 8 chunks of 8 columns
    const int r = id >> 3, c = (id & 7) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (r0 + r < R && c0 + c < C) v = *reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + r) * C + c0 + c);
    *reinterpret_cast<bf16x8*>(&tile[r][c]) = v;
  }
  __syncthreads();
  for (int id = threadIdx.x; id < 64 * 8; id += 256) {     // 64 output rows (source columns) x 8 chunks of 8 source rows
    const int c = id >> 3, r = (id & 7) * 8;
    if (c0 + c < C && r0 + r < R) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
      *reinterpret_cast<bf16x8*>(dst + (size_t)(c0 + c) * R + r0 + r) = v;
    }
  }
}
extern "C" int evlm_transpose_grouped(const int64_t* table, int n_units, int total_tiles, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(table && n_units > 0 && total_tiles > 0, "evlm_transpose_grouped: bad args");
  hipLaunchKernelGGL(transpose_grouped_kernel, dim3(total_tiles), dim3(256), 0, stream, table, n_units);
  EVLM_LAUNCH_CHECK("evlm_transpose_grouped");
  return 0;
}

extern "C" int evlm_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(src && dst && n > 0, "evlm_cast: bad args");
  EVLM_REQUIRE((((uintptr_t)src) | ((uintptr_t)dst)) % 16 == 0, "evlm_cast: pointers must be 16-byte aligned");
  const int grid = imin(4096, (n / 8 + 255) / 256 + 1);
  if (src_dtype == EVLM_F32 && dst_dtype == EVLM_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, stream, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == EVLM_BF16 && dst_dtype == EVLM_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(grid), dim3(256), 0, stream, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == EVLM_F32 && dst_dtype == EVLM_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, stream, (const float*)src, (float*)dst, n);
  else if (src_dtype == EVLM_BF16 && dst_dtype == EVLM_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, n);
  else return evlm_set_error("evlm_cast: bad dtypes");
  EVLM_LAUNCH_CHECK("evlm_cast");
  return 0;
}

// ---- backward of the gated activations (only when L0 gates are active) -----------------------------
template <typename T>
__global__ __launch_bounds__(256) void gated_act_bwd_kernel(const T* __restrict__ da, const T* __restrict__ h,
                                                            const float* __restrict__ gate, int I, int J, int ld, int act,
                                                            int gate_pos, T* __restrict__ dh, float* __restrict__ dgate,
                                                            int rows_per_block) {
  __shared__ float red[8][32 * 9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j0 = blockIdx.x * 256 + tx * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(I, r0 + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gz[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gz[e] = (j0 + e < J) ? (gate ? gate[j0 + e] : 1.f) : 0.f;
  if (j0 < J) {
    for (int i = r0 + ty; i < r1; i += 8) {
      float a[8], hv[8], o[8];
      load8<T>(da + (size_t)i * ld + j0, a);
      load8<T>(h + (size_t)i * ld + j0, hv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (gate_pos == EVLM_GATE_PRE_ACT) {
          const float t = a[e] * act_grad(act, hv[e] * gz[e]);
          o[e] = t * gz[e];
          acc[e] += t * hv[e];
        } else {
          o[e] = a[e] * act_grad(act, hv[e]) * gz[e];
          acc[e] += a[e] * act_apply(act, hv[e]);
        }
      }
      store8<T>(dh + (size_t)i * ld + j0, o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 9 + e] = acc[e];
  __syncthreads();
  if (ty == 0 && j0 < J && dgate) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
      for (int y = 0; y < 8; ++y) s += red[y][tx * 9 + e];
      if (j0 + e < J) atomicAdd(dgate + j0 + e, s);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ x, int64_t n, int act, T* __restrict__ y) {
  const int64_t nv = n >> 3;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nv; c += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8<T>(x + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_apply(act, v[e]);
    store8<T>(y + c * 8, v);
  }
  if (blockIdx.x == 0) for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x) y[i] = from_f<T>(act_apply(act, to_f(x[i])));
}
extern "C" int evlm_act_fwd(int dtype, const void* x, int64_t n, int act, void* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && y && n > 0, "evlm_act_fwd: bad args");
  const int grid = imin(4096, (n / 8 + 255) / 256 + 1);
  EVLM_DISPATCH_DTYPE(dtype, "evlm_act_fwd",
    hipLaunchKernelGGL((act_fwd_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)x, n, act, (T*)y);)
  EVLM_LAUNCH_CHECK("evlm_act_fwd");
  return 0;
}
extern "C" int evlm_gated_act_bwd(int dtype, const void* da, const void* h, const float* gate, int I, int J, int ld,
                                  int act, int gate_pos, void* dh, float* dgate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(da && h && dh && ld % 8 == 0 && J % 8 == 0, "evlm_gated_act_bwd: bad args");
  EVLM_REQUIRE((gate == nullptr) == (dgate == nullptr), "evlm_gated_act_bwd: gate and dgate go together");
  // rows per block: 256 on large problems, fewer when that would leave most of the chip idle (a [512, 768] head
  // activation ran 68 us on six workgroups)
  int rpb = 256;
  while (rpb > 8 && ceil_div(J, 256) * ceil_div(I, rpb) < 512) rpb >>= 1;
  dim3 grid(ceil_div(J, 256), ceil_div(I, rpb)), block(256);
  EVLM_DISPATCH_DTYPE(dtype, "evlm_gated_act_bwd",
    hipLaunchKernelGGL((gated_act_bwd_kernel<T>), grid, block, 0, stream, (const T*)da, (const T*)h, gate, I, J, ld, act, gate_pos, (T*)dh, dgate, rpb);)
  EVLM_LAUNCH_CHECK("evlm_gated_act_bwd");
  return 0;
}

// ---- hard-concrete L0 gates ----------------------------------------------------------------------
#define L0_LIMIT_A (-0.1f)
#define L0_LIMIT_B (1.1f)
__global__ __launch_bounds__(256) void l0_sample_fwd_kernel(const float* __restrict__ loga, const float* __restrict__ eps,
                                                            int64_t n, float inv_t, float* __restrict__ z) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float u = eps[i];
    const float y = 1.0f / (1.0f + expf(-((logf(u) - logf(1.0f - u) + loga[i]) * inv_t)));
    const float s = y * (L0_LIMIT_B - L0_LIMIT_A) + L0_LIMIT_A;
    z[i] = fminf(fmaxf(s, 0.f), 1.f);
  }
}
__global__ __launch_bounds__(256) void l0_sample_bwd_kernel(const float* __restrict__ loga, const float* __restrict__ eps,
                                                            const float* __restrict__ dz, int64_t n, float inv_t,
                                                            float* __restrict__ dloga) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float u = eps[i];
    const float y = 1.0f / (1.0f + expf(-((logf(u) - logf(1.0f - u) + loga[i]) * inv_t)));
    const float s = y * (L0_LIMIT_B - L0_LIMIT_A) + L0_LIMIT_A;
    const float pass = (s > 0.f && s < 1.f) ? 1.f : 0.f;     // hardtanh backward: open interval
    dloga[i] = dz[i] * pass * (L0_LIMIT_B - L0_LIMIT_A) * y * (1.f - y) * inv_t;
  }
}
extern "C" int evlm_l0_sample_fwd(const float* loga, const float* eps, int64_t n, float temperature, float* z, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(loga && eps && z && n > 0, "evlm_l0_sample_fwd: bad args");
  hipLaunchKernelGGL(l0_sample_fwd_kernel, dim3(imin(1024, (n + 255) / 256)), dim3(256), 0, stream, loga, eps, n, 1.0f / temperature, z);
  EVLM_LAUNCH_CHECK("evlm_l0_sample_fwd");
  return 0;
}
extern "C" int evlm_l0_sample_bwd(const float* loga, const float* eps, const float* dz, int64_t n, float temperature,
                                  float* dloga, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(loga && eps && dz && dloga && n > 0, "evlm_l0_sample_bwd: bad args");
  hipLaunchKernelGGL(l0_sample_bwd_kernel, dim3(imin(1024, (n + 255) / 256)), dim3(256), 0, stream, loga, eps, dz, n, 1.0f / temperature, dloga);
  EVLM_LAUNCH_CHECK("evlm_l0_sample_bwd");
  return 0;
}

// ---- the Lagrangian sparsity term in ONE launch each way (round 5) --------------------------------------------------------
// xvlm_l0_module.py:get_num_parameters_and_constraint + lagrangian_regularization: per gate type t
//   S_t = sum_i (1 - clamp(sigmoid(c - loga_t[i]), eps, 1 - eps)),  n = sum_t S_t * w_t  (type order),
//   es = 1 - n / prunable,  ts = target sparsity (ramped by the step counter),  lag = l1 (es - ts) + l2 (es - ts)^2
// As tensor expressions this is ~40 launches of a few microseconds forward and ~50 backward on the critical path of every
// pruning step (six to eight gate tensors of 36 .. 18 432 elements).  table: int64 [ntypes][3] = {loga, n, f32 bits of w_t}.
__global__ __launch_bounds__(1024) void l0_lagrangian_fwd_kernel(const int64_t* __restrict__ table, int ntypes, float c, float eps,
                                                                 float prunable, float target_sp, float start_sp, float warmup,
                                                                 const float* __restrict__ steps_dev, float steps_host,
                                                                 const float* __restrict__ l1, const float* __restrict__ l2,
                                                                 float* __restrict__ out) {
  __shared__ float red[16];
  __shared__ float S[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = 0; t < ntypes; ++t) {
    const float* la = reinterpret_cast<const float*>(table[3 * t]);
    const int64_t n = table[3 * t + 1];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
      float s = 1.0f / (1.0f + expf(-(c - la[i])));
      s = fminf(fmaxf(s, eps), 1.0f - eps);
      acc += 1.0f - s;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += red[w];
      S[t] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float n = 0.f;
    for (int t = 0; t < ntypes; ++t) {
      const int32_t wb = (int32_t)table[3 * t + 2];
      n = n + S[t] * __int_as_float(wb);
    }
    const float es = 1.0f - n / prunable;
    float ts = target_sp;
    if (warmup > 0.f) {
      const float st = steps_dev ? steps_dev[0] : steps_host;
      ts = (target_sp - start_sp) * fminf(st / warmup, 1.0f) + start_sp;
    }
    const float d = es - ts;
    out[0] = l1[0] * d + l2[0] * (d * d);
    out[1] = es;
    out[2] = ts;
  }
}
// dloga_t[i] += g (l1 + 2 l2 d) (-1 / prunable) w_t [eps <= s <= 1 - eps] s (1 - s);  dl1 += g d;  dl2 += g d^2
// gtable: int64 [ntypes] = gradient buffers (accumulated)
__global__ __launch_bounds__(256) void l0_lagrangian_bwd_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ gtable,
                                                                float c, float eps, float prunable, const float* __restrict__ out_fwd,
                                                                const float* __restrict__ l1, const float* __restrict__ l2,
                                                                const float* __restrict__ gout, float* __restrict__ dl1,
                                                                float* __restrict__ dl2) {
  const int t = blockIdx.y;
  const float* la = reinterpret_cast<const float*>(table[3 * t]);
  float* dla = reinterpret_cast<float*>(gtable[t]);
  const int64_t n = table[3 * t + 1];
  const float w = __int_as_float((int32_t)table[3 * t + 2]);
  const float g = gout[0], d = out_fwd[1] - out_fwd[2];
  const float k = g * (l1[0] + 2.0f * l2[0] * d) * (-1.0f / prunable) * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = 1.0f / (1.0f + expf(-(c - la[i])));
    if (s >= eps && s <= 1.0f - eps) dla[i] += k * (s * (1.0f - s));
  }
  if (t == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    if (dl1) dl1[0] += g * d;
    if (dl2) dl2[0] += g * (d * d);
  }
}
extern "C" int evlm_l0_lagrangian_fwd(const int64_t* table, int ntypes, float logit_c, float eps, float prunable, float target_sp,
                                      float start_sp, float warmup, const float* steps_dev, float steps_host,
                                      const float* lambda1, const float* lambda2, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(table && ntypes > 0 && ntypes <= 16 && lambda1 && lambda2 && out && prunable > 0.f, "evlm_l0_lagrangian_fwd: bad args");
  hipLaunchKernelGGL(l0_lagrangian_fwd_kernel, dim3(1), dim3(1024), 0, stream, table, ntypes, logit_c, eps, prunable, target_sp,
                     start_sp, warmup, steps_dev, steps_host, lambda1, lambda2, out);
  EVLM_LAUNCH_CHECK("evlm_l0_lagrangian_fwd");
  return 0;
}
extern "C" int evlm_l0_lagrangian_bwd(const int64_t* table, const int64_t* gtable, int ntypes, int64_t n_max, float logit_c, float eps,
                                      float prunable, const float* out_fwd, const float* lambda1, const float* lambda2,
                                      const float* gout, float* dlambda1, float* dlambda2, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(table && gtable && ntypes > 0 && ntypes <= 16 && n_max > 0 && out_fwd && lambda1 && lambda2 && gout,
               "evlm_l0_lagrangian_bwd: bad args");
  hipLaunchKernelGGL(l0_lagrangian_bwd_kernel, dim3(imin(64, (int)((n_max + 255) / 256)), ntypes), dim3(256), 0, stream, table, gtable,
                     logit_c, eps, prunable, out_fwd, lambda1, lambda2, gout, dlambda1, dlambda2);
  EVLM_LAUNCH_CHECK("evlm_l0_lagrangian_bwd");
  return 0;
}

// eval masks: one block per layer row.  k = round_half_even(size - sum(1 - cdf_qz(0)));  the k smallest
// sigmoid(loga/T*magic) by ascending (value, index) become 0, everything else 1.
__global__ __launch_bounds__(256) void l0_det_kernel(const float* __restrict__ loga, int size, float temperature,
                                                     float magic, float* __restrict__ z) {
  extern __shared__ float soft[];
  __shared__ double dred[4];
  __shared__ int kzero;
  const float* la = loga + (size_t)blockIdx.x * size;
  float* zr = z + (size_t)blockIdx.x * size;
  const float xn = (0.f - L0_LIMIT_A) / (L0_LIMIT_B - L0_LIMIT_A);
  const float lg = (float)(log((double)xn) - log(1.0 - (double)xn));
  double nz = 0.0;
  for (int i = threadIdx.x; i < size; i += blockDim.x) {
    float c = 1.0f / (1.0f + expf(-(lg * temperature - la[i])));
    c = fminf(fmaxf(c, 1e-6f), 1.0f - 1e-6f);
    nz += (double)(1.0f - c);
    soft[i] = 1.0f / (1.0f + expf(-(la[i] / temperature * magic)));
  }
  for (int o = 32; o > 0; o >>= 1) nz += __shfl_xor(nz, o, 64);
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = nz;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = dred[0] + dred[1] + dred[2] + dred[3];
    const float expected_nonzeros = (float)tot;                    // the reference sums in fp32
    const double ez = (double)size - (double)expected_nonzeros;    // python float arithmetic
    kzero = (int)rint(ez);                                         // python round(): half to even
  }
  __syncthreads();
  const int k = kzero;
  for (int i = threadIdx.x; i < size; i += blockDim.x) {
    if (k <= 0) { zr[i] = 1.0f; continue; }
    const float v = soft[i];
    int rank = 0;
    for (int j = 0; j < size; ++j) {
      const float w = soft[j];
      rank += (w < v || (w == v && j < i)) ? 1 : 0;
    }
    zr[i] = rank < k ? 0.0f : 1.0f;
  }
}
extern "C" int evlm_l0_deterministic(const float* loga, int rows, int size, float temperature, float magical_number,
                                     float* z, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(loga && z && rows > 0 && size > 0 && size <= 16384, "evlm_l0_deterministic: bad args");
  hipLaunchKernelGGL(l0_det_kernel, dim3(rows), dim3(256), size * sizeof(float), stream, loga, size, temperature, magical_number, z);
  EVLM_LAUNCH_CHECK("evlm_l0_deterministic");
  return 0;
}

// ---- row-wise L2 normalisation (F.normalize(dim=-1), get_features xvlm.py:375-382) -------------------
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, int d, int ldx, float eps, T* __restrict__ y,
                                                         float* __restrict__ inv_norm, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) { const float v = to_f(xr[i]); s += v * v; }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), eps);
  for (int i = lane; i < d; i += 64) y[(size_t)row * d + i] = from_f<T>(to_f(xr[i]) * inv);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}
// dx = inv * (dy - y * sum(dy*y))
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy,
                                                         const float* __restrict__ inv_norm, int d, T* __restrict__ dx, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* yr = y + (size_t)row * d;
  const T* dr = dy + (size_t)row * d;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) s += to_f(yr[i]) * to_f(dr[i]);
  s = wave_sum(s);
  const float inv = inv_norm[row];
  for (int i = lane; i < d; i += 64) dx[(size_t)row * d + i] = from_f<T>(inv * (to_f(dr[i]) - to_f(yr[i]) * s));
}
extern "C" int evlm_l2norm_fwd(int dtype, const void* x, int rows, int d, int ldx, float eps, void* y, float* inv_norm, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && y && rows > 0 && d > 0, "evlm_l2norm_fwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_l2norm_fwd",
    hipLaunchKernelGGL((l2norm_fwd_kernel<T>), dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const T*)x, d, ldx, eps, (T*)y, inv_norm, rows);)
  EVLM_LAUNCH_CHECK("evlm_l2norm_fwd");
  return 0;
}
extern "C" int evlm_l2norm_bwd(int dtype, const void* y, const void* dy, const float* inv_norm, int rows, int d, void* dx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(y && dy && inv_norm && dx && rows > 0, "evlm_l2norm_bwd: bad args");
  EVLM_DISPATCH_DTYPE(dtype, "evlm_l2norm_bwd",
    hipLaunchKernelGGL((l2norm_bwd_kernel<T>), dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const T*)y, (const T*)dy, inv_norm, d, (T*)dx, rows);)
  EVLM_LAUNCH_CHECK("evlm_l2norm_bwd");
  return 0;
}

// ---- dropout (hidden states): y = x .* keep / (1 - p) (+ residual) --------------------------------
// The same kernel is its own backward (dx = dy .* the regenerated mask).  8 elements per thread = ONE Philox call (common.h).
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, int64_t n, float p,
                                                      const int64_t* __restrict__ state, uint32_t call, T* __restrict__ y) {
  const DropRng r = drop_rng(state, call, p);
  const int64_t nv = n >> 3;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nv; c += (int64_t)gridDim.x * blockDim.x) {
    float v[8], rr[8], f[8];
    load8<T>(x + c * 8, v);
    if (res) load8<T>(res + c * 8, rr);
    drop_factor8(r, (uint64_t)c, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = mul_rn(v[e], f[e]) + (res ? rr[e] : 0.f);      // (no fma: x .* m is rounded, then + residual)
    store8<T>(y + c * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 7)) {
    const int64_t i = (nv << 3) + threadIdx.x;
    y[i] = from_f<T>(mul_rn(to_f(x[i]), drop_factor(r, (uint64_t)i)) + (res ? to_f(res[i]) : 0.f));
  }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t n, float p, const int64_t* __restrict__ state, uint32_t call,
                                                           float* __restrict__ mask) {
  const DropRng r = drop_rng(state, call, p);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    mask[i] = drop_factor(r, (uint64_t)i);
}
extern "C" int evlm_dropout(int dtype, const void* x, const void* residual, int64_t n, float p, const int64_t* rng_state,
                            uint32_t call_id, void* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(x && y && rng_state && n > 0, "evlm_dropout: bad args");
  EVLM_REQUIRE(p >= 0.f && p < 1.f, "evlm_dropout: p = %f outside [0, 1)", (double)p);
  EVLM_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) % 16 == 0, "evlm_dropout: 16-byte alignment");
  const int blocks = imin(ceil_div(ceil_div(n, 8), 256), 2048);
  EVLM_DISPATCH_DTYPE(dtype, "evlm_dropout",
    hipLaunchKernelGGL((dropout_kernel<T>), dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, (const T*)x, (const T*)residual,
                       n, p, rng_state, call_id, (T*)y);)
  EVLM_LAUNCH_CHECK("evlm_dropout");
  return 0;
}
extern "C" int evlm_dropout_mask(int64_t n, float p, const int64_t* rng_state, uint32_t call_id, float* mask, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(mask && rng_state && n > 0 && p >= 0.f && p < 1.f, "evlm_dropout_mask: bad args");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(imin(ceil_div(n, 256), 2048)), dim3(256), 0, stream, n, p, rng_state, call_id, mask);
  EVLM_LAUNCH_CHECK("evlm_dropout_mask");
  return 0;
}

// ---- grouped copy: many (src, dst, bytes) pairs in ONE launch -------------------------------------------------------
// table: int64 [n][4] = {src, dst, bytes (a multiple of 16; both pointers 16-byte aligned), first block of the unit};
// a block copies 64 KiB (256 threads x 16 x 16 bytes).  The teacher pipeline parks ~45 small tensors per step in
// persistent buffers: one launch instead of 45 copy launches.
__global__ __launch_bounds__(256) void copy_grouped_kernel(const int64_t* __restrict__ table, int n_units) {
  int u = 0;
  while (u + 1 < n_units && (int64_t)blockIdx.x >= table[4 * (u + 1) + 3]) ++u;      // (block-uniform scan)
  const uint4* src = reinterpret_cast<const uint4*>(table[4 * u]);
  uint4* dst = reinterpret_cast<uint4*>(table[4 * u + 1]);
  const int64_t nvec = table[4 * u + 2] >> 4;
  const int64_t v0 = ((int64_t)blockIdx.x - table[4 * u + 3]) * 4096;
  if (!src) {                          // a unit without a source is a ZERO FILL (ops.zero_grouped: a step's gradient ranges)
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int64_t i = v0 + k * 256 + threadIdx.x;
      if (i < nvec) dst[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int64_t i = v0 + k * 256 + threadIdx.x;
    if (i < nvec) dst[i] = src[i];
  }
}
// ... and the same for up to 8 units whose addresses change from call to call (a step's input batch going into its static
// buffers): the units travel as KERNEL ARGUMENTS - no device table to upload.
struct CopyFew { const void* src[8]; void* dst[8]; int64_t nbytes[8]; int first_block[9]; int n; };
__global__ __launch_bounds__(256) void copy_few_kernel(CopyFew a) {
  int u = 0;
  while (u + 1 < a.n && (int)blockIdx.x >= a.first_block[u + 1]) ++u;
  const uint4* src = reinterpret_cast<const uint4*>(a.src[u]);
  uint4* dst = reinterpret_cast<uint4*>(a.dst[u]);
  const int64_t nvec = a.nbytes[u] >> 4;
  const int64_t v0 = ((int64_t)blockIdx.x - a.first_block[u]) * 4096;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int64_t i = v0 + k * 256 + threadIdx.x;
    if (i < nvec) dst[i] = src[i];
  }
}
extern "C" int evlm_copy_few(const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(src && dst && nbytes && n > 0 && n <= 8, "evlm_copy_few: 1..8 units");
  CopyFew a;
  int blocks = 0;
  for (int u = 0; u < n; ++u) {
    EVLM_REQUIRE(src[u] && dst[u] && nbytes[u] > 0 && nbytes[u] % 16 == 0 && ((uintptr_t)src[u]) % 16 == 0 && ((uintptr_t)dst[u]) % 16 == 0,
                 "evlm_copy_few: units of a multiple of 16 bytes, 16-byte aligned");
    a.src[u] = src[u]; a.dst[u] = dst[u]; a.nbytes[u] = nbytes[u]; a.first_block[u] = blocks;
    blocks += (int)((nbytes[u] + 65535) / 65536);
  }
  a.first_block[n] = blocks; a.n = n;
  hipLaunchKernelGGL(copy_few_kernel, dim3(blocks), dim3(256), 0, stream, a);
  EVLM_LAUNCH_CHECK("evlm_copy_few");
  return 0;
}
extern "C" int evlm_copy_grouped(const int64_t* table, int n_units, int total_blocks, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(table && n_units > 0 && total_blocks > 0, "evlm_copy_grouped: bad args");
  hipLaunchKernelGGL(copy_grouped_kernel, dim3(total_blocks), dim3(256), 0, stream, table, n_units);
  EVLM_LAUNCH_CHECK("evlm_copy_grouped");
  return 0;
}
