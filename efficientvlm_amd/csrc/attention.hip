// Multi-head attention core with the probability map as an output (general path).
//
//   S = scale * Q K^T + mask ;  P = softmax(S) ;  O = (P V) * gate[h]
//
// This is the shape-generic kernel family (any Lq/Lk, dh in {16,32,...,128}, f32 or bf16 storage, fp32 math):
// it is the exact-fp32 parity path and the fallback for shapes the MFMA kernels (attention_mfma.hip) do not take.
// One workgroup owns a 16-row query tile of one (batch, head); the whole score row block [16][Lk] lives in LDS
// (so the softmax is a single pass and P is written to HBM exactly once); K and V stream through LDS in 64-key
// chunks.
#include "common.h"

#define AQ 16     // query rows per workgroup
#define AKC 64    // keys per LDS chunk
#define AMAXU 8   // dh <= 128

struct AttnF {
  const void* Q; const void* K; const void* V; const int32_t* kv_index; const float* mask; const float* gate;
  void* O; void* P;
  int B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo, ldpr;
  float scale;
  int causal;
  float drop_p; const int64_t* rng; uint32_t call;      // attention-probability dropout (0: off)
};

// load rows [r0, r0+nrows) x dh of a [.., L, H, dh]-strided tensor (row stride ld) into LDS as fp32 [nrows][dh+1]
template <typename T>
__device__ __forceinline__ void load_rows(const T* base, int ld, int r0, int nrows, int L, int dh, float* s) {
  const int per = dh >> 3;   // 16-byte chunks per row (8 elements) for bf16 / two loads for f32
  for (int id = threadIdx.x; id < nrows * per; id += blockDim.x) {
    const int r = id / per, c = id - r * per;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + r < L) load8<T>(base + (size_t)(r0 + r) * ld + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[r * (dh + 1) + c * 8 + e] = v[e];
  }
}

template <typename T, typename TP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnF a) {
  extern __shared__ float sm[];
  const int dh = a.dh, Lk = a.Lk, Lkp = (Lk + 3) & ~3;
  float* Qs = sm;                       // [16][dh+1]
  float* KVs = Qs + AQ * (dh + 1);      // [64][dh+1]
  float* Ss = KVs + AKC * (dh + 1);     // [16][Lkp]
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const int tid = threadIdx.x, r = tid >> 4, kl = tid & 15;
  const T* Qb = reinterpret_cast<const T*>(a.Q) + (size_t)b * a.Lq * a.ldq + h * dh;
  const T* Kb = reinterpret_cast<const T*>(a.K) + (size_t)bkv * Lk * a.ldk + h * dh;
  const T* Vb = reinterpret_cast<const T*>(a.V) + (size_t)bkv * Lk * a.ldv + h * dh;
  const float* mk = a.mask ? a.mask + (size_t)b * Lk : nullptr;

  load_rows<T>(Qb, a.ldq, q0, AQ, a.Lq, dh, Qs);
  for (int kb = 0; kb < Lk; kb += AKC) {
    __syncthreads();
    load_rows<T>(Kb, a.ldk, kb, AKC, Lk, dh, KVs);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = kl + 16 * u;
      if (kb + key < Lk) {
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(Qs[r * (dh + 1) + d], KVs[key * (dh + 1) + d], s);
        s *= a.scale;
        float add = mk ? mk[kb + key] : 0.f;
        if (a.causal && kb + key > q0 + r) add = fminf(add, -10000.0f);      // (1 - causal * pad) * -10000: once, not twice
        s += add;
        Ss[r * Lkp + kb + key] = s;
      }
    }
  }
  __syncthreads();
  // softmax over each of the 16 rows: 16 consecutive lanes own one row
  {
    float m = -INFINITY;
    for (int k = kl; k < Lk; k += 16) m = fmaxf(m, Ss[r * Lkp + k]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int k = kl; k < Lk; k += 16) {
      const float e = __expf(Ss[r * Lkp + k] - m);
      Ss[r * Lkp + k] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    const bool rowok = (q0 + r < a.Lq);
    TP* Pr = reinterpret_cast<TP*>(a.P) + (((size_t)b * a.H + h) * a.Lq + (q0 + r)) * a.ldpr;
    // the map is returned BEFORE dropout (eff_bert.py:338-361); the context uses the dropped probabilities (:346-352)
    const bool drop = a.drop_p > 0.f;
    DropRng rng;
    if (drop) rng = drop_rng(a.rng, a.call, a.drop_p);
    const uint64_t drow = (((uint64_t)b * a.H + h) * a.Lq + (q0 + r)) * (uint64_t)((Lk + 7) & ~7);   // (rows padded to 8: common.h)
    for (int k = kl; k < Lk; k += 16) {
      const float p = Ss[r * Lkp + k] * inv;
      Ss[r * Lkp + k] = (drop && rowok) ? p * drop_factor(rng, drow + k) : p;
      if (rowok && a.P) Pr[k] = from_f<TP>(p);
    }
    if (rowok && a.P) for (int k = Lk + kl; k < a.ldpr; k += 16) Pr[k] = from_f<TP>(0.f);
  }
  // O = P V
  float acc[AMAXU];
#pragma unroll
  for (int u = 0; u < AMAXU; ++u) acc[u] = 0.f;
  const int nu = dh >> 4;
  for (int kb = 0; kb < Lk; kb += AKC) {
    __syncthreads();
    load_rows<T>(Vb, a.ldv, kb, AKC, Lk, dh, KVs);
    __syncthreads();
    const int kn = min(AKC, Lk - kb);
    for (int k = 0; k < kn; ++k) {
      const float p = Ss[r * Lkp + kb + k];
#pragma unroll
      for (int u = 0; u < AMAXU; ++u)
        if (u < nu) acc[u] = fmaf(p, KVs[k * (dh + 1) + kl + 16 * u], acc[u]);
    }
  }
  if (q0 + r < a.Lq) {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    T* Or = reinterpret_cast<T*>(a.O) + ((size_t)b * a.Lq + q0 + r) * a.ldo + h * dh;
#pragma unroll
    for (int u = 0; u < AMAXU; ++u)
      if (u < nu) Or[kl + 16 * u] = from_f<T>(acc[u] * gz);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, kernel A (per 16-row query tile):  dS = P .* (dP - rowsum(P .* dP)),  dP = gate*dO V^T + dP_ext
//   writes dS (workspace, for kernel B) and dQ = scale * dS K ; accumulates dgate[h] = sum dO .* (P V)
// ---------------------------------------------------------------------------------------------
struct AttnB {
  const void* Q; const void* K; const void* V; const void* P; const void* dO; const void* dPext;
  const int32_t* kv_index; const float* gate;
  void* dS; void* dQ; void* dK; void* dV; float* dK32; float* dV32; float* dgate;
  int B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo, lddq, lddk, lddv, ldpr;
  float scale;
  float drop_p; const int64_t* rng; uint32_t call;
};

template <typename T, typename TP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnB a) {
  extern __shared__ float sm[];
  const int dh = a.dh, Lk = a.Lk, Lkp = (Lk + 3) & ~3;
  float* dOs = sm;                      // [16][dh+1]
  float* KVs = dOs + AQ * (dh + 1);     // [64][dh+1]
  float* Ss = KVs + AKC * (dh + 1);     // [16][Lkp]   dP, then dS
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const int tid = threadIdx.x, r = tid >> 4, kl = tid & 15;
  const T* dOb = reinterpret_cast<const T*>(a.dO) + (size_t)b * a.Lq * a.ldo + h * dh;
  const T* Kb = reinterpret_cast<const T*>(a.K) + (size_t)bkv * Lk * a.ldk + h * dh;
  const T* Vb = reinterpret_cast<const T*>(a.V) + (size_t)bkv * Lk * a.ldv + h * dh;
  const bool rowok = (q0 + r < a.Lq);
  const size_t prow = (((size_t)b * a.H + h) * a.Lq + (q0 + r)) * a.ldpr;
  const TP* Pr = reinterpret_cast<const TP*>(a.P) + prow;
  const TP* Er = a.dPext ? reinterpret_cast<const TP*>(a.dPext) + prow : nullptr;

  load_rows<T>(dOb, a.ldo, q0, AQ, a.Lq, dh, dOs);
  for (int kb = 0; kb < Lk; kb += AKC) {     // dPo = dO V^T (ungated)
    __syncthreads();
    load_rows<T>(Vb, a.ldv, kb, AKC, Lk, dh, KVs);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = kl + 16 * u;
      if (kb + key < Lk) {
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(dOs[r * (dh + 1) + d], KVs[key * (dh + 1) + d], s);
        Ss[r * Lkp + kb + key] = s;
      }
    }
  }
  __syncthreads();
  {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    float dsum = 0.f, gsum = 0.f;
    const bool drop = a.drop_p > 0.f;
    DropRng rng;
    if (drop) rng = drop_rng(a.rng, a.call, a.drop_p);
    const uint64_t drow = (((uint64_t)b * a.H + h) * a.Lq + (q0 + r)) * (uint64_t)((Lk + 7) & ~7);
    if (rowok) {
      for (int k = kl; k < Lk; k += 16) {
        const float p = to_f(Pr[k]);
        // O = gate * (P .* M) V: the gradient reaches P through the same (regenerated) mask M = keep / (1 - p)
        const float dpo = drop ? Ss[r * Lkp + k] * drop_factor(rng, drow + k) : Ss[r * Lkp + k];
        gsum += p * dpo;
        const float dp = gz * dpo + (Er ? to_f(Er[k]) : 0.f);
        Ss[r * Lkp + k] = dp;
        dsum += p * dp;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { dsum += __shfl_xor(dsum, o, 64); gsum += __shfl_xor(gsum, o, 64); }
    if (a.dgate) {   // sum over the wave's 4 rows, then one atomic per wave
      float g4 = gsum;
      g4 += __shfl_xor(g4, 16, 64);
      g4 += __shfl_xor(g4, 32, 64);
      if ((tid & 63) == 0) atomicAdd(a.dgate + h, g4);
    }
    T* dSr = reinterpret_cast<T*>(a.dS) + prow;
    for (int k = kl; k < Lk; k += 16) {
      float ds = 0.f;
      if (rowok) {
        ds = to_f(Pr[k]) * (Ss[r * Lkp + k] - dsum);
        dSr[k] = from_f<T>(ds);
      }
      Ss[r * Lkp + k] = ds;
    }
    if (rowok) for (int k = Lk + kl; k < a.ldpr; k += 16) dSr[k] = from_f<T>(0.f);
  }
  float acc[AMAXU];
#pragma unroll
  for (int u = 0; u < AMAXU; ++u) acc[u] = 0.f;
  const int nu = dh >> 4;
  for (int kb = 0; kb < Lk; kb += AKC) {     // dQ = scale * dS K
    __syncthreads();
    load_rows<T>(Kb, a.ldk, kb, AKC, Lk, dh, KVs);
    __syncthreads();
    const int kn = min(AKC, Lk - kb);
    for (int k = 0; k < kn; ++k) {
      const float p = Ss[r * Lkp + kb + k];
#pragma unroll
      for (int u = 0; u < AMAXU; ++u)
        if (u < nu) acc[u] = fmaf(p, KVs[k * (dh + 1) + kl + 16 * u], acc[u]);
    }
  }
  if (rowok) {
    T* dQr = reinterpret_cast<T*>(a.dQ) + ((size_t)b * a.Lq + q0 + r) * a.lddq + h * dh;
#pragma unroll
    for (int u = 0; u < AMAXU; ++u)
      if (u < nu) dQr[kl + 16 * u] = from_f<T>(acc[u] * a.scale);
  }
}

// backward, kernel B (per 64-key chunk): dK = scale * dS^T Q ;  dV = gate * P^T dO
template <typename T, typename TP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnB a) {
  extern __shared__ float sm[];
  const int dh = a.dh, Lk = a.Lk;
  float* Qs = sm;                        // [16][dh+1]
  float* dOs = Qs + AQ * (dh + 1);       // [16][dh+1]
  float* dSs = dOs + AQ * (dh + 1);      // [16][65]
  float* Ps = dSs + AQ * (AKC + 1);      // [16][65]
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * AKC;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const int tid = threadIdx.x, key = tid >> 2, dq = tid & 3;
  const int per = dh >> 2;               // d values per thread (contiguous block)
  const T* Qb = reinterpret_cast<const T*>(a.Q) + (size_t)b * a.Lq * a.ldq + h * dh;
  const T* dOb = reinterpret_cast<const T*>(a.dO) + (size_t)b * a.Lq * a.ldo + h * dh;
  const size_t pbase = ((size_t)b * a.H + h) * a.Lq * a.ldpr;
  const TP* Pb = reinterpret_cast<const TP*>(a.P) + pbase;
  const T* dSb = reinterpret_cast<const T*>(a.dS) + pbase;
  float ak[32], av[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) { ak[e] = 0.f; av[e] = 0.f; }
  const bool drop = a.drop_p > 0.f;
  DropRng rng;
  if (drop) rng = drop_rng(a.rng, a.call, a.drop_p);
  for (int q0 = 0; q0 < a.Lq; q0 += AQ) {
    __syncthreads();
    load_rows<T>(Qb, a.ldq, q0, AQ, a.Lq, dh, Qs);
    load_rows<T>(dOb, a.ldo, q0, AQ, a.Lq, dh, dOs);
    for (int id = tid; id < AQ * AKC; id += 256) {
      const int rr = id >> 6, kk = id & 63;
      float ds = 0.f, p = 0.f;
      if (q0 + rr < a.Lq && k0 + kk < Lk) {
        ds = to_f(dSb[(size_t)(q0 + rr) * a.ldpr + k0 + kk]);
        p = to_f(Pb[(size_t)(q0 + rr) * a.ldpr + k0 + kk]);
        if (drop) p *= drop_factor(rng, (((uint64_t)b * a.H + h) * a.Lq + (q0 + rr)) * (uint64_t)((Lk + 7) & ~7) + k0 + kk);   // dV = gate (P .* M)^T dO
      }
      dSs[rr * (AKC + 1) + kk] = ds;
      Ps[rr * (AKC + 1) + kk] = p;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < AQ; ++rr) {
      const float ds = dSs[rr * (AKC + 1) + key], p = Ps[rr * (AKC + 1) + key];
#pragma unroll
      for (int e = 0; e < 32; ++e)
        if (e < per) {
          ak[e] = fmaf(ds, Qs[rr * (dh + 1) + dq * per + e], ak[e]);
          av[e] = fmaf(p, dOs[rr * (dh + 1) + dq * per + e], av[e]);
        }
    }
  }
  if (k0 + key < Lk) {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    const size_t row = (size_t)bkv * Lk + k0 + key;
    if (a.dK32) {
      float* dk = a.dK32 + row * a.lddk + h * dh + dq * per;
      float* dv = a.dV32 + row * a.lddv + h * dh + dq * per;
#pragma unroll
      for (int e = 0; e < 32; ++e)
        if (e < per) { atomicAdd(dk + e, ak[e] * a.scale); atomicAdd(dv + e, av[e] * gz); }
    } else {
      T* dk = reinterpret_cast<T*>(a.dK) + row * a.lddk + h * dh + dq * per;
      T* dv = reinterpret_cast<T*>(a.dV) + row * a.lddv + h * dh + dq * per;
#pragma unroll
      for (int e = 0; e < 32; ++e)
        if (e < per) { dk[e] = from_f<T>(ak[e] * a.scale); dv[e] = from_f<T>(av[e] * gz); }
    }
  }
}

static int attn_check(int dtype, int p_dtype, int dh, const char* name) {
  if (dtype != EVLM_F32 && dtype != EVLM_BF16) return evlm_set_error("%s: bad dtype", name);
  if (p_dtype != EVLM_F32 && p_dtype != EVLM_BF16) return evlm_set_error("%s: bad p_dtype", name);
  if (dh % 16 != 0 || dh > 128 || dh <= 0) return evlm_set_error("%s: head dim %d unsupported (multiple of 16, <= 128)", name, dh);
  return 0;
}

#define ATTN_DISPATCH(dt, pdt, KERNEL, grid, block, lds, stream, arg)                                         \
  if (dt == EVLM_F32 && pdt == EVLM_F32) hipLaunchKernelGGL((KERNEL<float, float>), grid, block, lds, stream, arg); \
  else if (dt == EVLM_BF16 && pdt == EVLM_BF16) hipLaunchKernelGGL((KERNEL<bf16, bf16>), grid, block, lds, stream, arg); \
  else if (dt == EVLM_BF16 && pdt == EVLM_F32) hipLaunchKernelGGL((KERNEL<bf16, float>), grid, block, lds, stream, arg); \
  else hipLaunchKernelGGL((KERNEL<float, bf16>), grid, block, lds, stream, arg);

int evlm_attention_fwd_mfma(const evlm_attn_fwd_args* a, hipStream_t stream, int* handled);   // attention_mfma.hip
int evlm_attention_bwd_mfma(const evlm_attn_bwd_args* a, hipStream_t stream, int* handled);

extern "C" int evlm_attention_fwd(const evlm_attn_fwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(a && a->Q && a->K && a->V && a->O, "evlm_attention_fwd: null operand");
  if (int e = attn_check(a->dtype, a->p_dtype, a->dh, "evlm_attention_fwd")) return e;
  EVLM_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0, "evlm_attention_fwd: bad shape");
  EVLM_REQUIRE((a->ldq | a->ldk | a->ldv) % 8 == 0, "evlm_attention_fwd: row strides must be multiples of 8");
  EVLM_REQUIRE(!a->P || (a->ldpr >= a->Lk && a->ldpr % 8 == 0), "evlm_attention_fwd: ldpr must be a multiple of 8 and >= Lk");
  EVLM_REQUIRE(!a->causal || a->Lq == a->Lk, "evlm_attention_fwd: a causal mask needs Lq == Lk");
  EVLM_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f && (a->dropout_p == 0.f || a->rng_state),
               "evlm_attention_fwd: dropout_p = %f needs 0 <= p < 1 and an rng_state", (double)a->dropout_p);
  int handled = 0;
  if (int e = evlm_attention_fwd_mfma(a, stream, &handled)) return e;     // (round 6: with or without probability dropout)
  if (handled) return 0;
  EVLM_REQUIRE(!a->lse, "evlm_attention_fwd: the lse form exists on the bf16 MFMA path only (evlm_attention_lse_supported)");
  EVLM_REQUIRE(!a->kd_teacher, "evlm_attention_fwd: the fused map distillation exists on the bf16 MFMA path only "
               "(head dim 64, Lk <= 928); use evlm_mse_fwd on the returned map");
  AttnF f;
  f.drop_p = a->dropout_p; f.rng = a->rng_state; f.call = a->call_id;
  f.Q = a->Q; f.K = a->K; f.V = a->V; f.kv_index = a->kv_index; f.mask = a->mask; f.gate = a->head_gate; f.causal = a->causal;
  f.O = a->O; f.P = a->P; f.B = a->B; f.H = a->H; f.Lq = a->Lq; f.Lk = a->Lk; f.dh = a->dh;
  f.ldq = a->ldq; f.ldk = a->ldk; f.ldv = a->ldv; f.ldo = a->ldo; f.ldpr = a->ldpr; f.scale = a->scale;
  const int Lkp = (a->Lk + 3) & ~3;
  const size_t lds = sizeof(float) * ((size_t)(AQ + AKC) * (a->dh + 1) + (size_t)AQ * Lkp);
  EVLM_REQUIRE(lds <= 160 * 1024, "evlm_attention_fwd: Lk=%d too long for the LDS score block", a->Lk);
  dim3 grid(ceil_div(a->Lq, AQ), a->H, a->B), block(256);
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<float, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  ATTN_DISPATCH(a->dtype, a->p_dtype, attn_fwd_kernel, grid, block, lds, stream, f)
  EVLM_LAUNCH_CHECK("evlm_attention_fwd");
  return 0;
}

extern "C" int evlm_attention_bwd(const evlm_attn_bwd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(a && a->Q && a->K && a->V && (a->P || a->lse) && a->dO && a->dQ, "evlm_attention_bwd: null operand");
  EVLM_REQUIRE(a->dK && a->dV, "evlm_attention_bwd: dK/dV required (f32 accumulators when kv_index is set)");
  if (int e = attn_check(a->dtype, a->p_dtype, a->dh, "evlm_attention_bwd")) return e;
  EVLM_REQUIRE(a->ldpr >= a->Lk && a->ldpr % 8 == 0, "evlm_attention_bwd: ldpr must be a multiple of 8 and >= Lk");
  EVLM_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f && (a->dropout_p == 0.f || a->rng_state),
               "evlm_attention_bwd: dropout_p = %f needs 0 <= p < 1 and an rng_state", (double)a->dropout_p);
  {
    int handled = 0;
    if (int e = evlm_attention_bwd_mfma(a, stream, &handled)) return e;
    if (handled) return 0;
  }
  EVLM_REQUIRE(!a->kd_teacher, "evlm_attention_bwd: the fused map distillation exists on the bf16 MFMA path only");
  EVLM_REQUIRE(!a->lse && a->P, "evlm_attention_bwd: the recomputing (lse) form exists on the bf16 MFMA path only "
               "(evlm_attention_lse_supported); this problem needs the stored map P");
  EVLM_REQUIRE(a->dS, "evlm_attention_bwd: this problem needs the dS workspace");
  AttnB g;
  g.drop_p = a->dropout_p; g.rng = a->rng_state; g.call = a->call_id;
  g.Q = a->Q; g.K = a->K; g.V = a->V; g.P = a->P; g.dO = a->dO; g.dPext = a->dP_ext; g.kv_index = a->kv_index;
  g.gate = a->head_gate; g.dS = a->dS; g.dQ = a->dQ; g.dgate = a->dgate;
  if (a->kv_index) { g.dK32 = (float*)a->dK; g.dV32 = (float*)a->dV; g.dK = nullptr; g.dV = nullptr; }
  else { g.dK32 = nullptr; g.dV32 = nullptr; g.dK = a->dK; g.dV = a->dV; }
  g.B = a->B; g.H = a->H; g.Lq = a->Lq; g.Lk = a->Lk; g.dh = a->dh;
  g.ldq = a->ldq; g.ldk = a->ldk; g.ldv = a->ldv; g.ldo = a->ldo; g.lddq = a->lddq; g.lddk = a->lddk; g.lddv = a->lddv;
  g.scale = a->scale; g.ldpr = a->ldpr;
  const int Lkp = (a->Lk + 3) & ~3;
  const size_t ldsA = sizeof(float) * ((size_t)(AQ + AKC) * (a->dh + 1) + (size_t)AQ * Lkp);
  EVLM_REQUIRE(ldsA <= 160 * 1024, "evlm_attention_bwd: Lk=%d too long", a->Lk);
  if (ldsA > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<bf16, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<bf16, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<float, bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 gridA(ceil_div(a->Lq, AQ), a->H, a->B), block(256);
  ATTN_DISPATCH(a->dtype, a->p_dtype, attn_bwd_dq_kernel, gridA, block, ldsA, stream, g)
  EVLM_LAUNCH_CHECK("evlm_attention_bwd(dq)");
  const size_t ldsB = sizeof(float) * ((size_t)2 * AQ * (a->dh + 1) + (size_t)2 * AQ * (AKC + 1));
  dim3 gridB(ceil_div(a->Lk, AKC), a->H, a->B);
  ATTN_DISPATCH(a->dtype, a->p_dtype, attn_bwd_dkv_kernel, gridB, block, ldsB, stream, g)
  EVLM_LAUNCH_CHECK("evlm_attention_bwd(dkv)");
  return 0;
}
