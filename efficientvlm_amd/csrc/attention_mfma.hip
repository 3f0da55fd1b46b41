// bf16 MFMA attention kernels for gfx950 (head dim 64): the fast path behind evlm_attention_fwd / _bwd.
//
// Everything is computed TRANSPOSED (keys on the MFMA row index, queries on the lane's column index):
//   S^T = K Q^T   -> each lane owns ONE query (lane & 15) and 4 keys per 16x16 tile, so a softmax row lives in one
//                    lane's registers plus a 2-step shuffle over the 4 lane groups (no LDS round trip);
//   O^T = V^T P^T -> the P^T accumulators are reused AS the MFMA B operand (same key<->k-slot assignment on both
//                    operands), V^T comes from a row-major [key][d] LDS tile through ds_read_b64_tr_b16.
// The probability map is an OUTPUT (the KD losses read it): rows are written once, with stride ldpr (padding zeroed).
#include "common.h"
#include <type_traits>

struct MAttnF {
  const bf16* Q; const bf16* K; const bf16* V; const int32_t* kv_index; const float* mask; const float* gate;
  bf16* O; bf16* P;
  int B, H, Lq, Lk, ldq, ldk, ldv, ldo, ldpr;
  float scale;
  int causal;
  const bf16* Pt; float* kd; float kd_coef;      // fused map distillation: *kd += kd_coef * sum((P - Pt)^2)
  float* lse;                                    // [B, H, Lq] log2-sum-exp of the scaled, masked scores (recomputing backward)
  float* rkd;                                    // [B, H, Lq] sum_k P (P - Pt) of the fused distillation (one-pass long backward)
  int skip_dead;                                 // heads with a gate of exactly 0: zero context, nothing staged (A/B switch)
  // ABI 8: the teacher's map REBUILT in the kernel instead of read (streaming kernels): its projected queries / keys
  // ([B, L, H, dh] bf16 inside its packed QKV buffer, row stride tld) and its row lse [B, H, L]
  const bf16* Tq; const bf16* Tk; int tld; const float* tlse;
  // round 6: dropout of the probabilities that form the context (eff_bert.py:346; the map / lse / distillation taps see the
  // un-dropped softmax).  DROP kernels regenerate keep / (1 - p) for the lane's 8 consecutive keys of a tile pair with ONE
  // Philox call (common.h: drop_factor8)
  float drop_p; const int64_t* rng; uint32_t call;
};

#define DH 64
extern "C" int evlm_attention_lse_supported(int dtype, int dh, int Lk, float dropout_p);
#define LOG2E 1.44269504088896341f
// 2^x for x <= 0 (softmax numerators, recomputed probabilities): the bare v_exp_f32 (1 ulp; results below 2^-126 flush to
// zero) instead of exp2f()'s range handling - 8 VALU issue slots per element saved in kernels that are VALU-bound
#ifdef EVLM_SLOW_EXP2
#define EXP2(x) exp2f(x)
#else
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

// 16-byte chunk swizzles of the two row-major [key][64] bf16 tiles (128-byte rows)
__device__ __forceinline__ int k_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }          // ds_read_b128 rows
__device__ __forceinline__ int v_swz(int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); }   // tr16 column reads
// both at once (the recomputing backward reads its K tile by rows for S^T = K Q^T AND by columns for dQ^T = K^T dS^T):
// the column reads of a half-wave touch rows 8i .. 8i+7, whose (row >> 3) bit is constant - the chunk PAIR slots stay
// those of v_swz -, and the 16 rows of a b128 row read get 16 distinct (row parity, chunk slot) bank groups
__device__ __forceinline__ int kv_swz(int row, int chunk) { return chunk ^ ((((row >> 1) & 3) << 1) | ((row >> 3) & 1)); }
enum { SW_K = 0, SW_V = 1, SW_KV = 2 };
template <int SW> __device__ __forceinline__ int swz(int row, int chunk) {
  return SW == SW_K ? k_swz(row, chunk) : (SW == SW_V ? v_swz(row, chunk) : kv_swz(row, chunk));
}

// Key order inside the LDS tiles.  Keys are PERMUTED within every block of 32: key 32s + 8g + 4h + r sits in LDS row
// (2s + h)*16 + 4g + r, i.e. MFMA tile 2s + h, row 4g + r.  An accumulator lane (g = lane >> 4) of the tile pair (2s, 2s+1)
// then holds keys 32s + 8g + 0..3 (tile 2s) and + 4..7 (tile 2s+1): EIGHT CONSECUTIVE keys, so the probability map, the
// external dP and dS move as 16-byte pieces (64 contiguous bytes per row per wave instruction) instead of 8-byte ones,
// and the k-slot <-> key map of the second MFMA's B operand becomes the natural one.  Fragment reads are unchanged.
__device__ __forceinline__ int key_row(int key) {
  return (((key >> 5) << 1) | ((key >> 2) & 1)) * 16 + ((key >> 3) & 3) * 4 + (key & 3);
}
// first key of accumulator tile t for lane group g (keys key0 .. key0 + 3)
__device__ __forceinline__ int tile_key0(int t, int g) { return (t >> 1) * 32 + g * 8 + (t & 1) * 4; }

// stage keys [0, nrows) of a [L][.. ld ..] tensor (64 columns at `base`) into a swizzled LDS tile; keys >= L are zero
// inverse of key_row: the key held by LDS row `row`
__device__ __forceinline__ int row_key(int row) {
  const int t = row >> 4;
  return (t >> 1) * 32 + ((row >> 2) & 3) * 8 + (t & 1) * 4 + (row & 3);
}

// Stage keys [0, nrows) of a [L][.. ld ..] tensor (64 columns at `base`) into a swizzled LDS tile by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, every instruction of a wave in flight at once).  One wave instruction
// fills 8 consecutive 128-byte LDS rows, lane-linear; the key permutation and the chunk swizzle (XOR: its own inverse)
// are applied to each lane's SOURCE address.  Keys >= L read key L-1: their scores carry the -1e30 of the mask row /
// their probabilities are zero, so any finite value serves.  The caller waits (stage_wait) before its barrier.
template <int SW>
__device__ __forceinline__ void stage_rows(const bf16* base, int ld, int L, int nrows, char* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r0 = wave * 8; r0 < nrows; r0 += nw * 8) {
    const int row = r0 + (lane >> 3), cs = lane & 7;
    const int c = swz<SW>(row, cs);
    const int key = min(row_key(row), L - 1);
    const bf16* src = base + (size_t)key * ld + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sm + r0 * 128), 16, 0, 0);
  }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A/B operand fragment (rows = keys, k = head dim) from a k_swz tile: row = 16*t + (lane&15), k = 32*ks + 8*(lane>>4)..+7
template <int SW = SW_K>
__device__ __forceinline__ bf16x8 krow_frag(const char* sm, int t, int ks, int lane) {
  const int row = t * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(sm + row * 128 + swz<SW>(row, c) * 16);
}
// A operand = (tile)^T for a sum over KEYS: row index d = 16*dt + (lane&15); k-slots j=0..3 <-> keys 16*t0+4g+j,
// j=4..7 <-> keys 16*t1+4g+(j-4)   (g = lane>>4).  `sm` is a v_swz tile.
template <int SW = SW_V>
__device__ __forceinline__ bf16x8 vcol_frag(const char* sm, int t0, int t1, int dt, int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  bf16x8 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = (h ? t1 : t0) * 16 + g * 4 + q;
    const int off = row * 128 + swz<SW>(row, dt * 2 + (p >> 1)) * 16 + ((p & 1) << 3);
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sm + off));
    out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
  }
  return out;
}

// The same fragment through INLINE-ASM transposing reads (round 5), for kernels that keep an LDS-DMA in flight while they
// read another buffer: hipcc models an LDS-DMA as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 while one is outstanding (found on the GEMM side in round 4, gemm_pp256_core.h:
// pp_frag) - the streaming attention kernels' "next block under the current block's arithmetic" ended at the first V / K^T
// fragment read of the block.  The asm reads are invisible to that pass AND to the compiler's lgkmcnt bookkeeping: the
// caller issues TR_WAIT() (s_waitcnt lgkmcnt(0) between two sched_barriers) before the MFMAs that consume them.  Counted
// waits the compiler emits for its own LDS reads stay correct: LDS operations complete in order, so extra operations in
// the queue can only make such a wait longer, never shorter.
template <int SW = SW_V>
__device__ __forceinline__ bf16x8 vcol_frag_a(const char* sm, int t0, int t1, int dt, int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  bf16x8 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = (h ? t1 : t0) * 16 + g * 4 + q;
    const uint32_t off = sb + row * 128 + swz<SW>(row, dt * 2 + (p >> 1)) * 16 + ((p & 1) << 3);
    bf16x4 t;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t) : "v"(off));
    out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
  }
  return out;
}
// (Measured and NOT kept, round 5: the teacher-map pieces of the streaming kernels as inline-asm global loads with counted
// waits - hipcc answers a wait for a tracked load that is older than an LDS-DMA with vmcnt(0) whatever it knows about the
// DMA count, draining the next block's staging at the piece's first use.  The asm form ran 2 % SLOWER on the forward with
// fused distillation (266 against 261 us at 577 keys) and level on forward + backward, and an asm load's destination must
// not be copied before its wait - a loop-carried set of pieces produced exactly that copy, and a memory fault.)
#define TR_WAIT()                                          \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

// 16-byte LDS reads through inline asm with an immediate offset (round 6, the batched form of the streaming backward: every
// read of a tile pair is issued up front and the consumers sit behind COUNTED waits - LDS operations complete in order).
// Same contract as vcol_frag_a: the destination must not be touched before the wait that covers it (tools/check_asm_loads.py).
template <int OFF> __device__ __forceinline__ bf16x8 lds_b128_a(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_f32x4_a(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int OFF0, int OFF1> __device__ __forceinline__ bf16x8 tr_two_a(uint32_t addr) {   // vcol_frag_a with immediate offsets
  bf16x4 t0, t1;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t0) : "v"(addr), "i"(OFF0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t1) : "v"(addr), "i"(OFF1));
  bf16x8 out;
  out[0] = t0[0]; out[1] = t0[1]; out[2] = t0[2]; out[3] = t0[3];
  out[4] = t1[0]; out[5] = t1[1]; out[6] = t1[2]; out[7] = t1[3];
  return out;
}
#define LGKM_WAIT(n)                                               \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)const_cast<char*>(p); }

// NT = number of 16-key tiles (even).  One workgroup = up to 16 waves x 16 queries of one (batch, head): with Lq <= 256
// (the ViT's 197 tokens) a single workgroup covers every query, so K and V are staged into LDS exactly once per
// (batch, head) instead of once per 64-query block on four different XCDs.
// SEQ (long key sequences: 480x480 images = 901 tokens): K and V do not fit in LDS together, so they take turns in ONE
// region - K for the scores, then (after the probabilities are in registers) V for P V.  Waves past the last query stay
// for the barriers.
// LSE: the per-row log2-sum-exp is written for a backward that recomputes P in fp32 (the map itself is then written only
// when a caller wants it), and the fused map distillation compares the fp32 probabilities - the ones that backward will
// rebuild - with the teacher map, before the P V product (while the un-normalised row is still in registers).
template <int NT, int MAXW, bool SEQ, bool LSE, bool DROP = false>
__global__ __launch_bounds__(64 * MAXW) void attn_fwd_mfma_kernel(MAttnF a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                                       // [NT*16][64] bf16, k_swz
  char* Vs = SEQ ? smem : smem + NT * 16 * 128;          // [NT*16][64] bf16, v_swz
  float* Ms = reinterpret_cast<float*>(smem + (SEQ ? 1 : 2) * NT * 16 * 128);   // [NT*16] additive mask (+ -1e30 beyond Lk)
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  // A head whose L0 gate is exactly 0 (masked-dense evaluation of a pruned-but-not-compacted model, BASELINE configs[4]:
  // "kernel skips masked heads") contributes a zero context: when nothing else is asked of the launch - no map, no row
  // lse for a backward, no distillation term - the workgroup writes the zeros and leaves before staging anything.
  // (Workgroup-uniform: ahead of every barrier.)
  if (!LSE && a.skip_dead && a.gate && !a.P && !a.Pt && a.gate[h] == 0.f) {
    const int q = (blockIdx.x * (blockDim.x >> 6) + wave) * 16 + ql;
    if (q < a.Lq) {
      bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH + g * 16;
      *reinterpret_cast<uint4*>(Or) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(Or + 8) = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const bf16* Kb = a.K + (size_t)bkv * a.Lk * a.ldk + h * DH;
  const bf16* Vb = a.V + (size_t)bkv * a.Lk * a.ldv + h * DH;
  stage_rows<SW_K>(Kb, a.ldk, a.Lk, NT * 16, Ks);
  if (!SEQ) stage_rows<SW_V>(Vb, a.ldv, a.Lk, NT * 16, Vs);
  for (int k = threadIdx.x; k < NT * 16; k += blockDim.x)
    Ms[k] = ((k < a.Lk) ? (a.mask ? a.mask[(size_t)b * a.Lk + k] : 0.f) : -1e30f) * LOG2E;      // (log2 domain, as stage_mask)
  float* kdw = Ms + NT * 16;                     // {partial sum, arrived waves} of the fused map distillation
  if (threadIdx.x < 2) kdw[threadIdx.x] = 0.f;
  stage_wait();
  __syncthreads();

  const int q0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 16;
  if (!SEQ && q0 >= a.Lq) return;
  const int q = q0 + ql;
  const bool qok = q < a.Lq;
  // B operand = Q^T: 8 consecutive head-dim values of this lane's query
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qok) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
    qf[ks] = *reinterpret_cast<bf16x8*>(&v);
  }
  // S^T tiles: acc[t][r] = S[q][key = tile_key0(t, g) + r]
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Ks, t, ks, lane), qf[ks], acc[t], 0, 0, 0);
  }
  // softmax over the keys of this lane's query: registers + the 4 lane groups
  const float sc = a.scale * 1.44269504088896341f;   // exp(x) = exp2(x * log2 e)
  float m = -3.0e38f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + tile_key0(t, g));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (round 6: the mask strip is staged already multiplied by log2 e - the same fp32 product, formed once per key instead
      // of once per (query, key); min commutes with a multiplication by a positive constant, bit for bit)
      float add = mk[r];
      if (a.causal && tile_key0(t, g) + r > q) add = fminf(add, -10000.0f * LOG2E);   // decoder: keys after the query
      acc[t][r] = fmaf(acc[t][r], sc, add);             // (explicit fma: the recomputing backward forms the same number)
      m = fmaxf(m, acc[t][r]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[t][r] = EXP2(acc[t][r] - m);
      sum += acc[t][r];
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (LSE && qok && g == 0) a.lse[((size_t)b * a.H + h) * a.Lq + q] = m + __log2f(sum);
  float sq = 0.f;
  if (LSE && a.Pt && qok) {
    const bf16* Tr = a.Pt + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr;
    float rk = 0.f;                                       // sum_k p (p - pt): the distillation term's share of the
#pragma unroll                                            // backward's row sum delta (one-pass long-sequence backward)
    for (int s = 0; s < NT / 2; ++s) {
      const int kcol = s * 32 + g * 8;
      if (kcol < a.ldpr) {
        const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(Tr + kcol);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = acc[2 * s][r] * inv, p1 = acc[2 * s + 1][r] * inv;
          const float d0 = p0 - (float)t8[r], d1 = p1 - (float)t8[4 + r];
          sq = fmaf(d0, d0, sq);
          sq = fmaf(d1, d1, sq);
          rk = fmaf(p0, d0, rk);
          rk = fmaf(p1, d1, rk);
        }
      }
    }
    if (a.rkd) {
      rk += __shfl_xor(rk, 16, 64);
      rk += __shfl_xor(rk, 32, 64);
      if (g == 0) a.rkd[((size_t)b * a.H + h) * a.Lq + q] = rk;
    }
  }
  // P (bf16): 4 consecutive keys per lane per tile -> 8-byte stores; also the PV B operand
  bf16x4 pk[NT];
  bf16x4 pd[DROP ? NT : 1];                      // DROP: the PV operand is P .* keep / (1 - p), rounded once from fp32
  bf16* Pr = a.P ? a.P + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr : nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[t][r] = (bf16)(acc[t][r] * inv);
  }
  if (DROP) {
    const DropRng rng = drop_rng(a.rng, a.call, a.drop_p);
    const uint64_t d8 = (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
#pragma unroll
    for (int s = 0; s < NT / 2; ++s) {
      float f[8];
      drop_factor8(rng, d8 + s * 4, f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pd[DROP ? 2 * s : 0][r] = (bf16)(acc[2 * s][r] * inv * f[r]);
        pd[DROP ? 2 * s + 1 : 0][r] = (bf16)(acc[2 * s + 1][r] * inv * f[4 + r]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) {            // 8 consecutive keys per lane: one 16-byte store per tile pair
    const int kcol = s * 32 + g * 8;
    if (Pr && qok && kcol < a.ldpr) {
      bf16x8 pp;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pp[r] = pk[2 * s][r]; pp[4 + r] = pk[2 * s + 1][r]; }
      *reinterpret_cast<bf16x8*>(Pr + kcol) = pp;
    }
  }
  if (SEQ) {                                             // the scores are done with K: V takes its place
    __syncthreads();
    stage_rows<SW_V>(Vb, a.ldv, a.Lk, NT * 16, Vs);
    stage_wait();
    __syncthreads();
  }
  // O^T[d][q] = sum_key V[key][d] * P[q][key]
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) {
    bf16x8 pb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pb[r] = DROP ? pd[DROP ? 2 * s : 0][r] : pk[2 * s][r];
      pb[4 + r] = DROP ? pd[DROP ? 2 * s + 1 : 0][r] : pk[2 * s + 1][r];
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcol_frag(Vs, 2 * s, 2 * s + 1, dt, lane), pb, o[dt], 0, 0, 0);
  }
  if (qok) {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ov = {(bf16)(o[dt][0] * gz), (bf16)(o[dt][1] * gz), (bf16)(o[dt][2] * gz), (bf16)(o[dt][3] * gz)};
      *reinterpret_cast<bf16x4*>(Or + dt * 16 + g * 4) = ov;
    }
  }
  if (a.Pt) {
    // attention-map distillation while the (bf16-rounded, as stored) probabilities are still in registers
    // (placed after the P V product: fewest live registers): the teacher's map is
    // read once, the student's not at all; one atomic per WORKGROUP (the waves meet in LDS, last arriver publishes)
    if (!LSE && qok) {
      const bf16* Tr = a.Pt + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr;
#pragma unroll
      for (int s = 0; s < NT / 2; ++s) {
        const int kcol = s * 32 + g * 8;
        if (kcol < a.ldpr) {
          const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(Tr + kcol);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d0 = (float)pk[2 * s][r] - (float)t8[r], d1 = (float)pk[2 * s + 1][r] - (float)t8[4 + r];
            sq = fmaf(d0, d0, sq);
            sq = fmaf(d1, d1, sq);
          }
        }
      }
    }
    sq = wave_sum(sq);
    if (lane == 0) {
      const int qb = blockIdx.x * (blockDim.x >> 6) * 16;
      const int active = SEQ ? (int)(blockDim.x >> 6) : min((int)(blockDim.x >> 6), (a.Lq - qb + 15) / 16);
      atomicAdd(&kdw[0], sq);                    // (LDS operations of one wave execute in order: the sum lands before the count)
      const float before = atomicAdd(&kdw[1], 1.0f);
      if ((int)before == active - 1) atomicAdd(a.kd, atomicAdd(&kdw[0], 0.f) * a.kd_coef);
    }
  }
}

// Cross-attention with a shared K/V index (kv_index: the image tokens of the positive / hard-negative / MLM fusion rows):
// ONE workgroup per (K/V row, head) stages K and V once and serves EVERY query batch that attends to them - found with a
// ballot over the index, 64 entries at a time; the (query batch, 16-query tile) tasks go round the waves.  The per-batch
// kernel above stages the same 50 KiB once per text row (4x per image in the GD step) with two waves per workgroup to use
// them.  Each wave keeps the additive mask row of its current query batch in a private LDS strip.
template <int NT, int NW, bool LSE, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd_grouped_kernel(MAttnF a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                                       // [NT*16][64] bf16, k_swz
  char* Vs = smem + NT * 16 * 128;                       // [NT*16][64] bf16, v_swz
  const int bkv = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  float* Ms = reinterpret_cast<float*>(smem + 2 * NT * 16 * 128) + wave * NT * 16;     // this wave's mask strip
  const bool dead = !LSE && a.skip_dead && a.gate && !a.P && a.gate[h] == 0.f;      // closed head, nothing but a zero context to deliver
  if (!dead) {
    stage_rows<SW_K>(a.K + (size_t)bkv * a.Lk * a.ldk + h * DH, a.ldk, a.Lk, NT * 16, Ks);
    stage_rows<SW_V>(a.V + (size_t)bkv * a.Lk * a.ldv + h * DH, a.ldv, a.Lk, NT * 16, Vs);
    stage_wait();
  }
  __syncthreads();
  const int qtiles = (a.Lq + 15) >> 4;
  const float sc = a.scale * 1.44269504088896341f;
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
  int task = 0;                                          // running (query batch, tile) counter: wave w takes task % NW == w
  for (int b0 = 0; b0 < a.B; b0 += 64) {
    unsigned long long hits = __ballot(b0 + lane < a.B && a.kv_index[min(b0 + lane, a.B - 1)] == bkv);
    while (hits) {                                       // wave-uniform, identical in every wave
      const int b = b0 + __builtin_amdgcn_readfirstlane(__ffsll((long long)hits) - 1);
      hits &= hits - 1;
      bool mask_ready = false;
      for (int qt = 0; qt < qtiles; ++qt, ++task) {
        if (task % NW != wave) continue;
        if (dead) {                                      // (wave-uniform)
          const int q = qt * 16 + ql;
          if (q < a.Lq) {
            bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH + g * 16;
            *reinterpret_cast<uint4*>(Or) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(Or + 8) = make_uint4(0, 0, 0, 0);
          }
          continue;
        }
        if (!mask_ready) {                               // (in-order LDS: the strip is complete before this wave reads it)
          for (int k = lane; k < NT * 16; k += 64)
            Ms[k] = ((k < a.Lk) ? (a.mask ? a.mask[(size_t)b * a.Lk + k] : 0.f) : -1e30f) * LOG2E;      // (log2 domain, as stage_mask)
          mask_ready = true;
        }
        const int q = qt * 16 + ql;
        const bool qok = q < a.Lq;
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (qok) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
          qf[ks] = *reinterpret_cast<bf16x8*>(&v);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Ks, t, ks, lane), qf[ks], acc[t], 0, 0, 0);
        }
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + tile_key0(t, g));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[t][r] = fmaf(acc[t][r], sc, mk[r]);
            m = fmaxf(m, acc[t][r]);
          }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[t][r] = EXP2(acc[t][r] - m);
            sum += acc[t][r];
          }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (LSE && qok && g == 0) a.lse[((size_t)b * a.H + h) * a.Lq + q] = m + __log2f(sum);
        bf16x4 pk[NT];
        bf16x4 pd[DROP ? NT : 1];                        // DROP: the PV operand P .* keep / (1 - p), rounded once from fp32
        bf16* Pr = a.P ? a.P + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr : nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[t][r] = (bf16)(acc[t][r] * inv);
        }
        if (DROP) {
          const uint64_t d8 = (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
#pragma unroll
          for (int s = 0; s < NT / 2; ++s) {
            float f[8];
            drop_factor8(rng, d8 + s * 4, f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pd[DROP ? 2 * s : 0][r] = (bf16)(acc[2 * s][r] * inv * f[r]);
              pd[DROP ? 2 * s + 1 : 0][r] = (bf16)(acc[2 * s + 1][r] * inv * f[4 + r]);
            }
          }
        }
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
          const int kcol = s * 32 + g * 8;
          if (Pr && qok && kcol < a.ldpr) {
            bf16x8 pp;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pp[r] = pk[2 * s][r]; pp[4 + r] = pk[2 * s + 1][r]; }
            *reinterpret_cast<bf16x8*>(Pr + kcol) = pp;
          }
        }
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
          bf16x8 pb;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pb[r] = DROP ? pd[DROP ? 2 * s : 0][r] : pk[2 * s][r];
            pb[4 + r] = DROP ? pd[DROP ? 2 * s + 1 : 0][r] : pk[2 * s + 1][r];
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcol_frag(Vs, 2 * s, 2 * s + 1, dt, lane), pb, o[dt], 0, 0, 0);
        }
        if (qok) {
          const float gz = a.gate ? a.gate[h] : 1.0f;
          bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ov = {(bf16)(o[dt][0] * gz), (bf16)(o[dt][1] * gz), (bf16)(o[dt][2] * gz), (bf16)(o[dt][3] * gz)};
            *reinterpret_cast<bf16x4*>(Or + dt * 16 + g * 4) = ov;
          }
        }
      }
    }
  }
}

#ifdef EVLM_EXPERIMENTAL_ATTN_PERSIST      // (make EXPERIMENTAL=1: 23.7 against 21.2 us on the GD shape - profiles/r05_xattn_persist.md)
// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the grouped cross-attention forward (round 5).  The kernel above is one workgroup per (K/V row, head):
// it stages 56 KiB, waits for them, then computes - two workgroups per CU, 768 of them on 512 slots, and while a workgroup
// computes nothing of its CU-share of the 62 MB is in flight (21 us = 2.9 TB/s on the GD shape).  Here ONE 8-wave workgroup
// per CU walks its items i, i + grid, i + 2 grid ... with K / V DOUBLE-BUFFERED: right behind the barrier that opens item
// i the LDS-DMA of item i + 1 goes out into the other buffer and stays in flight under item i's whole arithmetic, and the
// Q fragments of a wave's first task of item i + 1 are requested just ahead of that DMA (vmcnt counts in order: a load
// issued BEHIND the DMA could only be waited for together with it).  kv_index is copied into LDS once; a wave finds the
// batches of its tasks t = wave, wave + NW, ... with ballots over that copy.
// The transposing V reads are inline asm: hipcc models an LDS-DMA as a pending LDS write and puts `s_waitcnt vmcnt(0)` in
// front of every __builtin_amdgcn_ds_read_tr16_b64 while one is outstanding (gemm_pp256_core.h: pp_frag) - which here
// would drain the NEXT item's staging before the P V product of the current one.  RAW: buffer i & 1 is read only behind
// `s_waitcnt vmcnt(0)` + barrier of iteration i; WAR: its next DMA (item i + 2) is issued behind the barrier of iteration
// i + 1, which every wave reaches only after its last read of item i.  Same arithmetic in the same order as the kernels
// above: bit-identical outputs (tests/test_ops_gpu.py).
__device__ __forceinline__ bf16x8 vcol_frag_asm(const char* sm, int t0, int t1, int dt, int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  bf16x8 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = (h ? t1 : t0) * 16 + g * 4 + q;
    const uint32_t off = sb + row * 128 + v_swz(row, dt * 2 + (p >> 1)) * 16 + ((p & 1) << 3);
    bf16x4 t;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t) : "v"(off));
    out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
  }
  return out;
}

// the n-th query batch (0-based, index order) that attends to K/V row `bkv`, or -1 (wave-uniform: n, result)
__device__ __forceinline__ int nth_hit(const int* kvs, int B, int bkv, int n, int lane) {
  for (int b0 = 0; b0 < B; b0 += 64) {
    unsigned long long hits = __ballot(b0 + lane < B && kvs[min(b0 + lane, B - 1)] == bkv);
    const int c = __popcll(hits);
    if (n < c) {
      for (int i = 0; i < n; ++i) hits &= hits - 1;
      return b0 + __builtin_amdgcn_readfirstlane(__ffsll((long long)hits) - 1);
    }
    n -= c;
  }
  return -1;
}

template <int NT, int NW, bool LSE>
__global__ __launch_bounds__(64 * NW) void attn_fwd_grouped_persist_kernel(MAttnF a, int nitems) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = NT * 16 * 128;                    // one [NT*16][64] bf16 tile; a buffer = K tile | V tile
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, ql = lane & 15;
  float* Ms = reinterpret_cast<float*>(smem + 4 * TILE) + wave * NT * 16;         // this wave's mask strip
  int* kvs = reinterpret_cast<int*>(smem + 4 * TILE + NW * NT * 16 * sizeof(float));
  int it = blockIdx.x;
  if (it >= nitems) return;
  for (int i = threadIdx.x; i < a.B; i += blockDim.x) kvs[i] = a.kv_index[i];
  if (!a.mask)
    for (int k = lane; k < NT * 16; k += 64) Ms[k] = ((k < a.Lk) ? 0.f : -1e30f) * LOG2E;
  const bool skip = !LSE && a.skip_dead && a.gate && !a.P;       // closed heads deliver a zero context and stage nothing
  const int qtiles = (a.Lq + 15) >> 4;
  const float sc = a.scale * 1.44269504088896341f;
  __syncthreads();
  // first task of the first item: batch, Q fragments; then the item's K / V
  bf16x8 qn[2];
  int bn;
  {
    const int bkv = it / a.H, h = it - bkv * a.H;
    bn = nth_hit(kvs, a.B, bkv, wave / qtiles, lane);
    const int q = (wave % qtiles) * 16 + ql;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (bn >= 0 && q < a.Lq) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)bn * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
      qn[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
    if (!(skip && a.gate[h] == 0.f)) {
      stage_rows<SW_K>(a.K + (size_t)bkv * a.Lk * a.ldk + h * DH, a.ldk, a.Lk, NT * 16, smem);
      stage_rows<SW_V>(a.V + (size_t)bkv * a.Lk * a.ldv + h * DH, a.ldv, a.Lk, NT * 16, smem + TILE);
    }
  }
  int mb = -1;                                           // batch whose mask row sits in this wave's strip
  for (int i = 0; it < nitems; ++i, it += gridDim.x) {
    const int bkv = it / a.H, h = it - bkv * a.H;
    stage_wait();
    // (the prefetched Q fragments are "consumed" HERE for the compiler's wait-count pass: it cannot see the wait above, and
    // would otherwise place its own vmcnt(0) at their first use - behind the next item's DMA, draining it)
    asm volatile("" : "+v"(qn[0]), "+v"(qn[1]));
    __syncthreads();                                     // item i has landed; everyone is done with the other buffer
    bf16x8 qc[2] = {qn[0], qn[1]};
    const int bc = bn;
    const int itn = it + gridDim.x;
    if (itn < nitems) {
      const int bkvn = itn / a.H, hn = itn - bkvn * a.H;
      bn = nth_hit(kvs, a.B, bkvn, wave / qtiles, lane);
      const int q = (wave % qtiles) * 16 + ql;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (bn >= 0 && q < a.Lq) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)bn * a.Lq + q) * a.ldq + hn * DH + ks * 32 + g * 8);
        qn[ks] = *reinterpret_cast<bf16x8*>(&v);
      }
      if (!(skip && a.gate[hn] == 0.f)) {
        char* nb = smem + ((i + 1) & 1) * 2 * TILE;
        stage_rows<SW_K>(a.K + (size_t)bkvn * a.Lk * a.ldk + hn * DH, a.ldk, a.Lk, NT * 16, nb);
        stage_rows<SW_V>(a.V + (size_t)bkvn * a.Lk * a.ldv + hn * DH, a.ldv, a.Lk, NT * 16, nb + TILE);
      }
    }
    const char* Ks = smem + (i & 1) * 2 * TILE;
    const char* Vs = Ks + TILE;
    const bool dead = skip && a.gate[h] == 0.f;
    for (int t = wave;; t += NW) {
      int b;
      bf16x8 qf[2];
      const int qt = t % qtiles, q = qt * 16 + ql;
      const bool qok = q < a.Lq;
      if (t == wave) {
        b = bc; qf[0] = qc[0]; qf[1] = qc[1];
      } else {
        b = nth_hit(kvs, a.B, bkv, t / qtiles, lane);
        if (b >= 0 && !dead) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qok) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
            qf[ks] = *reinterpret_cast<bf16x8*>(&v);
          }
          // (a wave's second and later tasks of an item: this load sits behind the next item's DMA and waits for it - the
          // wait belongs in THIS branch, not at the join with the prefetched first task)
          asm volatile("" : "+v"(qf[0]), "+v"(qf[1]));
        }
      }
      if (b < 0) break;                                  // (wave-uniform)
      if (dead) {
        if (qok) {
          bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH + g * 16;
          *reinterpret_cast<uint4*>(Or) = make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(Or + 8) = make_uint4(0, 0, 0, 0);
        }
        continue;
      }
      if (a.mask && mb != b) {                           // (in-order LDS: the strip is complete before this wave reads it)
        for (int k = lane; k < NT * 16; k += 64) Ms[k] = ((k < a.Lk) ? a.mask[(size_t)b * a.Lk + k] : -1e30f) * LOG2E;
        mb = b;
      }
      f32x4 acc[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Ks, tt, ks, lane), qf[ks], acc[tt], 0, 0, 0);
      }
      float m = -3.0e38f;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + tile_key0(tt, g));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[tt][r] = fmaf(acc[tt][r], sc, mk[r]);
          m = fmaxf(m, acc[tt][r]);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[tt][r] = EXP2(acc[tt][r] - m);
          sum += acc[tt][r];
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      if (LSE && qok && g == 0) a.lse[((size_t)b * a.H + h) * a.Lq + q] = m + __log2f(sum);
      bf16x4 pk[NT];
      bf16* Pr = a.P ? a.P + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr : nullptr;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[tt][r] = (bf16)(acc[tt][r] * inv);
      }
#pragma unroll
      for (int s2 = 0; s2 < NT / 2; ++s2) {
        const int kcol = s2 * 32 + g * 8;
        if (Pr && qok && kcol < a.ldpr) {
          bf16x8 pp;
#pragma unroll
          for (int r = 0; r < 4; ++r) { pp[r] = pk[2 * s2][r]; pp[4 + r] = pk[2 * s2 + 1][r]; }
          *reinterpret_cast<bf16x8*>(Pr + kcol) = pp;
        }
      }
      f32x4 o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // V fragments of tile pair s2 + 1 are requested before the MFMAs of pair s2 (asm reads: the explicit lgkmcnt)
      bf16x8 vf[2][4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[0][dt] = vcol_frag_asm(Vs, 0, 1, dt, lane);
#pragma unroll
      for (int s2 = 0; s2 < NT / 2; ++s2) {
        bf16x8 pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pb[r] = pk[2 * s2][r]; pb[4 + r] = pk[2 * s2 + 1][r]; }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (s2 + 1 < NT / 2) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) vf[(s2 + 1) & 1][dt] = vcol_frag_asm(Vs, 2 * s2 + 2, 2 * s2 + 3, dt, lane);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[s2 & 1][dt], pb, o[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (qok) {
        const float gz = a.gate ? a.gate[h] : 1.0f;
        bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16x4 ov = {(bf16)(o[dt][0] * gz), (bf16)(o[dt][1] * gz), (bf16)(o[dt][2] * gz), (bf16)(o[dt][3] * gz)};
          *reinterpret_cast<bf16x4*>(Or + dt * 16 + g * 4) = ov;
        }
      }
    }
  }
}

#endif

// ---------------------------------------------------------------------------------------------------------------------
// Long key sequences (225..928 keys: the ViT at 384 x 384 / 480 x 480, cross-attention onto those image tokens) when
// nobody takes the map: a STREAMING forward.  The whole-row kernel above keeps all 38 / 58 score tiles of a query tile in
// registers (256 VGPRs + 197 AGPRs: one wave per SIMD) and K and V of a (batch, head) whole in LDS (152 KiB: one 4-wave
// workgroup per CU, every 64 queries stage the full 148 KiB again - 1.1 GB of staging per ViT layer at 577 tokens, 488 us
// for 65 GFLOP, 1 086 us with the fused distillation term).  Here K and V pass through LDS in BLOCKS of 128 keys,
// double-buffered (64 KiB; the next block's LDS-DMA is in flight under the current block's arithmetic, one barrier per
// block), and a wave keeps only the current block's 8 score tiles plus the running row maximum m, row sum l and the
// un-normalised context (online softmax): ~110 VGPRs, 16 waves = 256 queries per workgroup, so K / V of a (batch, head)
// are streamed 3 times at 577 queries instead of 10.  The row's lse = m + log2 l is what the recomputing backward wants.
// Fused map distillation in the same pass: sum_k (p - pt)^2 with p = e / l is  (sum e^2) / l^2 - 2 (sum e pt) / l + sum pt^2
// - three running sums over the un-normalised e, rescaled with the maximum like l; the backward's row term
// sum_k p (p - pt) = (sum e^2) / l^2 - (sum e pt) / l comes out of the same sums.
// Workgroups of one (batch, head) are mapped to ONE XCD (consecutive logical ids share an L2).
// (round 5: every caller passes compile-time nrows / nw with nrows % (8 nw) == 0, so the trip count nrows / (8 nw) is the
// SAME for every wave and known to the compiler - written `r0 = wave * 8; r0 < nrows` it looked wave-dependent, the loop
// stayed a loop, and the wait-count pass, unable to count the DMA instructions in flight, answered every wait for an older
// load - the teacher map's pieces requested just before the DMA - with vmcnt(0): draining the next block's staging)
template <int SW>
__device__ __forceinline__ void stage_block(const bf16* base, int ld, int L, int key0, int nrows, int nw, char* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < nrows / (nw * 8); ++k) {
    const int r0 = wave * 8 + k * nw * 8;
    const int row = r0 + (lane >> 3), cs = lane & 7;
    const int c = swz<SW>(row, cs);
    const int key = min(key0 + row_key(row), L - 1);
    const bf16* src = base + (size_t)key * ld + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sm + r0 * 128), 16, 0, 0);
  }
}

// KDR (round 5, ABI 8): the fused map distillation against a teacher map that is NOT in memory - the frozen teacher kept
// its projected queries / keys and its row lse (113 MB + 1.8 MB per ViT layer at 577 tokens) instead of writing a 517 MB
// bf16 map that this kernel and the backward would each read once.  The teacher's K block streams through LDS beside the
// student's (a third 16 KiB tile per buffer), its scores are rebuilt per key tile - 2 MFMAs, 4 exponentials per lane - and
// consumed at once by the running sums of the term: p_t = 2^(s_t log2e - lse_t) in fp32, not a bf16-rounded stored value.
template <int NW, bool LSE, int TQ, bool KDR = false, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd_stream_kernel(MAttnF a) {
  constexpr int KBT = 8, KB = KBT * 16;                  // 128 keys per block
  constexpr int BUF = (KDR ? 3 : 2) * KB * 128;          // one staging buffer: K | V [| teacher K]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = (a.Lk + KB - 1) / KB;
  float* Ms = reinterpret_cast<float*>(smem + 2 * BUF);          // [nblk * KB] additive mask (+ -1e30 beyond Lk)
  float* kdw = Ms + nblk * KB;
  // XCD-aware map: hardware workgroup id i runs on XCD i % 8; logical ids (i % 8) * (n / 8) + i / 8 are then consecutive
  // per XCD, and the gridDim.x query blocks of one (batch, head) - consecutive logical ids - share an L2
  const int gx = gridDim.x, nwg = gx * gridDim.y * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int qblk = lid % gx, h = (lid / gx) % a.H, b = lid / (gx * a.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  // TQ query tiles per wave (consecutive): every K / V fragment read out of LDS feeds TQ MFMAs, and the TQ independent
  // softmax chains interleave in the wave's instruction stream
  const int q0 = (qblk * NW + wave) * 16 * TQ;
  int q[TQ];
  bool qok[TQ];
#pragma unroll
  for (int j = 0; j < TQ; ++j) { q[j] = q0 + 16 * j + ql; qok[j] = q[j] < a.Lq; }
  const bool active = q0 < a.Lq;                         // (waves past the last query still stage and meet the barriers)
  if (!LSE && a.skip_dead && a.gate && !a.Pt && a.gate[h] == 0.f) {        // closed head, zero context (workgroup-uniform)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
      if (qok[j]) {
        bf16* Or = a.O + ((size_t)b * a.Lq + q[j]) * a.ldo + h * DH + g * 16;
        *reinterpret_cast<uint4*>(Or) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Or + 8) = make_uint4(0, 0, 0, 0);
      }
    return;
  }
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const bf16* Kb = a.K + (size_t)bkv * a.Lk * a.ldk + h * DH;
  const bf16* Vb = a.V + (size_t)bkv * a.Lk * a.ldv + h * DH;
  for (int k = threadIdx.x; k < nblk * KB; k += blockDim.x)
    Ms[k] = ((k < a.Lk) ? (a.mask ? a.mask[(size_t)b * a.Lk + k] : 0.f) : -1e30f) * LOG2E;      // (log2 domain, as stage_mask)
  if (threadIdx.x < 2) kdw[threadIdx.x] = 0.f;
  bf16x8 qf[TQ][2];
#pragma unroll
  for (int j = 0; j < TQ; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qok[j]) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q[j]) * a.ldq + h * DH + ks * 32 + g * 8);
      qf[j][ks] = *reinterpret_cast<bf16x8*>(&v);
    }
  stage_block<SW_K>(Kb, a.ldk, a.Lk, 0, KB, NW, smem);
  stage_block<SW_V>(Vb, a.ldv, a.Lk, 0, KB, NW, smem + KB * 128);
  const bf16* Tkb = KDR ? a.Tk + (size_t)b * a.Lk * a.tld + h * DH : nullptr;
  bf16x8 qt[KDR ? TQ : 1][2];
  float tl[KDR ? TQ : 1];
  if (KDR) {
    stage_block<SW_K>(Tkb, a.tld, a.Lk, 0, KB, NW, smem + 2 * KB * 128);
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      tl[j] = qok[j] ? a.tlse[((size_t)b * a.H + h) * a.Lq + q[j]] : 3.0e38f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qok[j]) v = *reinterpret_cast<const uint4*>(a.Tq + ((size_t)b * a.Lq + q[j]) * a.tld + h * DH + ks * 32 + g * 8);
        qt[j][ks] = *reinterpret_cast<bf16x8*>(&v);
      }
    }
  }
  const float sc = a.scale * 1.44269504088896341f;
  const bool kd_any = LSE && (KDR || a.Pt != nullptr);
  // DROP: the un-normalised e of a dropped key leaves the P V product (keep is 0 / 1 here; the kept keys' 1 / (1 - p)
  // joins the row's 1 / l on the context); l, the lse and the distillation sums see every key
  DropRng rng;
  uint64_t d8[TQ];
  float keep_scale = 1.0f;
  if (DROP) {
    rng = drop_rng(a.rng, a.call, a.drop_p);
    keep_scale = rng.scale;
    rng.scale = 1.0f;
#pragma unroll
    for (int j = 0; j < TQ; ++j) d8[j] = (((uint64_t)b * a.H + h) * a.Lq + (qok[j] ? q[j] : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
  }
  float m[TQ], l[TQ], se2[TQ], sep[TQ], spt[TQ];
  f32x4 o[TQ][4];
#pragma unroll
  for (int j = 0; j < TQ; ++j) {
    m[j] = -3.0e38f; l[j] = 0.f; se2[j] = 0.f; sep[j] = 0.f; spt[j] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int blk = 0; blk < nblk; ++blk) {
    stage_wait();
    __syncthreads();                                     // block blk has landed; everyone is done with the other buffer
    // the teacher map's piece for this block is requested BEFORE the next block's DMA: vmcnt counts in order, a wait for
    // a load issued behind the DMA would drain the DMA with it
    bf16x8 t8[KDR ? 1 : TQ][KBT / 2];
    if (!KDR && kd_any) {
#pragma unroll
      for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int s2 = 0; s2 < KBT / 2; ++s2) {
          const int kcol = blk * KB + s2 * 32 + g * 8;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (qok[j] && kcol < a.ldpr) v = *reinterpret_cast<const uint4*>(a.Pt + (((size_t)b * a.H + h) * a.Lq + q[j]) * a.ldpr + kcol);
          t8[j][s2] = *reinterpret_cast<bf16x8*>(&v);
        }
    }
    if (blk + 1 < nblk) {
      char* nb = smem + ((blk + 1) & 1) * BUF;
      stage_block<SW_K>(Kb, a.ldk, a.Lk, (blk + 1) * KB, KB, NW, nb);
      stage_block<SW_V>(Vb, a.ldv, a.Lk, (blk + 1) * KB, KB, NW, nb + KB * 128);
      if (KDR) stage_block<SW_K>(Tkb, a.tld, a.Lk, (blk + 1) * KB, KB, NW, nb + 2 * KB * 128);
    }
    if (!active) continue;                               // (wave-uniform)
    const char* Ks = smem + (blk & 1) * BUF;
    const char* Vs = Ks + KB * 128;
    const char* Kts = Ks + 2 * KB * 128;
    f32x4 acc[TQ][KBT];
#pragma unroll
    for (int t = 0; t < KBT; ++t) {
#pragma unroll
      for (int j = 0; j < TQ; ++j) acc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = krow_frag(Ks, t, ks, lane);
#pragma unroll
        for (int j = 0; j < TQ; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[j][ks], acc[j][t], 0, 0, 0);
      }
    }
    float bm[TQ];
#pragma unroll
    for (int j = 0; j < TQ; ++j) bm[j] = -3.0e38f;
#pragma unroll
    for (int t = 0; t < KBT; ++t) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + blk * KB + tile_key0(t, g));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ml = mk[r];
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          acc[j][t][r] = fmaf(acc[j][t][r], sc, ml);     // (the recomputing backward forms the same number)
          bm[j] = fmaxf(bm[j], acc[j][t][r]);
        }
      }
    }
    float alpha[TQ];
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      bm[j] = fmaxf(bm[j], __shfl_xor(bm[j], 16, 64));
      bm[j] = fmaxf(bm[j], __shfl_xor(bm[j], 32, 64));
      const float mn = fmaxf(m[j], bm[j]);
      alpha[j] = EXP2(m[j] - mn);                        // (first block: 2^(-3e38) = 0 on l = 0, o = 0)
      m[j] = mn;
      l[j] *= alpha[j];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[j][dt][r] *= alpha[j];
#pragma unroll
      for (int t = 0; t < KBT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[j][t][r] = EXP2(acc[j][t][r] - mn);
          l[j] += acc[j][t][r];                          // (this lane's keys; the lane groups meet after the last block)
        }
      if (KDR) {                       // (wave-uniform: the MFMAs run for every lane; rows beyond Lq carry lse = +3e38 -> p_t = 0)
        se2[j] *= alpha[j] * alpha[j];
        sep[j] *= alpha[j];
#pragma unroll
        for (int t = 0; t < KBT; ++t) {
          f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Kts, t, ks, lane), qt[j][ks], sa, 0, 0, 0);
          const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + blk * KB + tile_key0(t, g));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = acc[j][t][r];
            const float pt = EXP2(fmaf(sa[r], sc, mk[r]) - tl[j]);
            se2[j] = fmaf(e, e, se2[j]);
            sep[j] = fmaf(e, pt, sep[j]);
            spt[j] = fmaf(pt, pt, spt[j]);
          }
          // (two query tiles per wave sit at the register limit: keep one tile's teacher scores alive at a time)
          if (TQ == 2) __builtin_amdgcn_sched_barrier(0);
        }
      } else if (kd_any && qok[j]) {
        se2[j] *= alpha[j] * alpha[j];
        sep[j] *= alpha[j];
#pragma unroll
        for (int s2 = 0; s2 < KBT / 2; ++s2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e0 = acc[j][2 * s2][r], e1 = acc[j][2 * s2 + 1][r];
            const float p0 = (float)t8[j][s2][r], p1 = (float)t8[j][s2][4 + r];
            se2[j] = fmaf(e0, e0, se2[j]); se2[j] = fmaf(e1, e1, se2[j]);
            sep[j] = fmaf(e0, p0, sep[j]); sep[j] = fmaf(e1, p1, sep[j]);
            spt[j] = fmaf(p0, p0, spt[j]); spt[j] = fmaf(p1, p1, spt[j]);
          }
      }
    }
    // O^T += V^T E^T with the un-normalised e (<= 1) as bf16: the row's 1 / l is applied once, to the context
    // (asm fragment reads, one tile pair ahead of the MFMAs that consume them: the next block's DMA stays in flight)
    // (16 waves per workgroup run at 128 VGPRs: no room for the second fragment set - read, wait, multiply there)
    constexpr bool AHEAD = NW <= 8 && !(KDR && TQ == 2);
    bf16x8 vfr[AHEAD ? 2 : 1][4];
    __builtin_amdgcn_sched_barrier(0);
    if (AHEAD) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vfr[0][dt] = vcol_frag_a<SW_V>(Vs, 0, 1, dt, lane);
    }
#pragma unroll
    for (int s2 = 0; s2 < KBT / 2; ++s2) {
      bf16x8 pb[TQ];
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        if (DROP) {
          float f[8];
          drop_factor8(rng, d8[j] + (blk * KBT / 2 + s2) * 4, f);
#pragma unroll
          for (int r = 0; r < 4; ++r) { pb[j][r] = (bf16)(acc[j][2 * s2][r] * f[r]); pb[j][4 + r] = (bf16)(acc[j][2 * s2 + 1][r] * f[4 + r]); }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { pb[j][r] = (bf16)acc[j][2 * s2][r]; pb[j][4 + r] = (bf16)acc[j][2 * s2 + 1][r]; }
        }
      }
      if (!AHEAD) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vfr[0][dt] = vcol_frag_a<SW_V>(Vs, 2 * s2, 2 * s2 + 1, dt, lane);
      }
      TR_WAIT();
      if (AHEAD && s2 + 1 < KBT / 2) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vfr[(s2 + 1) & 1][dt] = vcol_frag_a<SW_V>(Vs, 2 * s2 + 2, 2 * s2 + 3, dt, lane);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int j = 0; j < TQ; ++j)
          o[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[AHEAD ? (s2 & 1) : 0][dt], pb[j], o[j][dt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sq = 0.f;
  if (active) {
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      float lj = l[j];
      lj += __shfl_xor(lj, 16, 64);
      lj += __shfl_xor(lj, 32, 64);
      const float inv = 1.0f / lj;
      if (LSE && qok[j] && g == 0) a.lse[((size_t)b * a.H + h) * a.Lq + q[j]] = m[j] + __log2f(lj);
      if (qok[j]) {
        const float gz = (a.gate ? a.gate[h] : 1.0f) * (DROP ? inv * keep_scale : inv);
        bf16* Or = a.O + ((size_t)b * a.Lq + q[j]) * a.ldo + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16x4 ov = {(bf16)(o[j][dt][0] * gz), (bf16)(o[j][dt][1] * gz), (bf16)(o[j][dt][2] * gz), (bf16)(o[j][dt][3] * gz)};
          *reinterpret_cast<bf16x4*>(Or + dt * 16 + g * 4) = ov;
        }
      }
      if (kd_any) {
        const bool on = qok[j];
        if (on) sq += fmaf(se2[j] * inv, inv, fmaf(-2.f * sep[j], inv, spt[j]));
        if (a.rkd) {
          float rk = on ? fmaf(se2[j] * inv, inv, -sep[j] * inv) : 0.f;
          rk += __shfl_xor(rk, 16, 64);
          rk += __shfl_xor(rk, 32, 64);
          if (on && g == 0) a.rkd[((size_t)b * a.H + h) * a.Lq + q[j]] = rk;
        }
      }
    }
  }
  if (kd_any) {                                          // one atomic per workgroup (the waves meet in LDS, last one publishes)
    sq = wave_sum(sq);
    if (lane == 0) {
      atomicAdd(&kdw[0], sq);
      const float before = atomicAdd(&kdw[1], 1.0f);
      if ((int)before == NW - 1) atomicAdd(a.kd, atomicAdd(&kdw[0], 0.f) * a.kd_coef);
    }
  }
}

// The same streaming scheme when the probability map IS an output (the teacher's kept layers, cross-attention maps of the
// pruning steps, the stored-map training form): TWO passes over the keys.  Pass 1 streams K alone and forms the row
// maximum m and sum l online; pass 2 streams K and V, recomputes the scores, writes p = 2^(s - m) / l as bf16 - 16-byte
// pieces, rows of `ldpr` - and multiplies the SAME rounded probabilities into V (what a reader of the map would
// recompute), with the distillation term formed from them (stored-map form) or from the fp32 values (lse form).
template <int NW, bool LSE, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd_stream_map_kernel(MAttnF a) {
  constexpr int KBT = 8, KB = KBT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = (a.Lk + KB - 1) / KB;
  float* Ms = reinterpret_cast<float*>(smem + 4 * KB * 128);
  float* kdw = Ms + nblk * KB;
  const int gx = gridDim.x, nwg = gx * gridDim.y * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int qblk = lid % gx, h = (lid / gx) % a.H, b = lid / (gx * a.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int q0 = (qblk * NW + wave) * 16;
  const int q = q0 + ql;
  const bool active = q0 < a.Lq, qok = q < a.Lq;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const bf16* Kb = a.K + (size_t)bkv * a.Lk * a.ldk + h * DH;
  const bf16* Vb = a.V + (size_t)bkv * a.Lk * a.ldv + h * DH;
  for (int k = threadIdx.x; k < nblk * KB; k += blockDim.x)
    Ms[k] = ((k < a.Lk) ? (a.mask ? a.mask[(size_t)b * a.Lk + k] : 0.f) : -1e30f) * LOG2E;      // (log2 domain, as stage_mask)
  if (threadIdx.x < 2) kdw[threadIdx.x] = 0.f;
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qok) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
    qf[ks] = *reinterpret_cast<bf16x8*>(&v);
  }
  const float sc = a.scale * 1.44269504088896341f;
  // ---- pass 1: m, l ----
  stage_block<SW_K>(Kb, a.ldk, a.Lk, 0, KB, NW, smem);
  float m = -3.0e38f, l = 0.f;
  for (int blk = 0; blk < nblk; ++blk) {
    stage_wait();
    __syncthreads();
    if (blk + 1 < nblk) stage_block<SW_K>(Kb, a.ldk, a.Lk, (blk + 1) * KB, KB, NW, smem + ((blk + 1) & 1) * 2 * KB * 128);
    if (!active) continue;
    const char* Ks = smem + (blk & 1) * 2 * KB * 128;
    f32x4 acc[KBT];
#pragma unroll
    for (int t = 0; t < KBT; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Ks, t, ks, lane), qf[ks], acc[t], 0, 0, 0);
    }
    float bm = -3.0e38f;
#pragma unroll
    for (int t = 0; t < KBT; ++t) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + blk * KB + tile_key0(t, g));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[t][r] = fmaf(acc[t][r], sc, mk[r]);
        bm = fmaxf(bm, acc[t][r]);
      }
    }
    bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m, bm);
    l *= EXP2(m - mn);
    m = mn;
#pragma unroll
    for (int t = 0; t < KBT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) l += EXP2(acc[t][r] - mn);
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = active ? 1.0f / l : 0.f;
  if (LSE && active && qok && g == 0) a.lse[((size_t)b * a.H + h) * a.Lq + q] = m + __log2f(l);
  __syncthreads();                                       // every wave is done with both buffers
  // ---- pass 2: p, P V ----
  stage_block<SW_K>(Kb, a.ldk, a.Lk, 0, KB, NW, smem);
  stage_block<SW_V>(Vb, a.ldv, a.Lk, 0, KB, NW, smem + KB * 128);
  const bool kd_on = a.Pt != nullptr && qok;
  const bf16* Tr = kd_on ? a.Pt + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr : nullptr;
  bf16* Pr = (a.P && qok) ? a.P + (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr : nullptr;
  float sq = 0.f, rk = 0.f;
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
  const uint64_t d8 = (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int blk = 0; blk < nblk; ++blk) {
    stage_wait();
    __syncthreads();
    bf16x8 t8[KBT / 2];
    if (kd_on) {                                         // (before the next block's DMA: see attn_fwd_stream_kernel)
#pragma unroll
      for (int s2 = 0; s2 < KBT / 2; ++s2) {
        const int kcol = blk * KB + s2 * 32 + g * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (kcol < a.ldpr) v = *reinterpret_cast<const uint4*>(Tr + kcol);
        t8[s2] = *reinterpret_cast<bf16x8*>(&v);
      }
    }
    if (blk + 1 < nblk) {
      char* nb = smem + ((blk + 1) & 1) * 2 * KB * 128;
      stage_block<SW_K>(Kb, a.ldk, a.Lk, (blk + 1) * KB, KB, NW, nb);
      stage_block<SW_V>(Vb, a.ldv, a.Lk, (blk + 1) * KB, KB, NW, nb + KB * 128);
    }
    if (!active) continue;
    const char* Ks = smem + (blk & 1) * 2 * KB * 128;
    const char* Vs = Ks + KB * 128;
    f32x4 acc[KBT];
#pragma unroll
    for (int t = 0; t < KBT; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Ks, t, ks, lane), qf[ks], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < KBT; ++t) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + blk * KB + tile_key0(t, g));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = EXP2(fmaf(acc[t][r], sc, mk[r]) - m) * inv;
    }
#pragma unroll
    for (int s2 = 0; s2 < KBT / 2; ++s2) {
      bf16x8 pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pb[r] = (bf16)acc[2 * s2][r]; pb[4 + r] = (bf16)acc[2 * s2 + 1][r]; }
      const int kcol = blk * KB + s2 * 32 + g * 8;
      if (Pr && kcol < a.ldpr) *reinterpret_cast<bf16x8*>(Pr + kcol) = pb;
      if (kd_on) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = LSE ? acc[2 * s2][r] : (float)pb[r], p1 = LSE ? acc[2 * s2 + 1][r] : (float)pb[4 + r];
          const float d0 = p0 - (float)t8[s2][r], d1 = p1 - (float)t8[s2][4 + r];
          sq = fmaf(d0, d0, sq); sq = fmaf(d1, d1, sq);
          rk = fmaf(p0, d0, rk); rk = fmaf(p1, d1, rk);
        }
      }
      if (DROP) {                                        // the context sees P .* keep / (1 - p); the map above stays un-dropped
        float f[8];
        drop_factor8(rng, d8 + (blk * KBT / 2 + s2) * 4, f);
#pragma unroll
        for (int r = 0; r < 4; ++r) { pb[r] = (bf16)(acc[2 * s2][r] * f[r]); pb[4 + r] = (bf16)(acc[2 * s2 + 1][r] * f[4 + r]); }
      }
      bf16x8 vfr[4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vfr[dt] = vcol_frag_a<SW_V>(Vs, 2 * s2, 2 * s2 + 1, dt, lane);
      TR_WAIT();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pb, o[dt], 0, 0, 0);
    }
  }
  if (active && qok) {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    bf16* Or = a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ov = {(bf16)(o[dt][0] * gz), (bf16)(o[dt][1] * gz), (bf16)(o[dt][2] * gz), (bf16)(o[dt][3] * gz)};
      *reinterpret_cast<bf16x4*>(Or + dt * 16 + g * 4) = ov;
    }
  }
  if (a.Pt) {
    if (LSE && a.rkd && active) {
      rk += __shfl_xor(rk, 16, 64);
      rk += __shfl_xor(rk, 32, 64);
      if (qok && g == 0) a.rkd[((size_t)b * a.H + h) * a.Lq + q] = rk;
    }
    sq = wave_sum(sq);
    if (lane == 0) {
      atomicAdd(&kdw[0], sq);
      const float before = atomicAdd(&kdw[1], 1.0f);
      if ((int)before == NW - 1) atomicAdd(a.kd, atomicAdd(&kdw[0], 0.f) * a.kd_coef);
    }
  }
}

static bool launch_fwd_stream(const MAttnF& f, hipStream_t stream) {
  // (A/B switch, read per call: the tests toggle it)  EVLM_ATTN_NO_STREAM=1: the whole-row kernels for every length
  const char* env = getenv("EVLM_ATTN_NO_STREAM");
  static const int min_keys = getenv("EVLM_ATTN_STREAM_MIN") ? atoi(getenv("EVLM_ATTN_STREAM_MIN")) : 225;   // (tuning aid)
  if ((env && atoi(env)) || f.Lk < min_keys || f.Lk > 1024 || f.causal) return false;
  constexpr int KB = 128;
  const int nblk = (f.Lk + KB - 1) / KB, qtiles = (f.Lq + 15) / 16;
  const size_t lds = (size_t)4 * KB * 128 + (size_t)nblk * KB * sizeof(float) + 16;
#define STREAM_LAUNCH_D(NW_, LSE_, TQ_, DROP_)                                                                           \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)attn_fwd_stream_kernel<NW_, LSE_, TQ_, false, DROP_>,                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
    dim3 grid((qtiles + NW_ * TQ_ - 1) / (NW_ * TQ_), f.H, f.B), block(64 * NW_);                                       \
    hipLaunchKernelGGL((attn_fwd_stream_kernel<NW_, LSE_, TQ_, false, DROP_>), grid, block, lds, stream, f);             \
  } while (0)
  // (probability dropout: text queries on image tokens - a few query tiles; one tile per wave, 8 waves)
#define STREAM_LAUNCH(NW_, LSE_, TQ_) do { if (drop) STREAM_LAUNCH_D(8, LSE_, 1, true); else STREAM_LAUNCH_D(NW_, LSE_, TQ_, false); } while (0)
  const bool drop = f.drop_p > 0.f;
  // 8 waves = 128 queries per workgroup: measured (tools/attn_long_bench.py) level with or ahead of 16 waves on the ViT's
  // 577 / 901 tokens (the last query block is fuller, two workgroups share a CU), and a few text queries on those image
  // tokens want the 8 waves for the K / V streaming itself (61 us against 82 us with one wave per query tile)
  static const int nw_env = getenv("EVLM_ATTN_STREAM_NW") ? atoi(getenv("EVLM_ATTN_STREAM_NW")) : 0;     // (tuning aid)
  const int nw = nw_env == 16 ? 16 : 8;
  if (f.P || (f.Pt && !f.lse)) {                         // the map is an output (or the stored-map distillation form): two passes
#define STREAM_MAP_LAUNCH(LSE_, DROP_)                                                                                   \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)attn_fwd_stream_map_kernel<8, LSE_, DROP_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                                 \
    dim3 grid((qtiles + 7) / 8, f.H, f.B), block(512);                                                                  \
    hipLaunchKernelGGL((attn_fwd_stream_map_kernel<8, LSE_, DROP_>), grid, block, lds, stream, f);                       \
  } while (0)
    if (drop) { if (f.lse) STREAM_MAP_LAUNCH(true, true); else STREAM_MAP_LAUNCH(false, true); }
    else if (f.lse) STREAM_MAP_LAUNCH(true, false); else STREAM_MAP_LAUNCH(false, false);
#undef STREAM_MAP_LAUNCH
    return true;
  }
  // two query tiles per wave once a (batch, head) has more than one workgroup's worth of single tiles (EVLM_ATTN_STREAM_TQ)
  static const int tq_env = getenv("EVLM_ATTN_STREAM_TQ") ? atoi(getenv("EVLM_ATTN_STREAM_TQ")) : 0;     // (tuning aid)
  // measured (tools/attn_long_bench.py, TQ = 1 -> 2): with lse 190 -> 158 us (577 keys) / 227 -> 162 (901), with lse + fused
  // distillation 289 -> 264 / 359 -> 289, no-grad 155 -> 166 / 177 -> 158; two text query tiles on image tokens 62 -> 111
  const int tq = tq_env == 1 || tq_env == 2 ? tq_env : ((qtiles > 8 && (f.lse || qtiles > 40)) ? 2 : 1);
  if (f.Tq) {                                            // the teacher's map rebuilt in the kernel (KDR): a third tile per buffer
    const size_t ldsr = (size_t)6 * KB * 128 + (size_t)nblk * KB * sizeof(float) + 16;
#define STREAM_KDR_LAUNCH(TQ_)                                                                                           \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)attn_fwd_stream_kernel<8, true, TQ_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)ldsr);                                                                                \
    dim3 grid((qtiles + 8 * TQ_ - 1) / (8 * TQ_), f.H, f.B), block(512);                                                \
    hipLaunchKernelGGL((attn_fwd_stream_kernel<8, true, TQ_, true>), grid, block, ldsr, stream, f);                      \
  } while (0)
    if (tq == 2) STREAM_KDR_LAUNCH(2); else STREAM_KDR_LAUNCH(1);
#undef STREAM_KDR_LAUNCH
    return true;
  }
  if (f.lse) {
    if (nw == 16) STREAM_LAUNCH(16, true, 1); else if (tq == 2) STREAM_LAUNCH(8, true, 2); else STREAM_LAUNCH(8, true, 1);
  } else {
    if (nw == 16) STREAM_LAUNCH(16, false, 1); else if (tq == 2) STREAM_LAUNCH(8, false, 2); else STREAM_LAUNCH(8, false, 1);
  }
#undef STREAM_LAUNCH
#undef STREAM_LAUNCH_D
  return true;
}

template <int NT>
static bool launch_fwd_grouped(const MAttnF& f, int Bkv, hipStream_t stream) {
  // the grouped form pays when several query batches share a K/V row and a batch is a few query tiles (text rows on image
  // tokens); causal masks and the fused map distillation stay on the per-batch kernel
  const char* env = getenv("EVLM_ATTN_NO_GROUP");         // (A/B switch, read per call: the tests toggle it)
  if ((env && atoi(env)) || !f.kv_index || Bkv <= 0 || Bkv >= f.B || f.Lq > 64 || f.causal || f.Pt) return false;
  constexpr int NW = 8;
  // persistent, double-buffered form (round 5): one workgroup per CU walks the (K/V row, head) items - EVLM_ATTN_GROUP_PERSIST=1.
  // Measured 23.7 us against 21.2 us for one workgroup per item on the GD shape (profiles/r05_xattn_persist.md): opt-in.
  // The index copy bounds B (a query batch count of 4 096 is 16 KiB of LDS)
#ifdef EVLM_EXPERIMENTAL_ATTN_PERSIST
  const char* pe = getenv("EVLM_ATTN_GROUP_PERSIST");     // (A/B switch, read per call; measured SLOWER: opt-in)
  if (pe && atoi(pe) == 1 && f.B <= 4096 && !(f.drop_p > 0.f)) {
    const int nitems = Bkv * f.H;
    const size_t ldsp = (size_t)4 * NT * 16 * 128 + (size_t)NW * NT * 16 * sizeof(float) + (size_t)f.B * sizeof(int);
    static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); (void)hipGetDeviceProperties(&p, d); return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }();
    dim3 gridp(imin(nitems, ncu)), blockp(64 * NW);
    if (f.lse) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_grouped_persist_kernel<NT, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      hipLaunchKernelGGL((attn_fwd_grouped_persist_kernel<NT, NW, true>), gridp, blockp, ldsp, stream, f, nitems);
    } else {
      (void)hipFuncSetAttribute((const void*)attn_fwd_grouped_persist_kernel<NT, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      hipLaunchKernelGGL((attn_fwd_grouped_persist_kernel<NT, NW, false>), gridp, blockp, ldsp, stream, f, nitems);
    }
    return true;
  }
#endif
  const size_t lds = (size_t)2 * NT * 16 * 128 + (size_t)NW * NT * 16 * sizeof(float);
  dim3 grid(1, f.H, Bkv), block(64 * NW);
#define GROUPED_LAUNCH(LSE_, DROP_)                                                                                       \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      (void)hipFuncSetAttribute((const void*)attn_fwd_grouped_kernel<NT, NW, LSE_, DROP_>,                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((attn_fwd_grouped_kernel<NT, NW, LSE_, DROP_>), grid, block, lds, stream, f);                     \
  } while (0)
  if (f.drop_p > 0.f) { if (f.lse) GROUPED_LAUNCH(true, true); else GROUPED_LAUNCH(false, true); }
  else if (f.lse) GROUPED_LAUNCH(true, false); else GROUPED_LAUNCH(false, false);
#undef GROUPED_LAUNCH
  return true;
}

template <int NT>
static int launch_fwd(const MAttnF& f, hipStream_t stream) {
  constexpr int MAXW = NT <= 14 ? 16 : (NT <= 26 ? 8 : 4);   // register budget: 16 (8) waves/workgroup need <= 128 (256) VGPRs
  constexpr bool SEQ = NT > 38;                        // 2 x NT x 2 KiB of K and V no longer fit in 160 KiB of LDS
  const size_t lds = (size_t)(SEQ ? 1 : 2) * NT * 16 * 128 + (size_t)NT * 16 * sizeof(float) + 16;
  static const int nw_cap = getenv("EVLM_ATTN_FWD_NW") ? atoi(getenv("EVLM_ATTN_FWD_NW")) : 0;     // (tuning aid)
  int nw = imin(MAXW, (f.Lq + 15) / 16);               // waves per workgroup (16 queries each)
  if (nw_cap > 0 && !SEQ) nw = imin(nw, nw_cap);
  dim3 grid((f.Lq + 16 * nw - 1) / (16 * nw), f.H, f.B), block(64 * nw);
#define FWD_LAUNCH(LSE_, DROP_)                                                                                           \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<NT, MAXW, SEQ, LSE_, DROP_>,                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<NT, MAXW, SEQ, LSE_, DROP_>), grid, block, lds, stream, f);                 \
  } while (0)
  if constexpr (NT <= 14) {                            // probability dropout: text-side problems (<= 224 keys) take this kernel,
    if (f.drop_p > 0.f) {                              // longer ones the streaming pair (evlm_attention_fwd_mfma)
      if (f.lse) FWD_LAUNCH(true, true); else FWD_LAUNCH(false, true);
      return 0;
    }
  }
  if constexpr (NT != 26) {                            // (225..416 keys keep the stored-map backward: no lse form)
    if (f.lse) {
      FWD_LAUNCH(true, false);
      return 0;
    }
  }
  FWD_LAUNCH(false, false);
#undef FWD_LAUNCH
  return 0;
}

// =============================================================================================
// backward
// =============================================================================================
struct MAttnB {
  const bf16* Q; const bf16* K; const bf16* V; const bf16* P; const bf16* dO; const bf16* E; const float* gate;
  const int32_t* kv_index;
  bf16* dS; bf16* dQ; bf16* dK; bf16* dV; float* dgate;
  int B, Bkv, H, Lq, Lk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, ldpr;
  float scale;
  const bf16* Pt; const float* kd_gout; float kd_coef;   // fused map distillation: dP += kd_coef * (*kd_gout) * (P - Pt)
  // recomputing form (RC kernels): P = 2^(s - lse) rebuilt in fp32 from Q, K, the forward's mask / causal flag; Pw = bf16
  // copy of it for kernel B of the two-kernel path (NULL in the single-pass kernel)
  const float* lse; const float* mask; int causal; bf16* Pw;
  // one-pass form of the long-sequence kernel: delta = rowsum(P .* dP) = dO . O + kd_coef * g * rkd (O: the forward's
  // output, gate included; rkd: the forward's sum_k P (P - Pt)), so no first pass over the keys is needed for it
  const bf16* O; const float* rkd;
  // ABI 8: the teacher's map rebuilt in the kernel (see MAttnF)
  const bf16* Tq; const bf16* Tk; int tld; const float* tlse;
  // round 6: dropout of the probabilities (see MAttnF): with M = keep / (1 - p) regenerated per tile pair,
  //   dP = gate (M .* dO V^T) + E ;  dgate += sum P M dPo ;  dV = gate (P .* M)^T dO ;  delta = dO . O as before.
  // p_dropped: the map kernel B reads (P = the workspace kernel A wrote) already carries M
  float drop_p; const int64_t* rng; uint32_t call; int p_dropped;
};

// mask row of one (batch): Ms[k] = additive mask of key k (0 without one), -1e30 beyond Lk - as the forward kernel builds it
__device__ __forceinline__ void stage_mask(const float* mask, int b, int Lk, int n, float* Ms) {
  for (int k = threadIdx.x; k < n; k += blockDim.x)
    Ms[k] = ((k < Lk) ? (mask ? mask[(size_t)b * Lk + k] : 0.f) : -1e30f) * LOG2E;      // (log2 domain: see recompute_p)
}
// the probabilities of this lane's query for the 8 keys of tile pair s (tiles 2s, 2s+1), recomputed exactly as the forward
// kernel formed them: S^T = K Q^T (K tile read by rows), scaled + masked in the log2 domain, minus the saved row lse
// (round 5: the backward kernels are VALU-bound on exactly this chain, so it is kept to mul-free form - the mask strip is
// staged already multiplied by log2 e: the forward's `mk * LOG2E`, the same fp32 product, formed once per key instead of
// once per (query, key); the causal clamp moves with it (min commutes with a multiplication by a positive constant, bit
// for bit); rows beyond Lq get lse = +3e38, i.e. p = 2^(-inf) = 0, instead of a select per element; NOCAUSAL drops the
// clamp's compare / min / select at compile time for the encoders, which never set the flag)
template <int SW, bool NOCAUSAL = false>
__device__ __forceinline__ void recompute_p(const char* Ks, const float* Ms, const bf16x8 (&qf)[2], int s, int g, int lane,
                                            float sc, float lse_q, bool qok, int causal, int q, float (&p)[8]) {
  const float lq = qok ? lse_q : 3.0e38f;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int t = 2 * s + hh;
    f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag<SW>(Ks, t, ks, lane), qf[ks], sa, 0, 0, 0);
    const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + tile_key0(t, g));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float add = mk[r];
      if (!NOCAUSAL && causal && tile_key0(t, g) + r > q) add = fminf(add, -10000.0f * LOG2E);
      p[hh * 4 + r] = EXP2(fmaf(sa[r], sc, add) - lq);
    }
  }
}

// kernel A: same shape as the forward (a wave owns 16 queries and ALL keys):
//   dPo^T = V dO^T ;  dP = gate*dPo + E ;  delta = rowsum(P .* dP) ;  dS = P .* (dP - delta) -> HBM (for kernel B)
//   dQ^T  = scale * K^T dS^T           (dS^T accumulators reused as the MFMA B operand, K^T through tr16 reads)
template <int NT, int MAXW, bool RC, bool DROP = false>
__global__ __launch_bounds__(64 * MAXW) void attn_bwd_dq_mfma_kernel(MAttnB a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KSW = RC ? SW_KV : SW_V;
  char* Ks = smem;                          // column reads (dQ); RC: also row reads (scores)
  char* Vs = smem + NT * 16 * 128;          // k_swz (row reads)
  float* Ms = reinterpret_cast<float*>(smem + 2 * NT * 16 * 128);   // RC: mask row
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  stage_rows<KSW>(a.K + (size_t)bkv * a.Lk * a.ldk + h * DH, a.ldk, a.Lk, NT * 16, Ks);
  stage_rows<SW_K>(a.V + (size_t)bkv * a.Lk * a.ldv + h * DH, a.ldv, a.Lk, NT * 16, Vs);
  if (RC) stage_mask(a.mask, b, a.Lk, NT * 16, Ms);
  stage_wait();
  __syncthreads();
  const int q0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 16;
  if (q0 >= a.Lq) return;
  const int q = q0 + ql;
  const bool qok = q < a.Lq;
  bf16x8 dof[2], qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
    if (qok) v = *reinterpret_cast<const uint4*>(a.dO + ((size_t)b * a.Lq + q) * a.ldo + h * DH + ks * 32 + g * 8);
    if (RC && qok) vq = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
    dof[ks] = *reinterpret_cast<bf16x8*>(&v);
    qf[ks] = *reinterpret_cast<bf16x8*>(&vq);
  }
  const size_t prow = (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr;
  const float gz = a.gate ? a.gate[h] : 1.0f;
  const float kdc = a.Pt ? a.kd_coef * a.kd_gout[0] : 0.f;
  const float sc = a.scale * LOG2E;
  const float lse_q = (RC && qok) ? a.lse[((size_t)b * a.H + h) * a.Lq + q] : 0.f;
  f32x4 acc[NT];
  f32x4 pf[RC ? NT : 1];
  bf16x4 pv[NT];
  float dsum = 0.f, gsum = 0.f;
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
  const uint64_t d8 = (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) {            // tile pair: this lane's 8 consecutive keys 32s + 8g .. + 7
    const int kcol = s * 32 + g * 8;
    const bool ok = qok && kcol < a.ldpr;
    float pr[8], ex[8];                          // ex: external gradient on the map (dP_ext and / or the fused distillation term)
    float fm[DROP ? 8 : 1];
    if constexpr (DROP) drop_factor8(rng, d8 + s * 4, fm);
#pragma unroll
    for (int r = 0; r < 8; ++r) { pr[r] = 0.f; ex[r] = 0.f; }
    if (RC) recompute_p<KSW>(Ks, Ms, qf, s, g, lane, sc, lse_q, qok, a.causal, q, pr);
    if (ok) {
      if (!RC) {
        const bf16x8 p8 = *reinterpret_cast<const bf16x8*>(a.P + prow + kcol);
#pragma unroll
        for (int r = 0; r < 8; ++r) pr[r] = (float)p8[r];
      }
      if (a.E) {
        const bf16x8 e8 = *reinterpret_cast<const bf16x8*>(a.E + prow + kcol);
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[r] = (float)e8[r];
      }
      if (a.Pt) {
        const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(a.Pt + prow + kcol);
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[r] = fmaf(kdc, pr[r] - (float)t8[r], ex[r]);
      }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int t = 2 * s + hh;
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Vs, t, ks, lane), dof[ks], acc[t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = pr[hh * 4 + r];
        const float dpo = DROP ? acc[t][r] * fm[DROP ? hh * 4 + r : 0] : acc[t][r];      // (the gradient reaches P through M)
        // (RC: pv feeds kernel B's dV through the workspace - with DROP it carries M; the stored-map form keeps the
        // un-dropped value for dS and kernel B applies M to the forward's map itself)
        pv[t][r] = (DROP && RC) ? (bf16)(p * fm[DROP ? hh * 4 + r : 0]) : (bf16)p;
        if (RC) pf[RC ? t : 0][r] = p;
        gsum = fmaf(p, dpo, gsum);                 // (explicit fma forms: the single-pass kernel and kernels A + B must
        const float dp = fmaf(gz, dpo, ex[hh * 4 + r]);   //  round identically whatever the compiler would contract)
        acc[t][r] = dp;
        dsum = fmaf(p, dp, dsum);
      }
    }
  }
  dsum += __shfl_xor(dsum, 16, 64); dsum += __shfl_xor(dsum, 32, 64);
  if (a.dgate) {
    const float gs = wave_sum(gsum);             // every lane holds a partial over its own keys: the wave sum is the total
    if (lane == 0) atomicAdd(a.dgate + h, gs);
  }
  bf16x4 dsk[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dsk[t][r] = (bf16)((RC ? pf[RC ? t : 0][r] : (float)pv[t][r]) * (acc[t][r] - dsum));
  }
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) {
    const int kcol = s * 32 + g * 8;
    if (qok && kcol < a.ldpr) {
      bf16x8 d8;
#pragma unroll
      for (int r = 0; r < 4; ++r) { d8[r] = dsk[2 * s][r]; d8[4 + r] = dsk[2 * s + 1][r]; }
      *reinterpret_cast<bf16x8*>(a.dS + prow + kcol) = d8;
      if (RC && a.Pw) {                          // kernel B reads the map from here (the forward did not store one)
        bf16x8 p8;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p8[r] = pv[2 * s][r]; p8[4 + r] = pv[2 * s + 1][r]; }
        *reinterpret_cast<bf16x8*>(a.Pw + prow + kcol) = p8;
      }
    }
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) {
    bf16x8 pb;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pb[r] = dsk[2 * s][r]; pb[4 + r] = dsk[2 * s + 1][r]; }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcol_frag<KSW>(Ks, 2 * s, 2 * s + 1, dt, lane), pb, o[dt], 0, 0, 0);
  }
  if (qok) {
    bf16* dQr = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ov = {(bf16)(o[dt][0] * a.scale), (bf16)(o[dt][1] * a.scale), (bf16)(o[dt][2] * a.scale), (bf16)(o[dt][3] * a.scale)};
      *reinterpret_cast<bf16x4*>(dQr + dt * 16 + g * 4) = ov;
    }
  }
}

// kernel A for LONG key sequences (608 < Lk <= 960: 480x480 images are 901 tokens).  Holding a whole row of dP / P / dS
// in registers (kernel A above) needs ~460 VGPRs at this length and spilled ~1 KiB per lane, so this variant makes two
// passes over the keys and keeps nothing but the running sums:
//   pass 1: V (all keys, one 120 KiB LDS tile) -> delta = rowsum(P .* dP), gate gradient
//   pass 2: per key HALF, V_h and K_h side by side in the same LDS space: dP recomputed (2 MFMAs per tile), dS = P .* (dP -
//           delta) stored for kernel B and consumed at once by dQ^T += K_h^T dS^T.
// P and E are read twice (the second time mostly from the last-level cache); ~70 VGPRs, 8 waves = 128 queries per workgroup.
template <int NT, int MAXW, bool RC, bool DROP = false>
__global__ __launch_bounds__(64 * MAXW) void attn_bwd_dq_long_kernel(MAttnB a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HT = NT / 2;                               // key tiles per half (even: tile pairs stay together)
  constexpr int KSW = RC ? SW_KV : SW_V;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const bf16* Kb = a.K + (size_t)bkv * a.Lk * a.ldk + h * DH;
  const bf16* Vb = a.V + (size_t)bkv * a.Lk * a.ldv + h * DH;
  char* Vs = smem;                                         // k_swz rows (dP = V dO^T)
  char* Ks = smem + HT * 16 * 128;                         // columns (dQ^T = K^T dS^T); RC: rows too (scores)
  float* Ms = reinterpret_cast<float*>(smem + NT * 16 * 128);       // RC: mask row of this batch (all keys)
  if (!RC) {
    stage_rows<SW_K>(Vb, a.ldv, a.Lk, NT * 16, smem);      // pass 1 of the stored-map form: V of ALL keys
    stage_wait();
  } else {
    stage_mask(a.mask, b, a.Lk, NT * 16, Ms);
  }
  __syncthreads();
  const int q = (blockIdx.x * (blockDim.x >> 6) + wave) * 16 + ql;
  const bool qok = q < a.Lq;                               // (waves past the last query stay for the barriers)
  bf16x8 dof[2], qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
    if (qok) v = *reinterpret_cast<const uint4*>(a.dO + ((size_t)b * a.Lq + q) * a.ldo + h * DH + ks * 32 + g * 8);
    if (RC && qok) vq = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
    dof[ks] = *reinterpret_cast<bf16x8*>(&v);
    qf[ks] = *reinterpret_cast<bf16x8*>(&vq);
  }
  const size_t prow = (((size_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * a.ldpr;
  const float gz = a.gate ? a.gate[h] : 1.0f;
  const float kdc = a.Pt ? a.kd_coef * a.kd_gout[0] : 0.f;
  const float sc = a.scale * LOG2E;
  const float lse_q = (RC && qok) ? a.lse[((size_t)b * a.H + h) * a.Lq + q] : 0.f;
  float dsum = 0.f, gsum = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // ONE-PASS form (recomputing, no external dP, the forward's O - and its rkd with a fused distillation term - at hand):
  //   delta = sum_k p (gz dpo + kdc (p - pt)) = dO . O + kdc * rkd      (O = gz * sum_k p V_k, as stored by the forward)
  // so pass 0 - a second score product, exponentials and a dP product per key just for this row sum - is not run; the gate
  // gradient's sum_k p dpo is collected in the one pass that remains.  (Wave-uniform: kernel arguments only.)
  const bool one_pass = RC && a.O != nullptr && a.E == nullptr && (a.Pt == nullptr || a.rkd != nullptr);
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
  const uint64_t d8 = (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
  if (one_pass) {
    float d = 0.f;
    if (qok) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH + ks * 32 + g * 8);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(&v);
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf((float)of[e], (float)dof[ks][e], d);
      }
    }
    d += __shfl_xor(d, 16, 64); d += __shfl_xor(d, 32, 64);
    dsum = d;
    if (a.Pt && qok) dsum = fmaf(kdc, a.rkd[((size_t)b * a.H + h) * a.Lq + q], dsum);
  }
  // stored-map form: pass 0 over all keys out of the one V tile staged above, then pass 1 per key half (V_h | K_h);
  // recomputing form: BOTH passes per key half (the scores need K_h beside V_h): pass 0 the row sums, pass 1 dS / dQ
#pragma unroll 1
  for (int pass = one_pass ? 1 : 0; pass < 2; ++pass) {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int key0 = half * HT * 16;
      const bool staged_all = !RC && pass == 0;            // (every key's V is in LDS: no restaging, one sweep)
      if (staged_all && half == 1) break;
      if (!staged_all) {
        __syncthreads();                                   // every wave is done with the previous contents
        stage_rows<SW_K>(Vb + (size_t)key0 * a.ldv, a.ldv, a.Lk - key0, HT * 16, Vs);
        stage_rows<KSW>(Kb + (size_t)key0 * a.ldk, a.ldk, a.Lk - key0, HT * 16, Ks);
        stage_wait();
        __syncthreads();
      }
      const int npairs = staged_all ? NT / 2 : HT / 2;
#pragma unroll 2
      for (int s = 0; s < npairs; ++s) {
        const int kcol = (staged_all ? 0 : key0) + s * 32 + g * 8;
        const bool ok = qok && kcol < a.ldpr;
        float pr[8], ex[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { pr[r] = 0.f; ex[r] = 0.f; }
        if (RC) {
          // (the mask strip covers all keys: tile s of this half is tile HT/2*half + s of the row; recompute_p indexes the
          // strip by tile through `Ms + key0`)
          recompute_p<KSW>(Ks, Ms + key0, qf, s, g, lane, sc, lse_q, qok, a.causal, q - key0, pr);
        }
        if (ok) {
          if (!RC) {
            const bf16x8 p8 = *reinterpret_cast<const bf16x8*>(a.P + prow + kcol);
#pragma unroll
            for (int r = 0; r < 8; ++r) pr[r] = (float)p8[r];
          }
          if (a.E) {
            const bf16x8 e8 = *reinterpret_cast<const bf16x8*>(a.E + prow + kcol);
#pragma unroll
            for (int r = 0; r < 8; ++r) ex[r] = (float)e8[r];
          }
          if (a.Pt) {
            const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(a.Pt + prow + kcol);
#pragma unroll
            for (int r = 0; r < 8; ++r) ex[r] = fmaf(kdc, pr[r] - (float)t8[r], ex[r]);
          }
        }
        float fm[DROP ? 8 : 1];
        if constexpr (DROP) drop_factor8(rng, d8 + (kcol >> 5) * 4, fm);
        bf16x8 ds8, p8o;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Vs, 2 * s + hh, ks, lane), dof[ks], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = pr[hh * 4 + r];
            const float dpo = DROP ? acc[r] * fm[DROP ? hh * 4 + r : 0] : acc[r];
            const float dp = fmaf(gz, dpo, ex[hh * 4 + r]);
            if (pass == 0) {
              gsum = fmaf(p, dpo, gsum);
              dsum = fmaf(p, dp, dsum);
            } else {
              if (one_pass) gsum = fmaf(p, dpo, gsum);
              ds8[hh * 4 + r] = (bf16)(p * (dp - dsum));
              p8o[hh * 4 + r] = DROP ? (bf16)(p * fm[DROP ? hh * 4 + r : 0]) : (bf16)p;
            }
          }
        }
        if (pass == 1) {
          if (ok) {
            *reinterpret_cast<bf16x8*>(a.dS + prow + kcol) = ds8;
            if (RC && a.Pw) *reinterpret_cast<bf16x8*>(a.Pw + prow + kcol) = p8o;
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcol_frag<KSW>(Ks, 2 * s, 2 * s + 1, dt, lane), ds8, o[dt], 0, 0, 0);
        }
      }
    }
    if (pass == 0) dsum += __shfl_xor(dsum, 16, 64), dsum += __shfl_xor(dsum, 32, 64);
    if (pass == (one_pass ? 1 : 0) && a.dgate) {
      const float gs = wave_sum(gsum);
      if (lane == 0) atomicAdd(a.dgate + h, gs);
    }
  }
  if (qok) {
    bf16* dQr = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ov = {(bf16)(o[dt][0] * a.scale), (bf16)(o[dt][1] * a.scale), (bf16)(o[dt][2] * a.scale), (bf16)(o[dt][3] * a.scale)};
      *reinterpret_cast<bf16x4*>(dQr + dt * 16 + g * 4) = ov;
    }
  }
}

// The ONE-PASS recomputing form of the kernel above with the keys STREAMED (the backward counterpart of
// attn_fwd_stream_kernel): K and V pass through LDS in double-buffered blocks of 128 keys (64 KiB instead of the 80 KiB
// key half staged behind two barriers with nothing in flight), the teacher map's piece of a block is requested before the
// next block's DMA, workgroups of one (batch, head) share an XCD.  Arithmetic per tile pair as above.
// KDR: the teacher's probabilities rebuilt from its Q, K and row lse (see attn_fwd_stream_kernel) - its K block streams
// through a third LDS tile per buffer, recompute_p forms p_t exactly as it forms the student's p.
// BATCH (round 6): every LDS read of a tile pair issued up front through inline asm (K rows + mask strip, the teacher's K
// rows, V rows, then the K^T fragments), the MFMAs behind COUNTED waits - the compiler-scheduled form drained the LDS queue five
// times per tile pair (a `s_waitcnt lgkmcnt(0)` in front of every group of MFMAs: it cannot count across the asm reads), and in
// the flavour without the teacher's recipe it waited `vmcnt(0)` - the next block's DMA - at the stored teacher map's first use
// whether or not a map was given (PT: those loads exist only in the instantiation that serves a stored teacher map).
// Arithmetic unchanged: bit-identical to BATCH = false (EVLM_ATTN_DQ_NO_BATCH=1), which stays for the dropout flavour.
template <int NW, bool KDR = false, bool DROP = false, bool BATCH = false, bool PT = true>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_stream_kernel(MAttnB a) {
  static_assert(!(BATCH && DROP), "the batched form carries no dropout mask");
  static_assert(!(KDR && PT && BATCH), "teacher recipe and stored teacher map are alternatives");
  constexpr int KBT = 8, KB = KBT * 16;
  constexpr int BUF = (KDR ? 3 : 2) * KB * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = (a.Lk + KB - 1) / KB;
  float* Ms = reinterpret_cast<float*>(smem + 2 * BUF);
  const int gx = gridDim.x, nwg = gx * gridDim.y * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int qblk = lid % gx, h = (lid / gx) % a.H, b = lid / (gx * a.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int q0 = (qblk * NW + wave) * 16;
  const int q = q0 + ql;
  const bool active = q0 < a.Lq, qok = q < a.Lq;
  const int bkv = a.kv_index ? a.kv_index[b] : b;
  const bf16* Kb = a.K + (size_t)bkv * a.Lk * a.ldk + h * DH;
  const bf16* Vb = a.V + (size_t)bkv * a.Lk * a.ldv + h * DH;
  stage_mask(a.mask, b, a.Lk, nblk * KB, Ms);
  bf16x8 dof[2], qf[2];
  float d = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0), vo = make_uint4(0, 0, 0, 0);
    if (qok) {
      v = *reinterpret_cast<const uint4*>(a.dO + ((size_t)b * a.Lq + q) * a.ldo + h * DH + ks * 32 + g * 8);
      vq = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
      vo = *reinterpret_cast<const uint4*>(a.O + ((size_t)b * a.Lq + q) * a.ldo + h * DH + ks * 32 + g * 8);
    }
    dof[ks] = *reinterpret_cast<bf16x8*>(&v);
    qf[ks] = *reinterpret_cast<bf16x8*>(&vq);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(&vo);
#pragma unroll
    for (int e = 0; e < 8; ++e) d = fmaf((float)of[e], (float)dof[ks][e], d);
  }
  d += __shfl_xor(d, 16, 64); d += __shfl_xor(d, 32, 64);
  const size_t prow = (((size_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * a.ldpr;
  const float gz = a.gate ? a.gate[h] : 1.0f;
  const float kdc = (KDR || a.Pt) ? a.kd_coef * a.kd_gout[0] : 0.f;
  const float sc = a.scale * LOG2E;
  const float lse_q = qok ? a.lse[((size_t)b * a.H + h) * a.Lq + q] : 0.f;
  // delta = sum_k p (gz dpo + kdc (p - pt)) = dO . O + kdc * rkd   (see attn_bwd_dq_long_kernel)
  const float dsum = ((KDR || a.Pt) && qok) ? fmaf(kdc, a.rkd[((size_t)b * a.H + h) * a.Lq + q], d) : d;
  const bool kd_on = !KDR && PT && a.Pt != nullptr && qok;
  const bf16* Tkb = KDR ? a.Tk + (size_t)b * a.Lk * a.tld + h * DH : nullptr;
  bf16x8 qt[2];
  float tl = 0.f;
  if (KDR) {
    tl = qok ? a.tlse[((size_t)b * a.H + h) * a.Lq + q] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qok) v = *reinterpret_cast<const uint4*>(a.Tq + ((size_t)b * a.Lq + q) * a.tld + h * DH + ks * 32 + g * 8);
      qt[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
  }
  float gsum = 0.f;
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
  const uint64_t dr8 = (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + g;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // BATCH: the lane's byte offsets inside a 16-row tile (every swizzle here depends on the row's low 4 bits only, so the tile
  // index is an immediate): row fragments of the SW_KV / SW_K tiles (krow_frag) and the K^T column fragments (vcol_frag_a)
  uint32_t offK[2], offV[2], offT[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offK[ks] = ql * 128 + swz<SW_KV>(ql, ks * 4 + g) * 16;
    offV[ks] = ql * 128 + swz<SW_K>(ql, ks * 4 + g) * 16;
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int row = g * 4 + (ql >> 2), pq = ql & 3;
    offT[dt] = row * 128 + swz<SW_KV>(row, dt * 2 + (pq >> 1)) * 16 + ((pq & 1) << 3);
  }
  stage_block<SW_K>(Vb, a.ldv, a.Lk, 0, KB, NW, smem);
  stage_block<SW_KV>(Kb, a.ldk, a.Lk, 0, KB, NW, smem + KB * 128);
  if (KDR) stage_block<SW_K>(Tkb, a.tld, a.Lk, 0, KB, NW, smem + 2 * KB * 128);
  for (int blk = 0; blk < nblk; ++blk) {
    // (measured and not kept, round 6: a counted `vmcnt(4)` + raw s_barrier here, leaving the previous block's four dS stores in
    // flight instead of draining their acknowledgements - 323 / 353 us against 305 / 364 at 577 / 901 keys: no gain; nor did a
    // chunk swizzle of kernel B's tiles that spreads its Q row reads over all bank groups: SQ_LDS_BANK_CONFLICT unchanged)
    stage_wait();
    __syncthreads();
    bf16x8 t8[KBT / 2];
    if (kd_on) {
#pragma unroll
      for (int s2 = 0; s2 < KBT / 2; ++s2) {
        const int kcol = blk * KB + s2 * 32 + g * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (kcol < a.ldpr) v = *reinterpret_cast<const uint4*>(a.Pt + prow + kcol);
        t8[s2] = *reinterpret_cast<bf16x8*>(&v);
      }
    }
    if (blk + 1 < nblk) {
      char* nb = smem + ((blk + 1) & 1) * BUF;
      stage_block<SW_K>(Vb, a.ldv, a.Lk, (blk + 1) * KB, KB, NW, nb);
      stage_block<SW_KV>(Kb, a.ldk, a.Lk, (blk + 1) * KB, KB, NW, nb + KB * 128);
      if (KDR) stage_block<SW_K>(Tkb, a.tld, a.Lk, (blk + 1) * KB, KB, NW, nb + 2 * KB * 128);
    }

    if (!active) continue;
    const char* Vs = smem + (blk & 1) * BUF;
    const char* Ks = Vs + KB * 128;
    const char* Kts = Vs + 2 * KB * 128;
    if constexpr (BATCH) {
      // per-lane read addresses of this block's buffer; the tile (2 s + hh) and the row pair of a K^T fragment are immediates
      const uint32_t vb0 = lds_addr(Vs) + offV[0], vb1 = lds_addr(Vs) + offV[1];
      const uint32_t kb0 = lds_addr(Ks) + offK[0], kb1 = lds_addr(Ks) + offK[1];
      const uint32_t tb0 = KDR ? lds_addr(Kts) + offV[0] : 0u, tb1 = KDR ? lds_addr(Kts) + offV[1] : 0u;
      const uint32_t mb = lds_addr(reinterpret_cast<const char*>(Ms + blk * KB)) + g * 32;
      const uint32_t ktb = lds_addr(Ks);
      // software pipeline over the block's four tile pairs: the K rows + mask strip of pair S + 1 are requested in front of pair
      // S's dQ MFMAs (the teacher's K rows right behind them), so a pair starts with its first operands already in flight; at
      // most 14 LDS reads are outstanding (the counter has 4 bits)
      bf16x8 kc[4], tc[4];
      f32x4 mkc[2];
      auto req_k = [&](auto S_) {
        constexpr int S = decltype(S_)::value, T0 = 2 * S * 2048, T1 = T0 + 2048;
        kc[0] = lds_b128_a<T0>(kb0); kc[1] = lds_b128_a<T1>(kb0); kc[2] = lds_b128_a<T0>(kb1); kc[3] = lds_b128_a<T1>(kb1);
        mkc[0] = lds_f32x4_a<S * 128>(mb); mkc[1] = lds_f32x4_a<S * 128 + 16>(mb);
      };
      auto req_t = [&](auto S_) {
        constexpr int S = decltype(S_)::value, T0 = 2 * S * 2048, T1 = T0 + 2048;
        if constexpr (KDR) { tc[0] = lds_b128_a<T0>(tb0); tc[1] = lds_b128_a<T1>(tb0); tc[2] = lds_b128_a<T0>(tb1); tc[3] = lds_b128_a<T1>(tb1); }
      };
      __builtin_amdgcn_sched_barrier(0);
      req_k(std::integral_constant<int, 0>());
      req_t(std::integral_constant<int, 0>());
      auto pair = [&](auto S_) {
        constexpr int S = decltype(S_)::value, T0 = 2 * S * 2048, T1 = T0 + 2048;
        const int kcol = blk * KB + S * 32 + g * 8;
        const bool ok = qok && kcol < a.ldpr;
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 v00 = lds_b128_a<T0>(vb0), v10 = lds_b128_a<T1>(vb0), v01 = lds_b128_a<T0>(vb1), v11 = lds_b128_a<T1>(vb1);
        if constexpr (KDR) LGKM_WAIT(8); else LGKM_WAIT(4);
        const f32x4 mk0 = mkc[0], mk1 = mkc[1];
        f32x4 sa0 = (f32x4){0.f, 0.f, 0.f, 0.f}, sa1 = sa0;
        sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[0], qf[0], sa0, 0, 0, 0);
        sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[1], qf[0], sa1, 0, 0, 0);
        sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[2], qf[1], sa0, 0, 0, 0);
        sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[3], qf[1], sa1, 0, 0, 0);
        f32x4 ta0 = (f32x4){0.f, 0.f, 0.f, 0.f}, ta1 = ta0;
        if constexpr (KDR) {
          LGKM_WAIT(4);
          ta0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tc[0], qt[0], ta0, 0, 0, 0);
          ta1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tc[1], qt[0], ta1, 0, 0, 0);
          ta0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tc[2], qt[1], ta0, 0, 0, 0);
          ta1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tc[3], qt[1], ta1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 kfr[4];
        kfr[0] = tr_two_a<T0, T1>(ktb + offT[0]); kfr[1] = tr_two_a<T0, T1>(ktb + offT[1]);
        kfr[2] = tr_two_a<T0, T1>(ktb + offT[2]); kfr[3] = tr_two_a<T0, T1>(ktb + offT[3]);
        __builtin_amdgcn_sched_barrier(0);
        const float lq = qok ? lse_q : 3.0e38f, ltq = qok ? tl : 3.0e38f;
        float pr[8], ptr[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = EXP2(fmaf(sa0[r], sc, mk0[r]) - lq);
          pr[4 + r] = EXP2(fmaf(sa1[r], sc, mk1[r]) - lq);
          if constexpr (KDR) {
            ptr[r] = EXP2(fmaf(ta0[r], sc, mk0[r]) - ltq);
            ptr[4 + r] = EXP2(fmaf(ta1[r], sc, mk1[r]) - ltq);
          }
        }
        LGKM_WAIT(8);
        f32x4 ac0 = (f32x4){0.f, 0.f, 0.f, 0.f}, ac1 = ac0;
        ac0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v00, dof[0], ac0, 0, 0, 0);
        ac1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v10, dof[0], ac1, 0, 0, 0);
        ac0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v01, dof[1], ac0, 0, 0, 0);
        ac1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v11, dof[1], ac1, 0, 0, 0);
        bf16x8 d8, p8o;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pp = pr[hh * 4 + r];
            const float ex = KDR ? (ok ? kdc * (pp - ptr[hh * 4 + r]) : 0.f)
                                 : ((kd_on && ok) ? kdc * (pp - (float)t8[S][hh * 4 + r]) : 0.f);
            const float dpo = hh ? ac1[r] : ac0[r];
            const float dp = fmaf(gz, dpo, ex);
            gsum = fmaf(pp, dpo, gsum);
            d8[hh * 4 + r] = (bf16)(pp * (dp - dsum));
            p8o[hh * 4 + r] = (bf16)pp;
          }
        if (ok) {
          *reinterpret_cast<bf16x8*>(a.dS + prow + kcol) = d8;
          if (a.Pw) *reinterpret_cast<bf16x8*>(a.Pw + prow + kcol) = p8o;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S < 3) {
          req_k(std::integral_constant<int, (S < 3 ? S + 1 : 0)>());
          LGKM_WAIT(6);
        } else {
          LGKM_WAIT(0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[dt], d8, o[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S < 3) req_t(std::integral_constant<int, (S < 3 ? S + 1 : 0)>());
      };
      pair(std::integral_constant<int, 0>()); pair(std::integral_constant<int, 1>());
      pair(std::integral_constant<int, 2>()); pair(std::integral_constant<int, 3>());
      continue;
    }
#pragma unroll
    for (int s2 = 0; s2 < KBT / 2; ++s2) {
      const int kcol = blk * KB + s2 * 32 + g * 8;
      const bool ok = qok && kcol < a.ldpr;
      // (K^T fragments of this tile pair through asm reads, requested here and consumed behind the softmax-backward
      // arithmetic below: the next block's DMA stays in flight through the whole block)
      // (DROP: requested right in front of their wait instead - with the mask arithmetic in between the register allocator
      // moved the loads' destinations before the wait, which tools/check_asm_loads.py rejects: an asm load's destination
      // must not be touched before its explicit wait)
      bf16x8 kfr[4];
      if (!DROP) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) kfr[dt] = vcol_frag_a<SW_KV>(Ks, 2 * s2, 2 * s2 + 1, dt, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      float pr[8], ptr[8];
      recompute_p<SW_KV>(Ks, Ms + blk * KB, qf, s2, g, lane, sc, lse_q, qok, 0, 0, pr);
      if (KDR) recompute_p<SW_K>(Kts, Ms + blk * KB, qt, s2, g, lane, sc, tl, qok, 0, 0, ptr);
      float fm[DROP ? 8 : 1];
      if constexpr (DROP) drop_factor8(rng, dr8 + (blk * KBT / 2 + s2) * 4, fm);
      bf16x8 d8, p8o;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Vs, 2 * s2 + hh, ks, lane), dof[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pp = pr[hh * 4 + r];
          const float ex = KDR ? (ok ? kdc * (pp - ptr[hh * 4 + r]) : 0.f)
                               : ((kd_on && ok) ? kdc * (pp - (float)t8[s2][hh * 4 + r]) : 0.f);
          const float dpo = DROP ? acc[r] * fm[DROP ? hh * 4 + r : 0] : acc[r];
          const float dp = fmaf(gz, dpo, ex);
          gsum = fmaf(pp, dpo, gsum);
          d8[hh * 4 + r] = (bf16)(pp * (dp - dsum));
          p8o[hh * 4 + r] = DROP ? (bf16)(pp * fm[DROP ? hh * 4 + r : 0]) : (bf16)pp;
        }
      }
      if (ok) {
        *reinterpret_cast<bf16x8*>(a.dS + prow + kcol) = d8;
        if (a.Pw) *reinterpret_cast<bf16x8*>(a.Pw + prow + kcol) = p8o;
      }
      if (DROP) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) kfr[dt] = vcol_frag_a<SW_KV>(Ks, 2 * s2, 2 * s2 + 1, dt, lane);
      }
      TR_WAIT();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[dt], d8, o[dt], 0, 0, 0);
    }
  }
  if (a.dgate) {
    const float gs = wave_sum(gsum);
    if (lane == 0) atomicAdd(a.dgate + h, gs);
  }
  if (qok) {
    bf16* dQr = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ov = {(bf16)(o[dt][0] * a.scale), (bf16)(o[dt][1] * a.scale), (bf16)(o[dt][2] * a.scale), (bf16)(o[dt][3] * a.scale)};
      *reinterpret_cast<bf16x4*>(dQr + dt * 16 + g * 4) = ov;
    }
  }
}

// does the one-pass streaming dQ kernel take this call?  (ops._Attention.backward mirrors the predicate: it allocates no
// [B, H, Lq, Lk] probability workspace when it holds)
static bool bwd_dq_stream_applies(const MAttnB& f) {
  const char* env = getenv("EVLM_ATTN_NO_STREAM");         // (A/B switch, read per call)
  return !((env && atoi(env)) || f.Lk <= 224 || f.Lk > 1024 || f.causal || !f.lse || !f.O || f.E || ((f.Pt || f.Tq) && !f.rkd));
}
static bool launch_bwd_dq_stream(MAttnB& f, hipStream_t stream) {
  if (!bwd_dq_stream_applies(f)) return false;
  const char* keep = getenv("EVLM_ATTN_STREAM_PWS");       // (A/B switch) 1: kernel B reads the map from the workspace
  if (!(keep && atoi(keep))) { f.P = nullptr; f.Pw = nullptr; }
  constexpr int KB = 128, NW = 8;      // (NW = 4, two workgroups per CU: 388 against 367 us at 577 keys, round 6)
  const int nblk = (f.Lk + KB - 1) / KB, qtiles = (f.Lq + 15) / 16;
  dim3 grid((qtiles + NW - 1) / NW, f.H, f.B), block(64 * NW);
  const char* nb = getenv("EVLM_ATTN_DQ_NO_BATCH");        // (A/B switch, read per call) the compiler-scheduled form of round 4
  const bool batch = !(nb && atoi(nb));
  auto go = [&](auto kern, size_t bytes) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(kern, grid, block, bytes, stream, f);
    return true;
  };
  if (f.Tq) {                                            // the teacher's map rebuilt in the kernel: a third tile per buffer
    const size_t ldsr = (size_t)6 * KB * 128 + (size_t)nblk * KB * sizeof(float);
    return batch ? go(attn_bwd_dq_stream_kernel<NW, true, false, true, false>, ldsr) : go(attn_bwd_dq_stream_kernel<NW, true>, ldsr);
  }
  const size_t lds = (size_t)4 * KB * 128 + (size_t)nblk * KB * sizeof(float);
  if (f.drop_p > 0.f) return go(attn_bwd_dq_stream_kernel<NW, false, true>, lds);
  if (!batch) return go(attn_bwd_dq_stream_kernel<NW>, lds);
  return f.Pt ? go(attn_bwd_dq_stream_kernel<NW, false, false, true, true>, lds)
              : go(attn_bwd_dq_stream_kernel<NW, false, false, true, false>, lds);
}

// kernel B: one workgroup = 64 keys of one (batch, head); wave w owns key tile w.  Sums over the queries in chunks of 32:
//   dK^T[d][key] = scale * sum_q Q[q][d] dS[q][key] ;  dV^T[d][key] = gate * sum_q dO[q][d] P[q][key]
// All four operands are [32 q][64] bf16 LDS tiles read by COLUMNS (ds_read_b64_tr_b16): the reduction index (q) is
// the row index of every tile.
__device__ __forceinline__ int p_swz(int row, int chunk) { return chunk ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1); }

// stage a [32 rows][64 cols] tile: row r = source row (r0 + r) (zero if >= nrows), 8 chunks of 8 columns, col chunk c valid
// iff c0 + 8c < ncols
__device__ __forceinline__ void stage_qtile(const bf16* base, size_t ld, int r0, int nrows, int c0, int ncols, char* sm) {
  const int id = threadIdx.x;            // 256 threads = 32 rows x 8 chunks
  const int row = id >> 3, c = id & 7;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (r0 + row < nrows && c0 + c * 8 < ncols) v = *reinterpret_cast<const uint4*>(base + (size_t)(r0 + row) * ld + c0 + c * 8);
  *reinterpret_cast<uint4*>(sm + row * 128 + p_swz(row, c) * 16) = v;
}
// operand with k-slot j <-> tile row 8g + j and row/col index = tile column 16*ct + (lane&15)
__device__ __forceinline__ bf16x8 qcol_frag(const char* sm, int ct, int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  bf16x8 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q;
    const int off = row * 128 + p_swz(row, ct * 2 + (p >> 1)) * 16 + ((p & 1) << 3);
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sm + off));
    out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
  }
  return out;
}

template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(MAttnB a) {
  __shared__ __attribute__((aligned(16))) char sm[4 * 32 * 128];
  char* Qs = sm; char* dOs = sm + 4096; char* Ps = sm + 8192; char* Ss = sm + 12288;
  const int bkv = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  // No stored map and a row lse: the probabilities are REBUILT here (K Q^T again: 4 MFMAs and 8 exponentials per lane per
  // 32-query chunk) instead of read from a [B, H, Lq, Lk] workspace that kernel A would have to write - the streaming
  // kernel A of long sequences then moves 517 MB less per ViT layer at 577 tokens, and this kernel reads 517 MB less
  const bool rcp = a.P == nullptr && a.lse != nullptr;
  // DROP: dV = gate (P .* M)^T dO.  A workspace map written by kernel A carries M already (p_dropped); the forward's stored
  // map gets it as its 16-byte pieces are staged (8 consecutive keys of one query = one Philox call); a rebuilt map gets it
  // per lane - 4 keys of one query, i.e. one half of a call's 8 factors
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
  const bool mask_staged = DROP && !rcp && !a.p_dropped;
  const uint64_t lk8 = (uint64_t)((a.Lk + 7) >> 3);
  bf16x8 kf[2];
  float mk[4] = {0.f, 0.f, 0.f, 0.f};
  const float sc = a.scale * LOG2E;
  if (rcp) {
    const int krow = min(k0 + wave * 16 + ql, a.Lk - 1);             // A operand: row = key, k-slots = 8 head-dim values
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 v = *reinterpret_cast<const uint4*>(a.K + ((size_t)bkv * a.Lk + krow) * a.ldk + h * DH + ks * 32 + g * 8);
      kf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  // every query batch that attends to this K/V row (one without kv_index; the positive / hard-negative / MLM passes
  // that share an image with it): their contributions are summed here, in registers, in batch order (deterministic)
  // (the index is scanned 64 entries at a time: one vector load + ballot per wave instead of B dependent scalar loads)
  const int nscan = a.kv_index ? a.B : 1;
  for (int b0 = 0; b0 < nscan; b0 += 64) {
   unsigned long long hits = 1ull;
   if (a.kv_index) hits = __ballot(b0 + lane < a.B && a.kv_index[min(b0 + lane, a.B - 1)] == bkv);
   while (hits) {                                            // wave-uniform, identical in every wave of the block
    const int b = a.kv_index ? b0 + __builtin_amdgcn_readfirstlane(__ffsll((long long)hits) - 1) : bkv;
    hits &= hits - 1;
    const bf16* Qb = a.Q + (size_t)b * a.Lq * a.ldq + h * DH;
    const bf16* dOb = a.dO + (size_t)b * a.Lq * a.ldo + h * DH;
    const size_t pbase = ((size_t)b * a.H + h) * a.Lq * a.ldpr;
    if (rcp) {                                           // additive mask of this query batch for the lane's 4 keys (log2 domain)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + wave * 16 + 4 * g + r;
        mk[r] = (key < a.Lk ? (a.mask ? a.mask[(size_t)b * a.Lk + key] : 0.f) : -1e30f) * LOG2E;
      }
    }
    // software pipeline over the 32-query chunks: chunk i+1's global loads (Q, dO, dS [, P] pieces and the row lse) are in
    // flight in registers while chunk i is multiplied - before, every chunk was a bare round trip to memory between two
    // barriers
    const int srow = threadIdx.x >> 3, scol = threadIdx.x & 7;          // this thread's 16-byte piece of a [32][64] tile
    uint4 nq, ndo, np, ns;
    float nlq[2];
    auto fetch = [&](int q0) {
      const bool rv = q0 + srow < a.Lq;
      nq = ndo = np = ns = make_uint4(0, 0, 0, 0);
      nlq[0] = nlq[1] = 0.f;
      if (rv) {
        nq = *reinterpret_cast<const uint4*>(Qb + (size_t)(q0 + srow) * a.ldq + scol * 8);
        ndo = *reinterpret_cast<const uint4*>(dOb + (size_t)(q0 + srow) * a.ldo + scol * 8);
        if (k0 + scol * 8 < a.ldpr) {
          ns = *reinterpret_cast<const uint4*>(a.dS + pbase + (size_t)(q0 + srow) * a.ldpr + k0 + scol * 8);
          if (!rcp) np = *reinterpret_cast<const uint4*>(a.P + pbase + (size_t)(q0 + srow) * a.ldpr + k0 + scol * 8);
        }
      }
      if (rcp) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (q0 + 16 * j + ql < a.Lq) nlq[j] = a.lse[((size_t)b * a.H + h) * a.Lq + q0 + 16 * j + ql];
      }
    };
    fetch(0);
    for (int q0 = 0; q0 < a.Lq; q0 += 32) {
      __syncthreads();                                   // every wave is done with the previous chunk's tiles
      const int so = srow * 128 + p_swz(srow, scol) * 16;
      *reinterpret_cast<uint4*>(Qs + so) = nq;
      *reinterpret_cast<uint4*>(dOs + so) = ndo;
      if (DROP && mask_staged) {
        float f[8];
        drop_factor8(rng, (((uint64_t)b * a.H + h) * a.Lq + min(q0 + srow, a.Lq - 1)) * lk8 + (k0 >> 3) + scol, f);
        bf16x8 p8 = *reinterpret_cast<bf16x8*>(&np);
#pragma unroll
        for (int e = 0; e < 8; ++e) p8[e] = (bf16)((float)p8[e] * f[e]);
        np = *reinterpret_cast<uint4*>(&p8);
      }
      if (!rcp) *reinterpret_cast<uint4*>(Ps + so) = np;
      *reinterpret_cast<uint4*>(Ss + so) = ns;
      const float lq[2] = {nlq[0], nlq[1]};
      __syncthreads();
      if (q0 + 32 < a.Lq) fetch(q0 + 32);
      if (rcp) {
        // P of this wave's 16 keys x the chunk's 32 queries, rebuilt as kernel A built it (same operands in the same
        // k-slots, same fma, same lse: the same bf16 values it multiplied into dS) and written into the wave's OWN 16
        // columns of the [32 q][64 key] tile - which only this wave reads back (column tile `wave`): no barrier
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f};
          const int row = 16 * j + ql;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(Qs + row * 128 + p_swz(row, ks * 4 + g) * 16);
            sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qfr, sa, 0, 0, 0);
          }
          const bool qv = q0 + row < a.Lq;
          bf16x4 p4;
          if (DROP) {
            float f[4];
            drop_factor4(rng, (((uint64_t)b * a.H + h) * a.Lq + min(q0 + row, a.Lq - 1)) * lk8 + ((k0 + wave * 16 + 4 * g) >> 3), (g & 1) != 0, f);
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = (bf16)(qv ? EXP2(fmaf(sa[r], sc, mk[r]) - lq[j]) * f[r] : 0.f);
          } else
#pragma unroll
          for (int r = 0; r < 4; ++r) p4[r] = (bf16)(qv ? EXP2(fmaf(sa[r], sc, mk[r]) - lq[j]) : 0.f);
          *reinterpret_cast<bf16x4*>(Ps + row * 128 + p_swz(row, 2 * wave + (g >> 1)) * 16 + (g & 1) * 8) = p4;
        }
      }
      const bf16x8 bS = qcol_frag(Ss, wave, lane), bP = qcol_frag(Ps, wave, lane);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qcol_frag(Qs, dt, lane), bS, dk[dt], 0, 0, 0);
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qcol_frag(dOs, dt, lane), bP, dv[dt], 0, 0, 0);
      }
    }
   }
  }
  const int key = k0 + wave * 16 + (lane & 15);
  if (key < a.Lk) {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    bf16* dKr = a.dK + ((size_t)bkv * a.Lk + key) * a.lddk + h * DH;
    bf16* dVr = a.dV + ((size_t)bkv * a.Lk + key) * a.lddv + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 kv = {(bf16)(dk[dt][0] * a.scale), (bf16)(dk[dt][1] * a.scale), (bf16)(dk[dt][2] * a.scale), (bf16)(dk[dt][3] * a.scale)};
      bf16x4 vv = {(bf16)(dv[dt][0] * gz), (bf16)(dv[dt][1] * gz), (bf16)(dv[dt][2] * gz), (bf16)(dv[dt][3] * gz)};
      *reinterpret_cast<bf16x4*>(dKr + dt * 16 + g * 4) = kv;
      *reinterpret_cast<bf16x4*>(dVr + dt * 16 + g * 4) = vv;
    }
  }
}

// Kernel B, STREAMED (round 6): the form of the kernel above for self-attention whose map is rebuilt (no stored P, row lse
// given, no kv_index, no dropout - the ViT layers at 577 / 901 tokens of the ITR / VQA steps).  Same arithmetic, same operands
// in the same k-slots, same order of the sums over the query chunks (bit-identical results); what changes is how a chunk
// reaches the MFMAs.  The old loop was  barrier / 4 register-staged ds_writes / barrier / 20 transposing reads each followed
// by its own `s_waitcnt lgkmcnt(0)` and one MFMA  (rocprofv3 at 64 x 12 x 577: 28 % of the wave cycles in issue stalls, 41 %
// parked at a wait, matrix pipe 11 % busy).  Here
//   * the Q / dO / dS tiles of chunk i+1 arrive by LDS-DMA into the other half of a double buffer while chunk i is multiplied
//     (no staging registers, no ds_writes, ONE barrier per chunk);
//   * all 20 transposing reads of a chunk are issued back to back (inline asm, counted waits: the dK operands are complete at
//     lgkmcnt(10), the dV operands at 0), the 8 MFMAs follow in two groups;
//   * workgroups of one (batch, head) share an XCD (its Q and dO are read by ceil(Lk / 64) workgroups).
// Rows of the last chunk beyond Lq: the DMA reads row Lq-1 again; their P is zero by construction and their dS rows are
// zeroed in LDS before use.
template <int OFF> __device__ __forceinline__ bf16x8 tr_pair_a(uint32_t addr) {   // two transposing reads, rows 4 apart
  bf16x4 t0, t1;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t0) : "v"(addr), "i"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t1) : "v"(addr), "i"(OFF + 512));
  bf16x8 out;
  out[0] = t0[0]; out[1] = t0[1]; out[2] = t0[2]; out[3] = t0[3];
  out[4] = t1[0]; out[5] = t1[1]; out[6] = t1[2]; out[7] = t1[3];
  return out;
}
#define DKVS_TILE 4096
#define DKVS_BUF (3 * DKVS_TILE)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_dkv_stream_kernel(MAttnB a) {
  __shared__ __attribute__((aligned(16))) char sm[2 * DKVS_BUF + DKVS_TILE];      // 2 x (Q, dO, dS) + P
  const int gx = gridDim.x, nwg = gx * gridDim.y * gridDim.z;
  int lid = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int k0 = (lid % gx) * 64, h = (lid / gx) % a.H, b = lid / (gx * a.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const float sc = a.scale * LOG2E;
  bf16x8 kf[2];
  {
    const int krow = min(k0 + wave * 16 + ql, a.Lk - 1);             // A operand: row = key, k-slots = 8 head-dim values
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 v = *reinterpret_cast<const uint4*>(a.K + ((size_t)b * a.Lk + krow) * a.ldk + h * DH + ks * 32 + g * 8);
      kf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
  }
  float mk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = k0 + wave * 16 + 4 * g + r;
    mk[r] = (key < a.Lk ? (a.mask ? a.mask[(size_t)b * a.Lk + key] : 0.f) : -1e30f) * LOG2E;
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const bf16* Qb = a.Q + (size_t)b * a.Lq * a.ldq + h * DH;
  const bf16* dOb = a.dO + (size_t)b * a.Lq * a.ldo + h * DH;
  const bf16* dSb = a.dS + ((size_t)b * a.H + h) * a.Lq * a.ldpr;
  const float* lseb = a.lse + ((size_t)b * a.H + h) * a.Lq;
  // DMA: this wave fills rows 8 wave .. 8 wave + 7 of each tile; the lane's LDS slot (row, cs) takes source chunk p_swz(row, cs)
  const int srow = wave * 8 + (lane >> 3), sc8 = p_swz(srow, lane & 7) * 8;
  const int sck = min(k0 + sc8, a.ldpr - 8);                          // (columns past the map's row: any in-bounds piece)
  auto stage = [&](int q0, char* buf) {
    const size_t r = (size_t)min(q0 + srow, a.Lq - 1);
    char* dst = buf + wave * 8 * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Qb + r * a.ldq + sc8),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dOb + r * a.ldo + sc8),
                                     (__attribute__((address_space(3))) void*)(dst + DKVS_TILE), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dSb + r * a.ldpr + sck),
                                     (__attribute__((address_space(3))) void*)(dst + 2 * DKVS_TILE), 16, 0, 0);
  };
  // transposing-read addresses of buffer 0 (qcol_frag's map): column tile ct -> chunk slot p_swz(row, 2 ct + (p >> 1)); the
  // second read of a pair sits 4 rows = 512 bytes further (same swizzle: p_swz ignores that row bit)
  const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  uint32_t tra[4];
  {
    const int q = ql >> 2, pp = ql & 3, row = g * 8 + q;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) tra[ct] = sb + row * 128 + p_swz(row, ct * 2 + (pp >> 1)) * 16 + ((pp & 1) << 3);
  }
  const uint32_t trw = sb + (g * 8 + (ql >> 2)) * 128 + p_swz(g * 8 + (ql >> 2), wave * 2 + ((ql & 3) >> 1)) * 16 + ((ql & 1) << 3);
  float nlq[2] = {0.f, 0.f};
  auto fetch_lse = [&](int q0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) nlq[j] = (q0 + 16 * j + ql < a.Lq) ? lseb[q0 + 16 * j + ql] : 0.f;
  };
  stage(0, sm);
  fetch_lse(0);
  auto chunk = [&](int q0, auto bufc) {
    constexpr int BO = decltype(bufc)::value * DKVS_BUF;
    stage_wait();
    __syncthreads();                          // chunk q0 has landed; every wave is done with the other buffer
    const float lq[2] = {nlq[0], nlq[1]};
    if (q0 + 32 < a.Lq) {
      stage(q0 + 32, sm + (DKVS_BUF - BO));
      fetch_lse(q0 + 32);
    } else if (q0 + 32 > a.Lq) {              // ragged last chunk: zero the dS rows of the queries that do not exist
      const int zr = threadIdx.x >> 3;
      if (q0 + zr >= a.Lq) *reinterpret_cast<uint4*>(sm + BO + 2 * DKVS_TILE + zr * 128 + (threadIdx.x & 7) * 16) = make_uint4(0, 0, 0, 0);
      __syncthreads();
    }
    const char* Qs = sm + BO;
    char* Ps = sm + 2 * DKVS_BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {             // P of this wave's 16 keys x 32 queries, into the wave's OWN columns of the P tile
      f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int row = 16 * j + ql;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(Qs + row * 128 + p_swz(row, ks * 4 + g) * 16);
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qfr, sa, 0, 0, 0);
      }
      const bool qv = q0 + row < a.Lq;
      bf16x4 p4;
#pragma unroll
      for (int r = 0; r < 4; ++r) p4[r] = (bf16)(qv ? EXP2(fmaf(sa[r], sc, mk[r]) - lq[j]) : 0.f);
      *reinterpret_cast<bf16x4*>(Ps + row * 128 + p_swz(row, 2 * wave + (g >> 1)) * 16 + (g & 1) * 8) = p4;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fq[4], fo[4];
    const bf16x8 bS = tr_pair_a<BO + 2 * DKVS_TILE>(trw);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) fq[dt] = tr_pair_a<BO>(tra[dt]);
    const bf16x8 bP = tr_pair_a<2 * DKVS_BUF>(trw);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) fo[dt] = tr_pair_a<BO + DKVS_TILE>(tra[dt]);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[dt], bS, dk[dt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo[dt], bP, dv[dt], 0, 0, 0);
  };
  for (int q0 = 0; q0 < a.Lq; q0 += 64) {
    chunk(q0, std::integral_constant<int, 0>());
    if (q0 + 32 < a.Lq) chunk(q0 + 32, std::integral_constant<int, 1>());
  }
  const int key = k0 + wave * 16 + ql;
  if (key < a.Lk) {
    const float gz = a.gate ? a.gate[h] : 1.0f;
    bf16* dKr = a.dK + ((size_t)b * a.Lk + key) * a.lddk + h * DH;
    bf16* dVr = a.dV + ((size_t)b * a.Lk + key) * a.lddv + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 kv = {(bf16)(dk[dt][0] * a.scale), (bf16)(dk[dt][1] * a.scale), (bf16)(dk[dt][2] * a.scale), (bf16)(dk[dt][3] * a.scale)};
      bf16x4 vv = {(bf16)(dv[dt][0] * gz), (bf16)(dv[dt][1] * gz), (bf16)(dv[dt][2] * gz), (bf16)(dv[dt][3] * gz)};
      *reinterpret_cast<bf16x4*>(dKr + dt * 16 + g * 4) = kv;
      *reinterpret_cast<bf16x4*>(dVr + dt * 16 + g * 4) = vv;
    }
  }
}
static bool bwd_dkv_stream_applies(const MAttnB& f) {
  const char* env = getenv("EVLM_ATTN_DKV_NO_STREAM");      // (A/B switch, read per call)
  return !(env && atoi(env)) && !f.P && f.lse && !f.kv_index && f.drop_p == 0.f && f.ldpr >= 64;
}

// =============================================================================================
// single-pass backward: every query and every key of one (batch, head) in ONE workgroup (self-attention with Lq, Lk <= 16 NT:
// the ViT's 197 tokens, the 30-token text passes).  dS never goes to HBM and P / dO / Q are read once:
//   phase 1  (kernel A's arithmetic, a wave owns 16 queries x all keys, G query tiles per wave): dS^T, P^T in registers,
//            dQ written; dS^T is transposed into LDS as [32 q][32 key] tiles;
//   phase 2  K and V are dead: Q and dO are staged in their place; wave w owns key tile w:
//            dK^T[d][key] = scale * sum_q Q[q][d] dS[q][key]   (both operands by column reads, as in kernel B);
//   phase 3  the P^T registers replace dS in the same LDS tiles:  dV^T[d][key] = gate * sum_q dO[q][d] P[q][key].
// LDS: (NT/2)^2 tiles of 2 KiB + 2 x NT x 2 KiB = 154 KiB at NT = 14 (one workgroup per CU; the kernel is HBM-bound and
// moves ~half the bytes of kernels A + B).
// =============================================================================================
__device__ __forceinline__ int s_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 3); }
// B operand from a [32 q][32 key] tile (64-byte rows): k-slot j <-> tile row 8g + j, column = key 16*ct + (lane&15)
__device__ __forceinline__ bf16x8 scol_frag(const char* tile, int ct, int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  bf16x8 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q;
    const int off = row * 64 + s_swz(row, ct * 2 + (p >> 1)) * 16 + ((p & 1) << 3);
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(tile + off));
    out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
  }
  return out;
}

template <int NT, int NW, bool RC, bool NOCAUSAL = false, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void attn_bwd_fused_kernel(MAttnB a) {
  static_assert(!DROP || RC, "the single-pass kernel carries the dropout mask in its recomputing form only");
  constexpr int G = (NT + NW - 1) / NW, KC = NT / 2, QC = NT / 2;
  constexpr int KSW = RC ? SW_KV : SW_V;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool EARLY = NT <= 4;           // short sequences: LDS to spare - Q and dO get their own tiles, staged up front
  char* Ss = smem;                          // [QC][KC] tiles of [32 q][32 key]: dS, then P
  char* Ks = smem + QC * KC * 2048;         // phase 1: K (column reads; RC: row reads too);  phases 2-3: Q as [QC] tiles of [32 q][64] (p_swz)
  char* Vs = Ks + NT * 16 * 128;            // phase 1: V (k_swz);  phases 2-3: dO likewise
  char* Qs = EARLY ? Vs + NT * 16 * 128 : Ks;
  char* dOs = EARLY ? Qs + NT * 16 * 128 : Vs;
  float* Ms = reinterpret_cast<float*>((EARLY ? dOs : Vs) + NT * 16 * 128);      // RC: mask row of this batch
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ql = lane & 15;
  auto stage_q_do = [&]() {                          // Q, dO as [32 q][64] tiles (p_swz), rows >= Lq zero
    for (int i = tid; i < QC * 256; i += 64 * NW) {
      const int tile = i >> 8, row = (i >> 3) & 31, c = i & 7, q = tile * 32 + row;
      uint4 vq = make_uint4(0, 0, 0, 0), vo = make_uint4(0, 0, 0, 0);
      if (q < a.Lq) {
        vq = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + c * 8);
        vo = *reinterpret_cast<const uint4*>(a.dO + ((size_t)b * a.Lq + q) * a.ldo + h * DH + c * 8);
      }
      const int off = tile * 4096 + row * 128 + p_swz(row, c) * 16;
      *reinterpret_cast<uint4*>(Qs + off) = vq;
      *reinterpret_cast<uint4*>(dOs + off) = vo;
    }
  };
  stage_rows<KSW>(a.K + (size_t)b * a.Lk * a.ldk + h * DH, a.ldk, a.Lk, NT * 16, Ks);
  stage_rows<SW_K>(a.V + (size_t)b * a.Lk * a.ldv + h * DH, a.ldv, a.Lk, NT * 16, Vs);
  if (EARLY) stage_q_do();
  if (RC) stage_mask(a.mask, b, a.Lk, NT * 16, Ms);
  stage_wait();
  __syncthreads();
  const float gz = a.gate ? a.gate[h] : 1.0f;
  const float kdc = a.Pt ? a.kd_coef * a.kd_gout[0] : 0.f;
  const float sc = a.scale * LOG2E;
  bf16x4 pvg[G][NT];
  float gsum = 0.f;
  DropRng rng;
  if (DROP) rng = drop_rng(a.rng, a.call, a.drop_p);
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int qt = gi * NW + wave;
    if (qt >= NT) continue;                         // wave-uniform
    const int q = qt * 16 + ql;
    const bool qok = q < a.Lq;
    bf16x8 dof[2], qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
      if (qok) v = *reinterpret_cast<const uint4*>(a.dO + ((size_t)b * a.Lq + q) * a.ldo + h * DH + ks * 32 + g * 8);
      if (RC && qok) vq = *reinterpret_cast<const uint4*>(a.Q + ((size_t)b * a.Lq + q) * a.ldq + h * DH + ks * 32 + g * 8);
      dof[ks] = *reinterpret_cast<bf16x8*>(&v);
      qf[ks] = *reinterpret_cast<bf16x8*>(&vq);
    }
    const size_t prow = (((size_t)b * a.H + h) * a.Lq + q) * a.ldpr;
    const float lse_q = (RC && qok) ? a.lse[((size_t)b * a.H + h) * a.Lq + q] : 0.f;
    f32x4 acc[NT];
    f32x4 pf[RC ? NT : 1];
    float dsum = 0.f;
    // The 16-byte pieces of the teacher map / the external dP of tile pair s + 1 are requested BEFORE pair s is worked on
    // (round 5): read where they are used, every pair was a bare round trip to memory inside a wave that shares its SIMD
    // with one other wave - 7 pairs x 2 query tiles of exposed latency per workgroup.
    auto piece = [&](const bf16* base, int s_) -> uint4 {
      const int kc = s_ * 32 + g * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (base && qok && kc < a.ldpr) v = *reinterpret_cast<const uint4*>(base + prow + kc);
      return v;
    };
    uint4 t_nx = piece(a.Pt, 0), e_nx = piece(a.E, 0);
#pragma unroll
    for (int s = 0; s < NT / 2; ++s) {
      const int kcol = s * 32 + g * 8;
      const bool ok = qok && kcol < a.ldpr;
      const uint4 t_cu = t_nx, e_cu = e_nx;
      if (s + 1 < NT / 2) { t_nx = piece(a.Pt, s + 1); e_nx = piece(a.E, s + 1); }
      float pr[8], ex[8];
      float fm[DROP ? 8 : 1];
      if constexpr (DROP) drop_factor8(rng, (((uint64_t)b * a.H + h) * a.Lq + (qok ? q : 0)) * (uint64_t)((a.Lk + 7) >> 3) + s * 4 + g, fm);
#pragma unroll
      for (int r = 0; r < 8; ++r) { pr[r] = 0.f; ex[r] = 0.f; }
      if (RC) recompute_p<KSW, NOCAUSAL>(Ks, Ms, qf, s, g, lane, sc, lse_q, qok, a.causal, q, pr);
      if (ok) {
        if (!RC) {
          const bf16x8 p8 = *reinterpret_cast<const bf16x8*>(a.P + prow + kcol);
#pragma unroll
          for (int r = 0; r < 8; ++r) pr[r] = (float)p8[r];
        }
        if (a.E) {
          const bf16x8 e8 = *reinterpret_cast<const bf16x8*>(&e_cu);
#pragma unroll
          for (int r = 0; r < 8; ++r) ex[r] = (float)e8[r];
        }
        if (a.Pt) {
          const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&t_cu);
#pragma unroll
          for (int r = 0; r < 8; ++r) ex[r] = fmaf(kdc, pr[r] - (float)t8[r], ex[r]);
        }
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int t = 2 * s + hh;
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(krow_frag(Vs, t, ks, lane), dof[ks], acc[t], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = pr[hh * 4 + r];
          const float dpo = DROP ? acc[t][r] * fm[DROP ? hh * 4 + r : 0] : acc[t][r];
          pvg[gi][t][r] = DROP ? (bf16)(p * fm[DROP ? hh * 4 + r : 0]) : (bf16)p;     // (phase 3: dV from P .* M)
          if (RC) pf[RC ? t : 0][r] = p;
          gsum = fmaf(p, dpo, gsum);
          const float dp = fmaf(gz, dpo, ex[hh * 4 + r]);
          acc[t][r] = dp;
          dsum = fmaf(p, dp, dsum);
        }
      }
    }
    dsum += __shfl_xor(dsum, 16, 64); dsum += __shfl_xor(dsum, 32, 64);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    char* srow = Ss + (size_t)((qt >> 1) * KC) * 2048 + ((qt & 1) * 16 + ql) * 64;
    const int sr = (qt & 1) * 16 + ql;
#pragma unroll
    for (int s = 0; s < NT / 2; ++s) {
      bf16x8 d8;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = RC ? pf[RC ? 2 * s : 0][r] : (float)pvg[gi][2 * s][r];
        const float p1 = RC ? pf[RC ? 2 * s + 1 : 0][r] : (float)pvg[gi][2 * s + 1][r];
        d8[r] = (bf16)(p0 * (acc[2 * s][r] - dsum));
        d8[4 + r] = (bf16)(p1 * (acc[2 * s + 1][r] - dsum));
      }
      *reinterpret_cast<bf16x8*>(srow + s * 2048 + s_swz(sr, g) * 16) = d8;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcol_frag<KSW>(Ks, 2 * s, 2 * s + 1, dt, lane), d8, o[dt], 0, 0, 0);
    }
    if (qok) {
      bf16* dQr = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + h * DH;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov = {(bf16)(o[dt][0] * a.scale), (bf16)(o[dt][1] * a.scale), (bf16)(o[dt][2] * a.scale), (bf16)(o[dt][3] * a.scale)};
        *reinterpret_cast<bf16x4*>(dQr + dt * 16 + g * 4) = ov;
      }
    }
  }
  if (a.dgate) {
    const float gs = wave_sum(gsum);
    if (lane == 0) atomicAdd(a.dgate + h, gs);
  }
  __syncthreads();                                   // K and V are dead, dS is complete
  if (!EARLY) {
    stage_q_do();
    __syncthreads();
  }
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {             // 0: dK from dS and Q;  1: dV from P and dO
    if (pass == 1) {
      __syncthreads();                               // every wave has read its dS columns
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const int qt = gi * NW + wave;
        if (qt >= NT) continue;
        const int sr = (qt & 1) * 16 + ql;
        char* srow = Ss + (size_t)((qt >> 1) * KC) * 2048 + sr * 64;
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
          bf16x8 d8;
#pragma unroll
          for (int r = 0; r < 4; ++r) { d8[r] = pvg[gi][2 * s][r]; d8[4 + r] = pvg[gi][2 * s + 1][r]; }
          *reinterpret_cast<bf16x8*>(srow + s * 2048 + s_swz(sr, g) * 16) = d8;
        }
      }
      __syncthreads();
    }
    const char* Xs = pass ? dOs : Qs;
    const float mul = pass ? gz : a.scale;
    bf16* out = pass ? a.dV : a.dK;
    const int ldx = pass ? a.lddv : a.lddk;
    for (int kt = wave; kt < NT; kt += NW) {
      f32x4 d[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) d[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qc = 0; qc < QC; ++qc) {
        const bf16x8 bS = scol_frag(Ss + (qc * KC + (kt >> 1)) * 2048, kt & 1, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          d[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qcol_frag(Xs + qc * 4096, dt, lane), bS, d[dt], 0, 0, 0);
      }
      const int key = kt * 16 + (lane & 15);
      if (key < a.Lk) {
        bf16* r = out + ((size_t)b * a.Lk + key) * ldx + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16x4 v = {(bf16)(d[dt][0] * mul), (bf16)(d[dt][1] * mul), (bf16)(d[dt][2] * mul), (bf16)(d[dt][3] * mul)};
          *reinterpret_cast<bf16x4*>(r + dt * 16 + g * 4) = v;
        }
      }
    }
  }
}

template <int NT, bool RC>
static bool launch_bwd_fused(const MAttnB& f, hipStream_t stream) {
  constexpr int NW = NT < 8 ? NT : 8;
  const char* env = getenv("EVLM_ATTN_BWD_SPLIT");          // (debug / A-B switch: force kernels A + B)
  if ((env && atoi(env)) || f.kv_index || f.Lq > NT * 16 || f.Lk > NT * 16) return false;
  const size_t lds = (size_t)(NT / 2) * (NT / 2) * 2048 + (size_t)(NT <= 4 ? 4 : 2) * NT * 16 * 128 + (RC ? NT * 16 * sizeof(float) : 0);
#define FUSED_LAUNCH(NOC_, DROP_)                                                                                         \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<NT, NW, RC, NOC_, DROP_>,                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((attn_bwd_fused_kernel<NT, NW, RC, NOC_, DROP_>), dim3(f.H, f.B), dim3(64 * NW), lds, stream, f); \
  } while (0)
  if (f.drop_p > 0.f) {                     // probability dropout: the recomputing form only (the stored-map form takes kernels A + B),
    if constexpr (RC && NT <= 4) {          // text-length problems only (at NT = 14 the mask registers spill: kernels A + B)
      if (!f.causal) FUSED_LAUNCH(true, true); else FUSED_LAUNCH(false, true);
      return true;
    }
    return false;
  }
  if (RC && !f.causal) FUSED_LAUNCH(true, false);      // (the encoders: no causal clamp in the recomputation)
  else FUSED_LAUNCH(false, false);
#undef FUSED_LAUNCH
  return true;
}

template <int NT, bool RC>
static void launch_bwd_dq(const MAttnB& f, hipStream_t stream) {
  constexpr int MAXW = NT <= 14 ? 8 : 4;
  const size_t lds = (size_t)2 * NT * 16 * 128 + (RC ? NT * 16 * sizeof(float) : 0);
  const int nw = imin(MAXW, (f.Lq + 15) / 16);
  dim3 grid((f.Lq + 16 * nw - 1) / (16 * nw), f.H, f.B), block(64 * nw);
  if (f.drop_p > 0.f) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_kernel<NT, MAXW, RC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<NT, MAXW, RC, true>), grid, block, lds, stream, f);
    return;
  }
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_kernel<NT, MAXW, RC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<NT, MAXW, RC>), grid, block, lds, stream, f);
}

template <int NT, bool RC>
static void launch_bwd_dq_long(const MAttnB& f, hipStream_t stream) {
  constexpr int MAXW = 8;
  const size_t lds = (size_t)NT * 16 * 128 + (RC ? NT * 16 * sizeof(float) : 0);
  const int nw = imin(MAXW, (f.Lq + 15) / 16);
  dim3 grid((f.Lq + 16 * nw - 1) / (16 * nw), f.H, f.B), block(64 * nw);
  if (f.drop_p > 0.f) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_long_kernel<NT, MAXW, RC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_bwd_dq_long_kernel<NT, MAXW, RC, true>), grid, block, lds, stream, f);
    return;
  }
  (void)hipFuncSetAttribute((const void*)attn_bwd_dq_long_kernel<NT, MAXW, RC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_bwd_dq_long_kernel<NT, MAXW, RC>), grid, block, lds, stream, f);
}

int evlm_attention_bwd_mfma(const evlm_attn_bwd_args* a, hipStream_t stream, int* handled) {
  *handled = 0;
  if (a->dtype != EVLM_BF16 || a->p_dtype != EVLM_BF16 || a->dh != DH || a->Lk > 928) return 0;
  if ((a->ldq | a->ldk | a->ldv | a->ldo | a->lddq | a->lddk | a->lddv | a->ldpr) % 8 != 0) return 0;
  MAttnB f;
  f.Q = (const bf16*)a->Q; f.K = (const bf16*)a->K; f.V = (const bf16*)a->V; f.P = (const bf16*)a->P;
  f.dO = (const bf16*)a->dO; f.E = (const bf16*)a->dP_ext; f.gate = a->head_gate; f.kv_index = a->kv_index;
  f.Bkv = a->kv_index ? a->Bkv : a->B;
  f.dS = (bf16*)a->dS; f.dQ = (bf16*)a->dQ; f.dK = (bf16*)a->dK; f.dV = (bf16*)a->dV; f.dgate = a->dgate;
  f.B = a->B; f.H = a->H; f.Lq = a->Lq; f.Lk = a->Lk; f.ldq = a->ldq; f.ldk = a->ldk; f.ldv = a->ldv; f.ldo = a->ldo;
  f.lddq = a->lddq; f.lddk = a->lddk; f.lddv = a->lddv; f.ldpr = a->ldpr; f.scale = a->scale;
  f.Pt = (const bf16*)a->kd_teacher; f.kd_gout = a->kd_gout;
  f.kd_coef = a->kd_teacher ? 2.0f * a->kd_weight / ((float)a->B * a->H * a->Lq * a->Lk) : 0.f;
  if (a->kd_teacher && !a->kd_gout) return evlm_set_error("evlm_attention_bwd: kd_teacher without kd_gout");
  f.lse = a->lse; f.mask = a->mask; f.causal = a->causal; f.Pw = nullptr;
  f.O = (const bf16*)a->O; f.rkd = a->kd_rowdot;
  f.Tq = (const bf16*)a->kd_tq; f.Tk = (const bf16*)a->kd_tk; f.tld = a->kd_tld; f.tlse = a->kd_tlse;
  f.drop_p = a->dropout_p; f.rng = a->rng_state; f.call = a->call_id; f.p_dropped = 0;
  if (f.drop_p > 0.f && (f.Tq || (a->Lk > 224 && a->Lk <= 416 && false)))
    return evlm_set_error("evlm_attention_bwd: probability dropout does not combine with kd_tq");
  if (f.Tq) {
    if (!(f.Tk && f.tlse && a->kd_gout && !a->kd_teacher && !a->mask && !a->kv_index && a->Lq == a->Lk && a->kd_tld % 8 == 0))
      return evlm_set_error("evlm_attention_bwd: kd_tq needs kd_tk, kd_tlse and kd_gout, self-attention without a mask, no kd_teacher");
    f.kd_coef = 2.0f * a->kd_weight / ((float)a->B * a->H * a->Lq * a->Lk);
    if (!bwd_dq_stream_applies(f))
      return evlm_set_error("evlm_attention_bwd: kd_tq is served by the one-pass streaming kernel only (225..928 keys, lse, O, "
                            "kd_rowdot, no dP_ext, no causal mask)");
  }
  const bool rc = a->lse != nullptr;
  if (rc && !evlm_attention_lse_supported(a->dtype, a->dh, a->Lk, 0.f))
    return evlm_set_error("evlm_attention_bwd: the recomputing form serves Lk <= 224 and 417..928 (got %d)", a->Lk);
  if (!rc && !a->P) return evlm_set_error("evlm_attention_bwd: neither the probability map nor the row lse was given");
  bool fused = false;                       // whole (batch, head) problems that fit one workgroup: one launch, no dS in HBM
  if (rc) {
    if (a->Lk <= 32) fused = launch_bwd_fused<2, true>(f, stream);
    else if (a->Lk <= 64) fused = launch_bwd_fused<4, true>(f, stream);
    else if (a->Lk <= 224) fused = launch_bwd_fused<14, true>(f, stream);
  } else {
    if (a->Lk <= 32) fused = launch_bwd_fused<2, false>(f, stream);
    else if (a->Lk <= 64) fused = launch_bwd_fused<4, false>(f, stream);
    else if (a->Lk <= 224) fused = launch_bwd_fused<14, false>(f, stream);
  }
  if (fused) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return evlm_set_error("evlm_attention_bwd(mfma, fused): %s", hipGetErrorString(e));
    *handled = 1;
    return 0;
  }
  if (!f.dS) return evlm_set_error("evlm_attention_bwd: the two-kernel path needs the dS workspace");
  if (rc) {
    // kernel B reads the map kernel A rebuilds (the same bf16 values the single-pass kernel hands its third phase); only
    // without a workspace does it fall back to the map the forward stored
    if (a->P_ws) {
      f.Pw = (bf16*)a->P_ws;
      f.P = (const bf16*)a->P_ws;
      f.p_dropped = 1;                                    // (kernel A writes P .* M into the workspace)
    } else if (!a->P && !bwd_dq_stream_applies(f) && a->Lk > 224)      // (the streaming pair needs neither: kernel B rebuilds the map;
      return evlm_set_error("evlm_attention_bwd: the two-kernel recomputing path needs P or the P_ws workspace");   // round 6: so do <= 224 keys)
    // (a grouped-by-K/V-row form of kernel A - the backward counterpart of attn_fwd_grouped_kernel - was measured 20 %
    // SLOWER than this per-batch launch on the GD shape, tools/attn_bench.py: a backward task is VALU-bound work of ~20 k
    // cycles per wave, so the staging it would share is a small part of it, and 256 registers leave one workgroup per CU)
    if (a->Lk <= 32) launch_bwd_dq<2, true>(f, stream);
    else if (a->Lk <= 64) launch_bwd_dq<4, true>(f, stream);
    else if (a->Lk <= 224) launch_bwd_dq<14, true>(f, stream);
    else if (launch_bwd_dq_stream(f, stream)) {}      // (kernel B rebuilds the map itself: f.P / f.Pw were cleared)
    else if (a->Lk <= 640) launch_bwd_dq_long<40, true>(f, stream);
    else launch_bwd_dq_long<60, true>(f, stream);
  } else if (a->Lk <= 32) launch_bwd_dq<2, false>(f, stream);
  else if (a->Lk <= 64) launch_bwd_dq<4, false>(f, stream);
  else if (a->Lk <= 224) launch_bwd_dq<14, false>(f, stream);
  else if (a->Lk <= 416) launch_bwd_dq<26, false>(f, stream);
  else if (a->Lk <= 640) launch_bwd_dq_long<40, false>(f, stream);      // long sequences: two passes over the keys, nothing spilled
  else launch_bwd_dq_long<60, false>(f, stream);
  dim3 gridB((a->Lk + 63) / 64, a->H, f.Bkv), block(256);
  if (bwd_dkv_stream_applies(f)) hipLaunchKernelGGL(attn_bwd_dkv_stream_kernel, gridB, block, 0, stream, f);
  else if (f.drop_p > 0.f) hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<true>, gridB, block, 0, stream, f);
  else hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<false>, gridB, block, 0, stream, f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return evlm_set_error("evlm_attention_bwd(mfma): %s", hipGetErrorString(e));
  *handled = 1;
  return 0;
}

extern "C" int evlm_attention_lse_supported(int dtype, int dh, int Lk, float dropout_p) {
  // Lk <= 224: one workgroup per (batch, head) holds a whole row in registers.  417..928 keys (384 x 384 / 480 x 480 images):
  // the long-sequence kernel recomputes per key half; given the forward's O (and kd_rowdot with a fused distillation term)
  // and no external dP it needs ONE pass over the keys (round 4; the two-pass recomputing form of round 3 cost 4.6 % of the
  // ITR-384 step and stays for calls with a dP_ext).  Whether to ask for the lse form is the caller's choice per call
  // (ops._Attention: when nobody wants the map).  225..416 keys: stored-map form only (the one-pass kernel A of that bucket
  // holds a whole row in registers).
  // (round 6: with or without probability dropout - the MFMA kernels regenerate the mask)
  if (dtype != EVLM_BF16 || dh != DH || dropout_p < 0.f || dropout_p >= 1.f) return 0;
  return Lk <= 224 || (Lk > 416 && Lk <= 928);
}

// returns 0 and sets *handled = 1 when a specialised kernel took the call
int evlm_attention_fwd_mfma(const evlm_attn_fwd_args* a, hipStream_t stream, int* handled) {
  *handled = 0;
  if (a->dtype != EVLM_BF16 || a->p_dtype != EVLM_BF16 || a->dh != DH || a->Lk > 928) return 0;
  if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8 != 0 || (a->P && a->ldpr % 8 != 0)) return 0;
  MAttnF f;
  f.Q = (const bf16*)a->Q; f.K = (const bf16*)a->K; f.V = (const bf16*)a->V; f.kv_index = a->kv_index;
  f.mask = a->mask; f.gate = a->head_gate; f.O = (bf16*)a->O; f.P = (bf16*)a->P;
  f.B = a->B; f.H = a->H; f.Lq = a->Lq; f.Lk = a->Lk; f.ldq = a->ldq; f.ldk = a->ldk; f.ldv = a->ldv; f.ldo = a->ldo;
  f.ldpr = a->ldpr; f.scale = a->scale; f.causal = a->causal;
  f.Pt = (const bf16*)a->kd_teacher; f.kd = a->kd_loss;
  f.kd_coef = a->kd_teacher ? a->kd_weight / ((float)a->B * a->H * a->Lq * a->Lk) : 0.f;
  f.lse = a->lse; f.rkd = a->kd_rowdot;
  f.Tq = (const bf16*)a->kd_tq; f.Tk = (const bf16*)a->kd_tk; f.tld = a->kd_tld; f.tlse = a->kd_tlse;
  f.drop_p = a->dropout_p; f.rng = a->rng_state; f.call = a->call_id;
  if (f.drop_p > 0.f) {
    // probability dropout (round 6): <= 224 keys on the whole-row / grouped kernels, 225..928 on the streaming pair; the
    // combinations nothing on the path issues stay with the shape-generic kernels (which then refuse lse / kd_teacher)
    if (f.Tq) return evlm_set_error("evlm_attention_fwd: probability dropout does not combine with kd_tq");
    if (a->Lk > 224 && (a->causal || getenv("EVLM_ATTN_NO_STREAM"))) return 0;
  }
  if (f.Tq) {
    if (!(f.Tk && f.tlse && a->lse && a->kd_loss && !a->kd_teacher && !a->P && !a->mask && !a->kv_index && !a->causal &&
          a->Lq == a->Lk && a->Lk > 224 && a->Lk <= 928 && a->kd_tld % 8 == 0))
      return evlm_set_error("evlm_attention_fwd: kd_tq (the teacher's map rebuilt in the kernel) needs kd_tk, kd_tlse, lse and "
                            "kd_loss, self-attention on 225..928 keys without a mask, and neither P nor kd_teacher");
    f.kd_coef = a->kd_weight / ((float)a->B * a->H * a->Lq * a->Lk);
  }
  static const bool no_head_skip = getenv("EVLM_ATTN_NO_HEAD_SKIP") != nullptr;      // (A/B switch)
  f.skip_dead = no_head_skip ? 0 : 1;
  if (a->kd_rowdot && !(a->lse && (a->kd_teacher || a->kd_tq)))
    return evlm_set_error("evlm_attention_fwd: kd_rowdot needs lse and kd_teacher");
  if (a->lse && !evlm_attention_lse_supported(a->dtype, a->dh, a->Lk, 0.f))
    return evlm_set_error("evlm_attention_fwd: the lse form serves Lk <= 224 and 417..928 (got %d)", a->Lk);
  if (a->kd_teacher && !a->kd_loss) return evlm_set_error("evlm_attention_fwd: kd_teacher without kd_loss");
  if (a->Lk > 64 && a->Lk <= 224 && launch_fwd_grouped<14>(f, a->Bkv, stream)) {}
  else if (launch_fwd_stream(f, stream)) {}
  else if (f.Tq) return evlm_set_error("evlm_attention_fwd: kd_tq is served by the streaming kernels only");
  else if (a->Lk <= 32) launch_fwd<2>(f, stream);
  else if (a->Lk <= 64) launch_fwd<4>(f, stream);
  else if (a->Lk <= 224) launch_fwd<14>(f, stream);
  else if (a->Lk <= 416) launch_fwd<26>(f, stream);
  else if (a->Lk <= 608) launch_fwd<38>(f, stream);
  else launch_fwd<58>(f, stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return evlm_set_error("evlm_attention_fwd(mfma): %s", hipGetErrorString(e));
  *handled = 1;
  return 0;
}
