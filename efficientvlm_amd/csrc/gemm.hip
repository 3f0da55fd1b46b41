// GEMM with fused epilogue for gfx950 (MI355X).
//
//   C[i,j] = epi( alpha * sum_k P(i,k) * Q(j,k) )
//
// bf16 path : 128x128x64 workgroup tile, 4 waves (2x2) of 64x64, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//             Operands are staged global -> VGPR -> LDS (double buffered, one barrier per K tile).
//             K-contiguous operands ([rows][K]) live in LDS as 128-byte rows with a 16-byte-chunk XOR swizzle
//             and are read with ds_read_b128; reduction-major operands ([K][rows], the backward products) live
//             as 256-byte rows (XOR swizzle) and are read TRANSPOSED with ds_read_b64_tr_b16, so dX = dY*W and
//             dW = dY^T*X need no transposed copies in HBM.
//             The MFMA takes Q as its A operand and P as its B operand, so each lane ends up holding FOUR
//             CONSECUTIVE j of one output row i: bias/gate loads and the C store are 8/16-byte vectors.
// f32 path  : exact fp32 (v_mfma_f32_16x16x4_f32), 64x64x16 tile; the parity path.
//
// Both paths share one epilogue (bias, pre/post-activation gate, GELU/quick-GELU, activation backward,
// residual add, optional pre-activation store).
#include <stdlib.h>
#include "common.h"

#include "gemm_common.h"

// 256x256 ping-pong kernel (gemm_pp256.hip)
int evlm_gemm_pp256_launch(GemmP& g, int pt, int qt, hipStream_t stream);
int evlm_gemm_pp256_splits(const GemmP& g);
bool evlm_gemm_pp256_eligible(const GemmP& g, int pt, int qt);
bool evlm_gemm_pp256_streamk(const GemmP& g, int pt);
bool evlm_gemm_w4_eligible(const GemmP& g, int pt, int qt);
int evlm_gemm_w4_launch(GemmP& g, int qt, hipStream_t stream);
bool evlm_gemm_pp128_eligible(const GemmP& g, int pt, int qt);
int evlm_gemm_pp128_launch(GemmP& g, int qt, hipStream_t stream);
bool evlm_gemm_pp192_eligible(const GemmP& g, int pt, int qt);
int evlm_gemm_pp192_launch(GemmP& g, int qt, hipStream_t stream);


// =============================================================================================
// bf16 kernel
// =============================================================================================
#define BT 128   // tile rows (i) and cols (j)
#define BK 64    // k per LDS tile
#define TILE_BYTES (BT * BK * 2)   // 16 KiB per operand per stage

// global -> registers: 4 x 16-byte chunks per thread per operand
template <bool TR, bool FULL>
__device__ __forceinline__ void stage_load(const bf16* base, int ld, int rows, int K, int row0, int k0, int tid,
                                           uint4 r[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = tid + 256 * c;
    const bf16* p;
    bool ok;
    if (!TR) {  // [rows][K]: tile = 128 rows x 8 chunks (64 k)
      const int row = id >> 3, kc = id & 7;
      ok = (row0 + row < rows) && (k0 + kc * 8 < K);
      p = base + (size_t)(row0 + row) * ld + k0 + kc * 8;
    } else {    // [K][rows]: tile = 64 k-rows x 16 chunks (128 rows)
      const int kr = id >> 4, cc = id & 15;
      ok = (k0 + kr < K) && (row0 + cc * 8 < rows);
      p = base + (size_t)(k0 + kr) * ld + row0 + cc * 8;
    }
    r[c] = (FULL || ok) ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
  }
}

template <bool TR>
__device__ __forceinline__ void stage_store(char* sm, int tid, const uint4 r[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = tid + 256 * c;
    int off;
    if (!TR) {
      const int row = id >> 3, kc = id & 7;
      off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
    } else {
      const int kr = id >> 4, cc = id & 15;
      off = kr * 256 + ((cc ^ (((kr & 3) << 2) | ((kr >> 2) & 3))) << 4);
    }
    *reinterpret_cast<uint4*>(sm + off) = r[c];
  }
}

// fragment of 16 rows (rt-th 16-row group of the 128-row tile) x 32 k (k-step ks) for lane `lane`:
// 8 bf16 = row (lane&15), k = 8*(lane>>4) + 0..7    (MFMA 16x16x32 A/B operand map)
template <bool TR>
__device__ __forceinline__ bf16x8 frag_read(const char* sm, int rt, int ks, int lane) {
  if (!TR) {
    const int row = rt * 16 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(sm + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
    bf16x8 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kr = ks * 32 + g * 8 + h * 4 + q;
      const int ch = rt * 2 + (p >> 1);
      const int off = kr * 256 + ((ch ^ (((kr & 3) << 2) | ((kr >> 2) & 3))) << 4) + ((p & 1) << 3);
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (bf16x4 __attribute__((address_space(3)))*)(sm + off));
      out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
    }
    return out;
  }
}

// ---------------------------------------------------------------------------------------------
// fast path (K a multiple of 64): every tile, interior or edge, runs the same unpredicated main loop.  Rows / columns
// beyond the matrix are CLAMPED to the last valid row / 16-byte chunk (they load valid memory and compute garbage
// that is never stored), so there is no zero-fill branch and the LDS-DMA can be used everywhere.
//
// MT = 16x16 accumulator tiles per wave per dimension: MT=4 -> 128x128 workgroup tile (large GEMMs),
//                                                      MT=2 ->  64x64  workgroup tile (text-side GEMMs with < ~400
//                                                               big tiles, which would leave most of the 256 CUs idle).
// Split-K (f32 outputs only: weight gradients, whose output is small and whose K is the token count): grid.y slices K and
// the partial tiles are combined with f32 atomics into the zero-initialised output.
// ---------------------------------------------------------------------------------------------
template <int MT> struct TileCfg {
  static constexpr int BTm = 32 * MT;                 // tile rows / cols
  static constexpr int TB = BTm * BK * 2;             // bytes per operand tile per stage
  static constexpr int NPW = MT;                      // 1-KiB pieces staged per wave per operand
  static constexpr int TRCH = 4 * MT;                 // 16-byte chunks per row of a reduction-major tile ([64][BTm])
};
// swizzle of the 16-byte chunks of a reduction-major tile row (conflict-free ds_read_b64_tr_b16, see frag_read)
template <int MT> __device__ __forceinline__ int tr_swz(int kr) {
  return MT == 4 ? (((kr & 3) << 2) | ((kr >> 2) & 3)) : ((((kr >> 1) & 1) | (((kr >> 3) & 1) << 1)) << 1);
}
// per-lane source offsets (elements from the operand base; 32-bit: the host routes operands of >= 2^31 elements to the
// generic kernel) of the MT x 16-byte pieces this thread stages for one operand
template <bool TR, int MT>
__device__ __forceinline__ void src_offs(int ld, int rows, int row0, int tid, int (&src)[MT]) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    const int id = (c * 4 + wave) * 64 + lane;      // LDS slot id (16 bytes each), linear per wave instruction
    if (!TR) {
      const int row = id >> 3, cp = id & 7;
      src[c] = min(row0 + row, rows - 1) * ld + ((cp ^ ((row >> 1) & 7)) << 3);
    } else {
      constexpr int NC = TileCfg<MT>::TRCH;
      const int kr = id / NC, cp = id % NC;
      const int col = row0 + ((cp ^ tr_swz<MT>(kr)) << 3);
      src[c] = kr * ld + min(col, ((rows + 7) & ~7) - 8);
    }
  }
}
// LDS-DMA: global -> LDS directly (global_load_lds_dwordx4).  One wave instruction writes 1 KiB of LDS linearly
// (wave-uniform base + lane*16); the XOR swizzle therefore lives in the per-lane SOURCE chunk (src_offs), the same
// involution the fragment reads apply.
template <int MT>
__device__ __forceinline__ void stage_glds(const bf16* base, const int (&src)[MT], int koff, char* sm, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < MT; ++c)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (src[c] + koff)),
                                     (__attribute__((address_space(3))) void*)(sm + (c * 4 + wave) * 1024), 16, 0, 0);
}
template <int MT> struct StageR { uint4 r[MT]; };   // by-value register bundle
template <int MT>
__device__ __forceinline__ StageR<MT> stage_regs_load(const bf16* base, const int (&src)[MT], int koff) {
  StageR<MT> r;
#pragma unroll
  for (int c = 0; c < MT; ++c) r.r[c] = *reinterpret_cast<const uint4*>(base + (src[c] + koff));
  return r;
}
template <int MT>
__device__ __forceinline__ void stage_regs_store(char* sm, int tid, const StageR<MT>& r) {
#pragma unroll
  for (int c = 0; c < MT; ++c) *reinterpret_cast<uint4*>(sm + c * 4096 + tid * 16) = r.r[c];   // slot id = c*256 + tid
}

// fragment reads for the MT-parametrised tiles
template <bool TR, int MT>
__device__ __forceinline__ bf16x8 frag_read_t(const char* sm, int rt, int ks, int lane) {
  if (!TR) {
    const int row = rt * 16 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(sm + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    constexpr int RB = TileCfg<MT>::TRCH * 16;       // bytes per reduction row
    const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
    bf16x8 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kr = ks * 32 + g * 8 + h * 4 + q;
      const int ch = rt * 2 + (p >> 1);
      const int off = kr * RB + ((ch ^ tr_swz<MT>(kr)) << 4) + ((p & 1) << 3);
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sm + off));
      out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
    }
    return out;
  }
}

// stream a swizzled [BTm][BTm] bf16 LDS tile to global memory as full rows (16 bytes per lane, consecutive lanes on
// consecutive addresses)
template <bool FULL, int MT>
__device__ __forceinline__ void copy_tile_out(const char* sT, bf16* dst, int ld, int i0, int j0, int I, int J, int tid) {
  constexpr int NC = 4 * MT, RPP = 256 / NC, NPASS = 32 * MT / RPP;   // chunks per row, rows per pass, passes
  const int ch = tid % NC;
  const int jc = j0 + ch * 8;
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int r = tid / NC + RPP * k;
    if (!FULL && (i0 + r >= I || jc >= J)) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(sT + r * (NC * 16) + ((ch ^ (r & (NC - 1))) << 4));
    bf16* d = dst + (size_t)(i0 + r) * ld + jc;
    if (FULL || jc + 8 <= J) *reinterpret_cast<uint4*>(d) = v;
    else {
      const bf16x8 e = *reinterpret_cast<const bf16x8*>(&v);
#pragma unroll
      for (int x = 0; x < 8; ++x) if (x < J - jc) d[x] = e[x];
    }
  }
}

template <bool PT, bool QT, bool FULL, int MT>
__device__ __forceinline__ void gemm_bf16_fast(const GemmP& g, char* smem, int i0, int j0) {
  using TC = TileCfg<MT>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  // LDS-DMA staging pays for NT and NN; the dW product (both operands reduction-major, long K) measured 20 % faster
  // through VGPRs (tools/gemm_bench.py)
  constexpr bool DMA = !(PT && QT);
  const bf16* Pb = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qb = reinterpret_cast<const bf16*>(g.Q);
  int sp_[MT], sq_[MT];
  src_offs<PT, MT>(g.ldp, g.I, i0, tid, sp_);
  src_offs<QT, MT>(g.ldq, g.J, j0, tid, sq_);
  const int kp = PT ? BK * g.ldp : BK, kq = QT ? BK * g.ldq : BK;   // element advance per K tile

  f32x4 acc[MT][MT];   // [a: j tile][b: i tile] : D rows = j (Q side), D cols = i (P side)
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // bias gradient riding on the dW GEMM: sum_k P(i,k) = (ones x P^T) -> one extra MFMA per i tile for the waves that own
  // the first j tile of the first tile column (every row of the result tile is the same column sum)
#ifdef EVLM_GEMM_STAMP
  const bool do_psum = false;
#else
  const bool do_psum = (g.psum != nullptr) && (j0 == 0) && (wj == 0);
#endif
  f32x4 ps[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) ps[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

#ifdef EVLM_GEMM_STAMP
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st2 = 0, st3 = 0;
#endif
  StageR<MT> rp, rq;
  const int nt_all = g.K / BK;
  const int t0 = blockIdx.y * g.kt_per_split, t1 = min(nt_all, t0 + g.kt_per_split);   // this split's K tiles
  if (DMA) {
    stage_glds<MT>(Pb, sp_, t0 * kp, smem, tid);
    stage_glds<MT>(Qb, sq_, t0 * kq, smem + TC::TB, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    rp = stage_regs_load<MT>(Pb, sp_, t0 * kp);
    rq = stage_regs_load<MT>(Qb, sq_, t0 * kq);
    stage_regs_store<MT>(smem, tid, rp);
    stage_regs_store<MT>(smem + TC::TB, tid, rq);
  }
  __syncthreads();
#ifdef EVLM_GEMM_STAMP
  st1 = __builtin_amdgcn_s_memtime();
#endif

  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    const char* sp = smem + cur * 2 * TC::TB;
    const char* sq = sp + TC::TB;
    char* nb = smem + (cur ^ 1) * 2 * TC::TB;
    if (t + 1 < t1) {   // next tile's loads are issued before the MFMA block: their latency hides under it
      if (DMA) { stage_glds<MT>(Pb, sp_, (t + 1) * kp, nb, tid); stage_glds<MT>(Qb, sq_, (t + 1) * kq, nb + TC::TB, tid); }
      else { rp = stage_regs_load<MT>(Pb, sp_, (t + 1) * kp); rq = stage_regs_load<MT>(Qb, sq_, (t + 1) * kq); }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fp[MT], fq[MT];
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        fp[a] = frag_read_t<PT, MT>(sp, wi * MT + a, ks, lane);
        fq[a] = frag_read_t<QT, MT>(sq, wj * MT + a, ks, lane);
      }
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[a], fp[b], acc[a][b], 0, 0, 0);
      if (do_psum) {
#pragma unroll
        for (int b = 0; b < MT; ++b) ps[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fp[b], ps[b], 0, 0, 0);
      }
    }
    if (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA of tile t+1 has landed
    } else if (t + 1 < t1) {
      stage_regs_store<MT>(nb, tid, rp);
      stage_regs_store<MT>(nb + TC::TB, tid, rq);
    }
    __syncthreads();
  }
#ifdef EVLM_DEBUG_NO_EPILOGUE
  {
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) keep += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (keep == 123.456f) reinterpret_cast<float*>(g.C)[0] = keep;
    return;
  }
#endif
#ifdef EVLM_GEMM_STAMP
  st2 = __builtin_amdgcn_s_memtime();
#endif
  const int ib = i0 + wi * 16 * MT, jb = j0 + wj * 16 * MT;
#ifndef EVLM_GEMM_STAMP
  if (do_psum && lane < 16) {       // D rows are identical: lanes 0..15 (row group 0) hold the 16 columns of each i tile
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      const int i = ib + b * 16 + lane;
      if (FULL || i < g.I) atomicAdd(g.psum + i, ps[b][0]);
    }
  }
#endif
  if (g.bare_f32) {
    // weight gradients: the f32 tile goes through LDS so that each wave instruction covers 256 contiguous bytes
    // (plain 16-byte stores, or f32 atomics for split-K partials: scattered dword atomics run ~17x slower)
    constexpr int BTm = TC::BTm, NCF = BTm / 4;          // 16-byte chunks per f32 tile row
    const int il = lane & 15, jl = (lane >> 4) * 4;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int r = (ib - i0) + b * 16 + il, c = (jb - j0) + a * 16 + jl;
        f32x4 v = acc[a][b] * g.alpha;
        *reinterpret_cast<f32x4*>(smem + r * (BTm * 4) + ((((c >> 2) ^ r) & (NCF - 1)) << 4)) = v;
      }
    __syncthreads();
    float* Cf = reinterpret_cast<float*>(g.C);
    if (gridDim.y > 1 || g.accumulate) {
#pragma unroll 4
      for (int k = 0; k < BTm * BTm / 256; ++k) {
        const int id = k * 256 + tid, r = id / BTm, c = id % BTm;
        if (FULL || (i0 + r < g.I && j0 + c < g.J)) {
          const float v = *reinterpret_cast<const float*>(smem + r * (BTm * 4) + ((((c >> 2) ^ r) & (NCF - 1)) << 4) + ((c & 3) << 2));
          atomicAdd(Cf + (size_t)(i0 + r) * g.ldc + j0 + c, v);
        }
      }
    } else {
#pragma unroll 4
      for (int k = 0; k < BTm * NCF / 256; ++k) {
        const int id = k * 256 + tid, r = id / NCF, ch = id % NCF, jc = j0 + ch * 4;
        if (!FULL && (i0 + r >= g.I || jc >= g.J)) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * (BTm * 4) + (((ch ^ r) & (NCF - 1)) << 4));
        float* d = Cf + (size_t)(i0 + r) * g.ldc + jc;
        if (FULL || jc + 4 <= g.J) *reinterpret_cast<f32x4*>(d) = v;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (jc + e < g.J) d[e] = v[e];
        }
      }
    }
  } else if (!g.c_f32) {
    // all staging buffers are dead after the last barrier: reuse them as the output tiles
    char* sC = smem;
    char* sH = smem + 2 * TC::TB;
    tile_epilogue<bf16, MT, MT, FULL, true, MT>(g, acc, ib, jb, lane, sC, sH, i0, j0);
    __syncthreads();
    copy_tile_out<FULL, MT>(sC, reinterpret_cast<bf16*>(g.C), g.ldc, i0, j0, g.I, g.J, tid);
    if (g.preact) copy_tile_out<FULL, MT>(sH, reinterpret_cast<bf16*>(g.preact), g.ldx, i0, j0, g.I, g.J, tid);
  } else {
    tile_epilogue<bf16, MT, MT, FULL>(g, acc, ib, jb, lane);
  }
#ifdef EVLM_GEMM_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st3 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && g.psum) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(g.psum) + (size_t)blockIdx.x * 4;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
  }
#endif
}

// generic path (K not a multiple of 64: only the vocabulary-sized reduction of the MLM decoder backward)
template <bool PT, bool QT>
__device__ __forceinline__ void gemm_bf16_generic(const GemmP& g, char* smem, int i0, int j0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  const bf16* P = reinterpret_cast<const bf16*>(g.P);
  const bf16* Q = reinterpret_cast<const bf16*>(g.Q);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 rp[4], rq[4];
  const int nt = (g.K + BK - 1) / BK;
  stage_load<PT, false>(P, g.ldp, g.I, g.K, i0, 0, tid, rp);
  stage_load<QT, false>(Q, g.ldq, g.J, g.K, j0, 0, tid, rq);
  stage_store<PT>(smem, tid, rp);
  stage_store<QT>(smem + TILE_BYTES, tid, rq);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char* sp = smem + (t & 1) * 2 * TILE_BYTES;
    const char* sq = sp + TILE_BYTES;
    if (t + 1 < nt) {
      stage_load<PT, false>(P, g.ldp, g.I, g.K, i0, (t + 1) * BK, tid, rp);
      stage_load<QT, false>(Q, g.ldq, g.J, g.K, j0, (t + 1) * BK, tid, rq);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fp[4], fq[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        fp[a] = frag_read<PT>(sp, wi * 4 + a, ks, lane);
        fq[a] = frag_read<QT>(sq, wj * 4 + a, ks, lane);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[a], fp[b], acc[a][b], 0, 0, 0);
    }
    if (t + 1 < nt) {
      char* np_ = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      stage_store<PT>(np_, tid, rp);
      stage_store<QT>(np_ + TILE_BYTES, tid, rq);
    }
    __syncthreads();
  }
  tile_epilogue<bf16, 4, 4, false>(g, acc, i0 + wi * 64, j0 + wj * 64, lane);
}

template <bool PT, bool QT, int MT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][P tile | Q tile]
  constexpr int BTm = 32 * MT;
  int ti, tj;
  tile_coords(g, ti, tj);
  const int i0 = ti * BTm, j0 = tj * BTm;
  // wave-uniform: interior tiles skip every bounds check in the epilogue
  const bool full = (i0 + BTm <= g.I) && (j0 + BTm <= g.J);
  if (full) gemm_bf16_fast<PT, QT, true, MT>(g, smem, i0, j0);
  else gemm_bf16_fast<PT, QT, false, MT>(g, smem, i0, j0);
}
// ---------------------------------------------------------------------------------------------
// persistent, cross-tile pipelined variant for bf16-output NT / NN GEMMs (the forward and dX products, K = 768..3072:
// only 12..48 K tiles per output tile, so the per-tile fixed cost - first-stage DMA latency, epilogue, workgroup
// relaunch - is what separates them from the steady-state MFMA rate).  One workgroup per resident slot walks the tile
// list; the first K tile of the NEXT output tile is DMA'd into the free half of LDS while the current tile's
// epilogue runs out of the other half.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_ij(const GemmP& g, int t_lin, int ntiles, int& ti, int& tj) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = t_lin & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t_lin >> 3);
  ti = t / g.tiles_j;
  tj = t - ti * g.tiles_j;
}

template <bool PT, bool QT, int MT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_persist_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // region 0 | region 1, each [P tile | Q tile]
  using TC = TileCfg<MT>;
  constexpr int BTm = TC::BTm, RB = 2 * TC::TB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  const bf16* Pb = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qb = reinterpret_cast<const bf16*>(g.Q);
  const int kp = PT ? BK * g.ldp : BK, kq = QT ? BK * g.ldq : BK;
  const int nt = g.K / BK, ntiles = g.tiles_i * g.tiles_j;
  // the epilogue needs one [BTm][BTm] bf16 tile per output (C, optionally the pre-activation): with MT = 4 a tile is a
  // whole region, so a pre-activation output leaves no room for the prefetch
  const bool overlap = (MT == 2) || (g.preact == nullptr);

  int tile = blockIdx.x, ti, tj;
  tile_ij(g, tile, ntiles, ti, tj);
  int i0 = ti * BTm, j0 = tj * BTm;
  int sp_[MT], sq_[MT];
  src_offs<PT, MT>(g.ldp, g.I, i0, tid, sp_);
  src_offs<QT, MT>(g.ldq, g.J, j0, tid, sq_);
  stage_glds<MT>(Pb, sp_, 0, smem, tid);
  stage_glds<MT>(Qb, sq_, 0, smem + TC::TB, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int first = 0;                     // region that holds K tile 0 of the current output tile

  while (true) {
    f32x4 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nt; ++t) {
      const int cur = first ^ (t & 1);
      const char* sp = smem + cur * RB;
      const char* sq = sp + TC::TB;
      char* nb = smem + (cur ^ 1) * RB;
      if (t + 1 < nt) { stage_glds<MT>(Pb, sp_, (t + 1) * kp, nb, tid); stage_glds<MT>(Qb, sq_, (t + 1) * kq, nb + TC::TB, tid); }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fp[MT], fq[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          fp[a] = frag_read_t<PT, MT>(sp, wi * MT + a, ks, lane);
          fq[a] = frag_read_t<QT, MT>(sq, wj * MT + a, ks, lane);
        }
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[a], fp[b], acc[a][b], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // every LDS region is free here.  Prefetch the next output tile's first K tile into region 1 ...
    const int next = tile + gridDim.x;
    const bool has_next = next < ntiles;
    int ni0 = 0, nj0 = 0;
    if (has_next) {
      tile_ij(g, next, ntiles, ti, tj);
      ni0 = ti * BTm; nj0 = tj * BTm;
      src_offs<PT, MT>(g.ldp, g.I, ni0, tid, sp_);
      src_offs<QT, MT>(g.ldq, g.J, nj0, tid, sq_);
      if (overlap) { stage_glds<MT>(Pb, sp_, 0, smem + RB, tid); stage_glds<MT>(Qb, sq_, 0, smem + RB + TC::TB, tid); }
    }
    // ... while this tile's epilogue runs out of region 0 (and region 1 too when there is no overlap)
    const int ib = i0 + wi * 16 * MT, jb = j0 + wj * 16 * MT;
    char* sC = smem;
    char* sH = (MT == 2) ? smem + TC::TB : smem + RB;
    const bool full = (i0 + BTm <= g.I) && (j0 + BTm <= g.J);
    if (full) tile_epilogue<bf16, MT, MT, true, true, MT>(g, acc, ib, jb, lane, sC, sH, i0, j0);
    else tile_epilogue<bf16, MT, MT, false, true, MT>(g, acc, ib, jb, lane, sC, sH, i0, j0);
    __syncthreads();
    if (full) {
      copy_tile_out<true, MT>(sC, reinterpret_cast<bf16*>(g.C), g.ldc, i0, j0, g.I, g.J, tid);
      if (g.preact) copy_tile_out<true, MT>(sH, reinterpret_cast<bf16*>(g.preact), g.ldx, i0, j0, g.I, g.J, tid);
    } else {
      copy_tile_out<false, MT>(sC, reinterpret_cast<bf16*>(g.C), g.ldc, i0, j0, g.I, g.J, tid);
      if (g.preact) copy_tile_out<false, MT>(sH, reinterpret_cast<bf16*>(g.preact), g.ldx, i0, j0, g.I, g.J, tid);
    }
    if (!has_next) break;
    if (!overlap) {
      __syncthreads();                 // the output tiles have been read out of LDS
      stage_glds<MT>(Pb, sp_, 0, smem + RB, tid);
      stage_glds<MT>(Qb, sq_, 0, smem + RB + TC::TB, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (full) {
      // wait for the prefetch DMA only: vmcnt retires in issue order and the only younger operations still in flight
      // are this tile's output stores (MT*MT/2 per thread and output), which may drain under the next main loop
      if (MT == 4) { if (g.preact) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else { if (g.preact) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    tile = next; i0 = ni0; j0 = nj0; first = 1;
  }
}

// ---------------------------------------------------------------------------------------------
// 128x128 tile, K step 32, THREE-stage LDS ring (48 KiB -> three workgroups per CU) with the LDS-DMA of K step t+2 left
// in flight across the barrier (counted vmcnt).  In-kernel stamps of the two-stage kernel above showed ~1700 cycles per
// 64-deep step against 1024 of MFMA issue: one step is shorter than the HBM/L2 round trip, so `vmcnt(0)` at every
// barrier exposed load latency.  Two steps of prefetch distance + 12 waves per CU cover it.
// ---------------------------------------------------------------------------------------------
#define BK3 32
#define TB3 (128 * BK3 * 2)          // 8 KiB per operand per stage

// per-lane source offsets of the 2 x 16-byte pieces per operand this thread stages per K step
template <bool TR>
__device__ __forceinline__ void src_offs3(int ld, int rows, int row0, int tid, int (&src)[2]) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = (c * 4 + wave) * 64 + lane;
    if (!TR) {     // [128 rows][32 k]: 64-byte rows, 4 chunks, swizzle (row>>2)&3
      const int row = id >> 2, cp = id & 3;
      src[c] = min(row0 + row, rows - 1) * ld + ((cp ^ ((row >> 2) & 3)) << 3);
    } else {       // [32 k][128 cols]: 256-byte rows, 16 chunks
      const int kr = id >> 4, cp = id & 15;
      const int col = row0 + ((cp ^ (((kr & 3) << 2) | ((kr >> 2) & 3))) << 3);
      src[c] = kr * ld + min(col, ((rows + 7) & ~7) - 8);
    }
  }
}
__device__ __forceinline__ void stage_glds3(const bf16* base, const int (&src)[2], int koff, char* sm, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (src[c] + koff)),
                                     (__attribute__((address_space(3))) void*)(sm + (c * 4 + wave) * 1024), 16, 0, 0);
}
template <bool TR>
__device__ __forceinline__ bf16x8 frag_read3(const char* sm, int rt, int lane) {
  if (!TR) {
    const int row = rt * 16 + (lane & 15), c = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(sm + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
  } else {
    const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
    bf16x8 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kr = g * 8 + h * 4 + q;
      const int ch = rt * 2 + (p >> 1);
      const int off = kr * 256 + ((ch ^ (((kr & 3) << 2) | ((kr >> 2) & 3))) << 4) + ((p & 1) << 3);
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sm + off));
      out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
    }
    return out;
  }
}

template <bool PT, bool QT>
__global__ __launch_bounds__(256, 3) void gemm_bf16_s3_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 3 stages x [P tile | Q tile] = 48 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  int ti, tj;
  tile_coords(g, ti, tj);
  const int i0 = ti * 128, j0 = tj * 128;
  const bf16* Pb = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qb = reinterpret_cast<const bf16*>(g.Q);
  int sp_[2], sq_[2];
  src_offs3<PT>(g.ldp, g.I, i0, tid, sp_);
  src_offs3<QT>(g.ldq, g.J, j0, tid, sq_);
  const int kp = PT ? BK3 * g.ldp : BK3, kq = QT ? BK3 * g.ldq : BK3;
  const int nt = g.K / BK3;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // prologue: K steps 0 and 1 in flight, wait for step 0 only
  stage_glds3(Pb, sp_, 0, smem, tid);
  stage_glds3(Qb, sq_, 0, smem + TB3, tid);
  if (nt > 1) {
    stage_glds3(Pb, sp_, kp, smem + 2 * TB3, tid);
    stage_glds3(Qb, sq_, kq, smem + 3 * TB3, tid);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sp = smem + cur * 2 * TB3;
    const char* sq = sp + TB3;
    int nxt2 = cur + 2; if (nxt2 >= 3) nxt2 -= 3;
    if (t + 2 < nt) {      // stage (t+2)%3 was last read in step t-1: every wave passed that step's closing barrier
      stage_glds3(Pb, sp_, (t + 2) * kp, smem + nxt2 * 2 * TB3, tid);
      stage_glds3(Qb, sq_, (t + 2) * kq, smem + nxt2 * 2 * TB3 + TB3, tid);
    }
    bf16x8 fp[4], fq[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      fp[a] = frag_read3<PT>(sp, wi * 4 + a, lane);
      fq[a] = frag_read3<QT>(sq, wj * 4 + a, lane);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[a], fp[b], acc[a][b], 0, 0, 0);
    // step t+1 must have landed before anyone reads it; step t+2 (the 4 youngest DMAs) stays in flight
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = cur + 1; if (cur >= 3) cur = 0;
  }
  // epilogue through LDS (32 KiB of the 48): same as the two-stage kernel
  const int ib = i0 + wi * 64, jb = j0 + wj * 64;
  const bool full = (i0 + 128 <= g.I) && (j0 + 128 <= g.J);
  char* sC = smem;
  char* sH = nullptr;     // a pre-activation output does not fit beside C: the host does not route such GEMMs here
  if (full) tile_epilogue<bf16, 4, 4, true, true, 4>(g, acc, ib, jb, lane, sC, sH, i0, j0);
  else tile_epilogue<bf16, 4, 4, false, true, 4>(g, acc, ib, jb, lane, sC, sH, i0, j0);
  __syncthreads();
  if (full) copy_tile_out<true, 4>(sC, reinterpret_cast<bf16*>(g.C), g.ldc, i0, j0, g.I, g.J, tid);
  else copy_tile_out<false, 4>(sC, reinterpret_cast<bf16*>(g.C), g.ldc, i0, j0, g.I, g.J, tid);
}

template <bool PT, bool QT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_generic_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int ti, tj;
  tile_coords(g, ti, tj);
  gemm_bf16_generic<PT, QT>(g, smem, ti * BT, tj * BT);
}

// =============================================================================================
// exact-fp32 kernel (parity path): 64x64x16 tile, 4 waves (2x2) of 32x32, v_mfma_f32_16x16x4_f32
// =============================================================================================
#define FT 64
#define FK 16
#define FLD (FT + 4)

template <bool TR>
__device__ __forceinline__ void f32_stage(const float* base, int ld, int rows, int K, int row0, int k0, int tid,
                                          float (*s)[FLD]) {
  if (!TR) {  // [rows][K]: thread loads 4 consecutive k of one row
    const int row = tid >> 2, kc = (tid & 3) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (row0 + row < rows) {
      const float* p = base + (size_t)(row0 + row) * ld + k0 + kc;
      if (k0 + kc + 3 < K) Vec4<float>::load(p, v);
      else for (int e = 0; e < 4; ++e) if (k0 + kc + e < K) v[e] = p[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[kc + e][row] = v[e];
  } else {    // [K][rows]: thread loads 4 consecutive rows of one k
    const int kr = tid >> 4, c4 = (tid & 15) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (k0 + kr < K) {
      const float* p = base + (size_t)(k0 + kr) * ld + row0 + c4;
      if (row0 + c4 + 3 < rows) Vec4<float>::load(p, v);
      else for (int e = 0; e < 4; ++e) if (row0 + c4 + e < rows) v[e] = p[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[kr][c4 + e] = v[e];
  }
}

template <bool PT, bool QT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP g) {
  __shared__ __attribute__((aligned(16))) float sp[FK][FLD];
  __shared__ __attribute__((aligned(16))) float sq[FK][FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  int ti, tj;
  tile_coords(g, ti, tj);
  const int i0 = ti * FT, j0 = tj * FT;
  const float* P = reinterpret_cast<const float*>(g.P);
  const float* Q = reinterpret_cast<const float*>(g.Q);
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < g.K; k0 += FK) {
    f32_stage<PT>(P, g.ldp, g.I, g.K, i0, k0, tid, sp);
    f32_stage<QT>(Q, g.ldq, g.J, g.K, j0, k0, tid, sq);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < FK / 4; ++ks) {
      const int kk = ks * 4 + (lane >> 4);
      float fp[2], fq[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        fp[a] = sp[kk][wi * 32 + a * 16 + (lane & 15)];
        fq[a] = sq[kk][wj * 32 + a * 16 + (lane & 15)];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[a], fp[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
  tile_epilogue<float, 2, 2, false>(g, acc, i0 + wi * 32, j0 + wj * 32, lane);
}

// =============================================================================================
// column sums (bias gradients)
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, int I, int J, int ld, float* out, int rows_per_block) {
  __shared__ float red[8][32 * 9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 column-octets x 8 row lanes
  const int j0 = blockIdx.x * 256 + tx * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(I, r0 + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (j0 < J) {
    for (int i = r0 + ty; i < r1; i += 8) {
      float v[8];
      if (j0 + 8 <= ld) load8<T>(X + (size_t)i * ld + j0, v);
      else for (int e = 0; e < 8; ++e) v[e] = (j0 + e < J) ? to_f(X[(size_t)i * ld + j0 + e]) : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e + tx] = acc[e];   // +tx: skew against bank conflicts
  __syncthreads();
  if (ty == 0 && j0 < J) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
      for (int y = 0; y < 8; ++y) s += red[y][tx * 8 + e + tx];
      if (j0 + e < J) atomicAdd(out + j0 + e, s);
    }
  }
}

// =============================================================================================
// C ABI
// =============================================================================================
// which kernel served the calling thread's last evlm_gemm (profiling aid: bench.py attributes launch times to kernels)
static thread_local const char* g_last_kernel = "";
extern "C" const char* evlm_gemm_last_kernel(void) { return g_last_kernel; }

// Zero fill of a split reduction's f32 output: a KERNEL, not hipMemsetAsync.  A memset node captured into a hipGraph wrote
// a 16-byte pattern whose first word was stale (0x..25be80: the low half of a host address) from its SECOND launch on, on
// ROCm 7.2 - the VQA step's vocabulary-head dX came back as 1e13..inf in every fourth column (round 4,
// profiles/r04_graph_memset.md).
__global__ void zero_f32_kernel(float* __restrict__ c, int64_t n) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  if (((uintptr_t)c & 15) == 0) {
    float4* c4 = reinterpret_cast<float4*>(c);
    for (int64_t i = i0; i < n / 4; i += step) c4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (n / 4) * 4 + i0; i < n; i += step) c[i] = 0.f;
  } else {
    for (int64_t i = i0; i < n; i += step) c[i] = 0.f;
  }
}
static inline int zero_f32(float* c, int64_t n, hipStream_t stream) {
  if (n <= 0) return 0;
  const int blocks = imin(ceil_div(ceil_div(n, 4), 256), 2048);
  hipLaunchKernelGGL(zero_f32_kernel, dim3(blocks), dim3(256), 0, stream, c, n);
  return hipGetLastError() == hipSuccess ? 0 : evlm_set_error("evlm_gemm: zero fill launch failed");
}

extern "C" int evlm_gemm(const evlm_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(a && a->P && a->Q && a->C, "evlm_gemm: null operand");
  EVLM_REQUIRE(a->I > 0 && a->J > 0 && a->K > 0, "evlm_gemm: bad shape I=%d J=%d K=%d", a->I, a->J, a->K);
  EVLM_REQUIRE(!(a->dact && a->act), "evlm_gemm: dact excludes act");
  EVLM_REQUIRE(!(a->dact && a->gate) || a->dgate, "evlm_gemm: dact with a gate needs dgate (the gated activation backward, ABI 8)");
  EVLM_REQUIRE(!a->dgate || (a->dact && a->gate && a->dtype == EVLM_BF16 && !a->residual && !a->bias && !a->preact && !a->c_f32),
               "evlm_gemm: dgate goes with dact + gate on a plain bf16 product");
  EVLM_REQUIRE(!a->dact || a->aux, "evlm_gemm: dact needs aux");
  const int vec = a->dtype == EVLM_BF16 ? 8 : 4;
  EVLM_REQUIRE(a->ldp % vec == 0 && a->ldq % vec == 0, "evlm_gemm: ldp/ldq must be multiples of %d", vec);
  EVLM_REQUIRE(a->ldc % 4 == 0, "evlm_gemm: ldc must be a multiple of 4");
  EVLM_REQUIRE((!a->preact && !a->aux && !a->residual) || a->ldx % 4 == 0, "evlm_gemm: ldx must be a multiple of 4");
  EVLM_REQUIRE(((uintptr_t)a->P | (uintptr_t)a->Q | (uintptr_t)a->C) % 16 == 0, "evlm_gemm: operands must be 16-byte aligned");
  GemmP g;
  g.P = a->P; g.Q = a->Q; g.C = a->C; g.bias = a->bias; g.gate = a->gate;
  g.preact = a->preact; g.aux = a->aux; g.residual = a->residual;
  g.I = a->I; g.J = a->J; g.K = a->K; g.ldp = a->ldp; g.ldq = a->ldq; g.ldc = a->ldc; g.ldx = a->ldx;
  g.c_f32 = (a->dtype == EVLM_F32) ? 1 : a->c_f32;
  g.accumulate = a->accumulate;
  g.psum = a->psum;
  g.sk_ws = a->sk_workspace; g.sk = 0;
  g.dgate = a->dgate;
  g.drop_p = a->dropout_p; g.rng = a->rng_state; g.call = a->call_id;
  EVLM_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "evlm_gemm: dropout_p = %f outside [0, 1)", (double)a->dropout_p);
  EVLM_REQUIRE(a->dropout_p == 0.f || (a->rng_state && a->residual && a->J % 8 == 0 && !a->c_f32 && !a->act && !a->dact && !a->gate &&
                                        !a->preact && !a->accumulate && !a->psum),
               "evlm_gemm: dropout_p needs rng_state, a residual, J a multiple of 8, an output of `dtype` and no act / gate / dact / preact");
  EVLM_REQUIRE(!a->psum || (a->dtype == EVLM_BF16 && a->K % 64 == 0), "evlm_gemm: psum needs bf16 operands and K a multiple of 64");
  EVLM_REQUIRE(!a->accumulate || (a->dtype == EVLM_BF16 && a->c_f32 && !a->bias && !a->gate && !a->preact && !a->aux &&
                                   !a->residual && a->act == EVLM_ACT_NONE && a->K % 64 == 0),
               "evlm_gemm: accumulate needs a bare f32-output bf16 GEMM with K a multiple of 64");
  g.act = a->act; g.gate_pos = a->gate_pos; g.dact = a->dact; g.alpha = a->alpha;
  const int pt = a->p_trans ? 1 : 0, qt = a->q_trans ? 1 : 0;
  if (a->dtype == EVLM_BF16) {
    const bool fast = (g.K % BK == 0) && ((int64_t)(pt ? g.K : g.I) * g.ldp < (1ll << 31)) &&
                      ((int64_t)(qt ? g.K : g.J) * g.ldq < (1ll << 31));
    dim3 block(256);
    // large K-contiguous bf16 products: 256x256 ping-pong kernel, one persistent workgroup per CU (gemm_pp256.hip), when
    // its tiles fill at least ~30 % of the CU rounds: alone on the chip the crossover against the 128x128 kernels is near
    // 55 % (tools/gemm_pp256.py), but the step keeps two streams busy (student + pipelined teacher), so CUs a launch
    // leaves free are taken by the other stream's kernels and the per-tile efficiency decides (measured: +1.3 % step)
    static const int pp_pct = getenv("EVLM_PP256_PCT") ? atoi(getenv("EVLM_PP256_PCT")) : 30;   // tuning aid; > 100 disables
#ifdef EVLM_EXPERIMENTAL_W4
    // (build option, `make EXPERIMENTAL=1`: the one-wave-per-SIMD 256 x 256 kernel of round 4 - gemm_w4.hip, then opt-in through
    // EVLM_W4=1.  Measured 8-35 % slower than the ping-pong flavours on every shape of the step (profiles/r04_w4_kloop.md),
    // so it is NOT part of the default library: no routing rule would pick it.)
    if (evlm_gemm_w4_eligible(g, pt, qt)) {
      const int w4rows = evlm_gemm_w4_launch(g, qt, stream);
      if (w4rows < 0) return -1;
      g_last_kernel = w4rows == 192 ? (qt ? "gemm_bf16_w4_kernel<true,6>" : "gemm_bf16_w4_kernel<false,6>")
                                    : (qt ? "gemm_bf16_w4_kernel<true,8>" : "gemm_bf16_w4_kernel<false,8>");
      EVLM_LAUNCH_CHECK("evlm_gemm");
      return 0;
    }
#endif
    if (evlm_gemm_pp128_eligible(g, pt, qt) && !evlm_gemm_pp256_streamk(g, pt)) {   // 128 x 256 tiles: thinly filled launches
      if (evlm_gemm_pp128_launch(g, qt, stream)) return -1;
      g_last_kernel = qt ? "gemm_bf16_pp128_kernel<true>" : "gemm_bf16_pp128_kernel<false>";
      EVLM_LAUNCH_CHECK("evlm_gemm");
      return 0;
    }
    if (evlm_gemm_pp192_eligible(g, pt, qt) && !evlm_gemm_pp256_streamk(g, pt)) {   // 192 x 256 tiles: one round filled 50-80 %
      if (evlm_gemm_pp192_launch(g, qt, stream)) return -1;
      g_last_kernel = g.dgate ? "gemm_bf16_pp192_kernel<gated dact>" : (qt ? "gemm_bf16_pp192_kernel<true>" : "gemm_bf16_pp192_kernel<false>");
      EVLM_LAUNCH_CHECK("evlm_gemm");
      return 0;
    }
    if (evlm_gemm_pp256_eligible(g, pt, qt)) {
      const int items = ceil_div(g.I, 256) * ceil_div(g.J, 256) * evlm_gemm_pp256_splits(g);
      if (!g.dgate && (items * 100 >= pp_pct * ceil_div(items, 256) * 256 || evlm_gemm_pp256_streamk(g, pt))) {
        if (g.c_f32 && evlm_gemm_pp256_splits(g) > 1 && !g.accumulate) {
          if (zero_f32((float*)g.C, (int64_t)g.I * g.ldc, stream)) return -1;
        }
        if (evlm_gemm_pp256_launch(g, pt, qt, stream)) return -1;
        g_last_kernel = g.sk ? (qt ? "gemm_bf16_pp256_sk_kernel<true>" : "gemm_bf16_pp256_sk_kernel<false>")
                             : g.c_f32 ? "gemm_bf16_pp256_kernel<true,true,1>"
                                       : (qt ? "gemm_bf16_pp256_kernel<false,true,0>" : "gemm_bf16_pp256_kernel<false,false,0>");
        EVLM_LAUNCH_CHECK("evlm_gemm");
        return 0;
      }
    }
    if (g.dgate)      // (only the ping-pong family's epilogue forms the gate gradient: gemm_pp256_epi.h)
      return evlm_set_error("evlm_gemm: the gated activation backward (dgate) needs a product the 256-column ping-pong kernels "
                            "take: K a multiple of 64 and >= 128, J / ldc / ldx multiples of 8, P not transposed");
    if (!fast) {
      g.tiles_i = ceil_div(g.I, BT); g.tiles_j = ceil_div(g.J, BT); g.kt_per_split = 0; g.bare_f32 = 0;
      dim3 grid(g.tiles_i * g.tiles_j);
      const size_t lds = 4 * TILE_BYTES;
#define LAUNCH_GEN(PT_, QT_) hipLaunchKernelGGL((gemm_bf16_generic_kernel<PT_, QT_>), grid, block, lds, stream, g)
      if (!pt && !qt) LAUNCH_GEN(false, false);
      else if (!pt && qt) LAUNCH_GEN(false, true);
      else if (pt && qt) LAUNCH_GEN(true, true);
      else LAUNCH_GEN(true, false);
#undef LAUNCH_GEN
      g_last_kernel = "gemm_bf16_generic_kernel";
    } else {
      // tile choice: 128x128 unless that leaves most of the 256 CUs (x2 resident workgroups) without work
      const int t128 = ceil_div(g.I, 128) * ceil_div(g.J, 128);
      const bool bare = g.c_f32 && !g.bias && !g.gate && !g.preact && !g.aux && !g.residual && g.act == EVLM_ACT_NONE;
      // weight gradients with a long reduction keep the big tile and get their parallelism from split-K instead
      // 128x128 tiles unless their wave quantisation on the 512 resident slots (2 per CU) wastes more than the ~12 % the
      // 64x64 tile loses in arithmetic intensity (tools/gemm_mt.py)
      const float eff4 = (float)t128 / (float)(ceil_div(t128, 512) * 512);
      int mt = (eff4 >= 0.65f || (bare && g.K >= 64 * BK && t128 >= 100)) ? 4 : 2;
      { static const char* force = getenv("EVLM_FORCE_MT"); if (force) mt = atoi(force); }   // tuning aid
      const int bt = 32 * mt;
      g.tiles_i = ceil_div(g.I, bt); g.tiles_j = ceil_div(g.J, bt);
      const int tiles = g.tiles_i * g.tiles_j, nt = g.K / BK;
      // split-K only for bare f32 outputs (weight gradients): combine by f32 atomics into the zeroed output
      int splits = 1;
      g.bare_f32 = bare ? 1 : 0;
      if (bare && tiles < 512 && nt >= 32) {
        splits = imin(imin(ceil_div(768, tiles), nt / 8), 32);
        if (splits < 1) splits = 1;
      }
      g.kt_per_split = ceil_div(nt, splits);
      splits = ceil_div(nt, g.kt_per_split);
      if (splits > 1 && !g.accumulate) {
        if (zero_f32((float*)g.C, (int64_t)g.I * g.ldc, stream)) return -1;
      }
      dim3 grid(tiles, splits);
      const size_t lds = (size_t)4 * bt * BK * 2;
      // bf16-output NT / NN products: persistent cross-tile pipelined kernel, one workgroup per resident slot
      static const bool no_persist = getenv("EVLM_NO_PERSIST") != nullptr;      // tuning aid
      // (measured, tools/gemm_mt.py: +4 % on the 64x64-tile shapes, -10 % on the 128x128 ones, where the hardware
      //  dispatcher's relaunch already overlaps a neighbour's MFMA phase with the epilogue)
      const bool persist = !g.c_f32 && !(pt && qt) && !no_persist && mt == 2;
      if (persist) grid = dim3(imin(tiles, 256 * (mt == 4 ? 2 : 4)), 1);
      static const bool no_s3 = getenv("EVLM_NO_S3") != nullptr;                 // tuning aid
      // short reductions only: measured +5..15 % at K = 768, -20 % at K >= 3072 (tools/gemm_mt.py, gemm_bench.py)
      const bool s3 = mt == 4 && !g.c_f32 && !g.preact && !no_s3 && splits == 1 && g.K <= 1024;
      const size_t lds3 = 6 * TB3;
#define LAUNCH_FAST(PT_, QT_)                                                                                \
  do {                                                                                                       \
    if (s3) hipLaunchKernelGGL((gemm_bf16_s3_kernel<PT_, QT_>), grid, block, lds3, stream, g);               \
    else if (persist && mt == 4) hipLaunchKernelGGL((gemm_bf16_persist_kernel<PT_, QT_, 4>), grid, block, lds, stream, g); \
    else if (persist) hipLaunchKernelGGL((gemm_bf16_persist_kernel<PT_, QT_, 2>), grid, block, lds, stream, g);       \
    else if (mt == 4) hipLaunchKernelGGL((gemm_bf16_kernel<PT_, QT_, 4>), grid, block, lds, stream, g);       \
    else hipLaunchKernelGGL((gemm_bf16_kernel<PT_, QT_, 2>), grid, block, lds, stream, g);                   \
  } while (0)
      if (!pt && !qt) LAUNCH_FAST(false, false);
      else if (!pt && qt) LAUNCH_FAST(false, true);
      else if (pt && qt) LAUNCH_FAST(true, true);
      else LAUNCH_FAST(true, false);
#undef LAUNCH_FAST
      g_last_kernel = s3 ? "gemm_bf16_s3_kernel" : (persist ? "gemm_bf16_persist_kernel<2>"
                                                    : (mt == 4 ? "gemm_bf16_kernel<4>" : "gemm_bf16_kernel<2>"));
    }
  } else if (a->dtype == EVLM_F32) {
    g.tiles_i = ceil_div(g.I, FT); g.tiles_j = ceil_div(g.J, FT); g.kt_per_split = 0; g.bare_f32 = 0;
    dim3 grid(g.tiles_i * g.tiles_j), block(256);
#define LAUNCH_F32(PT_, QT_) hipLaunchKernelGGL((gemm_f32_kernel<PT_, QT_>), grid, block, 0, stream, g)
    if (!pt && !qt) LAUNCH_F32(false, false);
    else if (!pt && qt) LAUNCH_F32(false, true);
    else if (pt && qt) LAUNCH_F32(true, true);
    else LAUNCH_F32(true, false);
#undef LAUNCH_F32
    g_last_kernel = "gemm_f32_kernel";
  } else {
    return evlm_set_error("evlm_gemm: bad dtype %d", a->dtype);
  }
  EVLM_LAUNCH_CHECK("evlm_gemm");
  return 0;
}

extern "C" int evlm_colsum(int dtype, const void* X, int I, int J, int ldx, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(X && out && I > 0 && J > 0, "evlm_colsum: bad args");
  const int rpb = 256;
  dim3 grid(ceil_div(J, 256), ceil_div(I, rpb)), block(256);
  EVLM_DISPATCH_DTYPE(dtype, "evlm_colsum",
    hipLaunchKernelGGL((colsum_kernel<T>), grid, block, 0, stream, (const T*)X, I, J, ldx, out, rpb);)
  EVLM_LAUNCH_CHECK("evlm_colsum");
  return 0;
}
