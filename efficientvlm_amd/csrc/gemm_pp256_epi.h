// Fused epilogues of the 256x256 ping-pong GEMM family (moved out of gemm_pp256.hip in round 4 so that the four-wave kernel
// of gemm_w4.hip runs the SAME epilogue code): a wave converts its accumulator block through a wave-private 4 KiB LDS
// window and stores full 128-byte row segments; bias, activation, activation backward, residual, L0 gates, the optional
// pre-activation output; the bare f32 form of the weight gradients; the XCD-aware tile map.
#pragma once
#include "gemm_pp256_core.h"

// ---------------------------------------------------------------------------------------------
// epilogue.  Each wave converts its own 128 x 64 block through a wave-PRIVATE 4 KiB LDS window (32 rows x 64 columns at
// a time): no workgroup barrier, the staging buffers stay free for the next tile's prologue DMAs, and the global stores
// are full 128-byte row segments.  Per-column vectors and the aux / residual fragments are fetched in ONE batch per
// 64-row half before they are consumed - with a single resident workgroup nothing else would hide a chain of dependent
// global loads.
// ---------------------------------------------------------------------------------------------

template <bool FULL>
__device__ __forceinline__ void pp_epi_cols(const GemmP& g, int jb, int lane, f32x4 (&bz)[4]) {
  const int jl = (lane >> 4) * 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int j = jb + a * 16 + jl;
    const int jc = FULL ? j : min(j, g.J - 4);
    bz[a] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + jc) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

// MODE 1: the pre-activation output (alpha * acc + bias); MODE 2: C
struct PPRows { uint4 r[4]; };     // 32 rows x 128 bytes of aux / residual for one wave: 16 bytes per lane, 8 rows per entry

template <bool FULL>
__device__ __forceinline__ PPRows pp_epi_rows(const GemmP& g, const bf16* xb, int i0, int jb, int lane) {
  PPRows x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + k * 8 + (lane >> 3), j = jb + (lane & 7) * 8;
    const size_t o = (size_t)(FULL ? i : min(i, g.I - 1)) * g.ldx + (FULL ? j : min(j, g.J - 8));
    x.r[k] = *reinterpret_cast<const uint4*>(xb + o);
  }
  return x;
}

// one 32-row chunk (i fragments b0, b0 + 1) of the wave's block through its LDS window.
// ACT / DACT are COMPILE-TIME activation codes (-1: read g.act / g.dact at run time - the L0-gated flavours of the pruning
// fine-tune only).  With the run-time form hipcc keeps a chain of scalar compares and branches around every single element
// (128 per lane and tile): the bias-only epilogue took 7.6 k cycles per tile with or without its global stores
// (in-kernel stamps), most of it branch issue.
// stream-K owner (SKP): up to two partial tiles parked by other workgroups are ADDED HERE, where the accumulators are only
// read - eight 16-byte system-scope loads per chunk and producer (slot layout [wave][half][b][a][lane]).  Summing them into
// the accumulator registers ahead of the epilogue (VALU adds, f32 MFMAs against the identity: both tried) made the
// register allocator spill 250-500 VGPRs, reloaded inside the K loop.
struct PPSk { const char* p0; int np; };     // per-lane address of this half's registers in the FIRST producer's slot; producers
#define PP_SK_NEXT (8 * 262144)               // (they are consecutive workgroups of one XCD: slots 8 apart)
__device__ __forceinline__ void pp_sk_load8(const char* base, f32x4 (&v)[2][4]) {
  asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
               "global_load_dwordx4 %1, %8, off offset:1024 sc0 sc1\n\t"
               "global_load_dwordx4 %2, %8, off offset:2048 sc0 sc1\n\t"
               "global_load_dwordx4 %3, %8, off offset:3072 sc0 sc1\n\t"
               "global_load_dwordx4 %4, %9, off sc0 sc1\n\t"
               "global_load_dwordx4 %5, %9, off offset:1024 sc0 sc1\n\t"
               "global_load_dwordx4 %6, %9, off offset:2048 sc0 sc1\n\t"
               "global_load_dwordx4 %7, %9, off offset:3072 sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]),
                 "=&v"(v[1][3])
               : "v"(base), "v"(base + 4096) : "memory");
}

// GATED && XM == 1 (round 5): the backward of an L0-GATED activation in the dX product's epilogue (what evlm_gated_act_bwd did
// in a second pass over [rows, ffn]: read dA, read the pre-activation, write dH, column-sum the gate gradient).  With h = the
// pre-activation rows (aux), z = the gate row, dA = this tile:
//   gate before the activation (CLIP MLP, eff_vit.py:214-220):  t = dA act'(h z);  dH = t z;  dgate += sum_rows t h
//   gate after it (BERT FFN, eff_bert.py:552-557):              dH = dA act'(h) z;            dgate += sum_rows dA act(h)
// `dgs` = the wave's per-lane partial column sums (4 column groups x 4 columns), reduced and added to g.dgate once per tile.
template <bool FULL, int MODE, bool GATED, int XM, int ACT, int DACT, bool SKP>
__device__ __forceinline__ void pp_epi_chunk(const GemmP& g, f32x4 (&acc)[4][4], const f32x4 (&bz)[4], int b0, const PPRows& xr,
                                             int ic, int jb, int lane, char* sw, bf16* dst, int ldd, const PPSk& sk,
                                             f32x4* dgs = nullptr) {
  constexpr bool need_h = MODE == 2 && XM == 1, need_r = MODE == 2 && XM == 2;
  // XM == 3 (ABI 9): hidden-state dropout ahead of the residual.  The residual rows stay in the registers they were fetched
  // into - their [row][16-byte chunk] layout IS the store stage's - and the mask meets the result there: a lane then holds 8
  // consecutive columns of one row = one Philox call (in the accumulator layout it would hold 4: half a call)
  constexpr bool drop_r = MODE == 2 && XM == 3;
  constexpr bool gdact = MODE == 2 && GATED && XM == 1;
  const int il = lane & 15, jl = (lane >> 4) * 4;
  const int act = ACT >= 0 ? ACT : g.act, dact = DACT >= 0 ? DACT : g.dact;
  f32x4 gz[4];                             // L0 FFN gate (per output column), fetched per chunk: L2-resident, 16 registers
  constexpr bool gated = MODE == 2 && GATED;   // (compile-time: the ungated instantiation carries no gate registers)
  if (gated) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int j = jb + a * 16 + jl;
      gz[a] = *reinterpret_cast<const f32x4*>(g.gate + (FULL ? j : min(j, g.J - 4)));
    }
  }
  if (need_h || need_r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = k * 8 + (lane >> 3), ch = lane & 7;
      *reinterpret_cast<uint4*>(sw + r * 128 + ((ch ^ (r & 7)) << 4)) = xr.r[k];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  f32x4 pk[2][4];                          // SKP: [bb][a] partial sums of the other workgroups for this chunk
  if (SKP) {
    pp_sk_load8(sk.p0 + b0 * 4096, pk);
    for (int k = 1; k < sk.np; ++k) {      // (wave-uniform)
      f32x4 pk1[2][4];
      pp_sk_load8(sk.p0 + (size_t)k * PP_SK_NEXT + b0 * 4096, pk1);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int a = 0; a < 4; ++a) pk[bb][a] += pk1[bb][a];
    }
  }
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    const int r = bb * 16 + il;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ch = a * 2 + (jl >> 3);
      bf16* cell = reinterpret_cast<bf16*>(sw + r * 128 + ((ch ^ (r & 7)) << 4) + ((jl & 4) << 1));
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (SKP ? acc[a][b0 + bb][e] + pk[bb][a][e] : acc[a][b0 + bb][e]) * g.alpha + bz[a][e];
      if (gdact) {
        const bf16x4 xx = *reinterpret_cast<const bf16x4*>(cell);
        const bool live = FULL || (ic + r < g.I);             // (clamped rows of an edge tile carry no gate gradient)
        if (g.gate_pos == EVLM_GATE_PRE_ACT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = (float)xx[e];
            const float t = v[e] * act_grad_fast(dact, hv * gz[a][e]);
            v[e] = t * gz[a][e];
            if (live) dgs[a][e] = fmaf(t, hv, dgs[a][e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = (float)xx[e];
            if (live) dgs[a][e] = fmaf(v[e], act_apply_fast(dact, hv), dgs[a][e]);
            v[e] = v[e] * act_grad_fast(dact, hv) * gz[a][e];
          }
        }
      } else if (MODE == 2) {
        if (gated && g.gate_pos == EVLM_GATE_PRE_ACT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gz[a][e];
        }
        if (ACT != EVLM_ACT_NONE) {        // (ACT < 0: run-time code, may be NONE)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply_fast(act, v[e]);
        }
        if (gated && g.gate_pos != EVLM_GATE_PRE_ACT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gz[a][e];
        }
        if (need_h || need_r) {
          const bf16x4 xx = *reinterpret_cast<const bf16x4*>(cell);
          if (need_h) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad_fast(dact, (float)xx[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)xx[e];
          }
        }
      }
      Vec4<bf16>::store(cell, v);
    }
  }
  asm volatile("" ::: "memory");           // same-wave LDS traffic is in order; keep the compiler from reordering it
  __builtin_amdgcn_wave_barrier();
  DropRng rng;
  if (drop_r) rng = drop_rng(g.rng, g.call, g.drop_p);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = k * 8 + (lane >> 3), ch = lane & 7;
    uint4 v16 = *reinterpret_cast<const uint4*>(sw + r * 128 + ((ch ^ (r & 7)) << 4));
    const int i = ic + r, j = jb + ch * 8;
    if (drop_r) {
      float f[8];
      drop_factor8(rng, ((uint64_t)(FULL ? i : min(i, g.I - 1)) * g.J + (FULL ? j : min(j, g.J - 8))) >> 3, f);
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(&v16), rr = *reinterpret_cast<const bf16x8*>(&xr.r[k]);
      bf16x8 oo;
#pragma unroll
      for (int e = 0; e < 8; ++e) oo[e] = (bf16)(mul_rn((float)vv[e], f[e]) + (float)rr[e]);
      v16 = *reinterpret_cast<const uint4*>(&oo);
    }
#if defined(PP_EXP_NOSTORE)        // diagnostic builds only (tools/): what the epilogue costs without its global stores
    if (g.alpha == 12345.f) *reinterpret_cast<uint4*>(dst + (size_t)i * ldd + j) = v16;
#else
    if (FULL || (i < g.I && j < g.J)) *reinterpret_cast<uint4*>(dst + (size_t)i * ldd + j) = v16;
#endif
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// MODE 1: the pre-activation output (alpha * acc + bias); MODE 2: C.  aux (activation backward) OR residual rows (the host
// never routes both here) are fetched as FULL 128-byte row segments; they reach the fragment layout through the wave's LDS
// window, where the result then overwrites them in place.
// XM (compile-time, so that the plain instantiation carries no row registers): 0 none, 1 aux (activation backward),
// 2 residual
template <bool FULL, int MODE, bool GATED, int XM, int ACT, int DACT, bool SKP>
__device__ __forceinline__ void pp_epi_half(const GemmP& g, f32x4 (&acc)[4][4], const f32x4 (&bz)[4],
                                            int ib, int jb, int lane, char* sw, bf16* dst, int ldd, const PPSk& sk,
                                            f32x4* dgs = nullptr) {
  constexpr bool need_h = MODE == 2 && XM == 1, need_r = MODE == 2 && (XM == 2 || XM == 3);
  // (rows are requested per 64-row half: requesting all four chunks of the tile up front - 64 registers - was measured
  // SLOWER, 12.7 k against 10.6 k cycles per tile, the extra registers spill around the epilogue)
  PPRows x0, x1;
  if (need_h || need_r) {
    const bf16* xb = reinterpret_cast<const bf16*>(need_h ? g.aux : g.residual);
    x0 = pp_epi_rows<FULL>(g, xb, ib, jb, lane);
    x1 = pp_epi_rows<FULL>(g, xb, ib + 32, jb, lane);
  }
  pp_epi_chunk<FULL, MODE, GATED, XM, ACT, DACT, SKP>(g, acc, bz, 0, x0, ib, jb, lane, sw, dst, ldd, sk, dgs);
  pp_epi_chunk<FULL, MODE, GATED, XM, ACT, DACT, SKP>(g, acc, bz, 2, x1, ib + 32, jb, lane, sw, dst, ldd, sk, dgs);
}

__device__ __forceinline__ PPSk pp_sk_high(const PPSk& sk) {      // the slot addresses of the H half (i rows 64..127)
  PPSk h;
  h.p0 = sk.p0 + 16384; h.np = sk.np;
  return h;
}
// 32 more rows (i fragments 0, 1 of `acc`): the 192 x 256 tile flavour's third piece
template <bool FULL, int MODE, bool GATED, int XM, int ACT, int DACT>
__device__ __forceinline__ void pp_epi_third(const GemmP& g, f32x4 (&acc)[4][4], const f32x4 (&bz)[4],
                                             int ib, int jb, int lane, char* sw, bf16* dst, int ldd, f32x4* dgs = nullptr) {
  constexpr bool need_h = MODE == 2 && XM == 1, need_r = MODE == 2 && (XM == 2 || XM == 3);
  PPRows x0;
  if (need_h || need_r) x0 = pp_epi_rows<FULL>(g, reinterpret_cast<const bf16*>(need_h ? g.aux : g.residual), ib, jb, lane);
  pp_epi_chunk<FULL, MODE, GATED, XM, ACT, DACT, false>(g, acc, bz, 0, x0, ib, jb, lane, sw, dst, ldd, PPSk{nullptr, 0}, dgs);
}
// HI: 0 = the wave owns 64 rows (accL), 1 = 128 rows (accL, accH), 2 = 96 rows (accL + i fragments 0, 1 of accH)
template <bool FULL, bool GATED, int XM, int ACT, int DACT, bool SKP, int HI>
__device__ __forceinline__ void pp_epi_c(const GemmP& g, f32x4 (&accL)[4][4], f32x4 (&accH)[4][4], const f32x4 (&bz)[4], int ib,
                                         int jb, int lane, char* sw, const PPSk& sk) {
  constexpr bool gdact = GATED && XM == 1;
  f32x4 dgs[gdact ? 4 : 1];
  if (gdact) {
#pragma unroll
    for (int a = 0; a < 4; ++a) dgs[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4* dp = gdact ? dgs : nullptr;
  pp_epi_half<FULL, 2, GATED, XM, ACT, DACT, SKP>(g, accL, bz, ib, jb, lane, sw, reinterpret_cast<bf16*>(g.C), g.ldc, sk, dp);
  if (HI == 1)
    pp_epi_half<FULL, 2, GATED, XM, ACT, DACT, SKP>(g, accH, bz, ib + 64, jb, lane, sw, reinterpret_cast<bf16*>(g.C), g.ldc,
                                                    pp_sk_high(sk), dp);
  if (HI == 2)
    pp_epi_third<FULL, 2, GATED, XM, ACT, DACT>(g, accH, bz, ib + 64, jb, lane, sw, reinterpret_cast<bf16*>(g.C), g.ldc, dp);
  if (gdact) {
    // the wave's column sums: lanes il = 0..15 of one column quad hold different rows - reduce over them, then one f32 atomic
    // per column from the lane with il == 0 (64 columns per wave and tile; a column collects one atomic per row block)
    const int il = lane & 15, jl = (lane >> 4) * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = dgs[a][e];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        const int j = jb + a * 16 + jl + e;
        if (il == 0 && (FULL || j < g.J)) atomicAdd(g.dgate + j, v);
      }
  }
}

// HI = 0: the wave owns a 64 x 64 block only (128 x 256 tile flavour: accH is not touched); 2: 96 x 64 (192 x 256 flavour)
// GD: the kernel instantiation that serves the gated activation backward (evlm_gemm_args.dgate) - a SEPARATE instantiation
// (gemm_bf16_pp192_kernel<QT, true>): compiled into the shared kernels the flavour's 16 column-sum registers pushed the
// 256-row kernel over 256 VGPRs (37-47 spilled registers in the kernel that carries a third of the step's GEMM time).
template <bool FULL, bool SKP = false, int HI = 1, bool GD = false>
__device__ __forceinline__ void pp_epilogue(const GemmP& g, f32x4 (&accL)[4][4], f32x4 (&accH)[4][4], int ib, int jb, int lane,
                                            char* sw, const PPSk& sk = PPSk{nullptr, 0}) {
  f32x4 bz[4];
  pp_epi_cols<FULL>(g, jb, lane, bz);
  if (g.preact) {
    pp_epi_half<FULL, 1, false, 0, 0, 0, SKP>(g, accL, bz, ib, jb, lane, sw, reinterpret_cast<bf16*>(g.preact), g.ldx, sk);
    if (HI == 1)
      pp_epi_half<FULL, 1, false, 0, 0, 0, SKP>(g, accH, bz, ib + 64, jb, lane, sw, reinterpret_cast<bf16*>(g.preact), g.ldx,
                                                pp_sk_high(sk));
    if (HI == 2)
      pp_epi_third<FULL, 1, false, 0, 0, 0>(g, accH, bz, ib + 64, jb, lane, sw, reinterpret_cast<bf16*>(g.preact), g.ldx);
  }
  // one instantiation per epilogue flavour (ONE wave-uniform dispatch per tile): the plain one carries neither gate nor
  // row registers, and every flavour of the training path has its activation code as a compile-time constant
  constexpr int G = EVLM_ACT_GELU, QG = EVLM_ACT_QUICK_GELU, N = EVLM_ACT_NONE;
  if (GD) {                          // backward of the L0-gated activation (round 5): dH and the gate gradient, run-time codes
    pp_epi_c<FULL, true, 1, N, -1, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
    return;
  }
  if (g.gate) {                      // L0-gated FFN (pruning fine-tune only): activation code read at run time
    if (g.residual) pp_epi_c<FULL, true, 2, -1, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
    else pp_epi_c<FULL, true, 0, -1, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
  } else if (g.drop_p > 0.f) pp_epi_c<FULL, false, 3, N, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);      // (+ residual)
  else if (g.dact == G) pp_epi_c<FULL, false, 1, N, G, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
  else if (g.dact == QG) pp_epi_c<FULL, false, 1, N, QG, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
  else if (g.residual) {
    if (g.act == N) pp_epi_c<FULL, false, 2, N, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
    else pp_epi_c<FULL, false, 2, -1, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
  } else if (g.act == G) pp_epi_c<FULL, false, 0, G, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
  else if (g.act == QG) pp_epi_c<FULL, false, 0, QG, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
  else pp_epi_c<FULL, false, 0, N, N, SKP, HI>(g, accL, accH, bz, ib, jb, lane, sw, sk);
}

// weight gradients: f32 tile out of the same 4 KiB window, 16 rows x 64 columns at a time.  Plain 16-byte stores when the
// workgroup owns the whole reduction and the output is not accumulated; otherwise f32 atomics issued so that one wave
// instruction covers 256 contiguous bytes (scattered dword atomics run an order of magnitude slower).
// mode 0: plain stores, 1: f32 atomics, 2: C += tile by load / add / store (the tile has ONE owner in the launch: grouped
// weight gradients accumulating into the gradient slab)
template <bool FULL>
__device__ __forceinline__ void pp_epi_f32_half(const GemmP& g, f32x4 (&acc)[4][4], int ib, int jb, int lane, char* sw,
                                                int mode) {
  const int il = lane & 15, jq = lane >> 4;
  float* Cf = reinterpret_cast<float*>(g.C);
  const bool atomic = mode == 1;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f32x4 cold[4];
    if (mode == 2) {                         // issue the C loads first: they fly under the LDS transpose below
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = k * 4 + (lane >> 4), ch = lane & 15;
        const int i = ib + b * 16 + r, j = jb + ch * 4;
        cold[k] = (FULL || (i < g.I && j < g.J)) ? *reinterpret_cast<const f32x4*>(Cf + (size_t)i * g.ldc + j)
                                                 : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4 v = acc[a][b];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= g.alpha;
      *reinterpret_cast<f32x4*>(sw + il * 256 + (((a * 4 + jq) ^ il) << 4)) = v;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (atomic) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = *reinterpret_cast<const float*>(sw + r * 256 + (((lane >> 2) ^ r) << 4) + ((lane & 3) << 2));
        const int i = ib + b * 16 + r, j = jb + lane;
        if (FULL || (i < g.I && j < g.J)) atomicAdd(Cf + (size_t)i * g.ldc + j, v);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = k * 4 + (lane >> 4), ch = lane & 15;
        f32x4 v = *reinterpret_cast<const f32x4*>(sw + r * 256 + ((ch ^ r) << 4));
        if (mode == 2) v += cold[k];
        const int i = ib + b * 16 + r, j = jb + ch * 4;
        if (FULL || (i < g.I && j < g.J)) *reinterpret_cast<f32x4*>(Cf + (size_t)i * g.ldc + j) = v;   // J % 4 == 0
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// XCD-aware bijective map of a virtual block id onto the tile grid (see tile_coords): blocks that share an XCD (id mod 8)
// get a contiguous range of tile ids, j fastest
__device__ __forceinline__ void pp_tile_ij(const GemmP& g, int vb, int ntiles, int& ti, int& tj) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
  ti = t / g.tiles_j;
  tj = t - ti * g.tiles_j;
}

