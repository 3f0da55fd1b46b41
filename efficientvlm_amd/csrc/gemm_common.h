// shared by the GEMM translation units: launch parameter block, fused tile epilogue, XCD-aware tile map
#pragma once
#include "common.h"

struct GemmP {
  const void* P; const void* Q; void* C;
  const float* bias; const float* gate;
  void* preact; const void* aux; const void* residual;
  int I, J, K, ldp, ldq, ldc, ldx;
  int c_f32, act, gate_pos, dact;
  float alpha;
  int tiles_i, tiles_j;
  int kt_per_split;   // K tiles (of 64) handled by one grid.y slice
  int bare_f32;       // f32 output with no epilogue terms (weight gradients): LDS-staged coalesced store / atomics
  int accumulate;     // C += result (f32 atomics), no zero-fill
  float* psum;        // [I] += sum_k P(i,k) (bias gradient), or nullptr
  void* sk_ws;        // stream-K workspace (gemm_pp256.hip): 256 flag words, then 256 f32 tile slots of 256 KiB; or nullptr
  int sk;             // launch form chosen by the host: 1 = stream-K
  float* dgate;       // [J] f32, accumulated: gate gradient of a gated activation backward folded into this dX product (ABI 8)
  float drop_p; const int64_t* rng; uint32_t call;      // ABI 9: C = (..) .* keep / (1 - p) + residual (hidden-state dropout)
};

// ---------------------------------------------------------------------------------------------
// tile epilogue.  Each lane owns, for every (a, b) accumulator, FOUR CONSECUTIVE j of one row i:
//     i = ibase + b*16 + (lane & 15),   j = jbase + a*16 + (lane >> 4)*4 + 0..3
// Per-column vectors (bias, gate) are fetched once per `a` as 16-byte loads; aux / residual rows are fetched as
// 8/16-byte vectors for all `b` of one `a` before they are consumed, so the loads overlap instead of serialising.
// FULL = the whole workgroup tile is inside the matrix (no bounds checks at all).
// ---------------------------------------------------------------------------------------------
// (static register indices only: a runtime-indexed v[e] would push the arrays to scratch)
template <typename T>
__device__ __forceinline__ void ld4(const T* p, int nv, float v[4]) {
  if (nv == 4) Vec4<T>::load(p, v);
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (e < nv) ? to_f(p[e]) : 0.f;
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, int nv, const float v[4]) {
  if (nv == 4) Vec4<T>::store(p, v);
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e < nv) p[e] = from_f<T>(v[e]);
  }
}

// LDS_OUT (bf16 interior tiles): C (and the pre-activation) are first written into swizzled [128][128] bf16 LDS tiles
// (sC, sH; local row/col = global - (i0, j0)) and then streamed out by the whole workgroup as 256-byte rows
// (copy_tile_out), instead of 8-byte stores scattered over 16 rows per wave instruction.
template <int MT>
__device__ __forceinline__ int ctile_off(int r, int c) {
  constexpr int NC = 4 * MT;   // 16-byte chunks per LDS tile row
  return r * (NC * 16) + ((((c >> 3) ^ r) & (NC - 1)) << 4) + ((c & 4) << 1);
}

// MODE 0: pre-activation and C in one pass; 1: only the pre-activation (into sH); 2: only C (the 256x256 kernel has one
// LDS tile to stage through, so it runs the two outputs as two passes)
// ACT / DACT: compile-time activation codes, -1 = read g.act / g.dact at run time (tile_epilogue below dispatches ONCE per
// tile: with run-time codes hipcc keeps scalar compares and branches around every single element)
// DROP (ABI 9): v = v .* keep / (1 - p) ahead of the residual.  A lane owns 4 consecutive columns of a row - one half of a
// Philox call's 8 factors (the 256-column ping-pong kernels apply the mask at their 16-byte store stage, a full call each)
template <typename T, int NA, int NB, bool FULL, bool LDS_OUT, int MT, int MODE, int ACT, int DACT, bool DROP = false>
__device__ __forceinline__ void tile_epilogue_impl(const GemmP& g, f32x4 (&acc)[NA][NB], int ibase, int jbase, int lane,
                                                   char* sC, char* sH, int i0, int j0) {
  const int il = lane & 15, jl = (lane >> 4) * 4;
  const int act = ACT >= 0 ? ACT : g.act, dact = DACT >= 0 ? DACT : g.dact;
  constexpr bool LOWP = sizeof(T) == 2;      // bf16 path: fast activation math (common.h); the f32 parity path stays exact
  DropRng rng;
  if (DROP) rng = drop_rng(g.rng, g.call, g.drop_p);
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int j = jbase + a * 16 + jl;
    const int nv = FULL ? 4 : max(0, min(4, g.J - j));
    if (!FULL && nv == 0) continue;
    float bz[4] = {0.f, 0.f, 0.f, 0.f}, gz[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.bias) ld4<float>(g.bias + j, nv, bz);
    if (g.gate) ld4<float>(g.gate + j, nv, gz);
    float hx[NB][4], rx[NB][4];
    if (FULL && MODE != 1 && dact != EVLM_ACT_NONE) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        ld4<T>(reinterpret_cast<const T*>(g.aux) + (size_t)(ibase + b * 16 + il) * g.ldx + j, 4, hx[b]);
    }
    if (FULL && MODE != 1 && g.residual) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        ld4<T>(reinterpret_cast<const T*>(g.residual) + (size_t)(ibase + b * 16 + il) * g.ldx + j, 4, rx[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = ibase + b * 16 + il;
      if (!FULL && i >= g.I) continue;
      if (!FULL && MODE != 1) {   // edge tiles: fetch per element block (rare; keeps the register budget of the interior path)
        if (dact != EVLM_ACT_NONE) ld4<T>(reinterpret_cast<const T*>(g.aux) + (size_t)i * g.ldx + j, nv, hx[b]);
        if (g.residual) ld4<T>(reinterpret_cast<const T*>(g.residual) + (size_t)i * g.ldx + j, nv, rx[b]);
      }
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[a][b][e] * g.alpha + bz[e];
      if (MODE != 2 && g.preact) {
        if (LDS_OUT) Vec4<bf16>::store(reinterpret_cast<bf16*>(sH + ctile_off<MT>(i - i0, j - j0)), v);
        else st4<T>(reinterpret_cast<T*>(g.preact) + (size_t)i * g.ldx + j, nv, v);
      }
      if (MODE == 1) continue;
      if (act != EVLM_ACT_NONE) {
        if (g.gate_pos == EVLM_GATE_PRE_ACT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = LOWP ? act_apply_fast(act, v[e] * gz[e]) : act_apply(act, v[e] * gz[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (LOWP ? act_apply_fast(act, v[e]) : act_apply(act, v[e])) * gz[e];
        }
      } else if (g.gate) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gz[e];
      }
      if (dact != EVLM_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= LOWP ? act_grad_fast(dact, hx[b][e]) : act_grad(dact, hx[b][e]);
      }
      if (DROP && MODE != 1) {
        float f[4];
        drop_factor4(rng, ((uint64_t)i * g.J + j) >> 3, (j & 4) != 0, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mul_rn(to_f(from_f<T>(v[e])), f[e]);      // (the product as it would be STORED, then
                                                                                       // x .* m rounded before the residual: what the
                                                                                       // 256-column kernels and evlm_dropout compute)
      }
      if (g.residual) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rx[b][e];
      }
      const size_t co = (size_t)i * g.ldc + j;
      if (LDS_OUT) Vec4<bf16>::store(reinterpret_cast<bf16*>(sC + ctile_off<MT>(i - i0, j - j0)), v);
      else if (g.c_f32) st4<float>(reinterpret_cast<float*>(g.C) + co, nv, v);
      else st4<T>(reinterpret_cast<T*>(g.C) + co, nv, v);
    }
  }
}

template <typename T, int NA, int NB, bool FULL, bool LDS_OUT = false, int MT = 4, int MODE = 0>
__device__ __forceinline__ void tile_epilogue(const GemmP& g, f32x4 (&acc)[NA][NB], int ibase, int jbase, int lane,
                                              char* sC = nullptr, char* sH = nullptr, int i0 = 0, int j0 = 0) {
#define EVLM_EPI(A_, D_) tile_epilogue_impl<T, NA, NB, FULL, LDS_OUT, MT, MODE, A_, D_>(g, acc, ibase, jbase, lane, sC, sH, i0, j0)
  constexpr int G = EVLM_ACT_GELU, QG = EVLM_ACT_QUICK_GELU, N = EVLM_ACT_NONE;
  if (g.drop_p > 0.f) {                              // hidden-state dropout + residual (the host admits no act / gate / dact with it)
    tile_epilogue_impl<T, NA, NB, FULL, LDS_OUT, MT, MODE, N, N, true>(g, acc, ibase, jbase, lane, sC, sH, i0, j0);
    return;
  }
  if (sizeof(T) == 2 && FULL && !g.gate) {           // interior bf16 tiles of the training path: compile-time activation codes
    if (g.dact == N) {
      if (g.act == N) EVLM_EPI(N, N);
      else if (g.act == G) EVLM_EPI(G, N);
      else EVLM_EPI(QG, N);
    } else if (g.act == N) {
      if (g.dact == G) EVLM_EPI(N, G);
      else EVLM_EPI(N, QG);
    } else EVLM_EPI(-1, -1);
  } else EVLM_EPI(-1, -1);
#undef EVLM_EPI
}

// XCD-aware, bijective block -> tile map: blocks b and b+8 share an XCD (and its L2), so give each
// XCD a contiguous range of tile ids; inside the range j runs fastest (tiles sharing a P row panel).
__device__ __forceinline__ void tile_coords(const GemmP& g, int& ti, int& tj) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  ti = t / g.tiles_j;
  tj = t - ti * g.tiles_j;
}
