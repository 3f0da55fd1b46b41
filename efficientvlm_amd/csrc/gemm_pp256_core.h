// Shared core of the 256x256x64 ping-pong bf16 GEMM schedule (gemm_pp256.hip has the full description): staging-unit
// geometry, LDS-DMA source offsets, fragment reads and the four-phase K-tile macros.  Included by gemm_pp256.hip and by
// xattn_fused.hip (the fused cross-attention kernel runs its K/V projection on the same K loop).
#pragma once
#include "gemm_common.h"

#define PPU 16384                     // bytes per staging unit: 128 operand rows x 64 k x 2 B
#define PPB 65536                     // bytes per buffer: PL | PH | QL | QH
#define OFF_PL 0
#define OFF_PH PPU
#define OFF_QL (2 * PPU)
#define OFF_QH (3 * PPU)
#define PP_EPI_OFF (2 * PPB)          // 8 x 4 KiB epilogue windows behind the staging buffers (160 KiB of LDS in total)

// per-lane source BYTE offsets (unsigned 32-bit) of the 2 x 16 bytes staged per unit.  The LDS-DMA address is formed as
// (wave-uniform 64-bit base: operand + K tile) + zero-extended per-lane offset, i.e. the SGPR-base + VGPR-offset form of
// global_load_lds: no 64-bit vector address arithmetic in the K loop and no address registers carried through it
struct PPSrc { uint32_t pl[2], ph[2], ql[2], qh[2]; };

// tile row (P units) / tile column (Q units) of unit row u
__device__ __forceinline__ int pp_prow(int u, int hi) { return (u >> 6) * 128 + (u & 63) + hi * 64; }
__device__ __forceinline__ int pp_qcol(int u, int hi) { return (u >> 5) * 64 + (u & 31) + hi * 32; }
// 16-byte-chunk swizzle of a reduction-major unit row ([64 k][128 rows], 256-byte rows): conflict-free tr reads
__device__ __forceinline__ int pp_trswz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }

// K-contiguous unit: [128 rows][64 k], 128-byte rows, chunk ^= (row >> 1) & 7.  Reduction-major unit: [64 k][128 rows].
// One wave instruction fills 1 KiB of LDS linearly, so the swizzle is applied to the per-lane SOURCE chunk.
template <bool PT, bool QT>
__device__ __forceinline__ void pp_src(const GemmP& g, int i0, int j0, int tid, PPSrc& s) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = (c * 8 + wave) * 64 + lane;          // 16-byte LDS slot, linear per wave instruction
    {
      const int u = id >> 3, cp = id & 7;
      const int koff = (cp ^ ((u >> 1) & 7)) << 3;
      if (!PT) {
        s.pl[c] = (uint32_t)(min(i0 + pp_prow(u, 0), g.I - 1) * g.ldp + koff) * 2u;
        s.ph[c] = (uint32_t)(min(i0 + pp_prow(u, 1), g.I - 1) * g.ldp + koff) * 2u;
      }
      if (!QT) {
        s.ql[c] = (uint32_t)(min(j0 + pp_qcol(u, 0), g.J - 1) * g.ldq + koff) * 2u;
        s.qh[c] = (uint32_t)(min(j0 + pp_qcol(u, 1), g.J - 1) * g.ldq + koff) * 2u;
      }
    }
    {
      const int kr = id >> 4, cp = id & 15;
      const int u0 = (cp ^ pp_trswz(kr)) << 3;
      if (PT) {
        const int lim = ((g.I + 7) & ~7) - 8;
        s.pl[c] = (uint32_t)(kr * g.ldp + min(i0 + pp_prow(u0, 0), lim)) * 2u;
        s.ph[c] = (uint32_t)(kr * g.ldp + min(i0 + pp_prow(u0, 1), lim)) * 2u;
      }
      if (QT) {
        const int lim = ((g.J + 7) & ~7) - 8;
        s.ql[c] = (uint32_t)(kr * g.ldq + min(j0 + pp_qcol(u0, 0), lim)) * 2u;
        s.qh[c] = (uint32_t)(kr * g.ldq + min(j0 + pp_qcol(u0, 1), lim)) * 2u;
      }
    }
  }
}

// base: wave-uniform operand pointer, kel: wave-uniform element offset of the K tile, so: per-lane byte offsets
#define PP_GLDS(base, so, kel, ldsoff)                                                                              \
  do {                                                                                                              \
    const char* ub__ = reinterpret_cast<const char*>((base) + (size_t)(kel));                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub__ + (so)[0]),               \
                                     (__attribute__((address_space(3))) void*)(smem + (ldsoff) + wave * 1024), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub__ + (so)[1]),               \
                                     (__attribute__((address_space(3))) void*)(smem + (ldsoff) + (8 + wave) * 1024), 16, 0, 0); \
  } while (0)

// Fragment f (16 operand rows) x k sub-step ks (32 deep) of the unit at byte offset `uo`.  `lb` is ONE per-lane base
// (pp_lane_base): every other address term is a compile-time XOR / immediate, so the K loop carries no address registers.
//   K-contiguous unit : row (lane & 15) of fragment f, chunk (ks*4 + (lane >> 4)) ^ swizzle   -> (lb ^ ks*64) + f*2048
//   reduction-major   : two transposing reads (h = 0, 1) of k rows ks*32 + (lane >> 4)*8 + h*4 + q; the swizzled chunk
//                       differs from the lane's base chunk only in the bits (f >> 1, f & 1, h)  -> lb ^ (those bits << 4)
//
// The transposing read is INLINE ASM, not __builtin_amdgcn_ds_read_tr16_b64 (round 4).  hipcc (ROCm 7.2) models an LDS-DMA
// as a pending LDS write and the builtin as an LDS read it cannot disambiguate from it, so it put `s_waitcnt vmcnt(0)` in
// front of the first transposed read of EVERY phase (7-8 per two K tiles in the .s of every PT / QT kernel of this family;
// the ds_read_b128 form - a plain pointer load - gets none): each phase then drained the whole staging pipeline, i.e. every
// unit had to land within the one phase after its issue instead of the 3-6 phases the counted vmcnt schedule gives it -
// the weight-gradient kernel (both operands reduction-major) ran at 0.30 of the MFMA peak on a 197-K-tile reduction where
// the K-contiguous loop of the same family holds 0.45-0.49.  An asm statement is invisible to that pass (and to its
// lgkmcnt bookkeeping: PP_MFMA_BEGIN carries the `s_waitcnt lgkmcnt(0)` the MFMAs need, behind a sched_barrier as rule 18
// of the guide asks); the RAW / WAR ordering of staged units against these reads is the counted-vmcnt + barrier
// argument of the file header, exactly as for the ds_read_b128 reads.  The immediates are compile-time after inlining
// (16-bit offset field; bit 16 of the unit offset - the second staging buffer - goes into the address register).
template <bool TR>
__device__ __forceinline__ bf16x8 pp_frag(const char* smem, int uo, int lb, int f, int ks) {
  if (!TR) {
    return *reinterpret_cast<const bf16x8*>(smem + (lb ^ (ks << 6)) + (uo + f * 2048));
  } else {
    bf16x8 out;
    const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cz = ((f >> 1) << 6) | ((f & 1) << 5) | (h << 4);
      const int off = uo + ks * 8192 + h * 1024;
      const uint32_t a = sb + (uint32_t)(lb ^ cz) + (uint32_t)(off & ~0xFFFF);
      bf16x4 t;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t) : "v"(a), "n"(off & 0xFFFF));
      out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
    }
    return out;
  }
}
// per-lane base for pp_frag; `wsel` = wave row (P units: 2 x 64 unit rows) or wave column (Q units: 4 x 32)
template <bool TR, bool IS_P>
__device__ __forceinline__ int pp_lane_base(int lane, int wsel) {
  if (!TR) {
    const int l15 = lane & 15;
    return (wsel * (IS_P ? 64 : 32) + l15) * 128 + (((lane >> 4) ^ ((l15 >> 1) & 7)) << 4);
  } else {
    const int g4 = lane >> 4, w = lane & 15, q = w >> 2, pp = w & 3;
    const int hi = IS_P ? (((wsel ^ (q >> 1)) << 1) | (q & 1)) : (wsel ^ q);     // chunk bits 3:2 at f = 0
    return g4 * 2048 + q * 256 + ((pp & 1) << 3) + (hi << 6) + ((g4 & 1) << 5) + ((pp >> 1) << 4);
  }
}

// [memory work] | barrier | MFMAs | barrier : the sched_barriers keep hipcc from moving MFMAs (pure register ops) across
// (the explicit lgkmcnt(0): the asm transposing reads of pp_frag<true> are not in the compiler's own count)
#define PP_MFMA_BEGIN()                                       \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_barrier();                               \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_setprio(1)
#define PP_MFMA_END()                   \
  __builtin_amdgcn_s_setprio(0);        \
  __builtin_amdgcn_sched_barrier(0);    \
  __builtin_amdgcn_s_barrier();         \
  __builtin_amdgcn_sched_barrier(0)

#define PP_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// 16 MFMAs: acc[AO + a][b] += Qf[a][ks] x Pf[b][ks]
#define PP_QUAD(ACC, AO, QF)                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                           \
  _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                           \
      ACC[AO + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QF[a][ks], pf[b][ks], ACC[AO + a][b], 0, 0, 0)

// bias gradient: per-lane partial row sums of the P fragments in registers (lane holds k = 8*(lane>>4)..+7 of row lane&15).
// All four wave columns of a wave row hold the same eight P fragments, so the sums are SHARED OUT: wave column c sums i
// fragments 2c, 2c + 1 (columns 0, 1 in the phase that holds fragments 0..3, columns 2, 3 in the one that holds 4..7).
// Left to wave column 0 alone (rounds 1-3) the 128 conversions + adds per K tile sat inside ONE wave's MFMA sections and
// every barrier of the workgroup waited for it: a tile with a bias gradient (a third of the ViT's) ran at that wave's pace.
#define PP_PSUM(PS, BO)                                                                                    \
  if (OUT == 1 && do_psum && (wc >> 1) == ((BO) >> 2)) {                                                   \
    if (wc & 1) {                                                                                          \
      _Pragma("unroll") for (int b = 2; b < 4; ++b)                                                        \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) PS[b - 2] += (float)pf[b][ks][e];                      \
    } else {                                                                                               \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                        \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) PS[b] += (float)pf[b][ks][e];                          \
    }                                                                                                      \
  }

// one K tile out of buffer BUF (compile-time LDS offsets); K tile index t of nt, operands at Pk / Qk.
// N1 / N2 = "K tile t+1 / t+2 exists": LITERAL true in the steady-state loops (round 4) - as run-time flags every staging
// piece and every counted wait sat behind a scalar compare-and-branch inside the memory sections (on the one-wave-per-SIMD
// kernel of gemm_w4.hip, where nothing hides them, those branches alone cost 8 % of the K loop).
// The j-lo Q fragments are read TWICE (phases 0 and 3) instead of being carried through phases 1-2: 16 registers less at
// the peak, which is what keeps the kernel's long-lived values (next-tile offsets, epilogue pointers) out of scratch, and
// phase 3 gets LDS reads of its own (it had none), evening out the memory sections the partner wave's MFMAs have to cover.
#define PP_KTILE(BUF, t, N1, N2)                                                                                 \
  do {                                                                                                     \
    constexpr int B0 = (BUF) * PPB, B1 = ((BUF) ^ 1) * PPB;                                                \
    const bool n1 = (N1), n2 = (N2);                                                                              \
    /* ---- phase 0: i-lo x j-lo ; stage PH(t+1) ---- */                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qf[a][0] = pp_frag<QT>(smem, B0 + OFF_QL, qlb, a, 0);                                                \
      qf[a][1] = pp_frag<QT>(smem, B0 + OFF_QL, qlb, a, 1); }                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
      pf[b][0] = pp_frag<PT>(smem, B0 + OFF_PL, plb, b, 0);                                                \
      pf[b][1] = pp_frag<PT>(smem, B0 + OFF_PL, plb, b, 1); }                                              \
    if (n1) PP_GLDS(Pk, src.ph, ((t) + 1) * kp, B1 + OFF_PH);                                              \
    if (n1) PP_WAIT(10); else PP_WAIT(4);                                                                  \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 0, qf); PP_PSUM(ps, 0); PP_MFMA_END();                                  \
    /* ---- phase 1: i-lo x j-hi ; stage QL(t+1) ---- */                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qf[a][0] = pp_frag<QT>(smem, B0 + OFF_QH, qlb, a, 0);                                                \
      qf[a][1] = pp_frag<QT>(smem, B0 + OFF_QH, qlb, a, 1); }                                              \
    if (n1) PP_GLDS(Qk, src.ql, ((t) + 1) * kq, B1 + OFF_QL);                                              \
    if (n1) PP_WAIT(10); else PP_WAIT(2);                                                                  \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 2, qf); PP_MFMA_END();                                                  \
    /* ---- phase 2: i-hi x j-hi ; stage PL(t+2) ---- */                                                   \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
      pf[b][0] = pp_frag<PT>(smem, B0 + OFF_PH, plb, b, 0);                                                \
      pf[b][1] = pp_frag<PT>(smem, B0 + OFF_PH, plb, b, 1); }                                              \
    if (n2) PP_GLDS(Pk, src.pl, ((t) + 2) * kp, B0 + OFF_PL);                                              \
    PP_MFMA_BEGIN(); PP_QUAD(accH, 2, qf); PP_PSUM(ps, 4); PP_MFMA_END();                                  \
    /* ---- phase 3: i-hi x j-lo (j-lo fragments read again) ; stage QH(t+2) ---- */                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qf[a][0] = pp_frag<QT>(smem, B0 + OFF_QL, qlb, a, 0);                                                \
      qf[a][1] = pp_frag<QT>(smem, B0 + OFF_QL, qlb, a, 1); }                                              \
    if (n2) PP_GLDS(Qk, src.qh, ((t) + 2) * kq, B0 + OFF_QH);                                              \
    if (n2) PP_WAIT(4); else if (n1) PP_WAIT(0);                                                           \
    PP_MFMA_BEGIN(); PP_QUAD(accH, 0, qf); PP_MFMA_END();                                                  \
  } while (0)

// Variant that HOLDS the j-lo Q fragments through phases 1-2 (no second read; staging order PL QL QH PH, vmcnt(8)): every
// unit keeps a lead of five phases, which the weight-gradient kernels need - there BOTH operands are activations streamed
// from HBM, and the three-phase lead the re-read schedule leaves QL is too short for them (measured: +3.5 % on the grouped
// launch).  Their epilogue is small enough that the extra 16 registers do not spill.
#define PP_KTILE_HOLD(BUF, t, N1, N2)                                                                                 \
  do {                                                                                                     \
    constexpr int B0 = (BUF) * PPB, B1 = ((BUF) ^ 1) * PPB;                                                \
    const bool n1 = (N1), n2 = (N2);                                                                              \
    /* ---- phase 0: i-lo x j-lo ---- */                                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      ql[a][0] = pp_frag<QT>(smem, B0 + OFF_QL, qlb, a, 0);                                     \
      ql[a][1] = pp_frag<QT>(smem, B0 + OFF_QL, qlb, a, 1); }                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
      pf[b][0] = pp_frag<PT>(smem, B0 + OFF_PL, plb, b, 0);                                     \
      pf[b][1] = pp_frag<PT>(smem, B0 + OFF_PL, plb, b, 1); }                                   \
    if (n1) PP_GLDS(Qk, src.qh, ((t) + 1) * kq, B1 + OFF_QH);                                              \
    if (n1) PP_WAIT(8); else PP_WAIT(2);                                                                   \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 0, ql); PP_PSUM(ps, 0); PP_MFMA_END();                                  \
    /* ---- phase 1: i-lo x j-hi ---- */                                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qh[a][0] = pp_frag<QT>(smem, B0 + OFF_QH, qlb, a, 0);                                     \
      qh[a][1] = pp_frag<QT>(smem, B0 + OFF_QH, qlb, a, 1); }                                   \
    if (n1) PP_GLDS(Pk, src.ph, ((t) + 1) * kp, B1 + OFF_PH);                                              \
    if (n1) PP_WAIT(8); else PP_WAIT(0);                                                                   \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 2, qh); PP_MFMA_END();                                                  \
    /* ---- phase 2: i-hi x j-hi ---- */                                                                   \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
      pf[b][0] = pp_frag<PT>(smem, B0 + OFF_PH, plb, b, 0);                                     \
      pf[b][1] = pp_frag<PT>(smem, B0 + OFF_PH, plb, b, 1); }                                   \
    if (n2) PP_GLDS(Pk, src.pl, ((t) + 2) * kp, B0 + OFF_PL);                                              \
    PP_MFMA_BEGIN(); PP_QUAD(accH, 2, qh); PP_PSUM(ps, 4); PP_MFMA_END();                                  \
    /* ---- phase 3: i-hi x j-lo ---- */                                                                   \
    if (n2) PP_GLDS(Qk, src.ql, ((t) + 2) * kq, B0 + OFF_QL);                                              \
    if (n2) PP_WAIT(8); else if (n1) PP_WAIT(4);                                                           \
    PP_MFMA_BEGIN(); PP_QUAD(accH, 0, ql); PP_MFMA_END();                                                  \
  } while (0)

