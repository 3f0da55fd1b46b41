// 256x256x64 bf16 GEMM with ONE wave per SIMD (round 4): C[i,j] = epi( alpha * sum_k P(i,k) * Q(j,k) ), P K-contiguous,
// Q K-contiguous or reduction-major (transposing LDS reads), bf16 output through the fused epilogues of gemm_pp256_epi.h.
//
// Why a third schedule.  The ping-pong kernels of gemm_pp256.hip run two waves per SIMD, 128 x 64 of C each (256 registers
// per wave is all two co-resident waves can have), one barrier apart: every 16 MFMAs are fenced by two workgroup barriers,
// and a K tile costs 56 ds_read_b128 + 16 LDS-DMA issues per SIMD.  In-kernel stamps put that loop at 72-74 % MFMA duty
// (2 780 cycles per K tile against 2 048), with or without the staging DMAs in it (profiles/r03_pp192v_staging.md).  Here a
// wave owns 128 x 128 of C (256 accumulator registers in the unified 512-entry file, the compiler places them in AGPRs):
//   * 32 fragment reads per K tile and SIMD instead of 56, every read issued INSIDE the MFMA stream (one or two behind
//     every 4 MFMAs of the first three quarters of a step), into the register set of the NEXT 32-deep step (two sets of 16
//     fragments) - no read burst, nothing waits on a read;
//   * ONE workgroup barrier per K tile (128 MFMAs) instead of eight;
//   * the 16 LDS-DMA pieces a wave stages per K tile sit in the first half of each step, 8 MFMAs apart.
// LDS: the two 64 KiB staging buffers and the unit layout of the ping-pong kernels (PL | PH | QL | QH, gemm_pp256_core.h),
// so fragment addressing, source swizzles and the fused epilogues are shared.
//
// Schedule.  K tile t lives in buffer t & 1; step 2t works on its k 0..31 out of fragment set A, step 2t+1 on k 32..63
// out of set B:
//   even step 2t  : [64 MFMAs on set A]  reads (t, k-hi) -> set B        DMA Q(t+1) -> buffer (t+1) & 1
//   odd step 2t+1 : vmcnt(0) lgkmcnt(0) BARRIER
//                   [64 MFMAs on set B]  reads (t+1, k-lo) -> set A      DMA P(t+2) -> buffer t & 1
// RAW: K tile t+1 is read from the odd step 2t+1 on; its P units were issued in step 2t-1 (two steps of lead: the
// streamed operand), its Q units in step 2t (one step: the weight operand, L2-resident); every wave waits for ITS pieces
// before the barrier of step 2t+1, so every piece has landed once any wave is past it.  WAR: buffer t & 1 is overwritten
// by P(t+2) from step 2t+1 on, behind the barrier that every wave reaches only after its last reads of that buffer (issued in
// step 2t) have retired (lgkmcnt(0)); Q(t+1) goes into buffer (t+1) & 1 in step 2t, whose last reads - K tile t-1, issued
// in step 2t-2 - retired before the barrier of step 2t-1.
#include <stdlib.h>
#include "gemm_common.h"
#include "gemm_pp256_core.h"
#include "gemm_pp256_epi.h"

#define W4_EPI_OFF (2 * PPB)          // 4 x 4 KiB epilogue windows behind the staging buffers (144 KiB of LDS)

struct W4Src { uint32_t pl[4], ph[4], ql[4], qh[4]; };      // per-lane source byte offsets: 4 pieces per unit and lane

template <bool QT>
__device__ __forceinline__ void w4_src(const GemmP& g, int i0, int j0, int tid, W4Src& s) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = (c * 4 + wave) * 64 + lane;          // 16-byte LDS slot of the unit, linear per wave instruction
    {
      const int u = id >> 3, cp = id & 7;
      const int koff = (cp ^ ((u >> 1) & 7)) << 3;
      s.pl[c] = (uint32_t)(min(i0 + pp_prow(u, 0), g.I - 1) * g.ldp + koff) * 2u;
      s.ph[c] = (uint32_t)(min(i0 + pp_prow(u, 1), g.I - 1) * g.ldp + koff) * 2u;
      if (!QT) {
        s.ql[c] = (uint32_t)(min(j0 + pp_qcol(u, 0), g.J - 1) * g.ldq + koff) * 2u;
        s.qh[c] = (uint32_t)(min(j0 + pp_qcol(u, 1), g.J - 1) * g.ldq + koff) * 2u;
      }
    }
    if (QT) {
      const int kr = id >> 4, cp = id & 15;
      const int u0 = (cp ^ pp_trswz(kr)) << 3;
      const int lim = ((g.J + 7) & ~7) - 8;
      s.ql[c] = (uint32_t)(kr * g.ldq + min(j0 + pp_qcol(u0, 0), lim)) * 2u;
      s.qh[c] = (uint32_t)(kr * g.ldq + min(j0 + pp_qcol(u0, 1), lim)) * 2u;
    }
  }
}

// one 1 KiB piece: base = wave-uniform operand pointer, kel = wave-uniform element offset of the K tile.  INLINE ASM in the
// SGPR-base + 32-bit-VGPR-offset form: the builtin made hipcc form a per-lane 64-bit address with two v_lshl_add_u64 per
// piece (zero-extended lane offset + K tile offset + operand base) - 32 two-pass VALU operations per K tile in the one
// instruction stream that also has to issue the MFMAs.  M0 (the LDS destination) is written in the same statement; no
// other LDS-DMA of this kernel goes through the builtin, so nothing of the compiler's lives in M0 across it.  The pieces
// are absent from hipcc's vmcnt bookkeeping: every wait on them is explicit (the prologue's and the odd steps').
#define W4_GLDS(base, so, c, kel, ldsoff)                                                                            \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                       \
               :: "v"((so)[c]), "s"(reinterpret_cast<const char*>((base) + (size_t)(kel))),                          \
                  "s"(lds0 + (uint32_t)((ldsoff) + ((c) * 4 + wave) * 1024)) : "memory")

// q fragment index f: 0..3 = QL fragments 0..3, 4..7 = QH fragments 0..3 of this wave's 64 unit columns.  Unit columns
// 0..31 / 32..63 of the wave are tile columns 0..31 / 64..95 (QL) and 32..63 / 96..127 (QH) of its 128: the 64-column
// block `blk` and the j fragment `a` inside it (the epilogue's accumulator layout [a][b], j = jb + 16 a + ...)
__device__ constexpr int w4_blk(int f) { return (f & 3) >> 1; }
__device__ constexpr int w4_a(int f) { return ((f >> 2) << 1) | (f & 1); }

#define W4_READ_P(DST, B0, b, KS) \
  DST[b] = pp_frag<false>(smem, (B0) + ((b) < 4 ? OFF_PL : OFF_PH), plb, (b) & 3, (KS))
#define W4_READ_Q(DST, B0, f, KS) \
  DST[f] = pp_frag<QT>(smem, (B0) + ((f) < 4 ? OFF_QL : OFF_QH), qlb, (f) & 3, (KS))

// DIAG (timing experiments, wrong results for 1 and 2): 1 = no LDS-DMA inside the K loop, 2 = no fragment reads inside it,
// 3 = the eight pieces of a step behind every SECOND group (all sixteen groups) instead of behind each of the first eight
template <bool QT, int DIAG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w4_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 64 KiB staging + 4 x 4 KiB epilogue windows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int ti, tj;
  pp_tile_ij(g, blockIdx.x, g.tiles_i * g.tiles_j, ti, tj);
  const int i0 = ti * 256, j0 = tj * 256;
  const int nt = g.K >> 6;
  const int kp = 64, kq = QT ? 64 * g.ldq : 64;                 // elements per K tile step
  const bf16* Pk = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qk = reinterpret_cast<const bf16*>(g.Q);
  W4Src src;
  w4_src<QT>(g, i0, j0, tid, src);
  const int plb = pp_lane_base<false, true>(lane, wr), qlb = pp_lane_base<QT, true>(lane, wc);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of the staging area

  f32x4 acc[2][2][4][4];            // [64-column block][i rows 0..63 / 64..127][j fragment a][i fragment b]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[x][y][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 pA[8], qA[8], pB[8], qB[8];

  // prologue: K tile 0 (P and Q units) into buffer 0, the P units of K tile 1 into buffer 1
#pragma unroll
  for (int c = 0; c < 4; ++c) { W4_GLDS(Pk, src.pl, c, 0, OFF_PL); W4_GLDS(Pk, src.ph, c, 0, OFF_PH); }
#pragma unroll
  for (int c = 0; c < 4; ++c) { W4_GLDS(Qk, src.ql, c, 0, OFF_QL); W4_GLDS(Qk, src.qh, c, 0, OFF_QH); }
  if (nt > 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { W4_GLDS(Pk, src.pl, c, kp, PPB + OFF_PL); W4_GLDS(Pk, src.ph, c, kp, PPB + OFF_PH); }
    PP_WAIT(8);
  } else {
    PP_WAIT(0);
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < 8; ++b) W4_READ_P(pA, 0, b, 0);
#pragma unroll
  for (int f = 0; f < 8; ++f) W4_READ_Q(qA, 0, f, 0);
  if (DIAG == 2) {
#pragma unroll
    for (int b = 0; b < 8; ++b) { W4_READ_P(pB, 0, b, 1); W4_READ_Q(qB, 0, b, 1); }
  }

  // 4 MFMAs of group gq on the current sets: q fragment gq >> 1, p fragments 4 (gq & 1) .. + 3.
  // The MFMA is INLINE ASM with the accumulator as a tied "+a" operand: the 64 accumulators of a 128 x 128 wave block are all
  // 256 AGPRs, and hipcc's allocator, left to itself, renames MFMA destinations - with no free AGPR it kept half of the
  // accumulators in VGPRs and moved them through a[20:23] around every MFMA (v_accvgpr_read / _write + s_nop 7 per MFMA in
  // the .s of the builtin form).  Tied operands pin each accumulator to its registers for the whole K loop; the A / B
  // fragments come from compiler-counted LDS reads (it waits for them ahead of the statement that names them) or from the
  // asm transposing reads behind the explicit lgkmcnt(0) of each step.
#define W4_MFMA4(PC, QC, gq)                                                                                        \
  _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) {                                                                \
    constexpr int f_ = (gq) >> 1;                                                                                    \
    const int b_ = ((gq) & 1) * 4 + bb;                                                                              \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                                           \
                 : "+a"(acc[w4_blk(f_)][((gq) & 1)][w4_a(f_)][bb]) : "v"(QC[f_]), "v"(PC[b_]));                       \
  }

  // One group = 4 MFMAs + its share of the step's memory work.  n1 / n2 ("K tile t+1 / t+2 exists") are LITERALS in the
  // steady-state loop below: a run-time flag around every piece is a scalar branch per group in the MFMA stream.
  // even step of K tile t (buffer B0): MFMAs on set A; reads (t, k-hi) -> set B; stages Q(t+1) into the other buffer
#define W4_G_EVEN(gq, B0, B1, t, n1)                                                                                 \
  W4_MFMA4(pA, qA, gq);                                                                                              \
  if (DIAG != 2) {                                                                                                   \
    if ((gq) < 8) W4_READ_P(pB, B0, (gq) & 7, 1);                                                                    \
    else if ((gq) < 12) { W4_READ_Q(qB, B0, ((gq) & 3) * 2, 1); W4_READ_Q(qB, B0, ((gq) & 3) * 2 + 1, 1); }          \
  }                                                                                                                  \
  if (DIAG != 1 && (n1)) {                                                                                           \
    constexpr int n_ = DIAG == 4 ? ((gq) < 4 ? 2 : 0) : DIAG == 5 ? ((gq) == 0 ? 8 : 0)                              \
                     : DIAG == 3 ? (((gq) & 1) == 0 ? 1 : 0) : ((gq) < 8 ? 1 : 0);                                   \
    constexpr int c0_ = DIAG == 4 ? 2 * (gq) : DIAG == 5 ? 0 : DIAG == 3 ? (gq) >> 1 : (gq);                         \
    _Pragma("unroll") for (int cc = 0; cc < n_; ++cc) {                                                              \
      const int c_ = c0_ + cc;                                                                                       \
      if (c_ < 4) W4_GLDS(Qk, src.ql, c_ & 3, ((t) + 1) * kq, B1 + OFF_QL);                                          \
      else W4_GLDS(Qk, src.qh, c_ & 3, ((t) + 1) * kq, B1 + OFF_QH);                                                 \
    }                                                                                                                \
  }                                                                                                                  \
  __builtin_amdgcn_sched_barrier(0)
#define W4_EVEN(BUF, t, n1)                                                                                          \
  do {                                                                                                               \
    constexpr int B0 = (BUF) * PPB, B1 = ((BUF) ^ 1) * PPB;                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* set A (asm transposing reads are not the compiler's) */ \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    W4_G_EVEN(0, B0, B1, t, n1); W4_G_EVEN(1, B0, B1, t, n1); W4_G_EVEN(2, B0, B1, t, n1); W4_G_EVEN(3, B0, B1, t, n1); \
    W4_G_EVEN(4, B0, B1, t, n1); W4_G_EVEN(5, B0, B1, t, n1); W4_G_EVEN(6, B0, B1, t, n1); W4_G_EVEN(7, B0, B1, t, n1); \
    W4_G_EVEN(8, B0, B1, t, n1); W4_G_EVEN(9, B0, B1, t, n1); W4_G_EVEN(10, B0, B1, t, n1); W4_G_EVEN(11, B0, B1, t, n1); \
    W4_G_EVEN(12, B0, B1, t, n1); W4_G_EVEN(13, B0, B1, t, n1); W4_G_EVEN(14, B0, B1, t, n1); W4_G_EVEN(15, B0, B1, t, n1); \
  } while (0)

  // odd step: barrier; MFMAs on set B; reads (t+1, k-lo) -> set A from the other buffer; stages P(t+2) into this one
#define W4_G_ODD(gq, B0, B1, t, n1, n2)                                                                              \
  W4_MFMA4(pB, qB, gq);                                                                                              \
  if (DIAG != 2 && (n1)) {                                                                                           \
    if ((gq) < 8) W4_READ_P(pA, B1, (gq) & 7, 0);                                                                    \
    else if ((gq) < 12) { W4_READ_Q(qA, B1, ((gq) & 3) * 2, 0); W4_READ_Q(qA, B1, ((gq) & 3) * 2 + 1, 0); }          \
  }                                                                                                                  \
  if (DIAG != 1 && (n2)) {                                                                                           \
    constexpr int n_ = DIAG == 4 ? ((gq) < 4 ? 2 : 0) : DIAG == 5 ? ((gq) == 0 ? 8 : 0)                              \
                     : DIAG == 3 ? (((gq) & 1) == 0 ? 1 : 0) : ((gq) < 8 ? 1 : 0);                                   \
    constexpr int c0_ = DIAG == 4 ? 2 * (gq) : DIAG == 5 ? 0 : DIAG == 3 ? (gq) >> 1 : (gq);                         \
    _Pragma("unroll") for (int cc = 0; cc < n_; ++cc) {                                                              \
      const int c_ = c0_ + cc;                                                                                       \
      if (c_ < 4) W4_GLDS(Pk, src.pl, c_ & 3, ((t) + 2) * kp, B0 + OFF_PL);                                          \
      else W4_GLDS(Pk, src.ph, c_ & 3, ((t) + 2) * kp, B0 + OFF_PH);                                                 \
    }                                                                                                                \
  }                                                                                                                  \
  __builtin_amdgcn_sched_barrier(0)
#define W4_ODD(BUF, t, n1, n2)                                                                                       \
  do {                                                                                                               \
    constexpr int B0 = (BUF) * PPB, B1 = ((BUF) ^ 1) * PPB;                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    W4_G_ODD(0, B0, B1, t, n1, n2); W4_G_ODD(1, B0, B1, t, n1, n2); W4_G_ODD(2, B0, B1, t, n1, n2);                  \
    W4_G_ODD(3, B0, B1, t, n1, n2); W4_G_ODD(4, B0, B1, t, n1, n2); W4_G_ODD(5, B0, B1, t, n1, n2);                  \
    W4_G_ODD(6, B0, B1, t, n1, n2); W4_G_ODD(7, B0, B1, t, n1, n2); W4_G_ODD(8, B0, B1, t, n1, n2);                  \
    W4_G_ODD(9, B0, B1, t, n1, n2); W4_G_ODD(10, B0, B1, t, n1, n2); W4_G_ODD(11, B0, B1, t, n1, n2);                \
    W4_G_ODD(12, B0, B1, t, n1, n2); W4_G_ODD(13, B0, B1, t, n1, n2); W4_G_ODD(14, B0, B1, t, n1, n2);               \
    W4_G_ODD(15, B0, B1, t, n1, n2);                                                                                 \
  } while (0)

  __builtin_amdgcn_sched_barrier(0);
  int t = 0;
  for (; t + 3 < nt; t += 2) {          // steady state: K tiles t+1 .. t+3 exist - no conditions in the stream
    W4_EVEN(0, t, true); W4_ODD(0, t, true, true);
    W4_EVEN(1, t + 1, true); W4_ODD(1, t + 1, true, true);
  }
  for (; t + 1 < nt; t += 2) {          // the last K tiles (at most three of them), run-time flags
    const bool a1 = t + 2 < nt, a2 = t + 3 < nt;
    W4_EVEN(0, t, true); W4_ODD(0, t, true, a1);
    W4_EVEN(1, t + 1, a1); W4_ODD(1, t + 1, a1, a2);
  }
  if (t < nt) { W4_EVEN(0, t, false); W4_ODD(0, t, false, false); }
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // the last MFMAs' results are read by compiler code (accvgpr reads) below
  __builtin_amdgcn_sched_barrier(0);

  // epilogue: the wave's two 128 x 64 blocks through its private 4 KiB LDS window (gemm_pp256_epi.h)
  const int ib = i0 + wr * 128, jb = j0 + wc * 128;
  const bool full = (i0 + 256 <= g.I) && (j0 + 256 <= g.J);
  char* swin = smem + W4_EPI_OFF + wave * 4096;
  if (full) {
    pp_epilogue<true, false, 1>(g, acc[0][0], acc[0][1], ib, jb, lane, swin);
    pp_epilogue<true, false, 1>(g, acc[1][0], acc[1][1], ib, jb + 64, lane, swin);
  } else {
    pp_epilogue<false, false, 1>(g, acc[0][0], acc[0][1], ib, jb, lane, swin);
    pp_epilogue<false, false, 1>(g, acc[1][0], acc[1][1], ib, jb + 64, lane, swin);
  }
}

bool evlm_gemm_pp256_eligible(const GemmP& g, int pt, int qt);

// opt-in while it is being measured (EVLM_W4=1): every bf16-output product the 256 x 256 family serves
bool evlm_gemm_w4_eligible(const GemmP& g, int pt, int qt) {
  static const int on = getenv("EVLM_W4") ? atoi(getenv("EVLM_W4")) : 0;
  if (!on || pt || g.c_f32 || g.accumulate || g.psum) return false;
  return evlm_gemm_pp256_eligible(g, pt, qt);
}

int evlm_gemm_w4_launch(GemmP& g, int qt, hipStream_t stream) {
  const int lds = 2 * PPB + 4 * 4096;
  g.tiles_i = ceil_div(g.I, 256); g.tiles_j = ceil_div(g.J, 256); g.bare_f32 = 0; g.sk = 0; g.kt_per_split = g.K / 64;
  const dim3 grid(g.tiles_i * g.tiles_j), block(256);
#define W4_LAUNCH(QT_)                                                                                        \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<QT_>),             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
      if (e != hipSuccess) return evlm_set_error("evlm_gemm: cannot reserve 144 KiB LDS: %s", hipGetErrorString(e)); \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_w4_kernel<QT_>), grid, block, lds, stream, g);                              \
  } while (0)
  static const int diag = getenv("EVLM_W4_DIAG") ? atoi(getenv("EVLM_W4_DIAG")) : 0;      // (timing experiments only)
  if (diag) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (diag == 1) hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, 1>), grid, block, lds, stream, g);
    else if (diag == 3) hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, 3>), grid, block, lds, stream, g);
    else if (diag == 4) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, 4>), grid, block, lds, stream, g); }
    else if (diag == 5) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<false, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, 5>), grid, block, lds, stream, g); }
    else hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, 2>), grid, block, lds, stream, g);
    return 0;
  }
  if (qt) W4_LAUNCH(true); else W4_LAUNCH(false);
#undef W4_LAUNCH
  return 0;
}
