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

// Tile flavours: NB = 8 -> 256 x 256 (wave block 128 x 128, units PL | PH | QL | QH, 64 KiB per buffer);
//                NB = 6 -> 192 x 256 (wave block  96 x 128, units PL | PX | QL | QH, 56 KiB per buffer: the unit geometry of
//                the 192-row ping-pong kernel) for launches whose 256-row tiles leave a quarter of the chip idle.
// PERSISTENT: gridDim.x = min(tiles, CUs); a workgroup walks tiles b, b + grid, ...; the next tile's first pieces are issued
// before the current tile's epilogue.

template <int NB> struct W4Geo {
  static constexpr int SB = NB == 8 ? 65536 : 57344;            // bytes per staging buffer
  static constexpr int O_PL = 0, O_PH = 16384;                  // PH (64 rows per wave row) or PX (32 rows per wave row)
  static constexpr int O_QL = NB == 8 ? 32768 : 24576, O_QH = O_QL + 16384;
  static constexpr int NPH = NB == 8 ? 4 : 2;                   // pieces of the second P unit per wave
  static constexpr int ROWS = NB * 32;                          // tile rows
  static constexpr int EPI = 2 * SB;                            // 4 x 4 KiB epilogue windows behind the buffers
};

struct W4Src { uint32_t pl[4], ph[4], ql[4], qh[4]; };      // per-lane source byte offsets of the pieces this lane's wave stages

template <bool QT, int NB>
__device__ __forceinline__ void w4_src(const GemmP& g, int i0, int j0, int tid, W4Src& s) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = (c * 4 + wave) * 64 + lane;          // 16-byte LDS slot of the unit, linear per wave instruction
    {
      const int u = id >> 3, cp = id & 7;
      const int koff = (cp ^ ((u >> 1) & 7)) << 3;
      if (NB == 8) {
        s.pl[c] = (uint32_t)(min(i0 + pp_prow(u, 0), g.I - 1) * g.ldp + koff) * 2u;
        s.ph[c] = (uint32_t)(min(i0 + pp_prow(u, 1), g.I - 1) * g.ldp + koff) * 2u;
      } else {                                          // wave row r: tile rows 96 r .. + 95 = PL rows 0..63, PX rows 64..95
        s.pl[c] = (uint32_t)(min(i0 + (u >> 6) * 96 + (u & 63), g.I - 1) * g.ldp + koff) * 2u;
        s.ph[c] = (uint32_t)(min(i0 + ((u & 63) >> 5) * 96 + 64 + (u & 31), g.I - 1) * g.ldp + koff) * 2u;   // (c < 2 used)
      }
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
// SGPR-base + 32-bit-VGPR-offset form (the builtin made hipcc form a per-lane 64-bit address with two v_lshl_add_u64 per
// piece).  M0 (the LDS destination) is written in the same statement; no LDS-DMA of this kernel goes through the builtin,
// so nothing of the compiler's lives in M0 across it.  The pieces are absent from hipcc's vmcnt bookkeeping: every wait on
// them is explicit.
#define W4_GLDS(base, so, c, kel, ldsoff)                                                                            \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                       \
               :: "v"((so)[c]), "s"(reinterpret_cast<const char*>((base) + (size_t)(kel))),                          \
                  "s"(lds0 + (uint32_t)((ldsoff) + ((c) * 4 + wave) * 1024)) : "memory")

// q fragment index f: 0..3 = QL fragments 0..3, 4..7 = QH fragments 0..3 of this wave's 64 unit columns.  Unit columns
// 0..31 / 32..63 of the wave are tile columns 0..31 / 64..95 (QL) and 32..63 / 96..127 (QH) of its 128: the 64-column
// block `blk` and the j fragment `a` inside it (the epilogue's accumulator layout [a][b], j = jb + 16 a + ...)
__device__ constexpr int w4_blk(int f) { return (f & 3) >> 1; }
__device__ constexpr int w4_a(int f) { return ((f >> 2) << 1) | (f & 1); }

// p fragment b: 0..3 out of PL (unit rows 64 wr + 16 b), 4.. out of PH (64 wr + 16 (b - 4)) or PX (32 wr + 16 (b - 4))
#define W4_READ_P(DST, B0, b, KS)                                                                                    \
  DST[b] = pp_frag<false>(smem, (B0) + ((b) < 4 ? GEO::O_PL : GEO::O_PH), ((b) < 4 || NB == 8) ? plb : pxb, (b) & 3, (KS))
#define W4_READ_Q(DST, B0, f, KS) \
  DST[f] = pp_frag<QT>(smem, (B0) + ((f) < 4 ? GEO::O_QL : GEO::O_QH), qlb, (f) & 3, (KS))

// the P / Q pieces of one K tile (kel: its element offset) into the buffer at B: piece index c 0..7 (P: 0..NPH+3)
#define W4_P_PIECE(c, kel, B)                                                                                        \
  do { if ((c) < 4) W4_GLDS(Pk, src.pl, (c) & 3, kel, (B) + GEO::O_PL);                                              \
       else if ((c) < 4 + GEO::NPH) W4_GLDS(Pk, src.ph, (c) & 3, kel, (B) + GEO::O_PH); } while (0)
#define W4_Q_PIECE(c, kel, B)                                                                                        \
  do { if ((c) < 4) W4_GLDS(Qk, src.ql, (c) & 3, kel, (B) + GEO::O_QL);                                              \
       else W4_GLDS(Qk, src.qh, (c) & 3, kel, (B) + GEO::O_QH); } while (0)

template <bool QT, int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w4_kernel(GemmP g) {
  using GEO = W4Geo<NB>;
  constexpr int SB = GEO::SB;
  constexpr int NM = NB / 2;                                    // MFMAs per group: q fragment gq >> 1 x p fragments NM (gq & 1) ..
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 staging buffers + 4 x 4 KiB epilogue windows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = g.tiles_i * g.tiles_j;
  const int nt = g.K >> 6;
  const int kp = 64, kq = QT ? 64 * g.ldq : 64;                 // elements per K tile step
  const bf16* Pk = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qk = reinterpret_cast<const bf16*>(g.Q);
  const int plb = pp_lane_base<false, true>(lane, wr), qlb = pp_lane_base<QT, true>(lane, wc);
  const int pxb = pp_lane_base<false, false>(lane, wr);         // PX: 32 unit rows per wave row
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of the staging area

  f32x4 acc[2][2][4][4];            // [64-column block][i rows 0..63 / 64..][j fragment a][i fragment b]
  bf16x8 pA[NB], qA[8], pB[NB], qB[8];

  // the first pieces of a tile: K tile 0 (P and Q units) into buffer 0, the P units of K tile 1 into buffer 1
#define W4_PROLOGUE()                                                                                                \
  do {                                                                                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) W4_P_PIECE(c, 0, 0);                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) W4_Q_PIECE(c, 0, 0);                                               \
    if (nt > 1) { _Pragma("unroll") for (int c = 0; c < 8; ++c) W4_P_PIECE(c, kp, SB); }                             \
  } while (0)

  // The MFMA is INLINE ASM with the accumulator as a tied "+a" operand: the accumulators of a 128 x 128 wave block are all
  // 256 AGPRs, and hipcc's allocator, left to itself, renames MFMA destinations - with no free AGPR it kept half of the
  // accumulators in VGPRs and moved them through a[20:23] around every MFMA (v_accvgpr_read / _write + s_nop 7 per MFMA in
  // the .s of the builtin form).  Tied operands pin each accumulator to its registers for the whole K loop; the A / B
  // fragments come from compiler-counted LDS reads (it waits for them ahead of the statement that names them) or from the
  // asm transposing reads behind the explicit lgkmcnt(0) of each step.  ZERO: the first step of a tile writes the
  // accumulators (C = 0) instead of zero-filling 256 registers beforehand.
#define W4_MFMAS(PC, QC, gq, ZERO)                                                                                  \
  _Pragma("unroll") for (int bb = 0; bb < NM; ++bb) {                                                               \
    constexpr int f_ = (gq) >> 1;                                                                                    \
    const int b_ = ((gq) & 1) * NM + bb;                                                                             \
    if (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0"                                                  \
                           : "=a"(acc[w4_blk(f_)][b_ >> 2][w4_a(f_)][b_ & 3]) : "v"(QC[f_]), "v"(PC[b_]));           \
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                                      \
                      : "+a"(acc[w4_blk(f_)][b_ >> 2][w4_a(f_)][b_ & 3]) : "v"(QC[f_]), "v"(PC[b_]));                \
  }

  // One group = NM MFMAs + its share of the step's memory work: p fragment reads behind groups 0 .. NB-1, q fragment reads
  // (two each) behind the next four, one staging piece behind each of the first eight.  n1 / n2 ("K tile t+1 / t+2
  // exists") are LITERALS in the steady-state loop: a run-time flag around every piece is a scalar branch per group in
  // the MFMA stream (measured: 8 % of the K loop).
  // even step of K tile t (buffer B0): MFMAs on set A; reads (t, k-hi) -> set B; stages Q(t+1) into the other buffer
#define W4_G_EVEN(gq, B0, B1, t, n1, ZERO)                                                                           \
  W4_MFMAS(pA, qA, gq, ZERO);                                                                                        \
  if ((gq) < NB) W4_READ_P(pB, B0, (gq) % NB, 1);                                                                    \
  else if ((gq) < NB + 4) { W4_READ_Q(qB, B0, (((gq) - NB) & 3) * 2, 1); W4_READ_Q(qB, B0, (((gq) - NB) & 3) * 2 + 1, 1); } \
  if ((gq) < 8 && (n1)) W4_Q_PIECE((gq) & 7, ((t) + 1) * kq, B1);                                                    \
  __builtin_amdgcn_sched_barrier(0)
#define W4_EVEN(BUF, t, n1, ZERO)                                                                                    \
  do {                                                                                                               \
    constexpr int B0 = (BUF) * SB, B1 = ((BUF) ^ 1) * SB;                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* set A (asm transposing reads are not the compiler's) */ \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    W4_G_EVEN(0, B0, B1, t, n1, ZERO); W4_G_EVEN(1, B0, B1, t, n1, ZERO); W4_G_EVEN(2, B0, B1, t, n1, ZERO);         \
    W4_G_EVEN(3, B0, B1, t, n1, ZERO); W4_G_EVEN(4, B0, B1, t, n1, ZERO); W4_G_EVEN(5, B0, B1, t, n1, ZERO);         \
    W4_G_EVEN(6, B0, B1, t, n1, ZERO); W4_G_EVEN(7, B0, B1, t, n1, ZERO); W4_G_EVEN(8, B0, B1, t, n1, ZERO);         \
    W4_G_EVEN(9, B0, B1, t, n1, ZERO); W4_G_EVEN(10, B0, B1, t, n1, ZERO); W4_G_EVEN(11, B0, B1, t, n1, ZERO);       \
    W4_G_EVEN(12, B0, B1, t, n1, ZERO); W4_G_EVEN(13, B0, B1, t, n1, ZERO); W4_G_EVEN(14, B0, B1, t, n1, ZERO);      \
    W4_G_EVEN(15, B0, B1, t, n1, ZERO);                                                                              \
  } while (0)

  // odd step: barrier; MFMAs on set B; reads (t+1, k-lo) -> set A from the other buffer; stages P(t+2) into this one
#define W4_G_ODD(gq, B0, B1, t, n1, n2)                                                                              \
  W4_MFMAS(pB, qB, gq, false);                                                                                       \
  if (n1) {                                                                                                          \
    if ((gq) < NB) W4_READ_P(pA, B1, (gq) % NB, 0);                                                                  \
    else if ((gq) < NB + 4) { W4_READ_Q(qA, B1, (((gq) - NB) & 3) * 2, 0); W4_READ_Q(qA, B1, (((gq) - NB) & 3) * 2 + 1, 0); } \
  }                                                                                                                  \
  if ((gq) < 8 && (n2)) W4_P_PIECE((gq) & 7, ((t) + 2) * kp, B0);                                                    \
  __builtin_amdgcn_sched_barrier(0)
#define W4_ODD(BUF, t, n1, n2)                                                                                       \
  do {                                                                                                               \
    constexpr int B0 = (BUF) * SB, B1 = ((BUF) ^ 1) * SB;                                                            \
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

  int vb = blockIdx.x;
  int ti, tj;
  pp_tile_ij(g, vb, ntiles, ti, tj);
  int i0 = ti * GEO::ROWS, j0 = tj * 256;
  W4Src src;
  w4_src<QT, NB>(g, i0, j0, tid, src);
  W4_PROLOGUE();
  while (true) {
    // this tile's first pieces were issued before the previous tile's epilogue (or just above): K tile 0 has landed once
    // everything but the eight youngest pieces (P of K tile 1) - and, one counter, the previous epilogue's stores - has
    if (nt > 1) { if (NB == 8) PP_WAIT(8); else PP_WAIT(6); } else PP_WAIT(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) W4_READ_P(pA, 0, b, 0);
#pragma unroll
    for (int f = 0; f < 8; ++f) W4_READ_Q(qA, 0, f, 0);
    __builtin_amdgcn_sched_barrier(0);
    // K tile 0 (its even step writes the accumulators), then the steady state, then at most three K tiles with run-time flags
    int t = 0;
    if (nt >= 4) {
      W4_EVEN(0, 0, true, true); W4_ODD(0, 0, true, true);
      W4_EVEN(1, 1, true, false); W4_ODD(1, 1, true, true);
      t = 2;
      for (; t + 3 < nt; t += 2) {        // K tiles t+1 .. t+3 exist - no conditions in the stream
        W4_EVEN(0, t, true, false); W4_ODD(0, t, true, true);
        W4_EVEN(1, t + 1, true, false); W4_ODD(1, t + 1, true, true);
      }
      for (; t + 1 < nt; t += 2) {
        const bool a1 = t + 2 < nt, a2 = t + 3 < nt;
        W4_EVEN(0, t, true, false); W4_ODD(0, t, true, a1);
        W4_EVEN(1, t + 1, a1, false); W4_ODD(1, t + 1, a1, a2);
      }
      if (t < nt) { W4_EVEN(0, t, false, false); W4_ODD(0, t, false, false); }
    } else {                              // two or three K tiles (the host guarantees K >= 128)
      const bool a1 = 2 < nt;
      W4_EVEN(0, 0, true, true); W4_ODD(0, 0, true, a1);
      W4_EVEN(1, 1, a1, false); W4_ODD(1, 1, a1, false);
      if (a1) { W4_EVEN(0, 2, false, false); W4_ODD(0, 2, false, false); }
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // the last MFMAs' results are read by compiler code (accvgpr reads) below
    __builtin_amdgcn_sched_barrier(0);

    const int ib = i0 + wr * (GEO::ROWS / 2), jb = j0 + wc * 128;
    const bool full = (i0 + GEO::ROWS <= g.I) && (j0 + 256 <= g.J);
    vb += gridDim.x;
    const bool more = vb < ntiles;
    if (more) {                           // next tile: its first pieces fly under this tile's epilogue (every LDS read of
      pp_tile_ij(g, vb, ntiles, ti, tj);  // the K loop has retired: the last odd step's wait + barrier)
      i0 = ti * GEO::ROWS; j0 = tj * 256;
      int tid_p = tid;                    // (no hoisting of the lane terms over the K loop)
      asm volatile("" : "+v"(tid_p));
      w4_src<QT, NB>(g, i0, j0, tid_p, src);
      W4_PROLOGUE();
    }
    // epilogue: the wave's two 64-column blocks through its private 4 KiB LDS window (gemm_pp256_epi.h)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    char* swin = smem + GEO::EPI + wave * 4096;
    constexpr int HI = NB == 8 ? 1 : 2;
    if (full) {
      pp_epilogue<true, false, HI>(g, acc[0][0], acc[0][1], ib, jb, lane_e, swin);
      pp_epilogue<true, false, HI>(g, acc[1][0], acc[1][1], ib, jb + 64, lane_e, swin);
    } else {
      pp_epilogue<false, false, HI>(g, acc[0][0], acc[0][1], ib, jb, lane_e, swin);
      pp_epilogue<false, false, HI>(g, acc[1][0], acc[1][1], ib, jb + 64, lane_e, swin);
    }
    if (!more) break;
  }
}

bool evlm_gemm_pp256_eligible(const GemmP& g, int pt, int qt);

// opt-in while it is being measured (EVLM_W4=1): every bf16-output product the 256 x 256 family serves
bool evlm_gemm_w4_eligible(const GemmP& g, int pt, int qt) {
  static const int on = getenv("EVLM_W4") ? atoi(getenv("EVLM_W4")) : 0;
  if (!on || pt || g.c_f32 || g.accumulate || g.psum) return false;
  return evlm_gemm_pp256_eligible(g, pt, qt);
}

// returns the tile rows used (256 / 192)
int evlm_gemm_w4_launch(GemmP& g, int qt, hipStream_t stream) {
  g.bare_f32 = 0; g.sk = 0; g.kt_per_split = g.K / 64;
  // tile flavour: 192-row tiles when they fill rounds better (the ViT's 12 608 x 768 outputs: 150 tiles -> 198)
  const int t256 = ceil_div(g.I, 256) * ceil_div(g.J, 256), t192 = ceil_div(g.I, 192) * ceil_div(g.J, 256);
  const int r256 = ceil_div(t256, 256), r192 = ceil_div(t192, 256);
  static const int force = getenv("EVLM_W4_ROWS") ? atoi(getenv("EVLM_W4_ROWS")) : 0;
  const bool use192 = force ? force == 192 : (r192 * 3 < r256 * 4 || (r192 == r256 && t192 > t256 && t256 % 256 != 0 && (t256 % 256) < 200 && r256 == 1));
  const int rows = use192 ? 192 : 256;
  g.tiles_i = ceil_div(g.I, rows); g.tiles_j = ceil_div(g.J, 256);
  const int tiles = g.tiles_i * g.tiles_j;
  const int lds = (use192 ? 2 * 57344 : 2 * 65536) + 4 * 4096;
  const dim3 grid(imin(tiles, 256)), block(256);
#define W4_LAUNCH(QT_, NB_)                                                                                   \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel<QT_, NB_>),        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
      if (e != hipSuccess) return evlm_set_error("evlm_gemm: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e)); \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_w4_kernel<QT_, NB_>), grid, block, lds, stream, g);                         \
  } while (0)
  if (use192) { if (qt) W4_LAUNCH(true, 6); else W4_LAUNCH(false, 6); }
  else { if (qt) W4_LAUNCH(true, 8); else W4_LAUNCH(false, 8); }
#undef W4_LAUNCH
  return rows;
}
