// 256x256x64 "ping-pong" bf16 GEMM for gfx950 (MI355X):  C[i,j] = epi( alpha * sum_k P(i,k) * Q(j,k) ).
// Operands are K-contiguous ([rows][K]: forward products x W^T) or reduction-major ([K][rows]: W in dX = dY W, both
// operands in dW = dY^T X); the latter are read from LDS with the transposing ds_read_b64_tr_b16, so no transposed copy
// ever exists in HBM.  bf16 output goes through the fused epilogue; weight gradients leave as f32 (plain stores, or f32
// atomics into the flat gradient slab when the reduction is split over workgroups / accumulated in place), with the
// bias gradient (row sums of dY^T) computed on the VALU from the fragments that are in registers anyway.
//
// Why a second kernel: with a 128x128 tile one K step needs as many cycles of the CU's vector-memory pipe (32 KiB of
// LDS-DMA at 64 B/clk) as of MFMA issue, so that kernel tops out near 60 % MFMA utilisation inside its K loop.  A
// 256x256 tile halves the bytes per flop; what is left is to keep the matrix cores fed from ONE resident workgroup:
//
//   * 8 waves = 2 groups (wr) x 4 (wc); a wave owns 128 (i) x 64 (j) of C = 8 x 4 accumulators of 16x16 (128 VGPRs).
//     SIMD s hosts wave s of group 0 and wave s of group 1.
//   * The groups run ONE BARRIER APART (group 1 takes an extra s_barrier on entry, group 0 one on exit): every phase is
//     [LDS reads + LDS-DMA issue] barrier [16 MFMAs] barrier, so while one group's wave owns the SIMD's matrix core the
//     other group's wave on that SIMD issues its memory work - the MFMA pipe alternates between the two waves.
//   * One K tile (64 deep) = 4 phases = the four 64x32 quadrants of the wave's C block (i-lo x j-lo, i-lo x j-hi,
//     i-hi x j-hi, i-hi x j-lo): phase 0 reads the i-lo P fragments and the j-lo Q fragments, phase 1 the j-hi Q
//     fragments (over the j-lo registers), phase 2 the i-hi P fragments (over the i-lo registers), phase 3 the j-lo Q
//     fragments again.
//   * LDS: 2 buffers x 4 units of 16 KiB.  A unit is what ONE phase consumes - PL / PH: the i-lo / i-hi 64 rows of both
//     wave groups, QL / QH: the j-lo / j-hi 32 columns of all four wave columns - as [128 rows][64 k] with the 16-byte
//     chunks XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128).  It is filled by 2 global_load_lds_dwordx4 per
//     wave (the swizzle is applied to the per-lane SOURCE address; the LDS image of a wave instruction is linear).
//   * Staging order in time is PL(t) QH(t) PH(t) QL(t) PL(t+1) ... one unit per phase: phase 0 of K tile t issues PH(t+1),
//     phase 1 QL(t+1), phase 2 PL(t+2), phase 3 QH(t+2) - each into the region whose last reader finished at least one
//     full phase earlier (WAR: a unit read in phase r is free after both groups' reads retired, i.e. from phase r+2;
//     QL is read in phases 0 AND 3) and 3-6 phases before its consumer (the short lead is QL's: the weight operand,
//     L2-resident).  RAW: every wave waits a counted `vmcnt` (10, 10, -, 4: the younger units stay in flight) in the
//     phase BEFORE the consuming one, ahead of that phase's first barrier, so both groups have passed a barrier that
//     follows every wave's wait before either reads the unit.
//
// Persistent: one workgroup per CU walks tiles b, b + grid, ...; the next tile's first six units are issued before the
// current tile's epilogue (see below), so neither the launch nor the first HBM round trip is paid per tile.
#include <stdlib.h>
#include <string.h>
#include "gemm_common.h"

#include "gemm_pp256_core.h"

#include "gemm_pp256_epi.h"

// Grouped launch (weight gradients of many layers in ONE launch, no split-K): up to PP_MAXG problems that share the
// reduction length K; item ids run over the concatenated tile lists (tile0 = prefix sums).
#define PP_MAXG 40
struct PPGroup {
  int n;
  int dup[PP_MAXG];                // 1: this problem's C is written by another problem of the launch too: f32 atomics
  int tile0[PP_MAXG + 1];
  const void* P[PP_MAXG]; const void* Q[PP_MAXG]; void* C[PP_MAXG]; float* psum[PP_MAXG];
  int I[PP_MAXG], J[PP_MAXG], ldp[PP_MAXG], ldq[PP_MAXG], ldc[PP_MAXG];
  int assign[PP_MAXG];             // 1: C = tile (this step's first contribution: no zero-fill before, no read here)
};

// select the problem item `vb0` belongs to; returns the tile id inside it.  Item ids are first remapped so that the
// workgroups of one XCD (ids = x mod 8) walk a CONTIGUOUS range of the concatenated tile lists: the 32 tiles an XCD has in
// flight then belong to one or two problems and share their dY / X panels through that XCD's L2 (without the remap every
// XCD touches every panel: 3.9 GB fetched per launch, the kernel ran at the HBM / Infinity-Cache rate)
__device__ __forceinline__ int pp_group_select(const PPGroup& grp, int vb0, GemmP& g) {
  const int nit = grp.tile0[grp.n];
  const int q = nit >> 3, r = nit & 7, xcd = vb0 & 7;
  const int vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb0 >> 3);
  int p = 0;
  while (p + 1 < grp.n && vb >= grp.tile0[p + 1]) ++p;
  g.P = grp.P[p]; g.Q = grp.Q[p]; g.C = grp.C[p]; g.psum = grp.psum[p];
  g.I = grp.I[p]; g.J = grp.J[p]; g.ldp = grp.ldp[p]; g.ldq = grp.ldq[p]; g.ldc = grp.ldc[p];
  g.tiles_i = (g.I + 255) >> 8; g.tiles_j = (g.J + 255) >> 8;
  g.accumulate = grp.dup[p] ? 2 : (grp.assign[p] ? 0 : 1);      // 0: C = tile, 1: C += tile (load / add / store), 2: atomics
  return vb - grp.tile0[p];
}

// persistent: gridDim.x = min(work items, CUs); workgroup b runs items b, b + grid, ...  An item is (K split, tile);
// split s covers K tiles [s * kt_per_split, ...).  The next item's first six staging units are issued BEFORE the epilogue
// of the current one.   OUT 0: bf16 through the fused epilogue;  OUT 1: bare f32 (weight gradients).
//
// SK (stream-K, bf16 output): a launch whose tiles fill only part of one round of the 256 workgroups is cut along K
// instead.  The tiles are first dealt to the 8 XCDs in contiguous runs (as pp_tile_ij does); inside an XCD the (tile, K
// tile) iterations are laid end to end and split evenly - in even chunks of sk_q K tiles - over its 32 workgroups
// (blockIdx.x = xcd + 8 m).  A workgroup whose chunk ends inside a tile stores its f32 accumulators in ITS slot of the
// workspace (lane-linear: 32 coalesced 16-byte stores per lane) and raises ITS flag; the workgroup that finishes the
// tile's last K tile - always a later one of the same XCD, so one that was dispatched after its producers - waits for
// them in ascending order, adds their slots to its registers, clears their flags and runs the normal fused epilogue.
// Sums are formed in a fixed order (deterministic), nothing is zero-filled between launches.
#define PP_SK_FLAGS_BYTES 4096
#define PP_SK_SLOT_F4 (32 * 512)                    /* float4 per slot: 32 accumulator registers x 512 lanes */
// this workgroup's share under stream-K: first tile of its XCD, iterations per workgroup (even), end of its range.
// Recomputed where needed instead of being carried through the K loop (every scalar held there costs: the kernel is
// at the SGPR limit, and spilled SGPRs live in VGPR lanes)
struct PPSkRange { int t0, q, end; };
__device__ __forceinline__ PPSkRange pp_sk_range(int ntiles, int nt_all) {
  const int qt = ntiles >> 3, r = ntiles & 7, x = blockIdx.x & 7, m = blockIdx.x >> 3;
  PPSkRange o;
  o.t0 = x < r ? x * (qt + 1) : r * (qt + 1) + (x - r) * qt;
  const int tot = (x < r ? qt + 1 : qt) * nt_all;
  o.q = (((tot + 31) >> 5) + 1) & ~1;             // even: every segment holds >= 2 K tiles (nt_all is even)
  o.end = min((m + 1) * o.q, tot);
  return o;
}

template <bool PT, bool QT, int OUT, bool GROUPED, bool SK = false>
__device__ __forceinline__ void pp256_body(GemmP& g, const PPGroup* grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 64 KiB staging + 8 x 4 KiB epilogue windows; ALL LDS
  // (the wave index as a SCALAR: the LDS destinations of the staging DMA (m0) and the wave-group branches then cost no
  // VGPRs - every kernel of this file compiles without a spill; GD step -0.27 ms in a same-box A/B)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nt_all = g.K >> 6;
  const int splits = GROUPED ? 1 : (nt_all + g.kt_per_split - 1) / g.kt_per_split;
  int vb = blockIdx.x;
  int ntiles, nitems, ti, tj, sp = 0;
  int sk_it = 0, sk_k0 = 0;                          // SK: cursor in the XCD's iteration space, first K tile of the segment
  if (SK) {
    ntiles = g.tiles_i * g.tiles_j;
    nitems = 0;
    const PPSkRange rg = pp_sk_range(ntiles, nt_all);
    sk_it = (int)(blockIdx.x >> 3) * rg.q;
    if (sk_it >= rg.end) return;                      // (whole workgroup, before any barrier)
    const int tl_ = sk_it / nt_all;
    sk_k0 = sk_it - tl_ * nt_all;
    ti = (rg.t0 + tl_) / g.tiles_j; tj = (rg.t0 + tl_) - ti * g.tiles_j;
  } else if (GROUPED) {
    nitems = grp->tile0[grp->n];
    const int t = pp_group_select(*grp, vb, g);
    ntiles = g.tiles_i * g.tiles_j;
    ti = t / g.tiles_j; tj = t - ti * g.tiles_j;      // (each XCD's consecutive ids share a P row panel)
  } else {
    ntiles = g.tiles_i * g.tiles_j;
    nitems = ntiles * splits;
    sp = vb / ntiles;
    pp_tile_ij(g, vb - sp * ntiles, ntiles, ti, tj);
  }
  int kp = PT ? 64 * g.ldp : 64, kq = QT ? 64 * g.ldq : 64;           // elements per K tile step
  int i0 = ti * 256, j0 = tj * 256;
  int nt = SK ? min(nt_all - sk_k0, pp_sk_range(g.tiles_i * g.tiles_j, nt_all).end - sk_it)
              : min(nt_all - sp * g.kt_per_split, g.kt_per_split);
  const bf16* Pk = reinterpret_cast<const bf16*>(g.P) + (size_t)(SK ? sk_k0 : sp * g.kt_per_split) * kp;
  const bf16* Qk = reinterpret_cast<const bf16*>(g.Q) + (size_t)(SK ? sk_k0 : sp * g.kt_per_split) * kq;
  PPSrc src;
  pp_src<PT, QT>(g, i0, j0, tid, src);
  int plb = pp_lane_base<PT, true>(lane, wr), qlb = pp_lane_base<QT, false>(lane, wc);

  f32x4 accL[4][4], accH[4][4];     // [j fragment][i fragment]; L: i rows 0..63 of the wave's block, H: 64..127
  bf16x8 pf[4][2], qf[2][2];
  bf16x8 ql[2][2], qh[2][2];        // PP_KTILE_HOLD only
  float ps[2];                      // OUT 1: per-lane partial row sums of P (bias gradient), i fragments 2 wc, 2 wc + 1

  // prologue (host guarantees >= 2 K tiles per item): six units in flight, in the steady-state issue order of the schedule
  constexpr bool HOLD = OUT == 1;                 // weight gradients: PP_KTILE_HOLD (see there)
  if (HOLD) {
    PP_GLDS(Pk, src.pl, 0, OFF_PL); PP_GLDS(Qk, src.ql, 0, OFF_QL); PP_GLDS(Qk, src.qh, 0, OFF_QH);
    PP_GLDS(Pk, src.ph, 0, OFF_PH); PP_GLDS(Pk, src.pl, kp, PPB + OFF_PL); PP_GLDS(Qk, src.ql, kq, PPB + OFF_QL);
    PP_WAIT(8);                                   // PL0, QL0 have landed (this wave's share)
  } else {
    PP_GLDS(Pk, src.pl, 0, OFF_PL); PP_GLDS(Qk, src.qh, 0, OFF_QH); PP_GLDS(Pk, src.ph, 0, OFF_PH);
    PP_GLDS(Qk, src.ql, 0, OFF_QL); PP_GLDS(Pk, src.pl, kp, PPB + OFF_PL); PP_GLDS(Qk, src.qh, kq, PPB + OFF_QH);
    PP_WAIT(4);                                   // PL0, QL0 have landed
  }
#ifdef PP_STAMP
  unsigned long long stp[5]; int stn = 0;
  stp[4] = __builtin_amdgcn_s_memtime();
#endif
  while (true) {
    if (SK) {
      // the per-lane K-loop values (LDS fragment bases, staging source offsets) are RE-DERIVED here from a laundered thread
      // id: kept live across the stream-K epilogue (64 more registers than the plain one) they were the allocator's
      // spill candidates, and ONE scratch reload inside the K loop (its s_waitcnt vmcnt(0) drains the whole staging
      // pipeline) costs more than the stream-K cut saves
      int tid_k = tid;
      asm volatile("" : "+v"(tid_k));
      plb = pp_lane_base<PT, true>(tid_k & 63, wr);
      qlb = pp_lane_base<QT, false>(tid_k & 63, wc);
      pp_src<PT, QT>(g, i0, j0, tid_k, src);
    }
#ifdef PP_STAMP
    stp[0] = __builtin_amdgcn_s_memtime();
#endif
#ifdef PP_STAMP
    const bool do_psum = false;
#else
    const bool do_psum = OUT == 1 && g.psum != nullptr && tj == 0;
#endif
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) { accL[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; accH[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    ps[0] = ps[1] = 0.f;
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
    // (steady state: K tiles t+1 .. t+3 exist, no run-time flags in the stream; then at most three K tiles with flags)
    if (HOLD) {
      for (; t + 3 < nt; t += 2) {
        PP_KTILE_HOLD(0, t, true, true);
        PP_KTILE_HOLD(1, t + 1, true, true);
      }
      for (; t + 1 < nt; t += 2) {
        PP_KTILE_HOLD(0, t, true, t + 2 < nt);
        PP_KTILE_HOLD(1, t + 1, t + 2 < nt, t + 3 < nt);
      }
      if (t < nt) PP_KTILE_HOLD(0, t, false, false);
    } else {
      for (; t + 3 < nt; t += 2) {
        PP_KTILE(0, t, true, true);
        PP_KTILE(1, t + 1, true, true);
      }
      for (; t + 1 < nt; t += 2) {
        PP_KTILE(0, t, true, t + 2 < nt);
        PP_KTILE(1, t + 1, t + 2 < nt, t + 3 < nt);
      }
      if (t < nt) PP_KTILE(0, t, false, false);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();    // re-align the groups: every LDS read of the K loop has retired
    __builtin_amdgcn_sched_barrier(0);
#ifdef PP_STAMP
    stp[1] = __builtin_amdgcn_s_memtime();
#endif
    const int ib = i0 + wr * 128, jb = j0 + wc * 64;
    const bool full = (i0 + 256 <= g.I) && (j0 + 256 <= g.J);
    GemmP gc = g;                                 // the finished item's problem (the grouped form switches g below)
    const int seg_k0 = sk_k0, seg_nt = nt, seg_first_it = sk_it - sk_k0;     // SK: the finished segment
    const PPSkRange rg = SK ? pp_sk_range(ntiles, nt_all) : PPSkRange{0, 2, 0};
    bool more;
    if (SK) {
      sk_it += nt;
      more = sk_it < rg.end;
    } else {
      vb += gridDim.x;
      more = vb < nitems;
    }
    if (more) {                                   // next item: its first six units fly under this item's epilogue
      if (SK) {                                   // (a later segment of a workgroup always starts a tile)
        const int tl_ = sk_it / nt_all;
        sk_k0 = 0;
        ti = (rg.t0 + tl_) / g.tiles_j; tj = (rg.t0 + tl_) - ti * g.tiles_j;
      } else if (GROUPED) {
        const int t = pp_group_select(*grp, vb, g);
        ti = t / g.tiles_j; tj = t - ti * g.tiles_j;
        kp = PT ? 64 * g.ldp : 64; kq = QT ? 64 * g.ldq : 64;
      } else {
        sp = vb / ntiles;
        pp_tile_ij(g, vb - sp * ntiles, ntiles, ti, tj);
      }
      i0 = ti * 256; j0 = tj * 256;
      nt = SK ? min(nt_all, rg.end - sk_it) : min(nt_all - sp * g.kt_per_split, g.kt_per_split);
      Pk = reinterpret_cast<const bf16*>(g.P) + (size_t)(SK ? 0 : sp * g.kt_per_split) * kp;
      Qk = reinterpret_cast<const bf16*>(g.Q) + (size_t)(SK ? 0 : sp * g.kt_per_split) * kq;
      int tid_p = tid;                          // (same reason as for the epilogue below: no hoisting of the lane terms)
      asm volatile("" : "+v"(tid_p));
      pp_src<PT, QT>(g, i0, j0, tid_p, src);
      if (HOLD) {
        PP_GLDS(Pk, src.pl, 0, OFF_PL); PP_GLDS(Qk, src.ql, 0, OFF_QL); PP_GLDS(Qk, src.qh, 0, OFF_QH);
        PP_GLDS(Pk, src.ph, 0, OFF_PH); PP_GLDS(Pk, src.pl, kp, PPB + OFF_PL); PP_GLDS(Qk, src.ql, kq, PPB + OFF_QL);
      } else {
        PP_GLDS(Pk, src.pl, 0, OFF_PL); PP_GLDS(Qk, src.qh, 0, OFF_QH); PP_GLDS(Pk, src.ph, 0, OFF_PH);
        PP_GLDS(Qk, src.ql, 0, OFF_QL); PP_GLDS(Pk, src.pl, kp, PPB + OFF_PL); PP_GLDS(Qk, src.qh, kq, PPB + OFF_QH);
      }
    }
#ifdef PP_STAMP
    stp[2] = __builtin_amdgcn_s_memtime();
#endif
    // the epilogue's per-lane address arithmetic must NOT be hoisted out of the persistent loop (it would sit in VGPRs -
    // in scratch, in practice - through every K loop): launder the thread id it is derived from
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    char* swin_e = smem + PP_EPI_OFF + (tid_e >> 6) * 4096;
    bool sk_partial = false;
    if (SK) {
      int* flags = reinterpret_cast<int*>(gc.sk_ws);
      // The partial sums and flags move with SYSTEM-scope accesses (sc0 sc1: written through to memory, read past L1 and
      // L2), not with agent-scope fences: a release / acquire pair at agent scope writes back and invalidates the whole
      // L2 under every other workgroup's operand panels (measured: 10x on the launch).  Slot layout: [wave][register]
      // [lane] x 16 bytes, so four registers of a lane sit within one 13-bit instruction offset of each other; the
      // accesses are hand-written in groups of 4 / 8 (the register allocator otherwise spills INTO THE K LOOP).
      // Slot layout: [wave][half L/H][b][a][lane] x 16 bytes (a lane's four a-registers of one b sit within one 13-bit
      // instruction offset).  Partial sums and flags move with SYSTEM-scope accesses (sc0 sc1: written through to memory,
      // read past L1 and L2), not with agent-scope fences: a release / acquire pair at agent scope writes back and
      // invalidates the whole L2 under every other workgroup's operand panels (measured: 10x on the launch).
      char* lane_base = reinterpret_cast<char*>(gc.sk_ws) + PP_SK_FLAGS_BYTES + (size_t)(tid_e >> 6) * 32768 + lane_e * 16;
      if (seg_k0 + seg_nt < nt_all) {             // the tile ends in a later workgroup: park the accumulators, raise the flag
        sk_partial = true;
        char* A = lane_base + (size_t)blockIdx.x * 262144;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\t"
                       "global_store_dwordx4 %0, %2, off offset:1024 sc0 sc1\n\t"
                       "global_store_dwordx4 %0, %3, off offset:2048 sc0 sc1\n\t"
                       "global_store_dwordx4 %0, %4, off offset:3072 sc0 sc1"
                       :: "v"(A + b * 4096), "v"(accL[0][b]), "v"(accL[1][b]), "v"(accL[2][b]), "v"(accL[3][b]) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\t"
                       "global_store_dwordx4 %0, %2, off offset:1024 sc0 sc1\n\t"
                       "global_store_dwordx4 %0, %3, off offset:2048 sc0 sc1\n\t"
                       "global_store_dwordx4 %0, %4, off offset:3072 sc0 sc1"
                       :: "v"(A + 16384 + b * 4096), "v"(accH[0][b]), "v"(accH[1][b]), "v"(accH[2][b]), "v"(accH[3][b]) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are acknowledged
        __syncthreads();
        if (tid_e == 0) __hip_atomic_store(flags + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else if (seg_k0 > 0) {                    // owner of a tile other workgroups started: wait for their parts
        const int m_first = seg_first_it / rg.q, m_own = (int)(blockIdx.x >> 3);
        if (tid_e == 0) {
          for (int m = m_first; m < m_own; ++m) {
            int spins = 0;
            while (__hip_atomic_load(flags + (blockIdx.x & 7) + 8 * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > (1 << 22)) break;     // (a producer that never arrives: wrong numbers rather than a hung GPU)
            }
          }
        }
        __syncthreads();
        PPSk skp;
        skp.p0 = lane_base + (size_t)((blockIdx.x & 7) + 8 * m_first) * 262144;
        skp.np = m_own - m_first;
        if (full) pp_epilogue<true, true>(gc, accL, accH, ib, jb, lane_e, swin_e, skp);
        else pp_epilogue<false, true>(gc, accL, accH, ib, jb, lane_e, swin_e, skp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                          // every wave has read the slots: hand them back
        if (tid_e == 0)
          for (int m = m_first; m < m_own; ++m)
            __hip_atomic_store(flags + (blockIdx.x & 7) + 8 * m, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sk_partial = true;                        // (= nothing left to do below)
      }
    }
    if (OUT == 0) {
      if (sk_partial) {}
      else if (full) pp_epilogue<true>(gc, accL, accH, ib, jb, lane_e, swin_e);
      else pp_epilogue<false>(gc, accL, accH, ib, jb, lane_e, swin_e);
    } else {
      const int mode = GROUPED ? (gc.accumulate == 2 ? 1 : (gc.accumulate ? 2 : 0)) : ((gc.accumulate || splits > 1) ? 1 : 0);
      if (full) { pp_epi_f32_half<true>(gc, accL, ib, jb, lane_e, swin_e, mode); pp_epi_f32_half<true>(gc, accH, ib + 64, jb, lane_e, swin_e, mode); }
      else { pp_epi_f32_half<false>(gc, accL, ib, jb, lane_e, swin_e, mode); pp_epi_f32_half<false>(gc, accH, ib + 64, jb, lane_e, swin_e, mode); }
      if (do_psum) {                              // lanes l, l+16, l+32, l+48 hold the four k quarters of row l
#pragma unroll
        for (int b = 0; b < 2; ++b) {             // (this wave's share: i fragments 2 wc + b, see PP_PSUM)
          float v = ps[b];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          const int i = ib + (2 * wc + b) * 16 + lane_e;
          if (lane_e < 16 && i < gc.I) atomicAdd(gc.psum + i, v);
        }
      }
    }
#ifdef PP_STAMP
    stp[3] = __builtin_amdgcn_s_memtime();
    PP_WAIT(0);
    if (tid == 0 && g.psum) {     // [start, k-loop end, next prologue issued, epilogue issued, all retired, kernel entry]
      unsigned long long* o = reinterpret_cast<unsigned long long*>(g.psum) + ((size_t)blockIdx.x * 4 + stn) * 6;
      o[0] = stp[0]; o[1] = stp[1]; o[2] = stp[2]; o[3] = stp[3]; o[4] = __builtin_amdgcn_s_memtime(); o[5] = stp[4];
    }
    ++stn;
#endif
    if (!more) break;
    PP_WAIT(0);                                   // the six units (and this item's stores: one counter) have retired
  }
}

// =============================================================================================
// 128 x 256 tile flavour (gemm_bf16_pp128_kernel): for products whose 256 x 256 tiles would fill well under half of one
// round of the chip (the 7 680-row text-side products x 768 columns: 90 tiles on 256 CUs).  Same eight waves in two
// groups one barrier apart, same staging units and fragment reads, but a wave owns a 64 x 64 block: a K tile is TWO phases
// (i x j-lo, i x j-hi), its three 16 KiB units (P: 128 rows; QL, QH) sit in one of THREE 48 KiB stages, each unit issued two
// K tiles ahead (a four-phase lead).  One tile per workgroup (no persistence), so the epilogue windows alias stage 0.
//   issue order   PL(t+2) QL(t+2) in phase A of K tile t, QH(t+2) in phase B
//   waits         phase A: QH(t) landed   -> vmcnt(10)  [PL QL QH(t+1), PL QL(t+2) may be in flight];  6 / 0 in the tail
//                 phase B: PL QL(t+1)     -> vmcnt(8)   [QH(t+1), PL QL QH(t+2)];                      2 in the tail
// =============================================================================================
#define PPH_STAGE (3 * PPU)
#define HOFF_PL 0
#define HOFF_QL PPU
#define HOFF_QH (2 * PPU)
template <bool QT>
__device__ __forceinline__ void pp_src_half(const GemmP& g, int i0, int j0, int tid, int wave, PPSrc& s) {
  const int lane = tid & 63;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = (c * 8 + wave) * 64 + lane;
    {
      const int u = id >> 3, cp = id & 7;
      const int koff = (cp ^ ((u >> 1) & 7)) << 3;
      s.pl[c] = (uint32_t)(min(i0 + u, g.I - 1) * g.ldp + koff) * 2u;       // unit row u IS tile row u
      s.ph[c] = 0;
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

#define PPH_KTILE(S, t, N1, N2)                                                                                  \
  do {                                                                                                     \
    constexpr int B0 = (S) * PPH_STAGE, B2 = (((S) + 2) % 3) * PPH_STAGE;                                  \
    const bool n1 = (N1), n2 = (N2);                                                                              \
    /* ---- phase A: i x j-lo ; stage PL(t+2), QL(t+2) ---- */                                             \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qf[a][0] = pp_frag<QT>(smem, B0 + HOFF_QL, qlb, a, 0);                                               \
      qf[a][1] = pp_frag<QT>(smem, B0 + HOFF_QL, qlb, a, 1); }                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
      pf[b][0] = pp_frag<false>(smem, B0 + HOFF_PL, plb, b, 0);                                            \
      pf[b][1] = pp_frag<false>(smem, B0 + HOFF_PL, plb, b, 1); }                                          \
    if (n2) { PP_GLDS(Pk, src.pl, ((t) + 2) * kp, B2 + HOFF_PL); PP_GLDS(Qk, src.ql, ((t) + 2) * kq, B2 + HOFF_QL); } \
    if (n2) PP_WAIT(10); else if (n1) PP_WAIT(6); else PP_WAIT(0);                                         \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 0, qf); PP_MFMA_END();                                                  \
    /* ---- phase B: i x j-hi ; stage QH(t+2) ---- */                                                      \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qf[a][0] = pp_frag<QT>(smem, B0 + HOFF_QH, qlb, a, 0);                                               \
      qf[a][1] = pp_frag<QT>(smem, B0 + HOFF_QH, qlb, a, 1); }                                             \
    if (n2) PP_GLDS(Qk, src.qh, ((t) + 2) * kq, B2 + HOFF_QH);                                             \
    if (n2) PP_WAIT(8); else if (n1) PP_WAIT(2);                                                           \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 2, qf); PP_MFMA_END();                                                  \
  } while (0)

template <bool QT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp128_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 3 stages x 48 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nt = g.K >> 6;
  int ti, tj;
  pp_tile_ij(g, blockIdx.x, g.tiles_i * g.tiles_j, ti, tj);
  const int i0 = ti * 128, j0 = tj * 256;
  const int kp = 64, kq = QT ? 64 * g.ldq : 64;
  const bf16* Pk = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qk = reinterpret_cast<const bf16*>(g.Q);
  PPSrc src;
  pp_src_half<QT>(g, i0, j0, tid, wave, src);
  const int plb = pp_lane_base<false, true>(lane, wr), qlb = pp_lane_base<QT, false>(lane, wc);
  f32x4 accL[4][4], accH[4][4];
  bf16x8 pf[4][2], qf[2][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) accL[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // prologue: K tiles 0 and 1 (the host guarantees nt >= 2)
  PP_GLDS(Pk, src.pl, 0, HOFF_PL); PP_GLDS(Qk, src.ql, 0, HOFF_QL); PP_GLDS(Qk, src.qh, 0, HOFF_QH);
  PP_GLDS(Pk, src.pl, kp, PPH_STAGE + HOFF_PL); PP_GLDS(Qk, src.ql, kq, PPH_STAGE + HOFF_QL);
  PP_GLDS(Qk, src.qh, kq, PPH_STAGE + HOFF_QH);
  PP_WAIT(8);                                   // PL0, QL0 have landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);
  int t = 0;
  for (; t + 4 < nt; t += 3) {                    // steady state: K tiles up to t + 4 exist, no run-time flags
    PPH_KTILE(0, t, true, true);
    PPH_KTILE(1, t + 1, true, true);
    PPH_KTILE(2, t + 2, true, true);
  }
  for (; t + 2 < nt; t += 3) {
    PPH_KTILE(0, t, true, true);
    PPH_KTILE(1, t + 1, true, t + 3 < nt);
    PPH_KTILE(2, t + 2, t + 3 < nt, t + 4 < nt);
  }
  if (t < nt) PPH_KTILE(0, t, t + 1 < nt, false);
  if (t + 1 < nt) PPH_KTILE(1, t + 1, false, false);
  if (wr == 0) __builtin_amdgcn_s_barrier();    // re-align the groups: every LDS read of the K loop has retired
  __builtin_amdgcn_sched_barrier(0);
  const int ib = i0 + wr * 64, jb = j0 + wc * 64;
  const bool full = (i0 + 128 <= g.I) && (j0 + 256 <= g.J);
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63;
  char* swin = smem + wave * 4096;              // (stage 0 is dead: one tile per workgroup)
  if (full) pp_epilogue<true, false, 0>(g, accL, accH, ib, jb, lane_e, swin);
  else pp_epilogue<false, false, 0>(g, accL, accH, ib, jb, lane_e, swin);
}

// =============================================================================================
// 192 x 256 tile flavour (gemm_bf16_pp192_kernel): for products whose 256 x 256 tiles fill 50-80 % of ONE round (the ViT's
// 12 608 rows x 768 columns: 150 tiles -> 198; the 3 840-row text pass x 3 072 / 2 304: 180 / 135 -> 240 / 180).  A wave
// owns a 96 x 64 block, a K tile is THREE phases of 16 MFMAs:  A  i[0..63] x j-lo,  B  i[0..63] x j-hi,  C  i[64..95] x
// (j-lo, j-hi: both Q fragment sets are held from A / B).  Units per K tile: PL (2 x 64 rows, 16 KiB), PX (2 x 32 rows,
// 8 KiB), QL, QH: 56 KiB, two stages.  Every unit is issued two K tiles - four phases - ahead, into the buffer its
// predecessor was read out of one phase earlier:
//   A(t): PX(t+1)            wait QH(t)          -> vmcnt(8)   [PX(t), PL QL QH(t+1), PX(t+1) may be in flight]
//   B(t): PL(t+2), QL(t+2)   wait PX(t)          -> vmcnt(11)  [PL QL QH PX(t+1), PL QL(t+2)]
//   C(t): QH(t+2)            wait PL(t+1) QL(t+1)-> vmcnt(9)   [QH PX(t+1), PL QL QH(t+2)]
// One tile per workgroup: the epilogue windows alias stage 0.
// =============================================================================================
#define PPX_STAGE 57344
#define XOFF_PL 0
#define XOFF_PX 16384
#define XOFF_QL 24576
#define XOFF_QH 40960
template <bool QT>
__device__ __forceinline__ void pp_src_192(const GemmP& g, int i0, int j0, int tid, int wave, PPSrc& s) {
  const int lane = tid & 63;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = (c * 8 + wave) * 64 + lane;
    {
      const int u = id >> 3, cp = id & 7;
      const int koff = (cp ^ ((u >> 1) & 7)) << 3;
      s.pl[c] = (uint32_t)(min(i0 + (u >> 6) * 96 + (u & 63), g.I - 1) * g.ldp + koff) * 2u;
      // PX: 64 unit rows only (c = 0): unit row u -> tile row (u >> 5) * 96 + 64 + (u & 31)
      s.ph[c] = (uint32_t)(min(i0 + ((u & 63) >> 5) * 96 + 64 + (u & 31), g.I - 1) * g.ldp + koff) * 2u;
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
// one 8 KiB unit: ONE LDS-DMA instruction per wave
#define PP_GLDS1(base, so, kel, ldsoff)                                                                             \
  do {                                                                                                              \
    const char* ub__ = reinterpret_cast<const char*>((base) + (size_t)(kel));                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub__ + (so)[0]),               \
                                     (__attribute__((address_space(3))) void*)(smem + (ldsoff) + wave * 1024), 16, 0, 0); \
  } while (0)

#define PPX_KTILE(BUF, t, N1, N2)                                                                                \
  do {                                                                                                     \
    constexpr int B0 = (BUF) * PPX_STAGE, B1 = ((BUF) ^ 1) * PPX_STAGE;                                    \
    const bool n1 = (N1), n2 = (N2);                                                                              \
    /* ---- phase A: i[0..63] x j-lo ; stage PX(t+1) ---- */                                               \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qlo[a][0] = pp_frag<QT>(smem, B0 + XOFF_QL, qlb, a, 0);                                              \
      qlo[a][1] = pp_frag<QT>(smem, B0 + XOFF_QL, qlb, a, 1); }                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                        \
      pf[b][0] = pp_frag<false>(smem, B0 + XOFF_PL, plb, b, 0);                                            \
      pf[b][1] = pp_frag<false>(smem, B0 + XOFF_PL, plb, b, 1); }                                          \
    if (n1) PP_GLDS1(Pk, src.ph, ((t) + 1) * kp, B1 + XOFF_PX);                                            \
    if (n1) PP_WAIT(8); else PP_WAIT(1);                                                                   \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 0, qlo); PP_MFMA_END();                                                 \
    /* ---- phase B: i[0..63] x j-hi ; stage PL(t+2), QL(t+2) ---- */                                      \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
      qhi[a][0] = pp_frag<QT>(smem, B0 + XOFF_QH, qlb, a, 0);                                              \
      qhi[a][1] = pp_frag<QT>(smem, B0 + XOFF_QH, qlb, a, 1); }                                            \
    if (n2) { PP_GLDS(Pk, src.pl, ((t) + 2) * kp, B0 + XOFF_PL); PP_GLDS(Qk, src.ql, ((t) + 2) * kq, B0 + XOFF_QL); } \
    if (n2) PP_WAIT(11); else if (n1) PP_WAIT(7); else PP_WAIT(0);                                         \
    PP_MFMA_BEGIN(); PP_QUAD(accL, 2, qhi); PP_MFMA_END();                                                 \
    /* ---- phase C: i[64..95] x (j-lo, j-hi) ; stage QH(t+2) ---- */                                      \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
      px[b][0] = pp_frag<false>(smem, B0 + XOFF_PX, pxb, b, 0);                                            \
      px[b][1] = pp_frag<false>(smem, B0 + XOFF_PX, pxb, b, 1); }                                          \
    if (n2) PP_GLDS(Qk, src.qh, ((t) + 2) * kq, B0 + XOFF_QH);                                             \
    if (n2) PP_WAIT(9); else if (n1) PP_WAIT(3);                                                           \
    PP_MFMA_BEGIN();                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                          \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
      accH[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qlo[a][ks], px[b][ks], accH[a][b], 0, 0, 0);    \
      accH[2 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qhi[a][ks], px[b][ks], accH[2 + a][b], 0, 0, 0); } \
    PP_MFMA_END();                                                                                         \
  } while (0)

template <bool QT, bool GD = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp192_kernel(GemmP g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x 56 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nt = g.K >> 6;
  int ti, tj;
  pp_tile_ij(g, blockIdx.x, g.tiles_i * g.tiles_j, ti, tj);
  const int i0 = ti * 192, j0 = tj * 256;
  const int kp = 64, kq = QT ? 64 * g.ldq : 64;
  const bf16* Pk = reinterpret_cast<const bf16*>(g.P);
  const bf16* Qk = reinterpret_cast<const bf16*>(g.Q);
  PPSrc src;
  pp_src_192<QT>(g, i0, j0, tid, wave, src);
  const int plb = pp_lane_base<false, true>(lane, wr), qlb = pp_lane_base<QT, false>(lane, wc);
  const int pxb = pp_lane_base<false, false>(lane, wr);          // PX: 2 x 32 unit rows, as a Q unit's wave columns
  f32x4 accL[4][4], accH[4][4];                                   // accH[.][0..1]: i fragments 4, 5
  bf16x8 pf[4][2], px[2][2], qlo[2][2], qhi[2][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { accL[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; accH[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  // prologue: K tile 0 whole, K tile 1 without its PX (issued in phase A of tile 0); nt >= 2
  PP_GLDS(Pk, src.pl, 0, XOFF_PL); PP_GLDS(Qk, src.ql, 0, XOFF_QL); PP_GLDS(Qk, src.qh, 0, XOFF_QH);
  PP_GLDS1(Pk, src.ph, 0, XOFF_PX);
  PP_GLDS(Pk, src.pl, kp, PPX_STAGE + XOFF_PL); PP_GLDS(Qk, src.ql, kq, PPX_STAGE + XOFF_QL);
  PP_GLDS(Qk, src.qh, kq, PPX_STAGE + XOFF_QH);
  PP_WAIT(9);                                   // PL0, QL0 have landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);
  int t = 0;
  for (; t + 3 < nt; t += 2) {                    // steady state: no run-time flags in the stream
    PPX_KTILE(0, t, true, true);
    PPX_KTILE(1, t + 1, true, true);
  }
  for (; t + 1 < nt; t += 2) {
    PPX_KTILE(0, t, true, t + 2 < nt);
    PPX_KTILE(1, t + 1, t + 2 < nt, t + 3 < nt);
  }
  if (t < nt) PPX_KTILE(0, t, false, false);
  if (wr == 0) __builtin_amdgcn_s_barrier();    // re-align the groups: every LDS read of the K loop has retired
  __builtin_amdgcn_sched_barrier(0);
  const int ib = i0 + wr * 96, jb = j0 + wc * 64;
  const bool full = (i0 + 192 <= g.I) && (j0 + 256 <= g.J);
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63;
  char* swin = smem + wave * 4096;              // (stage 0 is dead: one tile per workgroup)
  if (full) pp_epilogue<true, false, 2, GD>(g, accL, accH, ib, jb, lane_e, swin);
  else pp_epilogue<false, false, 2, GD>(g, accL, accH, ib, jb, lane_e, swin);
}

template <bool PT, bool QT, int OUT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp256_kernel(GemmP g) {
  pp256_body<PT, QT, OUT, false>(g, nullptr);
}
#ifdef EVLM_EXPERIMENTAL_SK      // (make EXPERIMENTAL=1: measured 3.6 x slower - register allocation around the partial-tile epilogue)
template <bool QT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp256_sk_kernel(GemmP g) {
  pp256_body<false, QT, 0, false, true>(g, nullptr);
}
#endif
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp256_grouped_kernel(GemmP g, PPGroup grp) {
  pp256_body<true, true, 1, true>(g, &grp);
}

// bf16 operands, K a multiple of 64 with >= 2 K tiles per workgroup item, 32-bit operand offsets, 16-byte rows.
// bf16 output with the fused epilogue (no L0 gates, not aux AND residual), or the bare f32 weight-gradient form.
bool evlm_gemm_pp256_eligible(const GemmP& g, int pt, int qt) {
  if (g.K % 64 != 0 || g.K < 128) return false;
  if ((int64_t)(pt ? g.K : g.I) * g.ldp >= (1ll << 31) || (int64_t)(qt ? g.K : g.J) * g.ldq >= (1ll << 31)) return false;
  if (g.c_f32) {
    // measured (tools/gemm_pp256.py): with one workgroup per CU the f32 atomics of a split reduction (1.3 TB/s chip-wide)
    // are fully exposed, so weight gradients stay on the 128x128 kernels unless asked for
    static const bool wgrad = getenv("EVLM_PP256_WGRAD") != nullptr;
    if (!wgrad) return false;
    const bool bare = !g.bias && !g.gate && !g.preact && !g.aux && !g.residual && g.act == EVLM_ACT_NONE && g.dact == EVLM_ACT_NONE;
    return bare && pt && qt && g.J % 4 == 0 && g.ldc % 4 == 0;
#ifdef PP_STAMP
  }
#else
  }
  if (g.psum) return false;
#endif
  if (pt || g.accumulate) return false;
  if (g.dact != EVLM_ACT_NONE && (g.residual || (g.gate && !g.dgate))) return false;
  if (g.dgate && g.sk_ws) return false;            // (the stream-K owner epilogue does not form the gate gradient)
  if (g.J % 8 != 0 || g.ldc % 8 != 0) return false;
  if ((g.preact || g.aux || g.residual) && g.ldx % 8 != 0) return false;
  return true;
}

// number of K splits the launch below will use (the caller zero-fills C when it is > 1 and C is not accumulated into)
int evlm_gemm_pp256_splits(const GemmP& g) {
  if (!g.c_f32) return 1;
  const int tiles = ceil_div(g.I, 256) * ceil_div(g.J, 256), nt = g.K / 64;
  if (tiles >= 200 || nt < 8) return 1;
  int splits = imin(ceil_div(256, tiles), nt / 4);           // >= 4 K tiles per item
  if (splits < 1) splits = 1;
  int kps = ceil_div(nt, splits);
  splits = ceil_div(nt, kps);
  while (splits > 1 && nt - (splits - 1) * kps < 2) { ++kps; splits = ceil_div(nt, kps); }   // last item keeps >= 2 K tiles
  return splits;
}

// stream-K pays when one round of 256x256 tiles leaves a good part of the chip idle and K is long enough to cut
bool evlm_gemm_pp256_streamk(const GemmP& g, int pt) {
  // OPT-IN (EVLM_PP256_SK=1): correct and deterministic (tests), but not yet faster - profiles/r02_streamk.md
#ifndef EVLM_EXPERIMENTAL_SK
  return false;                            // (the kernel is out of the default build: make EXPERIMENTAL=1)
#endif
  static const int on = getenv("EVLM_PP256_SK") ? atoi(getenv("EVLM_PP256_SK")) : 0;
  if (!on || !g.sk_ws || g.c_f32 || pt) return false;
  const int tiles = ceil_div(g.I, 256) * ceil_div(g.J, 256), nt = g.K / 64;
  return nt % 2 == 0 && nt >= 8 && tiles >= 16 && tiles <= 200;
}

// 128 x 256 flavour: bf16 output, K-contiguous P, products whose 256 x 256 tiles fill less than ~40 % of one round
bool evlm_gemm_pp128_eligible(const GemmP& g, int pt, int qt) {
  if (g.dgate) return false;                      // (the gated activation backward lives in the 192-row kernel only)
  static const int on = getenv("EVLM_PP128") ? atoi(getenv("EVLM_PP128")) : 1;
  if (!on || pt || g.c_f32 || g.accumulate || g.psum) return false;
  if (!evlm_gemm_pp256_eligible(g, pt, qt)) return false;
  const int t256 = ceil_div(g.I, 256) * ceil_div(g.J, 256), t128 = ceil_div(g.I, 128) * ceil_div(g.J, 256);
  // (measured, tools/step_gemm_breakdown.py: 180 half tiles - the 7 680-row products - gain 1.35-1.5x over 90 full tiles;
  // 90 half tiles - the 3 840-row text pass - do not beat the 64 x 64 persistent kernel)
  static const int tmin = getenv("EVLM_PP128_MIN") ? atoi(getenv("EVLM_PP128_MIN")) : 128;     // (tuning aid)
  return t256 <= 100 && t128 >= tmin && t128 <= 256;
}
int evlm_gemm_pp128_launch(GemmP& g, int qt, hipStream_t stream) {
  const int lds = 3 * PPH_STAGE;
  g.tiles_i = ceil_div(g.I, 128); g.tiles_j = ceil_div(g.J, 256); g.bare_f32 = 0; g.sk = 0; g.kt_per_split = g.K / 64;
  const dim3 grid(g.tiles_i * g.tiles_j), block(512);
#define PP_LAUNCH_H(QT_)                                                                                      \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp128_kernel<QT_>),          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
      if (e != hipSuccess) return evlm_set_error("evlm_gemm: cannot reserve 144 KiB LDS: %s", hipGetErrorString(e)); \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_pp128_kernel<QT_>), grid, block, lds, stream, g);                           \
  } while (0)
  if (qt) PP_LAUNCH_H(true); else PP_LAUNCH_H(false);
#undef PP_LAUNCH_H
  return 0;
}

// 192 x 256 flavour: one round of 256 x 256 tiles filled 50-80 %, and 192-row tiles still fit one round
bool evlm_gemm_pp192_eligible(const GemmP& g, int pt, int qt) {
  static const int on = getenv("EVLM_PP192") ? atoi(getenv("EVLM_PP192")) : 1;
  if (g.dgate) return !pt && !g.c_f32 && !g.accumulate && !g.psum && evlm_gemm_pp256_eligible(g, pt, qt);   // (its only home)
  if (!on || pt || g.c_f32 || g.accumulate || g.psum) return false;
  if (!evlm_gemm_pp256_eligible(g, pt, qt)) return false;
  const int t256 = ceil_div(g.I, 256) * ceil_div(g.J, 256), t192 = ceil_div(g.I, 192) * ceil_div(g.J, 256);
  if (t256 <= 100) return false;
  const int r256 = ceil_div(t256, 256), r192 = ceil_div(t192, 256);
  if (r256 == 1) return t256 <= 200 && t192 <= 256 && t192 * 10 >= t256 * 12;
  // several rounds: a 192-row tile costs ~0.85 of a 256-row one (measured), so it pays when it does not add a round -
  // 270 / 300 / 360 tiles (1.05-1.4 rounds, run as 2) become 360 / 396 / 480 (2 rounds of cheaper tiles)
  static const int multi = getenv("EVLM_PP192_MULTI") ? atoi(getenv("EVLM_PP192_MULTI")) : 1;
  return multi && r192 * 85 < r256 * 100;
}
int evlm_gemm_pp192_launch(GemmP& g, int qt, hipStream_t stream) {
  const int lds = 2 * PPX_STAGE;
  g.tiles_i = ceil_div(g.I, 192); g.tiles_j = ceil_div(g.J, 256); g.bare_f32 = 0; g.sk = 0; g.kt_per_split = g.K / 64;
  const dim3 grid(g.tiles_i * g.tiles_j), block(512);
#define PP_LAUNCH_X(QT_, GD_)                                                                                 \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp192_kernel<QT_, GD_>),     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
      if (e != hipSuccess) return evlm_set_error("evlm_gemm: cannot reserve 112 KiB LDS: %s", hipGetErrorString(e)); \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_pp192_kernel<QT_, GD_>), grid, block, lds, stream, g);                      \
  } while (0)
  if (g.dgate) { if (qt) PP_LAUNCH_X(true, true); else PP_LAUNCH_X(false, true); }
  else if (qt) PP_LAUNCH_X(true, false); else PP_LAUNCH_X(false, false);
#undef PP_LAUNCH_X
  return 0;
}

int evlm_gemm_pp256_launch(GemmP& g, int pt, int qt, hipStream_t stream) {
  const int lds = 2 * PPB + 8 * 4096;
  g.tiles_i = ceil_div(g.I, 256); g.tiles_j = ceil_div(g.J, 256); g.bare_f32 = g.c_f32;
  g.sk = evlm_gemm_pp256_streamk(g, pt) ? 1 : 0;
#ifdef EVLM_EXPERIMENTAL_SK
  if (g.sk) {
    g.kt_per_split = g.K / 64;
#define PP_LAUNCH_SK(QT_)                                                                                     \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp256_sk_kernel<QT_>),       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
      if (e != hipSuccess) return evlm_set_error("evlm_gemm: cannot reserve 160 KiB LDS: %s", hipGetErrorString(e)); \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_pp256_sk_kernel<QT_>), dim3(256), dim3(512), lds, stream, g);               \
  } while (0)
    if (qt) PP_LAUNCH_SK(true); else PP_LAUNCH_SK(false);
#undef PP_LAUNCH_SK
    return 0;
  }
#endif
  const int tiles = g.tiles_i * g.tiles_j, nt = g.K / 64;
  const int splits = evlm_gemm_pp256_splits(g);
  g.kt_per_split = ceil_div(nt, splits);
  const dim3 grid(imin((int64_t)tiles * ceil_div(nt, g.kt_per_split), 256)), block(512);
#define PP_LAUNCH(PT_, QT_, OUT_)                                                                             \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {     /* 160 KiB of dynamic LDS needs the opt-in */                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp256_kernel<PT_, QT_, OUT_>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                    \
      if (e != hipSuccess) return evlm_set_error("evlm_gemm: cannot reserve 160 KiB LDS: %s", hipGetErrorString(e)); \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_pp256_kernel<PT_, QT_, OUT_>), grid, block, lds, stream, g);                \
  } while (0)
  if (g.c_f32) PP_LAUNCH(true, true, 1);
  else if (qt) PP_LAUNCH(false, true, 0);
  else PP_LAUNCH(false, false, 0);
#undef PP_LAUNCH
  return 0;
}

// ---------------------------------------------------------------------------------------------
// grouped weight gradients: C_n (+)= P_n^T Q_n for problems sharing the reduction length K, one persistent launch per
// <= PP_MAXG problems, every output tile owned by exactly one workgroup item (no split-K, no atomics unless two problems
// of a launch accumulate into the same C).
// ---------------------------------------------------------------------------------------------
extern "C" int evlm_wgrad_grouped(const evlm_wgrad_problem* pr, int n, int K, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(pr && n > 0, "evlm_wgrad_grouped: no problems");
  EVLM_REQUIRE(K % 64 == 0 && K >= 128, "evlm_wgrad_grouped: K=%d must be a multiple of 64, >= 128", K);
  const int lds = 2 * PPB + 8 * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp256_grouped_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return evlm_set_error("evlm_wgrad_grouped: cannot reserve 160 KiB LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  for (int base = 0; base < n; base += PP_MAXG) {
    const int m = imin(PP_MAXG, n - base);
    PPGroup grp;
    grp.n = m; grp.tile0[0] = 0;
    for (int k = 0; k < m; ++k) {
      const evlm_wgrad_problem& q = pr[base + k];
      EVLM_REQUIRE(q.P && q.Q && q.C && q.I > 0 && q.J > 0, "evlm_wgrad_grouped: bad problem %d", base + k);
      EVLM_REQUIRE(q.ldp % 8 == 0 && q.ldq % 8 == 0 && q.ldc % 4 == 0 && q.J % 4 == 0,
                   "evlm_wgrad_grouped: problem %d: ldp/ldq must be multiples of 8, ldc and J of 4", base + k);
      EVLM_REQUIRE(((uintptr_t)q.P | (uintptr_t)q.Q | (uintptr_t)q.C) % 16 == 0, "evlm_wgrad_grouped: 16-byte alignment");
      EVLM_REQUIRE((int64_t)K * q.ldp < (1ll << 31) && (int64_t)K * q.ldq < (1ll << 31), "evlm_wgrad_grouped: operand too large");
      grp.P[k] = q.P; grp.Q[k] = q.Q; grp.C[k] = q.C; grp.psum[k] = q.psum;
      grp.I[k] = q.I; grp.J[k] = q.J; grp.ldp[k] = q.ldp; grp.ldq[k] = q.ldq; grp.ldc[k] = q.ldc;
      grp.assign[k] = q.assign ? 1 : 0;
      grp.dup[k] = 0;
      grp.tile0[k + 1] = grp.tile0[k] + ceil_div(q.I, 256) * ceil_div(q.J, 256);
    }
    for (int k = 0; k < m; ++k)                       // two contributions to one C in this launch: both run on atomics
      for (int o = 0; o < k; ++o)
        if (grp.C[o] == grp.C[k]) { grp.dup[o] = grp.dup[k] = 1; }
    for (int k = 0; k < m; ++k)
      EVLM_REQUIRE(!(grp.dup[k] && grp.assign[k]), "evlm_wgrad_grouped: problem %d assigns a C another problem of the call writes", base + k);
    GemmP g;
    memset(&g, 0, sizeof(g));
    g.K = K; g.alpha = 1.0f; g.c_f32 = 1; g.bare_f32 = 1; g.accumulate = 1; g.kt_per_split = K / 64;
    g.act = EVLM_ACT_NONE; g.dact = EVLM_ACT_NONE;
    hipLaunchKernelGGL(gemm_bf16_pp256_grouped_kernel, dim3(imin(grp.tile0[m], 256)), dim3(512), lds, stream, g, grp);
  }
  EVLM_LAUNCH_CHECK("evlm_wgrad_grouped");
  return 0;
}
