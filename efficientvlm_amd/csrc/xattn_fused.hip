// Fused cross-attention forward for gfx950: K/V projection + Q K^T + softmax + P V (+ the probability map) in ONE launch
// (reference: BertSelfAttention.forward with encoder_hidden_states, efficient_models/eff_bert.py:277-364 - key / value
// Linear over the image tokens, matmul / sqrt(d) + mask, softmax, matmul, *= head_z; SURVEY.md 2.3 "BERT cross-attn").
//
// One workgroup = one image x one PAIR of heads.  Its 8 waves first run the K/V projection of that image's N <= 224 tokens
// for the pair's 4 x 64 output columns (K head 0, K head 1, V head 0, V head 1) as ONE 256 x 256 tile of the ping-pong
// bf16 GEMM schedule (gemm_pp256_core.h: LDS-DMA staging, two wave groups alternating on the matrix cores) - the image
// tokens are read once per workgroup, the K/V tiles never go to HBM.  The accumulators (+ bias) are then written as bf16
// into the LDS space the staging buffers occupied, in exactly the layouts the MFMA attention kernels stage K and V in
// (attention_mfma.hip: permuted key rows, swizzled 16-byte chunks), and every (query batch, head) that attends to this
// image - found through kv_index: the positive, hard-negative and MLM text rows share an image - is handled by one wave:
// S^T = K Q^T with one query per lane column, softmax in registers, P written once (optional), O^T = V^T P^T.
//
// No K/V output: the backward pass needs K, V and P in HBM, so this kernel serves the forwards that keep nothing - the
// frozen teacher's six fusion layers of every distillation step, and inference.  The training forward keeps the
// two-launch form (packed K/V GEMM + attention kernel sharing K/V through kv_index).
#include "gemm_pp256_core.h"

struct XAttnP {
  const bf16* X; const bf16* W; const float* bias; const bf16* Q; const int32_t* kv_index; const float* mask; const float* gate;
  bf16* O; bf16* P;
  int Bimg, Bq, N, Lq, d, H, ldx, ldq, ldo, ldpr;
  float scale;
};

#define XA_NT 14                      // 16-key tiles held per head: 224 keys >= the 197 tokens of a 224 x 224 image
#define XA_TILE (XA_NT * 16 * 128)    // bytes of one [224][64] bf16 K or V tile

// ---- the LDS layouts of attention_mfma.hip (kept in step with it: the fragment readers below are the same) -----------
__device__ __forceinline__ int xa_k_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int xa_v_swz(int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); }
__device__ __forceinline__ int xa_key_row(int key) {
  return (((key >> 5) << 1) | ((key >> 2) & 1)) * 16 + ((key >> 3) & 3) * 4 + (key & 3);
}
__device__ __forceinline__ int xa_tile_key0(int t, int g) { return (t >> 1) * 32 + g * 8 + (t & 1) * 4; }
__device__ __forceinline__ bf16x8 xa_krow_frag(const char* sm, int t, int ks, int lane) {
  const int row = t * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(sm + row * 128 + xa_k_swz(row, c) * 16);
}
__device__ __forceinline__ bf16x8 xa_vcol_frag(const char* sm, int t0, int t1, int dt, int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  bf16x8 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = (h ? t1 : t0) * 16 + g * 4 + q;
    const int off = row * 128 + xa_v_swz(row, dt * 2 + (p >> 1)) * 16 + ((p & 1) << 3);
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(sm + off));
    out[4 * h + 0] = t[0]; out[4 * h + 1] = t[1]; out[4 * h + 2] = t[2]; out[4 * h + 3] = t[3];
  }
  return out;
}

// one (query batch bq, head h) against the K / V tiles in LDS: 16 queries per pass (lane & 15), all keys in registers
__device__ __forceinline__ void xa_attend(const XAttnP& a, const char* Ks, const char* Vs, const float* Ms, int bq, int h,
                                          int lane) {
  const int g = lane >> 4, ql = lane & 15;
  const float* mrow = a.mask ? a.mask + (size_t)bq * a.N : nullptr;
  const float sc = a.scale * 1.44269504088896341f;
  const float gz = a.gate ? a.gate[h] : 1.0f;
  for (int q0 = 0; q0 < a.Lq; q0 += 16) {
    const int q = q0 + ql;
    const bool qok = q < a.Lq;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qok) v = *reinterpret_cast<const uint4*>(a.Q + ((size_t)bq * a.Lq + q) * a.ldq + h * 64 + ks * 32 + g * 8);
      qf[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
    f32x4 acc[XA_NT];
#pragma unroll
    for (int t = 0; t < XA_NT; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_krow_frag(Ks, t, ks, lane), qf[ks], acc[t], 0, 0, 0);
    }
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < XA_NT; ++t) {
      const int key0 = xa_tile_key0(t, g);
      f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + key0);          // 0 for keys < N, -1e30 beyond (LDS)
      if (mrow && key0 + 3 < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mk[r] += mrow[key0 + r];
      } else if (mrow) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (key0 + r < a.N) mk[r] += mrow[key0 + r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[t][r] = fmaf(acc[t][r], sc, mk[r] * 1.44269504088896341f);   // (the forms of attention_mfma.hip: bit-identical maps)
        m = fmaxf(m, acc[t][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < XA_NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[t][r] = __builtin_amdgcn_exp2f(acc[t][r] - m);
        sum += acc[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    bf16x4 pk[XA_NT];
#pragma unroll
    for (int t = 0; t < XA_NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[t][r] = (bf16)(acc[t][r] * inv);
    }
    if (a.P && qok) {
      bf16* Pr = a.P + (((size_t)bq * a.H + h) * a.Lq + q) * a.ldpr;
#pragma unroll
      for (int s = 0; s < XA_NT / 2; ++s) {
        const int kcol = s * 32 + g * 8;
        if (kcol < a.ldpr) {
          bf16x8 pp;
#pragma unroll
          for (int r = 0; r < 4; ++r) { pp[r] = pk[2 * s][r]; pp[4 + r] = pk[2 * s + 1][r]; }
          *reinterpret_cast<bf16x8*>(Pr + kcol) = pp;
        }
      }
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < XA_NT / 2; ++s) {
      bf16x8 pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pb[r] = pk[2 * s][r]; pb[4 + r] = pk[2 * s + 1][r]; }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_vcol_frag(Vs, 2 * s, 2 * s + 1, dt, lane), pb, o[dt], 0, 0, 0);
    }
    if (qok) {
      bf16* Or = a.O + ((size_t)bq * a.Lq + q) * a.ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov = {(bf16)(o[dt][0] * gz), (bf16)(o[dt][1] * gz), (bf16)(o[dt][2] * gz), (bf16)(o[dt][3] * gz)};
        *reinterpret_cast<bf16x4*>(Or + dt * 16 + g * 4) = ov;
      }
    }
  }
}

__global__ __launch_bounds__(512, 1) void xattn_fused_kernel(XAttnP a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 x 64 KiB staging, reused for the four K / V tiles
  constexpr bool PT = false, QT = false;
  constexpr int OUT = 0;
  const bool do_psum = false;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  // workgroups of one image (its H / 2 head pairs read the same token rows) sit on one XCD: blocks that share blockIdx & 7
  // share an XCD under round-robin placement (a speed choice, never a correctness one)
  const int HP = a.H >> 1, nwg = a.Bimg * HP;
  int b, hp;
  {
    const int vb = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = vb & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    b = t / HP; hp = t - b * HP;
  }
  const int i0 = b * a.N, ilim = i0 + a.N - 1;
  const int nt = a.d >> 6;
  const int kp = 64, kq = 64;
  const bf16* Pk = a.X;
  const bf16* Qk = a.W;
  PPSrc src;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = (c * 8 + wave) * 64 + lane;
    const int u = id >> 3, cp = id & 7;
    const int koff = (cp ^ ((u >> 1) & 7)) << 3;
    src.pl[c] = (uint32_t)(min(i0 + pp_prow(u, 0), ilim) * a.ldx + koff) * 2u;
    src.ph[c] = (uint32_t)(min(i0 + pp_prow(u, 1), ilim) * a.ldx + koff) * 2u;
    // tile column -> packed weight row: columns [64 wq, 64 wq + 64) are K (wq < 2) or V (wq >= 2) of head 2 hp + (wq & 1)
    const int cl = pp_qcol(u, 0), ch = pp_qcol(u, 1);
    const int rl = ((cl >> 6) >= 2 ? a.d : 0) + (2 * hp + ((cl >> 6) & 1)) * 64 + (cl & 63);
    const int rh = ((ch >> 6) >= 2 ? a.d : 0) + (2 * hp + ((ch >> 6) & 1)) * 64 + (ch & 63);
    src.ql[c] = (uint32_t)(rl * a.d + koff) * 2u;
    src.qh[c] = (uint32_t)(rh * a.d + koff) * 2u;
  }
  const int plb = pp_lane_base<PT, true>(lane, wr), qlb = pp_lane_base<QT, false>(lane, wc);
  f32x4 accL[4][4], accH[4][4];
  bf16x8 pf[4][2], qf[2][2];
  float ps[8];
  PP_GLDS(Pk, src.pl, 0, OFF_PL); PP_GLDS(Qk, src.qh, 0, OFF_QH); PP_GLDS(Pk, src.ph, 0, OFF_PH);
  PP_GLDS(Qk, src.ql, 0, OFF_QL); PP_GLDS(Pk, src.pl, kp, PPB + OFF_PL); PP_GLDS(Qk, src.qh, kq, PPB + OFF_QH);
  PP_WAIT(4);
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) { accL[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f}; accH[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int y = 0; y < 8; ++y) ps[y] = 0.f;
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    PP_KTILE(0, t, (t) + 1 < nt, (t) + 2 < nt);
    PP_KTILE(1, t + 1, (t) + 2 < nt, (t) + 3 < nt);
  }
  if (t < nt) PP_KTILE(0, t, (t) + 1 < nt, (t) + 2 < nt);
  if (wr == 0) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  PP_WAIT(0);
  __syncthreads();                     // every staging read and DMA has retired: the space becomes the K / V tiles

  // (the per-lane address arithmetic of everything below must NOT be hoisted above the K loop, where every register is
  // spoken for: launder the thread id it derives from - the same measure as in gemm_pp256.hip's persistent loop)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, wave_e = tid_e >> 6, wc_e = wave_e & 3, wr_e = wave_e >> 2;

  // ---- accumulators (+ bias) -> bf16 K / V tiles in the attention kernels' LDS layouts --------------------------------
  {
    char* tile = smem + wc_e * XA_TILE;                     // tiles 0, 1: K of heads 2hp, 2hp+1; tiles 2, 3: V
    const bool is_v = wc_e >= 2;
    const int il = lane_e & 15, jl = (lane_e >> 4) * 4;
    const int col0 = (is_v ? a.d : 0) + (2 * hp + (wc_e & 1)) * 64;
    f32x4 bz[4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
      bz[x] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + col0 + x * 16 + jl) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int key = wr_e * 128 + half * 64 + y * 16 + il;
        if (key < XA_NT * 16) {
          const int row = xa_key_row(key);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const f32x4 v = half ? accH[x][y] : accL[x][y];
            const int dcol = x * 16 + jl, chunk = dcol >> 3;
            const int sw = is_v ? xa_v_swz(row, chunk) : xa_k_swz(row, chunk);
            bf16x4 o4 = {(bf16)(v[0] + bz[x][0]), (bf16)(v[1] + bz[x][1]), (bf16)(v[2] + bz[x][2]), (bf16)(v[3] + bz[x][3])};
            *reinterpret_cast<bf16x4*>(tile + row * 128 + sw * 16 + ((dcol & 4) << 1)) = o4;
          }
        }
      }
    }
  }
  float* Ms = reinterpret_cast<float*>(smem + 4 * XA_TILE);    // [224]: 0 for real tokens, -1e30 for the padding keys
  if (tid_e < XA_NT * 16) Ms[tid_e] = tid_e < a.N ? 0.f : -1e30f;
  __syncthreads();

  // ---- attention: one wave per (query batch, head) that attends to this image ----------------------------------------
  // the query batches of this image are found 64 at a time (one index per lane, a ballot), not by a scalar scan: 256
  // dependent scalar loads per wave were most of this phase's time
  int unit = 0;
  for (int base = 0; base < a.Bq; base += 64) {
    const int bl = base + lane_e;
    int img = -1;
    if (bl < a.Bq) img = a.kv_index ? a.kv_index[bl] : bl;
    unsigned long long hit = __ballot(img == b);
    while (hit) {
      const int bq = base + __ffsll((long long)hit) - 1;
      hit &= hit - 1;
#pragma unroll 1
      for (int hl = 0; hl < 2; ++hl, ++unit) {
        if ((unit & 7) != wave_e) continue;
        xa_attend(a, smem + hl * XA_TILE, smem + (2 + hl) * XA_TILE, Ms, bq, 2 * hp + hl, lane_e);
      }
    }
  }
}

extern "C" int evlm_xattn_fused_fwd(const evlm_xattn_fused_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVLM_REQUIRE(a && a->X && a->Wkv && a->Q && a->O, "evlm_xattn_fused_fwd: null operand");
  EVLM_REQUIRE(a->dtype == EVLM_BF16, "evlm_xattn_fused_fwd: bf16 only");
  EVLM_REQUIRE(a->dh == 64 && a->H > 0 && a->H % 2 == 0 && a->d == a->H * 64, "evlm_xattn_fused_fwd: head dim 64, an even head count, d = 64 H");
  EVLM_REQUIRE(a->d % 64 == 0 && a->d >= 128, "evlm_xattn_fused_fwd: d must be a multiple of 64, >= 128");
  EVLM_REQUIRE(a->N > 0 && a->N <= XA_NT * 16, "evlm_xattn_fused_fwd: at most %d image tokens (got %d)", XA_NT * 16, a->N);
  EVLM_REQUIRE(a->Bimg > 0 && a->Bq > 0 && a->Lq > 0, "evlm_xattn_fused_fwd: bad shape");
  EVLM_REQUIRE((a->ldx | a->ldq | a->ldo) % 8 == 0 && (!a->P || (a->ldpr % 8 == 0 && a->ldpr >= a->N)), "evlm_xattn_fused_fwd: strides");
  EVLM_REQUIRE((int64_t)a->Bimg * a->N * a->ldx < (1ll << 30) && (int64_t)2 * a->d * a->d < (1ll << 30), "evlm_xattn_fused_fwd: operand too large");
  XAttnP p;
  p.X = (const bf16*)a->X; p.W = (const bf16*)a->Wkv; p.bias = a->bias_kv; p.Q = (const bf16*)a->Q; p.kv_index = a->kv_index;
  p.mask = a->mask; p.gate = a->head_gate; p.O = (bf16*)a->O; p.P = (bf16*)a->P;
  p.Bimg = a->Bimg; p.Bq = a->Bq; p.N = a->N; p.Lq = a->Lq; p.d = a->d; p.H = a->H;
  p.ldx = a->ldx; p.ldq = a->ldq; p.ldo = a->ldo; p.ldpr = a->ldpr; p.scale = a->scale;
  const int lds = 2 * PPB;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return evlm_set_error("evlm_xattn_fused_fwd: cannot reserve 128 KiB LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(xattn_fused_kernel, dim3(a->Bimg * (a->H / 2)), dim3(512), lds, stream, p);
  EVLM_LAUNCH_CHECK("evlm_xattn_fused_fwd");
  return 0;
}
