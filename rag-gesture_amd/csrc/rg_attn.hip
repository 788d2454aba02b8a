// Linear ("efficient") attention kernels of the denoiser, fp32 VALU with wavefront shuffles.
//
// The reference's attention has no TxT score matrix: per head (head_dim 32)
//   P = softmax_over_tokens(K + mask),  A = P^T V  (32x32),  y = softmax_hd(Q) A.
// One wave owns one head of one batch row; a 256-thread workgroup = 4 adjacent heads
// (128 output columns), so a row's LayerNorm statistics leave the kernel as 4 partial
// (sum, sumsq) pairs that the consumer GEMM folds into its A-operand prologue.
//
//  rg_sa_attention : efficient_attention.py:23-41  (EfficientSelfAttention, up to `y`)
//  rg_ca_attention : efficient_attention.py:90-98  (y = Q A for precomputed A, + query mask)
//  rg_kv_reduce    : efficient_attention.py:82-90  (A = softmax_N(K)^T V over conditioning tokens;
//                    depends only on the conditioning, so it runs once per clip, not per step)
#include "rg_common.h"

namespace {

constexpr int HD = 32;       // head dim
constexpr int WAVES = 4;     // heads per workgroup
constexpr int TMAX = 64;     // max tokens per batch row handled by the per-step kernels

__device__ __forceinline__ float wave_sum32(float v) {  // sum over the 32 lanes of each half-wave
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  return v;
}

// y[n][l] = sum_d q[n][d] * A[d][l] for the T tokens of one head; lane -> (l = lane&31, half).
// Areg = column l of A.  Writes y (optionally quantised like the reference's "-1e6 + y" on
// masked query rows) to global memory and, in place of the consumed q row, to LDS; the per-token
// (sum, sumsq) over the head's 32 columns is then one lane per token reading its row with a
// rotated column order (bank-conflict free) -- no cross-lane shuffles, fixed summation order, so
// results are bitwise reproducible run to run.  sstat[T][2] is this wave's own LDS slice.
__device__ __forceinline__ void qa_and_store(float* sq, const float (&Areg)[HD], float* yout, int ldy,
                                             int T, int lane, const float* qmask, float* sstat) {
  const int l = lane & 31, half = lane >> 5;
#pragma unroll 2
  for (int n = half; n < T; n += 2) {
    const float* qr = sq + n * HD;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < HD; d4 += 4) {
      float4 q = *reinterpret_cast<const float4*>(qr + d4);
      a0 = fmaf(q.x, Areg[d4], a0);
      a1 = fmaf(q.y, Areg[d4 + 1], a1);
      a2 = fmaf(q.z, Areg[d4 + 2], a2);
      a3 = fmaf(q.w, Areg[d4 + 3], a3);
    }
    float acc = (a0 + a1) + (a2 + a3);
    if (qmask && qmask[n] == 0.f) {
      // reference: y + (1 - query_mask) * -1e6 in fp32, then LayerNorm (shift invariant).
      // z is y rounded onto the fp32 grid near -1e6 (spacing 1/16); z + 1e6 is exact.
      const float z = acc + (-1000000.0f);
      acc = z + 1000000.0f;
    }
    yout[(size_t)n * ldy + l] = acc;
    sq[n * HD + l] = acc;   // row n is read by this half-wave only, and all its reads fed acc
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  if (lane < T) {
    const float* yr = sq + lane * HD;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < HD; j += 2) {
      const float v0 = yr[(j + lane) & 31], v1 = yr[(j + 1 + lane) & 31];
      s0 += v0;
      s1 += v1;
      q0 = fmaf(v0, v0, q0);
      q1 = fmaf(v1, v1, q1);
    }
    sstat[2 * lane] = s0 + s1;
    sstat[2 * lane + 1] = q0 + q1;
  }
}

// MFMA_ = false: exact fp32 VALU products (precision = "fp32" path);  true: both products on the matrix
// cores with bf16 hi + lo operand pairs and fp32 accumulation (~2^-17 relative per product).
#ifdef RG_STAMPS
// Diagnostic build only (RG_DIAG=1 -> librg_gesture_diag.so): wall-clock stamps (100 MHz) of the phases of wave 0 of
// workgroup 0 of the attention kernels; no output depends on them.
__device__ unsigned long long* g_stamp_buf3 = nullptr;
#define RG_STAMP3(slot)                                                                   \
  do {                                                                                    \
    if (g_stamp_buf3 && threadIdx.x == 0 && blockIdx.x == 0)                              \
      g_stamp_buf3[slot] = __builtin_amdgcn_s_memrealtime();                              \
  } while (0)
#else
#define RG_STAMP3(slot)
#endif

// TCAP = token capacity of the softmax's register column (48 or 64): T = 43 needs 24 values per lane, not 32
// YBF: y leaves as bf16 (the A operand of the SA-out GEMM, which stylizes it in LDS); the statistics are those of the
// fp32 values
template <bool MFMA_, int TCAP = TMAX, bool YBF = false>
__global__ void __launch_bounds__(256, 2) sa_attention_kernel(const float* __restrict__ qkv, int ldqkv, int D,
                                                          const float* __restrict__ src_mask, float* __restrict__ y,
                                                          int ldy, float* __restrict__ stats, int T,
                                                          const int* __restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // perm: block -> work item, so that a row group is processed on the XCD whose L2 holds its rows
  // (the GEMMs put M-tile t on XCD t % 8); -1 = padding block
  RG_STAMP3(0);
  const int item = perm ? perm[blockIdx.x] : (int)blockIdx.x;
  if (item < 0) return;
  const int b = item / (D / (HD * WAVES));
  const int hg = item % (D / (HD * WAVES));
  const int h = hg * WAVES + wave;
  const int Tp = (T + 7) & ~7;
  float* sstat = sm;                       // [WAVES][Tp][2]
  float* smask = sm + WAVES * 2 * Tp;      // [Tp]  (token mask staged once: no global loads in the loops)
  float* sq = smask + Tp + wave * (3 * Tp * HD);
  float* sk = sq + Tp * HD;
  float* sv = sk + Tp * HD;
  constexpr int AS = HD + 1;
  float* sA = sk;                          // [32][33] reuses the P tile once A is in registers (Tp*32 >= 32*33)
  const float mval = src_mask[(size_t)b * T + (lane < T ? lane : 0)];   // requested first, used after the tiles
  // q, k, v tiles [T][32] straight into LDS: one 1-KiB LDS-DMA per 8 rows and tile (register staging -- 24 plain
  // 16-B loads, then ds_write_b128 -- was measured 0.4 us slower at R = 32)
  {
    typedef __attribute__((address_space(3))) void lds_void;
    const float* src = qkv + (size_t)b * T * ldqkv + h * HD + (lane & 7) * 4;
    for (int r0 = 0; r0 < Tp; r0 += 8) {
      int r = r0 + (lane >> 3);
      r = r < T ? r : T - 1;   // pad rows duplicate the last token (never read back)
      const float* rp = src + (size_t)r * ldqkv;
      __builtin_amdgcn_global_load_lds((const void*)(rp + D), (lds_void*)(sk + r0 * HD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(rp + 2 * D), (lds_void*)(sv + r0 * HD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)rp, (lds_void*)(sq + r0 * HD), 16, 0, 0);
    }
  }
  RG_STAMP3(1);
  // token validity as a wave-uniform bit set (bit n: token n takes part); a wave touches only its own q/k/v tiles,
  // so no workgroup barrier is needed before the softmax
  const unsigned long long vbits = __ballot(lane < T && mval != 0.f);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RG_STAMP3(2);
  __builtin_amdgcn_wave_barrier();
  RG_STAMP3(3);

  // softmax over tokens for column d = lane&31; the two half-waves split the tokens.  The column
  // lives in registers: ONE batch of independent, unconditional LDS reads (rows are clamped into the tile and
  // deselected by the bit set -- a guarded read would serialise into read / wait / branch per token)
  {
    const int d = lane & 31, half = lane >> 5;
    constexpr int NH = TCAP / 2;
    float kr[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int n = half + 2 * i;
      kr[i] = sk[(n < Tp ? n : Tp - 1) * HD + d];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int n = half + 2 * i;
      // key + (1-mask)*-1e6: a masked token's weight underflows to exactly 0 in fp32
      kr[i] = ((vbits >> n) & 1ull) ? kr[i] : -INFINITY;
      mx = fmaxf(mx, kr[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      // bf16 (MFMA) mode: exp2-based fast exponential (relative error ~1e-6 at these arguments, far below the
      // bf16 operand rounding of the GEMMs around); fp32 mode keeps expf
      kr[i] = (kr[i] == -INFINITY) ? 0.f : (MFMA_ ? __expf(kr[i] - mx) : expf(kr[i] - mx));
      sum += kr[i];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int n = half + 2 * i;
      if (n < T) sk[n * HD + d] = kr[i] * inv;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
  __builtin_amdgcn_wave_barrier();
  RG_STAMP3(4);
  if constexpr (MFMA_) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    auto pk = [](float x, float y) {
      return rg_pack2_bf16(x, y);
    };
    auto split = [&](const float (&x)[8], bf16x8& hi, bf16x8& lo) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = x[e] - (float)(__bf16)x[e];
      hi = __builtin_bit_cast(bf16x8, u32x4{pk(x[0], x[1]), pk(x[2], x[3]), pk(x[4], x[5]), pk(x[6], x[7])});
      lo = __builtin_bit_cast(bf16x8, u32x4{pk(r[0], r[1]), pk(r[2], r[3]), pk(r[4], r[5]), pk(r[6], r[7])});
    };
    const int l15 = lane & 15, g = lane >> 4;
    // ---- A[d][l] = sum_n P[n][d] V[n][l]: rows d (2 blocks), columns l (2 blocks), K = tokens (2 x 32, zero
    // beyond T; masked tokens have P = 0 exactly).  lane -> row/col 16*blk + l15, tokens 32*ks + 8*g + j
    f32x4 accA[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) accA[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ph[2], pl[2], vh[2], vl[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        float pv[8], vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = 32 * ks + 8 * g + j;
          const int nn = n < Tp ? n : Tp - 1;        // unconditional in-tile reads, deselected below
          pv[j] = sk[nn * HD + 16 * blk + l15];
          vv[j] = sv[nn * HD + 16 * blk + l15];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = 32 * ks + 8 * g + j < T;
          pv[j] = ok ? pv[j] : 0.f;
          vv[j] = ok ? vv[j] : 0.f;
        }
        split(pv, ph[blk], pl[blk]);
        split(vv, vh[blk], vl[blk]);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          accA[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[rb], vh[nb], accA[rb][nb], 0, 0, 0);
          accA[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rb], vl[nb], accA[rb][nb], 0, 0, 0);
          accA[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rb], vh[nb], accA[rb][nb], 0, 0, 0);
        }
    }
    RG_STAMP3(5);
    // ---- y = q A.  accA[rb][nb][e] = A[16 rb + 4 g + e][16 nb + l15] is already a B fragment if the
    // contraction index is enumerated as slot (g, j) <-> d = 16*(j/4) + 4*g + j%4; q is read with the same
    // permutation (two 16-B LDS reads per fragment), so A never leaves the registers.
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float av[8] = {accA[0][nb][0], accA[0][nb][1], accA[0][nb][2], accA[0][nb][3],
                           accA[1][nb][0], accA[1][nb][1], accA[1][nb][2], accA[1][nb][3]};
      split(av, ah[nb], al[nb]);
    }
    f32x4 accY[4][2];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      accY[tb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      accY[tb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (16 * tb < Tp) {
        int row = 16 * tb + l15;
        row = row < Tp ? row : Tp - 1;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sq + row * HD + 4 * g);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(sq + row * HD + 16 + 4 * g);
        const float qv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        bf16x8 qh, ql;
        split(qv, qh, ql);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          accY[tb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql, ah[nb], accY[tb][nb], 0, 0, 0);
          accY[tb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, al[nb], accY[tb][nb], 0, 0, 0);
          accY[tb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, ah[nb], accY[tb][nb], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // every lane's q reads are complete before y overwrites the tile
    __builtin_amdgcn_wave_barrier();
    RG_STAMP3(6);
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * tb + 4 * g + e;
        if (row < T) {
          sq[row * HD + l15] = accY[tb][0][e];
          sq[row * HD + 16 + l15] = accY[tb][1][e];
        }
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    RG_STAMP3(7);
    // coalesced y store (8 rows x 128 B per wave-instruction) and per-token statistics from the tile
    {
      if constexpr (YBF) {   // 16 rows x 64 B per wave-instruction
        unsigned short* yout = reinterpret_cast<unsigned short*>(y) + (size_t)b * T * ldy + h * HD;
        for (int r0 = 0; r0 < T; r0 += 16) {
          const int r = r0 + (lane >> 2);
          if (r < T) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(sq + r * HD + (lane & 3) * 8);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(sq + r * HD + (lane & 3) * 8 + 4);
            *reinterpret_cast<u32x4*>(yout + (size_t)r * ldy + (lane & 3) * 8) =
                u32x4{pk(x0[0], x0[1]), pk(x0[2], x0[3]), pk(x1[0], x1[1]), pk(x1[2], x1[3])};
          }
        }
      } else {
      float* yout = y + (size_t)b * T * ldy + h * HD;
      for (int r0 = 0; r0 < T; r0 += 8) {
        const int r = r0 + (lane >> 3);
        if (r < T)
          *reinterpret_cast<f32x4*>(yout + (size_t)r * ldy + (lane & 7) * 4) =
              *reinterpret_cast<const f32x4*>(sq + r * HD + (lane & 7) * 4);
      }
      }
      float* mystat = sstat + wave * 2 * Tp;
      if (lane < T) {
        const float* yr = sq + lane * HD;
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int j = 0; j < HD; j += 2) {
          const float v0 = yr[(j + lane) & 31], v1 = yr[(j + 1 + lane) & 31];
          s0 += v0;
          s1 += v1;
          q0 = fmaf(v0, v0, q0);
          q1 = fmaf(v1, v1, q1);
        }
        mystat[2 * lane] = s0 + s1;
        mystat[2 * lane + 1] = q0 + q1;
      }
    }
  } else {
  // A[d][l] = sum_n P[n][d] * (V[n][l] * mask[n]); lane -> (d = lane&31, 16 l's)
  {
    const int d = lane & 31, l0 = (lane >> 5) * 16;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int n = 0; n < T; ++n) {
      const float pn = sk[n * HD + d];
      const float* vr = sv + n * HD + l0;
#pragma unroll
      for (int j4 = 0; j4 < 16; j4 += 4) {
        float4 vv = *reinterpret_cast<const float4*>(vr + j4);
        acc[j4] = fmaf(pn, vv.x, acc[j4]);
        acc[j4 + 1] = fmaf(pn, vv.y, acc[j4 + 1]);
        acc[j4 + 2] = fmaf(pn, vv.z, acc[j4 + 2]);
        acc[j4 + 3] = fmaf(pn, vv.w, acc[j4 + 3]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // every lane's reads of P are done before A overwrites it
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) sA[d * AS + l0 + j] = acc[j];   // row stride 33: conflict-free transposed write
  }
  // (same wave wrote sA: a wave-level LDS fence is enough)
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  float Areg[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) Areg[d] = sA[d * AS + (lane & 31)];
  qa_and_store(sq, Areg, y + (size_t)b * T * ldy + h * HD, ldy, T, lane, nullptr, sstat + wave * 2 * Tp);
  }
  RG_STAMP3(8);
  __syncthreads();
  RG_STAMP3(9);
  const int ngroups = D / (HD * WAVES);
  for (int n = threadIdx.x; n < T; n += 256) {
    float* so = stats + (((size_t)b * T + n) * ngroups + hg) * 2;
    so[0] = (sstat[2 * n] + sstat[2 * Tp + 2 * n]) + (sstat[4 * Tp + 2 * n] + sstat[6 * Tp + 2 * n]);
    so[1] = (sstat[2 * n + 1] + sstat[2 * Tp + 2 * n + 1]) + (sstat[4 * Tp + 2 * n + 1] + sstat[6 * Tp + 2 * n + 1]);
  }
}

// grid = R * ncond * (D/128); Apre layout [cond][R][H][32][32]; q3/y3 layout [M][ncond*D];
// stats layout [cond][M][D/128][2]; qmask layout [cond][R][T].
__global__ void __launch_bounds__(256, 4) ca_attention_kernel(const float* __restrict__ q3, const float* __restrict__ Apre,
                                                          const float* __restrict__ qmask, float* __restrict__ y3,
                                                          float* __restrict__ stats, int R, int T, int D, int ncond,
                                                          int Rc, const float* __restrict__ Aunc,
                                                          const int* __restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ngroups = D / (HD * WAVES);
  const int H = D / HD;
  int bid = perm ? perm[blockIdx.x] : (int)blockIdx.x;
  if (bid < 0) return;
  const int hg = bid % ngroups;
  bid /= ngroups;
  const int c = bid % ncond;
  const int b = bid / ncond;
  const int h = hg * WAVES + wave;
  const int Tp = (T + 7) & ~7;
  float* sstat = sm;  // [WAVES][Tp][2]
  float* smask = sm + WAVES * 2 * Tp;  // [Tp]
  float* sq = smask + Tp + wave * (Tp * HD);
  const int ld = ncond * D;
  // q tile [T][32] straight into LDS: one 1-KiB LDS-DMA per 8 rows (8 lanes x 16 B per row), no
  // staging registers, so 4 workgroups fit a CU and the grid runs in one round
  {
    typedef __attribute__((address_space(3))) void lds_void;
    const float* src = q3 + (size_t)b * T * ld + c * D + h * HD + (lane & 7) * 4;
    for (int r0 = 0; r0 < Tp; r0 += 8) {
      int r = r0 + (lane >> 3);
      r = r < T ? r : T - 1;   // pad rows duplicate the last token (never read back)
      __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)r * ld), (lds_void*)(sq + r0 * HD), 16, 0, 0);
    }
  }
  // rows [0,Rc) carry per-row conditioning; rows [Rc,R) are the classifier-free "no condition"
  // branch whose A depends only on the weights (Aunc[cond][H][32][32])
  const float* Ap = (b < Rc) ? Apre + ((((size_t)c * Rc + b) * H + h) * HD) * HD
                             : Aunc + (((size_t)c * H + h) * HD) * HD;
  float Areg[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) Areg[d] = Ap[d * HD + (lane & 31)];
  if (qmask)
    for (int n = threadIdx.x; n < T; n += 256) smask[n] = qmask[((size_t)c * R + b) * T + n];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* qm = qmask ? smask : nullptr;
  qa_and_store(sq, Areg, y3 + (size_t)b * T * ld + c * D + h * HD, ld, T, lane, qm, sstat + wave * 2 * Tp);
  __syncthreads();
  for (int n = threadIdx.x; n < T; n += 256) {
    float* so = stats + ((((size_t)c * R + b) * T + n) * ngroups + hg) * 2;
    so[0] = (sstat[2 * n] + sstat[2 * Tp + 2 * n]) + (sstat[4 * Tp + 2 * n] + sstat[6 * Tp + 2 * n]);
    so[1] = (sstat[2 * n + 1] + sstat[2 * Tp + 2 * n + 1]) + (sstat[4 * Tp + 2 * n + 1] + sstat[6 * Tp + 2 * n + 1]);
  }
}

// One workgroup per (batch row, head): A[d][l] = sum_n softmax_n(K[n][d]) * V[n][l] over N tokens.
__global__ void __launch_bounds__(256) kv_reduce_kernel(const float* __restrict__ kv, int ldkv, int D, int N,
                                                       float* __restrict__ A, int H) {
  // thread -> (key column d, token residue class `part` of 8).  Two passes over K (column max, then the
  // exponentials), ONE over V: every thread accumulates all 32 value columns for its tokens (the 32 lanes of
  // a part read the same 128-B V row: one request), then the 8 partial (sum, A-row) sets are added through LDS
  // in a fixed order.
  __shared__ float red[8][HD];
  __shared__ float smax[HD];
  __shared__ float sacc[8][HD][HD + 1];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = threadIdx.x & 31, part = threadIdx.x >> 5;
  const float* kb = kv + (size_t)b * N * ldkv + h * HD;
  const float* vb = kb + D;
  float mx = -INFINITY;
#pragma unroll 4
  for (int n = part; n < N; n += 8) mx = fmaxf(mx, kb[(size_t)n * ldkv + d]);
  red[part][d] = mx;
  __syncthreads();
  if (part == 0) {
    float m = red[0][d];
    for (int q = 1; q < 8; ++q) m = fmaxf(m, red[q][d]);
    smax[d] = m;
  }
  __syncthreads();
  mx = smax[d];
  float sum = 0.f;
  float acc[HD];
#pragma unroll
  for (int l = 0; l < HD; ++l) acc[l] = 0.f;
#pragma unroll 2
  for (int n = part; n < N; n += 8) {
    const float pn = expf(kb[(size_t)n * ldkv + d] - mx);
    sum += pn;
    const float4* vr = reinterpret_cast<const float4*>(vb + (size_t)n * ldkv);
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const float4 vv = vr[q];
      acc[4 * q] = fmaf(pn, vv.x, acc[4 * q]);
      acc[4 * q + 1] = fmaf(pn, vv.y, acc[4 * q + 1]);
      acc[4 * q + 2] = fmaf(pn, vv.z, acc[4 * q + 2]);
      acc[4 * q + 3] = fmaf(pn, vv.w, acc[4 * q + 3]);
    }
  }
  __syncthreads();
  red[part][d] = sum;
#pragma unroll
  for (int l = 0; l < HD; ++l) sacc[part][d][l] = acc[l];
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += red[q][d];
  const float inv = 1.0f / s;
  const int l0 = part * 4;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += sacc[q][d][l0 + e];
  *reinterpret_cast<float4*>(A + (((size_t)b * H + h) * HD + d) * HD + l0) =
      make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
}

// Exact fp32 linear for tiny, load-time problems (time-embedding tables): out[m][n] = a[m].w[n] + bias,
// one wave per output column, optional SiLU on the input and/or output.
__global__ void __launch_bounds__(256) linear_f32_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int M, int N, int K, int silu_in, int silu_out) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= N) return;
  const float* wr = w + (size_t)wave * K;
  for (int m = 0; m < M; ++m) {
    const float* ar = a + (size_t)m * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
      float x = ar[k];
      if (silu_in) x = x / (1.0f + expf(-x));
      acc = fmaf(x, wr[k], acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 8);
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (lane == 0) {
      float r = acc + (bias ? bias[wave] : 0.f);
      if (silu_out) r = r / (1.0f + expf(-r));
      out[(size_t)m * N + wave] = r;
    }
  }
}

// stats[row][p] = (sum, sumsq) of x[row][64p .. 64p+63]; one wave per (row, 64-column part).
__global__ void __launch_bounds__(256) row_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                       int rows, int dim) {
  const int parts = dim / 64;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= rows * parts) return;
  const float v = x[(size_t)(wave / parts) * dim + (wave % parts) * 64 + lane];
  float s = v, ss = v * v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  if (lane == 0) {
    stats[2 * (size_t)wave] = s;
    stats[2 * (size_t)wave + 1] = ss;
  }
}

// max over rows of mean^2 / var from partial (sum, sumsq) statistics: how far a row's mean sits from zero in units of
// its spread.  The folded LayerNorm (rg_gemm: ln_stats / ln_c1) multiplies the bf16 copy of the UN-normalised row, whose
// rounding error scales with |mean|; the host reads this figure once per session and falls back to the LayerNorm
// pre-pass when rows are far off centre.  Non-negative floats order like their bit patterns: atomicMax on the bits.
__global__ void __launch_bounds__(256) ln_guard_kernel(const float* __restrict__ stats, int rows, int nparts, int K,
                                                      unsigned* __restrict__ max_ratio_bits) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  float ratio = 0.f;
  if (row < rows) {
    const float* sp = stats + (size_t)row * nparts * 2;
    float su = 0.f, sq = 0.f;
    for (int q = 0; q < nparts; ++q) {
      su += sp[2 * q];
      sq += sp[2 * q + 1];
    }
    const float inv = 1.0f / (float)K;
    const float mu = su * inv;
    float var = sq * inv - mu * mu;
    var = var < 0.f ? 0.f : var;
    ratio = mu * mu / (var + 1e-5f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, o));
  if ((threadIdx.x & 63) == 0 && ratio > 0.f) atomicMax(max_ratio_bits, __float_as_uint(ratio));
}

// out[row, s*seg_len + k] = bf16( f_s(src_s[row, k]) ), f_s = identity / LayerNorm / LN*(1+scale)+shift->SiLU.
// The same transforms rg_gemm applies in its A prologue, done ONCE per element: used where a GEMM
// would otherwise redo an expensive prologue in every column tile (the K = 4*512 ca_mix GEMM).
// One wave per (row, segment), 8 elements per lane per 512 columns; HBM-bound.
struct StylizeArgs {
  rg_a_segment seg[RG_MAX_SEG];
  int nseg, seg_len, M, ldo;
  unsigned short* out;
  // classifier-free rows [m_cond, M): the first unc_nseg segments do not read their source but copy a
  // precomputed bf16 row (cross-attention output is a weights-only constant there):
  // unc_tab[flag][s*seg_len + k], flag = 1 on masked-query rows (qmask[s][row] == 0)
  int m_cond, unc_nseg;
  const unsigned short* unc_tab;
  const float* qmask;   // [nseg][M] or null
  // two groups of sequences at different diffusion steps in one batch (rows are [2 CFG halves][nseq sequences][T tokens]):
  // sequences >= split of either half take scale_shift from ss_b instead of seg[s].scale_shift; split >= nseq: one group
  const float* ss_b[RG_MAX_SEG];
  int T, nseq, split;
};

__global__ void __launch_bounds__(256) stylize_kernel(const StylizeArgs a) {
  typedef __attribute__((ext_vector_type(4))) float v4;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= a.M * a.nseg) return;
  const int row = wave / a.nseg, s = wave % a.nseg;
  rg_a_segment sg = a.seg[0];
  if (s == 1) sg = a.seg[1];
  if (s == 2) sg = a.seg[2];
  if (s == 3) sg = a.seg[3];
  unsigned short* dst = a.out + (size_t)row * a.ldo + s * a.seg_len;
  if (row >= a.m_cond && s < a.unc_nseg) {
    const int flag = (a.qmask && a.qmask[(size_t)s * a.M + row] == 0.f) ? 1 : 0;
    const unsigned short* tab = a.unc_tab + (size_t)flag * a.unc_nseg * a.seg_len + s * a.seg_len;
    for (int k = lane * 8; k < a.seg_len; k += 512) *reinterpret_cast<u4*>(dst + k) = *reinterpret_cast<const u4*>(tab + k);
    return;
  }
  const float* src = sg.src + (size_t)row * sg.ld;
  const bool norm = sg.mode != RG_A_IDENT, styl = sg.mode == RG_A_STYL;
  if (a.split < a.nseq && (row / a.T) % a.nseq >= a.split) {   // wave-uniform: a wave owns one row
    if (s == 0) sg.scale_shift = a.ss_b[0];
    if (s == 1) sg.scale_shift = a.ss_b[1];
    if (s == 2) sg.scale_shift = a.ss_b[2];
    if (s == 3) sg.scale_shift = a.ss_b[3];
  }
  float mu = 0.f, rs = 1.f;
  bool have_stats = !norm;
  for (int k = lane * 8; k < a.seg_len; k += 512) {
    // every load of the iteration is issued before the first value is needed: the row statistics (a second,
    // dependent round trip otherwise) are requested together with the row itself
    const v4 x0 = *reinterpret_cast<const v4*>(src + k), x1 = *reinterpret_cast<const v4*>(src + k + 4);
    v4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, c0 = b0, c1 = b0, h0 = b0, h1 = b0;
    if (norm) {
      g0 = *reinterpret_cast<const v4*>(sg.gamma + k); g1 = *reinterpret_cast<const v4*>(sg.gamma + k + 4);
      b0 = *reinterpret_cast<const v4*>(sg.beta + k); b1 = *reinterpret_cast<const v4*>(sg.beta + k + 4);
    }
    if (styl) {
      c0 = *reinterpret_cast<const v4*>(sg.scale_shift + k); c1 = *reinterpret_cast<const v4*>(sg.scale_shift + k + 4);
      h0 = *reinterpret_cast<const v4*>(sg.scale_shift + a.seg_len + k);
      h1 = *reinterpret_cast<const v4*>(sg.scale_shift + a.seg_len + k + 4);
    }
    if (!have_stats) {
      const float* sp = sg.stats + (size_t)row * sg.nparts * 2;
      float su = 0.f, sq = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {   // nparts <= 8 (checked by the wrapper): all requests go out together
        const float2 t = q < sg.nparts ? *reinterpret_cast<const float2*>(sp + 2 * q) : make_float2(0.f, 0.f);
        su += t.x;
        sq += t.y;
      }
      const float inv = 1.0f / (float)a.seg_len;
      mu = su * inv;
      float var = sq * inv - mu * mu;
      var = var < 0.f ? 0.f : var;
      rs = rsqrtf(var + 1e-5f);
      have_stats = true;
    }
    float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    if (norm) {
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const float be[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] - mu) * rs * g[e] + be[e];
      if (styl) {
        const float sc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = v[e] * (1.0f + sc[e]) + sh[e];
          v[e] = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.44269504088896340736f));
        }
      }
    }
    u4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = rg_pack2_bf16(v[2 * e], v[2 * e + 1]);
    }
    *reinterpret_cast<u4*>(dst + k) = o;
  }
}

// ------------------------------------------------------------------------------ cross attention + stylization
// One workgroup (8 waves) per (conditional row group b, condition c): all H heads of the group, so the
// LayerNorm statistics of the 512-wide cross-attention output never leave the CU:
//   y = softmax_hd(Q_c) A_c  (fp32 VALU, as rg_ca_attention), "-1e6" quantisation on masked query rows,
//   LN over D, * (1 + scale) + shift, SiLU, bf16  ->  out[b*T + n][c*D + col]
// i.e. rg_ca_attention + the cross-attention segments of rg_stylize in one launch: y3, its statistics and
// one kernel boundary disappear.  Workgroups [Rc*ncond, Rc*ncond + Ru) fill the classifier-free rows from
// the (step, layer) table like rg_stylize does.
struct CaStylizeArgs {
  const float* q3;        // [Rc*T][ncond*D] softmaxed queries (conditional rows only)
  const unsigned short* At;  // [ncond][Rc][H][2 (hi, lo)][32 l][32 d] bf16: A^T split (rg_split_transpose_bf16)
  const float* qmask;     // [ncond][Rc+Ru][T] or null
  const float* gamma;     // [ncond][D]   proj_out.norm of each condition
  const float* beta;      // [ncond][D]
  const float* scale_shift;  // [ncond][2*D]: (scale | shift) of each condition at this step
  const unsigned short* unc_tab;  // [2][ncond*D] bf16
  unsigned short* out;    // bf16 [(Rc+Ru)*T][ldo]
  int ldo, Rc, Ru, T, D, ncond;
  // row groups (sequences) >= split of either half are at another diffusion step: their scale/shift and table
  const float* scale_shift_b;
  const unsigned short* unc_tab_b;
  int split;
};

// At[m][0][l][d] = bf16(A[m][d][l]), At[m][1][l][d] = bf16(A[m][d][l] - float(hi)) for n_mat 32x32 matrices
__global__ void __launch_bounds__(256) split_transpose_kernel(const float* __restrict__ A, unsigned short* __restrict__ At,
                                                             int n_mat) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_mat * HD * HD) return;
  const int m = i / (HD * HD), l = (i / HD) % HD, d = i % HD;
  const float x = A[(size_t)m * HD * HD + d * HD + l];
  const __bf16 hi = (__bf16)x;
  const __bf16 lo = (__bf16)(x - (float)hi);
  At[(size_t)m * 2 * HD * HD + l * HD + d] = __builtin_bit_cast(unsigned short, hi);
  At[(size_t)m * 2 * HD * HD + HD * HD + l * HD + d] = __builtin_bit_cast(unsigned short, lo);
}

constexpr int CS_WAVES = 16;   // one wave per head at D = 512

__global__ void __launch_bounds__(CS_WAVES * 64) ca_stylize_kernel(const CaStylizeArgs a) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  constexpr int NTH = CS_WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T, D = a.D, H = D / HD, ncond = a.ncond;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Tp = (T + 7) & ~7;
  if ((int)blockIdx.x >= a.Rc * ncond) {
    // ---- classifier-free row group: copy the tabulated rows
    const int u = blockIdx.x - a.Rc * ncond;
    const int R = a.Rc + a.Ru;
    int* sflag = reinterpret_cast<int*>(sm);   // [ncond][T]: which of the two table rows a token takes
    for (int i = threadIdx.x; i < ncond * T; i += NTH) {
      const int cseg = i / T, n = i % T;
      sflag[i] = (a.qmask && a.qmask[((size_t)cseg * R + a.Rc + u) * T + n] == 0.f) ? 1 : 0;
    }
    __syncthreads();
    const int vec = ncond * D / 8;  // 16-B vectors per row
    unsigned short* orow = a.out + (size_t)(a.Rc + u) * T * a.ldo;
#pragma unroll 4
    for (int i = threadIdx.x; i < T * vec; i += NTH) {
      const int n = i / vec, k8 = (i % vec) * 8;
      const int flag = sflag[(k8 / D) * T + n];
      *reinterpret_cast<u4*>(orow + (size_t)n * a.ldo + k8) =
          *reinterpret_cast<const u4*>((u >= a.split ? a.unc_tab_b : a.unc_tab) + (size_t)flag * ncond * D + k8);
    }
    return;
  }
  RG_STAMP3(0);
  const int b = blockIdx.x / ncond, c = blockIdx.x % ncond;
  const float* scale_shift = b >= a.split ? a.scale_shift_b : a.scale_shift;
  float* sstat = sm;                           // [CS_WAVES][Tp][2]
  float* srow = sstat + CS_WAVES * 2 * Tp;     // [Tp][2] (mean, rstd)
  float* smask = srow + 2 * Tp;                // [Tp]
  float* tiles = smask + Tp;                   // [H][Tp][32]
  const int ld = ncond * D;
  const int hpw = H / CS_WAVES;                // heads per wave (1 at D = 512)
  // ---- q tiles of this wave's heads straight into LDS (8 rows x 128 B per 1-KiB LDS-DMA; register staging measured
  // equal: the 2.4 us before the first MFMA are the latency of the first dependent load, and the last of the 16
  // waves reaches the statistics barrier 2 us after the first -- in-kernel stamps, profiles/ca_stamps.py)
  for (int hh = 0; hh < hpw; ++hh) {
    const int h = wave * hpw + hh;
    const float* src = a.q3 + (size_t)b * T * ld + c * D + h * HD + (lane & 7) * 4;
    float* sq = tiles + (size_t)h * Tp * HD;
    for (int r0 = 0; r0 < Tp; r0 += 8) {
      int r = r0 + (lane >> 3);
      r = r < T ? r : T - 1;
      __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)r * ld), (lds_void*)(sq + r0 * HD), 16, 0, 0);
    }
  }
  // this wave's first head: B fragments of y = q A (A^T rows, bf16 hi/lo), in flight together with the tile
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int l15 = lane & 15, g = lane >> 4;
  bf16x8 bh[2], bl[2];
  auto load_b = [&](int h) {
    const unsigned short* Ap = a.At + ((((size_t)c * a.Rc + b) * H + h) * 2) * HD * HD;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      bh[nb] = *reinterpret_cast<const bf16x8*>(Ap + (16 * nb + l15) * HD + 8 * g);
      bl[nb] = *reinterpret_cast<const bf16x8*>(Ap + HD * HD + (16 * nb + l15) * HD + 8 * g);
    }
  };
  load_b(wave * hpw);
  // stylization parameters of this thread's column pair: requested now, used after the attention
  const int col0 = (threadIdx.x & 255) * 2;
  float g0 = 0.f, g1 = 0.f, b0 = 0.f, b1 = 0.f, sc0 = 0.f, sc1 = 0.f, sh0 = 0.f, sh1 = 0.f;
  if (col0 < D) {
    const float* ss = scale_shift + (size_t)c * 2 * D;
    const float2 gg = *reinterpret_cast<const float2*>(a.gamma + c * D + col0);
    const float2 bb = *reinterpret_cast<const float2*>(a.beta + c * D + col0);
    const float2 sc = *reinterpret_cast<const float2*>(ss + col0);
    const float2 sh = *reinterpret_cast<const float2*>(ss + D + col0);
    g0 = gg.x; g1 = gg.y; b0 = bb.x; b1 = bb.y; sc0 = 1.0f + sc.x; sc1 = 1.0f + sc.y; sh0 = sh.x; sh1 = sh.y;
  }
  const int R = a.Rc + a.Ru;
  // query mask of this (condition, row group) as a wave-uniform bit set (bit n: token n is NOT masked); a wave reads
  // only the tiles of its own heads until the statistics are combined, so no workgroup barrier is needed here
  unsigned long long qbits = ~0ull;
  if (a.qmask) {
    const float mv = a.qmask[((size_t)c * R + b) * T + (lane < T ? lane : 0)];
    qbits = __ballot(lane < T && mv != 0.f);
  }
  float* mystat = sstat + wave * 2 * Tp;
  for (int n = lane; n < Tp; n += 64) {
    mystat[2 * n] = 0.f;
    mystat[2 * n + 1] = 0.f;
  }
  RG_STAMP3(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  RG_STAMP3(2);
  auto pk = [](float x, float y) {
    return rg_pack2_bf16(x, y);
  };
  for (int hh = 0; hh < hpw; ++hh) {
    const int h = wave * hpw + hh;
    float* sq = tiles + (size_t)h * Tp * HD;
    if (hh > 0) load_b(h);
    // y[16 rb + 4g + e][16 nb + l15] = sum_d q[row][d] A[d][col]: v_mfma_f32_16x16x32_bf16 with q and A as
    // bf16 hi + lo pairs (hi*hi + hi*lo + lo*hi, fp32 accumulate: error ~2^-17 relative per product)
    const int nrb = Tp / 16;   // Tp is a multiple of 8; the tile has whole 16-row blocks when Tp % 16 == 0
    f32x4 acc[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      acc[rb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rb * 16 < Tp) {
        int row = 16 * rb + l15;
        row = row < Tp ? row : Tp - 1;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sq + row * HD + 8 * g);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(sq + row * HD + 8 * g + 4);
        const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = x[e] - (float)(__bf16)x[e];
        const bf16x8 ah = __builtin_bit_cast(bf16x8, u32x4{pk(x[0], x[1]), pk(x[2], x[3]), pk(x[4], x[5]), pk(x[6], x[7])});
        const bf16x8 al = __builtin_bit_cast(bf16x8, u32x4{pk(r[0], r[1]), pk(r[2], r[3]), pk(r[4], r[5]), pk(r[6], r[7])});
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nb], acc[rb][nb], 0, 0, 0);
          acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nb], acc[rb][nb], 0, 0, 0);
          acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nb], acc[rb][nb], 0, 0, 0);
        }
      }
    }
    (void)nrb;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // every lane's q reads are complete before y overwrites the tile
    __builtin_amdgcn_wave_barrier();
    RG_STAMP3(3);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * rb + 4 * g + e;
        if (row < T) {
          const bool masked = !((qbits >> row) & 1ull);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            float v = acc[rb][nb][e];
            if (masked) {   // fp32 rounding of the reference's y + (1 - query_mask) * -1e6
              const float z = v + (-1000000.0f);
              v = z + 1000000.0f;
            }
            sq[row * HD + 16 * nb + l15] = v;
          }
        }
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    RG_STAMP3(4);
    if (lane < T) {
      const float* yr = sq + lane * HD;
      float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int j = 0; j < HD; j += 2) {
        const float v0 = yr[(j + lane) & 31], v1 = yr[(j + 1 + lane) & 31];
        s0 += v0;
        s1 += v1;
        q0 = fmaf(v0, v0, q0);
        q1 = fmaf(v1, v1, q1);
      }
      mystat[2 * lane] += s0 + s1;
      mystat[2 * lane + 1] += q0 + q1;
    }
  }
  RG_STAMP3(5);
  __syncthreads();
  RG_STAMP3(6);
  for (int n = threadIdx.x; n < T; n += NTH) {
    float su = 0.f, sq2 = 0.f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; ++w) {
      su += sstat[w * 2 * Tp + 2 * n];
      sq2 += sstat[w * 2 * Tp + 2 * n + 1];
    }
    const float inv = 1.0f / (float)D;
    const float mu = su * inv;
    float var = sq2 * inv - mu * mu;
    var = var < 0.f ? 0.f : var;
    srow[2 * n] = mu;
    srow[2 * n + 1] = rsqrtf(var + 1e-5f);
  }
  __syncthreads();
  RG_STAMP3(7);
  // ---- LN + stylization + SiLU -> bf16; thread -> (two adjacent columns, every 4th token)
  const int rq = threadIdx.x >> 8;   // NTH / 256 = 4 row phases
  for (int col = col0; col < D; col += 512) {
    if (col != col0) {
      const float* ss = scale_shift + (size_t)c * 2 * D;
      g0 = a.gamma[c * D + col]; g1 = a.gamma[c * D + col + 1];
      b0 = a.beta[c * D + col]; b1 = a.beta[c * D + col + 1];
      sc0 = 1.0f + ss[col]; sc1 = 1.0f + ss[col + 1];
      sh0 = ss[D + col]; sh1 = ss[D + col + 1];
    }
    const float* yt = tiles + (size_t)(col >> 5) * Tp * HD + (col & 31);
    unsigned short* op = a.out + (size_t)b * T * a.ldo + c * D + col;
#pragma unroll 4
    for (int n = rq; n < T; n += NTH / 256) {
      const float mu = srow[2 * n], rs = srow[2 * n + 1];
      const float2 y = *reinterpret_cast<const float2*>(yt + n * HD);
      float t0 = ((y.x - mu) * rs * g0 + b0) * sc0 + sh0;
      float t1 = ((y.y - mu) * rs * g1 + b1) * sc1 + sh1;
      t0 = t0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t0 * -1.44269504088896340736f));
      t1 = t1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t1 * -1.44269504088896340736f));
      *reinterpret_cast<unsigned*>(op + (size_t)n * a.ldo) = rg_pack2_bf16(t0, t1);
    }
  }
  RG_STAMP3(8);
}

}  // namespace

extern "C" int rg_stylize(rg_handle* h, const rg_a_segment* segs_host, int nseg, int seg_len, int M, void* out_bf16,
                          int ldo, int m_cond, int unc_nseg, const void* unc_tab_bf16, const float* qmask,
                          void* stream) {
  return rg_stylize_groups(h, segs_host, nseg, seg_len, M, out_bf16, ldo, m_cond, unc_nseg, unc_tab_bf16, qmask, nullptr, 1, 1,
                           1, stream);
}

extern "C" int rg_stylize_groups(rg_handle* h, const rg_a_segment* segs_host, int nseg, int seg_len, int M, void* out_bf16,
                                 int ldo, int m_cond, int unc_nseg, const void* unc_tab_bf16, const float* qmask,
                                 const float* const* scale_shift_b_host, int T, int nseq, int split, void* stream) {
  RG_REQUIRE(h, segs_host && out_bf16, "null pointer");
  RG_REQUIRE(h, T > 0 && nseq > 0 && split >= 0 && (split >= nseq || scale_shift_b_host), "bad sequence groups");
  RG_REQUIRE(h, m_cond >= 0 && m_cond <= M && unc_nseg >= 0 && unc_nseg <= nseg && (m_cond == M || unc_nseg == 0 || unc_tab_bf16),
             "classifier-free rows need unc_tab");
  RG_REQUIRE(h, nseg >= 1 && nseg <= RG_MAX_SEG && seg_len % 8 == 0 && M > 0 && ldo % 8 == 0, "bad shape");
  StylizeArgs a;
  for (int s = 0; s < RG_MAX_SEG; ++s) a.seg[s] = segs_host[s < nseg ? s : 0];
  for (int s = 0; s < nseg; ++s) {
    RG_REQUIRE(h, a.seg[s].src && (a.seg[s].ld % 4) == 0, "segment source must be 16-B aligned rows");
    if (a.seg[s].mode != RG_A_IDENT)
      RG_REQUIRE(h, a.seg[s].stats && a.seg[s].gamma && a.seg[s].beta && a.seg[s].nparts >= 1 && a.seg[s].nparts <= 8,
                 "LN/STYL needs stats (1..8 partial sums per row)");
    if (a.seg[s].mode == RG_A_STYL) RG_REQUIRE(h, a.seg[s].scale_shift, "STYL needs scale_shift");
  }
  a.nseg = nseg; a.seg_len = seg_len; a.M = M; a.ldo = ldo;
  a.out = reinterpret_cast<unsigned short*>(out_bf16);
  a.m_cond = m_cond; a.unc_nseg = unc_nseg;
  a.unc_tab = reinterpret_cast<const unsigned short*>(unc_tab_bf16);
  a.qmask = qmask;
  a.T = T; a.nseq = nseq; a.split = split;
  for (int s = 0; s < RG_MAX_SEG; ++s) a.ss_b[s] = (split < nseq && s < nseg) ? scale_shift_b_host[s] : nullptr;
  const int64_t waves = (int64_t)M * nseg;
  hipLaunchKernelGGL(stylize_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_row_stats(rg_handle* h, const float* x, float* stats, int rows, int dim, void* stream) {
  RG_REQUIRE(h, x && stats, "null pointer");
  RG_REQUIRE(h, rows > 0 && dim > 0 && dim % 64 == 0, "bad shape");
  const int64_t waves = (int64_t)rows * (dim / 64);
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, rg_stream(stream), x,
                     stats, rows, dim);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_ln_guard(rg_handle* h, const float* stats, int rows, int nparts, int K, float* max_ratio, void* stream) {
  RG_REQUIRE(h, stats && max_ratio, "null pointer");
  RG_REQUIRE(h, rows > 0 && nparts > 0 && K > 0, "bad shape");
  hipLaunchKernelGGL(ln_guard_kernel, dim3((rows + 255) / 256), dim3(256), 0, rg_stream(stream), stats, rows, nparts, K,
                     reinterpret_cast<unsigned*>(max_ratio));
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

#ifdef RG_STAMPS
extern "C" int rg_debug_set_stamp_buffer3(void* dev_ptr) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf3), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int rg_sa_attention(rg_handle* h, const float* qkv, int ldqkv, const float* src_mask, void* y_out, int ldy,
                               float* stats, int R, int T, int D, const int* perm, int nperm, int mode,
                               void* stream) {
  float* y = reinterpret_cast<float*>(y_out);
  const bool use_mfma = mode != 0;
  RG_REQUIRE(h, qkv && src_mask && y && stats, "null pointer");
  RG_REQUIRE(h, mode >= 0 && mode <= 2, "mode: 0 fp32 VALU, 1 matrix cores, 2 matrix cores with a bf16 y");
  RG_REQUIRE(h, mode != 2 || (T <= 48 && ldy % 8 == 0), "bf16 y: T <= 48 and ldy % 8 == 0");
  RG_REQUIRE(h, T >= 33 && T <= TMAX && D % (HD * WAVES) == 0 && ldqkv % 4 == 0 && R > 0, "bad shape (33 <= T <= 64)");
  const int Tp = (T + 7) & ~7;
  const size_t lds = (WAVES * 2 * Tp + Tp + WAVES * (3 * Tp * HD)) * sizeof(float);
  RG_REQUIRE(h, !perm || nperm >= R * (D / (HD * WAVES)), "perm shorter than the work list");
  RG_REQUIRE(h, !use_mfma || (ldy % 4 == 0), "ldy must be a multiple of 4 floats");
  const dim3 grid(perm ? nperm : R * (D / (HD * WAVES)));
  if (mode == 2)
    hipLaunchKernelGGL((sa_attention_kernel<true, 48, true>), grid, dim3(256), lds, rg_stream(stream), qkv, ldqkv, D, src_mask,
                       y, ldy, stats, T, perm);
  else if (use_mfma && T <= 48)
    hipLaunchKernelGGL((sa_attention_kernel<true, 48>), grid, dim3(256), lds, rg_stream(stream), qkv, ldqkv, D, src_mask, y,
                       ldy, stats, T, perm);
  else if (use_mfma)
    hipLaunchKernelGGL(sa_attention_kernel<true>, grid, dim3(256), lds, rg_stream(stream), qkv, ldqkv, D, src_mask, y, ldy,
                       stats, T, perm);
  else
    hipLaunchKernelGGL(sa_attention_kernel<false>, grid, dim3(256), lds, rg_stream(stream), qkv, ldqkv, D, src_mask, y,
                       ldy, stats, T, perm);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_ca_attention(rg_handle* h, const float* q3, const float* Apre, const float* Aunc,
                               const float* qmask, float* y3, float* stats, int R, int Rc, int T, int D, int ncond,
                               const int* perm, int nperm, void* stream) {
  RG_REQUIRE(h, q3 && Apre && y3 && stats, "null pointer");
  RG_REQUIRE(h, Rc >= 0 && Rc <= R && (Rc == R || Aunc), "rows beyond Rc need Aunc");
  RG_REQUIRE(h, T > 0 && T <= TMAX && D % (HD * WAVES) == 0 && R > 0 && ncond > 0, "bad shape");
  const int Tp = (T + 7) & ~7;
  const size_t lds = (WAVES * 2 * Tp + Tp + WAVES * Tp * HD) * sizeof(float);
  RG_REQUIRE(h, !perm || nperm >= R * ncond * (D / (HD * WAVES)), "perm shorter than the work list");
  hipLaunchKernelGGL(ca_attention_kernel, dim3(perm ? nperm : R * ncond * (D / (HD * WAVES))), dim3(256), lds,
                     rg_stream(stream), q3, Apre, qmask, y3, stats, R, T, D, ncond, Rc, Aunc, perm);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_kv_reduce(rg_handle* h, const float* kv, int ldkv, float* A, int B, int N, int D, void* stream) {
  RG_REQUIRE(h, kv && A, "null pointer");
  RG_REQUIRE(h, B > 0 && N > 0 && D % HD == 0 && ldkv % 4 == 0, "bad shape");
  hipLaunchKernelGGL(kv_reduce_kernel, dim3(B * (D / HD)), dim3(256), 0, rg_stream(stream), kv, ldkv, D, N, A,
                     D / HD);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_linear_f32(rg_handle* h, const float* a, const float* w, const float* bias, float* out, int M,
                             int N, int K, int silu_in, int silu_out, void* stream) {
  RG_REQUIRE(h, a && w && out, "null pointer");
  RG_REQUIRE(h, M > 0 && N > 0 && K > 0, "bad shape");
  hipLaunchKernelGGL(linear_f32_kernel, dim3((N * 64 + 255) / 256), dim3(256), 0, rg_stream(stream), a, w, bias,
                     out, M, N, K, silu_in, silu_out);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_ca_stylize(rg_handle* h, const float* q3, const void* At_bf16, const float* qmask, const float* gamma,
                             const float* beta, const float* scale_shift, const void* unc_tab_bf16, void* out_bf16,
                             int ldo, int Rc, int Ru, int T, int D, int ncond, void* stream) {
  return rg_ca_stylize_groups(h, q3, At_bf16, qmask, gamma, beta, scale_shift, unc_tab_bf16, out_bf16, ldo, Rc, Ru, T, D, ncond,
                              nullptr, nullptr, Rc + Ru, stream);
}

extern "C" int rg_ca_stylize_groups(rg_handle* h, const float* q3, const void* At_bf16, const float* qmask, const float* gamma,
                                    const float* beta, const float* scale_shift, const void* unc_tab_bf16, void* out_bf16,
                                    int ldo, int Rc, int Ru, int T, int D, int ncond, const float* scale_shift_b,
                                    const void* unc_tab_b_bf16, int split, void* stream) {
  RG_REQUIRE(h, q3 && At_bf16 && gamma && beta && scale_shift && out_bf16, "null pointer");
  RG_REQUIRE(h, split >= 0 && ((split >= Rc && split >= Ru) || (scale_shift_b && (Ru == 0 || unc_tab_b_bf16))),
             "row groups from `split` on need their own scale_shift / table");
  RG_REQUIRE(h, Rc > 0 && Ru >= 0 && (Ru == 0 || unc_tab_bf16), "classifier-free rows need the table");
  RG_REQUIRE(h, T > 0 && T <= TMAX && D % (HD * CS_WAVES) == 0 && ncond > 0 && ldo % 8 == 0, "bad shape");
  const int Tp = (T + 7) & ~7;
  const size_t lds = ((size_t)CS_WAVES * 2 * Tp + 2 * Tp + Tp + (size_t)(D / HD) * Tp * HD) * sizeof(float);
  RG_REQUIRE(h, lds <= 160 * 1024, "row group does not fit LDS (D <= 768)");
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (ca_stylize_kernel), 160 * 1024);
  CaStylizeArgs a;
  a.q3 = q3; a.At = reinterpret_cast<const unsigned short*>(At_bf16); a.qmask = qmask; a.gamma = gamma; a.beta = beta; a.scale_shift = scale_shift;
  a.unc_tab = reinterpret_cast<const unsigned short*>(unc_tab_bf16);
  a.out = reinterpret_cast<unsigned short*>(out_bf16);
  a.ldo = ldo; a.Rc = Rc; a.Ru = Ru; a.T = T; a.D = D; a.ncond = ncond;
  a.scale_shift_b = scale_shift_b; a.unc_tab_b = reinterpret_cast<const unsigned short*>(unc_tab_b_bf16); a.split = split;
  hipLaunchKernelGGL(ca_stylize_kernel, dim3(Rc * ncond + Ru), dim3(CS_WAVES * 64), lds, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_split_transpose_bf16(rg_handle* h, const float* A, void* At_bf16, int n_mat, void* stream) {
  RG_REQUIRE(h, A && At_bf16 && n_mat > 0, "bad arguments");
  hipLaunchKernelGGL(split_transpose_kernel, dim3((n_mat * HD * HD + 255) / 256), dim3(256), 0, rg_stream(stream), A,
                     reinterpret_cast<unsigned short*>(At_bf16), n_mat);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

