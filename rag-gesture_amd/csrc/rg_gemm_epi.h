// Shared pieces of the two GEMM kernels (rg_gemm.hip: register-staged generic kernel,
// rg_gemm_dma.hip: LDS-DMA kernel for aligned shapes): tile constants, bf16 helpers, and the
// epilogue that turns the fp32 accumulator tile sC[64][132] into the fused output.
#pragma once
#include "rg_common.h"

namespace rg_gemm_detail {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // native vectors: HIP's uint4/float4 wrapper
                                                             // structs keep staging sets in scratch

constexpr int BM = 64, BN = 128, BK = 64, NT = 256;
constexpr int ROW_BYTES = BK * 2;               // 128
constexpr int A_TILE = BM * ROW_BYTES;          // 8 KiB  (bf16)
constexpr int W_TILE = BN * ROW_BYTES;          // 16 KiB (bf16)
constexpr int SC_LD = BN + 4;                   // fp32 epilogue tile row stride (floats)
constexpr int SEG_MAX = 512;                    // widest fp32 segment with LN/STYL parameters

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return rg_pack2_bf16(lo, hi); }
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// SiLU on the A-operand prologue path: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of libm expf +
// IEEE divide; the result is rounded to bf16 (or split hi/lo) right after.
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// GELU for GEMMs whose operands are already bf16-rounded (relative 4e-3): erf by Abramowitz-Stegun 7.1.26 (absolute
// error 1.5e-7) on v_rcp_f32 / v_exp_f32 -- ~14 instructions instead of libm erff's ~45, 32 times per epilogue thread.
__device__ __forceinline__ float gelu_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = 1.0f - pl * t * __builtin_amdgcn_exp2f(x * x * -1.44269504088896340736f);   // erf(|v|/sqrt 2)
  return 0.5f * v + 0.5f * fabsf(v) * e;
}

// bf16 A whose rows are stylized inside the LDS-DMA kernel (see rg_gemm_desc.seg in the header)
inline bool rg_gemm_a_styl(const rg_gemm_desc* d) { return d->a_is_bf16 && d->nseg == 1 && d->seg[0].mode == RG_A_STYL; }

struct SegInfo {      // LDS copy of one rg_a_segment (dynamic indexing of kernargs would go to scratch)
  const float* src;
  int ld, mode;
};

// XCD-aware tile mapping: one XCD walks the N-tiles of one 64-row A panel.
__device__ __forceinline__ void tile_of_block(int bid, int mt, int nt, int& tile_m, int& tile_n) {
  const int grp = bid / (8 * nt);
  int rem_m = mt - grp * 8;
  if (rem_m > 8) rem_m = 8;
  const int rr = bid - grp * 8 * nt;
  tile_m = grp * 8 + rr % rem_m;
  tile_n = rr / rem_m;
}

// Every thread owns 32 consecutive columns of one row of sC (= one attention head).
// Residual values of this thread's 32 outputs, optionally loaded before the K loop so that their
// (cold) latency is hidden under the main loop instead of being exposed in the epilogue.
constexpr int RG_MAX_LN_PARTS = 8;   // partial (sum, sumsq) pairs per row a kernel requests in one batch

struct ResidualPrefetch {
  bool valid = false;
  float r[32];
};

// All partial (sum, sumsq) pairs of one row in ONE batch of independent loads (a loop over a runtime count compiles to
// one load + s_waitcnt per pair: nparts dependent round trips of ~1-2 us each).  nparts <= RG_MAX_LN_PARTS.
__device__ __forceinline__ void load_row_stats(const float* sp, int nparts, float (&raw)[2 * RG_MAX_LN_PARTS]) {
  // unconditional loads (index clamped into the row's partials, surplus values deselected afterwards): a guarded load
  // would put every request into a basic block of its own, with its own s_waitcnt
#pragma unroll
  for (int q = 0; q < RG_MAX_LN_PARTS; ++q) {
    const float2 t = *reinterpret_cast<const float2*>(sp + 2 * (q < nparts ? q : nparts - 1));
    raw[2 * q] = t.x;
    raw[2 * q + 1] = t.y;
  }
}
// (sum, sumsq) partials of one row in LDS -> (mean, rstd)
__device__ __forceinline__ void lds_row_stats(const float* sp, int nparts, int K, float& mu, float& rs) {
  float su = 0.f, sq = 0.f;
#pragma unroll
  for (int q = 0; q < RG_MAX_LN_PARTS; ++q) {
    const float2 t = *reinterpret_cast<const float2*>(sp + 2 * (q < nparts ? q : nparts - 1));
    su += q < nparts ? t.x : 0.f;
    sq += q < nparts ? t.y : 0.f;
  }
  const float inv = 1.0f / (float)K;
  mu = su * inv;
  float var = sq * inv - mu * mu;
  var = var < 0.f ? 0.f : var;
  rs = rsqrtf(var + 1e-5f);
}

// Up to RG_GEMM_GROUP independent GEMMs of the SAME shape signature in one launch (blockIdx.y picks the descriptor): the four
// body-part VAEs run the same layer on different weights and rows (rg_gemm_grouped).  Every kernel takes the group by
// value and reads its descriptor from the kernel arguments, exactly as it read the single descriptor before.
#define RG_GEMM_GROUP 4
struct rg_gemm_group { rg_gemm_desc d[RG_GEMM_GROUP]; };
inline thread_local int rg_group_n = 1;      // descriptors behind the pointer every launcher receives (set by rg_gemm_grouped)
inline rg_gemm_group rg_group_of(const rg_gemm_desc* d) {
  rg_gemm_group g;
  for (int i = 0; i < RG_GEMM_GROUP; ++i) g.d[i] = d[i < rg_group_n ? i : 0];
  return g;
}
inline dim3 rg_group_grid(dim3 grid) { grid.y = rg_group_n; return grid; }

// statistics partials per row that the LDS-DMA kernel stages for a descriptor with a bf16 A operand
__host__ __device__ inline int dma_stat_parts(const rg_gemm_desc& d) {
  if (d.a_is_bf16 && d.nseg == 1 && d.seg[0].mode == RG_A_STYL) return d.seg[0].nparts;
  return (d.ln_stats && d.ln_nparts <= RG_MAX_LN_PARTS) ? d.ln_nparts : 0;
}

// Pin a loaded value to this point of the program: the compiler otherwise hoists its first arithmetic use up to the
// load (and waits for the round trip there, in front of everything issued afterwards).
__device__ __forceinline__ void use_here(float& x) { asm volatile("" : "+v"(x)); }

// (the deselection lives here, not next to the loads: the first use of a loaded value is where the compiler waits)
__device__ __forceinline__ void reduce_row_stats(float (&raw)[2 * RG_MAX_LN_PARTS], int nparts, int K, float& mu,
                                                 float& rs) {
#pragma unroll
  for (int q = 0; q < 2 * RG_MAX_LN_PARTS; ++q) use_here(raw[q]);
  float su = 0.f, sq = 0.f;
#pragma unroll
  for (int q = 0; q < RG_MAX_LN_PARTS; ++q) {
    su += q < nparts ? raw[2 * q] : 0.f;
    sq += q < nparts ? raw[2 * q + 1] : 0.f;
  }
  const float inv = 1.0f / (float)K;
  mu = su * inv;
  float var = sq * inv - mu * mu;
  var = var < 0.f ? 0.f : var;
  rs = rsqrtf(var + 1e-5f);
}

__device__ __forceinline__ void prefetch_residual(const rg_gemm_desc& p, int tid, int m0, int n0, ResidualPrefetch& pre,
                                                  int width = BN) {
  const int grow = m0 + (tid >> 2), gcol = n0 + (tid & 3) * 32;
  pre.valid = false;
  if (p.residual && grow < p.M && gcol + 32 <= p.N && gcol + 32 <= n0 + width && (p.ldr & 3) == 0) {
    const float* rp = p.residual + (size_t)grow * p.ldr + gcol;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(rp + 4 * q);
      pre.r[4 * q] = t[0]; pre.r[4 * q + 1] = t[1]; pre.r[4 * q + 2] = t[2]; pre.r[4 * q + 3] = t[3];
    }
    pre.valid = true;
  }
}

// `width` = columns of sC that belong to this tile (128, or 64 for the narrow-tile kernel: the threads of the
// upper two column groups then only take part in the shuffles); nt / tile_n count tiles of that width.
// ln_row: null, or LDS [BM][ln_nparts][2]: the partial statistics of this tile's rows for the folded LayerNorm (the LDS-DMA
// kernel fetches them with its operand tiles); otherwise they are fetched here, all partials of the row in one batch.
__device__ __forceinline__ void epilogue(const rg_gemm_desc& p, const float* sC, int tid, int m0, int n0, int tile_n,
                                         int nt, const ResidualPrefetch* pre = nullptr, int width = BN,
                                         const float* ln_row = nullptr) {
  const int erow = tid >> 2;           // 0..63
  const int ecol = (tid & 3) * 32;     // 0,32,64,96 : one 32-column head per thread
  const int grow = m0 + erow;
  const int gcol = n0 + ecol;
  const bool row_ok = grow < p.M;
  const int Nlim = min(p.N, n0 + width);   // first column that is not this tile's
  float v[32];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(sC + erow * SC_LD + ecol + 4 * q);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
  const bool full = (gcol + 32 <= Nlim);
  if (p.ln_stats) {   // LayerNorm of the A rows, folded: rstd * (acc - mean * c1[n]); the bias below carries beta
    const float* sp = p.ln_stats + (size_t)(row_ok ? grow : 0) * p.ln_nparts * 2;
    float mu, rs;
    if (ln_row) {
      lds_row_stats(ln_row + erow * 2 * p.ln_nparts, p.ln_nparts, p.K, mu, rs);
    } else if (p.ln_nparts <= RG_MAX_LN_PARTS) {
      float raw[2 * RG_MAX_LN_PARTS];
      load_row_stats(sp, p.ln_nparts, raw);
      reduce_row_stats(raw, p.ln_nparts, p.K, mu, rs);
    } else {
      float su = 0.f, sq = 0.f;
      for (int q = 0; q < p.ln_nparts; ++q) {
        su += sp[2 * q];
        sq += sp[2 * q + 1];
      }
      const float inv = 1.0f / (float)p.K;
      mu = su * inv;
      float var = sq * inv - mu * mu;
      var = var < 0.f ? 0.f : var;
      rs = rsqrtf(var + 1e-5f);
    }
    if (full) {   // whole 32-column group inside the matrix: eight 16-B loads instead of 32 guarded ones
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 c = *reinterpret_cast<const float4*>(p.ln_c1 + gcol + 4 * q);
        v[4 * q] = rs * (v[4 * q] - mu * c.x);
        v[4 * q + 1] = rs * (v[4 * q + 1] - mu * c.y);
        v[4 * q + 2] = rs * (v[4 * q + 2] - mu * c.z);
        v[4 * q + 3] = rs * (v[4 * q + 3] - mu * c.w);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const float c1 = (gcol + e < Nlim) ? p.ln_c1[gcol + e] : 0.f;
        v[e] = rs * (v[e] - mu * c1);
      }
    }
  }
  if (p.bias) {
    if (full) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p.bias + gcol + 4 * q);
        v[4 * q] += t.x; v[4 * q + 1] += t.y; v[4 * q + 2] += t.z; v[4 * q + 3] += t.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) v[e] += (gcol + e < Nlim) ? p.bias[gcol + e] : 0.f;
    }
  }
  if (p.tbias && row_ok) {
    const float* tb = p.tbias + (size_t)(grow % p.tb_period) * p.N + gcol;
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] += (gcol + e < Nlim) ? tb[e] : 0.f;
  }
  // bf16-operand GEMMs (no hi/lo weight planes): exp / erf to ~1e-6, far below the operand rounding; the fp32-
  // equivalent mode keeps libm
  const bool fast_math = p.W_lo == nullptr;
  if (gcol < p.softmax_cols) {  // this thread's 32 columns are exactly one head
    float mx = v[0];
#pragma unroll
    for (int e = 1; e < 32; ++e) mx = fmaxf(mx, v[e]);
    float sum = 0.f;
    if (fast_math) {
#pragma unroll
      for (int e = 0; e < 32; ++e) { v[e] = __expf(v[e] - mx); sum += v[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) { v[e] = expf(v[e] - mx); sum += v[e]; }
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] *= inv;
  }
  if (p.act == 1) {
    if (fast_math) {
#pragma unroll
      for (int e = 0; e < 32; ++e) v[e] = gelu_fast(v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) v[e] = gelu_f(v[e]);
    }
  } else if (p.act == 2) {
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (pre && pre->valid) {
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] += pre->r[e];
  } else if (p.residual && row_ok) {
    const float* rp = p.residual + (size_t)grow * p.ldr + gcol;
    if (full && (p.ldr & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(rp + 4 * q);
        v[4 * q] += t.x; v[4 * q + 1] += t.y; v[4 * q + 2] += t.z; v[4 * q + 3] += t.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) v[e] += (gcol + e < Nlim) ? rp[e] : 0.f;
    }
  }
  if (p.stats_out) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const float t = (gcol + e < Nlim) ? v[e] : 0.f;
      s += t; ss += t * t;
    }
    s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
    s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
    if ((tid & 3) == 0 && row_ok) {
      float* so = p.stats_out + ((size_t)grow * nt + tile_n) * 2;
      so[0] = s; so[1] = ss;
    }
  }
  // ---- stores.  Fast path (whole 128-column tile in range, aligned rows): the final values go back to this
  // thread's own 32 floats of sC; the wave's 64 threads own rows [16w, 16w+16) x all 128 columns of the
  // tile, so after a wave-level fence the wave streams them out with lanes running ALONG the row: every
  // wave-instruction writes whole contiguous rows (2 x 512 B fp32 or 4 x 256 B bf16) instead of 64
  // scattered 16-B pieces at a 128-B stride (measured 3 us -> 1 us per 64x128 tile).
  const bool tile_full = (n0 + width <= p.N);
  // split output: this tile's columns go either to out (below split_col) or to out2 (bf16, from split_col on)
  const bool to_o2_only = p.split_col > 0 && n0 >= p.split_col;
  const bool want_main = !to_o2_only;
  const bool want_o2 = p.out2 && (p.split_col == 0 || to_o2_only);
  const int o2c0 = p.split_col > 0 ? p.split_col : 0;   // column origin of out2
  const bool fast_f32 = want_main && tile_full && !p.out_bf16 && (p.ldo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0;
  const bool fast_bf16 = want_main && tile_full && p.out_bf16 && (p.ldo & 7) == 0 && (((uintptr_t)p.out) & 15) == 0;
  const bool fast_o2 = want_o2 && tile_full && (p.ldo2 & 7) == 0 && (((uintptr_t)p.out2) & 15) == 0 && (o2c0 & 7) == 0;
  if (fast_f32 || fast_bf16 || fast_o2) {
    float* mine = const_cast<float*>(sC) + erow * SC_LD + ecol;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      *reinterpret_cast<float4*>(mine + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const int lane = tid & 63, wrow0 = (tid >> 6) * 16;
    if (fast_f32) {
      const int c4 = (lane & 31) * 4, rpar = lane >> 5;
      float* o = reinterpret_cast<float*>(p.out) + n0 + c4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = wrow0 + 2 * i + rpar;
        if (m0 + r < p.M && c4 < width)
          *reinterpret_cast<float4*>(o + (size_t)(m0 + r) * p.ldo) = *reinterpret_cast<const float4*>(sC + r * SC_LD + c4);
      }
    }
    if (fast_bf16 || fast_o2) {
      const int c8 = (lane & 15) * 8, rq = lane >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wrow0 + 4 * i + rq;
        if (m0 + r < p.M && c8 < width) {
          const float4 x0 = *reinterpret_cast<const float4*>(sC + r * SC_LD + c8);
          const float4 x1 = *reinterpret_cast<const float4*>(sC + r * SC_LD + c8 + 4);
          const uint4 pk = make_uint4(pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w));
          if (fast_bf16)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)(m0 + r) * p.ldo + n0 + c8) = pk;
          if (fast_o2)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.out2) + (size_t)(m0 + r) * p.ldo2 + n0 - o2c0 + c8) = pk;
        }
      }
    }
  }
  if (!row_ok) return;
  if (want_o2 && !fast_o2) {
    unsigned short* o2 = reinterpret_cast<unsigned short*>(p.out2) + (size_t)grow * p.ldo2 + gcol - o2c0;
    if (full && (p.ldo2 & 7) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        reinterpret_cast<uint4*>(o2)[q] = make_uint4(pack2(v[8 * q], v[8 * q + 1]), pack2(v[8 * q + 2], v[8 * q + 3]),
                                                     pack2(v[8 * q + 4], v[8 * q + 5]), pack2(v[8 * q + 6], v[8 * q + 7]));
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) if (gcol + e < Nlim) o2[e] = f2bf(v[e]);
    }
  }
  if (fast_f32 || fast_bf16 || !want_main) return;
  if (p.out_bf16) {
    unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)grow * p.ldo + gcol;
    if (full && (p.ldo & 7) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        reinterpret_cast<uint4*>(o)[q] = make_uint4(pack2(v[8 * q], v[8 * q + 1]), pack2(v[8 * q + 2], v[8 * q + 3]),
                                                    pack2(v[8 * q + 4], v[8 * q + 5]), pack2(v[8 * q + 6], v[8 * q + 7]));
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) if (gcol + e < Nlim) o[e] = f2bf(v[e]);
    }
  } else {
    float* o = reinterpret_cast<float*>(p.out) + (size_t)grow * p.ldo + gcol;
    if (full && (p.ldo & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        reinterpret_cast<float4*>(o)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
      for (int e = 0; e < 32; ++e) if (gcol + e < Nlim) o[e] = v[e];
    }
  }
}

}  // namespace rg_gemm_detail
