// Fused bf16-MFMA GEMM for gfx950:  out[M,N] = epilogue( A'[M,K] * W[N,K]^T ).
//
// Shape regime (denoiser): M = rows*43 tokens (86 .. ~11k), N,K in {512..2048}.  These are
// small GEMMs: one 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each wave a
// 32x32 sub-tile = 2x2 MFMA 16x16x32 bf16 tiles), BK = 64, double-buffered LDS with register
// staging (global loads of tile t+1 are in flight under the MFMAs of tile t, the LDS write
// lands after them, one barrier per K-tile).
//
// LDS image: [64 rows][64 bf16] per operand per buffer, 128-B rows; the 16-B chunk index is
// XOR-swizzled with (row>>1)&7 so the ds_read_b128 fragment reads of a 16-lane group (16
// distinct rows, two adjacent chunks) hit 16 distinct 16-B slots of the 256-B bank row.
//
// A' is either bf16 in HBM or is built on the fly from fp32 sources while staging
// (identity cast / LayerNorm / StylizationBlock front half), see include/rg_gesture.h.
// The epilogue goes through LDS (reusing the staging buffers) so every thread owns 16
// consecutive columns of one row: bias, token-periodic bias, per-head softmax (32 columns =
// two threads, one shuffle), GELU, residual, per-row partial LayerNorm statistics, and 16-B
// coalesced stores.
//
// blockIdx -> tile: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs;
// tiles are numbered so that one XCD walks the N-tiles of one 64-row A panel (A panel stays in
// that XCD's L2; W is shared by all).  Pure speed: any placement is correct.
#include "rg_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 64, BN = 64, BK = 64, NT = 256;
constexpr int ROW_BYTES = BK * 2;            // 128
constexpr int TILE_BYTES = BM * ROW_BYTES;   // 8 KiB per operand per buffer
constexpr int SC_LD = 68;                    // fp32 epilogue tile row stride (floats)

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// SiLU on the A-operand prologue path: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the libm
// expf + IEEE divide; the result is rounded to bf16 (or split hi/lo) right after.
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

struct AStage {            // per-thread staging registers for one K-tile
  uint4 w[2];              // W chunks (bf16 x8)
  uint4 wlo[2];            // low-order W chunks (SPLIT)
  uint4 abf[2];            // A chunks when A is bf16
  float4 af[2][2];         // A chunks when A is fp32 (8 floats per chunk)
  float4 g[2], b[2], sc[2], sh[2];  // gamma / beta / scale / shift for this thread's 8 columns
};

// SPLIT: "bf16x3" precise mode.  Both operands are split into hi = bf16(v) and lo = bf16(v - hi)
// and the product is accumulated as hi*hi + hi*lo + lo*hi (fp32 accumulate): ~2^-17 relative
// operand precision at 3 MFMAs per tile, used to check parity against the fp32 reference.
template <bool A_BF16, bool SPLIT>
__global__ void __launch_bounds__(NT) gemm_kernel(const rg_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [buf][A|W(|Alo|Wlo)][8 KiB] = 32 (64) KiB; the epilogue reuses it as fp32 [64][68]
  constexpr int NPLANE = SPLIT ? 4 : 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware tile mapping
  const int mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  int grp = bid / (8 * nt);
  int rem_m = mt - grp * 8;
  if (rem_m > 8) rem_m = 8;
  int r = bid - grp * 8 * nt;
  const int tile_m = grp * 8 + r % rem_m;
  const int tile_n = r / rem_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging assignment: chunk id c = tid + 256*j -> row = c/8 (tid/8 + 32 j), kchunk = tid%8
  const int srow = tid >> 3;
  const int kch = tid & 7;
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(p.W);
  const int nk = (p.K + BK - 1) / BK;
  const int gboff = p.gb_group > 0 ? (n0 / p.gb_group) * p.gb_stride : 0;

  int arow[2];
  bool arow_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int gr = m0 + srow + 32 * j;
    arow_ok[j] = gr < p.M;
    if (!arow_ok[j]) gr = p.M - 1;
    arow[j] = p.a_row_mod > 0 ? gr % p.a_row_mod : gr;
  }

  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
  int cur_seg = -1;

  AStage st;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK + kch * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + srow + 32 * j;  // W is zero padded to a multiple of 64 rows / 64 cols
      st.w[j] = *reinterpret_cast<const uint4*>(Wb + (size_t)n * p.ldw + k0);
      if constexpr (SPLIT)
        st.wlo[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.W_lo) +
                                                    (size_t)n * p.ldw + k0);
    }
    if constexpr (A_BF16) {
      const unsigned short* Ab = reinterpret_cast<const unsigned short*>(p.A);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (arow_ok[j] && k0 + 8 <= p.K)
          st.abf[j] = *reinterpret_cast<const uint4*>(Ab + (size_t)arow[j] * p.lda + k0);
        else
          st.abf[j] = make_uint4(0, 0, 0, 0);
      }
    } else {
      const int sidx = (kt * BK) / p.seg_len;
      const rg_a_segment& sg = p.seg[sidx];
      const int ks = k0 - sidx * p.seg_len;  // column inside the segment
      if (sidx != cur_seg) {
        cur_seg = sidx;
        if (sg.mode != RG_A_IDENT) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float s = 0.f, ss = 0.f;
            const float* sp = sg.stats + (size_t)arow[j] * sg.nparts * 2;
            for (int q = 0; q < sg.nparts; ++q) {
              s += sp[2 * q];
              ss += sp[2 * q + 1];
            }
            const float inv = 1.0f / (float)p.seg_len;
            float mu = s * inv;
            float var = ss * inv - mu * mu;
            var = var < 0.f ? 0.f : var;
            mean[j] = mu;
            rstd[j] = rsqrtf(var + 1e-5f);
          }
        }
      }
      const bool fast = ((sg.ld & 3) == 0) && (k0 + 8 <= p.K);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* src = sg.src + (size_t)arow[j] * sg.ld + ks;
        if (fast) {
          st.af[j][0] = *reinterpret_cast<const float4*>(src);
          st.af[j][1] = *reinterpret_cast<const float4*>(src + 4);
        } else {
          float t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = (k0 + e < p.K) ? src[e] : 0.f;
          st.af[j][0] = make_float4(t[0], t[1], t[2], t[3]);
          st.af[j][1] = make_float4(t[4], t[5], t[6], t[7]);
        }
      }
      if (sg.mode != RG_A_IDENT) {
        const float* gp = sg.gamma + gboff + ks;
        const float* bp = sg.beta + gboff + ks;
        st.g[0] = *reinterpret_cast<const float4*>(gp);
        st.g[1] = *reinterpret_cast<const float4*>(gp + 4);
        st.b[0] = *reinterpret_cast<const float4*>(bp);
        st.b[1] = *reinterpret_cast<const float4*>(bp + 4);
        if (sg.mode == RG_A_STYL) {
          const float* sp = sg.scale_shift + ks;
          st.sc[0] = *reinterpret_cast<const float4*>(sp);
          st.sc[1] = *reinterpret_cast<const float4*>(sp + 4);
          st.sh[0] = *reinterpret_cast<const float4*>(sp + p.seg_len);
          st.sh[1] = *reinterpret_cast<const float4*>(sp + p.seg_len + 4);
        }
      }
    }
  };

  auto store_tile = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * NPLANE * TILE_BYTES;
    unsigned char* sW = sA + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = srow + 32 * j;
      *reinterpret_cast<uint4*>(sW + lds_off(row, kch)) = st.w[j];
      if constexpr (SPLIT) *reinterpret_cast<uint4*>(sW + 2 * TILE_BYTES + lds_off(row, kch)) = st.wlo[j];
      if constexpr (A_BF16) {
        *reinterpret_cast<uint4*>(sA + lds_off(row, kch)) = st.abf[j];
      } else {
        const int sidx = (kt * BK) / p.seg_len;
        const int mode = p.seg[sidx].mode;
        float v[8] = {st.af[j][0].x, st.af[j][0].y, st.af[j][0].z, st.af[j][0].w,
                      st.af[j][1].x, st.af[j][1].y, st.af[j][1].z, st.af[j][1].w};
        if (mode != RG_A_IDENT) {
          const float g[8] = {st.g[0].x, st.g[0].y, st.g[0].z, st.g[0].w, st.g[1].x, st.g[1].y, st.g[1].z, st.g[1].w};
          const float b[8] = {st.b[0].x, st.b[0].y, st.b[0].z, st.b[0].w, st.b[1].x, st.b[1].y, st.b[1].z, st.b[1].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean[j]) * rstd[j] * g[e] + b[e];
          if (mode == RG_A_STYL) {
            const float sc[8] = {st.sc[0].x, st.sc[0].y, st.sc[0].z, st.sc[0].w, st.sc[1].x, st.sc[1].y, st.sc[1].z, st.sc[1].w};
            const float sh[8] = {st.sh[0].x, st.sh[0].y, st.sh[0].z, st.sh[0].w, st.sh[1].x, st.sh[1].y, st.sh[1].z, st.sh[1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e] * (1.0f + sc[e]) + sh[e]);
          }
        }
        if (!arow_ok[j]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
        *reinterpret_cast<uint4*>(sA + lds_off(row, kch)) = o;
        if constexpr (SPLIT) {
          float r[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = v[e] - bf2f(f2bf(v[e]));
          uint4 ol = make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
          *reinterpret_cast<uint4*>(sA + 2 * TILE_BYTES + lds_off(row, kch)) = ol;
        }
      }
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile(0, 0);
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const unsigned char* sA = smem + buf * NPLANE * TILE_BYTES;
    const unsigned char* sW = sA + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sA + lds_off(wr * 32 + i * 16 + frow, 4 * s + fq));
        bfr[i] = *reinterpret_cast<const bf16x8*>(sW + lds_off(wc * 32 + i * 16 + frow, 4 * s + fq));
      }
      if constexpr (SPLIT) {
        bf16x8 al[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          al[i] = *reinterpret_cast<const bf16x8*>(sA + 2 * TILE_BYTES + lds_off(wr * 32 + i * 16 + frow, 4 * s + fq));
          bl[i] = *reinterpret_cast<const bf16x8*>(sW + 2 * TILE_BYTES + lds_off(wc * 32 + i * 16 + frow, 4 * s + fq));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bfr[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(kt + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue through LDS: sC[64][SC_LD] fp32
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        sC[(wr * 32 + i * 16 + fq * 4 + e) * SC_LD + wc * 32 + j * 16 + frow] = acc[i][j][e];
  __syncthreads();

  const int erow = tid >> 2;           // 0..63
  const int ecol = (tid & 3) * 16;     // 0,16,32,48
  const int grow = m0 + erow;
  const int gcol = n0 + ecol;
  const bool row_ok = grow < p.M;
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 t = *reinterpret_cast<const float4*>(sC + erow * SC_LD + ecol + 4 * q);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (gcol + e < p.N) ? p.bias[gcol + e] : 0.f;
  }
  if (p.tbias && row_ok) {
    const float* tb = p.tbias + (size_t)(grow % p.tb_period) * p.N + gcol;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (gcol + e < p.N) ? tb[e] : 0.f;
  }
  if (gcol < p.softmax_cols) {  // uniform over the pair of threads that share a 32-column head
    float mx = v[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, v[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { v[e] = expf(v[e] - mx); sum += v[e]; }
    sum += __shfl_xor(sum, 1);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] *= inv;
  }
  if (p.act == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = gelu_f(v[e]);
  } else if (p.act == 2) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (p.residual && row_ok) {
    const float* rp = p.residual + (size_t)grow * p.ldr + gcol;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (gcol + e < p.N) ? rp[e] : 0.f;
  }
  if (p.stats_out) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float t = (gcol + e < p.N) ? v[e] : 0.f;
      s += t; ss += t * t;
    }
    s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
    s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
    if ((tid & 3) == 0 && row_ok) {
      float* so = p.stats_out + ((size_t)grow * nt + tile_n) * 2;
      so[0] = s; so[1] = ss;
    }
  }
  if (!row_ok) return;
  const bool full = (gcol + 16 <= p.N);
  if (p.out_bf16) {
    unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)grow * p.ldo + gcol;
    if (full && (p.ldo & 7) == 0) {
      uint4 a = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
      uint4 b = make_uint4(pack2(v[8], v[9]), pack2(v[10], v[11]), pack2(v[12], v[13]), pack2(v[14], v[15]));
      reinterpret_cast<uint4*>(o)[0] = a;
      reinterpret_cast<uint4*>(o)[1] = b;
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) if (gcol + e < p.N) o[e] = f2bf(v[e]);
    }
  } else {
    float* o = reinterpret_cast<float*>(p.out) + (size_t)grow * p.ldo + gcol;
    if (full && (p.ldo & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        reinterpret_cast<float4*>(o)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) if (gcol + e < p.N) o[e] = v[e];
    }
  }
}

}  // namespace

extern "C" int rg_gemm(rg_handle* h, const rg_gemm_desc* d, void* stream) {
  RG_REQUIRE(h, d != nullptr, "null descriptor");
  RG_REQUIRE(h, d->M > 0 && d->N > 0 && d->K > 0, "empty problem");
  RG_REQUIRE(h, d->W && d->out, "null W/out");
  RG_REQUIRE(h, (d->ldw % 8) == 0 && d->ldw >= ((d->K + 63) / 64) * 64, "W must be K-padded to a multiple of 64");
  RG_REQUIRE(h, d->softmax_cols % 32 == 0, "softmax_cols must be a multiple of 32");
  if (d->a_is_bf16) {
    RG_REQUIRE(h, d->A != nullptr && (d->lda % 8) == 0, "bf16 A must have lda % 8 == 0");
  } else {
    RG_REQUIRE(h, d->nseg >= 1 && d->nseg <= RG_MAX_SEG && d->seg_len > 0 && d->seg_len % 64 == 0 ||
                      (d->nseg == 1 && d->seg_len >= d->K),
               "bad fp32 segment layout");
    RG_REQUIRE(h, (long)d->nseg * d->seg_len >= d->K, "segments do not cover K");
    for (int s = 0; s < d->nseg; ++s) {
      RG_REQUIRE(h, d->seg[s].src != nullptr, "null segment source");
      if (d->seg[s].mode != RG_A_IDENT)
        RG_REQUIRE(h, d->seg[s].stats && d->seg[s].gamma && d->seg[s].beta && d->seg[s].nparts > 0 &&
                          d->seg_len % 8 == 0 && d->K % 8 == 0,
                   "LN/STYL segment needs stats, gamma, beta");
      if (d->seg[s].mode == RG_A_STYL) RG_REQUIRE(h, d->seg[s].scale_shift, "STYL segment needs scale_shift");
    }
  }
  const int mt = (d->M + BM - 1) / BM, nt = (d->N + BN - 1) / BN;
  dim3 grid(mt * nt), block(NT);
  const size_t lds = 2 * 2 * TILE_BYTES;  // 32 KiB (>= 64*68*4 epilogue tile)
  rg_prof_rec rec;
  if (h->profiling) {
    auto get_ev = [&]() {
      hipEvent_t e;
      if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else { (void)hipEventCreate(&e); }
      return e;
    };
    rec.start = get_ev(); rec.stop = get_ev();
    rec.variant = d->W_lo ? 2 : (d->a_is_bf16 ? 1 : 0);
    rec.flops = 2.0 * (double)d->M * (double)d->N * (double)d->K;
    (void)hipEventRecord(rec.start, rg_stream(stream));
  }
  if (d->W_lo) {
    RG_REQUIRE(h, !d->a_is_bf16, "the split (bf16x3) mode needs fp32 A segments");
    hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 2 * lds, rg_stream(stream), *d);
  } else if (d->a_is_bf16) {
    hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, lds, rg_stream(stream), *d);
  } else {
    hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, lds, rg_stream(stream), *d);
  }
  RG_CHECK_LAUNCH(h);
  if (h->profiling) {
    (void)hipEventRecord(rec.stop, rg_stream(stream));
    h->prof.push_back(rec);
  }
  return RG_OK;
}

// HIP-event instrumentation of the GEMM launches between begin and end (bench.py's roofline
// figure).  Not capturable: use on eager launches only.
extern "C" int rg_profile_begin(rg_handle* h) {
  if (!h) return RG_ERR_INVALID;
  h->profiling = true;
  return RG_OK;
}

extern "C" int rg_profile_end(rg_handle* h, int variant, int64_t* launches, double* total_ms, double* total_flops) {
  if (!h) return RG_ERR_INVALID;
  h->profiling = false;
  if (hipDeviceSynchronize() != hipSuccess) return RG_ERR_HIP;
  int64_t n = 0;
  double ms = 0.0, fl = 0.0;
  for (auto& r : h->prof) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.start, r.stop);
    if (r.variant == variant) { n++; ms += t; fl += r.flops; }
    h->ev_pool.push_back(r.start);
    h->ev_pool.push_back(r.stop);
  }
  h->prof.clear();
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return RG_OK;
}
