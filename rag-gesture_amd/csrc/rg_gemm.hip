// Fused bf16-MFMA GEMM for gfx950:  out[M,N] = epilogue( A'[M,K] * W[N,K]^T ).
// GENERIC (register-staged) kernel: handles every shape incl. ragged K / unaligned rows; aligned
// shapes are routed to the LDS-DMA kernel in rg_gemm_dma.hip (same tile, same epilogue).
//
// Shape regime (denoiser / VAE): M = rows*43 tokens (86 .. ~11k), N,K in {512..2048}: small GEMMs
// whose cost is latency and staging, not MFMA issue.  Design for that regime:
//   * 64(M) x 128(N) output tile per 256-thread workgroup, 4 waves as 2x2, each wave 32x64 =
//     2x4 MFMA 16x16x32 bf16 tiles (fp32 accumulate);  BK = 64.
//   * A' is either bf16 in HBM or is built on the fly from fp32 sources while staging
//     (identity cast / LayerNorm / StylizationBlock front half LN*(1+scale)+shift -> SiLU).
//     With BN = 128 each fp32 element is transformed N/128 times, and the per-column
//     parameters (gamma, beta, scale, shift) and per-row (mean, rstd) are staged in LDS once per
//     workgroup instead of being re-fetched per K-tile.
//   * register-staged software pipeline, prefetch distance 2: global loads of K-tile t+2 are
//     issued before the MFMAs of tile t, the LDS write of tile t+1 lands after them; two LDS
//     buffers, one barrier per K-tile; two statically named register sets (no runtime-indexed
//     register arrays).
//   * LDS image [rows][64 bf16], 128-B rows, 16-B chunk index XOR-swizzled with (row>>1)&7 so
//     the ds_read_b128 fragment reads of a 16-lane group hit 16 distinct 16-B slots.
//   * epilogue through LDS (reusing the staging buffers): every thread owns 32 consecutive
//     columns of one row = one attention head for the per-head softmax; bias, token-periodic
//     bias, GELU/ReLU, residual, per-row partial LayerNorm statistics (one (sum,sumsq) pair per
//     128 columns) and 16-B coalesced stores.
//   * blockIdx -> tile: consecutive workgroups are dealt round-robin over the 8 XCDs; tiles are
//     numbered so that one XCD walks the N-tiles of one 64-row A panel (the panel stays in that
//     XCD's L2; W is shared by all).  Pure speed: any placement is correct.
//   * SPLIT ("bf16x3") precise mode: operands split into hi = bf16(v), lo = bf16(v - hi) and the
//     product accumulated as lo*hi + hi*lo + hi*hi: ~fp32 products at 3 MFMAs per tile, used to
//     check parity against the fp32 reference.
#include "rg_gemm_epi.h"

namespace {
using namespace rg_gemm_detail;

#ifdef RG_STAMPS
__device__ unsigned long long* g_stamp_buf2 = nullptr;
#define RG_STAMP2(slot)                                                                        \
  do {                                                                                         \
    if (g_stamp_buf2 && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))  \
      g_stamp_buf2[(blockIdx.x == 0 ? 0 : 64) + (slot)] = __builtin_amdgcn_s_memrealtime();    \
  } while (0)
#else
#define RG_STAMP2(slot)
#endif

template <bool A_BF16, bool SPLIT>
struct Stage {                 // one K-tile in flight in registers
  u32x4 w[4];
  u32x4 wlo[SPLIT ? 4 : 1];
  u32x4 abf[A_BF16 ? 2 : 1];
  f32x4 af[A_BF16 ? 1 : 2][2];
};

// FAST: K % 256 == 0 and 16-B aligned rows.  Loads are unconditional (rows beyond M are clamped, their
// results discarded by the epilogue) and the K loop is peeled so that no global load sits under a
// branch: hipcc's waitcnt insertion then emits counted vmcnt(N) waits; with predicated or conditional
// loads it falls back to vmcnt(0) at the LDS store and the software pipeline degenerates into one
// exposed memory latency per K-tile (measured ~1 us per tile).
template <bool A_BF16, bool SPLIT, bool FAST>
__global__ void __launch_bounds__(NT) gemm_kernel(const rg_gemm_group grp) {
  const rg_gemm_desc& p = grp.d[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NPL = SPLIT ? 2 : 1;                       // hi (+ lo) planes
  constexpr int STAGE_BYTES = NPL * (A_TILE + W_TILE);
  // LDS: [2 stages][A hi | W hi | (A lo | W lo)] | params [4 seg][4][SEG_MAX] f32 | rowstat [4][64][2] | seginfo[4]
  unsigned char* stage_base = smem;
  float* sPar = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);   // [nseg][4][SEG_MAX]
  float* sRow = sPar + (A_BF16 ? 0 : p.nseg) * 4 * SEG_MAX;         // [nseg][64][2]
  SegInfo* sSeg = reinterpret_cast<SegInfo*>(sRow + RG_MAX_SEG * BM * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware tile mapping
  const int mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
  int tile_m, tile_n;
  tile_of_block(blockIdx.x, mt, nt, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int srow = tid >> 3;   // staging row 0..31 (+32 j)
  const int kch = tid & 7;     // 16-B chunk (8 bf16) inside the 64-wide K-tile
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(p.W);
  const unsigned short* Wl = reinterpret_cast<const unsigned short*>(p.W_lo);
  const int nk = (p.K + BK - 1) / BK;

  int arow[2];
  bool arow_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int gr = m0 + srow + 32 * j;
    arow_ok[j] = gr < p.M;
    if (!arow_ok[j]) gr = p.M - 1;
    arow[j] = p.a_row_mod > 0 ? gr % p.a_row_mod : gr;
  }

  // ---- one-time LDS tables for the fp32 prologue
  if constexpr (!A_BF16) {
    if (tid == 0) {
      sSeg[0] = SegInfo{p.seg[0].src, p.seg[0].ld, p.seg[0].mode};
      sSeg[1] = SegInfo{p.seg[1].src, p.seg[1].ld, p.seg[1].mode};
      sSeg[2] = SegInfo{p.seg[2].src, p.seg[2].ld, p.seg[2].mode};
      sSeg[3] = SegInfo{p.seg[3].src, p.seg[3].ld, p.seg[3].mode};
    }
    const int gboff = p.gb_group > 0 ? (n0 / p.gb_group) * p.gb_stride : 0;
#pragma unroll
    for (int s = 0; s < RG_MAX_SEG; ++s) {
      if (s < p.nseg && p.seg[s].mode != RG_A_IDENT) {
        const rg_a_segment sg = p.seg[s];   // s is a compile-time constant here (unrolled)
        float* par = sPar + s * 4 * SEG_MAX;
        for (int i = tid; i < p.seg_len; i += NT) {
          par[i] = sg.gamma[gboff + i];
          par[SEG_MAX + i] = sg.beta[gboff + i];
          if (sg.mode == RG_A_STYL) {
            par[2 * SEG_MAX + i] = 1.0f + sg.scale_shift[i];
            par[3 * SEG_MAX + i] = sg.scale_shift[p.seg_len + i];
          }
        }
        if (tid < BM) {
          int gr = m0 + tid;
          if (gr >= p.M) gr = p.M - 1;
          const int ar = p.a_row_mod > 0 ? gr % p.a_row_mod : gr;
          const float* sp = sg.stats + (size_t)ar * sg.nparts * 2;
          float su = 0.f, sq = 0.f;
          for (int q = 0; q < sg.nparts; ++q) {
            su += sp[2 * q];
            sq += sp[2 * q + 1];
          }
          const float inv = 1.0f / (float)p.seg_len;
          const float mu = su * inv;
          float var = sq * inv - mu * mu;
          var = var < 0.f ? 0.f : var;
          sRow[(s * BM + tid) * 2] = mu;
          sRow[(s * BM + tid) * 2 + 1] = rsqrtf(var + 1e-5f);
        }
      }
    }
    __syncthreads();
  }

  using St = Stage<A_BF16, SPLIT>;

  auto load_tile = [&](St& st, int kt) {
    const int k0 = kt * BK + kch * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t off = (size_t)(n0 + srow + 32 * j) * p.ldw + k0;  // W zero-padded to 128 rows / 64 cols
      st.w[j] = *reinterpret_cast<const u32x4*>(Wb + off);
      if constexpr (SPLIT) st.wlo[j] = *reinterpret_cast<const u32x4*>(Wl + off);
    }
    if constexpr (A_BF16) {
      const unsigned short* Ab = reinterpret_cast<const unsigned short*>(p.A) +
                                 (p.gb_group > 0 ? (n0 / p.gb_group) * p.gb_stride : 0);   // grouped A (see header)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (FAST || (arow_ok[j] && k0 + 8 <= p.K))
          st.abf[j] = *reinterpret_cast<const u32x4*>(Ab + (size_t)arow[j] * p.lda + k0);
        else
          st.abf[j] = u32x4{0u, 0u, 0u, 0u};
      }
    } else {
      const int sidx = (kt * BK) / p.seg_len;
      const SegInfo sg = sSeg[sidx];
      const int ks = k0 - sidx * p.seg_len;
      const bool fast = FAST || (((sg.ld & 3) == 0) && (k0 + 8 <= p.K));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* src = sg.src + (size_t)arow[j] * sg.ld + ks;
        if (fast) {
          st.af[j][0] = *reinterpret_cast<const f32x4*>(src);
          st.af[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
        } else {
          const int rem = p.K - k0;
          st.af[j][0] = f32x4{rem > 0 ? src[0] : 0.f, rem > 1 ? src[1] : 0.f, rem > 2 ? src[2] : 0.f,
                              rem > 3 ? src[3] : 0.f};
          st.af[j][1] = f32x4{rem > 4 ? src[4] : 0.f, rem > 5 ? src[5] : 0.f, rem > 6 ? src[6] : 0.f,
                              rem > 7 ? src[7] : 0.f};
        }
      }
    }
  };

  auto store_tile = [&](const St& st, int kt, int buf) {
    unsigned char* sA = stage_base + buf * STAGE_BYTES;
    unsigned char* sW = sA + A_TILE;
    unsigned char* sAl = sW + W_TILE;
    unsigned char* sWl = sAl + A_TILE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = srow + 32 * j;
      *reinterpret_cast<u32x4*>(sW + lds_off(row, kch)) = st.w[j];
      if constexpr (SPLIT) *reinterpret_cast<u32x4*>(sWl + lds_off(row, kch)) = st.wlo[j];
    }
    if constexpr (A_BF16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(sA + lds_off(srow + 32 * j, kch)) = st.abf[j];
    } else {
      const int sidx = (kt * BK) / p.seg_len;
      const int mode = sSeg[sidx].mode;
      const int ks = kt * BK + kch * 8 - sidx * p.seg_len;
      float4 g0, g1, b0, b1, c0, c1, h0, h1;
      g0 = g1 = b0 = b1 = c0 = c1 = h0 = h1 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mode != RG_A_IDENT) {
        const float* par = sPar + sidx * 4 * SEG_MAX + ks;
        g0 = *reinterpret_cast<const float4*>(par);
        g1 = *reinterpret_cast<const float4*>(par + 4);
        b0 = *reinterpret_cast<const float4*>(par + SEG_MAX);
        b1 = *reinterpret_cast<const float4*>(par + SEG_MAX + 4);
        if (mode == RG_A_STYL) {
          c0 = *reinterpret_cast<const float4*>(par + 2 * SEG_MAX);
          c1 = *reinterpret_cast<const float4*>(par + 2 * SEG_MAX + 4);
          h0 = *reinterpret_cast<const float4*>(par + 3 * SEG_MAX);
          h1 = *reinterpret_cast<const float4*>(par + 3 * SEG_MAX + 4);
        }
      }
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = srow + 32 * j;
        float v[8] = {st.af[j][0][0], st.af[j][0][1], st.af[j][0][2], st.af[j][0][3],
                      st.af[j][1][0], st.af[j][1][1], st.af[j][1][2], st.af[j][1][3]};
        if (mode != RG_A_IDENT) {
          const float mu = sRow[(sidx * BM + row) * 2], rs = sRow[(sidx * BM + row) * 2 + 1];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (v[e] - mu) * rs * g[e] + be[e];
          if (mode == RG_A_STYL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e] * sc[e] + sh[e]);
          }
        }
        if (!arow_ok[j]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        const uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
        *reinterpret_cast<uint4*>(sA + lds_off(row, kch)) = o;
        if constexpr (SPLIT) {
          float r[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = v[e] - bf2f(f2bf(v[e]));
          const uint4 ol = make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
          *reinterpret_cast<uint4*>(sAl + lds_off(row, kch)) = ol;
        }
      }
    }
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const unsigned char* sA = stage_base + buf * STAGE_BYTES;
    const unsigned char* sW = sA + A_TILE;
    const unsigned char* sAl = sW + W_TILE;
    const unsigned char* sWl = sAl + A_TILE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2], bfr[4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sA + lds_off(wr * 32 + i * 16 + frow, 4 * s + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sW + lds_off(wc * 64 + j * 16 + frow, 4 * s + fq));
      if constexpr (SPLIT) {
        bf16x8 al[2], bl[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          al[i] = *reinterpret_cast<const bf16x8*>(sAl + lds_off(wr * 32 + i * 16 + frow, 4 * s + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bl[j] = *reinterpret_cast<const bf16x8*>(sWl + lds_off(wc * 64 + j * 16 + frow, 4 * s + fq));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bfr[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  RG_STAMP2(0);
  if constexpr (FAST) {
    // ---- prefetch distance 4, four statically named register sets, K loop in quads (nk % 4 == 0);
    //      the last quad is peeled (nothing left to prefetch), so every load is unconditional
    St s0, s1, s2, s3;
    load_tile(s0, 0); load_tile(s1, 1); load_tile(s2, 2); load_tile(s3, 3);
    store_tile(s0, 0, 0);
    __syncthreads();
    RG_STAMP2(1);
#define RG_STEP(FREE, NEXT, T)              \
    load_tile(FREE, (T) + 4);               \
    compute((T) & 1);                       \
    store_tile(NEXT, (T) + 1, ((T) + 1) & 1); \
    __syncthreads();
#define RG_TAIL(NEXT, T)                    \
    compute((T) & 1);                       \
    store_tile(NEXT, (T) + 1, ((T) + 1) & 1); \
    __syncthreads();
    int kt = 0;
    for (; kt + 4 < nk; kt += 4) {
      if (kt < 40) RG_STAMP2(2 + kt);
      RG_STEP(s0, s1, kt)
      RG_STEP(s1, s2, kt + 1)
      RG_STEP(s2, s3, kt + 2)
      RG_STEP(s3, s0, kt + 3)
    }
    if (kt < 40) RG_STAMP2(2 + kt);
    RG_TAIL(s1, kt)
    RG_TAIL(s2, kt + 1)
    RG_TAIL(s3, kt + 2)
    compute((kt + 3) & 1);
    __syncthreads();
#undef RG_STEP
#undef RG_TAIL
  } else {
  // ---- software pipeline: tile t lives in register set (t & 1); prefetch distance 2
  St st0, st1;
  load_tile(st0, 0);
  if (nk > 1) load_tile(st1, 1);
  store_tile(st0, 0, 0);
  __syncthreads();
  RG_STAMP2(1);
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt < 40) RG_STAMP2(2 + kt);
    if (kt + 2 < nk) load_tile(st0, kt + 2);
    compute(0);
    if (kt + 1 < nk) store_tile(st1, kt + 1, 1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt < 39) RG_STAMP2(3 + kt);
    if (kt + 3 < nk) load_tile(st1, kt + 3);
    compute(1);
    if (kt + 2 < nk) store_tile(st0, kt + 2, 0);
    __syncthreads();
  }
  }

  RG_STAMP2(60);
  // ---- epilogue through LDS: sC[64][SC_LD] fp32 (33.8 KiB <= 2 * STAGE_BYTES)
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        sC[(wr * 32 + i * 16 + fq * 4 + e) * SC_LD + wc * 64 + j * 16 + frow] = acc[i][j][e];
  __syncthreads();

  RG_STAMP2(61);
  epilogue(p, sC, tid, m0, n0, tile_n, nt);
  RG_STAMP2(62);
#ifdef RG_STAMPS
  if (g_stamp_buf2 && threadIdx.x == 0 && blockIdx.x == 0) g_stamp_buf2[126] = __builtin_amdgcn_s_memtime();
#endif
}

template <bool A_BF16, bool SPLIT>
size_t lds_bytes(int nseg) {
  const size_t stages = 2 * (SPLIT ? 2 : 1) * (A_TILE + W_TILE);
  const size_t tables = A_BF16 ? 0 : ((size_t)nseg * 4 * SEG_MAX * 4 + RG_MAX_SEG * BM * 2 * 4 + RG_MAX_SEG * sizeof(SegInfo));
  return stages + tables;
}

template <bool A_BF16, bool SPLIT, bool FAST>
void launch(const rg_gemm_desc* d, dim3 grid, hipStream_t s) {
  static rg_attr_once lds_once;
  const size_t lds = lds_bytes<A_BF16, SPLIT>(d->nseg);
  (void)rg_reserve_lds(lds_once, (gemm_kernel<A_BF16, SPLIT, FAST>), (int)lds_bytes<A_BF16, SPLIT>(RG_MAX_SEG));
  hipLaunchKernelGGL((gemm_kernel<A_BF16, SPLIT, FAST>), rg_group_grid(grid), dim3(NT), lds, s, rg_group_of(d));
}

// aligned shapes whose K loop splits into quads of 64-wide tiles
bool reg_fast_eligible(const rg_gemm_desc* d) {
  if (d->K % 256 != 0 || d->W_lo) return false;
  if (d->a_is_bf16) return (d->lda % 8) == 0;
  if (d->seg_len % 64 != 0) return false;
  for (int s = 0; s < d->nseg; ++s)
    if ((d->seg[s].ld % 4) != 0) return false;
  return true;
}

}  // namespace

bool rg_gemm_dma_eligible(const rg_gemm_desc* d);          // rg_gemm_dma.hip
void rg_gemm_dma_launch(const rg_gemm_desc* d, int num_cus, int waves, void* stream);
bool rg_gemm_big_eligible(const rg_gemm_desc* d);          // rg_gemm_big.hip
int rg_gemm_big_width(const rg_gemm_desc* d, int num_cus); // 0 = do not use, else 128 / 256
void rg_gemm_big_launch(const rg_gemm_desc* d, int bn, void* stream);

static int gemm_validate(rg_handle* h, const rg_gemm_desc* d) {
  RG_REQUIRE(h, d != nullptr, "null descriptor");
  RG_REQUIRE(h, d->M > 0 && d->N > 0 && d->K > 0, "empty problem");
  RG_REQUIRE(h, d->W && d->out, "null W/out");
  RG_REQUIRE(h, (d->ldw % 8) == 0 && d->ldw >= ((d->K + 63) / 64) * 64, "W must be K-padded to a multiple of 64");
  RG_REQUIRE(h, d->softmax_cols % 32 == 0, "softmax_cols must be a multiple of 32");
  if (d->a_is_bf16) {
    RG_REQUIRE(h, d->A != nullptr && (d->lda % 8) == 0, "bf16 A must have lda % 8 == 0");
    if (d->nseg != 0) {   // stylized bf16 A: seg[0] carries the LayerNorm statistics and parameters (header)
      RG_REQUIRE(h, rg_gemm_a_styl(d), "bf16 A takes nseg = 0, or nseg = 1 with a RG_A_STYL segment");
      const rg_a_segment& sg = d->seg[0];
      RG_REQUIRE(h, sg.stats && sg.gamma && sg.beta && !sg.scale_shift && sg.nparts > 0 && sg.nparts <= RG_MAX_LN_PARTS &&
                        ((uintptr_t)sg.gamma % 16) == 0 && ((uintptr_t)sg.beta % 16) == 0,
                 "stylized A needs stats (<= 8 partials per row) and 16-byte aligned folded gain / offset vectors");
      RG_REQUIRE(h, d->K <= SEG_MAX && d->K % 64 == 0 && d->a_row_mod == 0 && d->gb_group == 0 && d->tile_n != 64 &&
                        !d->W_lo && rg_gemm_dma_eligible(d),
                 "stylized A: K <= 512, K % 64 == 0, 16-byte aligned rows, default tiles");
    }
  } else {
    RG_REQUIRE(h, (d->nseg >= 1 && d->nseg <= RG_MAX_SEG && d->seg_len > 0 && d->seg_len % 64 == 0) ||
                      (d->nseg == 1 && d->seg_len >= d->K),
               "bad fp32 segment layout");
    RG_REQUIRE(h, (long)d->nseg * d->seg_len >= d->K, "segments do not cover K");
    for (int s = 0; s < d->nseg; ++s) {
      RG_REQUIRE(h, d->seg[s].src != nullptr, "null segment source");
      if (d->seg[s].mode != RG_A_IDENT)
        RG_REQUIRE(h, d->seg[s].stats && d->seg[s].gamma && d->seg[s].beta && d->seg[s].nparts > 0 &&
                          d->seg[s].nparts <= RG_MAX_LN_PARTS &&
                          d->seg_len % 8 == 0 && d->K % 8 == 0 && d->seg_len <= SEG_MAX,
                   "LN/STYL segment needs stats, gamma, beta and seg_len <= 512");
      if (d->seg[s].mode == RG_A_STYL) RG_REQUIRE(h, d->seg[s].scale_shift, "STYL segment needs scale_shift");
    }
  }
  if (d->W_lo) RG_REQUIRE(h, !d->a_is_bf16, "the split (bf16x3) mode needs fp32 A segments");
  if (d->tile_n == 64)
    RG_REQUIRE(h, d->a_is_bf16 && !d->W_lo && rg_gemm_dma_eligible(d) && d->N % 64 == 0 && d->split_col == 0,
               "tile_n = 64 needs a bf16 A operand, aligned shapes and N % 64 == 0");
  return RG_OK;
}

extern "C" int rg_gemm(rg_handle* h, const rg_gemm_desc* d, void* stream) {
  if (int rc = gemm_validate(h, d)) return rc;
  const int mt = (d->M + BM - 1) / BM, nt = (d->N + BN - 1) / BN;
  dim3 grid(mt * nt);
  rg_prof_rec rec;
  if (h->profiling) {
    auto get_ev = [&]() {
      hipEvent_t e;
      if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else { (void)hipEventCreate(&e); }
      return e;
    };
    rec.start = get_ev(); rec.stop = get_ev();
    rec.variant = d->W_lo ? 2 : (d->a_is_bf16 ? 1 : 0);
    rec.flops = 2.0 * (double)d->M * (double)d->N * (double)d->K * rg_group_n;
    (void)hipEventRecord(rec.start, rg_stream(stream));
  }
  // gemm_path: 0 = auto (LDS-DMA kernel where eligible, else generic),
  //            1 = generic only, 2 = prefer LDS-DMA, 3 = prefer register-staged FAST
  //            4 = prefer the 128-row big-tile kernel (bf16 A), 5 = never use it
  const int path = h->gemm_path;
  const bool fast_ok = reg_fast_eligible(d), dma_ok = rg_gemm_dma_eligible(d);
  const bool big_ok = rg_gemm_big_eligible(d);
  const int big_bn = big_ok ? (path == 4 ? (d->N >= 256 ? 256 : 128) : path == 6 ? 128 : path == 7 ? 129 : (path == 0 ? rg_gemm_big_width(d, h->num_cus) : 0)) : 0;
  if (d->tile_n == 64) {
    rg_gemm_dma_launch(d, h->num_cus, h->gemm_waves, stream);
  } else if (big_bn) {
    rg_gemm_big_launch(d, big_bn, stream);
  } else if (fast_ok && path == 3) {
    if (d->a_is_bf16) launch<true, false, true>(d, grid, rg_stream(stream));
    else launch<false, false, true>(d, grid, rg_stream(stream));
  } else if (path != 1 && dma_ok) {
    rg_gemm_dma_launch(d, h->num_cus, h->gemm_waves, stream);
  } else if (d->W_lo) {
    launch<false, true, false>(d, grid, rg_stream(stream));
  } else if (d->a_is_bf16) {
    launch<true, false, false>(d, grid, rg_stream(stream));
  } else {
    launch<false, false, false>(d, grid, rg_stream(stream));
  }
  RG_CHECK_LAUNCH(h);
  if (h->profiling) {
    (void)hipEventRecord(rec.stop, rg_stream(stream));
    h->prof.push_back(rec);
  }
  return RG_OK;
}

// n GEMMs of one shape signature (same M, N, K, operand kinds and epilogue features; pointers, leading dimensions,
// activation and output type per descriptor) in ONE launch: descriptor i is computed by the workgroups with blockIdx.y = i.
// Descriptors that do not share a signature (or more than RG_GEMM_GROUP of them) are launched one by one.
extern "C" int rg_gemm_grouped(rg_handle* h, const rg_gemm_desc* descs, int n, void* stream) {
  RG_REQUIRE(h, descs != nullptr && n >= 1, "rg_gemm_grouped: no descriptors");
  for (int i = 0; i < n; ++i)
    if (int rc = gemm_validate(h, &descs[i])) return rc;
  auto key = [&](const rg_gemm_desc& d, int* k) {
    int j = 0;
    k[j++] = d.M; k[j++] = d.N; k[j++] = d.K; k[j++] = d.a_is_bf16; k[j++] = d.nseg; k[j++] = d.seg_len; k[j++] = d.a_row_mod;
    k[j++] = d.gb_group; k[j++] = d.tile_n; k[j++] = d.split_col; k[j++] = d.ln_stats ? d.ln_nparts : -1; k[j++] = d.W_lo != nullptr;
    k[j++] = d.residual != nullptr; k[j++] = d.stats_out != nullptr; k[j++] = d.out2 != nullptr; k[j++] = d.softmax_cols;
    k[j++] = d.tbias != nullptr; k[j++] = d.bias != nullptr;
    for (int sg = 0; sg < RG_MAX_SEG; ++sg) { k[j++] = sg < d.nseg ? d.seg[sg].mode : -1; k[j++] = sg < d.nseg ? d.seg[sg].nparts : -1; }
    k[j++] = reg_fast_eligible(&d); k[j++] = rg_gemm_dma_eligible(&d); k[j++] = rg_gemm_big_eligible(&d);
    k[j++] = rg_gemm_big_eligible(&d) ? rg_gemm_big_width(&d, h->num_cus) : 0;
    return j;
  };
  bool same = n <= RG_GEMM_GROUP;
  int k0[64], ki[64];
  const int nk = key(descs[0], k0);
  for (int i = 1; i < n && same; ++i) {
    key(descs[i], ki);
    for (int j = 0; j < nk; ++j) same = same && ki[j] == k0[j];
  }
  if (!same || n == 1) {
    for (int i = 0; i < n; ++i)
      if (int rc = rg_gemm(h, &descs[i], stream)) return rc;
    return RG_OK;
  }
  rg_group_n = n;
  const int rc = rg_gemm(h, descs, stream);
  rg_group_n = 1;
  return rc;
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

// Test / tuning hook, see the dispatch in rg_gemm.
extern "C" int rg_set_gemm_path(rg_handle* h, int path) {
  if (!h || path < 0 || path > 7) return RG_ERR_INVALID;
  h->gemm_path = path;
  return RG_OK;
}

#ifdef RG_STAMPS
extern "C" int rg_debug_set_stamp_buffer2(void* dev_ptr) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf2), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -2;
}
#endif
