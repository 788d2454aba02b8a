// LDS-DMA variant of the fused bf16-MFMA GEMM (same 64x128x64 tile and epilogue as rg_gemm.hip)
// for aligned shapes (K % 64 == 0, 16-B aligned rows).
//
// Why: at M <= ~3k token rows these GEMMs launch fewer workgroups than the chip has CUs, so the K
// loop of a workgroup is a pure latency chain (measured: ~1 us per K-tile with a 2-deep register
// pipeline, i.e. HBM/L2 latency / prefetch depth).  Deep prefetch needs bytes in flight without
// VGPRs: every operand tile is fetched with global_load_lds_dwordx4 (1 KiB per wave-instruction,
// lane-linear LDS destination) into a 4-stage LDS ring, with a counted s_waitcnt vmcnt(N) and a
// raw s_barrier per K-tile (a __syncthreads() would drain the DMA queue).
//   * bf16 operands (W always, A when it is bf16): the XOR swizzle of the ds_read_b128 fragment
//     reads is applied on the SOURCE address (which 16-B chunk of the row a lane fetches), the LDS
//     image stays lane-linear.
//   * fp32 A sources are DMA'd raw ([64][64] fp32 per stage, chunk index XOR row&15) and
//     converted when the MFMA fragment is read: LayerNorm / stylization / SiLU run on the 32
//     values a lane feeds to the matrix core per K-tile, with gamma/beta/(1+scale)/shift and the
//     per-row (rstd, -mean*rstd) staged in LDS once per workgroup.
//   * SPLIT (bf16x3) precise mode: W hi/lo planes are both DMA'd, A hi/lo are formed in registers.
#include "rg_gemm_epi.h"

namespace {
using namespace rg_gemm_detail;

typedef __attribute__((address_space(3))) void lds_void;

#ifdef RG_STAMPS
// Diagnostic build only (build.py RG_DIAG=1 -> librg_gesture_diag.so): wall-clock stamps of the phases
// of workgroup 0 and of the last workgroup go to a buffer of their own; no output depends on them.
__device__ unsigned long long* g_stamp_buf = nullptr;
#define RG_STAMP(slot)                                                                         \
  do {                                                                                         \
    if (g_stamp_buf && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))   \
      g_stamp_buf[(blockIdx.x == 0 ? 0 : 64) + (slot)] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
__device__ int g_dbg_mode = 0;   // bit0: skip MFMA/fragment reads, bit1: skip in-loop DMA, bit2: skip epilogue
#define RG_DBG(bit) (dbg_mode & (bit))
#define RG_DBG_LOAD() const int dbg_mode = g_dbg_mode
#else
#define RG_STAMP(slot)
#define RG_DBG(bit) 0
#define RG_DBG_LOAD()
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// wait until at most `younger` tiles (PT DMA instructions each) issued after the current one are still in flight
template <int PT, int MAXY>
__device__ __forceinline__ void wait_tiles(int younger) {
  if constexpr (MAXY <= 0) {
    wait_vmcnt<0>();
  } else {
    if (younger >= MAXY) wait_vmcnt<MAXY * PT>();
    else wait_tiles<PT, MAXY - 1>(younger);
  }
}

// NW = waves per workgroup.  With 4 waves (one per SIMD) a K-tile costs ~1500 cycles although its
// MFMAs need 256: the DMA issue (60-185 cycles per piece) and the LDS fragment-read latency are fully
// exposed because nothing else can issue on the SIMD.  With 8 waves (2x4 over the 64x128 tile, 32x32
// per wave) each SIMD holds two waves that cover each other's DMA issue and LDS waits; the epilogue
// is still run by the first 256 threads (32 columns per thread).
// BN_ = tile width: 128, or 64 for single-round N = 512 GEMMs at small M.  There a workgroup is alone on its CU
// and pulls its weight panel from the Infinity Cache at the per-CU rate (~33 GB/s: every XCD sees a panel once,
// L2 reuse is nil at 1-2 M-tiles per XCD); 64-wide tiles put twice as many CUs on the same bytes.
// STYL (bf16 A only; p.seg[0] carries stats / gamma / beta / scale_shift, K <= SEG_MAX): the A rows are the bf16 copy of a
// block output y and the operand is SiLU(LN(y) * (1 + scale) + shift) (StylizationBlock front half).  A landed K-tile
// is rewritten in place in LDS, ONCE per element by the whole workgroup (8 values per thread and K-tile), one tile ahead
// of the MFMAs and between the same two barriers -- instead of a separate elementwise launch in front of the GEMM
// (4.9 us per launch at M = 1376) or of the fragment-read prologue of the fp32-A kernel, which redoes the two
// transcendentals in every wave column (x4) and fetches fp32 rows.
template <bool A_BF16, bool SPLIT, int NS, int NW, int BN_ = BN, bool STYL = false, int MINW = 1>
__global__ void __launch_bounds__(NW * 64, MINW) gemm_dma_kernel(const rg_gemm_group grp) {
  const rg_gemm_desc& p = grp.d[blockIdx.y];
  static_assert(!STYL || (A_BF16 && !SPLIT && NS >= 3), "STYL: bf16 A, one weight plane, ring of >= 3");
  constexpr int NTH = NW * 64;
  constexpr int WN = NW / 2;            // waves along N (2 along M)
  constexpr int TN = (BN_ / WN) / 16;   // 16-column MFMA tiles per wave: 4 (NW=4) or 2 (NW=8) at BN_ = 128
  constexpr int CPW = 16 / NW;          // 1-KiB DMA chunks of a 16-chunk fp32 A tile per wave
  constexpr int W_TILE_ = BN_ * ROW_BYTES;
  constexpr int CPW_W = (BN_ / 8) / NW; // 1-KiB chunks of the W tile per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_STAGE = A_BF16 ? A_TILE : 2 * A_TILE;           // fp32 tile = 16 KiB
  constexpr int W_PLANES = SPLIT ? 2 : 1;
  constexpr int STAGE = A_STAGE + W_PLANES * W_TILE_;
  constexpr int ACH = A_BF16 ? (NW == 4 ? 2 : 1) : CPW;        // A chunks per wave
  constexpr int PER_TILE = ACH + CPW_W * W_PLANES;                 // DMA instructions per wave per K-tile
  // behind the ring: sStat [BM][parts][2] = the partial (sum, sumsq) row statistics of this tile's rows (folded
  // LayerNorm of the epilogue, or the STYL pass), DMA'd like the operand tiles; then the per-variant tables
  const int st_parts = dma_stat_parts(p);
  float* sStat = reinterpret_cast<float*>(smem + NS * STAGE);
  float* sPar = sStat + BM * 2 * st_parts;                       // [nseg][4][SEG_MAX]
  float* sRow = sPar + (A_BF16 ? 0 : p.nseg) * 4 * SEG_MAX;      // [RG_MAX_SEG][64][2] = (rstd, -mean*rstd)
  SegInfo* sSeg = reinterpret_cast<SegInfo*>(sRow + RG_MAX_SEG * BM * 2);

  RG_STAMP(0);
  RG_DBG_LOAD();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int mt = (p.M + BM - 1) / BM, nt = (p.N + BN_ - 1) / BN_;
  int tile_m, tile_n;
  tile_of_block(blockIdx.x, mt, nt, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN_;
  const int nk = p.K / BK;

  auto a_row = [&](int r) {
    int gr = m0 + r;
    if (gr >= p.M) gr = p.M - 1;       // duplicate the last row; discarded by the epilogue's row guard
    return p.a_row_mod > 0 ? gr % p.a_row_mod : gr;
  };

  // ---- per-lane DMA source rows (fixed for the whole kernel)
  // W: 16 chunks of 1 KiB (8 rows x 128 B); wave w issues chunks w, w+4, w+8, w+12.
  //    lane -> row = 8c + (lane>>3), physical 16-B slot lane&7 holds logical chunk slot ^ ((row>>1)&7)
  unsigned w_off[CPW_W];
#pragma unroll
  for (int i = 0; i < CPW_W; ++i) {
    const int row = (wave + NW * i) * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    w_off[i] = (unsigned)(n0 + row) * (unsigned)p.ldw + lc * 8;   // bf16 elements; + kt*64 per tile
  }
  // A bf16: 8 chunks (same row geometry), wave w issues chunks w, w+4.
  // A fp32: 16 chunks of 1 KiB (4 rows x 256 B), wave w issues chunks w, w+4, w+8, w+12;
  //    lane -> row = 4c + (lane>>4), physical slot lane&15 holds logical chunk slot ^ (row & 15)
  int a_rowidx[ACH];
  int a_lc[ACH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    if constexpr (A_BF16) {
      const int row = (wave + NW * i) * 8 + (lane >> 3);   // 8 chunks: NW=4 -> 2 per wave, NW=8 -> 1
      a_rowidx[i] = a_row(row);
      a_lc[i] = (lane & 7) ^ ((row >> 1) & 7);
    } else {
      const int row = (wave + NW * i) * 4 + (lane >> 4);
      a_rowidx[i] = a_row(row);
      a_lc[i] = (lane & 15) ^ (row & 15);
    }
  }
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(p.W);
  const unsigned short* Wl = reinterpret_cast<const unsigned short*>(p.W_lo);

  // `first`: tiles of the prologue are issued before the LDS segment table exists; they lie in
  // segment 0 (seg_len >= (NS-1)*BK is checked on the host), read straight from the kernel arguments
  auto issue = [&](int kt, bool first) {
    unsigned char* st = smem + (kt % NS) * STAGE;
    const int k0 = kt * BK;
    if constexpr (A_BF16) {
      const unsigned short* Ab = reinterpret_cast<const unsigned short*>(p.A) +
                                 (p.gb_group > 0 ? (n0 / p.gb_group) * p.gb_stride : 0);   // grouped A (see header)
#pragma unroll
      for (int i = 0; i < ACH; ++i)
        __builtin_amdgcn_global_load_lds((const void*)(Ab + (size_t)a_rowidx[i] * p.lda + k0 + a_lc[i] * 8),
                                         (lds_void*)(st + (wave + NW * i) * 1024), 16, 0, 0);
    } else {
      const int sidx = first ? 0 : k0 / p.seg_len;
      const SegInfo sg = first ? SegInfo{p.seg[0].src, p.seg[0].ld, p.seg[0].mode} : sSeg[sidx];
      const int ks = k0 - sidx * p.seg_len;
#pragma unroll
      for (int i = 0; i < ACH; ++i)
        __builtin_amdgcn_global_load_lds((const void*)(sg.src + (size_t)a_rowidx[i] * sg.ld + ks + a_lc[i] * 4),
                                         (lds_void*)(st + (wave + NW * i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CPW_W; ++i) {
      __builtin_amdgcn_global_load_lds((const void*)(Wb + (size_t)w_off[i] + k0),
                                       (lds_void*)(st + A_STAGE + (wave + NW * i) * 1024), 16, 0, 0);
      if constexpr (SPLIT)
        __builtin_amdgcn_global_load_lds((const void*)(Wl + (size_t)w_off[i] + k0),
                                         (lds_void*)(st + A_STAGE + W_TILE_ + (wave + NW * i) * 1024), 16, 0, 0);
    }
  };

  // residual values for the epilogue and the first operand tiles are requested up front: their
  // latency overlaps the table setup and the K loop instead of adding to it
  ResidualPrefetch pre;
  if constexpr (MINW == 1)   // (the two-per-CU variant has 128 VGPRs: its residual is fetched in the epilogue)
    if (tid < NT) prefetch_residual(p, tid, m0, n0, pre, BN_);
  // Row statistics and (STYL) the folded gain / offset vectors go into LDS by DMA as well, IN FRONT of the operand
  // tiles in every wave's queue: the counted waits below then cover them, and no register-destination load (whose
  // first use the compiler would wait for with vmcnt(0), draining the ring's prefetch) is involved.
  //   sStat: 2 * parts wave-instructions of 64 floats (4 B per lane; rows clamped into the matrix)
  //   sGB  : [2][SEG_MAX] gain, offset: K / 256 wave-instructions of 16 B per lane each
  float* sGB = sPar;
  {
    const float* stat_src = STYL ? p.seg[0].stats : p.ln_stats;
    const int n_stat = 2 * st_parts;                       // instructions
    const int n_vec = STYL ? (p.K + 255) / 256 : 0;        // per vector
    for (int i = wave; i < n_stat + 2 * n_vec; i += NW) {
      if (i < n_stat) {
        const int f = i * 64 + lane, row = f / (2 * st_parts), w = f - row * 2 * st_parts;
        const int gr = min(m0 + row, p.M - 1);
        __builtin_amdgcn_global_load_lds((const void*)(stat_src + (size_t)gr * 2 * st_parts + w), (lds_void*)(sStat + i * 64), 4, 0,
                                         0);
      } else if constexpr (STYL) {
        const int v = (i - n_stat) / n_vec, piece = (i - n_stat) % n_vec;
        const float* src = (v == 0 ? p.seg[0].gamma : p.seg[0].beta) + min(piece * 256 + lane * 4, p.K - 4);
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_void*)(sGB + v * SEG_MAX + piece * 256), 16, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nk) issue(t, true);

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
        const rg_a_segment sg = p.seg[s];
        float* par = sPar + s * 4 * SEG_MAX;
        for (int i = tid; i < p.seg_len; i += NTH) {
          par[i] = sg.gamma[gboff + i];
          par[SEG_MAX + i] = sg.beta[gboff + i];
          if (sg.mode == RG_A_STYL) {
            par[2 * SEG_MAX + i] = 1.0f + sg.scale_shift[i];
            par[3 * SEG_MAX + i] = sg.scale_shift[p.seg_len + i];
          }
        }
        if (tid < BM) {
          float raw[2 * RG_MAX_LN_PARTS], mu, rs;
          load_row_stats(sg.stats + (size_t)a_row(tid) * sg.nparts * 2, sg.nparts, raw);
          reduce_row_stats(raw, sg.nparts, p.seg_len, mu, rs);
          sRow[(s * BM + tid) * 2] = rs;
          sRow[(s * BM + tid) * 2 + 1] = -mu * rs;
        }
      }
    }
    __syncthreads();
  }

  f32x4 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  int cur_seg = -1;
  float rc1[2] = {1.f, 1.f}, rc0[2] = {0.f, 0.f};
  int seg_mode = RG_A_IDENT;

  auto compute = [&](int kt) {
    const unsigned char* st = smem + (kt % NS) * STAGE;
    const unsigned char* sW = st + A_STAGE;
    const unsigned char* sWl = sW + W_TILE_;
    int sidx = 0, ks0 = 0;
    if constexpr (!A_BF16) {
      sidx = (kt * BK) / p.seg_len;
      ks0 = kt * BK - sidx * p.seg_len;
      if (sidx != cur_seg) {        // wave-uniform
        cur_seg = sidx;
        seg_mode = sSeg[sidx].mode;
        if (seg_mode != RG_A_IDENT) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int r = wr * 32 + i * 16 + frow;
            rc1[i] = sRow[(sidx * BM + r) * 2];
            rc0[i] = sRow[(sidx * BM + r) * 2 + 1];
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2], al[2], bfr[TN];
      if constexpr (A_BF16) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(st + lds_off(wr * 32 + i * 16 + frow, 4 * s + fq));
      } else {
        float g[8], be[8], sc[8], sh[8];
        if (seg_mode != RG_A_IDENT) {
          const float* par = sPar + sidx * 4 * SEG_MAX + ks0 + 32 * s + 8 * fq;
          *reinterpret_cast<f32x4*>(g) = *reinterpret_cast<const f32x4*>(par);
          *reinterpret_cast<f32x4*>(g + 4) = *reinterpret_cast<const f32x4*>(par + 4);
          *reinterpret_cast<f32x4*>(be) = *reinterpret_cast<const f32x4*>(par + SEG_MAX);
          *reinterpret_cast<f32x4*>(be + 4) = *reinterpret_cast<const f32x4*>(par + SEG_MAX + 4);
          if (seg_mode == RG_A_STYL) {
            *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(par + 2 * SEG_MAX);
            *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(par + 2 * SEG_MAX + 4);
            *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(par + 3 * SEG_MAX);
            *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(par + 3 * SEG_MAX + 4);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wr * 32 + i * 16 + frow;
          const int lc = 8 * s + 2 * fq;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + r * 256 + (((lc) ^ (r & 15)) << 4));
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(st + r * 256 + (((lc + 1) ^ (r & 15)) << 4));
          float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          if (seg_mode != RG_A_IDENT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(fmaf(v[e], rc1[i], rc0[i]), g[e], be[e]);
            if (seg_mode == RG_A_STYL) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = silu_f(fmaf(v[e], sc[e], sh[e]));
            }
          }
          const u32x4 hi = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
          af[i] = __builtin_bit_cast(bf16x8, hi);
          if constexpr (SPLIT) {
            float rr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) rr[e] = v[e] - bf2f(f2bf(v[e]));
            const u32x4 lo = {pack2(rr[0], rr[1]), pack2(rr[2], rr[3]), pack2(rr[4], rr[5]), pack2(rr[6], rr[7])};
            al[i] = __builtin_bit_cast(bf16x8, lo);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sW + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s + fq));
      if constexpr (SPLIT) {
        bf16x8 bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bl[j] = *reinterpret_cast<const bf16x8*>(sWl + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s + fq));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bfr[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  // STYL: the A K-tile `kt` (8 KiB: 512 chunks of 16 B; chunk c = row c >> 3, physical slot c & 7, which holds the
  // logical 8-column group slot ^ ((row >> 1) & 7)) rewritten in place.  A thread owns the same chunk(s) of every
  // K-tile, i.e. the same row(s): their (mean, rstd) live in its registers.
  constexpr int CH = (BM * 8) / NTH;     // chunks per thread and K-tile: 1 (8 waves) or 2 (4 waves)
  float row_mu[CH], row_rs[CH];
  auto transform = [&](int kt) {
    unsigned char* st = smem + (kt % NS) * STAGE;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = tid + j * NTH;
      const int row = c >> 3;
      const int k = kt * BK + (((c & 7) ^ ((row >> 1) & 7)) << 3);
      const u32x4 raw = *reinterpret_cast<const u32x4*>(st + c * 16);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(sGB + k), g1 = *reinterpret_cast<const f32x4*>(sGB + k + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sGB + SEG_MAX + k);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(sGB + SEG_MAX + k + 4);
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const float b[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(raw[e] << 16);
        v[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = silu_f(fmaf((v[e] - row_mu[j]) * row_rs[j], g[e], b[e]));
      *reinterpret_cast<u32x4*>(st + c * 16) = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    }
  };

  if constexpr (STYL) {
    // Iteration t: [this wave's pieces of tile t+1 landed, its rewrite of tile t written] -> barrier -> issue tile
    // t+NS-1 into stage (t-1) % NS -> rewrite tile t+1, MFMAs of tile t (different stages).
    RG_STAMP(1);
    wait_tiles<PER_TILE, NS - 2>(min(NS - 2, nk - 1));
    __builtin_amdgcn_s_barrier();          // statistics, gain / offset and everyone's pieces of tile 0 have landed
#pragma unroll
    for (int j = 0; j < CH; ++j) lds_row_stats(sStat + ((tid + j * NTH) >> 3) * 2 * st_parts, st_parts, p.K, row_mu[j], row_rs[j]);
    transform(0);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_tiles<PER_TILE, NS - 3>(min(NS - 3, nk - 2 - kt));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + NS - 1 < nk) issue(kt + NS - 1, false);
      if (kt + 1 < nk) transform(kt + 1);
      compute(kt);
    }
  }
  // ---- ring pipeline.  Iteration t: [wait until this wave's pieces of tile t have landed]
  //      -> barrier (everyone's pieces landed; everyone is done reading stage (t-1) % NS)
  //      -> issue tile t+NS-1 into stage (t-1) % NS -> MFMAs of tile t.
  RG_STAMP(1);
  for (int kt = 0; !STYL && kt < nk; ++kt) {
    const int younger = min(NS - 2, nk - 1 - kt);   // tiles issued after tile kt and still in flight
    wait_tiles<PER_TILE, NS - 2>(younger);
    __builtin_amdgcn_s_barrier();
#ifdef RG_STAMPS_LOOP
    if (kt < 40) RG_STAMP(2 + kt);
#endif
    if (kt + NS - 1 < nk && !RG_DBG(2)) issue(kt + NS - 1, false);
    if (!RG_DBG(1)) compute(kt);
  }
  RG_STAMP(60);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if (RG_DBG(4)) return;
  // ---- epilogue through LDS: sC[64][SC_LD] fp32 (33.8 KiB, fits in the ring)
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        sC[(wr * 32 + i * 16 + fq * 4 + e) * SC_LD + wc * (TN * 16) + j * 16 + frow] = acc[i][j][e];
  __syncthreads();
  RG_STAMP(61);
  if (tid < NT) epilogue(p, sC, tid, m0, n0, tile_n, nt, MINW == 1 ? &pre : nullptr, BN_, (!STYL && st_parts > 0) ? sStat : nullptr);
  RG_STAMP(62);
}

constexpr size_t LDS_MAX = 160 * 1024;
// bytes of the row-statistics block behind the ring (bf16-A kernels: as many partials as the descriptor uses)
size_t stat_bytes(const rg_gemm_desc* d) { return (size_t)BM * 2 * sizeof(float) * dma_stat_parts(*d); }

template <bool A_BF16, bool SPLIT>
size_t dma_lds_bytes(int nseg, int ns, size_t LN_ROWS) {
  const size_t a_stage = A_BF16 ? A_TILE : 2 * A_TILE;
  const size_t stage = a_stage + (SPLIT ? 2 : 1) * W_TILE;
  const size_t tables = A_BF16 ? 0 : ((size_t)nseg * 4 * SEG_MAX * 4 + RG_MAX_SEG * BM * 2 * 4 + RG_MAX_SEG * sizeof(SegInfo));
  return ns * stage + LN_ROWS + tables;
}

template <bool A_BF16, bool SPLIT, int NS, int NW>
void dma_launch(const rg_gemm_desc* d, dim3 grid, hipStream_t s) {
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (gemm_dma_kernel<A_BF16, SPLIT, NS, NW>), (int)LDS_MAX);
  const size_t lds = dma_lds_bytes<A_BF16, SPLIT>(d->nseg, NS, stat_bytes(d));
  hipLaunchKernelGGL((gemm_dma_kernel<A_BF16, SPLIT, NS, NW>), rg_group_grid(grid), dim3(NW * 64), lds, s, rg_group_of(d));
}

// 64x64 tiles, bf16 A, 4 waves (2x2, 32x32 each), 4-stage ring of 16 KiB; the epilogue tile keeps its 128-column stride
template <int NS>
void dma_launch_narrow_ns(const rg_gemm_desc* d, hipStream_t s) {
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (gemm_dma_kernel<true, false, NS, 4, 64>), (int)LDS_MAX);
  size_t lds = (size_t)NS * (A_TILE + 64 * ROW_BYTES) + stat_bytes(d);
  const size_t epi = (size_t)BM * SC_LD * sizeof(float);
  if (epi > lds) lds = epi;
  const int mt = (d->M + BM - 1) / BM, nt = (d->N + 63) / 64;
  hipLaunchKernelGGL((gemm_dma_kernel<true, false, NS, 4, 64>), rg_group_grid(dim3(mt * nt)), dim3(256), lds, s, rg_group_of(d));
}

// Ring depth: 4.  Measured (M = 688..2064, K = 512..2048, graph-replayed chains on rotating operands): a 6-stage
// 64x128 ring / 8-stage 64x64 ring (a K = 512 panel requested whole before the first MFMA) is 7-10 % SLOWER --
// the K loop runs at the per-CU LDS-DMA intake (~72 GB/s, 0.34 us per 24.5 KB K-tile), not at the prefetch depth.
// Same for two K-tiles per barrier (ring of 2 / 3 stage pairs: +4..12 % / +-0 %) and for L2-hot operands (-0.5 us per
// launch at most): neither the barrier count nor the cache level of the source sets the per-tile time.
void dma_launch_narrow(const rg_gemm_desc* d, hipStream_t s) { dma_launch_narrow_ns<4>(d, s); }

// bf16 A with the in-LDS stylization pass (p.seg[0]): a ring of 5 where one workgroup owns the CU (the rewrite runs one
// tile ahead, so one stage less is in flight than the ring is deep), 3 stages + tables = 76.5 KiB where two share it
template <int NS, int NW>
void dma_launch_styl(const rg_gemm_desc* d, dim3 grid, hipStream_t s) {
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (gemm_dma_kernel<true, false, NS, NW, BN, true>), (int)LDS_MAX);
  size_t lds = (size_t)NS * (A_TILE + W_TILE) + stat_bytes(d) + 2 * SEG_MAX * sizeof(float);
  const size_t epi = (size_t)BM * SC_LD * sizeof(float);
  if (epi > lds) lds = epi;
  hipLaunchKernelGGL((gemm_dma_kernel<true, false, NS, NW, BN, true>), rg_group_grid(grid), dim3(NW * 64), lds, s, rg_group_of(d));
}

// 8 waves, ring of 3 (76.5 KiB), at most 128 VGPRs (no residual prefetch): two workgroups per CU
void dma_launch_pair(const rg_gemm_desc* d, dim3 grid, hipStream_t s) {
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (gemm_dma_kernel<true, false, 3, 8, BN, false, 4>), (int)LDS_MAX);
  const size_t lds = dma_lds_bytes<true, false>(0, 3, stat_bytes(d));
  hipLaunchKernelGGL((gemm_dma_kernel<true, false, 3, 8, BN, false, 4>), rg_group_grid(grid), dim3(512), lds, s, rg_group_of(d));
}

// ring depth for this descriptor (0: does not fit the 160 KiB LDS at all).  Grids with more
// workgroups than CUs use a 2-stage ring (<= 80 KiB) so two workgroups share a CU and hide each
// other's DMA latency; smaller grids are a pure latency chain and take the deepest ring that fits.
int dma_depth(const rg_gemm_desc* d, int num_cus) {
  const int wgs = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  if (!d->W_lo && wgs > num_cus) {
    const size_t need2 = d->a_is_bf16 ? dma_lds_bytes<true, false>(0, 2, stat_bytes(d)) : dma_lds_bytes<false, false>(d->nseg, 2, stat_bytes(d));
    if (need2 <= LDS_MAX / 2) return 2;
  }
  for (int ns = (d->W_lo ? 3 : 4); ns >= 3; --ns) {
    const size_t need = d->W_lo ? dma_lds_bytes<false, true>(d->nseg, ns, stat_bytes(d))
                                : (d->a_is_bf16 ? dma_lds_bytes<true, false>(0, ns, stat_bytes(d))
                                                : dma_lds_bytes<false, false>(d->nseg, ns, stat_bytes(d)));
    if (need <= LDS_MAX) return ns;
  }
  return 0;
}

}  // namespace

// true if the descriptor satisfies the DMA kernel's alignment contract and its LDS ring fits
bool rg_gemm_dma_eligible(const rg_gemm_desc* d) {
  if (d->K % 64 != 0) return false;
  if (dma_depth(d, 256) == 0) return false;
  if (d->a_is_bf16) return (d->lda % 8) == 0 && ((uintptr_t)d->A % 16) == 0;
  if (d->seg_len % 64 != 0) return false;
  if (d->nseg > 1 && d->seg_len < 3 * 64) return false;   // the prologue tiles must lie in segment 0
  for (int s = 0; s < d->nseg; ++s)
    if ((d->seg[s].ld % 4) != 0 || ((uintptr_t)d->seg[s].src % 16) != 0) return false;
  return true;
}

void rg_gemm_dma_launch(const rg_gemm_desc* d, int num_cus, int waves, void* stream) {
  if (d->tile_n == 64) {
    dma_launch_narrow(d, rg_stream(stream));
    return;
  }
  const int mt = (d->M + BM - 1) / BM, nt = (d->N + BN - 1) / BN;
  dim3 grid(mt * nt);
  hipStream_t s = rg_stream(stream);
  if (rg_gemm_a_styl(d)) {
    if ((int)grid.x <= num_cus) dma_launch_styl<5, 8>(d, grid, s);
    else dma_launch_styl<3, 4>(d, grid, s);
    return;
  }
  const int ns = dma_depth(d, num_cus);
  // bf16 A, more workgroups than CUs: two 8-wave workgroups per CU (ring of 3, <= 128 VGPRs) instead of two 4-wave ones
  // (ring of 2): 16 waves per CU hide each other's round trips (M = 4128 forward 1456 -> 1385 us); waves = 16 forces it
  if (d->a_is_bf16 && !d->W_lo && (waves == 16 || (waves == 0 && ns == 2))) {
    dma_launch_pair(d, grid, s);
    return;
  }
  // measured (MI355X, graph-replayed): 8 waves win ~7% on single-round grids with plain or bf16 A
  // (7.8 vs 8.4 us at 2752x512x512); with the LN/stylization prologue or multi-round grids the
  // extra per-wave prologue math / lower workgroup residency loses 15-50%
  bool plain = true;
  if (!d->a_is_bf16)
    for (int i = 0; i < d->nseg; ++i) plain = plain && d->seg[i].mode == RG_A_IDENT;
  const bool use8 = waves == 8 || (waves == 0 && plain && (int)grid.x <= num_cus);
  if (d->W_lo) {
    dma_launch<false, true, 3, 4>(d, grid, s);
  } else if (use8) {
    if (d->a_is_bf16) {
      if (ns == 2) dma_launch<true, false, 2, 8>(d, grid, s);
      else dma_launch<true, false, 4, 8>(d, grid, s);
    } else if (ns == 2) {
      dma_launch<false, false, 2, 8>(d, grid, s);
    } else if (ns == 4) {
      dma_launch<false, false, 4, 8>(d, grid, s);
    } else {
      dma_launch<false, false, 3, 8>(d, grid, s);
    }
  } else if (d->a_is_bf16) {
    if (ns == 2) dma_launch<true, false, 2, 4>(d, grid, s);
    else dma_launch<true, false, 4, 4>(d, grid, s);
  } else if (ns == 2) {
    dma_launch<false, false, 2, 4>(d, grid, s);
  } else if (ns == 4) {
    dma_launch<false, false, 4, 4>(d, grid, s);
  } else {
    dma_launch<false, false, 3, 4>(d, grid, s);
  }
}

extern "C" int rg_set_gemm_waves(rg_handle* h, int waves) {
  RG_REQUIRE(h, h != nullptr, "null handle");
  RG_REQUIRE(h, waves == 0 || waves == 4 || waves == 8 || waves == 16, "waves must be 0, 4, 8 or 16 (8 waves, two workgroups per CU)");
  h->gemm_waves = waves;
  return RG_OK;
}

#ifdef RG_STAMPS
extern "C" int rg_debug_set_mode(int mode) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_mode), &mode, sizeof(int)) == hipSuccess ? 0 : -2;
}
extern "C" int rg_debug_set_stamp_buffer(void* dev_ptr) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -2;
}
#endif
