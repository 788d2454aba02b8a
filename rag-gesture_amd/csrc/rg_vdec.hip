// Block-fused body-part VAE DECODER of the "all_encoder" architecture (round 4): the skip-transformer stack over the
// 10 latent tokens + 150 frame queries of a clip (gesture_vae.py:195-239 `decode`, detr_utils.py:101-152, :335-393 with
// pos = query_pos: q = k = x + pos, v = x; num_heads * 8 = 32 heads of 16) as ONE launch per block instead of nine.
//
// Why: as per-op launch chains (72 grouped launches) the decode of a batch costs 4.9 ms alone and 8-15 ms beside the other
// lane's denoiser chain, and it sits on the lane's stream between two chains (profiles/r04i_lane_timeline.txt).  A 160-token
// sequence does not fit one workgroup (320 KB of fp32), so it is cut into four tiles of 40 rows; a tile's workgroup keeps its
// rows on chip for a whole block exactly as rg_venc.hip / rg_seq.hip do (fp32 residual in registers, bf16 operand panels in
// LDS, weights streamed by LDS-DMA), and what the tiles of a sequence exchange -- every tile's keys and values -- goes through
// L2 at the one point per block where it is needed.  A launch therefore runs from the attention of block k - 1 to the
// Q / K / V projections of block k:
//     [attention(k-1) -> out_proj + residual -> norm1 -> FFN -> norm2] -> [skip push | skip linear(k)] -> Q, K, V (k)
// with the launch boundary as the sequence-wide synchronisation.  Launch 0 only projects; the last launch ends with the
// stack's final LayerNorm.  Between launches a tile hands itself x (fp32 rows) and its Q panel (bf16, the LDS image as is).
//
// Attention: wave w owns features [64 w, 64 w + 64) = heads 4 w .. 4 w + 3.  Per head and 16-query block: the 160 scores of a
// query live in 40 registers of its lane (S^T = K Q^T, one MFMA per 16 keys: the head's 16 dims fill half of the k = 32
// contraction, the other half is zero on both operands), softmax over them, then O^T = V^T P with V^T fragments loaded
// feature-major (the producer stores V transposed) in the order the score registers enumerate the keys.
#define RG_PACK2_ONE      // (the kernel owns its SIMDs, RG_OWN_THE_SIMD: rg_common.h rg_pack2_bf16)
#include "rg_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int DM = 512;
constexpr int TP = 48;         // panel rows
constexpr int TR = 40;         // rows of a tile (a 160-token sequence = 4 tiles); panel rows [TR, TP) repeat row TR - 1
constexpr int NW = 8;
constexpr int NTH = NW * 64;
constexpr int RD = 7;          // ring slots (1 KiB) per wave
constexpr int OFF_P0 = 0;
constexpr int OFF_P1 = TP * 1024;
constexpr int OFF_RING = 2 * TP * 1024;
constexpr int MAX_UNITS = 8 * 17 + 2 * 8;                // up to 17 blocks (num_layers <= 16)
constexpr int MAX_SEG = 2 * MAX_UNITS + 2;               // (P, W) per unit + final norm P + sentinel
constexpr int OFF_DESC = OFF_RING + NW * RD * 1024;
constexpr int OFF_STAT = OFF_DESC + MAX_SEG * 16;
constexpr int LDS_BYTES = OFF_STAT + NW * TP * 2 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return rg_pack2_bf16(lo, hi); }
__device__ __forceinline__ float gelu_fast(float v) { return rg_gelu_erf(v); }
// 8 fp32 values -> bf16 hi fragment and the bf16 residual fragment
__device__ __forceinline__ void split_hl(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned short a = f2bf(v[2 * q]), b = f2bf(v[2 * q + 1]);
    h[q] = (unsigned)a | ((unsigned)b << 16);
    l[q] = pack2(v[2 * q] - bf2f(a), v[2 * q + 1] - bf2f(b));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  return __builtin_bit_cast(bf16x8, u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])});
}
__device__ __forceinline__ float xsum4(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float xmax4(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

typedef f32x4 Acc[4][3];   // [16-feature block of the wave's 64][16-token block]

__device__ __forceinline__ void zero(Acc& a) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) a[j][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace

struct rg_vdec_group { rg_vdec_args a[4]; };   // up to four stacks (the four body parts) in one launch: blockIdx.y picks

// Diagnostic build only (build.py RG_DIAG=1): wall-clock (100 MHz) stamps at the phase boundaries of a launch, written per wave
// to a.dump[(tile * 8 + wave) * 16 + i] (ticks since stamp 0) when pad_ == 99 (profiles/dbg/vdec_stamps.py).  Stamps: 0 start,
// 1 descriptors + Q image in LDS, 2 attention, 3 x rows loaded, 4 out_proj + norm1, 5 FFN + norm2 (+ skip push), 6 skip linear +
// x rows stored, 7 Q / K / V projections, 8 K / V / Q image stored (launch 0: 1 -> 3 -> 6 ...; last launch: ... 5 -> 8).
#ifdef RG_STAMPS
#define VSTAMP(i) ts_[i] = __builtin_amdgcn_s_memrealtime()
#else
#define VSTAMP(i)
#endif

__global__ void __launch_bounds__(NTH) rg_vdec_kernel(const rg_vdec_group grp) {
  RG_OWN_THE_SIMD();
  const rg_vdec_args& a = grp.a[blockIdx.y];
  if ((int)blockIdx.x >= 4 * a.nseq) return;
#ifdef RG_STAMPS
  unsigned long long ts_[9];
  const bool stamps = a.pad_ == 99 && a.dump;
  VSTAMP(0);
#pragma unroll
  for (int i = 1; i < 9; ++i) ts_[i] = 0;
#else
  constexpr bool stamps = false;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const P0 = smem + OFF_P0;
  unsigned char* const P1 = smem + OFF_P1;
  float* const sStat = reinterpret_cast<float*>(smem + OFF_STAT);
  const int tid = threadIdx.x, lane0 = tid & 63;
#define LANE_LOCAL()                      \
  int ln_ = lane0;                        \
  asm volatile("" : "+v"(ln_));         \
  const int lane = ln_, l15 = ln_ & 15, g4 = ln_ >> 4; \
  (void)lane; (void)l15; (void)g4
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const ring = smem + OFF_RING + wave * (RD * 1024);
  const int nb = a.nb, n_blocks = 2 * nb + 1;
  const int NU = 8 * n_blocks + 2 * nb;
  const int step = a.step;                            // 0 .. n_blocks
  const bool first = step == 0, last = step == n_blocks;
  const int tile = blockIdx.x, seq = tile >> 2, row0 = TR * (tile & 3);
  // units of this launch: [out_proj, FF1_0, FF2_0, FF1_1, FF2_1 of block step - 1] [skip linear (2), Q, K, V of block step]
  auto base = [&](int b) { return 8 * b + 2 * max(0, b - nb - 1); };
  const int skip_now = (!last && step > nb) ? 2 : 0;
  const int u0 = first ? 0 : base(step - 1) + (step - 1 > nb ? 2 : 0) + 3;
  const int n_units = (first ? 0 : 5) + (last ? 0 : skip_now + 3);
  const int n_seg = 2 * n_units + (last ? 1 : 0);

  if (tid <= n_seg) {
    const unsigned char* adr = reinterpret_cast<const unsigned char*>(a.wstream);
    unsigned cnt = 1u << 30, stride = 0;
    if (tid < n_seg) {
      const int u = tid < 2 * n_units ? u0 + (tid >> 1) : NU;      // (the last launch ends with the final norm's parameters)
      if ((tid & 1) == 0) {
        adr = reinterpret_cast<const unsigned char*>(a.pstream) + ((size_t)u * 8 << 10);
        cnt = 1; stride = 1;
      } else {
        adr = reinterpret_cast<const unsigned char*>(a.wstream) + ((size_t)u * 512 << 10);
        cnt = 64; stride = 64;
      }
    }
    const unsigned long long av = reinterpret_cast<unsigned long long>(adr);
    *reinterpret_cast<u32x4*>(smem + OFF_DESC + tid * 16) = u32x4{(unsigned)av, (unsigned)(av >> 32), cnt, stride};
  }
  // the tile's Q panel image of the previous launch: 48 KiB, as it lay in LDS
  if (!first) {
    const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.qimg) + (size_t)tile * (TP * 1024));
#pragma unroll
    for (int i = 0; i < (TP * 64) / NTH; ++i) reinterpret_cast<u32x4*>(P1)[tid + NTH * i] = src[tid + NTH * i];
  }
  __syncthreads();
  VSTAMP(1);

  auto panel_store = [&](unsigned char* panel, int l15, int g4, int j, int tb, float v0, float v1, float v2, float v3) {
    const int s = 2 * wave + (j >> 1), gq = 2 * (j & 1) + (g4 >> 1);
    *reinterpret_cast<u32x2*>(panel + ((tb * 16 + s) << 10) + ((l15 + 16 * gq) << 4) + 8 * (g4 & 1)) = u32x2{pack2(v0, v1), pack2(v2, v3)};
  };
  auto write_raw = [&](unsigned char* panel, const Acc& v) {
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) panel_store(panel, l15, g4, j, tb, v[j][tb][0], v[j][tb][1], v[j][tb][2], v[j][tb][3]);
  };

  // ---- the wave's fetch cursor (all wave-uniform), as rg_venc.hip
  int ie = 0, ir = 0;
  int cur_cnt = 0;
  __amdgpu_buffer_rsrc_t cur_rsrc;
  const int lane16 = lane0 * 16;
  auto load_seg = [&]() {
    const u32x4 d = *reinterpret_cast<const u32x4*>(smem + OFF_DESC + ie * 16);
    const unsigned lo = __builtin_amdgcn_readfirstlane(d[0]), hi = __builtin_amdgcn_readfirstlane(d[1]);
    cur_cnt = __builtin_amdgcn_readfirstlane(d[2]);
    const unsigned stride = __builtin_amdgcn_readfirstlane(d[3]);
    unsigned char* bs = reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo) + ((size_t)(wave * stride) << 10);
    cur_rsrc = __builtin_amdgcn_make_buffer_rsrc(bs, 0, 0x7fffffff, 0x00020000);
  };
  auto issue = [&](int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_rsrc, (lds_void*)(ring + slot * 1024), 16, lane16, ir << 10, 0, 0);
    if (++ir == cur_cnt) {
      ir = 0;
      ++ie;
      load_seg();
    }
  };
  int head = 0;
  auto consume = [&]() -> const unsigned char* {
    wait_vmcnt<RD - 1>();
    return ring + head * 1024;
  };
  auto release = [&]() {
    wait_lds();
    issue(head);
    head = head + 1 == RD ? 0 : head + 1;
  };
  load_seg();
#pragma unroll
  for (int s = 0; s < RD; ++s) issue(s);

  // unit GEMM (see rg_seq.hip: gemm_frags_reg): the unit's first RD fragments -- issued before the unit starts, across its
  // epilogue -- come through the LDS ring, the other 64 - RD straight into registers (eight in rotation, RD in flight as before);
  // the last RD iterations refill the ring's slots for whatever the stream holds next.  `head` leaves as it came.
  auto issue_reg = [&](u32x4& dst) {
    dst = __builtin_amdgcn_raw_buffer_load_b128(cur_rsrc, lane16, ir << 10, 0);
    if (++ir == cur_cnt) {
      ir = 0;
      ++ie;
      load_seg();
    }
  };
#ifndef RGD_REG
#define RGD_REG 0x7
#endif
#define SITE(n) std::integral_constant<int, n>()
  auto gemm_unit_reg = [&](Acc& acc, const unsigned char* panel, auto std_tag) {
    constexpr bool STD = decltype(std_tag)::value;
    constexpr int NJ = 4;
    static_assert(RD <= 8 && RD >= 2, "RD fragments in flight, eight registers in rotation");
    LANE_LOCAL();
    const unsigned char* pl = panel + lane * 16;
    const unsigned char* rl = ring + lane * 16;
    bf16x8 pf[3];
    u32x4 wr[8];
    int hs = head;
    wait_vmcnt<RD - 1>();
    wr[0] = *reinterpret_cast<const u32x4*>(rl + hs * 1024);
    hs = hs + 1 == RD ? 0 : hs + 1;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) pf[tb] = *reinterpret_cast<const bf16x8*>(pl + ((tb * 16) << 10));
    auto group = [&](const int s0, auto first_tag, auto last_tag) {      // fragments [4 s0, 4 s0 + 8)
      constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int j = f % NJ, s = s0 + f / NJ;
        if (FIRST && f + 1 < RD) {      // the next fragment sits in the ring: landed when at most RD - 2 younger loads are outstanding
          wait_vmcnt<RD - 2>();
          wr[f + 1] = *reinterpret_cast<const u32x4*>(rl + hs * 1024);
          hs = hs + 1 == RD ? 0 : hs + 1;
        }
        const bf16x8 wv = __builtin_bit_cast(bf16x8, wr[f]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          acc[j][tb] = STD ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[tb], wv, acc[j][tb], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, pf[tb], acc[j][tb], 0, 0, 0);
          if (j == NJ - 1) {           // re-read for the next k-step right behind its last use (behind the panel's end: valid LDS, unused)
            pf[tb] = *reinterpret_cast<const bf16x8*>(pl + ((tb * 16 + s + 1) << 10));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (LAST && f >= 8 - RD) {
          issue(hs);
          hs = hs + 1 == RD ? 0 : hs + 1;
        } else {
          issue_reg(wr[(f + RD) & 7]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    group(0, std::true_type(), std::false_type());
#pragma unroll 1
    for (int s0 = 2; s0 < 14; s0 += 2) group(s0, std::false_type(), std::false_type());
    group(14, std::false_type(), std::true_type());
  };
  // (site: bit n of RGD_REG = this call site has the 32 VGPRs for the register form)
  auto gemm_unit = [&](Acc& acc, const unsigned char* panel, auto std_tag, auto site) {
    if constexpr (((RGD_REG) >> decltype(site)::value) & 1) { gemm_unit_reg(acc, panel, std_tag); return; }
    constexpr bool STD = decltype(std_tag)::value;
    constexpr int NJ = 4;
    LANE_LOCAL();
    const unsigned char* pl = panel + lane * 16;
    const unsigned char* rl = ring + lane * 16;
    bf16x8 w[2], pf[2][3];
    wait_vmcnt<RD - 1>();
    w[0] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) pf[0][tb] = *reinterpret_cast<const bf16x8*>(pl + ((tb * 16) << 10));
#pragma unroll 1
    for (int s2 = 0; s2 < 16; s2 += 2) {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bool lastf = ss == 1 && j == NJ - 1 && s2 == 14;
          wait_lds();
          issue(head);
          head = head + 1 == RD ? 0 : head + 1;
          if (!lastf) {
            wait_vmcnt<RD - 1>();
            w[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
          }
          if (j == NJ - 1 && !lastf) {
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) pf[ss ^ 1][tb] = *reinterpret_cast<const bf16x8*>(pl + ((tb * 16 + s2 + ss + 1) << 10));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tb = 0; tb < 3; ++tb)
            acc[j][tb] = STD ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[ss][tb], w[j & 1], acc[j][tb], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j & 1], pf[ss][tb], acc[j][tb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  std::false_type TL;
  std::true_type STDL;
  auto par_t = [&](const unsigned char* slot, int p, int j, int g4) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(slot + (p * 64 + 16 * j + 4 * g4) * 4);
  };
  auto add_bias_t = [&](Acc& acc, const unsigned char* slot) {
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b = par_t(slot, 0, j, g4);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) acc[j][tb] += b;
    }
  };
  auto unit = [&](Acc& acc, const unsigned char* panel) {
    const unsigned char* ps = consume();
    add_bias_t(acc, ps);
    release();
    gemm_unit(acc, panel, TL, SITE(0));
  };
  auto row_stats = [&](const Acc& v, float (&mean)[3], float (&rstd)[3]) {
    LANE_LOCAL();
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      // (one pass: per-wave sum and sum of squares, variance = E[x^2] - mean^2 in fp32; rg_seq.hip row_stats)
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += (v[j][tb][0] + v[j][tb][1]) + (v[j][tb][2] + v[j][tb][3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) ss = fmaf(v[j][tb][r], v[j][tb][r], ss);
      }
      s = xsum4(s);
      ss = xsum4(ss);
      if (g4 == 0) *reinterpret_cast<float2*>(sStat + (wave * TP + 16 * tb + l15) * 2) = make_float2(s, ss);
    }
    bar();
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      float tot = 0.f, tot2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float2 p = *reinterpret_cast<const float2*>(sStat + (w * TP + 16 * tb + l15) * 2);
        tot += p.x;
        tot2 += p.y;
      }
      const float mu = tot * (1.0f / DM);
      mean[tb] = mu;
      rstd[tb] = rsqrtf(fmaxf(fmaf(-mu, mu, tot2 * (1.0f / DM)), 0.f) + 1e-5f);
    }
  };
  // rows of the tile, T layout: row 16 tb + l15 (clamped into the tile for loads, skipped for stores)
  auto rows_io = [&](Acc& v, float* basep, bool store) {
    LANE_LOCAL();
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const int r = 16 * tb + l15;
      float* p = basep + ((size_t)seq * 160 + row0 + min(r, TR - 1)) * DM + 64 * wave + 4 * g4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!store) v[j][tb] = *reinterpret_cast<const f32x4*>(p + 16 * j);
        else if (r < TR) *reinterpret_cast<f32x4*>(p + 16 * j) = v[j][tb];
      }
    }
    wait_vmcnt<0>();
  };
  float* const xbuf = a.xbuf + (size_t)tile * nb * (NW * 12 * 64 * 4);
  auto skip_io = [&](Acc& v, int slot, bool store) {
    LANE_LOCAL();
    float* bs = xbuf + ((size_t)slot * NW + wave) * (12 * 64 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        f32x4* p = reinterpret_cast<f32x4*>(bs + ((j * 3 + tb) * 64 + lane) * 4);
        if (store) *p = v[j][tb]; else v[j][tb] = *p;
      }
    wait_vmcnt<0>();
  };

  Acc xr;
  if (!first) {
    // ======================================================= attention of block step - 1: heads 4 wave .. 4 wave + 3
    Acc oo;
    {
      LANE_LOCAL();
      // (keys / values are double-buffered by launch parity: this launch's faster tiles write block `step`'s K / V while the
      //  slower tiles of the same sequence still read block step - 1's)
      const size_t par_r = (size_t)((step - 1) & 1) * a.nseq * 160 * DM;
      const unsigned short* Kg = reinterpret_cast<const unsigned short*>(a.kbuf) + par_r + (size_t)seq * 160 * DM + 64 * wave + 8 * (g4 & 1);
      const unsigned short* Vg = reinterpret_cast<const unsigned short*>(a.vt) + par_r + ((size_t)seq * DM + 64 * wave + l15) * 160 + 4 * g4;
      const bf16x8 zero8 = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const bool sel = (g4 >> 1) == (h & 1);       // the lane groups whose 8 dims belong to this head in the 32-wide k-step
        bf16x8 kf[10], vf[5];
#pragma unroll
        for (int kb = 0; kb < 10; ++kb)
          kf[kb] = sel ? *reinterpret_cast<const bf16x8*>(Kg + (size_t)(16 * kb + l15) * DM + 16 * h) : zero8;
#pragma unroll
        for (int kp = 0; kp < 5; ++kp) {
          const u32x2 v0 = *reinterpret_cast<const u32x2*>(Vg + (size_t)(16 * h) * 160 + 32 * kp);
          const u32x2 v1 = *reinterpret_cast<const u32x2*>(Vg + (size_t)(16 * h) * 160 + 32 * kp + 16);
          vf[kp] = __builtin_bit_cast(bf16x8, u32x4{v0[0], v0[1], v1[0], v1[1]});
        }
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) {
          const bf16x8 qraw = *reinterpret_cast<const bf16x8*>(P1 + ((qb * 16 + 2 * wave + (h >> 1)) << 10) + lane * 16);
          const bf16x8 qf = sel ? qraw : zero8;
          f32x4 sc[10];
#pragma unroll
          for (int kb = 0; kb < 10; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb], qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          float mx = -INFINITY;
#pragma unroll
          for (int kb = 0; kb < 10; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kb][r]);
          mx = xmax4(mx);
          const float nm2 = mx * -1.44269504088896340736f;
          float sum = 0.f;
#pragma unroll
          for (int kb = 0; kb < 10; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sc[kb][r] = rg_exp_sub(sc[kb][r], nm2);
              sum += sc[kb][r];
            }
          sum = xsum4(sum);
          const float inv = __builtin_amdgcn_rcpf(sum);
          f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kp = 0; kp < 5; ++kp) {
            const float p8[8] = {sc[2 * kp][0] * inv, sc[2 * kp][1] * inv, sc[2 * kp][2] * inv, sc[2 * kp][3] * inv,
                                 sc[2 * kp + 1][0] * inv, sc[2 * kp + 1][1] * inv, sc[2 * kp + 1][2] * inv, sc[2 * kp + 1][3] * inv};
            bf16x8 ph, pl;
            split_hl(p8, ph, pl);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kp], pl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kp], ph, d, 0, 0, 0);
          }
          oo[h][qb] = d;
        }
      }
    }
    VSTAMP(2);
    bar();                                        // everyone is done with the Q panel
    write_raw(P1, oo);
    bar();
    rows_io(xr, a.x, false);
    VSTAMP(3);
    // ======================================================= x = LayerNorm1(x + out_proj(attention))
    {
      const unsigned char* ps = consume();
      f32x4 ga[4], be[4];
      {
        LANE_LOCAL();
#pragma unroll
        for (int j = 0; j < 4; ++j) { ga[j] = par_t(ps, 1, j, g4); be[j] = par_t(ps, 2, j, g4); }
      }
      add_bias_t(xr, ps);
      release();
      gemm_unit(xr, P1, TL, SITE(1));
      float mean[3], rstd[3];
      row_stats(xr, mean, rstd);
      LANE_LOCAL();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xr[j][tb][r] = fmaf((xr[j][tb][r] - mean[tb]) * rstd[tb], ga[j][r], be[j][r]);
          panel_store(P0, l15, g4, j, tb, xr[j][tb][0], xr[j][tb][1], xr[j][tb][2], xr[j][tb][3]);
        }
      bar();
    }
    VSTAMP(4);
    // ======================================================= x = LayerNorm2(x + linear2(gelu(linear1(x))))
    {
      f32x4 ga[4], be[4];
#pragma unroll 1
      for (int jh = 0; jh < 2; ++jh) {
        Acc gg;
        zero(gg);
        unit(gg, P0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tb = 0; tb < 3; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) gg[j][tb][r] = gelu_fast(gg[j][tb][r]);
        bar();
        write_raw(P1, gg);
        bar();
        const unsigned char* ps = consume();
        if (jh == 0) {
          LANE_LOCAL();
#pragma unroll
          for (int j = 0; j < 4; ++j) { ga[j] = par_t(ps, 1, j, g4); be[j] = par_t(ps, 2, j, g4); }
        }
        add_bias_t(xr, ps);
        release();
        gemm_unit(xr, P1, TL, SITE(2));
      }
      float mean[3], rstd[3];
      row_stats(xr, mean, rstd);
      LANE_LOCAL();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) xr[j][tb][r] = fmaf((xr[j][tb][r] - mean[tb]) * rstd[tb], ga[j][r], be[j][r]);
    }
    if (step - 1 < nb) skip_io(xr, step - 1, true);
    VSTAMP(5);
  } else {
    rows_io(xr, a.x, false);
    VSTAMP(3);
  }
#ifdef RG_STAMPS
  auto stamps_out = [&]() {
    if (stamps && lane0 == 0) {
      VSTAMP(8);
      for (int i = 0; i < 9; ++i) a.dump[((size_t)tile * 8 + wave) * 16 + i] = ts_[i] ? (float)(ts_[i] - ts_[0]) : 0.f;
    }
  };
#endif
  if (a.dump && !stamps) {      // diagnostics: the state behind block step - 1 (launch 0: the input)
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb)
        *reinterpret_cast<f32x4*>(a.dump + ((size_t)tile * TP + 16 * tb + l15) * DM + 64 * wave + 16 * j + 4 * g4) = xr[j][tb];
    wait_vmcnt<0>();
  }
  if (last) {
    // =========================================================== final LayerNorm of the stack, rows back to x
    const unsigned char* ps = consume();
    float mean[3], rstd[3];
    row_stats(xr, mean, rstd);
    {
      LANE_LOCAL();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 ga = par_t(ps, 0, j, g4), be = par_t(ps, 1, j, g4);
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) xr[j][tb][r] = fmaf((xr[j][tb][r] - mean[tb]) * rstd[tb], ga[r], be[r]);
      }
    }
    release();
    rows_io(xr, a.x, true);
#ifdef RG_STAMPS
    stamps_out();
#endif
    return;
  }
  // ======================================================= skip concatenation + Linear(2 D -> D) in front of output block `step`
  bar();                                          // P0 / P1 are free (their last readers are behind the row-statistics barrier)
  write_raw(P0, xr);
  if (step > nb) {
    Acc xs;
    skip_io(xs, 2 * nb - step, false);
    write_raw(P1, xs);
    bar();
    Acc xn;
    zero(xn);
    unit(xn, P0);
    unit(xn, P1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) xr[j][tb] = xn[j][tb];
    bar();
    write_raw(P0, xr);
  }
  rows_io(xr, a.x, true);                         // the residual stream of the next launch
  VSTAMP(6);
  // ======================================================= Q, K (from x + pos) and V (from x) of block `step`
  {
    Acc xp;
    rows_io(xp, const_cast<float*>(a.pos), false);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) xp[j][tb] += xr[j][tb];
    write_raw(P1, xp);
  }
  bar();                                          // P0 = bf16(x), P1 = bf16(x + pos)
  Acc qq, kk, vv;
  zero(qq);
  unit(qq, P1);
  zero(kk);
  unit(kk, P1);
  {
    LANE_LOCAL();
    const unsigned char* ps = consume();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = *reinterpret_cast<const float*>(ps + (16 * j + l15) * 4);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) vv[j][tb] = f32x4{b, b, b, b};
    }
    release();
  }
  gemm_unit(vv, P0, STDL, SITE(3));
  VSTAMP(7);
  {
    LANE_LOCAL();
    // K rows (T layout: row 16 tb + l15, 4 consecutive features) and V transposed (standard layout: feature 16 j + l15,
    // 4 consecutive rows 16 tb + 4 g4 + r)
    const size_t par_w = (size_t)(step & 1) * a.nseq * 160 * DM;
    unsigned short* Kg = reinterpret_cast<unsigned short*>(a.kbuf) + par_w + ((size_t)seq * 160 + row0) * DM + 64 * wave + 4 * g4;
    unsigned short* Vg = reinterpret_cast<unsigned short*>(a.vt) + par_w + ((size_t)seq * DM + 64 * wave + l15) * 160 + row0 + 4 * g4;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const bool ok_t = 16 * tb + l15 < TR, ok_s = 16 * tb + 4 * g4 < TR;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ok_t)
          *reinterpret_cast<u32x2*>(Kg + (size_t)(16 * tb + l15) * DM + 16 * j) = u32x2{pack2(kk[j][tb][0], kk[j][tb][1]), pack2(kk[j][tb][2], kk[j][tb][3])};
        if (ok_s)
          *reinterpret_cast<u32x2*>(Vg + (size_t)(16 * j) * 160 + 16 * tb) = u32x2{pack2(vv[j][tb][0], vv[j][tb][1]), pack2(vv[j][tb][2], vv[j][tb][3])};
      }
    }
  }
  bar();                                          // everyone is done reading P1 (= x + pos)
  write_raw(P1, qq);
  bar();
  {
    u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.qimg) + (size_t)tile * (TP * 1024));
#pragma unroll
    for (int i = 0; i < (TP * 64) / NTH; ++i) dst[tid + NTH * i] = reinterpret_cast<const u32x4*>(P1)[tid + NTH * i];
  }
  wait_vmcnt<0>();
#ifdef RG_STAMPS
  stamps_out();
#endif
}

static int vdec_check(rg_handle* h, const rg_vdec_args& a) {
  RG_REQUIRE(h, a.wstream && a.pstream && a.x && a.pos && a.qimg && a.kbuf && a.vt && a.xbuf, "null pointer");
  RG_REQUIRE(h, a.nseq >= 1 && a.nb >= 1 && 8 * (2 * a.nb + 1) + 2 * a.nb <= MAX_UNITS, "unsupported shape");
  RG_REQUIRE(h, a.step >= 0 && a.step <= 2 * a.nb + 1, "step out of range (0 .. 2 nb + 1)");
  return RG_OK;
}

extern "C" int rg_vdec_step_grouped(rg_handle* h, const rg_vdec_args* args_host, int n, void* stream) {
  RG_REQUIRE(h, args_host && n >= 1 && n <= 4, "1..4 argument blocks");
  rg_vdec_group g;
  int wgs = 0;
  for (int i = 0; i < 4; ++i) {
    g.a[i] = args_host[i < n ? i : 0];
    if (i < n) {
      if (int rc = vdec_check(h, g.a[i])) return rc;
      wgs = max(wgs, 4 * g.a[i].nseq);
    }
  }
  static rg_attr_once lds_once;
  if (!rg_reserve_lds(lds_once, rg_vdec_kernel, LDS_BYTES)) {
    h->err = "rg_vdec_step: cannot reserve LDS";
    return RG_ERR_HIP;
  }
  hipLaunchKernelGGL(rg_vdec_kernel, dim3(wgs, n), dim3(NTH), LDS_BYTES, rg_stream(stream), g);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_vdec_step(rg_handle* h, const rg_vdec_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  return rg_vdec_step_grouped(h, args_host, 1, stream);
}
