// Sequence-stationary body-part VAE encoder (round 4): ONE workgroup owns TWO chunk sequences (S <= 24 token rows each: the
// 15 frames of a chunk + the two distribution tokens) for the WHOLE skip-transformer stack -- every input, middle and output
// block, the skip concatenation linears and the final LayerNorm (gesture_vae.py:111-193 `encode_to_dist`,
// detr_utils.py:101-152 SkipTransformerEncoder, :335-393 TransformerEncoderLayer.forward_post).
//
// Why: as a chain of per-op launches (QKV GEMM, attention, out-projection, LayerNorm, two FFN GEMMs, LayerNorm: 72 grouped
// launches for the nine blocks of the four parts) the exemplar encode of a guided batch costs 5.7 ms alone on the chip and
// 12-27 ms beside the denoiser chains (profiles/r03b_lane_timeline.txt): every launch is a full-chip grid of small tiles that
// waits for compute units, and the activations make a round trip through L2 between any two of them.  A chunk sequence is
// 17 x 512 fp32 = 34 KB: two of them fill the 48-row operand panels of the denoiser's sequence-stationary kernel
// (rg_seq.hip), whose machinery this kernel shares: the fp32 residual stream lives in registers (lane = token row, 4
// consecutive features), the bf16 operand panels in LDS, and the only traffic is the WEIGHT STREAM -- 8 unit GEMMs of
// 512 x 512 per block (+ 2 per skip linear), packed on the host in the order and MFMA-fragment layout a wave consumes them,
// fetched by every wave for itself with LDS-DMA into a private ring behind a counted vmcnt.
//
// Layouts (as rg_seq.hip).  MFMA 16x16x32 bf16: lane L = (l15 = L & 15, g4 = L >> 4) holds A[i = l15][k = 8 g4 + e],
// B[k = 8 g4 + e][j = l15], D[i = 4 g4 + r][j = l15].
//   "T layout"  out[n][t]: A = weight fragment, B = panel fragment; the lane holds token t = l15 of a 16-token block and 4
//               consecutive features; wave w owns features [64 w, 64 w + 64).
//   "standard"  out[t][n]: roles swapped; used for V, whose accumulators then ARE the A operand of O^T = V^T P.
// Rows: sequence 0 sits in panel rows [0, S), sequence 1 in rows [24, 24 + S); the other rows repeat a valid row (finite,
// never stored) and are masked as keys.
// Softmax attention (torch.nn.MultiheadAttention, 4 heads of 128): Q (pre-scaled by 1/sqrt(128), folded into its weights)
// and K go to the two panels as bf16; head h = the wave pair (2 h, 2 h + 1); each wave computes the head's 48 x 48 score
// blocks S^T = K Q^T from the panels (contraction over the head's 128 features), masks keys of the other sequence / padding,
// takes the softmax over keys per query (lane), and forms O^T = V^T P for its own 64 features from V's accumulators.
#define RG_PACK2_ONE      // (two waves x >= 247 registers per SIMD: nothing shares this kernel's SIMDs; rg_common.h rg_pack2_bf16)
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
constexpr int SQ = 24;         // row stride of the two sequences of a workgroup
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

struct rg_venc_group { rg_venc_args a[4]; };   // up to four stacks (the four body parts) in one launch: blockIdx.y picks

__global__ void __launch_bounds__(NTH) rg_venc_kernel(const rg_venc_group grp) {
  RG_OWN_THE_SIMD();
  const rg_venc_args& a = grp.a[blockIdx.y];
  if ((int)blockIdx.x >= (a.nseq + 1) / 2) return;
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
  const int S = a.S, nb = a.nb;                       // tokens per sequence; input (= output) blocks
  const int n_blocks = 2 * nb + 1;
  const int NU = 8 * n_blocks + 2 * nb;               // unit GEMMs in the stream
  const int n_seg = 2 * NU + 1;                       // (P, W) per unit + the final norm's parameter fragment
  const int wg = blockIdx.x;
  const int seq0 = 2 * wg, seq1 = min(2 * wg + 1, a.nseq - 1);      // (an odd tail repeats its sequence; stored once)
  const bool store1 = 2 * wg + 1 < a.nseq;

  // ---- fetch program: {address for wave 0, fragment count, wave stride in fragments} per segment + sentinel
  if (tid <= n_seg) {
    const unsigned char* adr = reinterpret_cast<const unsigned char*>(a.wstream);
    unsigned cnt = 1u << 30, stride = 0;
    if (tid < n_seg) {
      const int u = tid >> 1;
      if ((tid & 1) == 0) {       // parameter fragment: pstream [NU + 1][8][1 KiB]
        adr = reinterpret_cast<const unsigned char*>(a.pstream) + ((size_t)u * 8 << 10);
        cnt = 1; stride = 1;
      } else {                    // weights: wstream [NU][8][64][1 KiB]
        adr = reinterpret_cast<const unsigned char*>(a.wstream) + ((size_t)u * 512 << 10);
        cnt = 64; stride = 64;
      }
    }
    const unsigned long long av = reinterpret_cast<unsigned long long>(adr);
    *reinterpret_cast<u32x4*>(smem + OFF_DESC + tid * 16) = u32x4{(unsigned)av, (unsigned)(av >> 32), cnt, stride};
  }

  // row -> (sequence, position): T layout rows 16 tb + l15, standard layout rows 16 tb + 4 g4 + r
  auto row_seq = [&](int r) { return r >= SQ ? seq1 : seq0; };
  auto row_pos = [&](int r) { return min(r >= SQ ? r - SQ : r, S - 1); };

  // ---- residual stream, T layout: xr[j][tb][r] = x[row 16 tb + l15][feature 64 wave + 16 j + 4 g4 + r]
  Acc xr;
  // key masks for the attention, per lane: bit (12 qb + 4 kb + r) = key row 16 kb + 4 g4 + r may be attended by query row
  // 16 qb + l15 (same sequence, not padding)
  unsigned long long kbits0 = 0;
  {
    LANE_LOCAL();
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const int r = 16 * tb + l15;
      const float* xp = a.x + ((size_t)row_seq(r) * S + row_pos(r)) * DM + 64 * wave + 4 * g4;
#pragma unroll
      for (int j = 0; j < 4; ++j) xr[j][tb] = *reinterpret_cast<const f32x4*>(xp + 16 * j);
    }
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
      const int q = 16 * qb + l15;
#pragma unroll
      for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * kb + 4 * g4 + r;
          const bool ok = ((k >= SQ) == (q >= SQ)) && ((k >= SQ ? k - SQ : k) < S);
          if (ok) kbits0 |= 1ull << (12 * qb + 4 * kb + r);
        }
    }
  }

  // ---- T-layout values -> bf16 panel fragments (8-byte stores)
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
  write_raw(P0, xr);
  __syncthreads();     // descriptors + P0 written

  // ---- the wave's fetch cursor (all wave-uniform)
  int ie = 0, ir = 0;
  int cur_cnt = 0;
  __amdgpu_buffer_rsrc_t cur_rsrc;
  const int lane16 = lane0 * 16;
  auto load_seg = [&]() {
    const u32x4 d = *reinterpret_cast<const u32x4*>(smem + OFF_DESC + ie * 16);
    const unsigned lo = __builtin_amdgcn_readfirstlane(d[0]), hi = __builtin_amdgcn_readfirstlane(d[1]);
    cur_cnt = __builtin_amdgcn_readfirstlane(d[2]);
    const unsigned stride = __builtin_amdgcn_readfirstlane(d[3]);
    unsigned char* base = reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo) + ((size_t)(wave * stride) << 10);
    cur_rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
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

  // ---- unit GEMM over K = 512 (16 steps of 32), 4 weight fragments per step (see rg_seq.hip: gemm_frags_reg): the unit's first
  // RD fragments -- issued before the unit starts, across its epilogue -- come through the LDS ring, the other 64 - RD straight
  // into registers (eight in rotation, RD in flight as before); the last RD iterations refill the ring's slots for whatever the
  // stream holds next.  `head` leaves as it came.
  auto issue_reg = [&](u32x4& dst) {
    dst = __builtin_amdgcn_raw_buffer_load_b128(cur_rsrc, lane16, ir << 10, 0);
    if (++ir == cur_cnt) {
      ir = 0;
      ++ie;
      load_seg();
    }
  };
  auto gemm_unit = [&](Acc& acc, const unsigned char* panel, auto std_tag) {
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
  std::false_type TL;
  std::true_type STDL;

  auto par_t = [&](const unsigned char* slot, int p, int j, int g4) -> f32x4 {   // T layout: features 16 j + 4 g4 + r
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
  // plain unit: acc += bias (vector 0 of the unit's parameter fragment), then acc += W x panel
  auto unit = [&](Acc& acc, const unsigned char* panel) {
    const unsigned char* ps = consume();
    add_bias_t(acc, ps);
    release();
    gemm_unit(acc, panel, TL);
  };

  // ---- LayerNorm statistics of the three token rows a lane holds (rg_seq.hip: row_stats)
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
  // x = LayerNorm(x) * gamma + beta in place (gamma, beta = vectors gi, gi + 1 of parameter fragment ps), P0 = bf16(x)
  auto layer_norm = [&](Acc& v, const unsigned char* ps, int gi, bool to_panel) {
    float mean[3], rstd[3];
    row_stats(v, mean, rstd);        // (its barrier also separates the panel's last readers from the write below)
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 ga = par_t(ps, gi, j, g4), be = par_t(ps, gi + 1, j, g4);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][tb][r] = fmaf((v[j][tb][r] - mean[tb]) * rstd[tb], ga[r], be[r]);
        if (to_panel) panel_store(P0, l15, g4, j, tb, v[j][tb][0], v[j][tb][1], v[j][tb][2], v[j][tb][3]);
      }
    }
  };

  float* const xbuf = a.xbuf + (size_t)wg * nb * (NW * 12 * 64 * 4);
  auto skip_io = [&](Acc& v, int slot, bool store) {     // the skip stack: 12 wave-instructions of 1 KiB, lane-linear
    LANE_LOCAL();
    float* base = xbuf + ((size_t)slot * NW + wave) * (12 * 64 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        f32x4* p = reinterpret_cast<f32x4*>(base + ((j * 3 + tb) * 64 + lane) * 4);
        if (store) *p = v[j][tb]; else v[j][tb] = *p;
      }
    wait_vmcnt<0>();
  };

#pragma unroll 1
  for (int blk = 0; blk < n_blocks; ++blk) {
    // ======================================================= skip concatenation + Linear(2 D -> D) in front of an output block
    if (blk > nb) {
      Acc xs;
      skip_io(xs, 2 * nb - blk, false);          // output block i pops the state saved behind input block nb - 1 - i
      // (the vmcnt(0) behind these register-destination loads also lands every ring fragment in flight: harmless, consume()
      //  only ever waits for the OLDEST one and every slot is re-issued as it is read)
      bar();                                      // P1's last readers (FF2 of the previous block) are done
      write_raw(P1, xs);
      bar();
      Acc xn;
      zero(xn);
      unit(xn, P0);                               // W[:, :D] x      (+ bias)
      unit(xn, P1);                               // W[:, D:] xs     (this slot's bias vector is zero)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) xr[j][tb] = xn[j][tb];
      bar();                                      // everyone is done reading P0 / P1
      write_raw(P0, xr);
      bar();
    }
    // ======================================================= self attention (detr_utils.py:364-366, nn.MultiheadAttention)
    {
      Acc qq;
      zero(qq);
      unit(qq, P0);                               // Q (1 / sqrt(128) folded into the weights and the bias)
      write_raw(P1, qq);                          // (P1's last readers -- FF2 of the previous block / the skip linear -- are behind a barrier)
    }
    Acc vv;
    {
      Acc kk;
      zero(kk);
      unit(kk, P0);                               // K
      // V in the standard layout: vv[j][tb][r] = V[row 16 tb + 4 g4 + r][feature 64 wave + 16 j + l15]
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
      gemm_unit(vv, P0, STDL);
      bar();                                      // everyone is done reading P0 (= x) for Q, K, V
      write_raw(P0, kk);
      bar();                                      // P0 = K, P1 = Q complete
    }
    Acc oo;
    {
      // scores of head h = wave >> 1: sc[kb][qb] = D[key 16 kb + 4 g4 + r][query 16 qb + l15], contraction over the head's
      // 128 features = panel fragments s = 4 h .. 4 h + 3
      LANE_LOCAL();
      const int hs = (wave >> 1) * 4;
      f32x4 sc[3][3];
#pragma unroll
      for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) sc[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 kf[3], qf[3];
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          kf[tb] = *reinterpret_cast<const bf16x8*>(P0 + ((tb * 16 + hs + s) << 10) + lane * 16);
          qf[tb] = *reinterpret_cast<const bf16x8*>(P1 + ((tb * 16 + hs + s) << 10) + lane * 16);
        }
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
          for (int qb = 0; qb < 3; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb], qf[qb], sc[kb][qb], 0, 0, 0);
      }
      unsigned long long kbits = kbits0;
      asm volatile("" : "+v"(kbits));
      // softmax over the keys of each query (lane): keys = 3 blocks x 4 registers x 4 lane groups
#pragma unroll
      for (int qb = 0; qb < 3; ++qb) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((kbits >> (12 * qb + 4 * kb + r)) & 1ull) mx = fmaxf(mx, sc[kb][qb][r]);
        mx = xmax4(mx);
        const float nm2 = mx * -1.44269504088896340736f;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = ((kbits >> (12 * qb + 4 * kb + r)) & 1ull) ? rg_exp_sub(sc[kb][qb][r], nm2) : 0.f;
            sc[kb][qb][r] = e;
            sum += e;
          }
        sum = xsum4(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) sc[kb][qb] *= inv;
      }
      // O^T[feature][query] = sum_key V[key][feature] P[key][query]: k-step 0 = key blocks 0 | 1, k-step 1 = block 2 | zeros;
      // A = V's accumulators (bf16), B = P as bf16 hi + lo
#pragma unroll
      for (int qb = 0; qb < 3; ++qb) {
        const float p0[8] = {sc[0][qb][0], sc[0][qb][1], sc[0][qb][2], sc[0][qb][3], sc[1][qb][0], sc[1][qb][1], sc[1][qb][2], sc[1][qb][3]};
        const float p1[8] = {sc[2][qb][0], sc[2][qb][1], sc[2][qb][2], sc[2][qb][3], 0.f, 0.f, 0.f, 0.f};
        bf16x8 ph0, pl0, ph1, pl1;
        split_hl(p0, ph0, pl0);
        split_hl(p1, ph1, pl1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0[8] = {vv[j][0][0], vv[j][0][1], vv[j][0][2], vv[j][0][3], vv[j][1][0], vv[j][1][1], vv[j][1][2], vv[j][1][3]};
          const float v1[8] = {vv[j][2][0], vv[j][2][1], vv[j][2][2], vv[j][2][3], 0.f, 0.f, 0.f, 0.f};
          const bf16x8 va = pack8(v0), vb = pack8(v1);
          f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pl0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pl1, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, ph0, d, 0, 0, 0);
          oo[j][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, ph1, d, 0, 0, 0);
        }
      }
    }
    bar();                                        // everyone is done with the K / Q panels
    write_raw(P1, oo);
    bar();
    // ======================================================= x = LayerNorm1(x + out_proj(attention))
    {
      const unsigned char* ps = consume();        // vectors: 0 bias, 1 gamma1, 2 beta1
      f32x4 ga[4], be[4];
      {
        LANE_LOCAL();
#pragma unroll
        for (int j = 0; j < 4; ++j) { ga[j] = par_t(ps, 1, j, g4); be[j] = par_t(ps, 2, j, g4); }
      }
      add_bias_t(xr, ps);
      release();
      gemm_unit(xr, P1, TL);
      float mean[3], rstd[3];
      row_stats(xr, mean, rstd);                  // (barrier inside: P0's readers -- the scores -- are long done)
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
    // ======================================================= x = LayerNorm2(x + linear2(gelu(linear1(x)))): 1024 hidden in two halves
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
        bar();                                    // P1 is free
        write_raw(P1, gg);
        bar();
        const unsigned char* ps = consume();      // first half: 0 bias2, 1 gamma2, 2 beta2; second half: zeros
        if (jh == 0) {
          LANE_LOCAL();
#pragma unroll
          for (int j = 0; j < 4; ++j) { ga[j] = par_t(ps, 1, j, g4); be[j] = par_t(ps, 2, j, g4); }
        }
        add_bias_t(xr, ps);
        release();
        gemm_unit(xr, P1, TL);                    // accumulates onto the residual
      }
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
    if (blk < nb) skip_io(xr, blk, true);         // input block: push its output on the skip stack
    if (a.dump && blk == a.dump_block) {
      LANE_LOCAL();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          *reinterpret_cast<f32x4*>(a.dump + ((size_t)wg * TP + 16 * tb + l15) * DM + 64 * wave + 16 * j + 4 * g4) = xr[j][tb];
      wait_vmcnt<0>();
    }
  }

  // =========================================================== final LayerNorm (encoder.norm) and the store of the valid rows
  {
    const unsigned char* ps = consume();
    layer_norm(xr, ps, 0, false);
    release();
    LANE_LOCAL();
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const int r = 16 * tb + l15;
      const bool second = r >= SQ;
      const int pos = second ? r - SQ : r;
      if (pos < S && (!second || store1)) {
        float* op = a.out + ((size_t)(second ? seq1 : seq0) * S + pos) * DM + 64 * wave + 4 * g4;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(op + 16 * j) = xr[j][tb];
      }
    }
  }
  wait_vmcnt<0>();
}

static int venc_check(rg_handle* h, const rg_venc_args& a) {
  RG_REQUIRE(h, a.wstream && a.pstream && a.x && a.out && a.xbuf, "null pointer");
  RG_REQUIRE(h, a.nseq >= 1 && a.S >= 1 && a.S <= SQ, "unsupported shape (1 <= S <= 24 tokens per sequence)");
  RG_REQUIRE(h, a.nb >= 1 && 8 * (2 * a.nb + 1) + 2 * a.nb <= MAX_UNITS, "unsupported depth (1 <= blocks per side <= 8)");
  RG_REQUIRE(h, a.dump_block < 0 || a.dump, "dump_block needs a dump buffer");
  return RG_OK;
}

extern "C" int rg_venc_forward_grouped(rg_handle* h, const rg_venc_args* args_host, int n, void* stream) {
  RG_REQUIRE(h, args_host && n >= 1 && n <= 4, "1..4 argument blocks");
  rg_venc_group g;
  int wgs = 0;
  for (int i = 0; i < 4; ++i) {
    g.a[i] = args_host[i < n ? i : 0];
    if (i < n) {
      if (int rc = venc_check(h, g.a[i])) return rc;
      wgs = max(wgs, (g.a[i].nseq + 1) / 2);
    }
  }
  static rg_attr_once lds_once;
  if (!rg_reserve_lds(lds_once, rg_venc_kernel, LDS_BYTES)) {
    h->err = "rg_venc_forward: cannot reserve LDS";
    return RG_ERR_HIP;
  }
  hipLaunchKernelGGL(rg_venc_kernel, dim3(wgs, n), dim3(NTH), LDS_BYTES, rg_stream(stream), g);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_venc_forward(rg_handle* h, const rg_venc_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  return rg_venc_forward_grouped(h, args_host, 1, stream);
}
