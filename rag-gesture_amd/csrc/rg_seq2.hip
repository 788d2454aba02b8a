// Sequence-stationary denoiser forward, TWO sequences per workgroup (round 5).
//
// rg_seq.hip (round 3) keeps one sequence per workgroup and is bound by what a compute unit can take in: every streamed 1-KiB
// weight fragment feeds 3 MFMAs (48 token rows), 0.47 of the matrix pipe at best (profiles/r05a_seq_pair_probe.txt: 5.0 us per
// 512 x 512 unit).  Here a workgroup owns two sequences OF THE SAME KIND (two conditional sequences, or two classifier-free
// ones, of the same diffusion step): they consume the same units in the same order, so each fragment feeds 6 MFMAs (96 rows,
// 6.7 us per unit = 0.77 of the pipe) and the weight bytes per row halve.
// What had to move for that (NOTEBOOK 9.4 counted the registers): a wave's 256 VGPRs hold ONE fp32 [2 x 48 x 64] tile, so
//   * the residual stream no longer lives in registers across a block: it takes the fp32 round trip through `xbuf` (L2, lane-
//     linear 1-KiB wave instructions) twice per layer (self attention, FFN) -- stored when a block starts, loaded back as the accumulator
//     initialiser of the block's output projection while the stylization in front of it is being computed;
//   * one bf16 panel per sequence (2 x 48 KiB) instead of two.  Operands that the old kernel parked in the second panel either
//     wait in registers as packed bf16 (48 VGPRs for both sequences: the two GELU halves of the FFN, units re-ordered FF1_0,
//     FF1_1, FF2_0, FF2_1) or go through `gbuf` (L2) as bf16 panel images, every wave writing and later restoring its own
//     fragments: the stylized rows of the three conditions (held in registers beside the queries' accumulator they spilled
//     ~120 registers each).  Cross-attention order (round 6): Q3_0, Q3_1, Q3_2, MIXX | MIX_0, MIX_1, MIX_2 -- every unit that
//     reads the normalised x first (the residual stream is dead there, so nothing but the queries is live), then the three
//     units that read the stylized rows: no second copy of xhat, no round trip of the accumulator (round 5 had both);
//   * the six panel fragments of a k-step live in one set of registers, re-read in place behind their last MFMA;
//   * everything that touches memory besides the ring is issued at the START of an epilogue (statistics, barriers), never in
//     front of a GEMM loop: the ring's counted vmcnt then rarely waits for it.  No scratch, 251 VGPRs.
// Arithmetic per sequence is the old kernel's, operation for operation (same MFMA accumulation order, same epilogues): the two
// kernels agree bit for bit (tests/test_denoiser_gpu.py), and a sequence's result does not depend on its partner.
// Layouts, weight / parameter / table streams: rg_seq.hip and include/rg_gesture.h (rg_seq_args).
// reference: mogen/models/transformers/diffusion_transformer.py:105-127 (DecoderLayer), :74-87 (FFN), :620-668 (forward);
// mogen/models/attentions/efficient_attention.py:23-45, 62-102; mogen/models/utils/stylization_block.py:29-40;
// mogen/models/transformers/raggesture.py:1041-1085 (classifier-free row doubling).
#ifndef RG2_PACK_TWO       // (experiment switch: the two-conversion form here too)
#define RG_PACK2_ONE      // the one-instruction bf16 pair (rg_common.h rg_pack2_bf16): ONLY because rg_seq2_kernel owns its SIMDs (RG_OWN_THE_SIMD below)
#endif
#include "rg_common.h"
#include "rg_tail.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int DM = 512;        // model width
constexpr int TP = 48;         // token rows of a panel (T <= 48; rows >= T repeat token T - 1)
constexpr int NW = 8;          // waves per workgroup; wave w owns features [64 w, 64 w + 64) = heads 2 w, 2 w + 1 of BOTH sequences
constexpr int NTH = NW * 64;
constexpr int RD = 6;          // ring slots (1 KiB) per wave
constexpr int UPL = 16;        // unit slots per layer in the weight stream
constexpr int PANEL = TP * 1024;                         // bytes of one sequence's bf16 panel
constexpr int OFF_RING = 2 * PANEL;
constexpr int NSEG_COND = 36, NSEG_UNC = 19;             // fetch segments per layer
constexpr int MAX_SEG = 8 * NSEG_COND + 5;               // + embed (2), head (2), sentinel
constexpr int OFF_DESC = OFF_RING + NW * RD * 1024;      // [MAX_SEG] x 8 B fetch segments {address of wave 0 (48 bits), count (8), wave stride (8)}
constexpr int OFF_STAT = OFF_DESC + ((MAX_SEG * 8 + 15) & ~15);   // [parity 2][sequence 2][NW][TP][2] fp32 partial (sum, M2)
constexpr int STAT_HALF = 2 * NW * TP * 2;               // floats of one parity's partials
constexpr int LDS_BYTES = OFF_STAT + 2 * STAT_HALF * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

enum { U_KV = 0, U_KV2, U_Q, U_SAO, U_MIXX, U_Q3_0, U_MIX_0, U_Q3_1, U_MIX_1, U_Q3_2, U_MIX_2, U_FF1_0, U_FF2_0, U_FF1_1, U_FF2_1, U_FFO };
// fetch segments of one layer IN CONSUMPTION ORDER: unit slot << 2 | kind (0 = parameter fragment, 1 = weights, 2 = extra of
// the first sequence (its clip's A fragments; the classifier-free table), 3 = A fragments of the second sequence's clip)
#define SEG(u, k) ((u) << 2 | (k))
__constant__ const unsigned char SEG2_COND[NSEG_COND + 1] = {
    SEG(U_KV, 0), SEG(U_KV, 1), SEG(U_Q, 0), SEG(U_Q, 1), SEG(U_SAO, 0), SEG(U_SAO, 1),
    SEG(U_Q3_0, 0), SEG(U_Q3_0, 1), SEG(U_Q3_0, 2), SEG(U_Q3_0, 3), SEG(U_MIX_0, 0),
    SEG(U_Q3_1, 0), SEG(U_Q3_1, 1), SEG(U_Q3_1, 2), SEG(U_Q3_1, 3), SEG(U_MIX_1, 0),
    SEG(U_Q3_2, 0), SEG(U_Q3_2, 1), SEG(U_Q3_2, 2), SEG(U_Q3_2, 3), SEG(U_MIX_2, 0),
    SEG(U_MIXX, 0), SEG(U_MIXX, 1), SEG(U_MIX_0, 1), SEG(U_MIX_1, 1), SEG(U_MIX_2, 1),
    SEG(U_FF1_0, 0), SEG(U_FF1_0, 1), SEG(U_FF1_1, 0), SEG(U_FF1_1, 1), SEG(U_FF2_0, 0), SEG(U_FF2_0, 1), SEG(U_FF2_1, 0), SEG(U_FF2_1, 1),
    SEG(U_FFO, 0), SEG(U_FFO, 1), 0};
__constant__ const unsigned char SEG2_UNC[NSEG_UNC + 1] = {
    SEG(U_KV, 0), SEG(U_KV, 1), SEG(U_Q, 0), SEG(U_Q, 1), SEG(U_SAO, 0), SEG(U_SAO, 1), SEG(U_MIXX, 0), SEG(U_MIXX, 1), SEG(U_MIXX, 2),
    SEG(U_FF1_0, 0), SEG(U_FF1_0, 1), SEG(U_FF1_1, 0), SEG(U_FF1_1, 1), SEG(U_FF2_0, 0), SEG(U_FF2_0, 1), SEG(U_FF2_1, 0), SEG(U_FF2_1, 1),
    SEG(U_FFO, 0), SEG(U_FFO, 1), 0};
#undef SEG

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return rg_pack2_bf16(lo, hi); }
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
__device__ __forceinline__ float gelu_fast(float v) { return rg_gelu_erf(v); }
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  return __builtin_bit_cast(bf16x8, u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])});
}
// sum / max over the four 16-lane groups of a wave on the VALU (rg_seq.hip)
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
// the BUILTIN forms: the compiler's wait-count bookkeeping sees them (rg_seq.hip: wait_lds)
__device__ __forceinline__ void wait_lds() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  asm volatile("" ::: "memory");
}
// every vector-memory operation of the wave has completed.  Placed where the ring's fragments have long landed (behind an
// epilogue's barriers): it tells the compiler that registers loaded from xbuf are ready, so that it does not put a counted
// wait of its own INSIDE the GEMM loop that first uses them (which would drain the ring on every iteration).
__device__ __forceinline__ void wait_vm_all() {
  __builtin_amdgcn_s_waitcnt(0x0f70);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

#ifdef RG_STAMPS
// Diagnostic build only (build.py RG_DIAG=1): wall-clock (100 MHz) time per category, summed per wave, written to
// a.dump[(workgroup * 8 + wave) * 12 + category] when dump_stage == 99 (8: panel writes outside the units' epilogues; 9: the
// mix_x scalings and classifier-free table adds; 10: the prologue up to the ring's first fill).  Categories: 0 unit GEMMs, 1 row statistics (with their
// barrier), 2 other barriers, 3 parameter fragments + panel writes, 4 attention math, 5 whole pass, 6 xbuf / gbuf traffic,
// 7 waiting in consume() (counted inside whatever category encloses it).
#define TSTART() const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime()
#define TSTOP(cat) tacc[cat] += __builtin_amdgcn_s_memrealtime() - t0_
// + the duration of every gemm_frags call, in call order: a.dump[(1 << 20) + (workgroup * 8 + wave) * 512 + call]
#define TLOG() if (a.dump_stage == 99 && lane0 == 0 && ncall < 512) a.dump[(1 << 20) + (blockIdx.x * 8 + wave) * 512 + ncall++] = (float)(__builtin_amdgcn_s_memrealtime() - t0_)
#else
#define TSTART()
#define TSTOP(cat)
#define TLOG()
#endif

typedef f32x4 Acc[4][3];       // one sequence: [16-feature block of the wave's 64][16-token block]
typedef f32x4 Acc2[2][4][3];   // both sequences
typedef u32x2 Held[2][4][3];   // both sequences' T-layout values as packed bf16 (the panel image of the wave's 64 features)

__device__ __forceinline__ void zero(Acc2& a) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) a[q][j][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace

// One forward of two sequences sA, sB of the same kind and step group (sB == sA: a lone sequence, computed twice).
__device__ __forceinline__ void run_pair(const rg_seq_args& a, const int sA, const int sB, unsigned char* const smem) {
  float* const sStat = reinterpret_cast<float*>(smem + OFF_STAT);
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // (opaque: nothing derived from the thread id is an invariant of the caller's pass loop)
  const int tid = tid_, lane0 = tid & 63;
  // Lane-derived values are re-derived from an opaque copy of the lane id wherever they are used (rg_seq.hip: LANE_LOCAL)
#define LANE_LOCAL()                      \
  int ln_ = lane0;                        \
  asm volatile("" : "+v"(ln_));         \
  const int lane = ln_, l15 = ln_ & 15, g4 = ln_ >> 4; \
  (void)lane; (void)l15; (void)g4
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const ring = smem + OFF_RING + wave * (RD * 1024);
  const int T = a.T, B = a.B, L = a.L, R = 2 * a.B;
  const bool cond = sA < B;
  const int seqs[2] = {sA, sB};
  const int clips[2] = {cond ? sA : sA - B, cond ? sB : sB - B};
  const int st = clips[0] >= a.split ? a.step_b : a.step;
#ifdef RG_STAMPS
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ncall = 0;
  const unsigned long long tk0 = __builtin_amdgcn_s_memrealtime();
#endif
  auto barx = [&]() {
    TSTART();
    bar();
    TSTOP(2);
  };
  const int NU = UPL * L + 2;
  const int nspl = cond ? NSEG_COND : NSEG_UNC;
  const int n_seg = 2 + nspl * L + 2;                     // embed (P, W), layers, head (P, W)

  // ---- fetch program: one {address for wave 0, fragment count, wave stride in fragments} per segment, in consumption order,
  // + a sentinel that keeps the in-flight count invariant behind the end
  if (tid <= n_seg) {
    const unsigned char* adr = reinterpret_cast<const unsigned char*>(a.wstream);
    unsigned cnt = 0, stride = 0;                        // (the sentinel: a count the cursor never reaches)
    if (tid < n_seg) {
      int uid, kind, idx = 0, l = 0;
      if (tid < 2) { uid = 0; kind = tid; }
      else if (tid >= 2 + nspl * L) { uid = NU - 1; kind = tid - (2 + nspl * L); }
      else {
        const int q = tid - 2;
        l = q / nspl;
        const unsigned char e = cond ? SEG2_COND[q - l * nspl] : SEG2_UNC[q - l * nspl];
        idx = e >> 2;
        kind = e & 3;
        uid = 1 + UPL * l + idx;
      }
      if (kind == 0) {            // parameter fragment: pstream [S][NU][8][1 KiB]
        adr = reinterpret_cast<const unsigned char*>(a.pstream) + ((size_t)(st * NU + uid) * 8 << 10);
        cnt = 1; stride = 1;
      } else if (kind == 1) {     // weights: wstream [NU][8][64][1 KiB] (U_KV: [8][128] over two slots)
        adr = reinterpret_cast<const unsigned char*>(a.wstream) + ((size_t)uid * 512 << 10);
        cnt = (uid > 0 && uid < NU - 1 && idx == U_KV) ? 128 : 64;
        stride = cnt;
      } else if (cond) {          // A fragments of (layer, condition, clip): afrag [L][3][B][8][4 KiB]
        const int c = (idx - U_Q3_0) >> 1;
        adr = reinterpret_cast<const unsigned char*>(a.afrag) + ((size_t)((l * 3 + c) * B + (kind == 2 ? clips[0] : clips[1])) * 32 << 10);
        cnt = 4; stride = 4;
      } else {                    // classifier-free cross-attention contribution: ustream [S][L][8][2 KiB]
        adr = reinterpret_cast<const unsigned char*>(a.ustream) + ((size_t)(st * L + l) * 16 << 10);
        cnt = 2; stride = 2;
      }
    }
    const unsigned long long av = reinterpret_cast<unsigned long long>(adr);      // (device addresses: 48 bits)
    *reinterpret_cast<u32x2*>(smem + OFF_DESC + tid * 8) = u32x2{(unsigned)av, ((unsigned)(av >> 32) & 0xffffu) | cnt << 16 | stride << 24};
  }

  // ---- token masks, per lane and sequence: bit (4 tb + r) of tokbits = token 16 tb + 4 g4 + r takes part in the self
  // attention (standard layout); bit (3 c + tb) of qbits = query token 16 tb + l15 of condition c is masked (T layout)
  unsigned tokbits0[2] = {0, 0}, qbits0[2] = {0, 0};
  Acc2 X;                       // the block's fp32 [token][feature] tile: residual stream / accumulator of the output projections
  {
    LANE_LOCAL();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = 16 * tb + 4 * g4 + r;
          if (t < T && a.src_mask[(size_t)seqs[q] * T + t] != 0.f) tokbits0[q] |= 1u << (4 * tb + r);
        }
        const int tq = min(16 * tb + l15, T - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (a.qmask[((size_t)c * R + seqs[q]) * T + tq] == 0.f) qbits0[q] |= 1u << (3 * c + tb);
      }
    // residual stream, T layout: X[q][j][tb][r] = x[token 16 tb + l15][feature 64 wave + 16 j + 4 g4 + r]; starts as the
    // positional tables (diffusion_transformer.py:646-659), the embedding GEMM accumulates onto it
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const int t = min(16 * tb + l15, T - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        X[0][j][tb] = *reinterpret_cast<const f32x4*>(a.tbias + (size_t)t * DM + 64 * wave + 16 * j + 4 * g4);
        X[1][j][tb] = X[0][j][tb];
      }
    }
  }
  // ---- panels = bf16(x_in): fragment (tb, s) = tokens [16 tb, +16) x features [32 s, +32): lane (l15, g) holds the 8
  // features [32 s + 8 g, +8) of token 16 tb + l15
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int it = 0; it < (TP * 64) / NTH; ++it) {
      const int slot = tid + NTH * it, t = slot >> 6, c8 = slot & 63;
      const float* xp = a.x + ((size_t)clips[q] * T + min(t, T - 1)) * DM + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + 4);
      *reinterpret_cast<u32x4*>(smem + q * PANEL + (((t >> 4) * 16 + (c8 >> 2)) << 10) + (((t & 15) + 16 * (c8 & 3)) << 4)) =
          u32x4{pack2(v0[0], v0[1]), pack2(v0[2], v0[3]), pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
    }
  __syncthreads();     // descriptors + panels written; every register-destination load above has been waited for

  // ---- the wave's fetch cursor (all state wave-uniform): buffer descriptor of the segment + a scalar offset (rg_seq.hip)
  int ie = 0, ir = 0;
  int cur_cnt = 0;
  __amdgpu_buffer_rsrc_t cur_rsrc;
  const int lane16 = lane0 * 16;
  auto load_seg = [&]() {
    const u32x2 d = *reinterpret_cast<const u32x2*>(smem + OFF_DESC + ie * 8);
    const unsigned lo = __builtin_amdgcn_readfirstlane(d[0]), w1 = __builtin_amdgcn_readfirstlane(d[1]);
    cur_cnt = (w1 >> 16) & 0xffu;
    const unsigned stride = w1 >> 24;
    unsigned char* base = reinterpret_cast<unsigned char*>(((unsigned long long)(w1 & 0xffffu) << 32) | lo) + ((size_t)(wave * stride) << 10);
    cur_rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  };
  auto issue = [&](int slot) {
    // (aux 0: cached in L2.  Non-temporal loads cut the HBM-side traffic of a launch from x8.4 to x6.3 of the algorithmic bytes but
    //  cost 11 % of its time: the four workgroups of an XCD share one fetch of the stream through its L2)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_rsrc, (lds_void*)(ring + slot * 1024), 16, lane16, ir << 10, 0, 0);
    if (__builtin_expect(++ir == cur_cnt, 0)) {
      ir = 0;
      ++ie;
      load_seg();
    }
  };
  int head = 0;                                  // ring slot of the oldest fragment in flight
  // The oldest fragment has landed when at most RD - 1 younger vector-memory operations are outstanding (vmcnt retires in issue
  // order).  Behind a burst of xbuf / gbuf transfers the first waits also cover the burst: measured cheaper than a second,
  // relaxed wait chosen per fragment by a wave-uniform counter (1 846 vs 1 924 us per 128-sequence launch).
  auto ring_wait = [&]() { wait_vmcnt<RD - 1>(); };
  auto consume = [&]() -> const unsigned char* {
#ifdef RG_STAMPS
    const unsigned long long tc0_ = __builtin_amdgcn_s_memrealtime();
    ring_wait();
    tacc[7] += __builtin_amdgcn_s_memrealtime() - tc0_;      // (category 7: waiting for a parameter / table fragment, i.e. behind a burst)
#else
    ring_wait();
#endif
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
#ifdef RG_STAMPS
  tacc[10] = __builtin_amdgcn_s_memrealtime() - tk0;
#endif

  // ---- fp32 round trip of the wave's [2 x 48 x 64] tile through xbuf, and the bf16 panel image in gbuf
  float* const Rw = a.xbuf + ((size_t)blockIdx.x * 2 * NW + wave) * (12 * 64 * 4);      // + q * NW * 12 * 256 floats
  unsigned char* const Gw = reinterpret_cast<unsigned char*>(a.gbuf) + (size_t)blockIdx.x * (8 * PANEL);      // [slot 4][sequence 2][PANEL]
  auto store_R = [&](const Acc2& v) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          *reinterpret_cast<f32x4*>(Rw + ((size_t)(q * NW * 12 + j * 3 + tb) * 64 + lane) * 4) = v[q][j][tb];
    TSTOP(6);
  };
  auto load_R = [&](Acc2& v) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          v[q][j][tb] = *reinterpret_cast<const f32x4*>(Rw + ((size_t)(q * NW * 12 + j * 3 + tb) * 64 + lane) * 4);
    TSTOP(6);
  };

  // ---- unit GEMM over K = 512 for BOTH sequences: per fragment 6 MFMAs.  NJ = 4: the wave's 64 features, NJ = 2: the 32
  // features of one head.  STD = false: T layout (A = weights); true: standard layout (A = panel).
  // Per fragment f: [its LDS read has landed -> refill its ring slot] [fragment f + 1 has landed in LDS -> issue its LDS read]
  // [the 6 MFMAs of f].  The six panel fragments of a k-step live in ONE set of registers (24 VGPRs; a second set does not fit
  // beside a 96-register accumulator and a held operand): each is re-read for the next k-step right behind its last MFMA of
  // this one, five MFMAs and the next fragment's bookkeeping ahead of its first use.  (The re-read of the last k-step falls
  // behind the panel's end: valid LDS, never used.)
  // INIT: the accumulators START as bi[j] (the unit's bias, one f32x4 per 16-feature block, the same for every token block of
  // both sequences): the first k-step's MFMAs take it as their C operand, so nothing copies it into the 96 registers first.
  auto gemm_frags = [&](auto& acc, auto nj_tag, auto std_tag, auto init_tag, const f32x4* const bi) {
    constexpr int NJ = decltype(nj_tag)::value;
    constexpr bool STD = decltype(std_tag)::value;
    constexpr bool INIT = decltype(init_tag)::value;
    static_assert(NJ % 2 == 0, "fragments per step alternate between two registers");
    LANE_LOCAL();
    TSTART();
    const unsigned char* pl = smem + lane * 16;
    const unsigned char* rl = ring + lane * 16;
    bf16x8 w[2], pf[6];
    ring_wait();
    w[0] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
#pragma unroll
    for (int b = 0; b < 6; ++b) pf[b] = *reinterpret_cast<const bf16x8*>(pl + (b / 3) * PANEL + (((b % 3) * 16) << 10));
    auto kstep = [&](const int s, auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        // w[j & 1] is in registers: the oldest LDS read outstanding (behind it at most the six panel re-reads)
        if (j == 0) __builtin_amdgcn_s_waitcnt(0xc67f); else __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("" ::: "memory");
        issue(head);                                              // refill the slot it came from
        head = head + 1 == RD ? 0 : head + 1;
        if (j < NJ - 1 || s != 15) {                              // (not behind the unit's last fragment)
          ring_wait();
          w[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          f32x4& c = acc[b / 3][j][b % 3];
          const f32x4 cin = (INIT && FIRST) ? bi[j] : c;
          c = STD ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b], w[j & 1], cin, 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j & 1], pf[b], cin, 0, 0, 0);
          if (j == NJ - 1) {
            pf[b] = *reinterpret_cast<const bf16x8*>(pl + (b / 3) * PANEL + (((b % 3) * 16 + s + 1) << 10));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (INIT) {
      kstep(0, std::true_type());
#pragma unroll 1
      for (int s = 1; s < 16; ++s) kstep(s, std::false_type());
    } else {
#pragma unroll 1
      for (int s = 0; s < 16; ++s) kstep(s, std::false_type());
    }
    TSTOP(0);
    TLOG();
  };
  // ---- the same unit GEMM with the weight stream's fragments 6 ... of the unit loaded STRAIGHT INTO REGISTERS (eight of them in
  // rotation, six fragments in flight as before): only the unit's first six fragments -- issued before the unit starts, across
  // its epilogue, when the registers are busy -- come through the LDS ring; the last six iterations refill the ring's slots for
  // whatever the stream holds next (parameter fragment, next unit).  One in-order pipeline, the destination depends on the
  // fragment's position only; `head` leaves as it came.  Why: the LDS array carries 2 x 512 KiB less per unit and CU (DMA writes
  // + weight reads), and the weight operand no longer waits for an LDS read (profiles/dbg/seq_pair_probe.hip PF = 3: 5.7 against
  // 7.3 us per unit on 64 CUs, 7.3 against 9.0 with the chip full).  32 more VGPRs: for the call sites that have them.
  auto issue_reg = [&](u32x4& dst) {
    dst = __builtin_amdgcn_raw_buffer_load_b128(cur_rsrc, lane16, ir << 10, 0);
    if (__builtin_expect(++ir == cur_cnt, 0)) {
      ir = 0;
      ++ie;
      load_seg();
    }
  };
  auto gemm_frags_reg = [&](auto& acc, auto nj_tag, auto std_tag, auto init_tag, const f32x4* const bi) {
    constexpr int NJ = decltype(nj_tag)::value;
    constexpr bool STD = decltype(std_tag)::value;
    constexpr bool INIT = decltype(init_tag)::value;
    constexpr int NF = 16 * NJ;                    // fragments of the unit (64 or 32)
    static_assert(NF % 8 == 0 && 8 % NJ == 0 && RD == 6, "groups of eight fragments, six in flight");
    // (round 6 measured seven in flight -- the quad of fragment f - 1 refilled at once -- and every unit streaming the same
    //  512 KiB, i.e. always from L2: 1 446 / 1 439 against 1 443 us per launch.  The stream's latency is not what the loop waits for.)
    LANE_LOCAL();
    TSTART();
    const unsigned char* pl = smem + lane * 16;
    const unsigned char* rl = ring + lane * 16;
    bf16x8 pf[6];
    u32x4 wr[8];
    int hs = head;                                  // ring slot of the next LDS-resident fragment (first group) / to refill (last group)
    ring_wait();
    wr[0] = *reinterpret_cast<const u32x4*>(rl + hs * 1024);
    hs = hs + 1 == RD ? 0 : hs + 1;
#pragma unroll
    for (int b = 0; b < 6; ++b) pf[b] = *reinterpret_cast<const bf16x8*>(pl + (b / 3) * PANEL + (((b % 3) * 16) << 10));
    auto group = [&](const int s0, auto first_tag, auto last_tag) {      // fragments [NJ s0, NJ s0 + 8)
      constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int j = f % NJ, s = s0 + f / NJ;
        if (FIRST && f + 1 < 6) {      // the next fragment sits in the ring: landed when at most 4 younger loads are outstanding
          wait_vmcnt<4>();
          wr[f + 1] = *reinterpret_cast<const u32x4*>(rl + hs * 1024);
          hs = hs + 1 == RD ? 0 : hs + 1;
        }
        const bf16x8 wv = __builtin_bit_cast(bf16x8, wr[f]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          f32x4& c = acc[b / 3][j][b % 3];
          const f32x4 cin = (INIT && FIRST && f < NJ) ? bi[j] : c;      // (the unit's first k-step: fragments 0 ... NJ - 1)
          c = STD ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b], wv, cin, 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, pf[b], cin, 0, 0, 0);
          if (j == NJ - 1) {
            pf[b] = *reinterpret_cast<const bf16x8*>(pl + (b / 3) * PANEL + (((b % 3) * 16 + s + 1) << 10));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (LAST && f >= 2) {          // the stream's next six items go to the ring (slots in the order they were read from)
          issue(hs);
          hs = hs + 1 == RD ? 0 : hs + 1;
        } else {
          issue_reg(wr[(f + 6) & 7]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    group(0, std::true_type(), std::false_type());
#pragma unroll 1
    for (int s0 = 8 / NJ; s0 < 16 - 8 / NJ; s0 += 8 / NJ) group(s0, std::false_type(), std::false_type());
    group(16 - 8 / NJ, std::false_type(), std::true_type());
    TSTOP(0);
    TLOG();
  };
  auto drained = [&]() {        // every vector-memory operation of the wave has completed (see wait_vm_all)
    wait_vm_all();
  };
  // call sites of the unit GEMM (SITE<n>): bit n of RG2_REG_SITES = the site's weights go straight into registers (gemm_frags_reg)
#ifndef RG2_REG_SITES
#define RG2_REG_SITES 0xE7F      // (round 6: site 2 -- the Q unit beside the held A operands -- fits too since the epilogues shrank; sites 7 and 8 -- a GELU half held -- still spill 71 / 47 registers)
#endif
#define SITE(n) std::integral_constant<int, n>()
  // (Round 6 tried the same loop k-step-major with the panel fragment outermost -- four weight quads per k-step, every panel
  //  fragment re-read 20 MFMAs ahead of its next use instead of 5, no read bursts: 1 408 against 1 364 us per launch, the
  //  weights' lead shrinks from 1.5 k-steps to 1; NOTEBOOK 11.)
  auto gemm_unit = [&](Acc2& acc, auto site) {
    if constexpr (((RG2_REG_SITES) >> decltype(site)::value) & 1) gemm_frags_reg(acc, std::integral_constant<int, 4>(), std::false_type(), std::false_type(), nullptr);
    else gemm_frags(acc, std::integral_constant<int, 4>(), std::false_type(), std::false_type(), nullptr);
  };
  auto gemm_unit_init = [&](Acc2& acc, auto site, const f32x4 (&bi)[4]) {      // acc = bi + W x panel
    if constexpr (((RG2_REG_SITES) >> decltype(site)::value) & 1) gemm_frags_reg(acc, std::integral_constant<int, 4>(), std::false_type(), std::true_type(), bi);
    else gemm_frags(acc, std::integral_constant<int, 4>(), std::false_type(), std::true_type(), bi);
  };
  auto gemm_head_std = [&](f32x4 (&acc)[2][2][3], const f32x4 (&bi)[2]) {       // acc = bi + panel x W (one head, standard layout)
    if constexpr (((RG2_REG_SITES) >> 1) & 1) gemm_frags_reg(acc, std::integral_constant<int, 2>(), std::true_type(), std::true_type(), bi);
    else gemm_frags(acc, std::integral_constant<int, 2>(), std::true_type(), std::true_type(), bi);
  };

  // parameter fragment [4][64] fp32 at the head of every unit: vector p for this wave's 64 features
  auto par_t = [&](const unsigned char* slot, int p, int j, int g4) -> f32x4 {   // T layout: features 16 j + 4 g4 + r
    return *reinterpret_cast<const f32x4*>(slot + (p * 64 + 16 * j + 4 * g4) * 4);
  };
  auto add_bias_t = [&](Acc2& acc, const unsigned char* slot) {
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b = par_t(slot, 0, j, g4);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) acc[q][j][tb] += b;
    }
  };
  // plain unit: acc += bias, then acc += W x panel
  auto unit = [&](Acc2& acc, auto site) {
    const unsigned char* ps = consume();
    add_bias_t(acc, ps);
    release();
    gemm_unit(acc, site);
  };
  // ... acc = bias + W x panel: the bias is the C operand of the first k-step's MFMAs (no zero-fill + add, no copies)
  auto unit_init = [&](Acc2& acc, auto site) {
    f32x4 bi[4];
    {
      LANE_LOCAL();
      const unsigned char* ps = consume();
#pragma unroll
      for (int j = 0; j < 4; ++j) bi[j] = par_t(ps, 0, j, g4);
      release();
    }
    gemm_unit_init(acc, site, bi);
  };
  // ... acc += W x panel; the unit's parameter fragment (a zero bias: the second half of FFN linear2) is only taken off the ring
  auto unit_more = [&](Acc2& acc, auto site) {
    (void)consume();
    release();
    gemm_unit(acc, site);
  };

  // ---- LayerNorm statistics of the three tokens a lane holds, both sequences: per-wave (sum, sum of squares) in one pass over
  // the registers, added up across the waves behind ONE barrier for both sequences (formulas: rg_seq.hip row_stats, bit for bit).
  // The partials alternate between two halves of sStat, call by call: a wave that is still reading one call's partials is
  // never overtaken by another wave's writes of the next call (between two calls of the same parity lies the other call's barrier).
  int stat_par = 0;
  auto row_stats = [&](const Acc2& v, float (&mean)[2][3], float (&rstd)[2][3]) {
    LANE_LOCAL();
    TSTART();
    float* const sSt = sStat + stat_par * STAT_HALF;
    stat_par ^= 1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        float s, ss;
        rg_sum_sq16(v[q][0][tb], v[q][1][tb], v[q][2][tb], v[q][3][tb], s, ss);
        s = xsum4(s);
        ss = xsum4(ss);
        if (g4 == 0) *reinterpret_cast<float2*>(sSt + ((q * NW + wave) * TP + 16 * tb + l15) * 2) = make_float2(s, ss);
      }
    bar();   // (inside the row-statistics stamp)
    int so = l15;               // (row of the partials; see below)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // the partials of ONE sequence in flight at a time (48 registers, not 96: the compiler would hoist and pair all 48 reads):
      // the second sequence's addresses formally depend on the first one's result
      if (q == 1) asm volatile("" : "+v"(so) : "v"(rstd[0][2]));
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        float2 p[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) p[w] = *reinterpret_cast<const float2*>(sSt + ((q * NW + w) * TP + 16 * tb + so) * 2);
        float tot = 0.f, tot2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          tot += p[w].x;
          tot2 += p[w].y;
        }
        const float mu = tot * (1.0f / DM);
        mean[q][tb] = mu;
        rstd[q][tb] = rsqrtf(fmaxf(fmaf(-mu, mu, tot2 * (1.0f / DM)), 0.f) + 1e-5f);
      }
    }
    TSTOP(1);
  };
  // ---- packed T-layout values -> panel fragments (8-byte stores): features 64 wave + 16 j + 4 g4 + [0, 4) of token 16 tb + l15
  auto panel_off = [&](int l15, int g4, int j, int tb) -> int {
    const int s = 2 * wave + (j >> 1), gq = 2 * (j & 1) + (g4 >> 1);
    return ((tb * 16 + s) << 10) + ((l15 + 16 * gq) << 4) + 8 * (g4 & 1);
  };
  auto store_held = [&](const Held& hd) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) *reinterpret_cast<u32x2*>(smem + q * PANEL + panel_off(l15, g4, j, tb)) = hd[q][j][tb];
    TSTOP(8);
  };
  auto write_raw = [&](const Acc2& v) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          *reinterpret_cast<u32x2*>(smem + q * PANEL + panel_off(l15, g4, j, tb)) =
              u32x2{pack2(v[q][j][tb][0], v[q][j][tb][1]), pack2(v[q][j][tb][2], v[q][j][tb][3])};
    TSTOP(8);
  };
  // panel = (v - mean) rstd
  auto write_norm = [&](const Acc2& v, const float (&mean)[2][3], const float (&rstd)[2][3]) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          const float r = rstd[q][tb], nm = -mean[q][tb] * r;      // (v - mean) rstd as ONE fused multiply-add per value
          *reinterpret_cast<u32x2*>(smem + q * PANEL + panel_off(l15, g4, j, tb)) = rg_norm4_bf16(v[q][j][tb], r, nm);
        }
    TSTOP(8);
  };
  // the wave's own fragments of both panels back from gbuf (16 bytes per lane and fragment)
  typedef u32x4 PanelRegs[2][3][2];
  auto restore_issue = [&](PanelRegs& t, const int slot) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          t[q][tb][k] = *reinterpret_cast<const u32x4*>(Gw + (2 * slot + q) * PANEL + ((tb * 16 + 2 * wave + k) << 10) + lane * 16);
    TSTOP(6);
  };
  auto restore_finish = [&](const PanelRegs& t) {
    LANE_LOCAL();
    TSTART();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          *reinterpret_cast<u32x4*>(smem + q * PANEL + ((tb * 16 + 2 * wave + k) << 10) + lane * 16) = t[q][tb][k];
    TSTOP(6);
  };
  // StylizationBlock front half: SiLU(LN(y) * (1 + scale) + shift) with gain = gamma (1 + scale), off = beta (1 + scale)
  // + shift = vectors 1, 2 of the consuming unit's parameter fragment `ps`, as packed bf16
  // ... to a slot of gbuf (the wave's own fragments of the panel image; slot 0 is unused since round 6)
  auto styl_gbuf = [&](const int slot, const Acc2& v, const float (&mean)[2][3], const float (&rstd)[2][3], const unsigned char* ps) {
    LANE_LOCAL();
    const float nmr[2][3] = {{-mean[0][0] * rstd[0][0], -mean[0][1] * rstd[0][1], -mean[0][2] * rstd[0][2]},
                             {-mean[1][0] * rstd[1][0], -mean[1][1] * rstd[1][1], -mean[1][2] * rstd[1][2]}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 gain = par_t(ps, 1, j, g4), off = par_t(ps, 2, j, g4);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          *reinterpret_cast<u32x2*>(Gw + (2 * slot + q) * PANEL + panel_off(l15, g4, j, tb)) = rg_styl4_bf16(v[q][j][tb], rstd[q][tb], nmr[q][tb], gain, off);
        }
    }
  };
  // ... or straight into the panels
  auto write_styl = [&](const Acc2& v, const float (&mean)[2][3], const float (&rstd)[2][3], const unsigned char* ps) {
    LANE_LOCAL();
    const float nmr[2][3] = {{-mean[0][0] * rstd[0][0], -mean[0][1] * rstd[0][1], -mean[0][2] * rstd[0][2]},
                             {-mean[1][0] * rstd[1][0], -mean[1][1] * rstd[1][1], -mean[1][2] * rstd[1][2]}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 gain = par_t(ps, 1, j, g4), off = par_t(ps, 2, j, g4);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          *reinterpret_cast<u32x2*>(smem + q * PANEL + panel_off(l15, g4, j, tb)) = rg_styl4_bf16(v[q][j][tb], rstd[q][tb], nmr[q][tb], gain, off);
        }
    }
  };
  // output projection of a block: acc = the residual stream (back from xbuf) + bias + W x stylize(y)
  auto styl_unit = [&](Acc2& acc, const Acc2& y, auto site) {
    float m3[2][3], r3[2][3];
    row_stats(y, m3, r3);            // (its barrier: every wave is done with the panels' previous content)
    load_R(acc);                     // the residual stream, landing while y is stylized
    {
      TSTART();
      const unsigned char* ps = consume();
      write_styl(y, m3, r3, ps);
      add_bias_t(acc, ps);
      release();
      TSTOP(3);
    }
    barx();
    gemm_unit(acc, site);
  };
  // softmax over the 32 features of each of the wave's two heads, T layout (features: 8 in the lane x 4 lane groups)
  auto softmax_q = [&](Acc& q) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        rg_softmax32(q[2 * h][tb], q[2 * h + 1][tb]);
      }
  };
  // y = softmax(q) A for one head, IN PLACE: blocks 2 h, 2 h + 1 of q become those of y (the contraction runs over the head's
  // 32 features = exactly the two blocks that are overwritten)
  auto qa_head = [&](Acc& q, int h, const bf16x8 (&ah)[2]) {
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const float b8[8] = {q[2 * h][tb][0], q[2 * h][tb][1], q[2 * h][tb][2], q[2 * h][tb][3],
                           q[2 * h + 1][tb][0], q[2 * h + 1][tb][1], q[2 * h + 1][tb][2], q[2 * h + 1][tb][3]};
      const bf16x8 bh = pack8(b8);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) q[2 * h + jb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[jb], bh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
  };
  auto dump = [&](const Acc2& v) {      // diagnostics: T-layout registers -> a.dump [R][TP][512]
    LANE_LOCAL();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          *reinterpret_cast<f32x4*>(a.dump + ((size_t)seqs[q] * TP + 16 * tb + l15) * DM + 64 * wave + 16 * j + 4 * g4) = v[q][j][tb];
    wait_vmcnt<0>();
  };

  // =========================================================== embedding: x = joint_embed(x_in) + tables
  unit(X, SITE(0));
  if (a.dump_stage == 1) dump(X);

#pragma unroll 1
  for (int layer = 0; layer < L; ++layer) {
    const bool dl = a.dump && layer == a.dump_layer;
    float mean[2][3], rstd[2][3];
    // ======================================================= self attention (efficient_attention.py:23-45)
    store_R(X);                                       // x comes back as the accumulator of the output projection
    row_stats(X, mean, rstd);
    write_norm(X, mean, rstd);     // panels = xhat; gamma is folded into the weights, beta into the bias
    barx();
    {
      bf16x8 Af[2][2][2];                             // [sequence][head][16-column block]: A_h = softmax_N(K_h)^T V_h as A operands
      {
        float bk[4], bv[4];                           // biases of the K / V double unit; standard layout: feature 16 j + l15
        unsigned tokbits[2] = {tokbits0[0], tokbits0[1]};
        asm volatile("" : "+v"(tokbits[0]), "+v"(tokbits[1]));
        {
          LANE_LOCAL();
          const unsigned char* ps = consume();
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bk[j] = *reinterpret_cast<const float*>(ps + (16 * j + l15) * 4);
            bv[j] = *reinterpret_cast<const float*>(ps + (64 + 16 * j + l15) * 4);
          }
          release();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 kk[2][2][3], vv[2][2][3];
          const f32x4 bki[2] = {f32x4{bk[2 * h], bk[2 * h], bk[2 * h], bk[2 * h]}, f32x4{bk[2 * h + 1], bk[2 * h + 1], bk[2 * h + 1], bk[2 * h + 1]}};
          const f32x4 bvi[2] = {f32x4{bv[2 * h], bv[2 * h], bv[2 * h], bv[2 * h]}, f32x4{bv[2 * h + 1], bv[2 * h + 1], bv[2 * h + 1], bv[2 * h + 1]}};
          gemm_head_std(kk, bki);
          {
            TSTART();
            // softmax over the tokens, per feature column (lane): tokens 16 tb + 4 g4 + r; masked / padded tokens weigh 0
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                float mx = -INFINITY;
#pragma unroll
                for (int tb = 0; tb < 3; ++tb)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if ((tokbits[q] >> (4 * tb + r)) & 1u) mx = fmaxf(mx, kk[q][j][tb][r]);
                mx = xmax4(mx);
                const float nm2 = mx * -1.44269504088896340736f;
                float sum = 0.f;
#pragma unroll
                for (int tb = 0; tb < 3; ++tb)
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const float e = ((tokbits[q] >> (4 * tb + r)) & 1u) ? rg_exp_sub(kk[q][j][tb][r], nm2) : 0.f;
                    kk[q][j][tb][r] = e;
                    sum += e;
                  }
                sum = xsum4(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int tb = 0; tb < 3; ++tb) kk[q][j][tb] *= inv;
              }
            TSTOP(4);
          }
          gemm_head_std(vv, bvi);
          // A_h[i][jc] = sum_t P[t][i] V[t][jc] (contraction over tokens: step 0 = token blocks 0 | 1, step 1 = block 2 | zeros)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            bf16x8 v0f[2], v1f[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
              const f32x4* vb = vv[q][jb];
              const float v0[8] = {vb[0][0], vb[0][1], vb[0][2], vb[0][3], vb[1][0], vb[1][1], vb[1][2], vb[1][3]};
              const float v1[8] = {vb[2][0], vb[2][1], vb[2][2], vb[2][3], 0.f, 0.f, 0.f, 0.f};
              v0f[jb] = pack8(v0);
              v1f[jb] = pack8(v1);
            }
            f32x4 Ab[2][2];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
              const f32x4* kb = kk[q][ib];
              const float k0[8] = {kb[0][0], kb[0][1], kb[0][2], kb[0][3], kb[1][0], kb[1][1], kb[1][2], kb[1][3]};
              const float k1[8] = {kb[2][0], kb[2][1], kb[2][2], kb[2][3], 0.f, 0.f, 0.f, 0.f};
              const bf16x8 k0f = pack8(k0), k1f = pack8(k1);
#pragma unroll
              for (int jb = 0; jb < 2; ++jb) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, v0f[jb], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                Ab[ib][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, v1f[jb], d, 0, 0, 0);
              }
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
              const float a8[8] = {Ab[0][jb][0], Ab[0][jb][1], Ab[0][jb][2], Ab[0][jb][3], Ab[1][jb][0], Ab[1][jb][1], Ab[1][jb][2], Ab[1][jb][3]};
              Af[q][h][jb] = pack8(a8);
            }
          }
        }
      }
      Acc2 yy;                                        // queries, then (in place) the attention output
      unit_init(yy, SITE(2));
      {
        TSTART();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          softmax_q(yy[q]);
#pragma unroll
          for (int h = 0; h < 2; ++h) qa_head(yy[q], h, Af[q][h]);
        }
        TSTOP(4);
      }
      if (dl && a.dump_stage == 10) dump(yy);
      styl_unit(X, yy, SITE(3));                               // x += proj_out(...)  (stylization_block.py:40, efficient_attention.py:44)
    }
    if (dl && a.dump_stage == 2) dump(X);

    // ======================================================= three cross attentions + ca_mix (efficient_attention.py:62-102,
    // diffusion_transformer.py:110-122), as [h_text | h_audio | h_spk | x] @ W_fused^T (rg_gesture.h: ca_mix fusion)
    row_stats(X, mean, rstd);
    write_norm(X, mean, rstd);     // xhat: the queries' operand (their gamma / beta folded) and the x segment
    barx();
    // x W_x^T + b = sd * (xhat W_x^T + rstd * (mean * rowsum(W_x) + b)),  sd = 1 / rstd; X becomes the block's accumulator
    auto mix_x = [&]() {
      {
        LANE_LOCAL();
        TSTART();
        const unsigned char* ps = consume();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b = par_t(ps, 0, j, g4), c1 = par_t(ps, 1, j, g4);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) X[q][j][tb] = (c1 * mean[q][tb] + b) * rstd[q][tb];
        }
        release();
        TSTOP(9);
      }
      gemm_unit(X, SITE(4));
      TSTART();
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          const float sd = __builtin_amdgcn_rcpf(rstd[q][tb]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            X[q][j][tb] *= sd;
            // (rounded here in both forward kernels: no contraction with the additions of the classifier-free tables behind it)
            asm volatile("" : "+v"(X[q][j][tb]));
          }
        }
      TSTOP(9);
    };
    if (!cond) {
      mix_x();
      // classifier-free rows: + sum_c W_c h_c with h_c one of two tabulated rows per (step, layer, condition)
      LANE_LOCAL();
      TSTART();
      unsigned qbits[2] = {qbits0[0], qbits0[1]};
      asm volatile("" : "+v"(qbits[0]), "+v"(qbits[1]));
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const unsigned char* us = consume();   // fragment 0: (c, flag) = (0,0) (0,1) (1,0) (1,1); fragment 1: (2,0) (2,1)
#pragma unroll
        for (int c2 = 0; c2 < (f == 0 ? 2 : 1); ++c2) {
          const int c = 2 * f + c2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(us + ((c2 * 2 + 0) * 64 + 16 * j + 4 * g4) * 4);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(us + ((c2 * 2 + 1) * 64 + 16 * j + 4 * g4) * 4);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int tb = 0; tb < 3; ++tb) X[q][j][tb] += ((qbits[q] >> (3 * c + tb)) & 1u) ? u1 : u0;
          }
        }
        release();
      }
      TSTOP(9);
    } else {
      // stylized cross-attention rows of condition c -> gbuf slot 1 + c: query projection (operand: xhat in the panels),
      // softmax, y = q A_clip, masked rows, LayerNorm + stylization with the parameters of MIX_c
      auto cross = [&](const int c) {
        Acc2 yy;
        unit_init(yy, SITE(5));
        LANE_LOCAL();
        unsigned qbits[2] = {qbits0[0], qbits0[1]};
        asm volatile("" : "+v"(qbits[0]), "+v"(qbits[1]));
        {
          TSTART();
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            softmax_q(yy[q]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              bf16x8 ah[2];
#pragma unroll
              for (int jb = 0; jb < 2; ++jb) {
                const unsigned char* s0 = consume();
                ah[jb] = *reinterpret_cast<const bf16x8*>(s0 + lane * 16);
                release();
              }
              qa_head(yy[q], h, ah);
            }
            // masked queries: the reference adds -1e6 before the LayerNorm; keep its fp32 rounding (DESIGN: masked query rows)
#pragma unroll
            for (int tb = 0; tb < 3; ++tb)
              if ((qbits[q] >> (3 * c + tb)) & 1u) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    float z = yy[q][j][tb][r] + (-1000000.0f);
                    asm volatile("" : "+v"(z));
                    yy[q][j][tb][r] = z + 1000000.0f;
                  }
              }
          }
          TSTOP(4);
        }
        if (dl && a.dump_stage == 11 + c) dump(yy);
        float m3[2][3], r3[2][3];
        row_stats(yy, m3, r3);
        TSTART();
        const unsigned char* ps = consume();
        styl_gbuf(1 + c, yy, m3, r3, ps);
        release();
        TSTOP(3);
      };
      // All four units that read xhat first (the residual stream is dead between the block's LayerNorm and its accumulator:
      // nothing of X is live beside the queries), then the three MIX units.  The stylized rows of the three conditions go to
      // gbuf slots 1-3 and come back into the panels when their MIX unit runs (held in registers beside the queries'
      // accumulator they cost ~120 spilled registers per condition); each image is requested BEFORE the barrier that frees the
      // panels, so that its L2 latency runs while the wave waits for the others.  Accumulation order into X as in rg_seq.hip.
      PanelRegs t;
      cross(0);
      cross(1);
      cross(2);
      mix_x();
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        restore_issue(t, 1 + c);
        barx();                                 // every wave is done reading the panels (xhat; h of condition c - 1)
        restore_finish(t);
        drained();
        barx();
        gemm_unit(X, SITE(6));                  // += W_c h_c  (text, audio, speaker)
      }
    }
    if (dl && a.dump_stage == 3) dump(X);

    // ======================================================= FFN (diffusion_transformer.py:74-87): 1024 hidden units in two halves
    store_R(X);
    barx();                                     // every wave is done reading the panels
    write_raw(X);
    barx();
    {
      Acc2 yf;
      {
        Held g0, g1;
        auto ff1 = [&](Held& gd, auto site) {      // one half of linear1 + GELU, kept as packed bf16
          Acc2 gg;
          unit_init(gg, site);
          TSTART();
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int tb = 0; tb < 3; ++tb)
                gd[q][j][tb] = rg_gelu4_bf16(gg[q][j][tb]);
          TSTOP(3);
        };
        ff1(g0, SITE(11));                      // (nothing held yet: the register form of the unit fits)
        ff1(g1, SITE(7));
        barx();                                 // every wave is done reading x
        store_held(g0);
        barx();
        unit_init(yf, SITE(8));                          // (the bias of linear2 rides with the first half)
        barx();
        store_held(g1);
        barx();
        unit_more(yf, SITE(9));
      }
      styl_unit(X, yf, SITE(3));
    }
    if (dl && a.dump_stage == 4) dump(X);
  }

  // =========================================================== output head (diffusion_transformer.py:662-666)
  LANE_LOCAL();
  barx();
  write_raw(X);
  barx();
  Acc2 out;
  unit_init(out, SITE(10));
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q == 1 && sB == sA) break;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const int t = 16 * tb + l15;
      if (t < T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const size_t o = ((size_t)seqs[q] * T + t) * DM + 64 * wave + 16 * j + 4 * g4;
          rg_tail::store_head(a.head, (unsigned)(o * 4), out[q][j][tb]);     // (written through: with the tail on, another workgroup of this launch reads it)
        }
      }
    }
  }
  wait_vmcnt<0>();
#ifdef RG_STAMPS
  if (a.dump_stage == 99 && lane0 == 0) {
    tacc[5] = __builtin_amdgcn_s_memrealtime() - tk0;
    for (int i = 0; i < 12; ++i) a.dump[(blockIdx.x * 8 + wave) * 12 + i] = (float)tacc[i];
  }
#endif
#undef LANE_LOCAL
}

// Pairs of a launch: the clips [0, split) and [split, B) run at different diffusion steps (sampler.cobatched_loop), so pairs
// are formed inside each group; a group with an odd count ends in a lone sequence (paired with itself).  Pair p of the
// conditional half: clips (c, c') below; its classifier-free twin pair: the same clips + B.
__device__ __forceinline__ void pair_clips(const int p, const int split, const int B, int& c0, int& c1) {
  const int n0 = (split + 1) >> 1;
  if (p < n0) { c0 = 2 * p; c1 = min(2 * p + 1, split - 1); }
  else { c0 = split + 2 * (p - n0); c1 = min(c0 + 1, B - 1); }
}

// workgroups of a launch: pairs per kind x (pairs ? 1 : 2)
__device__ __host__ __forceinline__ int seq2_grid(const int B, const int split_in, const int pairs) {
  const int split = split_in < 0 ? 0 : (split_in > B ? B : split_in);
  const int npc = ((split + 1) >> 1) + ((B - split + 1) >> 1);
  return pairs ? npc : 2 * npc;
}
// The work of workgroup `block` of such a launch.
__device__ __forceinline__ void seq2_block(const rg_seq_args& a, const int block, unsigned char* const smem) {
  const int B = a.B, split = min(max(a.split, 0), B);
  const int npc = ((split + 1) >> 1) + ((B - split + 1) >> 1);     // pairs per kind
  // pairs == 0: one workgroup per pair (2 npc workgroups); dealt round-robin over the 8 XCDs, the conditional pairs to four of
  // them and the classifier-free pairs (which skip a third of the weight stream) to the other four, so that the workgroups
  // sharing an L2 walk the stream together (speed only).  pairs == 1: npc workgroups, each runs a conditional pair and then
  // the classifier-free pair of the same clips.
  int p = block, kind0 = 0;
  if (!a.pairs) {
    if ((npc & 3) == 0) {
      const int x = block & 7, q = block >> 3;
      kind0 = x >= 4;
      p = 4 * q + (x & 3);
    } else {
      kind0 = block >= npc;
      p = block - kind0 * npc;
    }
  }
  int c0, c1;
  pair_clips(p, split, B, c0, c1);
  const int npass = a.pairs ? 2 : 1;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    const int kind = a.pairs ? pass : kind0;
    run_pair(a, c0 + kind * B, c1 + kind * B, smem);
    __syncthreads();     // descriptors, panels and statistics of the pass are dead in every wave
    int* const ctr = rg_tail::late_ctr();
    if (ctr && !a.dump_stage)              // the loop step's update of these clips, where their other sequence is done (rg_tail.h)
      rg_tail::arrive_and_glue<NTH>(ctr, c0, c1, reinterpret_cast<int*>(smem), __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  }
}

#ifndef RG_SEQ_BODY_ONLY      // (rg_seqx.hip includes this file for run_pair / seq2_block only)
__global__ void __launch_bounds__(NTH) rg_seq2_kernel(const rg_seq_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifndef RG2_SHARE_SIMD     // (experiment switch)
  RG_OWN_THE_SIMD();
#endif
  seq2_block(a, blockIdx.x, smem);
}

extern "C" int rg_seq2_forward(rg_handle* h, const rg_seq_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  const rg_seq_args& a = *args_host;
  RG_REQUIRE(h, a.wstream && a.pstream && a.ustream && a.afrag && a.x && a.tbias && a.src_mask && a.qmask && a.head, "null pointer");
  RG_REQUIRE(h, a.xbuf && a.gbuf, "the two-sequence forward needs its scratch buffers xbuf and gbuf");
  RG_REQUIRE(h, a.L >= 1 && a.L <= 8 && a.B >= 1 && a.T >= 1 && a.T <= TP, "unsupported shape (T <= 48, L <= 8)");
  RG_REQUIRE(h, a.step >= 0 && a.step < a.S && a.step_b >= 0 && a.step_b < a.S, "step out of range");
  RG_REQUIRE(h, a.dump_stage == 0 || a.dump, "dump_stage needs a dump buffer");
  RG_REQUIRE(h, a.pairs == 0 || a.pairs == 1, "pairs must be 0 or 1");
  RG_REQUIRE(h, rg_tail::args_ok(a), "glue_ctr: glue must cover the B clips (n_a + n_b == B, T, D = 512), its pointers set, no dump");
  static rg_attr_once lds_once;
  if (!rg_reserve_lds(lds_once, rg_seq2_kernel, LDS_BYTES)) {
    h->err = "rg_seq2_forward: cannot reserve LDS";
    return RG_ERR_HIP;
  }
  rg_prof_rec rec;
  if (h->profiling) {   // bench.py roofline: HIP events around the launch (variant 3), algorithmic FLOPs of the T token rows
    auto get_ev = [&]() {
      hipEvent_t e;
      if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else { (void)hipEventCreate(&e); }
      return e;
    };
    rec.start = get_ev(); rec.stop = get_ev();
    rec.variant = 3;
    const double unit = 2.0 * a.T * DM * DM, att = 2.0 * a.T * 32 * 32 * 16;      // one 512 x 512 GEMM; one q A (or K^T V) over 16 heads
    const double cond = (UPL * a.L + 2) * unit + a.L * (2 + 3) * att, unc = (10 * a.L + 2) * unit + a.L * 2 * att;
    rec.flops = a.B * (cond + unc);
    (void)hipEventRecord(rec.start, rg_stream(stream));
  }
  hipLaunchKernelGGL(rg_seq2_kernel, dim3(seq2_grid(a.B, a.split, a.pairs)), dim3(NTH), LDS_BYTES, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  if (h->profiling) {
    (void)hipEventRecord(rec.stop, rg_stream(stream));
    h->prof.push_back(rec);
  }
  return RG_OK;
}
#endif  // RG_SEQ_BODY_ONLY
