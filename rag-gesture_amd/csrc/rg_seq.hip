// Sequence-stationary denoiser forward (round 3): ONE workgroup owns ONE sequence (T <= 48 token rows of one
// classifier-free branch of one clip) for the WHOLE forward -- embedding, L decoder layers, output head.
//
// Why: the layer chain of per-op launches (rg_gemm + attention + stylization kernels, ~90 launches per forward) moves
// ~330 MB of activations per layer through L2 / HBM at M = 5504 rows and runs at 0.10 of the MFMA peak; every launch pays
// its boundary, its first dependent load and its epilogue.  The sequences of a batch are independent chains (the linear
// attention only mixes the tokens of one sequence, everything else is row-local), so here the activations never leave
// the CU: the fp32 residual stream lives in the accumulator registers (48 per lane), the bf16 MFMA operand panels in
// LDS (2 x 48 KiB), and the only memory traffic is the WEIGHT STREAM -- 16 unit GEMMs of 512 x 512 per layer, packed on
// the host in exactly the order and MFMA-fragment layout a wave consumes them (1-KiB fragments, lane-linear), fetched by
// every wave for itself with LDS-DMA (global_load_lds_dwordx4) into a wave-private ring of 7 slots behind a counted
// vmcnt.  No workgroup barrier, no hand-off and no global round trip sits between a weight fragment and its MFMAs; the
// eight waves only meet when a panel changes hands (~20 s_barrier per layer).
// Bound: per-CU LDS-DMA intake (measured 95 GB/s per CU with 256 workgroups streaming the same 64 MB,
// profiles/dbg/stream_probe.hip): 8 MB of weights per layer and sequence -> ~90-100 us per layer, for up to 256
// sequences at once (one per CU), against 192 us per layer and 128 sequences for the launch chain.
//
// Layouts.  MFMA 16x16x32 bf16: lane L = (l15 = L & 15, g4 = L >> 4) holds A[i = l15][k = 8 g4 + e], B[k = 8 g4 + e][j = l15],
// D[i = 4 g4 + r][j = l15].  A and B fragments have the same shape, so one fragment image serves either role:
//   "T layout"  out[n][t] = sum_k W[n][k] P[t][k]: A = weight fragment, B = panel fragment; the lane holds token t = l15 of
//               a 16-token block and 4 CONSECUTIVE features -> panels are written with 8-byte LDS stores, per-token
//               scalars (mean, rstd, masks) are per-lane, per-feature vectors come as 16-byte broadcast reads;
//   "standard"  out[t][n]: the same two fragments with the roles swapped; used for K and V of the self attention, whose
//               accumulators then ARE the operands of K^T V (contraction over tokens = over registers and lane groups).
// The attention products never touch LDS: A_h = softmax_N(K_h)^T V_h and y = softmax(q) A_h take their operands straight
// from accumulator registers (the contraction index is enumerated identically on both sides).
// reference: mogen/models/transformers/diffusion_transformer.py:105-127 (DecoderLayer), :74-87 (FFN), :620-668 (forward);
// mogen/models/attentions/efficient_attention.py:23-45, 62-102; mogen/models/utils/stylization_block.py:29-40;
// mogen/models/transformers/raggesture.py:1041-1085 (classifier-free row doubling).
#define RG_PACK2_ONE      // (the kernel owns its SIMDs, RG_OWN_THE_SIMD: rg_common.h rg_pack2_bf16)
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
constexpr int TP = 48;         // token rows of the panels (T <= 48; rows >= T repeat token T - 1)
constexpr int NW = 8;          // waves per workgroup; wave w owns features [64 w, 64 w + 64) = heads 2 w, 2 w + 1
constexpr int NTH = NW * 64;
constexpr int RD = 7;          // ring slots (1 KiB) per wave: all of them in flight except the one being read
constexpr int UPL = 16;        // unit GEMMs per layer in the weight stream
constexpr int OFF_P0 = 0;
constexpr int OFF_P1 = TP * 1024;
constexpr int OFF_RING = 2 * TP * 1024;
constexpr int MAX_SEG = 8 * 34 + 5;                      // fetch segments of a conditional sequence at L = 8, + sentinel
constexpr int OFF_DESC = OFF_RING + NW * RD * 1024;      // [MAX_SEG] x 16 B fetch segments {address (wave 0), count, wave stride}
constexpr int OFF_STAT = OFF_DESC + MAX_SEG * 16;         // [NW][TP][2] fp32 partial (sum, sum of squares)
constexpr int LDS_BYTES = OFF_STAT + NW * TP * 2 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// unit slots of a layer in the weight / parameter streams (U_KV is a double unit: the wave's K and V fragments of head
// 2 w, then of head 2 w + 1; its parameter fragment carries both biases)
enum { U_KV = 0, U_KV2, U_Q, U_SAO, U_MIXX, U_Q3_0, U_MIX_0, U_Q3_1, U_MIX_1, U_Q3_2, U_MIX_2, U_FF1_0, U_FF2_0, U_FF1_1, U_FF2_1, U_FFO };
// fetch segments of one layer, in consumption order: unit slot << 2 | kind (0 = parameter fragment, 1 = weights, 2 = extra)
#define SEG(u, k) ((u) << 2 | (k))
__constant__ const unsigned char SEG_COND[34] = {
    SEG(U_KV, 0), SEG(U_KV, 1), SEG(U_Q, 0), SEG(U_Q, 1), SEG(U_SAO, 0), SEG(U_SAO, 1), SEG(U_MIXX, 0), SEG(U_MIXX, 1),
    SEG(U_Q3_0, 0), SEG(U_Q3_0, 1), SEG(U_Q3_0, 2), SEG(U_MIX_0, 0), SEG(U_MIX_0, 1),
    SEG(U_Q3_1, 0), SEG(U_Q3_1, 1), SEG(U_Q3_1, 2), SEG(U_MIX_1, 0), SEG(U_MIX_1, 1),
    SEG(U_Q3_2, 0), SEG(U_Q3_2, 1), SEG(U_Q3_2, 2), SEG(U_MIX_2, 0), SEG(U_MIX_2, 1),
    SEG(U_FF1_0, 0), SEG(U_FF1_0, 1), SEG(U_FF2_0, 0), SEG(U_FF2_0, 1), SEG(U_FF1_1, 0), SEG(U_FF1_1, 1),
    SEG(U_FF2_1, 0), SEG(U_FF2_1, 1), SEG(U_FFO, 0), SEG(U_FFO, 1), 0};
__constant__ const unsigned char SEG_UNC[20] = {
    SEG(U_KV, 0), SEG(U_KV, 1), SEG(U_Q, 0), SEG(U_Q, 1), SEG(U_SAO, 0), SEG(U_SAO, 1), SEG(U_MIXX, 0), SEG(U_MIXX, 1), SEG(U_MIXX, 2),
    SEG(U_FF1_0, 0), SEG(U_FF1_0, 1), SEG(U_FF2_0, 0), SEG(U_FF2_0, 1), SEG(U_FF1_1, 0), SEG(U_FF1_1, 1),
    SEG(U_FF2_1, 0), SEG(U_FF2_1, 1), SEG(U_FFO, 0), SEG(U_FFO, 1), 0};
#undef SEG
constexpr int NSEG_COND = 33, NSEG_UNC = 19;

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return rg_pack2_bf16(lo, hi); }
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
__device__ __forceinline__ float gelu_fast(float v) { return rg_gelu_erf(v); }
// Attention products (softmax_N(K)^T V, softmax(q) A): ATT_HL = true feeds the matrix cores bf16 hi + lo operand pairs
// (hi * hi + hi * lo + lo * hi ~ fp32 products, 3 MFMAs and the residual arithmetic per fragment), false = plain bf16
// operands with fp32 accumulation (what every GEMM around them does; y is rounded to bf16 right after its stylization).
// (Off, and the conditions' A fragments come as plain bf16: measured in round 3, no build has used the pairs since.)
constexpr bool ATT_HL = false;
// 8 fp32 values -> bf16 hi fragment and the bf16 residual fragment
__device__ __forceinline__ void split_hl(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  u32x4 h, l = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned short a = f2bf(v[2 * q]), b = f2bf(v[2 * q + 1]);
    h[q] = (unsigned)a | ((unsigned)b << 16);
    if (ATT_HL) l[q] = pack2(v[2 * q] - bf2f(a), v[2 * q + 1] - bf2f(b));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ f32x4 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
  if (ATT_HL) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
// sum / max over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: v_permlane16_swap exchanges
// the odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the
// lower half of the second; with both operands the same value the two results are the value and its partner's
// (ds_bpermute-based shuffles cost an LDS round trip each: 12 dependent ones per LayerNorm call)
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
// lgkmcnt(0) as the BUILTIN: the compiler's wait-count bookkeeping sees it, so it does not add a wait of its own in front of
// the first use of a register that this wait already covers (after an inline-asm wait it does: a full lgkmcnt(0) right behind
// the next fragment's LDS read, which exposes that read's latency).  The empty asm keeps memory operations from crossing.
__device__ __forceinline__ void wait_lds() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

#ifdef RG_STAMPS
// Diagnostic build only (build.py RG_DIAG=1): wall-clock (100 MHz) time per category, summed per wave, written to
// a.dump[(seq * 8 + wave) * 8 + category] when dump_stage == 99.  Categories: 0 unit GEMMs, 1 row statistics (with their
// barrier), 2 other barriers, 3 parameter fragments + panel writes, 4 attention math, 5 whole kernel.
#define TSTART() const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime()
#define TSTOP(cat) tacc[cat] += __builtin_amdgcn_s_memrealtime() - t0_
#else
#define TSTART()
#define TSTOP(cat)
#endif

typedef f32x4 Acc[4][3];   // [16-feature block of the wave's 64][16-token block]

__device__ __forceinline__ void zero(Acc& a) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) a[j][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace

// One forward of one sequence (a whole workgroup): the body of rg_seq_kernel.
__device__ __forceinline__ void run_sequence(const rg_seq_args& a, const int seq, unsigned char* const smem) {
  unsigned char* const P0 = smem + OFF_P0;
  unsigned char* const P1 = smem + OFF_P1;
  float* const sStat = reinterpret_cast<float*>(smem + OFF_STAT);
  // (an opaque copy: nothing derived from the thread id is an invariant of the caller's pass loop, where it would stay live
  //  -- spilled -- across the whole forward)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane0 = tid & 63;
  // Lane-derived values are re-derived from an opaque copy of the lane id wherever they are used: as loop invariants of
  // the layer loop the address arithmetic of every unrolled LDS access would otherwise be hoisted in front of the loop
  // and live (spilled) across it.
#define LANE_LOCAL()                      \
  int ln_ = lane0;                        \
  asm volatile("" : "+v"(ln_));         \
  const int lane = ln_, l15 = ln_ & 15, g4 = ln_ >> 4; \
  (void)lane; (void)l15; (void)g4
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const ring = smem + OFF_RING + wave * (RD * 1024);
  const int T = a.T, B = a.B, L = a.L, R = 2 * a.B;
  const bool cond = seq < B;
  const int clip = cond ? seq : seq - B;
  const int st = clip >= a.split ? a.step_b : a.step;
#ifdef RG_STAMPS
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tk0 = __builtin_amdgcn_s_memrealtime();
#endif
  auto barx = [&]() {      // workgroup barrier (stamped as category 2 in the diagnostic build)
    TSTART();
    bar();
    TSTOP(2);
  };
  const int NU = UPL * L + 2;
  const int nspl = cond ? NSEG_COND : NSEG_UNC;          // fetch segments per layer
  const int pre = 2;
  const int n_seg = pre + nspl * L + 2;   // embed (P, W), layers, head (P, W)

  // ---- fetch program of this sequence: one {address for wave 0, fragment count, wave stride in fragments} per segment,
  // in consumption order, + a sentinel that keeps the in-flight count invariant behind the end.  Classifier-free
  // sequences skip the three query projections and their ca_mix segments: their stylized cross-attention rows are
  // constants of (step, layer) and enter as a tabulated vector (extra segment of mix_x).
  if (tid <= n_seg) {
    const unsigned char* adr = reinterpret_cast<const unsigned char*>(a.wstream);
    unsigned cnt = 1u << 30, stride = 0;
    if (tid < n_seg) {
      int uid, kind, idx = 0, l = 0;
      if (tid < pre) { uid = 0; kind = tid; }
      else if (tid >= pre + nspl * L) { uid = NU - 1; kind = tid - (pre + nspl * L); }
      else {
        const int q = tid - pre;
        l = q / nspl;
        const unsigned char e = cond ? SEG_COND[q - l * nspl] : SEG_UNC[q - l * nspl];
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
        adr = reinterpret_cast<const unsigned char*>(a.afrag) + ((size_t)((l * 3 + c) * B + clip) * 32 << 10);
        cnt = 4; stride = 4;
      } else {                    // classifier-free cross-attention contribution: ustream [S][L][8][2 KiB]
        adr = reinterpret_cast<const unsigned char*>(a.ustream) + ((size_t)(st * L + l) * 16 << 10);
        cnt = 2; stride = 2;
      }
    }
    const unsigned long long av = reinterpret_cast<unsigned long long>(adr);
    *reinterpret_cast<u32x4*>(smem + OFF_DESC + tid * 16) = u32x4{(unsigned)av, (unsigned)(av >> 32), cnt, stride};
  }

  // ---- token masks, per lane: bit (4 tb + r) of tokbits = token 16 tb + 4 g4 + r takes part in the self attention
  // (standard layout); bit (3 c + tb) of qbits = query token 16 tb + l15 of condition c is masked (T layout)
  // (efficient_attention.py:34, 95-97)
  unsigned tokbits0 = 0, qbits0 = 0;
  Acc xr;
  {
  LANE_LOCAL();
#pragma unroll
  for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 16 * tb + 4 * g4 + r;
      if (t < T && a.src_mask[(size_t)seq * T + t] != 0.f) tokbits0 |= 1u << (4 * tb + r);
    }
    const int tq = min(16 * tb + l15, T - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (a.qmask[((size_t)c * R + seq) * T + tq] == 0.f) qbits0 |= 1u << (3 * c + tb);
  }

  // ---- residual stream, T layout: xr[j][tb][r] = x[token 16 tb + l15][feature 64 wave + 16 j + 4 g4 + r];
  // starts as the positional tables (diffusion_transformer.py:646-659), the embedding GEMM accumulates onto it
#pragma unroll
  for (int tb = 0; tb < 3; ++tb) {
    const int t = min(16 * tb + l15, T - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      xr[j][tb] = *reinterpret_cast<const f32x4*>(a.tbias + (size_t)t * DM + 64 * wave + 16 * j + 4 * g4);
  }
  }
  // ---- panel P0 = bf16(x_in): fragment (tb, s) = tokens [16 tb, +16) x features [32 s, +32): lane (l15, g) holds the 8
  // features [32 s + 8 g, +8) of token 16 tb + l15
#pragma unroll
  for (int it = 0; it < (TP * 64) / NTH; ++it) {
    const int slot = tid + NTH * it, t = slot >> 6, c8 = slot & 63;
    const float* xp = a.x + ((size_t)clip * T + min(t, T - 1)) * DM + c8 * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + 4);
    *reinterpret_cast<u32x4*>(P0 + (((t >> 4) * 16 + (c8 >> 2)) << 10) + (((t & 15) + 16 * (c8 & 3)) << 4)) =
        u32x4{pack2(v0[0], v0[1]), pack2(v0[2], v0[3]), pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
  }
  __syncthreads();     // descriptors + P0 written; every register-destination load above has been waited for

  // ---- the wave's fetch cursor: segment, fragment inside it.  All state is wave-uniform (scalar registers): the source of
  // a fragment is a buffer descriptor of the segment (for this wave) + a scalar offset, the lane only adds its 16 bytes.
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
  int head = 0;                                  // ring slot of the oldest fragment in flight
  // consume(): the oldest fragment has landed; returns its slot.  release(): the slot's bytes are in registers -> refill it.
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


  // ---- unit GEMM: acc += W_unit x panel over K = 512 (16 steps of 32); NJ weight fragments per step (NJ = 4: the wave's 64
  // features, NJ = 2: the 32 features of one head).  STD = false: T layout (A = weights); true: standard layout (A = panel).
  // The weight stream's fragments RD ... of the unit are loaded STRAIGHT INTO REGISTERS (eight in
  // rotation, RD fragments in flight): only the unit's first RD fragments -- issued before the unit starts, across
  // its epilogue -- come through the LDS ring; the last RD iterations refill the ring's slots for whatever the stream holds
  // next.  One in-order pipeline, the destination depends on the fragment's position only; `head` leaves as it came (rg_seq2.hip).
  // (Until late in round 5 every fragment went through the ring: 1 030 instead of 901 us per forward, same bits.)
  auto issue_reg = [&](u32x4& dst) {
    dst = __builtin_amdgcn_raw_buffer_load_b128(cur_rsrc, lane16, ir << 10, 0);
    if (++ir == cur_cnt) {
      ir = 0;
      ++ie;
      load_seg();
    }
  };
  auto gemm_frags_reg = [&](auto& acc, const unsigned char* panel, auto nj_tag, auto std_tag) {
    constexpr int NJ = decltype(nj_tag)::value;
    constexpr bool STD = decltype(std_tag)::value;
    static_assert((16 * NJ) % 8 == 0 && 8 % NJ == 0 && RD <= 8 && RD >= 2, "groups of eight fragments, RD in flight");
    LANE_LOCAL();
    TSTART();
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
    auto group = [&](const int s0, auto first_tag, auto last_tag) {      // fragments [NJ s0, NJ s0 + 8)
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
        if (LAST && f >= 8 - RD) {     // the stream's next RD items go to the ring (slots in the order they were read from)
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
    for (int s0 = 8 / NJ; s0 < 16 - 8 / NJ; s0 += 8 / NJ) group(s0, std::false_type(), std::false_type());
    group(16 - 8 / NJ, std::false_type(), std::true_type());
    TSTOP(0);
  };
  auto gemm_unit = [&](Acc& acc, const unsigned char* panel, auto std_tag) {
    gemm_frags_reg(acc, panel, std::integral_constant<int, 4>(), std_tag);
  };
  // half unit, standard layout: the 32 features of ONE head (32 fragments: per step the head's two 16-feature blocks)
  auto gemm_head_std = [&](f32x4 (&acc)[2][3], const unsigned char* panel) {
    gemm_frags_reg(acc, panel, std::integral_constant<int, 2>(), std::true_type());
  };
  std::false_type TL;

  // parameter fragment [4][64] fp32 at the head of every unit: vector p for this wave's 64 features
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
  // plain unit: acc += bias, then acc += W x panel
  auto unit = [&](Acc& acc, const unsigned char* panel, auto std_tag) {
    const unsigned char* ps = consume();
    add_bias_t(acc, ps);
    release();
    gemm_unit(acc, panel, std_tag);
  };

  // ---- LayerNorm statistics of the three tokens a lane holds (all 512 features: 16 in the lane, x 4 lane groups,
  // x 8 waves): per-wave (sum, sum of squares) in ONE pass over the registers, added up across the waves behind one barrier;
  // variance = E[x^2] - mean^2 in fp32 (relative error ~1e-7 (1 + mean^2 / variance): the rows normalised here -- residual
  // stream, attention and FFN outputs -- have |mean| of the order of their deviation or below).  Until round 5: per-wave M2
  // about the wave's own mean, combined by Chan's formula -- 1.8x the vector instructions, a fifth of a layer's epilogue work.
  // (Between two calls lies a workgroup barrier at every call site: one buffer of partials is enough here; rg_seq2.hip alternates two.)
  auto row_stats = [&](const Acc& v, float (&mean)[3], float (&rstd)[3]) {
    LANE_LOCAL();
    TSTART();
    float* const sSt = sStat;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      float s, ss;
      rg_sum_sq16(v[0][tb], v[1][tb], v[2][tb], v[3][tb], s, ss);
      s = xsum4(s);
      ss = xsum4(ss);
      if (g4 == 0) *reinterpret_cast<float2*>(sSt + (wave * TP + 16 * tb + l15) * 2) = make_float2(s, ss);
    }
    bar();   // (inside the row-statistics stamp)
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      float tot = 0.f, tot2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float2 p = *reinterpret_cast<const float2*>(sSt + (w * TP + 16 * tb + l15) * 2);
        tot += p.x;
        tot2 += p.y;
      }
      const float mu = tot * (1.0f / DM);
      mean[tb] = mu;
      rstd[tb] = rsqrtf(fmaxf(fmaf(-mu, mu, tot2 * (1.0f / DM)), 0.f) + 1e-5f);
    }
    TSTOP(1);
  };
  // ---- T-layout values -> bf16 panel fragments (8-byte stores): features 64 wave + 16 j + 4 g4 + [0, 4) of token 16 tb + l15
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
  auto write_norm = [&](unsigned char* panel, const Acc& v, const float (&mean)[3], const float (&rstd)[3]) {
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb)
        {
          const float r = rstd[tb], nm = -mean[tb] * r;      // (v - mean) rstd as ONE fused multiply-add per value (as rg_seq2.hip)
          panel_store(panel, l15, g4, j, tb, fmaf(v[j][tb][0], r, nm), fmaf(v[j][tb][1], r, nm), fmaf(v[j][tb][2], r, nm), fmaf(v[j][tb][3], r, nm));
        }
  };
  // StylizationBlock front half: SiLU(LN(y) * (1 + scale) + shift) with gain = gamma (1 + scale), off = beta (1 + scale)
  // + shift = vectors 1, 2 of the consuming unit's parameter fragment `ps`
  auto write_styl = [&](unsigned char* panel, const Acc& v, const float (&mean)[3], const float (&rstd)[3], const unsigned char* ps) {
    LANE_LOCAL();
    const float nmr[3] = {-mean[0] * rstd[0], -mean[1] * rstd[1], -mean[2] * rstd[2]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 gain = par_t(ps, 1, j, g4), off = par_t(ps, 2, j, g4);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_f(fmaf(fmaf(v[j][tb][r], rstd[tb], nmr[tb]), gain[r], off[r]));
        panel_store(panel, l15, g4, j, tb, o[0], o[1], o[2], o[3]);
      }
    }
  };
  // stylizing unit: panel = stylize(y) with the unit's own parameters, acc += bias, barrier, acc += W x panel
  auto styl_unit = [&](Acc& acc, unsigned char* panel, const Acc& y) {
    float m3[3], r3[3];
    row_stats(y, m3, r3);
    {
      TSTART();
      const unsigned char* ps = consume();
      write_styl(panel, y, m3, r3, ps);
      add_bias_t(acc, ps);
      release();
      TSTOP(3);
    }
    barx();
    gemm_unit(acc, panel, TL);
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
  // y = softmax(q) A for one head: T-layout out blocks 2 h, 2 h + 1 <- A fragments (hi, lo) of the head's two column
  // blocks, q blocks 2 h, 2 h + 1 (the contraction runs over the head's 32 features)
  auto qa_head = [&](Acc& y, const Acc& q, int h, const bf16x8 (&ah)[2], const bf16x8 (&al)[2]) {
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const float b8[8] = {q[2 * h][tb][0], q[2 * h][tb][1], q[2 * h][tb][2], q[2 * h][tb][3],
                           q[2 * h + 1][tb][0], q[2 * h + 1][tb][1], q[2 * h + 1][tb][2], q[2 * h + 1][tb][3]};
      bf16x8 bh, bl;
      split_hl(b8, bh, bl);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) y[2 * h + jb][tb] = mfma3(ah[jb], al[jb], bh, bl, f32x4{0.f, 0.f, 0.f, 0.f});
    }
  };
  auto dump = [&](const Acc& v) {       // diagnostics: T-layout registers -> a.dump [R][TP][512]
    LANE_LOCAL();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb)
        *reinterpret_cast<f32x4*>(a.dump + ((size_t)seq * TP + 16 * tb + l15) * DM + 64 * wave + 16 * j + 4 * g4) = v[j][tb];
    wait_vmcnt<0>();
  };

  // =========================================================== embedding: x = joint_embed(x_in) + tables
  unit(xr, P0, TL);
  if (a.dump_stage == 1) dump(xr);

#pragma unroll 1
  for (int layer = 0; layer < L; ++layer) {
    const bool dl = a.dump && layer == a.dump_layer;
    float mean[3], rstd[3];
    // ======================================================= self attention (efficient_attention.py:23-45)
    row_stats(xr, mean, rstd);
    write_norm(P0, xr, mean, rstd);            // P0 = xhat; gamma is folded into the weights, beta into the bias
    barx();
    {
      f32x4 Ab[2][2][2];                       // [head][16-row block i][16-column block j] of A_h = softmax_N(K_h)^T V_h
      {
        // both biases of the K / V double unit: vector 0 = key, vector 1 = value; standard layout: feature 16 j + l15
        float bk[4], bv[4];
        unsigned tokbits = tokbits0;
        asm volatile("" : "+v"(tokbits));
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
          f32x4 kk[2][3], vv[2][3];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) {
              kk[j][tb] = f32x4{bk[2 * h + j], bk[2 * h + j], bk[2 * h + j], bk[2 * h + j]};
              vv[j][tb] = f32x4{bv[2 * h + j], bv[2 * h + j], bv[2 * h + j], bv[2 * h + j]};
            }
          gemm_head_std(kk, P0);
          TSTART();
          // softmax over the tokens, per feature column (lane): tokens 16 tb + 4 g4 + r; masked / padded tokens weigh 0
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int tb = 0; tb < 3; ++tb)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if ((tokbits >> (4 * tb + r)) & 1u) mx = fmaxf(mx, kk[j][tb][r]);
            mx = xmax4(mx);
            const float nm2 = mx * -1.44269504088896340736f;
            float sum = 0.f;
#pragma unroll
            for (int tb = 0; tb < 3; ++tb)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float e = ((tokbits >> (4 * tb + r)) & 1u) ? rg_exp_sub(kk[j][tb][r], nm2) : 0.f;
                kk[j][tb][r] = e;
                sum += e;
              }
            sum = xsum4(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) kk[j][tb] *= inv;
          }
          TSTOP(4);
          gemm_head_std(vv, P0);
          // A_h[i][jc] = sum_t P[t][i] V[t][jc] (contraction over tokens: step 0 = token blocks 0 | 1, step 1 = block 2 | zeros)
          bf16x8 vh0[2], vl0[2], vh1[2], vl1[2];
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) {
            const f32x4* vb = vv[jb];
            const float v0[8] = {vb[0][0], vb[0][1], vb[0][2], vb[0][3], vb[1][0], vb[1][1], vb[1][2], vb[1][3]};
            const float v1[8] = {vb[2][0], vb[2][1], vb[2][2], vb[2][3], 0.f, 0.f, 0.f, 0.f};
            split_hl(v0, vh0[jb], vl0[jb]);
            split_hl(v1, vh1[jb], vl1[jb]);
          }
#pragma unroll
          for (int ib = 0; ib < 2; ++ib) {
            const f32x4* kb = kk[ib];
            const float k0[8] = {kb[0][0], kb[0][1], kb[0][2], kb[0][3], kb[1][0], kb[1][1], kb[1][2], kb[1][3]};
            const float k1[8] = {kb[2][0], kb[2][1], kb[2][2], kb[2][3], 0.f, 0.f, 0.f, 0.f};
            bf16x8 kh0, kl0, kh1, kl1;
            split_hl(k0, kh0, kl0);
            split_hl(k1, kh1, kl1);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
              f32x4 d = mfma3(kh0, kl0, vh0[jb], vl0[jb], f32x4{0.f, 0.f, 0.f, 0.f});
              Ab[h][ib][jb] = mfma3(kh1, kl1, vh1[jb], vl1[jb], d);
            }
          }
        }
      }
      Acc qq, yy;
      zero(qq);
      unit(qq, P0, TL);
      softmax_q(qq);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const float a8[8] = {Ab[h][0][jb][0], Ab[h][0][jb][1], Ab[h][0][jb][2], Ab[h][0][jb][3],
                               Ab[h][1][jb][0], Ab[h][1][jb][1], Ab[h][1][jb][2], Ab[h][1][jb][3]};
          split_hl(a8, ah[jb], al[jb]);
        }
        qa_head(yy, qq, h, ah, al);
      }
      if (dl && a.dump_stage == 10) dump(yy);
      styl_unit(xr, P1, yy);                   // x += proj_out(...)  (stylization_block.py:40, efficient_attention.py:44)
    }
    if (dl && a.dump_stage == 2) dump(xr);

    // ======================================================= three cross attentions + ca_mix (efficient_attention.py:62-102,
    // diffusion_transformer.py:110-122), as [h_text | h_audio | h_spk | x] @ W_fused^T (rg_gesture.h: ca_mix fusion)
    row_stats(xr, mean, rstd);
    write_norm(P0, xr, mean, rstd);            // xhat: query projections (per-condition gamma / beta folded) and the x segment
    barx();
    {
      // x W_x^T + b = sd * (xhat W_x^T + rstd * (mean * rowsum(W_x) + b)),  sd = 1 / rstd
      LANE_LOCAL();
      const unsigned char* ps = consume();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b = par_t(ps, 0, j, g4), c1 = par_t(ps, 1, j, g4);
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) xr[j][tb] = (c1 * mean[tb] + b) * rstd[tb];
      }
      release();
      gemm_unit(xr, P0, TL);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const float sd = __builtin_amdgcn_rcpf(rstd[tb]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xr[j][tb] *= sd;
          // (rounded here in both forward kernels: no contraction with the additions of the classifier-free tables behind it)
          asm volatile("" : "+v"(xr[j][tb]));
        }
      }
    }
    if (!cond) {
      // classifier-free rows: + sum_c W_c h_c with h_c one of two tabulated rows per (step, layer, condition)
      LANE_LOCAL();
      unsigned qbits = qbits0;
      asm volatile("" : "+v"(qbits));
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const unsigned char* us = consume();   // fragment 0: (c, flag) = (0,0) (0,1) (1,0) (1,1); fragment 1: (2,0) (2,1)
#pragma unroll
        for (int c2 = 0; c2 < (f == 0 ? 2 : 1); ++c2) {
          const int c = 2 * f + c2;
#pragma unroll
          for (int tb = 0; tb < 3; ++tb) {
            const int flag = (int)((qbits >> (3 * c + tb)) & 1u);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              xr[j][tb] += *reinterpret_cast<const f32x4*>(us + ((c2 * 2 + flag) * 64 + 16 * j + 4 * g4) * 4);
          }
        }
        release();
      }
    } else {
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        Acc qq, yy;
        zero(qq);
        unit(qq, P0, TL);
        softmax_q(qq);
        LANE_LOCAL();
        unsigned qbits = qbits0;
        asm volatile("" : "+v"(qbits));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 ah[2], al[2];
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) {
            const unsigned char* s0 = consume();
            ah[jb] = *reinterpret_cast<const bf16x8*>(s0 + lane * 16);
            release();
            al[jb] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};      // (ATT_HL is off: the stream does not carry low-order halves)
          }
          qa_head(yy, qq, h, ah, al);
        }
        // masked queries: the reference adds -1e6 before the LayerNorm; keep its fp32 rounding (DESIGN: masked query rows)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          if ((qbits >> (3 * c + tb)) & 1u) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float z = yy[j][tb][r] + (-1000000.0f);
                asm volatile("" : "+v"(z));
                yy[j][tb][r] = z + 1000000.0f;
              }
          }
        if (dl && a.dump_stage == 11 + c) dump(yy);
        styl_unit(xr, P1, yy);
      }
    }
    if (dl && a.dump_stage == 3) dump(xr);

    // ======================================================= FFN (diffusion_transformer.py:74-87): 1024 hidden units in two halves
    barx();                                     // every wave is done reading P0
    write_raw(P0, xr);
    barx();
    {
      Acc yf;
      zero(yf);
#pragma unroll 1
      for (int jh = 0; jh < 2; ++jh) {
        Acc gg;
        zero(gg);
        unit(gg, P0, TL);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tb = 0; tb < 3; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) gg[j][tb][r] = gelu_fast(gg[j][tb][r]);
        barx();                                 // P1 is free
        write_raw(P1, gg);
        barx();
        unit(yf, P1, TL);                      // (the bias of linear2 rides with the first half)
      }
      styl_unit(xr, P1, yf);
    }
    if (dl && a.dump_stage == 4) dump(xr);
  }

  // =========================================================== output head (diffusion_transformer.py:662-666)
  LANE_LOCAL();
  barx();
  write_raw(P0, xr);
  barx();
  Acc out;
  zero(out);
  unit(out, P0, TL);
#pragma unroll
  for (int tb = 0; tb < 3; ++tb) {
    const int t = 16 * tb + l15;
    if (t < T) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t o = ((size_t)seq * T + t) * DM + 64 * wave + 16 * j + 4 * g4;
        rg_tail::store_head(a.head, (unsigned)(o * 4), out[j][tb]);     // (written through: with the tail on, another workgroup of this launch reads it)
      }
    }
  }
  wait_vmcnt<0>();
#ifdef RG_STAMPS
  if (a.dump_stage == 99 && lane0 == 0) {
    tacc[5] = __builtin_amdgcn_s_memrealtime() - tk0;
    for (int i = 0; i < 6; ++i) a.dump[(seq * 8 + wave) * 8 + i] = (float)tacc[i];
  }
#endif
}

#undef LANE_LOCAL

// The work of workgroup `block` of a launch of (pairs ? B : 2 B) workgroups.
__device__ __forceinline__ void seq_block(const rg_seq_args& a, const int block, const int pairs, unsigned char* const smem) {
  const int B = a.B;
  // Workgroup -> sequence(s).
  // pairs == 0: one workgroup per sequence.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one,
  // speed only, never correctness): the conditional sequences go to four of the eight groups and the classifier-free ones
  // (which skip a third of the weight stream and run ahead) to the other four, so the workgroups that share an L2 walk the
  // stream together and each L2 pulls one copy of it instead of two.
  // pairs == 1: one workgroup per clip: its conditional sequence, then its classifier-free twin (0.7 of the time).  A launch
  // of B workgroups then holds B compute units for 1.7 units of time instead of 2 B for 1.0, of which the classifier-free
  // half idles the last 0.3: 15 % less CU time per forward, for callers whose launches are narrow enough to run side by
  // side.  Every workgroup of the launch walks the same part of the stream at the same time.
  int seq0 = block;
  if (!pairs && (B & 3) == 0) {
    const int x = block & 7, q = block >> 3;
    seq0 = x < 4 ? 4 * q + x : B + 4 * q + (x - 4);
  }
  const int npass = pairs ? 2 : 1;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    run_sequence(a, seq0 + pass * B, smem);
    __syncthreads();     // descriptors, panels and statistics of the pass are dead in every wave
    int* const ctr = rg_tail::late_ctr();
    if (ctr && !a.dump_stage) {            // the loop step's update of this clip, if its other sequence is done (rg_tail.h)
      const int c = seq0 + pass * B - (seq0 + pass * B >= B ? B : 0);
      rg_tail::arrive_and_glue<NTH>(ctr, c, c, reinterpret_cast<int*>(smem), __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    }
  }
}

#ifndef RG_SEQ_BODY_ONLY      // (rg_seqx.hip includes this file for run_sequence / seq_block only)
__global__ void __launch_bounds__(NTH) rg_seq_kernel(const rg_seq_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RG_OWN_THE_SIMD();
  seq_block(a, blockIdx.x, a.pairs, smem);
}

extern "C" int rg_seq_forward(rg_handle* h, const rg_seq_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  const rg_seq_args& a = *args_host;
  RG_REQUIRE(h, a.wstream && a.pstream && a.ustream && a.afrag && a.x && a.tbias && a.src_mask && a.qmask && a.head, "null pointer");
  RG_REQUIRE(h, a.L >= 1 && a.L <= 8 && a.B >= 1 && a.T >= 1 && a.T <= TP, "unsupported shape (T <= 48, L <= 8)");
  RG_REQUIRE(h, a.step >= 0 && a.step < a.S && a.step_b >= 0 && a.step_b < a.S, "step out of range");
  RG_REQUIRE(h, a.dump_stage == 0 || a.dump, "dump_stage needs a dump buffer");
  RG_REQUIRE(h, a.pairs == 0 || a.pairs == 1, "pairs must be 0 or 1");
  RG_REQUIRE(h, rg_tail::args_ok(a), "glue_ctr: glue must cover the B clips (n_a + n_b == B, T, D = 512), its pointers set, no dump");
  static rg_attr_once lds_once;
  if (!rg_reserve_lds(lds_once, rg_seq_kernel, LDS_BYTES)) {
    h->err = "rg_seq_forward: cannot reserve LDS";
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
    const int nl = a.L, ends = 2;
    const double cond = (UPL * nl + ends) * unit + nl * (2 + 3) * att, unc = (10 * nl + ends) * unit + nl * 2 * att;
    rec.flops = a.B * (cond + unc);
    (void)hipEventRecord(rec.start, rg_stream(stream));
  }
  hipLaunchKernelGGL(rg_seq_kernel, dim3(a.pairs ? a.B : 2 * a.B), dim3(NTH), LDS_BYTES, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  if (h->profiling) {
    (void)hipEventRecord(rec.stop, rg_stream(stream));
    h->prof.push_back(rec);
  }
  return RG_OK;
}
#endif  // RG_SEQ_BODY_ONLY
