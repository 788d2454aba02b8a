// One denoiser forward (both classifier-free branches of B clips, 8 layers) as ONE persistent dataflow launch.
//
// Why: at M = 2*B*43 <= ~4k token rows every per-op launch of the layer chain is latency, not throughput (launch
// boundary + first dependent load + per-CU intake).  The sequences of a batch are independent
// chains (linear attention only mixes the 43 tokens of one sequence, everything else is row-local), so the chain is
// cut into tiles (sequence x 64 output columns, or sequence x head pair) that depend only on earlier tiles of the
// SAME sequence.  Workgroups (4 waves, 2-3 resident per CU so one tile's latencies hide behind another's work) pull
// tiles from per-shard queues in topological order; a tile
//   1. requests its weight slice straight into registers (no dependency: the loads fly while the tile waits; a wave
//      owns 16 output columns and holds their whole K = 512 slice in 64 VGPRs, so the K loop has no LDS staging of
//      weights and no barriers),
//   2. waits until the sequence's completion counter says its inputs exist (one lane polls with sc1 loads),
//   3. builds its bf16 A panel [48 x 512] in LDS from the sequence's activations (sc1 loads; LayerNorm or
//      LN * (1 + scale) + shift -> SiLU evaluated in fp32 on the way in),
//   4. runs the K loop on the matrix cores (v_mfma_f32_16x16x32_bf16, A fragments from LDS, B from registers),
//   5. applies the stage's epilogue (bias, GELU, residual, the two attention softmaxes and the 32x32 linear
//      attention products, row statistics) and writes its outputs with write-through (sc1) 16-byte stores,
//   6. waits for its stores (vmcnt(0) in every wave, workgroup barrier) and adds 1 to the sequence's counter.
// Hand-off form: MI355X guide "inter-workgroup visibility", row 1 of the sc1 table (sc1 stores, one agent-scope
// atomic add per workgroup behind a barrier, sc1-load poll, barrier, sc1 loads); no fences, no grid barrier.
// Deadlock freedom: tickets are taken in queue order and a tile only waits for tiles earlier in its own queue, so the
// oldest incomplete tile is always being worked on, whatever number of workgroups is resident.
#include "rg_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int TP = 48;          // token rows of a tile (T <= 48; rows >= T repeat the last token and are never stored)
constexpr int DM = 512;         // model width
constexpr int HD = 32;          // head dim
constexpr int NTH = 256;        // 4 waves
constexpr int SC1 = 16;         // cache-policy bit of the buffer builtins: sc1
constexpr int EPLD = 68;        // fp32 staging row stride (floats)
constexpr int AH_LD = 33;
constexpr int NSHARD = 8;
constexpr int CTRL_ABORT = NSHARD * 32;        // word index of the abort flag
constexpr int CTRL_DONE = CTRL_ABORT + 1;       // workgroups that have drained the queues
constexpr int CTRL_CNT = CTRL_ABORT + 32;      // first completion counter; one per sequence, 16 words apart
constexpr unsigned SPIN_LIMIT = 1u << 20;      // polls before a wait gives up (~1 s)

// ---- LDS map (bytes).  The A panel region doubles as the epilogue's fp32 staging once the K loop is done:
//      sQ [48][68] | sK [48][68] | sV [48][68] | A_h [2][32][33]
constexpr int OFF_A = 0;                            // bf16 A panel [48][512], 16-byte chunks XOR-swizzled by row & 15
constexpr int STG = TP * EPLD * 4;                  // 13056 bytes: one fp32 staging tile
constexpr int OFF_PAR = OFF_A + TP * 1024;          // fp32 [2][512]: LayerNorm gain / offset (stylization folded in)
constexpr int OFF_ROW = OFF_PAR + 2 * DM * 4;       // fp32 [48][2]: mean, rstd
constexpr int OFF_MASK = OFF_ROW + TP * 2 * 4;      // fp32 [48] token mask
constexpr int OFF_CTL = OFF_MASK + TP * 4;          // ints: [0] ticket, [1] abort, [2] last workgroup
constexpr int LDS_BYTES = OFF_CTL + 64;
static_assert(3 * STG + 2 * HD * AH_LD * 4 <= TP * 1024, "epilogue staging must fit the A panel");
static_assert(3 * LDS_BYTES <= 160 * 1024, "three workgroups per CU");

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
// GELU (erf form) with erf by Abramowitz-Stegun 7.1.26 (abs. error 1.5e-7), as rg_gemm's bf16 path
__device__ __forceinline__ float gelu_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = 1.0f - pl * t * __builtin_amdgcn_exp2f(x * x * -1.44269504088896340736f);
  return 0.5f * v + 0.5f * fabsf(v) * e;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// activation traffic; POL = SC1 inside the persistent launch (hand-offs between running workgroups), 0 = default cache
// policy when every stage is a launch of its own (the kernel boundary orders producers and consumers, rows stay in L2)
template <int POL>
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, POL);
}
template <int POL>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, int byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, POL);
}
template <int POL>
__device__ __forceinline__ void st8(__amdgpu_buffer_rsrc_t r, int byte_off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, POL);
}
__device__ __forceinline__ f32x4 asf(u32x4 v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ u32x4 asu(f32x4 v) { return __builtin_bit_cast(u32x4, v); }

// plain (cached) loads of launch inputs -- weights, biases, tables -- through a global-address-space pointer: pointers that
// come out of the layer table in memory would otherwise be treated as generic (flat_load: slower, ties up lgkmcnt)
template <class V>
__device__ __forceinline__ V gld(const void* p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) V*>(reinterpret_cast<uintptr_t>(p));
}

enum { PM_F32 = 0, PM_F32_LN = 1, PM_BF16 = 2, PM_BF16_STYL = 3, PM_TAB = 4 };

// One K = 512 panel of the A operand
struct Panel {
  const void* src;       // fp32 or bf16 rows (or the bf16 table for PM_TAB)
  int ld;                // row stride, elements
  int col0;              // first column, elements
  int row0;              // global row of token 0
  int mode;
  const float* stats;    // partial (sum, M2) pairs: stats + (row * stats_ld + part0) * 2, 8 parts
  int stats_ld, part0;
  const float* gamma;    // [512]
  const float* beta;
  const float* ss;       // scale [512] | shift [512] (PM_BF16_STYL)
  unsigned long long tabflags;   // PM_TAB: bit n set -> token n takes table row 1
};

struct EpiOut {
  float* o32;            // fp32 [rows][ld32] or null
  int ld32;
  unsigned short* o16;   // bf16 or null
  int ld16;
  float* stats;          // (sum, M2) partial per row: stats + (row * stats_ld + part) * 2, or null
  int stats_ld, part;
  bool stats_bf16;       // statistics of the bf16-rounded values (the consumer normalises the bf16 tensor)
  const float* residual; // fp32 [rows][512] added to the tile (sc1 loads), or null
  const float* tbias;    // [T][512] or null
  int col0;              // first output column
};

}  // namespace

// PERSIST = true: the whole forward, tiles pulled from the queues (a.sched) with dependency waits and publishes.
// PERSIST = false: ONE stage per launch -- workgroup b runs tile tile0 + b of the stage-major list `stage_tiles`
// (same tile code; the kernel boundary replaces the hand-off protocol, activations use the default cache policy).
template <bool PERSIST>
__global__ void __launch_bounds__(NTH, 2) rg_fwd_kernel(const rg_fwd_args a, const int* __restrict__ stage_tiles, int tile0) {
  constexpr int POL = PERSIST ? SC1 : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem + OFF_A;
  float* sQ = reinterpret_cast<float*>(smem + OFF_A);             // staging tile 0 (generic epilogue: the output tile)
  float* sK = reinterpret_cast<float*>(smem + OFF_A + STG);       // staging tile 1 (attention: k, then y)
  float* sV = reinterpret_cast<float*>(smem + OFF_A + 2 * STG);
  float* sAh = reinterpret_cast<float*>(smem + OFF_A + 3 * STG);
  float* sPar = reinterpret_cast<float*>(smem + OFF_PAR);
  float* sRow = reinterpret_cast<float*>(smem + OFF_ROW);
  float* sMask = reinterpret_cast<float*>(smem + OFF_MASK);
  volatile int* sCtl = reinterpret_cast<volatile int*>(smem + OFF_CTL);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int T = a.T, B = a.B;
  unsigned* ctrl = a.ctrl;
  const int* sched = a.sched;

  // ---- ticket queue (thread 0): own shard first, then the others (work stealing; any workgroup may run any tile)
  int tries = 0;
  const int shard0 = blockIdx.x & (NSHARD - 1);
  auto take = [&]() -> int {
    while (tries < NSHARD) {
      const int q = (shard0 + tries) & (NSHARD - 1);
      const int first = sched[q], n = sched[q + 1] - first;
      if (n > 0) {
        const unsigned k = __hip_atomic_fetch_add(ctrl + q * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)k < n) return first + (int)k;
      }
      ++tries;
    }
    return -1;
  };
  int next_ticket = -1;
  if (tid == 0) {
    sCtl[1] = 0;
    next_ticket = PERSIST ? take() : tile0 + (int)blockIdx.x;
  }

  // ---- weight slice of this wave's 16 columns for one K = 512 panel, straight into registers: lane (l15, g4) holds,
  // for each 64-wide k block, the 32 bytes W[row l15][64 kb + 16 g4 .. + 16): the 4 lanes of a row read one whole
  // 128-byte line.  The first 8 values feed MFMA step 0, the last 8 step 1; the A fragments are read from LDS with the
  // same enumeration of the contraction index (chunk 8 kb + 2 g4 + s).
  auto load_h = [&](u32x4(&wb)[8], const unsigned short* wp) {   // one half panel: k in [wp, wp + 256)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      wb[2 * kb] = gld<u32x4>(wp + kb * 64);
      wb[2 * kb + 1] = gld<u32x4>(wp + kb * 64 + 8);
    }
  };
  // A fragment of row tile r, k block kbt (0..7), MFMA step s: chunk 8 kbt + 2 g4 + s of row 16 r + l15, stored at
  // chunk ^ l15.  Its low four bits are (2 g4 ^ l15) ^ (8 (kbt & 1) + s): four per-lane base offsets, everything
  // else is an immediate.
  const int lx = (2 * g4) ^ l15;
  const int ab0 = l15 * 1024 + (lx << 4), ab1 = l15 * 1024 + ((lx ^ 1) << 4);
  const int ab8 = l15 * 1024 + ((lx ^ 8) << 4), ab9 = l15 * 1024 + ((lx ^ 9) << 4);
  // The fragments of k block kb are read one block ahead of their MFMAs (two register sets); the scheduling barriers
  // keep the compiler from hoisting every LDS read of a panel in front of the first MFMA (hundreds of VGPRs)
  auto mma_h = [&](const u32x4(&wb)[8], f32x4(&ac)[3], int kb0) {   // kb0 = 0 / 4: first / second half of the A panel
    bf16x8 af[2][6];
    auto frags = [&](bf16x8(&f)[6], int kb) {
      const int kbt = kb0 + kb;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ab = (kb & 1) ? (s ? ab9 : ab8) : (s ? ab1 : ab0);   // kb0 is even: kbt & 1 == kb & 1
#pragma unroll
        for (int r = 0; r < 3; ++r) f[s * 3 + r] = *reinterpret_cast<const bf16x8*>(sA + ab + r * 16384 + (kbt >> 1) * 256);
      }
    };
    frags(af[0], 0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb + 1 < 4) frags(af[(kb + 1) & 1], kb + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, wb[2 * kb + s]);
#pragma unroll
        for (int r = 0; r < 3; ++r) ac[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kb & 1][s * 3 + r], bfr, ac[r], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  while (true) {
    if (tid == 0) sCtl[0] = next_ticket;
    __syncthreads();
    const int ticket = sCtl[0];
    if (ticket < 0) break;
    const int4 td = *reinterpret_cast<const int4*>((PERSIST ? sched + 16 : stage_tiles) + 4 * (size_t)ticket);
    const int type = __builtin_amdgcn_readfirstlane(td.x & 0xff);
    if (!PERSIST && type == 0xff) break;   // padding slot of the stage list
    const int layer = __builtin_amdgcn_readfirstlane(td.x >> 8);
    const int seq = __builtin_amdgcn_readfirstlane(td.y);
    const int nt = __builtin_amdgcn_readfirstlane(td.z);
    const unsigned target = (unsigned)__builtin_amdgcn_readfirstlane(td.w);
    if (tid == 0) next_ticket = PERSIST ? take() : -1;   // in flight during the tile
    unsigned long long* stamp = a.stamps ? a.stamps + 4 * (size_t)ticket : nullptr;
    if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();

    // index arithmetic below uses an opaque copy of the thread id: the loop-invariant address math of every unrolled
    // load / store would otherwise be hoisted out of the tile loop and live (spilled) across it
    int vt = tid;
    asm volatile("" : "+v"(vt));
    const bool cond = seq < B;
    const int clip = cond ? seq : seq - B;
    const int row0 = seq * T;                       // global row of token 0
    const rg_fwd_layer* LW = a.layers + layer;
    const float* ssl = a.ss + ((size_t)a.step * a.L + layer) * 5 * 1024;

    // ---- weight panels of the stage: panel p = rows [n0, n0 + 64) of W, k in [wk0 + 512 p, + 512)
    const unsigned short* wbase;
    int ldw = DM, nwp = 1;
    switch (type) {
      case RG_FWD_EMBED: wbase = reinterpret_cast<const unsigned short*>(a.w_embed); break;
      case RG_FWD_QKV_SA: wbase = reinterpret_cast<const unsigned short*>(LW->w_qkv); nwp = 3; break;   // q, k, v row blocks
      case RG_FWD_SAOUT: wbase = reinterpret_cast<const unsigned short*>(LW->w_sao); break;
      case RG_FWD_Q3_CA: wbase = reinterpret_cast<const unsigned short*>(LW->w_q3); break;   // nt = cond * 8 + head pair
      case RG_FWD_MIX: wbase = reinterpret_cast<const unsigned short*>(LW->w_mix); ldw = 4 * DM; nwp = 4; break;
      case RG_FWD_FF1: wbase = reinterpret_cast<const unsigned short*>(LW->w_ff1); break;
      case RG_FWD_FF2: wbase = reinterpret_cast<const unsigned short*>(LW->w_ff2); ldw = 2 * DM; nwp = 2; break;
      case RG_FWD_FFOUT: wbase = reinterpret_cast<const unsigned short*>(LW->w_ffo); break;
      default: wbase = reinterpret_cast<const unsigned short*>(a.w_out); break;
    }
    // this lane's row of the slice; QKV: panel p is the q / k / v row block, others: panel p is the next 512 k
    const unsigned short* wlane = wbase + (size_t)(nt * 64 + 16 * wave + l15) * ldw + 16 * g4;
    const size_t wstep = type == RG_FWD_QKV_SA ? (size_t)DM * DM : (size_t)DM;

    // two half-panel buffers (k 0-255 / 256-511 of a panel: 32 VGPRs each), refilled as soon as their MFMAs are issued.
    // Waves 1-3 request the first panel now; wave 0 polls first (its loads would sit in front of the poll's return
    // in the in-order vmcnt queue) and requests it when the dependency has resolved
    u32x4 wa[8], wb[8];
    if (!PERSIST || wave != 0) {
      load_h(wa, wlane);
      load_h(wb, wlane + 256);
    }

    // ---- token mask of this sequence (input of the launch: plain loads)
    if (tid < TP) sMask[tid] = tid < T ? gld<float>(a.src_mask + (size_t)seq * T + tid) : 0.f;
    // A_h of the cross attention (two heads of one condition of this clip): requested now, parked in LDS later
    f32x4 apre[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (type == RG_FWD_Q3_CA) {
      const int c = nt >> 3, hp = nt & 7;
      const float* ap = LW->a_pre + ((((size_t)c * B + clip) * 16 + hp * 2) * HD) * HD;   // two consecutive heads: 2048 floats
      apre[0] = gld<f32x4>(ap + tid * 8);
      apre[1] = gld<f32x4>(ap + tid * 8 + 4);
    }

    // ---- dependency: every earlier tile of this sequence has published
    if (PERSIST && tid == 0) {
      const unsigned* cnt = ctrl + CTRL_CNT + seq * 16;
      unsigned spins = 0;
      int bad = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0 &&
            (spins > SPIN_LIMIT || __hip_atomic_load(ctrl + CTRL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          __hip_atomic_store(ctrl + CTRL_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = 1;
          break;
        }
      }
      sCtl[1] = bad;
    }
    if (PERSIST) {
      __syncthreads();
      if (sCtl[1]) break;
    }
    if (stamp && tid == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
    if (PERSIST && wave == 0) {
      load_h(wa, wlane);
      load_h(wb, wlane + 256);
    }

    // ---- A panel builder
    auto fill_panel = [&](const Panel& p) {
      const bool norm = p.mode == PM_F32_LN || p.mode == PM_BF16_STYL;
      if (norm) {
        // LayerNorm gain / offset with the stylization folded in: LN(x) * (1 + scale) + shift = xhat * g' + b',
        // g' = gamma (1 + scale), b' = beta (1 + scale) + shift  (inputs of the launch: plain loads)
        if (vt < 128) {
          const int i4 = vt * 4;
          f32x4 ga = gld<f32x4>(p.gamma + i4);
          f32x4 be = gld<f32x4>(p.beta + i4);
          if (p.mode == PM_BF16_STYL) {
            const f32x4 sc = gld<f32x4>(p.ss + i4) + 1.0f;
            const f32x4 sh = gld<f32x4>(p.ss + DM + i4);
            ga = ga * sc;
            be = be * sc + sh;
          }
          *reinterpret_cast<f32x4*>(sPar + i4) = ga;
          *reinterpret_cast<f32x4*>(sPar + DM + i4) = be;
        } else if (vt < 128 + TP) {
          // row statistics -> (mean, rstd): 8 partial (sum, M2) pairs per row, combined exactly (Chan)
          const int r = vt - 128;
          const int rl = r < T ? r : T - 1;
          const __amdgpu_buffer_rsrc_t rs = rsrc_of(p.stats);
          const int off = (((p.row0 + rl) * p.stats_ld) + p.part0) * 8;
          f32x4 s[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) s[q] = asf(ld16<POL>(rs, off + 16 * q));
          float su = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) su += s[q][0] + s[q][2];
          const float mean = su * (1.0f / DM);
          float m2 = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d0 = s[q][0] * (1.0f / 64) - mean, d1 = s[q][2] * (1.0f / 64) - mean;
            m2 += s[q][1] + s[q][3] + 64.0f * (d0 * d0 + d1 * d1);
          }
          sRow[2 * r] = mean;
          sRow[2 * r + 1] = rsqrtf(m2 * (1.0f / DM) + 1e-5f);
        }
      }
      if (p.mode == PM_F32 || p.mode == PM_F32_LN) {
        // unit = 4 floats (one 16-byte load) -> 4 bf16; 24 units per thread in 2 batches of 12
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(reinterpret_cast<const float*>(p.src) + (size_t)p.row0 * p.ld + p.col0);
#pragma unroll 1
        for (int b8 = 0; b8 < 2; ++b8) {
          f32x4 v[12];
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const int u = vt + NTH * (b8 * 12 + j), row = u >> 7, q = u & 127;
            const int rl = row < T ? row : T - 1;
            v[j] = asf(ld16<POL>(rs, (rl * p.ld + q * 4) * 4));
          }
          if (norm && b8 == 0) __syncthreads();   // parameters and row statistics are in LDS
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const int u = vt + NTH * (b8 * 12 + j), row = u >> 7, q = u & 127;
            f32x4 x = v[j];
            if (p.mode == PM_F32_LN) {
              const float mean = sRow[2 * row], rstd = sRow[2 * row + 1];
              const f32x4 ga = *reinterpret_cast<const f32x4*>(sPar + q * 4);
              const f32x4 be = *reinterpret_cast<const f32x4*>(sPar + DM + q * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = (x[e] - mean) * rstd * ga[e] + be[e];
            }
            const u32x2 o = {pack2(x[0], x[1]), pack2(x[2], x[3])};
            *reinterpret_cast<u32x2*>(sA + row * 1024 + ((((q >> 1) ^ (row & 15))) << 4) + (q & 1) * 8) = o;
          }
        }
      } else if (p.mode == PM_BF16 || p.mode == PM_BF16_STYL) {
        // unit = 8 bf16 (one 16-byte load); 12 units per thread
        const __amdgpu_buffer_rsrc_t rs =
            rsrc_of(reinterpret_cast<const unsigned short*>(p.src) + (size_t)p.row0 * p.ld + p.col0);
        u32x4 v[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const int u = vt + NTH * j, row = u >> 6, ch = u & 63;
          const int rl = row < T ? row : T - 1;
          v[j] = ld16<POL>(rs, (rl * p.ld + ch * 8) * 2);
        }
        if (norm) __syncthreads();
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const int u = vt + NTH * j, row = u >> 6, ch = u & 63;
          u32x4 o = v[j];
          if (p.mode == PM_BF16_STYL) {
            const float mean = sRow[2 * row], rstd = sRow[2 * row + 1];
            float x[8] = {bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1]), bflo(o[2]), bfhi(o[2]), bflo(o[3]), bfhi(o[3])};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 ga = *reinterpret_cast<const f32x4*>(sPar + ch * 8 + 4 * h);
              const f32x4 be = *reinterpret_cast<const f32x4*>(sPar + DM + ch * 8 + 4 * h);
#pragma unroll
              for (int e = 0; e < 4; ++e) x[4 * h + e] = silu_f(fmaf((x[4 * h + e] - mean) * rstd, ga[e], be[e]));
            }
            o = u32x4{pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7])};
          }
          *reinterpret_cast<u32x4*>(sA + row * 1024 + ((ch ^ (row & 15)) << 4)) = o;
        }
      } else {   // PM_TAB: every token row is one of two tabulated rows
        const unsigned short* tab = reinterpret_cast<const unsigned short*>(p.src) + p.col0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const int u = vt + NTH * j, row = u >> 6, ch = u & 63;
          const int rl = row < T ? row : T - 1;
          const int var = (int)((p.tabflags >> rl) & 1ull);
          const u32x4 o = gld<u32x4>(tab + (size_t)var * p.ld + ch * 8);
          *reinterpret_cast<u32x4*>(sA + row * 1024 + ((ch ^ (row & 15)) << 4)) = o;
        }
      }
      __syncthreads();
    };

    // query-mask bit set of (condition c, this sequence): bit n set -> token n is masked (qmask == 0)
    auto qmask_bits = [&](int c) -> unsigned long long {
      const float mv = lane < T ? gld<float>(a.qmask + ((size_t)c * 2 * B + seq) * T + lane) : 1.0f;
      return __ballot(mv == 0.f);
    };

    auto panel_of = [&](int pi) -> Panel {
      Panel p;
      p.src = nullptr; p.ld = DM; p.col0 = 0; p.row0 = row0; p.mode = PM_BF16; p.stats = nullptr; p.stats_ld = 8; p.part0 = 0;
      p.gamma = p.beta = p.ss = nullptr; p.tabflags = 0ull;
      switch (type) {
        case RG_FWD_EMBED: p.src = a.x; p.row0 = clip * T; p.mode = PM_F32; break;
        case RG_FWD_QKV_SA: p.src = a.xa; p.mode = PM_F32_LN; p.stats = a.st_a; p.gamma = LW->sa_g; p.beta = LW->sa_b; break;
        case RG_FWD_SAOUT:
          p.src = a.ysa; p.mode = PM_BF16_STYL; p.stats = a.st_sa; p.gamma = LW->sa_sg; p.beta = LW->sa_sb; p.ss = ssl; break;
        case RG_FWD_Q3_CA: {
          const int c = nt >> 3;
          p.src = a.xb; p.mode = PM_F32_LN; p.stats = a.st_b; p.gamma = LW->ca_g + c * DM; p.beta = LW->ca_b + c * DM;
        } break;
        case RG_FWD_MIX:
          if (pi == 3) { p.src = a.xb_bf; p.mode = PM_BF16; }
          else if (cond) {
            p.src = a.y3; p.ld = 3 * DM; p.col0 = pi * DM; p.mode = PM_BF16_STYL; p.stats = a.st3; p.stats_ld = 24; p.part0 = pi * 8;
            p.gamma = LW->ca_sg + pi * DM; p.beta = LW->ca_sb + pi * DM; p.ss = ssl + (1 + pi) * 1024;
          } else {
            p.src = reinterpret_cast<const unsigned short*>(LW->unc_tab) + (size_t)a.step * 2 * 3 * DM;
            p.ld = 3 * DM; p.col0 = pi * DM; p.mode = PM_TAB; p.tabflags = qmask_bits(pi);
          }
          break;
        case RG_FWD_FF1: p.src = a.xc_bf; p.mode = PM_BF16; break;
        case RG_FWD_FF2: p.src = a.g; p.ld = 2 * DM; p.col0 = pi * DM; p.mode = PM_BF16; break;
        case RG_FWD_FFOUT:
          p.src = a.yf; p.mode = PM_BF16_STYL; p.stats = a.st_f; p.gamma = LW->ff_sg; p.beta = LW->ff_sb; p.ss = ssl + 4 * 1024; break;
        default: p.src = a.xa; p.mode = PM_F32; break;   // head
      }
      return p;
    };

    // ---- K loops.  Panel p accumulates into acc[p]: the three row blocks of QKV stay apart (q, k, v), the K panels of
    // the other stages are summed in the epilogue -- one straight-line schedule for every stage
    const bool qkv = type == RG_FWD_QKV_SA;
    f32x4 acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 3; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    fill_panel(panel_of(0));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p < nwp) {
        if (p > 0 && !qkv) {   // next 512 columns of A
          __syncthreads();
          fill_panel(panel_of(p));
        }
        mma_h(wa, acc[p], 0);
        if (p + 1 < nwp) load_h(wa, wlane + (p + 1) * wstep);
        mma_h(wb, acc[p], 4);
        if (p + 1 < nwp) load_h(wb, wlane + (p + 1) * wstep + 256);
      }
    }
    __syncthreads();   // every wave is done with the A panel: the region becomes epilogue staging
    if (stamp && vt == 0) stamp[2] = __builtin_amdgcn_s_memrealtime();

    // ---- row-major output pass shared by all stages: src[48][68] fp32 holds the tile (before residual / tbias)
    auto store_tile = [&](float* src, const EpiOut& eo) {
      // fp32 pass: thread -> (row, 4 columns) x 3; adds residual / positional table, stores fp32, leaves the value in src
      if (eo.o32 || eo.residual || eo.tbias) {
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(eo.o32 ? eo.o32 + (size_t)row0 * eo.ld32 + eo.col0 : nullptr);
        f32x4 resv[3];
        if (eo.residual) {
          const __amdgpu_buffer_rsrc_t rr = rsrc_of(eo.residual + (size_t)row0 * DM + eo.col0);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int it = vt + NTH * j, row = it >> 4, c4 = it & 15;
            resv[j] = asf(ld16<POL>(rr, ((row < T ? row : T - 1) * DM + c4 * 4) * 4));
          }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int it = vt + NTH * j, row = it >> 4, c4 = it & 15;
          f32x4 v = *reinterpret_cast<const f32x4*>(src + row * EPLD + c4 * 4);
          if (eo.residual) v = v + resv[j];
          if (eo.tbias) {
            const int rl = row < T ? row : T - 1;
            v = v + gld<f32x4>(eo.tbias + (size_t)rl * DM + eo.col0 + c4 * 4);
          }
          if (eo.residual || eo.tbias) *reinterpret_cast<f32x4*>(src + row * EPLD + c4 * 4) = v;
          if (eo.o32 && row < T) st16<POL>(ro, (row * eo.ld32 + c4 * 4) * 4, asu(v));
        }
        if ((eo.residual || eo.tbias) && (eo.o16 || eo.stats)) __syncthreads();
      }
      if (eo.o16) {
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(eo.o16 + (size_t)row0 * eo.ld16 + eo.col0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int it = vt + NTH * j, row = it >> 3, c8 = it & 7;
          if (it < TP * 8 && row < T) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + row * EPLD + c8 * 8);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + row * EPLD + c8 * 8 + 4);
            st16<POL>(ro, (row * eo.ld16 + c8 * 8) * 2, u32x4{pack2(x0[0], x0[1]), pack2(x0[2], x0[3]), pack2(x1[0], x1[1]), pack2(x1[2], x1[3])});
          }
        }
      }
      if (eo.stats && vt < TP * 4) {
        // (sum, M2 about the part's own mean) over the tile's 64 columns: 4 threads per row, 16 columns each
        const int row = vt >> 2, part = vt & 3;
        float x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(src + row * EPLD + part * 16 + 4 * q);
          x[4 * q] = t[0]; x[4 * q + 1] = t[1]; x[4 * q + 2] = t[2]; x[4 * q + 3] = t[3];
        }
        if (eo.stats_bf16) {
#pragma unroll
          for (int e = 0; e < 16; ++e) x[e] = bf2f(f2bf(x[e]));
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s += x[e];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mp = s * (1.0f / 64);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) m2 = fmaf(x[e] - mp, x[e] - mp, m2);
        m2 += __shfl_xor(m2, 1);
        m2 += __shfl_xor(m2, 2);
        if (part == 0 && row < T) {
          const __amdgpu_buffer_rsrc_t rs = rsrc_of(eo.stats);
          st8<POL>(rs, ((row0 + row) * eo.stats_ld + eo.part) * 8, u32x2{__float_as_uint(s), __float_as_uint(m2)});
        }
      }
    };

    // accumulators (+ bias, activation) -> staging tile; wave w owns columns [16w, 16w + 16)
    auto acc_to_lds = [&](f32x4(&ac)[3], float* dst, const float* bias, int act) {
      const float bv = gld<float>(bias + 16 * wave + l15);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = ac[r][e] + bv;
          if (act == 1) v = gelu_fast(v);
          dst[(r * 16 + 4 * g4 + e) * EPLD + 16 * wave + l15] = v;
        }
    };

    // y[n][c] = sum_d qs[n][h*32 + d] * A_h[h][d][c & 31] for the 64 columns (2 heads) of the tile; q rows in sQ,
    // A_h in sAh; result -> sK.  maskbits: rows whose y is rounded like the reference's y - 1e6.
    auto qa_to_sk = [&](unsigned long long maskbits) {
      const int c = vt & 63, h = c >> 5, l = c & 31;
      float Ar[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) Ar[d] = sAh[(h * HD + d) * AH_LD + l];
#pragma unroll 2
      for (int n = wave; n < TP; n += 4) {
        const float* qr = sQ + n * EPLD + h * HD;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < HD; d4 += 4) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(qr + d4);
          a0 = fmaf(q[0], Ar[d4], a0);
          a1 = fmaf(q[1], Ar[d4 + 1], a1);
          a2 = fmaf(q[2], Ar[d4 + 2], a2);
          a3 = fmaf(q[3], Ar[d4 + 3], a3);
        }
        float y = (a0 + a1) + (a2 + a3);
        if ((maskbits >> (n < T ? n : T - 1)) & 1ull) {   // fp32 rounding of the reference's y + (1 - query_mask) * -1e6
          const float z = y + (-1000000.0f);
          y = z + 1000000.0f;
        }
        sK[n * EPLD + c] = y;
      }
    };
    // softmax over each head's 32 columns of the q rows in sQ: 4 lanes per (row, head), 8 values each
    auto q_softmax = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int it = vt + NTH * j;
        if (it < TP * 8) {
          float* qr = sQ + (it >> 3) * EPLD + (it & 7) * 8;
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(qr), t1 = *reinterpret_cast<const f32x4*>(qr + 4);
          float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
          float mx = v[0];
#pragma unroll
          for (int e = 1; e < 8; ++e) mx = fmaxf(mx, v[e]);
          mx = fmaxf(mx, __shfl_xor(mx, 1));
          mx = fmaxf(mx, __shfl_xor(mx, 2));
          float sum = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] = __expf(v[e] - mx); sum += v[e]; }
          sum += __shfl_xor(sum, 1);
          sum += __shfl_xor(sum, 2);
          const float inv = 1.0f / sum;
          *reinterpret_cast<f32x4*>(qr) = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
          *reinterpret_cast<f32x4*>(qr + 4) = f32x4{v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv};
        }
      }
    };

    EpiOut eo;
    eo.o32 = nullptr; eo.ld32 = DM; eo.o16 = nullptr; eo.ld16 = DM; eo.stats = nullptr; eo.stats_ld = 8; eo.part = nt;
    eo.stats_bf16 = false; eo.residual = nullptr; eo.tbias = nullptr; eo.col0 = nt * 64;

    if (type == RG_FWD_QKV_SA) {
      // ---- q, k, v (+ bias) -> fp32 staging
      acc_to_lds(acc[0], sQ, LW->b_qkv + nt * 64, 0);
      acc_to_lds(acc[1], sK, LW->b_qkv + DM + nt * 64, 0);
      acc_to_lds(acc[2], sV, LW->b_qkv + 2 * DM + nt * 64, 0);
      __syncthreads();
      // softmax of q over head_dim; softmax of k over the tokens: 4 lanes per column, 12 tokens each
      // (efficient_attention.py:32-36: key + (1 - mask) * -1e6 underflows to weight 0 exactly, so P = 0 on masked
      // tokens and the value mask is implied)
      q_softmax();
      {
        const int c = vt >> 2, part = vt & 3;
        float kr[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int n = part + 4 * i;
          const float kv = sK[(n < TP ? n : TP - 1) * EPLD + c];
          kr[i] = (n < T && sMask[n < TP ? n : TP - 1] != 0.f) ? kv : -INFINITY;
        }
        float mx = kr[0];
#pragma unroll
        for (int i = 1; i < 12; ++i) mx = fmaxf(mx, kr[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          kr[i] = kr[i] == -INFINITY ? 0.f : __expf(kr[i] - mx);
          sum += kr[i];
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int n = part + 4 * i;
          if (n < TP) sK[n * EPLD + c] = kr[i] * inv;
        }
      }
      __syncthreads();
      // A_h[d][l] = sum_n P[n][d] V[n][l]: thread -> (head, d, 8 columns)
      {
        const int h = vt >> 7, d = (vt >> 2) & 31, l0 = (vt & 3) * 8;
        float s8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = 0.f;
#pragma unroll 4
        for (int n = 0; n < T; ++n) {
          const float pn = sK[n * EPLD + h * HD + d];
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sV + n * EPLD + h * HD + l0);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sV + n * EPLD + h * HD + l0 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s8[e] = fmaf(pn, v0[e], s8[e]);
            s8[4 + e] = fmaf(pn, v1[e], s8[4 + e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sAh[(h * HD + d) * AH_LD + l0 + e] = s8[e];
      }
      __syncthreads();
      qa_to_sk(0ull);
      __syncthreads();
      eo.o16 = reinterpret_cast<unsigned short*>(a.ysa);
      eo.stats = a.st_sa; eo.stats_bf16 = true;
      store_tile(sK, eo);
    } else if (type == RG_FWD_Q3_CA) {
      const int c = nt >> 3, hp = nt & 7;
      acc_to_lds(acc[0], sQ, LW->b_q3 + nt * 64, 0);
      // park A_h: thread vt holds floats [8 vt, 8 vt + 8) of the two heads' [2][32][32]
      {
        const int f0 = vt * 8, h = f0 >> 10, d = (f0 >> 5) & 31, l0 = f0 & 31;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sAh[(h * HD + d) * AH_LD + l0 + e] = apre[0][e];
          sAh[(h * HD + d) * AH_LD + l0 + 4 + e] = apre[1][e];
        }
      }
      const unsigned long long mb = qmask_bits(c);
      __syncthreads();
      q_softmax();
      __syncthreads();
      qa_to_sk(mb);
      __syncthreads();
      eo.o16 = reinterpret_cast<unsigned short*>(a.y3);
      eo.ld16 = 3 * DM; eo.col0 = c * DM + hp * 64;
      eo.stats = a.st3; eo.stats_ld = 24; eo.part = c * 8 + hp; eo.stats_bf16 = true;
      store_tile(sK, eo);
    } else {
      const float* bias;
      int act = 0;
      switch (type) {
        case RG_FWD_EMBED: bias = a.b_embed; eo.tbias = a.tbias; eo.o32 = a.xa; eo.stats = a.st_a; break;
        case RG_FWD_SAOUT:
          bias = LW->b_sao; eo.residual = a.xa; eo.o32 = a.xb; eo.o16 = reinterpret_cast<unsigned short*>(a.xb_bf);
          eo.stats = a.st_b; break;
        case RG_FWD_MIX: bias = LW->b_mix; eo.o32 = a.xc; eo.o16 = reinterpret_cast<unsigned short*>(a.xc_bf); break;
        case RG_FWD_FF1: bias = LW->b_ff1; act = 1; eo.o16 = reinterpret_cast<unsigned short*>(a.g); eo.ld16 = 2 * DM; break;
        case RG_FWD_FF2:
          bias = LW->b_ff2; eo.o16 = reinterpret_cast<unsigned short*>(a.yf); eo.stats = a.st_f; eo.stats_bf16 = true; break;
        case RG_FWD_FFOUT: bias = LW->b_ffo; eo.residual = a.xc; eo.o32 = a.xa; eo.stats = a.st_a; break;
        default: bias = a.b_out; eo.o32 = a.head; break;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) acc[0][r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
      acc_to_lds(acc[0], sQ, bias + nt * 64, act);
      __syncthreads();
      store_tile(sQ, eo);
    }

    // ---- publish: every wave's stores have completed, then one agent-scope add on the sequence's counter
    if (PERSIST) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(ctrl + CTRL_CNT + seq * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp && tid == 0) stamp[3] = __builtin_amdgcn_s_memrealtime();
    if (!PERSIST) break;
  }
  if (!PERSIST) return;
  // ---- the last workgroup to drain the queues (every tile has been published by then) leaves the control block
  // zeroed for the next launch: no memset between launches, nothing for a graph to reorder.  After an abort the
  // block stays as it is (the abort word is the caller's evidence; the caller re-zeroes it).
  if (tid == 0) {
    int last = 0;
    if (sCtl[1] == 0) {
      const unsigned d = __hip_atomic_fetch_add(ctrl + CTRL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = d == gridDim.x - 1 &&
             __hip_atomic_load(ctrl + CTRL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    }
    sCtl[2] = last;
  }
  __syncthreads();
  if (sCtl[2]) {
    const int nwords = CTRL_CNT + 2 * B * 16;
    for (int i = tid; i < nwords; i += NTH) __hip_atomic_store(ctrl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int rg_fwd_ctrl_words(int B) { return CTRL_CNT + 2 * B * 16; }

static int fwd_prepare(rg_handle* h, const void* fn, int* wgs_per_cu) {
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
    h->err = "rg_denoiser_forward: cannot reserve LDS";
    return RG_ERR_HIP;
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, NTH, LDS_BYTES) != hipSuccess || n < 1) n = 1;
  *wgs_per_cu = n > 3 ? 3 : n;
  return RG_OK;
}

extern "C" int rg_denoiser_forward(rg_handle* h, const rg_fwd_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  const rg_fwd_args& a = *args_host;
  RG_REQUIRE(h, a.layers && a.sched && a.ctrl && a.x && a.xa && a.head, "null pointer");
  RG_REQUIRE(h, a.L >= 1 && a.B >= 1 && a.T >= 1 && a.T <= TP && a.step >= 0, "unsupported shape (T <= 48)");
  static int wgs_per_cu = 0;
  if (wgs_per_cu == 0) {
    const int rc = fwd_prepare(h, (const void*)rg_fwd_kernel<true>, &wgs_per_cu);
    if (rc != RG_OK) return rc;
  }
  // one workgroup per resident slot; correctness does not depend on the grid size (any number of workgroups drains
  // the queues), only the overlap of one tile's latencies with another tile's work does
  hipLaunchKernelGGL(rg_fwd_kernel<true>, dim3(h->num_cus * wgs_per_cu), dim3(NTH), LDS_BYTES, rg_stream(stream), a,
                     (const int*)nullptr, 0);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_denoiser_forward_stages(rg_handle* h, const rg_fwd_args* args_host, const int* stage_tiles,
                                          const int* stage_first_host, int n_stages, void* stream) {
  RG_REQUIRE(h, args_host && stage_tiles && stage_first_host, "null pointer");
  const rg_fwd_args& a = *args_host;
  RG_REQUIRE(h, a.layers && a.x && a.xa && a.head, "null pointer");
  RG_REQUIRE(h, a.L >= 1 && a.B >= 1 && a.T >= 1 && a.T <= TP && a.step >= 0 && n_stages >= 1, "unsupported shape (T <= 48)");
  static int wgs_per_cu = 0;
  if (wgs_per_cu == 0) {
    const int rc = fwd_prepare(h, (const void*)rg_fwd_kernel<false>, &wgs_per_cu);
    if (rc != RG_OK) return rc;
  }
  for (int s = 0; s < n_stages; ++s) {
    const int n = stage_first_host[s + 1] - stage_first_host[s];
    if (n <= 0) continue;
    hipLaunchKernelGGL(rg_fwd_kernel<false>, dim3(n), dim3(NTH), LDS_BYTES, rg_stream(stream), a, stage_tiles, stage_first_host[s]);
  }
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}
