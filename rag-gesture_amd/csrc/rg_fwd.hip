// One denoiser forward (both classifier-free branches of B clips, 8 layers) as ONE persistent dataflow launch.
//
// Why: at M = 2*B*43 <= ~4k token rows every per-op launch of the layer chain is latency, not throughput (launch
// boundary + first dependent load + per-CU L2->LDS intake).  The sequences of a batch are
// independent chains (linear attention only mixes the 43 tokens of one sequence, everything else is row-local), so
// the chain is cut into tiles (sequence x 64 output columns, or sequence x head pair) that depend only on earlier
// tiles of the SAME sequence.  Workgroups pull tiles from per-shard queues in topological order; a tile
//   1. requests its weight panel by LDS-DMA (no dependency: the stream runs while the tile waits),
//   2. waits until the sequence's completion counter says its inputs exist (one lane polls with sc1 loads),
//   3. builds its bf16 A panel [48 x 512] in LDS from the sequence's activations (sc1 loads; LayerNorm or
//      LN * (1 + scale) + shift -> SiLU evaluated in fp32 on the way in),
//   4. runs the K loop on the matrix cores (v_mfma_f32_16x16x32_bf16) against the weight ring,
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
typedef __attribute__((address_space(3))) void lds_void;

constexpr int TP = 48;          // token rows of a tile (T <= 48; rows >= T repeat the last token and are never stored)
constexpr int DM = 512;         // model width
constexpr int HD = 32;          // head dim
constexpr int NTH = 256;        // 4 waves
constexpr int NS = 8;           // weight ring depth (64 x 64 bf16 tiles of 8 KiB)
constexpr int WT_BYTES = 8192;
constexpr int SC1 = 16;         // cache-policy bit of the buffer builtins: sc1
constexpr int EPLD = 68;        // fp32 staging row stride (floats)
constexpr int AH_LD = 33;
constexpr int NSHARD = 8;
constexpr int CTRL_ABORT = NSHARD * 32;        // word index of the abort flag
constexpr int CTRL_DONE = CTRL_ABORT + 1;       // workgroups that have drained the queues
constexpr int CTRL_CNT = CTRL_ABORT + 32;      // first completion counter; one per sequence, 16 words apart
constexpr unsigned SPIN_LIMIT = 1u << 20;      // polls before a wait gives up (~1 s)

// ---- LDS map (bytes)
constexpr int OFF_A = 0;                            // bf16 A panel [48][512], chunk-swizzled; reused as q/k/v fp32 staging
constexpr int OFF_EPI = OFF_A + TP * 1024;          // epilogue staging: fp32 [48][68] (+ A_h [2][32][33] behind it)
constexpr int EPI_BYTES = 24576;
constexpr int OFF_W = OFF_EPI + EPI_BYTES;          // weight ring
constexpr int OFF_PAR = OFF_W + NS * WT_BYTES;      // fp32 [4][512]: gamma, beta, 1 + scale, shift
constexpr int OFF_ROW = OFF_PAR + 4 * DM * 4;       // fp32 [48][2]: mean, rstd
constexpr int OFF_MASK = OFF_ROW + TP * 2 * 4;      // fp32 [48] token mask
constexpr int OFF_CTL = OFF_MASK + 256;             // ints: [0] ticket, [1] abort, [2] last workgroup
constexpr int LDS_BYTES = OFF_CTL + 64;
static_assert(3 * TP * EPLD * 4 <= TP * 1024, "q/k/v staging must fit the A panel");
static_assert(TP * EPLD * 4 + 2 * HD * AH_LD * 4 <= EPI_BYTES, "epilogue staging");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

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
__device__ __forceinline__ int w_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, SC1);
}
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, int byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, SC1);
}
__device__ __forceinline__ void st8(__amdgpu_buffer_rsrc_t r, int byte_off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, SC1);
}
__device__ __forceinline__ f32x4 asf(u32x4 v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ u32x4 asu(f32x4 v) { return __builtin_bit_cast(u32x4, v); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// wait until at most `younger` weight tiles (2 DMA instructions per wave each) issued after the current one are in flight
template <int MAXY>
__device__ __forceinline__ void wait_tiles(int younger) {
  if constexpr (MAXY <= 0) {
    wait_vmcnt<0>();
  } else {
    if (younger >= MAXY) wait_vmcnt<MAXY * 2>();
    else wait_tiles<MAXY - 1>(younger);
  }
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

// One stream of 64 x 64 weight tiles: tile i = rows [n0[j], n0[j] + 64) of W[j], k in [64 * (i / nsub), +64), j = i % nsub
struct WStream {
  const unsigned short* w[3];
  int ldw;
  int nsub;
  int ntiles;
};

struct EpiOut {
  float* o32;            // fp32 [rows][ld32] or null
  int ld32;
  unsigned short* o16;   // bf16 or null
  int ld16;
  float* stats;          // (sum, M2) partial per row: stats + (row * stats_ld + part) * 2, or null
  int stats_ld, part;
  bool stats_bf16;       // statistics of the bf16-rounded values (the consumer normalises the bf16 tensor)
  const float* residual; // fp32 [rows][512] or null (sc1 loads)
  const float* tbias;    // [T][512] or null
  const float* bias;
  int col0;              // first output column
  int act;               // 1 = GELU
};

}  // namespace

__global__ void __launch_bounds__(NTH, 1) rg_fwd_kernel(const rg_fwd_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem + OFF_A;
  float* sEpi = reinterpret_cast<float*>(smem + OFF_EPI);
  float* sAh = sEpi + TP * EPLD;
  unsigned char* sW = smem + OFF_W;
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
    next_ticket = take();
  }

  // per-lane weight DMA geometry: a tile is 8 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w and w + 4
  int wrow[2], wlc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    wrow[q] = (wave + 4 * q) * 8 + (lane >> 3);
    wlc[q] = (lane & 7) ^ ((wrow[q] >> 1) & 7);
  }

  while (true) {
    if (tid == 0) sCtl[0] = next_ticket;
    __syncthreads();
    const int ticket = sCtl[0];
    if (ticket < 0) break;
    const int4 td = *reinterpret_cast<const int4*>(sched + 16 + 4 * (size_t)ticket);
    const int type = __builtin_amdgcn_readfirstlane(td.x & 0xff);
    const int layer = __builtin_amdgcn_readfirstlane(td.x >> 8);
    const int seq = __builtin_amdgcn_readfirstlane(td.y);
    const int nt = __builtin_amdgcn_readfirstlane(td.z);
    const unsigned target = (unsigned)__builtin_amdgcn_readfirstlane(td.w);
    if (tid == 0) next_ticket = take();   // in flight during the tile
    unsigned long long* stamp = a.stamps ? a.stamps + 4 * (size_t)ticket : nullptr;
    if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();

    const bool cond = seq < B;
    const int clip = cond ? seq : seq - B;
    const int row0 = seq * T;                       // global row of token 0
    const rg_fwd_layer* LW = a.layers + layer;
    const float* ssl = a.ss + ((size_t)a.step * a.L + layer) * 5 * 1024;

    // ---- stage description
    WStream ws;
    ws.w[0] = ws.w[1] = ws.w[2] = nullptr;
    ws.nsub = 1;
    int npanels = 1;
    switch (type) {
      case RG_FWD_EMBED:
        ws.w[0] = reinterpret_cast<const unsigned short*>(a.w_embed) + (size_t)nt * 64 * DM; ws.ldw = DM; break;
      case RG_FWD_QKV_SA: {
        const unsigned short* w = reinterpret_cast<const unsigned short*>(LW->w_qkv);
        ws.w[0] = w + (size_t)(nt * 64) * DM;
        ws.w[1] = w + (size_t)(DM + nt * 64) * DM;
        ws.w[2] = w + (size_t)(2 * DM + nt * 64) * DM;
        ws.ldw = DM; ws.nsub = 3;
      } break;
      case RG_FWD_SAOUT:
        ws.w[0] = reinterpret_cast<const unsigned short*>(LW->w_sao) + (size_t)nt * 64 * DM; ws.ldw = DM; break;
      case RG_FWD_Q3_CA:   // nt = cond * 8 + head pair: rows cond * 512 + hp * 64 = nt * 64
        ws.w[0] = reinterpret_cast<const unsigned short*>(LW->w_q3) + (size_t)nt * 64 * DM; ws.ldw = DM; break;
      case RG_FWD_MIX:
        ws.w[0] = reinterpret_cast<const unsigned short*>(LW->w_mix) + (size_t)nt * 64 * (4 * DM); ws.ldw = 4 * DM; npanels = 4; break;
      case RG_FWD_FF1:
        ws.w[0] = reinterpret_cast<const unsigned short*>(LW->w_ff1) + (size_t)nt * 64 * DM; ws.ldw = DM; break;
      case RG_FWD_FF2:
        ws.w[0] = reinterpret_cast<const unsigned short*>(LW->w_ff2) + (size_t)nt * 64 * (2 * DM); ws.ldw = 2 * DM; npanels = 2; break;
      case RG_FWD_FFOUT:
        ws.w[0] = reinterpret_cast<const unsigned short*>(LW->w_ffo) + (size_t)nt * 64 * DM; ws.ldw = DM; break;
      default:
        ws.w[0] = reinterpret_cast<const unsigned short*>(a.w_out) + (size_t)nt * 64 * DM; ws.ldw = DM; break;
    }
    ws.ntiles = npanels * 8 * ws.nsub;

    // ---- weight ring: tile i -> stage i % NS
    auto issue = [&](int i) {
      unsigned char* st = sW + (i % NS) * WT_BYTES;
      const int kt = i / ws.nsub, j = i - kt * ws.nsub;
      const unsigned short* wj = j == 0 ? ws.w[0] : (j == 1 ? ws.w[1] : ws.w[2]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        __builtin_amdgcn_global_load_lds((const void*)(wj + (size_t)wrow[q] * ws.ldw + kt * 64 + wlc[q] * 8),
                                         (lds_void*)(st + (wave + 4 * q) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
      if (i < ws.ntiles) issue(i);

    // ---- token masks of this sequence (inputs of the launch: plain loads)
    if (tid < TP) sMask[tid] = tid < T ? a.src_mask[(size_t)seq * T + tid] : 0.f;
    // A_h of the cross attention (two heads of one condition of this clip): requested now, parked in LDS later
    f32x4 apre[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (type == RG_FWD_Q3_CA) {
      const int c = nt >> 3, hp = nt & 7;
      const float* ap = LW->a_pre + ((((size_t)c * B + clip) * 16 + hp * 2) * HD) * HD;   // two consecutive heads: 2048 floats
      apre[0] = *reinterpret_cast<const f32x4*>(ap + tid * 8);
      apre[1] = *reinterpret_cast<const f32x4*>(ap + tid * 8 + 4);
    }

    // ---- dependency: every earlier tile of this sequence has published
    if (tid == 0) {
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
    __syncthreads();
    if (sCtl[1]) break;
    if (stamp && tid == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();

    // ---- residual rows for the epilogue (thread -> (row, 16-byte piece) x 3), requested before the K loop
    const float* res_src = type == RG_FWD_SAOUT ? a.xa : (type == RG_FWD_FFOUT ? a.xc : nullptr);
    f32x4 resv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) resv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (res_src) {
      const __amdgpu_buffer_rsrc_t rr = rsrc_of(res_src + (size_t)row0 * DM);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int it = tid + NTH * j, row = it >> 4, c4 = it & 15;
        const int rl = row < T ? row : T - 1;
        resv[j] = asf(ld16(rr, (rl * DM + nt * 64 + c4 * 4) * 4));
      }
    }

    // ---- A panel builder
    auto fill_panel = [&](const Panel& p) {
      const bool norm = p.mode == PM_F32_LN || p.mode == PM_BF16_STYL;
      if (norm) {
        // parameters -> LDS (inputs of the launch: plain loads), row statistics -> (mean, rstd)
        {
          const int i4 = (tid & 127) * 4;
          const float* src = tid < 128 ? p.gamma : p.beta;
          *reinterpret_cast<f32x4*>(sPar + (tid < 128 ? 0 : DM) + i4) = *reinterpret_cast<const f32x4*>(src + i4);
          if (p.mode == PM_BF16_STYL) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p.ss + (tid < 128 ? 0 : DM) + i4);
            if (tid < 128) v = v + 1.0f;
            *reinterpret_cast<f32x4*>(sPar + (tid < 128 ? 2 * DM : 3 * DM) + i4) = v;
          }
        }
        if (tid < TP) {
          const int rl = tid < T ? tid : T - 1;
          const __amdgpu_buffer_rsrc_t rs = rsrc_of(p.stats);
          const int off = (((p.row0 + rl) * p.stats_ld) + p.part0) * 8;
          f32x4 s[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) s[q] = asf(ld16(rs, off + 16 * q));
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
          sRow[2 * tid] = mean;
          sRow[2 * tid + 1] = rsqrtf(m2 * (1.0f / DM) + 1e-5f);
        }
        __syncthreads();
      }
      if (p.mode == PM_F32 || p.mode == PM_F32_LN) {
        // unit = 4 floats (one 16-byte load) -> 4 bf16; 24 units per thread in 3 batches of 8
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(reinterpret_cast<const float*>(p.src) + (size_t)p.row0 * p.ld + p.col0);
#pragma unroll 1
        for (int b8 = 0; b8 < 3; ++b8) {
          f32x4 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int u = tid + NTH * (b8 * 8 + j), row = u >> 7, q = u & 127;
            const int rl = row < T ? row : T - 1;
            v[j] = asf(ld16(rs, (rl * p.ld + q * 4) * 4));
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int u = tid + NTH * (b8 * 8 + j), row = u >> 7, q = u & 127;
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
          const int u = tid + NTH * j, row = u >> 6, ch = u & 63;
          const int rl = row < T ? row : T - 1;
          v[j] = ld16(rs, (rl * p.ld + ch * 8) * 2);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const int u = tid + NTH * j, row = u >> 6, ch = u & 63;
          u32x4 o = v[j];
          if (p.mode == PM_BF16_STYL) {
            const float mean = sRow[2 * row], rstd = sRow[2 * row + 1];
            float x[8] = {bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1]), bflo(o[2]), bfhi(o[2]), bflo(o[3]), bfhi(o[3])};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 ga = *reinterpret_cast<const f32x4*>(sPar + ch * 8 + 4 * h);
              const f32x4 be = *reinterpret_cast<const f32x4*>(sPar + DM + ch * 8 + 4 * h);
              const f32x4 sc = *reinterpret_cast<const f32x4*>(sPar + 2 * DM + ch * 8 + 4 * h);
              const f32x4 sh = *reinterpret_cast<const f32x4*>(sPar + 3 * DM + ch * 8 + 4 * h);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float ln = (x[4 * h + e] - mean) * rstd * ga[e] + be[e];
                x[4 * h + e] = silu_f(ln * sc[e] + sh[e]);
              }
            }
            o = u32x4{pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7])};
          }
          *reinterpret_cast<u32x4*>(sA + row * 1024 + ((ch ^ (row & 15)) << 4)) = o;
        }
      } else {   // PM_TAB: every token row is one of two tabulated rows
        const unsigned short* tab = reinterpret_cast<const unsigned short*>(p.src) + p.col0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const int u = tid + NTH * j, row = u >> 6, ch = u & 63;
          const int rl = row < T ? row : T - 1;
          const int var = (int)((p.tabflags >> rl) & 1ull);
          const u32x4 o = *reinterpret_cast<const u32x4*>(tab + (size_t)var * p.ld + ch * 8);
          *reinterpret_cast<u32x4*>(sA + row * 1024 + ((ch ^ (row & 15)) << 4)) = o;
        }
      }
      __syncthreads();
    };

    // query-mask bit set of (condition c, this sequence): bit n set -> token n is masked (qmask == 0)
    auto qmask_bits = [&](int c) -> unsigned long long {
      const float mv = lane < T ? a.qmask[((size_t)c * 2 * B + seq) * T + lane] : 1.0f;
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

    // ---- K loop
    f32x4 acc[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 3; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const unsigned char* st, int ktp, f32x4(&ac)[3]) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(st + w_lds_off(16 * wave + l15, 4 * s + g4));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + (r * 16 + l15) * 1024 + (((ktp * 8 + 4 * s + g4) ^ l15) << 4));
          ac[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, ac[r], 0, 0, 0);
        }
      }
    };

    {
      const int per_panel = 8 * ws.nsub;
      int i = 0;
      for (int pi = 0; pi < npanels; ++pi) {
        const Panel p = panel_of(pi);
        fill_panel(p);
        for (int ip = 0; ip < per_panel; ++ip, ++i) {
          const int younger = min(NS - 2, ws.ntiles - 1 - i);
          wait_tiles<NS - 2>(younger);
          __builtin_amdgcn_s_barrier();
          if (i + NS - 1 < ws.ntiles) issue(i + NS - 1);
          const unsigned char* st = sW + (i % NS) * WT_BYTES;
          const int ktp = ip / ws.nsub, j = ip - ktp * ws.nsub;
          if (j == 0) mma(st, ktp, acc[0]);
          else if (j == 1) mma(st, ktp, acc[1]);
          else mma(st, ktp, acc[2]);
        }
        __syncthreads();   // every wave is done with this A panel (and, after the last panel, with LDS operands at all)
      }
    }
    if (stamp && tid == 0) stamp[2] = __builtin_amdgcn_s_memrealtime();

    // ---- row-major output pass shared by all stages: sEpi[48][68] fp32 holds the tile (before residual / tbias)
    auto store_tile = [&](const EpiOut& eo) {
      // fp32 pass: thread -> (row, 4 columns) x 3; adds residual / positional table, stores fp32, leaves the value in sEpi
      if (eo.o32 || eo.residual || eo.tbias) {
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(eo.o32 ? eo.o32 + (size_t)row0 * eo.ld32 + eo.col0 : nullptr);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int it = tid + NTH * j, row = it >> 4, c4 = it & 15;
          f32x4 v = *reinterpret_cast<const f32x4*>(sEpi + row * EPLD + c4 * 4);
          if (eo.residual) v = v + resv[j];
          if (eo.tbias) {
            const int rl = row < T ? row : T - 1;
            v = v + *reinterpret_cast<const f32x4*>(eo.tbias + (size_t)rl * DM + eo.col0 + c4 * 4);
          }
          if (eo.residual || eo.tbias) *reinterpret_cast<f32x4*>(sEpi + row * EPLD + c4 * 4) = v;
          if (eo.o32 && row < T) st16(ro, (row * eo.ld32 + c4 * 4) * 4, asu(v));
        }
        if ((eo.residual || eo.tbias) && (eo.o16 || eo.stats)) __syncthreads();
      }
      if (eo.o16) {
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(eo.o16 + (size_t)row0 * eo.ld16 + eo.col0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int it = tid + NTH * j, row = it >> 3, c8 = it & 7;
          if (it < TP * 8 && row < T) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(sEpi + row * EPLD + c8 * 8);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(sEpi + row * EPLD + c8 * 8 + 4);
            st16(ro, (row * eo.ld16 + c8 * 8) * 2, u32x4{pack2(x0[0], x0[1]), pack2(x0[2], x0[3]), pack2(x1[0], x1[1]), pack2(x1[2], x1[3])});
          }
        }
      }
      if (eo.stats && tid < TP * 4) {
        // (sum, M2 about the part's own mean) over the tile's 64 columns: 4 threads per row, 16 columns each
        const int row = tid >> 2, part = tid & 3;
        float x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(sEpi + row * EPLD + part * 16 + 4 * q);
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
          st8(rs, ((row0 + row) * eo.stats_ld + eo.part) * 8, u32x2{__float_as_uint(s), __float_as_uint(m2)});
        }
      }
    };

    // accumulators (+ bias, activation) -> sEpi; wave w owns columns [16w, 16w + 16)
    auto acc_to_lds = [&](f32x4(&ac)[3], float* dst, const float* bias, int act) {
      const float bv = bias ? bias[16 * wave + l15] : 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = ac[r][e] + bv;
          if (act == 1) v = gelu_fast(v);
          dst[(r * 16 + 4 * g4 + e) * EPLD + 16 * wave + l15] = v;
        }
    };

    // y[n][c] = sum_d qs[n][h*32 + d] * A_h[h][d][c & 31] for the 64 columns (2 heads) of the tile; q rows in sQ
    // (fp32 [48][68]), A_h in sAh; result -> sEpi.  maskbits: rows whose y is rounded like the reference's y - 1e6.
    auto qa_to_epi = [&](const float* sQ, unsigned long long maskbits) {
      const int c = tid & 63, h = c >> 5, l = c & 31;
      float Ar[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) Ar[d] = sAh[(h * HD + d) * AH_LD + l];
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
        sEpi[n * EPLD + c] = y;
      }
    };
    // softmax over each head's 32 columns of the q rows in sQ: thread -> (row, head), 96 threads
    auto q_softmax = [&](float* sQ) {
      if (tid < TP * 2) {
        float* qr = sQ + (tid >> 1) * EPLD + (tid & 1) * HD;
        float v[HD];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(qr + 4 * q);
          v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
        float mx = v[0];
#pragma unroll
        for (int e = 1; e < HD; ++e) mx = fmaxf(mx, v[e]);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) { v[e] = __expf(v[e] - mx); sum += v[e]; }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          *reinterpret_cast<f32x4*>(qr + 4 * q) = f32x4{v[4 * q] * inv, v[4 * q + 1] * inv, v[4 * q + 2] * inv, v[4 * q + 3] * inv};
      }
    };

    EpiOut eo;
    eo.o32 = nullptr; eo.ld32 = DM; eo.o16 = nullptr; eo.ld16 = DM; eo.stats = nullptr; eo.stats_ld = 8; eo.part = nt;
    eo.stats_bf16 = false; eo.residual = nullptr; eo.tbias = nullptr; eo.bias = nullptr; eo.col0 = nt * 64; eo.act = 0;

    if (type == RG_FWD_QKV_SA) {
      // ---- q, k, v (+ bias) -> fp32 staging in the A panel region
      float* sQ = reinterpret_cast<float*>(sA);
      float* sK = sQ + TP * EPLD;
      float* sV = sK + TP * EPLD;
      acc_to_lds(acc[0], sQ, LW->b_qkv + nt * 64, 0);
      acc_to_lds(acc[1], sK, LW->b_qkv + DM + nt * 64, 0);
      acc_to_lds(acc[2], sV, LW->b_qkv + 2 * DM + nt * 64, 0);
      __syncthreads();
      // softmax of q over head_dim (waves 0-1), of k over the tokens (wave 2), v * mask (wave 3)
      // (efficient_attention.py:32-36: key + (1 - mask) * -1e6 underflows to weight 0 exactly)
      q_softmax(sQ);
      if (wave == 2) {
        const int c = lane;
        float mx = -INFINITY;
        for (int n = 0; n < T; ++n)
          if (sMask[n] != 0.f) mx = fmaxf(mx, sK[n * EPLD + c]);
        float sum = 0.f;
        for (int n = 0; n < T; ++n) {
          const float e = sMask[n] != 0.f ? __expf(sK[n * EPLD + c] - mx) : 0.f;
          sK[n * EPLD + c] = e;
          sum += e;
        }
        const float inv = 1.0f / sum;
        for (int n = 0; n < T; ++n) sK[n * EPLD + c] *= inv;
      } else if (wave == 3) {
        const int c = lane;
        for (int n = 0; n < T; ++n) sV[n * EPLD + c] *= sMask[n];
      }
      __syncthreads();
      // A_h[d][l] = sum_n P[n][d] V[n][l]: thread -> (head, d, 8 columns)
      {
        const int h = tid >> 7, d = (tid >> 2) & 31, l0 = (tid & 3) * 8;
        float s8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = 0.f;
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
      qa_to_epi(sQ, 0ull);
      __syncthreads();
      eo.o16 = reinterpret_cast<unsigned short*>(a.ysa);
      eo.stats = a.st_sa; eo.stats_bf16 = true;
      store_tile(eo);
    } else if (type == RG_FWD_Q3_CA) {
      const int c = nt >> 3, hp = nt & 7;
      float* sQ = reinterpret_cast<float*>(sA);
      acc_to_lds(acc[0], sQ, LW->b_q3 + nt * 64, 0);
      // park A_h: thread tid holds floats [8 tid, 8 tid + 8) of the two heads' [2][32][32]
      {
        const int f0 = tid * 8, h = f0 >> 10, d = (f0 >> 5) & 31, l0 = f0 & 31;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sAh[(h * HD + d) * AH_LD + l0 + e] = apre[0][e];
          sAh[(h * HD + d) * AH_LD + l0 + 4 + e] = apre[1][e];
        }
      }
      const unsigned long long mb = qmask_bits(c);
      __syncthreads();
      q_softmax(sQ);
      __syncthreads();
      qa_to_epi(sQ, mb);
      __syncthreads();
      eo.o16 = reinterpret_cast<unsigned short*>(a.y3);
      eo.ld16 = 3 * DM; eo.col0 = c * DM + hp * 64;
      eo.stats = a.st3; eo.stats_ld = 24; eo.part = c * 8 + hp; eo.stats_bf16 = true;
      store_tile(eo);
    } else {
      switch (type) {
        case RG_FWD_EMBED: eo.bias = a.b_embed; eo.tbias = a.tbias; eo.o32 = a.xa; eo.stats = a.st_a; break;
        case RG_FWD_SAOUT:
          eo.bias = LW->b_sao; eo.residual = a.xa; eo.o32 = a.xb; eo.o16 = reinterpret_cast<unsigned short*>(a.xb_bf);
          eo.stats = a.st_b; break;
        case RG_FWD_MIX: eo.bias = LW->b_mix; eo.o32 = a.xc; eo.o16 = reinterpret_cast<unsigned short*>(a.xc_bf); break;
        case RG_FWD_FF1: eo.bias = LW->b_ff1; eo.act = 1; eo.o16 = reinterpret_cast<unsigned short*>(a.g); eo.ld16 = 2 * DM; break;
        case RG_FWD_FF2:
          eo.bias = LW->b_ff2; eo.o16 = reinterpret_cast<unsigned short*>(a.yf); eo.stats = a.st_f; eo.stats_bf16 = true; break;
        case RG_FWD_FFOUT: eo.bias = LW->b_ffo; eo.residual = a.xc; eo.o32 = a.xa; eo.stats = a.st_a; break;
        default: eo.bias = a.b_out; eo.o32 = a.head; break;
      }
      acc_to_lds(acc[0], sEpi, eo.bias + nt * 64, eo.act);
      __syncthreads();
      store_tile(eo);
    }

    // ---- publish: every wave's stores have completed, then one agent-scope add on the sequence's counter
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(ctrl + CTRL_CNT + seq * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
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

extern "C" int rg_denoiser_forward(rg_handle* h, const rg_fwd_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  const rg_fwd_args& a = *args_host;
  RG_REQUIRE(h, a.layers && a.sched && a.ctrl && a.x && a.xa && a.head, "null pointer");
  RG_REQUIRE(h, a.L >= 1 && a.B >= 1 && a.T >= 1 && a.T <= TP && a.step >= 0, "unsupported shape (T <= 48)");
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)rg_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      h->err = "rg_denoiser_forward: cannot reserve LDS";
      return RG_ERR_HIP;
    }
    attr = true;
  }
  hipStream_t s = rg_stream(stream);
  hipLaunchKernelGGL(rg_fwd_kernel, dim3(h->num_cus), dim3(NTH), LDS_BYTES, s, a);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}
