// What lies between two denoiser forwards of a diffusion loop, per token row -- shared by the launch that does it for a whole
// batch (rg_sampler.hip cobatch_glue_kernel) and by the TAIL of the sequence-stationary forwards (rg_seq.hip, rg_seq2.hip):
// with rg_seq_args.glue_ctr set, the workgroup that finishes the SECOND of a clip's two sequences (conditional, classifier-free)
// does the clip's update itself, so a loop step is one launch instead of two (the update launch and the two launch boundaries
// around it are ~40 us of a ~1.5 ms step, beside a chip that is full of other lanes' workgroups).
//
// Coherence.  The two sequences of a clip run in different workgroups, in general on different XCDs, whose L2s do not see each
// other's lines inside a launch.  So with the tail on, the forwards store their head rows write-through (buffer stores with
// sc0 sc1), every wave waits for its stores (vmcnt(0)) before the workgroup barrier in front of the arrival, the arrival is
// an agent-scope atomic add on the clip's counter, and the last arriver reads both head rows with sc0 sc1 loads (they miss
// every cache).  No cache-wide write-back or invalidate is involved.  Everything else the update reads was written by
// earlier launches, and what it writes is read by later ones.
// The counters only ever grow: two arrivals per clip and launch, the one that finds an odd count is the second.
#pragma once
#include "rg_common.h"
#include <cstddef>

namespace rg_tail {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int COHERENT = 17;          // cache policy of the buffer instructions: sc0 | sc1 (bits 0 and 4 on gfx940+)

// The reference evaluates eps and the update as separate fp32 torch ops: same operation order, no FMA contraction (the fp32
// oracle agrees bit for bit, tests/test_sampler_gpu.py).
__device__ __forceinline__ float ddim_one(float x, float x0, float c_recip, float c_recipm1, float c_a, float c_b) {
#pragma clang fp contract(off)
  float eps = (c_recip * x - x0) / c_recipm1;
  return x0 * c_a + c_b * eps;
}
__device__ __forceinline__ float cfg_one(float oc, float ou, float jc, float ju, float w_c, float w_u) {
#pragma clang fp contract(off)
  // reference order: out_text*both*js + out_text*text*js + out_none*retr/js + out_none*none/js;
  // with (both,text) and (retr,none) pre-summed on the host (one of each pair is 0 for t>100).
  return oc * w_c * jc + ou * w_u * ju;
}

template <typename P>
__device__ __forceinline__ P* uniform_ptr(P* p) {       // a wave-uniform pointer the compiler does not know to be one
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  return reinterpret_cast<P*>((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32 |
                              (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v));
}
// 16 bytes of a head row past every cache: base (wave-uniform) + row_bytes (wave-uniform, a scalar operand) + 16 j
__device__ __forceinline__ float4 load_out(const float* base, const unsigned row_bytes, const int j) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, j * 16, row_bytes, COHERENT);
  return float4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// The group's coefficients of a step.
struct step_coef { float w_c, w_u, c_recip, c_recipm1, c_a, c_b; };
__device__ __forceinline__ step_coef coef_of(const rg_glue_args& a, const bool ga) {
  return ga ? step_coef{a.wc_a, a.wu_a, a.c_recip_a, a.c_recipm1_a, a.ca_a, a.cb_a}
            : step_coef{a.wc_b, a.wu_b, a.c_recip_b, a.c_recipm1_b, a.ca_b, a.cb_b};
}

// Four consecutive features of one token row: x <- cfg_ddim(out_c, out_u, x) of this step; then, where the NEXT step inserts on
// the row (ins; y = its in_seq, e = its noise draw), the insertion-guidance update and the in-sequence replacement
// (rg_cfg_ddim_update_rows, rg_guidance_update, rg_inseq_replace, in that order, operation for operation).
__device__ __forceinline__ float4 glue_elem(const rg_glue_args& a, const step_coef& k, const float4 c, const float4 u, const float4 x,
                                            const float jc, const float ju, const bool ins, const float4 y, const float4 e,
                                            const float two_over_numel) {
#pragma clang fp contract(off)
  float4 p, v;
  p.x = cfg_one(c.x, u.x, jc, ju, k.w_c, k.w_u);
  p.y = cfg_one(c.y, u.y, jc, ju, k.w_c, k.w_u);
  p.z = cfg_one(c.z, u.z, jc, ju, k.w_c, k.w_u);
  p.w = cfg_one(c.w, u.w, jc, ju, k.w_c, k.w_u);
  v.x = ddim_one(x.x, p.x, k.c_recip, k.c_recipm1, k.c_a, k.c_b);
  v.y = ddim_one(x.y, p.y, k.c_recip, k.c_recipm1, k.c_a, k.c_b);
  v.z = ddim_one(x.z, p.z, k.c_recip, k.c_recipm1, k.c_a, k.c_b);
  v.w = ddim_one(x.w, p.w, k.c_recip, k.c_recipm1, k.c_a, k.c_b);
  if (ins) {
    for (int it = 0; it < a.g_iter_next; ++it) {          // (rg_guidance_update; what follows overwrites it, as in the reference)
      v.x = v.x - a.lr * (two_over_numel * (v.x - y.x));
      v.y = v.y - a.lr * (two_over_numel * (v.y - y.y));
      v.z = v.z - a.lr * (two_over_numel * (v.z - y.z));
      v.w = v.w - a.lr * (two_over_numel * (v.w - y.w));
    }
    v.x = a.s_ab_next * y.x + a.s_1mab_next * e.x;
    v.y = a.s_ab_next * y.y + a.s_1mab_next * e.y;
    v.z = a.s_ab_next * y.z + a.s_1mab_next * e.z;
    v.w = a.s_ab_next * y.w + a.s_1mab_next * e.w;
  }
  return v;
}

// One token row (group a: ga, row rl of the group = clip * T + token) by one wave (rg_cobatch_glue's launch):
//   sampling rows (a):  this step's update, then the next step's guidance update and in-sequence replacement on marked rows;
//   inverting rows (b): this step's update with the inversion's coefficients, and a second copy (the level kept).
__device__ __forceinline__ void glue_row(const rg_glue_args& a, const bool ga, const int rl, const int lane, const float two_over_numel) {
  const int d4 = a.D >> 2;
  const float jc = a.js[rl % a.T], ju = 1.0f / jc;
  const float4* oc = reinterpret_cast<const float4*>((ga ? a.out_c_a : a.out_c_b) + (int64_t)rl * a.D);
  const float4* ou = reinterpret_cast<const float4*>((ga ? a.out_u_a : a.out_u_b) + (int64_t)rl * a.D);
  float4* xr = reinterpret_cast<float4*>((ga ? a.x_a : a.x_b) + (int64_t)rl * a.D);
  const step_coef k = coef_of(a, ga);
  bool ins = false;                                         // the next step inserts on this row
  const float4* s = nullptr;
  if (ga && a.in_seq_next) {
    s = reinterpret_cast<const float4*>(a.in_seq_next + (int64_t)rl * a.D);
    bool nz = false;
    for (int j = lane; j < d4; j += 64) {
      float4 v = s[j];
      nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    ins = __ballot(nz) != 0ull;
  }
  const float4* nn = ins ? reinterpret_cast<const float4*>(a.noise_next + (int64_t)rl * a.D) : nullptr;
  float4* x2 = (!ga && a.x_b_copy) ? reinterpret_cast<float4*>(a.x_b_copy + (int64_t)rl * a.D) : nullptr;
  const float4 zero{0.f, 0.f, 0.f, 0.f};
  for (int j = lane; j < d4; j += 64) {
    const float4 v = glue_elem(a, k, oc[j], ou[j], xr[j], jc, ju, ins, ins ? s[j] : zero, ins ? nn[j] : zero, two_over_numel);
    xr[j] = v;
    if (x2) x2[j] = v;
  }
}

// A whole clip (group a: ga, clip cl of the group) by the NW waves of a workgroup, at the end of a forward (D = 512: a row is
// two 16-byte pieces per lane; T <= 48: NB batches of RPW rows per wave).  The same arithmetic; what differs is the order of the
// memory operations: every load of a batch of rows is issued before the first result is needed -- row after row, each a chain of
// loads that miss every cache (the head rows) in front of its stores, the clip took ~18 us of a compute unit's time per step.
// 16 bytes of global memory (the pointers below have lost their provenance: say which address space they point to)
typedef __attribute__((address_space(1))) f32x4 gf32x4;
__device__ __forceinline__ float4 gload(const float* p, const int j) {
  const f32x4 v = ((const gf32x4*)p)[j];
  return float4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void gstore(float* p, const int j, const float4 v) { ((gf32x4*)p)[j] = f32x4{v.x, v.y, v.z, v.w}; }
template <int NW>
__device__ __forceinline__ void glue_clip(const rg_glue_args& a, const bool ga, const int cl, const int wave, const int lane,
                                          const float two_over_numel) {
  constexpr int RPW = 3, NB = (48 + NW * RPW - 1) / (NW * RPW);     // rows per wave and batch; batches (120 registers of loads each)
  const int T = a.T;
  const step_coef k = coef_of(a, ga);
  const float* const oc0 = uniform_ptr(ga ? a.out_c_a : a.out_c_b);
  const float* const ou0 = uniform_ptr(ga ? a.out_u_a : a.out_u_b);
  float* const x0 = uniform_ptr(ga ? a.x_a : a.x_b);
  const bool marks = ga && a.in_seq_next;
  // (without marks the two loads below read x instead: valid memory, values unused -- no branch between the loads of a batch)
  const float* const y0 = uniform_ptr(marks ? a.in_seq_next : x0);
  const float* const e0 = uniform_ptr(marks ? a.noise_next : x0);
  float* const x2_0 = uniform_ptr((!ga && a.x_b_copy) ? a.x_b_copy : nullptr);
#pragma unroll 1
  for (int nb = 0; nb < NB; ++nb) {
    float4 C[RPW][2], U[RPW][2], X[RPW][2], Y[RPW][2], E[RPW][2];
    float jc[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int t = min(wave + NW * (RPW * nb + i), T - 1);                // (rows beyond T: a valid row again, nothing stored)
      const int row = __builtin_amdgcn_readfirstlane(cl * T + t);         // (uniform by construction; says so to the compiler)
      const int64_t o = (int64_t)row * 512;
      jc[i] = a.js[t];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        C[i][h] = load_out(oc0, (unsigned)row * 2048u, lane + 64 * h);
        U[i][h] = load_out(ou0, (unsigned)row * 2048u, lane + 64 * h);
        X[i][h] = gload(x0 + o, lane + 64 * h);
        Y[i][h] = gload(y0 + o, lane + 64 * h);
        E[i][h] = gload(e0 + o, lane + 64 * h);
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int t = wave + NW * (RPW * nb + i);
      const int64_t o = (int64_t)__builtin_amdgcn_readfirstlane(cl * T + min(t, T - 1)) * 512;
      bool nz = false;
#pragma unroll
      for (int h = 0; h < 2; ++h) nz |= (Y[i][h].x != 0.f) | (Y[i][h].y != 0.f) | (Y[i][h].z != 0.f) | (Y[i][h].w != 0.f);
      const bool ins = marks && __ballot(nz) != 0ull;                     // the next step inserts on this row
      const float ju = 1.0f / jc[i];
      if (t < T) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 v = glue_elem(a, k, C[i][h], U[i][h], X[i][h], jc[i], ju, ins, Y[i][h], E[i][h], two_over_numel);
          gstore(x0 + o, lane + 64 * h, v);
          if (x2_0) gstore(x2_0 + o, lane + 64 * h, v);
        }
      }
    }
  }
}

__device__ __forceinline__ float two_over_numel(const rg_glue_args& a) {
#pragma clang fp contract(off)
  const int rows_a = a.n_a * a.T;
  return 2.0f / ((float)rows_a * (float)a.D);
}

// A member of the launch's rg_seq_args (the kernel's ONLY parameter: offset 0 of the kernarg segment) read WHERE THE CALL
// STANDS.  As plain kernel-parameter accesses the ~45 scalars of `glue` are loaded at the kernel's entry and stay live -- that
// is, spilled -- across the whole forward (rg_seq2: 191 -> 1 288 scalar spills and 6 vector ones); behind the opaque copy of
// the segment pointer the compiler can neither hoist nor merge the loads.
template <typename T>
__device__ __forceinline__ T late_arg(const unsigned offset) {
  auto p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  T v;
  __builtin_memcpy(&v, p + offset, sizeof(T));
  return v;
}
__device__ __forceinline__ int* late_ctr() { return late_arg<int*>(offsetof(rg_seq_args, glue_ctr)); }

// A head row of a forward whose tail is on: 16 bytes, written through every cache.
__device__ __forceinline__ void store_head(float* head, const unsigned byte_offset, const f32x4 v) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(head, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_offset, 0, COHERENT);
}

// The workgroup has stored the head rows of the sequences of clips c0 and c1 (c1 == c0: one sequence), every wave has waited
// for its stores, and a workgroup barrier lies behind that.  `flag`: one int of LDS nobody else touches until the call returns.
// Called by every thread of the workgroup (NTH threads).
// (wave: the caller's wave-uniform index; the lane index is taken from the hardware here, so that no vector register of the
//  forward has to live until its end for the tail's sake)
template <int NTH>
__device__ __forceinline__ void arrive_and_glue(int* const ctr, const int c0, const int c1, int* const flag, const int wave) {
  const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  if (wave == 0 && lane == 0) {
    int m = 0;
    if (__hip_atomic_fetch_add(ctr + c0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) m |= 1;
    if (c1 != c0 && (__hip_atomic_fetch_add(ctr + c1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1)) m |= 2;
    *flag = m;
  }
  __syncthreads();
  const int m = *flag;
  __syncthreads();
  if (m == 0) return;
  const rg_glue_args g = late_arg<rg_glue_args>(offsetof(rg_seq_args, glue));
  const float ton = two_over_numel(g);
#pragma unroll 1
  for (int q = 0; q < 2; ++q) {
    if (!(m >> q & 1)) continue;
    const int c = q ? c1 : c0;
    const bool ga = c < g.n_a;
    const int cl = ga ? c : c - g.n_a;
    glue_clip<NTH / 64>(g, ga, cl, wave, lane, ton);
  }
}

// Host side: what the entry points require of args->glue when args->glue_ctr is set.
inline bool args_ok(const rg_seq_args& a) {
  if (!a.glue_ctr) return true;
  const rg_glue_args& g = a.glue;
  if (a.dump_stage || g.n_a < 0 || g.n_b < 0 || g.n_a + g.n_b != a.B || g.T != a.T || g.D != 512 || !g.js || g.g_iter_next < 0) return false;
  if (g.n_a && !(g.out_c_a && g.out_u_a && g.x_a)) return false;
  if (g.n_b && !(g.out_c_b && g.out_u_b && g.x_b)) return false;
  return !g.in_seq_next || g.noise_next;
}

}  // namespace rg_tail
