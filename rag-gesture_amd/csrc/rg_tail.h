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

template <bool COH>
__device__ __forceinline__ float4 load_out(const float* row, const int j) {
  if (COH) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, 0x7fffffff, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, j * 16, 0, COHERENT);
    return float4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  }
  return reinterpret_cast<const float4*>(row)[j];
}

// One token row (group a: ga, row rl of the group = clip * T + token) by one wave:
//   sampling rows (a):  x <- cfg_ddim(out_c, out_u, x) of this step; then, for the NEXT step, the insertion-guidance update and
//                       the in-sequence replacement where its in_seq marks the row (rg_cfg_ddim_update_rows, rg_guidance_update,
//                       rg_inseq_replace, in that order, operation for operation);
//   inverting rows (b): x <- cfg_ddim(...) of this step with the inversion's coefficients, and a second copy (the level kept).
template <bool COH>
__device__ __forceinline__ void glue_row(const rg_glue_args& a, const bool ga, const int rl, const int lane,
                                         const float two_over_numel) {
#pragma clang fp contract(off)
  const int d4 = a.D >> 2;
  const float jc = a.js[rl % a.T], ju = 1.0f / jc;
  const float* oc = (ga ? a.out_c_a : a.out_c_b) + (int64_t)rl * a.D;
  const float* ou = (ga ? a.out_u_a : a.out_u_b) + (int64_t)rl * a.D;
  float4* xr = reinterpret_cast<float4*>((ga ? a.x_a : a.x_b) + (int64_t)rl * a.D);
  const float w_c = ga ? a.wc_a : a.wc_b, w_u = ga ? a.wu_a : a.wu_b;
  const float c_recip = ga ? a.c_recip_a : a.c_recip_b, c_recipm1 = ga ? a.c_recipm1_a : a.c_recipm1_b;
  const float c_a = ga ? a.ca_a : a.ca_b, c_b = ga ? a.cb_a : a.cb_b;
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
  for (int j = lane; j < d4; j += 64) {
    const float4 c = load_out<COH>(oc, j), u = load_out<COH>(ou, j), x = xr[j];
    float4 p, v;
    p.x = cfg_one(c.x, u.x, jc, ju, w_c, w_u);
    p.y = cfg_one(c.y, u.y, jc, ju, w_c, w_u);
    p.z = cfg_one(c.z, u.z, jc, ju, w_c, w_u);
    p.w = cfg_one(c.w, u.w, jc, ju, w_c, w_u);
    v.x = ddim_one(x.x, p.x, c_recip, c_recipm1, c_a, c_b);
    v.y = ddim_one(x.y, p.y, c_recip, c_recipm1, c_a, c_b);
    v.z = ddim_one(x.z, p.z, c_recip, c_recipm1, c_a, c_b);
    v.w = ddim_one(x.w, p.w, c_recip, c_recipm1, c_a, c_b);
    if (ins) {
      const float4 y = s[j], e = nn[j];
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
    xr[j] = v;
    if (x2) x2[j] = v;
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
#pragma unroll 1
    for (int t = wave; t < g.T; t += NTH / 64) glue_row<true>(g, ga, cl * g.T + t, lane, ton);
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
