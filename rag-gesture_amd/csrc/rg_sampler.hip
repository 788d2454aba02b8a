// Sampler elementwise kernels: DDIM update / inversion update, CFG mix + DDIM, in-sequence
// replacement, insertion-guidance update, exemplar splice.  All HBM-bandwidth bound:
// 16 B per lane coalesced accesses, grid-stride, no LDS.  (include/rg_gesture.h cites the
// reference lines each one replaces.)
#include "rg_common.h"
#include "rg_tail.h"

namespace {

constexpr int kBlock = 256;

// The reference evaluates eps and the update as separate fp32 torch ops; keep the same
// operation order and forbid FMA contraction so results match the fp32 oracle bit for bit.
#pragma clang fp contract(off)
using rg_tail::ddim_one;     // (rg_tail.h: the denoiser kernels' tails do the same arithmetic)
using rg_tail::cfg_one;

__global__ void __launch_bounds__(kBlock) ddim_update_kernel(const float4* __restrict__ x,
                                                            const float4* __restrict__ x0,
                                                            float4* __restrict__ xo, int64_t n4,
                                                            float c_recip, float c_recipm1,
                                                            float c_a, float c_b) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    float4 a = x[i], b = x0[i], r;
    r.x = ddim_one(a.x, b.x, c_recip, c_recipm1, c_a, c_b);
    r.y = ddim_one(a.y, b.y, c_recip, c_recipm1, c_a, c_b);
    r.z = ddim_one(a.z, b.z, c_recip, c_recipm1, c_a, c_b);
    r.w = ddim_one(a.w, b.w, c_recip, c_recipm1, c_a, c_b);
    xo[i] = r;
  }
}

// out / out_u: conditional and classifier-free rows of the denoiser output for the same clips; xo2: optional second copy of
// the updated latent (the inversion keeps every level AND feeds the next step from the session's contiguous rows)
__global__ void __launch_bounds__(kBlock) cfg_ddim_kernel(const float4* __restrict__ out, const float4* __restrict__ out_u,
                                                         const float4* __restrict__ x,
                                                         float4* __restrict__ xo, float4* __restrict__ xo2, float4* __restrict__ x0o,
                                                         const float* __restrict__ js, int64_t n4, int td4,
                                                         int d4, float w_c, float w_u, float c_recip,
                                                         float c_recipm1, float c_a, float c_b) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    int t = (int)((i % td4) / d4);
    float jc = js[t];
    float ju = 1.0f / jc;
    float4 oc = out[i], ou = out_u[i], a = x[i], p, r;
    p.x = cfg_one(oc.x, ou.x, jc, ju, w_c, w_u);
    p.y = cfg_one(oc.y, ou.y, jc, ju, w_c, w_u);
    p.z = cfg_one(oc.z, ou.z, jc, ju, w_c, w_u);
    p.w = cfg_one(oc.w, ou.w, jc, ju, w_c, w_u);
    r.x = ddim_one(a.x, p.x, c_recip, c_recipm1, c_a, c_b);
    r.y = ddim_one(a.y, p.y, c_recip, c_recipm1, c_a, c_b);
    r.z = ddim_one(a.z, p.z, c_recip, c_recipm1, c_a, c_b);
    r.w = ddim_one(a.w, p.w, c_recip, c_recipm1, c_a, c_b);
    xo[i] = r;
    if (xo2) xo2[i] = r;
    if (x0o) x0o[i] = p;
  }
}

// CFG mix + ancestral (DDPM) step: mean = c1 * x0 + c2 * x, sample = mean + sigma * noise (sigma = 0 at the last step).
__global__ void __launch_bounds__(kBlock) cfg_ddpm_kernel(const float4* __restrict__ out, const float4* __restrict__ x,
                                                         const float4* __restrict__ noise, float4* __restrict__ xo,
                                                         const float* __restrict__ js, int64_t n4, int td4, int d4,
                                                         float w_c, float w_u, float c1, float c2, float sigma) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    int t = (int)((i % td4) / d4);
    float jc = js[t];
    float ju = 1.0f / jc;
    float4 oc = out[i], ou = out[i + n4], a = x[i], z = noise[i], r;
    r.x = (c1 * cfg_one(oc.x, ou.x, jc, ju, w_c, w_u) + c2 * a.x) + sigma * z.x;
    r.y = (c1 * cfg_one(oc.y, ou.y, jc, ju, w_c, w_u) + c2 * a.y) + sigma * z.y;
    r.z = (c1 * cfg_one(oc.z, ou.z, jc, ju, w_c, w_u) + c2 * a.z) + sigma * z.z;
    r.w = (c1 * cfg_one(oc.w, ou.w, jc, ju, w_c, w_u) + c2 * a.w) + sigma * z.w;
    xo[i] = r;
  }
}

// One wave per token row: the wave first decides m = any(in_seq[row] != 0) with a ballot,
// then rewrites the row.  dim is a multiple of 4.
__global__ void __launch_bounds__(kBlock) inseq_replace_kernel(float* __restrict__ x,
                                                              const float* __restrict__ in_seq,
                                                              const float* __restrict__ noise, int rows,
                                                              int dim, float s_ab, float s_1mab) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * kBlock) >> 6;
  const int d4 = dim >> 2;
  for (int r = wave; r < rows; r += nwaves) {
    const float4* s = reinterpret_cast<const float4*>(in_seq + (int64_t)r * dim);
    bool nz = false;
    for (int j = lane; j < d4; j += 64) {
      float4 v = s[j];
      nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    if (__ballot(nz) == 0ull) continue;
    const float4* nn = reinterpret_cast<const float4*>(noise + (int64_t)r * dim);
    float4* xr = reinterpret_cast<float4*>(x + (int64_t)r * dim);
    for (int j = lane; j < d4; j += 64) {
      float4 v = s[j], e = nn[j], o;
      o.x = s_ab * v.x + s_1mab * e.x;
      o.y = s_ab * v.y + s_1mab * e.y;
      o.z = s_ab * v.z + s_1mab * e.z;
      o.w = s_ab * v.w + s_1mab * e.w;
      xr[j] = o;
    }
  }
}

__global__ void __launch_bounds__(kBlock) guidance_kernel(float* __restrict__ x, const float* __restrict__ in_seq,
                                                         int rows, int dim, int g_iter, float lr,
                                                         float two_over_numel) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * kBlock) >> 6;
  const int d4 = dim >> 2;
  for (int r = wave; r < rows; r += nwaves) {
    const float4* s = reinterpret_cast<const float4*>(in_seq + (int64_t)r * dim);
    bool nz = false;
    for (int j = lane; j < d4; j += 64) {
      float4 v = s[j];
      nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    if (__ballot(nz) == 0ull) continue;  // gradient is exactly 0 on unmasked rows
    float4* xr = reinterpret_cast<float4*>(x + (int64_t)r * dim);
    for (int j = lane; j < d4; j += 64) {
      float4 y = s[j], v = xr[j];
      for (int it = 0; it < g_iter; ++it) {
        v.x = v.x - lr * (two_over_numel * (v.x - y.x));
        v.y = v.y - lr * (two_over_numel * (v.y - y.y));
        v.z = v.z - lr * (two_over_numel * (v.z - y.z));
        v.w = v.w - lr * (two_over_numel * (v.w - y.w));
      }
      xr[j] = v;
    }
  }
}

// Everything between two denoiser forwards of a co-batched chain (sampler.cobatched_loop) in ONE launch, one wave per token row:
//   sampling rows:  x <- cfg_ddim(out_c, out_u, x) of this step; then, for the NEXT step, the insertion-guidance update and the
//                   in-sequence replacement on the rows its in_seq marks (cfg_ddim_kernel, guidance_kernel, inseq_replace_kernel,
//                   in that order, the same arithmetic operation for operation);
//   inverting rows: x <- cfg_ddim(...) of this step with the inversion's coefficients, and a second copy (the level kept).
// Four launches of ~15 us each beside a full chip become one (3-4 % of a chain's time).
__global__ void __launch_bounds__(kBlock) cobatch_glue_kernel(const rg_glue_args a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * kBlock) >> 6;
  const int rows_a = a.n_a * a.T, rows = (a.n_a + a.n_b) * a.T;
  const float two_over_numel = rg_tail::two_over_numel(a);
  for (int r = wave; r < rows; r += nwaves) {
    const bool ga = r < rows_a;
    rg_tail::glue_row(a, ga, ga ? r : r - rows_a, lane, two_over_numel);      // (row within its group)
  }
}

__global__ void __launch_bounds__(kBlock) splice_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                       int d4, int nrows, int src_row0, int dst_row0,
                                                       int hands_off, int nrep, int64_t src_rep_rows,
                                                       int64_t dst_rep_rows) {
  // rows [0,nrows): upper block; rows [nrows, 2*nrows): hands block (offset hands_off token rows)
  const int64_t per = (int64_t)2 * nrows * d4;
  const int64_t total = per * nrep;
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int s = (int)(i / per);
    const int64_t w = i % per;
    int r = (int)(w / d4), c = (int)(w % d4);
    int off = (r >= nrows) ? hands_off : 0;
    int rr = (r >= nrows) ? r - nrows : r;
    dst[(s * dst_rep_rows + dst_row0 + off + rr) * d4 + c] = src[(s * src_rep_rows + src_row0 + off + rr) * d4 + c];
  }
}

// Every exemplar splice of a lane's batch in ONE launch (rg_splice_many): entry i = blockIdx.y copies the rows of exemplar
// e[i] (upper block and hands block) into clip b[i] -- from inversion level `lvl` into the start noise, and, when the guidance
// target is given, from every level into it.  One launch instead of two per exemplar (96 behind a chain of 48 exemplars).
__global__ void __launch_bounds__(kBlock) splice_many_kernel(const rg_splice_table tab, const float4* __restrict__ inv,
                                                            float4* __restrict__ start_noise, float4* __restrict__ invl, int d4,
                                                            int T, int hands_off, int lvl, int S, int Ep, int B) {
  const int i = blockIdx.y;
  const int nrows = tab.nrows[i], e = tab.e[i], b = tab.b[i], r0 = tab.r0[i], q0 = tab.q0[i];
  const int64_t per = (int64_t)2 * nrows * d4;
  const int nslab = invl ? S + 1 : 1;                       // slab S (or 0): level lvl -> start noise
  const int64_t total = per * nslab;
  for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < total; k += (int64_t)gridDim.x * kBlock) {
    const int sl = (int)(k / per);
    const int64_t w = k % per;
    const int r = (int)(w / d4), c = (int)(w % d4);
    const int off = (r >= nrows) ? hands_off : 0;
    const int rr = (r >= nrows) ? r - nrows : r;
    const bool to_noise = sl == nslab - 1;
    const int level = to_noise ? lvl : sl;
    const float4 v = inv[(((int64_t)level * Ep + e) * T + r0 + off + rr) * d4 + c];
    if (to_noise) start_noise[(((int64_t)b) * T + q0 + off + rr) * d4 + c] = v;
    else invl[(((int64_t)sl * B + b) * T + q0 + off + rr) * d4 + c] = v;
  }
}

__global__ void __launch_bounds__(kBlock) gather_rows_kernel(const float4* __restrict__ table,
                                                            const int64_t* __restrict__ idx, float4* __restrict__ out,
                                                            int n, int d4) {
  int64_t total = (int64_t)n * d4;
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int r = (int)(i / d4), c = (int)(i % d4);
    out[i] = table[idx[r] * d4 + c];
  }
}

}  // namespace

extern "C" int rg_gather_rows(rg_handle* h, const float* table, const int64_t* idx, float* out, int n, int dim,
                              void* stream) {
  RG_REQUIRE(h, table && idx && out, "null pointer");
  RG_REQUIRE(h, n > 0 && dim > 0 && dim % 4 == 0, "bad shape");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rg_grid_1d((int64_t)n * dim / 4, kBlock)), dim3(kBlock), 0,
                     rg_stream(stream), (const float4*)table, idx, (float4*)out, n, dim / 4);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_ddim_update(rg_handle* h, const float* x, const float* x0, float* x_out, int64_t n,
                              float c_recip, float c_recipm1, float c_a, float c_b, void* stream) {
  RG_REQUIRE(h, x && x0 && x_out, "null pointer");
  RG_REQUIRE(h, n > 0 && n % 4 == 0, "n must be a positive multiple of 4");
  int64_t n4 = n / 4;
  hipLaunchKernelGGL(ddim_update_kernel, dim3(rg_grid_1d(n4, kBlock)), dim3(kBlock), 0, rg_stream(stream),
                     (const float4*)x, (const float4*)x0, (float4*)x_out, n4, c_recip, c_recipm1, c_a, c_b);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_cfg_ddim_update(rg_handle* h, const float* out, const float* x, float* x_out, float* x0_out,
                                  const float* js, int B, int T, int D, float w_c, float w_u, float c_recip,
                                  float c_recipm1, float c_a, float c_b, void* stream) {
  RG_REQUIRE(h, out && x && x_out && js, "null pointer");
  RG_REQUIRE(h, B > 0 && T > 0 && D > 0 && D % 4 == 0, "bad shape");
  int64_t n4 = (int64_t)B * T * D / 4;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(rg_grid_1d(n4, kBlock)), dim3(kBlock), 0, rg_stream(stream),
                     (const float4*)out, (const float4*)out + n4, (const float4*)x, (float4*)x_out, (float4*)nullptr,
                     (float4*)x0_out, js, n4, T * D / 4, D / 4, w_c, w_u, c_recip, c_recipm1, c_a, c_b);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_cfg_ddim_update_rows(rg_handle* h, const float* out_cond, const float* out_uncond, const float* x,
                                       float* x_out, float* x_out2, const float* js, int B, int T, int D, float w_c, float w_u,
                                       float c_recip, float c_recipm1, float c_a, float c_b, void* stream) {
  RG_REQUIRE(h, out_cond && out_uncond && x && x_out && js, "null pointer");
  RG_REQUIRE(h, B > 0 && T > 0 && D > 0 && D % 4 == 0, "bad shape");
  int64_t n4 = (int64_t)B * T * D / 4;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(rg_grid_1d(n4, kBlock)), dim3(kBlock), 0, rg_stream(stream),
                     (const float4*)out_cond, (const float4*)out_uncond, (const float4*)x, (float4*)x_out, (float4*)x_out2,
                     (float4*)nullptr, js, n4, T * D / 4, D / 4, w_c, w_u, c_recip, c_recipm1, c_a, c_b);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_cfg_ddpm_update(rg_handle* h, const float* out, const float* x, const float* noise, float* x_out,
                                  const float* js, int B, int T, int D, float w_c, float w_u, float c1, float c2,
                                  float sigma, void* stream) {
  RG_REQUIRE(h, out && x && noise && x_out && js, "null pointer");
  RG_REQUIRE(h, B > 0 && T > 0 && D > 0 && D % 4 == 0, "bad shape");
  int64_t n4 = (int64_t)B * T * D / 4;
  hipLaunchKernelGGL(cfg_ddpm_kernel, dim3(rg_grid_1d(n4, kBlock)), dim3(kBlock), 0, rg_stream(stream), (const float4*)out,
                     (const float4*)x, (const float4*)noise, (float4*)x_out, js, n4, T * D / 4, D / 4, w_c, w_u, c1, c2, sigma);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_cobatch_glue(rg_handle* h, const rg_glue_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  const rg_glue_args& a = *args_host;
  RG_REQUIRE(h, a.out_c_a && a.out_u_a && a.x_a && a.out_c_b && a.out_u_b && a.x_b && a.js, "null pointer");
  RG_REQUIRE(h, a.n_a > 0 && a.n_b > 0 && a.T > 0 && a.D > 0 && a.D % 4 == 0 && a.g_iter_next >= 0, "bad shape");
  RG_REQUIRE(h, !a.in_seq_next || a.noise_next, "in_seq_next needs noise_next");
  const int rows = (a.n_a + a.n_b) * a.T;
  hipLaunchKernelGGL(cobatch_glue_kernel, dim3(rg_grid_1d((int64_t)rows * 64, kBlock)), dim3(kBlock), 0, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_inseq_replace(rg_handle* h, float* x, const float* in_seq, const float* noise, int rows,
                                int dim, float s_ab, float s_1mab, void* stream) {
  RG_REQUIRE(h, x && in_seq && noise, "null pointer");
  RG_REQUIRE(h, rows > 0 && dim > 0 && dim % 4 == 0, "bad shape");
  int grid = rg_grid_1d((int64_t)rows * 64, kBlock);
  hipLaunchKernelGGL(inseq_replace_kernel, dim3(grid), dim3(kBlock), 0, rg_stream(stream), x, in_seq, noise,
                     rows, dim, s_ab, s_1mab);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_guidance_update(rg_handle* h, float* x, const float* in_seq, int rows, int dim, int g_iter,
                                  float lr, void* stream) {
  RG_REQUIRE(h, x && in_seq, "null pointer");
  RG_REQUIRE(h, rows > 0 && dim > 0 && dim % 4 == 0 && g_iter >= 0, "bad shape");
  if (g_iter == 0) return RG_OK;
  int grid = rg_grid_1d((int64_t)rows * 64, kBlock);
  float two_over_numel = 2.0f / ((float)rows * (float)dim);
  hipLaunchKernelGGL(guidance_kernel, dim3(grid), dim3(kBlock), 0, rg_stream(stream), x, in_seq, rows, dim,
                     g_iter, lr, two_over_numel);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_splice_rows_rep(rg_handle* h, const float* src, float* dst, int T, int D, int n_lat, int b_src,
                                  int b_dst, int r0, int r1, int q0, int q1, int nrep, int src_rep_stride,
                                  int dst_rep_stride, void* stream) {
  RG_REQUIRE(h, src && dst, "null pointer");
  RG_REQUIRE(h, D % 4 == 0 && r1 - r0 == q1 - q0 && r0 >= 0 && q0 >= 0 && r1 <= n_lat && q1 <= n_lat && nrep > 0,
             "bad row ranges");
  int nrows = r1 - r0;
  if (nrows <= 0) return RG_OK;
  int d4 = D / 4;
  hipLaunchKernelGGL(splice_kernel, dim3(rg_grid_1d((int64_t)2 * nrows * d4 * nrep, kBlock)), dim3(kBlock), 0,
                     rg_stream(stream), (const float4*)src, (float4*)dst, d4, nrows, b_src * T + r0,
                     b_dst * T + q0, n_lat + 1, nrep, (int64_t)src_rep_stride * T, (int64_t)dst_rep_stride * T);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_splice_many(rg_handle* h, const rg_splice_table* tab_host, const float* inv, float* start_noise, float* invl,
                              int T, int D, int n_lat, int lvl, int S, int Ep, int B, void* stream) {
  RG_REQUIRE(h, tab_host && inv && start_noise, "null pointer");
  const rg_splice_table& tab = *tab_host;
  RG_REQUIRE(h, tab.n >= 0 && tab.n <= RG_SPLICE_MAX && D % 4 == 0 && S >= 1 && lvl >= 0 && lvl < S && Ep >= 1 && B >= 1, "bad shape");
  int most = 0;
  for (int i = 0; i < tab.n; ++i) {
    RG_REQUIRE(h, tab.e[i] >= 0 && tab.e[i] < Ep && tab.b[i] >= 0 && tab.b[i] < B && tab.nrows[i] >= 0 && tab.r0[i] >= 0 && tab.q0[i] >= 0
                   && tab.r0[i] + tab.nrows[i] <= n_lat && tab.q0[i] + tab.nrows[i] <= n_lat, "bad row ranges");
    most = tab.nrows[i] > most ? tab.nrows[i] : most;
  }
  if (tab.n == 0 || most == 0) return RG_OK;
  const int d4 = D / 4;
  const int64_t work = (int64_t)2 * most * d4 * (invl ? S + 1 : 1);
  hipLaunchKernelGGL(splice_many_kernel, dim3(rg_grid_1d(work, kBlock) > 64 ? 64 : rg_grid_1d(work, kBlock), tab.n), dim3(kBlock), 0,
                     rg_stream(stream), tab, (const float4*)inv, (float4*)start_noise, (float4*)invl, d4, T, n_lat + 1, lvl, S, Ep, B);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_splice_rows(rg_handle* h, const float* src, float* dst, int T, int D, int n_lat, int b_src,
                              int b_dst, int r0, int r1, int q0, int q1, void* stream) {
  return rg_splice_rows_rep(h, src, dst, T, D, n_lat, b_src, b_dst, r0, r1, q0, q1, 1, 0, 0, stream);
}
