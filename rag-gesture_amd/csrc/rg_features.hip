// Conditioning feature extraction per window (tools/longform_synthesis.py:64-94): the small kernels the
// wav2vec2-base and BERT-base forwards need besides rg_gemm / rg_mha_bf16 / rg_layernorm_res -- all HBM-bound byte movers.
//   rg_embed_sum3          BERT embeddings: word[id] + position[i] + token_type[0]
//   rg_time_groupnorm_gelu wav2vec2 feature extractor layer 0: GroupNorm(512 groups = per channel, over time) -> GELU -> bf16
//   rg_im2col_grouped      wav2vec2 positional convolution (k = 128, 16 groups, padding 64): per-group patch matrix, bf16
#include "rg_common.h"

namespace {

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

__global__ void __launch_bounds__(256) embed_sum3_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0,
                                                        float* __restrict__ out, int L, int dim) {
  const int64_t total = (int64_t)L * dim;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i / dim), c = (int)(i % dim);
    out[i] = word[(size_t)ids[t] * dim + c] + type0[c] + pos[(size_t)t * dim + c];   // HF order: inputs + token_type, + position
  }
}

// column sums of x [T][C] (or of (x - mean)^2 when mean != null) into acc[C]; thread -> column, block -> 64 rows
__global__ void __launch_bounds__(256) col_reduce_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        float* __restrict__ acc, int T, int C, float inv_t) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int t0 = blockIdx.x * 64, t1 = min(T, t0 + 64);
  const float m = mean ? mean[c] * inv_t : 0.f;
  float s = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float d = x[(size_t)t * C + c] - m;
    s += mean ? d * d : d;
  }
  atomicAdd(acc + c, s);
}

__global__ void __launch_bounds__(256) groupnorm_gelu_kernel(const float* __restrict__ x, const float* __restrict__ sum,
                                                            const float* __restrict__ sq, const float* __restrict__ g,
                                                            const float* __restrict__ b, unsigned short* __restrict__ out,
                                                            float* __restrict__ out32, int64_t total, int C, float inv_t,
                                                            float eps) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float mean = sum[c] * inv_t;
    const float rstd = 1.0f / sqrtf(sq[c] * inv_t + eps);
    const float v = gelu_erf((x[i] - mean) * rstd * g[c] + b[c]);
    if (out) out[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
    if (out32) out32[i] = v;
  }
}

// out[g][t][k * Cg + ci] = xpad[t + k][g * Cg + ci], xpad = x padded with `pad` zero rows in front (and behind)
__global__ void __launch_bounds__(256) im2col_grouped_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                            int T, int C, int groups, int ksize, int pad) {
  const int Cg = C / groups;
  const int64_t row_len = (int64_t)ksize * Cg, total = (int64_t)groups * T * row_len;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % Cg);
    const int k = (int)((i / Cg) % ksize);
    const int t = (int)((i / row_len) % T);
    const int g = (int)(i / (row_len * T));
    const int ts = t + k - pad;
    const float v = (ts >= 0 && ts < T) ? x[(size_t)ts * C + g * Cg + ci] : 0.f;
    out[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

}  // namespace

extern "C" int rg_embed_sum3(rg_handle* h, const int64_t* ids, const float* word, const float* pos, const float* type0,
                             float* out, int L, int dim, void* stream) {
  RG_REQUIRE(h, ids && word && pos && type0 && out, "null pointer");
  RG_REQUIRE(h, L > 0 && dim > 0, "bad shape");
  hipLaunchKernelGGL(embed_sum3_kernel, dim3(rg_grid_1d((int64_t)L * dim, 256)), dim3(256), 0, rg_stream(stream), ids, word, pos,
                     type0, out, L, dim);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_time_groupnorm_gelu(rg_handle* h, const float* x, const float* gamma, const float* beta, void* out_bf16,
                                      float* out_f32, int T, int C, float eps, float* workspace, void* stream) {
  RG_REQUIRE(h, x && gamma && beta && (out_bf16 || out_f32) && workspace, "null pointer");
  RG_REQUIRE(h, T > 0 && C > 0, "bad shape");
  hipStream_t s = rg_stream(stream);
  if (hipMemsetAsync(workspace, 0, sizeof(float) * 2 * C, s) != hipSuccess) return RG_ERR_HIP;
  const dim3 grid((T + 63) / 64, (C + 255) / 256);
  const float inv_t = 1.0f / (float)T;
  hipLaunchKernelGGL(col_reduce_kernel, grid, dim3(256), 0, s, x, (const float*)nullptr, workspace, T, C, inv_t);
  hipLaunchKernelGGL(col_reduce_kernel, grid, dim3(256), 0, s, x, (const float*)workspace, workspace + C, T, C, inv_t);
  hipLaunchKernelGGL(groupnorm_gelu_kernel, dim3(rg_grid_1d((int64_t)T * C, 256)), dim3(256), 0, s, x, workspace, workspace + C,
                     gamma, beta, reinterpret_cast<unsigned short*>(out_bf16), out_f32, (int64_t)T * C, C, inv_t, eps);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_im2col_grouped(rg_handle* h, const float* x, void* out_bf16, int T, int C, int groups, int ksize, int pad,
                                 void* stream) {
  RG_REQUIRE(h, x && out_bf16, "null pointer");
  RG_REQUIRE(h, T > 0 && C > 0 && groups > 0 && C % groups == 0 && ksize > 0 && pad >= 0, "bad shape");
  hipLaunchKernelGGL(im2col_grouped_kernel, dim3(rg_grid_1d((int64_t)T * C * ksize, 256)), dim3(256), 0, rg_stream(stream), x,
                     reinterpret_cast<unsigned short*>(out_bf16), T, C, groups, ksize, pad);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}
