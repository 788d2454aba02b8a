// Kernels of the four body-part TransformerVAEs (everything that is not an nn.Linear; the
// linears go through rg_gemm) and the SMPL-X rotation conversions around them.
//
//  rg_mha            : torch.nn.MultiheadAttention core, softmax(Q K^T / sqrt(hd)) V, short sequences
//                      (reference: utils/detr_utils.py:364-366, 427-433 via nn.MultiheadAttention)
//  rg_layernorm      : nn.LayerNorm over the last dim (detr_utils.py:368,371,...)
//  rg_add_rows       : out = a + b[row % period]   (positional embeddings, `with_pos_embed`)
//  rg_vae_reparam    : z = mu + exp(0.5*logvar) * eps scattered into the [B,43,D] latent
//                      (gesture_vae.py:173-193, diffusion_transformer.py:239-254)
//  rg_aa_to_6d / rg_6d_to_aa : rotation_conversions.py:416-550 (quaternion route of the old
//                      PyTorch3D fork, incl. the 1e-6 small-angle branches)
#include "rg_common.h"

namespace {

// ------------------------------------------------------------------------------ MHA
// grid = (B*H, ceil(Sq/QB)); 256 threads; K/V of one (batch, head) staged in LDS
// (K rows padded by one float: lanes read different keys at the same d).
constexpr int QB = 32;

__global__ void __launch_bounds__(256) mha_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                 int ldk, const float* __restrict__ v, int ldv, float* __restrict__ o,
                                                 int ldo, int H, int Sq, int Sk, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int q0 = blockIdx.y * QB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kst = hd + 1;
  float* sK = sm;                        // [Sk][hd+1]
  float* sV = sK + Sk * kst;             // [Sk][hd]
  float* sQ = sV + Sk * hd;              // [4 waves][hd]
  float* sP = sQ + 4 * hd;               // [4 waves][Sk]
  for (int i = threadIdx.x; i < Sk * hd; i += 256) {
    const int j = i / hd, d = i % hd;
    sK[j * kst + d] = k[((size_t)b * Sk + j) * ldk + h * hd + d];
    sV[j * hd + d] = v[((size_t)b * Sk + j) * ldv + h * hd + d];
  }
  __syncthreads();
  const int qend = min(q0 + QB, Sq);
  float* myQ = sQ + wave * hd;
  float* myP = sP + wave * Sk;
  for (int r = q0 + wave; r < qend; r += 4) {
    const float* qr = q + ((size_t)b * Sq + r) * ldq + h * hd;
    for (int d = lane; d < hd; d += 64) myQ[d] = qr[d] * scale;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float sc[3];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int j = lane + 64 * t;
      float s = -INFINITY;
      if (j < Sk) {
        s = 0.f;
        const float* kr = sK + j * kst;
        for (int d = 0; d < hd; ++d) s = fmaf(myQ[d], kr[d], s);
      }
      sc[t] = s;
      mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int j = lane + 64 * t;
      const float e = (j < Sk) ? expf(sc[t] - mx) : 0.f;
      sc[t] = e;
      sum += e;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int j = lane + 64 * t;
      if (j < Sk) myP[j] = sc[t] * inv;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float* orow = o + ((size_t)b * Sq + r) * ldo + h * hd;
    for (int d = lane; d < hd; d += 64) {
      float acc = 0.f;
      for (int j = 0; j < Sk; ++j) acc = fmaf(myP[j], sV[j * hd + d], acc);
      orow[d] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// ------------------------------------------------------------------------------ MHA on the matrix cores
// bf16-path variant (the VAE linears around it already round their operands to bf16): K and V^T of
// one (batch, head) are staged once per workgroup in LDS as bf16; every wave owns 16 query rows.
//   S^T = K Q^T      v_mfma_f32_16x16x32_bf16, A = K rows from LDS (b128, row stride hd+8: conflict
//                    free), B = Q (hi + lo bf16 split, registers);  C layout: lane -> query lane&15,
//                    keys 16*nb + 4*(lane>>4) + e  ->  the softmax over keys is in-register plus two
//                    cross-lane steps (xor 16, 32), and
//   O   = P V        needs NO transposition of P: a lane's 4+4 probabilities of key blocks 2s, 2s+1
//                    are exactly an A fragment of the k-step s if V^T is read with the same key
//                    permutation (two b64 LDS reads per fragment).  P is split hi + lo as well.
typedef __attribute__((ext_vector_type(8))) __bf16 mh_bf16x8;
typedef __attribute__((ext_vector_type(4))) float mh_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int mh_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int mh_u32x2;

constexpr int MH_MAX_SK = 192;   // key capacity of the default instantiation (VAE stacks, BERT windows)
constexpr int MH_BIG_SK = 512;   // wav2vec2 windows (499 frames): 32 score blocks in registers, 140 KiB of K / V^T in LDS

__device__ __forceinline__ unsigned short mh_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float mh_bf16_val(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ unsigned int mh_pack(float a, float b) {
  return (unsigned int)mh_bf16_bits(a) | ((unsigned int)mh_bf16_bits(b) << 16);
}

// HD_ = head dim padded to a multiple of 32 (the MFMA k-step); HDR = real head dim (16 -> zero padded)
template <int HD_, int HDR, int MAXSK = MH_MAX_SK>
__device__ __forceinline__ void mha_mfma_body(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                              int ldk, const float* __restrict__ v, int ldv,
                                              float* __restrict__ o, int ldo, int H, int Sq, int Sk, float scale,
                                              int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  constexpr int KST = HD_ + 8;                 // bf16 elements per K row (pad: 16 key rows hit 16 bank groups)
  constexpr int KS = HD_ / 32;                 // k-steps of S^T = K Q^T
  constexpr int ND = HDR / 16;                 // 16-column blocks of O
  constexpr int NB = MAXSK / 16;               // key blocks held in registers
  const int Skp = (Sk + 31) & ~31;             // keys padded to whole PV k-steps (P = 0, V^T = 0 there)
  const int VST = Skp + 8;                     // bf16 elements per V^T row
  unsigned short* sK = reinterpret_cast<unsigned short*>(smraw);           // [Skp][KST]
  unsigned short* sVt = sK + (size_t)Skp * KST;                            // [HD_][VST]
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- stage K (row major) and V^T as bf16
  for (int i = threadIdx.x; i < Skp * (HD_ / 4); i += 256) {
    const int j = i / (HD_ / 4), d4 = (i % (HD_ / 4)) * 4;
    mh_f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (j < Sk && d4 < HDR) {
      kv = *reinterpret_cast<const mh_f32x4*>(k + ((size_t)b * Sk + j) * ldk + h * HDR + d4);
      vv = *reinterpret_cast<const mh_f32x4*>(v + ((size_t)b * Sk + j) * ldv + h * HDR + d4);
    }
    *reinterpret_cast<mh_u32x2*>(sK + (size_t)j * KST + d4) = mh_u32x2{mh_pack(kv[0], kv[1]), mh_pack(kv[2], kv[3])};
#pragma unroll
    for (int e = 0; e < 4; ++e) sVt[(size_t)(d4 + e) * VST + j] = mh_bf16_bits(vv[e]);
  }
  __syncthreads();

  const int q0 = blockIdx.y * 64 + wave * 16;
  if (q0 >= Sq) return;
  // ---- Q fragments (B operand): lane -> query q0 + l15, d = 32*ks + 8*g .. +7, scaled, hi/lo split
  mh_bf16x8 qh[KS], ql[KS];
  {
    const int qr = min(q0 + l15, Sq - 1);
    const float* qp = q + ((size_t)b * Sq + qr) * ldq + h * HDR + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      mh_f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (32 * ks + 8 * g < HDR) {
        a = *reinterpret_cast<const mh_f32x4*>(qp + 32 * ks);
        c = *reinterpret_cast<const mh_f32x4*>(qp + 32 * ks + 4);
      }
      float x[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale,
                    c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = x[e] - mh_bf16_val(x[e]);
      qh[ks] = __builtin_bit_cast(mh_bf16x8, mh_u32x4{mh_pack(x[0], x[1]), mh_pack(x[2], x[3]), mh_pack(x[4], x[5]), mh_pack(x[6], x[7])});
      ql[ks] = __builtin_bit_cast(mh_bf16x8, mh_u32x4{mh_pack(r[0], r[1]), mh_pack(r[2], r[3]), mh_pack(r[4], r[5]), mh_pack(r[6], r[7])});
    }
  }
  // ---- S^T blocks
  const int nblk = Skp / 16;
  mh_f32x4 sc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    sc[nb] = mh_f32x4{0.f, 0.f, 0.f, 0.f};
    if (nb < nblk) {
      const unsigned short* kr = sK + (size_t)(16 * nb + l15) * KST + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const mh_bf16x8 kf = *reinterpret_cast<const mh_bf16x8*>(kr + 32 * ks);
        sc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qh[ks], sc[nb], 0, 0, 0);
        sc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql[ks], sc[nb], 0, 0, 0);
      }
    }
  }
  // ---- softmax over keys for query l15: keys 16*nb + 4*g + e
  float mx = -INFINITY;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool valid = nb < nblk && 16 * nb + 4 * g + e < Sk;
      sc[nb][e] = valid ? sc[nb][e] : -INFINITY;
      mx = fmaxf(mx, sc[nb][e]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = (sc[nb][e] == -INFINITY) ? 0.f : expf(sc[nb][e] - mx);
      sc[nb][e] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  // ---- O = P V
  mh_f32x4 oc[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) oc[nd] = mh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s2 = 0; s2 < NB / 2; ++s2) {
    if (2 * s2 < nblk) {
      float p[8], r[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[e] = sc[2 * s2][e] * inv;
        p[4 + e] = sc[2 * s2 + 1][e] * inv;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = p[e] - mh_bf16_val(p[e]);
      const mh_bf16x8 ph = __builtin_bit_cast(mh_bf16x8, mh_u32x4{mh_pack(p[0], p[1]), mh_pack(p[2], p[3]), mh_pack(p[4], p[5]), mh_pack(p[6], p[7])});
      const mh_bf16x8 pl = __builtin_bit_cast(mh_bf16x8, mh_u32x4{mh_pack(r[0], r[1]), mh_pack(r[2], r[3]), mh_pack(r[4], r[5]), mh_pack(r[6], r[7])});
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        const unsigned short* vr = sVt + (size_t)(16 * nd + l15) * VST + 32 * s2 + 4 * g;
        const mh_u32x2 v0 = *reinterpret_cast<const mh_u32x2*>(vr);
        const mh_u32x2 v1 = *reinterpret_cast<const mh_u32x2*>(vr + 16);
        const mh_bf16x8 vf = __builtin_bit_cast(mh_bf16x8, mh_u32x4{v0[0], v0[1], v1[0], v1[1]});
        oc[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vf, oc[nd], 0, 0, 0);
        oc[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vf, oc[nd], 0, 0, 0);
      }
    }
  }
  // ---- store: lane -> rows q0 + 4*g + e, column 16*nd + l15
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = q0 + 4 * g + e;
    if (r < Sq) {
      if (out_bf16) {   // the output only feeds the out-projection GEMM: hand it over as its bf16 A operand
        unsigned short* orow = reinterpret_cast<unsigned short*>(o) + ((size_t)b * Sq + r) * ldo + h * HDR;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) orow[16 * nd + l15] = mh_bf16_bits(oc[nd][e]);
      } else {
        float* orow = o + ((size_t)b * Sq + r) * ldo + h * HDR;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) orow[16 * nd + l15] = oc[nd][e];
      }
    }
  }
}

template <int HD_, int HDR, int MAXSK = MH_MAX_SK>
__global__ void __launch_bounds__(256) mha_mfma_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                      int ldk, const float* __restrict__ v, int ldv,
                                                      float* __restrict__ o, int ldo, int H, int Sq, int Sk, float scale,
                                                      int out_bf16) {
  mha_mfma_body<HD_, HDR, MAXSK>(q, ldq, k, ldk, v, ldv, o, ldo, H, Sq, Sk, scale, out_bf16);
}

// up to 4 independent problems of one shape in a launch (blockIdx.z picks the pointers): the four body-part VAEs
struct mha_group { const float* q[4]; const float* k[4]; const float* v[4]; float* o[4]; };
template <int HD_, int HDR, int MAXSK = MH_MAX_SK>
__global__ void __launch_bounds__(256) mha_mfma_group_kernel(const mha_group g, int ldq, int ldk, int ldv, int ldo, int H, int Sq,
                                                            int Sk, float scale, int out_bf16) {
  const int z = blockIdx.z;
  mha_mfma_body<HD_, HDR, MAXSK>(g.q[z], ldq, g.k[z], ldk, g.v[z], ldv, g.o[z], ldo, H, Sq, Sk, scale, out_bf16);
}

template <int HD_, int HDR, int MAXSK = MH_MAX_SK>
static int launch_mha_mfma_group(const mha_group& g, int n, int ldq, int ldk, int ldv, int ldo, int B, int H, int Sq, int Sk,
                                 int out_bf16, hipStream_t s) {
  const int Skp = (Sk + 31) & ~31;
  const size_t lds = ((size_t)Skp * (HD_ + 8) + (size_t)HD_ * (Skp + 8)) * sizeof(unsigned short);
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (mha_mfma_group_kernel<HD_, HDR, MAXSK>), 160 * 1024);
  hipLaunchKernelGGL((mha_mfma_group_kernel<HD_, HDR, MAXSK>), dim3(B * H, (Sq + 63) / 64, n), dim3(256), lds, s, g, ldq, ldk, ldv, ldo,
                     H, Sq, Sk, 1.0f / sqrtf((float)HDR), out_bf16);
  return 0;
}

template <int HD_, int HDR, int MAXSK = MH_MAX_SK>
static int launch_mha_mfma(rg_handle* h, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           float* o, int ldo, int B, int H, int Sq, int Sk, int out_bf16, hipStream_t s) {
  const int Skp = (Sk + 31) & ~31;
  const size_t lds = ((size_t)Skp * (HD_ + 8) + (size_t)HD_ * (Skp + 8)) * sizeof(unsigned short);
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (mha_mfma_kernel<HD_, HDR, MAXSK>), 160 * 1024);
  hipLaunchKernelGGL((mha_mfma_kernel<HD_, HDR, MAXSK>), dim3(B * H, (Sq + 63) / 64), dim3(256), lds, s, q, ldq, k, ldk, v, ldv, o, ldo,
                     H, Sq, Sk, 1.0f / sqrtf((float)HDR), out_bf16);
  return 0;
}

// ------------------------------------------------------------------------------ LayerNorm, one wave per row
__device__ __forceinline__ void layernorm_body(const float* __restrict__ x, const float* __restrict__ g,
                                               const float* __restrict__ b, float* __restrict__ out, int rows,
                                               int dim, float eps, unsigned short* __restrict__ out_bf16) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * dim;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += xr[i];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)dim;
  float ss = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float d = xr[i] - mean;
    ss += d * d;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) ss += __shfl_xor(ss, off);
  const float rstd = rsqrtf(ss / (float)dim + eps);
  float* orow = out + (size_t)row * dim;
  for (int i = lane; i < dim; i += 64) {
    const float v = (xr[i] - mean) * rstd * g[i] + b[i];
    orow[i] = v;
    if (out_bf16) out_bf16[(size_t)row * dim + i] = __builtin_bit_cast(unsigned short, (__bf16)v);   // next GEMM's A operand
  }
}

__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ b, float* __restrict__ out, int rows,
                                                       int dim, float eps, unsigned short* __restrict__ out_bf16) {
  layernorm_body(x, g, b, out, rows, dim, eps, out_bf16);
}
struct ln_group { const float* x[4]; const float* g[4]; const float* b[4]; float* out[4]; unsigned short* o16[4]; };
__global__ void __launch_bounds__(256) layernorm_group_kernel(const ln_group p, int rows, int dim, float eps) {
  const int z = blockIdx.y;
  layernorm_body(p.x[z], p.g[z], p.b[z], p.out[z], rows, dim, eps, p.o16[z]);
}

// LN(x + r) with a caller-supplied eps (post-norm encoder layers of BERT / wav2vec2); one wave per row
__global__ void __launch_bounds__(256) layernorm_res_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ g, const float* __restrict__ b,
                                                           float* __restrict__ out, int rows, int dim, float eps,
                                                           unsigned short* __restrict__ out_bf16) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * dim;
  const float* rr = res ? res + (size_t)row * dim : nullptr;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += xr[i] + (rr ? rr[i] : 0.f);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)dim;
  float ss = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float d = xr[i] + (rr ? rr[i] : 0.f) - mean;
    ss += d * d;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) ss += __shfl_xor(ss, off);
  const float rstd = 1.0f / sqrtf(ss / (float)dim + eps);
  float* orow = out + (size_t)row * dim;
  for (int i = lane; i < dim; i += 64) {
    const float v = (xr[i] + (rr ? rr[i] : 0.f) - mean) * rstd * g[i] + b[i];
    orow[i] = v;
    if (out_bf16) out_bf16[(size_t)row * dim + i] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

__device__ __forceinline__ void add_rows_body(const float4* __restrict__ a, const float4* __restrict__ b,
                                              float4* __restrict__ out, int64_t n4, int64_t period4) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 x = a[i], y = b[i % period4];
    out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

__global__ void __launch_bounds__(256) add_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                      float4* __restrict__ out, int64_t n4, int64_t period4) {
  add_rows_body(a, b, out, n4, period4);
}
struct ptr3_group { const float* a[4]; const float* b[4]; float* out[4]; };
__global__ void __launch_bounds__(256) add_rows_group_kernel(const ptr3_group p, int64_t n4, int64_t period4) {
  const int z = blockIdx.y;
  add_rows_body((const float4*)p.a[z], (const float4*)p.b[z], (float4*)p.out[z], n4, period4);
}

// copy `nrows_per` rows per group from src (row stride ld_src, rows_src_per rows per group, starting at
// src_row0 inside the group) to dst (rows_dst_per rows per group, starting at dst_row0).
__device__ __forceinline__ void copy_rows_body(const float* __restrict__ src, float* __restrict__ dst,
                                               int groups, int nrows_per, int dim, int rows_src_per,
                                               int src_row0, int rows_dst_per, int dst_row0) {
  const int64_t total = (int64_t)groups * nrows_per * dim;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % dim);
    const int64_t r = i / dim;
    const int g = (int)(r / nrows_per), rr = (int)(r % nrows_per);
    dst[((int64_t)g * rows_dst_per + dst_row0 + rr) * dim + c] =
        src[((int64_t)g * rows_src_per + src_row0 + rr) * dim + c];
  }
}

__global__ void __launch_bounds__(256) copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       int groups, int nrows_per, int dim, int rows_src_per,
                                                       int src_row0, int rows_dst_per, int dst_row0) {
  copy_rows_body(src, dst, groups, nrows_per, dim, rows_src_per, src_row0, rows_dst_per, dst_row0);
}
__global__ void __launch_bounds__(256) copy_rows_group_kernel(const ptr3_group p, int groups, int nrows_per, int dim,
                                                             int rows_src_per, int src_row0, int rows_dst_per, int dst_row0) {
  const int z = blockIdx.y;
  copy_rows_body(p.a[z], p.out[z], groups, nrows_per, dim, rows_src_per, src_row0, rows_dst_per, dst_row0);
}

// z[b, row_off + c, :] = mu + exp(logvar)^0.5 * eps  with mu = enc[(b*n_chunks+c), 0, :], logvar = enc[.., 1, :]
__global__ void __launch_bounds__(256) vae_reparam_kernel(const float* __restrict__ enc, int seq, const float* __restrict__ eps,
                                                         float* __restrict__ latent, int B, int n_chunks, int D, int T,
                                                         int row_off) {
  const int64_t total = (int64_t)B * n_chunks * D;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t bc = i / D;
    const int b = (int)(bc / n_chunks), c = (int)(bc % n_chunks);
    const float mu = enc[(bc * seq + 0) * D + d];
    const float lv = enc[(bc * seq + 1) * D + d];
    const float sd = sqrtf(expf(lv));  // logvar.exp().pow(0.5)
    latent[((int64_t)b * T + row_off + c) * D + d] = mu + sd * eps[i];
  }
}

// ------------------------------------------------------------------------------ rotations
#pragma clang fp contract(off)
__device__ __forceinline__ void aa_to_6d_one(const float* a, float* o) {
  const float x = a[0], y = a[1], z = a[2];
  const float angle = sqrtf(x * x + y * y + z * z);
  const float half = 0.5f * angle;
  const float s = (fabsf(angle) < 1e-6f) ? (0.5f - (angle * angle) / 48.0f) : (sinf(half) / angle);
  const float r = cosf(half), i = x * s, j = y * s, k = z * s;
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  o[0] = 1 - two_s * (j * j + k * k);
  o[1] = two_s * (i * j - k * r);
  o[2] = two_s * (i * k + j * r);
  o[3] = two_s * (i * j + k * r);
  o[4] = 1 - two_s * (i * i + k * k);
  o[5] = two_s * (j * k - i * r);
}

__device__ __forceinline__ float sqrt_pos(float v) { return v > 0.f ? sqrtf(v) : 0.f; }
__device__ __forceinline__ float copysign_ref(float a, float b) { return ((a < 0.f) != (b < 0.f)) ? -a : a; }

__device__ __forceinline__ void sixd_to_aa_one(const float* d6, float* o) {
  // rotation_6d_to_matrix: Gram-Schmidt, F.normalize eps 1e-12
  float a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
  float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float dot = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float b2[3] = {a2[0] - dot * b1[0], a2[1] - dot * b1[1], a2[2] - dot * b1[2]};
  float n2 = fmaxf(sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]), 1e-12f);
  b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
  float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  // matrix rows are b1, b2, b3: m[r][c]
  const float m00 = b1[0], m11 = b2[1], m22 = b3[2];
  const float o0 = 0.5f * sqrt_pos(1 + m00 + m11 + m22);
  const float qx = 0.5f * sqrt_pos(1 + m00 - m11 - m22);
  const float qy = 0.5f * sqrt_pos(1 - m00 + m11 - m22);
  const float qz = 0.5f * sqrt_pos(1 - m00 - m11 + m22);
  const float o1 = copysign_ref(qx, b3[1] - b2[2]);  // m21 - m12
  const float o2 = copysign_ref(qy, b1[2] - b3[0]);  // m02 - m20
  const float o3 = copysign_ref(qz, b2[0] - b1[1]);  // m10 - m01
  const float norm = sqrtf(o1 * o1 + o2 * o2 + o3 * o3);
  const float half = atan2f(norm, o0);
  const float angle = 2.0f * half;
  const float s = (fabsf(angle) < 1e-6f) ? (0.5f - (angle * angle) / 48.0f) : (sinf(half) / angle);
  o[0] = o1 / s; o[1] = o2 / s; o[2] = o3 / s;
}

__global__ void __launch_bounds__(256) aa_to_6d_kernel(const float* __restrict__ aa, int ld_in, float* __restrict__ out,
                                                      int ld_out, int col_off, int rows, int joints) {
  const int64_t total = (int64_t)rows * joints;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / joints), j = (int)(i % joints);
    float a[3], o[6];
    const float* p = aa + (size_t)r * ld_in + j * 3;
    a[0] = p[0]; a[1] = p[1]; a[2] = p[2];
    aa_to_6d_one(a, o);
    float* q = out + (size_t)r * ld_out + col_off + j * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) q[e] = o[e];
  }
}

__global__ void __launch_bounds__(256) sixd_to_aa_kernel(const float* __restrict__ d6, int ld_in, int col_off,
                                                        float* __restrict__ out, int ld_out, int rows, int joints) {
  const int64_t total = (int64_t)rows * joints;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / joints), j = (int)(i % joints);
    float a[6], o[3];
    const float* p = d6 + (size_t)r * ld_in + col_off + j * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) a[e] = p[e];
    sixd_to_aa_one(a, o);
    float* q = out + (size_t)r * ld_out + j * 3;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
  }
}

// ------------------------------------------------------------------------------ caller-side packing
// (tools/visualize.py:204-291, tools/longform_synthesis.py:413-476, 714-741)
//
// scatter_joints: pred_motion[..., part_mask] = pred_part for the four body parts in one pass.  The boolean
//   masks are whole joints, so the map is per joint: src_part[j] (0..3, -1 = joint in no part -> 0) and
//   src_joint[j] (index inside the part).
struct PartPtrs {
  const float* p[4];
  int ld[4];
};
__global__ void __launch_bounds__(256) scatter_joints_kernel(PartPtrs parts, const int* __restrict__ src_part,
                                                            const int* __restrict__ src_joint, float* __restrict__ out,
                                                            int rows, int joints) {
  const int64_t total = (int64_t)rows * joints;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / joints), j = (int)(i % joints);
    const int pt = src_part[j];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (pt >= 0) {
      const float* sp = (pt == 0 ? parts.p[0] : pt == 1 ? parts.p[1] : pt == 2 ? parts.p[2] : parts.p[3]) +
                        (size_t)r * (pt == 0 ? parts.ld[0] : pt == 1 ? parts.ld[1] : pt == 2 ? parts.ld[2] : parts.ld[3]) +
                        src_joint[j] * 3;
      a0 = sp[0]; a1 = sp[1]; a2 = sp[2];
    }
    float* q = out + (size_t)r * joints * 3 + j * 3;
    q[0] = a0; q[1] = a1; q[2] = a2;
  }
}

// F.interpolate(x.permute(0,2,1), scale_factor=s, mode='linear') source coordinates (align_corners=False):
// src = (dst + 0.5) / s - 0.5 clamped at 0, i1 = min(i0 + 1, n - 1), w1 = src - i0, out = w0*x[i0] + w1*x[i1]
__device__ __forceinline__ void lerp_coord(int t, int n, float inv_scale, int& i0, int& i1, float& w0, float& w1) {
  float src = ((float)t + 0.5f) * inv_scale - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  w1 = src - (float)i0;
  w0 = 1.0f - w1;
}

// aa -> 6D, temporal linear interpolation in 6D, 6D -> aa, fused per (clip, output frame, joint)
__global__ void __launch_bounds__(256) interp_aa_kernel(const float* __restrict__ aa, float* __restrict__ out, int B, int n,
                                                       int joints, int scale) {
  const int n_out = n * scale;
  const int64_t total = (int64_t)B * n_out * joints;
  const float inv_scale = 1.0f / (float)scale;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % joints);
    const int t = (int)((i / joints) % n_out);
    const int b = (int)(i / ((int64_t)joints * n_out));
    int i0, i1;
    float w0, w1;
    lerp_coord(t, n, inv_scale, i0, i1, w0, w1);
    const float* p0 = aa + ((size_t)b * n + i0) * joints * 3 + j * 3;
    const float* p1 = aa + ((size_t)b * n + i1) * joints * 3 + j * 3;
    float a0[3] = {p0[0], p0[1], p0[2]}, a1[3] = {p1[0], p1[1], p1[2]}, d0[6], d1[6], d[6], o[3];
    aa_to_6d_one(a0, d0);
    aa_to_6d_one(a1, d1);
#pragma unroll
    for (int e = 0; e < 6; ++e) d[e] = w0 * d0[e] + w1 * d1[e];
    sixd_to_aa_one(d, o);
    float* q = out + ((size_t)b * n_out + t) * joints * 3 + j * 3;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
  }
}

__global__ void __launch_bounds__(256) interp_linear_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int n,
                                                           int dim, int scale) {
  const int n_out = n * scale;
  const int64_t total = (int64_t)B * n_out * dim;
  const float inv_scale = 1.0f / (float)scale;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % dim);
    const int t = (int)((i / dim) % n_out);
    const int b = (int)(i / ((int64_t)dim * n_out));
    int i0, i1;
    float w0, w1;
    lerp_coord(t, n, inv_scale, i0, i1, w0, w1);
    out[i] = w0 * x[((size_t)b * n + i0) * dim + c] + w1 * x[((size_t)b * n + i1) * dim + c];
  }
}

// long-form overlap blend (longform_synthesis.py:431-476): the whole new window goes aa -> 6D -> aa; on its
// first `overlap` frames the 6D values are prev6D * (1 - w) + new6D * w with w = linspace(0, 1, overlap)[t]
__global__ void __launch_bounds__(256) blend_aa_kernel(const float* __restrict__ prev_tail, const float* __restrict__ cur,
                                                      float* __restrict__ out, int B, int n, int joints, int overlap) {
  const int64_t total = (int64_t)B * n * joints;
  const float step = overlap > 1 ? 1.0f / (float)(overlap - 1) : 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % joints);
    const int t = (int)((i / joints) % n);
    const int b = (int)(i / ((int64_t)joints * n));
    const float* pc = cur + ((size_t)b * n + t) * joints * 3 + j * 3;
    float a[3] = {pc[0], pc[1], pc[2]}, d[6], o[3];
    aa_to_6d_one(a, d);
    if (t < overlap) {
      const float* pp = prev_tail + ((size_t)b * overlap + t) * joints * 3 + j * 3;
      float ap[3] = {pp[0], pp[1], pp[2]}, dp[6];
      aa_to_6d_one(ap, dp);
      // torch.linspace(0, 1, overlap): start + step*t on the lower half, end - step*(overlap-1-t) on the upper half
      const float wn = (t < overlap / 2) ? step * (float)t : 1.0f - step * (float)(overlap - 1 - t);
      const float wp = 1.0f - wn;
#pragma unroll
      for (int e = 0; e < 6; ++e) d[e] = dp[e] * wp + d[e] * wn;
    }
    sixd_to_aa_one(d, o);
    float* q = out + ((size_t)b * n + t) * joints * 3 + j * 3;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
  }
}

__global__ void __launch_bounds__(256) blend_linear_kernel(const float* __restrict__ prev_tail, float* __restrict__ cur, int B,
                                                          int n, int dim, int overlap) {
  const int64_t total = (int64_t)B * overlap * dim;
  const float step = overlap > 1 ? 1.0f / (float)(overlap - 1) : 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % dim);
    const int t = (int)((i / dim) % overlap);
    const int b = (int)(i / ((int64_t)dim * overlap));
    const float wn = (t < overlap / 2) ? step * (float)t : 1.0f - step * (float)(overlap - 1 - t);
    const float wp = 1.0f - wn;
    float* q = cur + ((size_t)b * n + t) * dim + c;
    *q = prev_tail[((size_t)b * overlap + t) * dim + c] * wp + *q * wn;
  }
}

// generic strided 2-D copy of fp32 columns: dst[r, dcol + c] = src[r, scol + c] (+ optional
// "subtract first frame" on selected columns, used for trans x/z re-zeroing)
__global__ void __launch_bounds__(256) copy_cols_kernel(const float* __restrict__ src, int ld_src, int scol,
                                                       float* __restrict__ dst, int ld_dst, int dcol, int rows,
                                                       int ncols, int frames, unsigned rel_mask) {
  const int64_t total = (int64_t)rows * ncols;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / ncols), c = (int)(i % ncols);
    float vv = src[(size_t)r * ld_src + scol + c];
    if (frames > 0 && c < 32 && ((rel_mask >> c) & 1u)) vv = vv - src[(size_t)(r / frames) * frames * ld_src + scol + c];
    dst[(size_t)r * ld_dst + dcol + c] = vv;
  }
}

}  // namespace

static inline int grid_for(int64_t n) { return rg_grid_1d(n, 256); }

extern "C" int rg_mha(rg_handle* h, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                      int ldo, int B, int H, int Sq, int Sk, int hd, void* stream) {
  RG_REQUIRE(h, q && k && v && o, "null pointer");
  RG_REQUIRE(h, B > 0 && H > 0 && Sq > 0 && Sk > 0 && Sk <= 192 && hd > 0 && hd <= 256, "bad shape (Sk <= 192)");
  const size_t lds = ((size_t)Sk * (hd + 1) + (size_t)Sk * hd + 4 * hd + 4 * Sk) * sizeof(float);
  RG_REQUIRE(h, lds <= 160 * 1024, "K/V of one head do not fit LDS");
  static rg_attr_once lds_once;
  (void)rg_reserve_lds(lds_once, (mha_kernel), 160 * 1024);
  hipLaunchKernelGGL(mha_kernel, dim3(B * H, (Sq + QB - 1) / QB), dim3(256), lds, rg_stream(stream), q, ldq, k, ldk, v,
                     ldv, o, ldo, H, Sq, Sk, hd, 1.0f / sqrtf((float)hd));
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_mha_bf16(rg_handle* h, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           void* o, int ldo, int out_is_bf16, int B, int H, int Sq, int Sk, int hd, void* stream) {
  RG_REQUIRE(h, q && k && v && o, "null pointer");
  RG_REQUIRE(h, B > 0 && H > 0 && Sq > 0 && Sk > 0 && (Sk <= MH_MAX_SK || (Sk <= MH_BIG_SK && hd == 64)),
             "bad shape (Sk <= 192, or Sk <= 512 at head dim 64)");
  RG_REQUIRE(h, hd == 128 || hd == 64 || hd == 32 || hd == 16, "head dim must be 16, 32, 64 or 128");
  RG_REQUIRE(h, ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "row strides must be multiples of 4 floats");
  hipStream_t s = rg_stream(stream);
  if (hd == 128) launch_mha_mfma<128, 128>(h, q, ldq, k, ldk, v, ldv, reinterpret_cast<float*>(o), ldo, B, H, Sq, Sk, out_is_bf16, s);
  else if (hd == 64 && Sk > MH_MAX_SK)
    launch_mha_mfma<64, 64, MH_BIG_SK>(h, q, ldq, k, ldk, v, ldv, reinterpret_cast<float*>(o), ldo, B, H, Sq, Sk, out_is_bf16, s);
  else if (hd == 64) launch_mha_mfma<64, 64>(h, q, ldq, k, ldk, v, ldv, reinterpret_cast<float*>(o), ldo, B, H, Sq, Sk, out_is_bf16, s);
  else if (hd == 32) launch_mha_mfma<32, 32>(h, q, ldq, k, ldk, v, ldv, reinterpret_cast<float*>(o), ldo, B, H, Sq, Sk, out_is_bf16, s);
  else launch_mha_mfma<32, 16>(h, q, ldq, k, ldk, v, ldv, reinterpret_cast<float*>(o), ldo, B, H, Sq, Sk, out_is_bf16, s);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_layernorm(rg_handle* h, const float* x, const float* gamma, const float* beta, float* out, int rows,
                            int dim, void* out_bf16, void* stream) {
  RG_REQUIRE(h, x && gamma && beta && out, "null pointer");
  RG_REQUIRE(h, rows > 0 && dim > 0, "bad shape");
  hipLaunchKernelGGL(layernorm_kernel, dim3(((int64_t)rows * 64 + 255) / 256), dim3(256), 0, rg_stream(stream), x, gamma,
                     beta, out, rows, dim, 1e-5f, reinterpret_cast<unsigned short*>(out_bf16));
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_layernorm_res(rg_handle* h, const float* x, const float* residual, const float* gamma, const float* beta,
                                float* out, int rows, int dim, float eps, void* out_bf16, void* stream) {
  RG_REQUIRE(h, x && gamma && beta && out, "null pointer");
  RG_REQUIRE(h, rows > 0 && dim > 0 && eps >= 0.f, "bad shape");
  hipLaunchKernelGGL(layernorm_res_kernel, dim3(((int64_t)rows * 64 + 255) / 256), dim3(256), 0, rg_stream(stream), x, residual,
                     gamma, beta, out, rows, dim, eps, reinterpret_cast<unsigned short*>(out_bf16));
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_add_rows(rg_handle* h, const float* a, const float* b, float* out, int64_t n, int64_t period,
                           void* stream) {
  RG_REQUIRE(h, a && b && out, "null pointer");
  RG_REQUIRE(h, n > 0 && period > 0 && n % 4 == 0 && period % 4 == 0, "sizes must be multiples of 4");
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(n / 4)), dim3(256), 0, rg_stream(stream), (const float4*)a,
                     (const float4*)b, (float4*)out, n / 4, period / 4);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_copy_rows(rg_handle* h, const float* src, float* dst, int groups, int nrows_per, int dim,
                            int rows_src_per, int src_row0, int rows_dst_per, int dst_row0, void* stream) {
  RG_REQUIRE(h, src && dst, "null pointer");
  RG_REQUIRE(h, groups > 0 && nrows_per > 0 && dim > 0, "bad shape");
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((int64_t)groups * nrows_per * dim)), dim3(256), 0,
                     rg_stream(stream), src, dst, groups, nrows_per, dim, rows_src_per, src_row0, rows_dst_per, dst_row0);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

// ---- up to 4 problems of one shape per launch (the four body-part VAEs; same bits as the single calls)
extern "C" int rg_layernorm_grouped(rg_handle* h, int n, const float* const* x, const float* const* gamma, const float* const* beta,
                                    float* const* out, int rows, int dim, void* const* out_bf16, void* stream) {
  RG_REQUIRE(h, n >= 1 && n <= 4 && x && gamma && beta && out && out_bf16, "1..4 problems");
  RG_REQUIRE(h, rows > 0 && dim > 0, "bad shape");
  ln_group p{};
  for (int i = 0; i < n; ++i) {
    RG_REQUIRE(h, x[i] && gamma[i] && beta[i] && out[i], "null pointer");
    p.x[i] = x[i]; p.g[i] = gamma[i]; p.b[i] = beta[i]; p.out[i] = out[i]; p.o16[i] = reinterpret_cast<unsigned short*>(out_bf16[i]);
  }
  hipLaunchKernelGGL(layernorm_group_kernel, dim3(((int64_t)rows * 64 + 255) / 256, n), dim3(256), 0, rg_stream(stream), p, rows,
                     dim, 1e-5f);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_add_rows_grouped(rg_handle* h, int n, const float* const* a, const float* const* b, float* const* out,
                                   int64_t nel, int64_t period, void* stream) {
  RG_REQUIRE(h, n >= 1 && n <= 4 && a && b && out, "1..4 problems");
  RG_REQUIRE(h, nel > 0 && period > 0 && nel % 4 == 0 && period % 4 == 0, "sizes must be multiples of 4");
  ptr3_group p{};
  for (int i = 0; i < n; ++i) {
    RG_REQUIRE(h, a[i] && b[i] && out[i], "null pointer");
    p.a[i] = a[i]; p.b[i] = b[i]; p.out[i] = out[i];
  }
  hipLaunchKernelGGL(add_rows_group_kernel, dim3(grid_for(nel / 4), n), dim3(256), 0, rg_stream(stream), p, nel / 4, period / 4);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_copy_rows_grouped(rg_handle* h, int n, const float* const* src, float* const* dst, int groups, int nrows_per,
                                    int dim, int rows_src_per, int src_row0, int rows_dst_per, int dst_row0, void* stream) {
  RG_REQUIRE(h, n >= 1 && n <= 4 && src && dst, "1..4 problems");
  RG_REQUIRE(h, groups > 0 && nrows_per > 0 && dim > 0, "bad shape");
  ptr3_group p{};
  for (int i = 0; i < n; ++i) {
    RG_REQUIRE(h, src[i] && dst[i], "null pointer");
    p.a[i] = src[i]; p.out[i] = dst[i];
  }
  hipLaunchKernelGGL(copy_rows_group_kernel, dim3(grid_for((int64_t)groups * nrows_per * dim), n), dim3(256), 0,
                     rg_stream(stream), p, groups, nrows_per, dim, rows_src_per, src_row0, rows_dst_per, dst_row0);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_mha_bf16_grouped(rg_handle* h, int n, const float* const* q, int ldq, const float* const* k, int ldk,
                                   const float* const* v, int ldv, void* const* o, int ldo, int out_is_bf16, int B, int H, int Sq,
                                   int Sk, int hd, void* stream) {
  RG_REQUIRE(h, n >= 1 && n <= 4 && q && k && v && o, "1..4 problems");
  RG_REQUIRE(h, B > 0 && H > 0 && Sq > 0 && Sk > 0 && (Sk <= MH_MAX_SK || (Sk <= MH_BIG_SK && hd == 64)),
             "bad shape (Sk <= 192, or Sk <= 512 at head dim 64)");
  RG_REQUIRE(h, hd == 128 || hd == 64 || hd == 32 || hd == 16, "head dim must be 16, 32, 64 or 128");
  RG_REQUIRE(h, ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "row strides must be multiples of 4 floats");
  mha_group g{};
  for (int i = 0; i < n; ++i) {
    RG_REQUIRE(h, q[i] && k[i] && v[i] && o[i], "null pointer");
    g.q[i] = q[i]; g.k[i] = k[i]; g.v[i] = v[i]; g.o[i] = reinterpret_cast<float*>(o[i]);
  }
  hipStream_t s = rg_stream(stream);
  if (hd == 128) launch_mha_mfma_group<128, 128>(g, n, ldq, ldk, ldv, ldo, B, H, Sq, Sk, out_is_bf16, s);
  else if (hd == 64 && Sk > MH_MAX_SK) launch_mha_mfma_group<64, 64, MH_BIG_SK>(g, n, ldq, ldk, ldv, ldo, B, H, Sq, Sk, out_is_bf16, s);
  else if (hd == 64) launch_mha_mfma_group<64, 64>(g, n, ldq, ldk, ldv, ldo, B, H, Sq, Sk, out_is_bf16, s);
  else if (hd == 32) launch_mha_mfma_group<32, 32>(g, n, ldq, ldk, ldv, ldo, B, H, Sq, Sk, out_is_bf16, s);
  else launch_mha_mfma_group<32, 16>(g, n, ldq, ldk, ldv, ldo, B, H, Sq, Sk, out_is_bf16, s);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_vae_reparam(rg_handle* h, const float* enc, int seq, const float* eps, float* latent, int B,
                              int n_chunks, int D, int T, int row_off, void* stream) {
  RG_REQUIRE(h, enc && eps && latent, "null pointer");
  RG_REQUIRE(h, B > 0 && n_chunks > 0 && D > 0 && seq >= 2 && row_off + n_chunks <= T, "bad shape");
  hipLaunchKernelGGL(vae_reparam_kernel, dim3(grid_for((int64_t)B * n_chunks * D)), dim3(256), 0, rg_stream(stream), enc,
                     seq, eps, latent, B, n_chunks, D, T, row_off);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_aa_to_6d(rg_handle* h, const float* aa, int ld_in, float* out, int ld_out, int col_off, int rows,
                           int joints, void* stream) {
  RG_REQUIRE(h, aa && out, "null pointer");
  RG_REQUIRE(h, rows > 0 && joints > 0, "bad shape");
  hipLaunchKernelGGL(aa_to_6d_kernel, dim3(grid_for((int64_t)rows * joints)), dim3(256), 0, rg_stream(stream), aa, ld_in,
                     out, ld_out, col_off, rows, joints);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_6d_to_aa(rg_handle* h, const float* d6, int ld_in, int col_off, float* out, int ld_out, int rows,
                           int joints, void* stream) {
  RG_REQUIRE(h, d6 && out, "null pointer");
  RG_REQUIRE(h, rows > 0 && joints > 0, "bad shape");
  hipLaunchKernelGGL(sixd_to_aa_kernel, dim3(grid_for((int64_t)rows * joints)), dim3(256), 0, rg_stream(stream), d6,
                     ld_in, col_off, out, ld_out, rows, joints);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_copy_cols(rg_handle* h, const float* src, int ld_src, int scol, float* dst, int ld_dst, int dcol,
                            int rows, int ncols, int frames, unsigned rel_mask, void* stream) {
  RG_REQUIRE(h, src && dst, "null pointer");
  RG_REQUIRE(h, rows > 0 && ncols > 0 && ncols <= 32 * 8, "bad shape");
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for((int64_t)rows * ncols)), dim3(256), 0, rg_stream(stream), src,
                     ld_src, scol, dst, ld_dst, dcol, rows, ncols, frames, rel_mask);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_scatter_joints(rg_handle* h, const float* upper, int ld_u, const float* lower, int ld_l, const float* hands,
                                 int ld_h, const float* face, int ld_f, const int* src_part, const int* src_joint, float* out,
                                 int rows, int joints, void* stream) {
  RG_REQUIRE(h, upper && lower && hands && face && src_part && src_joint && out, "null pointer");
  RG_REQUIRE(h, rows > 0 && joints > 0, "bad shape");
  PartPtrs pp;
  pp.p[0] = upper; pp.p[1] = lower; pp.p[2] = hands; pp.p[3] = face;
  pp.ld[0] = ld_u; pp.ld[1] = ld_l; pp.ld[2] = ld_h; pp.ld[3] = ld_f;
  hipLaunchKernelGGL(scatter_joints_kernel, dim3(grid_for((int64_t)rows * joints)), dim3(256), 0, rg_stream(stream), pp,
                     src_part, src_joint, out, rows, joints);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_interp_aa(rg_handle* h, const float* aa, float* out, int B, int n, int joints, int scale, void* stream) {
  RG_REQUIRE(h, aa && out, "null pointer");
  RG_REQUIRE(h, B > 0 && n > 0 && joints > 0 && scale >= 1, "bad shape");
  hipLaunchKernelGGL(interp_aa_kernel, dim3(grid_for((int64_t)B * n * scale * joints)), dim3(256), 0, rg_stream(stream), aa,
                     out, B, n, joints, scale);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_interp_linear(rg_handle* h, const float* x, float* out, int B, int n, int dim, int scale, void* stream) {
  RG_REQUIRE(h, x && out, "null pointer");
  RG_REQUIRE(h, B > 0 && n > 0 && dim > 0 && scale >= 1, "bad shape");
  hipLaunchKernelGGL(interp_linear_kernel, dim3(grid_for((int64_t)B * n * scale * dim)), dim3(256), 0, rg_stream(stream), x,
                     out, B, n, dim, scale);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_blend_aa(rg_handle* h, const float* prev_tail, const float* cur, float* out, int B, int n, int joints,
                           int overlap, void* stream) {
  RG_REQUIRE(h, cur && out && (prev_tail || overlap == 0), "null pointer");
  RG_REQUIRE(h, B > 0 && n > 0 && joints > 0 && overlap >= 0 && overlap <= n, "bad shape");
  hipLaunchKernelGGL(blend_aa_kernel, dim3(grid_for((int64_t)B * n * joints)), dim3(256), 0, rg_stream(stream), prev_tail,
                     cur, out, B, n, joints, overlap);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_blend_linear(rg_handle* h, const float* prev_tail, float* cur, int B, int n, int dim, int overlap,
                               void* stream) {
  RG_REQUIRE(h, prev_tail && cur, "null pointer");
  RG_REQUIRE(h, B > 0 && n > 0 && dim > 0 && overlap > 0 && overlap <= n, "bad shape");
  hipLaunchKernelGGL(blend_linear_kernel, dim3(grid_for((int64_t)B * overlap * dim)), dim3(256), 0, rg_stream(stream),
                     prev_tail, cur, B, n, dim, overlap);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}
