// Internal helpers shared by the HIP translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/rg_gesture.h"
#include "rg_once.h"

#include <vector>

struct rg_prof_rec {
  hipEvent_t start, stop;
  int variant;      // 0: fp32-A bf16 GEMM, 1: bf16-A GEMM, 2: bf16x3 GEMM
  double flops;
};

struct rg_handle {
  int device = 0;
  int num_cus = 256;
  std::string err;
  int gemm_path = 0;               // rg_set_gemm_path
  int gemm_waves = 0;              // rg_set_gemm_waves: 0 = auto, or 4 / 8 waves per LDS-DMA GEMM workgroup
  bool profiling = false;          // rg_profile_begin/end: HIP events around every rg_gemm launch
  std::vector<rg_prof_rec> prof;
  std::vector<hipEvent_t> ev_pool;
};

#define RG_REQUIRE(h, cond, msg)                                   \
  do {                                                             \
    if (!(cond)) {                                                 \
      if (h) (h)->err = std::string(__func__) + ": " + (msg);      \
      return RG_ERR_INVALID;                                       \
    }                                                              \
  } while (0)

#define RG_CHECK_LAUNCH(h)                                                         \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      if (h) (h)->err = std::string(__func__) + ": " + hipGetErrorString(e__);     \
      return RG_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

static inline hipStream_t rg_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// the device the calling thread launches on (the handle's device for every entry point that has one)
static inline int rg_current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}
// LDS reservation of a kernel, once per device (rg_once.h)
template <class K>
static inline bool rg_reserve_lds(rg_attr_once& once, K kernel, size_t bytes) {
  return once(rg_current_device(), [&]() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
  });
}

// grid for a grid-stride memory-bound kernel: enough blocks to fill 256 CUs x 8, no more.
static inline int rg_grid_1d(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// A kernel that says this is allocated all 256 vector registers per wave whatever it uses (the clobber raises its register
// count): with two waves per SIMD (512 threads, one workgroup per compute unit) NOTHING else can be resident on its SIMDs.
// Why (round 6, NOTEBOOK 11.3): rg_seq2_kernel went from 251 to 247 registers, 16 per SIMD came free, waves of the pipeline's
// small kernels (8-16 registers) moved in beside it -- and one workgroup in ~10^5 (two clips of a batch) came out wrong, in 10
// of 10 full-depth runs of profiles/race_stress.py; with all 256 allocated: 0 of 8, same code.  Which instruction misbehaves
// beside a foreign wave is not known (the two-conversion bf16 pack fails the same way, so it is not the packed conversion;
// round 4 met the same signature and blamed packed fp32 arithmetic, build.py).  Every sequence-stationary kernel owns its SIMDs.
#define RG_OWN_THE_SIMD() asm volatile("; v255 reserved: the kernel's waves fill their SIMDs" ::: "v255")

// Two fp32 -> one dword of packed bf16 (lo in bits 0-15), round-to-nearest-even: ONE v_cvt_pk_bf16_f32.  (Written as two scalar
// conversions + shift + or, the compiler emits two v_cvt_pk_bf16_f32, a shift and an SDWA or: four VALU instructions per pair,
// ~10 % of the vector instructions of the sequence-stationary kernels' epilogues.)  Same rounding, same bits.
typedef __attribute__((ext_vector_type(2))) float rg_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 rg_bf16x2;
__device__ __forceinline__ unsigned rg_pack2_bf16_one(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(rg_f32x2{lo, hi}, rg_bf16x2));
}
// The same value from two conversions whose HIGH result halves are never used (what every kernel did until round 6).
__device__ __forceinline__ unsigned rg_pack2_bf16_two(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
// Kernels whose workgroups own their SIMDs outright (two waves x 256 registers: rg_seq2, rg_venc) define RG_PACK2_ONE before
// including this header; every other kernel keeps the two-conversion form -- see build.py NO_PACKED_FP32 for why.
__device__ __forceinline__ unsigned rg_pack2_bf16(float lo, float hi) {
#if defined(RG_PACK2_ONE) || defined(RG_PACK2_ONE_EVERYWHERE)
  return rg_pack2_bf16_one(lo, hi);
#else
  return rg_pack2_bf16_two(lo, hi);
#endif
}

// GELU (erf form, torch's default) for the fused kernels' epilogues:  GELU(v) = max(v, 0) - |v| h(x),  x = |v| / sqrt(2),
// h = erfc(x) / 2 = 2^-g(x) with g a degree-7 polynomial (g(0) = 1), fitted on x in [0, 4.5] (profiles/dbg/gelu_fit.py); beyond
// that g keeps growing (positive leading coefficient) and h underflows to 0.  Seven fused multiply-adds and ONE transcendental
// (Abramowitz-Stegun 7.1.26, used until round 5, needs a reciprocal AND an exponential and five more vector instructions).
// Max abs error 7.6e-7 (A-S: 6.9e-7), and RELATIVE error <= 2e-5 of erfc down to 1e-9, i.e. also in GELU's negative tail.
__device__ __forceinline__ float rg_gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  float p = fmaf(1.904678831e-05f, x, -4.679475024e-04f);
  p = fmaf(p, x, 5.123828382e-03f);
  p = fmaf(p, x, -3.364521737e-02f);
  p = fmaf(p, x, 1.520822882e-01f);
  p = fmaf(p, x, 9.172845077e-01f);
  p = fmaf(p, x, 1.628025418e+00f);
  p = fmaf(p, -x, -1.0f);                                  // -g(x)
  return fmaf(-fabsf(v), __builtin_amdgcn_exp2f(p), fmaxf(v, 0.0f));
}
// e^(q - m) as one fused multiply-add and the hardware's 2^x (the caller passes nm2 = -m log2(e))
__device__ __forceinline__ float rg_exp_sub(float q, float nm2) {
  return __builtin_amdgcn_exp2f(fmaf(q, 1.44269504088896340736f, nm2));
}

// ---- two-wide fp32 epilogue arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 where the translation unit is built with
// packed fp32 -- build.py PACKED_FP32_UNITS: the kernels that own their SIMDs -- and the same operations one by one elsewhere:
// every operation is an exactly rounded IEEE multiply / add / fused multiply-add per element either way, so the results do not
// depend on how a unit is built).  The vector ALU issues one instruction per 4 cycles and wave whatever its width: the
// epilogues of the sequence-stationary kernels are bound by that count (profiles/r06a_pmc_sq.txt).
typedef __attribute__((ext_vector_type(4))) float rg_f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned rg_u32x2;
__device__ __forceinline__ rg_f32x2 rg_lo(const rg_f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ rg_f32x2 rg_hi(const rg_f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ rg_f32x2 rg_splat2(float x) { return rg_f32x2{x, x}; }
__device__ __forceinline__ rg_f32x2 rg_fma2(rg_f32x2 a, rg_f32x2 b, rg_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ rg_f32x2 rg_exp2_2(rg_f32x2 t) { return rg_f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])}; }
__device__ __forceinline__ rg_f32x2 rg_rcp2(rg_f32x2 t) { return rg_f32x2{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])}; }
__device__ __forceinline__ unsigned rg_pack2v(rg_f32x2 v) { return rg_pack2_bf16(v[0], v[1]); }
// SiLU(z) = z / (1 + 2^(-z log2 e))
__device__ __forceinline__ rg_f32x2 rg_silu2(rg_f32x2 z) {
  return z * rg_rcp2(rg_exp2_2(z * -1.44269504088896340736f) + 1.0f);
}
// GELU, two at a time: rg_gelu_erf's arithmetic, operation for operation
__device__ __forceinline__ rg_f32x2 rg_gelu2(rg_f32x2 v) {
  const rg_f32x2 av = rg_f32x2{fabsf(v[0]), fabsf(v[1])};
  const rg_f32x2 x = av * 0.70710678118654752440f;
  rg_f32x2 p = rg_fma2(rg_splat2(1.904678831e-05f), x, rg_splat2(-4.679475024e-04f));
  p = rg_fma2(p, x, rg_splat2(5.123828382e-03f));
  p = rg_fma2(p, x, rg_splat2(-3.364521737e-02f));
  p = rg_fma2(p, x, rg_splat2(1.520822882e-01f));
  p = rg_fma2(p, x, rg_splat2(9.172845077e-01f));
  p = rg_fma2(p, x, rg_splat2(1.628025418e+00f));
  p = rg_fma2(p, -x, rg_splat2(-1.0f));
  return rg_fma2(-av, rg_exp2_2(p), rg_f32x2{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)});
}
// four values (v - mean) rstd as bf16 (r = rstd, nm = -mean rstd: one fused multiply-add per value)
__device__ __forceinline__ rg_u32x2 rg_norm4_bf16(const rg_f32x4 v, float r, float nm) {
  const rg_f32x2 r2 = rg_splat2(r), n2 = rg_splat2(nm);
  return rg_u32x2{rg_pack2v(rg_fma2(rg_lo(v), r2, n2)), rg_pack2v(rg_fma2(rg_hi(v), r2, n2))};
}
// StylizationBlock front half on four values: SiLU(((v - mean) rstd) gain + off) as bf16
__device__ __forceinline__ rg_u32x2 rg_styl4_bf16(const rg_f32x4 v, float r, float nm, const rg_f32x4 gain, const rg_f32x4 off) {
  const rg_f32x2 r2 = rg_splat2(r), n2 = rg_splat2(nm);
  const rg_f32x2 lo = rg_silu2(rg_fma2(rg_fma2(rg_lo(v), r2, n2), rg_lo(gain), rg_lo(off)));
  const rg_f32x2 hi = rg_silu2(rg_fma2(rg_fma2(rg_hi(v), r2, n2), rg_hi(gain), rg_hi(off)));
  return rg_u32x2{rg_pack2v(lo), rg_pack2v(hi)};
}
// GELU of four values as bf16
__device__ __forceinline__ rg_u32x2 rg_gelu4_bf16(const rg_f32x4 v) {
  return rg_u32x2{rg_pack2v(rg_gelu2(rg_lo(v))), rg_pack2v(rg_gelu2(rg_hi(v)))};
}
// sum and sum of squares of a lane's 16 values of one token row (four 16-feature blocks), in a FIXED association (both
// denoiser kernels use this: their LayerNorm statistics agree bit for bit): pairwise over the element pairs, then across
__device__ __forceinline__ void rg_sum_sq16(const rg_f32x4 a, const rg_f32x4 b, const rg_f32x4 c, const rg_f32x4 d, float& s, float& ss) {
  rg_f32x2 t = rg_lo(a) + rg_hi(a), q = rg_lo(a) * rg_lo(a);
  q = rg_fma2(rg_hi(a), rg_hi(a), q);
  t = t + (rg_lo(b) + rg_hi(b));
  q = rg_fma2(rg_lo(b), rg_lo(b), q);
  q = rg_fma2(rg_hi(b), rg_hi(b), q);
  t = t + (rg_lo(c) + rg_hi(c));
  q = rg_fma2(rg_lo(c), rg_lo(c), q);
  q = rg_fma2(rg_hi(c), rg_hi(c), q);
  t = t + (rg_lo(d) + rg_hi(d));
  q = rg_fma2(rg_lo(d), rg_lo(d), q);
  q = rg_fma2(rg_hi(d), rg_hi(d), q);
  s = t[0] + t[1];
  ss = q[0] + q[1];
}
// sum / max over the four 16-lane groups of a wave on the VALU (v_permlane16_swap, v_permlane32_swap)
__device__ __forceinline__ float rg_xsum4(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float rg_xmax4(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
// softmax over the 32 features of one head for the lane's token: the 8 values in the lane (q0, q1) x the 4 lane groups
__device__ __forceinline__ void rg_softmax32(rg_f32x4& q0, rg_f32x4& q1) {
  float mx = fmaxf(fmaxf(fmaxf(q0[0], q0[1]), fmaxf(q0[2], q0[3])), fmaxf(fmaxf(q1[0], q1[1]), fmaxf(q1[2], q1[3])));
  mx = rg_xmax4(mx);
  const rg_f32x2 l2 = rg_splat2(1.44269504088896340736f), nm2 = rg_splat2(mx * -1.44269504088896340736f);
  const rg_f32x2 e0 = rg_exp2_2(rg_fma2(rg_lo(q0), l2, nm2)), e1 = rg_exp2_2(rg_fma2(rg_hi(q0), l2, nm2));
  const rg_f32x2 e2 = rg_exp2_2(rg_fma2(rg_lo(q1), l2, nm2)), e3 = rg_exp2_2(rg_fma2(rg_hi(q1), l2, nm2));
  const rg_f32x2 t = (e0 + e1) + (e2 + e3);
  const float inv = __builtin_amdgcn_rcpf(rg_xsum4(t[0] + t[1]));
  const rg_f32x2 i2 = rg_splat2(inv);
  const rg_f32x2 a = e0 * i2, b = e1 * i2, c = e2 * i2, d = e3 * i2;
  q0 = rg_f32x4{a[0], a[1], b[0], b[1]};
  q1 = rg_f32x4{c[0], c[1], d[0], d[1]};
}

