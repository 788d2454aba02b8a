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
