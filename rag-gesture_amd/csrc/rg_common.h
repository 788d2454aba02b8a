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
