// Handle lifetime + error reporting for the C ABI (include/rg_gesture.h).
#include "rg_common.h"

extern "C" int rg_version(void) { return RG_VERSION; }

extern "C" int rg_create(rg_handle** out, int device) {
  if (!out) return RG_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RG_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return RG_ERR_INVALID;
  rg_handle* h = new rg_handle();
  h->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cus = prop.multiProcessorCount;
  *out = h;
  return RG_OK;
}

extern "C" void rg_destroy(rg_handle* h) { delete h; }

extern "C" const char* rg_last_error(rg_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int rg_num_cus(rg_handle* h) { return h ? h->num_cus : 0; }

// ---- launch-form arbitration between the lanes of a pipeline (include/rg_gesture.h: rg_lane_form; the consumer is
// csrc/rg_seqx.hip).  One thread; lives here because rg_seqx.hip is built with packed fp32 and holds only kernels that own
// their SIMDs (build.py PACKED_FP32_UNITS).
namespace {
// state: one 128-byte record (RG_LANE_STRIDE ints) per lane, written by that lane's arbitration kernel only: [0] = workgroups
// its current launch form holds (read by the other lanes' arbitration), [1] = its form flag (read by its own rg_seqx launches)
__global__ void rg_lane_form_kernel(int* state, int lane, int nlanes, int narrow_wgs, int wide_wgs, int budget) {
  if (threadIdx.x != 0) return;
  int others = 0;
  for (int l = 0; l < nlanes; ++l)
    if (l != lane) others += __hip_atomic_load(state + l * RG_LANE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int wide = wide_wgs > narrow_wgs && others + wide_wgs <= budget;
  __hip_atomic_store(state + lane * RG_LANE_STRIDE, wide ? wide_wgs : narrow_wgs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(state + lane * RG_LANE_STRIDE + 1, wide, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

extern "C" int rg_lane_form(rg_handle* h, int* state, int lane, int nlanes, int narrow_wgs, int wide_wgs, int budget, void* stream) {
  RG_REQUIRE(h, state, "null state");
  RG_REQUIRE(h, nlanes >= 1 && nlanes <= 64 && lane >= 0 && lane < nlanes, "lane out of range");
  RG_REQUIRE(h, narrow_wgs >= 0 && wide_wgs >= 0 && budget >= 0, "negative workgroup count");
  hipLaunchKernelGGL(rg_lane_form_kernel, dim3(1), dim3(64), 0, rg_stream(stream), state, lane, nlanes, narrow_wgs, wide_wgs, budget);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}


