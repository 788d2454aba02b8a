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
