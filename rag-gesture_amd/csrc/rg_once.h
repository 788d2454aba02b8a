// A one-time-per-DEVICE guard for function attributes (hipFuncSetAttribute is per device, and the C ABI has no process-global
// state to rely on: a second handle on another device of the same process must not find the attribute "already set"; two
// threads arriving together may both set it, which is harmless).  Host-only, no HIP types: tests/test_capi_cpu.py compiles it.
#pragma once
#include <atomic>

struct rg_attr_once {
  std::atomic<unsigned long long> done{0};
  // set() -> bool (true = the attribute is in place); called at most once per device unless it fails or two threads race
  template <class F>
  bool operator()(int device, F&& set) {
    const unsigned long long bit = 1ull << (device & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (!set()) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
  }
};
