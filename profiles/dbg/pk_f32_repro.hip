// Stand-alone reproducer attempt for the packed-fp32 hazard of NOTEBOOK 9.2: rg_6d_to_aa's arithmetic,
// compiled WITH packed fp32 VALU instructions (the compiler's default on gfx950: this file is built without the product's
// -packed-fp32-ops switch), launched 10^5 times on stream A while an LDS-free elementwise kernel keeps the SIMDs shared on
// stream B; every launch's output is compared on the device with the output of the SAME kernel's first launch, made alone on the
// chip (the hazard of NOTEBOOK 9.2 never showed alone: 0 in 1 400 launches), and that reference is itself checked against a copy of
// the function whose adds / multiplies are hidden from the packer behind opaque registers (-ffp-contract=off, so both round alike).
// Counts mismatching joints by lane index mod 64.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 pk_f32_repro.hip -o pk_f32_repro && ./pk_f32_repro [launches] [busy streams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float sqrt_pos(float v) { return v > 0.f ? sqrtf(v) : 0.f; }
__device__ __forceinline__ float copysign_ref(float a, float b) { return ((a < 0.f) != (b < 0.f)) ? -a : a; }
// OPAQUE = true: every intermediate passes through an empty asm, so no two operations can be paired into a v_pk_* instruction
template <bool OPAQUE>
__device__ __forceinline__ float op(float v) {
  if (OPAQUE) asm volatile("" : "+v"(v));
  return v;
}
template <bool Q>
__device__ __forceinline__ void sixd_to_aa_one(const float* d6, float* o) {      // csrc/rg_vae.hip: sixd_to_aa_one
  float a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
  float n1 = fmaxf(sqrtf(op<Q>(op<Q>(a1[0] * a1[0]) + op<Q>(a1[1] * a1[1])) + op<Q>(a1[2] * a1[2])), 1e-12f);
  float b1[3] = {op<Q>(a1[0] / n1), op<Q>(a1[1] / n1), op<Q>(a1[2] / n1)};
  const float dot = op<Q>(op<Q>(op<Q>(b1[0] * a2[0]) + op<Q>(b1[1] * a2[1])) + op<Q>(b1[2] * a2[2]));
  float b2[3] = {op<Q>(a2[0] - op<Q>(dot * b1[0])), op<Q>(a2[1] - op<Q>(dot * b1[1])), op<Q>(a2[2] - op<Q>(dot * b1[2]))};
  float n2 = fmaxf(sqrtf(op<Q>(op<Q>(b2[0] * b2[0]) + op<Q>(b2[1] * b2[1])) + op<Q>(b2[2] * b2[2])), 1e-12f);
  b2[0] = op<Q>(b2[0] / n2); b2[1] = op<Q>(b2[1] / n2); b2[2] = op<Q>(b2[2] / n2);
  float b3[3] = {op<Q>(op<Q>(b1[1] * b2[2]) - op<Q>(b1[2] * b2[1])), op<Q>(op<Q>(b1[2] * b2[0]) - op<Q>(b1[0] * b2[2])),
                 op<Q>(op<Q>(b1[0] * b2[1]) - op<Q>(b1[1] * b2[0]))};
  const float m00 = b1[0], m11 = b2[1], m22 = b3[2];
  const float o0 = 0.5f * sqrt_pos(op<Q>(op<Q>(op<Q>(1 + m00) + m11) + m22));
  const float qx = 0.5f * sqrt_pos(op<Q>(op<Q>(op<Q>(1 + m00) - m11) - m22));
  const float qy = 0.5f * sqrt_pos(op<Q>(op<Q>(op<Q>(1 - m00) + m11) - m22));
  const float qz = 0.5f * sqrt_pos(op<Q>(op<Q>(op<Q>(1 - m00) - m11) + m22));
  const float o1 = copysign_ref(qx, op<Q>(b3[1] - b2[2]));
  const float o2 = copysign_ref(qy, op<Q>(b1[2] - b3[0]));
  const float o3 = copysign_ref(qz, op<Q>(b2[0] - b1[1]));
  const float norm = sqrtf(op<Q>(op<Q>(op<Q>(o1 * o1) + op<Q>(o2 * o2)) + op<Q>(o3 * o3)));
  const float half = atan2f(norm, o0);
  const float angle = 2.0f * half;
  const float s = (fabsf(angle) < 1e-6f) ? (0.5f - (angle * angle) / 48.0f) : (sinf(half) / angle);
  o[0] = o1 / s; o[1] = o2 / s; o[2] = o3 / s;
}
template <bool Q>
__global__ void __launch_bounds__(256) conv(const float* __restrict__ d6, float* __restrict__ out, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float a[6], o[3];
#pragma unroll
    for (int e = 0; e < 6; ++e) a[e] = d6[(size_t)i * 6 + e];
    sixd_to_aa_one<Q>(a, o);
    out[(size_t)i * 3] = o[0]; out[(size_t)i * 3 + 1] = o[1]; out[(size_t)i * 3 + 2] = o[2];
  }
}
__global__ void __launch_bounds__(256) busy(float* x, int n, int iters) {      // stream B: LDS-free elementwise work that shares the SIMDs
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float v = x[i];
    for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0000001f, 1e-7f);
    x[i] = v;
  }
}
__global__ void __launch_bounds__(256) compare(const float* a, const float* b, int n, unsigned* hist, unsigned* first) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    bool bad = false;
    for (int e = 0; e < 3; ++e) bad |= __float_as_uint(a[(size_t)i * 3 + e]) != __float_as_uint(b[(size_t)i * 3 + e]);
    if (bad) { atomicAdd(&hist[i & 63], 1u); atomicMin(first, (unsigned)i); }
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 100000;
  const int n = 16 * 150 * 55;                    // joints of a 16-clip decode (as the product launches it)
  std::vector<float> h((size_t)n * 6);
  unsigned x = 2024u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) % 20001 - 10000) * 1e-4f; }
  float *d6, *ref, *out, *junk; unsigned *hist, *first;
  CK(hipMalloc(&d6, h.size() * 4)); CK(hipMalloc(&ref, (size_t)n * 12)); CK(hipMalloc(&out, (size_t)n * 12));
  CK(hipMalloc(&junk, (size_t)(1 << 18) * 4 * 64)); CK(hipMalloc(&hist, 64 * 4)); CK(hipMalloc(&first, 4));
  CK(hipMemcpy(d6, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(junk, 0, (size_t)(1 << 18) * 4 * 64)); CK(hipMemset(hist, 0, 64 * 4)); CK(hipMemset(first, 0xff, 4));
  const int nbusy = argc > 2 ? (atoi(argv[2]) < 1 ? 1 : (atoi(argv[2]) > 64 ? 64 : atoi(argv[2]))) : 1;      // busy streams: more than the runtime's hardware queues multiplexes them
  hipStream_t sa;
  std::vector<hipStream_t> sbs(nbusy);
  CK(hipStreamCreate(&sa));
  for (auto& q : sbs) CK(hipStreamCreate(&q));
  hipLaunchKernelGGL(conv<false>, dim3(516), dim3(256), 0, sa, d6, ref, n);     // the reference: the packed build, alone on the chip
  hipLaunchKernelGGL(conv<true>, dim3(516), dim3(256), 0, sa, d6, out, n);      // ... == the unpacked arithmetic?
  hipLaunchKernelGGL(compare, dim3(516), dim3(256), 0, sa, out, ref, n, hist, first);
  CK(hipStreamSynchronize(sa));
  {
    unsigned hh[64]; unsigned long long t = 0;
    CK(hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) t += hh[l];
    printf("alone on the chip: packed build vs opaque (unpacked) arithmetic: %llu mismatching joints of %d\n", t, n);
    CK(hipMemset(hist, 0, 64 * 4)); CK(hipMemset(first, 0xff, 4));
  }
  for (int it = 0; it < launches; ++it) {
    if (it % 4 == 0) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, sbs[(it / 4) % nbusy], junk + (size_t)((it / 4) % nbusy) * (1 << 18), 1 << 18, 40 + it % 37);
    hipLaunchKernelGGL(conv<false>, dim3(516), dim3(256), 0, sa, d6, out, n);
    hipLaunchKernelGGL(compare, dim3(516), dim3(256), 0, sa, out, ref, n, hist, first);
  }
  CK(hipDeviceSynchronize());
  unsigned hh[64], f;
  CK(hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
  unsigned long long tot = 0, hi = 0;
  for (int l = 0; l < 64; ++l) { tot += hh[l]; if (l >= 48) hi += hh[l]; }
  printf("%d launches x %d joints beside %d busy stream(s): %llu mismatching joints (%llu in lanes 48-63), first index %d\n", launches, n, nbusy, tot, hi, (int)f);
  if (tot) { printf("per lane:"); for (int l = 0; l < 64; ++l) printf(" %u", hh[l]); printf("\n"); }
  return 0;
}
