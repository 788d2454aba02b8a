// Feasibility probe for the sequence-stationary denoiser engine (round 3): how fast can ONE workgroup per CU stream a
// shared weight stream (1-KiB MFMA B fragments, lane-linear) through wave-private rings while it multiplies them with a
// 48-row activation panel that stays in LDS?  Reports GB/s per CU and chip-wide.
//   hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// MODE 0: LDS-DMA into a wave-private ring of D 1-KiB slots; MODE 1: global_load_dwordx4 into a register ring of D
template <int D, int MODE>
__global__ void __launch_bounds__(512, 2) probe(const u32x4* __restrict__ stream, int nfrag, long stream_frags, int skew_groups,
                                                 float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* panel = smem;                       // 48 KiB: [3 row blocks][16 k32 steps][1 KiB fragment]
  unsigned char* ring = smem + 49152 + wave * (D * 1024);
  for (int i = tid; i < 49152 / 4; i += 512) reinterpret_cast<unsigned*>(panel)[i] = 0x3c003c00u + (i & 0xff);
  __syncthreads();
  // every wave reads fragments f = base + i * 8 + wave (mod stream_frags); workgroups of different skew groups start apart
  const long base = (skew_groups > 1) ? (long)(blockIdx.x % skew_groups) * (stream_frags / skew_groups) : 0;
  auto src = [&](int i) { return stream + (((base + (long)i * 8 + wave) & (stream_frags - 1)) * 64 + lane); };
  f32x4 acc[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3];
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < D - 1; ++j) __builtin_amdgcn_global_load_lds((const void*)src(j), (lds_void*)(ring + j * 1024), 16, 0, 0);
    for (int i0 = 0; i0 < nfrag; i0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int i = i0 + j;
        wait_vmcnt<D - 2>();     // (the tail over-waits nothing: loads past nfrag wrap around the stream)
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(ring + j * 1024 + lane * 16);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_global_load_lds((const void*)src(i + D - 1), (lds_void*)(ring + ((j + D - 1) % D) * 1024), 16, 0, 0);
        if ((j & 3) == 0) {
          const int s = (i >> 2) & 15;
#pragma unroll
          for (int r = 0; r < 3; ++r) a[r] = *reinterpret_cast<const bf16x8*>(panel + (r * 16 + s) * 1024 + lane * 16);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b, acc[r][j & 3], 0, 0, 0);
      }
    }
    wait_vmcnt<0>();
  } else {
    u32x4 rg[D];
#pragma unroll
    for (int j = 0; j < D - 1; ++j) rg[j] = *src(j);
    for (int i0 = 0; i0 < nfrag; i0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int i = i0 + j;
        rg[(j + D - 1) % D] = *src(i + D - 1);
        const bf16x8 b = __builtin_bit_cast(bf16x8, rg[j]);
        if ((j & 3) == 0) {
          const int s = (i >> 2) & 15;
#pragma unroll
          for (int r = 0; r < 3; ++r) a[r] = *reinterpret_cast<const bf16x8*>(panel + (r * 16 + s) * 1024 + lane * 16);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b, acc[r][j & 3], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int D, int MODE>
void run(const u32x4* stream, long stream_frags, int grid, int skew, float* out, double mb_per_wg) {
  const int nfrag = (int)(mb_per_wg * 1024 * 1024 / 1024 / 8) / D * D;     // fragments per wave
  const size_t lds = 49152 + (MODE == 0 ? 8 * D * 1024 : 0);
  CK(hipFuncSetAttribute((const void*)probe<D, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<D, MODE>), dim3(grid), dim3(512), lds, 0, stream, nfrag, stream_frags, skew, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes_wg = (double)nfrag * 8 * 1024;
  const int conc = grid < 256 ? grid : 256;
  const double rounds = (grid + 255) / 256;
  printf("mode=%s D=%2d grid=%4d skew=%d  %8.1f us  per-WG %6.1f GB/s  chip %6.2f TB/s  (%.1f MB per WG)\n", MODE ? "reg" : "dma", D, grid,
         skew, best * 1e3, bytes_wg * rounds / (best * 1e-3) / 1e9, bytes_wg * grid / (best * 1e-3) / 1e12, bytes_wg / 1048576.0);
  (void)conc;
}

int main() {
  const long stream_bytes = 64L << 20;
  const long stream_frags = stream_bytes / 1024;
  std::vector<unsigned short> h(stream_bytes / 2);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u | ((x >> 16) & 0x1ff) | ((x >> 9) & 0x8000u)); }
  u32x4* stream; float* out;
  CK(hipMalloc(&stream, stream_bytes));
  CK(hipMalloc(&out, 1024 * 512 * sizeof(float)));
  CK(hipMemcpy(stream, h.data(), stream_bytes, hipMemcpyHostToDevice));
  const double mb = 16.0;
  for (int grid : {256, 32, 8}) {
    for (int skew : {1, 2}) {
      run<4, 0>(stream, stream_frags, grid, skew, out, mb);
      run<8, 0>(stream, stream_frags, grid, skew, out, mb);
      run<12, 0>(stream, stream_frags, grid, skew, out, mb);
      run<4, 1>(stream, stream_frags, grid, skew, out, mb);
      run<8, 1>(stream, stream_frags, grid, skew, out, mb);
      run<12, 1>(stream, stream_frags, grid, skew, out, mb);
      run<16, 1>(stream, stream_frags, grid, skew, out, mb);
    }
  }
  return 0;
}
