// Go / no-go probe for "two sequences per streamed weight fragment" in rg_seq (round 5, VERDICT r04 task 1).
// Unit GEMMs only: a workgroup streams 512 x 512 bf16 weight units (1-KiB MFMA fragments, wave-private LDS-DMA rings, counted
// vmcnt -- the loop of rg_seq.hip's gemm_frags) against NS resident 48-row bf16 panels, NS * 3 MFMAs per fragment, and
// (RT = 1) round-trips the fp32 accumulators of every unit through a per-sequence L2 buffer (load, add, store) the way a
// residual stream that no longer lives in VGPRs would.  Reports us per unit, rows x units per us and CU, the shader clock held.
//   hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage seq_pair_probe.hip -o seq_pair_probe && ./seq_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xc07f); asm volatile("" ::: "memory"); }

constexpr int UNIT_BYTES = 512 * 512 * 2;

// NS sequences per workgroup; NJ 16-feature blocks per wave (4: 8 waves x 64 features, 8: 4 waves x 128 features, one per
// SIMD with 512 registers); RD ring slots per wave; RT: 0 accumulators stay in registers, 1 load + add + store per unit
template <int NS, int NJ, int RD, int RT, int PF = 0, int BAR = 0, int PRIO = 0>
__global__ void __launch_bounds__(32 / NJ * 64) probe(const unsigned char* __restrict__ wstream, float* __restrict__ R, int nunits,
                                                       int stream_units, float* out, unsigned long long* clk) {
  constexpr int NW = 32 / NJ, NB = NS * 3, FPU = 16 * NJ;     // waves, token blocks, fragments per unit and wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const ring = smem + NS * 49152 + wave * (RD * 1024);
  for (int i = tid; i < NS * 49152 / 4; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + ((i * 2654435761u >> 20) & 0x7f007f);
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wstream), 0, 0x7fffffff, 0x00020000);
  int ir = 0, iu = 0, soff = wave * (FPU * 1024), head = 0;
  auto issue = [&](int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(ring + slot * 1024), 16, lane * 16, soff, 0, 0);
    soff += 1024;
    if (++ir == FPU) {
      ir = 0;
      soff += (NW - 1) * FPU * 1024;
      if (++iu == stream_units) { iu = 0; soff = wave * (FPU * 1024); }
    }
  };
  auto dma_frag = [&](int slot, int unit, int f) {     // (PF == 3) fragment f of a unit by LDS-DMA into a ring slot
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(ring + slot * 1024), 16, lane * 16,
                                             (((unit % stream_units) * NW + wave) * FPU + f) * 1024, 0, 0);
  };
  if constexpr (PF == 3) {
#pragma unroll
    for (int s = 0; s < 6; ++s) dma_frag(s, 0, s);
  } else if constexpr (PF != 2) {
#pragma unroll
    for (int s = 0; s < RD; ++s) issue(s);
  }
  f32x4 acc[NJ][NB];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* pl = smem + lane * 16;
  const unsigned char* rl = ring + lane * 16;
  f32x4* Rw = reinterpret_cast<f32x4*>(R) + ((size_t)(blockIdx.x * NW + wave) * (NJ * NB)) * 64 + lane;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 wr[8];
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(wstream) + lane;
  auto frag_ptr = [&](int unit, int f) { return wsrc + ((size_t)((unit % stream_units) * NW + wave) * FPU + f) * 64; };
  if constexpr (PF == 2) {
#pragma unroll
    for (int f = 0; f < 8; ++f) wr[f] = *frag_ptr(0, f);
  }
  unsigned long long tg = 0;
#pragma unroll 1
  for (int u = 0; u < nunits; ++u) {
    const unsigned long long tu0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (PF == 0) {
    bf16x8 w[2], pf[2][NB];
      wait_vmcnt<RD - 1>();
      w[0] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
  #pragma unroll
      for (int b = 0; b < NB; ++b) pf[0][b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16) << 10));
  #pragma unroll 1
      for (int s2 = 0; s2 < 16; s2 += 2) {
  #pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
  #pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const bool last = ss == 1 && j == NJ - 1 && s2 == 14;
            wait_lds();
            issue(head);
            head = head + 1 == RD ? 0 : head + 1;
            if (!last) {
              wait_vmcnt<RD - 1>();
              w[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
            }
            if (j == NJ - 1 && !last) {
  #pragma unroll
              for (int b = 0; b < NB; ++b) pf[ss ^ 1][b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16 + s2 + ss + 1) << 10));
            }
            __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
            for (int b = 0; b < NB; ++b) acc[j][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j & 1], pf[ss][b], acc[j][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
  }
    if constexpr (PF == 1) {     // one set of panel fragments, each re-read for the next k-step behind its last MFMA (rg_seq2.hip)
      bf16x8 w[2], pf[NB];
      wait_vmcnt<RD - 1>();
      w[0] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
#pragma unroll
      for (int b = 0; b < NB; ++b) pf[b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16) << 10));
#pragma unroll 1
      for (int s = 0; s < 16; ++s) {
        if (PRIO == 1 && s == 0 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
        if (PRIO >= 2 && wave >= NW / 2) { if ((s >> (PRIO - 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (j == 0) __builtin_amdgcn_s_waitcnt(0xc07f | (NB << 8)); else __builtin_amdgcn_s_waitcnt(0xc07f);
          asm volatile("" ::: "memory");
          issue(head);
          head = head + 1 == RD ? 0 : head + 1;
          if (j < NJ - 1 || s != 15) {
            wait_vmcnt<RD - 1>();
            w[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(rl + head * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[j][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j & 1], pf[b], acc[j][b], 0, 0, 0);
            if (j == NJ - 1) {
              pf[b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16 + s + 1) << 10));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (PF == 2) {     // weights straight into registers (no LDS ring): 8 fragments (two k-steps) in flight per wave
      static_assert(NJ == 4, "register ring: 8 fragments = 2 k-steps");
      bf16x8 pf[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) pf[b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16) << 10));
#pragma unroll 1
      for (int s2 = 0; s2 < 16; s2 += 2) {
        if (PRIO == 1 && s2 == 0 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
        if (PRIO >= 2 && wave >= NW / 2) { if ((s2 >> (PRIO - 1)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int j = f & 3;
          const bf16x8 wv = __builtin_bit_cast(bf16x8, wr[f]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[j][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, pf[b], acc[j][b], 0, 0, 0);
            if (j == NJ - 1) {
              pf[b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16 + s2 + (f >> 2) + 1) << 10));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          wr[f] = *frag_ptr(u + (s2 == 14), ((s2 + 2) & 15) * 4 + f);      // the same fragment slot, two k-steps on
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (PF == 3) {     // hybrid: a unit's first six fragments through the LDS ring (issued before the unit starts: by the
      // previous unit's last six iterations -- in the kernel, across its epilogue), the other 58 straight into registers; one
      // in-order pipeline of six fragments in flight, the destination depends on the fragment's position only.  The first and
      // the last pair of k-steps are peeled so that no group has a branch (the compiler's own vmcnt placement stays exact).
      static_assert(NJ == 4 && RD == 6, "hybrid ring");
      bf16x8 pf[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) pf[b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16) << 10));
      wait_vmcnt<5>();
      wr[0] = *reinterpret_cast<const u32x4*>(rl);
      auto group = [&](const int s2, auto first_tag, auto last_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        // PRIO 2: the later-dispatched half of the waves at priority 1 in every other group (two k-steps); 3: every other k-step;
        // 4: the later half at priority 1 throughout; 5: the EARLIER half at priority 1 in every other group
        if (PRIO == 2 && wave >= NW / 2) { if ((s2 >> 1) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (PRIO == 4 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
        if (PRIO == 5 && wave < NW / 2) { if ((s2 >> 1) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int j = f & 3;
          if (PRIO == 3 && wave >= NW / 2 && (f & 3) == 0) { if ((f >> 2) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
          if (FIRST && f + 1 < 6) {      // the next fragment comes through LDS: it has landed when at most 4 younger loads are out
            wait_vmcnt<4>();
            wr[f + 1] = *reinterpret_cast<const u32x4*>(rl + (f + 1) * 1024);
          }
          const bf16x8 wv = __builtin_bit_cast(bf16x8, wr[f]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[j][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, pf[b], acc[j][b], 0, 0, 0);
            if (j == NJ - 1) {
              pf[b] = *reinterpret_cast<const bf16x8*>(pl + ((b * 16 + s2 + (f >> 2) + 1) << 10));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (LAST && f >= 2) dma_frag(f - 2, u + 1, f - 2);      // fragments 0-5 of the next unit: LDS (slot last read by this unit's fragment f - 2)
          else wr[(f + 6) & 7] = *frag_ptr(u, s2 * 4 + f + 6);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      group(0, std::true_type(), std::false_type());
#pragma unroll 1
      for (int s2 = 2; s2 < 14; s2 += 2) group(s2, std::false_type(), std::false_type());
      group(14, std::false_type(), std::true_type());
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    tg += __builtin_amdgcn_s_memrealtime() - tu0;
    if (BAR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (RT) {      // residual round trip: 12 KiB per sequence and wave each way, lane-linear 1-KiB wave instructions
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        f32x4 r[NJ][3];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int b = 0; b < 3; ++b) r[j][b] = Rw[((s * NJ + j) * 3 + b) * 64];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const f32x4 v = acc[j][s * 3 + b] * 0.5f + r[j][b];
            Rw[((s * NJ + j) * 3 + b) * 64] = v;
            acc[j][s * 3 + b] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
  }
  wait_vmcnt<0>();
  if (lane == 0) clk[512 + blockIdx.x * 8 + wave] = tg;
  if (tid == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b) s += acc[j][b][0] + acc[j][b][1] + acc[j][b][2] + acc[j][b][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int NS, int NJ, int RD, int RT, int PF = 0, int BAR = 0, int PRIO = 0>
void run(const unsigned char* stream, int stream_units, float* R, int grid, float* out, unsigned long long* clk) {
  constexpr int NW = 32 / NJ;
  const int nunits = 260;
  const size_t lds = NS * 49152 + NW * RD * 1024;
  auto k = probe<NS, NJ, RD, RT, PF, BAR, PRIO>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, stream, R, nunits, stream_units, out, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned long long c[2]; CK(hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> wt(grid * 8); CK(hipMemcpy(wt.data(), clk + 512, wt.size() * 8, hipMemcpyDeviceToHost));
  double told = 0, tyoung = 0;
  for (int g = 0; g < grid; ++g) for (int w = 0; w < NW; ++w) (w < NW / 2 ? told : tyoung) += (double)wt[g * 8 + w] / 100.0 / nunits / (grid * NW / 2);
  const double us_unit = best * 1e3 / nunits, mhz = (double)c[0] / ((double)c[1] / 100.0);
  const double flop = 2.0 * 512 * 512 * 48 * NS;
  printf("NS=%d waves=%d RD=%2d RT=%d PF=%d BAR=%d PRIO=%d grid=%3d | %6.2f us/unit (in-loop: first half of the waves %5.2f, second half %5.2f) | rows x units / us / CU: %5.2f (48-row) %5.2f (43-row) | %5.1f GB/s/CU | "
         "MFMA %5.1f TF/CU-set = %.3f of peak(grid CUs) | clk %4.0f MHz\n", NS, NW, RD, RT, PF, BAR, PRIO, grid, us_unit, told, tyoung, 48.0 * NS / us_unit, 43.0 * NS / us_unit,
         UNIT_BYTES / us_unit / 1e3, flop * grid / us_unit / 1e6, flop * grid / us_unit / 1e6 / (2500.0 * grid / 256), mhz);
}

int main() {
  const int stream_units = 130;     // one conditional forward's worth of distinct units (65 MB: streams through L2 like the real one)
  std::vector<unsigned short> h((size_t)stream_units * UNIT_BYTES / 2);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u | ((x >> 16) & 0x1ff) | ((x >> 9) & 0x8000u)); }
  unsigned char* stream; float *out, *R; unsigned long long* clk;
  CK(hipMalloc(&stream, h.size() * 2));
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  CK(hipMalloc(&R, (size_t)256 * 2 * 48 * 512 * sizeof(float)));
  CK(hipMemset(R, 0, (size_t)256 * 2 * 48 * 512 * sizeof(float)));
  CK(hipMalloc(&clk, (512 + 256 * 8) * sizeof(unsigned long long)));
  CK(hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  printf("today's rg_seq GEMM phases: 130 units x 43 rows / 720 us = 7.76 rows x units / us / CU; go >= 11.6 (43-row)\n");
  for (int grid : {64, 256}) {
    run<2, 4, 6, 0, 1, 1>(stream, stream_units, R, grid, out, clk);       // LDS ring, barrier per unit
    run<2, 4, 6, 0, 3, 1>(stream, stream_units, R, grid, out, clk);       // hybrid: six fragments per unit through LDS, 58 into registers
    run<2, 4, 6, 0, 3, 1, 2>(stream, stream_units, R, grid, out, clk);    // ... with priority alternation (see PRIO in the kernel)
    run<2, 4, 6, 0, 3, 1, 3>(stream, stream_units, R, grid, out, clk);
    run<2, 4, 6, 0, 3, 1, 4>(stream, stream_units, R, grid, out, clk);
    run<2, 4, 6, 0, 3, 1, 5>(stream, stream_units, R, grid, out, clk);
    run<2, 4, 6, 0, 2, 1>(stream, stream_units, R, grid, out, clk);       // register ring
  }
  return 0;
}
