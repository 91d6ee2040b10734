// Bare inner loop of conv_h2_kernel (LDS fragment reads + the three-product fp16 MFMA block, nothing else) on the two
// fp16 MFMA shapes of gfx950, same occupancy and register budget as the kernel: what bounds its matrix rate, measured.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe tools/mfma_shape_probe.hip && ./mfma_shape_probe
//
// Per workgroup (256 threads = 4 waves, 2 workgroups per CU through a 68 KB LDS request, like conv_h2_kernel<2,2,...>):
//   shape 0   v_mfma_f32_32x32x16_f16: per tap 2 planes x (2 A + 2 B) ds_read_b128, 2 x 2 x 3 MFMAs  (K = 16)
//   shape 1   v_mfma_f32_16x16x32_f16: per tap 2 planes x (4 A + 4 B) ds_read_b128, 4 x 4 x 3 MFMAs  (K = 32)
// Both move the same LDS bytes per FLOP and hold 128 accumulator registers (two sets: leading / cross terms).
// Reports TFLOP/s issued (fp16), "algorithmic" TFLOP/s (/3), cycles per MFMA from s_memtime, and the in-kernel clock
// (s_memtime / s_memrealtime), on random and on all-zero operands (MI355X_MICROARCH.md: DVFS give-back, item 7).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

union Unit {
  uint4 u;
  f16x8 b;
};

constexpr int LDS_UNITS = 4352;   // 68 KB of 16-byte units

template <int SHAPE, int READS, int ORDER = 0>
__global__ __launch_bounds__(256, 2) void probe(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                               unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) uint4 L[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS_UNITS; i += 256) L[i] = src[i];
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (lane == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (SHAPE == 0) {
    f32x16 acc[2][2], acx[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = acx[a][b][r] = 0.f;
    const uint4* const xB = L + (lane >> 5) * 340 + wave * 68 + (lane & 31);    // B: 32 consecutive units per k-half
    const uint4* const wA = L + 2720 + (lane >> 5) * 32 + (lane & 31);          // A: [.. planes][2 halves][32]
    for (int it = 0; it < iters; ++it) {
      const int rot = (it & 3) * 2;          // (the address depends on the trip count: nothing is hoisted)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        __builtin_amdgcn_sched_barrier(0);
        Unit av[2][2], bv[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            if (READS || it == 0) av[p][mt].u = wA[((mt * 3 + kw) * 2 + p) * 64 + rot * 64];
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (READS || it == 0) bv[p][q].u = xB[p * 680 + q * 34 + kw + rot];
        }
        if constexpr (ORDER == 0) {          // the kernel's order: per output tile, the two cross terms then the leading one
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              f32x16 cx = acx[mt][q];
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0][q].b, cx, 0, 0, 0);
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1][q].b, cx, 0, 0, 0);
              acx[mt][q] = cx;
              acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
            }
        } else if constexpr (ORDER == 1) {   // by product type: consecutive MFMAs never share an accumulator
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) acx[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0][q].b, acx[mt][q], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) acx[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1][q].b, acx[mt][q], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
        } else if constexpr (ORDER == 2) {   // A-stationary: the A operand changes every 2 - 4 MFMAs
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
              acx[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1][q].b, acx[mt][q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) acx[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0][q].b, acx[mt][q], 0, 0, 0);
          }
        } else {                              // B-stationary
#pragma unroll
          for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
              acx[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0][q].b, acx[mt][q], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acx[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1][q].b, acx[mt][q], 0, 0, 0);
          }
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][b][r] + acx[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    f32x4 acc[4][4], acx[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = acx[a][b][r] = 0.f;
    // B: [plane][k-quarter][pixels]: a fragment = 16 consecutive units per k-quarter; A: [..][plane][4 quarters][16]
    const uint4* const xB = L + (lane >> 4) * 340 + wave * 68 + (lane & 15);
    const uint4* const wA = L + 2720 + (lane >> 4) * 16 + (lane & 15);
    for (int it = 0; it < iters; ++it) {
      const int rot = (it & 3);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        __builtin_amdgcn_sched_barrier(0);
        Unit av[2][4], bv[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            if (READS || it == 0) av[p][mt].u = wA[((mt * 3 + kw) * 2 + p) * 64 + rot * 64];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (READS || it == 0) bv[p][q].u = xB[p * 1360 + (q >> 1) * 34 + (q & 1) * 16 + kw + rot];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 cx = acx[mt][q];
            cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[1][mt].b, bv[0][q].b, cx, 0, 0, 0);
            cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt].b, bv[1][q].b, cx, 0, 0, 0);
            acx[mt][q] = cx;
            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
          }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[a][b][r] + acx[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
  }
  if (lane == 0 && wave == 0) {
    stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <int SHAPE, int READS, int ORDER = 0>
static void run(const char* name, const uint4* src, float* out, unsigned long long* stamps, int blocks, int iters) {
  auto k = probe<SHAPE, READS, ORDER>;
  const size_t lds = LDS_UNITS * 16;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, out, iters, stamps);   // warm: the clock settles
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, out, iters, stamps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(blocks * 2);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk(blocks), cyc(blocks);
  for (int b = 0; b < blocks; ++b) {
    clk[b] = (double)st[2 * b] / (double)st[2 * b + 1] * 100e6;
    cyc[b] = (double)st[2 * b];
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  // per wave and iteration: 3 taps x 12 MFMAs of 32x32x16 (16384 MAC) or 48 of 16x16x32 (8192 MAC)
  const double mac_it = SHAPE == 0 ? 3.0 * 12 * 16384 : 3.0 * 48 * 8192;
  const double flops = 2.0 * mac_it * iters * 4.0 * blocks * reps;
  const double tf = flops / (ms * 1e-3) / 1e12;
  const double mfma_per_wave = (SHAPE == 0 ? 36.0 : 144.0) * iters;
  printf("{\"loop\": \"%s\", \"mfma\": \"%s\", \"order\": %d, \"lds_reads\": %d, \"issued_tflops\": %.1f, \"algorithmic_tflops_3_products\": %.1f, "
         "\"frac_of_2500\": %.3f, \"us_per_launch\": %.1f, \"inkernel_clock_ghz_median\": %.3f, "
         "\"s_memtime_ticks_per_mfma_per_wave_median\": %.2f}\n",
         name, SHAPE == 0 ? "32x32x16_f16" : "16x16x32_f16", ORDER, READS, tf, tf / 3, tf / 2500.0, 1e3 * ms / reps,
         clk[blocks / 2] / 1e9, cyc[blocks / 2] / mfma_per_wave);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int blocks = 512, iters = argc > 1 ? atoi(argv[1]) : 3000;
  uint4* src;
  float* out;
  unsigned long long* stamps;
  CK(hipMalloc(&src, LDS_UNITS * 16));
  CK(hipMalloc(&out, blocks * 256 * 4));
  CK(hipMalloc(&stamps, blocks * 16));
  std::vector<uint16_t> h(LDS_UNITS * 8);
  uint32_t s = 12345;
  for (auto& v : h) {   // random fp16 in +-[2^-3, 2^1): sign, exponent 12..15, random mantissa (products stay finite in fp32)
    s = s * 1664525u + 1013904223u;
    v = (uint16_t)(((s >> 31) << 15) | ((12 + ((s >> 20) & 3)) << 10) | ((s >> 8) & 0x3FF));
  }
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) std::fill(h.begin(), h.end(), 0);
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const char* tag = zero ? "zeros" : "random";
    run<0, 1>(tag, src, out, stamps, blocks, iters);
    if (!zero) {   // issue orders of the same twelve MFMAs per tap (order 0 is the kernel's)
      run<0, 1, 1>(tag, src, out, stamps, blocks, iters);
      run<0, 1, 2>(tag, src, out, stamps, blocks, iters);
      run<0, 1, 3>(tag, src, out, stamps, blocks, iters);
    }
    run<1, 1>(tag, src, out, stamps, blocks, iters);
    run<0, 0>(tag, src, out, stamps, blocks, iters);
    run<1, 0>(tag, src, out, stamps, blocks, iters);
  }
  return 0;
}
