// Issue cost of the dropout hash on gfx950: the round-1 form (two 32-bit multiplies) against a 24-bit-multiply mixer of the
// same statistical quality (tools/hash_stats.py).   hipcc --offload-arch=gfx950 -O3 -o hash_probe tools/hash_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t h32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint32_t h24(uint32_t x) {
  x ^= x >> 15; x = __umul24(x, 0x6B43A9u); x ^= x >> 13; x = __umul24(x, 0x52DCE7u); x ^= x >> 16; return x;
}
template <int V>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t acc = 0, x = blockIdx.x * blockDim.x + threadIdx.x + seed;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t v = V ? h24(x + u * 7919u) : h32(x + u * 7919u);
      acc += v >= 214748364u ? 1u : 0u;
    }
    x += 65537u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  uint32_t* out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) {
    for (int w = 0; w < 3; ++w) { if (v) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, 2000, 1u); else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, 2000, 1u); }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) { if (v) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, 2000, 1u); else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, 2000, 1u); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double hashes = 10.0 * 1024 * 256 * 2000 * 8;
    printf("%s: %.3f ms per launch, %.1f G hashes/s\n", v ? "24-bit multiplies" : "32-bit multiplies", ms / 10, hashes / (ms * 1e-3) / 1e9);
  }
  return 0;
}
