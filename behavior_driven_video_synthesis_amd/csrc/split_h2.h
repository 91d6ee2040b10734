// The "h2" operand split shared by the fp16-matrix-core kernels (conv_h2_kernel.h, conv_wgrad_h2.hip) and the weight
// pack (weightnorm.hip):  s x = h + l / 2^11  with h = f16(s x), l = f16((s x - h) * 2^11), both round-to-nearest-even.
#pragma once
#include "common.h"

typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));

constexpr int H2_SLAB = 384;          // 16-byte units of one (chunk, kh, m-tile) weight slab: [3 kw][2 planes][2 halves][32]
constexpr int H2_AMAX_PARTS = 1024;   // partial |x| maxima handed to a kernel: 512 per source (vunet_absmax_partials)
constexpr int H2_TOP = 14;            // a tensor's largest magnitude is scaled into [2^13, 2^14): 4x headroom below fp16's 65504

// exponent e such that  m * 2^e  lies in [2^(H2_TOP-1), 2^H2_TOP)  (m > 0; 0 for an all-zero tensor).  The largest fp32
// magnitudes need e = 14 - 128 = -114 (never clamped: a too-small down-scale would overflow fp16); tiny tensors are capped
// at 2^126 (below ~2^-112 the data is under-scaled and loses relative precision gradually, as fp32 subnormals do).
__host__ __device__ __forceinline__ int h2_scale_exp(float m) {
  if (!(m > 0.f) || !(m < 3.4e38f)) return 0;      // zeros, NaN, Inf: leave the data alone (they propagate as in fp32)
  int e;
  frexpf(m, &e);                                   // m = f * 2^e, f in [0.5, 1)
  e = H2_TOP - e;
  return e > 126 ? 126 : e;
}

__host__ __device__ __forceinline__ float h2_pow2(int e) {   // exact 2^e, |e| <= 126
  return __uint_as_float((uint32_t)(127 + e) << 23);
}

// 2^etot for |etot| <= 252 as two factors of the same sign (each a normal fp32 power of two): v * f1 * f2 overflows /
// underflows only if v * 2^etot itself does
__host__ __device__ __forceinline__ void h2_pow2_pair(int etot, float& f1, float& f2) {
  const int e1 = etot < -126 ? -126 : (etot > 126 ? 126 : etot);
  f1 = h2_pow2(e1);
  f2 = h2_pow2(etot - e1);
}

// (a, b), already scaled -> packed fp16 pairs of the two split terms (low half = a)
__device__ __forceinline__ void h2_split2(float a, float b, uint32_t& h, uint32_t& l) {
  h2_f16x2 ph;
  ph[0] = (_Float16)a;
  ph[1] = (_Float16)b;
  h2_f16x2 pl;
  pl[0] = (_Float16)((a - (float)ph[0]) * 2048.f);
  pl[1] = (_Float16)((b - (float)ph[1]) * 2048.f);
  h = __builtin_bit_cast(uint32_t, ph);
  l = __builtin_bit_cast(uint32_t, pl);
}

// The 1024 partial |x| maxima of a convolution's input as ONE buffer [x1: 512 | x2: 512] -- or, when the two sources'
// maxima live in different buffers (producer-side tags, ops.py), as two: `amax` holds source 1's 512, `amax2` source 2's
// (no concatenation launch).  i indexes float4s (0..255) / float2s (0..511).
__device__ __forceinline__ float4 h2_amax4(const float* __restrict__ amax, const float* __restrict__ amax2, int i) {
  const float* p = (amax2 != nullptr && i >= 128) ? amax2 - 512 : amax;
  return reinterpret_cast<const float4*>(p)[i];
}
__device__ __forceinline__ float2 h2_amax2(const float* __restrict__ amax, const float* __restrict__ amax2, int i) {
  const float* p = (amax2 != nullptr && i >= 256) ? amax2 - 512 : amax;
  return reinterpret_cast<const float2*>(p)[i];
}
