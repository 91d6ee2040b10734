// Pointwise kernels of the "p2" VGG19 pass on pre-split activations (layout: conv_p2.hip): the L1 taps of the perceptual loss
// (lib/losses.py:98-102), the 2x2 max-pools of the stack and their backward, directly on the (hi, lo) planes -- HBM-bound,
// 16-byte units (8 channels of one pixel), two-stage deterministic sums for the loss terms.
//
// Values:   x = (hi + lo / 2^11) * 2^-e   with e = meta[0] of the tensor.
// Gradients are planes tensors too; their scale comes from a bound of the result (the upstream maximum + the L1 step),
// like the convolution's (conv_p2.hip).
#include "common.h"
#include "split_h2.h"

namespace {

constexpr int P2_AMAX0 = 16, P2_NSLOT = 64;

__device__ __forceinline__ void p2_unpack8(const uint4& h, const uint4& l, float inv, float* v) {
  const uint32_t hs[4] = {h.x, h.y, h.z, h.w}, ls[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const h2_f16x2 a = __builtin_bit_cast(h2_f16x2, hs[j]), b = __builtin_bit_cast(h2_f16x2, ls[j]);
    v[2 * j] = ((float)a[0] + (float)b[0] * (1.f / 2048.f)) * inv;
    v[2 * j + 1] = ((float)a[1] + (float)b[1] * (1.f / 2048.f)) * inv;
  }
}
__device__ __forceinline__ void p2_pack8(const float* v, float s, uint4& h, uint4& l) {
  h2_split2(v[0] * s, v[1] * s, h.x, l.x);
  h2_split2(v[2] * s, v[3] * s, h.y, l.y);
  h2_split2(v[4] * s, v[5] * s, h.z, l.z);
  h2_split2(v[6] * s, v[7] * s, h.w, l.w);
}
__device__ __forceinline__ float block_sum4(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float meta_amax(const int* meta) {   // every lane of the wave gets the maximum of the 64 slots
  return wave_max(__int_as_float(meta[P2_AMAX0 + (threadIdx.x & (P2_NSLOT - 1))]));
}
__device__ __forceinline__ void publish_amax(int* meta, float vmax) {
  const float m = wave_max(vmax);
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned*>(meta) + P2_AMAX0 + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (P2_NSLOT - 1)), __float_as_uint(m));
}

__global__ __launch_bounds__(256) void p2_finish_sum_kernel(const float* partial, int nb, float* out, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  const float t = block_sum4(s, red);
  if (threadIdx.x == 0) out[0] += scale * t;
}

// sum |t - p| over the whole padded buffers (the borders hold zeros in both): linear, coalesced
__global__ __launch_bounds__(256) void p2_l1_partial_kernel(const uint4* __restrict__ t, const int* __restrict__ tmeta,
                                                            const uint4* __restrict__ p, const int* __restrict__ pmeta,
                                                            float* __restrict__ partial, size_t plane) {
  __shared__ float red[4];
  const float it = h2_pow2(-tmeta[0]), ip = h2_pow2(-pmeta[0]);
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
    float a[8], b[8];
    p2_unpack8(t[i], t[plane + i], it, a);
    p2_unpack8(p[i], p[plane + i], ip, b);
    s += ((fabsf(a[0] - b[0]) + fabsf(a[1] - b[1])) + (fabsf(a[2] - b[2]) + fabsf(a[3] - b[3]))) +
         ((fabsf(a[4] - b[4]) + fabsf(a[5] - b[5])) + (fabsf(a[6] - b[6]) + fabsf(a[7] - b[7])));
  }
  const float tot = block_sum4(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// 2x2 max-pool of p (per channel: the FIRST maximum in window scan order, as ATen's max_pool2d) and, with t, the L1 partial
// sums of the same pass.  One thread per pooled unit.  The pooled tensor inherits p's scale and maximum.
template <bool L1>
__global__ __launch_bounds__(256) void p2_pool_fwd_kernel(const uint4* __restrict__ t, const int* __restrict__ tmeta,
                                                          const uint4* __restrict__ p, const int* __restrict__ pmeta,
                                                          float* __restrict__ partial, uint4* __restrict__ y, int* __restrict__ ymeta,
                                                          int NCB, int H, int W) {
  __shared__ float red[4];
  const int Ho = H >> 1, Wo = W >> 1, Wp = W + 2, Wpo = Wo + 2;
  const size_t plane = (size_t)NCB * (H + 2) * Wp, plane_o = (size_t)NCB * (Ho + 2) * Wpo, total = (size_t)NCB * Ho * Wo;
  const float ip = h2_pow2(-pmeta[0]);
  const float it = L1 ? h2_pow2(-tmeta[0]) : 0.f;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    if (threadIdx.x == 0) ymeta[0] = pmeta[0];
    ymeta[P2_AMAX0 + threadIdx.x] = pmeta[P2_AMAX0 + threadIdx.x];
  }
  float s = 0.f;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const int xo = (int)(o % Wo);
    size_t r = o / Wo;
    const int yo = (int)(r % Ho);
    const size_t ncb = r / Ho;
    const size_t i0 = (ncb * (H + 2) + (2 * yo + 1)) * Wp + (2 * xo + 1);
    const size_t off[4] = {i0, i0 + 1, i0 + Wp, i0 + Wp + 1};
    uint32_t hs[4][4], ls[4][4];
    float v[4][8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint4 h = p[off[e]], l = p[plane + off[e]];
      hs[e][0] = h.x; hs[e][1] = h.y; hs[e][2] = h.z; hs[e][3] = h.w;
      ls[e][0] = l.x; ls[e][1] = l.y; ls[e][2] = l.z; ls[e][3] = l.w;
      p2_unpack8(h, l, ip, v[e]);
    }
    if (L1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a[8];
        p2_unpack8(t[off[e]], t[plane + off[e]], it, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += fabsf(a[j] - v[e][j]);
      }
    }
    // per channel: pick the window's maximum and carry its (hi, lo) halves over unchanged
    uint32_t oh[4] = {0, 0, 0, 0}, ol[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
      const int k = v[0][j] == m ? 0 : (v[1][j] == m ? 1 : (v[2][j] == m ? 2 : 3));
      const uint32_t hsel = k == 0 ? hs[0][j >> 1] : (k == 1 ? hs[1][j >> 1] : (k == 2 ? hs[2][j >> 1] : hs[3][j >> 1]));
      const uint32_t lsel = k == 0 ? ls[0][j >> 1] : (k == 1 ? ls[1][j >> 1] : (k == 2 ? ls[2][j >> 1] : ls[3][j >> 1]));
      const uint32_t keep = (j & 1) ? 0xFFFF0000u : 0x0000FFFFu;
      oh[j >> 1] |= hsel & keep;
      ol[j >> 1] |= lsel & keep;
    }
    const size_t oo = (ncb * (Ho + 2) + (yo + 1)) * Wpo + (xo + 1);
    y[oo] = make_uint4(oh[0], oh[1], oh[2], oh[3]);
    y[plane_o + oo] = make_uint4(ol[0], ol[1], ol[2], ol[3]);
  }
  if (L1) {
    const float tot = block_sum4(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
  }
}

// The step of a tap's gradient:  g = [add] + c sign(p - t), zeroed where p == 0 (p is a ReLU output: the ReLU backward of the
// layer that produced it); c = gscale * gout[0].  The output's scale: bound = max|add| + |c|.
__global__ __launch_bounds__(256) void p2_l1_bwd_kernel(const uint4* __restrict__ t, const int* __restrict__ tmeta,
                                                        const uint4* __restrict__ p, const int* __restrict__ pmeta,
                                                        const uint4* __restrict__ add, const int* __restrict__ addmeta,
                                                        uint4* __restrict__ g, int* __restrict__ gmeta, float gscale,
                                                        const float* __restrict__ gout, int NCB, int H, int W) {
  const float c = gscale * (gout ? gout[0] : 1.f);
  const float bound = (add ? meta_amax(addmeta) : 0.f) + fabsf(c);
  const int eg = h2_scale_exp(bound);
  const float sg = h2_pow2(eg);
  if (blockIdx.x == 0 && threadIdx.x == 0) gmeta[0] = eg;
  const float it = h2_pow2(-tmeta[0]), ip = h2_pow2(-pmeta[0]), ia = add ? h2_pow2(-addmeta[0]) : 0.f;
  const int Wp = W + 2;
  const size_t plane = (size_t)NCB * (H + 2) * Wp, total = (size_t)NCB * H * W;
  float vmax = 0.f;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const int xw = (int)(o % W);
    size_t r = o / W;
    const int yh = (int)(r % H);
    const size_t i = ((r / H) * (H + 2) + (yh + 1)) * Wp + (xw + 1);
    float a[8], b[8], d[8];
    p2_unpack8(t[i], t[plane + i], it, a);
    p2_unpack8(p[i], p[plane + i], ip, b);
    if (add) p2_unpack8(add[i], add[plane + i], ia, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float df = b[j] - a[j];
      float v = (add ? d[j] : 0.f) + (df > 0.f ? c : (df < 0.f ? -c : 0.f));
      if (!(b[j] > 0.f)) v = 0.f;
      d[j] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
    uint4 h, l;
    p2_pack8(d, sg, h, l);
    g[i] = h;
    g[plane + i] = l;
  }
  publish_amax(gmeta, vmax);
}

// Backward of the 2x2 max-pool on planes, optionally with the L1 step of a tap that feeds the pool (relu1_2, relu2_2):
//   g = route(dy) [+ c sign(p - t)], zeroed where p == 0.  One thread per pooled unit (four output units).
template <bool L1>
__global__ __launch_bounds__(256) void p2_pool_bwd_kernel(const uint4* __restrict__ t, const int* __restrict__ tmeta,
                                                          const uint4* __restrict__ p, const int* __restrict__ pmeta,
                                                          const uint4* __restrict__ dy, const int* __restrict__ dymeta,
                                                          uint4* __restrict__ g, int* __restrict__ gmeta, float gscale,
                                                          const float* __restrict__ gout, int NCB, int H, int W) {
  const float c = L1 ? gscale * (gout ? gout[0] : 1.f) : 0.f;
  const float bound = meta_amax(dymeta) + fabsf(c);
  const int eg = h2_scale_exp(bound);
  const float sg = h2_pow2(eg);
  if (blockIdx.x == 0 && threadIdx.x == 0) gmeta[0] = eg;
  const float ip = h2_pow2(-pmeta[0]), idy = h2_pow2(-dymeta[0]);
  const float it = L1 ? h2_pow2(-tmeta[0]) : 0.f;
  const int Ho = H >> 1, Wo = W >> 1, Wp = W + 2, Wpo = Wo + 2;
  const size_t plane = (size_t)NCB * (H + 2) * Wp, plane_o = (size_t)NCB * (Ho + 2) * Wpo, total = (size_t)NCB * Ho * Wo;
  float vmax = 0.f;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const int xo = (int)(o % Wo);
    size_t r = o / Wo;
    const int yo = (int)(r % Ho);
    const size_t ncb = r / Ho;
    const size_t i0 = (ncb * (H + 2) + (2 * yo + 1)) * Wp + (2 * xo + 1);
    const size_t off[4] = {i0, i0 + 1, i0 + Wp, i0 + Wp + 1};
    const size_t oo = (ncb * (Ho + 2) + (yo + 1)) * Wpo + (xo + 1);
    float v[4][8], gy[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) p2_unpack8(p[off[e]], p[plane + off[e]], ip, v[e]);
    p2_unpack8(dy[oo], dy[plane_o + oo], idy, gy);
    float res[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
      const int k = v[0][j] == m ? 0 : (v[1][j] == m ? 1 : (v[2][j] == m ? 2 : 3));
#pragma unroll
      for (int e = 0; e < 4; ++e) res[e][j] = k == e ? gy[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a[8];
      if (L1) p2_unpack8(t[off[e]], t[plane + off[e]], it, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = res[e][j];
        if (L1) {
          const float df = v[e][j] - a[j];
          x += df > 0.f ? c : (df < 0.f ? -c : 0.f);
        }
        if (!(v[e][j] > 0.f)) x = 0.f;
        res[e][j] = x;
        vmax = fmaxf(vmax, fabsf(x));
      }
      uint4 h, l;
      p2_pack8(res[e], sg, h, l);
      g[off[e]] = h;
      g[plane + off[e]] = l;
    }
  }
  publish_amax(gmeta, vmax);
}

inline unsigned grid_for(size_t n, unsigned cap) {
  size_t b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int vunet_p2_l1_fwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, float* partial, float* out,
                               float weight, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  if (!t || !tmeta || !p || !pmeta || !partial || !out || C % 8 || N < 1) return VUNET_ERR_ARG;
  const size_t plane = (size_t)N * (C / 8) * (H + 2) * (W + 2);
  const unsigned nb = grid_for(plane, 1024);
  hipStream_t st = (hipStream_t)stream;
  VUNET_LAUNCH(p2_l1_partial_kernel, dim3(nb), dim3(256), 0, st, (const uint4*)t, (const int*)tmeta, (const uint4*)p, (const int*)pmeta,
               partial, plane);
  VUNET_LAUNCH(p2_finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, (int)nb, out,
               weight / ((float)N * C * H * W));
  return vunet_check_launch();
}

extern "C" int vunet_p2_pool_fwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, float* partial, float* out,
                                 float weight, void* y, int32_t* ymeta, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  if (!p || !pmeta || !y || !ymeta || C % 8 || N < 1 || (H & 1) || (W & 1)) return VUNET_ERR_ARG;
  if (t && (!tmeta || !partial || !out)) return VUNET_ERR_ARG;
  const int NCB = N * (C / 8);
  const size_t total = (size_t)NCB * (H / 2) * (W / 2);
  const unsigned nb = grid_for(total, 1024);
  hipStream_t st = (hipStream_t)stream;
  if (t) {
    VUNET_LAUNCH(p2_pool_fwd_kernel<true>, dim3(nb), dim3(256), 0, st, (const uint4*)t, (const int*)tmeta, (const uint4*)p,
                 (const int*)pmeta, partial, (uint4*)y, (int*)ymeta, NCB, (int)H, (int)W);
    VUNET_LAUNCH(p2_finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, (int)nb, out,
                 weight / ((float)N * C * H * W));
  } else {
    VUNET_LAUNCH(p2_pool_fwd_kernel<false>, dim3(nb), dim3(256), 0, st, (const uint4*)nullptr, (const int*)nullptr, (const uint4*)p,
                 (const int*)pmeta, (float*)nullptr, (uint4*)y, (int*)ymeta, NCB, (int)H, (int)W);
  }
  return vunet_check_launch();
}

extern "C" int vunet_p2_l1_bwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, const void* add,
                               const int32_t* addmeta, void* g, int32_t* gmeta, float gscale, const float* gout, int32_t N, int32_t C,
                               int32_t H, int32_t W, void* stream) {
  if (!t || !tmeta || !p || !pmeta || !g || !gmeta || C % 8 || N < 1 || (add && !addmeta)) return VUNET_ERR_ARG;
  const int NCB = N * (C / 8);
  const size_t total = (size_t)NCB * H * W;
  VUNET_LAUNCH(p2_l1_bwd_kernel, dim3(grid_for(total, 2048)), dim3(256), 0, (hipStream_t)stream, (const uint4*)t, (const int*)tmeta,
               (const uint4*)p, (const int*)pmeta, (const uint4*)add, (const int*)addmeta, (uint4*)g, (int*)gmeta, gscale, gout, NCB,
               (int)H, (int)W);
  return vunet_check_launch();
}

extern "C" int vunet_p2_pool_bwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, const void* dy,
                                 const int32_t* dymeta, void* g, int32_t* gmeta, float gscale, const float* gout, int32_t N, int32_t C,
                                 int32_t H, int32_t W, void* stream) {
  if (!p || !pmeta || !dy || !dymeta || !g || !gmeta || C % 8 || N < 1 || (H & 1) || (W & 1) || (t && !tmeta)) return VUNET_ERR_ARG;
  const int NCB = N * (C / 8);
  const size_t total = (size_t)NCB * (H / 2) * (W / 2);
  const unsigned nb = grid_for(total, 2048);
  hipStream_t st = (hipStream_t)stream;
  if (t)
    VUNET_LAUNCH(p2_pool_bwd_kernel<true>, dim3(nb), dim3(256), 0, st, (const uint4*)t, (const int*)tmeta, (const uint4*)p,
                 (const int*)pmeta, (const uint4*)dy, (const int*)dymeta, (uint4*)g, (int*)gmeta, gscale, gout, NCB, (int)H, (int)W);
  else
    VUNET_LAUNCH(p2_pool_bwd_kernel<false>, dim3(nb), dim3(256), 0, st, (const uint4*)nullptr, (const int*)nullptr, (const uint4*)p,
                 (const int*)pmeta, (const uint4*)dy, (const int*)dymeta, (uint4*)g, (int*)gmeta, 0.f, (const float*)nullptr, NCB, (int)H,
                 (int)W);
  return vunet_check_launch();
}
