// HBM-bound pointwise / index / reduction kernels of the VUnet hot path.
// Grid-stride loops over <= 2048 blocks (256 CUs x 8), 16-B accesses where the layout allows,
// wavefront-shuffle reductions (64 lanes) with deterministic two-stage sums.
#include "common.h"

static inline dim3 ew_grid(int64_t n, int per_thread = 1) {
  int64_t b = (n + 256ll * per_thread - 1) / (256ll * per_thread);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

#define EW_LOOP(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------ depth/space shuffles
// out[n,c,2h+i,2w+j] = in[n,(2i+j)*Cq+c,h,w]
__global__ void d2s_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W) {
  const int Cq = C >> 2, H2 = 2 * H, W2 = 2 * W;
  const int64_t total = (int64_t)N * C * H * W;
  EW_LOOP(o, total) {
    const int ow = (int)(o % W2);
    int64_t t = o / W2;
    const int oh = (int)(t % H2);
    t /= H2;
    const int c = (int)(t % Cq), n = (int)(t / Cq);
    const int blk = ((oh & 1) << 1) | (ow & 1);
    y[o] = x[(((int64_t)n * C + blk * Cq + c) * H + (oh >> 1)) * W + (ow >> 1)];
  }
}
// out[n,(2i+j)*C+c,h,w] = in[n,c,2h+i,2w+j]     (x: [N,C,H,W], H and W even)
__global__ void s2d_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W) {
  const int Hh = H >> 1, Wh = W >> 1, C4 = 4 * C;
  const int64_t total = (int64_t)N * C * H * W;
  EW_LOOP(o, total) {
    const int w = (int)(o % Wh);
    int64_t t = o / Wh;
    const int hh = (int)(t % Hh);
    t /= Hh;
    const int cc = (int)(t % C4), n = (int)(t / C4);
    const int blk = cc / C, c = cc - blk * C;
    y[o] = x[(((int64_t)n * C + c) * H + 2 * hh + (blk >> 1)) * W + 2 * w + (blk & 1)];
  }
}
// The same shuffle for W % 8 == 0 (every map of the training step): one thread = 8 input columns of two input rows = four
// output columns of the four sub-pixel planes: 16-byte loads and stores, 32-bit index arithmetic once per 16 elements.  The
// per-element form above does six 64-bit divisions per float and reads every other column: 154 us for the 268 MB of the
// 256^2 up-convolution's gradient (1.7 TB/s, profiles/r04_rocprofv3_kernel_stats_1stream.csv), 235 us per step.
__global__ __launch_bounds__(256) void s2d_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H,
                                                      int W) {
  const unsigned Hh = H >> 1, Wh = W >> 1, Wq = Wh >> 2;
  const unsigned total = (unsigned)N * C * Hh * Wq;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned w4 = i % Wq;
    unsigned t = i / Wq;
    const unsigned hh = t % Hh;
    t /= Hh;
    const unsigned c = t % (unsigned)C, n = t / (unsigned)C;
    const float* r0 = x + ((size_t)(n * C + c) * H + 2 * hh) * W + 8 * w4;
    const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(r0 + W), b1 = *reinterpret_cast<const float4*>(r0 + W + 4);
    const size_t plane = (size_t)C * Hh * Wh;
    float* o = y + ((size_t)n * 4 * C + c) * Hh * Wh + (size_t)hh * Wh + 4 * w4;
    *reinterpret_cast<float4*>(o) = make_float4(a0.x, a0.z, a1.x, a1.z);               // (i, j) = (0, 0)
    *reinterpret_cast<float4*>(o + plane) = make_float4(a0.y, a0.w, a1.y, a1.w);       // (0, 1)
    *reinterpret_cast<float4*>(o + 2 * plane) = make_float4(b0.x, b0.z, b1.x, b1.z);   // (1, 0)
    *reinterpret_cast<float4*>(o + 3 * plane) = make_float4(b0.y, b0.w, b1.y, b1.w);   // (1, 1)
  }
}
extern "C" int vunet_depth_to_space(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* st) {
  if (!x || !y || C % 4) return VUNET_ERR_ARG;
  VUNET_LAUNCH(d2s_kernel, ew_grid((int64_t)N * C * H * W), dim3(256), 0, (hipStream_t)st, x, y, N, C, H, W);
  return vunet_check_launch();
}
extern "C" int vunet_space_to_depth(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* st) {
  if (!x || !y || (H & 1) || (W & 1)) return VUNET_ERR_ARG;
  const int64_t total = (int64_t)N * C * H * W;
  if (W % 8 == 0 && total < (1ll << 31) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
    VUNET_LAUNCH(s2d_vec_kernel, ew_grid(total / 16), dim3(256), 0, (hipStream_t)st, x, y, N, C, H, W);
    return vunet_check_launch();
  }
  VUNET_LAUNCH(s2d_kernel, ew_grid(total), dim3(256), 0, (hipStream_t)st, x, y, N, C, H, W);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ simple elementwise
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ yin, float* __restrict__ y,
                             float a, float b, int64_t n) {
  EW_LOOP(i, n) y[i] = a * x[i] + (yin ? b * yin[i] : 0.f);
}
extern "C" int vunet_axpby(const float* x, const float* y_in, float* y, float a, float b, int64_t n, void* st) {
  if (!x || !y) return VUNET_ERR_ARG;
  VUNET_LAUNCH(axpby_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, x, y_in, y, a, b, n);
  return vunet_check_launch();
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int act, float slope, int64_t n) {
  EW_LOOP(i, n) {
    float v = x[i];
    if (act == ACT_ELU) v = elu_f(v);
    else if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
    else if (act == ACT_LRELU) v = v > 0.f ? v : v * slope;
    y[i] = v;
  }
}
extern "C" int vunet_act_fwd(const float* x, float* y, int32_t act, float slope, int64_t n, void* st) {
  if (!x || !y) return VUNET_ERR_ARG;
  VUNET_LAUNCH(act_fwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, x, y, act, slope, n);
  return vunet_check_launch();
}
__global__ void act_bwd_out_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                   int act, float slope, int64_t n) {
  EW_LOOP(i, n) {
    const float o = y[i];
    float g = 1.f;
    if (act == ACT_SIGMOID) g = o * (1.f - o);
    else if (act == ACT_RELU) g = o > 0.f ? 1.f : 0.f;
    else if (act == ACT_LRELU) g = o > 0.f ? 1.f : slope;
    else if (act == ACT_ELU) g = o > 0.f ? 1.f : o + 1.f;
    dx[i] = dy[i] * g;
  }
}
extern "C" int vunet_act_bwd_from_out(const float* y, const float* dy, float* dx, int32_t act, float slope, int64_t n,
                                      void* st) {
  if (!y || !dy || !dx) return VUNET_ERR_ARG;
  VUNET_LAUNCH(act_bwd_out_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, y, dy, dx, act, slope, n);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ reparametrisation
__global__ void reparam_fwd_kernel(const float* mu, const float* ls, const float* eps, float* z, int64_t n) {
  EW_LOOP(i, n) z[i] = eps[i] * __expf(ls[i]) + mu[i];
}
__global__ void reparam_bwd_kernel(const float* dz, const float* ls, const float* eps, float* dmu, float* dls,
                                   int64_t n) {
  EW_LOOP(i, n) {
    const float g = dz[i];
    dmu[i] = g;
    dls[i] = g * eps[i] * __expf(ls[i]);
  }
}
extern "C" int vunet_reparam_fwd(const float* mu, const float* ls, const float* eps, float* z, int64_t n, void* st) {
  if (!mu || !ls || !eps || !z) return VUNET_ERR_ARG;
  VUNET_LAUNCH(reparam_fwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, mu, ls, eps, z, n);
  return vunet_check_launch();
}
extern "C" int vunet_reparam_bwd(const float* dz, const float* ls, const float* eps, float* dmu, float* dls, int64_t n,
                                 void* st) {
  if (!dz || !ls || !eps || !dmu || !dls) return VUNET_ERR_ARG;
  VUNET_LAUNCH(reparam_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, dz, ls, eps, dmu, dls, n);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ L1 mean (perceptual loss taps)
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ partial, int64_t n) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 u = a4[i], v = b4[i];
    s += fabsf(u.x - v.x) + fabsf(u.y - v.y) + fabsf(u.z - v.z) + fabsf(u.w - v.w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += fabsf(a[i] - b[i]);
  const float t = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* partial, int nb, float* out, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  const float t = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] += scale * t;
}
extern "C" int vunet_l1_mean_fwd(const float* a, const float* b, float* partial, float* out, float weight, int64_t n,
                                 void* st) {
  if (!a || !b || !partial || !out || n <= 0) return VUNET_ERR_ARG;
  if ((((uintptr_t)a) | ((uintptr_t)b)) & 15) return VUNET_ERR_ARG;
  int64_t nb = (n / 4 + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  VUNET_LAUNCH(l1_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)st, a, b, partial, n);
  VUNET_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)st, partial, (int)nb, out,
                     weight / (float)n);
  return vunet_check_launch();
}
// L1 tap + the 2x2 max-pool of the same tensor, forward in ONE pass: partial sums of |a - b| (as l1_partial_kernel) and
// y = maxpool2(b); one thread per pool window (b is read once instead of by two kernels)
__global__ __launch_bounds__(256) void l1_pool_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ partial, float* __restrict__ y, int H, int W,
                                                          int64_t n) {
  __shared__ float red[4];
  const int Ho = H >> 1, Wo = W >> 1;
  float s = 0.f;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
    const int ow = (int)(o % Wo);
    const int64_t t = o / Wo;
    const int oh = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const int64_t i0 = (nc * H + 2 * oh) * W + 2 * ow;
    const float2 b0 = *reinterpret_cast<const float2*>(b + i0), b1 = *reinterpret_cast<const float2*>(b + i0 + W);
    const float2 a0 = *reinterpret_cast<const float2*>(a + i0), a1 = *reinterpret_cast<const float2*>(a + i0 + W);
    s += (fabsf(a0.x - b0.x) + fabsf(a0.y - b0.y)) + (fabsf(a1.x - b1.x) + fabsf(a1.y - b1.y));
    y[o] = fmaxf(fmaxf(b0.x, b0.y), fmaxf(b1.x, b1.y));
  }
  const float tsum = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tsum;
}
extern "C" int vunet_l1_pool_fwd(const float* a, const float* b, float* partial, float* out, float* y, float weight,
                                 int32_t NC, int32_t H, int32_t W, void* st) {
  if (!a || !b || !partial || !out || !y || (H & 1) || (W & 1)) return VUNET_ERR_ARG;
  const int64_t n = (int64_t)NC * (H / 2) * (W / 2);
  int64_t nb = (n + 1023) / 1024;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  VUNET_LAUNCH(l1_pool_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)st, a, b, partial, y, H, W, n);
  VUNET_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)st, partial, (int)nb, out,
               weight / (float)(4 * n));
  return vunet_check_launch();
}
__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ add,
                              float* __restrict__ db, float gs, const float* __restrict__ gout, int64_t n,
                              float* __restrict__ amax_out, int relu_mask) {
  if (gout) gs *= gout[0];
  float vmax = 0.f;
  EW_LOOP(i, n) {
    const float bi = b[i];
    const float d = bi - a[i];
    const float g = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    float v = (add ? add[i] : 0.f) + g;
    if (relu_mask && !(bi > 0.f)) v = 0.f;   // b is a ReLU output: the layer that produced it would zero this entry anyway
    db[i] = v;
    vmax = fmaxf(vmax, fabsf(v));
  }
  if (amax_out) {   // partial maxima of |db| for the convolution that reads it (as the conv epilogues publish theirs)
    const float m = wave_max(vmax);
    if ((threadIdx.x & 63) == 0)
      atomicMax(reinterpret_cast<unsigned*>(amax_out) + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & 511u), __float_as_uint(m));
  }
}
extern "C" int vunet_l1_mean_bwd_amax(const float* a, const float* b, const float* add, float* db, float gscale,
                                      const float* gout, int64_t n, float* amax_out, int32_t relu_mask, void* st) {
  if (!a || !b || !db) return VUNET_ERR_ARG;
  VUNET_LAUNCH(l1_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, a, b, add, db, gscale, gout, n, amax_out,
               (int)relu_mask);
  return vunet_check_launch();
}
extern "C" int vunet_l1_mean_bwd(const float* a, const float* b, const float* add, float* db, float gscale,
                                 const float* gout, int64_t n, void* st) {
  return vunet_l1_mean_bwd_amax(a, b, add, db, gscale, gout, n, nullptr, 0, st);
}

// ------------------------------------------------------------------ KL / squared-difference (latents: small)
// two-stage (deterministic) sums over up to 256 workgroups; kind 0: -l + .5(e^{2l} + m^2), kind 1: .5 (p-q)^2
__global__ __launch_bounds__(256) void pair_sum_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               float* __restrict__ partial, int kind, int64_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (kind == 0) {
      const float m = a[i], l = b[i], e = __expf(l);
      s += -l + 0.5f * (e * e + m * m);
    } else {
      const float d = a[i] - b[i];
      s += 0.5f * d * d;
    }
  }
  const float t = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void pair_sum_finish_kernel(const float* partial, int nb, float* out, float scale,
                                                              float offset) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  const float t = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] += scale * t - offset;
}
static int pair_sum(const float* a, const float* b, float* partial, float* out, int kind, float scale, float offset,
                    int64_t n, hipStream_t st) {
  int64_t nb = (n + 2047) / 2048;
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  VUNET_LAUNCH(pair_sum_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, b, partial, kind, n);
  VUNET_LAUNCH(pair_sum_finish_kernel, dim3(1), dim3(256), 0, st, partial, (int)nb, out, scale, offset);
  return vunet_check_launch();
}
extern "C" int vunet_kl_fwd(const float* mu, const float* ls, float* partial, float* out, float weight, int32_t N,
                            int64_t D, void* st) {
  if (!mu || !ls || !partial || !out || N <= 0) return VUNET_ERR_ARG;
  // weight * ( (1/N) sum_all(...) - 0.5 D )
  return pair_sum(mu, ls, partial, out, 0, weight / (float)N, weight * 0.5f * (float)D, (int64_t)N * D, (hipStream_t)st);
}
__global__ void kl_bwd_kernel(const float* mu, const float* ls, float* dmu, float* dls, float gs, const float* gout,
                              int64_t n) {
  if (gout) gs *= gout[0];
  EW_LOOP(i, n) {
    const float e = __expf(ls[i]);
    dmu[i] = gs * mu[i];
    dls[i] = gs * (e * e - 1.f);
  }
}
extern "C" int vunet_kl_bwd(const float* mu, const float* ls, float* dmu, float* dls, float gscale, const float* gout,
                            int64_t n, void* st) {
  if (!mu || !ls || !dmu || !dls) return VUNET_ERR_ARG;
  VUNET_LAUNCH(kl_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, mu, ls, dmu, dls, gscale, gout, n);
  return vunet_check_launch();
}
extern "C" int vunet_sqdiff_fwd(const float* p, const float* q, float* partial, float* out, float weight, int32_t N,
                                int64_t D, void* st) {
  if (!p || !q || !partial || !out || N <= 0) return VUNET_ERR_ARG;
  return pair_sum(p, q, partial, out, 1, weight / (float)N, 0.f, (int64_t)N * D, (hipStream_t)st);
}
__global__ void sqdiff_bwd_kernel(const float* p, const float* q, float* dp, float* dq, float gs, const float* gout,
                                  int64_t n) {
  if (gout) gs *= gout[0];
  EW_LOOP(i, n) {
    const float d = gs * (p[i] - q[i]);
    if (dp) dp[i] = d;
    if (dq) dq[i] = -d;
  }
}
extern "C" int vunet_sqdiff_bwd(const float* p, const float* q, float* dp, float* dq, float gscale, const float* gout,
                                int64_t n, void* st) {
  if (!p || !q) return VUNET_ERR_ARG;
  VUNET_LAUNCH(sqdiff_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, p, q, dp, dq, gscale, gout, n);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ VGG input affine, max-pool
__global__ void vgg_pre_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int64_t n) {
  EW_LOOP(i, n) {
    const int c = (int)((i / HW) % 3);
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
    const float std = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    y[i] = ((x[i] + 1.0f) / 2.0f - mean) / std;
  }
}
extern "C" int vunet_vgg_preprocess(const float* x, float* y, int32_t N, int32_t H, int32_t W, void* st) {
  if (!x || !y) return VUNET_ERR_ARG;
  const int64_t n = (int64_t)N * 3 * H * W;
  VUNET_LAUNCH(vgg_pre_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, x, y, H * W, n);
  return vunet_check_launch();
}
__global__ void vgg_pre_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ add, float* __restrict__ dx,
                                   int HW, int64_t n) {
  EW_LOOP(i, n) {
    const int c = (int)((i / HW) % 3);
    const float std = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    dx[i] = (add ? add[i] : 0.f) + dy[i] * (0.5f / std);
  }
}
extern "C" int vunet_vgg_preprocess_bwd(const float* dy, const float* add, float* dx, int32_t N, int32_t H, int32_t W,
                                        void* st) {
  if (!dy || !dx) return VUNET_ERR_ARG;
  const int64_t n = (int64_t)N * 3 * H * W;
  VUNET_LAUNCH(vgg_pre_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, dy, add, dx, H * W, n);
  return vunet_check_launch();
}
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int64_t n) {
  const int Ho = H >> 1, Wo = W >> 1;
  EW_LOOP(o, n) {
    const int ow = (int)(o % Wo);
    const int64_t t = o / Wo;
    const int oh = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const float* r0 = x + (nc * H + 2 * oh) * W + 2 * ow;
    const float2 a = *reinterpret_cast<const float2*>(r0);
    const float2 b = *reinterpret_cast<const float2*>(r0 + W);
    y[o] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
  }
}
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                    const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int64_t n,
                                    int relu_mask) {
  const int Ho = H >> 1, Wo = W >> 1;
  EW_LOOP(o, n) {
    const int ow = (int)(o % Wo);
    const int64_t t = o / Wo;
    const int oh = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const int64_t i0 = (nc * H + 2 * oh) * W + 2 * ow;
    const float m = y[o];
    const float g = (relu_mask && !(m > 0.f)) ? 0.f : dy[o];   // x is a ReLU output: its layer's backward zeroes x <= 0
    const float2 a = *reinterpret_cast<const float2*>(x + i0);
    const float2 b = *reinterpret_cast<const float2*>(x + i0 + W);
    // first maximum in window scan order gets the gradient (ATen max_pool2d semantics)
    const int k = a.x == m ? 0 : (a.y == m ? 1 : (b.x == m ? 2 : 3));
    *reinterpret_cast<float2*>(dx + i0) = make_float2(k == 0 ? g : 0.f, k == 1 ? g : 0.f);
    *reinterpret_cast<float2*>(dx + i0 + W) = make_float2(k == 2 ? g : 0.f, k == 3 ? g : 0.f);
  }
}
extern "C" int vunet_maxpool2_fwd(const float* x, float* y, int32_t NC, int32_t H, int32_t W, void* st) {
  if (!x || !y || (H & 1) || (W & 1)) return VUNET_ERR_ARG;
  const int64_t n = (int64_t)NC * (H / 2) * (W / 2);
  VUNET_LAUNCH(maxpool2_fwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, x, y, H, W, n);
  return vunet_check_launch();
}
// L1 tap + the 2x2 max-pool that reads the same tensor, backward in ONE pass (VGG19 relu1_2 / relu2_2, which feed both a loss
// term and a pool): db = route(dy_pool) + gs * sign(b - a), optionally zeroed where b <= 0 (b is a ReLU output), partial
// maxima of |db| published.  One thread per pool window; the window maximum is recomputed from b (no read of the pooled y).
__global__ void l1_pool_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dyp,
                                   float* __restrict__ db, float gs, const float* __restrict__ gout, int H, int W, int64_t n,
                                   float* __restrict__ amax_out, int relu_mask) {
  if (gout) gs *= gout[0];
  const int Ho = H >> 1, Wo = W >> 1;
  float vmax = 0.f;
  EW_LOOP(o, n) {
    const int ow = (int)(o % Wo);
    const int64_t t = o / Wo;
    const int oh = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const int64_t i0 = (nc * H + 2 * oh) * W + 2 * ow;
    const float2 b0 = *reinterpret_cast<const float2*>(b + i0), b1 = *reinterpret_cast<const float2*>(b + i0 + W);
    const float2 a0 = *reinterpret_cast<const float2*>(a + i0), a1 = *reinterpret_cast<const float2*>(a + i0 + W);
    const float m = fmaxf(fmaxf(b0.x, b0.y), fmaxf(b1.x, b1.y));
    const float g = dyp ? dyp[o] : 0.f;
    const int k = b0.x == m ? 0 : (b0.y == m ? 1 : (b1.x == m ? 2 : 3));   // first maximum in scan order (ATen max_pool2d)
    const float bv[4] = {b0.x, b0.y, b1.x, b1.y}, av[4] = {a0.x, a0.y, a1.x, a1.y};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = bv[e] - av[e];
      float r = (k == e ? g : 0.f) + (d > 0.f ? gs : (d < 0.f ? -gs : 0.f));
      if (relu_mask && !(bv[e] > 0.f)) r = 0.f;
      v[e] = r;
      vmax = fmaxf(vmax, fabsf(r));
    }
    *reinterpret_cast<float2*>(db + i0) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2*>(db + i0 + W) = make_float2(v[2], v[3]);
  }
  if (amax_out) {
    const float m_ = wave_max(vmax);
    if ((threadIdx.x & 63) == 0)
      atomicMax(reinterpret_cast<unsigned*>(amax_out) + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & 511u), __float_as_uint(m_));
  }
}
extern "C" int vunet_l1_pool_bwd(const float* a, const float* b, const float* dy_pool, float* db, float gscale,
                                 const float* gout, int32_t NC, int32_t H, int32_t W, float* amax_out, int32_t relu_mask,
                                 void* st) {
  if (!a || !b || !db || (H & 1) || (W & 1)) return VUNET_ERR_ARG;
  const int64_t n = (int64_t)NC * (H / 2) * (W / 2);
  VUNET_LAUNCH(l1_pool_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, a, b, dy_pool, db, gscale, gout, H, W, n,
               amax_out, (int)relu_mask);
  return vunet_check_launch();
}

static int maxpool2_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t NC, int32_t H, int32_t W,
                        int relu_mask, void* st) {
  if (!x || !y || !dy || !dx || (H & 1) || (W & 1)) return VUNET_ERR_ARG;
  const int64_t n = (int64_t)NC * (H / 2) * (W / 2);
  VUNET_LAUNCH(maxpool2_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, x, y, dy, dx, H, W, n, relu_mask);
  return vunet_check_launch();
}
extern "C" int vunet_maxpool2_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t NC, int32_t H,
                                  int32_t W, void* st) {
  return maxpool2_bwd(x, y, dy, dx, NC, H, W, 0, st);
}
extern "C" int vunet_maxpool2_bwd_relu(const float* x, const float* y, const float* dy, float* dx, int32_t NC, int32_t H,
                                       int32_t W, void* st) {
  return maxpool2_bwd(x, y, dy, dx, NC, H, W, 1, st);
}

// ------------------------------------------------------------------ InstanceNorm2d (affine=False)
// one workgroup per (n,c) plane; two passes over the plane (L2-resident), wavefront-shuffle sums
__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           float* __restrict__ stats, int HW, float eps) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * HW;
  float* yr = y + (size_t)blockIdx.x * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) s += xr[i];
  const float mean = block_sum_256(s, red) / (float)HW;
  float v = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) { const float d = xr[i] - mean; v += d * d; }
  const float var = block_sum_256(v, red) / (float)HW;
  const float rstd = rsqrtf(var + eps);
  for (int i = threadIdx.x; i < HW; i += 256) yr[i] = (xr[i] - mean) * rstd;
  if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, float* __restrict__ dx,
                                                           int HW) {
  __shared__ float red[4];
  const size_t o = (size_t)blockIdx.x * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) { const float g = dy[o + i]; s1 += g; s2 += g * y[o + i]; }
  const float m1 = block_sum_256(s1, red) / (float)HW;
  const float m2 = block_sum_256(s2, red) / (float)HW;
  const float rstd = stats[2 * blockIdx.x + 1];
  for (int i = threadIdx.x; i < HW; i += 256) dx[o + i] = rstd * (dy[o + i] - m1 - y[o + i] * m2);
}
extern "C" int vunet_instnorm_fwd(const float* x, float* y, float* stats, int32_t NC, int32_t HW, float eps, void* st) {
  if (!x || !y || !stats || NC <= 0 || HW <= 0) return VUNET_ERR_ARG;
  VUNET_LAUNCH(instnorm_fwd_kernel, dim3(NC), dim3(256), 0, (hipStream_t)st, x, y, stats, HW, eps);
  return vunet_check_launch();
}
extern "C" int vunet_instnorm_bwd(const float* y, const float* dy, const float* stats, float* dx, int32_t NC,
                                  int32_t HW, void* st) {
  if (!y || !dy || !stats || !dx) return VUNET_ERR_ARG;
  VUNET_LAUNCH(instnorm_bwd_kernel, dim3(NC), dim3(256), 0, (hipStream_t)st, y, dy, stats, dx, HW);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ dropout step counter (see InAct::step)
const uint32_t* g_vunet_drop_step = nullptr;

extern "C" int vunet_set_dropout_step(const uint32_t* step_dev) {
  g_vunet_drop_step = step_dev;
  return VUNET_OK;
}

// ------------------------------------------------------------------ tuning knobs (tests / kernel tuning)
int g_vunet_tune[16] = {0};

extern "C" int vunet_set_tuning(int32_t key, int32_t value) {
  if (key < 0 || key >= VUNET_TUNE_COUNT) return VUNET_ERR_ARG;
  g_vunet_tune[key] = value;
  return VUNET_OK;
}

// ------------------------------------------------------------------ fused Adam over a flat buffer
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float step_size, float b1, float b2, float eps,
                            float wd, float inv_sqrt_bc2, float gscale) {
  EW_LOOP(i, n) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}
extern "C" int vunet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t step,
                               float grad_scale, void* st) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return VUNET_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  VUNET_LAUNCH(adam_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, param, grad, exp_avg, exp_avg_sq, n,
                     (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale);
  return vunet_check_launch();
}

// The same update with the learning rate and the step count read from DEVICE memory (float64 / int64 scalars): nothing
// that changes from step to step is a launch argument, so the launch can sit in a captured hipGraph.  Bias corrections
// are formed in double exactly as the host version forms them.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, int64_t n, const double* __restrict__ lr,
                                const int64_t* __restrict__ step, float b1, float b2, float eps, float wd, float gscale) {
  const double t = (double)*step;
  const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
  const float step_size = (float)((double)(float)*lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  EW_LOOP(i, n) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}
extern "C" int vunet_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const double* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                   const int64_t* step_dev, float grad_scale, void* st) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return VUNET_ERR_ARG;
  VUNET_LAUNCH(adam_dev_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, param, grad, exp_avg, exp_avg_sq, n, lr_dev,
               step_dev, beta1, beta2, eps, weight_decay, grad_scale);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ |x| maxima for the split-fp16 convolution scale
// 1024 workgroups: 0..511 walk x1, 512..1023 walk x2; each writes ONE partial maximum (no atomics, nothing to clear).
// NaNs are ignored by fmaxf (they reach the convolution's output through the data itself).
__global__ __launch_bounds__(256) void absmax_partials_kernel(const float* __restrict__ x1, int64_t n1,
                                                              const float* __restrict__ x2, int64_t n2,
                                                              float* __restrict__ out) {
  __shared__ float red[4];
  const bool second = blockIdx.x >= 512;
  const float* __restrict__ x = second ? x2 : x1;
  const int64_t n = second ? n2 : n1;
  const int b = second ? blockIdx.x - 512 : blockIdx.x;
  float m = 0.f;
  if (x) {
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < n4; i += 512 * 256) {
      const float4 v = x4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t i = 4 * n4 + (int64_t)b * 256 + threadIdx.x; i < n; i += 512 * 256) m = fmaxf(m, fabsf(x[i]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
extern "C" int vunet_absmax_partials(const float* x1, int64_t n1, const float* x2, int64_t n2, float* out, void* st) {
  if (!x1 || !out || n1 < 0 || n2 < 0) return VUNET_ERR_ARG;
  VUNET_LAUNCH(absmax_partials_kernel, dim3(1024), dim3(256), 0, (hipStream_t)st, x1, n1, x2, x2 ? n2 : 0, out);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ misc
__global__ void dropout_mask_kernel(float* mask, int64_t n, uint32_t thresh, uint32_t seed, const uint32_t* step) {
  if (step) seed += (*step) * VUNET_DROP_STEP_MUL;
  EW_LOOP(i, n) mask[i] = vunet_hash_u32((uint32_t)i + seed) >= thresh ? 1.f : 0.f;
}
extern "C" int vunet_dropout_mask(float* mask, int64_t n, float p, uint32_t seed, void* st) {
  if (!mask) return VUNET_ERR_ARG;
  const InAct a = make_inact(ACT_NONE, 0.f, p, seed);
  VUNET_LAUNCH(dropout_mask_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, mask, n, a.thresh, seed, a.step);
  return vunet_check_launch();
}
__global__ void u8_to_unit_kernel(const uint8_t* in, float* out, int64_t n) {
  EW_LOOP(i, n) out[i] = ((float)in[i] / 255.0f) * 2.0f - 1.0f;
}
extern "C" int vunet_u8_to_unit(const uint8_t* in, float* out, int64_t n, void* st) {
  if (!in || !out) return VUNET_ERR_ARG;
  VUNET_LAUNCH(u8_to_unit_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, in, out, n);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ bilinear 2x up-sampling
// nn.Upsample(scale_factor=2, mode="bilinear") with align_corners False (lib/modules.py:172-175, the non-sub-pixel
// branch of Upsample): source coordinate (o + 0.5) / 2 - 0.5 clamped at 0, neighbours i0 = floor, i1 = min(i0 + 1, n - 1).
// For output 2i: (i - 1, i) with weights (0.25, 0.75) (i = 0: all on 0); for 2i + 1: (i, i + 1) with (0.75, 0.25).
__device__ __forceinline__ void bil2_src(int o, int n, int& i0, int& i1, float& l) {
  const int i = o >> 1;
  if (o & 1) { i0 = i; l = 0.25f; }
  else if (i == 0) { i0 = 0; l = 0.f; }
  else { i0 = i - 1; l = 0.75f; }
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
}

__global__ void bilinear2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t NC, int H, int W) {
  const int H2 = 2 * H, W2 = 2 * W;
  EW_LOOP(o, NC * H2 * W2) {
    const int ox = (int)(o % W2);
    const int64_t t = o / W2;
    const int oy = (int)(t % H2);
    const float* p = x + (t / H2) * H * W;
    int y0, y1, x0, x1;
    float ly, lx;
    bil2_src(oy, H, y0, y1, ly);
    bil2_src(ox, W, x0, x1, lx);
    // same association as ATen's upsample_bilinear2d: h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (...)
    const float top = (1.f - lx) * p[y0 * W + x0] + lx * p[y0 * W + x1];
    const float bot = (1.f - lx) * p[y1 * W + x0] + lx * p[y1 * W + x1];
    y[o] = (1.f - ly) * top + ly * bot;
  }
}

// gather form of the adjoint: input pixel i receives from outputs 2i-1 (0.25), 2i (0.75; 1 if i == 0), 2i+1 (0.75; 1 if
// i == n-1), 2i+2 (0.25) -- deterministic, no atomics
__device__ __forceinline__ void bil2_adj(int i, int n, int o[4], float w[4]) {
  o[0] = 2 * i - 1; w[0] = i >= 1 ? 0.25f : 0.f;
  o[1] = 2 * i;     w[1] = i == 0 ? 1.f : 0.75f;
  o[2] = 2 * i + 1; w[2] = i == n - 1 ? 1.f : 0.75f;
  o[3] = 2 * i + 2; w[3] = i <= n - 2 ? 0.25f : 0.f;
  if (o[0] < 0) o[0] = 0;
  if (o[3] > 2 * n - 1) o[3] = 2 * n - 1;
}

__global__ void bilinear2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t NC, int H, int W) {
  const int W2 = 2 * W;
  EW_LOOP(e, NC * H * W) {
    const int ix = (int)(e % W);
    const int64_t t = e / W;
    const int iy = (int)(t % H);
    const float* p = dy + (t / H) * 4 * H * W;
    int oy[4], ox[4];
    float wy[4], wx[4];
    bil2_adj(iy, H, oy, wy);
    bil2_adj(ix, W, ox, wx);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float r = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) r += wx[b] * p[oy[a] * W2 + ox[b]];
      s += wy[a] * r;
    }
    dx[e] = s;
  }
}

extern "C" int vunet_upsample_bilinear2x_fwd(const float* x, float* y, int64_t NC, int32_t H, int32_t W, void* st) {
  if (!x || !y || NC < 1 || H < 1 || W < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(bilinear2x_fwd_kernel, ew_grid(NC * 4 * H * W), dim3(256), 0, (hipStream_t)st, x, y, NC, H, W);
  return vunet_check_launch();
}

extern "C" int vunet_upsample_bilinear2x_bwd(const float* dy, float* dx, int64_t NC, int32_t H, int32_t W, void* st) {
  if (!dy || !dx || NC < 1 || H < 1 || W < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(bilinear2x_bwd_kernel, ew_grid(NC * H * W), dim3(256), 0, (hipStream_t)st, dy, dx, NC, H, W);
  return vunet_check_launch();
}

extern "C" int vunet_abi_version(void) { return 12; }   // 11: include/vunet_seq_train.h (config 4 training); 12: vunet_seq_dx_finish

// ------------------------------------------------------------------ window crop with a device-resident corner
__global__ void crop_window_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W, int P,
                                   const int32_t* __restrict__ off, int bwd) {
  const int oy = off[0], ox = off[1];
  const int64_t n = (int64_t)planes * P * P;
  EW_LOOP(i, n) {
    const int j = (int)(i % P), r = (int)((i / P) % P);
    const int64_t pl = i / ((int64_t)P * P);
    const int64_t src = (pl * H + (oy + r)) * W + (ox + j);
    if (bwd) y[src] = x[i];     // (x = dy, y = dx)
    else y[i] = x[src];
  }
}
extern "C" int vunet_crop_window(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t P, const int32_t* off,
                                 void* st) {
  if (!x || !y || !off || planes < 1 || P < 1 || P > H || P > W) return VUNET_ERR_ARG;
  VUNET_LAUNCH(crop_window_kernel, ew_grid((int64_t)planes * P * P), dim3(256), 0, (hipStream_t)st, x, y, planes, H, W, P, off, 0);
  return vunet_check_launch();
}
extern "C" int vunet_crop_window_bwd(const float* dy, float* dx, int32_t planes, int32_t H, int32_t W, int32_t P,
                                     const int32_t* off, void* st) {
  if (!dy || !dx || !off || planes < 1 || P < 1 || P > H || P > W) return VUNET_ERR_ARG;
  VUNET_LAUNCH(crop_window_kernel, ew_grid((int64_t)planes * P * P), dim3(256), 0, (hipStream_t)st, dy, dx, planes, H, W, P, off, 1);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ the step's scalar glue as single launches
// loss = ll_weight * sum_i term[i] (+ gamma * kl) -- experiments/shape_and_pose_net.py:391-405 -- and its gradient.  One
// thread: six to eight scalars.  The sum runs in double and is rounded once (any fp32 summation order is within 1 ulp of it).
struct LossTermPtrs {
  const float* t[VUNET_LOSS_MAX_TERMS];
};
__global__ void total_loss_kernel(LossTermPtrs a, int n, const float* __restrict__ kl, const float* __restrict__ gamma,
                                  float gamma_const, float ll_weight, int use_kl, float* __restrict__ loss,
                                  float* __restrict__ llo) {
  if (threadIdx.x | blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += (double)a.t[i][0];
  const float ll = __fmul_rn(ll_weight, (float)s);
  llo[0] = ll;
  const float g = gamma ? gamma[0] : gamma_const;
  loss[0] = use_kl ? __fadd_rn(ll, __fmul_rn(g, kl[0])) : ll;   // (separately rounded, as the reference's mul / add)
}
__global__ void total_loss_bwd_kernel(const float* __restrict__ g_loss, const float* __restrict__ g_ll, int n,
                                      const float* __restrict__ gamma, float gamma_const, float ll_weight, int use_kl,
                                      float* __restrict__ d) {
  const int i = threadIdx.x;
  if (blockIdx.x || i > n) return;
  const float gl = g_loss ? g_loss[0] : 0.f, gll = g_ll ? g_ll[0] : 0.f;
  if (i < n) d[i] = __fmul_rn(ll_weight, __fadd_rn(gl, gll));
  else d[n] = use_kl ? gl * (gamma ? gamma[0] : gamma_const) : 0.f;
}
extern "C" int vunet_total_loss(const float* const* terms, int32_t n, const float* kl, const float* gamma, float gamma_const,
                                float ll_weight, int32_t use_kl, float* loss, float* ll, void* st) {
  if (!terms || n < 1 || n > VUNET_LOSS_MAX_TERMS || !loss || !ll || (use_kl && !kl)) return VUNET_ERR_ARG;
  LossTermPtrs a;
  for (int i = 0; i < VUNET_LOSS_MAX_TERMS; ++i) a.t[i] = i < n ? terms[i] : nullptr;
  for (int i = 0; i < n; ++i)
    if (!a.t[i]) return VUNET_ERR_ARG;
  VUNET_LAUNCH(total_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, a, n, kl, gamma, gamma_const, ll_weight, use_kl, loss,
               ll);
  return vunet_check_launch();
}
extern "C" int vunet_total_loss_bwd(const float* g_loss, const float* g_ll, int32_t n, const float* gamma, float gamma_const,
                                    float ll_weight, int32_t use_kl, float* d, void* st) {
  if (n < 1 || n > VUNET_LOSS_MAX_TERMS || !d) return VUNET_ERR_ARG;
  VUNET_LAUNCH(total_loss_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, g_loss, g_ll, n, gamma, gamma_const, ll_weight,
               use_kl, d);
  return vunet_check_launch();
}

// gamma <- max(gamma - gamma_step * (information_max - avg_kl), 0), in place (experiments/shape_and_pose_net.py:82-85, 442)
__global__ void gamma_update_kernel(float* __restrict__ gamma, const float* __restrict__ imax, const float* __restrict__ kl,
                                    float gamma_step) {
  if (threadIdx.x | blockIdx.x) return;
  const float g = __fsub_rn(gamma[0], __fmul_rn(gamma_step, __fsub_rn(imax[0], kl[0])));   // no fma: the reference's rounding
  gamma[0] = g > 0.f ? g : (g != g ? g : 0.f);     // torch.clamp(min=0) passes NaN through
}
extern "C" int vunet_gamma_update(float* gamma, const float* imax, const float* avg_kl, float gamma_step, void* st) {
  if (!gamma || !imax || !avg_kl) return VUNET_ERR_ARG;
  VUNET_LAUNCH(gamma_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, gamma, imax, avg_kl, gamma_step);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ unit-variance latent sample with in-kernel noise
// z = mu + eps, eps ~ N(0, 1) drawn here (models/vunets.py:151-156: `p + torch.randn_like(p)`): Box-Muller on two 32-bit hashes of
// (element index, seed, step) -- the step read from the device counter of vunet_set_dropout_step when one is set, so the launch
// arguments stay constant from step to step (captured hipGraph) while every replay draws fresh noise.  One launch instead of
// a zero-fill, a generator launch and the reparametrisation kernel; the backward is the identity.
__global__ void unit_sample_kernel(const float* __restrict__ mu, const float* __restrict__ ls, float* __restrict__ z,
                                   float* __restrict__ eps_out, int64_t n, uint32_t seed, const uint32_t* __restrict__ step) {
  if (step) seed += step[0] * VUNET_DROP_STEP_MUL;
  EW_LOOP(i, n) {
    const uint32_t a = vunet_hash_u32((uint32_t)(2 * i) + seed), b = vunet_hash_u32((uint32_t)(2 * i + 1) + (seed ^ 0x68E31DA4u));
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);    // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);             // [0, 1)
    const float e = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958648f * u2);
    if (eps_out) eps_out[i] = e;
    z[i] = ls ? e * __expf(ls[i]) + mu[i] : mu[i] + e;     // (with a log-std: vunet_reparam_fwd's expression)
  }
}
extern "C" int vunet_unit_sample(const float* mu, const float* logstd, float* z, float* eps_out, int64_t n, uint32_t seed,
                                 void* st) {
  if (!mu || !z || n < 0) return VUNET_ERR_ARG;
  if (n == 0) return VUNET_OK;
  VUNET_LAUNCH(unit_sample_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)st, mu, logstd, z, eps_out, n, seed, g_vunet_drop_step);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ this step's schedule values -> device, one launch
__global__ void set_schedule_kernel(double* lr, double lr_v, float* imax, float imax_v, int32_t* step, int32_t step_v) {
  if (threadIdx.x | blockIdx.x) return;
  if (lr) lr[0] = lr_v;
  if (imax) imax[0] = imax_v;
  if (step) step[0] = step_v;
}
extern "C" int vunet_set_schedule(double* lr_dev, double lr, float* imax_dev, float imax, int32_t* step_dev, int32_t step,
                                  void* st) {
  VUNET_LAUNCH(set_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, lr_dev, lr, imax_dev, imax, step_dev, step);
  return vunet_check_launch();
}

// ------------------------------------------------------------------ a fan-out's gradients summed in one launch, maxima included
// out = ((g0 + g1) + g2) + g3 over up to four tensors of one shape, in that order (the autograd engine's accumulation order
// is the caller's to reproduce), with the partial |out| maxima the split-fp16 kernels scale by published like a convolution's
// (amax_out: 512 zero-initialised floats, atomic max on the bit patterns; may be NULL): replaces the engine's aten::add_
// launches and the vunet_absmax_partials pass of the next data gradient on the small maps of the bottleneck.
struct SumSrcs {
  const float* s[4];
};
__global__ __launch_bounds__(256) void sum_amax_kernel(SumSrcs a, int n, float* __restrict__ out, float* __restrict__ amax_out,
                                                       int64_t numel) {
  float m = 0.f;
  EW_LOOP(i, numel) {
    float v = a.s[0][i] + a.s[1][i];
    if (n > 2) v += a.s[2][i];
    if (n > 3) v += a.s[3][i];
    out[i] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (amax_out) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0)
      atomicMax(reinterpret_cast<unsigned*>(amax_out) + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & 511u), __float_as_uint(m));
  }
}
extern "C" int vunet_sum_amax(const float* const* srcs, int32_t n, float* out, float* amax_out, int64_t numel, void* st) {
  if (!srcs || n < 2 || n > 4 || !out || numel < 0) return VUNET_ERR_ARG;
  SumSrcs a;
  for (int i = 0; i < 4; ++i) {
    a.s[i] = i < n ? srcs[i] : nullptr;
    if (i < n && !a.s[i]) return VUNET_ERR_ARG;
  }
  if (numel == 0) return VUNET_OK;
  VUNET_LAUNCH(sum_amax_kernel, ew_grid(numel), dim3(256), 0, (hipStream_t)st, a, n, out, amax_out, numel);
  return vunet_check_launch();
}
