// Convolutions with a 3-channel side: plain fp32 VALU kernels (no MFMA).
//
// The image ends of the networks -- VGG19 conv1_1 (3 -> 64) and its data gradient (64 -> 3), the generator's
// out_conv (32 -> 3, models/vunets.py:188), the 1x1 input layers (3 -> 32, :118) -- put 3 channels on one side of
// the product.  On the MFMA kernels that side is padded to a 32-row tile (>= 90 % of the matrix work multiplies
// zeros: 8.7 TFLOP/s on 64 -> 3 @ 256^2), while the layer itself is memory bound (256^2 x 64 channels of
// activations for 1728 FMAs per pixel).  Here one lane owns one pixel:
//   thin_m : M <= 4 outputs per pixel, K walks the input channels; the weights of a (channel, tap) are wave-uniform
//            (scalar loads, SGPR operands of v_fmac), the 9 taps are row-coalesced loads served by L1;
//   thin_k : 3 input channels, the (<= 27) input values of a pixel stay in registers and M walks the outputs
//            in blocks of 8 accumulators.
// Same weight layout (K-major wt_f / wt_d), same epilogue (store_out) as the MFMA kernels.
#include <string.h>

#include "conv_common.h"
#include "split_h2.h"

// ---- M <= 4, 3x3 / stride 1 / pad 1, forward (MODE 0) or data gradient (MODE 1: mirrored taps) ----------------
template <int MODE>
__global__ __launch_bounds__(256) void conv_thin_m_kernel(const GatherArgs a) {
  const vunet_conv_desc& d = a.d;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;
  const int tiles_w = (W + 31) / 32, tiles_h = (H + 7) / 8;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_w;
  const int ty = (bid / tiles_w) % tiles_h;
  const int n = bid / (tiles_w * tiles_h);
  const int ow = tx * 32 + (threadIdx.x & 31), oh = ty * 8 + (threadIdx.x >> 5);
  const bool inside = ow < W && oh < H;
  // tap offsets and validity of this pixel (chunk-invariant)
  int off[9];
  uint32_t ok = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
    const bool v = inside && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    off[t] = v ? ih * W + iw : 0;
    ok |= (v ? 1u : 0u) << t;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int C = d.C1, Cp = (C + 1) & ~1;
  const float* __restrict__ xs = a.x1 + (size_t)n * C * HW;
  const float* __restrict__ wt = a.wt + d.m_off;
  for (int c = 0; c < C; ++c) {
    float xv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float v = xs[(size_t)c * HW + off[t]];
      xv[t] = ((ok >> t) & 1u) ? v : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int tap = MODE == 0 ? t : 8 - t;   // data gradient: the taps mirrored
      const float* __restrict__ w = wt + (size_t)(tap * Cp + c) * d.Mpad;   // wave-uniform: scalar loads
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = fmaf(w[m], xv[t], acc[m]);
    }
  }
  if (!inside) return;
  PixGeo g;
  g.n = n;
  g.oh = oh;
  g.ow = ow;
  g.valid = true;
#pragma unroll
  for (int m = 0; m < 4; ++m)
    if (m < d.M) store_out(a, g, m, acc[m]);
}

// ---- conv_thin_m_kernel with R vertically adjacent output pixels per lane: the (R + 2) x 3 input window of a channel is
//      loaded once for the R pixels (R = 4: 4.5 L1 loads per pixel and channel instead of 9) and the next channel's window
//      is requested before this channel's FMAs.  Same fmaf chain per output as the one-pixel form: bit-identical results.
template <int MODE, int R>
__global__ __launch_bounds__(256) void conv_thin_mv_kernel(const GatherArgs a) {
  const vunet_conv_desc& d = a.d;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;
  const int tiles_w = (W + 31) / 32, tiles_h = (H + 8 * R - 1) / (8 * R);
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_w;
  const int ty = (bid / tiles_w) % tiles_h;
  const int n = bid / (tiles_w * tiles_h);
  const int ow = tx * 32 + (threadIdx.x & 31), oh0 = ty * 8 * R + (threadIdx.x >> 5) * R;
  int off[R + 2][3];
  uint32_t ok = 0;
#pragma unroll
  for (int r = 0; r < R + 2; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ih = oh0 - 1 + r, iw = ow - 1 + k;
      const bool v = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      off[r][k] = v ? ih * W + iw : 0;
      ok |= (v ? 1u : 0u) << (r * 3 + k);
    }
  float acc[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[r][m] = 0.f;
  const int C = d.C1, Cp = (C + 1) & ~1;
  const float* __restrict__ xs = a.x1 + (size_t)n * C * HW;
  const float* __restrict__ wt = a.wt + d.m_off;
  float xn[R + 2][3];
#pragma unroll
  for (int r = 0; r < R + 2; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) xn[r][k] = xs[off[r][k]];
  for (int c = 0; c < C; ++c) {
    float xv[R + 2][3];
#pragma unroll
    for (int r = 0; r < R + 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) xv[r][k] = ((ok >> (r * 3 + k)) & 1u) ? xn[r][k] : 0.f;
    const int cn = c + 1 < C ? c + 1 : c;   // (the last iteration re-reads its own window: unconditional loads)
#pragma unroll
    for (int r = 0; r < R + 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) xn[r][k] = xs[(size_t)cn * HW + off[r][k]];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int tap = MODE == 0 ? t : 8 - t;   // data gradient: the taps mirrored
      const float* __restrict__ w = wt + (size_t)(tap * Cp + c) * d.Mpad;   // wave-uniform: scalar loads
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[r][m] = fmaf(w[m], xv[r + t / 3][t % 3], acc[r][m]);
    }
  }
  if (ow >= W) return;
  PixGeo g;
  g.n = n;
  g.ow = ow;
  g.valid = true;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    g.oh = oh0 + r;
    if (g.oh < H) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < d.M) store_out(a, g, m, acc[r][m]);
    }
  }
}

// ---- CI (= 3: RGB) input channels, 3x3 pad 1 (KS 3) or 1x1 (KS 1), stride 1; forward, or (3x3) the data gradient of
//      a layer with 3 OUTPUT channels (dd.out_conv, models/vunets.py:281: dy has 3 channels) with the taps mirrored ------
template <int KS, int CI>
__global__ __launch_bounds__(256, KS == 3 ? 2 : 4) void conv_thin_k_kernel(const GatherArgs a) {
  constexpr int T = KS * KS;
  const vunet_conv_desc& d = a.d;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;
  const int tiles_w = (W + 31) / 32, tiles_h = (H + 7) / 8;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_w;
  const int ty = (bid / tiles_w) % tiles_h;
  const int n = bid / (tiles_w * tiles_h);
  const int ow = tx * 32 + (threadIdx.x & 31), oh = ty * 8 + (threadIdx.x >> 5);
  const bool inside = ow < W && oh < H;
  constexpr int C = CI, Cp = (C + 1) & ~1;
  float xv[CI][T];
  const float* __restrict__ xs = a.x1 + (size_t)n * C * HW;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int ih = oh + (KS == 3 ? t / 3 - 1 : 0), iw = ow + (KS == 3 ? t % 3 - 1 : 0);
    const bool v = inside && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const int off = v ? ih * W + iw : 0;
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const float x = xs[c * HW + off];
      xv[c][t] = v ? x : 0.f;
    }
  }
  const bool relu = d.mode == 0 && d.out_act == ACT_RELU;
  // the whole [T*CI][M] weight block (<= 27 x 128 floats) goes to LDS once; every lane then reads the same
  // address (LDS broadcast).  Scalar loads are not an option here: the kernel stores to y between the passes.
  __shared__ __attribute__((aligned(16))) float wL[T * CI * 128];
  const int Mr = (d.M + 7) & ~7;
  for (int e = threadIdx.x; e < T * CI * Mr; e += 256) {
    const int m = e % Mr, r = e / Mr;           // r = t*CI + c
    const int t = r / CI, c = r - t * CI;
    const int tw = d.mode == 0 ? t : T - 1 - t;   // data gradient: the weight of tap t is stored under the mirrored tap
    wL[r * Mr + m] = m < d.M ? a.wt[(size_t)(tw * Cp + c) * d.Mpad + d.m_off + m] : 0.f;
  }
  __syncthreads();
  constexpr int MBK = 8;                   // outputs per pass
  float ymax = 0.f;                        // max |y| of what this lane stores (published below: GatherArgs::amax_out)
#pragma unroll 1
  for (int m0 = 0; m0 < d.M; m0 += MBK) {
    float acc[MBK];
#pragma unroll
    for (int m = 0; m < MBK; ++m) acc[m] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const float4 w0 = *reinterpret_cast<const float4*>(&wL[(t * CI + c) * Mr + m0]);
        const float4 w1 = *reinterpret_cast<const float4*>(&wL[(t * CI + c) * Mr + m0 + 4]);
        const float x = xv[c][t];
        acc[0] = fmaf(w0.x, x, acc[0]);
        acc[1] = fmaf(w0.y, x, acc[1]);
        acc[2] = fmaf(w0.z, x, acc[2]);
        acc[3] = fmaf(w0.w, x, acc[3]);
        acc[4] = fmaf(w1.x, x, acc[4]);
        acc[5] = fmaf(w1.y, x, acc[5]);
        acc[6] = fmaf(w1.z, x, acc[6]);
        acc[7] = fmaf(w1.w, x, acc[7]);
      }
      asm volatile("" ::: "memory");   // keep one tap's weight reads in flight, not all 27 rows (register pressure)
    }
    if (inside) {   // lean epilogue (this kernel is only chosen for: + shift, optional ReLU, no residual)
      float* __restrict__ yp = a.y + (size_t)(n * d.M + m0) * HW + oh * W + ow;
#pragma unroll
      for (int m = 0; m < MBK; ++m)
        if (m0 + m < d.M) {
          float v = acc[m] + (a.shift ? a.shift[m0 + m] : 0.f);
          if (relu) v = fmaxf(v, 0.f);
          yp[(size_t)m * HW] = v;
          ymax = fmaxf(ymax, fabsf(v));
        }
    }
  }
  if (a.amax_out) publish_amax(a, ymax);   // (whole waves reach this point: `inside` only guards the stores)
}

// ---- the 3x3 form of conv_thin_k_kernel with R vertically adjacent output pixels per lane --------------------------------
// conv_thin_k_kernel is bound by the LDS return path, not by its FMAs: a broadcast ds_read_b128 still delivers 1 KiB to the
// wave, and a lane needs 54 of them per 8 output channels for 216 FMAs -- at bs 16, 256^2, 64 output channels that is 92 us
// of LDS time per CU against 46 us of fp32 FMA time (130 us measured, r03).  Here a lane owns a column strip of R pixels:
// one weight read feeds R FMAs per output channel (54 reads : 216 R FMAs), and the (R + 2) x 3 input window per channel is
// loaded once for the R pixels (R = 4: 54 loads instead of 108).  Same arithmetic order per pixel as the one-pixel form
// (taps outer, channels inner, one fmaf chain per output): bit-identical results.
// PL (planes output, csrc/conv_p2.hip): the 8 output channels of a pass are exactly one 16-byte unit of a planes tensor -- the
// result is scaled by the power of two derived from the bound  wk[0] * max|x| + wk[1], split and stored as (hi, lo) units into
// the padded planes at a.y (P2Out below), its maximum published into the planes' meta: the first layer of the VGG19 stack
// hands relu1_1 to conv1_2 without an fp32 round trip through HBM.
struct P2Out {
  const float* amax_x;   // partial |x| maxima of the input (n_amax floats)
  int n_amax;
  const float* wk;       // [0] max row sum of |w|, [1] max |shift|
  int* ymeta;
};
template <int CI, int R, bool PL = false>
__global__ __launch_bounds__(256, 2) void conv_thin_kv_kernel(const GatherArgs a, const P2Out po) {
  constexpr int T = 9;
  const vunet_conv_desc& d = a.d;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;
  const int tiles_w = (W + 31) / 32, tiles_h = (H + 8 * R - 1) / (8 * R);
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_w;
  const int ty = (bid / tiles_w) % tiles_h;
  const int n = bid / (tiles_w * tiles_h);
  const int ow = tx * 32 + (threadIdx.x & 31), oh0 = ty * 8 * R + (threadIdx.x >> 5) * R;
  constexpr int C = CI, Cp = (C + 1) & ~1;
  float xv[CI][R + 2][3];
  const float* __restrict__ xs = a.x1 + (size_t)n * C * HW;
#pragma unroll
  for (int r = 0; r < R + 2; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ih = oh0 - 1 + r, iw = ow - 1 + k;
      const bool v = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      const int off = v ? ih * W + iw : 0;
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const float x = xs[c * HW + off];
        xv[c][r][k] = v ? x : 0.f;
      }
    }
  const bool relu = d.mode == 0 && d.out_act == ACT_RELU;
  __shared__ __attribute__((aligned(16))) float wL[T * CI * 128];
  const int Mr = (d.M + 7) & ~7;
  for (int e = threadIdx.x; e < T * CI * Mr; e += 256) {
    const int m = e % Mr, r = e / Mr;           // r = t*CI + c
    const int t = r / CI, c = r - t * CI;
    const int tw = d.mode == 0 ? t : T - 1 - t;   // data gradient: the weight of tap t is stored under the mirrored tap
    wL[r * Mr + m] = m < d.M ? a.wt[(size_t)(tw * Cp + c) * d.Mpad + d.m_off + m] : 0.f;
  }
  __syncthreads();
  constexpr int MBK = 8;                   // outputs per pass
  float ymax = 0.f;
  float sy = 1.f;
  if constexpr (PL) {
    float m_ = 0.f;
    for (int i = threadIdx.x & 63; i < po.n_amax; i += 64) m_ = fmaxf(m_, po.amax_x[i]);
    m_ = wave_max(m_);
    const int ey = h2_scale_exp(po.wk[0] * m_ + po.wk[1]);
    sy = h2_pow2(ey);
    if (blockIdx.x == 0 && threadIdx.x == 0) po.ymeta[0] = ey;
  }
#pragma unroll 1
  for (int m0 = 0; m0 < d.M; m0 += MBK) {
    float acc[R][MBK];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MBK; ++m) acc[r][m] = 0.f;
    int woff = m0 / 4;   // in 16-byte units
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const float4 w0 = reinterpret_cast<const float4*>(wL)[(t * CI + c) * (Mr / 4) + woff];
        const float4 w1 = reinterpret_cast<const float4*>(wL)[(t * CI + c) * (Mr / 4) + woff + 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float x = xv[c][r + t / 3][t % 3];
          acc[r][0] = fmaf(w0.x, x, acc[r][0]);
          acc[r][1] = fmaf(w0.y, x, acc[r][1]);
          acc[r][2] = fmaf(w0.z, x, acc[r][2]);
          acc[r][3] = fmaf(w0.w, x, acc[r][3]);
          acc[r][4] = fmaf(w1.x, x, acc[r][4]);
          acc[r][5] = fmaf(w1.y, x, acc[r][5]);
          acc[r][6] = fmaf(w1.z, x, acc[r][6]);
          acc[r][7] = fmaf(w1.w, x, acc[r][7]);
        }
      }
      // one tap's weight reads in flight: hipcc otherwise hoists all 54 reads of a pass above the FMAs and spills them.
      // The next tap's LDS offset is made to depend on this tap's last FMA of every accumulator chain.
#pragma unroll
      for (int r = 0; r < R; ++r)
        asm volatile("" : "+v"(woff), "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]), "+v"(acc[r][4]),
                          "+v"(acc[r][5]), "+v"(acc[r][6]), "+v"(acc[r][7]));
    }
    if constexpr (PL) {   // one planes unit per pixel and pass (d.M % 8 == 0, W % 32 == 0, H % (8 R) == 0: checked by the caller)
      float sh[MBK];
#pragma unroll
      for (int m = 0; m < MBK; ++m) sh[m] = a.shift ? a.shift[m0 + m] : 0.f;
      const int Hp = H + 2, Wp = W + 2, MB8 = d.M >> 3;
      uint4* const yu = reinterpret_cast<uint4*>(a.y);
      const size_t plane = (size_t)d.N * MB8 * Hp * Wp;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float v[MBK];
#pragma unroll
        for (int m = 0; m < MBK; ++m) {
          float t = acc[r][m] + sh[m];
          if (relu) t = fmaxf(t, 0.f);
          ymax = fmaxf(ymax, fabsf(t));
          v[m] = t * sy;
        }
        uint4 hu, lu;
        h2_split2(v[0], v[1], hu.x, lu.x);
        h2_split2(v[2], v[3], hu.y, lu.y);
        h2_split2(v[4], v[5], hu.z, lu.z);
        h2_split2(v[6], v[7], hu.w, lu.w);
        const size_t o = (((size_t)n * MB8 + (m0 >> 3)) * Hp + (oh0 + r + 1)) * Wp + (ow + 1);
        yu[o] = hu;
        yu[plane + o] = lu;
      }
    } else if (ow < W) {   // lean epilogue (this kernel is only chosen for: + shift, optional ReLU, no residual)
#pragma unroll
      for (int m = 0; m < MBK; ++m)
        if (m0 + m < d.M) {
          const float sh = a.shift ? a.shift[m0 + m] : 0.f;
          float* __restrict__ yp = a.y + (size_t)(n * d.M + m0 + m) * HW + oh0 * W + ow;
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (oh0 + r < H) {
              float v = acc[r][m] + sh;
              if (relu) v = fmaxf(v, 0.f);
              yp[r * W] = v;
              ymax = fmaxf(ymax, fabsf(v));
            }
        }
    }
  }
  if constexpr (PL) {
    const float m_ = wave_max(ymax);
    if ((threadIdx.x & 63) == 0)
      atomicMax(reinterpret_cast<unsigned*>(po.ymeta) + 16 + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u), __float_as_uint(m_));
  } else {
    if (a.amax_out) publish_amax(a, ymax);   // (whole waves reach this point)
  }
}

// which thin kernel (if any) takes this problem: 1 = thin_m, 2 = thin_k, 0 = none
VUNET_ENV_FLAG(env_no_thin, "VUNET_NO_THIN")
int vunet_conv_thin_kind(const vunet_conv_desc* d, int pro, bool has_aux, bool has_res) {
  if (env_no_thin() || pro != 0 || has_aux || d->C2 != 0 || d->stride != 1 || d->Hs != d->Ho ||
      d->Ws != d->Wo || d->d2s || (long)d->N * d->Hs * d->Ws < 64 * 1024)
    return 0;
  const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
  if (k3 && d->M <= 4 && d->C1 >= 8) return 1;
  // (the 3x3 form runs at two waves per SIMD: hipcc keeps all 27 weight rows of a pass in flight, ~250 VGPRs)
  if ((k1 || k3) && d->C1 == 3 && (d->mode == 0 || k3) && d->M <= 128 && !has_res &&
      (d->mode == 1 || d->out_act == ACT_NONE || d->out_act == ACT_RELU))
    return 2;
  return 0;
}

VUNET_ENV_FLAG(env_no_thin_kv, "VUNET_NO_THIN_KV")   // (A/B timing: the one-pixel-per-lane form everywhere)
static bool thin_kv(const vunet_conv_desc& d) { return d.KH == 3 && d.Hs >= 32 && !env_no_thin_kv(); }   // the column-strip form

int vunet_conv_thin_name(const vunet_conv_desc* d, int kind, char* name, int len) {
  if (kind == 1 && thin_kv(*d)) return snprintf(name, len, "conv_thin_mv_kernel<%d, 4>", d->mode);
  if (kind == 1) return snprintf(name, len, "conv_thin_m_kernel<%d>", d->mode);
  if (thin_kv(*d)) return snprintf(name, len, "conv_thin_kv_kernel<3, 4>");
  return snprintf(name, len, "conv_thin_k_kernel<%d, 3>", d->KH);
}

int vunet_conv_thin_launch(const GatherArgs& ga, int kind, hipStream_t st) {
  const vunet_conv_desc& d = ga.d;
  dim3 grid((unsigned)(d.N * ((d.Hs + 7) / 8) * ((d.Ws + 31) / 32))), block(256);
  if (kind == 1 && thin_kv(d)) {
    dim3 gv((unsigned)(d.N * ((d.Hs + 31) / 32) * ((d.Ws + 31) / 32)));
    if (d.mode == 0) VUNET_LAUNCH((conv_thin_mv_kernel<0, 4>), gv, block, 0, st, ga);
    else VUNET_LAUNCH((conv_thin_mv_kernel<1, 4>), gv, block, 0, st, ga);
  } else if (kind == 1) {
    if (d.mode == 0) VUNET_LAUNCH((conv_thin_m_kernel<0>), grid, block, 0, st, ga);
    else VUNET_LAUNCH((conv_thin_m_kernel<1>), grid, block, 0, st, ga);
  } else {
    if (thin_kv(d)) {
      dim3 gv((unsigned)(d.N * ((d.Hs + 31) / 32) * ((d.Ws + 31) / 32)));
      VUNET_LAUNCH((conv_thin_kv_kernel<3, 4>), gv, block, 0, st, ga, P2Out{});
    } else if (d.KH == 3) VUNET_LAUNCH((conv_thin_k_kernel<3, 3>), grid, block, 0, st, ga);
    else VUNET_LAUNCH((conv_thin_k_kernel<1, 3>), grid, block, 0, st, ga);
  }
  return vunet_check_launch();
}

// The first layer of the p2 VGG19 pass (include/vunet_hip.h: vunet_p2_conv_first): 3 -> M channels, 3x3 / pad 1, + shift, ReLU,
// fp32 NCHW in, planes out.
extern "C" int vunet_p2_conv_first(const float* x, const float* amax_x, int32_t n_amax, const float* wt_f, int32_t Mpad,
                                   const float* shift, const float* wk, void* y, int32_t* ymeta, int32_t N, int32_t H, int32_t W,
                                   int32_t M, void* stream) {
  if (!x || !amax_x || n_amax < 1 || !wt_f || !wk || !y || !ymeta) return VUNET_ERR_ARG;
  if (N < 1 || M < 8 || M % 8 || M > 128 || H % 32 || W % 32 || ((uintptr_t)y & 15)) return VUNET_ERR_UNSUPPORTED;
  GatherArgs ga;
  memset(&ga, 0, sizeof(ga));
  vunet_conv_desc& d = ga.d;
  d.N = N; d.C1 = 3; d.C2 = 0; d.Hs = H; d.Ws = W; d.M = M; d.m_off = 0; d.Mpad = Mpad; d.Ho = H; d.Wo = W; d.KH = 3; d.KW = 3;
  d.stride = 1; d.pad = 1; d.mode = 0; d.out_act = ACT_RELU;
  ga.x1 = x; ga.wt = wt_f; ga.shift = shift; ga.y = reinterpret_cast<float*>(y);
  ga.NP = N * H * W; ga.HoWo = H * W; ga.HsWs = H * W;
  ga.ph = ga.pw = -1;
  P2Out po{amax_x, (int)n_amax, wk, (int*)ymeta};
  dim3 gv((unsigned)(N * (H / 32) * (W / 32)));
  VUNET_LAUNCH((conv_thin_kv_kernel<3, 4, true>), gv, dim3(256), 0, (hipStream_t)stream, ga, po);
  return vunet_check_launch();
}
