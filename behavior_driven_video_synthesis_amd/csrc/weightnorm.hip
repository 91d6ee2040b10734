// Weight normalisation folded with NormConv2d's learned affine, and its backward.
//
//   scale[co] = gamma[co] * g[co] / ||v[co]||_2        shift[co] = gamma[co]*bias[co] + beta[co]
//   w_eff[co][ci][tap] = scale[co] * v[co][ci][tap]
//
// vunet_weightnorm_fwd packs w_eff into the two K-major matrices the MFMA conv kernels read
// (forward: rows (source, tap, ci) x cols co ; data-gradient: rows (tap, co) x cols ci).
// vunet_weightnorm_bwd reduces the split-K slabs of vunet_conv2d_wgrad (deterministic order) and
// turns dL/dw_eff, dL/dshift into the gradients of v, g, bias, gamma, beta.
// Reductions are wavefront shuffles (64 lanes) + one LDS hop.
#include "common.h"
#include "split_h2.h"

#define X6_SLAB_UNITS 576   // 16-byte units per (chunk, kh, m-tile): [3 kw][3 planes][2 k-halves][32]

struct WnArgs {
  vunet_wn_desc d;
  const float *v, *g, *bias, *gamma, *beta;
  float *wt_f, *wt_d, *scale, *shift, *invnorm;
  uint4 *wx_f, *wx_d;   // split images for conv_x6_kernel / conv_h2_kernel (NULL: not wanted / geometry not covered)
  float* wmax;          // [Cout] max |w_eff| of every row (d.split == 2: the layer's scale comes from it)
  int T, Ctot, C1p, C2p, Kf, Mpad_f, Coutp2, Kd, Mpad_d;
};

// ---- split-bf16 weight images (conv_x6_kernel.h): units of 8 bf16 laid out
//        [chunk of 16 K-channels][kh][m-tile of 32][kw][plane h/m/l][k-half][32 channels of the m-tile]
//      forward image : K = input channels (source 1 chunks, then source 2 chunks), M = output channels
//      dgrad image   : K = output channels, M = input channels (both sources, as the columns of wt_d)
//      m-tiles are padded (zeros) to vunet_x6_mtiles(M): every workgroup may read MT = 2 tiles from any tile.
__host__ __device__ __forceinline__ int x6_mtiles(int m) { return (((m + 31) / 32 + 1) + 1) & ~1; }

__device__ __forceinline__ uint32_t wn_pack2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, p);
}

__device__ __forceinline__ void wn_split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = wn_pack2(a, b);
  float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = wn_pack2(ra, rb);
  ra -= __uint_as_float(m << 16);
  rb -= __uint_as_float(m & 0xffff0000u);
  l = wn_pack2(ra, rb);
}

// one thread = one (chunk, kh, m-tile, kw, k-half, j) position = three 16-byte units (planes)
__device__ __forceinline__ void wn_pack_x6_body(const WnArgs& a, size_t start, size_t stride) {
  const int T = a.T;
  if (T != 9) return;
  for (int img = 0; img < 2; ++img) {
    uint4* out = img == 0 ? a.wx_f : a.wx_d;
    if (!out) continue;
    const int nch1 = img == 0 ? a.d.C1 / 16 : a.d.Cout / 16;
    const int nch = img == 0 ? nch1 + a.d.C2 / 16 : nch1;
    const int Mdim = img == 0 ? a.d.Cout : a.Ctot;
    const int mtp = x6_mtiles(Mdim);
    // (32-bit index arithmetic: a layer's image has < 2^31 units, and a 64-bit division by a run-time value costs ~100
    // instructions -- five of them per unit were most of this kernel's 300 us per step in round 3)
    const uint32_t total = (uint32_t)nch * 3u * (uint32_t)mtp * 3u * 64u;
    for (uint32_t i = (uint32_t)start; i < total; i += (uint32_t)stride) {
      const int j = (int)(i & 31u), half = (int)((i >> 5) & 1u);
      uint32_t q = i >> 6;
      const int kw = (int)(q % 3u); q /= 3u;
      const int mt = (int)(q % (uint32_t)mtp); q /= (uint32_t)mtp;
      const int kh = (int)(q % 3u);
      const int ch = (int)(q / 3u);
      const int tap = kh * 3 + kw, m = mt * 32 + j;
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float val = 0.f;
        if (m < Mdim) {
          if (img == 0) {
            const int cg = ch < nch1 ? ch * 16 + half * 8 + e : a.d.C1 + (ch - nch1) * 16 + half * 8 + e;
            val = a.scale[m] * a.v[((size_t)m * a.Ctot + cg) * T + tap];
          } else {
            const int co = ch * 16 + half * 8 + e;
            val = a.scale[co] * a.v[((size_t)co * a.Ctot + m) * T + tap];
          }
        }
        w[e] = val;
      }
      uint4 ph, pm, pl;
      wn_split2(w[0], w[1], ph.x, pm.x, pl.x);
      wn_split2(w[2], w[3], ph.y, pm.y, pl.y);
      wn_split2(w[4], w[5], ph.z, pm.z, pl.z);
      wn_split2(w[6], w[7], ph.w, pm.w, pl.w);
      // unit index: (((ch*3 + kh)*mtp + mt)*3 + kw)*3 + plane)*64 + half*32 + j
      const size_t base = ((((size_t)(ch * 3 + kh) * mtp + mt) * 3 + kw) * 3) * 64 + half * 32 + j;
      out[base] = ph;
      out[base + 64] = pm;
      out[base + 128] = pl;
    }
  }
}

// ---- split-fp16 ("h2") weight images (conv_h2_kernel.h): unit 0 = header (int32 exponent ew of the layer's power-of-two
//      scale), then units of 8 fp16 laid out
//        [chunk of 16 K-channels][kh][m-tile of 32][kw][plane h / l][k-half][32 channels of the m-tile]
//      with  2^ew * w = h + l / 2^11.  `ew` comes from the largest |w_eff| of the layer (wmax, written by the scale
//      kernel), reduced here by every workgroup.
__device__ __forceinline__ int wn_layer_exp(const WnArgs& a) {
  __shared__ float s_wmax[4];
  float m = 0.f;
  for (int k = threadIdx.x; k < a.d.Cout; k += blockDim.x) m = fmaxf(m, a.wmax[k]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  m = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, s_wmax[w]);
  return h2_scale_exp(m);
}

__device__ __forceinline__ void wn_pack_h2_body(const WnArgs& a, size_t start, size_t stride) {
  const int T = a.T;
  if (T != 9 || (!a.wx_f && !a.wx_d)) return;
  const int ew = wn_layer_exp(a);
  const float sw = h2_pow2(ew);
  for (int img = 0; img < 2; ++img) {
    uint4* out = img == 0 ? a.wx_f : a.wx_d;
    if (!out) continue;
    if (start == 0) out[0] = make_uint4((uint32_t)ew, 0u, 0u, 0u);
    ++out;
    const int nch1 = img == 0 ? a.d.C1 / 16 : a.d.Cout / 16;
    const int nch = img == 0 ? nch1 + a.d.C2 / 16 : nch1;
    const int Mdim = img == 0 ? a.d.Cout : a.Ctot;
    const int mtp = x6_mtiles(Mdim);
    // (32-bit index arithmetic: a layer's image has < 2^31 units, and a 64-bit division by a run-time value costs ~100
    // instructions -- five of them per unit were most of this kernel's 300 us per step in round 3)
    const uint32_t total = (uint32_t)nch * 3u * (uint32_t)mtp * 3u * 64u;
    for (uint32_t i = (uint32_t)start; i < total; i += (uint32_t)stride) {
      const int j = (int)(i & 31u), half = (int)((i >> 5) & 1u);
      uint32_t q = i >> 6;
      const int kw = (int)(q % 3u); q /= 3u;
      const int mt = (int)(q % (uint32_t)mtp); q /= (uint32_t)mtp;
      const int kh = (int)(q % 3u);
      const int ch = (int)(q / 3u);
      const int tap = kh * 3 + kw, m = mt * 32 + j;
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float val = 0.f;
        if (m < Mdim) {
          if (img == 0) {
            const int cg = ch < nch1 ? ch * 16 + half * 8 + e : a.d.C1 + (ch - nch1) * 16 + half * 8 + e;
            val = a.scale[m] * a.v[((size_t)m * a.Ctot + cg) * T + tap];
          } else {
            const int co = ch * 16 + half * 8 + e;
            val = a.scale[co] * a.v[((size_t)co * a.Ctot + m) * T + tap];
          }
        }
        w[e] = val * sw;
      }
      uint4 ph, pl;
      h2_split2(w[0], w[1], ph.x, pl.x);
      h2_split2(w[2], w[3], ph.y, pl.y);
      h2_split2(w[4], w[5], ph.z, pl.z);
      h2_split2(w[6], w[7], ph.w, pl.w);
      const size_t base = ((((size_t)(ch * 3 + kh) * mtp + mt) * 3 + kw) * 2) * 64 + half * 32 + j;
      out[base] = ph;
      out[base + 64] = pl;
    }
  }
}

__device__ __forceinline__ void wn_pack_split_body(const WnArgs& a, size_t start, size_t stride) {
  if (a.d.split == 2) wn_pack_h2_body(a, start, stride);
  else wn_pack_x6_body(a, start, stride);
}

extern "C" int vunet_x6_mtiles(int32_t M) { return x6_mtiles(M); }

// does the split-bf16 kernel family cover this layer's forward / data-gradient weights?  (3x3, channel counts in 16s)
static bool x6_fwd_ok(const vunet_wn_desc* d) { return d->KH == 3 && d->KW == 3 && d->C1 % 16 == 0 && d->C2 % 16 == 0; }
static bool x6_dgrad_ok(const vunet_wn_desc* d) { return d->KH == 3 && d->KW == 3 && d->Cout % 16 == 0; }

extern "C" int vunet_x6_image_bytes(const vunet_wn_desc* d, int32_t dgrad) {
  if (!d) return 0;
  const int slab = d->split == 2 ? H2_SLAB : X6_SLAB_UNITS, hdr = d->split == 2 ? 16 : 0;
  if (dgrad) return x6_dgrad_ok(d) ? (d->Cout / 16) * 3 * x6_mtiles(d->C1 + d->C2) * slab * 16 + hdr : 0;
  return x6_fwd_ok(d) ? ((d->C1 + d->C2) / 16) * 3 * x6_mtiles(d->Cout) * slab * 16 + hdr : 0;
}

__global__ __launch_bounds__(64) void wn_scale_kernel(const WnArgs a) {
  const int co = blockIdx.x, lane = threadIdx.x;
  const int K = a.Ctot * a.T;
  const float* vr = a.v + (size_t)co * K;
  float invn = 1.f, scale = 1.f, mx = 0.f;
  if (a.d.kind != 1 || a.wmax) {
    float ss = 0.f;
    for (int k = lane; k < K; k += 64) { const float t = vr[k]; ss += t * t; mx = fmaxf(mx, fabsf(t)); }
    if (a.wmax) mx = wave_max(mx);
  if (a.d.kind != 1) {
    ss = wave_sum(ss);
    float nrm = sqrtf(ss);
    if (a.d.kind == 2) nrm = fmaxf(nrm, 1e-12f);
    invn = 1.f / nrm;
    const float gm = a.gamma ? a.gamma[co] : 1.f;
    scale = a.d.kind == 0 ? gm * a.g[co] * invn : gm * invn;
  }
  }
  if (lane == 0) {
    if (a.wmax) a.wmax[co] = fabsf(scale) * mx;
    const float gm = (a.d.kind != 1 && a.gamma) ? a.gamma[co] : 1.f;
    const float b = a.bias ? a.bias[co] : 0.f;
    const float be = (a.d.kind != 1 && a.beta) ? a.beta[co] : 0.f;
    a.scale[co] = scale;
    a.invnorm[co] = invn;
    a.shift[co] = gm * b + be;
  }
}

__global__ __launch_bounds__(256) void wn_pack_kernel(const WnArgs a) {
  const size_t nf = (size_t)a.Kf * a.Mpad_f;
  const size_t nd = a.wt_d ? (size_t)a.Kd * a.Mpad_d : 0;
  const int T = a.T;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (size_t)gridDim.x * 256) {
    if (i < nf) {
      const int krow = (int)(i / a.Mpad_f), m = (int)(i - (size_t)krow * a.Mpad_f);
      int tap, c, cg;
      bool ok;
      const int k1 = T * a.C1p;
      if (krow < k1) { tap = krow / a.C1p; c = krow - tap * a.C1p; ok = c < a.d.C1; cg = c; }
      else { const int r = krow - k1; tap = r / a.C2p; c = r - tap * a.C2p; ok = c < a.d.C2; cg = a.d.C1 + c; }
      float w = 0.f;
      if (ok && m < a.d.Cout) w = a.scale[m] * a.v[((size_t)m * a.Ctot + cg) * T + tap];
      a.wt_f[i] = w;
    } else {
      const size_t e = i - nf;
      const int row = (int)(e / a.Mpad_d), ci = (int)(e - (size_t)row * a.Mpad_d);
      const int tap = row / a.Coutp2, co = row - tap * a.Coutp2;
      float w = 0.f;
      if (co < a.d.Cout && ci < a.Ctot) w = a.scale[co] * a.v[((size_t)co * a.Ctot + ci) * T + tap];
      a.wt_d[e] = w;
    }
  }
  wn_pack_split_body(a, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

static void wn_geometry(const vunet_wn_desc* d, WnArgs& a) {
  a.T = d->KH * d->KW;
  a.Ctot = d->C1 + d->C2;
  a.C1p = (d->C1 + 1) & ~1;
  a.C2p = (d->C2 + 1) & ~1;
  a.Kf = a.T * (a.C1p + a.C2p);
  a.Mpad_f = (d->Cout + 31) / 32 * 32;
  a.Coutp2 = (d->Cout + 1) & ~1;
  a.Kd = a.T * a.Coutp2;
  a.Mpad_d = (a.Ctot + 31) / 32 * 32;
}

extern "C" int vunet_weightnorm_fwd(const vunet_wn_desc* d, const float* v, const float* g, const float* bias,
                                    const float* gamma, const float* beta, float* wt_f, float* wt_d, void* wx_f,
                                    void* wx_d, float* scale, float* shift, float* invnorm, float* wmax, void* stream) {
  if (!d || !v || !wt_f || !scale || !shift || !invnorm) return VUNET_ERR_ARG;
  if (d->kind == 0 && !g) return VUNET_ERR_ARG;
  if (d->Cout < 1 || d->C1 < 1 || d->C2 < 0) return VUNET_ERR_ARG;
  WnArgs a;
  a.d = *d;
  a.v = v; a.g = g; a.bias = bias; a.gamma = gamma; a.beta = beta;
  a.wt_f = wt_f; a.wt_d = wt_d; a.scale = scale; a.shift = shift; a.invnorm = invnorm;
  if ((wx_f && !x6_fwd_ok(d)) || (wx_d && !x6_dgrad_ok(d))) return VUNET_ERR_UNSUPPORTED;
  if (d->split == 2 && (wx_f || wx_d) && !wmax) return VUNET_ERR_ARG;
  a.wx_f = (uint4*)wx_f; a.wx_d = (uint4*)wx_d; a.wmax = wmax;
  wn_geometry(d, a);
  hipStream_t st = (hipStream_t)stream;
  VUNET_LAUNCH(wn_scale_kernel, dim3(d->Cout), dim3(64), 0, st, a);
  const size_t n = (size_t)a.Kf * a.Mpad_f + (wt_d ? (size_t)a.Kd * a.Mpad_d : 0);
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  VUNET_LAUNCH(wn_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  return vunet_check_launch();
}

// ---- batched form: every layer of a model in two launches (one workgroup row per layer) ----------
struct WnItemDev {  // mirrors vunet_wn_item (include/vunet_hip.h)
  const float *v, *g, *bias, *gamma, *beta;
  float *wt_f, *wt_d, *scale, *shift, *invnorm;
  uint4 *wx_f, *wx_d;
  float* wmax;
  vunet_wn_desc d;
};

__device__ __forceinline__ WnArgs item_args(const WnItemDev& it) {
  WnArgs a;
  a.d = it.d;
  a.v = it.v; a.g = it.g; a.bias = it.bias; a.gamma = it.gamma; a.beta = it.beta;
  a.wt_f = it.wt_f; a.wt_d = it.wt_d; a.scale = it.scale; a.shift = it.shift; a.invnorm = it.invnorm;
  a.wx_f = it.wx_f; a.wx_d = it.wx_d; a.wmax = it.wmax;
  a.T = it.d.KH * it.d.KW;
  a.Ctot = it.d.C1 + it.d.C2;
  a.C1p = (it.d.C1 + 1) & ~1;
  a.C2p = (it.d.C2 + 1) & ~1;
  a.Kf = a.T * (a.C1p + a.C2p);
  a.Mpad_f = (it.d.Cout + 31) / 32 * 32;
  a.Coutp2 = (it.d.Cout + 1) & ~1;
  a.Kd = a.T * a.Coutp2;
  a.Mpad_d = (a.Ctot + 31) / 32 * 32;
  return a;
}

__device__ __forceinline__ void wn_scale_body(const WnArgs& a, int co, int lane) {
  const int K = a.Ctot * a.T;
  const float* vr = a.v + (size_t)co * K;
  float invn = 1.f, scale = 1.f, mx = 0.f;
  if (a.d.kind != 1 || a.wmax) {
    float ss = 0.f;
    for (int k = lane; k < K; k += 64) { const float t = vr[k]; ss += t * t; mx = fmaxf(mx, fabsf(t)); }
    if (a.wmax) mx = wave_max(mx);
  if (a.d.kind != 1) {
    ss = wave_sum(ss);
    float nrm = sqrtf(ss);
    if (a.d.kind == 2) nrm = fmaxf(nrm, 1e-12f);
    invn = 1.f / nrm;
    const float gm = a.gamma ? a.gamma[co] : 1.f;
    scale = a.d.kind == 0 ? gm * a.g[co] * invn : gm * invn;
  }
  }
  if (lane == 0) {
    if (a.wmax) a.wmax[co] = fabsf(scale) * mx;
    const float gm = (a.d.kind != 1 && a.gamma) ? a.gamma[co] : 1.f;
    const float b = a.bias ? a.bias[co] : 0.f;
    const float be = (a.d.kind != 1 && a.beta) ? a.beta[co] : 0.f;
    a.scale[co] = scale;
    a.invnorm[co] = invn;
    a.shift[co] = gm * b + be;
  }
}

__device__ __forceinline__ void wn_pack_body(const WnArgs& a, size_t start, size_t stride) {
  const uint32_t nf = (uint32_t)a.Kf * (uint32_t)a.Mpad_f;
  const uint32_t nd = a.wt_d ? (uint32_t)a.Kd * (uint32_t)a.Mpad_d : 0u;
  const int T = a.T;
  for (uint32_t i = (uint32_t)start; i < nf + nd; i += (uint32_t)stride) {   // (32-bit index arithmetic, as above)
    if (i < nf) {
      const int krow = (int)(i / (uint32_t)a.Mpad_f), m = (int)(i - (uint32_t)krow * (uint32_t)a.Mpad_f);
      int tap, c, cg;
      bool ok;
      const int k1 = T * a.C1p;
      if (krow < k1) { tap = krow / a.C1p; c = krow - tap * a.C1p; ok = c < a.d.C1; cg = c; }
      else { const int r = krow - k1; tap = r / a.C2p; c = r - tap * a.C2p; ok = c < a.d.C2; cg = a.d.C1 + c; }
      float w = 0.f;
      if (ok && m < a.d.Cout) w = a.scale[m] * a.v[((size_t)m * a.Ctot + cg) * T + tap];
      a.wt_f[i] = w;
    } else {
      const uint32_t e = i - nf;
      const int row = (int)(e / (uint32_t)a.Mpad_d), ci = (int)(e - (uint32_t)row * (uint32_t)a.Mpad_d);
      const int tap = row / a.Coutp2, co = row - tap * a.Coutp2;
      float w = 0.f;
      if (co < a.d.Cout && ci < a.Ctot) w = a.scale[co] * a.v[((size_t)co * a.Ctot + ci) * T + tap];
      a.wt_d[e] = w;
    }
  }
}

__global__ __launch_bounds__(64) void wn_scale_multi_kernel(const WnItemDev* __restrict__ items) {
  const WnItemDev& it = items[blockIdx.y];
  if ((int)blockIdx.x >= it.d.Cout) return;
  wn_scale_body(item_args(it), blockIdx.x, threadIdx.x);
}

// ---- batched pack, one grid row per layer.
// Round 3 gave every layer 64 workgroups that each GATHERED their outputs from v (one float per output element, neighbouring
// lanes K floats apart: 64 cache lines per load instruction, every line of v fetched again for each of its nine taps and for each
// of the four images): 300 us at the top of every step with nothing beside it (profiles/r04_timeline_graph.txt).  A 3x3 layer of
// the fp16 scheme is now packed by TILES: one workgroup = 32 output channels x one 16-channel chunk x 9 taps of v, read once,
// coalesced (32 rows of 576 contiguous bytes), scaled, parked in LDS (row stride 145 floats: conflict-free along either axis),
// and written out four times -- the two fp32 K-major matrices and the two split images -- each in its own coalesced order.
// Regions no tile owns (padding m-tiles of the images, columns past Ctot of wt_d) are never touched: the host allocates these
// buffers zero-filled (ops.py).  Other layers (1x1, ragged channel counts, the bf16 scheme) keep the gathering loop, with
// workgroups per layer by its size.
constexpr int WN_TILE_LD = 145;

__device__ __forceinline__ bool wn_tiled_ok(const WnArgs& a) {
  return a.T == 9 && a.d.split == 2 && a.wx_f != nullptr && a.d.C1 % 16 == 0 && a.d.C2 % 16 == 0;
}

__device__ __forceinline__ void wn_pack_tile(const WnArgs& a, int mtile, int ch, float* tile) {
  const int tid = threadIdx.x;
  const int nch1 = a.d.C1 / 16;
  const bool second = ch >= nch1;
  const int cg0 = second ? a.d.C1 + (ch - nch1) * 16 : ch * 16;   // first channel of the chunk among all input channels
  const int m0 = mtile * 32;
  const int ew = wn_layer_exp(a);                                 // (cooperative: every thread of the workgroup)
  const float sw = h2_pow2(ew);
  // ---- v tile -> LDS, scaled: tile[co][ci * 9 + tap]
  for (int e = tid; e < 32 * 144; e += 256) {
    const int r = e / 144, k = e - r * 144;
    const int co = m0 + r;
    tile[r * WN_TILE_LD + k] = co < a.d.Cout ? a.scale[co] * a.v[((size_t)co * a.Ctot + cg0) * 9 + k] : 0.f;
  }
  __syncthreads();
  // ---- wt_f: rows (source, tap, channel), columns m -- 32 consecutive floats per row
  {
    const int cl0 = second ? (ch - nch1) * 16 : ch * 16;          // first channel of the chunk inside its source
    const int krow0 = second ? 9 * a.C1p : 0, Cp = second ? a.C2p : a.C1p;
    for (int e = tid; e < 32 * 144; e += 256) {
      const int m = e & 31, q = e >> 5;
      const int ci = q & 15, tap = q >> 4;
      a.wt_f[(size_t)(krow0 + tap * Cp + cl0 + ci) * a.Mpad_f + m0 + m] = tile[m * WN_TILE_LD + ci * 9 + tap];
    }
  }
  // ---- wt_d: rows (tap, co), columns input channel -- 16 consecutive floats per row
  if (a.wt_d) {
    for (int e = tid; e < 32 * 144; e += 256) {
      const int ci = e & 15, q = e >> 4;
      const int r = q & 31, tap = q >> 5;
      const int co = m0 + r;
      if (co < a.Coutp2) a.wt_d[(size_t)(tap * a.Coutp2 + co) * a.Mpad_d + cg0 + ci] = tile[r * WN_TILE_LD + ci * 9 + tap];
    }
  }
  // ---- forward image: K chunk `ch`, m-tile `mtile`: units (tap, plane, half, j): 32 consecutive units per (tap, plane, half)
  {
    uint4* const out = a.wx_f + 1;
    const int mtp = x6_mtiles(a.d.Cout);
    if (mtile == 0 && ch == 0 && tid == 0) a.wx_f[0] = make_uint4((uint32_t)ew, 0u, 0u, 0u);
    for (int e = tid; e < 9 * 2 * 32; e += 256) {
      const int j = e & 31, half = (e >> 5) & 1, tap = e >> 6;
      const int kh = tap / 3, kw = tap - 3 * kh;
      float w[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = tile[j * WN_TILE_LD + (half * 8 + c) * 9 + tap] * sw;
      uint4 ph, pl;
      h2_split2(w[0], w[1], ph.x, pl.x);
      h2_split2(w[2], w[3], ph.y, pl.y);
      h2_split2(w[4], w[5], ph.z, pl.z);
      h2_split2(w[6], w[7], ph.w, pl.w);
      const size_t base = ((((size_t)(ch * 3 + kh) * mtp + mtile) * 3 + kw) * 2) * 64 + half * 32 + j;
      out[base] = ph;
      out[base + 64] = pl;
    }
  }
  // ---- data-gradient image: K = output channels (this tile: chunks 2 mtile, 2 mtile + 1), M = input channels
  if (a.wx_d) {
    uint4* const out = a.wx_d + 1;
    const int mtp = x6_mtiles(a.Ctot);
    if (mtile == 0 && ch == 0 && tid == 0) a.wx_d[0] = make_uint4((uint32_t)ew, 0u, 0u, 0u);
    for (int e = tid; e < 2 * 9 * 2 * 16; e += 256) {
      const int ci = e & 15, half = (e >> 4) & 1;
      const int q = e >> 5;
      const int tap = q % 9, chl = q / 9;
      const int chd = 2 * mtile + chl;
      if (chd * 16 >= a.d.Cout) continue;
      const int kh = tap / 3, kw = tap - 3 * kh;
      float w[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = tile[(chl * 16 + half * 8 + c) * WN_TILE_LD + ci * 9 + tap] * sw;
      uint4 ph, pl;
      h2_split2(w[0], w[1], ph.x, pl.x);
      h2_split2(w[2], w[3], ph.y, pl.y);
      h2_split2(w[4], w[5], ph.z, pl.z);
      h2_split2(w[6], w[7], ph.w, pl.w);
      const int cgt = cg0 + ci;
      const size_t base = ((((size_t)(chd * 3 + kh) * mtp + (cgt >> 5)) * 3 + kw) * 2) * 64 + half * 32 + (cgt & 31);
      out[base] = ph;
      out[base + 64] = pl;
    }
  }
}

__global__ __launch_bounds__(256) void wn_pack_multi_kernel(const WnItemDev* __restrict__ items) {
  __shared__ float tile[32 * WN_TILE_LD];
  const WnArgs a = item_args(items[blockIdx.y]);
  if (wn_tiled_ok(a)) {
    const int ntm = (a.d.Cout + 31) / 32, nch = a.Ctot / 16;
    for (int t = blockIdx.x; t < ntm * nch; t += gridDim.x) {   // (wave-uniform trip count)
      wn_pack_tile(a, t % ntm, t / ntm, tile);
      __syncthreads();                                          // the tile is rewritten by the next round
    }
    return;
  }
  const size_t n = (size_t)a.Kf * a.Mpad_f + (a.wt_d ? (size_t)a.Kd * a.Mpad_d : 0);
  size_t nb = (n + 1023) / 1024;
  if (nb > gridDim.x) nb = gridDim.x;
  if (nb < 1) nb = 1;
  if (blockIdx.x >= nb) return;
  wn_pack_body(a, (size_t)blockIdx.x * 256 + threadIdx.x, nb * 256);
  wn_pack_split_body(a, (size_t)blockIdx.x * 256 + threadIdx.x, nb * 256);
}

extern "C" int vunet_weightnorm_fwd_multi(const vunet_wn_item* items_dev, int32_t n_items, int32_t max_cout,
                                          void* stream) {
  if (!items_dev || n_items < 1 || max_cout < 1) return VUNET_ERR_ARG;
  static_assert(sizeof(WnItemDev) == sizeof(vunet_wn_item), "vunet_wn_item layout");
  const WnItemDev* items = reinterpret_cast<const WnItemDev*>(items_dev);
  hipStream_t st = (hipStream_t)stream;
  VUNET_LAUNCH(wn_scale_multi_kernel, dim3((unsigned)max_cout, (unsigned)n_items), dim3(64), 0, st, items);
  VUNET_LAUNCH(wn_pack_multi_kernel, dim3(512, (unsigned)n_items), dim3(256), 0, st, items);
  return vunet_check_launch();
}

struct WnBwdArgs {
  vunet_wn_desc d;
  const float *slabs, *dshift, *v, *g, *bias, *gamma, *invnorm;
  float *dv, *dg, *dbias, *dgamma, *dbeta;
  float *dwred, *dsred;  // workspace: reduced dW in v's (ci, tap) order [Cout][K], reduced dshift [Cout]
  int nsplit, T, Ctot, Coutp, accumulate;
};

// Stage 1: sum the split-K slabs.  One workgroup = 64 consecutive K entries of one output channel (coalesced
// 256-B rows of every slab) x 4 waves that each take a quarter of the slabs; fixed summation order.
__device__ __forceinline__ void wn_slab_reduce_body(const WnBwdArgs& a, int block, float (*part)[64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.T * a.Ctot;
  const int kchunks = (K + 63) / 64;
  const int co = block / kchunks, kc = block - co * kchunks;
  const int k = kc * 64 + lane;  // slab order (tap, ci)
  float s = 0.f;
  if (k < K) {
    const float* p = a.slabs + (size_t)co * K + k;
    const size_t stride = (size_t)a.Coutp * K;
#pragma unroll 8
    for (int sl = wave; sl < a.nsplit; sl += 4) s += p[(size_t)sl * stride];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && k < K) {
    const float tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int tap = k / a.Ctot, ci = k - tap * a.Ctot;
    a.dwred[(size_t)co * K + ci * a.T + tap] = tot;
  }
  if (kc == 0 && wave == 1) {  // the per-channel sum of dy
    float t = 0.f;
    for (int sl = lane; sl < a.nsplit; sl += 64) t += a.dshift[(size_t)sl * a.Coutp + co];
    t = wave_sum(t);
    if (lane == 0) a.dsred[co] = t;
  }
}

__global__ __launch_bounds__(256) void wn_slab_reduce_kernel(const WnBwdArgs a) {
  __shared__ float part[4][64];
  wn_slab_reduce_body(a, blockIdx.x, part);
}

// Stage 2: per output channel, D = <dW, v> and the parameter gradients.
__device__ __forceinline__ void wn_bwd_body(const WnBwdArgs& a, int co, float* red) {
  const int tid = threadIdx.x;
  const int K = a.T * a.Ctot;
  const float* vr = a.v + (size_t)co * K;
  const float* dwr = a.dwred + (size_t)co * K;
  float dot = 0.f;
  for (int k = tid; k < K; k += 256) dot += dwr[k] * vr[k];
  dot = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  const float D = red[0] + red[1] + red[2] + red[3];
  const float dsh = a.dsred[co];
  // non-accumulate outputs may be uninitialised memory (torch.empty): select, never `0 * old` (0 * NaN = NaN)
#define WN_OLD(ptr) (a.accumulate ? (ptr)[co] : 0.f)

  const int kind = a.d.kind;
  const float invn = kind == 1 ? 1.f : a.invnorm[co];
  const float gm = (kind != 1 && a.gamma) ? a.gamma[co] : 1.f;
  const float gg = kind == 0 ? a.g[co] : 1.f;
  const float b = a.bias ? a.bias[co] : 0.f;
  if (tid == 0) {
    if (kind == 1) {
      if (a.dbias) a.dbias[co] = WN_OLD(a.dbias) + dsh;
    } else {
      if (a.dgamma) a.dgamma[co] = WN_OLD(a.dgamma) + (gg * invn * D + b * dsh);
      if (a.dbeta) a.dbeta[co] = WN_OLD(a.dbeta) + dsh;
      if (a.dbias) a.dbias[co] = WN_OLD(a.dbias) + gm * dsh;
      if (a.dg && kind == 0) a.dg[co] = WN_OLD(a.dg) + gm * invn * D;
    }
  }
  if (a.dv) {
    const float c0 = gm * gg * invn, c1 = invn * invn * D;
    float* dvr = a.dv + (size_t)co * K;
    for (int k = tid; k < K; k += 256) {
      const float dw = dwr[k];
      const float val = kind == 1 ? dw : c0 * (dw - c1 * vr[k]);
      dvr[k] = a.accumulate ? dvr[k] + val : val;
    }
  }
#undef WN_OLD
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const WnBwdArgs a) {
  __shared__ float red[4];
  wn_bwd_body(a, blockIdx.x, red);
}

// ---- batched form: the slab reduction + parameter gradients of MANY layers in two launches (one grid row per layer).
// The per-layer launch pair was 188 launches and 1.56 ms of a 26 ms step (r02 profile), almost all of it launch latency.
struct WnBwdItemDev {  // mirrors vunet_wn_bwd_item (include/vunet_hip.h)
  const float *slabs, *dshift, *v, *g, *bias, *gamma, *invnorm;
  float *dv, *dg, *dbias, *dgamma, *dbeta;
  float* workspace;
  vunet_wn_desc d;
  int32_t nsplit, accumulate;
};

__device__ __forceinline__ WnBwdArgs bwd_item_args(const WnBwdItemDev& it) {
  WnBwdArgs a;
  a.d = it.d;
  a.slabs = it.slabs; a.dshift = it.dshift; a.v = it.v; a.g = it.g; a.bias = it.bias; a.gamma = it.gamma;
  a.invnorm = it.invnorm;
  a.dv = it.dv; a.dg = it.dg; a.dbias = it.dbias; a.dgamma = it.dgamma; a.dbeta = it.dbeta;
  a.nsplit = it.nsplit;
  a.T = it.d.KH * it.d.KW;
  a.Ctot = it.d.C1 + it.d.C2;
  a.Coutp = (it.d.Cout + 31) / 32 * 32;
  a.accumulate = it.accumulate;
  a.dwred = it.workspace;
  a.dsred = it.workspace + (size_t)it.d.Cout * a.T * a.Ctot;
  return a;
}

__global__ __launch_bounds__(256) void wn_slab_reduce_multi_kernel(const WnBwdItemDev* __restrict__ items) {
  __shared__ float part[4][64];
  const WnBwdArgs a = bwd_item_args(items[blockIdx.y]);
  const int nb = a.d.Cout * ((a.T * a.Ctot + 63) / 64);
  for (int b = blockIdx.x; b < nb; b += gridDim.x) {
    wn_slab_reduce_body(a, b, part);
    __syncthreads();   // `part` is reused by the next block of work
  }
}

__global__ __launch_bounds__(256) void wn_bwd_multi_kernel(const WnBwdItemDev* __restrict__ items) {
  __shared__ float red[4];
  const WnBwdItemDev& it = items[blockIdx.y];
  if ((int)blockIdx.x >= it.d.Cout) return;
  wn_bwd_body(bwd_item_args(it), blockIdx.x, red);
}

extern "C" int vunet_weightnorm_bwd_multi(const vunet_wn_bwd_item* items_dev, int32_t n_items, int32_t max_cout,
                                          int32_t max_reduce_blocks, void* stream) {
  if (!items_dev || n_items < 1 || max_cout < 1 || max_reduce_blocks < 1) return VUNET_ERR_ARG;
  static_assert(sizeof(WnBwdItemDev) == sizeof(vunet_wn_bwd_item), "vunet_wn_bwd_item layout");
  const WnBwdItemDev* items = reinterpret_cast<const WnBwdItemDev*>(items_dev);
  hipStream_t st = (hipStream_t)stream;
  // every layer gets the same number of workgroups, each walking its layer's (channel, K chunk) blocks with a stride:
  // enough of them in total to fill the chip several times over, no empty workgroups for the small layers
  int gx = 4096 / n_items;
  if (gx < 16) gx = 16;
  if (gx > max_reduce_blocks) gx = max_reduce_blocks;
  VUNET_LAUNCH(wn_slab_reduce_multi_kernel, dim3((unsigned)gx, (unsigned)n_items), dim3(256), 0, st, items);
  VUNET_LAUNCH(wn_bwd_multi_kernel, dim3((unsigned)max_cout, (unsigned)n_items), dim3(256), 0, st, items);
  return vunet_check_launch();
}

extern "C" int vunet_weightnorm_bwd(const vunet_wn_desc* d, const float* slabs, const float* dshift, int32_t nsplit,
                                    const float* v, const float* g, const float* bias, const float* gamma,
                                    const float* invnorm, float* dv, float* dg, float* dbias, float* dgamma,
                                    float* dbeta, float* workspace, int32_t accumulate, void* stream) {
  if (!d || !slabs || !dshift || !v || !workspace || nsplit < 1) return VUNET_ERR_ARG;
  if (d->kind != 1 && !invnorm) return VUNET_ERR_ARG;
  if (d->kind == 0 && !g) return VUNET_ERR_ARG;
  WnBwdArgs a;
  a.d = *d;
  a.slabs = slabs; a.dshift = dshift; a.v = v; a.g = g; a.bias = bias; a.gamma = gamma; a.invnorm = invnorm;
  a.dv = dv; a.dg = dg; a.dbias = dbias; a.dgamma = dgamma; a.dbeta = dbeta;
  a.nsplit = nsplit;
  a.T = d->KH * d->KW;
  a.Ctot = d->C1 + d->C2;
  a.Coutp = (d->Cout + 31) / 32 * 32;
  a.accumulate = accumulate;
  const int K = a.T * a.Ctot;
  a.dwred = workspace;                       // [Cout][K]
  a.dsred = workspace + (size_t)d->Cout * K; // [Cout]
  const int kchunks = (K + 63) / 64;
  VUNET_LAUNCH(wn_slab_reduce_kernel, dim3((unsigned)(d->Cout * kchunks)), dim3(256), 0, (hipStream_t)stream, a);
  VUNET_LAUNCH(wn_bwd_kernel, dim3(d->Cout), dim3(256), 0, (hipStream_t)stream, a);
  return vunet_check_launch();
}
