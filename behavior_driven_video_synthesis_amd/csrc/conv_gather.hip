// Generic implicit-GEMM convolution on fp32 MFMA, operands gathered straight from L1/L2.
//
//   D[m][px] = sum_k  W[k][m] * f(X)[k][px]          k = (source, tap, channel)
//
// One wave owns an (MT*32 channels) x (NT*32 pixels) output tile and issues
// v_mfma_f32_32x32x2_f32: A = weights (lane: channel l&31, k-half l>>5), B = activations
// (lane: pixel l&31, k-half l>>5).  Pixels are the MFMA column so that every accumulator
// register stores 32 consecutive NCHW pixels of one channel (128-B segments).
//
// This kernel makes no assumption on the geometry (any kernel size / stride / padding / map
// size, single or dual source, forward or transposed gather), so it is the fallback for the
// shapes the LDS-tiled kernel (conv_tiled.hip) does not cover, and the data-gradient kernel.
#include "common.h"

struct GatherArgs {
  vunet_conv_desc d;
  const float* x1;
  const float* x2;
  const float* wt;
  const float* shift;
  const float* res;
  const float* aux;
  float* y;
  int NP, HoWo, HsWs;
  InAct in1, in2, auxa;
};

struct PixGeo {
  int n, oh, ow;
  bool valid;
};

__device__ __forceinline__ PixGeo decompose(int P, int NP, int HoWo, int Wo) {
  PixGeo g;
  g.valid = P < NP;
  const int Pc = g.valid ? P : 0;
  g.n = Pc / HoWo;
  const int rem = Pc - g.n * HoWo;
  g.oh = rem / Wo;
  g.ow = rem - g.oh * Wo;
  return g;
}

// PRO: 0 = no prologue activation, 1 = ELU, 2 = ELU + dropout, 3 = generic (runtime InAct)
template <int PRO>
__device__ __forceinline__ float prologue(const InAct& a, float v, uint32_t idx) {
  if (PRO == 0) return v;
  if (PRO == 1) return elu_f(v);
  if (PRO == 2) {
    v = elu_f(v);
    return (vunet_hash_u32(idx + a.seed) >= a.thresh) ? v * a.keep_scale : 0.f;
  }
  return apply_in_act(a, v, idx);
}

template <int MT, int NT, int PRO>
__global__ __launch_bounds__(256) void conv_gather_kernel(const GatherArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const vunet_conv_desc& d = a.d;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = (d.M + 32 * MT - 1) / (32 * MT);
  const int mb = bid % mblocks, pb = bid / mblocks;
  const int m0 = mb * 32 * MT;
  const int tile0 = (pb * 4 + wave) * NT;
  if (tile0 * 32 >= a.NP) return;  // whole wave out of range (no barriers in this kernel)

  // ---- per-lane pixel geometry for each pixel tile
  int base1[NT], base2[NT];
  uint32_t vmask[NT];
  const int s = d.stride, p = d.pad;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const PixGeo g = decompose((tile0 + t) * 32 + j, a.NP, a.HoWo, d.Wo);
    uint32_t rb = 0, cb = 0;
    int sp;
    if (d.mode == 0) {
      const int ih0 = g.oh * s - p, iw0 = g.ow * s - p;
      for (int k = 0; k < d.KH; ++k) rb |= ((unsigned)(ih0 + k) < (unsigned)d.Hs) << k;
      for (int k = 0; k < d.KW; ++k) cb |= ((unsigned)(iw0 + k) < (unsigned)d.Ws) << k;
      sp = ih0 * d.Ws + iw0;
    } else {
      const int ah = (g.oh + p) / s, rh = (g.oh + p) - ah * s;
      const int aw = (g.ow + p) / s, rw = (g.ow + p) - aw * s;
      for (int k = 0; k < d.KH; ++k) rb |= ((k % s == rh) && (unsigned)(ah - k / s) < (unsigned)d.Hs) << k;
      for (int k = 0; k < d.KW; ++k) cb |= ((k % s == rw) && (unsigned)(aw - k / s) < (unsigned)d.Ws) << k;
      sp = ah * d.Ws + aw;
    }
    base1[t] = g.n * d.C1 * a.HsWs + sp;
    base2[t] = g.n * d.C2 * a.HsWs + sp;
    vmask[t] = g.valid ? (rb | (cb << 8)) : 0u;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;

  bool mok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) mok[mt] = (m0 + mt * 32 + j) < d.M;

  int krow = 0;
  for (int src = 0; src < 2; ++src) {
    const int C = src ? d.C2 : d.C1;
    if (C == 0) continue;
    const int Cp = (C + 1) & ~1;
    const float* __restrict__ xs = src ? a.x2 : a.x1;
    const InAct ia = src ? a.in2 : a.in1;
    for (int kh = 0; kh < d.KH; ++kh) {
      for (int kw = 0; kw < d.KW; ++kw) {
        const int toff = d.mode == 0 ? kh * d.Ws + kw : -((kh / s) * d.Ws + kw / s);
        int off[NT];
        bool tv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          tv[t] = ((vmask[t] >> kh) & (vmask[t] >> (8 + kw)) & 1u) != 0;
          off[t] = (src ? base2[t] : base1[t]) + toff + h * a.HsWs;
        }
        const float* __restrict__ wp = a.wt + (size_t)(krow + h) * d.Mpad + d.m_off + m0 + j;
        const bool odd_tail = (C & 1) && h;  // last k-step: the h=1 half has no channel
        for (int c2 = 0; c2 < Cp; c2 += 2) {
          float av[MT], bv[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[mt] = mok[mt] ? wp[mt * 32] : 0.f;
          const bool cok = !(odd_tail && (c2 + 2 >= Cp));
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float v = 0.f;
            if (tv[t] && cok) v = xs[off[t]];
            bv[t] = prologue<PRO>(ia, v, (uint32_t)off[t]);
            off[t] += 2 * a.HsWs;
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[t], acc[mt][t], 0, 0, 0);
          wp += 2 * (size_t)d.Mpad;
        }
        krow += Cp;
      }
    }
  }

  // ---- epilogue
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const PixGeo g = decompose((tile0 + t) * 32 + j, a.NP, a.HoWo, d.Wo);
    if (!g.valid) continue;
    const int pix = g.oh * d.Wo + g.ow;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= d.M) continue;
        float v = acc[mt][t][r];
        if (d.mode == 0) {
          if (a.shift) v += a.shift[m];
          if (d.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (d.out_act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          else if (d.out_act == ACT_ELU) v = elu_f(v);
          else if (d.out_act == ACT_LRELU) v = v > 0.f ? v : v * d.in_slope;
          size_t o;
          if (d.d2s) {
            const int Cq = d.M >> 2, blk = m / Cq, c = m - blk * Cq;
            o = ((size_t)(g.n * Cq + c) * (2 * d.Ho) + (2 * g.oh + (blk >> 1))) * (2 * d.Wo) + 2 * g.ow + (blk & 1);
          } else {
            o = (size_t)(g.n * d.M + m) * a.HoWo + pix;
          }
          if (a.res) v += a.res[o];
          a.y[o] = v;
        } else {
          const size_t o = (size_t)(g.n * d.M + m) * a.HoWo + pix;
          if (a.aux) v *= in_act_grad(a.auxa, a.aux[o], (uint32_t)o);
          if (a.res) v += a.res[o];
          a.y[o] = v;
        }
      }
    }
  }
}

template <int MT, int NT>
static int launch_gather(const GatherArgs& ga, int pro, hipStream_t st) {
  const int ntiles = (ga.NP + 31) / 32;
  const int pblocks = (ntiles + 4 * NT - 1) / (4 * NT);
  const int mblocks = (ga.d.M + 32 * MT - 1) / (32 * MT);
  dim3 grid((unsigned)(pblocks * mblocks)), block(256);
  switch (pro) {
    case 0: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 0>), grid, block, 0, st, ga); break;
    case 1: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 1>), grid, block, 0, st, ga); break;
    case 2: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 2>), grid, block, 0, st, ga); break;
    default: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 3>), grid, block, 0, st, ga); break;
  }
  return vunet_check_launch();
}

extern "C" int vunet_conv2d_gather(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt,
                                   const float* shift, const float* res, const float* aux, float* y,
                                   void* stream) {
  if (!d || !x1 || !wt || !y) return VUNET_ERR_ARG;
  if (d->N <= 0 || d->C1 <= 0 || d->C2 < 0 || d->M <= 0 || d->KH <= 0 || d->KW <= 0 || d->KH > 8 || d->KW > 8 ||
      d->stride <= 0 || d->Mpad % 32 != 0)
    return VUNET_ERR_ARG;
  if (d->C2 > 0 && !x2) return VUNET_ERR_ARG;
  if (d->d2s && (d->M % 4 != 0 || d->mode != 0)) return VUNET_ERR_ARG;
  const int64_t in_elems = (int64_t)d->N * (d->C1 > d->C2 ? d->C1 : d->C2) * d->Hs * d->Ws;
  const int64_t out_elems = (int64_t)d->N * d->M * d->Ho * d->Wo;
  if (in_elems >= (1ll << 31) || out_elems >= (1ll << 31)) return VUNET_ERR_UNSUPPORTED;

  GatherArgs ga;
  ga.d = *d;
  ga.x1 = x1; ga.x2 = x2; ga.wt = wt; ga.shift = shift; ga.res = res; ga.aux = aux; ga.y = y;
  ga.NP = d->N * d->Ho * d->Wo;
  ga.HoWo = d->Ho * d->Wo;
  ga.HsWs = d->Hs * d->Ws;
  ga.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  ga.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  ga.auxa = make_inact(d->aux_act, d->aux_slope, d->aux_drop_p, d->aux_drop_seed);
  int pro = 3;
  if (d->in_act == ACT_NONE && d->drop_p <= 0.f) pro = 0;
  else if (d->in_act == ACT_ELU && d->drop_p <= 0.f) pro = 1;
  else if (d->in_act == ACT_ELU) pro = 2;
  hipStream_t st = (hipStream_t)stream;
  if (d->M <= 32) return launch_gather<1, 4>(ga, pro, st);
  return launch_gather<2, 2>(ga, pro, st);
}
