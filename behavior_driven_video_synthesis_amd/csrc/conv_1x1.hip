// Streaming 1x1 / stride-1 convolution on fp32 MFMA (forward and data gradient): the `nin` layers of the
// residual blocks (lib/modules.py:224-226) and the network-in-network heads (models/vunets.py:118,231,273).
//
// A 1x1 convolution is a [M x K] x [K x pixels] product with no spatial reuse: every input element is used
// once per output channel, so with M <= 128 it sits at or below the HBM balance point (128->128: 32 flop/B) and
// the job is to stream x through the MFMA exactly once.  A workgroup (4 waves) owns ALL output channels of its
// m-block (up to 128 = 4 m-tiles) for 4 x 32 consecutive pixels of one image, one 32-pixel tile per wave:
//   * the B operand (activations) goes global -> register -> MFMA with no LDS hop: lane (pixel j, half h) of
//     k-pair p loads x[c = 2p + h][pixel], i.e. each load instruction reads two 128-byte NCHW row segments;
//     the prologue (ELU / dropout hash) runs on the loaded registers;
//   * the A operand (K-major effective weights, [32-channel chunk][MB]) is staged through LDS once per chunk
//     per workgroup, double buffered, and read with conflict-free ds_read_b32;
//   * loads of chunk c+1 (activations and weights) are issued before the MFMAs of chunk c.
// The data gradient of a 1x1 convolution is the same product with the transposed weights (wt_d) and the
// epilogue  y = acc * act'(aux) + res  (store_out, shared with the other kernels).
#include <type_traits>

#include "conv_common.h"

template <int MT, int MODE, int PRO>
__global__ __launch_bounds__(256, 3) void conv_1x1_kernel(const GatherArgs a_in, const int ntiles, const int tiles_per_img) {
  GatherArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  constexpr int CK = 32, MB = 32 * MT;
  constexpr int NW = CK * MB / 256;   // weight floats staged per thread per chunk
  __shared__ float wL[2][CK * MB];

  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int HW = a.HsWs;
  const int mblocks = (d.M + MB - 1) / MB;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mb = bid % mblocks;
  const int tile_raw = (int)(bid / mblocks) * 4 + wave;
  const bool live = tile_raw < ntiles;            // a trailing wave without a tile still takes part in the barriers
  const int tile = live ? tile_raw : ntiles - 1;
  const int n = tile / tiles_per_img;
  const int hw0 = (tile - n * tiles_per_img) * 32;
  const int m0 = mb * MB;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const int nch1 = d.C1 / CK, nch = nch1 + d.C2 / CK;
  float xc[CK / 2], xn[CK / 2], wv[NW];

  auto issue_loads = [&](int ch) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * CK : ch * CK;
    const int C = second ? d.C2 : d.C1;
    const float* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C + cs + h) * HW + hw0 + j;
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) xn[p] = xs[(size_t)(2 * p) * HW];
    const int krow0 = second ? ((d.C1 + 1) & ~1) : 0;
    const float* __restrict__ wp = a.wt + (size_t)(krow0 + cs) * d.Mpad + d.m_off + m0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + 256 * i;   // [c][m]
      const int m = e % MB, c = e / MB;
      const bool mv = m0 + m < d.M;
      const float w = wp[(size_t)c * d.Mpad + (mv ? m : 0)];
      wv[i] = mv ? w : 0.f;
    }
  };
  auto commit = [&](int ch, float* buf) {   // weights -> LDS, activations -> prologue -> current registers
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * CK : ch * CK;
    const int C = second ? d.C2 : d.C1;
    const InAct& ia = second ? a.in2 : a.in1;
#pragma unroll
    for (int i = 0; i < NW; ++i) buf[tid + 256 * i] = wv[i];
    const uint32_t idx0 = (uint32_t)((n * C + cs + h) * HW + hw0 + j);
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) {
      float v = xn[p];
      if (PRO != 0) v = prologue<PRO>(ia, v, idx0 + (uint32_t)(2 * p * HW));
      xc[p] = v;
    }
  };

  issue_loads(0);
  commit(0, wL[0]);
  __syncthreads();

  for (int ch = 0; ch < nch; ++ch) {
    const float* buf = wL[ch & 1] + h * MB + j;   // A: wL[(2p + h)*MB + mt*32 + j]
    if (ch + 1 < nch) issue_loads(ch + 1);
#pragma unroll
    for (int p = 0; p < CK / 2; ++p)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[2 * p * MB + mt * 32], xc[p], acc[mt], 0, 0, 0);
    if (ch + 1 < nch) commit(ch + 1, wL[(ch + 1) & 1]);
    __syncthreads();
  }

  if (!live) return;
  PixGeo g;
  g.n = n;
  const int pix = hw0 + j;
  g.oh = pix / d.Wo;
  g.ow = pix - g.oh * d.Wo;
  g.valid = true;
  if (a.wide) {
    // 16-byte stores after a lane-quad transpose, the common cases as straight-line code (conv_common.h): the general
    // per-element form below cost the split-fp16 kernel 7 - 8 us per tile in scalar branches (tools/h2_timeline.py)
    const int k = lane & 3;
    float ymax = 0.f;
    auto tiles = [&](auto form_c) {
      constexpr int FORM = decltype(form_c)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        TileSide4 side;
        load_tile_side4<MODE, FORM>(a, g, m0 + mt * 32, h, k, side);
        ymax = fmaxf(ymax, store_tile_side4<MODE, FORM>(a, g, m0 + mt * 32, h, k, acc[mt], side));
      }
    };
    const int form = side4_form<MODE>(a);
    if (form == 1) tiles(std::integral_constant<int, 1>{});
    else if (MODE == 1 && form == 2) tiles(std::integral_constant<int, MODE == 1 ? 2 : 0>{});
    else if (MODE == 1 && form == 3) tiles(std::integral_constant<int, MODE == 1 ? 3 : 0>{});
    else tiles(std::integral_constant<int, 0>{});
    if (a.amax_out) publish_amax(a, ymax);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) store_tile16(a, g, m0 + mt * 32, h, acc[mt]);
}

VUNET_ENV_FLAG(env_no_1x1, "VUNET_NO_1X1")
bool vunet_conv_1x1_applicable(const vunet_conv_desc* d, int pro) {
  return !env_no_1x1() && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 &&
         d->Hs == d->Ho && d->Ws == d->Wo && (d->Hs * d->Ws) % 32 == 0 && d->C1 % 32 == 0 && d->C2 % 32 == 0 &&
         pro != 3 && (d->mode == 0 || pro == 0);
}

// m-tiles per workgroup: all of M (x is then read once) unless that leaves the chip short of workgroups -- the 128-channel
// layers of the 32^2 / 64^2 levels are 128 / 512 workgroups of four chunk-serial steps each (56 us for 0.5 GFLOP at 32^2,
// r03); with fewer m-tiles per workgroup x is re-read from L2, which at these sizes costs nothing
static int mt_of(const vunet_conv_desc* d) {
  int MT = d->M <= 32 ? 1 : d->M <= 64 ? 2 : 4;
#ifndef VUNET_AB_1X1_MT_BY_M   // (A/B baseline, tools/ab_build.sh)
  const long groups = ((long)d->N * d->Hs * d->Ws / 32 + 3) / 4;
  while (MT > 1 && groups * ((d->M + 32 * MT - 1) / (32 * MT)) < 1024) MT >>= 1;
#endif
  return MT;
}

int vunet_conv_1x1_name(const vunet_conv_desc* d, int pro, char* name, int len) {
  return snprintf(name, len, "conv_1x1_kernel<%d, %d, %d>", mt_of(d), d->mode, d->mode == 1 ? 0 : pro);
}

template <int MT, int MODE>
static int launch_1x1(const GatherArgs& ga, int pro, hipStream_t st) {
  const vunet_conv_desc& d = ga.d;
  const int tiles_per_img = d.Hs * d.Ws / 32, ntiles = d.N * tiles_per_img;
  const int mblocks = (d.M + 32 * MT - 1) / (32 * MT);
  dim3 grid((unsigned)(((ntiles + 3) / 4) * mblocks)), block(256);
  if constexpr (MODE == 1) {
    VUNET_LAUNCH((conv_1x1_kernel<MT, 1, 0>), grid, block, 0, st, ga, ntiles, tiles_per_img);
  } else {
    switch (pro) {
      case 0: VUNET_LAUNCH((conv_1x1_kernel<MT, 0, 0>), grid, block, 0, st, ga, ntiles, tiles_per_img); break;
      case 1: VUNET_LAUNCH((conv_1x1_kernel<MT, 0, 1>), grid, block, 0, st, ga, ntiles, tiles_per_img); break;
      case 2: VUNET_LAUNCH((conv_1x1_kernel<MT, 0, 2>), grid, block, 0, st, ga, ntiles, tiles_per_img); break;
      default: return VUNET_ERR_UNSUPPORTED;
    }
  }
  return vunet_check_launch();
}

int vunet_conv_1x1_launch(const GatherArgs& ga, int pro, hipStream_t st) {
  const int MT = mt_of(&ga.d);
  if (ga.d.mode == 1) return MT == 1 ? launch_1x1<1, 1>(ga, pro, st) : MT == 2 ? launch_1x1<2, 1>(ga, pro, st) : launch_1x1<4, 1>(ga, pro, st);
  return MT == 1 ? launch_1x1<1, 0>(ga, pro, st) : MT == 2 ? launch_1x1<2, 0>(ga, pro, st) : launch_1x1<4, 0>(ga, pro, st);
}
