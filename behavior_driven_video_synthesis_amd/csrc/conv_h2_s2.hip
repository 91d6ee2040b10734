// Forward of the 3x3 / stride 2 / pad 1 convolution (Downsample, lib/modules.py:148-161) on the fp16 matrix cores, "h2"
// scheme (two scaled fp16 terms, three products, two accumulators: conv_h2_kernel.h) -- the last convolution of the step's
// critical path that still ran on the fp32-input MFMA kernel (conv_tiled_kernel: 67-79 TFLOP/s).
//
// A workgroup (four waves) owns 4 output rows x 32 output columns x 32*MT output channels; wave w owns output row w.
// Tap (kh, kw) of output (q, j) reads input (2q + kh, 2j + kw) of the (9 x 65)-pixel input tile, i.e. pixel
// (q + (kh >> 1), j + (kw >> 1)) of the PARITY PLANE (kh & 1, kw & 1): the tile is staged into four such planes
// ([row parity][column parity][5][33] 16-byte units of 8 channels, per k-half and per fp16 term), so that the B fragment of
// a tap is 32 consecutive units for 32 consecutive output columns -- the same conflict-free ds_read_b128 as the stride-1
// kernel, where reading every other pixel of a row-major tile would put the lanes of a read on half the banks.
// K loop: chunks of 16 input channels; one input buffer (42 KB) and one buffer for the chunk's weight slabs (18 KB per
// m-tile, copied from the image vunet_weightnorm_fwd* writes for conv_h2_kernel: [chunk][kh][m-tile][kw][term][k-half][32]):
// two workgroups per CU cover each other's staging.
#include "conv_common.h"

#include "split_h2.h"

union S2Unit {
  uint4 u;
  h2_f16x8 b;
};

template <int MT>
__global__ __launch_bounds__(256, 2) void conv_h2_s2_kernel(const GatherArgs a, const uint4* __restrict__ wx, int mtiles_pad,
                                                            const float* __restrict__ amax) {
  constexpr int TH = 4, TW = 32, IR = 2 * TH + 1, IC = 2 * TW + 1;   // input tile: 9 rows x 65 columns
  constexpr int PR = TH + 1, PC = TW + 1, PLANE = PR * PC;           // one parity plane: 5 x 33 pixels
  constexpr int HALF = 4 * PLANE;                                    // the four planes of one k-half
  constexpr int XU = 2 * IR * IC;                                    // staging units per chunk (k-half, row, column)
  constexpr int NX = (XU + 255) / 256;
  constexpr int WU = 3 * MT * H2_SLAB;                               // weight units of one chunk: [kh][m-tile][kw][term][k-half][32]
  constexpr int NW = (WU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];      // [term][k-half][plane][PR][PC] | weights of the chunk
  uint4* const xL = smem4;
  uint4* const wL = smem4 + 4 * HALF;

  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = (d.M + 32 * MT - 1) / (32 * MT);
  const int mb = bid % mblocks;
  int t = bid / mblocks;
  const int tiles_w = d.Wo / TW, tiles_h = d.Ho / TH;
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * TH, col0 = tx * TW, m0 = mb * 32 * MT;

  // ---- chunk-invariant staging geometry: unit u = (k-half, tile row r, tile column c); lanes walk columns
  unsigned rel[NX];
  int lds_x[NX];
  uint32_t vbits = 0, ubits = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = tid + 256 * i;
    const bool used = u < XU;
    const int uu = used ? u : 0;
    const int c8 = uu / (IR * IC);
    const int rem = uu - c8 * (IR * IC);
    const int r = rem / IC, c = rem - r * IC;
    const int ih = 2 * row0 - 1 + r, iw = 2 * col0 - 1 + c;
    const bool ok = used && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    rel[i] = ok ? (unsigned)(8 * c8 * HW + ih * W + iw) : 0u;
    lds_x[i] = c8 * HALF + ((r & 1) * 2 + (c & 1)) * PLANE + (r >> 1) * PC + (c >> 1);
    vbits |= (ok ? 1u : 0u) << i;
    ubits |= (used ? 1u : 0u) << i;
  }

  f32x16 acc[MT], acx[MT];   // leading term / the two cross terms (scaled by 2^11)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = acx[mt][r] = 0.f;

  // ---- scales: x by 2^ex from the tensor maximum (1024 partial maxima), w by the exponent in the image header
  float sx, descale, descale2;
  {
    const float4 pm = h2_amax4(amax, a.amax2, tid);
    float m_ = wave_max(fmaxf(fmaxf(pm.x, pm.y), fmaxf(pm.z, pm.w)));
    float* const redm = reinterpret_cast<float*>(smem4);
    if (lane == 0) redm[wave] = m_;
    __syncthreads();
    m_ = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    const int ex = h2_scale_exp(m_);
    const int ew = reinterpret_cast<const int*>(wx)[0];
    sx = h2_pow2(ex);
    h2_pow2_pair(-(ex + ew), descale, descale2);
  }

  const int nch = d.C1 / 16;
  const int mt0 = (d.m_off + m0) >> 5;   // first m-tile of this workgroup in the image
  const uint4* const xB = xL + h * HALF + wave * PC + j;   // + term * 2 * HALF + plane * PLANE + (kh >> 1) * PC + (kw >> 1)
  float xv[NX][8];
  auto issue_x = [&](int ch) {
    const float* __restrict__ xs = a.x1 + (size_t)(n * d.C1 + ch * 16) * HW;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        xv[i][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xs) +
                                                   4u * (rel[i] + (((vbits >> i) & 1u) ? (unsigned)(k * HW) : 0u)));
  };
  issue_x(0);
  for (int ch = 0; ch < nch; ++ch) {
    // ---- stage the chunk's input tile (requested during the previous chunk's MFMAs): 8 channels per unit, scaled and
    //      split into the two fp16 terms
    __syncthreads();   // the previous chunk's fragment reads (and the scale reduction) are done
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if ((ubits >> i) & 1u) {
        const bool ok = (vbits >> i) & 1u;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ok ? xv[i][k] * sx : 0.f;
        uint4 ph, pl;
        h2_split2(v[0], v[1], ph.x, pl.x);
        h2_split2(v[2], v[3], ph.y, pl.y);
        h2_split2(v[4], v[5], ph.z, pl.z);
        h2_split2(v[6], v[7], ph.w, pl.w);
        xL[lds_x[i]] = ph;
        xL[2 * HALF + lds_x[i]] = pl;
      }
    }
    // ---- the chunk's weight slabs (three kernel rows x MT m-tiles), shared by the four waves through LDS: read as A
    //      fragments straight from global memory they were 36 KB per wave and chunk through the L1 -- more than the
    //      54 MFMAs they feed could cover
    {
      uint4 wv[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int w = tid + 256 * i;
        const int kh = w / (MT * H2_SLAB), r = w - kh * (MT * H2_SLAB);
        wv[i] = wx[1 + ((size_t)(ch * 3 + (w < WU ? kh : 0)) * mtiles_pad + mt0) * H2_SLAB + (w < WU ? r : 0)];
      }
#pragma unroll
      for (int i = 0; i < NW; ++i)
        if (tid + 256 * i < WU) wL[tid + 256 * i] = wv[i];
    }
    if (ch + 1 < nch) issue_x(ch + 1);   // in flight while this chunk multiplies
    __syncthreads();
    // ---- nine taps x three products
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const uint4* const wp = wL + kh * (MT * H2_SLAB) + h * 32 + j;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        S2Unit av[2][MT], bv[2];
        const int po = ((kh & 1) * 2 + (kw & 1)) * PLANE + (kh >> 1) * PC + (kw >> 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[p][mt].u = wp[((mt * 3 + kw) * 2 + p) * 64];
          bv[p].u = xB[p * 2 * HALF + po];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x16 cx = acx[mt];
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0].b, cx, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1].b, cx, 0, 0, 0);
          acx[mt] = cx;
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0].b, acc[mt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue (conv_common.h): combine the accumulators, de-scale, shift / activation, 16-byte stores, |y| maxima
  PixGeo g;
  g.n = n;
  g.oh = row0 + wave;
  g.ow = col0 + j;
  g.valid = true;
  float ymax = 0.f;
  const int k4 = lane & 3;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = (acc[mt][r] + acx[mt][r] * (1.f / 2048.f)) * descale * descale2;
    if (a.wide) {
      TileSide4 side;
      load_tile_side4<0>(a, g, m0 + 32 * mt, h, k4, side);
      if (side4_form<0>(a) == 1) ymax = fmaxf(ymax, store_tile_side4<0, 1>(a, g, m0 + 32 * mt, h, k4, c, side));
      else ymax = fmaxf(ymax, store_tile_side4<0>(a, g, m0 + 32 * mt, h, k4, c, side));
    } else {
      TileSide side;
      load_tile_side(a, g, m0 + 32 * mt, h, side);
      ymax = fmaxf(ymax, store_tile_side(a, g, m0 + 32 * mt, h, c, side));
    }
  }
  if (a.amax_out) publish_amax(a, ymax);
}

// ---- host side ------------------------------------------------------------------------------------------------------
bool vunet_conv_h2_s2_ok(const vunet_conv_desc* d, int pro) {
  return d->mode == 0 && d->stride == 2 && d->KH == 3 && d->KW == 3 && d->pad == 1 && pro == 0 && d->C2 == 0 &&
         d->C1 > 0 && d->C1 % 16 == 0 && d->M % 32 == 0 && d->m_off % 32 == 0 && !d->d2s && d->Hs == 2 * d->Ho &&
         d->Ws == 2 * d->Wo && d->Wo % 32 == 0 && d->Ho % 4 == 0 &&
         (long)d->N * (d->Ho / 4) * (d->Wo / 32) * (d->M / 32) >= 64;   // (smaller launches: the fp32 / small-map kernels)
}

// two m-tiles per workgroup where that still leaves two workgroups per CU; else one (twice the workgroups, the input tile
// staged by both: the 128 -> 128 layer at 64 -> 32 has 256 workgroups with two m-tiles)
static int s2_mt(const vunet_conv_desc* d) {
  return (d->M % 64 == 0 && (long)d->N * (d->Ho / 4) * (d->Wo / 32) * (d->M / 64) >= 512) ? 2 : 1;
}

int vunet_conv_h2_s2_name(const vunet_conv_desc* d, char* name, int len) {
  return snprintf(name, len, "conv_h2_s2_kernel<%d>", s2_mt(d));
}

int vunet_conv_h2_s2_launch(const GatherArgs& ga_in, const void* wx, int mtiles_pad, const float* amax, hipStream_t st) {
  GatherArgs ga = ga_in;
  const vunet_conv_desc& d = ga.d;
  const uintptr_t al = reinterpret_cast<uintptr_t>(ga.y) | reinterpret_cast<uintptr_t>(ga.res);
  ga.wide = (al & 15) == 0 && d.Wo % 4 == 0;   // (fill_args ties `wide` to stride 1: the OUTPUT map is what the stores walk)
  const int MT = s2_mt(&d);
  const int blocks = d.N * (d.Ho / 4) * (d.Wo / 32) * (d.M / (32 * MT));
  const size_t lds = (size_t)(2 * 2 * 4 * 5 * 33 + 3 * MT * H2_SLAB) * 16;
  if (MT == 2) {
    if (hipFuncSetAttribute((const void*)conv_h2_s2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VUNET_ERR_LAUNCH;
    VUNET_LAUNCH((conv_h2_s2_kernel<2>), dim3((unsigned)blocks), dim3(256), lds, st, ga, (const uint4*)wx, mtiles_pad, amax);
  } else {
    VUNET_LAUNCH((conv_h2_s2_kernel<1>), dim3((unsigned)blocks), dim3(256), lds, st, ga, (const uint4*)wx, mtiles_pad, amax);
  }
  return vunet_check_launch();
}
