// bf16-operand forward 3x3 / stride 1 / pad 1 convolution for the inference (transfer / render) path.
//
// The training path multiplies in fp32 (v_mfma_f32_32x32x2_f32) because its contract is fp32 parity with the
// reference.  The render loop (models/vunets.py:508-515 driven per frame by experiments/behavior_net.py's
// sequence sampling -- BASELINE config 5) is specified in bf16: operands are rounded to
// bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS, products accumulate in fp32
// on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate), activations stay fp32 NCHW in HBM.  At that MFMA
// rate the layers are no longer MFMA-bound (a 64-channel layer moves 512 B per pixel for 73.7 kflop); what
// limits them is how many bytes a CU keeps in flight between the staging barriers, so the SMALLEST tile wins:
// 4 x 32 pixels, ~100-128 VGPRs, 3-4 workgroups per CU (measured on the 50-frame render, MI355X: 2.3 TB/s of
// algorithmic traffic with 4-row tiles vs 2.0 / 1.85 TB/s with 8- / 16-row tiles, tools/bench_render.py).
//
// Workgroup tile: MT*32 output channels x (4*NT rows x 32 columns) of one image, NT = 1; K walks the input
// channels in chunks of 16 = one MFMA K step per tap.  LDS images, per chunk, double buffered:
//   xL[(TH+2) x 34 pixels][16 channels] bf16 -- 32 B per pixel: the B fragment of lane (pixel j, half h) is the
//                                               16-byte unit 2*pixel + h, so 64 lanes read 2 KiB contiguously;
//   wL[9 taps][MT*32 channels][16] bf16       -- same unit structure for the A fragment.
// The MFMA lane maps (cdna guide): A[row = l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][col = l&31]; the
// accumulator layout equals the fp32 kernel's, so the epilogue (store_out: + shift, activation, + residual,
// depth-to-space) is shared with it.
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

union Unit16 {
  uint4 u;
  bf16x8 b;
};

template <int MT, int NT, int PRO>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const GatherArgs a, const uint4* __restrict__ wb) {
  constexpr int TW = 32, TH = 4 * NT, IH = TH + 2, IW = TW + 2, MB = 32 * MT;
  constexpr int XU = IH * IW * 2;   // 16-byte units of the input tile: (pixel, channel half)
  constexpr int WU = 9 * MB * 2;    // 16-byte units of a weight chunk: (tap, channel, k half)
  constexpr int NX = (XU + 255) / 256, NW = (WU + 255) / 256;
  constexpr int BUF = XU + WU;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = (d.M + MB - 1) / MB;
  const int mb = bid % mblocks;
  int t = bid / mblocks;
  const int tiles_w = W / TW, tiles_h = H / TH;
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * TH, col0 = tx * TW, m0 = mb * MB;

  // ---- chunk-invariant staging geometry: unit u = (half c8, halo row r, halo column col), lanes walk columns
  int rel[NX], lds_x[NX];
  uint32_t vbits = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = tid + 256 * i;
    const int c8 = u / (IH * IW);
    const int rem = u - c8 * (IH * IW);
    const int r = rem / IW, col = rem - r * IW;
    const int ih = row0 - 1 + r, iw = col0 - 1 + col;
    const bool ok = u < XU && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    rel[i] = ok ? 8 * c8 * HW + ih * W + iw : 0;
    lds_x[i] = 2 * rem + c8;
    vbits |= (ok ? 1u : 0u) << i;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;

  const int nch1 = d.C1 / 16, nch = nch1 + d.C2 / 16;
  float xv[NX][8];
  uint4 wv[NW];

  auto issue_loads = [&](int ch) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    const float* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C + cs) * HW;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) xv[i][k] = xs[rel[i] + (((vbits >> i) & 1u) ? k * HW : 0)];
    const uint4* __restrict__ wp = wb + ((size_t)ch * 9 * d.Mpad + d.m_off + m0) * 2;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int w = tid + 256 * i;   // (tap, m, half)
      const int tap = w / (2 * MB), rem = w - tap * 2 * MB;
      const bool ok = w < WU && d.m_off + m0 + (rem >> 1) < d.Mpad;
      const uint4 v = wp[ok ? (size_t)tap * d.Mpad * 2 + rem : 0];
      wv[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_lds = [&](int ch, uint4* buf) {
    const bool second = ch >= nch1;
    const InAct& ia = second ? a.in2 : a.in1;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (tid + 256 * i < XU) {
        Unit16 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v = xv[i][k];
          if (PRO != 0) v = prologue<PRO>(ia, v, 0u);
          o.b[k] = (__bf16)(((vbits >> i) & 1u) ? v : 0.f);
        }
        buf[lds_x[i]] = o.u;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (tid + 256 * i < WU) buf[XU + tid + 256 * i] = wv[i];
  };

  issue_loads(0);
  write_lds(0, smem4);
  __syncthreads();

  for (int ch = 0; ch < nch; ++ch) {
    const uint4* buf = smem4 + (ch & 1) * BUF;
    if (ch + 1 < nch) issue_loads(ch + 1);
    const uint4* xL = buf + 2 * ((wave * NT) * IW + j) + h;
    const uint4* wL = buf + XU + 2 * j + h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dr = tap / 3, dc = tap % 3;
      Unit16 av[MT], bv[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt].u = wL[2 * (tap * MB + mt * 32)];
#pragma unroll
      for (int q = 0; q < NT; ++q) bv[q].u = xL[2 * ((q + dr) * IW + dc)];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < NT; ++q)
          acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt].b, bv[q].b, acc[mt][q], 0, 0, 0);
    }
    if (ch + 1 < nch) write_lds(ch + 1, smem4 + ((ch + 1) & 1) * BUF);
    __syncthreads();
  }

  // ---- epilogue (shared with the fp32 kernels): lane j = pixel (row0 + wave*NT + q, col0 + j)
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    PixGeo g;
    g.n = n;
    g.oh = row0 + wave * NT + q;
    g.ow = col0 + j;
    g.valid = true;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < d.M) store_out(a, g, m, acc[mt][q][r]);
      }
  }
}

// wt_f [9*(C1p + C2p)][Mpad] fp32 (rows: source, tap, channel)  ->  wb [chunk][tap][Mpad][16] bf16
__global__ void pack_bf16_kernel(const float* __restrict__ wt, __bf16* __restrict__ wb, int C1, int C2, int Mpad) {
  const long total = (long)(C1 + C2) * 9 * Mpad;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i & 15);
  long q = i >> 4;
  const int m = (int)(q % Mpad);
  q /= Mpad;
  const int tap = (int)(q % 9);
  const int ch = (int)(q / 9);
  const int nch1 = C1 / 16;
  const int C1p = (C1 + 1) & ~1, C2p = (C2 + 1) & ~1;
  const long krow = ch < nch1 ? (long)tap * C1p + ch * 16 + c : 9L * C1p + (long)tap * C2p + (ch - nch1) * 16 + c;
  wb[i] = (__bf16)wt[krow * Mpad + m];
}

template <int MT, int NT>
static int launch_bf16(const GatherArgs& ga, const void* wb, int pro, hipStream_t st) {
  constexpr int BUF = (4 * NT + 2) * 34 * 2 + 9 * 32 * MT * 2;
  const size_t lds = 2 * (size_t)BUF * sizeof(uint4);
  const vunet_conv_desc& d = ga.d;
  const int blocks = d.N * (d.Hs / (4 * NT)) * (d.Ws / 32) * ((d.M + 32 * MT - 1) / (32 * MT));
  dim3 grid((unsigned)blocks), block(256);
  if (lds > 64 * 1024) {
    hipFuncSetAttribute((const void*)conv_bf16_kernel<MT, NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)conv_bf16_kernel<MT, NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (pro == 0) VUNET_LAUNCH((conv_bf16_kernel<MT, NT, 0>), grid, block, lds, st, ga, (const uint4*)wb);
  else VUNET_LAUNCH((conv_bf16_kernel<MT, NT, 1>), grid, block, lds, st, ga, (const uint4*)wb);
  return vunet_check_launch();
}

extern "C" int vunet_conv2d_bf16_supported(const vunet_conv_desc* d) {
  if (!d) return 0;
  const bool act_ok = (d->in_act == ACT_NONE || d->in_act == ACT_ELU) && d->drop_p == 0.f;
  return d->mode == 0 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Hs == d->Ho && d->Ws == d->Wo &&
         d->C1 > 0 && d->C1 % 16 == 0 && d->C2 % 16 == 0 && d->Ws % 32 == 0 && d->Hs % 4 == 0 && d->Mpad % 32 == 0 && act_ok;
}

extern "C" int vunet_pack_bf16(const float* wt_f, void* wb, int C1, int C2, int Mpad, void* stream) {
  if (!wt_f || !wb || C1 <= 0 || C1 % 16 || C2 % 16 || Mpad % 32) return VUNET_ERR_ARG;
  const long total = (long)(C1 + C2) * 9 * Mpad;
  VUNET_LAUNCH(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt_f,
               (__bf16*)wb, C1, C2, Mpad);
  return vunet_check_launch();
}

extern "C" int vunet_conv2d_bf16(const vunet_conv_desc* d, const float* x1, const float* x2, const void* wb,
                                 const float* shift, const float* res, float* y, void* stream) {
  if (!d || !x1 || !wb || !y || (d->C2 > 0 && !x2)) return VUNET_ERR_ARG;
  if (!vunet_conv2d_bf16_supported(d)) return VUNET_ERR_UNSUPPORTED;
  GatherArgs ga;
  ga.wide = 0;
  ga.res2 = nullptr;
  ga.amax2 = nullptr;
  ga.amax_out = nullptr;
  ga.amax_out = nullptr;
  ga.d = *d;
  ga.x1 = x1;
  ga.x2 = x2;
  ga.wt = nullptr;
  ga.shift = shift;
  ga.res = res;
  ga.aux = nullptr;
  ga.y = y;
  ga.NP = d->N * d->Ho * d->Wo;
  ga.HoWo = d->Ho * d->Wo;
  ga.HsWs = d->Hs * d->Ws;
  ga.in1 = make_inact(d->in_act, d->in_slope, 0.f, 0u);
  ga.in2 = ga.in1;
  ga.auxa = make_inact(ACT_NONE, 0.f, 0.f, 0u);
  ga.ph = ga.pw = -1;
  ga.subW = ga.subHW = 0;
  ga.mask = nullptr;
  const int pro = d->in_act == ACT_ELU ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  return d->M <= 32 ? launch_bf16<1, 1>(ga, wb, pro, st) : launch_bf16<2, 1>(ga, wb, pro, st);
}
