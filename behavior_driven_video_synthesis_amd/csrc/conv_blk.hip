// Inference convolutions on CHANNEL-BLOCKED bf16 activations -- the render loop of BASELINE config 5
// (models/vunets.py:508-515 `transfer`, driven frame by frame by data/data_conversions_3d.py:1130-1185; "bf16").
//
// Layout "blk": [N][C/8][H][W][8] bf16 -- 16-byte units of 8 consecutive channels of one pixel.  That unit IS the B
// fragment of v_mfma_f32_32x32x16_bf16 for lane (k-half, pixel), so staging an input tile is a 16-byte copy per unit
// (one load instruction where the fp32 NCHW path of conv_bf16.hip needs eight dword loads and a pack), every layer reads
// and writes HALF the bytes of fp32, and the epilogue writes 8 bytes per lane into 512-byte contiguous runs.  Products
// accumulate in fp32; activations are rounded to bf16 once, when stored (the operands of the next convolution would be
// rounded to bf16 anyway: measured on the CPU oracle, storing the block outputs in bf16 costs 0.7 dB of PSNR on top of
// bf16 operands).  Two kernels cover every layer of VunetAlter.transfer:
//   conv_blk_tiled_kernel   3x3 / stride 1 / pad 1 on maps >= 32 wide: LDS-tiled like conv_bf16_kernel (4-row tiles, the
//                           ELU prologue applied once per staged element), two sources (the skip concat), residual,
//                           depth-to-space store, optional fp32 NCHW store (the 3-channel output layer)
//   conv_blk_direct_kernel  everything else -- 1x1 `nin` layers, the stride-2 Downsample convolutions, 3x3 on the small
//                           maps: no LDS, the B fragment of a tap is ONE 16-byte load per lane straight from the tensor
//                           (lane = pixel of the flattened batch, per-lane tap offsets computed once)
// plus fp32 NCHW <-> blk converters and the 3 -> C first layer, which reads the fp32 stickman planes directly.
// Weights: [chunk of 16 K-channels][k half][tap][Mpad][8] bf16 (vunet_pack_bf16_taps): 16-byte A-fragment units, k-half
// planes -- like the LDS images, so that the 16 lanes a ds_read_b128 serves together read 16 consecutive units
// (interleaving the halves put every lane group on 8 of the 16 slots of the 256-byte bank row: 2-way conflicts).
#include "common.h"

typedef __bf16 bk_bf16x8 __attribute__((ext_vector_type(8)));

union BkUnit {
  uint4 u;
  bk_bf16x8 b;
};

struct BlkArgs {
  vunet_conv_desc d;
  const uint4* x1;
  const uint4* x2;
  const uint4* wb;
  const float* shift;
  const void* res;
  void* y;
  int y_nchw;   // 1: y is fp32 NCHW (the network's output layer)
  int NP;
};

__device__ __forceinline__ float bk_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bk_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t bk_pack(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ uint32_t bk_elu2(uint32_t u) { return bk_pack(elu_f(bk_lo(u)), elu_f(bk_hi(u))); }
__device__ __forceinline__ uint4 bk_elu8(uint4 v) { return make_uint4(bk_elu2(v.x), bk_elu2(v.y), bk_elu2(v.z), bk_elu2(v.w)); }

// Epilogue.  Lane = pixel; accumulator register r of a 32 x 32 tile is channel (r&3) + 8(r>>2) + 4h of the m-tile, so per
// register quad the lane owns 4 consecutive channels of its pixel = 8 bytes of a blk unit.  bk_store_quad finishes one
// quad: + shift, activation, + residual, rounding to bf16, plain / depth-to-space / fp32 NCHW addressing.
__device__ __forceinline__ void bk_store_quad(const BlkArgs& a, int n, int oh, int ow, int cb, const float (&q)[4]) {
  const vunet_conv_desc& d = a.d;
  if (cb >= d.M) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = q[e];
    if (a.shift && cb + e < d.M) t += a.shift[cb + e];
    if (d.out_act == ACT_RELU) t = t > 0.f ? t : 0.f;
    else if (d.out_act == ACT_SIGMOID) t = 1.f / (1.f + __expf(-t));
    else if (d.out_act == ACT_ELU) t = elu_f(t);
    v[e] = t;
  }
  if (a.y_nchw) {
    float* y = reinterpret_cast<float*>(a.y);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (cb + e < d.M) y[((size_t)(n * d.M + cb + e) * d.Ho + oh) * d.Wo + ow] = v[e];
    return;
  }
  size_t unit;
  int sub;   // bf16 index inside the unit: 0 or 4
  if (d.d2s) {
    const int Cq = d.M >> 2, blk = cb / Cq, c = cb - blk * Cq;
    unit = ((size_t)(n * (Cq >> 3) + (c >> 3)) * (2 * d.Ho) + 2 * oh + (blk >> 1)) * (2 * d.Wo) + 2 * ow + (blk & 1);
    sub = c & 7;
  } else {
    unit = ((size_t)(n * (d.M >> 3) + (cb >> 3)) * d.Ho + oh) * d.Wo + ow;
    sub = cb & 7;
  }
  if (a.res) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.res) + unit * 16 + sub * 2);
    v[0] += bk_lo(r.x);
    v[1] += bk_hi(r.x);
    v[2] += bk_lo(r.y);
    v[3] += bk_hi(r.y);
  }
  *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.y) + unit * 16 + sub * 2) = make_uint2(bk_pack(v[0], v[1]), bk_pack(v[2], v[3]));
}

__device__ __forceinline__ void bk_store_tile(const BlkArgs& a, int n, int oh, int ow, bool valid, int m_tile0, int h,
                                              const f32x16& acc) {
  if (!valid) return;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float q[4] = {acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
    bk_store_quad(a, n, oh, ow, m_tile0 + 8 * q4 + 4 * h, q);
  }
}

// ---- 3x3 / stride 1 / pad 1, maps a multiple of 32 wide and of 4 high: LDS-tiled
template <int MT, int NT, int PRO>
__global__ __launch_bounds__(256, 2) void conv_blk_tiled_kernel(const BlkArgs a) {
  constexpr int TW = 32, TH = 4 * NT, IH = TH + 2, IW = TW + 2, MB = 32 * MT;   // wave w owns rows w*NT .. w*NT + NT-1
  constexpr int XU = IH * IW * 2;   // 16-byte units of the input tile: (pixel, channel half)
  constexpr int WU = 9 * MB * 2;    // 16-byte units of a weight chunk: (tap, channel, k half)
  constexpr int NX = (XU + 255) / 256, NW = (WU + 255) / 256;
  constexpr int BUF = XU + WU;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int H = d.Hs, W = d.Ws;

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

  // staging geometry: unit u = (half c8, halo row r, halo column col); consecutive lanes walk columns (16 B apart)
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
    rel[i] = ok ? (c8 * H + ih) * W + iw : 0;   // in units, relative to the chunk's first channel block
    lds_x[i] = c8 * (IH * IW) + rem;             // k-half planes: consecutive lanes -> consecutive 16-byte units
    vbits |= (ok ? 1u : 0u) << i;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nch1 = d.C1 / 16, nch = nch1 + d.C2 / 16;
  auto issue_loads = [&](int ch, uint4 (&xv)[NX], uint4 (&wv)[NW]) {
    const bool second = ch >= nch1;
    const int cb = second ? (ch - nch1) * 2 : ch * 2;   // first channel block of the chunk
    const int C8 = (second ? d.C2 : d.C1) >> 3;
    const uint4* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C8 + cb) * H * W;
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = xs[rel[i]];
    const uint4* __restrict__ wp = a.wb + (size_t)ch * 18 * d.Mpad + d.m_off + m0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int w = tid + 256 * i;   // (half, tap, m): the order of the packed image and of the LDS image
      const int ht = w / MB, m = w - ht * MB;
      const bool ok = w < WU && d.m_off + m0 + m < d.Mpad;
      const uint4 v = wp[ok ? (size_t)ht * d.Mpad + m : 0];
      wv[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_lds = [&](uint4* buf, const uint4 (&xv)[NX], const uint4 (&wv)[NW]) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (tid + 256 * i < XU) {
        uint4 v = xv[i];
        if (PRO != 0) v = bk_elu8(v);
        buf[lds_x[i]] = ((vbits >> i) & 1u) ? v : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (tid + 256 * i < WU) buf[XU + tid + 256 * i] = wv[i];
  };
  auto multiply = [&](const uint4* buf) {
    const uint4* wL = buf + XU + h * (9 * MB) + j;
    const uint4* xL = buf + h * (IH * IW) + wave * NT * IW + j;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dr = tap / 3, dc = tap % 3;
      BkUnit av[MT], bv[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt].u = wL[tap * MB + mt * 32];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt].u = xL[(dr + nt) * IW + dc];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt].b, bv[nt].b, acc[mt][nt], 0, 0, 0);
    }
  };

  // One chunk of lookahead.  Measured and rejected on the render shapes (tools/time_blk.py): two chunks of lookahead
  // through a second register set (no faster, +40-60 VGPRs), and staging the next chunk piecewise between the taps
  // (5-25 % slower); staggering the start of co-resident workgroups (no effect).  Removing any ONE of {matrix
  // instructions + LDS reads, weight staging, input loads, ELU} from the 128-channel 64x64 layer shortens it by
  // 25 / 16 / 11 / 8 %: no single resource bounds the kernel.
  uint4 xv[NX], wv[NW];
  issue_loads(0, xv, wv);
  write_lds(smem4, xv, wv);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) issue_loads(ch + 1, xv, wv);
    multiply(smem4 + (ch & 1) * BUF);
    if (ch + 1 < nch) write_lds(smem4 + ((ch + 1) & 1) * BUF, xv, wv);
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bk_store_tile(a, n, row0 + wave * NT + nt, col0 + j, true, m0 + mt * 32, h, acc[mt][nt]);
}

// ---- a whole residual block with a skip input in ONE launch (lib/modules.py:221-233, eval mode):
//        y = x + conv3x3(elu(cat(x, nin(elu(a)))))
// The 1x1 `nin` of the skip tensor is computed on the tile's halo region in a pre-phase -- ELU(a) staged once for all of its
// channels, one small MFMA product per 32 halo pixels, + shift, rounded to bf16 exactly where the separate launch rounds it
// when it stores, zeroed outside the image (the 3x3 pads ITS input) -- and left in LDS as the second source's image, already
// through the 3x3's ELU.  The main loop is conv_blk_tiled_kernel's; the second source's chunks stage weights only.  Saves the
// store and the re-load of nin(elu(a)): 4 of the block's ~10 bytes per pixel and channel.  C1 = C2 = M in {32, 64}.
struct BlkRnbArgs {
  BlkArgs b;           // the 3x3: x1 = x, x2 = a (the RAW skip tensor), wb / shift of the 3x3, res, y
  const uint4* wb_nin; // [chunk of 16][k half][Mpad_nin][8] bf16 (vunet_pack_bf16_taps, taps = 1)
  const float* shift_nin;
  int Mpad_nin;
};

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_blk_rnb_kernel(const BlkRnbArgs ar) {
  constexpr int TW = 32, TH = 4 * NT, IH = TH + 2, IW = TW + 2, MB = 32 * MT, PIX = IH * IW;
  constexpr int XU = PIX * 2, WU = 9 * MB * 2;
  constexpr int NX = (XU + 255) / 256, NW = (WU + 255) / 256;
  constexpr int BUF = XU + WU;
  constexpr int C2 = 32 * MT, AU = (C2 / 8) * PIX;   // the skip tensor's tile: all channels
  constexpr int NPT = (PIX + 31) / 32;
  static_assert(AU <= 2 * BUF, "the staged skip tile aliases the two chunk buffers");
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* const aT = smem4 + 2 * BUF;   // nin(elu(a)) after the 3x3's ELU, [channel block][halo pixel]

  const BlkArgs& a = ar.b;
  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int H = d.Hs, W = d.Ws;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int t = bid;
  const int tiles_w = W / TW, tiles_h = H / TH;
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * TH, col0 = tx * TW;

  // ---- pre-phase 1: ELU(a) of the halo tile, all channels, into the (not yet used) chunk buffers; every load issued first
  {
    const uint4* __restrict__ as = a.x2 + (size_t)n * (C2 / 8) * H * W;
    constexpr int NA = (AU + 255) / 256;
    uint4 av[NA];
    bool aok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + 256 * i;
      const int c8 = u / PIX, rem = u - c8 * PIX;
      const int r = rem / IW, col = rem - r * IW;
      const int ih = row0 - 1 + r, iw = col0 - 1 + col;
      aok[i] = u < AU && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      av[i] = as[aok[i] ? ((size_t)c8 * H + ih) * W + iw : 0];
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (tid + 256 * i < AU) smem4[tid + 256 * i] = aok[i] ? bk_elu8(av[i]) : make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  // ---- pre-phase 2: the 1x1 product per 32 halo pixels; wave w takes pixel tiles w, w + 4, ...
  for (int pt = wave; pt < NPT; pt += 4) {
    const int px = pt * 32 + j, pxc = px < PIX ? px : PIX - 1;
    const int r = pxc / IW, col = pxc - r * IW;
    const int ih = row0 - 1 + r, iw = col0 - 1 + col;
    const bool inside = px < PIX && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int kc = 0; kc < C2 / 16; ++kc) {
        BkUnit av, bv;
        av.u = ar.wb_nin[(size_t)(kc * 2 + h) * ar.Mpad_nin + mt * 32 + j];
        bv.u = smem4[(kc * 2 + h) * PIX + pxc];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.b, bv.b, acc, 0, 0, 0);
      }
      if (px < PIX) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int cb = mt * 32 + 8 * q4 + 4 * h;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * q4 + e] + (ar.shift_nin ? ar.shift_nin[cb + e] : 0.f);
          // the separate launch stores bf16(v); the 3x3 then stages bf16(elu(that)) -- the same two roundings here
          uint2 o = make_uint2(bk_elu2(bk_pack(v[0], v[1])), bk_elu2(bk_pack(v[2], v[3])));
          if (!inside) o = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(reinterpret_cast<char*>(aT) + ((size_t)(cb >> 3) * PIX + px) * 16 + (cb & 7) * 2) = o;
        }
      }
    }
  }
  __syncthreads();   // the chunk buffers are free again; aT is complete

  // ---- the 3x3 itself: conv_blk_tiled_kernel's loop; chunks of the second source read aT and stage weights only
  const int m0 = 0;
  int rel[NX], lds_x[NX];
  uint32_t vbits = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = tid + 256 * i;
    const int c8 = u / PIX;
    const int rem = u - c8 * PIX;
    const int r = rem / IW, col = rem - r * IW;
    const int ih = row0 - 1 + r, iw = col0 - 1 + col;
    const bool ok = u < XU && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    rel[i] = ok ? (c8 * H + ih) * W + iw : 0;
    lds_x[i] = c8 * PIX + rem;
    vbits |= (ok ? 1u : 0u) << i;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  constexpr int nch1 = C2 / 16, nch = 2 * nch1;
  auto issue_loads = [&](int ch, uint4 (&xv)[NX], uint4 (&wv)[NW]) {
    if (ch < nch1) {
      const uint4* __restrict__ xs = a.x1 + (size_t)(n * (C2 / 8) + ch * 2) * H * W;
#pragma unroll
      for (int i = 0; i < NX; ++i) xv[i] = xs[rel[i]];
    }
    const uint4* __restrict__ wp = a.wb + (size_t)ch * 18 * d.Mpad + m0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int w = tid + 256 * i;
      const int ht = w / MB, m = w - ht * MB;
      const bool ok = w < WU && m0 + m < d.Mpad;
      const uint4 v = wp[ok ? (size_t)ht * d.Mpad + m : 0];
      wv[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_lds = [&](int ch, uint4* buf, const uint4 (&xv)[NX], const uint4 (&wv)[NW]) {
    if (ch < nch1) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (tid + 256 * i < XU) buf[lds_x[i]] = ((vbits >> i) & 1u) ? bk_elu8(xv[i]) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (tid + 256 * i < WU) buf[XU + tid + 256 * i] = wv[i];
  };
  auto multiply = [&](int ch, const uint4* buf) {
    const uint4* wL = buf + XU + h * (9 * MB) + j;
    const uint4* xL = (ch < nch1 ? buf + h * PIX : aT + ((ch - nch1) * 2 + h) * PIX) + wave * NT * IW + j;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dr = tap / 3, dc = tap % 3;
      BkUnit av[MT], bv[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt].u = wL[tap * MB + mt * 32];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt].u = xL[(dr + nt) * IW + dc];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt].b, bv[nt].b, acc[mt][nt], 0, 0, 0);
    }
  };
  uint4 xv[NX], wv[NW];
  issue_loads(0, xv, wv);
  write_lds(0, smem4, xv, wv);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) issue_loads(ch + 1, xv, wv);
    multiply(ch, smem4 + (ch & 1) * BUF);
    if (ch + 1 < nch) write_lds(ch + 1, smem4 + ((ch + 1) & 1) * BUF, xv, wv);
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bk_store_tile(a, n, row0 + wave * NT + nt, col0 + j, true, m0 + mt * 32, h, acc[mt][nt]);
}

// ---- the same tile, WAVE-SPECIALISED: 512 threads = four matrix waves (w = 0..3: the taps of chunk c from LDS buffer c & 1,
// rows w*NT .. as above) beside four staging waves (loads two chunks ahead through two register sets, ELU, the LDS writes
// of chunk c + 1 into the other buffer), one barrier per chunk for all eight.  In the uniform kernel above every wave does
// load -> multiply -> convert -> write -> barrier in sequence and the phases of co-resident workgroups do not interleave
// by themselves (ablations: each of {MFMA + LDS reads, weight staging, input loads, ELU} costs 8-25 %, their sum is the
// kernel); here a SIMD always holds a wave of each kind, so its matrix pipe and its VALU / LDS-write path run side by side.
template <int MT, int NT, int PRO>
__global__ __launch_bounds__(512, 4) void conv_blk_ws_kernel(const BlkArgs a) {
  constexpr int TW = 32, TH = 4 * NT, IH = TH + 2, IW = TW + 2, MB = 32 * MT;
  constexpr int XU = IH * IW * 2, WU = 9 * MB * 2;
  constexpr int NX = (XU + 255) / 256, NW = (WU + 255) / 256;
  constexpr int BUF = XU + WU;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

  const vunet_conv_desc& d = a.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = d.Hs, W = d.Ws;
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
  const int nch1 = d.C1 / 16, nch = nch1 + d.C2 / 16;

  if (wave >= 4) {
    // ------------------------------------------------------------------ staging waves
    const int tid = threadIdx.x - 256;
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
      rel[i] = ok ? (c8 * H + ih) * W + iw : 0;
      lds_x[i] = c8 * (IH * IW) + rem;
      vbits |= (ok ? 1u : 0u) << i;
    }
    auto issue_loads = [&](int ch, uint4 (&xv)[NX], uint4 (&wv)[NW]) {
      const bool second = ch >= nch1;
      const int cb = second ? (ch - nch1) * 2 : ch * 2;
      const int C8 = (second ? d.C2 : d.C1) >> 3;
      const uint4* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C8 + cb) * H * W;
#pragma unroll
      for (int i = 0; i < NX; ++i) xv[i] = xs[rel[i]];
      const uint4* __restrict__ wp = a.wb + (size_t)ch * 18 * d.Mpad + d.m_off + m0;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int w = tid + 256 * i;   // (half, tap, m)
        const int ht = w / MB, m = w - ht * MB;
        const bool ok = w < WU && d.m_off + m0 + m < d.Mpad;
        const uint4 v = wp[ok ? (size_t)ht * d.Mpad + m : 0];
        wv[i] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    };
    auto write_lds = [&](uint4* buf, const uint4 (&xv)[NX], const uint4 (&wv)[NW]) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        if (tid + 256 * i < XU) {
          uint4 v = xv[i];
          if (PRO != 0) v = bk_elu8(v);
          buf[lds_x[i]] = ((vbits >> i) & 1u) ? v : make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NW; ++i)
        if (tid + 256 * i < WU) buf[XU + tid + 256 * i] = wv[i];
    };
    // chunk c travels in register set c & 1 (A: even, B: odd); iteration c writes chunk c + 1 and requests chunk c + 3
    uint4 xa[NX], wa[NW], xb[NX], wb2[NW];
    issue_loads(0, xa, wa);
    if (nch > 1) issue_loads(1, xb, wb2);
    write_lds(smem4, xa, wa);
    if (nch > 2) issue_loads(2, xa, wa);
    __syncthreads();
    for (int ch = 0; ch < nch; ch += 2) {
      if (ch + 1 < nch) {
        write_lds(smem4 + BUF, xb, wb2);
        if (ch + 3 < nch) issue_loads(ch + 3, xb, wb2);
      }
      __syncthreads();
      if (ch + 1 >= nch) break;
      if (ch + 2 < nch) {
        write_lds(smem4, xa, wa);
        if (ch + 4 < nch) issue_loads(ch + 4, xa, wa);
      }
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------- matrix waves
  const int j = lane & 31, h = lane >> 5;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const uint4* buf = smem4 + (ch & 1) * BUF;
    const uint4* wL = buf + XU + h * (9 * MB) + j;
    const uint4* xL = buf + h * (IH * IW) + wave * NT * IW + j;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dr = tap / 3, dc = tap % 3;
      BkUnit av[MT], bv[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt].u = wL[tap * MB + mt * 32];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt].u = xL[(dr + nt) * IW + dc];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt].b, bv[nt].b, acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bk_store_tile(a, n, row0 + wave * NT + nt, col0 + j, true, m0 + mt * 32, h, acc[mt][nt]);
}

// ---- every other geometry: taps 1 or 9, stride 1 or 2, any map size; operands straight from global memory.
// KS = 1: the four waves of a workgroup own four 32-pixel tiles and walk all of K.  KS = 4 (small maps: a 4x4 .. 16x16
// map of 25 frames has 13 .. 200 pixel tiles for 256 CUs): the four waves share ONE pixel tile, wave w walks the K chunks
// w, w+4, ... and the partial sums meet in LDS; each wave then finishes one quarter of the accumulator registers.
// In the K-split and 1x1 forms the loads of the next chunk are in flight while the current one multiplies.
template <int MT, int T, int PRO, int KS>
__global__ __launch_bounds__(256) void conv_blk_direct_kernel(const BlkArgs a) {
  static_assert(KS == 1 || (KS == 4 && MT == 1), "the K-split form reduces one 32 x 32 tile");
  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int mblocks = (d.M + 32 * MT - 1) / (32 * MT);
  const int mb = blockIdx.x % mblocks, pb = blockIdx.x / mblocks;
  const int m0 = mb * 32 * MT;
  const int ptile = KS == 1 ? pb * 4 + wave : pb;
  if (KS == 1 && ptile * 32 >= a.NP) return;   // whole wave out of range (no barriers in the KS == 1 form)
  const int P = ptile * 32 + j;
  const bool valid = P < a.NP;
  const int Pc = valid ? P : 0;
  const int HoWo = d.Ho * d.Wo;
  const int n = Pc / HoWo, rem = Pc - n * HoWo;
  const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
  const int H = d.Hs, W = d.Ws, HW = H * W;

  int toff[T];   // unit offset of the tap inside a channel block, -1: outside the map
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int ih = oh * d.stride - d.pad + (T == 1 ? 0 : t / 3), iw = ow * d.stride - d.pad + (T == 1 ? 0 : t % 3);
    const bool ok = valid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    toff[t] = ok ? ih * W + iw : -1;
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const int nch1 = d.C1 / 16, nch = nch1 + d.C2 / 16;
  bool mok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) mok[mt] = d.m_off + m0 + mt * 32 + j < d.Mpad;

  auto load = [&](int ch, uint4 (&A)[T][MT], uint4 (&B)[T]) {
    const bool second = ch >= nch1;
    const int cb = second ? (ch - nch1) * 2 : ch * 2;
    const int C8 = (second ? d.C2 : d.C1) >> 3;
    const uint4* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C8 + cb + h) * HW;   // this lane's k-half
    const uint4* __restrict__ wp = a.wb + ((size_t)(ch * 2 + h) * T) * d.Mpad + d.m_off + m0 + j;
#pragma unroll
    for (int t = 0; t < T; ++t) B[t] = xs[toff[t] >= 0 ? toff[t] : 0];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) A[t][mt] = wp[mok[mt] ? (size_t)t * d.Mpad + mt * 32 : 0];
  };
  auto mma = [&](uint4 (&A)[T][MT], uint4 (&B)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      BkUnit bv;
      bv.u = toff[t] >= 0 ? (PRO != 0 ? bk_elu8(B[t]) : B[t]) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        BkUnit av;
        av.u = mok[mt] ? A[t][mt] : make_uint4(0, 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.b, bv.b, acc[mt], 0, 0, 0);
      }
    }
  };

  if (KS == 1 && T == 9) {
    // large maps: occupancy hides the latency; one chunk in registers keeps 3-4 waves per SIMD resident
    uint4 A0[T][MT], B0[T];
    for (int ch = 0; ch < nch; ++ch) {
      load(ch, A0, B0);
      mma(A0, B0);
    }
  } else {
    uint4 A0[T][MT], B0[T], A1[T][MT], B1[T];
    int ch = KS == 1 ? 0 : wave;
    if (ch < nch) load(ch, A0, B0);
    for (; ch < nch; ch += 2 * KS) {
      const bool more = ch + KS < nch;
      if (more) load(ch + KS, A1, B1);
      mma(A0, B0);
      if (more) {
        if (ch + 2 * KS < nch) load(ch + 2 * KS, A0, B0);
        mma(A1, B1);
      }
    }
  }

  if (KS == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bk_store_tile(a, n, oh, ow, valid, m0 + mt * 32, h, acc[mt]);
  } else {
    __shared__ float red[4][16][64];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[0][r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * wave + e;   // this wave finishes register quad `wave`
      v[e] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    }
    if (valid) bk_store_quad(a, n, oh, ow, m0 + 8 * wave + 4 * h, v);
  }
}

// ---- the KS = 1 form with the WEIGHTS THROUGH LDS.  Read as A fragments straight from global memory they pass through
// the L1 once per wave: 2 KB per wave and chunk for the two MFMAs of a 1x1 layer, 18 KB for the eighteen of a 3x3 one --
// twice what the L1 delivers in the time those MFMAs take (the same finding as in conv_h2_s2.hip).  Here the four waves of
// a workgroup share them: a 1x1 layer copies ALL its chunks once (one barrier), a 3x3 layer one chunk at a time.
template <int MT, int T, int PRO>
__global__ __launch_bounds__(256) void conv_blk_direct_lds_kernel(const BlkArgs a) {
  constexpr int MB = 32 * MT;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* const wL = smem4;   // T == 1: [chunk][k half][MB]   T == 9: [k half][tap][MB] of the current chunk
  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int mblocks = (d.M + MB - 1) / MB;
  const int mb = blockIdx.x % mblocks, pb = blockIdx.x / mblocks;
  const int m0 = mb * MB;
  const int ptile = pb * 4 + wave;
  const bool active = ptile * 32 < a.NP;   // (wave-uniform; idle waves still copy weights and meet the barriers)
  const int P = ptile * 32 + j;
  const bool valid = P < a.NP;
  const int Pc = valid ? P : 0;
  const int HoWo = d.Ho * d.Wo;
  const int n = Pc / HoWo, rem = Pc - n * HoWo;
  const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
  const int H = d.Hs, W = d.Ws, HW = H * W;
  int toff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int ih = oh * d.stride - d.pad + (T == 1 ? 0 : t / 3), iw = ow * d.stride - d.pad + (T == 1 ? 0 : t % 3);
    const bool ok = valid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    toff[t] = ok ? ih * W + iw : -1;
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  const int nch1 = d.C1 / 16, nch = nch1 + d.C2 / 16;

  auto copy_w = [&](int row0, int nrows) {   // rows of the packed image [chunk][k half][tap] -> wL, MB units each
    for (int u = tid; u < nrows * MB; u += 256) {
      const int r = u / MB, m = u - r * MB;
      const bool ok = d.m_off + m0 + m < d.Mpad;
      const uint4 v = a.wb[ok ? (size_t)(row0 + r) * d.Mpad + d.m_off + m0 + m : 0];
      wL[u] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto load_b = [&](int ch, uint4 (&B)[T]) {
    const bool second = ch >= nch1;
    const int cb = second ? (ch - nch1) * 2 : ch * 2;
    const int C8 = (second ? d.C2 : d.C1) >> 3;
    const uint4* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C8 + cb + h) * HW;
#pragma unroll
    for (int t = 0; t < T; ++t) B[t] = xs[toff[t] >= 0 ? toff[t] : 0];
  };
  auto mma = [&](const uint4* wrow, uint4 (&B)[T]) {   // wrow: this lane's k-half of the chunk, + t * MB + mt * 32 + j
#pragma unroll
    for (int t = 0; t < T; ++t) {
      BkUnit bv;
      bv.u = toff[t] >= 0 ? (PRO != 0 ? bk_elu8(B[t]) : B[t]) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        BkUnit av;
        av.u = wrow[t * MB + mt * 32 + j];
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.b, bv.b, acc[mt], 0, 0, 0);
      }
    }
  };

  if (T == 1) {
    uint4 B0[T], B1[T];
    if (active) load_b(0, B0);
    copy_w(0, nch * 2);
    __syncthreads();
    if (active) {
      for (int ch = 0; ch < nch; ch += 2) {
        const bool more = ch + 1 < nch;
        if (more) load_b(ch + 1, B1);
        mma(wL + (ch * 2 + h) * MB, B0);
        if (more) {
          if (ch + 2 < nch) load_b(ch + 2, B0);
          mma(wL + ((ch + 1) * 2 + h) * MB, B1);
        }
      }
    }
  } else {
    uint4 B0[T];
    for (int ch = 0; ch < nch; ++ch) {
      if (active) load_b(ch, B0);
      __syncthreads();   // the previous chunk's fragment reads are done
      copy_w(ch * 2 * T, 2 * T);
      __syncthreads();
      if (active) mma(wL + h * T * MB, B0);
    }
  }
  if (active) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bk_store_tile(a, n, oh, ow, valid, m0 + mt * 32, h, acc[mt]);
  }
}

// ---- converters and the first layer
__global__ void nchw_to_blk_kernel(const float* __restrict__ x, uint4* __restrict__ y, int C8, int HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one unit: (n, c8, pixel)
  if (i >= total) return;
  const int p = (int)(i % HW);
  const long q = i / HW;   // n * C8 + c8
  const float* s = x + (q * 8) * HW + p;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = s[(long)e * HW];
  y[i] = make_uint4(bk_pack(v[0], v[1]), bk_pack(v[2], v[3]), bk_pack(v[4], v[5]), bk_pack(v[6], v[7]));
}

__global__ void blk_to_nchw_kernel(const uint4* __restrict__ x, float* __restrict__ y, int C8, int HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int p = (int)(i % HW);
  const long q = i / HW;
  const uint4 u = x[i];
  float* o = y + (q * 8) * HW + p;
  o[0] = bk_lo(u.x); o[(long)HW] = bk_hi(u.x); o[2L * HW] = bk_lo(u.y); o[3L * HW] = bk_hi(u.y);
  o[4L * HW] = bk_lo(u.z); o[5L * HW] = bk_hi(u.z); o[6L * HW] = bk_lo(u.w); o[7L * HW] = bk_hi(u.w);
}

// 1x1 convolution from a fp32 NCHW tensor with C <= 4 channels (the stickman planes) to a blk tensor:
// wt_f rows = input channel (row pitch Mpad, fp32 effective weights), one thread per (pixel, output octet)
__global__ void conv1x1_few_to_blk_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ shift,
                                          uint4* __restrict__ y, int C, int M8, int Mpad, int HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (n, m8, pixel)
  if (i >= total) return;
  const int p = (int)(i % HW);
  const long q = i / HW;
  const int m8 = (int)(q % M8);
  const long n = q / M8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = shift ? shift[m8 * 8 + e] : 0.f;
  for (int c = 0; c < C; ++c) {
    const float xc = x[(n * C + c) * HW + p];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += xc * wt[(long)c * Mpad + m8 * 8 + e];
  }
  y[i] = make_uint4(bk_pack(v[0], v[1]), bk_pack(v[2], v[3]), bk_pack(v[4], v[5]), bk_pack(v[6], v[7]));
}

// wt_f [T*(C1p + C2p)][Mpad] fp32 (rows: source, tap, channel)  ->  wb [chunk of 16][k half][tap][Mpad][8] bf16
__global__ void pack_bf16_taps_kernel(const float* __restrict__ wt, __bf16* __restrict__ wb, int C1, int C2, int Mpad, int T) {
  const long total = (long)(C1 + C2) * T * Mpad;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7);
  long q = i >> 3;
  const int m = (int)(q % Mpad);
  q /= Mpad;
  const int tap = (int)(q % T);
  q /= T;
  const int half = (int)(q & 1);
  const int ch = (int)(q >> 1);
  const int c = 8 * half + e;
  const int nch1 = C1 / 16;
  const int C1p = (C1 + 1) & ~1, C2p = (C2 + 1) & ~1;
  const long krow = ch < nch1 ? (long)tap * C1p + ch * 16 + c : (long)T * C1p + (long)tap * C2p + (ch - nch1) * 16 + c;
  wb[i] = (__bf16)wt[krow * Mpad + m];
}

// ---- host side ------------------------------------------------------------------------------------------------------
extern "C" int vunet_pack_bf16_taps(const float* wt_f, void* wb, int32_t C1, int32_t C2, int32_t Mpad, int32_t taps,
                                    void* stream) {
  if (!wt_f || !wb || C1 <= 0 || C1 % 16 || C2 % 16 || Mpad % 32 || (taps != 1 && taps != 9)) return VUNET_ERR_ARG;
  const long total = (long)(C1 + C2) * taps * Mpad;
  VUNET_LAUNCH(pack_bf16_taps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt_f,
               (__bf16*)wb, C1, C2, Mpad, taps);
  return vunet_check_launch();
}

extern "C" int vunet_nchw_to_blk(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  if (!x || !y || N < 1 || C < 8 || C % 8 || H < 1 || W < 1) return VUNET_ERR_ARG;
  const long total = (long)N * (C / 8) * H * W;
  VUNET_LAUNCH(nchw_to_blk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint4*)y,
               C / 8, H * W, total);
  return vunet_check_launch();
}

extern "C" int vunet_blk_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  if (!x || !y || N < 1 || C < 8 || C % 8 || H < 1 || W < 1) return VUNET_ERR_ARG;
  const long total = (long)N * (C / 8) * H * W;
  VUNET_LAUNCH(blk_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
               (const uint4*)x, y, C / 8, H * W, total);
  return vunet_check_launch();
}

extern "C" int vunet_conv1x1_few_to_blk(const float* x, const float* wt_f, const float* shift, void* y, int32_t N, int32_t C,
                                        int32_t H, int32_t W, int32_t M, int32_t Mpad, void* stream) {
  if (!x || !wt_f || !y || N < 1 || C < 1 || C > 4 || M < 8 || M % 8 || Mpad < M) return VUNET_ERR_ARG;
  const long total = (long)N * (M / 8) * H * W;
  VUNET_LAUNCH(conv1x1_few_to_blk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, wt_f,
               shift, (uint4*)y, C, M / 8, Mpad, H * W, total);
  return vunet_check_launch();
}

static bool blk_tiled_ok(const vunet_conv_desc* d) {
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Hs == d->Ho && d->Ws == d->Wo &&
         d->Ws % 32 == 0 && d->Hs % 4 == 0;
}

extern "C" int vunet_conv2d_blk_tiled(const vunet_conv_desc* d) { return d && blk_tiled_ok(d) ? 1 : 0; }

static bool blk_rnb_ok(const vunet_conv_desc* d) {
  return blk_tiled_ok(d) && d->mode == 0 && (d->C1 == 32 || d->C1 == 64) && d->C2 == d->C1 && d->M == d->C1 && d->m_off == 0 &&
         d->Mpad % 32 == 0 && d->in_act == ACT_ELU && d->drop_p == 0.f && d->out_act == ACT_NONE && !d->d2s;
}

extern "C" int vunet_conv2d_blk_rnb_supported(const vunet_conv_desc* d) { return d && blk_rnb_ok(d) ? 1 : 0; }

extern "C" int vunet_conv2d_blk_rnb(const vunet_conv_desc* d, const void* x, const void* skip, const void* wb_nin, const float* shift_nin,
                                    int32_t Mpad_nin, const void* wb, const float* shift, const void* res, void* y, void* stream) {
  if (!d || !x || !skip || !wb_nin || !wb || !y || Mpad_nin < d->C2 || Mpad_nin % 32) return VUNET_ERR_ARG;
  if (!blk_rnb_ok(d)) return VUNET_ERR_UNSUPPORTED;
  BlkRnbArgs ar;
  BlkArgs& a = ar.b;
  a.d = *d;
  a.x1 = (const uint4*)x; a.x2 = (const uint4*)skip; a.wb = (const uint4*)wb; a.shift = shift; a.res = res; a.y = y;
  a.y_nchw = 0;
  a.NP = d->N * d->Ho * d->Wo;
  ar.wb_nin = (const uint4*)wb_nin;
  ar.shift_nin = shift_nin;
  ar.Mpad_nin = Mpad_nin;
  hipStream_t st = (hipStream_t)stream;
  const int MT = d->M / 32;
  int NT = (MT == 1 && d->Hs % 8 == 0) ? 2 : 1;   // the tile heights conv_blk_tiled_kernel measured best per channel count
  if (NT == 2 && (size_t)d->N * (d->Hs / 8) * (d->Ws / 32) < 1024) NT = 1;
  if (g_vunet_tune[VUNET_TUNE_BLK_FORCE_NT] == 1 || (g_vunet_tune[VUNET_TUNE_BLK_FORCE_NT] == 2 && MT == 1 && d->Hs % 8 == 0))
    NT = g_vunet_tune[VUNET_TUNE_BLK_FORCE_NT];
  const int blocks = d->N * (d->Hs / (4 * NT)) * (d->Ws / 32);
  const int pix = (4 * NT + 2) * 34;
  const size_t lds = (2 * (size_t)(pix * 2 + 9 * 32 * MT * 2) + (size_t)(4 * MT) * pix) * sizeof(uint4);
  if (MT == 1 && NT == 1) VUNET_LAUNCH((conv_blk_rnb_kernel<1, 1>), dim3(blocks), dim3(256), lds, st, ar);
  else if (MT == 1) VUNET_LAUNCH((conv_blk_rnb_kernel<1, 2>), dim3(blocks), dim3(256), lds, st, ar);
  else VUNET_LAUNCH((conv_blk_rnb_kernel<2, 1>), dim3(blocks), dim3(256), lds, st, ar);
  return vunet_check_launch();
}

extern "C" int vunet_conv2d_blk(const vunet_conv_desc* d, const void* x1, const void* x2, const void* wb, const float* shift,
                                const void* res, void* y, int32_t y_fp32_nchw, void* stream) {
  if (!d || !x1 || !wb || !y || (d->C2 > 0 && !x2)) return VUNET_ERR_ARG;
  if (d->mode != 0 || d->C1 <= 0 || d->C1 % 16 || d->C2 % 16 || d->Mpad % 32 || d->m_off != 0) return VUNET_ERR_UNSUPPORTED;
  if (!((d->KH == 1 && d->KW == 1 && d->pad == 0) || (d->KH == 3 && d->KW == 3 && d->pad == 1))) return VUNET_ERR_UNSUPPORTED;
  if (d->stride != 1 && d->stride != 2) return VUNET_ERR_UNSUPPORTED;
  if ((d->in_act != ACT_NONE && d->in_act != ACT_ELU) || d->drop_p != 0.f) return VUNET_ERR_UNSUPPORTED;
  if (!y_fp32_nchw && (d->M % 8 != 0 || (d->d2s && (d->M % 32 != 0)))) return VUNET_ERR_UNSUPPORTED;
  if (d->d2s && (res || y_fp32_nchw)) return VUNET_ERR_UNSUPPORTED;
  if (d->Ho != (d->Hs + 2 * d->pad - d->KH) / d->stride + 1 || d->Wo != (d->Ws + 2 * d->pad - d->KW) / d->stride + 1)
    return VUNET_ERR_ARG;
  BlkArgs a;
  a.d = *d;
  a.x1 = (const uint4*)x1; a.x2 = (const uint4*)x2; a.wb = (const uint4*)wb; a.shift = shift; a.res = res; a.y = y;
  a.y_nchw = y_fp32_nchw ? 1 : 0;
  a.NP = d->N * d->Ho * d->Wo;
  hipStream_t st = (hipStream_t)stream;
  const int pro = d->in_act == ACT_ELU ? 1 : 0;
  const int MT = d->M <= 32 ? 1 : 2;
  if (blk_tiled_ok(d)) {
    // Tile height, measured per render shape (tools/time_blk.py, profiles/r03_time_blk.txt): 8-row tiles (two rows per
    // wave) win 3-9 % on the 32-wide layers (one m-tile: the weight image is staged half as often per pixel); with two
    // m-tiles the 4-row tile is 0-18 % faster (3 workgroups per CU instead of 2)
    int NT = (MT == 1 && d->Hs % 8 == 0) ? 2 : 1;
    if (NT == 2 && (size_t)d->N * (d->Hs / 8) * (d->Ws / 32) * ((d->M + 32 * MT - 1) / (32 * MT)) < 1024) NT = 1;
    if (g_vunet_tune[VUNET_TUNE_BLK_FORCE_NT] == 1 || (g_vunet_tune[VUNET_TUNE_BLK_FORCE_NT] == 2 && d->Hs % 8 == 0))
      NT = g_vunet_tune[VUNET_TUNE_BLK_FORCE_NT];
    const int blocks = d->N * (d->Hs / (4 * NT)) * (d->Ws / 32) * ((d->M + 32 * MT - 1) / (32 * MT));
    const size_t lds = 2 * (size_t)((4 * NT + 2) * 34 * 2 + 9 * 32 * MT * 2) * sizeof(uint4);
    // wave-specialised form: measured 3-18 % faster on the 128-channel layers (64^2, 32^2 maps), 10-20 % SLOWER on the
    // HBM-bound 32- / 64-channel ones (half the threads of a workgroup issue loads); tools/time_blk.py [--uniform].
    // VUNET_TUNE_BLK_WS: 1 = never, 2 = always (tests, A/B)
    const int ws_knob = g_vunet_tune[VUNET_TUNE_BLK_WS];
    const bool ws = ws_knob == 2 || (ws_knob != 1 && d->C1 >= 128);
#define BLK_TILED(MT_, NT_)                                                                                   \
  do {                                                                                                        \
    if (ws) {                                                                                                 \
      if (pro) VUNET_LAUNCH((conv_blk_ws_kernel<MT_, NT_, 1>), dim3(blocks), dim3(512), lds, st, a);          \
      else VUNET_LAUNCH((conv_blk_ws_kernel<MT_, NT_, 0>), dim3(blocks), dim3(512), lds, st, a);              \
    } else {                                                                                                  \
      if (pro) VUNET_LAUNCH((conv_blk_tiled_kernel<MT_, NT_, 1>), dim3(blocks), dim3(256), lds, st, a);       \
      else VUNET_LAUNCH((conv_blk_tiled_kernel<MT_, NT_, 0>), dim3(blocks), dim3(256), lds, st, a);           \
    }                                                                                                         \
  } while (0)
    if (MT == 1 && NT == 1) BLK_TILED(1, 1);
    else if (MT == 1) BLK_TILED(1, 2);
    else if (NT == 1) BLK_TILED(2, 1);
    else BLK_TILED(2, 2);
#undef BLK_TILED
    return vunet_check_launch();
  }
  const int ptiles = (a.NP + 31) / 32;
  // small launches (the 4x4 .. 16x16 maps): one pixel tile per workgroup, K split over its four waves
  const bool ksplit = (d->C1 + d->C2) >= 64 && (size_t)((ptiles + 3) / 4) * ((d->M + 63) / 64) < 1024;
  const int MTd = ksplit ? 1 : MT;
  const int blocks = (ksplit ? ptiles : (ptiles + 3) / 4) * ((d->M + 32 * MTd - 1) / (32 * MTd));
#define BLK_DIRECT(MT_, T_, PRO_, KS_) \
  VUNET_LAUNCH((conv_blk_direct_kernel<MT_, T_, PRO_, KS_>), dim3(blocks), dim3(256), 0, st, a)
#define BLK_DIRECT_T(T_)                                                      \
  do {                                                                        \
    if (ksplit) { if (pro) BLK_DIRECT(1, T_, 1, 4); else BLK_DIRECT(1, T_, 0, 4); } \
    else if (MT == 1) { if (pro) BLK_DIRECT(1, T_, 1, 1); else BLK_DIRECT(1, T_, 0, 1); } \
    else { if (pro) BLK_DIRECT(2, T_, 1, 1); else BLK_DIRECT(2, T_, 0, 1); }   \
  } while (0)
  // weights through LDS (conv_blk_direct_lds_kernel) for the KS = 1 launches whose image fits; VUNET_TUNE_BLK_WS == 3 keeps
  // the all-global form (A/B, tests)
  const int nch = (d->C1 + d->C2) / 16, MBd = 32 * MTd;
  const size_t wlds = (size_t)(d->KH == 1 ? nch * 2 : 18) * MBd * sizeof(uint4);
  if (!ksplit && wlds <= 64 * 1024 && g_vunet_tune[VUNET_TUNE_BLK_WS] != 3) {
#define BLK_DLDS(MT_, T_, PRO_) VUNET_LAUNCH((conv_blk_direct_lds_kernel<MT_, T_, PRO_>), dim3(blocks), dim3(256), wlds, st, a)
    if (d->KH == 1) {
      if (MT == 1) { if (pro) BLK_DLDS(1, 1, 1); else BLK_DLDS(1, 1, 0); }
      else { if (pro) BLK_DLDS(2, 1, 1); else BLK_DLDS(2, 1, 0); }
    } else {
      if (MT == 1) { if (pro) BLK_DLDS(1, 9, 1); else BLK_DLDS(1, 9, 0); }
      else { if (pro) BLK_DLDS(2, 9, 1); else BLK_DLDS(2, 9, 0); }
    }
#undef BLK_DLDS
    return vunet_check_launch();
  }
  if (d->KH == 1) BLK_DIRECT_T(1);
  else BLK_DIRECT_T(9);
#undef BLK_DIRECT_T
#undef BLK_DIRECT
  return vunet_check_launch();
}
