// fp32-accurate 3x3 / stride 1 / pad 1 convolution (forward and data gradient) on the bf16 matrix cores.
//
// On gfx950 the fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA
// rate.  A convolution that must match an fp32 reference is therefore fastest when every fp32 operand is split,
// exactly, into three bf16 terms
//       x = xh + xm + xl      (xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm); round-to-nearest-even:
//                              8 + 8 + 8 significand bits cover fp32's 24, the split loses nothing)
// and the product  w * x  is formed from the six partial products whose weight is >= 2^-16 of the leading one,
//       wh xh + (wh xm + wm xh) + (wm xm + wh xl + wl xh),
// each exact in fp32 (8 x 8 significand bits) and accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped
// terms (wm xl, wl xm, wl xl) are below 2^-24 of the product, i.e. below fp32's own rounding: measured against an
// fp64 convolution the result is as close as (K-blocked accumulation: slightly closer than) the fp32 fmaf chain of
// the fp32 MFMA kernel (tests/test_hip_x6.py).  Six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of K = 2:
// 192 instead of 512 matrix-pipe cycles per 32 x 32 x 16 block -- a 2.67x higher roof (2.5 PFLOP/s / 6 = 417 TFLOP/s
// fp32-equivalent against 157 TFLOP/s).
//
// Work decomposition (256 threads = 4 waves, 2 workgroups per CU):
//   tile      MT*32 output channels x (4*NT rows x 32 columns) of one image; wave w owns rows w*NT .. w*NT+NT-1
//   K loop    input channels in chunks of 16 (one bf16 MFMA K step per tap); per chunk three PHASES, one per
//             kernel row kh, each 3 taps x 6 products x MT x NT MFMAs per wave
//   xL        the chunk's input tile with halo, split at staging (prologue -- ELU / dropout hash / ReLU mask --
//             and split run ONCE per element): [3 planes][2 k-halves][(TH+2) x 34 pixels] 16-byte units of 8 bf16.
//             A B-fragment read is 32 consecutive units per k-half: conflict-free ds_read_b128, tap shift = immediate.
//   wL        the phase's weights, pre-split by the weight-norm pack kernel (vunet_weightnorm_fwd*: wx image
//             [chunk][kh][m-tile][kw][plane][k-half][32 channels] units), double buffered: the slab of phase p+1 is
//             loaded to registers before the MFMA block of phase p and written after it.
//   barriers  one per phase, plus one per chunk to retire the single-buffered input tile.
// The accumulator layout is that of every 32x32 MFMA, so the fused epilogue (shift, activation, residual,
// depth-to-space; data gradient: * act'(aux) + res) is shared with the fp32 kernels (conv_common.h).
#pragma once
#include <type_traits>

#include "conv_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

union X6Unit {
  uint4 u;
  bf16x8_t b;
};

__device__ __forceinline__ uint32_t x6_pack2(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
  bf16x2_t p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, p);
}

// (a, b) -> packed bf16 pairs of the three split terms; a == ah + am + al exactly (same for b)
__device__ __forceinline__ void x6_split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = x6_pack2(a, b);
  float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = x6_pack2(ra, rb);
  ra -= __uint_as_float(m << 16);
  rb -= __uint_as_float(m & 0xffff0000u);
  l = x6_pack2(ra, rb);
}

constexpr int X6_SLAB = 576;  // 16-byte units of one (chunk, kh, m-tile) weight slab: [3 kw][3 planes][2 halves][32]

// MODE 0 forward, 1 data gradient (taps mirrored).  PRO 0 none, 1 ELU, 2 ELU + dropout, 4 ReLU mask (MODE 1).
// PHW >= 0 (MODE 1 only): data gradient of the STRIDE-2 convolution (Downsample, lib/modules.py:152-158) for the output
// parity (ph, pw) = (PHW >> 1, PHW & 1): dx[2a+ph][2b+pw] = sum over the taps kh = ph+1 (mod 2), kw = pw+1 (mod 2) of
// w[kh][kw] * dy[a + (ph+1-kh)/2][b + (pw+1-kw)/2] -- a stride-1 problem on the dy map with 1, 2, 2 or 4 of the 9
// taps, stored to every other pixel of dx.  Four launches cover the four parities; no MFMA is spent on the structural
// zeros of a transposed strided convolution.
template <int MT, int NT, int MODE, int PRO, int PHW = -1>
__global__ __launch_bounds__(256, 2) void conv_x6_kernel(const GatherArgs a_in, const uint4* __restrict__ wx, int mtiles_pad) {
  GatherArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  static_assert(PHW < 0 || MODE == 1, "parity phases exist for the data gradient only");
  constexpr int PH = PHW >= 0 ? (PHW >> 1) : 0, PW = PHW >= 0 ? (PHW & 1) : 0;
  constexpr int NKH = PHW < 0 ? 3 : (PH ? 2 : 1);   // kernel rows visited per chunk
  constexpr int TW = 32, TH = 4 * NT, IH = TH + 2, IW = TW + 2, PIX = IH * IW, MB = 32 * MT;
  constexpr int XU = 2 * PIX;             // staging units of the input tile: (k-half, pixel)
  constexpr int NX = (XU + 255) / 256;
  constexpr int WU = X6_SLAB * MT;        // units of one weight slab
  constexpr int NW = (WU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* const xL = smem4;                // [3][2][PIX]
  uint4* const wL = smem4 + 6 * PIX;      // [2 buffers][WU]

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

  // ---- chunk-invariant staging geometry: unit u = (k-half c8, halo row r, halo column col), lanes walk columns
  unsigned rel[NX];   // element offsets; loads address as scalar base + unsigned 32-bit BYTE offset (no 64-bit pairs)
  int lds_x[NX];
  uint32_t vbits = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = tid + 256 * i;
    const int c8 = u / PIX;
    const int rem = u - c8 * PIX;
    const int r = rem / IW, col = rem - r * IW;
    const int ih = row0 - 1 + r, iw = col0 - 1 + col;
    const bool ok = u < XU && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    rel[i] = ok ? (unsigned)(8 * c8 * HW + ih * W + iw) : 0u;   // invalid: a safe in-bounds address, masked afterwards
    lds_x[i] = c8 * PIX + rem;
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
  float mk[PRO == 4 ? NX : 1][8];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 struct did not)
  u32x4 wv[NW];
  const int mt0 = (d.m_off + m0) >> 5;   // first m-tile of this workgroup in the wx image

  auto issue_x = [&](int ch) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    const float* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C + cs) * HW;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        xv[i][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xs) +
                                                   4u * (rel[i] + (((vbits >> i) & 1u) ? (unsigned)(k * HW) : 0u)));
  };
  // PRO 4: the ReLU mask travels separately and late (after the MFMA block, when the fragment registers are free):
  // prefetched with the input values it costs 8*NX more live registers across the MFMAs and the tall tiles spill.
  auto issue_mask = [&](int ch) {
    if constexpr (PRO == 4) {
      const float* __restrict__ ms = a.mask + (size_t)(n * d.C1 + ch * 16) * HW;   // mode 1, single source
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k)
          mk[i][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ms) +
                                                     4u * (rel[i] + (((vbits >> i) & 1u) ? (unsigned)(k * HW) : 0u)));
    }
  };
  auto write_x = [&](int ch) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    InAct ia = a.in1;                       // the two sources differ in the dropout seed only
    ia.seed = second ? a.in2.seed : a.in1.seed;
    const int gbase = (n * C + cs) * HW;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (tid + 256 * i < XU) {
        const bool ok = (vbits >> i) & 1u;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float t_ = xv[i][k];
          if constexpr (PRO == 4) t_ = mk[i][k] > 0.f ? t_ : 0.f;
          else if constexpr (PRO != 0) t_ = prologue<PRO>(ia, t_, (uint32_t)gbase + rel[i] + (uint32_t)(k * HW));
          v[k] = ok ? t_ : 0.f;
        }
        uint4 ph, pm, pl;
        x6_split2(v[0], v[1], ph.x, pm.x, pl.x);
        x6_split2(v[2], v[3], ph.y, pm.y, pl.y);
        x6_split2(v[4], v[5], ph.z, pm.z, pl.z);
        x6_split2(v[6], v[7], ph.w, pm.w, pl.w);
        xL[lds_x[i]] = ph;
        xL[2 * PIX + lds_x[i]] = pm;
        xL[4 * PIX + lds_x[i]] = pl;
      }
    }
  };
  // Weight slab staging: WU units over 256 threads, every load and LDS write unconditional -- the threads past the end
  // of the slab wrap around and rewrite its first units with the same bytes.  (A load whose only use sits in a
  // conditional block is sunk into that block by the compiler, next to the ds_write: no prefetch.)
  auto issue_w = [&](int phase) {   // phase = chunk * 3 + kh
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(wx) + ((size_t)phase * mtiles_pad + mt0) * X6_SLAB;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      unsigned w = tid + 256 * i;
      if (w >= (unsigned)WU) w -= WU;
      wv[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wp) + 16u * w);
    }
  };
  auto write_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      int w = tid + 256 * i;
      if (w >= WU) w -= WU;
      reinterpret_cast<u32x4*>(wL)[buf * WU + w] = wv[i];
    }
  };

  // step -> weight slab (chunk * 3 + kh): all three kernel rows, or only those of this output parity
  auto slab_of = [&](int step) {
    if constexpr (PHW < 0) return step;
    else if constexpr (PH == 0) return step * 3 + 1;                     // kh = 1
    else return (step >> 1) * 3 + ((step & 1) ? 2 : 0);                  // kh = 0, 2
  };
  issue_x(0);
  issue_w(slab_of(0));
  issue_mask(0);
  write_x(0);
  write_w(0);
  __syncthreads();

  const uint4* const xB = xL + h * PIX + (wave * NT) * IW + j;   // + plane*2*PIX + (q + dr)*IW + dc
  const uint4* const wA = wL + h * 32 + j;                       // + buf*WU + ((mt*3 + kw)*3 + plane)*64
  // One chunk = three phases (kernel rows).  LAST is a compile-time flag so that every prefetch is unconditional code.
  auto chunk = [&](int ch, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
    for (int ki = 0; ki < NKH; ++ki) {
      const int kh = PHW < 0 ? ki : (PH ? 2 * ki : 1);
      const int phase = ch * NKH + ki;
      const int buf = phase & 1;
      const bool more_w = !(LAST && ki == NKH - 1);
      const bool more_x = !LAST && ki == NKH - 1;
      if (more_x) issue_x(ch + 1);
      if (more_w) issue_w(slab_of(phase + 1));
      // row / column of the staged tile (origin row0-1, col0-1) that tap (kh, kw) reads for output row q, column j
      const int dr = PHW >= 0 ? (PH + 1 - kh) / 2 + 1 : (MODE == 0 ? kh : 2 - kh);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if (PHW >= 0 && ((kw + PW) & 1) == 0) continue;   // this parity's taps only: kw = pw + 1 (mod 2)
        // two waves share a SIMD: the partner's MFMAs cover this wave's fragment reads, so nothing is gained by
        // letting the scheduler hoist the next tap's 6*(MT+NT) fragment registers above this tap's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const int dc = PHW >= 0 ? (PW + 1 - kw) / 2 + 1 : (MODE == 0 ? kw : 2 - kw);
        X6Unit av[3][MT], bv[3][NT];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[p][mt].u = wA[buf * WU + ((mt * 3 + kw) * 3 + p) * 64];
#pragma unroll
          for (int q = 0; q < NT; ++q) bv[p][q].u = xB[p * 2 * PIX + (q + dr) * IW + dc];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            f32x16 c = acc[mt][q];
            // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2][mt].b, bv[0][q].b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][mt].b, bv[2][q].b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1][mt].b, bv[1][q].b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1][mt].b, bv[0][q].b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][mt].b, bv[1][q].b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][mt].b, bv[0][q].b, c, 0, 0, 0);
            acc[mt][q] = c;
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more_w) write_w(buf ^ 1);
      if (more_x) {
        issue_mask(ch + 1);
        __syncthreads();          // every wave has finished reading this chunk's input tile
        write_x(ch + 1);
      }
      if (more_w) __syncthreads();
    }
  };
  for (int ch = 0; ch + 1 < nch; ++ch) chunk(ch, std::false_type{});
  chunk(nch - 1, std::true_type{});

  // ---- epilogue (shared with the fp32 kernels): lane j = pixel (row0 + wave*NT + q, col0 + j).  The tiles are named
  // at compile time (a runtime q would put the accumulators in scratch: the unroller gives up on this body's size).
  auto epilogue = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    PixGeo g;
    g.n = n;
    g.oh = PHW >= 0 ? 2 * (row0 + wave * NT + q) + PH : row0 + wave * NT + q;   // parity phase: every other pixel of dx
    g.ow = PHW >= 0 ? 2 * (col0 + j) + PW : col0 + j;
    g.valid = true;
    store_tile16(a, g, m0, h, acc[0][q]);
    if constexpr (MT > 1) store_tile16(a, g, m0 + 32, h, acc[1][q]);
  };
  epilogue(std::integral_constant<int, 0>{});
  if constexpr (NT > 1) epilogue(std::integral_constant<int, 1>{});
  if constexpr (NT > 2) {
    epilogue(std::integral_constant<int, 2>{});
    epilogue(std::integral_constant<int, 3>{});
  }
}

template <int MT, int NT, int MODE, int PRO, int PHW = -1>
static int launch_x6_one(const GatherArgs& ga, const void* wx, int mtiles_pad, hipStream_t st) {
  constexpr int PIX = (4 * NT + 2) * 34;
  constexpr size_t lds = (size_t)(6 * PIX + 2 * X6_SLAB * MT) * 16;
  const vunet_conv_desc& d = ga.d;
  const int blocks = d.N * (d.Hs / (4 * NT)) * (d.Ws / 32) * ((d.M + 32 * MT - 1) / (32 * MT));
  auto kern = conv_x6_kernel<MT, NT, MODE, PRO, PHW>;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  VUNET_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds, st, ga, (const uint4*)wx, mtiles_pad);
  return vunet_check_launch();
}

template <int MT, int NT>
static int launch_x6(const GatherArgs& ga, const void* wx, int mtiles_pad, int pro, hipStream_t st) {
  if (ga.d.mode == 1 && ga.d.stride == 2) {   // one launch per output parity
    if (pro != 0) return VUNET_ERR_UNSUPPORTED;
    int rc = launch_x6_one<MT, NT, 1, 0, 0>(ga, wx, mtiles_pad, st);
    if (rc == VUNET_OK) rc = launch_x6_one<MT, NT, 1, 0, 1>(ga, wx, mtiles_pad, st);
    if (rc == VUNET_OK) rc = launch_x6_one<MT, NT, 1, 0, 2>(ga, wx, mtiles_pad, st);
    if (rc == VUNET_OK) rc = launch_x6_one<MT, NT, 1, 0, 3>(ga, wx, mtiles_pad, st);
    return rc;
  }
  if (ga.d.mode == 1) {
    if (pro == 4) return launch_x6_one<MT, NT, 1, 4>(ga, wx, mtiles_pad, st);
    if (pro == 0) return launch_x6_one<MT, NT, 1, 0>(ga, wx, mtiles_pad, st);
    return VUNET_ERR_UNSUPPORTED;
  }
  switch (pro) {
    case 0: return launch_x6_one<MT, NT, 0, 0>(ga, wx, mtiles_pad, st);
    case 1: return launch_x6_one<MT, NT, 0, 1>(ga, wx, mtiles_pad, st);
    case 2: return launch_x6_one<MT, NT, 0, 2>(ga, wx, mtiles_pad, st);
    default: return VUNET_ERR_UNSUPPORTED;
  }
}
