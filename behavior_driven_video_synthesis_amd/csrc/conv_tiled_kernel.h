// LDS-tiled implicit-GEMM 3x3 / stride 1 / pad 1 convolution on fp32 MFMA (forward and data gradient).
//
// A workgroup (4 waves) owns MT*32 output channels x (4*NT rows x 32 columns) of one image.  The K
// loop walks the input channels in chunks of CK.  Per chunk the workgroup stages, exactly once per
// element,
//   xL[CK][TH+2][34]  -- the input tile with its halo: coalesced NCHW row segments, the prologue
//                        (ELU / dropout hash) applied here instead of once per tap per consumer,
//   wL[9][CK][MT*32]  -- the chunk's K-major effective weights,
// into one of two LDS buffers, while the MFMAs of the previous chunk run from the other buffer: the
// global loads of chunk c+1 are issued before the MFMA block of chunk c and written to LDS after it
// (register-staged, write-late), so HBM/L2 latency hides under 36*MT*NT MFMAs and there is one
// barrier per chunk.  Every LDS fragment read is a conflict-free ds_read_b32 with a compile-time
// offset from one per-lane base (lanes 0-31 read 32 consecutive dwords; the tap shift is an
// immediate), so the loop carries no address arithmetic.
//
// v_mfma_f32_32x32x2_f32: A = wL (lane: channel l&31, k-half l>>5), B = xL (lane: pixel column
// l&31, k-half l>>5).  Pixels are the MFMA column: each accumulator register stores 32 consecutive
// NCHW pixels of one channel.  The data gradient is the same kernel with the taps mirrored
// (MODE 1) and the epilogue  y = acc * act'(aux) + res.
#pragma once
#include "conv_common.h"

// TW = 32: an MFMA pixel tile is one 32-pixel row segment; TW = 16 (16-wide maps): two 16-pixel row segments.
// S = 2: the stride-2 Downsample conv (forward only): the staged input tile is (2*TH+1) x (2*TW+1).
template <int MT, int NT, int CK, int MODE, int PRO, int TW = 32, int S = 1>
__global__ __launch_bounds__(256, 2) void conv_tiled_kernel(const GatherArgs a_in) {
  GatherArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  constexpr int RPT = 32 / TW;                  // rows per MFMA pixel tile
  constexpr int TH = 4 * NT * RPT, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int MB = 32 * MT;
  constexpr int EI = CK * IH * IW;          // input elements per chunk
  constexpr int EW = 9 * CK * MB;           // weight elements per chunk
  constexpr int NI = (EI + 255) / 256;
  constexpr int NW = EW / 256;
  constexpr int BUF = EI + EW;              // floats per LDS buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;

  // ---- tile of this workgroup (m-block fastest so the blocks sharing an input tile are adjacent)
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = (d.M + MB - 1) / MB;
  const int mb = bid % mblocks;
  int t = bid / mblocks;
  const int tiles_w = d.Wo / TW, tiles_h = d.Ho / TH;   // tiles walk the OUTPUT map
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * TH, col0 = tx * TW, m0 = mb * MB;
  const int jr = j / TW, jc = j % TW;           // this lane's row / column inside an MFMA pixel tile

  // ---- chunk-invariant staging geometry of this thread
  int rel[NI];
  uint32_t vbits = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + 256 * i;
    const int c = e / (IH * IW);
    const int rem = e - c * (IH * IW);
    const int r = rem / IW, col = rem - r * IW;
    const int ih = row0 * S - 1 + r, iw = col0 * S - 1 + col;
    const bool ok = e < EI && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    rel[i] = ok ? c * HW + ih * W + iw : 0;  // invalid: a safe in-bounds address, the value is masked afterwards
    vbits |= (ok ? 1u : 0u) << i;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;

  const int nch1 = d.C1 / CK, nch = nch1 + d.C2 / CK;
  float xv[NI], wv[NW];
  float mk[PRO == 4 ? NI : 1];   // PRO 4 (MODE 1 only): ReLU mask values travelling with xv

  auto issue_loads = [&](int ch) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * CK : ch * CK;
    const int C = second ? d.C2 : d.C1;
    const float* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C + cs) * HW;
#pragma unroll
    for (int i = 0; i < NI; ++i) xv[i] = xs[rel[i]];  // unconditional (no branch per load): masked in write_lds
    if constexpr (PRO == 4) {
      const float* __restrict__ ms = a.mask + (size_t)(n * C + cs) * HW;
#pragma unroll
      for (int i = 0; i < NI; ++i) mk[i] = ms[rel[i]];
    }
    const int Cp = (C + 1) & ~1;
    const int krow0 = second ? 9 * ((d.C1 + 1) & ~1) : 0;
    const float* __restrict__ wp = a.wt + (size_t)(krow0 + cs) * d.Mpad + d.m_off + m0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + 256 * i;            // LDS order [tap][c][m]
      const int m = e % MB, rc = e / MB;
      const int c = rc % CK, tap = rc / CK;
      const bool mv = m0 + m < d.M;
      const float w = wp[(size_t)(tap * Cp + c) * d.Mpad + (mv ? m : 0)];
      wv[i] = mv ? w : 0.f;
    }
  };
  auto write_lds = [&](int ch, float* buf) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * CK : ch * CK;
    const int C = second ? d.C2 : d.C1;
    const InAct& ia = second ? a.in2 : a.in1;
    const int gbase = (n * C + cs) * HW;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      if (e < EI) {
        float v = xv[i];
        if constexpr (PRO == 4) v = mk[i] > 0.f ? v : 0.f;
        else if (PRO != 0) v = prologue<PRO>(ia, v, (uint32_t)(gbase + rel[i]));
        buf[e] = ((vbits >> i) & 1u) ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) buf[EI + tid + 256 * i] = wv[i];
  };

  issue_loads(0);
  write_lds(0, smem);
  __syncthreads();

  const int aoff = h * MB + j;                                  // A: wL[(tap*CK + 2p + h)*MB + mt*32 + j]
  const int boff = h * IH * IW + (wave * NT * RPT + jr) * S * IW + jc * S;  // B: xL[(2p+h)*IH*IW + (S*row + dr)*IW + S*col + dc]
  for (int ch = 0; ch < nch; ++ch) {
    const float* buf = smem + (ch & 1) * BUF;
    if (ch + 1 < nch) issue_loads(ch + 1);
    const float* xL = buf + boff;
    const float* wL = buf + EI + aoff;
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap % 3;
        const int dr = MODE == 0 ? kh : 2 - kh, dc = MODE == 0 ? kw : 2 - kw;
        float av[MT], bv[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = wL[(tap * CK + 2 * p) * MB + mt * 32];
#pragma unroll
        for (int q = 0; q < NT; ++q) bv[q] = xL[(2 * p) * IH * IW + (q * RPT * S + dr) * IW + dc];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < NT; ++q)
            acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[q], acc[mt][q], 0, 0, 0);
      }
    }
    if (ch + 1 < nch) write_lds(ch + 1, smem + ((ch + 1) & 1) * BUF);
    __syncthreads();
  }

  // ---- epilogue: lane j = pixel (row0 + (wave*NT + q)*RPT + jr, col0 + jc)
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    PixGeo g;
    g.n = n;
    g.oh = row0 + (wave * NT + q) * RPT + jr;
    g.ow = col0 + jc;
    g.valid = true;
    if constexpr (NT == 1) {
      // small tile (the VALU-heavy layers): issue the tile's aux / residual loads together
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) store_tile16(a, g, m0 + mt * 32, h, acc[mt][q]);
    } else {
      // tall tiles run at the VGPR limit: keep the register-lean per-element epilogue
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < d.M) store_out(a, g, m, acc[mt][q][r]);
        }
    }
  }
}

template <int MT, int NT, int CK, int MODE, int TW, int S>
static int launch_tiled_pro(const GatherArgs& ga, int pro, int blocks, hipStream_t st) {
  constexpr int BUF = CK * ((4 * NT * (32 / TW) - 1) * S + 3) * ((TW - 1) * S + 3) + 9 * CK * 32 * MT;
  const size_t lds = 2 * (size_t)BUF * sizeof(float);
  dim3 grid((unsigned)blocks), block(256);
  if (lds > 64 * 1024) {  // opt in to > 64 KiB of dynamic LDS (once per instantiation is enough; it is cheap)
    hipFuncSetAttribute((const void*)conv_tiled_kernel<MT, NT, CK, MODE, 0, TW, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (MODE == 1 && S == 1)
      hipFuncSetAttribute((const void*)conv_tiled_kernel<MT, NT, CK, 1, 4, TW, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (MODE == 0) {
      hipFuncSetAttribute((const void*)conv_tiled_kernel<MT, NT, CK, MODE, 1, TW, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute((const void*)conv_tiled_kernel<MT, NT, CK, MODE, 2, TW, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
  }
  if constexpr (MODE == 1) {
    if (pro == 4) {
      if constexpr (S == 1) VUNET_LAUNCH((conv_tiled_kernel<MT, NT, CK, 1, 4, TW, S>), grid, block, lds, st, ga);
      else return VUNET_ERR_UNSUPPORTED;
    } else {
      VUNET_LAUNCH((conv_tiled_kernel<MT, NT, CK, 1, 0, TW, S>), grid, block, lds, st, ga);
    }
  } else {
    switch (pro) {
      case 0: VUNET_LAUNCH((conv_tiled_kernel<MT, NT, CK, 0, 0, TW, S>), grid, block, lds, st, ga); break;
      case 1: VUNET_LAUNCH((conv_tiled_kernel<MT, NT, CK, 0, 1, TW, S>), grid, block, lds, st, ga); break;
      case 2: VUNET_LAUNCH((conv_tiled_kernel<MT, NT, CK, 0, 2, TW, S>), grid, block, lds, st, ga); break;
      default: return VUNET_ERR_UNSUPPORTED;
    }
  }
  return vunet_check_launch();
}

template <int MT, int NT, int CK, int TW = 32, int S = 1>
static int launch_tiled(const GatherArgs& ga, int pro, hipStream_t st) {
  const vunet_conv_desc& d = ga.d;
  const int mblocks = (d.M + 32 * MT - 1) / (32 * MT);
  const int blocks = d.N * (d.Ho / (4 * NT * (32 / TW))) * (d.Wo / TW) * mblocks;
  if (d.mode == 1) {
    if (pro != 0 && pro != 4) return VUNET_ERR_UNSUPPORTED;
    return launch_tiled_pro<MT, NT, CK, 1, TW, S>(ga, pro, blocks, st);
  }
  return launch_tiled_pro<MT, NT, CK, 0, TW, S>(ga, pro, blocks, st);
}

