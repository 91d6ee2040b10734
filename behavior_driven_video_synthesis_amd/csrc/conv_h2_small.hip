// fp32-accurate 3x3 / pad 1 convolution on SMALL maps (4 x 4 ... 32 x 32 with few pixels in the whole batch) on the fp16
// matrix cores: the "h2" operand split of conv_h2_kernel.h (two scaled fp16 terms, three partial products, the same
// weight image and |x| maxima protocol) for the layers whose launch cannot fill the chip with the row-tiled kernel --
// the bottleneck of the VUnet (models/vunets.py:520-597, :264-424: maps of 4 x 4, 8 x 8, 16 x 16 at 128 channels) and the
// Downsample convolutions that lead there (lib/modules.py:148-161).  Until round 3 these ran on fp32-input MFMA gather
// kernels: 15 - 250 us per launch for 0.04 - 1.2 GFLOP, a sixteenth of the fp16 rate and latency bound.
//
// Covers, with ONE code path (all index arithmetic is per-lane and done once, before the K loop):
//   mode 0  forward, stride 1 or stride 2                 ih = s * oh - 1 + kh
//   mode 1  data gradient, stride 1                       ih = oh + 1 - kh
//   mode 1  data gradient of the stride-2 convolution     ih = (oh + 1 - kh) / 2 where that is an integer; the output
//           pixels are enumerated one output PARITY CLASS (oh & 1, ow & 1) after the other, so that a 32-pixel tile has
//           one parity and the taps it cannot use are skipped by a wave-uniform vote: no MFMA on structural zeros
// prologue none / ELU / ELU + dropout, dual source, every epilogue of the generic store (shift, activation, residual,
// depth-to-space, act'(aux) of the data gradient).
//
// Work decomposition: the whole batch is ONE pixel axis.  A workgroup (8 waves) owns one 32-channel x 32-pixel output
// tile and SPLITS K over its waves: wave w sums the 16-channel chunks w, w + 8, ... of the input, on a staging area of
// its own in LDS -- no barrier inside the K loop -- and the eight partial tiles are summed through LDS in a fixed order.
// Per chunk a wave stages the input rows its tile can touch (whole images when the tile spans several) as 16-byte units
// of 8 fp16 [plane][k-half][slot], one lane per unit (8 strided loads, prologue, scale, split, two ds_write_b128), then
// runs the 9 taps: the B fragment of tap t is the unit at this lane's precomputed slot for t (the ZERO unit behind the
// last slot where the tap falls outside the map / has the wrong parity), the A fragment comes straight from the weight
// image in L2 (a wave reads 1 KiB contiguous per fragment: the image IS the fragment layout).
#include <type_traits>

#include "conv_common.h"
#include "split_h2.h"

struct SmallGeo {
  int ntile;      // 32-pixel tiles
  int ppc;        // pixels per parity class (stride-2 data gradient), else NP
  int cw, chw;    // width / size of the per-image pixel grid the classes enumerate (sub-grid for parity classes)
  int nclass;     // 1, or 4 parity classes
  int smax;       // upper bound of the staged slots of any tile (LDS sizing)
};


union SmUnit {
  uint4 u;
  h2_f16x8 b;
};

// pixel P of the enumeration -> output pixel.  Classes: P = (class, n, a, b) with (oh, ow) = (2a + ph, 2b + pw).
__device__ __forceinline__ PixGeo sm_pixel(const GatherArgs& a, const SmallGeo& sg, int P) {
  PixGeo g;
  g.valid = P < a.NP;
  const int Pc = g.valid ? P : 0;
  const int cls = sg.nclass > 1 ? Pc / sg.ppc : 0;
  const int rem = Pc - cls * sg.ppc;
  g.n = rem / sg.chw;
  const int r2 = rem - g.n * sg.chw;
  const int ra = r2 / sg.cw, rb = r2 - ra * sg.cw;
  if (sg.nclass > 1) {
    g.oh = 2 * ra + (cls >> 1);
    g.ow = 2 * rb + (cls & 1);
  } else {
    g.oh = ra;
    g.ow = rb;
  }
  return g;
}

// input rows output row oh can touch: [lo, hi]
__device__ __forceinline__ void sm_rows(const vunet_conv_desc& d, int oh, int& lo, int& hi) {
  if (d.mode == 0) {
    lo = oh * d.stride - 1;
    hi = lo + 2;
  } else if (d.stride == 1) {
    lo = oh - 1;
    hi = oh + 1;
  } else {
    lo = (oh - 1) >> 1;   // kh = 2 ... kh = 0 (arithmetic shift: -1 -> -1)
    hi = (oh + 1) >> 1;
  }
}

// input pixel of tap (kh, kw) for output (oh, ow); false: outside the map or not on the stride grid
__device__ __forceinline__ bool sm_tap(const vunet_conv_desc& d, int oh, int ow, int kh, int kw, int& ih, int& iw) {
  if (d.mode == 0) {
    ih = oh * d.stride - 1 + kh;
    iw = ow * d.stride - 1 + kw;
  } else if (d.stride == 1) {
    ih = oh + 1 - kh;
    iw = ow + 1 - kw;
  } else {
    const int th = oh + 1 - kh, tw = ow + 1 - kw;
    if ((th | tw) & 1) return false;
    ih = th >> 1;
    iw = tw >> 1;
  }
  return (unsigned)ih < (unsigned)d.Hs && (unsigned)iw < (unsigned)d.Ws;
}

// NWV waves per workgroup share K: 8, or 4 where five staging rounds need more than 256 registers per lane
template <int PRO, int NR, int NWV = (NR > 3 ? 4 : 8)>
__global__ __launch_bounds__(64 * NWV) void conv_h2_small_kernel(const GatherArgs a_in, const uint4* __restrict__ wx,
                                                                      int mtiles_pad, const float* __restrict__ amax,
                                                                      const SmallGeo sg) {
  GatherArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int mblocks = d.M >> 5;
  const int mb = blockIdx.x % mblocks, pt = blockIdx.x / mblocks;   // m fastest: the m-tiles of one pixel tile share L2 lines
  const int m0 = mb * 32;
  const int HW = a.HsWs, W = d.Ws;

  // ---- the staged region of this tile (wave-uniform): images n0 .. n1, rows r_lo .. r_hi - 1 of each
  const int P0 = pt * 32, P1 = min(P0 + 31, a.NP - 1);
  const PixGeo g0 = sm_pixel(a, sg, P0), g1 = sm_pixel(a, sg, P1);
  int r_lo = 0, r_hi = d.Hs;
  if (g0.n == g1.n) {
    int lo0, hi0, lo1, hi1;
    sm_rows(d, g0.oh, lo0, hi0);
    sm_rows(d, g1.oh, lo1, hi1);
    r_lo = max(0, min(lo0, lo1));
    r_hi = min(d.Hs, max(hi0, hi1) + 1);
  }
  const int n0 = g0.n, RH = r_hi - r_lo, per_img = RH * W;
  const int S = (g1.n - g0.n + 1) * per_img;   // slots; slot S is the zero unit
  const int SP = S + 1;

  // ---- this lane's output pixel and its nine tap slots
  const PixGeo g = sm_pixel(a, sg, P0 + j);
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    int ih, iw;
    const bool ok = g.valid && sm_tap(d, g.oh, g.ow, t / 3, t % 3, ih, iw);
    toff[t] = ok ? (g.n - n0) * per_img + (ih - r_lo) * W + iw : S;
  }

  uint4* const xw = smem4 + (size_t)wave * 4 * SP;   // this wave's staging area: [plane][k-half][SP]
  // zero unit (written once; the chunk loop never touches slot S)
  if (lane < 4) xw[lane * SP + S] = make_uint4(0u, 0u, 0u, 0u);

  // ---- staging geometry of this lane's units (chunk-invariant): unit u = lane + 64 r -> (k-half, slot)
  unsigned rel[NR];
  int lds_u[NR];
  unsigned vbits = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int u = lane + 64 * r;
    const bool ok = u < 2 * S;
    const int uu = ok ? u : 0;
    const int c8 = uu >= S ? 1 : 0;
    const int s = uu - c8 * S;
    const int img = s / per_img, rem = s - img * per_img;
    rel[r] = (unsigned)(8 * c8 * HW + (r_lo * W + rem));   // + (n0 + img) * C * HW + chunk base, added per chunk
    lds_u[r] = c8 * SP + s;
    vbits |= (ok ? 1u : 0u) << r;
    // the image index rides in the upper bits of rel's companion: keep it separately
    lds_u[r] |= img << 20;
  }

  f32x16 acc, acx;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acx[r] = 0.f;

  const int nch1 = d.C1 >> 4, nch = nch1 + (d.C2 >> 4);
  const int mt0 = (d.m_off + m0) >> 5;
  const uint4* const wA = wx + 1 + h * 32 + j;   // + ((((ch*3 + kh)*mtp + mt0)*3 + kw)*2 + plane)*64

  float sx, descale, descale2;
  // rounds 0 .. NRP-1 are register-prefetched one chunk ahead (live across the MFMA block); the rounds beyond -- only the
  // stride-2 forward of the larger maps has them -- are loaded at the top of their own chunk (two waves per SIMD leave 256
  // registers per lane: 72 of A fragments + 32 accumulators + 8 per prefetched round)
  constexpr int NRP = NR < 3 ? NR : 3, NRL = NR - NRP;
  float xv[NRP][8], xt[NRL > 0 ? NRL : 1][8];
  auto load_rounds = [&](int ch, auto r0c, auto r1c, auto& buf) {
    constexpr int R0 = decltype(r0c)::value, R1 = decltype(r1c)::value;
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    const float* __restrict__ xs = second ? a.x2 : a.x1;
#pragma unroll
    for (int r = R0; r < R1; ++r) {
      const int img = lds_u[r] >> 20;
      const float* p = xs + (size_t)((n0 + img) * C + cs) * HW + rel[r];
      const bool ok = (vbits >> r) & 1u;
#pragma unroll
      for (int k = 0; k < 8; ++k) buf[r - R0][k] = p[ok ? k * HW : 0];
    }
  };
  auto write_rounds = [&](int ch, auto r0c, auto r1c, auto& buf) {
    constexpr int R0 = decltype(r0c)::value, R1 = decltype(r1c)::value;
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    InAct ia = a.in1;
    ia.seed = second ? a.in2.seed : a.in1.seed;
#pragma unroll
    for (int r = R0; r < R1; ++r) {
      const int img = lds_u[r] >> 20;
      const uint32_t gbase = (uint32_t)((n0 + img) * C + cs) * (uint32_t)HW + rel[r];
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t_ = buf[r - R0][k];
        if constexpr (PRO != 0) t_ = prologue<PRO>(ia, t_, gbase + (uint32_t)(k * HW));
        v[k] = t_ * sx;
      }
      uint4 ph, pl;
      h2_split2(v[0], v[1], ph.x, pl.x);
      h2_split2(v[2], v[3], ph.y, pl.y);
      h2_split2(v[4], v[5], ph.z, pl.z);
      h2_split2(v[6], v[7], ph.w, pl.w);
      if ((vbits >> r) & 1u) {
        const int lu = lds_u[r] & 0xFFFFF;
        xw[lu] = ph;
        xw[2 * SP + lu] = pl;
      }
    }
  };
  using IC0 = std::integral_constant<int, 0>;
  using ICP = std::integral_constant<int, NRP>;
  using ICN = std::integral_constant<int, NR>;
  auto issue_x = [&](int ch) { load_rounds(ch, IC0{}, ICP{}, xv); };
  auto write_x = [&](int ch) {
    if constexpr (NRL > 0) load_rounds(ch, ICP{}, ICN{}, xt);
    write_rounds(ch, IC0{}, ICP{}, xv);
    if constexpr (NRL > 0) write_rounds(ch, ICP{}, ICN{}, xt);
  };

  // A fragments of one chunk, all nine taps (72 registers): requested in one go -- the weight image is cold in this
  // XCD's L2 on first touch, and nine dependent round trips (one per tap) were the kernel's whole run time
  SmUnit av[9][2];
  auto issue_w = [&](int ch) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const uint4* const wp = wA + ((size_t)((ch * 3 + t / 3) * mtiles_pad + mt0) * 3 + t % 3) * 128;
      av[t][0].u = wp[0];
      av[t][1].u = wp[64];
    }
  };

  if (wave < nch) {
    issue_x(wave);
    issue_w(wave);
  }
  // ---- scales (behind the first chunk's loads)
  {
    float m_;   // the 1024 partial maxima: 512 threads x 2 or 256 x 4
    if constexpr (NWV == 8) {
      const float2 pm = h2_amax2(amax, a.amax2, tid);
      m_ = fmaxf(pm.x, pm.y);
    } else {
      const float4 pm = h2_amax4(amax, a.amax2, tid);
      m_ = fmaxf(fmaxf(pm.x, pm.y), fmaxf(pm.z, pm.w));
    }
    m_ = wave_max(m_);
    __shared__ float redm[NWV];
    if (lane == 0) redm[wave] = m_;
    __syncthreads();
    m_ = redm[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m_ = fmaxf(m_, redm[w]);
    if (a.in1.thresh) m_ *= a.in1.keep_scale;
    const int ex = h2_scale_exp(m_);
    const int ew = reinterpret_cast<const int*>(wx)[0];
    sx = h2_pow2(ex);
    h2_pow2_pair(-(ex + ew), descale, descale2);
  }


  for (int ch = wave; ch < nch; ch += NWV) {
    // the previous chunk's fragment reads were issued before these writes (LDS is in order per wave); the asm keeps the
    // compiler from moving them across
    asm volatile("" ::: "memory");
    write_x(ch);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // prefetches are unconditional (the last chunk re-requests itself): branch-free code, so the B reads and the next
    // chunk's loads run ahead of the MFMAs
    const int chn = ch + NWV < nch ? ch + NWV : ch;
    issue_x(chn);
    // every tap, valid or not (an invalid tap reads the zero unit); at these sizes the matrix pipe is idle anyway
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      SmUnit bh, bl;
      bh.u = xw[h * SP + toff[t]];
      bl.u = xw[(2 + h) * SP + toff[t]];
      acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t][1].b, bh.b, acx, 0, 0, 0);
      acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t][0].b, bl.b, acx, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t][0].b, bh.b, acc, 0, 0, 0);
      // this tap's A registers are free again: request the next chunk's
      const uint4* const wp = wA + ((size_t)((chn * 3 + t / 3) * mtiles_pad + mt0) * 3 + t % 3) * 128;
      av[t][0].u = wp[0];
      av[t][1].u = wp[64];
    }
  }

  // ---- sum the NWV partial tiles (fixed order) and store: wave w finishes 16 / NWV accumulator registers
  __syncthreads();
  float* const red = reinterpret_cast<float*>(smem4);   // [NWV][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r] + acx[r] * (1.f / 2048.f);
  __syncthreads();
  float vmax = 0.f;
#pragma unroll
  for (int rr = 0; rr < 16 / NWV; ++rr) {
    const int r = (16 / NWV) * wave + rr;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += red[(w * 16 + r) * 64 + lane];
    v = v * descale * descale2;
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (g.valid) vmax = fmaxf(vmax, fabsf(store_out(a, g, m, v)));
  }
  if (a.amax_out) publish_amax(a, vmax);
}

// ---- host side ----------------------------------------------------------------------------------------------------
static bool small_geo(const vunet_conv_desc* d, SmallGeo& sg) {
  const int NP = d->N * d->Ho * d->Wo;
  sg.nclass = 1;
  sg.cw = d->Wo;
  sg.chw = d->Ho * d->Wo;
  sg.ppc = NP;
  if (d->mode == 1 && d->stride == 2) {
    if (d->Ho != 2 * d->Hs || d->Wo != 2 * d->Ws) return false;
    sg.nclass = 4;
    sg.cw = d->Ws;
    sg.chw = d->Hs * d->Ws;
    sg.ppc = d->N * sg.chw;
    if (sg.ppc % 32) return false;   // a tile must not straddle two parity classes
  }
  sg.ntile = (NP + 31) / 32;
  // staged slots of the worst tile (32 consecutive pixels of the class grid): an upper bound of the kernel's S
  int smax;
  if (sg.chw % 32 == 0) {   // tiles never leave their image: the input rows of the class-grid rows they touch
    int orows = 32 % sg.cw == 0 ? 32 / sg.cw : 31 / sg.cw + 2;
    if (orows > sg.chw / sg.cw) orows = sg.chw / sg.cw;
    int irows;
    if (d->mode == 0) irows = (orows - 1) * d->stride + 3;
    else if (d->stride == 1) irows = orows + 2;
    else irows = orows + 1;
    if (irows > d->Hs) irows = d->Hs;
    smax = irows * d->Ws;
  } else {                  // tiles may span images: whole images, as many as 32 consecutive pixels can touch
    int imgs = 32 % sg.chw == 0 ? 32 / sg.chw : 31 / sg.chw + 2;   // (aligned when an image divides the tile)
    if (imgs > d->N) imgs = d->N;
    smax = imgs * d->Hs * d->Ws;
  }
  sg.smax = smax;
  return true;
}

// geometry the small-map kernel covers (not whether it is the better choice: vunet_conv_h2_small_wanted)
bool vunet_conv_h2_small_ok(const vunet_conv_desc* d, int pro) {
  if (d->KH != 3 || d->KW != 3 || d->pad != 1) return false;
  if (d->C1 <= 0 || d->C1 % 16 || d->C2 % 16 || d->M <= 0 || d->M % 32 || d->m_off % 32) return false;
  if (d->stride != 1 && d->stride != 2) return false;
  if (d->mode == 0) {
    if (pro != 0 && pro != 1 && pro != 2) return false;
    if (d->Ho != (d->Hs + 2 - 3) / d->stride + 1 || d->Wo != (d->Ws + 2 - 3) / d->stride + 1) return false;
  } else {
    if (pro != 0 || d->C2 != 0) return false;
    if (d->stride == 1 && (d->Ho != d->Hs || d->Wo != d->Ws)) return false;
  }
  if (d->Ws > 32) return false;
  SmallGeo sg;
  if (!small_geo(d, sg)) return false;
  return sg.smax <= 160;
}

// the launch is small enough that the K-split workgroups beat every row-tiled kernel: few tiles in the whole batch
bool vunet_conv_h2_small_wanted(const vunet_conv_desc* d, int pro) {
  if (!vunet_conv_h2_small_ok(d, pro)) return false;
  const long tiles = ((long)d->N * d->Ho * d->Wo + 31) / 32 * (d->M / 32);
  return tiles <= 2048;
}

template <int PRO>
static int small_launch_pro(const GatherArgs& ga, const void* wx, int mtp, const float* amax, const SmallGeo& sg,
                            hipStream_t st) {
  const int nr = (2 * sg.smax + 63) / 64;
  const int nwv = nr > 3 ? 4 : 8;
  const size_t lds_stage = (size_t)nwv * 4 * (sg.smax + 1) * 16, lds_red = (size_t)nwv * 16 * 64 * 4;
  const size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
  const dim3 grid((unsigned)(sg.ntile * (ga.d.M / 32))), block(64 * nwv);
#define SM_LAUNCH(NR_)                                                                                              \
  do {                                                                                                              \
    auto kern = conv_h2_small_kernel<PRO, NR_>;                                                                      \
    if (lds > 64 * 1024) {                                                                                           \
      static bool attr_set = false;                                                                                  \
      if (!attr_set) {                                                                                               \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) \
          return VUNET_ERR_LAUNCH;                                                                                   \
        attr_set = true;                                                                                             \
      }                                                                                                              \
    }                                                                                                                \
    VUNET_LAUNCH(kern, grid, block, lds, st, ga, (const uint4*)wx, mtp, amax, sg);                                   \
  } while (0)
  if (nr <= 1) SM_LAUNCH(1);
  else if (nr <= 2) SM_LAUNCH(2);
  else if (nr <= 3) SM_LAUNCH(3);
  else SM_LAUNCH(5);
#undef SM_LAUNCH
  return vunet_check_launch();
}

int vunet_conv_h2_small_launch(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro,
                               hipStream_t st) {
  SmallGeo sg;
  if (!small_geo(&ga.d, sg) || sg.smax > 160) return VUNET_ERR_UNSUPPORTED;
  switch (pro) {
    case 0: return small_launch_pro<0>(ga, wx, mtiles_pad, amax, sg, st);
    case 1: return small_launch_pro<1>(ga, wx, mtiles_pad, amax, sg, st);
    case 2: return small_launch_pro<2>(ga, wx, mtiles_pad, amax, sg, st);
    default: return VUNET_ERR_UNSUPPORTED;
  }
}

int vunet_conv_h2_small_name(const vunet_conv_desc* d, int pro, char* name, int len) {
  SmallGeo sg;
  if (!small_geo(d, sg)) return VUNET_ERR_UNSUPPORTED;
  const int nr = (2 * sg.smax + 63) / 64;
  snprintf(name, len, "conv_h2_small_kernel<%d, %d>", pro, nr <= 1 ? 1 : nr <= 2 ? 2 : nr <= 3 ? 3 : 5);
  return VUNET_OK;
}
