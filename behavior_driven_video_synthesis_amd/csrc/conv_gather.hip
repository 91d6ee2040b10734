// Generic implicit-GEMM convolution on fp32 MFMA, operands gathered straight from L1/L2.
//
//   D[m][px] = sum_k  W[k][m] * f(X)[k][px]          k = (source, channel pair, tap)
//
// One wave owns an (MT*32 channels) x (NT*32 pixels) output tile and issues
// v_mfma_f32_32x32x2_f32: A = weights (lane: channel l&31, k-half l>>5), B = activations
// (lane: pixel l&31, k-half l>>5).  Pixels are the MFMA column so that every accumulator
// register stores 32 consecutive NCHW pixels of one channel (128-B segments).
//
// K order is channel-pair outer, tap inner: the taps of one channel pair re-read the same cache
// lines (L1 hits) and form one software-pipelined group -- all (MT+NT)*G loads of a group are
// issued before its MT*NT*G MFMAs, so a wave keeps tens of loads in flight instead of paying one
// L2 round trip per MFMA.
//
// This kernel makes no assumption on the geometry (any kernel size / stride / padding / map size,
// single or dual source, forward or transposed gather): it is the data-gradient kernel and the
// fallback for shapes the LDS-tiled kernel (conv_tiled.hip) does not take.  Small maps (few output
// tiles, long K) go to the split-K variant: 16 waves of one workgroup share one 32x32 output tile,
// each sums a strided subset of the channel pairs, partial tiles are reduced through LDS in a fixed
// order (deterministic).
#include "conv_common.h"

// per-lane geometry of one 32-pixel tile: source base offsets and the tap validity bits
struct TileGeo {
  int base1, base2;
  uint32_t vmask;  // bits 0..7 rows (kh), 8..15 cols (kw); 0 if the pixel is out of range
};

// output pixel of flattened index P (phase mode: P walks the sub-grid of one output parity)
__device__ __forceinline__ PixGeo out_pixel(const GatherArgs& a, int P) {
  if (a.ph < 0) return decompose(P, a.NP, a.HoWo, a.d.Wo);
  PixGeo g = decompose(P, a.NP, a.subHW, a.subW);
  g.oh = g.oh * a.d.stride + a.ph;
  g.ow = g.ow * a.d.stride + a.pw;
  return g;
}

__device__ __forceinline__ TileGeo tile_geometry(const GatherArgs& a, int P) {
  const vunet_conv_desc& d = a.d;
  const PixGeo g = out_pixel(a, P);
  const int s = d.stride, p = d.pad;
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
  TileGeo t;
  t.base1 = g.n * d.C1 * a.HsWs + sp;
  t.base2 = g.n * d.C2 * a.HsWs + sp;
  t.vmask = g.valid ? (rb | (cb << 8)) : 0u;
  return t;
}

__device__ __forceinline__ int tap_offset(const vunet_conv_desc& d, int kh, int kw) {
  return d.mode == 0 ? kh * d.Ws + kw : -((kh / d.stride) * d.Ws + kw / d.stride);
}

// One K sweep over [c2_begin, Cp) step c2_step of one source, all taps.
// KS: 3 -> 3x3 taps unrolled as one 9-step group per channel pair; 1 -> 1x1, groups of 8 channel
// pairs; 0 -> generic runtime KH x KW, one step at a time.
template <int MT, int NT, int PRO, int KS>
__device__ __forceinline__ void k_sweep(const GatherArgs& a, const float* __restrict__ xs, const InAct& ia, int C,
                                        int krow0, const int* base, const uint32_t* vmask, const bool* mok, int m0,
                                        int j, int h, int c2_begin, int c2_step, f32x16 (&acc)[MT][NT]) {
  const vunet_conv_desc& d = a.d;
  const int Cp = (C + 1) & ~1;
  const int HW2 = a.HsWs;
  const float* __restrict__ wbase = a.wt + (size_t)d.m_off + m0 + j;
  int wcol[MT];  // column offset of each m-tile; out-of-range channels read column 0 of the row and are masked
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) wcol[mt] = mok[mt] ? mt * 32 : -(d.m_off + m0 + j);

  if (KS == 3) {
    int toff[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) toff[t9] = tap_offset(d, t9 / 3, t9 % 3);
    for (int c2 = c2_begin; c2 < Cp; c2 += c2_step) {
      const int ci = c2 + h;
      const bool cok = ci < C;
      float av[9][MT], bv[9][NT];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const float* wp = wbase + (size_t)(krow0 + t9 * Cp + ci) * d.Mpad;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { const float w = wp[wcol[mt]]; av[t9][mt] = mok[mt] ? w : 0.f; }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bool ok = cok && (((vmask[t] >> (t9 / 3)) & (vmask[t] >> (8 + t9 % 3)) & 1u) != 0);
          const int off = base[t] + ci * HW2 + toff[t9];
          const float v = xs[ok ? off : 0];  // unconditional load (no branch / vmcnt(0) per element)
          bv[t9][t] = ok ? prologue<PRO>(ia, v, (uint32_t)off) : 0.f;
        }
      }
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t9][mt], bv[t9][t], acc[mt][t], 0, 0, 0);
    }
  } else if (KS == 1) {
    constexpr int G = 8;
    bool tv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tv[t] = (vmask[t] & (vmask[t] >> 8) & 1u) != 0;
    const int toff = tap_offset(d, 0, 0);
    for (int c2 = c2_begin; c2 < Cp; c2 += G * c2_step) {
      float av[G][MT], bv[G][NT];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int ci = c2 + u * c2_step + h;
        const bool cok = ci < C;
        const bool rok = ci < Cp;
        const float* wp = wbase + (size_t)(krow0 + (rok ? ci : 0)) * d.Mpad;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { const float w = wp[wcol[mt]]; av[u][mt] = (mok[mt] && rok) ? w : 0.f; }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int off = base[t] + ci * HW2 + toff;
          const bool ok = tv[t] && cok;
          const float v = xs[ok ? off : 0];
          bv[u][t] = ok ? prologue<PRO>(ia, v, (uint32_t)off) : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][mt], bv[u][t], acc[mt][t], 0, 0, 0);
    }
  } else {
    constexpr int G = 8;
    int tap = 0;
    for (int kh = 0; kh < d.KH; ++kh) {
      for (int kw = 0; kw < d.KW; ++kw, ++tap) {
        bool tv[NT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          tv[t] = ((vmask[t] >> kh) & (vmask[t] >> (8 + kw)) & 1u) != 0;
          any |= tv[t];
        }
        if (!__any(any)) continue;  // wave-uniform: a tap no lane of this wave can use (parity / border)
        const int toff = tap_offset(d, kh, kw);
        for (int c2 = c2_begin; c2 < Cp; c2 += G * c2_step) {
          float av[G][MT], bv[G][NT];
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int ci = c2 + u * c2_step + h;
            const bool cok = ci < C, rok = ci < Cp;
            const float* wp = wbase + (size_t)(krow0 + tap * Cp + (rok ? ci : 0)) * d.Mpad;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { const float w = wp[wcol[mt]]; av[u][mt] = (mok[mt] && rok) ? w : 0.f; }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const bool ok = tv[t] && cok;
              const int off = base[t] + ci * HW2 + toff;
              const float v = xs[ok ? off : 0];
              bv[u][t] = ok ? prologue<PRO>(ia, v, (uint32_t)off) : 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < G; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int t = 0; t < NT; ++t)
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][mt], bv[u][t], acc[mt][t], 0, 0, 0);
        }
      }
    }
  }
}

template <int MT, int NT, int PRO, int KS>
__device__ __forceinline__ void k_loop(const GatherArgs& a, const TileGeo* tg, const bool* mok, int m0, int j, int h,
                                       int c2_begin, int c2_step, f32x16 (&acc)[MT][NT]) {
  const vunet_conv_desc& d = a.d;
  int base[NT];
  uint32_t vmask[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { base[t] = tg[t].base1; vmask[t] = tg[t].vmask; }
  k_sweep<MT, NT, PRO, KS>(a, a.x1, a.in1, d.C1, 0, base, vmask, mok, m0, j, h, c2_begin, c2_step, acc);
  if (d.C2 > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) base[t] = tg[t].base2;
    const int krow2 = d.KH * d.KW * ((d.C1 + 1) & ~1);
    k_sweep<MT, NT, PRO, KS>(a, a.x2, a.in2, d.C2, krow2, base, vmask, mok, m0, j, h, c2_begin, c2_step, acc);
  }
}

template <int MT, int NT, int PRO, int KS>
__global__ __launch_bounds__(256) void conv_gather_kernel(const GatherArgs a_in) {
  GatherArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const vunet_conv_desc& d = a.d;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = (d.M + 32 * MT - 1) / (32 * MT);
  const int mb = bid % mblocks, pb = bid / mblocks;
  const int m0 = mb * 32 * MT;
  const int tile0 = (pb * 4 + wave) * NT;
  if (tile0 * 32 >= a.NP) return;  // whole wave out of range (no barriers in this kernel)

  TileGeo tg[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tg[t] = tile_geometry(a, (tile0 + t) * 32 + j);

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

  k_loop<MT, NT, PRO, KS>(a, tg, mok, m0, j, h, 0, 2, acc);

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const PixGeo g = out_pixel(a, (tile0 + t) * 32 + j);
    if (!g.valid) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < d.M) store_out(a, g, m, acc[mt][t][r]);
      }
  }
}

// ---- split-K variant for small maps: one 32x32 output tile per workgroup, SW waves share K
template <int SW, int PRO, int KS>
__global__ __launch_bounds__(SW * 64) void conv_gather_splitk_kernel(const GatherArgs a_in) {
  GatherArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  extern __shared__ __attribute__((aligned(16))) float red[];  // [SW][16][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const vunet_conv_desc& d = a.d;
  const int mblocks = (d.M + 31) / 32;
  const int mb = blockIdx.x % mblocks, pb = blockIdx.x / mblocks;
  const int m0 = mb * 32;

  TileGeo tg[1];
  tg[0] = tile_geometry(a, pb * 32 + j);
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  bool mok[1];
  mok[0] = (m0 + j) < d.M;

  k_loop<1, 1, PRO, KS>(a, tg, mok, m0, j, h, 2 * wave, 2 * SW, acc);

#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[0][0][r];
  __syncthreads();
  // wave w finishes accumulator register r = w, w + SW, ... (fixed summation order over the waves)
  const PixGeo g = out_pixel(a, pb * 32 + j);
  float vmax = 0.f;
  for (int r = wave; r < 16; r += SW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) v += red[(w * 16 + r) * 64 + lane];
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (g.valid && m < d.M) vmax = fmaxf(vmax, fabsf(store_out(a, g, m, v)));
  }
  if (a.amax_out) publish_amax(a, vmax);   // |y| maxima for the consumer's fp16 scale (GatherArgs::amax_out)
}

template <int MT, int NT, int KS>
static int launch_gather(const GatherArgs& ga, int pro, hipStream_t st) {
  const int ntiles = (ga.NP + 31) / 32;
  const int pblocks = (ntiles + 4 * NT - 1) / (4 * NT);
  const int mblocks = (ga.d.M + 32 * MT - 1) / (32 * MT);
  dim3 grid((unsigned)(pblocks * mblocks)), block(256);
  switch (pro) {
    case 0: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 0, KS>), grid, block, 0, st, ga); break;
    case 1: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 1, KS>), grid, block, 0, st, ga); break;
    case 2: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 2, KS>), grid, block, 0, st, ga); break;
    default: VUNET_LAUNCH((conv_gather_kernel<MT, NT, 3, KS>), grid, block, 0, st, ga); break;
  }
  return vunet_check_launch();
}

template <int KS>
static int launch_splitk(const GatherArgs& ga, int pro, hipStream_t st) {
  constexpr int SW = 16;
  const int ntiles = (ga.NP + 31) / 32, mblocks = (ga.d.M + 31) / 32;
  dim3 grid((unsigned)(ntiles * mblocks)), block(SW * 64);
  const size_t lds = (size_t)SW * 16 * 64 * sizeof(float);
  switch (pro) {
    case 0: VUNET_LAUNCH((conv_gather_splitk_kernel<SW, 0, KS>), grid, block, lds, st, ga); break;
    case 1: VUNET_LAUNCH((conv_gather_splitk_kernel<SW, 1, KS>), grid, block, lds, st, ga); break;
    case 2: VUNET_LAUNCH((conv_gather_splitk_kernel<SW, 2, KS>), grid, block, lds, st, ga); break;
    default: VUNET_LAUNCH((conv_gather_splitk_kernel<SW, 3, KS>), grid, block, lds, st, ga); break;
  }
  return vunet_check_launch();
}

// Few output tiles and a long K: share each tile's K loop between 16 waves.  Up to 768 tiles (3 per CU) any layer
// with >= 64 input channels; up to 2048 tiles only when K is at least 576 deep (3x3 over >= 64 channels) -- there
// the one-wave-per-tile kernel is latency bound (25 x 256->128 @ 16^2: 582 us vs ~60 us split).
static bool want_splitk(long ntiles, long mtiles, int kpairs, int taps) {
  const long tiles = ntiles * mtiles;
  return (tiles <= 768 && kpairs >= 32) || (tiles <= 2048 && (long)kpairs * taps >= 288);
}

template <int KS>
static int dispatch_gather(const GatherArgs& ga_in, int pro, hipStream_t st, float* amax_out = nullptr) {
  GatherArgs ga = ga_in;
  const vunet_conv_desc& d = ga.d;
  const int ntiles = (ga.NP + 31) / 32, mtiles = (d.M + 31) / 32;
  const int kpairs = (((d.C1 + 1) >> 1) + ((d.C2 + 1) >> 1));
  if (want_splitk(ntiles, mtiles, kpairs, d.KH * d.KW)) {
    ga.amax_out = amax_out;   // the split-K kernel publishes |y| maxima (the others of this dispatcher do not)
    return launch_splitk<KS>(ga, pro, st);
  }
  if (d.M <= 32) return launch_gather<1, 4, KS>(ga, pro, st);
  return launch_gather<2, 2, KS>(ga, pro, st);
}

static bool gather_uses_splitk(const vunet_conv_desc* d) {
  const long ntiles = ((long)d->N * d->Ho * d->Wo + 31) / 32, mtiles = (d->M + 31) / 32;
  const int kpairs = ((d->C1 + 1) >> 1) + ((d->C2 + 1) >> 1);
  return want_splitk(ntiles, mtiles, kpairs, d->KH * d->KW);
}

// conv_tiled.hip: LDS-tiled kernel for the 3x3 / stride-1 layers on maps at least 32 wide
bool vunet_conv_tiled_applicable(const vunet_conv_desc* d);
int vunet_conv_tiled_pick(const vunet_conv_desc* d, int* MT, bool valu_heavy);
int vunet_conv_tiled_launch(const GatherArgs& ga, int pro, hipStream_t st);

int vunet_conv_tiled_name(const vunet_conv_desc* d, int pro, bool has_aux, char* name, int len);

// conv_thin.hip: VALU kernels for the layers with <= 4 channels on one side
int vunet_conv_thin_kind(const vunet_conv_desc* d, int pro, bool has_aux, bool has_res);
int vunet_conv_thin_launch(const GatherArgs& ga, int kind, hipStream_t st);
int vunet_conv_thin_name(const vunet_conv_desc* d, int kind, char* name, int len);

// conv_1x1.hip: streaming kernel for the 1x1 / stride-1 layers
bool vunet_conv_1x1_applicable(const vunet_conv_desc* d, int pro);
int vunet_conv_1x1_launch(const GatherArgs& ga, int pro, hipStream_t st);
int vunet_conv_1x1_name(const vunet_conv_desc* d, int pro, char* name, int len);

VUNET_ENV_FLAG(env_no_tiled, "VUNET_NO_TILED")
VUNET_ENV_FLAG(env_no_phase, "VUNET_NO_PHASE")

static int prologue_code(const vunet_conv_desc* d) {
  if (d->in_act == ACT_NONE && d->drop_p <= 0.f) return 0;
  if (d->in_act == ACT_ELU && d->drop_p <= 0.f) return 1;
  if (d->in_act == ACT_ELU) return 2;
  return 3;
}

// the streaming 1x1 kernel, unless the problem is so small that splitting K over 16 waves is the better plan
static bool use_1x1(const vunet_conv_desc* d, int pro) {
  const long ntiles = ((long)d->N * d->Ho * d->Wo + 31) / 32, mtiles = (d->M + 31) / 32;
  const int kpairs = ((d->C1 + 1) >> 1) + ((d->C2 + 1) >> 1);
  return vunet_conv_1x1_applicable(d, pro) && !want_splitk(ntiles, mtiles, kpairs, 1);
}

static bool use_tiled(const vunet_conv_desc* d, int pro) {
  int mt_unused;
  return !env_no_tiled() && vunet_conv_tiled_applicable(d) &&
         (d->mode == 0 ? pro != 4 : (pro == 0 || (pro == 4 && d->stride == 1))) && pro != 3 &&
         vunet_conv_tiled_pick(d, &mt_unused, true) > 0;
}

// Name (rocprofv3 spelling) of the kernel vunet_conv2d_gather would launch for this problem.
extern "C" int vunet_conv2d_gather_variant(const vunet_conv_desc* d, int32_t has_aux, char* name, int32_t len) {
  if (!d || !name || len < 8) return VUNET_ERR_ARG;
  const int pro = prologue_code(d);
  if (d->mode == 1 && d->stride > 1 && !env_no_phase()) {
    snprintf(name, len, "conv_gather_kernel<phase x%d>", d->stride * d->stride);
    return VUNET_OK;
  }
  if (const int thin = vunet_conv_thin_kind(d, pro, has_aux != 0, false)) {
    vunet_conv_thin_name(d, thin, name, len);
    return VUNET_OK;
  }
  if (use_1x1(d, pro)) {
    vunet_conv_1x1_name(d, pro, name, len);
    return VUNET_OK;
  }
  if (use_tiled(d, pro)) {
    vunet_conv_tiled_name(d, pro, has_aux != 0, name, len);
    return VUNET_OK;
  }
  const int KS = (d->KH == 3 && d->KW == 3) ? 3 : ((d->KH == 1 && d->KW == 1) ? 1 : 0);
  const long ntiles = ((long)d->N * d->Ho * d->Wo + 31) / 32, mtiles = (d->M + 31) / 32;
  const int kpairs = ((d->C1 + 1) >> 1) + ((d->C2 + 1) >> 1);
  if (want_splitk(ntiles, mtiles, kpairs, d->KH * d->KW)) snprintf(name, len, "conv_gather_splitk_kernel<16, %d, %d>", pro, KS);
  else if (d->M <= 32) snprintf(name, len, "conv_gather_kernel<1, 4, %d, %d>", pro, KS);
  else snprintf(name, len, "conv_gather_kernel<2, 2, %d, %d>", pro, KS);
  return VUNET_OK;
}

extern "C" int vunet_conv2d_gather(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt,
                                   const float* shift, const float* res, const float* aux, float* y,
                                   void* stream) {
  return vunet_conv2d_gather_amax(d, x1, x2, wt, shift, res, aux, y, nullptr, stream);
}

// kernels of this dispatcher whose epilogue publishes |y| maxima (store_tile16): the streaming 1x1 and the LDS-tiled one
bool vunet_conv2d_gather_publishes(const vunet_conv_desc* d, bool has_aux, bool has_res) {
  if (d->d2s || (d->mode == 1 && d->stride > 1 && !env_no_phase())) return false;
  const int pro = prologue_code(d);
  if (const int thin = vunet_conv_thin_kind(d, pro, has_aux, has_res)) return thin == 2;   // the 3-channel-input kernel publishes
  return use_1x1(d, pro) || use_tiled(d, pro) || gather_uses_splitk(d);
}

int vunet_conv2d_gather_amax(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt, const float* shift,
                             const float* res, const float* aux, float* y, float* amax_out, void* stream) {
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
  ga.wide = 0;
  ga.amax_out = nullptr;
  ga.res2 = nullptr;
  ga.amax2 = nullptr;
  ga.d = *d;
  ga.x1 = x1; ga.x2 = x2; ga.wt = wt; ga.shift = shift; ga.res = res; ga.aux = aux; ga.y = y;
  ga.NP = d->N * d->Ho * d->Wo;
  ga.HoWo = d->Ho * d->Wo;
  ga.HsWs = d->Hs * d->Ws;
  ga.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  ga.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  ga.auxa = make_inact(d->aux_act, d->aux_slope, d->aux_drop_p, d->aux_drop_seed);
  const int pro = prologue_code(d);
  ga.ph = ga.pw = -1;
  ga.subW = ga.subHW = 0;
  ga.mask = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (d->mode == 1 && d->stride > 1 && !env_no_phase()) {
    // strided data gradient: one launch per output parity, each visiting only its own taps
    const int s = d->stride;
    for (int ph = 0; ph < s; ++ph)
      for (int pw = 0; pw < s; ++pw) {
        GatherArgs gp = ga;
        gp.ph = ph;
        gp.pw = pw;
        const int subH = (d->Ho - ph + s - 1) / s;
        gp.subW = (d->Wo - pw + s - 1) / s;
        gp.subHW = subH * gp.subW;
        gp.NP = d->N * gp.subHW;
        if (gp.NP <= 0) continue;
        const int rc = dispatch_gather<0>(gp, pro, st);
        if (rc != VUNET_OK) return rc;
      }
    return VUNET_OK;
  }
  if (const int thin = vunet_conv_thin_kind(d, pro, aux != nullptr, res != nullptr)) {
    if (thin == 2) ga.amax_out = amax_out;
    return vunet_conv_thin_launch(ga, thin, st);
  }
  if (!d->d2s) ga.amax_out = amax_out;   // read by the two kernels below only
  if (use_1x1(d, pro)) {
    // 16-byte epilogue (conv_common.h: store_tile_side4): a 32-pixel tile is 8 aligned groups of 4 consecutive pixels
    const uintptr_t al = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(aux);
    ga.wide = (al & 15) == 0 && !d->d2s && (d->Ho * d->Wo) % 32 == 0;
#ifdef VUNET_AB_NO_WIDE_1X1   // (A/B baseline, tools/ab_build.sh)
    ga.wide = 0;
#endif
    return vunet_conv_1x1_launch(ga, pro, st);
  }
  if (use_tiled(d, pro)) return vunet_conv_tiled_launch(ga, pro, st);
  float* const amax_keep = ga.amax_out;
  ga.amax_out = nullptr;
  ga.res2 = nullptr;
  ga.amax2 = nullptr;
  if (d->KH == 3 && d->KW == 3) return dispatch_gather<3>(ga, pro, st, amax_keep);
  if (d->KH == 1 && d->KW == 1) return dispatch_gather<1>(ga, pro, st, amax_keep);
  return dispatch_gather<0>(ga, pro, st, amax_keep);
}


// Data gradient through a layer whose forward epilogue was ReLU:  dx = dgrad(dy * [y > 0]) (+ res), the mask applied
// while dy is staged -- no separate pass over dy / y.  Only the LDS-tiled kernel implements it; every other geometry
// returns VUNET_ERR_UNSUPPORTED and the caller runs vunet_act_bwd_from_out + vunet_conv2d_gather(mode 1) instead.
extern "C" int vunet_conv2d_dgrad_relu(const vunet_conv_desc* d, const float* dy, const float* y, const float* wt,
                                       const float* res, float* dx, void* stream) {
  if (!d || !dy || !y || !wt || !dx || d->mode != 1 || d->C2 != 0 || d->Mpad % 32 != 0) return VUNET_ERR_ARG;
  if (d->drop_p > 0.f || d->in_act != ACT_NONE || d->aux_act != ACT_NONE || d->aux_drop_p > 0.f || !use_tiled(d, 4) ||
      vunet_conv_thin_kind(d, 0, false, res != nullptr) != 0)   // 3-channel side: the two-pass route ends in the VALU kernel
    return VUNET_ERR_UNSUPPORTED;
  GatherArgs ga;
  ga.wide = 0;
  ga.amax_out = nullptr;
  ga.res2 = nullptr;
  ga.amax2 = nullptr;
  ga.d = *d;
  ga.x1 = dy; ga.x2 = nullptr; ga.wt = wt; ga.shift = nullptr; ga.res = res; ga.aux = nullptr; ga.y = dx;
  ga.NP = d->N * d->Ho * d->Wo;
  ga.HoWo = d->Ho * d->Wo;
  ga.HsWs = d->Hs * d->Ws;
  ga.in1 = ga.in2 = ga.auxa = make_inact(ACT_NONE, 0.f, 0.f, 0u);
  ga.ph = ga.pw = -1;
  ga.subW = ga.subHW = 0;
  ga.mask = y;
  return vunet_conv_tiled_launch(ga, 4, (hipStream_t)stream);
}
