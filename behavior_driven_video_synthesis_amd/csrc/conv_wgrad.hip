// Weight-gradient of the fused convolution on fp32 MFMA (generic geometry).
//
//   dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap]        (K = pixels)
//
// A workgroup owns a (WM*32 output channels) x (all taps x 32 input channels) block of dW and a
// contiguous range of 32-pixel chunks (split-K over pixels).  Per chunk it stages
//   dyL[co][33]      -- the dy rows of its output channels (coalesced along pixels)
//   fxL[tap][ci][33] -- the im2col'ed, prologue-activated input (ELU / dropout applied here)
// into LDS (row pitch 33 dwords: conflict-free transposed reads), then every wave runs
// v_mfma_f32_32x32x2_f32 with A = dy (lane: co, k-half: pixel parity) and B = f(x) (lane: ci).
// Partial results go to per-split slabs (deterministic; summed by vunet_weightnorm_bwd).
// The per-channel sum of dy (gradient of the folded shift) falls out of the A fragments.
#include "common.h"

struct WgradArgs {
  vunet_wgrad_desc d;
  const float* x1;
  const float* x2;
  const float* dy;
  float* slabs;
  float* dshift;
  int NP, HoWo, HsWs, T, Ctot, Coutp, nchunks, cps;
  InAct in1, in2;
};

template <int WM, int WN, int NTW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a_in) {
  WgradArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MB = 32 * WM;
  constexpr int PITCH = 33;
  float* dyL = smem;                 // [MB][33]
  float* fxL = smem + MB * PITCH;    // [T*32][33]

  const vunet_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int q = tid & 31, rgrp = tid >> 5;

  const int split = blockIdx.x;
  const int ci0 = blockIdx.y * 32;
  const int co0 = blockIdx.z * MB;
  const int T = a.T;

  f32x16 acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float dsum = 0.f;

  const int c_begin = split * a.cps;
  const int c_end = min(c_begin + a.cps, a.nchunks);

  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    // ---- this thread's pixel
    const int P = chunk * 32 + q;
    const bool pv = P < a.NP;
    const int Pc = pv ? P : 0;
    const int n = Pc / a.HoWo;
    const int rem = Pc - n * a.HoWo;
    const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
    const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;

    __syncthreads();  // previous chunk's MFMA reads are done
    // ---- stage dy rows
#pragma unroll 4
    for (int i = 0; i < MB / 8; ++i) {
      const int col = rgrp + 8 * i, co = co0 + col;
      const bool ok = pv && co < d.Cout;
      const float v = a.dy[ok ? (size_t)(n * d.Cout + co) * a.HoWo + rem : 0];
      dyL[col * PITCH + q] = ok ? v : 0.f;
    }
    // ---- stage im2col'ed activated input rows
    int kh = 0, kw = 0;
    for (int tap = 0; tap < T; ++tap) {
      const int ih = ih0 + kh, iw = iw0 + kw;
      const bool tv = pv && (unsigned)ih < (unsigned)d.Hs && (unsigned)iw < (unsigned)d.Ws;
      const int sp = ih * d.Ws + iw;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int cl = rgrp + 8 * ii, ci = ci0 + cl;
        const bool ok = tv && ci < a.Ctot;
        const bool first = ci < d.C1;
        const int off = first ? (n * d.C1 + ci) * a.HsWs + sp : (n * d.C2 + (ci - d.C1)) * a.HsWs + sp;
        const float* __restrict__ xs = (first || !a.x2) ? a.x1 : a.x2;
        const float raw = xs[ok ? off : 0];  // unconditional load
        const float v = apply_in_act(first ? a.in1 : a.in2, raw, (uint32_t)off);
        fxL[(tap * 32 + cl) * PITCH + q] = ok ? v : 0.f;
      }
      if (++kw == d.KW) { kw = 0; ++kh; }
    }
    __syncthreads();

    // ---- MFMA over the 32 pixels of the chunk (16 k-steps of 2)
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const float av = dyL[(wm * 32 + j) * PITCH + 2 * ks + h];
      dsum += av;
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int nt = wn + WN * i;
        if (nt < T) {
          const float bv = fxL[(nt * 32 + j) * PITCH + 2 * ks + h];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
        }
      }
    }
  }

  // ---- write the partial slab  [split][Coutp][T*Ctot]
  const size_t KT = (size_t)T * a.Ctot;
  float* slab = a.slabs + (size_t)split * a.Coutp * KT;
  const int ci = ci0 + j;
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int nt = wn + WN * i;
    if (nt >= T || ci >= a.Ctot) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < d.Cout) slab[(size_t)co * KT + (size_t)nt * a.Ctot + ci] = acc[i][r];
    }
  }
  if (wn == 0 && blockIdx.y == 0) {
    const float tot = dsum + __shfl_xor(dsum, 32, 64);
    const int co = co0 + wm * 32 + j;
    if (h == 0 && co < a.Coutp) a.dshift[(size_t)split * a.Coutp + co] = co < d.Cout ? tot : 0.f;
  }
}

static void wgrad_geometry(const vunet_wgrad_desc* d, int& T, int& Ctot, int& Coutp, int& nchunks, int& WM) {
  T = d->KH * d->KW;
  Ctot = d->C1 + d->C2;
  Coutp = (d->Cout + 31) / 32 * 32;
  const int64_t NP = (int64_t)d->N * d->Ho * d->Wo;
  nchunks = (int)((NP + 31) / 32);
  WM = d->Cout > 64 ? 4 : (d->Cout > 32 ? 2 : 1);
  if (T > 9 && WM == 4) WM = 2;
}

// conv_wgrad_thin.hip: 3x3 layers with <= 4 output channels at full resolution (out_conv): fp32 FMA kernel
bool vunet_wgrad_thin_applicable(const vunet_wgrad_desc* d);
int vunet_wgrad_thin_nslabs(const vunet_wgrad_desc* d);
int vunet_wgrad_thin_name(const vunet_wgrad_desc* d, char* name, int len);
int vunet_wgrad_thin_launch(const vunet_wgrad_desc* d, const float* x1, const float* dy, float* slabs, float* dshift,
                            hipStream_t st);

// conv_wgrad_tiled.hip: LDS halo-tile kernel for the 3x3 / stride-1 layers that carry the FLOPs
bool vunet_wgrad_tiled_applicable(const vunet_wgrad_desc* d);
int vunet_wgrad_tiled_nslabs(const vunet_wgrad_desc* d);
int vunet_wgrad_tiled_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                             float* slabs, float* dshift, hipStream_t st);

// conv_wgrad_x6.hip: the same layers on the bf16 matrix cores (fp32-accurate split-bf16 operands)
bool vunet_wgrad_x6_applicable(const vunet_wgrad_desc* d);
int vunet_wgrad_x6_nslabs(const vunet_wgrad_desc* d);
int vunet_wgrad_x6_name(const vunet_wgrad_desc* d, char* name, int len);
int vunet_wgrad_x6_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs,
                          float* dshift, hipStream_t st);
// conv_wgrad_h2.hip: the same layers with two scaled fp16 terms / three products (d->flags bit 1)
int vunet_wgrad_h2_name(const vunet_wgrad_desc* d, char* name, int len);
int vunet_wgrad_h2_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs,
                          float* dshift, const float* amax_x, const float* amax_x2, const float* amax_dy, hipStream_t st);

// conv_wgrad_direct.hip: small maps, stride 2, 1x1 on the fp16 matrix cores (d->flags bit 1), operands straight from global
bool vunet_wgrad_direct_applicable(const vunet_wgrad_desc* d);
int vunet_wgrad_direct_nslabs(const vunet_wgrad_desc* d);
int vunet_wgrad_direct_name(const vunet_wgrad_desc* d, char* name, int len);
int vunet_wgrad_direct_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs,
                              float* dshift, const float* amax_x, const float* amax_x2, const float* amax_dy, hipStream_t st);

int vunet_wgrad_direct_form(const vunet_wgrad_desc* d);
int vunet_wgrad_direct_launch_multi(const vunet_wgrad_item* items, const int* idx, int cnt, hipStream_t st);

// layers whose weight gradient runs on the direct kernel and is a short launch: batched (vunet_conv2d_wgrad_multi)
extern "C" int vunet_conv2d_wgrad_batchable(const vunet_wgrad_desc* d) {
  if (!d || vunet_wgrad_x6_applicable(d) || !vunet_wgrad_direct_applicable(d)) return 0;
  // (A/B: VUNET_WGRAD_BATCH_PIX moves the bound between "latency" and "work")
  static const int64_t bound = [] { const char* e = getenv("VUNET_WGRAD_BATCH_PIX"); return e ? atoll(e) : 16384ll; }();
  return (int64_t)d->N * d->Ho * d->Wo <= bound ? 1 : 0;
}

extern "C" int vunet_conv2d_wgrad_multi(const vunet_wgrad_item* items, int32_t n, void* stream) {
  if (n < 0 || (n > 0 && !items)) return VUNET_ERR_ARG;
  constexpr int MAXN = 256;
  if (n > MAXN) return VUNET_ERR_ARG;
  int form[MAXN], idx[MAXN];
  for (int i = 0; i < n; ++i) {
    if (!vunet_conv2d_wgrad_batchable(&items[i].d) || items[i].d.nsplit < 1) return VUNET_ERR_UNSUPPORTED;
    form[i] = vunet_wgrad_direct_form(&items[i].d);
  }
  for (int i = 0; i < n; ++i) {
    if (form[i] < 0) continue;           // already launched with an earlier item of its form
    int cnt = 0;
    const int f = form[i];
    for (int k = i; k < n; ++k)
      if (form[k] == f) { idx[cnt++] = k; form[k] = -1; }
    const int rc = vunet_wgrad_direct_launch_multi(items, idx, cnt, (hipStream_t)stream);
    if (rc != VUNET_OK) return rc;
  }
  return VUNET_OK;
}

extern "C" int vunet_conv2d_wgrad_wants_split(const vunet_wgrad_desc* d) {
  return d && (vunet_wgrad_x6_applicable(d) || vunet_wgrad_direct_applicable(d)) ? 1 : 0;
}

extern "C" int vunet_conv2d_wgrad_variant(const vunet_wgrad_desc* d, char* name, int32_t len) {
  if (!d || !name || len < 8) return VUNET_ERR_ARG;
  if (vunet_wgrad_x6_applicable(d)) {
    if (d->flags & 2) vunet_wgrad_h2_name(d, name, len);
    else vunet_wgrad_x6_name(d, name, len);
    return VUNET_OK;
  }
  if (vunet_wgrad_direct_applicable(d)) {
    vunet_wgrad_direct_name(d, name, len);
    return VUNET_OK;
  }
  if (vunet_wgrad_thin_applicable(d)) {
    vunet_wgrad_thin_name(d, name, len);
    return VUNET_OK;
  }
  if (vunet_wgrad_tiled_applicable(d)) {
    if (d->stride == 2) snprintf(name, len, "conv_wgrad_tiled_kernel<2, 2, 3, 2>");
    else snprintf(name, len, "conv_wgrad_tiled_kernel<%d, 4, %d, 1>", d->Cout >= 64 ? 2 : 1, d->KH);
    return VUNET_OK;
  }
  int T, Ctot, Coutp, nchunks, WM;
  wgrad_geometry(d, T, Ctot, Coutp, nchunks, WM);
  const int WN = 4 / WM;
  const int NTW = T == 1 ? 1 : (T <= 9 ? (WM == 4 ? 9 : (WM == 2 ? 5 : 3)) : (WM == 2 ? 8 : 4));
  snprintf(name, len, "conv_wgrad_kernel<%d, %d, %d>", WM, WN, NTW);
  return VUNET_OK;
}

extern "C" int vunet_conv2d_wgrad_nsplit(const vunet_wgrad_desc* d) {
  if (!d) return VUNET_ERR_ARG;
  if (vunet_wgrad_x6_applicable(d)) return vunet_wgrad_x6_nslabs(d);
  if (vunet_wgrad_direct_applicable(d)) return vunet_wgrad_direct_nslabs(d);
  if (vunet_wgrad_thin_applicable(d)) return vunet_wgrad_thin_nslabs(d);
  if (vunet_wgrad_tiled_applicable(d)) return vunet_wgrad_tiled_nslabs(d);
  int T, Ctot, Coutp, nchunks, WM;
  wgrad_geometry(d, T, Ctot, Coutp, nchunks, WM);
  const int ciblocks = (Ctot + 31) / 32, coblocks = (d->Cout + 32 * WM - 1) / (32 * WM);
  int S = 768 / (ciblocks * coblocks);
  if (S > nchunks) S = nchunks;  // small maps: one 32-pixel chunk per workgroup rather than a serial chunk loop
  // (capping S to bound the slab volume was measured slower: the serial chunk loop costs more than the reduce)
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  return S;
}

template <int WM, int WN, int NTW>
static int launch_wgrad(const WgradArgs& wa, hipStream_t st) {
  const int ciblocks = (wa.Ctot + 31) / 32, coblocks = (wa.d.Cout + 32 * WM - 1) / (32 * WM);
  dim3 grid(wa.d.nsplit, ciblocks, coblocks), block(256);
  const size_t lds = (size_t)(32 * WM + wa.T * 32) * 33 * sizeof(float);
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)conv_wgrad_kernel<WM, WN, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
  VUNET_LAUNCH((conv_wgrad_kernel<WM, WN, NTW>), grid, block, lds, st, wa);
  return vunet_check_launch();
}

extern "C" int vunet_conv2d_wgrad_a2(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                                     float* slabs, float* dshift, const float* amax_x, const float* amax_x2,
                                     const float* amax_dy, void* stream);
extern "C" int vunet_conv2d_wgrad(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                                  float* slabs, float* dshift, const float* amax_x, const float* amax_dy, void* stream) {
  return vunet_conv2d_wgrad_a2(d, x1, x2, dy, slabs, dshift, amax_x, nullptr, amax_dy, stream);
}
extern "C" int vunet_conv2d_wgrad_a2(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                                     float* slabs, float* dshift, const float* amax_x, const float* amax_x2,
                                     const float* amax_dy, void* stream) {
  if (!d || !x1 || !dy || !slabs || !dshift) return VUNET_ERR_ARG;
  if (d->C2 > 0 && !x2) return VUNET_ERR_ARG;
  if (d->nsplit < 1 || d->KH < 1 || d->KW < 1) return VUNET_ERR_ARG;
  const int64_t in_elems = (int64_t)d->N * (d->C1 > d->C2 ? d->C1 : d->C2) * d->Hs * d->Ws;
  const int64_t out_elems = (int64_t)d->N * d->Cout * d->Ho * d->Wo;
  if (in_elems >= (1ll << 31) || out_elems >= (1ll << 31)) return VUNET_ERR_UNSUPPORTED;
  if (vunet_wgrad_x6_applicable(d)) {
    if (d->flags & 2) {
      if (!amax_x || !amax_dy) return VUNET_ERR_ARG;
      return vunet_wgrad_h2_launch(d, x1, x2, dy, slabs, dshift, amax_x, amax_x2, amax_dy, (hipStream_t)stream);
    }
    return vunet_wgrad_x6_launch(d, x1, x2, dy, slabs, dshift, (hipStream_t)stream);
  }
  if (vunet_wgrad_direct_applicable(d)) {
    if (!amax_x || !amax_dy) return VUNET_ERR_ARG;
    return vunet_wgrad_direct_launch(d, x1, x2, dy, slabs, dshift, amax_x, amax_x2, amax_dy, (hipStream_t)stream);
  }
  if (vunet_wgrad_thin_applicable(d)) return vunet_wgrad_thin_launch(d, x1, dy, slabs, dshift, (hipStream_t)stream);
  if (vunet_wgrad_tiled_applicable(d)) return vunet_wgrad_tiled_launch(d, x1, x2, dy, slabs, dshift, (hipStream_t)stream);
  WgradArgs wa;
  wa.d = *d;
  wa.x1 = x1; wa.x2 = x2; wa.dy = dy; wa.slabs = slabs; wa.dshift = dshift;
  int WM;
  wgrad_geometry(d, wa.T, wa.Ctot, wa.Coutp, wa.nchunks, WM);
  wa.NP = d->N * d->Ho * d->Wo;
  wa.HoWo = d->Ho * d->Wo;
  wa.HsWs = d->Hs * d->Ws;
  wa.cps = (wa.nchunks + d->nsplit - 1) / d->nsplit;
  wa.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  wa.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  hipStream_t st = (hipStream_t)stream;
  const int T = wa.T;
  if (T > 16) return VUNET_ERR_UNSUPPORTED;
  if (T == 1) {
    if (WM == 4) return launch_wgrad<4, 1, 1>(wa, st);
    if (WM == 2) return launch_wgrad<2, 2, 1>(wa, st);
    return launch_wgrad<1, 4, 1>(wa, st);
  }
  if (T <= 9) {
    if (WM == 4) return launch_wgrad<4, 1, 9>(wa, st);
    if (WM == 2) return launch_wgrad<2, 2, 5>(wa, st);
    return launch_wgrad<1, 4, 3>(wa, st);
  }
  if (WM == 2) return launch_wgrad<2, 2, 8>(wa, st);
  return launch_wgrad<1, 4, 4>(wa, st);
}
