// Weight gradient of the 3x3 / stride 1 / pad 1 convolution on fp32 MFMA -- LDS halo-tile kernel.
//
//   dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap]          (K = pixels)
//
// A workgroup owns MTW*32 output channels x 32 input channels x 9 taps and walks rectangular
// TH x 32 pixel tiles (split-K over tiles).  Per tile it stages into LDS, once per element,
//   fxL[32 ci][TH+2][34] (+1 dword per channel: odd pitch, conflict-free column reads)
//       = prologue-activated input with halo (ELU / dropout hash applied here, not per tap)
//   dyL[MTW*32 co][TH*32] (+1 dword per row)
// and every wave runs v_mfma_f32_32x32x2_f32 with A = dy (lane: co, k-half: pixel parity) and
// B = f(x) shifted by the tap (lane: ci) into 9 accumulator tiles.  The 4 waves split the
// (m-tile, pixel-row) space; waves that share an m-tile write separate slabs, which
// vunet_weightnorm_bwd sums in a fixed order (deterministic, no atomics).
#include "common.h"

struct WgradTiledArgs {
  vunet_wgrad_desc d;
  const float* x1;
  const float* x2;
  const float* dy;
  float* slabs;
  float* dshift;
  int HW, Ctot, Coutp, tiles_per_img_w, tiles_per_img, ntiles, tps, S;
  InAct in1, in2;
};

// S = 2 (the stride-2 Downsample convs): tiles walk the output map, the staged input tile is (2*TH+1) x 65.
template <int MTW, int TH, int KS, int S = 1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tiled_kernel(const WgradTiledArgs a_in) {
  WgradTiledArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  constexpr int WP = 4 / MTW;        // pixel parts (waves sharing one m-tile)
  constexpr int RW = TH / WP;        // tile rows per wave
  constexpr int HALO = KS / 2, NTAP = KS * KS;
  constexpr int IH = (TH - 1) * S + 1 + 2 * HALO, IW = 31 * S + 1 + 2 * HALO;
  static_assert(RW >= 1, "tile too small for the wave split");
  constexpr int CS = IH * IW + 1;    // fx channel pitch (odd)
  constexpr int PS = TH * 32 + 1;    // dy row pitch (odd)
  constexpr int MB = 32 * MTW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* fxL = smem;                 // [32][CS]
  float* dyL = smem + 32 * CS;       // [MB][PS]

  const vunet_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave / WP, pw = wave % WP;
  const int split = blockIdx.x;
  const int ci0 = blockIdx.y * 32;
  const int co0 = blockIdx.z * MB;
  const int H = d.Hs, W = d.Ws, Ho = d.Ho, Wo = d.Wo;

  // source of this input-channel block (C1 % 32 == 0 is a precondition of this kernel)
  const bool second = ci0 >= d.C1;
  const float* __restrict__ xs = second ? a.x2 : a.x1;
  const int Cs = second ? d.C2 : d.C1;
  const int cbase = second ? ci0 - d.C1 : ci0;
  const InAct ia = second ? a.in2 : a.in1;

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dsum = 0.f;

  const int t_begin = split * a.tps;
  const int t_end = min(t_begin + a.tps, a.ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n = tile / a.tiles_per_img;
    const int tr = tile - n * a.tiles_per_img;
    const int ty = tr / a.tiles_per_img_w, tx = tr - ty * a.tiles_per_img_w;
    const int row0 = ty * TH, col0 = tx * 32;

    __syncthreads();  // previous tile's LDS reads are done
    // ---- stage f(x) with halo: 32 x IH x 34 elements, in register batches of BF
    {
      constexpr int E = 32 * IH * IW;
      constexpr int NI = (E + 255) / 256;
      constexpr int BF = 9;
#pragma unroll 1
      for (int i0 = 0; i0 < NI; i0 += BF) {
        float v[BF];
        int go[BF];
#pragma unroll
        for (int u = 0; u < BF; ++u) {
          const int e = tid + 256 * (i0 + u);
          const int c = e / (IH * IW);
          const int rem = e - c * (IH * IW);
          const int r = rem / IW, col = rem - r * IW;
          const int ih = row0 * S - HALO + r, iw = col0 * S - HALO + col;
          const bool ok = e < E && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && cbase + c < Cs;
          go[u] = ok ? ((n * Cs + cbase + c) * H + ih) * W + iw : -1;
          v[u] = xs[ok ? go[u] : 0];  // unconditional load; invalid elements are zeroed below
        }
#pragma unroll
        for (int u = 0; u < BF; ++u) {
          const int e = tid + 256 * (i0 + u);
          if (e < E) {
            const int c = e / (IH * IW);
            fxL[e + c] = go[u] >= 0 ? apply_in_act(ia, v[u], (uint32_t)go[u]) : 0.f;  // c*CS + rem == e + c
          }
        }
      }
    }
    // ---- stage dy: MB x TH x 32 elements, in register batches of 8
    {
      constexpr int E = MB * TH * 32;
      constexpr int NI = E / 256;
#pragma unroll 1
      for (int i0 = 0; i0 < NI; i0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + 256 * (i0 + u);
          const int co = e / (TH * 32), p = e % (TH * 32);
          const int cog = co0 + co;
          const bool cv = cog < d.Cout;
          const float t = a.dy[((size_t)(n * d.Cout + (cv ? cog : 0)) * Ho + row0 + (p >> 5)) * Wo + col0 + (p & 31)];
          v[u] = cv ? t : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + 256 * (i0 + u);
          dyL[(e / (TH * 32)) * PS + e % (TH * 32)] = v[u];
        }
      }
    }
    __syncthreads();

    // ---- MFMA: this wave's RW rows x 32 columns = RW*16 k-steps, 9 taps each
    const float* aP = dyL + (wm * 32 + j) * PS + pw * RW * 32 + h;
    const float* bP = fxL + j * CS + pw * RW * S * IW + h * S;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float av = aP[rr * 32 + 2 * s];
        dsum += av;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const float bv = bP[(rr * S + t / KS) * IW + 2 * s * S + t % KS];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- fold the WP pixel-part waves of each m-tile into the pw == 0 wave through LDS (fixed order),
  //      one tap at a time (16 KiB), so that one slab per split leaves the workgroup
  if (WP > 1) {
    float* red = smem;  // [4 waves][16][64]
    float ds_other = 0.f;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      __syncthreads();
      if (pw != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (pw == 0) {
#pragma unroll
        for (int o = 1; o < WP; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[((wave + o) * 16 + r) * 64 + lane];
      }
    }
    __syncthreads();
    if (pw != 0) red[wave * 64 + lane] = dsum;
    __syncthreads();
    if (pw == 0)
      for (int o = 1; o < WP; ++o) ds_other += red[(wave + o) * 64 + lane];
    dsum += ds_other;
  }

  // ---- partial slab of this split:  [split][Coutp][NTAP*Ctot], k order (tap, ci)
  const size_t KT = (size_t)NTAP * a.Ctot;
  float* slab = a.slabs + (size_t)split * a.Coutp * KT;
  const int ci = ci0 + j;
  if (pw == 0) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < d.Cout && ci < a.Ctot) slab[(size_t)co * KT + (size_t)t * a.Ctot + ci] = acc[t][r];
      }
    }
    if (blockIdx.y == 0) {
      const float tot = dsum + __shfl_xor(dsum, 32, 64);
      const int co = co0 + wm * 32 + j;
      if (h == 0 && co < a.Coutp) a.dshift[(size_t)split * a.Coutp + co] = co < d.Cout ? tot : 0.f;
    }
  }
}

// ---- host side ----------------------------------------------------------------------------
bool vunet_wgrad_tiled_applicable(const vunet_wgrad_desc* d) {
  const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
  const bool chans = ((d->C1 % 32 == 0 && d->C2 % 32 == 0) || (d->C2 == 0 && d->C1 < 32)) &&
                     (d->Cout % 32 == 0 || d->Cout < 32);
  if (d->stride == 2)  // Downsample convs: 2 x 32 output tiles, two m-tiles per workgroup
    return k3 && chans && d->Cout >= 64 && d->Hs == 2 * d->Ho && d->Ws == 2 * d->Wo && d->Wo % 32 == 0 && d->Ho % 2 == 0;
  return (k3 || k1) && chans && d->stride == 1 && d->Ws % 32 == 0 && d->Hs % 4 == 0 && d->Ho == d->Hs && d->Wo == d->Ws;
}

static void tiled_geometry(const vunet_wgrad_desc* d, int& MTW, int& TH, int& ntiles, int& ciblocks, int& coblocks) {
  MTW = d->Cout >= 64 ? 2 : 1;
  TH = d->stride == 2 ? 2 : 4;
  ntiles = d->N * (d->Ho / TH) * (d->Wo / 32);
  ciblocks = (d->C1 + d->C2 + 31) / 32;
  coblocks = (d->Cout + 32 * MTW - 1) / (32 * MTW);
}

int vunet_wgrad_tiled_nslabs(const vunet_wgrad_desc* d) {
  int MTW, TH, ntiles, ciblocks, coblocks;
  tiled_geometry(d, MTW, TH, ntiles, ciblocks, coblocks);
  int S = 1024 / (ciblocks * coblocks);  // ~4 workgroups per CU over the whole grid
  if (S > ntiles / 2) S = ntiles / 2;
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  return S;
}

template <int MTW, int TH, int KS, int ST>
static int launch_wgrad_tiled(const WgradTiledArgs& a, dim3 grid, hipStream_t st) {
  constexpr int HALO = KS / 2;
  constexpr int IH = (TH - 1) * ST + 1 + 2 * HALO, IW = 31 * ST + 1 + 2 * HALO;
  size_t lds = (size_t)(32 * (IH * IW + 1) + 32 * MTW * (TH * 32 + 1)) * sizeof(float);
  if (lds < 16384 + 4096) lds = 16384 + 4096;  // the in-workgroup fold needs 16 KiB + 1 KiB
  VUNET_LAUNCH((conv_wgrad_tiled_kernel<MTW, TH, KS, ST>), grid, dim3(256), lds, st, a);
  return vunet_check_launch();
}

int vunet_wgrad_tiled_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                             float* slabs, float* dshift, hipStream_t st) {
  WgradTiledArgs a;
  a.d = *d;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.slabs = slabs; a.dshift = dshift;
  int MTW, TH, ciblocks, coblocks;
  tiled_geometry(d, MTW, TH, a.ntiles, ciblocks, coblocks);
  a.S = d->nsplit;
  a.HW = d->Hs * d->Ws;
  a.Ctot = d->C1 + d->C2;
  a.Coutp = (d->Cout + 31) / 32 * 32;
  a.tiles_per_img_w = d->Wo / 32;
  a.tiles_per_img = (d->Ho / TH) * a.tiles_per_img_w;
  a.tps = (a.ntiles + a.S - 1) / a.S;
  a.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  a.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  dim3 grid(a.S, ciblocks, coblocks);
  if (d->stride == 2) return launch_wgrad_tiled<2, 2, 3, 2>(a, grid, st);
  if (d->KH == 3) return MTW == 1 ? launch_wgrad_tiled<1, 4, 3, 1>(a, grid, st) : launch_wgrad_tiled<2, 4, 3, 1>(a, grid, st);
  return MTW == 1 ? launch_wgrad_tiled<1, 4, 1, 1>(a, grid, st) : launch_wgrad_tiled<2, 4, 1, 1>(a, grid, st);
}
