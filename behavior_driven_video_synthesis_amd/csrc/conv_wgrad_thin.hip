// Weight gradient of a 3x3 / stride 1 / pad 1 layer with at most 4 OUTPUT channels (dd.out_conv, models/vunets.py:281:
// 32 -> 3 at full resolution):   dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap].
//
// A 32-row MFMA tile would multiply >= 87 % zeros here, and the kernels that did so spent their time on the operand
// gathers, not the matrix pipe (conv_wgrad_tiled_kernel<1,4,3,1>: 470 us for 1.8 GFLOP at bs 16).  This is a plain fp32
// FMA kernel instead: a workgroup stages an (8 + 2) x (32 + 2) pixel tile of all input channels (prologue applied once
// per element) and the matching dy tile in LDS, every thread owns a (kernel row, ci) item and slides along the tile's rows
// -- one LDS read of f(x) per pixel (lanes = consecutive ci: the channel pitch is odd, conflict-free), one broadcast read
// of the 4 dy values, 3 taps x Cout FMAs -- and accumulates over the tiles of its split.  Partial sums go to the slabs
// vunet_weightnorm_bwd* reduces ([split][Coutp][T * Ctot], k order (tap, ci)); the per-channel sums of dy ride along.
#include "common.h"

struct WgradThinArgs {
  vunet_wgrad_desc d;
  const float* x1;
  const float* dy;
  float* slabs;
  float* dshift;
  int Coutp, tiles_w, tiles_per_img, ntiles;
  InAct in1;
};

// Work items (r03 rewrite).  The first form gave a thread (tap, ci) items and, per pixel, one LDS read of f(x), one
// broadcast float4 of dy and 4 FMAs -- bound by the LDS return path (a broadcast read still delivers 1 KiB to the wave),
// 193 us at bs 16.  Now a thread owns (kernel ROW kh, ci): it slides along the tile's rows keeping the last two f(x) values,
// so one new f(x) read and one dy read feed the THREE taps of the row -- 12 FMAs per pixel.  3 C items use at most 192
// threads; G = 256 / (3 C) (a power of two, <= 8) groups of threads split the tile's 8 rows and are summed through LDS at
// the end (fixed order).
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WgradThinArgs a_in) {
  WgradThinArgs a = a_in;
  inact_resolve(a.in1);
  constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 3;   // row pitch 35
  const vunet_wgrad_desc& d = a.d;
  const int C = d.C1, H = d.Hs, W = d.Ws, HW = H * W;
  const int CP = IH * IW + 1;                                  // channel pitch 351: odd -> lanes = ci hit 32 distinct banks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const xT = smem;                                      // [C][IH][IW] (+1 per channel)
  float4* const dyT = reinterpret_cast<float4*>(smem + ((C * CP + 3) & ~3));   // [TH*TW] (co 0..3)
  const int tid = threadIdx.x;
  const int nitems = 3 * C;                                    // (kh, ci)
  int G = 1;
  while (G < 8 && 2 * G * nitems <= 256) G *= 2;               // thread groups; each takes TH / G rows of every tile
  const int RG = TH / G;
  const int grp = tid / nitems, it = tid - grp * nitems;       // consecutive lanes: consecutive ci
  const bool worker = grp < G;
  const int kh = it / C, ci = it - kh * C;

  float acc[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[kw][c] = 0.f;
  float dpart[4] = {0.f, 0.f, 0.f, 0.f};   // this thread's share of sum_px dy[co]
  const int sg = tid / (TW + 2), scol = tid - sg * (TW + 2);   // staging group (0..6 used) and halo-tile column

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / a.tiles_per_img, tr = tile - n * a.tiles_per_img;
    const int ty = tr / a.tiles_w, tx = tr - ty * a.tiles_w;
    const int row0 = ty * TH, col0 = tx * TW;
    __syncthreads();   // the previous tile's reads are done
    // ---- stage f(x): [C][10][34]; seven groups of 34 threads each take one (channel, row) line per pass (coalesced)
    if (sg < 7) {
      const int iw = col0 - 1 + scol;
      const bool cok = (unsigned)iw < (unsigned)W;
      // eight lines per pass: all eight loads are issued before the first is used (a load -> convert -> store loop
      // pays a memory round trip per line: 46 of them per tile)
      for (int p0 = sg; p0 < C * IH; p0 += 7 * 8) {
        float v[8];
        uint32_t idx[8];
        bool ok[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int p = p0 + 7 * e;
          const int c = p / IH, r = p - c * IH;
          const int ih = row0 - 1 + r;
          ok[e] = p < C * IH && cok && (unsigned)ih < (unsigned)H;
          idx[e] = ok[e] ? (uint32_t)((n * C + c) * HW + ih * W + iw) : 0u;
          v[e] = a.x1[idx[e]];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int p = p0 + 7 * e;
          if (p < C * IH) {
            const int c = p / IH, r = p - c * IH;
            xT[c * CP + r * IW + scol] = ok[e] ? apply_in_act(a.in1, v[e], idx[e]) : 0.f;
          }
        }
      }
    }
    // ---- stage dy: one float4 (co 0..3) per pixel; its per-channel sums accumulate per thread
    {
      const int r = tid >> 5, col = tid & 31;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < d.Cout) v[co] = a.dy[(size_t)(n * d.Cout + co) * HW + (row0 + r) * W + col0 + col];
      dyT[tid] = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int co = 0; co < 4; ++co) dpart[co] += v[co];
    }
    __syncthreads();
    if (worker) {
      for (int rr = 0; rr < RG; ++rr) {
        const int r = grp * RG + rr;
        const float* xp = xT + ci * CP + (r + kh) * IW;   // halo-tile row r + kh, columns 0 .. 33 <-> taps kw = 0 .. 2
        float x0 = xp[0], x1 = xp[1];
#pragma unroll 8
        for (int col = 0; col < TW; ++col) {
          const float x2 = xp[col + 2];
          const float4 g = dyT[r * TW + col];   // the same address in every lane of a group: a broadcast read
          acc[0][0] = fmaf(g.x, x0, acc[0][0]);
          acc[0][1] = fmaf(g.y, x0, acc[0][1]);
          acc[0][2] = fmaf(g.z, x0, acc[0][2]);
          acc[0][3] = fmaf(g.w, x0, acc[0][3]);
          acc[1][0] = fmaf(g.x, x1, acc[1][0]);
          acc[1][1] = fmaf(g.y, x1, acc[1][1]);
          acc[1][2] = fmaf(g.z, x1, acc[1][2]);
          acc[1][3] = fmaf(g.w, x1, acc[1][3]);
          acc[2][0] = fmaf(g.x, x2, acc[2][0]);
          acc[2][1] = fmaf(g.y, x2, acc[2][1]);
          acc[2][2] = fmaf(g.z, x2, acc[2][2]);
          acc[2][3] = fmaf(g.w, x2, acc[2][3]);
          x0 = x1;
          x1 = x2;
        }
      }
    }
  }

  // ---- sum the G groups (fixed order) and write the slab of this split
  __syncthreads();
  float* const part = smem;   // [G][nitems][12]
  if (worker) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int c = 0; c < 4; ++c) part[(grp * nitems + it) * 12 + kw * 4 + c] = acc[kw][c];
  }
  __syncthreads();
  const size_t K = (size_t)9 * C;
  float* slab = a.slabs + (size_t)blockIdx.x * a.Coutp * K;
  if (tid < nitems) {   // group 0's threads: item `it` = tid
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        float t = 0.f;
        for (int g = 0; g < G; ++g) t += part[(g * nitems + tid) * 12 + kw * 4 + co];
        if (co < d.Cout) slab[(size_t)co * K + (size_t)(kh * 3 + kw) * C + ci] = t;
      }
  }
  // ---- per-channel sums of dy of this split: wave sums, then four waves through LDS
  __syncthreads();
  float* const red = smem;   // [4 waves][4]
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    const float s_ = wave_sum(dpart[co]);
    if ((tid & 63) == 0) red[(tid >> 6) * 4 + co] = s_;
  }
  __syncthreads();
  if (tid < a.Coutp)
    a.dshift[(size_t)blockIdx.x * a.Coutp + tid] =
        (tid < 4 && tid < d.Cout) ? (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]) : 0.f;
}

// ---- host side ------------------------------------------------------------------------------------------------------
bool vunet_wgrad_thin_applicable(const vunet_wgrad_desc* d) {
  return d->Cout >= 1 && d->Cout <= 4 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->C2 == 0 &&
         d->C1 >= 8 && d->C1 <= 64 && d->Ho == d->Hs && d->Wo == d->Ws && d->Hs % 8 == 0 && d->Ws % 32 == 0 &&
         (long)d->N * d->Hs * d->Ws >= 64 * 1024;
}

static int thin_tiles(const vunet_wgrad_desc* d) { return d->N * (d->Hs / 8) * (d->Ws / 32); }

int vunet_wgrad_thin_nslabs(const vunet_wgrad_desc* d) {
  const int t = thin_tiles(d);
  return t < 512 ? t : 512;   // two workgroups per CU; each walks ntiles / 512 tiles
}

int vunet_wgrad_thin_name(const vunet_wgrad_desc* d, char* name, int len) {
  return snprintf(name, len, "conv_wgrad_thin_kernel");
}

int vunet_wgrad_thin_launch(const vunet_wgrad_desc* d, const float* x1, const float* dy, float* slabs, float* dshift,
                            hipStream_t st) {
  WgradThinArgs a;
  a.d = *d;
  a.x1 = x1; a.dy = dy; a.slabs = slabs; a.dshift = dshift;
  a.Coutp = (d->Cout + 31) / 32 * 32;
  a.tiles_w = d->Ws / 32;
  a.tiles_per_img = (d->Hs / 8) * a.tiles_w;
  a.ntiles = thin_tiles(d);
  a.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  if (d->nsplit != vunet_wgrad_thin_nslabs(d)) return VUNET_ERR_ARG;
  const int CP = 10 * 35 + 1;
  const size_t lds = ((size_t)((d->C1 * CP + 3) & ~3) + 4 * 256) * sizeof(float);
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_wgrad_thin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess)
    return VUNET_ERR_LAUNCH;
  VUNET_LAUNCH(conv_wgrad_thin_kernel, dim3(d->nsplit), dim3(256), lds, st, a);
  return vunet_check_launch();
}
