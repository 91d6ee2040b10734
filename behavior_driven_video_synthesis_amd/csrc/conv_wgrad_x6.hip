// Weight gradient of the 3x3 / stride 1 / pad 1 convolution, fp32-accurate on the bf16 matrix cores.
//
//   dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap]          (K = pixels)
//
// Same split as conv_x6_kernel.h: both fp32 operands are split exactly into three bf16 terms and the six leading
// partial products accumulate in fp32 on v_mfma_f32_32x32x16_bf16 (A = dy: rows co, B = f(x): columns ci, K = 16
// pixels = two lane halves x 8 consecutive pixels of one image row).
//
// A workgroup (4 waves) owns MTW*32 output channels x 32 input channels x 9 taps and walks 4 x 32 pixel tiles
// (split-K over tiles; the partial slabs are summed in a fixed order by vunet_weightnorm_bwd -- no atomics).
//   f(x)   staged once per element into LDS, already split: xL[3 planes][6 rows][32 ci][5 units of 8 bf16]; unit 0 of
//          an entry holds the two halo columns (element 7 = column -1 of this entry, element 0 = column 32 of the
//          PREVIOUS entry), units 1..4 the 32 columns.  Lanes (= ci) are 5 units apart: conflict-free ds_read_b128.
//          The three horizontal taps of a k-step come from ONE aligned read plus the neighbouring dwords, shifted in
//          registers by v_alignbit_b32 (5 per plane); the vertical taps are whole-row offsets.
//   dy     never touches LDS: lane (co, k-half) reads its 8 consecutive pixels straight from global memory (32 aligned
//          bytes), all k-steps of the tile issued before the f(x) staging so the latency hides behind it, and is split
//          in registers once per k-step (shared by the 9 taps).  Their sum over pixels is the shift gradient.
// Waves split (m-tile, tile rows) and are folded through LDS at the end, one slab per split leaves the workgroup.
#include <type_traits>

#include "common.h"

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wg_u32x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

struct WgradX6Args {
  vunet_wgrad_desc d;
  const float* x1;
  const float* x2;
  const float* dy;
  float* slabs;
  float* dshift;
  int Ctot, Coutp, tiles_per_img_w, tiles_per_img, ntiles, tps;
  InAct in1, in2;
};

__device__ __forceinline__ uint32_t wg_pack2(float a, float b) {
  wg_bf16x2 p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, p);
}

__device__ __forceinline__ void wg_split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = wg_pack2(a, b);
  float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = wg_pack2(ra, rb);
  ra -= __uint_as_float(m << 16);
  rb -= __uint_as_float(m & 0xffff0000u);
  l = wg_pack2(ra, rb);
}

__device__ __forceinline__ wg_bf16x8 wg_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  wg_u32x4 u = {a, b, c, d};
  return __builtin_bit_cast(wg_bf16x8, u);
}

template <int MTW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x6_kernel(const WgradX6Args a_in) {
  WgradX6Args a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  constexpr int TH = 4;
  // optional register prefetch of the next tile's f(x) across the MFMA block
  constexpr bool PREFETCH = false;   // measured: no gain from the register prefetch (r02), and two m-tiles spill with it
  constexpr int WP = 4 / MTW;        // waves sharing one m-tile (they split the tile rows)
  constexpr int RW = TH / WP;        // tile rows per wave
  constexpr int ENT = 6 * 32;        // (row, ci) entries of the staged tile
  constexpr int PL = ENT * 5 + 1;    // units per plane (+1: the halo unit behind the last entry)
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  wg_u32x4* const xL = reinterpret_cast<wg_u32x4*>(smem4);   // [3][PL]

  const vunet_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave / WP, pw = wave % WP;
  const int split = blockIdx.x;
  const int ci0 = blockIdx.y * 32;
  const int co0 = blockIdx.z * 32 * MTW;
  const int H = d.Hs, W = d.Ws;

  const bool second = ci0 >= d.C1;   // C1 % 32 == 0 is a precondition
  const float* __restrict__ xs = second ? a.x2 : a.x1;
  const int Cs = second ? d.C2 : d.C1;
  const int cbase = second ? ci0 - d.C1 : ci0;
  InAct ia = a.in1;
  ia.seed = second ? a.in2.seed : a.in1.seed;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dsum = 0.f;

  // staging geometry that does not depend on the tile: main units u = tid + 256*i over (row, ci, octet), halo by entry
  int s_row[3], s_ci[3], s_oct[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = tid + 256 * i;       // 0 .. 767
    s_oct[i] = u & 3;
    s_ci[i] = (u >> 2) & 31;
    s_row[i] = u >> 7;
  }
  const int co_lane = co0 + wm * 32 + j;   // the output channel whose dy this lane feeds to the A operand

  const int t_begin = split * a.tps;
  const int t_end = min(t_begin + a.tps, a.ntiles);

  // ---- f(x) staging, register-prefetched one tile ahead: the loads of tile t+1 are issued before the MFMA block of
  //      tile t and split / written to LDS after it (the input tile is single-buffered: two barriers per tile)
  wg_f32x4 xv[3][2];      // main units: 8 consecutive columns each
  float hv[2];            // halo columns -1 / 32 of entry `tid` (tid < 192)
  uint32_t gidx[3], hidx = 0;
  uint32_t okbits = 0;    // bit i: main unit i in range; bit 4 / 5: left / right halo in range
  auto tile_origin = [&](int tile, int& n, int& row0, int& col0) {
    n = tile / a.tiles_per_img;
    const int tr = tile - n * a.tiles_per_img;
    const int ty = tr / a.tiles_per_img_w, tx = tr - ty * a.tiles_per_img_w;
    row0 = ty * TH;
    col0 = tx * 32;
  };
  auto issue_x = [&](int tile) {
    int n, row0, col0;
    tile_origin(tile, n, row0, col0);
    okbits = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ih = row0 - 1 + s_row[i];
      const bool ok = (unsigned)ih < (unsigned)H;
      okbits |= (ok ? 1u : 0u) << i;
      gidx[i] = (uint32_t)(((n * Cs + cbase + s_ci[i]) * H + (ok ? ih : 0)) * W + col0 + 8 * s_oct[i]);
      xv[i][0] = *reinterpret_cast<const wg_f32x4*>(xs + gidx[i]);
      xv[i][1] = *reinterpret_cast<const wg_f32x4*>(xs + gidx[i] + 4);
    }
    {  // every thread loads (threads >= 192 a harmless duplicate of entry tid - 64): no divergent load
      const int e = tid < ENT ? tid : tid - 64;
      const int r = e >> 5, c = e & 31;
      const int ih = row0 - 1 + r;
      const bool rok = (unsigned)ih < (unsigned)H;
      hidx = (uint32_t)(((n * Cs + cbase + c) * H + (rok ? ih : 0)) * W + col0);
      const bool okl = rok && col0 > 0, okr = rok && col0 + 32 < W;
      okbits |= (okl ? 16u : 0u) | (okr ? 32u : 0u);
      hv[0] = xs[okl ? hidx - 1 : hidx];
      hv[1] = xs[okr ? hidx + 32 : hidx];
    }
  };
  auto write_x = [&]() {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool ok = (okbits >> i) & 1u;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = e < 4 ? xv[i][0][e] : xv[i][1][e - 4];
        f[e] = ok ? apply_in_act(ia, t, gidx[i] + e) : 0.f;
      }
      uint32_t ph[4], pm[4], pl[4];
      wg_split2(f[0], f[1], ph[0], pm[0], pl[0]);
      wg_split2(f[2], f[3], ph[1], pm[1], pl[1]);
      wg_split2(f[4], f[5], ph[2], pm[2], pl[2]);
      wg_split2(f[6], f[7], ph[3], pm[3], pl[3]);
      const int unit = (s_row[i] * 32 + s_ci[i]) * 5 + 1 + s_oct[i];
      xL[unit] = wg_u32x4{ph[0], ph[1], ph[2], ph[3]};
      xL[PL + unit] = wg_u32x4{pm[0], pm[1], pm[2], pm[3]};
      xL[2 * PL + unit] = wg_u32x4{pl[0], pl[1], pl[2], pl[3]};
    }
    if (tid < ENT) {
      const float fl = (okbits & 16u) ? apply_in_act(ia, hv[0], hidx - 1) : 0.f;
      const float fr = (okbits & 32u) ? apply_in_act(ia, hv[1], hidx + 32) : 0.f;
      uint32_t ph, pm, pl;
      wg_split2(fr, fl, ph, pm, pl);   // low half = right halo (element 0 of the NEXT entry's unit 0), high = left
      unsigned short* const base = reinterpret_cast<unsigned short*>(xL);
      const int ul = tid * 5, ur = (tid + 1) * 5;   // halo units: this entry's (left, element 7), the next one's (right, element 0)
      base[(size_t)(ul) * 8 + 7] = (unsigned short)(ph >> 16);
      base[(size_t)(PL + ul) * 8 + 7] = (unsigned short)(pm >> 16);
      base[(size_t)(2 * PL + ul) * 8 + 7] = (unsigned short)(pl >> 16);
      base[(size_t)(ur) * 8] = (unsigned short)(ph & 0xffffu);
      base[(size_t)(PL + ur) * 8] = (unsigned short)(pm & 0xffffu);
      base[(size_t)(2 * PL + ur) * 8] = (unsigned short)(pl & 0xffffu);
    }
  };

  if (PREFETCH && t_begin < t_end) issue_x(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    int n, row0, col0;
    tile_origin(tile, n, row0, col0);

    // ---- dy of this wave's rows: RW rows x 2 k-steps x 8 pixels per lane, straight to registers
    wg_f32x4 dyv[RW][2][2];
    {
      const float* __restrict__ dp = a.dy + ((size_t)(n * d.Cout + co_lane) * H + row0 + pw * RW) * W + col0 + 8 * h;
#pragma unroll
      for (int rr = 0; rr < RW; ++rr)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          dyv[rr][s][0] = *reinterpret_cast<const wg_f32x4*>(dp + rr * W + 16 * s);
          dyv[rr][s][1] = *reinterpret_cast<const wg_f32x4*>(dp + rr * W + 16 * s + 4);
        }
    }

    __syncthreads();  // the previous tile's LDS reads are done
    if constexpr (!PREFETCH) issue_x(tile);
    write_x();        // PREFETCH: this tile's f(x) was loaded during the previous tile's MFMA block
    __syncthreads();
    if constexpr (PREFETCH) issue_x(tile + 1 < t_end ? tile + 1 : tile);   // unconditional (a conditional load is sunk to its use)

    // ---- MFMA: this wave's RW rows x 2 k-steps x 3 kernel rows = RW*6 steps of 18 MFMAs (3 taps x 6 products).
    //      The f(x) reads of step i+1 (one aligned 16-byte read + the two neighbouring edge dwords, per plane) are
    //      issued before the MFMAs of step i: LDS latency hides behind 18 MFMAs instead of stalling every step.
    const uint32_t* const xw = reinterpret_cast<const uint32_t*>(xL);
    struct Raw {
      wg_u32x4 c[3];
      uint32_t p3[3], n0[3];
    };
    auto load_raw = [&](int st) {
      const int rr = st / 6, s = (st / 3) & 1, kh = st % 3;
      // unit of (row pw*RW + rr + kh, ci j, octet 2s + h)
      const int unit = ((pw * RW + rr + kh) * 32 + j) * 5 + 1 + 2 * s + h;
      Raw r;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        r.c[p] = xL[p * PL + unit];
        r.p3[p] = xw[(size_t)(p * PL + unit - 1) * 4 + 3];
        r.n0[p] = xw[(size_t)(p * PL + unit + 1) * 4];
      }
      return r;
    };
    constexpr int NSTEP = RW * 6;
    Raw cur = load_raw(0);
    wg_bf16x8 Ah, Am, Al;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int rr = st / 6, s = (st / 3) & 1, kh = st % 3;
      __builtin_amdgcn_sched_barrier(0);
      Raw nxt = cur;
      if (st + 1 < NSTEP) nxt = load_raw(st + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads up here (the scheduler otherwise sinks them to the end of the step)
      if (kh == 0) {   // a new k-step: split this lane's 8 dy pixels once for the 9 taps
        const wg_f32x4 d0 = dyv[rr][s][0], d1 = dyv[rr][s][1];
        dsum += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
        uint32_t ah[4], am[4], al[4];
        wg_split2(d0[0], d0[1], ah[0], am[0], al[0]);
        wg_split2(d0[2], d0[3], ah[1], am[1], al[1]);
        wg_split2(d1[0], d1[1], ah[2], am[2], al[2]);
        wg_split2(d1[2], d1[3], ah[3], am[3], al[3]);
        Ah = wg_frag(ah[0], ah[1], ah[2], ah[3]);
        Am = wg_frag(am[0], am[1], am[2], am[3]);
        Al = wg_frag(al[0], al[1], al[2], al[3]);
      }
      // one plane of f(x) at a time, smallest terms first: only three shifted fragments are live
      //   plane l: Ah*Bl ; plane m: Am*Bm, Ah*Bm ; plane h: Al*Bh, Am*Bh, Ah*Bh
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) {
        const int p = 2 - pp;
        const wg_u32x4 c = cur.c[p];
        const uint32_t s01 = __builtin_amdgcn_alignbit(c.y, c.x, 16), s12 = __builtin_amdgcn_alignbit(c.z, c.y, 16),
                       s23 = __builtin_amdgcn_alignbit(c.w, c.z, 16);
        wg_bf16x8 B[3];
        B[0] = wg_frag(__builtin_amdgcn_alignbit(c.x, cur.p3[p], 16), s01, s12, s23);   // columns p - 1
        B[1] = __builtin_bit_cast(wg_bf16x8, c);                                        // columns p
        B[2] = wg_frag(s01, s12, s23, __builtin_amdgcn_alignbit(cur.n0[p], c.w, 16));   // columns p + 1
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          f32x16 cc = acc[kh * 3 + kw];
          if (p == 0) cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, B[kw], cc, 0, 0, 0);
          if (p <= 1) cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, B[kw], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B[kw], cc, 0, 0, 0);
          acc[kh * 3 + kw] = cc;
        }
      }
      cur = nxt;
    }
  }

  // ---- fold the WP row-part waves of each m-tile into the pw == 0 wave through LDS (fixed order), one tap at a
  //      time (16 KiB), so that one slab per split leaves the workgroup
  {
    float* const red = reinterpret_cast<float*>(smem4);  // [4 waves][16][64]
    auto fold = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
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
    };
    fold(std::integral_constant<int, 0>{});
    fold(std::integral_constant<int, 1>{});
    fold(std::integral_constant<int, 2>{});
    fold(std::integral_constant<int, 3>{});
    fold(std::integral_constant<int, 4>{});
    fold(std::integral_constant<int, 5>{});
    fold(std::integral_constant<int, 6>{});
    fold(std::integral_constant<int, 7>{});
    fold(std::integral_constant<int, 8>{});
    __syncthreads();
    if (pw != 0) red[wave * 64 + lane] = dsum;
    __syncthreads();
    if (pw == 0)
      for (int o = 1; o < WP; ++o) dsum += red[(wave + o) * 64 + lane];
  }

  // ---- partial slab of this split:  [split][Coutp][9*Ctot], k order (tap, ci)
  const size_t KT = (size_t)9 * a.Ctot;
  float* slab = a.slabs + (size_t)split * a.Coutp * KT;
  const int ci = ci0 + j;
  if (pw == 0) {
    auto store = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[(size_t)co * KT + (size_t)t * a.Ctot + ci] = acc[t][r];
      }
    };
    store(std::integral_constant<int, 0>{});
    store(std::integral_constant<int, 1>{});
    store(std::integral_constant<int, 2>{});
    store(std::integral_constant<int, 3>{});
    store(std::integral_constant<int, 4>{});
    store(std::integral_constant<int, 5>{});
    store(std::integral_constant<int, 6>{});
    store(std::integral_constant<int, 7>{});
    store(std::integral_constant<int, 8>{});
    if (blockIdx.y == 0) {
      const float tot = dsum + __shfl_xor(dsum, 32, 64);
      if (h == 0) a.dshift[(size_t)split * a.Coutp + co_lane] = tot;
    }
  }
}

// ---- host side ----------------------------------------------------------------------------
bool vunet_wgrad_x6_applicable(const vunet_wgrad_desc* d) {
  static const bool on = [] {
    const char* e = getenv("VUNET_CONV_PRECISION");
    return !(e && (e[0] == 'f' || e[0] == 'F'));
  }();
  if (!on || (d->flags & 1)) return false;
  return d->KH == 3 && d->KW == 3 && d->pad == 1 && d->stride == 1 && d->Ho == d->Hs && d->Wo == d->Ws &&
         d->Ws % 32 == 0 && d->Hs % 4 == 0 && d->C1 > 0 && d->C1 % 32 == 0 && d->C2 % 32 == 0 && d->Cout % 32 == 0;
}

static void x6_geometry(const vunet_wgrad_desc* d, int& MTW, int& ntiles, int& ciblocks, int& coblocks) {
  MTW = d->Cout % 64 == 0 ? 2 : 1;
  ntiles = d->N * (d->Ho / 4) * (d->Wo / 32);
  ciblocks = (d->C1 + d->C2) / 32;
  coblocks = (d->Cout + 32 * MTW - 1) / (32 * MTW);
}

int vunet_wgrad_x6_nslabs(const vunet_wgrad_desc* d) {
  int MTW, ntiles, ciblocks, coblocks;
  x6_geometry(d, MTW, ntiles, ciblocks, coblocks);
  // 512 workgroups = exactly the two resident per CU, one round: measured best with the fp16 kernel (r02: 1024 -> 512:
  // conv_wgrad_h2_kernel<2> 2.08 -> 1.87 ms per step, half the slab volume for the reduce; 768 and 256 are worse)
  static const int target = [] { const char* e = getenv("VUNET_WGRAD_SPLIT_WGS"); return e ? atoi(e) : 512; }();   // tuning
  // (fp16 scheme, round 4: a workgroup is 512 threads -- two groups in antiphase, conv_wgrad_h2.hip -- so ONE workgroup per
  // CU is resident: half the workgroups, each walking twice the tiles, half the slab volume again)
  int S = ((d->flags & 2) ? target / 2 : target) / (ciblocks * coblocks);
  if (S > ntiles / 2) S = ntiles / 2;
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  return S;
}

int vunet_wgrad_x6_name(const vunet_wgrad_desc* d, char* name, int len) {
  return snprintf(name, len, "conv_wgrad_x6_kernel<%d>", d->Cout % 64 == 0 ? 2 : 1);
}

int vunet_wgrad_x6_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs,
                          float* dshift, hipStream_t st) {
  WgradX6Args a;
  a.d = *d;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.slabs = slabs; a.dshift = dshift;
  int MTW, ciblocks, coblocks;
  x6_geometry(d, MTW, a.ntiles, ciblocks, coblocks);
  a.Ctot = d->C1 + d->C2;
  a.Coutp = (d->Cout + 31) / 32 * 32;
  a.tiles_per_img_w = d->Wo / 32;
  a.tiles_per_img = (d->Ho / 4) * a.tiles_per_img_w;
  a.tps = (a.ntiles + d->nsplit - 1) / d->nsplit;
  a.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  a.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  dim3 grid(d->nsplit, ciblocks, coblocks);
  constexpr size_t lds = (size_t)3 * (6 * 32 * 5 + 1) * 16;
  if (MTW == 1) VUNET_LAUNCH((conv_wgrad_x6_kernel<1>), grid, dim3(256), lds, st, a);
  else VUNET_LAUNCH((conv_wgrad_x6_kernel<2>), grid, dim3(256), lds, st, a);
  return vunet_check_launch();
}
