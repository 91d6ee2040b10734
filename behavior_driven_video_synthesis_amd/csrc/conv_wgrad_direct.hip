// Weight gradient on the fp16 matrix cores for the geometries conv_wgrad_h2.hip does not take: small maps (4 x 4 ... 16 x 16,
// the VUnet bottleneck: models/vunets.py:520-597), the stride-2 Downsample convolutions (lib/modules.py:148-161) at every
// map size, and the 1 x 1 `nin` layers (lib/modules.py:201-205) -- 3.5 ms and ~80 launches of fp32-input MFMA kernels at
// 10 - 50 TFLOP/s per bs-16 step before round 3.
//
//   dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap]          (K = output pixels)
//
// "Direct": no LDS.  Both MFMA operands of v_mfma_f32_32x32x16_f16 want, per lane, 8 consecutive K entries of one row /
// column -- here 8 consecutive output pixels of one channel, which is how NCHW stores them.  So a lane loads its dy octet
// (A: lane = output channel) and, for B (lane = input channel), the 3 (or S + 3) input rows its octet's taps touch -- the
// row segment plus a halo column, 16-byte loads -- applies the prologue and the operand scale ONCE per loaded element, and
// builds the nine tap fragments by selecting elements of those rows in registers (a tap is a column / row offset; at
// stride 2 every other element).  Same operand split as conv_wgrad_h2.hip: power-of-two scales from the tensor maxima, two
// fp16 terms with the low one at natural scale, three products into ONE accumulator per tap, nine accumulators per wave.
// One wave owns a 32 x 32 (co, ci) tile for all taps and a contiguous range of pixel octets (split-K); slabs in the layout
// vunet_weightnorm_bwd* reduces.  Channel counts need not be multiples of 32 (rows / columns past the end are zero).
#include <type_traits>

#include "common.h"
#include "split_h2.h"

struct WgradDirectArgs {
  vunet_wgrad_desc d;
  const float* x1;
  const float* x2;
  const float* dy;
  float* slabs;
  float* dshift;
  const float* amax_x;
  const float* amax_x2;   // the second source's 512 partial maxima in a buffer of their own, or nullptr (split_h2.h: h2_amax4)
  const float* amax_dy;
  int Ctot, Coutp, ncot, noct, ops;   // co tiles; pixel octets in all; octets per split (even)
  int opr, opi;                       // octets per output row / per image
  InAct in1, in2;
};

typedef uint32_t wd_u32x4 __attribute__((ext_vector_type(4)));
typedef float wd_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wd_split2(float a, float b, uint32_t& h, uint32_t& l) {   // low term at natural scale
  h2_f16x2 ph, pl;
  ph[0] = (_Float16)a;
  ph[1] = (_Float16)b;
  pl[0] = (_Float16)(a - (float)ph[0]);
  pl[1] = (_Float16)(b - (float)ph[1]);
  h = __builtin_bit_cast(uint32_t, ph);
  l = __builtin_bit_cast(uint32_t, pl);
}

__device__ __forceinline__ h2_f16x8 wd_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  wd_u32x4 u = {a, b, c, d};
  return __builtin_bit_cast(h2_f16x8, u);
}

// S: stride (1, 2).  W4: output maps 4 wide (an octet is two output rows of 4).  T: 9 (3x3, pad 1) or 1 (1x1, S = 1), or
// 3: the three taps of ONE kernel row -- a wave then owns (co tile, kernel row kh): it loads one input row per octet instead
// of three and keeps three accumulators instead of nine (~110 VGPRs instead of 243), so four waves per SIMD cover the
// load -> MFMA dependency that bounds the nine-tap form on the large stride-2 maps (r03: 53 TFLOP/s, two waves per SIMD
// waiting on 17 uncoalesced loads per 27 MFMAs).
// (bx, by): the workgroup's place in THIS layer's grid -- blockIdx for a launch of its own, a slice of the grid for a batched one
template <int S, bool W4, int T>
__device__ __forceinline__ void wgrad_direct_body(WgradDirectArgs& a, const int bx, const int by) {
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  // rows / loaded columns of the input one octet touches
  static_assert(T != 3 || !W4, "the kernel-row split is for maps >= 8 wide");
  constexpr int R = (T == 1 || T == 3) ? 1 : (W4 ? S + 3 : 3);
  constexpr int L = T == 1 ? 8 : (W4 ? 4 * S + 2 : 8 * S + 2);   // L - 1 used at S = 2 (no right halo)
  const vunet_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  // (split, co tile[, kernel row]), fastest last: the waves of a workgroup share x (and, row-split, dy)
  const int unit0 = bx * 4 + wave;
  const int krow = T == 3 ? unit0 % 3 : 0;
  const int unit = T == 3 ? unit0 / 3 : unit0;
  const int cot = unit % a.ncot, split = unit / a.ncot;
  const int ci0 = by * 32;
  const int H = d.Hs, W = d.Ws, HW = H * W, HoWo = d.Ho * d.Wo;

  // ---- operand scales
  float sx, sdy, descale, descale2;
  {
    __shared__ float redm[8];
    const float4 px = h2_amax4(a.amax_x, a.amax_x2, tid), pd = reinterpret_cast<const float4*>(a.amax_dy)[tid];
    float mx = wave_max(fmaxf(fmaxf(px.x, px.y), fmaxf(px.z, px.w)));
    float md = wave_max(fmaxf(fmaxf(pd.x, pd.y), fmaxf(pd.z, pd.w)));
    if (lane == 0) { redm[wave] = mx; redm[4 + wave] = md; }
    __syncthreads();
    mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    md = fmaxf(fmaxf(redm[4], redm[5]), fmaxf(redm[6], redm[7]));
    if (a.in1.thresh) mx *= a.in1.keep_scale;
    const int ex = h2_scale_exp(mx), ed = h2_scale_exp(md);
    sx = h2_pow2(ex);
    sdy = h2_pow2(ed);
    h2_pow2_pair(-(ex + ed), descale, descale2);
  }
  if (split >= d.nsplit) return;   // (grid rounded up to whole workgroups; after the only barrier)

  const int ci = ci0 + j;
  const bool ci_ok = ci < a.Ctot;
  const bool second = ci_ok && ci >= d.C1;
  const float* __restrict__ xs = second ? a.x2 : a.x1;
  const int Cs = second ? d.C2 : d.C1;
  const int cl = ci_ok ? (second ? ci - d.C1 : ci) : 0;
  InAct ia = a.in1;
  ia.seed = second ? a.in2.seed : a.in1.seed;
  const int co = cot * 32 + j;
  const bool co_ok = co < d.Cout;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dsum = 0.f;

  const int pro_form = in_act_form_of(a.in1);   // (wave-uniform; the two sources differ in the dropout seed only)
  const int o_begin = split * a.ops, o_end = min(o_begin + a.ops, a.noct);
  // (1x1: four K steps unrolled, so that their loads -- the whole cost of this HBM-bound case -- are in flight together)
#pragma unroll(T == 1 ? 4 : 1)
  for (int o2 = o_begin; o2 < o_end; o2 += 2) {
    const int o = o2 + h;                     // this lane's octet (the k-half of the lane)
    const bool ov = o < o_end;
    const int oc = ov ? o : o_begin;
    const int n = oc / a.opi, rem = oc - n * a.opi;
    int r0, c0;                               // first output row / column of the octet
    if (T == 1) { r0 = 0; c0 = rem * 8; }     // (1x1: a flat pixel offset inside the image)
    else if (W4) { r0 = rem * 2; c0 = 0; }
    else { r0 = rem / a.opr; c0 = (rem - r0 * a.opr) * 8; }

    // ---- A: 8 consecutive dy pixels of channel co
    h2_f16x8 Ah, Al;
    {
      const float* dp = a.dy + (size_t)(n * d.Cout + (co_ok ? co : 0)) * HoWo + (T == 1 ? c0 : r0 * d.Wo + c0);
      wd_f32x4 d0 = *reinterpret_cast<const wd_f32x4*>(dp), d1 = *reinterpret_cast<const wd_f32x4*>(dp + 4);
      if (!(ov && co_ok)) { d0 = wd_f32x4{0.f, 0.f, 0.f, 0.f}; d1 = d0; }
      dsum += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
      uint32_t ah[4], al[4];
      wd_split2(d0[0] * sdy, d0[1] * sdy, ah[0], al[0]);
      wd_split2(d0[2] * sdy, d0[3] * sdy, ah[1], al[1]);
      wd_split2(d1[0] * sdy, d1[1] * sdy, ah[2], al[2]);
      wd_split2(d1[2] * sdy, d1[3] * sdy, ah[3], al[3]);
      Ah = wd_frag(ah[0], ah[1], ah[2], ah[3]);
      Al = wd_frag(al[0], al[1], al[2], al[3]);
    }

    // ---- B source: R input rows x L columns of channel ci, prologue applied, scaled; seg[rr][0] is input column
    //      S * c0 - 1 (the left halo), rows start at S * r0 - 1.  The prologue as straight-line code, one instantiation per
    //      form behind a wave-uniform branch (common.h: in_act_form; out-of-range elements are zeroed BEFORE it and scaled
    //      by 0 -- "ok ? f(v) * sx : 0" puts every element's prologue in an exec-mask block of its own).
    float seg[R][L];
    auto stage = [&](auto pro_c) {
      constexpr int PRO = decltype(pro_c)::value;
      const uint32_t cbase = (uint32_t)(n * Cs + cl) * (uint32_t)HW;
      const float* xp = xs + cbase;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const int ih = T == 1 ? 0 : S * r0 - 1 + rr + krow;
        const bool rok = ov && ci_ok && (T == 1 || (unsigned)ih < (unsigned)H);
        const int ihc = rok ? ih : 0;
        const float sxr = rok ? sx : 0.f;
        if (T == 1) {
          const uint32_t off = (uint32_t)c0;
          const wd_f32x4 v0 = *reinterpret_cast<const wd_f32x4*>(xp + off), v1 = *reinterpret_cast<const wd_f32x4*>(xp + off + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t_ = e < 4 ? v0[e] : v1[e - 4];
            seg[rr][e] = in_act_form<PRO>(ia, rok ? t_ : 0.f, cbase + off + e) * sxr;
          }
        } else {
          constexpr int NV = (L - 2) / 4;      // aligned 16-byte loads per row: columns S*c0 .. S*c0 + 4*NV - 1
          const uint32_t off = (uint32_t)(ihc * W + S * c0);
          wd_f32x4 v[NV];
#pragma unroll
          for (int q = 0; q < NV; ++q) v[q] = *reinterpret_cast<const wd_f32x4*>(xp + off + 4 * q);
          const bool lok = rok && c0 > 0, hok = rok && S == 1 && S * c0 + 4 * NV < W;
          const float lv = xp[lok ? off - 1 : off];
          const float hv = xp[hok ? off + 4 * NV : off];
          seg[rr][0] = in_act_form<PRO>(ia, lok ? lv : 0.f, cbase + off - 1) * (lok ? sx : 0.f);
#pragma unroll
          for (int e = 0; e < 4 * NV; ++e)
            seg[rr][1 + e] = in_act_form<PRO>(ia, rok ? v[e >> 2][e & 3] : 0.f, cbase + off + e) * sxr;
          seg[rr][L - 1] = in_act_form<PRO>(ia, hok ? hv : 0.f, cbase + off + 4 * NV) * (hok ? sx : 0.f);
        }
      }
    };
    if (pro_form == 2) stage(std::integral_constant<int, 2>{});
    else if (pro_form == 1) stage(std::integral_constant<int, 1>{});
    else if (pro_form == 0) stage(std::integral_constant<int, 0>{});
    else stage(std::integral_constant<int, 3>{});

    // ---- nine taps: fragment = 8 elements selected from the rows, split, three products
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int kh = T == 3 ? 0 : t / 3, kw = t % 3;   // (row-split: the one staged row IS kernel row krow)
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (T == 1) f[e] = seg[0][e];
        else if (W4) f[e] = seg[(e >> 2) * S + kh][(e & 3) * S + kw];
        else f[e] = seg[kh][e * S + kw];
      }
      uint32_t bh[4], bl[4];
      wd_split2(f[0], f[1], bh[0], bl[0]);
      wd_split2(f[2], f[3], bh[1], bl[1]);
      wd_split2(f[4], f[5], bh[2], bl[2]);
      wd_split2(f[6], f[7], bh[3], bl[3]);
      const h2_f16x8 Bh = wd_frag(bh[0], bh[1], bh[2], bh[3]), Bl = wd_frag(bl[0], bl[1], bl[2], bl[3]);
      f32x16 cc = acc[t];
      cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, cc, 0, 0, 0);
      cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, cc, 0, 0, 0);
      cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, cc, 0, 0, 0);
      acc[t] = cc;
    }
  }

  // ---- partial slab of this split:  [split][Coutp][T * Ctot], k order (tap, ci)
  const size_t KT = (size_t)(T == 3 ? 9 : T) * a.Ctot;
  float* slab = a.slabs + (size_t)split * a.Coutp * KT;
  if (ci_ok) {
    auto store = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int tap = T == 3 ? krow * 3 + t : t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cr = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (cr < d.Cout) slab[(size_t)cr * KT + (size_t)tap * a.Ctot + ci] = acc[t][r] * descale * descale2;
      }
    };
    store(std::integral_constant<int, 0>{});
    if constexpr (T == 3) {
      store(std::integral_constant<int, 1>{});
      store(std::integral_constant<int, 2>{});
    }
    if constexpr (T == 9) {
      store(std::integral_constant<int, 1>{});
      store(std::integral_constant<int, 2>{});
      store(std::integral_constant<int, 3>{});
      store(std::integral_constant<int, 4>{});
      store(std::integral_constant<int, 5>{});
      store(std::integral_constant<int, 6>{});
      store(std::integral_constant<int, 7>{});
      store(std::integral_constant<int, 8>{});
    }
  }
  if (by == 0 && krow == 0) {
    const float tot = dsum + __shfl_xor(dsum, 32, 64);
    if (h == 0 && co < a.Coutp) a.dshift[(size_t)split * a.Coutp + co] = co_ok ? tot : 0.f;
  }
}

template <int S, bool W4, int T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_direct_kernel(const WgradDirectArgs a_in) {
  WgradDirectArgs a = a_in;
  wgrad_direct_body<S, W4, T>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- stride-2 3x3 on the large maps, BOTH operands staged through LDS (round 5) -------------------------------
// The direct form above reads x as 64-byte pieces of 64 different cache lines per wave instruction (lane = input channel), once
// per output-channel tile and per kernel row: 366 MB per launch against 201 MB of algorithmic bytes at the 256^2 level, 58 % of
// the wave cycles waiting (profiles/r04_pmc_sq.json).  Here a workgroup owns (pixel split, 32-channel tile); a step is 16 output
// pixels of one output row.  Per step the three input rows it touches are loaded ONCE with coalesced 16-byte loads along the row,
// the prologue (ELU, dropout) + operand scale + two-term split applied once per element, and written to LDS as COLUMN-PARITY
// planes: E[c] = x[2c], O[c] = x[2c - 1] (17 entries) -- a tap's eight stride-2 elements are then eight consecutive fp16, one
// aligned ds_read_b128 (kw = 0: O, kw = 1: E, kw = 2: O shifted by one entry with v_alignbit).  dy likewise.  Waves =
// (output-channel tile, kernel row); planes are [row][plane][channel][48 B]: 16 lanes x 16 B on 64 distinct banks.
struct S2Lds {
  static constexpr int PITCH = 48;                       // bytes per (plane, channel) row: 17 fp16 + padding
  static constexpr int XPLANE = 32 * PITCH;              // one plane of one input row
  static constexpr int XBYTES = 3 * 4 * XPLANE;          // rows x (E hi, E lo, O hi, O lo)
  static constexpr int DYCOT = 2 * 32 * PITCH;           // (hi, lo) x 32 channels of one co tile
};

// Loads of the ring as inline assembly: the compiler's own wait insertion loses the order of loads across the loop's back edge
// and waits for (nearly) all of them at the first use -- the ring then hides nothing.  The asm loads are invisible to it; the
// kernel waits itself, with a count (s2_wait): "all but the N newest".  Every thread issues the same number of loads per step.
__device__ __forceinline__ wd_f32x4 s2_load4(const float* p) {
  wd_f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float s2_load1(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void s2_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is six bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NWAVE, int NU, int NCOT>
__global__ __launch_bounds__(64 * NWAVE) void conv_wgrad_s2_kernel(const WgradDirectArgs a_in) {
  WgradDirectArgs a = a_in;
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  constexpr int NTHR = 64 * NWAVE;
  constexpr int NX = (3 * 32 * 8 + NTHR - 1) / NTHR;     // float4 loads of x per thread and step
  constexpr int NDY = (NCOT * 32 * 4 + NTHR - 1) / NTHR; // float4 loads of dy per thread and step
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_lds[];
  unsigned char* const xL = s2_lds;
  unsigned char* const dL = s2_lds + S2Lds::XBYTES;
  const vunet_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int split = blockIdx.x, ci0 = blockIdx.y * 32;
  const int H = d.Hs, W = d.Ws, HW = H * W, HoWo = d.Ho * d.Wo;
  const int ppr = d.Wo >> 4, ppi = d.Ho * ppr;            // pixel pairs (16 output pixels) per row / image

  // ---- operand scales (as the direct kernel)
  float sx, sdy, descale, descale2;
  {
    __shared__ float redm[2 * NWAVE];
    float mx = 0.f, md = 0.f;
    for (int i = tid; i < 256; i += NTHR) {   // 1024 partial maxima per tensor
      const float4 px = h2_amax4(a.amax_x, a.amax_x2, i), pd = reinterpret_cast<const float4*>(a.amax_dy)[i];
      mx = fmaxf(mx, fmaxf(fmaxf(px.x, px.y), fmaxf(px.z, px.w)));
      md = fmaxf(md, fmaxf(fmaxf(pd.x, pd.y), fmaxf(pd.z, pd.w)));
    }
    mx = wave_max(mx);
    md = wave_max(md);
    if (lane == 0) { redm[wave] = mx; redm[NWAVE + wave] = md; }
    __syncthreads();
    mx = md = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) { mx = fmaxf(mx, redm[w]); md = fmaxf(md, redm[NWAVE + w]); }
    if (a.in1.thresh) mx *= a.in1.keep_scale;
    const int ex = h2_scale_exp(mx), ed = h2_scale_exp(md);
    sx = h2_pow2(ex);
    sdy = h2_pow2(ed);
    h2_pow2_pair(-(ex + ed), descale, descale2);
  }
  const bool second = ci0 >= d.C1;                        // (C1 % 32 == 0: a channel tile lies in one source)
  const float* __restrict__ xs = second ? a.x2 : a.x1;
  const int Cs = second ? d.C2 : d.C1, cl0 = second ? ci0 - d.C1 : ci0;
  InAct ia = a.in1;
  ia.seed = second ? a.in2.seed : a.in1.seed;
  const int pro_form = in_act_form_of(a.in1);

  const int pairs = d.N * ppi, pps = (pairs + d.nsplit - 1) / d.nsplit;
  const int pb = split * pps, pe = min(pb + pps, pairs);

  // this wave's units: kernel row krow, co tiles cotb and (NU == 2) cotb + 2
  const int krow = wave % 3, cotb = wave / 3;
  f32x16 acc[NU][3];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;

  // global loads run D steps ahead of their use.  Measured (tools/time_wgrad_s2.py, bs 16): the four-deep ring wins on the
  // 128-channel layer (two units per wave: 80 -> 74 us) and loses on the 64-channel one (80 -> 101 us: 40 more registers for
  // a step that is bound by its conversion work, ~200 vector instructions per thread against 9 MFMAs per wave, not by latency)
  constexpr int D = NCOT == 4 ? 4 : 1;
  wd_f32x4 xv[D][NX], dv[D][NDY];
  float xhalo[D];
  float dsum[NDY];
#pragma unroll
  for (int i = 0; i < NDY; ++i) dsum[i] = 0.f;

  // (image, output row, first output column) of a step: divided out once, then advanced step by step -- the loads run D steps
  // ahead of the conversion, so there are two cursors
  struct Cursor { int n, r, cb; };
  auto cursor_at = [&](int p) {
    Cursor c;
    c.n = p / ppi;
    const int rem = p - c.n * ppi;
    c.r = rem / ppr;
    c.cb = (rem - c.r * ppr) << 4;
    return c;
  };
  auto advance = [&](Cursor& c) {
    c.cb += 16;
    if (c.cb == d.Wo) {
      c.cb = 0;
      if (++c.r == d.Ho) { c.r = 0; ++c.n; }
    }
  };
  Cursor cur_load = cursor_at(pb), cur_conv = cur_load;
  auto prefetch = [&](int p, auto slot_c) {
    constexpr int K_ = decltype(slot_c)::value;
    (void)p;
    const int n = cur_load.n, r = cur_load.r, cb = cur_load.cb;
    advance(cur_load);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int u = tid + NTHR * i;                      // (row rr, channel ci, quad q), q fastest
      const int q = u & 7, ci = (u >> 3) & 31, rr = u >> 8;
      const int ih = 2 * r - 1 + rr;
      const bool ok = (unsigned)ih < (unsigned)H;          // (u >= 768: rr = 3 reads row 2 r + 2 <= H - 1 or is clamped; unused)
      const float* xp = xs + ((size_t)(n * Cs + cl0 + ci) * H + (ok ? ih : 0)) * W + 2 * cb + 4 * q;
      xv[K_][i] = s2_load4(xp);   // (masked when converted)
    }
    {                                                     // the left halo column of (row, channel): threads 0 .. 95
      const int ci = tid & 31, rr = (tid >> 5) % 3;
      const int ih = 2 * r - 1 + rr;
      const bool ok = (unsigned)ih < (unsigned)H && cb > 0;
      xhalo[K_] = s2_load1(xs + ((size_t)(n * Cs + cl0 + ci) * H + (ok ? ih : 0)) * W + (ok ? 2 * cb - 1 : 0));
    }
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int u = tid + NTHR * i;                      // (co tile, channel, quad), quad fastest
      const int q4 = u & 3, co = (u >> 2) & 31, cot = u >> 7;
      const bool ok = cot < NCOT;
      const float* dp = a.dy + ((size_t)(n * d.Cout + (ok ? cot * 32 + co : 0)) * d.Ho + r) * d.Wo + cb + 4 * q4;
      dv[K_][i] = s2_load4(dp);
    }
  };
  auto convert = [&](int p, auto pro_c, auto slot_c) {
    constexpr int PRO = decltype(pro_c)::value;
    constexpr int K_ = decltype(slot_c)::value;
    (void)p;
    const int n = cur_conv.n, r = cur_conv.r, cb = cur_conv.cb;
    advance(cur_conv);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int u = tid + NTHR * i;
      if (u >= 768) continue;
      const int q = u & 7, ci = (u >> 3) & 31, rr = u >> 8;
      const int ih = 2 * r - 1 + rr;
      const bool ok = (unsigned)ih < (unsigned)H;
      const uint32_t idx = (uint32_t)(n * Cs + cl0 + ci) * (uint32_t)HW + (uint32_t)((ok ? ih : 0) * W + 2 * cb + 4 * q);
      float f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = ok ? in_act_form<PRO>(ia, xv[K_][i][e], idx + e) * sx : 0.f;
      uint32_t eh, el, oh, ol;
      wd_split2(f[0], f[2], eh, el);                      // even input columns -> E[2q], E[2q + 1]
      wd_split2(f[1], f[3], oh, ol);                      // odd input columns  -> O[2q + 1], O[2q + 2]
      unsigned char* base = xL + (size_t)rr * 4 * S2Lds::XPLANE + ci * S2Lds::PITCH;
      *reinterpret_cast<uint32_t*>(base + 0 * S2Lds::XPLANE + 4 * q) = eh;
      *reinterpret_cast<uint32_t*>(base + 1 * S2Lds::XPLANE + 4 * q) = el;
      uint16_t* po_h = reinterpret_cast<uint16_t*>(base + 2 * S2Lds::XPLANE) + 2 * q + 1;
      uint16_t* po_l = reinterpret_cast<uint16_t*>(base + 3 * S2Lds::XPLANE) + 2 * q + 1;
      po_h[0] = (uint16_t)(oh & 0xffffu);
      po_h[1] = (uint16_t)(oh >> 16);
      po_l[0] = (uint16_t)(ol & 0xffffu);
      po_l[1] = (uint16_t)(ol >> 16);
    }
    if (tid < 96) {
      const int ci = tid & 31, rr = tid >> 5;
      const int ih = 2 * r - 1 + rr;
      const bool ok = (unsigned)ih < (unsigned)H && cb > 0;
      const uint32_t idx = (uint32_t)(n * Cs + cl0 + ci) * (uint32_t)HW + (uint32_t)((ok ? ih : 0) * W + (ok ? 2 * cb - 1 : 0));
      const float f0 = ok ? in_act_form<PRO>(ia, xhalo[K_], idx) * sx : 0.f;
      uint32_t hh, hl;
      wd_split2(f0, 0.f, hh, hl);
      unsigned char* base = xL + (size_t)rr * 4 * S2Lds::XPLANE + ci * S2Lds::PITCH;
      *reinterpret_cast<uint16_t*>(base + 2 * S2Lds::XPLANE) = (uint16_t)(hh & 0xffffu);
      *reinterpret_cast<uint16_t*>(base + 3 * S2Lds::XPLANE) = (uint16_t)(hl & 0xffffu);
    }
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int u = tid + NTHR * i;
      const int q4 = u & 3, co = (u >> 2) & 31, cot = u >> 7;
      if (cot >= NCOT) continue;
      dsum[i] += (dv[K_][i][0] + dv[K_][i][1]) + (dv[K_][i][2] + dv[K_][i][3]);
      uint32_t h0, l0, h1, l1;
      wd_split2(dv[K_][i][0] * sdy, dv[K_][i][1] * sdy, h0, l0);
      wd_split2(dv[K_][i][2] * sdy, dv[K_][i][3] * sdy, h1, l1);
      unsigned char* base = dL + (size_t)cot * S2Lds::DYCOT + co * S2Lds::PITCH + 8 * q4;
      *reinterpret_cast<uint2*>(base) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(base + 32 * S2Lds::PITCH) = make_uint2(l0, l1);
    }
  };
  auto multiply = [&]() {
    const unsigned char* xb = xL + (size_t)krow * 4 * S2Lds::XPLANE + j * S2Lds::PITCH + 16 * h;
    const wd_u32x4 eh = *reinterpret_cast<const wd_u32x4*>(xb), el = *reinterpret_cast<const wd_u32x4*>(xb + S2Lds::XPLANE);
    const wd_u32x4 oh = *reinterpret_cast<const wd_u32x4*>(xb + 2 * S2Lds::XPLANE), ol = *reinterpret_cast<const wd_u32x4*>(xb + 3 * S2Lds::XPLANE);
    const uint32_t nh = *reinterpret_cast<const uint16_t*>(xb + 2 * S2Lds::XPLANE + 16), nl = *reinterpret_cast<const uint16_t*>(xb + 3 * S2Lds::XPLANE + 16);
    // kw = 2: the odd-column plane one entry further
    const wd_u32x4 sh = {__builtin_amdgcn_alignbit(oh[1], oh[0], 16), __builtin_amdgcn_alignbit(oh[2], oh[1], 16),
                         __builtin_amdgcn_alignbit(oh[3], oh[2], 16), __builtin_amdgcn_alignbit(nh, oh[3], 16)};
    const wd_u32x4 sl = {__builtin_amdgcn_alignbit(ol[1], ol[0], 16), __builtin_amdgcn_alignbit(ol[2], ol[1], 16),
                         __builtin_amdgcn_alignbit(ol[3], ol[2], 16), __builtin_amdgcn_alignbit(nl, ol[3], 16)};
    const h2_f16x8 Bh[3] = {__builtin_bit_cast(h2_f16x8, oh), __builtin_bit_cast(h2_f16x8, eh), __builtin_bit_cast(h2_f16x8, sh)};
    const h2_f16x8 Bl[3] = {__builtin_bit_cast(h2_f16x8, ol), __builtin_bit_cast(h2_f16x8, el), __builtin_bit_cast(h2_f16x8, sl)};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int cot = cotb + 2 * u;
      if (cot >= NCOT) continue;
      const unsigned char* db = dL + (size_t)cot * S2Lds::DYCOT + j * S2Lds::PITCH + 16 * h;
      const h2_f16x8 Ah = *reinterpret_cast<const h2_f16x8*>(db), Al = *reinterpret_cast<const h2_f16x8*>(db + 32 * S2Lds::PITCH);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f32x16 cc = acc[u][t];
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[t], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[t], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[t], cc, 0, 0, 0);
        acc[u][t] = cc;
      }
    }
  };

  // (not __syncthreads(): its fence waits for vmcnt(0), i.e. for the loads issued D steps ahead -- the whole point of the ring;
  // LDS traffic is what the two barriers of a step order, so lgkmcnt(0) + s_barrier)
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  constexpr int LPS = NX + 1 + NDY;                       // loads per thread and step
  static_assert((D - 1) * LPS < 64, "ring deeper than vmcnt counts");
  auto step = [&](int p, auto slot_c) {
    // this slot's loads have landed when at most the newer slots' are outstanding: D - 1 of them in the steady state, fewer
    // at the end of the range
    const int newer = min(D - 1, pe - 1 - p);
    if (D > 3 && newer == 3) s2_wait<(D > 3 ? 3 : 0) * LPS>();
    else if (D > 2 && newer == 2) s2_wait<(D > 2 ? 2 : 0) * LPS>();
    else if (D > 1 && newer == 1) s2_wait<(D > 1 ? 1 : 0) * LPS>();
    else s2_wait<0>();
    // The loads were inline assembly with "=v" outputs: to the compiler this slot's registers were valid from the load statement
    // on, so nothing in the source kept it from copying or spilling them BEFORE the wait above.  Re-defining them here ("+v": read
    // and written by an empty statement that sits after the wait and cannot be moved across it) makes the order explicit: every
    // use below depends on THIS statement, which depends on the wait.  No instruction is emitted.
    {
      constexpr int K_ = decltype(slot_c)::value;
#pragma unroll
      for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(xv[K_][i]) : : "memory");
#pragma unroll
      for (int i = 0; i < NDY; ++i) asm volatile("" : "+v"(dv[K_][i]) : : "memory");
      asm volatile("" : "+v"(xhalo[K_]) : : "memory");
    }
    lds_barrier();                                        // the previous step's fragment reads are done
    if (pro_form == 2) convert(p, std::integral_constant<int, 2>{}, slot_c);
    else if (pro_form == 1) convert(p, std::integral_constant<int, 1>{}, slot_c);
    else if (pro_form == 0) convert(p, std::integral_constant<int, 0>{}, slot_c);
    else convert(p, std::integral_constant<int, 3>{}, slot_c);
    lds_barrier();
    if (p + D < pe) prefetch(p + D, slot_c);              // into the slot just consumed: lands D - 1 steps from now
    if (cotb < NCOT) multiply();
  };
  if (pb + 0 < pe) prefetch(pb + 0, std::integral_constant<int, 0>{});
  if constexpr (D == 4) {
    if (pb + 1 < pe) prefetch(pb + 1, std::integral_constant<int, 1>{});
    if (pb + 2 < pe) prefetch(pb + 2, std::integral_constant<int, 2>{});
    if (pb + 3 < pe) prefetch(pb + 3, std::integral_constant<int, 3>{});
  }
  for (int p0 = pb; p0 < pe; p0 += D) {
    step(p0, std::integral_constant<int, 0>{});
    if constexpr (D == 4) {
      if (p0 + 1 < pe) step(p0 + 1, std::integral_constant<int, 1>{});
      if (p0 + 2 < pe) step(p0 + 2, std::integral_constant<int, 2>{});
      if (p0 + 3 < pe) step(p0 + 3, std::integral_constant<int, 3>{});
    }
  }

  // ---- partial slab of this split: [split][Coutp][9 * Ctot], k order (tap, ci) -- the direct kernel's layout
  const size_t KT = (size_t)9 * a.Ctot;
  float* slab = a.slabs + (size_t)split * a.Coutp * KT;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int cot = cotb + 2 * u;
    if (cot >= NCOT) continue;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cr = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[(size_t)cr * KT + (size_t)(krow * 3 + t) * a.Ctot + ci0 + j] = acc[u][t][r] * descale * descale2;
      }
  }
  if (blockIdx.y == 0) {
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int u = tid + NTHR * i;
      const int q4 = u & 3, co = (u >> 2) & 31, cot = u >> 7;
      float t = dsum[i];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      if (q4 == 0 && cot < NCOT) a.dshift[(size_t)split * a.Coutp + cot * 32 + co] = t;
    }
  }
}

// the LDS-staged form takes the stride-2 layers that are launched on their own (above the batching bound), with whole channel
// tiles; VUNET_TUNE_WGRAD_ROWSPLIT = 3 keeps them on the direct kernel (A/B, tests)
static bool s2_staged_ok(const vunet_wgrad_desc* d) {
  if (g_vunet_tune[VUNET_TUNE_WGRAD_ROWSPLIT] == 3) return false;
  if (!(d->flags & 2) || (d->flags & 1)) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1) return false;
  if (d->Ws != 2 * d->Wo || d->Hs != 2 * d->Ho || d->Wo % 16) return false;
  if (d->C1 % 32 || d->C2 % 32 || d->Cout % 32) return false;
  const int ncot = d->Cout / 32;
  if (ncot != 1 && ncot != 2 && ncot != 4) return false;
  // (the batching bound of vunet_conv2d_wgrad_batchable, conv_wgrad.hip: layers at or below it share a launch on the direct kernel.
  //  Both kernels accept any split count, so a bound moved by VUNET_WGRAD_BATCH_PIX only shifts which of the two runs a layer.)
  return (int64_t)d->N * d->Ho * d->Wo > 16384;
}

static int s2_staged_nslabs(const vunet_wgrad_desc* d) {
  const int ncit = (d->C1 + d->C2) / 32, pairs = d->N * d->Ho * (d->Wo / 16);
  int S = 512 / ncit;
  if (S > pairs / 4) S = pairs / 4;
  return S < 1 ? 1 : S;
}

// Several layers of one form in ONE launch (vunet_conv2d_wgrad_multi): the small-map layers' weight gradients are 20 - 30 us
// launches of ~2000 short waves each -- latency, not work -- and 60 of them per step interleaved with the data-gradient chain
// cost more than they overlap.  The layers' argument blocks travel by value (kernel arguments: the pointers change from step to
// step when the step is issued eagerly); a workgroup finds its layer by a scalar scan of the grid prefix sums.
constexpr int WD_MULTI = 12;
struct WgradDirectBatch {
  WgradDirectArgs it[WD_MULTI];
  int start[WD_MULTI + 1];   // first workgroup of layer i (start[n] = grid size)
  int gx[WD_MULTI];          // layer i's grid is gx[i] x gy[i], flattened x-fastest
  int n;
};
static_assert(sizeof(WgradDirectBatch) <= 4096, "kernel arguments: 4 KB");
template <int S, bool W4, int T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_direct_multi_kernel(const WgradDirectBatch b) {
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.start[i + 1]) ++i;
  WgradDirectArgs a = b.it[i];
  const int local = (int)blockIdx.x - b.start[i];
  wgrad_direct_body<S, W4, T>(a, local % b.gx[i], local / b.gx[i]);
}

// ---- host side ------------------------------------------------------------------------------------------------------
// geometry class: 0 not covered; else (S, W4, T) packed
static int direct_class(const vunet_wgrad_desc* d) {
  if (!(d->flags & 2) || (d->flags & 1)) return 0;   // the fp16 scheme only
  if (d->C1 <= 0 || d->Cout <= 0 || (d->C2 > 0 && d->C1 % 32 != 0)) return 0;
  if (d->Cout < 16 && d->KH == 3) return 0;          // (out_conv, 3 output channels: a 32-row tile would multiply 90 % zeros)
  if (d->KH == 1 && d->KW == 1) {
    if (d->stride != 1 || d->pad != 0 || d->Ho != d->Hs || d->Wo != d->Ws || (d->Hs * d->Ws) % 8) return 0;
    return 1;
  }
  if (d->KH != 3 || d->KW != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2)) return 0;
  if (d->Ho != (d->Hs - 1) / d->stride + 1 || d->Wo != (d->Ws - 1) / d->stride + 1) return 0;
  if (d->Ws % 4) return 0;                           // 16-byte row loads
  if (d->Wo == 4 && d->Ws == 4 * d->stride && d->Ho % 2 == 0) return d->stride == 1 ? 2 : 3;
  if (d->Wo % 8 == 0 && d->Ws == d->Wo * d->stride) return d->stride == 1 ? 4 : 5;
  return 0;
}

bool vunet_wgrad_direct_applicable(const vunet_wgrad_desc* d) { return direct_class(d) != 0; }

static void direct_geometry(const vunet_wgrad_desc* d, WgradDirectArgs& a) {
  const int cls = direct_class(d);
  a.Ctot = d->C1 + d->C2;
  a.Coutp = (d->Cout + 31) / 32 * 32;
  a.ncot = a.Coutp / 32;
  if (cls == 1) { a.opr = 0; a.opi = d->Hs * d->Ws / 8; }
  else if (cls == 2 || cls == 3) { a.opr = 0; a.opi = d->Ho / 2; }
  else { a.opr = d->Wo / 8; a.opi = d->Ho * a.opr; }
  a.noct = d->N * a.opi;
}

// Pixel splits: about 2048 waves in all (eight per CU), every wave at least four K steps (eight octets) where the
// problem has them; few splits for the tiny maps (a slab is written per split).
// kernel-row split (three waves per (split, co tile), one kernel row each): every 3x3 layer on maps >= 8 wide.  Round 4 split
// the stride-2 layers only; same-box A/B runs of the whole step at the end of round 5 (profiles/r05_rowsplit_ab.txt: five
// alternating pairs on two boxes) put the stride-1 direct layers on it too: + 0.4 - 0.7 %.
// VUNET_TUNE_WGRAD_ROWSPLIT: 1 = never, 2 = (the default, kept as a value for tests), 3 = the large stride-2 layers on the direct
// kernel instead of the LDS-staged one, 4 = the stride-2 layers only (round 4's choice)
static bool direct_rowsplit(int cls) {
  const int knob = g_vunet_tune[VUNET_TUNE_WGRAD_ROWSPLIT];
  if (knob == 1) return false;
  return cls == 5 || (knob != 4 && cls == 4);
}

int vunet_wgrad_direct_nslabs(const vunet_wgrad_desc* d) {
  if (s2_staged_ok(d)) return s2_staged_nslabs(d);
  WgradDirectArgs a;
  direct_geometry(d, a);
  const int ncit = (a.Ctot + 31) / 32;
  int S = 2048 / (a.ncot * ncit * (direct_rowsplit(direct_class(d)) ? 3 : 1));
  const int by_work = (a.noct + 7) / 8;
  if (S > by_work) S = by_work;
  if (S < 1) S = 1;
  const int cap = d->KH == 1 ? 2048 : 512;   // (a 1x1 slab is one 4 KB tile: many splits cost nothing)
  if (S > cap) S = cap;
  return S;
}

int vunet_wgrad_direct_name(const vunet_wgrad_desc* d, char* name, int len) {
  if (s2_staged_ok(d)) return snprintf(name, len, "conv_wgrad_s2_kernel");
  const int cls = direct_class(d);
  const int S = (cls == 3 || cls == 5) ? 2 : 1, W4 = (cls == 2 || cls == 3), T = cls == 1 ? 1 : (direct_rowsplit(cls) ? 3 : 9);
  return snprintf(name, len, "conv_wgrad_direct_kernel<%d, %s, %d>", S, W4 ? "true" : "false", T);
}

// the kernel's argument block and grid of one layer
static void direct_fill(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs, float* dshift,
                        const float* amax_x, const float* amax_x2, const float* amax_dy, WgradDirectArgs& a, int& gx, int& gy) {
  a.d = *d;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.slabs = slabs; a.dshift = dshift; a.amax_x = amax_x; a.amax_x2 = amax_x2; a.amax_dy = amax_dy;
  direct_geometry(d, a);
  a.ops = (a.noct + d->nsplit - 1) / d->nsplit;
  a.ops = (a.ops + 1) & ~1;   // whole K steps of two octets
  a.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  a.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  const int units = d->nsplit * a.ncot * (direct_rowsplit(direct_class(d)) ? 3 : 1);
  gx = (units + 3) / 4;
  gy = (a.Ctot + 31) / 32;
}

// form of a layer's kernel: class * 2 + (kernel-row split)
static int direct_form(const vunet_wgrad_desc* d) {
  const int cls = direct_class(d);
  return cls ? cls * 2 + (direct_rowsplit(cls) ? 1 : 0) : 0;
}

template <int S, bool W4, int T>
static void direct_launch_batch(const WgradDirectBatch& b, hipStream_t st) {
  VUNET_LAUNCH((conv_wgrad_direct_multi_kernel<S, W4, T>), dim3((unsigned)b.start[b.n]), dim3(256), 0, st, b);
}

// items[i] for i in idx[0..cnt): all of one form
int vunet_wgrad_direct_launch_multi(const vunet_wgrad_item* items, const int* idx, int cnt, hipStream_t st) {
  if (cnt < 1) return VUNET_OK;
  const int form = direct_form(&items[idx[0]].d);
  if (!form) return VUNET_ERR_UNSUPPORTED;
  for (int at = 0; at < cnt; at += WD_MULTI) {
    WgradDirectBatch b;
    b.n = cnt - at < WD_MULTI ? cnt - at : WD_MULTI;
    b.start[0] = 0;
    for (int k = 0; k < b.n; ++k) {
      const vunet_wgrad_item& it = items[idx[at + k]];
      if (direct_form(&it.d) != form) return VUNET_ERR_ARG;
      if (!it.x1 || !it.dy || !it.slabs || !it.dshift || !it.amax_x || !it.amax_dy || (it.d.C2 > 0 && !it.x2)) return VUNET_ERR_ARG;
      int gx, gy;
      direct_fill(&it.d, it.x1, it.x2, it.dy, it.slabs, it.dshift, it.amax_x, it.amax_x2, it.amax_dy, b.it[k], gx, gy);
      b.gx[k] = gx;
      b.start[k + 1] = b.start[k] + gx * gy;
    }
    for (int k = b.n; k < WD_MULTI; ++k) { b.it[k] = b.it[0]; b.gx[k] = 1; b.start[k + 1] = b.start[b.n]; }
    switch (form) {
      case 2: direct_launch_batch<1, false, 1>(b, st); break;
      case 4: direct_launch_batch<1, true, 9>(b, st); break;
      case 6: direct_launch_batch<2, true, 9>(b, st); break;
      case 8: direct_launch_batch<1, false, 9>(b, st); break;
      case 9: direct_launch_batch<1, false, 3>(b, st); break;
      case 10: direct_launch_batch<2, false, 9>(b, st); break;
      case 11: direct_launch_batch<2, false, 3>(b, st); break;
      default: return VUNET_ERR_UNSUPPORTED;
    }
    const int rc = vunet_check_launch();
    if (rc != VUNET_OK) return rc;
  }
  return VUNET_OK;
}

int vunet_wgrad_direct_form(const vunet_wgrad_desc* d) { return direct_form(d); }

int vunet_wgrad_direct_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs,
                              float* dshift, const float* amax_x, const float* amax_x2, const float* amax_dy, hipStream_t st) {
  const int cls = direct_class(d);
  if (!cls) return VUNET_ERR_UNSUPPORTED;
  WgradDirectArgs a;
  int gx, gy;
  direct_fill(d, x1, x2, dy, slabs, dshift, amax_x, amax_x2, amax_dy, a, gx, gy);
  if (s2_staged_ok(d)) {
    const int ncot = d->Cout / 32;
    dim3 grid((unsigned)d->nsplit, (unsigned)((d->C1 + d->C2) / 32));
    const size_t lds = (size_t)S2Lds::XBYTES + (size_t)ncot * S2Lds::DYCOT;
    if (ncot == 1) VUNET_LAUNCH((conv_wgrad_s2_kernel<3, 1, 1>), grid, dim3(192), lds, st, a);
    else if (ncot == 2) VUNET_LAUNCH((conv_wgrad_s2_kernel<6, 1, 2>), grid, dim3(384), lds, st, a);
    else VUNET_LAUNCH((conv_wgrad_s2_kernel<6, 2, 4>), grid, dim3(384), lds, st, a);
    return vunet_check_launch();
  }
  const bool rs = direct_rowsplit(cls);
  dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  if (rs) {
    if (cls == 5) VUNET_LAUNCH((conv_wgrad_direct_kernel<2, false, 3>), grid, block, 0, st, a);
    else VUNET_LAUNCH((conv_wgrad_direct_kernel<1, false, 3>), grid, block, 0, st, a);
    return vunet_check_launch();
  }
  switch (cls) {
    case 1: VUNET_LAUNCH((conv_wgrad_direct_kernel<1, false, 1>), grid, block, 0, st, a); break;
    case 2: VUNET_LAUNCH((conv_wgrad_direct_kernel<1, true, 9>), grid, block, 0, st, a); break;
    case 3: VUNET_LAUNCH((conv_wgrad_direct_kernel<2, true, 9>), grid, block, 0, st, a); break;
    case 4: VUNET_LAUNCH((conv_wgrad_direct_kernel<1, false, 9>), grid, block, 0, st, a); break;
    default: VUNET_LAUNCH((conv_wgrad_direct_kernel<2, false, 9>), grid, block, 0, st, a); break;
  }
  return vunet_check_launch();
}
