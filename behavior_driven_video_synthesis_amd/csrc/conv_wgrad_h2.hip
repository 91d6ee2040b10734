// Weight gradient of the 3x3 / stride 1 / pad 1 convolution, fp32-accurate on the fp16 matrix cores ("h2" scheme).
//
//   dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap]          (K = pixels)
//
// Structure of conv_wgrad_x6.hip (A = dy from registers, B = f(x) staged once per element into LDS with the +-1 column
// fragments built by v_alignbit_b32, split-K over pixel tiles, slabs summed by vunet_weightnorm_bwd) with the operand
// split of conv_h2_kernel.h: both operands scaled by a power of two from their tensor maxima (vunet_absmax_partials) and
// split into two fp16 terms,  s v = h + l,  h = f16(s v), l = f16(s v - h), and THREE partial products per tap
// (Ah Bl, Al Bh, Ah Bh) instead of six.  One accumulator per tap (nine per wave) leaves no room for a second set, so the
// low term keeps its natural scale here: it is exact down to |s v| = 2^-3 and loses relative precision gradually below
// that, i.e. for elements more than 2^16 below their tensor's maximum -- whose contribution to a sum over all pixels
// of the batch is far below the sum's own fp32 rounding (tests/test_hip_x6.py, wide-range case).
//
// Round 4 -- two groups in antiphase.  Timing ablations of the round-3 kernel (profiles/r04_time_wgrad_ablations.txt: the
// 128-channel 128^2 layer, 332 us) came out ADDITIVE: f(x) staged once per workgroup 231 us, dy loaded once 302, no MFMA
// block 156 -- nothing overlapped.  The two workgroups of a CU start together and run the same phases of the same length,
// so they load, convert and multiply in lockstep; the matrix phase itself runs at the rate the chip sustains (~0.6 of the
// fp16 peak on random data, profiles/r04_mfma_shape_probe.jsonl) but only half of the time.  Copying the next raw tile to
// LDS by LDS-DMA behind the MFMA block (no registers) made it SLOWER (383 us: the latency was not the cost, the serial
// phases were).  A workgroup is now 512 threads = two groups of four waves, each with its own staged tile, working on
// alternate pixel tiles half a step apart: while group 0 multiplies its tile, group 1 loads, converts and writes its next
// one, and vice versa -- one barrier per half step.  A SIMD holds one wave of each group (waves w and w + 4), so its
// matrix pipe always has a wave in the MFMA phase beside one in the staging phase; the phases overlap by construction,
// not by luck.  The groups' accumulators meet in the fixed-order fold at the end.  -DWG_H2_ONE_GROUP: the round-3 shape.
// With the phases in antiphase the exposed latencies are what is left (tools/wgrad_timeline.py, profiles/r04_wgrad_timeline.txt:
// f(x) loads 2.0 us, conversion 1.5 - 2.3, dy loads 1.4 - 2.0, the 108 MFMAs 2.4 - 2.7), so both load streams left the phase
// that consumes them.  The raw fp32 tile of a group's NEXT step is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4 /
// _dword: no registers) at the END of its staging phase and has the whole multiply phase to land; the staging phase is then
// a conversion from LDS.  Every wave copies exactly the 6 x 64 units (+ halo columns) its own lanes convert -- the DMA
// destination is the conversion's read order, [wave][unit i][half][lane], linear and conflict-free -- so the raw buffer
// needs no barrier of its own: a wave retires its copy (s_waitcnt vmcnt) at the top of its next staging phase and overwrites
// it only after it has consumed it.  dy is requested at the top of the staging phase and lands during the conversion: the
// multiply phase starts with everything in registers / LDS and is the 108 MFMAs.  -DWG_H2_NO_DMA: register staging (A/B).
//
// Edge plane (round 4).  The +-1 column fragments need, per (row, ci, octet), the pixel left of the octet and the pixel right
// of it.  Rounds 2-3 read them as the last dword of the previous unit and the first dword of the next one: 64 lanes reading
// the same dword position of 16-byte-aligned units, 5 units apart, land on the 16 banks = 0 (mod 4) -- two 4-way conflicted
// ds_read_b32 per plane and step (SQ_LDS_BANK_CONFLICT 9.2e6 cycles per launch, profiles/r03_pmc_sq.json).  They now come
// from a plane of their own, eL[plane][row][octet][ci] = (right neighbour | left neighbour << 16): ONE ds_read_b32 per plane
// and step, the 64 lanes (ci, octet half) on 64 consecutive dwords.  Staging writes the two halves as 16-bit stores when it
// writes the unit they belong to (a unit's first pixel is the right neighbour of the octet before it, its last the left
// neighbour of the octet behind it; the tile's halo columns fill octets 0 / 3).  Timing: +-2 % (the conflicts were never
// what bounded the kernel).  SQ_LDS_BANK_CONFLICT per launch 9.2e6 -> 4.2e6 (<1>) and 6.4e6 -> 1.7e6 (<2>) cycles
// (profiles/r04_pmc_sq.json): the fragment reads are conflict-free; what is left are the staging WRITES (the 16-bit edge
// stores, 2-way, and the 16-byte unit stores across the 5-unit entry stride), a ninth as frequent as the reads.
#include <type_traits>

#include "common.h"
#include "split_h2.h"

typedef h2_f16x8 wg_bf16x8;   // (the fragment type of this file: eight fp16)
typedef uint32_t wg_u32x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

// (diagnostic, tools/ab_build.sh -DWG_TSTAMP=<export name>: wave 0 of each group stamps the 100 MHz wall clock at the phase
// boundaries of its middle step; tools/wgrad_timeline.py reads them back.  Results are unchanged.)
#ifdef WG_TSTAMP
static __device__ unsigned long long wg_ts[4096 * 16];
#define WG_STAMP(k_)                                                                                                     \
  do {                                                                                                                   \
    if (lane == 0 && wave == 0 && step == (nsteps >> 1)) {                                                                           \
      const unsigned b_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                                \
      if (b_ < 4096) wg_ts[b_ * 16 + grp * 8 + (k_)] = __builtin_amdgcn_s_memrealtime();                                 \
    }                                                                                                                    \
  } while (0)
extern "C" int WG_TSTAMP(unsigned long long* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(wg_ts), sizeof(unsigned long long) * 16 * (size_t)nblocks) == hipSuccess ? 0 : -1;
}
#define WG_STAMP_AT(slot_)                                                                                               \
  do {                                                                                                                   \
    if (threadIdx.x == 0) {                                                                                              \
      const unsigned b_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                                \
      if (b_ < 4096) wg_ts[b_ * 16 + (slot_)] = __builtin_amdgcn_s_memrealtime();                                        \
    }                                                                                                                    \
  } while (0)
#else
#define WG_STAMP(k_)
#define WG_STAMP_AT(slot_)
#endif

// One LDS-DMA wave-instruction (see conv_h2_kernel.h: h2_dma16): 64 lanes x 16 (4) bytes from per-lane global addresses
// to 1 KiB (256 B) of LDS at the wave-uniform byte address `lds_addr`; inline assembly (the compiler does not track it: the
// kernel retires it itself with s_waitcnt vmcnt), M0 saved and restored.
__device__ __forceinline__ void wg_dma16(const void* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void wg_dma4(const void* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}

// The prologue f(x) runs as straight-line code (common.h: in_act_form): the generic form branches on the activation code and
// the dropout threshold PER ELEMENT -- 358 scalar branches in the round-3 kernel, whose conversion of one tile took 3.0 - 3.4 us
// against 2.6 us for the tile's 108 MFMAs (tools/wgrad_timeline.py, profiles/r04_wgrad_timeline.txt); the staging code is
// instantiated per form behind ONE wave-uniform branch per tile.
struct WgradH2Args {
  vunet_wgrad_desc d;
  const float* x1;
  const float* x2;
  const float* dy;
  float* slabs;
  float* dshift;
  int Ctot, Coutp, tiles_per_img_w, tiles_per_img, ntiles, tps;
  InAct in1, in2;
  const float* amax_x;    // 1024 partial maxima of |x1|, |x2|
  const float* amax_x2;   // or: the second source's 512 in a buffer of their own (split_h2.h: h2_amax4)
  const float* amax_dy;   // 1024 partial maxima of |dy| (second half zeros)
};

// (a, b), already scaled -> packed fp16 pairs of the two terms, the low one at its natural scale
__device__ __forceinline__ void wg_split2(float a, float b, uint32_t& h, uint32_t& l) {
  h2_f16x2 ph, pl;
  ph[0] = (_Float16)a;
  ph[1] = (_Float16)b;
  pl[0] = (_Float16)(a - (float)ph[0]);
  pl[1] = (_Float16)(b - (float)ph[1]);
  h = __builtin_bit_cast(uint32_t, ph);
  l = __builtin_bit_cast(uint32_t, pl);
}

__device__ __forceinline__ wg_bf16x8 wg_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  wg_u32x4 u = {a, b, c, d};
  return __builtin_bit_cast(wg_bf16x8, u);
}

#ifdef WG_H2_ONE_GROUP
constexpr int WG_GROUPS = 1;
#else
constexpr int WG_GROUPS = 2;
#endif

template <int MTW>
__global__ __launch_bounds__(256 * WG_GROUPS, WG_GROUPS == 2 ? 1 : 2) void conv_wgrad_h2_kernel(const WgradH2Args a_in) {
  WgradH2Args a = a_in;
  WG_STAMP_AT(7);    // (slot 7 of group 0: kernel entry; slot 15: the tile loop is over; slot 14: exit)
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  constexpr int TH = 4;
  // optional register prefetch of the next tile's f(x) across the MFMA block
#ifdef WG_H2_PREFETCH
  constexpr bool PREFETCH = true;
#else
  constexpr bool PREFETCH = false;   // measured: no gain from the register prefetch (r02), and two m-tiles spill with it
#endif
  constexpr int WP = 4 / MTW;        // waves sharing one m-tile (they split the tile rows)
  constexpr int RW = TH / WP;        // tile rows per wave
  constexpr int ENT = 6 * 32;        // (row, ci) entries of the staged tile
  constexpr int PL = ENT * 5 + 1;    // units per plane (+1: the halo unit behind the last entry)
  constexpr int EPL = 6 * 4 * 32;    // dwords per plane of the edge plane: (row, octet, ci)
  // The raw-tile DMA keeps dy live across the conversion: one m-tile per workgroup (RW = 1: 16 dy registers) has the room,
  // two m-tiles (RW = 2: 32) spill 18 registers in the staging phase and lose what the copy gains (same box, sustained
  // launches, profiles/r04_time_wgrad.txt: Cout = 32 at 256^2 112 -> 96 us with the copy; 64 channels at 256^2 333 -> 366 us,
  // 64 + 64 -> 64 at 128^2 136 -> 160 us): DMA for the one-m-tile form only.  -DWG_H2_NO_DMA / -DWG_H2_DMA_ALL: A/B.
#if defined(WG_H2_NO_DMA)
  constexpr bool DMA = false;
#elif defined(WG_H2_DMA_ALL)
  constexpr bool DMA = true;
#else
  constexpr bool DMA = MTW == 1;
#endif
  constexpr int SU = 2 * PL + 2 * EPL / 4;   // 16-byte units of one group's staged tile (two planes + two edge planes)
  constexpr int RU = ENT * 8 + 2 * ENT / 4;  // ... of its raw fp32 tile [48 groups][2 halves][16] + the halo columns [2][ENT]
  constexpr int GU = SU + (DMA ? RU : 0);
  constexpr int G = WG_GROUPS;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

  const vunet_wgrad_desc& d = a.d;
  const int grp = threadIdx.x >> 8;                    // group of four waves (0 / 1)
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // thread / wave index INSIDE the group
  wg_u32x4* const xL = reinterpret_cast<wg_u32x4*>(smem4 + grp * GU);   // this group's tile: [2][PL]
  uint32_t* const eL = reinterpret_cast<uint32_t*>(smem4 + grp * GU + 2 * PL);   // [2][EPL]
  const wg_f32x4* const rawL = reinterpret_cast<const wg_f32x4*>(smem4 + grp * GU + SU);
  const float* const haloL = reinterpret_cast<const float*>(smem4 + grp * GU + SU + ENT * 8);
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave / WP, pw = wave % WP;
  const int split = blockIdx.x;
  const int ci0 = blockIdx.y * 32;
  const int co0 = blockIdx.z * 32 * MTW;
  const int H = d.Hs, W = d.Ws;

  // ---- operand scales (powers of two) from the tensor maxima
  float sx, sdy, descale, descale2;
  {
    // (every group reduces all 1024 + 1024 partial maxima: the same values in both)
    const float4 px = h2_amax4(a.amax_x, a.amax_x2, tid), pd = reinterpret_cast<const float4*>(a.amax_dy)[tid];
    float mx = wave_max(fmaxf(fmaxf(px.x, px.y), fmaxf(px.z, px.w)));
    float md = wave_max(fmaxf(fmaxf(pd.x, pd.y), fmaxf(pd.z, pd.w)));
    float* const redm = reinterpret_cast<float*>(smem4) + grp * 8;
    if (lane == 0) { redm[wave] = mx; redm[4 + wave] = md; }
    __syncthreads();
    mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    md = fmaxf(fmaxf(redm[4], redm[5]), fmaxf(redm[6], redm[7]));
    __syncthreads();
    if (a.in1.thresh) mx *= a.in1.keep_scale;
    const int ex = h2_scale_exp(mx), ed = h2_scale_exp(md);
    sx = h2_pow2(ex);
    sdy = h2_pow2(ed);
    h2_pow2_pair(-(ex + ed), descale, descale2);
  }

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
  // DMA path: the raw tile travels global -> LDS without registers.  Wave w copies what its own lanes convert: for unit
  // i = 0..2 (u = tid + 256 i -> row, ci, octet) the two 16-byte halves of the octet, 64 lanes at a time, to
  // raw[((w * 3 + i) * 2 + half) * 64 + lane]; waves 0..2 also the halo columns of their 64 (row, ci) entries.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t raw_addr = (uint32_t)__builtin_amdgcn_readfirstlane(   // (wave-uniform: it goes to M0)
      (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(smem4 + grp * GU + SU));
  auto dma_x = [&](int tile) {
    int n, row0, col0;
    tile_origin(tile, n, row0, col0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ih = row0 - 1 + s_row[i];
      const bool ok = (unsigned)ih < (unsigned)H;
      const float* src = xs + (size_t)((n * Cs + cbase + s_ci[i]) * H + (ok ? ih : 0)) * W + col0 + 8 * s_oct[i];
      wg_dma16(src, raw_addr + (uint32_t)(((wave_u * 3 + i) * 2 + 0) * 64) * 16u);
      wg_dma16(src + 4, raw_addr + (uint32_t)(((wave_u * 3 + i) * 2 + 1) * 64) * 16u);
    }
    if (wave_u < 3) {   // entries tid = 64 wave + lane < ENT
      const int ih = row0 - 1 + (tid >> 5);
      const bool rok = (unsigned)ih < (unsigned)H;
      const float* base = xs + (size_t)((n * Cs + cbase + (tid & 31)) * H + (rok ? ih : 0)) * W + col0;
      wg_dma4(base + ((rok && col0 > 0) ? -1 : 0), raw_addr + (uint32_t)(ENT * 8) * 16u + (uint32_t)(wave_u * 64) * 4u);
      wg_dma4(base + ((rok && col0 + 32 < W) ? 32 : 0), raw_addr + (uint32_t)(ENT * 8) * 16u + (uint32_t)(ENT + wave_u * 64) * 4u);
    }
  };
  int f_n = 0, f_row0 = 0, f_col0 = 0;   // origin of the tile being converted (DMA path)
  auto fetch_unit = [&](int i) {
    const int ih = f_row0 - 1 + s_row[i];
    const bool ok = (unsigned)ih < (unsigned)H;
    okbits = (okbits & ~(1u << i)) | ((ok ? 1u : 0u) << i);
    gidx[i] = (uint32_t)(((f_n * Cs + cbase + s_ci[i]) * H + (ok ? ih : 0)) * W + f_col0 + 8 * s_oct[i]);
    xv[i][0] = rawL[((wave * 3 + i) * 2 + 0) * 64 + lane];
    xv[i][1] = rawL[((wave * 3 + i) * 2 + 1) * 64 + lane];
  };
  auto fetch_halo = [&]() {
    const int e = tid < ENT ? tid : tid - 64;   // (the last wave holds no entry: it reads a neighbour's, and drops it)
    const int r = e >> 5, c = e & 31;
    const int ih = f_row0 - 1 + r;
    const bool rok = (unsigned)ih < (unsigned)H;
    hidx = (uint32_t)(((f_n * Cs + cbase + c) * H + (rok ? ih : 0)) * W + f_col0);
    const bool okl = rok && f_col0 > 0, okr = rok && f_col0 + 32 < W;
    okbits = (okbits & 7u) | (okl ? 16u : 0u) | (okr ? 32u : 0u);
    hv[0] = haloL[e];
    hv[1] = haloL[ENT + e];
  };
  auto write_x_as = [&](auto pro_c, auto from_lds_c) {
    constexpr int PRO = decltype(pro_c)::value;
    constexpr bool FROM_LDS = decltype(from_lds_c)::value;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if constexpr (FROM_LDS) fetch_unit(i);   // (one unit's eight values live at a time: nine accumulator tiles leave no room for three)
      const bool ok = (okbits >> i) & 1u;
      // out-of-range rows: the value is selected to 0 BEFORE the prologue and the scale is 0 as well -- written as
      // "ok ? f(t) * sx : 0" the compiler wraps every element's prologue in an exec-mask block of its own
      const float sxk = ok ? sx : 0.f;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = e < 4 ? xv[i][0][e] : xv[i][1][e - 4];
        f[e] = in_act_form<PRO>(ia, ok ? t : 0.f, gidx[i] + e) * sxk;
      }
      uint32_t ph[4], pl[4];
      wg_split2(f[0], f[1], ph[0], pl[0]);
      wg_split2(f[2], f[3], ph[1], pl[1]);
      wg_split2(f[4], f[5], ph[2], pl[2]);
      wg_split2(f[6], f[7], ph[3], pl[3]);
      const int unit = (s_row[i] * 32 + s_ci[i]) * 5 + 1 + s_oct[i];
      xL[unit] = wg_u32x4{ph[0], ph[1], ph[2], ph[3]};
      xL[PL + unit] = wg_u32x4{pl[0], pl[1], pl[2], pl[3]};
      // this unit's first pixel is the right neighbour of octet o - 1 (low half), its last the left neighbour of o + 1 (high)
      unsigned short* const eb = reinterpret_cast<unsigned short*>(eL);
      const int e0 = (s_row[i] * 4 + s_oct[i]) * 32 + s_ci[i];
      if (s_oct[i] > 0) {
        eb[2 * (e0 - 32)] = (unsigned short)(ph[0] & 0xffffu);
        eb[2 * (EPL + e0 - 32)] = (unsigned short)(pl[0] & 0xffffu);
      }
      if (s_oct[i] < 3) {
        eb[2 * (e0 + 32) + 1] = (unsigned short)(ph[3] >> 16);
        eb[2 * (EPL + e0 + 32) + 1] = (unsigned short)(pl[3] >> 16);
      }
    }
    if constexpr (FROM_LDS) fetch_halo();
    if (tid < ENT) {
      const bool okl = okbits & 16u, okr = okbits & 32u;
      const float fl = in_act_form<PRO>(ia, okl ? hv[0] : 0.f, hidx - 1) * (okl ? sx : 0.f);
      const float fr = in_act_form<PRO>(ia, okr ? hv[1] : 0.f, hidx + 32) * (okr ? sx : 0.f);
      uint32_t ph, pl;
      wg_split2(fr, fl, ph, pl);   // low half = right halo, high = left
      unsigned short* const eb = reinterpret_cast<unsigned short*>(eL);
      const int r = tid >> 5, c = tid & 31;
      eb[2 * ((r * 4 + 0) * 32 + c) + 1] = (unsigned short)(ph >> 16);           // left of octet 0: column -1
      eb[2 * (EPL + (r * 4 + 0) * 32 + c) + 1] = (unsigned short)(pl >> 16);
      eb[2 * ((r * 4 + 3) * 32 + c)] = (unsigned short)(ph & 0xffffu);            // right of octet 3: column 32
      eb[2 * (EPL + (r * 4 + 3) * 32 + c)] = (unsigned short)(pl & 0xffffu);
    }
  };
  // the form of this layer's prologue (wave-uniform, the same for every tile)
  const int pro_form = in_act_form_of(ia);
  auto write_x = [&]() {
    constexpr std::integral_constant<bool, DMA> from_lds{};
    if (pro_form == 2) write_x_as(std::integral_constant<int, 2>{}, from_lds);
    else if (pro_form == 1) write_x_as(std::integral_constant<int, 1>{}, from_lds);
    else if (pro_form == 0) write_x_as(std::integral_constant<int, 0>{}, from_lds);
    else write_x_as(std::integral_constant<int, 3>{}, from_lds);
  };

  // Tiles t_begin + grp, + G, ...: the groups alternate.  Both groups run the SAME loop -- stage, barrier, multiply,
  // barrier -- group 1 half a step behind: it passes one barrier before its loop, group 0 one after (a barrier releases
  // when all eight waves have arrived at A barrier, not at the same one), so that between any two barriers one group
  // stages while the other multiplies.  Every wave executes 2 nsteps + 1 barriers.
  wg_f32x4 dyv[RW][2][2];
  const wg_u32x4* const xl_lane = xL + (pw * RW * 32 + j) * 5 + 1 + h;     // this lane's fragment unit of tile row pw * RW
  const uint32_t* const el_lane = eL + (pw * RW * 4 + h) * 32 + j;
  const int nsteps = (t_end - t_begin + G - 1) / G;
  // dy of a wave's rows: RW rows x 2 k-steps x 8 pixels per lane, straight to registers (4 RW load instructions)
  auto load_dy = [&](int tile) {
    int n, row0, col0;
    tile_origin(tile, n, row0, col0);
    const float* __restrict__ dp = a.dy + ((size_t)(n * d.Cout + co_lane) * H + row0 + pw * RW) * W + col0 + 8 * h;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dyv[rr][s][0] = *reinterpret_cast<const wg_f32x4*>(dp + rr * W + 16 * s);
        dyv[rr][s][1] = *reinterpret_cast<const wg_f32x4*>(dp + rr * W + 16 * s + 4);
      }
  };
  if constexpr (DMA) {   // the group's first tile: the copy is retired at the top of the first staging phase
    if (t_begin + grp < t_end) dma_x(t_begin + grp);
  }
  if (G == 2 && grp == 1) __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int mtile = t_begin + step * G + grp;          // this group's tile of this step
    const bool valid = mtile < t_end;
    const bool more = mtile + G < t_end;                 // ... and there is a next one
    WG_STAMP(0);
    // ---- staging phase: the tile's f(x) -> two fp16 planes + edge plane
    if (valid) {
#ifdef WG_ABL_NOX   // (timing ablations for tools/ab_build.sh: results are WRONG with any WG_ABL_* defined)
      if (step == 0 || descale == 12345.f) {
#endif
      if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copy of the tile (issued a phase ago) has landed
        load_dy(mtile);                                      // lands during the conversion
        tile_origin(mtile, f_n, f_row0, f_col0);
      } else {
        issue_x(mtile);
#ifdef WG_TSTAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      }
      WG_STAMP(1);
      write_x();
      if constexpr (DMA) {
        // dy retired HERE (naming its youngest register makes the compiler place its wait): the copy below is younger, and
        // the compiler's own waits further down would otherwise wait for it as well, in front of the MFMA block
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[RW - 1][1][1]) :: "memory");
        if (more) dma_x(mtile + G);                          // this wave is done with its raw units: the next tile's
      }
#ifdef WG_ABL_NOX
      }
#endif
    }
    WG_STAMP(2);
    __syncthreads();
    WG_STAMP(3);
    // ---- multiply phase
    if (valid) {
    if constexpr (!DMA) load_dy(mtile);
#ifdef WG_TSTAMP
    if constexpr (!DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(4);
#endif

    // ---- MFMA: this wave's RW rows x 2 k-steps x 3 kernel rows = RW*6 steps of 9 MFMAs (3 taps x 3 products).
    //      The f(x) reads of step i+1 (one aligned 16-byte read + the two neighbouring edge dwords, per plane) are
    //      issued before the MFMAs of step i: LDS latency hides behind 9 MFMAs instead of stalling every step.
    struct Raw {
      wg_u32x4 c[2];
      uint32_t e[2];   // (right neighbour | left neighbour << 16) of this lane's octet
    };
    auto load_raw = [&](int st) {
      const int rr = st / 6, s = (st / 3) & 1, kh = st % 3;
      // unit of (row pw*RW + rr + kh, ci j, octet 2s + h): this lane's base + a compile-time offset
      Raw r;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        r.c[p] = xl_lane[p * PL + (rr + kh) * 160 + 2 * s];
        r.e[p] = el_lane[p * EPL + ((rr + kh) * 4 + 2 * s) * 32];
      }
      return r;
    };
    constexpr int NSTEP = RW * 6;
    Raw cur = load_raw(0);
    wg_bf16x8 Ah, Al;
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
        uint32_t ah[4], al[4];
        wg_split2(d0[0] * sdy, d0[1] * sdy, ah[0], al[0]);
        wg_split2(d0[2] * sdy, d0[3] * sdy, ah[1], al[1]);
        wg_split2(d1[0] * sdy, d1[1] * sdy, ah[2], al[2]);
        wg_split2(d1[2] * sdy, d1[3] * sdy, ah[3], al[3]);
        Ah = wg_frag(ah[0], ah[1], ah[2], ah[3]);
        Al = wg_frag(al[0], al[1], al[2], al[3]);
      }
      // one plane of f(x) at a time, smallest terms first: only three shifted fragments are live
      //   plane l: Ah*Bl ; plane h: Al*Bh, Ah*Bh
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int p = 1 - pp;
        const wg_u32x4 c = cur.c[p];
        const uint32_t s01 = __builtin_amdgcn_alignbit(c.y, c.x, 16), s12 = __builtin_amdgcn_alignbit(c.z, c.y, 16),
                       s23 = __builtin_amdgcn_alignbit(c.w, c.z, 16);
        wg_bf16x8 B[3];
        B[0] = wg_frag(__builtin_amdgcn_alignbit(c.x, cur.e[p], 16), s01, s12, s23);   // columns p - 1
        B[1] = __builtin_bit_cast(wg_bf16x8, c);                                        // columns p
        B[2] = wg_frag(s01, s12, s23, __builtin_amdgcn_alignbit(cur.e[p], c.w, 16));   // columns p + 1
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          f32x16 cc = acc[kh * 3 + kw];
          if (p == 0) cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, B[kw], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, B[kw], cc, 0, 0, 0);
          acc[kh * 3 + kw] = cc;
        }
      }
      cur = nxt;
    }
    WG_STAMP(5);
    }   // valid
    __syncthreads();
    WG_STAMP(6);
  }
  if (G == 2 && grp == 0) __syncthreads();
  WG_STAMP_AT(15);

  // ---- fold the row-part waves (and the second group) of each m-tile into the (group 0, pw == 0) wave through LDS in a
  //      fixed order, one tap at a time (16 KiB per group), so that one slab per split leaves the workgroup
  {
    float* const red = reinterpret_cast<float*>(smem4);  // [G groups][4 waves][16][64]
    const bool owner = pw == 0 && grp == 0;
    auto fold = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      __syncthreads();
      if (!owner) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((grp * 4 + wave) * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (owner) {
        // partners in a fixed order: the other row parts of group 0, then every row part of group 1.  (Not unrolled: the
        // compiler otherwise requests all 7 x 16 values at once and spills the accumulators it is adding to.)
#pragma unroll 1
        for (int pi = 1; pi < G * WP; ++pi) {
          const int src = (pi / WP) * 4 + wave + pi % WP;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[(src * 16 + r) * 64 + lane];
        }
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
    if (!owner) red[(grp * 4 + wave) * 64 + lane] = dsum;
    __syncthreads();
    if (owner)
      for (int g2 = 0; g2 < G; ++g2)
        for (int o = (g2 == 0 ? 1 : 0); o < WP; ++o) dsum += red[(g2 * 4 + wave + o) * 64 + lane];
  }

  // ---- partial slab of this split:  [split][Coutp][9*Ctot], k order (tap, ci)
  const size_t KT = (size_t)9 * a.Ctot;
  float* slab = a.slabs + (size_t)split * a.Coutp * KT;
  const int ci = ci0 + j;
  if (pw == 0 && grp == 0) {
    auto store = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[(size_t)co * KT + (size_t)t * a.Ctot + ci] = acc[t][r] * descale * descale2;
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
  WG_STAMP_AT(14);
}

// ---- host side ----------------------------------------------------------------------------
static void h2_geometry(const vunet_wgrad_desc* d, int& MTW, int& ntiles, int& ciblocks, int& coblocks) {
  MTW = d->Cout % 64 == 0 ? 2 : 1;
  ntiles = d->N * (d->Ho / 4) * (d->Wo / 32);
  ciblocks = (d->C1 + d->C2) / 32;
  coblocks = (d->Cout + 32 * MTW - 1) / (32 * MTW);
}

int vunet_wgrad_h2_name(const vunet_wgrad_desc* d, char* name, int len) {
  return snprintf(name, len, "conv_wgrad_h2_kernel<%d>", d->Cout % 64 == 0 ? 2 : 1);
}

// same applicability and split count as the three-term kernel (vunet_wgrad_x6_applicable / _nslabs)
int vunet_wgrad_h2_launch(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy, float* slabs,
                          float* dshift, const float* amax_x, const float* amax_x2, const float* amax_dy, hipStream_t st) {
  WgradH2Args a;
  a.d = *d;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.slabs = slabs; a.dshift = dshift;
  a.amax_x = amax_x; a.amax_x2 = amax_x2; a.amax_dy = amax_dy;
  int MTW, ciblocks, coblocks;
  h2_geometry(d, MTW, a.ntiles, ciblocks, coblocks);
  a.Ctot = d->C1 + d->C2;
  a.Coutp = (d->Cout + 31) / 32 * 32;
  a.tiles_per_img_w = d->Wo / 32;
  a.tiles_per_img = (d->Ho / 4) * a.tiles_per_img_w;
  a.tps = (a.ntiles + d->nsplit - 1) / d->nsplit;
  a.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  a.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  dim3 grid(d->nsplit, ciblocks, coblocks);
  // per group: two planes (30 KiB) + the edge planes (6 KiB) [+ the raw fp32 tile and its halo columns, 25.5 KiB, for the
  // form that stages by LDS-DMA]; the fold through LDS reuses 16 KiB per group of it
  constexpr size_t staged = (size_t)2 * (6 * 32 * 5 + 1) * 16 + (size_t)2 * 6 * 4 * 32 * 4;
  constexpr size_t raw_bytes = (size_t)6 * 32 * 8 * 16 + (size_t)2 * 6 * 32 * 4;
#if defined(WG_H2_NO_DMA)
  constexpr size_t lds1 = WG_GROUPS * staged, lds2 = lds1;
#elif defined(WG_H2_DMA_ALL)
  constexpr size_t lds1 = WG_GROUPS * (staged + raw_bytes), lds2 = lds1;
#else
  constexpr size_t lds1 = WG_GROUPS * (staged + raw_bytes), lds2 = WG_GROUPS * staged;
#endif
  static const bool attr_ok = [] {
    return (lds1 <= 64 * 1024 || hipFuncSetAttribute((const void*)conv_wgrad_h2_kernel<1>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) == hipSuccess) &&
           (lds2 <= 64 * 1024 || hipFuncSetAttribute((const void*)conv_wgrad_h2_kernel<2>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) == hipSuccess);
  }();
  if (!attr_ok) return VUNET_ERR_LAUNCH;
  if (MTW == 1) VUNET_LAUNCH((conv_wgrad_h2_kernel<1>), grid, dim3(256 * WG_GROUPS), lds1, st, a);
  else VUNET_LAUNCH((conv_wgrad_h2_kernel<2>), grid, dim3(256 * WG_GROUPS), lds2, st, a);
  return vunet_check_launch();
}
