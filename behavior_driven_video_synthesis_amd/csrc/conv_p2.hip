// "p2": 3x3 / pad 1 / stride 1 convolution (forward and data gradient) on PRE-SPLIT activations -- the h2 arithmetic of
// conv_h2_kernel.h (two scaled fp16 terms per fp32 operand, three partial products, fp32 accumulation on the fp16 matrix
// cores) with the operand conversion done ONCE, by the producer of a tensor, instead of by every workgroup that stages it.
//
// Why.  conv_h2_kernel converts its input tile while staging it: loads to registers, scale, split, ds_write -- per element
// ~8 vector instructions, repeated by each of the M / 64 workgroups that read the tile and 1.25x by the halo: for VGG19
// conv4_x (512 -> 512 channels) every activation is converted TEN times per layer, and every workgroup first reduces 4 KB
// of partial |x| maxima to find the scale.  Here a tensor lives in HBM as two fp16 planes (hi, lo) in channel blocks of 8
// -- 16-byte units that ARE the B fragment of v_mfma_f32_16x16x32_f16 -- written by the producing kernel's epilogue
// (values are in its registers anyway), with a one-pixel zero border, so that the consumer's staging is a linear
// LDS-DMA copy (global_load_lds_dwordx4): no registers, no VALU, no ds_write, no bounds logic.  HBM bytes per element
// are the same as fp32 (2 + 2).
//
// Layout of a planes tensor ("P2"), logical shape [N][C][H][W], C % 8 == 0:
//     fp16  [2 planes][N][C / 8][H + 2][W + 2][8]        plane 0 = hi, plane 1 = lo;  s x = hi + lo / 2^11,  s = 2^e
//     int32 meta[128]:  [0] e (scale exponent)   [16 .. 79] 64 slots holding max |x| (fp32 bit patterns, atomicMax;
//                       zeroed by the caller before the producer runs): an UPPER BOUND of the largest stored magnitude
// The border rows / columns hold zeros and are never written by a producer: the caller zero-fills a buffer once and
// reuses it (ops.py keeps the VGG19 pass's buffers in a persistent arena).
//
// The scale of an OUTPUT cannot come from its maximum (unknown until the kernel has finished), so it comes from a bound:
//     |y| <= max_row( sum |w| ) * max|x| + max|shift|        (x's maximum IS known: its producer published it)
// rounded up to a power of two.  The bound is typically 2^5 .. 2^8 above the true maximum.  That costs nothing: the pair
// (hi, lo) represents s x with absolute error <= max(2^-23 |s x|, 2^-36) -- hi = f16(s x) has 11 bits, the remainder is
// scaled by 2^11 before it is rounded, and where hi goes subnormal the remainder still resolves 2^-36 -- so relative to the
// tensor's maximum the error stays at 2^-23 as long as that maximum sits above 2^-13 after scaling, i.e. for bounds up to
// 2^27 too large (tests/test_hip_p2.py: fp64 reference, inputs whose scale exponent is forced 2^20 too small).
//
// Work decomposition (NW waves; 16x16x32 MFMA: A = weights [16 rows][32 k], B = activations [32 k][16 pixels], the k of an
// instruction = 32 channels of ONE tap -- chunks of 32 input channels):
//   tile      16 NW output channels x (8 rows x CT columns) of one image; wave w: channel group w >> 2 (64 channels = 4
//             m-tiles), rows 2 (w & 3), 2 (w & 3) + 1 (2 CT / 16 pixel tiles): 4 x 4 accumulator tiles of 4 registers,
//             two sets (leading term / cross terms) = 128 VGPRs
//   xL        [XBUF][2 planes][4 k-quarters][10 x (CT + 2)] units, filled by LDS-DMA one chunk ahead (XBUF = 2: eight waves,
//             one workgroup per CU) or between chunks (XBUF = 1: four waves, two workgroups per CU cover each other)
//   wL        [2][16 NW / 16 m-tiles][2 planes][64] the weight slab of ONE tap, by LDS-DMA one tap ahead; one barrier per tap
//   epilogue  combine, de-scale, + shift, ReLU (forward) -- scale by the OUTPUT's 2^e, split, pack; two lane rows exchange
//             halves (v_permlane16_swap) so that a lane holds the 8 channels of one 16-byte unit; data gradient: the unit is
//             masked by [forward activation != 0] read as two 16-byte units of the forward planes (the ReLU backward of the
//             layer below); max |y| published into the output's meta
#include <stdio.h>

#include <type_traits>

#include "common.h"
#include "split_h2.h"

namespace {

constexpr int P2_META = 128, P2_AMAX0 = 16, P2_NSLOT = 64;

struct P2Args {
  const uint4* x;        // input planes
  const int* xmeta;
  const uint4* w;        // weight image (vunet_p2_pack_weights)
  const float* wk;       // [0] max row sum of |w|, [1] max |shift|, [2] weight scale exponent (as a float)
  const float* shift;    // [M] or null
  const uint4* mask;     // planes shaped like y (data gradient: the forward activation of the layer below) or null
  uint4* y;
  int* ymeta;
  int N, C, H, W, M;
  int relu;
};

__device__ __forceinline__ void p2_dma(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  // 64 lanes x 16 bytes from sbase + voff (per lane) to 1 KiB of LDS at the wave-uniform byte address lds_addr.  Inline
  // assembly for the reason given at h2_dma16 (conv_h2_kernel.h); retired by p2_dma_wait<N>() before the publishing barrier.
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
template <int N>
__device__ __forceinline__ void p2_dma_wait() {   // all but the N youngest transfers of this wave have landed
  static_assert(N == 0 || N == 1 || N == 2 || N == 4, "counts in use");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}
__device__ __forceinline__ const void* p2_uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const void*)(((uint64_t)hi << 32) | lo);
}

union P2Unit {
  uint4 u;
  h2_f16x8 b;
};

// upper bound of the stored magnitudes of a planes tensor: the 64 slots of its meta, one per lane
__device__ __forceinline__ float p2_meta_amax(const int* meta, int lane) {
  return wave_max(__int_as_float(meta[P2_AMAX0 + (lane & (P2_NSLOT - 1))]));
}

// two lanes of adjacent lane rows (lane, lane ^ 16) exchange halves: see the epilogue
__device__ __forceinline__ void p2_swap16(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// per 16-bit half: 0xFFFF where the half of t is non-zero (sign bits already cleared), else 0
__device__ __forceinline__ uint32_t p2_nonzero_halves(uint32_t t) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 v = __builtin_bit_cast(u16x2, t);
  const u16x2 one = {1, 1}, all = {0xFFFF, 0xFFFF};
  v = __builtin_elementwise_min(v, one) * all;
  return __builtin_bit_cast(uint32_t, v);
}

template <int NW, int CT, int XBUF>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void conv_p2_kernel(const P2Args a) {
  static_assert(NW == 4 || NW == 8, "waves");
  static_assert(CT == 32 || CT == 16, "tile width");
  constexpr int MW = 16 * NW;                     // output channels of a workgroup
  constexpr int RT = 8, IH = RT + 2, CW = CT + 2, XPIX = IH * CW, XU = 8 * XPIX;
  constexpr int NXI = (XU + 63) / 64, XS = NXI * 64;   // x tile of one chunk: wave-instructions / units incl. the overrun pad
  constexpr int NXW = (NXI + NW - 1) / NW;        // ... per wave
  constexpr int WU = MW * 8;                      // units of a weight slab: [MW / 16 m-tiles][2 planes][64]
  constexpr int NWI = WU / 64 / NW;               // wave-instructions per wave and slab (= 2)
  constexpr int PTR = CT / 16, NPT = 2 * PTR;     // 16-pixel tiles per tile row / per wave (two rows)
  static_assert(XBUF == 1 || NXW <= 8, "the next chunk's pieces ride on the first taps");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* const xL = smem;                // [XBUF][XS]
  uint4* const wL = smem + XBUF * XS;    // [2][WU]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mg = wave >> 2, rg = wave & 3;
  const int px = lane & 15, cg = lane >> 4;
  const int Hp = a.H + 2, Wp = a.W + 2, CB = a.C >> 3, MB = a.M >> 3;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = a.M / MW;
  const int mb = bid % mblocks;
  int t = bid / mblocks;
  const int tiles_w = a.W / CT, tiles_h = a.H / RT;
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * RT, col0 = tx * CT, m0 = mb * MW;

  // ---- chunk-invariant DMA geometry.  x: unit u of the tile = (plane, k-quarter, halo row, halo column); the tile's
  // origin in the PADDED map is (row0, col0) (its first halo row is padded row row0).  Byte offsets relative to the
  // (image, chunk) base; the wave-instructions past the end of the tile repeat its last piece (same bytes, same place).
  uint32_t xoff[NXW];
#pragma unroll
  for (int i = 0; i < NXW; ++i) {
    int j = wave + NW * i;
    if (j >= NXI) j = NXI - 1;
    int u = 64 * j + lane;
    if (u >= XU) u = XU - 1;             // (lands in the pad behind the tile)
    const int p = u / (4 * XPIX);
    int rem = u - p * 4 * XPIX;
    const int kq = rem / XPIX;
    rem -= kq * XPIX;
    const int r = rem / CW, c = rem - r * CW;
    xoff[i] = (uint32_t)((((size_t)p * a.N * CB + kq) * Hp + (row0 + r)) * Wp + col0 + c) * 16u;
  }
  const uint32_t xL_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)xL;
  const uint32_t wL_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)wL;
  const char* const xsrc0 = reinterpret_cast<const char*>(a.x) + (size_t)n * CB * Hp * Wp * 16;
  const size_t xchunk = (size_t)4 * Hp * Wp * 16;                  // bytes from one 32-channel chunk to the next
  const char* const wsrc0 = reinterpret_cast<const char*>(a.w) + (size_t)mb * (MW / 16) * 128 * 16;
  const size_t wslab = (size_t)(a.M / 16) * 128 * 16;             // bytes from one (chunk, tap) slab to the next
  const uint32_t lane16 = (uint32_t)lane * 16u;

  auto issue_x = [&](int ch, int i) {   // piece i of chunk ch's tile
    int j = wave + NW * i;
    if (j >= NXI) j = NXI - 1;
    const int xb = XBUF == 2 ? (ch & 1) : 0;
    p2_dma(xoff[i], p2_uniform_ptr(xsrc0 + (size_t)ch * xchunk), xL_addr + (uint32_t)(xb * XS + 64 * j) * 16u);
  };
  auto issue_w = [&](int g, int buf) {  // slab g = chunk * 9 + tap
    const char* src = wsrc0 + (size_t)g * wslab;
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int part = wave * NWI + i;
      p2_dma(lane16, p2_uniform_ptr(src + (size_t)part * 1024), wL_addr + (uint32_t)(buf * WU + part * 64) * 16u);
    }
  };

  // ---- prologue: the first stage is requested at once, the scales are worked out behind it
  issue_w(0, 0);
#pragma unroll
  for (int i = 0; i < NXW; ++i) issue_x(0, i);

  float descale, descale2, sy;
  {
    const int ex = a.xmeta[0];
    const float amax_x = p2_meta_amax(a.xmeta, lane);
    const int ew = (int)a.wk[2];
    h2_pow2_pair(-(ex + ew), descale, descale2);
    const float bound = a.wk[0] * amax_x + a.wk[1];
    const int ey = h2_scale_exp(bound);
    sy = h2_pow2(ey);
    if (blockIdx.x == 0 && tid == 0) a.ymeta[0] = ey;
  }

  f32x4 acc[4][NPT], acx[4][NPT];   // leading term / the two cross terms (scaled by 2^11)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < NPT; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][q][r] = acx[mt][q][r] = 0.f;

  p2_dma_wait<0>();
  __syncthreads();

  const int nch = a.C >> 5;
  const uint4* const wA0 = wL + (mg * 4) * 128 + lane;                      // + buf * WU + (mt * 2 + plane) * 64
  const uint4* const xB0 = xL + cg * XPIX + px + (2 * rg) * CW;            // + xbuf * XS + plane * 4 * XPIX + (row + kh) * CW + col + kw

  auto chunk = [&](int ch, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    const uint4* const xB = xB0 + (XBUF == 2 ? (ch & 1) * XS : 0);
    const int wpar = (ch * 9) & 1;     // slab parity of this chunk's tap 0
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int g = ch * 9 + tap;
      const int buf = wpar ^ (tap & 1);
      constexpr bool dummy = false;
      (void)dummy;
      const bool more_w = !(LAST && tap == 8);
      const bool xpiece = XBUF == 2 && !LAST && tap < NXW;
      if (more_w) issue_w(g + 1, buf ^ 1);      // (its buffer: slab g - 1, which every wave finished reading before the last barrier)
      if (xpiece) issue_x(ch + 1, tap);
      const uint4* const wA = wA0 + buf * WU;
      P2Unit av[2][4], bv[2][NPT];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) av[p][mt].u = wA[(mt * 2 + p) * 64];
#pragma unroll
        for (int q = 0; q < NPT; ++q) bv[p][q].u = xB[p * 4 * XPIX + (q / PTR + kh) * CW + (q % PTR) * 16 + kw];
      }
      // by product type: consecutive instructions never share an accumulator
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NPT; ++q)
          acx[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[1][mt].b, bv[0][q].b, acx[mt][q], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NPT; ++q)
          acx[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt].b, bv[1][q].b, acx[mt][q], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NPT; ++q)
          acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more_w) {
        // the weight slab of the next tap was requested BEFORE this tap's x piece: waiting for all but the newest
        // transfer retires it (and the previous tap's piece) and leaves the piece a second tap to land
        if (xpiece) p2_dma_wait<1>();
        else p2_dma_wait<0>();
        __syncthreads();
        if (XBUF == 1 && tap == 8) {   // single x buffer: every wave is through with it -- refill, wait, publish
#pragma unroll
          for (int i = 0; i < NXW; ++i) issue_x(ch + 1, i);
          p2_dma_wait<0>();
          __syncthreads();
        }
      }
    }
  };
  for (int ch = 0; ch + 1 < nch; ++ch) chunk(ch, std::false_type{});
  chunk(nch - 1, std::true_type{});

  // ---- epilogue.  Accumulator tile (mt, q): lane (px, cg) holds channels m0 + 64 mg + 16 mt + 4 cg + {0..3} of pixel
  // (row0 + 2 rg + q / PTR, col0 + 16 (q % PTR) + px).  Pixel tiles are taken in pairs (q, q + 1): after the packed halves
  // have been exchanged between lane rows cg and cg ^ 1, the lanes of an even row hold the 8-channel unit of tile q, those of
  // the odd row the unit of tile q + 1 (channel block (cg >> 1) of the m-tile): one 16-byte store per plane.
  const bool relu = a.relu != 0;
  const size_t plane = (size_t)a.N * MB * Hp * Wp;    // units of one output plane
  float ymax = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int co = m0 + 64 * mg + 16 * mt + 4 * cg;
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.shift) {
      const float4 s4 = *reinterpret_cast<const float4*>(a.shift + co);
      sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
    }
#pragma unroll
    for (int qp = 0; qp < NPT; qp += 2) {
      uint32_t hh[2][2], ll[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float c = (acc[mt][qp + s][r] + acx[mt][qp + s][r] * (1.f / 2048.f)) * descale * descale2 + sh[r];
          c = (relu && !(c > 0.f)) ? 0.f : c;
          ymax = fmaxf(ymax, fabsf(c));
          v[r] = c * sy;
        }
        h2_split2(v[0], v[1], hh[s][0], ll[s][0]);
        h2_split2(v[2], v[3], hh[s][1], ll[s][1]);
      }
      p2_swap16(hh[0][0], hh[1][0]);
      p2_swap16(hh[0][1], hh[1][1]);
      p2_swap16(ll[0][0], ll[1][0]);
      p2_swap16(ll[0][1], ll[1][1]);
      uint4 uh = make_uint4(hh[0][0], hh[0][1], hh[1][0], hh[1][1]);
      uint4 ul = make_uint4(ll[0][0], ll[0][1], ll[1][0], ll[1][1]);
      const int q = qp + (cg & 1);
      const int orow = row0 + 2 * rg + q / PTR, ocol = col0 + (q % PTR) * 16 + px;
      const int cb = ((m0 + 64 * mg + 16 * mt) >> 3) + (cg >> 1);
      const size_t o = (((size_t)n * MB + cb) * Hp + (orow + 1)) * Wp + (ocol + 1);
      if (a.mask) {
        const uint4 mh = a.mask[o], ml = a.mask[plane + o];
        const uint32_t k0 = p2_nonzero_halves((mh.x | ml.x) & 0x7FFF7FFFu), k1 = p2_nonzero_halves((mh.y | ml.y) & 0x7FFF7FFFu);
        const uint32_t k2 = p2_nonzero_halves((mh.z | ml.z) & 0x7FFF7FFFu), k3 = p2_nonzero_halves((mh.w | ml.w) & 0x7FFF7FFFu);
        uh.x &= k0; uh.y &= k1; uh.z &= k2; uh.w &= k3;
        ul.x &= k0; ul.y &= k1; ul.z &= k2; ul.w &= k3;
      }
      a.y[o] = uh;
      a.y[plane + o] = ul;
    }
  }
  {
    const float m_ = wave_max(ymax);
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned*>(a.ymeta) + P2_AMAX0 + ((blockIdx.x * NW + wave) & (P2_NSLOT - 1)), __float_as_uint(m_));
  }
}

// ---- the eight-wave form with its two wave groups in ANTIPHASE.
// In conv_p2_kernel every wave of a workgroup reads its fragments right after the tap's barrier and multiplies afterwards:
// eight waves want 128 KB from the LDS at once (1024 cycles at 128 B / clock), the matrix pipes wait, then the LDS idles
// while 2 x 768 cycles of MFMAs run -- measured: the phases ADD (0.45 of the fp16 peak: 63 % pipe occupancy at the clock
// the chip holds).  Here waves 0-3 (one per SIMD) and waves 4-7 (their partners on the same SIMDs) run half a step apart:
//      group 0:  R(t) | M(t) | R(t+1) | M(t+1) | ...          R = the tap's 16 ds_read_b128 (+ issuing LDS-DMA),
//      group 1:   -   | R(t) | M(t)   | R(t+1) | ...          M = its 48 MFMAs;  "|" = s_barrier of all eight waves
// so that in every segment one wave of each SIMD multiplies while its partner reads.  One register set of fragments is
// enough (a wave's R segment starts after its M segment has ended).  Group 0 copies the weight slabs (slab t+1 at the start
// of R(t): its buffer was last read by group 1 one segment earlier, reads retired before that barrier), group 1 the next
// chunk's activation pieces; each group retires its own transfers at the end of its M segment, i.e. a full two segments
// after issuing them and one barrier before anybody reads them.  (Measured and not kept, r05: three slab buffers with the
// transfers requested two taps ahead and counted waits that leave the newest in flight -- conv4_x 173 us against 168 - 170:
// the transfers' latency is not what the loop waits for.)
template <int CT>
__global__ __launch_bounds__(512, 1) void conv_p2a_kernel(const P2Args a) {
  constexpr int NW = 8, MW = 128;
  constexpr int RT = 8, IH = RT + 2, CW = CT + 2, XPIX = IH * CW, XU = 8 * XPIX;
  constexpr int NXI = (XU + 63) / 64, XS = NXI * 64;
  constexpr int NXG = (NXI + 3) / 4;               // x pieces per wave of group 1 and chunk
  constexpr int XPT = 2, NXT = (NXG + XPT - 1) / XPT;   // ... XPT per tap over the chunk's first NXT taps
  constexpr int WU = MW * 8;
  constexpr int PTR = CT / 16, NPT = 2 * PTR;
  static_assert(NXT <= 8, "the next chunk's pieces must be visible before its first tap");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* const xL = smem;                // [2][XS]
  uint4* const wL = smem + 2 * XS;       // [2][WU]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mg = wave >> 2, rg = wave & 3;          // mg is also the wave's phase group
  const int px = lane & 15, cg = lane >> 4;
  const int Hp = a.H + 2, Wp = a.W + 2, CB = a.C >> 3, MB = a.M >> 3;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = a.M / MW;
  const int mb = bid % mblocks;
  int t = bid / mblocks;
  const int tiles_w = a.W / CT, tiles_h = a.H / RT;
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * RT, col0 = tx * CT, m0 = mb * MW;

  uint32_t xoff[NXG];                               // piece rg + 4 i of the tile (group 1; group 0 uses them in the prologue too)
#pragma unroll
  for (int i = 0; i < NXG; ++i) {
    int j = rg + 4 * i;
    if (j >= NXI) j = NXI - 1;
    int u = 64 * j + lane;
    if (u >= XU) u = XU - 1;
    const int p = u / (4 * XPIX);
    int rem = u - p * 4 * XPIX;
    const int kq = rem / XPIX;
    rem -= kq * XPIX;
    const int r = rem / CW, c = rem - r * CW;
    xoff[i] = (uint32_t)((((size_t)p * a.N * CB + kq) * Hp + (row0 + r)) * Wp + col0 + c) * 16u;
  }
  const uint32_t xL_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)xL;
  const uint32_t wL_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)wL;
  const char* const xsrc0 = reinterpret_cast<const char*>(a.x) + (size_t)n * CB * Hp * Wp * 16;
  const size_t xchunk = (size_t)4 * Hp * Wp * 16;
  const char* const wsrc0 = reinterpret_cast<const char*>(a.w) + (size_t)mb * (MW / 16) * 128 * 16;
  const size_t wslab = (size_t)(a.M / 16) * 128 * 16;
  const uint32_t lane16 = (uint32_t)lane * 16u;

  auto issue_x = [&](int ch, int i) {   // this wave's piece i of chunk ch (waves with the same rg copy the same piece)
    int j = rg + 4 * i;
    if (j >= NXI) j = NXI - 1;
    p2_dma(xoff[i], p2_uniform_ptr(xsrc0 + (size_t)ch * xchunk), xL_addr + (uint32_t)((ch & 1) * XS + 64 * j) * 16u);
  };
  auto issue_w = [&](int g, int buf) {  // this wave's quarter of slab g (four wave-instructions)
    const char* src = wsrc0 + (size_t)g * wslab;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int part = rg * 4 + i;
      p2_dma(lane16, p2_uniform_ptr(src + (size_t)part * 1024), wL_addr + (uint32_t)(buf * WU + part * 64) * 16u);
    }
  };

  // ---- prologue: group 0 requests slab 0, group 1 the first chunk's tile; the scales are worked out behind the transfers
  if (mg == 0) issue_w(0, 0);
  else {
#pragma unroll
    for (int i = 0; i < NXG; ++i) issue_x(0, i);
  }
  float descale, descale2, sy;
  {
    const int ex = a.xmeta[0];
    const float amax_x = p2_meta_amax(a.xmeta, lane);
    const int ew = (int)a.wk[2];
    h2_pow2_pair(-(ex + ew), descale, descale2);
    const float bound = a.wk[0] * amax_x + a.wk[1];
    const int ey = h2_scale_exp(bound);
    sy = h2_pow2(ey);
    if (blockIdx.x == 0 && tid == 0) a.ymeta[0] = ey;
  }
  f32x4 acc[4][NPT], acx[4][NPT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < NPT; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][q][r] = acx[mt][q][r] = 0.f;
  p2_dma_wait<0>();
  __syncthreads();
  if (mg == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one segment behind

  const int nch = a.C >> 5;
  const uint4* const wA0 = wL + (mg * 4) * 128 + lane;
  const uint4* const xB0 = xL + cg * XPIX + px + (2 * rg) * CW;

  auto chunk = [&](int ch, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    const uint4* const xB = xB0 + (ch & 1) * XS;
    const int wpar = (ch * 9) & 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int g = ch * 9 + tap;
      const int buf = wpar ^ (tap & 1);
      const bool more_w = !(LAST && tap == 8);
      // ---- R segment: request what the NEXT steps need, then read this tap's fragments
      if (mg == 0) {
        if (more_w) issue_w(g + 1, buf ^ 1);
      } else if (!LAST && tap < NXT) {
#pragma unroll
        for (int i = 0; i < XPT; ++i)
          if (tap * XPT + i < NXG) issue_x(ch + 1, tap * XPT + i);
      }
      const uint4* const wA = wA0 + buf * WU;
      P2Unit av[2][4], bv[2][NPT];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) av[p][mt].u = wA[(mt * 2 + p) * 64];
#pragma unroll
        for (int q = 0; q < NPT; ++q) bv[p][q].u = xB[p * 4 * XPIX + (q / PTR + kh) * CW + (q % PTR) * 16 + kw];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are retired BEFORE the barrier: the partner group may refill
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M segment
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NPT; ++q)
          acx[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[1][mt].b, bv[0][q].b, acx[mt][q], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NPT; ++q)
          acx[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt].b, bv[1][q].b, acx[mt][q], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NPT; ++q)
          acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt].b, bv[0][q].b, acc[mt][q], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      p2_dma_wait<0>();                                     // this wave's transfers of the R segment: landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int ch = 0; ch + 1 < nch; ++ch) chunk(ch, std::false_type{});
  chunk(nch - 1, std::true_type{});
  if (mg == 0) __builtin_amdgcn_s_barrier();       // (both groups pass the same number of barriers)

  const bool relu = a.relu != 0;
  const size_t plane = (size_t)a.N * MB * Hp * Wp;
  float ymax = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int co = m0 + 64 * mg + 16 * mt + 4 * cg;
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.shift) {
      const float4 s4 = *reinterpret_cast<const float4*>(a.shift + co);
      sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
    }
#pragma unroll
    for (int qp = 0; qp < NPT; qp += 2) {
      uint32_t hh[2][2], ll[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float c = (acc[mt][qp + s][r] + acx[mt][qp + s][r] * (1.f / 2048.f)) * descale * descale2 + sh[r];
          c = (relu && !(c > 0.f)) ? 0.f : c;
          ymax = fmaxf(ymax, fabsf(c));
          v[r] = c * sy;
        }
        h2_split2(v[0], v[1], hh[s][0], ll[s][0]);
        h2_split2(v[2], v[3], hh[s][1], ll[s][1]);
      }
      p2_swap16(hh[0][0], hh[1][0]);
      p2_swap16(hh[0][1], hh[1][1]);
      p2_swap16(ll[0][0], ll[1][0]);
      p2_swap16(ll[0][1], ll[1][1]);
      uint4 uh = make_uint4(hh[0][0], hh[0][1], hh[1][0], hh[1][1]);
      uint4 ul = make_uint4(ll[0][0], ll[0][1], ll[1][0], ll[1][1]);
      const int q = qp + (cg & 1);
      const int orow = row0 + 2 * rg + q / PTR, ocol = col0 + (q % PTR) * 16 + px;
      const int cb = ((m0 + 64 * mg + 16 * mt) >> 3) + (cg >> 1);
      const size_t o = (((size_t)n * MB + cb) * Hp + (orow + 1)) * Wp + (ocol + 1);
      if (a.mask) {
        const uint4 mh = a.mask[o], ml = a.mask[plane + o];
        const uint32_t k0 = p2_nonzero_halves((mh.x | ml.x) & 0x7FFF7FFFu), k1 = p2_nonzero_halves((mh.y | ml.y) & 0x7FFF7FFFu);
        const uint32_t k2 = p2_nonzero_halves((mh.z | ml.z) & 0x7FFF7FFFu), k3 = p2_nonzero_halves((mh.w | ml.w) & 0x7FFF7FFFu);
        uh.x &= k0; uh.y &= k1; uh.z &= k2; uh.w &= k3;
        ul.x &= k0; ul.y &= k1; ul.z &= k2; ul.w &= k3;
      }
      a.y[o] = uh;
      a.y[plane + o] = ul;
    }
  }
  {
    const float m_ = wave_max(ymax);
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned*>(a.ymeta) + P2_AMAX0 + ((blockIdx.x * NW + wave) & (P2_NSLOT - 1)), __float_as_uint(m_));
  }
}

template <int CT>
int p2a_launch(const P2Args& a, hipStream_t st) {
  constexpr int XU = 8 * 10 * (CT + 2), XS = (XU + 63) / 64 * 64, WU = 128 * 8;
  constexpr size_t lds = (size_t)(2 * XS + 2 * WU) * 16;
  const int blocks = a.N * (a.H / 8) * (a.W / CT) * (a.M / 128);
  auto kern = conv_p2a_kernel<CT>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VUNET_ERR_LAUNCH;
  VUNET_LAUNCH(kern, dim3((unsigned)blocks), dim3(512), lds, st, a);
  return vunet_check_launch();
}

template <int NW, int CT, int XBUF>
int p2_launch(const P2Args& a, hipStream_t st) {
  constexpr int XU = 8 * 10 * (CT + 2), XS = (XU + 63) / 64 * 64, WU = 16 * NW * 8;
  constexpr size_t lds = (size_t)(XBUF * XS + 2 * WU) * 16;
  const int blocks = a.N * (a.H / 8) * (a.W / CT) * (a.M / (16 * NW));
  auto kern = conv_p2_kernel<NW, CT, XBUF>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VUNET_ERR_LAUNCH;
  VUNET_LAUNCH(kern, dim3((unsigned)blocks), dim3(64 * NW), lds, st, a);
  return vunet_check_launch();
}

}  // namespace

static bool p2_shape_ok(const vunet_p2_desc* d) {
  if (!d || d->N < 1 || d->C < 32 || d->C % 32 || d->M < 64 || d->M % 64 || d->H < 8 || d->H % 8) return false;
  if (!(d->W == 16 || (d->W >= 32 && d->W % 32 == 0))) return false;
  // 32-bit byte offsets inside one image's planes (the lo plane sits a whole plane behind the hi plane)
  const double plane_bytes = (double)d->N * (d->C / 8) * (d->H + 2) * (d->W + 2) * 16.0;
  return plane_bytes * 2.0 < 4.0e9;
}

extern "C" int vunet_p2_conv_supported(const vunet_p2_desc* d) { return p2_shape_ok(d) ? 1 : 0; }

// waves per workgroup: eight / 128 channels (x double-buffered, one workgroup per CU) where that still fills the chip;
// otherwise four / 64 channels (two workgroups per CU, x single-buffered).  VUNET_TUNE_P2_FORM: 1 = four, 2 = eight (tests, A/B)
static int p2_form(const vunet_p2_desc* d) {
  const long px_tiles = (long)d->N * (d->H / 8) * (d->W / (d->W == 16 ? 16 : 32));
  const int force = g_vunet_tune[VUNET_TUNE_P2_FORM];
  const bool wide = d->M % 128 == 0 && (force == 2 || force == 3 || (force != 1 && px_tiles * (d->M / 128) >= 256));
  return wide ? 8 : 4;
}

extern "C" int vunet_p2_conv(const vunet_p2_desc* d, const void* x, const int32_t* xmeta, const void* w_image, const float* wk,
                             const float* shift, const void* mask, void* y, int32_t* ymeta, void* stream) {
  if (!d || !x || !xmeta || !w_image || !wk || !y || !ymeta) return VUNET_ERR_ARG;
  if (!p2_shape_ok(d)) return VUNET_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w_image | (uintptr_t)y | (uintptr_t)mask | (uintptr_t)shift) & 15) return VUNET_ERR_ARG;
  P2Args a;
  a.x = (const uint4*)x; a.xmeta = xmeta; a.w = (const uint4*)w_image; a.wk = wk; a.shift = shift; a.mask = (const uint4*)mask;
  a.y = (uint4*)y; a.ymeta = ymeta;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->M; a.relu = d->relu;
  hipStream_t st = (hipStream_t)stream;
  const int form = p2_form(d);
  const bool lockstep = g_vunet_tune[VUNET_TUNE_P2_FORM] == 3;   // (A/B: the eight-wave form without the antiphase)
  if (d->W == 16) return form == 8 ? (lockstep ? p2_launch<8, 16, 2>(a, st) : p2a_launch<16>(a, st)) : p2_launch<4, 16, 1>(a, st);
  return form == 8 ? (lockstep ? p2_launch<8, 32, 2>(a, st) : p2a_launch<32>(a, st)) : p2_launch<4, 32, 1>(a, st);
}

// the kernel instantiation vunet_p2_conv launches for this problem, in rocprofv3's spelling (bench.py's roofline keys)
extern "C" int vunet_p2_conv_variant(const vunet_p2_desc* d, char* name, int32_t len) {
  if (!d || !name || len < 32) return VUNET_ERR_ARG;
  if (!p2_shape_ok(d)) return VUNET_ERR_UNSUPPORTED;
  const int form = p2_form(d);
  if (form == 8 && g_vunet_tune[VUNET_TUNE_P2_FORM] != 3) snprintf(name, (size_t)len, "conv_p2a_kernel<%d>", d->W == 16 ? 16 : 32);
  else snprintf(name, (size_t)len, "conv_p2_kernel<%d, %d, %d>", form, d->W == 16 ? 16 : 32, form == 8 ? 2 : 1);
  return VUNET_OK;
}

// ------------------------------------------------------------------------------------------------ weight image
// w [Cout][Cin][3][3] fp32 (the layer's effective weights) -> image [K / 32 chunks][9 taps][R / 16 m-tiles][2 planes][4
// k-quarters][16 rows] 16-byte units of 8 fp16 (k = 32 chunk + 8 quarter + j), scaled by 2^ew so that max |w| lands in
// [2^13, 2^14), split as the activations are.  forward: rows R = Cout, k = Cin, tap (kh, kw).  data gradient: rows = Cin,
// k = Cout, tap (kh', kw') holds w[..][2 - kh'][2 - kw'] (the mirrored tap: the kernel itself is the same).
// wk: [0] max over rows of sum |w| over (k, taps), [1] max |shift| (forward; 0 for the data gradient), [2] ew.
namespace {
__global__ __launch_bounds__(256) void p2_wstat_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad,
                                                       float* __restrict__ rows) {
  // one workgroup per row of the image's matrix: rows[2 r] = sum |w|, rows[2 r + 1] = max |w|
  __shared__ float red[8];
  const int r = blockIdx.x;
  float s = 0.f, m = 0.f;
  const int K = dgrad ? Cout : Cin;
  for (int i = threadIdx.x; i < K * 9; i += 256) {
    const int k = i / 9, t = i - 9 * k;
    const float v = fabsf(dgrad ? w[((size_t)k * Cin + r) * 9 + t] : w[((size_t)r * Cin + k) * 9 + t]);
    s += v;
    m = fmaxf(m, v);
  }
  s = wave_sum(s);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    rows[2 * r] = (red[0] + red[1]) + (red[2] + red[3]);
    rows[2 * r + 1] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  }
}
__global__ __launch_bounds__(256) void p2_wk_kernel(const float* __restrict__ rows, int R, const float* __restrict__ shift,
                                                    int nshift, float* __restrict__ wk) {
  __shared__ float red[12];
  float s = 0.f, m = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) {
    s = fmaxf(s, rows[2 * i]);
    m = fmaxf(m, rows[2 * i + 1]);
  }
  for (int i = threadIdx.x; i < nshift; i += 256) b = fmaxf(b, fabsf(shift[i]));
  s = wave_max(s);
  m = wave_max(m);
  b = wave_max(b);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = m;
    red[8 + (threadIdx.x >> 6)] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    // (the bound is used as  wk[0] * max|x| + wk[1]  in fp32: a relative 2^-20 covers the roundings of the sums)
    wk[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * (1.f + 1e-6f);
    wk[1] = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
    wk[2] = (float)h2_scale_exp(mx);
    wk[3] = mx;
  }
}
__global__ __launch_bounds__(256) void p2_wpack_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad,
                                                       const float* __restrict__ wk, uint4* __restrict__ img) {
  const int R = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  const int MT = R / 16;
  const size_t units = (size_t)(K / 32) * 9 * MT * 2 * 64;
  const float sw = h2_pow2((int)wk[2]);
  for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
    size_t v = u;
    const int l = (int)(v & 63);
    v >>= 6;
    const int p = (int)(v & 1);
    v >>= 1;
    const int mt = (int)(v % MT);
    v /= MT;
    const int tap = (int)(v % 9);
    const int ch = (int)(v / 9);
    const int row = mt * 16 + (l & 15), k0 = ch * 32 + (l >> 4) * 8;
    const int kh = tap / 3, kw = tap - 3 * kh;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      const float x = dgrad ? w[(((size_t)k * Cin + row) * 3 + (2 - kh)) * 3 + (2 - kw)] : w[(((size_t)row * Cin + k) * 3 + kh) * 3 + kw];
      e[j] = x * sw;
    }
    uint32_t h[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h2_split2(e[2 * j], e[2 * j + 1], h[j], lo[j]);
    img[u] = p ? make_uint4(lo[0], lo[1], lo[2], lo[3]) : make_uint4(h[0], h[1], h[2], h[3]);
  }
}
}  // namespace

extern "C" int vunet_p2_weight_image_bytes(int32_t Cout, int32_t Cin, int32_t dgrad) {
  const int R = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  if (R < 16 || R % 16 || K < 32 || K % 32 || (int64_t)R * K > (1 << 22)) return 0;
  return (K / 32) * 9 * (R / 16) * 2 * 64 * 16;
}

extern "C" int vunet_p2_pack_weights(const float* w, const float* shift, int32_t Cout, int32_t Cin, int32_t dgrad, void* image,
                                     float* wk, float* workspace, void* stream) {
  if (!w || !image || !wk || !workspace) return VUNET_ERR_ARG;
  if (vunet_p2_weight_image_bytes(Cout, Cin, dgrad) == 0) return VUNET_ERR_UNSUPPORTED;
  const int R = dgrad ? Cin : Cout;
  hipStream_t st = (hipStream_t)stream;
  VUNET_LAUNCH(p2_wstat_kernel, dim3((unsigned)R), dim3(256), 0, st, w, (int)Cout, (int)Cin, (int)dgrad, workspace);
  VUNET_LAUNCH(p2_wk_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, R, dgrad ? nullptr : shift,
               (dgrad || !shift) ? 0 : (int)Cout, wk);
  const size_t units = (size_t)vunet_p2_weight_image_bytes(Cout, Cin, dgrad) / 16;
  size_t nb = (units + 255) / 256;
  if (nb > 2048) nb = 2048;
  VUNET_LAUNCH(p2_wpack_kernel, dim3((unsigned)nb), dim3(256), 0, st, w, (int)Cout, (int)Cin, (int)dgrad, (const float*)wk,
               (uint4*)image);
  return vunet_check_launch();
}

// wk alone (no image): the bound constants of a layer the p2 kernels do not run themselves (the 3-channel first layer,
// csrc/conv_thin.hip: vunet_p2_conv_first).  w [Cout][Cin][3][3]; workspace: 2 * Cout floats.
extern "C" int vunet_p2_weight_bound(const float* w, const float* shift, int32_t Cout, int32_t Cin, float* wk, float* workspace,
                                     void* stream) {
  if (!w || !wk || !workspace || Cout < 1 || Cin < 1) return VUNET_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  VUNET_LAUNCH(p2_wstat_kernel, dim3((unsigned)Cout), dim3(256), 0, st, w, (int)Cout, (int)Cin, 0, workspace);
  VUNET_LAUNCH(p2_wk_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)Cout, shift, shift ? (int)Cout : 0, wk);
  return vunet_check_launch();
}

// ------------------------------------------------------------------------------------------------ converters
namespace {
// fp32 NCHW -> planes.  amax: partial |x| maxima (n_amax floats, e.g. the 512 of vunet_absmax_partials / a producer's tag):
// the scale comes from the true maximum.  One thread per unit: 8 channels of one pixel (coalesced across lanes in x).
__global__ __launch_bounds__(256) void p2_from_nchw_kernel(const float* __restrict__ x, const float* __restrict__ amax, int n_amax,
                                                           int relu, uint4* __restrict__ y, int* __restrict__ ymeta, int N, int C,
                                                           int H, int W) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < n_amax; i += 256) m = fmaxf(m, amax[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int e = h2_scale_exp(m);
  const float s = h2_pow2(e);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ymeta[0] = e;
    ymeta[P2_AMAX0] = __float_as_int(m);
  }
  const int CB = C >> 3, Hp = H + 2, Wp = W + 2;
  const size_t HW = (size_t)H * W, total = (size_t)N * CB * HW, plane = (size_t)N * CB * Hp * Wp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xw = (int)(i % W);
    size_t t = i / W;
    const int yh = (int)(t % H);
    t /= H;   // = n * CB + cb
    const float* src = x + (t * 8) * HW + (size_t)yh * W + xw;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = src[j * HW];
      if (relu) f = f > 0.f ? f : 0.f;
      v[j] = f * s;
    }
    uint4 h, l;
    h2_split2(v[0], v[1], h.x, l.x);
    h2_split2(v[2], v[3], h.y, l.y);
    h2_split2(v[4], v[5], h.z, l.z);
    h2_split2(v[6], v[7], h.w, l.w);
    const size_t o = (t * Hp + (yh + 1)) * Wp + (xw + 1);
    y[o] = h;
    y[plane + o] = l;
  }
}
__device__ __forceinline__ void p2_unpack(const uint4& h, const uint4& l, float inv, float* v) {
  const uint32_t hs[4] = {h.x, h.y, h.z, h.w}, ls[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const h2_f16x2 a = __builtin_bit_cast(h2_f16x2, hs[j]), b = __builtin_bit_cast(h2_f16x2, ls[j]);
    v[2 * j] = ((float)a[0] + (float)b[0] * (1.f / 2048.f)) * inv;
    v[2 * j + 1] = ((float)a[1] + (float)b[1] * (1.f / 2048.f)) * inv;
  }
}
__global__ __launch_bounds__(256) void p2_to_nchw_kernel(const uint4* __restrict__ x, const int* __restrict__ xmeta,
                                                         float* __restrict__ y, int N, int C, int H, int W) {
  const int CB = C >> 3, Hp = H + 2, Wp = W + 2;
  const float inv = h2_pow2(-xmeta[0]);
  const size_t HW = (size_t)H * W, total = (size_t)N * CB * HW, plane = (size_t)N * CB * Hp * Wp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xw = (int)(i % W);
    size_t t = i / W;
    const int yh = (int)(t % H);
    t /= H;
    const size_t o = (t * Hp + (yh + 1)) * Wp + (xw + 1);
    float v[8];
    p2_unpack(x[o], x[plane + o], inv, v);
    float* dst = y + (t * 8) * HW + (size_t)yh * W + xw;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j * HW] = v[j];
  }
}
}  // namespace

extern "C" int vunet_p2_from_nchw(const float* x, const float* amax, int32_t n_amax, int32_t relu, void* y, int32_t* ymeta,
                                  int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  if (!x || !amax || n_amax < 1 || !y || !ymeta || C % 8 || N < 1) return VUNET_ERR_ARG;
  const size_t total = (size_t)N * (C / 8) * H * W;
  size_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  VUNET_LAUNCH(p2_from_nchw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, amax, (int)n_amax, (int)relu, (uint4*)y,
               (int*)ymeta, (int)N, (int)C, (int)H, (int)W);
  return vunet_check_launch();
}

extern "C" int vunet_p2_to_nchw(const void* x, const int32_t* xmeta, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                void* stream) {
  if (!x || !xmeta || !y || C % 8 || N < 1) return VUNET_ERR_ARG;
  const size_t total = (size_t)N * (C / 8) * H * W;
  size_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  VUNET_LAUNCH(p2_to_nchw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const int*)xmeta, y, (int)N,
               (int)C, (int)H, (int)W);
  return vunet_check_launch();
}
