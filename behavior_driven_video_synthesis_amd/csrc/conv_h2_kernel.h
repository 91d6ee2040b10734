// fp32-accurate 3x3 / pad 1 convolution (forward and data gradient) on the fp16 matrix cores: the "h2" scheme.
//
// Every fp32 operand, scaled by a power of two so that its tensor's largest magnitude lands below 2^14, is split into two
// fp16 terms
//       s x = xh + xl / 2^11      (xh = f16(s x), xl = f16((s x - xh) * 2^11); round-to-nearest-even: 11 + 11 significand
//                                  bits and the sign of the remainder cover 23 of fp32's 24 bits)
// and the product  w * x  is formed from the three partial products  wh xh + (wh xl + wl xh) / 2^11 , each exact in fp32
// (11 x 11 significand bits) and accumulated in fp32 by v_mfma_f32_32x32x16_f16 -- the leading term in one accumulator,
// the two cross terms in a second one, combined (and de-scaled) in the epilogue.  The dropped term (wl xl) and the
// representation error are <= 2^-22 of the product: measured against an fp64 convolution the result is as close as the
// three-term bf16 split (conv_x6_kernel.h) and the fp32 fmaf chain -- with HALF the matrix instructions of the former:
// 3 MFMAs of K = 16 per 32 x 32 x 16 block, a roof of 2.5 PFLOP/s / 3 = 833 TFLOP/s fp32-equivalent.
// The remainder keeps its own exponent (scaled by 2^11 into fp16's normal range) as long as |s x| >= 2^-14, i.e. 28
// binades below the tensor's maximum; the scale is exact (power of two), so nothing else depends on the data.
//   scale of x   from the tensor's |x| maximum: vunet_absmax_partials writes 1024 partial maxima (512 per source), every
//                workgroup reduces them at entry (4 KB from L2).  The prologue (ELU / dropout / ReLU mask) never grows
//                |x| beyond  max|x| * keep_scale.
//   scale of w   per layer, from max |w_eff| (vunet_weightnorm_fwd*): unit 0 of the image holds its exponent.
//
// Work decomposition (256 threads = 4 waves; 2 workgroups per CU, 4 for the one-m-tile 4-row form):
//   tile      MT*32 output channels x (4*NT rows x 32 columns) of one image; wave w owns rows w*NT .. w*NT+NT-1
//             (TW = 16: a column tile is 16 pixels of two rows, see the template comment)
//   K loop    input channels in chunks of 16 (one fp16 MFMA K step per tap); per chunk three PHASES, one per kernel
//             row kh, each 3 taps x 3 products x MT x NT MFMAs per wave; one barrier per phase
//   xL        [2 buffers][2 planes][2 k-halves][(TH+2) x 34 pixels] 16-byte units of 8 fp16: a B-fragment read is 32
//             consecutive units per k-half (conflict-free ds_read_b128, tap shift = immediate).  The NEXT chunk's tile is
//             staged into the other buffer in three rounds, one per phase: loads before the phase's MFMAs, prologue +
//             scale + split + ds_write after them -- 8 live staging registers.
//   wL        [2 buffers] the phase's weight slab [3 kw][2 planes][2 k-halves][32], copied by LDS-DMA (no registers)
//   epilogue  combine the two accumulators, de-scale, then shift / activation / residual (data gradient: * act'(aux) +
//             res) through 16-byte accesses after a lane-quad transpose; publishes max|y| if asked (GatherArgs::amax_out)
// What was measured to bound it -- and what did not help -- is recorded at the epilogue below and in DESIGN.md section 5.
#pragma once
#include <type_traits>

#include "conv_common.h"

#include "split_h2.h"

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS starting at the
// wave-uniform byte address `lds_addr`.  Issued as inline assembly so that the compiler does not track it: the builtin
// form makes hipcc wait vmcnt(0) before the NEXT ds_read (it cannot tell the read from the DMA's destination), i.e.
// before the MFMA block the transfer is meant to hide behind.  The kernel retires the DMA itself (h2_dma_wait) before the
// barrier that publishes the slab.  M0 is saved and restored (compiler-reserved).
__device__ __forceinline__ void h2_dma16(const void* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void h2_dma_wait() {
#ifndef H2_ABL_NOWAIT   // (timing ablations for tools/ab_build.sh: results are wrong with any H2_ABL_* defined)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

union H2Unit {
  uint4 u;
  h2_f16x8 b;
};

// Timing-only builds (tools/ab_build.sh -DH2_ABL_* / -DH2_PROBE_* / -DH2_SINGLE / -DH2_PIPE) compile parts of the kernel out or
// change its arithmetic: their results are WRONG by construction.  Such a library refuses every launch of this kernel unless
// the process says it knows (VUNET_ALLOW_TIMING_BUILD=1, set by the timing tools) -- a product build defines none of them.
#if defined(H2_ABL_NOX) || defined(H2_ABL_NOW) || defined(H2_ABL_NOSTORE) || defined(H2_ABL_NOWAIT) || \
    defined(H2_PROBE_VLOAD) || defined(H2_SINGLE) || defined(H2_PIPE) || defined(H2_NO_TAP_BARRIER)
#define VUNET_H2_TIMING_BUILD 1
#else
#define VUNET_H2_TIMING_BUILD 0
#endif
static inline bool h2_launch_allowed() {
  if (!VUNET_H2_TIMING_BUILD) return true;
  static const bool ok = [] { const char* e = getenv("VUNET_ALLOW_TIMING_BUILD"); return e && e[0] == '1'; }();
  return ok;
}

// (diagnostic, tools/ab_build.sh -DH2_TSTAMP=<export name>: wave 0 of every workgroup records the 100 MHz wall clock at its
// phase boundaries; tools/h2_timeline.py reads them back through the exported function.  Results are unchanged.)
#ifdef H2_TSTAMP
static __device__ unsigned long long h2_ts[8192 * 8];
#define H2_STAMP(k_)                                                                                              \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 8192) h2_ts[blockIdx.x * 8 + (k_)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
extern "C" int H2_TSTAMP(unsigned long long* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(h2_ts), sizeof(unsigned long long) * 8 * (size_t)nblocks) == hipSuccess ? 0 : -1;
}
#else
#define H2_STAMP(k_)
#endif

// MODE 0 forward, 1 data gradient (taps mirrored).  PRO 0 none, 1 ELU, 2 ELU + dropout, 4 ReLU mask (MODE 1).
// PHW >= 0 (MODE 1 only): data gradient of the STRIDE-2 convolution (Downsample, lib/modules.py:152-158) for the output
// parity (ph, pw) = (PHW >> 1, PHW & 1): dx[2a+ph][2b+pw] = sum over the taps kh = ph+1 (mod 2), kw = pw+1 (mod 2) of
// w[kh][kw] * dy[a + (ph+1-kh)/2][b + (pw+1-kw)/2] -- a stride-1 problem on the dy map with 1, 2, 2 or 4 of the 9
// taps, stored to every other pixel of dx.  Four launches cover the four parities; no MFMA is spent on the structural
// zeros of a transposed strided convolution.
// PHW == 4: ALL FOUR parities in one launch (NT = 1).  Every tap (kh, kw) belongs to exactly one parity class
// (ph, pw) = ((kh+1) & 1, (kw+1) & 1), so a workgroup stages its dy tile ONCE, walks all nine taps like a stride-1 layer and
// keeps one accumulator tile per parity (the register budget of NT = 4); the epilogue pairs the pw = 0 / 1 tiles of an
// output row into 8-byte stores.  The four-launch form staged the dy tile four times for 1 + 2 + 2 + 4 taps and wrote
// every other dword of dx per launch: 54 TFLOP/s against ~250 for the stride-1 layers of the same size (r03).
// NWV waves per workgroup: 4 -> 256 threads, two workgroups per CU, one weight slab (kernel row) per barrier;
//                          8 -> 512 threads, ONE workgroup per CU owning most of the LDS: twice the tile rows (less halo,
//                               the weight slabs shared by twice the MFMAs), the weights of a whole chunk (all kernel rows)
//                               per stage and one barrier per chunk.  Measured on the bs-16 layers (r02, tools/time_conv.py):
//                               within 2 % of the four-wave form everywhere -- neither the barriers nor the weight staging
//                               are what bounds the kernel -- so only NWV = 4 is instantiated.
// TW: tile width in pixels.  32: an MFMA column tile is 32 consecutive pixels of one row; 16 (maps 16 wide: VGG19 conv5,
//     the decoder's 16 x 16 level): it is 16 pixels of two consecutive rows, lane j -> (row j / 16, column j % 16); the
//     staged rows are then 32 units apart (18 used), which keeps the two half-rows of a fragment read on distinct banks.
// MT * NT >= 8 (128 channels x 8 rows, or 64 channels x 16 rows, per four-wave workgroup): ONE workgroup per CU, one wave
// per SIMD with the whole 512-entry register file (256 accumulator registers in AGPRs).  Built and measured in round 3
// (tools/time_conv.py): 10 - 25 % SLOWER than the 64 x 8 tile at two waves per SIMD -- with nothing to switch to, every
// fragment read's latency is exposed unless the reads are pipelined by hand, and the compiler's schedule is read -> wait ->
// MFMA; not instantiated.
template <int MT, int NT, int MODE, int PRO, int PHW = -1, int NWV = 4, int TW = 32>
__global__ __launch_bounds__(64 * NWV, (NWV == 8 || MT * NT >= 8) ? 1 : 2) void conv_h2_kernel(const GatherArgs a_in,
                                                                             const uint4* __restrict__ wx, int mtiles_pad,
                                                                             const float* __restrict__ amax) {
  GatherArgs a = a_in;
  H2_STAMP(0);
  inact_resolve(a.in1);
  inact_resolve(a.in2);
  inact_resolve(a.auxa);
  static_assert(PHW < 0 || MODE == 1, "parity phases exist for the data gradient only");
  constexpr bool ALLP = PHW == 4;         // every output parity in this launch
  static_assert(!ALLP || (NT == 1 && TW == 32 && NWV == 4 && PRO == 0), "fused parities: one dy row per wave");
  constexpr int NQ = ALLP ? 4 : NT;       // accumulator tiles per m-tile: pixel-row tiles, or the four parities
  // PIPE (16-wide maps): the fragment reads of tap kw + 1 are issued before the MFMAs of tap kw.  VGG19 conv5_x at bs 16 is
  // 256 workgroups -- one wave per SIMD, no partner whose MFMAs cover this wave's LDS latency -- and the compiler's schedule
  // is read -> wait -> MFMA: 93 -> 82 us forward, 113 -> 103 us data gradient (r03).  With two waves per SIMD (every other
  // form) the same pipelining was 3 - 16 % SLOWER (r02, r03: -DH2_PIPE; conv3_x 242 -> 278 us), so it is tied to the tile width
  // (the 128-channel 32^2 layers, 256 workgroups of the 8-row tile, would gain 8 % -- 2 us on 14 launches: not instantiated).  Requesting all
  // staging rounds at the start of a chunk, dropping the weight-DMA wait (timing only) and one m-tile per workgroup (512
  // workgroups) each changed nothing here: the chunk time of this form is its fragment-read latency.
#ifdef H2_PIPE
  constexpr bool PIPE = PHW < 0;
#else
  constexpr bool PIPE = TW == 16 && PHW < 0;
#endif
  constexpr int PH = (PHW >= 0 && !ALLP) ? (PHW >> 1) : 0, PW = (PHW >= 0 && !ALLP) ? (PHW & 1) : 0;
  constexpr int NKH = (PHW < 0 || ALLP) ? 3 : (PH ? 2 : 1);   // kernel rows visited per chunk
  constexpr int NTHR = 64 * NWV;
  static_assert(TW == 32 || TW == 16, "tile width");
  constexpr int RPQ = 32 / TW;            // image rows per 32-pixel MFMA column tile
  constexpr int TH = NWV * NT * RPQ, IH = TH + 2, CW = TW + 2, IW = TW == 32 ? CW : 32, PIX = IH * IW, MB = 32 * MT;
  constexpr int XU = 2 * IH * CW;         // staging units of the input tile: (k-half, halo row, halo column)
  constexpr int NX = (XU + NTHR - 1) / NTHR;
  constexpr int WU = H2_SLAB * MT;        // units of one weight slab
  constexpr int GS = NWV == 8 ? NKH : 1;  // weight slabs (kernel rows) per stage = per barrier
  constexpr int WUS = WU * GS;            // units of one stage of weights
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* const xL = smem4;                // [2 buffers][2 planes][2 k-halves][PIX]
  uint4* const wL = smem4 + 8 * PIX;      // [2 buffers][WUS]

  const vunet_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int jr = j / TW, jc = j % TW;     // this lane's pixel inside its column tile
  const int H = d.Hs, W = d.Ws, HW = a.HsWs;

  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mblocks = (d.M + MB - 1) / MB;
  const int mb = bid % mblocks;
  int t = bid / mblocks;
  const int tiles_w = W / TW, tiles_h = H / TH;
  const int tx = t % tiles_w;
  t /= tiles_w;
  const int ty = t % tiles_h;
  const int n = t / tiles_h;
  const int row0 = ty * TH, col0 = tx * TW, m0 = mb * MB;

  float sx, descale, descale2;   // set in the prologue below, once the first loads are in flight

  // ---- chunk-invariant staging geometry: unit u = (k-half c8, halo row r, halo column col), lanes walk columns
  unsigned rel[NX];   // element offsets; loads address as scalar base + unsigned 32-bit BYTE offset (no 64-bit pairs)
  int lds_x[NX];
  uint32_t vbits = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    int u = tid + NTHR * i;
    if (u >= XU) u -= XU;   // the threads past the end of the tile redo its first units (same bytes): no divergent staging
    const int c8 = u / (IH * CW);
    const int rem = u - c8 * (IH * CW);
    const int r = rem / CW, col = rem - r * CW;
    const int ih = row0 - 1 + r, iw = col0 - 1 + col;
    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    rel[i] = ok ? (unsigned)(8 * c8 * HW + ih * W + iw) : 0u;   // invalid: a safe in-bounds address, masked afterwards
    lds_x[i] = c8 * PIX + r * IW + col;
    vbits |= (ok ? 1u : 0u) << i;
  }

#ifdef H2_SINGLE   // (experiment, tools/ab_build.sh: one accumulator set, low terms at natural scale)
  f32x16 acc[MT][NQ];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;
#else
  f32x16 acc[MT][NQ], acx[MT][NQ];   // leading term / the two cross terms (scaled by 2^11)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][q][r] = acx[mt][q][r] = 0.f;
#endif

  const int nch1 = d.C1 / 16, nch = nch1 + d.C2 / 16;
  float xv[NX][8];
  float mk[1][8];
  const int mt0 = (d.m_off + m0) >> 5;   // first m-tile of this workgroup in the wx image

  // The input tile of chunk c+1 is staged in NX rounds of 256 units spread over the phases of chunk c (round i in phase
  // i * NKH / NX): its loads are issued before the phase's MFMA block and converted / written to the OTHER xL buffer
  // after it, so only one round's 8 values are live across the MFMAs.
  auto issue_x = [&](int ch, int i) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    const float* __restrict__ xs = (second ? a.x2 : a.x1) + (size_t)(n * C + cs) * HW;
#ifdef H2_PROBE_VLOAD   // (timing probe, tools/ab_build.sh: two 16-byte loads instead of eight dword loads; WRONG data)
    {
      const float4 va = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs) + 4u * (rel[i] & ~3u));
      const float4 vb = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs) +
                                                         4u * ((rel[i] & ~3u) + (((vbits >> i) & 1u) ? (unsigned)(4 * HW) : 0u)));
      xv[i][0] = va.x; xv[i][1] = va.y; xv[i][2] = va.z; xv[i][3] = va.w;
      xv[i][4] = vb.x; xv[i][5] = vb.y; xv[i][6] = vb.z; xv[i][7] = vb.w;
    }
#else
#pragma unroll
    for (int k = 0; k < 8; ++k)
      xv[i][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xs) +
                                                 4u * (rel[i] + (((vbits >> i) & 1u) ? (unsigned)(k * HW) : 0u)));
#endif
  };
  // PRO 4: the ReLU mask travels separately and late (after the MFMA block, when the fragment registers are free)
  auto issue_mask = [&](int ch, int i) {
    if constexpr (PRO == 4) {
      const float* __restrict__ ms = a.mask + (size_t)(n * d.C1 + ch * 16) * HW;   // mode 1, single source
#pragma unroll
      for (int k = 0; k < 8; ++k)
        mk[0][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ms) +
                                                   4u * (rel[i] + (((vbits >> i) & 1u) ? (unsigned)(k * HW) : 0u)));
    }
  };
  auto write_x = [&](int ch, int i) {
    const bool second = ch >= nch1;
    const int cs = second ? (ch - nch1) * 16 : ch * 16;
    const int C = second ? d.C2 : d.C1;
    InAct ia = a.in1;                       // the two sources differ in the dropout seed only
    ia.seed = second ? a.in2.seed : a.in1.seed;
    const int gbase = (n * C + cs) * HW;
    const bool ok = (vbits >> i) & 1u;
    // (a unit outside the image: its INPUT is zeroed -- every prologue maps 0 to 0 -- instead of selecting the result: written
    //  as "ok ? f(x) * sx : 0" the compiler sinks each element's prologue into an exec-mask block of its own, eight
    //  saveexec / branch / restore sequences per unit around ~16 VALU instructions each)
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#ifdef H2_SELECT_AFTER   // (A/B, tools/ab_build.sh: the form before round 4 -- same values)
      float t_ = xv[i][k];
      if constexpr (PRO == 4) t_ = mk[0][k] > 0.f ? t_ : 0.f;
      else if constexpr (PRO != 0) t_ = prologue<PRO>(ia, t_, (uint32_t)gbase + rel[i] + (uint32_t)(k * HW));
      v[k] = ok ? t_ * sx : 0.f;
#else
      float t_ = ok ? xv[i][k] : 0.f;
      if constexpr (PRO == 4) t_ = mk[0][k] > 0.f ? t_ : 0.f;
      else if constexpr (PRO != 0) t_ = prologue<PRO>(ia, t_, (uint32_t)gbase + rel[i] + (uint32_t)(k * HW));
      v[k] = t_ * sx;
#endif
    }
    uint4 ph, pl;
    h2_split2(v[0], v[1], ph.x, pl.x);
    h2_split2(v[2], v[3], ph.y, pl.y);
    h2_split2(v[4], v[5], ph.z, pl.z);
    h2_split2(v[6], v[7], ph.w, pl.w);
    uint4* const xb = xL + (ch & 1) * 4 * PIX;
    xb[lds_x[i]] = ph;
    xb[2 * PIX + lds_x[i]] = pl;
  };
  // Weight slab staging by LDS-DMA (global_load_lds_dwordx4: one wave-instruction copies 64 consecutive units = 1 KiB
  // straight into LDS, no registers, no ds_write): a slab is a linear copy of WU / 64 = 6 * MT wave-instructions, a
  // stage is GS slabs, dealt round-robin to the NWV waves; the waves past the end wrap around and rewrite the first units
  // with the same bytes.  The DMA of stage s+1 is issued before the MFMA block of stage s and retired (h2_dma_wait) just
  // before the barrier that ends the stage.
  constexpr int WPS = WU / 64;               // wave-instructions per slab
  constexpr int NWI = (GS * WPS + NWV - 1) / NWV;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t wL_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)wL;
  // step -> weight slab (chunk * 3 + kh): all three kernel rows, or only those of this output parity
  auto slab_of = [&](int step) {
    if constexpr (PHW < 0 || ALLP) return step;
    else if constexpr (PH == 0) return step * 3 + 1;                     // kh = 1
    else return (step >> 1) * 3 + ((step & 1) ? 2 : 0);                  // kh = 0, 2
  };
  auto issue_w = [&](int stage, int buf) {   // stage = GS consecutive steps (step = chunk * NKH + ki)
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      int wi = wave_u + NWV * i;
      if (wi >= GS * WPS) wi -= GS * WPS;
      const int g = wi / WPS, r = wi - g * WPS;
      const char* wp = reinterpret_cast<const char*>(wx) + 16 +
                       ((size_t)slab_of(stage * GS + g) * mtiles_pad + mt0) * (H2_SLAB * 16);
      h2_dma16(wp + (size_t)(r * 64 + lane) * 16, wL_addr + (uint32_t)(buf * WUS + g * WU + r * 64) * 16u);
    }
  };

  // ---- prologue: everything the first stage needs is requested at once (weight DMA, all rounds of the input tile, the
  //      |x| maxima), converted as it arrives
  issue_w(0, 0);
#pragma unroll
  for (int i = 0; i < NX; ++i) issue_x(0, i);
  // ---- scales: x by 2^ex from the tensor maximum (1024 partial maxima, reduced here, behind the loads issued above),
  //      w by the exponent in the image header
  {
    const float4 pm = h2_amax4(amax, a.amax2, tid & 255);
    float m_ = fmaxf(fmaxf(pm.x, pm.y), fmaxf(pm.z, pm.w));
    m_ = wave_max(m_);
    float* const redm = reinterpret_cast<float*>(smem4);
    if (lane == 0) redm[wave] = m_;
    __syncthreads();
    m_ = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));   // waves 0..3 cover all 1024 entries
    __syncthreads();
    if (a.in1.thresh) m_ *= a.in1.keep_scale;        // dropout rescales the kept values
    const int ex = h2_scale_exp(m_);
    const int ew = reinterpret_cast<const int*>(wx)[0];
    sx = h2_pow2(ex);
    h2_pow2_pair(-(ex + ew), descale, descale2);
  }
  H2_STAMP(1);

#pragma unroll
  for (int i = 0; i < NX; ++i) {
    issue_mask(0, i);
    write_x(0, i);
  }
  h2_dma_wait();
  __syncthreads();
  H2_STAMP(2);

  const uint4* const xB0 = xL + h * PIX + (wave * NT * RPQ + jr) * IW + jc;  // + buffer*4*PIX + plane*2*PIX + (q*RPQ + dr)*IW + dc
  const uint4* const wA = wL + h * 32 + j;                       // + buf*WUS + g*WU + ((mt*3 + kw)*2 + plane)*64
  // One chunk = three phases (kernel rows).  LAST is a compile-time flag so that every prefetch is unconditional code.
  auto chunk = [&](int ch, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
    for (int ki = 0; ki < NKH; ++ki) {
      const int kh = (PHW < 0 || ALLP) ? ki : (PH ? 2 * ki : 1);
      const int ph_t = ALLP ? ((kh + 1) & 1) : PH;   // the output-row parity this kernel row feeds
      const int phase = ch * NKH + ki;
      const int stage = phase / GS, g = phase - stage * GS;   // GS == NKH: stage = chunk, g = ki
      const int buf = stage & 1;
      const bool stage_begin = g == 0, stage_end = g == GS - 1;
      const bool more_w = !(LAST && ki == NKH - 1);            // another stage follows this step's
      const uint4* const xB = xB0 + (ch & 1) * 4 * PIX;
#ifndef H2_ABL_NOX
      if constexpr (!LAST) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
          if (i * NKH / NX == ki) issue_x(ch + 1, i);
      }
#endif
#ifndef H2_ABL_NOW
      if (stage_begin && !(LAST && GS == NKH) && (GS == NKH || more_w)) issue_w(stage + 1, buf ^ 1);
#endif
      // row / column of the staged tile (origin row0-1, col0-1) that tap (kh, kw) reads for output row q, column j
      const int dr = PHW >= 0 ? (ph_t + 1 - kh) / 2 + 1 : (MODE == 0 ? kh : 2 - kh);
      // fragment reads software-pipelined inside the phase: tap kw + 1 is read while tap kw multiplies (PIPE above)
      if constexpr (PIPE) {
        H2Unit fa[2][2][MT], fb[2][2][NT];
        auto rd = [&](int kw, H2Unit (&av)[2][MT], H2Unit (&bv)[2][NT]) {
          const int dc = MODE == 0 ? kw : 2 - kw;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[p][mt].u = wA[buf * WUS + g * WU + ((mt * 3 + kw) * 2 + p) * 64];
#pragma unroll
            for (int q = 0; q < NT; ++q) bv[p][q].u = xB[p * 2 * PIX + (q * RPQ + dr) * IW + dc];
          }
        };
        rd(0, fa[0], fb[0]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          __builtin_amdgcn_sched_barrier(0);
          if (kw + 1 < 3) rd(kw + 1, fa[(kw + 1) & 1], fb[(kw + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
#ifdef H2_SINGLE
              f32x16 cx = acc[mt][q];
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kw & 1][1][mt].b, fb[kw & 1][0][q].b, cx, 0, 0, 0);
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kw & 1][0][mt].b, fb[kw & 1][1][q].b, cx, 0, 0, 0);
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kw & 1][0][mt].b, fb[kw & 1][0][q].b, cx, 0, 0, 0);
              acc[mt][q] = cx;
#else
              f32x16 cx = acx[mt][q];   // same products, same order and same accumulators as the plain loop below
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kw & 1][1][mt].b, fb[kw & 1][0][q].b, cx, 0, 0, 0);
              cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kw & 1][0][mt].b, fb[kw & 1][1][q].b, cx, 0, 0, 0);
              acx[mt][q] = cx;
              acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kw & 1][0][mt].b, fb[kw & 1][0][q].b, acc[mt][q], 0, 0, 0);
#endif
            }
        }
      } else
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if (PHW >= 0 && !ALLP && ((kw + PW) & 1) == 0) continue;   // this parity's taps only: kw = pw + 1 (mod 2)
        const int pw_t = ALLP ? ((kw + 1) & 1) : PW;
        const int par = ph_t * 2 + pw_t;   // (fused parities) the accumulator tile of this tap
        // two waves share a SIMD: the partner's MFMAs cover this wave's fragment reads, so nothing is gained by
        // letting the scheduler hoist the next tap's 4*(MT+NT) fragment registers above this tap's MFMAs
#ifndef H2_NO_TAP_BARRIER
        if constexpr (MT * NT < 8) __builtin_amdgcn_sched_barrier(0);
#endif
        const int dc = PHW >= 0 ? (pw_t + 1 - kw) / 2 + 1 : (MODE == 0 ? kw : 2 - kw);
        H2Unit av[2][MT], bv[2][NT];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[p][mt].u = wA[buf * WUS + g * WU + ((mt * 3 + kw) * 2 + p) * 64];
#pragma unroll
          for (int q = 0; q < NT; ++q) bv[p][q].u = xB[p * 2 * PIX + (q * RPQ + dr) * IW + dc];
        }
#ifdef H2_SETPRIO   // (experiment, tools/ab_build.sh: raised issue priority around the tap's MFMA cluster)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            const int qa = ALLP ? par : q;   // accumulator tile (compile-time after unrolling)
#ifdef H2_SINGLE
            f32x16 cx = acc[mt][qa];
            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0][q].b, cx, 0, 0, 0);
            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1][q].b, cx, 0, 0, 0);
            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0][q].b, cx, 0, 0, 0);
            acc[mt][qa] = cx;
#else
            f32x16 cx = acx[mt][qa];
            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1][mt].b, bv[0][q].b, cx, 0, 0, 0);
            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[1][q].b, cx, 0, 0, 0);
            acx[mt][qa] = cx;
            acc[mt][qa] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0][mt].b, bv[0][q].b, acc[mt][qa], 0, 0, 0);
#endif
          }
#ifdef H2_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#ifndef H2_ABL_NOX
      if constexpr (!LAST) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
          if (i * NKH / NX == ki) {
            issue_mask(ch + 1, i);
            write_x(ch + 1, i);
          }
      }
#endif
      if (stage_end && more_w) {
        h2_dma_wait();
        __syncthreads();
      }
    }
  };
  for (int ch = 0; ch + 1 < nch; ++ch) {
    chunk(ch, std::false_type{});
#ifdef H2_TSTAMP
    if (ch == 0) H2_STAMP(6);
#endif
  }
  chunk(nch - 1, std::true_type{});
  H2_STAMP(3);

  // What the output store costs (r02, tools/time_conv.py with -DH2_ABL_NOSTORE): without it every layer runs at ~305
  // TFLOP/s whatever its K; with 4-byte stores (lane = pixel, one store instruction per accumulator register) the
  // short-K layers paid 15 % (C = 128) to 36 % (C = 32 at 256^2: 135 us, 87 us without the store) -- ~2.8 TB/s of output
  // wherever the stores were placed in time.  It is the per-CU memory path, ~40 cycles per store INSTRUCTION whatever
  // its width: an address pattern of 16-byte stores (same bytes, a quarter of the instructions) recovered half of it, so
  // the epilogue now transposes 4 pixels x 4 channels blocks across lane quads and stores dwordx4 (conv_common.h:
  // store_tile_side4; conv2_2 281 -> 270 us, conv1_2 353 -> 335, the 64-channel RNB conv 82 -> 74, C = 32 131 -> 121).
  // Measured and discarded on the way (no gain): persistent workgroups prefetching the next tile's first chunk across
  // the epilogue; 512-thread workgroups with whole-chunk weight stages; starting the first round of workgroups in eight
  // phases spread over a tile period; non-temporal stores; s_setprio 1 around every tap's MFMA cluster (-DH2_SETPRIO, r03:
  // within +-3 % on every layer shape of tools/time_conv.py).
  // r03, tools/h2_timeline.py (wall-clock stamps per workgroup): a tile of the 64 x 8 form spends 2.5 us reducing the maxima
  // behind its first loads, ~1 us converting them, 5.3 - 5.6 us per 16-channel chunk and 7 - 8 us in the epilogue below --
  // VALU-bound on per-element scalar branches over the activation code, not on the stores (removing the stores or the
  // lane transposes changed nothing).  Straight-line forms of the common cases (store_tile_side4's FORM) took it to ~5 us:
  // 64-channel layers -5 %, the others -1 .. 3 %.  A tile-WALKING form (one workgroup per residency slot, the next tile's
  // first chunk staged behind this tile's last; bit-identical) was built for the 32-channel layers, whose tiles wait 4.5 of
  // their 11.5 us for the first loads -- and ran 20 - 30 % SLOWER (105 -> 126 - 140 us): every chunk then stages, and a
  // staging chunk takes 5.4 us against 1.4 for one that does not; requesting all of a chunk's rounds at its start instead
  // of one per kernel row did not change either form.  These layers move 268 - 400 MB at 2.7 - 2.8 TB/s whatever the
  // schedule.  Removed.
  // ---- epilogue (arithmetic shared with the fp32 kernels): lane j = pixel (row0 + wave*NT + q, col0 + j).  The MT * NT
  // tiles are named at compile time (a runtime index would put the accumulators in scratch) and software-pipelined: the
  // residual / aux / shift loads of tile t+1 are issued before the stores of tile t (conv_common.h: load_tile_side).
  auto geo = [&](int q) {
    PixGeo g;
    g.n = n;
    const int orow = row0 + (wave * NT + q) * RPQ + jr, ocol = col0 + jc;
    g.oh = PHW >= 0 ? 2 * orow + PH : orow;   // parity phase: every other pixel of dx
    g.ow = PHW >= 0 ? 2 * ocol + PW : ocol;   // (PHW == 4 never gets here: its epilogue is above)
    g.valid = true;
    return g;
  };
  auto combined = [&](auto tc) {
    constexpr int t = decltype(tc)::value, q = t / MT, mt = t % MT;
    f32x16 c;
#pragma unroll
#ifdef H2_SINGLE
    for (int r = 0; r < 16; ++r) c[r] = acc[mt][q][r] * descale * descale2;
#else
    for (int r = 0; r < 16; ++r) c[r] = (acc[mt][q][r] + acx[mt][q][r] * (1.f / 2048.f)) * descale * descale2;
#endif
    return c;
  };
  float ymax = 0.f;   // max |y| over what this lane stores (published below if the caller asked for it)
  auto publish = [&]() {
    if (a.amax_out) {
      const float m_ = wave_max(ymax);
      if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(a.amax_out) + ((blockIdx.x * 4u + wave) & 511u), __float_as_uint(m_));
    }
  };
  if constexpr (ALLP) {   // fused parities: lane = dy pixel (orow, ocol) -> dx pixels (2 orow + ph, 2 ocol + {0, 1})
    const int orow = row0 + wave + jr, ocol = col0 + jc;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        f32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifdef H2_SINGLE
          c0[r] = acc[mt][2 * ph][r] * descale * descale2;
          c1[r] = acc[mt][2 * ph + 1][r] * descale * descale2;
#else
          c0[r] = (acc[mt][2 * ph][r] + acx[mt][2 * ph][r] * (1.f / 2048.f)) * descale * descale2;
          c1[r] = (acc[mt][2 * ph + 1][r] + acx[mt][2 * ph + 1][r] * (1.f / 2048.f)) * descale * descale2;
#endif
        }
        ymax = fmaxf(ymax, store_tile_pair(a, n, 2 * orow + ph, 2 * ocol, m0 + 32 * mt, h, c0, c1));
      }
    publish();
    return;
  }
  if (PHW < 0 && a.wide) {   // 16-byte epilogue: four lanes transpose their 4 pixels x 4 channels blocks (conv_common.h)
    const int k = lane & 3;
    TileSide4 side[2];
    auto tile = [&](auto tc, auto form_c) {
      constexpr int t = decltype(tc)::value, q = t / MT, mt = t % MT, FORM = decltype(form_c)::value;
      if constexpr (t == 0) load_tile_side4<MODE, FORM>(a, geo(0), m0, h, k, side[0]);
      if constexpr (t + 1 < MT * NT)
        load_tile_side4<MODE, FORM>(a, geo((t + 1) / MT), m0 + 32 * ((t + 1) % MT), h, k, side[(t + 1) & 1]);
#ifdef H2_ABL_NOSTORE   // timing ablation (tools/ab_build.sh): the output is not written
      if (descale == 12345.f)
#endif
      ymax = fmaxf(ymax, store_tile_side4<MODE, FORM>(a, geo(q), m0 + 32 * mt, h, k, combined(tc), side[t & 1]));
    };
    auto tiles = [&](auto form_c) {
      tile(std::integral_constant<int, 0>{}, form_c);
      if constexpr (MT * NT > 1) tile(std::integral_constant<int, 1>{}, form_c);
      if constexpr (MT * NT > 2) {
        tile(std::integral_constant<int, 2>{}, form_c);
        tile(std::integral_constant<int, 3>{}, form_c);
      }
      if constexpr (MT * NT > 4) {
        tile(std::integral_constant<int, 4>{}, form_c);
        tile(std::integral_constant<int, 5>{}, form_c);
        tile(std::integral_constant<int, 6>{}, form_c);
        tile(std::integral_constant<int, 7>{}, form_c);
      }
    };
    // ONE wave-uniform branch around the whole epilogue: the common cases run straight-line code (conv_common.h:
    // store_tile_side4's FORM).  The side loads of the first tile sit behind the branch -- issued before it, their
    // registers were live across all four forms and the data-gradient kernels spilled 264 bytes in the epilogue.
    const int form = side4_form<MODE>(a);
    if (form == 1) tiles(std::integral_constant<int, 1>{});
    else if (MODE == 1 && form == 2) tiles(std::integral_constant<int, MODE == 1 ? 2 : 0>{});
    else if (MODE == 1 && form == 3) tiles(std::integral_constant<int, MODE == 1 ? 3 : 0>{});
    else tiles(std::integral_constant<int, 0>{});
    static_assert(MT * NT <= 8, "epilogue tiles are named at compile time");
    publish();
#ifdef H2_TSTAMP
    H2_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H2_STAMP(5);
#endif
    return;
  }
  TileSide side[2];
  load_tile_side(a, geo(0), m0, h, side[0]);
  auto tile = [&](auto tc) {
    constexpr int t = decltype(tc)::value, q = t / MT, mt = t % MT;
    if constexpr (t + 1 < MT * NT) load_tile_side(a, geo((t + 1) / MT), m0 + 32 * ((t + 1) % MT), h, side[(t + 1) & 1]);
    ymax = fmaxf(ymax, store_tile_side(a, geo(q), m0 + 32 * mt, h, combined(tc), side[t & 1]));
  };
  tile(std::integral_constant<int, 0>{});
  if constexpr (MT * NT > 1) tile(std::integral_constant<int, 1>{});
  if constexpr (MT * NT > 2) {
    tile(std::integral_constant<int, 2>{});
    tile(std::integral_constant<int, 3>{});
  }
  if constexpr (MT * NT > 4) {
    tile(std::integral_constant<int, 4>{});
    tile(std::integral_constant<int, 5>{});
    tile(std::integral_constant<int, 6>{});
    tile(std::integral_constant<int, 7>{});
  }
  publish();
}

template <int MT, int NT, int MODE, int PRO, int PHW, int NWV, int TW = 32>
static int launch_h2_one(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, hipStream_t st) {
  constexpr int RPQ = 32 / TW;
  constexpr int PIX = (NWV * NT * RPQ + 2) * (TW == 32 ? 34 : 32);
  constexpr int NKH = (PHW < 0 || PHW == 4) ? 3 : ((PHW >> 1) ? 2 : 1);
  constexpr size_t lds = (size_t)(8 * PIX + 2 * H2_SLAB * MT * (NWV == 8 ? NKH : 1)) * 16;
  const vunet_conv_desc& d = ga.d;
  const int blocks = d.N * (d.Hs / (NWV * NT * RPQ)) * (d.Ws / TW) * ((d.M + 32 * MT - 1) / (32 * MT));
  if (!h2_launch_allowed()) return VUNET_ERR_UNSUPPORTED;
  auto kern = conv_h2_kernel<MT, NT, MODE, PRO, PHW, NWV, TW>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VUNET_ERR_LAUNCH;
  VUNET_LAUNCH(kern, dim3((unsigned)blocks), dim3(64 * NWV), lds, st, ga, (const uint4*)wx, mtiles_pad, amax);
  return vunet_check_launch();
}

// 16-wide maps: forward (no prologue / ELU / ELU + dropout) and data gradient (plain / ReLU mask), NT = 1
template <int MT>
static int launch_h2_w16(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, hipStream_t st) {
  if (ga.d.mode == 1) {
    if (ga.d.stride != 1) return VUNET_ERR_UNSUPPORTED;
    if (pro == 4) return launch_h2_one<MT, 1, 1, 4, -1, 4, 16>(ga, wx, mtiles_pad, amax, st);
    if (pro == 0) return launch_h2_one<MT, 1, 1, 0, -1, 4, 16>(ga, wx, mtiles_pad, amax, st);
    return VUNET_ERR_UNSUPPORTED;
  }
  switch (pro) {
    case 0: return launch_h2_one<MT, 1, 0, 0, -1, 4, 16>(ga, wx, mtiles_pad, amax, st);
    case 1: return launch_h2_one<MT, 1, 0, 1, -1, 4, 16>(ga, wx, mtiles_pad, amax, st);
    case 2: return launch_h2_one<MT, 1, 0, 2, -1, 4, 16>(ga, wx, mtiles_pad, amax, st);
    default: return VUNET_ERR_UNSUPPORTED;
  }
}

template <int MT, int NT, int NWV = 4>
static int launch_h2(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, hipStream_t st) {
  if (ga.d.mode == 1 && ga.d.stride == 2) {   // one launch per output parity (tests / A-B: VUNET_TUNE_PARITY_LAUNCHES)
    if (pro != 0) return VUNET_ERR_UNSUPPORTED;
    int rc = launch_h2_one<MT, NT, 1, 0, 0, NWV>(ga, wx, mtiles_pad, amax, st);
    if (rc == VUNET_OK) rc = launch_h2_one<MT, NT, 1, 0, 1, NWV>(ga, wx, mtiles_pad, amax, st);
    if (rc == VUNET_OK) rc = launch_h2_one<MT, NT, 1, 0, 2, NWV>(ga, wx, mtiles_pad, amax, st);
    if (rc == VUNET_OK) rc = launch_h2_one<MT, NT, 1, 0, 3, NWV>(ga, wx, mtiles_pad, amax, st);
    return rc;
  }
  if (ga.d.mode == 1) {
    if (pro == 4) return launch_h2_one<MT, NT, 1, 4, -1, NWV>(ga, wx, mtiles_pad, amax, st);
    if (pro == 0) return launch_h2_one<MT, NT, 1, 0, -1, NWV>(ga, wx, mtiles_pad, amax, st);
    return VUNET_ERR_UNSUPPORTED;
  }
  switch (pro) {
    case 0: return launch_h2_one<MT, NT, 0, 0, -1, NWV>(ga, wx, mtiles_pad, amax, st);
    case 1: return launch_h2_one<MT, NT, 0, 1, -1, NWV>(ga, wx, mtiles_pad, amax, st);
    case 2: return launch_h2_one<MT, NT, 0, 2, -1, NWV>(ga, wx, mtiles_pad, amax, st);
    default: return VUNET_ERR_UNSUPPORTED;
  }
}
