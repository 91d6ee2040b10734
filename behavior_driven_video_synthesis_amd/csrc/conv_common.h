// Shared pieces of the convolution kernels (conv_gather.hip, conv_tiled.hip): launch arguments,
// pixel decomposition, the prologue (input activation / dropout) and the fused epilogue.
#pragma once
#include "common.h"

struct GatherArgs {
  vunet_conv_desc d;
  const float* x1;
  const float* x2;
  const float* wt;
  const float* shift;
  const float* res;
  const float* aux;
  float* y;
  int NP, HoWo, HsWs;
  InAct in1, in2, auxa;
  // internal (not part of the C ABI): output-parity phase of a strided transposed gather.  ph < 0: off.
  // With ph/pw set, pixel indices enumerate the sub-grid (n, a, b) of outputs (s*a + ph, s*b + pw), NP counts
  // that sub-grid, and only the taps whose parity matches are visited -- no MFMA is spent on structural zeros.
  int ph, pw, subW, subHW;
  // internal: y / res / aux are 16-byte aligned, no depth-to-space store, no output-parity phase: the split kernels may
  // use the lane-transposed 16-byte epilogue (load_tile_side4 / store_tile_side4)
  int wide;
  // optional (split-fp16 kernel): 512 zero-initialised floats that receive partial maxima of |y| (atomic max on the bit
  // patterns: |y| >= 0), so that the consumer of y needs no pass of its own over it (vunet_absmax_partials).  nullptr: off.
  float* amax_out;
  // internal: data gradient through a ReLU epilogue -- x1 (= dy) is multiplied by [mask > 0] while it is staged
  // (prologue code 4; mask = the forward output, shaped like x1).  nullptr: off.
  const float* mask;
  // internal (fp16 scheme): the second source's 512 partial |x| maxima when they live in a buffer of their own (the kernel's
  // `amax` argument then holds only the first source's); nullptr: `amax` holds all 1024.
  const float* amax2;
  // internal (data gradient): a SECOND tensor added in the epilogue, shaped like y -- the gradient that reaches the layer's
  // input through another reader (ops.py: pass-through alias) when the one residual slot is taken by dy itself (a
  // VunetRNB's x is both the convolution's source and the residual).  nullptr: off.  Mode 1 only.
  const float* res2;
};

struct PixGeo {
  int n, oh, ow;
  bool valid;
};

__device__ __forceinline__ PixGeo decompose(int P, int NP, int HoWo, int Wo) {
  PixGeo g;
  g.valid = P < NP;
  const int Pc = g.valid ? P : 0;
  g.n = Pc / HoWo;
  const int rem = Pc - g.n * HoWo;
  g.oh = rem / Wo;
  g.ow = rem - g.oh * Wo;
  return g;
}

// PRO: 0 = no prologue activation, 1 = ELU, 2 = ELU + dropout, 3 = generic (runtime InAct)
template <int PRO>
__device__ __forceinline__ float prologue(const InAct& a, float v, uint32_t idx) {
  if (PRO == 0) return v;
  if (PRO == 1) return elu_f(v);
  if (PRO == 2) {
    v = elu_f(v);
    return (vunet_hash_u32(idx + inact_seed(a)) >= a.thresh) ? v * a.keep_scale : 0.f;
  }
  return apply_in_act(a, v, idx);
}

// epilogue of one accumulator element -> the value stored
__device__ __forceinline__ float store_out(const GatherArgs& a, const PixGeo& g, int m, float v) {
  const vunet_conv_desc& d = a.d;
  const int pix = g.oh * d.Wo + g.ow;
  if (d.mode == 0) {
    if (a.shift) v += a.shift[m];
    if (d.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (d.out_act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
    else if (d.out_act == ACT_ELU) v = elu_f(v);
    else if (d.out_act == ACT_LRELU) v = v > 0.f ? v : v * d.in_slope;
    size_t o;
    if (d.d2s) {
      const int Cq = d.M >> 2, blk = m / Cq, c = m - blk * Cq;
      o = ((size_t)(g.n * Cq + c) * (2 * d.Ho) + (2 * g.oh + (blk >> 1))) * (2 * d.Wo) + 2 * g.ow + (blk & 1);
    } else {
      o = (size_t)(g.n * d.M + m) * a.HoWo + pix;
    }
    if (a.res) v += a.res[o];
    a.y[o] = v;
  } else {
    const size_t o = (size_t)(g.n * d.M + m) * a.HoWo + pix;
    if (a.aux) v *= in_act_grad(a.auxa, a.aux[o], (uint32_t)o);
    if (a.res) v += a.res[o];
    if (a.res2) v += a.res2[o];
    a.y[o] = v;
  }
  return v;
}


// Epilogue of one 32x32 accumulator tile held by a wave (16 registers per lane: rows (r&3)+8(r>>2)+4h of the
// m-tile, lane = pixel), in two halves so that a kernel can issue the side loads of tile t+1 BEFORE the stores of tile t:
// on gfx9 stores count in vmcnt like loads, so a load issued behind 16 stores is only "returned" (s_waitcnt) once those
// stores have been acknowledged -- a serial load -> store -> load -> store chain costs a write round trip per tile.
//   load_tile_side   issues every residual / aux / shift load of the tile (no per-element branch or wait)
//   store_tile_side  computes the 16 results and stores them
// m_tile0 = first channel of the m-tile.  (The sub-pixel store, d.d2s, is scattered and rare: one-pass store_out.)
struct TileSide {
  float aux[16], res[16], sh[16];
};

__device__ __forceinline__ void load_tile_side(const GatherArgs& a, const PixGeo& g, int m_tile0, int h, TileSide& s) {
  const vunet_conv_desc& d = a.d;
  if (d.d2s) return;
  const int mb = m_tile0 + 4 * h;
  const int pix = g.oh * d.Wo + g.ow;
  const size_t o0 = (size_t)(g.n * d.M + mb) * a.HoWo + pix;
  const size_t rs = (size_t)a.HoWo;
  const bool full = m_tile0 + 32 <= d.M;  // wave-uniform
  bool ok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ok[r] = g.valid && (full || mb + (r & 3) + 8 * (r >> 2) < d.M);
  if (a.aux) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s.aux[r] = a.aux[ok[r] ? o0 + ((r & 3) + 8 * (r >> 2)) * rs : 0];
  }
  if (a.res) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s.res[r] = a.res[ok[r] ? o0 + ((r & 3) + 8 * (r >> 2)) * rs : 0];
  }
  if (d.mode == 0 && a.shift) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s.sh[r] = a.shift[ok[r] ? mb + (r & 3) + 8 * (r >> 2) : 0];
  }
}

// -> max |stored value| of this lane
__device__ __forceinline__ float store_tile_side(const GatherArgs& a, const PixGeo& g, int m_tile0, int h,
                                                 const f32x16& acc, const TileSide& s) {
  const vunet_conv_desc& d = a.d;
  float vmax = 0.f;
  if (d.d2s) {  // sub-pixel store (up-convs, lib/modules.py: Upsample + DepthToSpace): scattered addresses
    const int Cq = d.M >> 2;
#ifdef VUNET_AB_NO_D2S_FAST   // (A/B baseline, tools/ab_build.sh)
    if (false) {
#else
    if ((Cq & 31) == 0 && m_tile0 + 32 <= d.M && g.valid && (d.out_act == ACT_NONE || d.out_act == ACT_RELU)) {
#endif
      // a 32-channel m-tile lies inside ONE sub-pixel block: (dy, dx) and the first output channel are tile-uniform,
      // so the per-element work is an add and a store (store_out divides by the runtime Cq for every element: the
      // up-conv layers ran 20 - 25 % below their stride-1 neighbours, r03 profiles/r03_layers_1stream.txt)
      const int blk = m_tile0 / Cq, c0 = m_tile0 - blk * Cq + 4 * h;
      const size_t rs = (size_t)(2 * d.Ho) * (2 * d.Wo);   // output channel stride
      const size_t o0 = ((size_t)(g.n * Cq + c0) * (2 * d.Ho) + (2 * g.oh + (blk >> 1))) * (2 * d.Wo) + 2 * g.ow + (blk & 1);
      const bool relu = d.out_act == ACT_RELU;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cr = (r & 3) + 8 * (r >> 2);
        const size_t o = o0 + cr * rs;
        float v = acc[r];
        if (a.shift) v += a.shift[m_tile0 + 4 * h + cr];
        v = (relu && !(v > 0.f)) ? 0.f : v;
        if (a.res) v += a.res[o];
        a.y[o] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
      return vmax;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_tile0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (g.valid && m < d.M) vmax = fmaxf(vmax, fabsf(store_out(a, g, m, acc[r])));
    }
    return vmax;
  }
  const int mb = m_tile0 + 4 * h;
  const int pix = g.oh * d.Wo + g.ow;
  const size_t o0 = (size_t)(g.n * d.M + mb) * a.HoWo + pix;
  const size_t rs = (size_t)a.HoWo;
  const bool full = m_tile0 + 32 <= d.M;  // wave-uniform
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const size_t o = o0 + ((r & 3) + 8 * (r >> 2)) * rs;
    const bool ok = g.valid && (full || mb + (r & 3) + 8 * (r >> 2) < d.M);
    float v = acc[r];
    if (d.mode == 0) {
      if (a.shift) v += s.sh[r];
      if (d.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (d.out_act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
      else if (d.out_act == ACT_ELU) v = elu_f(v);
      else if (d.out_act == ACT_LRELU) v = v > 0.f ? v : v * d.in_slope;
    } else if (a.aux) {
      v *= in_act_grad(a.auxa, s.aux[r], (uint32_t)o);
    }
    if (a.res) v += s.res[r];
    if (d.mode == 1 && a.res2 && ok) v += a.res2[o];
    if (ok) {
      a.y[o] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
  }
  return vmax;
}

// ---- the same epilogue with 16-byte accesses.  An accumulator tile has lane = pixel, register = channel, so a plain
// store writes 4 bytes per lane: 64 store instructions per wave and tile pair, and the per-CU store path takes ~40 cycles
// for each whatever its width (measured: the output store costs 15 - 36 % of the split kernels, tools/time_conv.py).
// Each group of four lanes (four consecutive pixels) therefore TRANSPOSES its 4 pixels x 4 channels block through two
// DPP exchanges (lane ^ 1, lane ^ 2), after which lane k of the group holds four consecutive pixels of channel
// base + k: one global_store_dwordx4 (and one dwordx4 load of the residual / aux tensor) per four registers.
//   k = lane & 3;  channels of register group q4 (registers 4 q4 .. 4 q4 + 3): m_tile0 + 8 q4 + 4 h + {0..3}
// Not for the depth-to-space store or the output-parity phases (their pixels are not contiguous): GatherArgs::wide.
__device__ __forceinline__ float dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
// in: lane k holds (r0..r3) = row k of a 4x4 block spread over four lanes; out: lane k holds column k
__device__ __forceinline__ void quad_transpose(float& r0, float& r1, float& r2, float& r3, int k) {
  const bool odd = k & 1, hi = k & 2;
  const float g0 = dpp_xor1(odd ? r0 : r1), g1 = dpp_xor1(odd ? r2 : r3);
  r0 = odd ? g0 : r0;
  r1 = odd ? r1 : g0;
  r2 = odd ? g1 : r2;
  r3 = odd ? r3 : g1;
  const float u0 = dpp_xor2(hi ? r0 : r2), u1 = dpp_xor2(hi ? r1 : r3);
  r0 = hi ? u0 : r0;
  r1 = hi ? u1 : r1;
  r2 = hi ? r2 : u0;
  r3 = hi ? r3 : u1;
}

__device__ __forceinline__ float plain_out(float t, bool relu) { return (relu && !(t > 0.f)) ? 0.f : t; }

struct TileSide4 {
  float4 aux[4], res[4];
  float sh[4];
};

// g: the pixel of THIS lane (as for store_tile_side); the group's four pixels start at g.ow - (lane & 3)
template <int MODE, int FORM = 0>   // FORM: see store_tile_side4 (1: there is no aux tensor)
__device__ __forceinline__ void load_tile_side4(const GatherArgs& a, const PixGeo& g, int m_tile0, int h, int k, TileSide4& s) {
  const vunet_conv_desc& d = a.d;
  const size_t pix0 = (size_t)g.oh * d.Wo + (g.ow - k);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const int ch = m_tile0 + 8 * q4 + 4 * h + k;
    const bool ok = ch < d.M;
    const size_t o = ok ? (size_t)(g.n * d.M + ch) * a.HoWo + pix0 : 0;
    if (MODE == 1 && FORM != 1 && a.aux) s.aux[q4] = *reinterpret_cast<const float4*>(a.aux + o);
    if (a.res) s.res[q4] = *reinterpret_cast<const float4*>(a.res + o);
    if (MODE == 0 && a.shift) s.sh[q4] = a.shift[ok ? ch : 0];
  }
}

// FORM: the caller picks it once for all its tiles with side4_form() (a wave-uniform branch around the whole epilogue):
//   0 general; 1 forward with no / ReLU output activation, data gradient without an activation derivative;
//   2 data gradient * ELU'(aux); 3 the same with the dropout mask of the forward pass
template <int MODE>
__device__ __forceinline__ int side4_form(const GatherArgs& a) {
#ifdef H2_EPI_GENERIC   // (A/B baseline for tools/ab_build.sh: always the general form)
  return 0;
#endif
  if (MODE == 0) return (a.d.out_act == ACT_NONE || a.d.out_act == ACT_RELU) ? 1 : 0;
  if (a.aux == nullptr) return 1;
  if (a.auxa.act == ACT_ELU) return a.auxa.thresh ? 3 : 2;
  return 0;
}
template <int MODE, int FORM = 0>
__device__ __forceinline__ float store_tile_side4(const GatherArgs& a, const PixGeo& g, int m_tile0, int h, int k, f32x16 acc,
                                                  const TileSide4& s) {   // -> max |stored value| of this lane
  float vmax = 0.f;
  const vunet_conv_desc& d = a.d;
  const size_t pix0 = (size_t)g.oh * d.Wo + (g.ow - k);
  // The common cases as straight-line code (r03, tools/h2_timeline.py: the general form below spends 7 - 8 us per workgroup
  // tile in per-element scalar branches on the activation code -- as long as 1.5 K-chunks of the matrix loop):
  //   forward with no / ReLU output activation (every NormConv2d of the VUnet, every VGG19 layer); data gradient without
  //   an activation derivative (the layers whose input is not pre-activated, and the ReLU-masked VGG19 chain).
  // second residual (data gradient only): requested here, added below -- the lane transposes cover most of its latency, and
  // it needs no second prefetched register set in the kernels' software-pipelined epilogues
  float4 r2[4];
  const bool has_r2 = MODE == 1 && a.res2 != nullptr;
  if (has_r2) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int ch = m_tile0 + 8 * q4 + 4 * h + k;
      r2[q4] = *reinterpret_cast<const float4*>(a.res2 + (ch < d.M ? (size_t)(g.n * d.M + ch) * a.HoWo + pix0 : 0));
    }
  }
  if constexpr (FORM != 0) {
    const bool relu = MODE == 0 && d.out_act == ACT_RELU;
    const bool has_sh = MODE == 0 && a.shift != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      float v[4] = {acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
#ifndef H2_ABL_EPI_NOTRANSPOSE
      quad_transpose(v[0], v[1], v[2], v[3], k);
#endif
      const int ch = m_tile0 + 8 * q4 + 4 * h + k;
      const size_t o = (size_t)(g.n * d.M + ch) * a.HoWo + pix0;
      const float4 rs = has_res ? s.res[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float resv[4] = {rs.x, rs.y, rs.z, rs.w};
      if constexpr (FORM == 1) {
        const float sh = has_sh ? s.sh[q4] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = plain_out(v[e] + sh, relu) + resv[e];
      } else {
        const float auxv[4] = {s.aux[q4].x, s.aux[q4].y, s.aux[q4].z, s.aux[q4].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float gr = elu_grad_f(auxv[e]);
          if constexpr (FORM == 3)   // (a.auxa was resolved at kernel entry: the seed is a register)
            gr = (vunet_hash_u32((uint32_t)(o + e) + a.auxa.seed) >= a.auxa.thresh) ? gr * a.auxa.keep_scale : 0.f;
          v[e] = v[e] * gr + resv[e];
        }
      }
      if (has_r2) {
        v[0] += r2[q4].x; v[1] += r2[q4].y; v[2] += r2[q4].z; v[3] += r2[q4].w;
      }
      if (ch < d.M) {
#ifdef H2_ABL_EPI_NOSTORE
        if (v[0] == 12345.f)
#endif
        *reinterpret_cast<float4*>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
    return vmax;
  }
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    float v[4] = {acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
    quad_transpose(v[0], v[1], v[2], v[3], k);
    const int ch = m_tile0 + 8 * q4 + 4 * h + k;
    const size_t o = (size_t)(g.n * d.M + ch) * a.HoWo + pix0;
    const float auxv[4] = {s.aux[q4].x, s.aux[q4].y, s.aux[q4].z, s.aux[q4].w};
    const float resv[4] = {s.res[q4].x, s.res[q4].y, s.res[q4].z, s.res[q4].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e];
      if (MODE == 0) {
        if (a.shift) t += s.sh[q4];
        if (d.out_act == ACT_RELU) t = t > 0.f ? t : 0.f;
        else if (d.out_act == ACT_SIGMOID) t = 1.f / (1.f + __expf(-t));
        else if (d.out_act == ACT_ELU) t = elu_f(t);
        else if (d.out_act == ACT_LRELU) t = t > 0.f ? t : t * d.in_slope;
      } else if (a.aux) {
        t *= in_act_grad(a.auxa, auxv[e], (uint32_t)(o + e));
      }
      if (a.res) t += resv[e];
      v[e] = t;
    }
    if (has_r2) {
      v[0] += r2[q4].x; v[1] += r2[q4].y; v[2] += r2[q4].z; v[3] += r2[q4].w;
    }
    if (ch < d.M) {
      *reinterpret_cast<float4*>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
  }
  return vmax;
}

// Data-gradient epilogue for TWO horizontally adjacent output pixels per lane (the fused stride-2 form of conv_h2_kernel:
// parities pw = 0 / 1 of output row oh, columns ow, ow + 1 with ow even): * act'(aux) + res through 8-byte accesses; 32
// lanes cover 256 contiguous bytes of a channel row.  -> max |stored value| of this lane.
__device__ __forceinline__ float store_tile_pair(const GatherArgs& a, int n, int oh, int ow, int m_tile0, int h,
                                                 const f32x16& c0, const f32x16& c1) {
  const vunet_conv_desc& d = a.d;
  float vmax = 0.f;
  const size_t pix = (size_t)oh * d.Wo + ow;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = m_tile0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (ch < d.M) {
      const size_t o = (size_t)(n * d.M + ch) * a.HoWo + pix;
      float v0 = c0[r], v1 = c1[r];
      if (a.aux) {
        const float2 x = *reinterpret_cast<const float2*>(a.aux + o);
        v0 *= in_act_grad(a.auxa, x.x, (uint32_t)o);
        v1 *= in_act_grad(a.auxa, x.y, (uint32_t)(o + 1));
      }
      if (a.res) {
        const float2 x = *reinterpret_cast<const float2*>(a.res + o);
        v0 += x.x;
        v1 += x.y;
      }
      if (a.res2) {
        const float2 x = *reinterpret_cast<const float2*>(a.res2 + o);
        v0 += x.x;
        v1 += x.y;
      }
      *reinterpret_cast<float2*>(a.y + o) = make_float2(v0, v1);
      vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
    }
  }
  return vmax;
}

__device__ __forceinline__ void publish_amax(const GatherArgs& a, float vmax) {   // whole wave; see GatherArgs::amax_out
  const float m_ = wave_max(vmax);
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned*>(a.amax_out) + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & 511u), __float_as_uint(m_));
}

__device__ __forceinline__ void store_tile16(const GatherArgs& a, const PixGeo& g, int m_tile0, int h,
                                             const f32x16& acc) {
  TileSide s;
  load_tile_side(a, g, m_tile0, h, s);
  const float vmax = store_tile_side(a, g, m_tile0, h, acc, s);
  if (a.amax_out) publish_amax(a, vmax);   // (wave-uniform branch; the 1x1 / LDS-tiled kernels publish through here)
}

// vunet_conv2d_gather with the optional |y| maxima output: set for the kernels that publish (1x1, LDS-tiled), ignored by
// the others -- vunet_conv2d_publishes_amax tells the caller which it will be
int vunet_conv2d_gather_amax(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt, const float* shift,
                             const float* res, const float* aux, float* y, float* amax_out, void* stream);
bool vunet_conv2d_gather_publishes(const vunet_conv_desc* d, bool has_aux, bool has_res);
