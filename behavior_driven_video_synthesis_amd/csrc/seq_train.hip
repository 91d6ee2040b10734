// Training of the behaviour path (BASELINE config 4): the flow's maximum-likelihood step, experiments/behavior_net.py:703-714 with
// the optimiser of :384-395 and the loss of lib/losses.py:294-331.  Interface and data flow: include/vunet_seq_train.h.
//
// The flow of config/behavior_net.yaml is 60 MLPs of 512-2048-2048-2048-512: 629 M parameters, 2.5 GB of fp32 weights, trained at 64
// rows per step.  Per step and parameter the products are 3 x 64 multiply-adds (forward, input gradient, weight gradient) against
// 28 bytes of unavoidable traffic (W read by the forward pass; W, exp_avg, exp_avg_sq read and written by the update): the step is
// bound by HBM, 17.6 GB per step, and the matrix work (240 GFLOP of exact fp32, v_mfma_f32_16x16x4_f32) has to hide under it.
//
//   * seq_dx_kernel: dX = dZ . W.  The reduction runs over W's ROWS, so a workgroup owns a 64-column stripe of W (256 contiguous
//     bytes per row; a lane loads 16 of them and feeds one component to each of four matrix instructions whose accumulators hold
//     the columns 4 i + e) over M / S rows, its waves take 16-row groups round robin and add through LDS in wave order.  S is
//     chosen so that the launch has >= 256 workgroups; the S partial slabs are raw, and what reads them adds them in slab order:
//     seq_dz_finish_kernel for a hidden layer (with the LeakyReLU derivative), seq_coupling_bwd_kernel for an MLP's first layer.
//   * seq_dw_kernel: the weight gradient is never written.  A workgroup forms one 64 x 64 tile of dW = dZ^T . X from LDS-staged
//     row tiles of dZ and X, turns the accumulator layout into rows through LDS and applies torch.optim.Adam's update to the tile
//     of W / exp_avg / exp_avg_sq with 16-byte accesses -- one read and one write of each per step.  One launch covers any list of
//     layers (a device table of descriptors), so the update of a whole coupling half (eight layers) is a single launch of 4 - 20
//     thousand independent tiles.  hp == NULL: the tile of dW is stored instead (autograd's view of the same kernels).
//   * seq_coupling_bwd_kernel, seq_actnorm_bwd_kernel, seq_flow_loss_kernel: the pointwise / reduction remainder.
// Nothing uses float atomics; every sum has a fixed order: a step is bit-reproducible.
#include <type_traits>

#include "common.h"
#include "../../include/vunet_seq_train.h"

namespace {

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ float comp4(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// weights and moments are streamed: each byte is touched once per kernel by one workgroup
#ifdef VUNET_SEQ_NO_NT   // (timing-ablation build, tools/ab_build.sh)
#define SEQ_NT_LOAD(ptr) (*reinterpret_cast<const float4*>(ptr))
#define SEQ_NT_STORE(ptr, val) (*reinterpret_cast<float4*>(ptr) = (val))
#else
typedef float seq_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 seq_nt_load(const float* p) {
  const seq_f4 t = __builtin_nontemporal_load(reinterpret_cast<const seq_f4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void seq_nt_store(float* p, const float4& v) {
  seq_f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<seq_f4*>(p));
}
#define SEQ_NT_LOAD(ptr) seq_nt_load(ptr)
#define SEQ_NT_STORE(ptr, val) seq_nt_store(ptr, val)
#endif

// ------------------------------------------------------------------------------------------------ dX = dZ . W
struct SeqDxArgs {
  const float* w[2];   // [M][K]
  const float* dz;     // [nets][Bp][M]
  float* raw;          // [nets][S][Bp][K]
  int M, K, Bp, S, ldw;
  // FIN (vunet_seq_dx_finish): the workgroup that completes a column stripe's S slabs finishes the stripe
  const float* y;      // [nets][Bp][K] the previous layer's saved output
  float* dzp;          // [nets][Bp][K] its dZ
  int* cnt;            // [nets][K / 64] arrival counters (zero between launches)
  float slope;
};

// grid (K / 64, S, nets), 64 WAVES threads.  Lane l: i = l & 15, q = l >> 4.  In a 16-row group starting at row mb the lane loads
// W[mb + 4 q + e'][k0 + 4 i .. + 3] (e' = 0..3: four 16-byte loads; a wave instruction covers four rows of 256 bytes) and
// dZ[16 nb + i][mb + 4 q .. + 3].  MFMA (e', e, nb): A = component e of load e' (row i of A <-> column k0 + 4 i + e), B = component
// e' of the dZ load (column j of B <-> batch row 16 nb + j), both at reduction slot q <-> W row mb + 4 q + e'.
// D: lane holds dX[16 nb + (l & 15)][k0 + 4 (4 q + r) + e] in acc[e][nb][r].
template <int NB, int WAVES, bool FIN = false>
__global__ __launch_bounds__(64 * WAVES) void seq_dx_kernel(SeqDxArgs a) {
  __shared__ float4 red[WAVES][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 64, s = blockIdx.y, net = blockIdx.z;
  // slab s takes the 16-row groups [s G / S, (s + 1) G / S) of W's G = M / 16: equal ranges where S divides G, else ranges that
  // differ by one group (17 column stripes x 15 slabs = 255 workgroups for the LSTM's gate matrix: one round on 256 CUs)
  const int G = a.M >> 4, g_lo = (int)(((long)s * G) / a.S), ngrp = (int)(((long)(s + 1) * G) / a.S) - g_lo, row0 = 16 * g_lo;
  const float* __restrict__ w = a.w[net] + (size_t)(row0 + 4 * q) * a.ldw + k0 + 4 * i;
  const float* __restrict__ dz = a.dz + ((size_t)net * a.Bp + i) * a.M + row0 + 4 * q;
  f32x4 acc[4][NB];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[e][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef SEQ_DX_RING
#define SEQ_DX_RING 2
#endif
#ifndef SEQ_DX_RING_U
#define SEQ_DX_RING_U 1
#endif
  constexpr int SU = (WAVES <= 8 && SEQ_DX_RING > 1) ? SEQ_DX_RING_U : 1;   // row groups per stage of the ring below
  struct Stage {
    float4 wv[SU][4], dv[SU][NB];
  };
  auto load = [&](int g, auto nu_c, Stage& r) {
    constexpr int NU = decltype(nu_c)::value;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2)
        r.wv[u][e2] = *reinterpret_cast<const float4*>(w + (size_t)(16 * (g + WAVES * u) + e2) * a.ldw);   // (default policy: the sweep re-reads W)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) r.dv[u][nb] = *reinterpret_cast<const float4*>(dz + (size_t)16 * nb * a.M + 16 * (g + WAVES * u));
    }
  };
  auto mfma = [&](auto nu_c, const Stage& r) {
    constexpr int NU = decltype(nu_c)::value;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[e][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(r.wv[u][e2], e), comp4(r.dv[u][nb], e2), acc[e][nb], 0, 0, 0);
  };
  constexpr std::integral_constant<int, SU> su_c{};
  constexpr std::integral_constant<int, 1> one_c{};
  int g = wave;
  // Four- / eight-wave workgroups (48 - 64 batch rows) are one or two waves per SIMD: the wave keeps a ring of SEQ_DX_RING stages (of SEQ_DX_RING_U
  // row groups) in flight, the loads of stage i + SEQ_DX_RING - 1 issued before the matrix steps of stage i (csrc/seq.hip:
  // seq_linear_kernel's ring).  Same groups, same order, same sums as the plain loop below.
  if constexpr (WAVES <= 8 && SEQ_DX_RING == 2 && SEQ_DX_RING_U == 1) {
    // depth 2, any number of row groups (uneven slabs leave a wave 4 or 5 of them: the last one stays in the ring)
    const int nw = g < ngrp ? (ngrp - g + WAVES - 1) / WAVES : 0;
    if (nw > 0) {
      Stage ra, rb;
      load(g, su_c, ra);
      int k = 0;
      for (; k + 2 < nw; k += 2) {   // ra holds group k
        load(g + WAVES * (k + 1), su_c, rb);
        __builtin_amdgcn_sched_barrier(0);
        mfma(su_c, ra);
        __builtin_amdgcn_sched_barrier(0);
        load(g + WAVES * (k + 2), su_c, ra);
        __builtin_amdgcn_sched_barrier(0);
        mfma(su_c, rb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nw - k == 2) {
        load(g + WAVES * (k + 1), su_c, rb);
        __builtin_amdgcn_sched_barrier(0);
        mfma(su_c, ra);
        __builtin_amdgcn_sched_barrier(0);
        mfma(su_c, rb);
      } else {
        mfma(su_c, ra);
      }
      g += WAVES * nw;
    }
  } else if constexpr (WAVES <= 8 && SEQ_DX_RING > 1) {
    constexpr int D = SEQ_DX_RING;
    const int nw = g < ngrp ? (ngrp - g + WAVES - 1) / WAVES : 0;
    const int ns = nw / SU, n_main = ns - ns % D;
    if (n_main > 0) {
      Stage ring[D];
#pragma unroll
      for (int dd = 0; dd < D - 1; ++dd) load(g + WAVES * SU * dd, su_c, ring[dd]);
      int base = 0;
      for (; base + D < n_main; base += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          load(g + WAVES * SU * (base + dd + D - 1), su_c, ring[(dd + D - 1) % D]);
          __builtin_amdgcn_sched_barrier(0);
          mfma(su_c, ring[dd]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      load(g + WAVES * SU * (n_main - 1), su_c, ring[D - 1]);
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        __builtin_amdgcn_sched_barrier(0);
        mfma(su_c, ring[dd]);
      }
      g += WAVES * SU * n_main;
    }
  }
  for (; g < ngrp; g += WAVES) {
    Stage r;
    load(g, one_c, r);
    __builtin_amdgcn_sched_barrier(0);
    mfma(one_c, r);
  }
  // the waves' partial sums meet in LDS, one batch tile per round; waves 0..3 (thread t: r = t >> 6) add them in wave order
  float* out = a.raw + ((size_t)net * a.S + s) * a.Bp * a.K;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = make_float4(acc[0][nb][r], acc[1][nb][r], acc[2][nb][r], acc[3][nb][r]);
    __syncthreads();
    if (threadIdx.x < 256) {
      const int r = threadIdx.x >> 6;
      float4 v = red[0][r][lane];
#pragma unroll 4
      for (int p = 1; p < WAVES; ++p) v = add4(v, red[p][r][lane]);
      *reinterpret_cast<float4*>(out + (size_t)(16 * nb + i) * a.K + k0 + 16 * q + 4 * r) = v;
    }
    __syncthreads();
  }
  if constexpr (FIN) {
    // The stripe's S workgroups (one per row range of W) count their arrivals; the last one adds the S slabs IN SLAB ORDER --
    // whoever it is, the sum is the one seq_dz_finish_kernel forms -- applies LeakyReLU' and writes the previous layer's dZ.
    // Release / acquire at device scope around the counter: the slabs were written by workgroups on other XCDs (other L2s).
    __shared__ int last;
    __threadfence();
    __syncthreads();
    int* const cnt = a.cnt + net * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) last = atomicAdd(cnt, 1) == a.S - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) *cnt = 0;   // (the next launch that uses the counters starts after this kernel has ended)
    __threadfence();
    const size_t slab = (size_t)a.Bp * a.K;
    const float* r0 = a.raw + (size_t)net * a.S * slab + k0;
    for (int idx = threadIdx.x; idx < a.Bp * 16; idx += 64 * WAVES) {
      const size_t off = (size_t)(idx >> 4) * a.K + 4 * (idx & 15);
      float4 v = *reinterpret_cast<const float4*>(r0 + off);
      for (int p = 1; p < a.S; ++p) v = add4(v, *reinterpret_cast<const float4*>(r0 + (size_t)p * slab + off));
      const float4 yv = *reinterpret_cast<const float4*>(a.y + (size_t)net * slab + k0 + off);
      v.x *= yv.x > 0.f ? 1.f : a.slope;
      v.y *= yv.y > 0.f ? 1.f : a.slope;
      v.z *= yv.z > 0.f ? 1.f : a.slope;
      v.w *= yv.w > 0.f ? 1.f : a.slope;
      *reinterpret_cast<float4*>(a.dzp + (size_t)net * slab + k0 + off) = v;
    }
  }
}

// dz[n][b][k] = (sum_s raw[n][s][b][k]) * LeakyReLU'(y[n][b][k]); one float4 per thread
__global__ __launch_bounds__(256) void seq_dz_finish_kernel(const float* __restrict__ raw, const float* __restrict__ y,
                                                            float* __restrict__ dz, int S, size_t per_net4, float slope) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // float4 index inside one net
  if (idx >= per_net4) return;
  const int net = blockIdx.y;
  const float4* r4 = reinterpret_cast<const float4*>(raw) + (size_t)net * S * per_net4 + idx;
  const float4 yv = reinterpret_cast<const float4*>(y)[(size_t)net * per_net4 + idx];
  float4 t[8];   // (up to eight slabs requested together; added in slab order)
#pragma unroll
  for (int s = 0; s < 8; ++s) t[s] = s < S ? r4[(size_t)s * per_net4] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v = t[0];
#pragma unroll
  for (int s = 1; s < 8; ++s)
    if (s < S) v = add4(v, t[s]);
  for (int s = 8; s < S; ++s) v = add4(v, r4[(size_t)s * per_net4]);
  v.x *= yv.x > 0.f ? 1.f : slope;
  v.y *= yv.y > 0.f ? 1.f : slope;
  v.z *= yv.z > 0.f ? 1.f : slope;
  v.w *= yv.w > 0.f ? 1.f : slope;
  reinterpret_cast<float4*>(dz)[(size_t)net * per_net4 + idx] = v;
}

// ------------------------------------------------------------------------------------------------ one flow step, backwards
struct SeqCouplingBwdArgs {
  const float* gbase;
  const float* gslabs;
  const int* inv_map;
  const float* scale;
  const float* in;
  const float* st;
  const float* bias_s;
  const float* bias_t;
  const float* dld;
  float* gfull;
  float* gout;
  float* dzh;
  int B, Bp, C, c1, ld_g, ld_in, ld_out, ld_full, Mp, S, n_sl, ld_sl, c1s;
};

// grid (B, ceil(C / 256)), 256 threads: batch row b, coupling index j.  What depends on j alone (the head's slabs, its bias, the
// step's input, d loss / d logdet) is requested first, beside the un-shuffle index; the gradient's slabs follow the index.
__global__ __launch_bounds__(256) void seq_coupling_bwd_kernel(SeqCouplingBwdArgs a) {
  const int b = blockIdx.x, j = blockIdx.y * 256 + threadIdx.x;
  if (j >= a.C) return;
  const bool cpl = a.st && j >= a.c1;
  const int q = cpl ? j - a.c1 : 0;
  float sv[8], bs = 0.f, xk = 0.f, dld = 0.f;
  {
    const size_t slab = (size_t)a.Bp * a.Mp;
    const float* ps = cpl ? a.st + (size_t)b * a.Mp + q : a.gbase;   // (not read unless cpl)
#pragma unroll
    for (int p = 0; p < 8; ++p) sv[p] = (cpl && p < a.S) ? ps[p * slab] : 0.f;
    if (cpl) {
      if (a.S > 1) bs = a.bias_s[q];
      xk = a.in[(size_t)b * a.ld_in + j];
      dld = a.dld[b];
    }
  }
  const int c = a.inv_map ? a.inv_map[j] : j;
  const float sc = a.scale ? a.scale[c] : 1.f;
  float g = a.gbase[(size_t)b * a.ld_g + c];
  if (a.gslabs && c < a.c1s) {   // all loads of a group of eight in flight, added in slab order
    const float* ps = a.gslabs + (size_t)b * a.ld_sl + c;
    const size_t slab = (size_t)a.Bp * a.ld_sl;
    int n = 0;
    for (; n + 8 <= a.n_sl; n += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ps[(size_t)(n + u) * slab];
#pragma unroll
      for (int u = 0; u < 8; ++u) g += v[u];
    }
    for (; n < a.n_sl; ++n) g += ps[(size_t)n * slab];
  }
  if (a.gfull) a.gfull[(size_t)b * a.ld_full + c] = g;
  if (a.scale) g *= sc;                               // out = scale (u + loc)   (lib/modules.py:307)
  if (!cpl) {
    a.gout[(size_t)b * a.ld_out + j] = g;
    return;
  }
  float s = sv[0];
  if (a.S > 1) {   // the forward pass left raw slabs: the same sum in the same order (csrc/seq.hip, seq_coupling_kernel)
#pragma unroll
    for (int p = 1; p < 8; ++p)
      if (p < a.S) s += sv[p];
    s = tanhf(s + bs);
  }
  const float e = expf(s);
  a.gout[(size_t)b * a.ld_out + j] = g * e;           // x_ = x_k exp(s) + t   (models/flow/blocks.py:304)
  const float ds = g * xk * e + dld;                  // ... and logdet += sum(s)   (:306)
  a.dzh[(size_t)b * a.Mp + q] = ds * (1.f - s * s);   // s = tanh(.)   (lib/modules.py:252-253)
  a.dzh[((size_t)a.Bp + b) * a.Mp + q] = g;
}

// ------------------------------------------------------------------------------------------------ Adam
struct AdamResolved {
  float step_size, inv_sqrt_bc2, b1, b2, eps, wd;
};

__device__ __forceinline__ AdamResolved adam_resolve(const vunet_seq_adam_hp& hp) {
  if (hp.resolved_dev) {   // formed once per step by seq_adam_tick_kernel (the same expressions): two scalar loads
    AdamResolved r;
    r.step_size = hp.resolved_dev[0];
    r.inv_sqrt_bc2 = hp.resolved_dev[1];
    r.b1 = hp.beta1;
    r.b2 = hp.beta2;
    r.eps = hp.eps;
    r.wd = hp.weight_decay;
    return r;
  }
  // bias corrections in double, as torch forms them on the host (torch/optim/adam.py: bias_correction1 / 2, step_size)
  const double t = (double)*hp.step_dev;
  const double bc1 = 1.0 - pow((double)hp.beta1, t), bc2 = 1.0 - pow((double)hp.beta2, t);
  AdamResolved r;
  r.step_size = (float)(*hp.lr_dev / bc1);
  r.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  r.b1 = hp.beta1;
  r.b2 = hp.beta2;
  r.eps = hp.eps;
  r.wd = hp.weight_decay;
  return r;
}

__device__ __forceinline__ void adam_update(const AdamResolved& h, float g, float& p, float& m, float& v) {
  if (h.wd != 0.f) g += h.wd * p;
  m = h.b1 * m + (1.f - h.b1) * g;
  v = h.b2 * v + (1.f - h.b2) * g * g;
  p -= h.step_size * (m / (sqrtf(v) * h.inv_sqrt_bc2 + h.eps));
}

// ------------------------------------------------------------------------------------------------ dW = dZ^T . X (+ Adam)

// LDS tiles are [rows][64] floats with the column XOR-ed by 16 * (row & 3) (operand tiles: the four rows 4 c + q of a matrix
// step fall into four disjoint bank groups) resp. 16 * ((row >> 2) & 3) (the tile of dW on its way from the accumulator layout
// -- rows 4 q + r -- to rows): conflict-free without padding, 32 KB per workgroup at 64 batch rows = five workgroups per CU.
__device__ __forceinline__ int dw_sw(int row, int col) { return row * 64 + (col ^ ((row & 3) << 4)); }
__device__ __forceinline__ int dw_swo(int row, int col) { return row * 64 + (col ^ (((row >> 2) & 3) << 4)); }

// grid: one workgroup per 64 x 64 tile of the launch's flat tile list, 256 threads.  Wave w owns rows 16 w .. 16 w + 15 of the
// tile: A = dZ^T (row i <-> W row m0 + 16 w + i), B = X (column j <-> W column k0 + 16 blk + j), reduction over the batch rows
// (step c, slot q <-> row 4 c + q).  D: lane holds dW[m0 + 16 w + 4 q + r][k0 + 16 blk + (l & 15)] in acc[blk][r].
template <int NB, bool ADAM>
__global__ __launch_bounds__(256) void seq_dw_kernel(const vunet_seq_dw_layer* __restrict__ tab, int n_layers, int first_tile,
                                                     vunet_seq_adam_hp hp) {
  __shared__ float dzs[16 * NB * 64];
  __shared__ float xs[64 * 64];      // (the tile of dW reuses this space)
  const int tile = first_tile + blockIdx.x;
  int lo = 0, hi = n_layers;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid].tile0 <= tile) lo = mid;
    else hi = mid;
  }
  const vunet_seq_dw_layer L = tab[lo];
  const int t = tile - L.tile0, tm = t / L.tiles_k, tk = t - tm * L.tiles_k;
  const int m0 = 64 * tm, k0 = 64 * tk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  AdamResolved h = AdamResolved{};
  if constexpr (ADAM) h = adam_resolve(hp);
  f32x4 acc[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) acc[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int nchunk = L.nchunk > 1 ? L.nchunk : 1;
  for (int ch = 0; ch < nchunk; ++ch) {   // (one block of Bp rows for a flow layer; one per time step for a recurrent layer)
    const float* dzp = L.dz + (size_t)ch * L.chunk_z;
    const float* xp = L.x + (size_t)ch * L.chunk_x;
    if (ch) __syncthreads();   // the previous block's operands have been read
    for (int idx = tid; idx < 16 * NB * 16; idx += 256) {
      const int b = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<float4*>(&dzs[dw_sw(b, 4 * c4)]) = *reinterpret_cast<const float4*>(dzp + (size_t)b * L.ldz + m0 + 4 * c4);
      *reinterpret_cast<float4*>(&xs[dw_sw(b, 4 * c4)]) = *reinterpret_cast<const float4*>(xp + (size_t)b * L.ldx + k0 + 4 * c4);
    }
    __syncthreads();
    float av[4 * NB];
#pragma unroll
    for (int c = 0; c < 4 * NB; ++c) av[c] = dzs[dw_sw(4 * c + q, 16 * wave + i)];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int c = 0; c < 4 * NB; ++c)
        acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], xs[dw_sw(4 * c + q, 16 * blk + i)], acc[blk], 0, 0, 0);
    // bias: d b[m] = sum_b dZ[b][m], rows in order (the k = 0 tiles)
    if (tk == 0 && tid < 64 && L.bias)
      for (int b = 0; b < 16 * NB; ++b) bsum += dzs[dw_sw(b, tid)];
  }
  __syncthreads();   // every wave has read its operands: the space of xs becomes the tile of dW, by rows
  float* dws = xs;
#pragma unroll
  for (int blk = 0; blk < 4; ++blk)
#pragma unroll
    for (int r = 0; r < 4; ++r) dws[dw_swo(16 * wave + 4 * q + r, 16 * blk + i)] = acc[blk][r];
  __syncthreads();
  // the tile leaves in rounds of GRP float4 per thread: the round's W / exp_avg / exp_avg_sq requested together, then updated and
  // stored (-DSEQ_DW_GRP=1: one float4 at a time, =4: the whole tile at once -- 140 registers, three workgroups per CU)
#ifndef SEQ_DW_GRP
#define SEQ_DW_GRP 2
#endif
  constexpr int GRP = ADAM ? SEQ_DW_GRP : 1;
#pragma unroll
  for (int it0 = 0; it0 < 4; it0 += GRP) {
    float4 g[GRP], p[GRP], m[GRP], v[GRP];
    size_t off[GRP];
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int idx = tid + 256 * (it0 + u), row = idx >> 4, c4 = idx & 15;
      off[u] = (size_t)(m0 + row) * L.K + k0 + 4 * c4;
      if constexpr (ADAM) {
        // (each of these bytes is touched once per step: keep them out of the L2 that holds the operand rows)
        p[u] = SEQ_NT_LOAD(L.w + off[u]);
        m[u] = SEQ_NT_LOAD(L.m + off[u]);
        v[u] = SEQ_NT_LOAD(L.v + off[u]);
      }
      g[u] = *reinterpret_cast<const float4*>(&dws[dw_swo(row, 4 * c4)]);
      if (k0 + 4 * c4 + 3 >= L.kv) {   // (padding columns of the image: only where K is not the layer's own width)
        const int kc = k0 + 4 * c4;
        if (kc >= L.kv) g[u].x = 0.f;
        if (kc + 1 >= L.kv) g[u].y = 0.f;
        if (kc + 2 >= L.kv) g[u].z = 0.f;
        g[u].w = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      if constexpr (ADAM) {
        adam_update(h, g[u].x, p[u].x, m[u].x, v[u].x);
        adam_update(h, g[u].y, p[u].y, m[u].y, v[u].y);
        adam_update(h, g[u].z, p[u].z, m[u].z, v[u].z);
        adam_update(h, g[u].w, p[u].w, m[u].w, v[u].w);
        SEQ_NT_STORE(L.w + off[u], p[u]);
        SEQ_NT_STORE(L.m + off[u], m[u]);
        SEQ_NT_STORE(L.v + off[u], v[u]);
      } else {
        *reinterpret_cast<float4*>(L.g + off[u]) = g[u];
      }
    }
  }
  if (tk == 0 && tid < 64 && L.bias) {
    const int m = m0 + tid;
    if constexpr (ADAM) {
      float p = L.bias[m], mm = L.bm[m], vv = L.bv[m];
      adam_update(h, bsum, p, mm, vv);
      L.bias[m] = p;
      L.bm[m] = mm;
      L.bv[m] = vv;
    } else {
      L.bg[m] = bsum;
    }
  }
}


// ------------------------------------------------------------------------------------------------ dW (+ Adam) AND dX in one pass over W
// The layer's weights are read ONCE in the backward pass: the workgroup that updates a tile of W also forms that tile's share of
// dX = dZ . W -- from the values it loaded, before the update -- so the input-gradient chain needs no pass of its own over W
// (seq_dx_kernel: +2.5 GB per step for the flow).  A workgroup owns a (up to) 256-row x 64-column tile as four 64 x 64 sub-tiles:
// per sub-tile dW and the update exactly as seq_dw_kernel does them, then the OLD 64 x 64 block of W goes to LDS (where the tile of
// dW just was) and the four waves add dX^T[k][b] += sum_m W[m][k] dZ[b][m] for their 16 columns each (A = W^T: row i <-> column
// k0 + 16 w + i; B = dZ^T: column j <-> batch row 16 bt + j; reduction slot q <-> row 4 c + q of the sub-tile).  The partial sums
// of the M / 256 tiles of a column stripe leave as RAW slabs [M / 256][Bp][K]; whoever reads them adds them in slab order
// (seq_dz_finish_kernel, seq_coupling_bwd_kernel), as with seq_dx_kernel's.
// D2: lane holds dX[16 bt + (l & 15)][k0 + 16 w + 4 q + r] in acc2[bt][r]: a float4 along k.
template <int NB, bool ADAM>
__global__ __launch_bounds__(256) void seq_dwx_kernel(const vunet_seq_dw_layer* __restrict__ tab, int n_layers, int first_tile,
                                                      vunet_seq_adam_hp hp) {
  __shared__ float dzs[16 * NB * 64];
  __shared__ float xs[16 * NB * 64];
  __shared__ float ws[64 * 64];       // the sub-tile of dW on its way to rows, then the old sub-tile of W
  const int tile = first_tile + blockIdx.x;
  int lo = 0, hi = n_layers;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid].tile0 <= tile) lo = mid;
    else hi = mid;
  }
  const vunet_seq_dw_layer L = tab[lo];
  const int t = tile - L.tile0, tm = t / L.tiles_k, tk = t - tm * L.tiles_k;
  const int tsub = L.nchunk > 0 ? L.nchunk : 4;                  // 64-row sub-tiles per tile (vunet_seq_dwx: nchunk = tile rows / 64)
  const int row0 = 64 * tsub * tm, nsub = min(tsub, (L.M - row0) >> 6);   // (the last row tile of a layer may be short)
  const int k0 = 64 * tk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  AdamResolved h = AdamResolved{};
  if constexpr (ADAM) h = adam_resolve(hp);
  for (int idx = tid; idx < 16 * NB * 16; idx += 256) {
    const int b = idx >> 4, c4 = idx & 15;
    *reinterpret_cast<float4*>(&xs[dw_sw(b, 4 * c4)]) = *reinterpret_cast<const float4*>(L.x + (size_t)b * L.ldx + k0 + 4 * c4);
  }
  f32x4 acc2[NB];
#pragma unroll
  for (int bt = 0; bt < NB; ++bt) acc2[bt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sub = 0; sub < nsub; ++sub) {
    const int m0 = row0 + 64 * sub;
    if (sub) __syncthreads();   // the previous sub-tile's dZ / W blocks have been read
    for (int idx = tid; idx < 16 * NB * 16; idx += 256) {
      const int b = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<float4*>(&dzs[dw_sw(b, 4 * c4)]) = *reinterpret_cast<const float4*>(L.dz + (size_t)b * L.ldz + m0 + 4 * c4);
    }
    __syncthreads();
    f32x4 acc[4];
    {
      float av[4 * NB];
#pragma unroll
      for (int c = 0; c < 4 * NB; ++c) av[c] = dzs[dw_sw(4 * c + q, 16 * wave + i)];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        acc[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4 * NB; ++c)
          acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], xs[dw_sw(4 * c + q, 16 * blk + i)], acc[blk], 0, 0, 0);
      }
    }
    float bsum = 0.f;
    if (tk == 0 && tid < 64 && L.bias)
      for (int b = 0; b < 16 * NB; ++b) bsum += dzs[dw_sw(b, tid)];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) ws[dw_swo(16 * wave + 4 * q + r, 16 * blk + i)] = acc[blk][r];
    __syncthreads();
    float4 pold[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 256 * it, row = idx >> 4, c4 = idx & 15;
      float4 g = *reinterpret_cast<const float4*>(&ws[dw_swo(row, 4 * c4)]);
      if (k0 + 4 * c4 + 3 >= L.kv) {   // (padding columns of the image: only where K is not the layer's own width)
        const int kc = k0 + 4 * c4;
        if (kc >= L.kv) g.x = 0.f;
        if (kc + 1 >= L.kv) g.y = 0.f;
        if (kc + 2 >= L.kv) g.z = 0.f;
        g.w = 0.f;
      }
      const size_t off = (size_t)(m0 + row) * L.K + k0 + 4 * c4;
      if constexpr (ADAM) {
        float4 p = SEQ_NT_LOAD(L.w + off), m = SEQ_NT_LOAD(L.m + off), v = SEQ_NT_LOAD(L.v + off);
        pold[it] = p;
        adam_update(h, g.x, p.x, m.x, v.x);
        adam_update(h, g.y, p.y, m.y, v.y);
        adam_update(h, g.z, p.z, m.z, v.z);
        adam_update(h, g.w, p.w, m.w, v.w);
        SEQ_NT_STORE(L.w + off, p);
        SEQ_NT_STORE(L.m + off, m);
        SEQ_NT_STORE(L.v + off, v);
      } else {
        pold[it] = L.dx_raw ? *reinterpret_cast<const float4*>(L.w + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(L.g + off) = g;
      }
    }
    if (tk == 0 && tid < 64 && L.bias) {
      const int m = m0 + tid;
      if constexpr (ADAM) {
        float p = L.bias[m], mm = L.bm[m], vv = L.bv[m];
        adam_update(h, bsum, p, mm, vv);
        L.bias[m] = p;
        L.bm[m] = mm;
        L.bv[m] = vv;
      } else {
        L.bg[m] = bsum;
      }
    }
    if (!L.dx_raw) continue;   // (uniform per launch entry)
    __syncthreads();           // every thread has taken its float4 of dW out of ws
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 256 * it, row = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<float4*>(&ws[dw_sw(row, 4 * c4)]) = pold[it];
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < 16; ++c) {
      const float wa = ws[dw_sw(4 * c + q, 16 * wave + i)];
#pragma unroll
      for (int bt = 0; bt < NB; ++bt)
        acc2[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, dzs[dw_sw(16 * bt + i, 4 * c + q)], acc2[bt], 0, 0, 0);
    }
  }
  if (L.dx_raw) {
    float* out = L.dx_raw + (size_t)tm * (16 * NB) * L.K + k0 + 16 * wave + 4 * q;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
      *reinterpret_cast<float4*>(out + (size_t)(16 * bt + i) * L.K) = make_float4(acc2[bt][0], acc2[bt][1], acc2[bt][2], acc2[bt][3]);
  }
}

// ------------------------------------------------------------------------------------------------ ActNorm loc / scale
// grid (ceil(C / 64), n_layers), 256 threads: channel bx 64 + (t & 63), row group t >> 6 takes rows rg, rg + 4, ...
template <bool ADAM>
__global__ __launch_bounds__(256) void seq_actnorm_bwd_kernel(const vunet_seq_actnorm_layer* __restrict__ tab, int C, int B,
                                                              const float* __restrict__ dld, vunet_seq_adam_hp hp) {
  __shared__ float red[2][4][64];
  const vunet_seq_actnorm_layer L = tab[blockIdx.y];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C)
    for (int b = rg; b < B; b += 4) {
      const float g = L.gfull[(size_t)b * L.ld + c];
      s0 += g;
      s1 += g * L.out[(size_t)b * L.ld + c];
    }
  red[0][rg][cl] = s0;
  red[1][rg][cl] = s1;
  __syncthreads();
  if (rg != 0 || c >= C) return;
  s0 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  s1 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  float sld = 0.f;
  for (int b = 0; b < B; ++b) sld += dld[b];
  const float sc = L.scale[c];
  // out = sc (u + loc): d sc = sum g (u + loc) = sum g out / sc; logdet = sum_c log|sc| on every row: d sc += sum_b dld / sc
  const float gsc = (s1 + sld) / sc, glo = sc * s0;
  if constexpr (ADAM) {
    const AdamResolved h = adam_resolve(hp);
    float p = sc, m = L.sm[c], v = L.sv[c];
    adam_update(h, gsc, p, m, v);
    L.scale[c] = p;
    L.sm[c] = m;
    L.sv[c] = v;
    p = L.loc[c];
    m = L.lm[c];
    v = L.lv[c];
    adam_update(h, glo, p, m, v);
    L.loc[c] = p;
    L.lm[c] = m;
    L.lv[c] = v;
  } else {
    L.gs[c] = gsc;
    L.gl[c] = glo;
  }
}

// ------------------------------------------------------------------------------------------------ FlowLoss
// one workgroup of 1024 threads: wave w takes rows w, w + 16, ... -- all of a wave's loads are independent (16-byte where the
// geometry allows: VEC), the row sums meet in LDS and thread 0 adds them in row order
template <bool VEC>
__global__ __launch_bounds__(1024) void seq_flow_loss_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ logdet,
                                                             const float* __restrict__ noise, int B, int Bp, int C,
                                                             float* __restrict__ scalars, float* __restrict__ dz, int ld_dz,
                                                             float* __restrict__ dld) {
  __shared__ float rs[2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_b = 1.f / (float)B;
  for (int b = wave; b < Bp; b += 16) {
    float s = 0.f, sn = 0.f;
    if (b < B) {
      if constexpr (VEC) {
        const float4* zr = reinterpret_cast<const float4*>(z + (size_t)b * ldz);
        const float4* nr = noise ? reinterpret_cast<const float4*>(noise + (size_t)b * C) : nullptr;
        float4* dr = dz ? reinterpret_cast<float4*>(dz + (size_t)b * ld_dz) : nullptr;
#pragma unroll 4
        for (int c = lane; c < C / 4; c += 64) {
          const float4 v = zr[c];
          s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          if (dr) dr[c] = make_float4(v.x * inv_b, v.y * inv_b, v.z * inv_b, v.w * inv_b);
          if (nr) {
            const float4 e = nr[c];
            sn += (e.x * e.x + e.y * e.y) + (e.z * e.z + e.w * e.w);
          }
        }
      } else {
        for (int c = lane; c < C; c += 64) {
          const float v = z[(size_t)b * ldz + c];
          s += v * v;
          if (dz) dz[(size_t)b * ld_dz + c] = v * inv_b;
          if (noise) {
            const float e = noise[(size_t)b * C + c];
            sn += e * e;
          }
        }
      }
    } else if (dz) {
      for (int c = lane; c < C; c += 64) dz[(size_t)b * ld_dz + c] = 0.f;
    }
    s = wave_sum(s);
    sn = wave_sum(sn);
    if (lane == 0) {
      rs[0][b] = 0.5f * s;      // nll (lib/losses.py:330-331)
      rs[1][b] = 0.5f * sn;
      if (dld) dld[b] = b < B ? -inv_b : 0.f;
    }
  }
  __syncthreads();
  if (wave == 0) {   // row sums in row order: lane b holds row b; the 64 values are added by a fixed tree
    float nll = lane < B ? rs[0][lane] : 0.f, ref = lane < B ? rs[1][lane] : 0.f, ld = lane < B ? logdet[lane] : 0.f;
    nll = wave_sum(nll);
    ref = wave_sum(ref);
    ld = wave_sum(ld);
    if (lane == 0) {
      nll *= inv_b;
      const float nld = -(ld * inv_b);
      scalars[0] = nll + nld;
      scalars[1] = ref * inv_b;
      scalars[2] = nld;
      scalars[3] = nll;
    }
  }
}

// ++step, and the step's bias-corrected constants for every kernel of the step to read (formed as adam_resolve forms them)
__global__ void seq_adam_tick_kernel(int64_t* step, const double* lr, float b1, float b2, float* resolved) {
  const int64_t t = *step + 1;
  *step = t;
  if (resolved) {
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    resolved[0] = (float)(*lr / bc1);
    resolved[1] = (float)(1.0 / sqrt(bc2));
  }
}

__global__ __launch_bounds__(256) void seq_unpack_rows_kernel(const float* __restrict__ src, int ld_src, int col_off, int row_off,
                                                              int row_mul, float* __restrict__ dst, int M, int K, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * K) return;
  const int m = (int)(idx / K), k = (int)(idx - (size_t)m * K);
  const float v = src[(size_t)(row_off + row_mul * m) * ld_src + col_off + k];
  dst[idx] = accumulate ? dst[idx] + v : v;
}

}  // namespace

static int seq_dx_launch(const vunet_seq_dx_desc* d, const float* w0, const float* w1, const float* dz, float* raw, const float* y,
                         float* dz_prev, int32_t* counters, float slope, bool fin, void* stream) {
  if (!d || !w0 || !dz || !raw) return VUNET_ERR_ARG;
  if (d->nets < 1 || d->nets > 2 || (d->nets == 2 && !w1)) return VUNET_ERR_ARG;
  if (d->B < 1 || d->B > 64 || d->K < 64 || d->K % 64 || d->S < 1 || d->M < 16 * d->S || d->M % 16) return VUNET_ERR_ARG;
  if (d->ldw && (d->ldw < d->K || d->ldw % 4)) return VUNET_ERR_ARG;
  if (fin && (!y || !dz_prev || !counters)) return VUNET_ERR_ARG;
  SeqDxArgs a;
  a.ldw = d->ldw ? d->ldw : d->K;
  a.w[0] = w0;
  a.w[1] = w1;
  a.dz = dz;
  a.raw = raw;
  a.M = d->M;
  a.K = d->K;
  a.Bp = (d->B + 15) / 16 * 16;
  a.S = d->S;
  a.y = y;
  a.dzp = dz_prev;
  a.cnt = counters;
  a.slope = slope;
  const dim3 grid(d->K / 64, d->S, d->nets);
  hipStream_t st = (hipStream_t)stream;
  const bool wide = d->M / d->S >= 16 * 16 && d->B <= 32;   // 16 waves: at least one 16-row group each (64 rows: see vunet_seq_linear)
  // a launch that cannot give every CU a workgroup (<= 160 of them, deep slabs) runs eight waves per workgroup
  const bool eight = !wide && (long)grid.x * grid.y * grid.z <= 160 && d->M / d->S >= 16 * 16;
#define SEQ_DX_LAUNCH(NB, FIN)                                                            \
  if (wide) VUNET_LAUNCH((seq_dx_kernel<NB, 16, FIN>), grid, dim3(1024), 0, st, a);       \
  else if (eight) VUNET_LAUNCH((seq_dx_kernel<NB, 8, FIN>), grid, dim3(512), 0, st, a);   \
  else VUNET_LAUNCH((seq_dx_kernel<NB, 4, FIN>), grid, dim3(256), 0, st, a);
#define SEQ_DX_CASE(NB)                  \
  case NB:                               \
    if (fin) { SEQ_DX_LAUNCH(NB, true) } \
    else { SEQ_DX_LAUNCH(NB, false) }    \
    break;
  switch (a.Bp / 16) {
    SEQ_DX_CASE(1)
    SEQ_DX_CASE(2)
    SEQ_DX_CASE(3)
    SEQ_DX_CASE(4)
    default: return VUNET_ERR_ARG;
  }
#undef SEQ_DX_CASE
#undef SEQ_DX_LAUNCH
  return vunet_check_launch();
}

extern "C" int vunet_seq_dx(const vunet_seq_dx_desc* d, const float* w0, const float* w1, const float* dz, float* raw, void* stream) {
  return seq_dx_launch(d, w0, w1, dz, raw, nullptr, nullptr, nullptr, 0.f, false, stream);
}

extern "C" int vunet_seq_dx_finish(const vunet_seq_dx_desc* d, const float* w0, const float* w1, const float* dz, float* raw,
                                   const float* y, float* dz_prev, int32_t* counters, float slope, void* stream) {
  return seq_dx_launch(d, w0, w1, dz, raw, y, dz_prev, counters, slope, true, stream);
}

extern "C" int vunet_seq_dz_finish(const float* raw, const float* y, float* dz, int32_t nets, int32_t S, int32_t Bp, int32_t K,
                                   float slope, void* stream) {
  if (!raw || !y || !dz || nets < 1 || nets > 2 || S < 1 || Bp < 16 || Bp % 16 || K < 4 || K % 4) return VUNET_ERR_ARG;
  const size_t per_net4 = (size_t)Bp * K / 4;
  VUNET_LAUNCH(seq_dz_finish_kernel, dim3((unsigned)((per_net4 + 255) / 256), nets), dim3(256), 0, (hipStream_t)stream, raw, y, dz, S,
               per_net4, slope);
  return vunet_check_launch();
}

extern "C" int vunet_seq_coupling_bwd(const vunet_seq_coupling_bwd_desc* d, const float* gbase, const float* gslabs,
                                      const int32_t* inv_map, const float* scale, const float* in, const float* st, const float* bias_s,
                                      const float* bias_t, const float* dld, float* gfull, float* gout, float* dzh, void* stream) {
  if (!d || !gbase || !gout || d->B < 1 || d->B > 64 || d->C < 1 || d->ld_g < d->C || d->ld_out < d->C) return VUNET_ERR_ARG;
  if (gslabs && (d->n_sl < 1 || d->ld_sl < d->c1s || d->c1s < 0 || d->c1s > d->C)) return VUNET_ERR_ARG;
  if (gfull && d->ld_full < d->C) return VUNET_ERR_ARG;
  if (st && (!in || !dzh || !dld || d->ld_in < d->C || d->c1 < 0 || d->c1 > d->C || d->Mp < d->C - d->c1 || d->S < 1 || d->S > 8))
    return VUNET_ERR_ARG;
  if (st && d->S > 1 && (!bias_s || !bias_t)) return VUNET_ERR_ARG;
  if (gout == gbase && inv_map) return VUNET_ERR_ARG;   // a gather cannot run in place
  SeqCouplingBwdArgs a;
  a.gbase = gbase;
  a.gslabs = gslabs;
  a.inv_map = inv_map;
  a.scale = scale;
  a.in = in;
  a.st = st;
  a.bias_s = bias_s;
  a.bias_t = bias_t;
  a.dld = dld;
  a.gfull = gfull;
  a.gout = gout;
  a.dzh = dzh;
  a.B = d->B;
  a.Bp = (d->B + 15) / 16 * 16;
  a.C = d->C;
  a.c1 = d->c1;
  a.ld_g = d->ld_g;
  a.ld_in = d->ld_in;
  a.ld_out = d->ld_out;
  a.ld_full = d->ld_full;
  a.Mp = d->Mp;
  a.S = st ? d->S : 1;
  a.n_sl = d->n_sl;
  a.ld_sl = d->ld_sl;
  a.c1s = d->c1s;
  VUNET_LAUNCH(seq_coupling_bwd_kernel, dim3(d->B, (d->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return vunet_check_launch();
}

static bool adam_hp_ok(const vunet_seq_adam_hp* hp) {
  return hp->lr_dev && hp->step_dev && hp->beta1 >= 0.f && hp->beta1 < 1.f && hp->beta2 >= 0.f && hp->beta2 < 1.f && hp->eps >= 0.f;
}

extern "C" int vunet_seq_dw(const vunet_seq_dw_layer* table_dev, int32_t n_layers, int32_t first_tile, int32_t n_tiles, int32_t B,
                            const vunet_seq_adam_hp* hp, void* stream) {
  if (!table_dev || n_layers < 1 || first_tile < 0 || n_tiles < 1 || B < 1 || B > 64) return VUNET_ERR_ARG;
  if (hp && !adam_hp_ok(hp)) return VUNET_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const vunet_seq_adam_hp h = hp ? *hp : vunet_seq_adam_hp{};
#define SEQ_DW_CASE(NB)                                                                                                  \
  case NB:                                                                                                               \
    if (hp) VUNET_LAUNCH((seq_dw_kernel<NB, true>), dim3(n_tiles), dim3(256), 0, st, table_dev, n_layers, first_tile, h); \
    else VUNET_LAUNCH((seq_dw_kernel<NB, false>), dim3(n_tiles), dim3(256), 0, st, table_dev, n_layers, first_tile, h);   \
    break;
  switch ((B + 15) / 16) {
    SEQ_DW_CASE(1)
    SEQ_DW_CASE(2)
    SEQ_DW_CASE(3)
    SEQ_DW_CASE(4)
    default: return VUNET_ERR_ARG;
  }
#undef SEQ_DW_CASE
  return vunet_check_launch();
}

extern "C" int vunet_seq_dwx(const vunet_seq_dw_layer* table_dev, int32_t n_layers, int32_t first_tile, int32_t n_tiles, int32_t B,
                             const vunet_seq_adam_hp* hp, void* stream) {
  if (!table_dev || n_layers < 1 || first_tile < 0 || n_tiles < 1 || B < 1 || B > 64) return VUNET_ERR_ARG;
  if (hp && !adam_hp_ok(hp)) return VUNET_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const vunet_seq_adam_hp h = hp ? *hp : vunet_seq_adam_hp{};
#define SEQ_DWX_CASE(NB)                                                                                                  \
  case NB:                                                                                                                \
    if (hp) VUNET_LAUNCH((seq_dwx_kernel<NB, true>), dim3(n_tiles), dim3(256), 0, st, table_dev, n_layers, first_tile, h); \
    else VUNET_LAUNCH((seq_dwx_kernel<NB, false>), dim3(n_tiles), dim3(256), 0, st, table_dev, n_layers, first_tile, h);   \
    break;
  switch ((B + 15) / 16) {
    SEQ_DWX_CASE(1)
    SEQ_DWX_CASE(2)
    SEQ_DWX_CASE(3)
    SEQ_DWX_CASE(4)
    default: return VUNET_ERR_ARG;
  }
#undef SEQ_DWX_CASE
  return vunet_check_launch();
}

extern "C" int vunet_seq_actnorm_bwd(const vunet_seq_actnorm_layer* table_dev, int32_t n_layers, int32_t C, int32_t B, const float* dld,
                                     const vunet_seq_adam_hp* hp, void* stream) {
  if (!table_dev || n_layers < 1 || C < 1 || B < 1 || B > 64 || !dld) return VUNET_ERR_ARG;
  if (hp && !adam_hp_ok(hp)) return VUNET_ERR_ARG;
  const dim3 grid((C + 63) / 64, n_layers);
  if (hp) VUNET_LAUNCH((seq_actnorm_bwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, table_dev, C, B, dld, *hp);
  else VUNET_LAUNCH((seq_actnorm_bwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, table_dev, C, B, dld, vunet_seq_adam_hp{});
  return vunet_check_launch();
}

extern "C" int vunet_seq_flow_loss(const float* z, int32_t ldz, const float* logdet, const float* noise, int32_t B, int32_t C,
                                   float* scalars, float* dz, int32_t ld_dz, float* dld, void* stream) {
  if (!z || !logdet || !scalars || B < 1 || B > 64 || C < 1 || ldz < C || (dz && ld_dz < C)) return VUNET_ERR_ARG;
  const int Bp = (B + 15) / 16 * 16;
  const bool vec = C % 4 == 0 && ldz % 4 == 0 && (!dz || ld_dz % 4 == 0) && ((uintptr_t)z & 15) == 0 && ((uintptr_t)dz & 15) == 0 &&
                   ((uintptr_t)noise & 15) == 0;
  if (vec) VUNET_LAUNCH((seq_flow_loss_kernel<true>), dim3(1), dim3(1024), 0, (hipStream_t)stream, z, ldz, logdet, noise, B, Bp, C,
                        scalars, dz, ld_dz, dld);
  else VUNET_LAUNCH((seq_flow_loss_kernel<false>), dim3(1), dim3(1024), 0, (hipStream_t)stream, z, ldz, logdet, noise, B, Bp, C,
                    scalars, dz, ld_dz, dld);
  return vunet_check_launch();
}

extern "C" int vunet_seq_adam_tick(int64_t* step_dev, const double* lr_dev, float beta1, float beta2, float* resolved_dev, void* stream) {
  if (!step_dev || (resolved_dev && !lr_dev)) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr_dev, beta1, beta2, resolved_dev);
  return vunet_check_launch();
}

extern "C" int vunet_seq_unpack_rows(const float* src, int32_t ld_src, int32_t col_off, int32_t row_off, int32_t row_mul, float* dst,
                                     int32_t M, int32_t K, int32_t accumulate, void* stream) {
  if (!src || !dst || M < 1 || K < 1 || col_off < 0 || ld_src < col_off + K || row_off < 0 || row_mul < 1) return VUNET_ERR_ARG;
  const size_t n = (size_t)M * K;
  VUNET_LAUNCH(seq_unpack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, col_off,
               row_off, row_mul, dst, M, K, accumulate);
  return vunet_check_launch();
}
