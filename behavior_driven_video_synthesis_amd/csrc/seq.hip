// Behaviour front half of BASELINE config 5: flow sample -> pose_behavior_rnn decode, on pose vectors [B, n_kps] and behaviour
// codes [B, C] (experiments/behavior_net.py:1086-1100, :1173-1184).
//
// Everything here is a chain of skinny matrix products  Y[B, M] = act(X[B, K] . W[M, K]^T + b)  with B <= 64 rows and weights of
// 0.2 - 17 MB per layer (the flow of config/behavior_net.yaml: 60 MLPs of 512-2048-2048-2048-512, 2.5 GB of fp32 weights per
// sample pass; the decoder: one [4096, 1075] gate matrix read 50 times).  Each weight is used once per pass, so the bound is
// the weight stream from HBM, not the matrix cores: the kernels are laid out for that.
//
//   * seq_linear_kernel: a workgroup owns a 16-row tile of W over the whole K; a K chunk of 32 is 128 contiguous bytes per weight
//     row, two 16-byte loads per lane, eight v_mfma_f32_16x16x4_f32 per tile (exact fp32: the reference computes in fp32)
//     against each 16-row batch tile.  Its 16 waves (4 where K is short) split K by chunks and add through LDS in wave order;
//     wave w then finishes batch tile w -- bias, activation, store -- so the next layer reads a plain operand.  K is NOT split
//     over workgroups, with one exception (below).  Two general split forms were built and measured
//     (profiles/r05_seq_time.txt): (1) partial slabs summed, with bias + LeakyReLU, by the consumer's loads: every 16-row tile of
//     the next layer re-read every slab of X from L2 -- 100 MB of L1 fills for a 33 MB layer, 17.5 us where the weights stream
//     in 5 (a CU fills its L1 at <= 70 GB/s from L2; MI355X_MICROARCH.md "Indexed rows"); (2) the last workgroup to arrive at
//     a tile's counter adds the slabs: behind __threadfence() +10 us per launch (buffer_wbl2 + buffer_inv), behind sc1 stores /
//     loads and an agent atomic wrong results on partial-line slabs.  Without a split the 2048-row layers still put 256
//     workgroups on the chip.  The exception: the flow's 512-row head layers, whose one consumer (seq_coupling_kernel) reads
//     every value exactly once -- there K is split four ways into RAW slabs and that kernel adds them in slab order, with the
//     bias and the scale net's tanh (64 CUs and 10.7 us -> 256 workgroups and 5.1 us).  The s and t nets of a coupling (same
//     input, same shapes) and the mu / logstd heads run as one launch (grid z).
//   * seq_coupling_kernel: everything between two MLP evaluations of the flow -- the affine coupling itself, the half swap,
//     ``Shuffle`` and ``ActNorm`` -- as "v = couple(in); out[c] = affine(v[map[c]])" with a host-composed index map.
//   * an LSTM step is seq_linear_kernel over the gate matrix [W_ih | 0 | W_hh] with its rows GATE-INTERLEAVED (row 4 j + q =
//     gate q of unit j): the lane that holds four consecutive output rows of a batch row holds the four gates of one unit and
//     finishes the cell in its registers -- c', h, the next operand row's h -- so the encoder is ONE launch per time step.  The
//     decoder adds seq_dec_out_kernel: the output layer + residual (``ResidualRNNDecoder.forward``), which writes the step's pose
//     and the next operand's x.  The decoder's optional input layer is folded into the gate matrix when the weights are packed.
#include "common.h"
#include "../../include/vunet_seq_train.h"
#include "../../include/vunet_seq_tiled.h"

namespace {

constexpr int SEQ_MAX_NB = 4;   // batch tiles of 16: B <= 64

struct SeqLinearArgs {
  const float* w[2];      // [M][K] row-major, M % 16 == 0, K % 32 == 0 (zero padded)
  const float* x;         // [nets_in][Bp][ldx]
  const float* bias[2];   // [M] (NULL: none)
  float* y;               // [nets][Bp][M]
  int M, K, Bp, ldx, act[2], shared_in;
  int S;                  // > 1: K split over grid.y workgroups, y = [nets][S][Bp][M] RAW partial slabs (no bias, no activation)
  // LSTM form (template flag): the rows are gate-interleaved -- row 4 j + q is gate q (i, f, g, o) of hidden unit j -- so the
  // lane that holds rows 4 kq .. 4 kq + 3 of batch row n holds all four gates of one unit and finishes the cell in registers
  const float* c_in;      // [Bp][H]
  float* c_out;           // [Bp][H] (another buffer)
  float* xh_next;         // [Bp][ldx] the NEXT step's operand rows (another buffer than x): h at hoff
  float* h_out;           // [Bp][H] or NULL
  const float* x_next;    // encoder: the next input pose, row b at + b * seq_stride (NULL: the decoder's second launch writes x)
  float* y2;              // tile-major output (LAY & 4) AND this row-major copy [nets][Bp][M] (training keeps both; NULL: none)
  float* gates_out;       // training: [Bp][H][4] = sigmoid(i), sigmoid(f), tanh(g), sigmoid(o) of every unit (NULL: not kept)
  const float* ht_in;     // LAY & 16: the h part of the operand rows as tiles [Bp / 16][H / 32][2][64][4] (the previous step's ht_out)
  float* ht_out;          // h ALSO written as the next step's ht_in (NULL: not wanted)
  long long seq_stride;
  int H, hoff, n, B;
};

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ float seq_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.01f * v);   // nn.LeakyReLU() default slope 0.01 (lib/modules.py:244)
  if (act == 2) return tanhf(v);              // the scale nets' head (lib/modules.py:252-253)
  return v;
}

// The weight stream is the cost and a wave has few chunks (2 - 8), so what matters is bytes in flight: a group of U chunks
// issues all of its loads (weights of RT tiles, the operand of NB batch tiles) before the first MFMA.  The scheduling barrier
// keeps it that way: without it the scheduler interleaves loads and uses to save registers and keeps ~7 loads in flight.
#ifdef SEQ_W_NO_NT   // (timing-ablation build, tools/ab_build.sh)
__device__ __forceinline__ float4 seq_w_load(const float* p) { return *reinterpret_cast<const float4*>(p); }
#else
typedef float seq_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 seq_w_load(const float* p) {
  const seq_f4v t = __builtin_nontemporal_load(reinterpret_cast<const seq_f4v*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
#endif

// LAY (include/vunet_seq_tiled.h): bit 0 -- the weights are tile-major images, bit 1 -- the operand is: a chunk is then 2 KB
// contiguous, [half][lane] float4, and each of a wave's loads covers 1 KB contiguous.
template <int NB, int RT, int U>
struct SeqGroupRegs {
  float4 wv[RT][U][2], xv[NB][U][2];
};

// the loads of a group of U chunks (c0, c0 + WAVES, ...): weights of RT tiles, the operand of NB batch tiles
template <int NB, int RT, int U, int WAVES, int LAY>
__device__ __forceinline__ void seq_group_load(const SeqLinearArgs& a, const float* __restrict__ w, const float* __restrict__ x, int c0,
                                               SeqGroupRegs<NB, RT, U>& r) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr ((LAY & 9) == 9) {   // (tile-major, re-read every time step from L2 / the Infinity Cache: default policy)
        const float* wp = w + (size_t)rt * 16 * a.K + 512 * (c0 + WAVES * u);
        r.wv[rt][u][0] = *reinterpret_cast<const float4*>(wp);
        r.wv[rt][u][1] = *reinterpret_cast<const float4*>(wp + 256);
      } else if constexpr (LAY & 1) {   // (read once per pass by this one workgroup: streamed past the caches' retention, SEQ_W_NT)
        const float* wp = w + (size_t)rt * 16 * a.K + 512 * (c0 + WAVES * u);
        r.wv[rt][u][0] = seq_w_load(wp);
        r.wv[rt][u][1] = seq_w_load(wp + 256);
      } else {
        // (row-major weights: the parameters themselves when a flow trains -- the input-gradient chain and the update sweep read
        //  them again within the same block's 168 MB, from the Infinity Cache: default policy.  Non-temporal here: 7.21 -> 7.40 ms
        //  per training step.)
        const float* wp = w + (size_t)rt * 16 * a.K + 32 * (c0 + WAVES * u);
        r.wv[rt][u][0] = *reinterpret_cast<const float4*>(wp);
        r.wv[rt][u][1] = *reinterpret_cast<const float4*>(wp + 4);
      }
    }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (LAY & 2) {
        const float* xp = x + (size_t)nb * 16 * a.K + 512 * (c0 + WAVES * u);
        r.xv[nb][u][0] = *reinterpret_cast<const float4*>(xp);
        r.xv[nb][u][1] = *reinterpret_cast<const float4*>(xp + 256);
      } else if constexpr (LAY & 16) {
        // the LSTM step's operand rows [x | 0 | h]: the chunks of x (and the padding) from the row-major rows, the chunks of h --
        // 32 of 34 at the reference size -- from the tile-major copy the previous step's epilogue left (1 KB contiguous per wave
        // load; the row-major form touches two half-used 128-byte lines per row and chunk)
        const int c = c0 + WAVES * u, ch0 = a.hoff >> 5;
        if (c >= ch0) {
          const float* xp = a.ht_in + ((size_t)(nb * (a.H >> 5) + (c - ch0)) * 2) * 256 + 4 * (threadIdx.x & 63);
          r.xv[nb][u][0] = *reinterpret_cast<const float4*>(xp);
          r.xv[nb][u][1] = *reinterpret_cast<const float4*>(xp + 256);
        } else {
          const float* xp = x + (size_t)nb * 16 * a.ldx + 32 * c;
          r.xv[nb][u][0] = *reinterpret_cast<const float4*>(xp);
          r.xv[nb][u][1] = *reinterpret_cast<const float4*>(xp + 4);
        }
      } else {
        const float* xp = x + (size_t)nb * 16 * a.ldx + 32 * (c0 + WAVES * u);
        r.xv[nb][u][0] = *reinterpret_cast<const float4*>(xp);
        r.xv[nb][u][1] = *reinterpret_cast<const float4*>(xp + 4);
      }
    }
}

// ... and its matrix steps: chunk by chunk, weight tile by weight tile, batch tile by batch tile, k ascending
template <int NB, int RT, int U>
__device__ __forceinline__ void seq_group_mfma(const SeqGroupRegs<NB, RT, U>& r, f32x4 (&acc)[RT][NB]) {
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float wa[8] = {r.wv[rt][u][0].x, r.wv[rt][u][0].y, r.wv[rt][u][0].z, r.wv[rt][u][0].w,
                           r.wv[rt][u][1].x, r.wv[rt][u][1].y, r.wv[rt][u][1].z, r.wv[rt][u][1].w};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float xb[8] = {r.xv[nb][u][0].x, r.xv[nb][u][0].y, r.xv[nb][u][0].z, r.xv[nb][u][0].w,
                             r.xv[nb][u][1].x, r.xv[nb][u][1].y, r.xv[nb][u][1].z, r.xv[nb][u][1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[rt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb[j], acc[rt][nb], 0, 0, 0);
      }
    }
}

template <int NB, int RT, int U, int WAVES, int LAY>
__device__ __forceinline__ void seq_linear_group(const SeqLinearArgs& a, const float* __restrict__ w, const float* __restrict__ x, int c0,
                                                 f32x4 (&acc)[RT][NB]) {
  SeqGroupRegs<NB, RT, U> r;
  seq_group_load<NB, RT, U, WAVES, LAY>(a, w, x, c0, r);
  __builtin_amdgcn_sched_barrier(0);
  seq_group_mfma<NB, RT, U>(r, acc);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// grid (M / 16, S, nets), 64 WAVES threads.  Lane l of a wave: i = l & 15 (weight row / batch row), kq = l >> 4; in a K chunk of 32
// it holds k = 8 kq .. 8 kq + 7 of its row -- the same k for the A (weights) and B (batch) operand of MFMA step j = 0..7.
// D: lane holds batch row n = l & 15 of its tile, output rows 4 (l >> 4) + r of the weight tile.
// RT (16-row weight tiles per wave): only 1 is instantiated -- 32-row tiles halve the operand traffic per weight byte but leave
// half the CUs without a workgroup at every layer size of the reference configuration.
// WAVES = 16 where K has >= 16 chunks, else 4.  Four waves with 16 chunks each and sixteen with 4 measured the same on the
// 2048 x 2048 layers (12 us: the bound is what a CU pulls through its L1, not the number of round trips); sixteen are kept
// because the batch tiles' epilogues then run on separate waves and the short layers request all their weights at once.
template <int NB, int RT, int WAVES, bool LSTM = false, int LAY = 0>
__global__ __launch_bounds__(64 * WAVES) void seq_linear_kernel(SeqLinearArgs a) {
  __shared__ float4 red[WAVES][RT][NB][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 16 * RT, net = blockIdx.z;
  const int kb = a.K / a.S, nchunk = kb >> 5;   // (S = 1 everywhere but the flow's 512-row head layers)
  const float* __restrict__ w = (LAY & 1) ? a.w[net] + (size_t)m0 * a.K + (size_t)blockIdx.y * kb * 16 + 4 * lane
                                          : a.w[net] + (size_t)(m0 + i) * a.K + (size_t)blockIdx.y * kb + 8 * kq;
  const float* __restrict__ x = (LAY & 2) ? a.x + (a.shared_in ? 0 : (size_t)net * a.Bp * a.K) + (size_t)blockIdx.y * kb * 16 + 4 * lane
                                          : a.x + (a.shared_in ? 0 : (size_t)net * a.Bp * a.ldx) + (size_t)i * a.ldx +
                                                (size_t)blockIdx.y * kb + 8 * kq;
  f32x4 acc[RT][NB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this wave's chunks: wave, wave + WAVES, ... (ascending k; the order of the sums is fixed by the geometry alone)
  int c = wave;
  // (16 waves share the CU's registers: 128 each)
  // (four-wave form at 3 - 4 batch tiles with groups of 4 instead of 2: 6.96 -> 7.10 ms per flow training step)
  constexpr int UMAX = WAVES == 16 ? (NB == 1 ? 4 : 2) : ((NB + RT <= 3) ? 4 : 2);
  // Four-wave workgroups at 3 - 4 batch tiles are one wave per SIMD (256 workgroups on 256 CUs): nothing else runs while a wave
  // waits for its loads, so the wave itself keeps a RING of SEQ_RING stages (of SEQ_RING_U chunks) in flight -- the loads of stage
  // i + SEQ_RING - 1 are issued before the matrix steps of stage i (the compiler's counted vmcnt waits retire them in order).
  // Same chunks, same order, same sums as the plain loop.  (flow training step at 64 rows: see DESIGN.md section 5.R6.)
#ifndef SEQ_RING
#define SEQ_RING 2
#endif
#ifndef SEQ_RING_U
#define SEQ_RING_U 1
#endif
  if constexpr (WAVES == 4 && NB >= 3 && SEQ_RING == 2 && SEQ_RING_U == 1) {
    // depth 2, any number of chunks (an odd count -- 34 chunks over four waves: 9, 9, 8, 8 -- keeps its last chunk in the ring)
    const int nw = c < nchunk ? (nchunk - c + WAVES - 1) / WAVES : 0;   // this wave's chunks
    if (nw > 0) {
      SeqGroupRegs<NB, RT, 1> ra, rb;
      seq_group_load<NB, RT, 1, WAVES, LAY>(a, w, x, c, ra);
      int k = 0;
      for (; k + 2 < nw; k += 2) {   // ra holds chunk k
        seq_group_load<NB, RT, 1, WAVES, LAY>(a, w, x, c + WAVES * (k + 1), rb);
        __builtin_amdgcn_sched_barrier(0);
        seq_group_mfma<NB, RT, 1>(ra, acc);
        __builtin_amdgcn_sched_barrier(0);
        seq_group_load<NB, RT, 1, WAVES, LAY>(a, w, x, c + WAVES * (k + 2), ra);
        __builtin_amdgcn_sched_barrier(0);
        seq_group_mfma<NB, RT, 1>(rb, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nw - k == 2) {
        seq_group_load<NB, RT, 1, WAVES, LAY>(a, w, x, c + WAVES * (k + 1), rb);
        __builtin_amdgcn_sched_barrier(0);
        seq_group_mfma<NB, RT, 1>(ra, acc);
        __builtin_amdgcn_sched_barrier(0);
        seq_group_mfma<NB, RT, 1>(rb, acc);
      } else {
        seq_group_mfma<NB, RT, 1>(ra, acc);
      }
      c += WAVES * nw;
    }
  } else if constexpr (WAVES == 4 && NB >= 3 && SEQ_RING > 1) {
    constexpr int D = SEQ_RING, SU = SEQ_RING_U;   // D stages of SU chunks each
    const int nw = c < nchunk ? (nchunk - c + WAVES - 1) / WAVES : 0;   // this wave's chunks
    const int ns = nw / SU, n_main = ns - ns % D;                        // ... stages, and those the ring takes
    if (n_main > 0) {
      SeqGroupRegs<NB, RT, SU> ring[D];
#pragma unroll
      for (int dd = 0; dd < D - 1; ++dd) seq_group_load<NB, RT, SU, WAVES, LAY>(a, w, x, c + WAVES * SU * dd, ring[dd]);
      int base = 0;
      for (; base + D < n_main; base += D) {   // steady state: every stage requests the one D - 1 ahead of the one it multiplies
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          seq_group_load<NB, RT, SU, WAVES, LAY>(a, w, x, c + WAVES * SU * (base + dd + D - 1), ring[(dd + D - 1) % D]);
          __builtin_amdgcn_sched_barrier(0);
          seq_group_mfma<NB, RT, SU>(ring[dd], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the last D stages: only the very last one is still to be requested
      seq_group_load<NB, RT, SU, WAVES, LAY>(a, w, x, c + WAVES * SU * (n_main - 1), ring[D - 1]);
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        __builtin_amdgcn_sched_barrier(0);
        seq_group_mfma<NB, RT, SU>(ring[dd], acc);
      }
      c += WAVES * SU * n_main;
    }
  }
  for (; c + WAVES * (UMAX - 1) < nchunk; c += WAVES * UMAX) seq_linear_group<NB, RT, UMAX, WAVES, LAY>(a, w, x, c, acc);
  if constexpr (UMAX == 4)
    if (c + WAVES < nchunk) {
      seq_linear_group<NB, RT, 2, WAVES, LAY>(a, w, x, c, acc);
      c += 2 * WAVES;
    }
  if constexpr (UMAX > 1)
    for (; c < nchunk; c += WAVES) seq_linear_group<NB, RT, 1, WAVES, LAY>(a, w, x, c, acc);
  // every wave's partial -> LDS; wave w < NB adds them in wave order for batch tile w and finishes that tile (the epilogues run
  // side by side: with tanh / the LSTM cell in them, one wave finishing four tiles was a 4 us tail)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      red[wave][rt][nb][lane] = make_float4(acc[rt][nb][0], acc[rt][nb][1], acc[rt][nb][2], acc[rt][nb][3]);
  __syncthreads();
  if (wave >= NB) return;
  const int nb = wave;
  float4 v[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    v[rt] = red[0][rt][nb][lane];
#pragma unroll 3
    for (int q = 1; q < WAVES; ++q) v[rt] = add4(v[rt], red[q][rt][nb][lane]);
  }
  const size_t col = (size_t)m0 + 4 * kq;   // + 16 rt; row n = 16 nb + i
  if constexpr (LSTM) {
    // gate order i, f, g, o (torch.nn.LSTMCell, models/pose_behavior_rnn.py:476, :498): c' = sig(f) c + sig(i) tanh(g),
    // h = sig(o) tanh(c')
    const int j = (int)(col >> 2);
    const float4 bv = *reinterpret_cast<const float4*>(a.bias[0] + col);
    {
      const int nrow = nb * 16 + i;
      const float4 g = add4(v[0], bv);
      const float si = sigmoid_f(g.x), sf = sigmoid_f(g.y), tg = tanhf(g.z), so = sigmoid_f(g.w);
      const float c2 = sf * a.c_in[(size_t)nrow * a.H + j] + si * tg;
      const float h = so * tanhf(c2);
      if (a.gates_out) *reinterpret_cast<float4*>(a.gates_out + ((size_t)nrow * a.H + j) * 4) = make_float4(si, sf, tg, so);
      a.c_out[(size_t)nrow * a.H + j] = c2;
      a.xh_next[(size_t)nrow * a.ldx + a.hoff + j] = h;
      if (a.h_out) a.h_out[(size_t)nrow * a.H + j] = h;
      if (a.ht_out)   // column j of the h part: chunk j / 32, slot (j % 32) / 8, half (j % 8) / 4 (include/vunet_seq_tiled.h)
        a.ht_out[((size_t)(nb * (a.H >> 5) + (j >> 5)) * 2 + ((j >> 2) & 1)) * 256 + ((((j >> 3) & 3) * 16 + i) * 4) + (j & 3)] = h;
    }
    // the encoder's next input rows: workgroup g copies rows g, g + gridDim.x, ...
    if (a.x_next && wave == 0)
      for (int b = blockIdx.x; b < a.B; b += gridDim.x)
        for (int r = lane; r < a.n; r += 64) a.xh_next[(size_t)b * a.ldx + r] = a.x_next[b * a.seq_stride + r];
    return;
  }
  if (a.S > 1) {   // a raw partial slab: the consumer (seq_coupling_kernel) adds the slabs in slab order, the bias and the tanh
    float* ys = a.y + ((size_t)net * a.S + blockIdx.y) * a.Bp * a.M + col;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(ys + (size_t)(nb * 16 + i) * a.M + 16 * rt) = v[rt];
    return;
  }
  float* y = a.y + (size_t)net * a.Bp * a.M + col;
  const float* bias = a.bias[net];
  const int act = a.act[net];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + col + 16 * rt);
    float4 t = add4(v[rt], bv);
    t = make_float4(seq_act(t.x, act), seq_act(t.y, act), seq_act(t.z, act), seq_act(t.w, act));
    if constexpr (LAY & 4) {
      // the next layer's operand tile: this lane's four columns m0 + 4 kq .. + 3 of batch row i are chunk m0 / 32, slot
      // 2 (m0 / 16 & 1) + kq / 2, half kq & 1 there -- one of its lane slots' float4
      const int mm = m0 + 16 * rt;
      float* yt = a.y + (size_t)net * a.Bp * a.M + ((size_t)(nb * (a.M >> 5) + (mm >> 5)) * 2 + (kq & 1)) * 256 +
                  (i + 16 * (2 * ((mm >> 4) & 1) + (kq >> 1))) * 4;
      *reinterpret_cast<float4*>(yt) = t;
      if (a.y2) *reinterpret_cast<float4*>(a.y2 + (size_t)net * a.Bp * a.M + col + (size_t)(nb * 16 + i) * a.M + 16 * rt) = t;
    } else {
      *reinterpret_cast<float4*>(y + (size_t)(nb * 16 + i) * a.M + 16 * rt) = t;
    }
  }
}

struct SeqCouplingArgs {
  const float* in;       // [B..][ld_in]: xa = in[b][0..c1), xk = in[b][c1..C)
  const float* st;       // [2][S][Bp][Mp]: the scale net's (0) and the translation net's (1) head; NULL: no coupling (v = in).
                         // S = 1: finished values (s already through tanh); S > 1: raw partial slabs, bias_s / bias_t apply
  const float* bias_s;
  const float* bias_t;
  const int* map;        // [C] out[c] = v[map[c]]; NULL: identity
  const float* scale;    // [C] ActNorm scale / loc (NULL: none)
  const float* loc;
  float* out;            // [B..][ld_out]
  float* logdet;         // [B] forward only: += sum(s) + sum log|scale|
  int B, Bp, C, c1, ld_in, ld_out, Mp, reverse, affine_on_src, S;
};

// grid (B, Y), 256 threads: batch row b, channels blockIdx.y * 256 + t, + 256 Y, ...  With a log-determinant to accumulate Y = 1
// (one workgroup owns the row's sum: fixed order, no atomics); otherwise the channels spread over the chip.
// A thread takes its channels four at a time and requests what they need together -- the shuffle indices, then inputs, head
// values / slabs, biases and ActNorm parameters -- two dependent round trips per four channels instead of two per channel (the
// forward pass with its log-determinant is one workgroup per row: 7.6 -> 5 us per launch); the row's sum keeps its order.
__global__ __launch_bounds__(256) void seq_coupling_kernel(SeqCouplingArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* __restrict__ in = a.in + (size_t)b * a.ld_in;
  float ld_sum = 0.f;
  constexpr int U = 4;
  const int stride = 256 * gridDim.y;
  const size_t slab = (size_t)a.Bp * a.Mp;
  for (int c0 = blockIdx.y * 256 + threadIdx.x; c0 < a.C; c0 += U * stride) {
    int j[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * stride;
      ok[u] = c < a.C;
      j[u] = ok[u] ? (a.map ? a.map[c] : c) : 0;
    }
    float v[U], sc[U], lo[U], bs[U], bt[U], vs[U][8], vt[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = in[j[u]];
      const bool cpl = ok[u] && a.st && j[u] >= a.c1;
      const int q = cpl ? j[u] - a.c1 : 0;
      const float* ps = cpl ? a.st + (size_t)b * a.Mp + q : in;   // (not read unless cpl)
      const float* pt = ps + (size_t)a.S * slab;
#pragma unroll
      for (int p = 0; p < 8; ++p) {   // (S <= 8; S = 1: finished values, s already through tanh)
        vs[u][p] = (cpl && p < a.S) ? ps[p * slab] : 0.f;
        vt[u][p] = (cpl && p < a.S) ? pt[p * slab] : 0.f;
      }
      bs[u] = (cpl && a.S > 1) ? a.bias_s[q] : 0.f;
      bt[u] = (cpl && a.S > 1) ? a.bias_t[q] : 0.f;
      const int pi = a.affine_on_src ? j[u] : c0 + u * stride;
      sc[u] = (ok[u] && a.scale) ? a.scale[pi] : 1.f;
      lo[u] = (ok[u] && a.scale) ? a.loc[pi] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float x = v[u];
      if (a.st && j[u] >= a.c1) {
        float s_ = vs[u][0], t_ = vt[u][0];
        if (a.S > 1) {   // raw partial slabs, added in slab order; then bias, and tanh for the scale net
#pragma unroll
          for (int p = 1; p < 8; ++p)
            if (p < a.S) {
              s_ += vs[u][p];
              t_ += vt[u][p];
            }
          s_ = tanhf(s_ + bs[u]);
          t_ += bt[u];
        }
        if (a.reverse) x = (x - t_) * expf(-s_);   // models/flow/blocks.py:316
        else {
          x = x * expf(s_) + t_;                   // :304
          ld_sum += s_;                            // :306
        }
      }
      if (a.scale) {
        if (a.reverse) x = x / sc[u] - lo[u];      // lib/modules.py:327
        else {
          x = sc[u] * (x + lo[u]);                 // :307
          ld_sum += logf(fabsf(sc[u]));            // :313-314 (H = W = 1)
        }
      }
      a.out[(size_t)b * a.ld_out + c0 + u * stride] = x;
    }
  }
  if (a.logdet) {
    ld_sum = wave_sum(ld_sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ld_sum;
    __syncthreads();
    if (threadIdx.x == 0) a.logdet[b] += (red[0] + red[1]) + (red[2] + red[3]);
  }
}

struct SeqDecOutArgs {
  const float* xh;      // [Bp][ldx] the operand rows the fused gate launch just wrote: h at hoff
  float* xh_w;          // the same buffer: x' goes to columns 0 .. n
  const float* w_out;   // [n][H]
  const float* b_out;
  float* xraw;          // [Bp][ldraw] the step's input pose (the residual), replaced by x'
  float* xs;            // row b at xs + b * seq_stride
  float* cs;
  long long seq_stride;
  int B, H, ldx, hoff, n, ldraw;
};

// The decoder's second launch of a step (models/pose_behavior_rnn.py:504-506): x' = n_out(h) + x, xs[b] = x', cs[b] = x, and x'
// becomes the next operand.  grid (B, ceil(n / 8)), 256 threads, dynamic LDS: H floats; wave w of workgroup y owns output rows
// 8 y + w and 8 y + w + 4.
__global__ __launch_bounds__(256) void seq_dec_out_kernel(SeqDecOutArgs a) {
  extern __shared__ float hs[];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < a.H; j += 256) hs[j] = a.xh[(size_t)b * a.ldx + a.hoff + j];
  __syncthreads();
  const int r0 = blockIdx.y * 8 + wave, r1 = r0 + 4;
  const bool v0 = r0 < a.n, v1 = r1 < a.n;
  const float* w0 = a.w_out + (size_t)(v0 ? r0 : 0) * a.H;
  const float* w1 = a.w_out + (size_t)(v1 ? r1 : 0) * a.H;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
  for (int j = lane; j < a.H; j += 64) {
    const float h = hs[j];
    s0 += w0[j] * h;
    s1 += w1[j] * h;
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = q ? r1 : r0;
      if (r >= a.n) continue;
      const float res = a.xraw[(size_t)b * a.ldraw + r];
      const float x2 = ((q ? s1 : s0) + a.b_out[r]) + res;
      a.xs[b * a.seq_stride + r] = x2;
      a.cs[b * a.seq_stride + r] = res;
      a.xraw[(size_t)b * a.ldraw + r] = x2;
      a.xh_w[(size_t)b * a.ldx + r] = x2;
    }
  }
}

// bias_perm[4 j + q] = b_ih[q H + j] + b_hh[q H + j] (+ fold[q H + j]): the gate-interleaved row order of the fused launch
__global__ __launch_bounds__(256) void seq_lstm_bias_kernel(const float* b_ih, const float* b_hh, const float* fold, int H, float* out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4 * H) return;
  const int j = idx >> 2, q = idx & 3;
  float v = b_ih[q * H + j] + b_hh[q * H + j];
  if (fold) v += fold[q * H + j];
  out[idx] = v;
}

// first operand row [x0 | 0 | h0] and the initial state; grid B, 256 threads
__global__ __launch_bounds__(256) void seq_start_kernel(const float* x0, long long x0_stride, const float* h0, const float* c0,
                                                        float* xraw, int ldraw, float* xh, int ldx, int hoff, float* c, int n, int H) {
  const int b = blockIdx.x;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float v = x0[b * x0_stride + r];
    if (xraw) xraw[(size_t)b * ldraw + r] = v;
    xh[(size_t)b * ldx + r] = v;
  }
  for (int j = threadIdx.x; j < H; j += 256) {
    xh[(size_t)b * ldx + hoff + j] = h0 ? h0[(size_t)b * H + j] : 0.f;
    c[(size_t)b * H + j] = c0 ? c0[(size_t)b * H + j] : 0.f;
  }
}

// ``linear_in_decoder``: x = n_in(x) in front of the cell (models/pose_behavior_rnn.py:494-495) folded into the gate matrix:
// W_ih (W_in x + b_in) = (W_ih W_in) x + W_ih b_in.  One thread per element of the [M][n] product; bias by column n.
__global__ __launch_bounds__(256) void seq_fold_input_kernel(const float* w_ih, const float* w_in, const float* b_in, int M, int n,
                                                             float* w_fold, float* bias_fold) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * (n + 1)) return;
  const int m = idx / (n + 1), q = idx - m * (n + 1);
  float s = 0.f;
  for (int r = 0; r < n; ++r) s += w_ih[(size_t)m * n + r] * (q < n ? w_in[(size_t)r * n + q] : b_in[r]);
  if (q < n) w_fold[(size_t)m * n + q] = s;
  else bias_fold[m] = s;
}

// heads: [2][Bp][Mp] = (mu, logstd) from seq_linear_kernel; b = eps * exp(logstd) + mu  (models/pose_behavior_rnn.py:180-201)
__global__ __launch_bounds__(256) void seq_bottleneck_kernel(const float* heads, int Bp, int Mp, const float* eps, float* mu,
                                                             float* logstd, float* bout, int B, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const float m = heads[(size_t)b * Mp + j], l = heads[((size_t)Bp + b) * Mp + j];
  mu[idx] = m;
  logstd[idx] = l;
  if (bout) bout[idx] = eps ? eps[idx] * expf(l) + m : m;
}

// dst[row_off + row_mul m][col_off + k] = src[m][k] * (row_scale ? row_scale[m] : 1) for m < M, k < K; dst is a zero-filled
// padded image (row_mul = 4, row_off = q: gate q of an LSTM into the gate-interleaved row order)
__global__ __launch_bounds__(256) void seq_pack_rows_kernel(const float* src, int M, int K, const float* row_scale, float* dst,
                                                            int ld_dst, int col_off, int row_off, int row_mul) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * K) return;
  const int m = (int)(idx / K), k = (int)(idx - (size_t)m * K);
  dst[(size_t)(row_off + row_mul * m) * ld_dst + col_off + k] = src[idx] * (row_scale ? row_scale[m] : 1.f);
}

// NormConv2d with a 1x1 kernel as a linear layer (lib/modules.py:135-145): row_scale[m] = gamma g / ||v_m||,
// bias_eff[m] = gamma bias + beta.  One wave per row.
__global__ __launch_bounds__(256) void seq_normlinear_rows_kernel(const float* v, const float* g, const float* bias, const float* gamma,
                                                                  const float* beta, int M, int K, float* row_scale, float* bias_eff) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float t = v[(size_t)m * K + k];
    s += t * t;
  }
  s = wave_sum(s);
  if (lane == 0) {
    row_scale[m] = gamma[m] * (g[m] / sqrtf(s));
    bias_eff[m] = gamma[m] * bias[m] + beta[m];
  }
}

// ActNorm's data-dependent initialisation (lib/modules.py:270-290): loc = -mean, scale = 1 / (std + 1e-6) per channel over the
// batch (unbiased std, torch.Tensor.std); one thread per channel
__global__ __launch_bounds__(256) void seq_actnorm_init_kernel(const float* x, int ld, int B, int C, float* loc, float* scale) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float m = 0.f;
  for (int b = 0; b < B; ++b) m += x[(size_t)b * ld + c];
  m /= (float)B;
  float v = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = x[(size_t)b * ld + c] - m;
    v += d * d;
  }
  loc[c] = -m;
  scale[c] = 1.f / (sqrtf(v / (float)(B - 1)) + 1e-6f);
}

// Decoded pose vectors -> pixel keypoints, what the reference does per frame in numpy (float64, since its camera matrix is):
// unNormalizeData (data/data_conversions_3d.py:178-211): full[dims_to_use] = x, full = full * std + mean;
// apply_affine_transform (:588-605): cam = [R | t] . (X, 1); camera_projection (:892-912): (u, v) = (fx X/Z + x0, fy Y/Z + y0);
// then the joint rescale to the synthesis resolution (:1139-1140).  One thread per (frame, joint).
__global__ __launch_bounds__(256) void seq_pose_project_kernel(const float* x, int n_use, const int* dims_to_use, const double* mean,
                                                               const double* stdv, int f32_math, const double* cam, float* kps, int T,
                                                               int J) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * J) return;
  const int t = idx / J, j = idx - t * J;
  double p[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int d = 3 * j + a;
    // position of dimension d among the used ones (dims_to_use is ascending); unused dimensions hold 0 * std + mean
    int lo = 0, hi = n_use;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (dims_to_use[mid] < d) lo = mid + 1;
      else hi = mid;
    }
    const float v = (lo < n_use && dims_to_use[lo] == d) ? x[(size_t)t * n_use + lo] : 0.f;
    // numpy's promotion: float32 arithmetic when the statistics are float32 arrays, float64 when they are float64
    p[a] = f32_math ? (double)(v * (float)stdv[d] + (float)mean[d]) : (double)v * stdv[d] + mean[d];
  }
  // cam: 12 doubles of [R | t] row-major, then fx, x0, fy, y0, then the two rescale factors
  double c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) c[r] = ((p[0] * cam[4 * r] + p[1] * cam[4 * r + 1]) + p[2] * cam[4 * r + 2]) + cam[4 * r + 3];
  const double xn = c[0] / c[2], yn = c[1] / c[2], zn = c[2] / c[2];
  const double u = (xn * cam[12] + yn * 0.0) + zn * cam[13], w = (xn * 0.0 + yn * cam[14]) + zn * cam[15];
  kps[2 * (size_t)idx] = (float)(u * cam[16]);
  kps[2 * (size_t)idx + 1] = (float)(w * cam[17]);
}

// wt[mt][c][h][l][e] = w[16 mt + (l & 15)][32 c + 8 (l >> 4) + 4 h + e]: one thread per float4 of the image
__global__ __launch_bounds__(256) void seq_pack_tiles_kernel(const float* __restrict__ w, int ld, int M, int K, float* __restrict__ wt) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // float4 index
  if (idx >= (size_t)M * K / 4) return;
  const int l = (int)(idx & 63), h = (int)((idx >> 6) & 1);
  const size_t tc = idx >> 7;                                   // mt * (K / 32) + c
  const int nck = K >> 5, mt = (int)(tc / nck), c = (int)(tc - (size_t)mt * nck);
  reinterpret_cast<float4*>(wt)[idx] =
      *reinterpret_cast<const float4*>(w + (size_t)(16 * mt + (l & 15)) * ld + 32 * c + 8 * (l >> 4) + 4 * h);
}

}  // namespace

extern "C" int vunet_seq_pack_tiles(const float* w, int32_t ld, int32_t M, int32_t K, float* wt, void* stream) {
  if (!w || !wt || M < 16 || M % 16 || K < 32 || K % 32 || ld < K || ld % 4) return VUNET_ERR_ARG;
  const size_t n4 = (size_t)M * K / 4;
  VUNET_LAUNCH(seq_pack_tiles_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ld, M, K, wt);
  return vunet_check_launch();
}

extern "C" int vunet_seq_pose_project(const float* x, int32_t n_use, const int32_t* dims_to_use, const double* mean, const double* stdv,
                                      int32_t D, int32_t f32_math, const double* cam, float* kps, int32_t T, int32_t J, void* stream) {
  if (!x || !dims_to_use || !mean || !stdv || !cam || !kps || n_use < 1 || T < 1 || J < 1 || D < 3 * J) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_pose_project_kernel, dim3((T * J + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n_use, dims_to_use, mean,
               stdv, f32_math, cam, kps, T, J);
  return vunet_check_launch();
}

extern "C" int vunet_seq_actnorm_init(const float* x, int32_t ld, int32_t B, int32_t C, float* loc, float* scale, void* stream) {
  if (!x || !loc || !scale || B < 2 || C < 1 || ld < C) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_actnorm_init_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ld, B, C, loc, scale);
  return vunet_check_launch();
}

static int seq_linear_launch(const vunet_seq_linear_desc* d, int layout, const float* w0, const float* w1, const float* x,
                             const float* bias0, const float* bias1, float* y, float* y2, void* stream) {
  if (!d || !w0 || !x || !y) return VUNET_ERR_ARG;
  if (layout < 0 || layout > 7 || layout == 1) return VUNET_ERR_UNSUPPORTED;   // (inference: 3 / 5 / 7; training: 2 / 4 / 6)
  if (y2 && !(layout & 4)) return VUNET_ERR_ARG;
  if ((layout & 4) && (d->S != 1 || d->M % 32)) return VUNET_ERR_ARG;
  if (d->nets < 1 || d->nets > 2 || (d->nets == 2 && !w1)) return VUNET_ERR_ARG;
  if (d->B < 1 || d->B > 16 * SEQ_MAX_NB || d->M < 16 || d->M % 16) return VUNET_ERR_ARG;
  if (d->S < 1 || d->S > 8 || d->K < 32 || d->K % (32 * d->S) || d->ldx < d->K || d->ldx % 4) return VUNET_ERR_ARG;
  if (d->act0 < 0 || d->act0 > 2 || d->act1 < 0 || d->act1 > 2) return VUNET_ERR_ARG;
  SeqLinearArgs a;
  a.w[0] = w0;
  a.w[1] = w1;
  a.x = x;
  a.bias[0] = bias0;
  a.bias[1] = bias1;
  a.y = y;
  a.M = d->M;
  a.K = d->K;
  a.Bp = (d->B + 15) / 16 * 16;
  a.ldx = d->ldx;
  a.act[0] = d->act0;
  a.act[1] = d->act1;
  a.shared_in = d->shared_in;
  a.S = d->S;
  a.c_in = nullptr; a.c_out = nullptr; a.xh_next = nullptr; a.h_out = nullptr; a.x_next = nullptr; a.gates_out = nullptr;
  a.y2 = y2;
  a.seq_stride = 0; a.H = a.hoff = a.n = a.B = 0;
  // (RT = 2, a 32-row tile per workgroup, halves the operand traffic per weight byte but leaves half the CUs without a workgroup
  // at every layer size of the reference configuration: not instantiated)
  const dim3 grid(d->M / 16, d->S, d->nets);
  hipStream_t st = (hipStream_t)stream;
  // 16 waves where K gives each at least one chunk -- up to 32 batch rows.  With 3 - 4 batch tiles (the training batch of 64) a
  // wave's operand loads are four times its weight loads and four waves with deeper groups are faster (flow training step at
  // 64 rows: 7.77 -> 7.41 ms; at 16 rows the two forms measured the same, profiles/r05_seq_time.txt)
  const bool wide = d->K / d->S >= 16 * 32 && d->B <= 32;
#define SEQ_LINEAR_LAY(NB, LAY)                                                                     \
  if (wide) VUNET_LAUNCH((seq_linear_kernel<NB, 1, 16, false, LAY>), grid, dim3(1024), 0, st, a);   \
  else VUNET_LAUNCH((seq_linear_kernel<NB, 1, 4, false, LAY>), grid, dim3(256), 0, st, a);
#define SEQ_LINEAR_CASE(NB)                        \
  case NB:                                         \
    if (layout == 0) { SEQ_LINEAR_LAY(NB, 0) }     \
    else if (layout == 2) { SEQ_LINEAR_LAY(NB, 2) } \
    else if (layout == 3) { SEQ_LINEAR_LAY(NB, 3) } \
    else if (layout == 4) { SEQ_LINEAR_LAY(NB, 4) } \
    else if (layout == 5) { SEQ_LINEAR_LAY(NB, 5) } \
    else if (layout == 6) { SEQ_LINEAR_LAY(NB, 6) } \
    else { SEQ_LINEAR_LAY(NB, 7) }                 \
    break;
  switch (a.Bp / 16) {
    SEQ_LINEAR_CASE(1)
    SEQ_LINEAR_CASE(2)
    SEQ_LINEAR_CASE(3)
    SEQ_LINEAR_CASE(4)
    default: return VUNET_ERR_ARG;
  }
#undef SEQ_LINEAR_CASE
#undef SEQ_LINEAR_LAY
  return vunet_check_launch();
}

extern "C" int vunet_seq_linear(const vunet_seq_linear_desc* d, const float* w0, const float* w1, const float* x, const float* bias0,
                                const float* bias1, float* y, void* stream) {
  return seq_linear_launch(d, 0, w0, w1, x, bias0, bias1, y, nullptr, stream);
}

extern "C" int vunet_seq_linear_tiled(const vunet_seq_linear_desc* d, int32_t layout, const float* w0, const float* w1, const float* x,
                                      const float* bias0, const float* bias1, float* y, float* y_rowmajor, void* stream) {
  return seq_linear_launch(d, layout, w0, w1, x, bias0, bias1, y, y_rowmajor, stream);
}

extern "C" int vunet_seq_coupling(const vunet_seq_coupling_desc* d, const float* in, const float* st, const float* bias_s,
                                  const float* bias_t, const int32_t* map, const float* scale, const float* loc, float* out,
                                  float* logdet, void* stream) {
  if (!d || !in || !out || d->B < 1 || d->C < 1 || d->ld_in < d->C || d->ld_out < d->C) return VUNET_ERR_ARG;
  if (st && (d->c1 < 0 || d->c1 > d->C || d->Mp < d->C - d->c1 || d->S < 1 || d->S > 8)) return VUNET_ERR_ARG;
  if (st && d->S > 1 && (!bias_s || !bias_t)) return VUNET_ERR_ARG;
  if ((scale == nullptr) != (loc == nullptr)) return VUNET_ERR_ARG;
  if (in == out && map) return VUNET_ERR_ARG;   // a gather cannot run in place
  SeqCouplingArgs a;
  a.in = in;
  a.st = st;
  a.bias_s = bias_s;
  a.bias_t = bias_t;
  a.S = st ? d->S : 1;
  a.map = map;
  a.scale = scale;
  a.loc = loc;
  a.out = out;
  a.logdet = d->reverse ? nullptr : logdet;
  a.B = d->B;
  a.Bp = (d->B + 15) / 16 * 16;
  a.C = d->C;
  a.c1 = d->c1;
  a.ld_in = d->ld_in;
  a.ld_out = d->ld_out;
  a.Mp = d->Mp;
  a.reverse = d->reverse;
  a.affine_on_src = d->affine_on_src;
  VUNET_LAUNCH(seq_coupling_kernel, dim3(d->B, a.logdet ? 1 : (d->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return vunet_check_launch();
}

extern "C" int vunet_seq_start(const float* x0, int64_t x0_stride, const float* h0, const float* c0, float* xraw, int32_t ldraw,
                               float* xh, int32_t ldx, int32_t hoff, float* c, int32_t B, int32_t n, int32_t H, void* stream) {
  if (!x0 || !xh || !c || B < 1 || n < 1 || H < 1 || hoff < n || ldx < hoff + H) return VUNET_ERR_ARG;
  if (xraw && ldraw < n) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_start_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x0, (long long)x0_stride, h0, c0, xraw, ldraw, xh, ldx,
               hoff, c, n, H);
  return vunet_check_launch();
}

static int seq_lstm_launch(const vunet_seq_lstm_desc* d, const float* w_perm, const float* xh, const float* bias_perm,
                           const float* c_in, float* c_out, float* xh_next, float* h_out, const float* x_next, float* gates_out,
                           bool w_tiled, void* stream, const float* ht_in = nullptr, float* ht_out = nullptr) {
  if (!d || !w_perm || !xh || !bias_perm || !c_in || !c_out || c_in == c_out || !xh_next || xh_next == xh) return VUNET_ERR_ARG;
  if (d->B < 1 || d->B > 16 * SEQ_MAX_NB || d->H < 4 || d->H % 4 || d->n < 1 || d->hoff < d->n || d->ldx < d->hoff + d->H) return VUNET_ERR_ARG;
  if (d->ldx % 32) return VUNET_ERR_ARG;
  SeqLinearArgs a;
  a.w[0] = w_perm;
  a.w[1] = nullptr;
  a.x = xh;
  a.bias[0] = bias_perm;
  a.bias[1] = nullptr;
  a.y = nullptr;
  a.M = 4 * d->H;
  a.K = d->ldx;
  a.Bp = (d->B + 15) / 16 * 16;
  a.ldx = d->ldx;
  a.act[0] = a.act[1] = 0;
  a.shared_in = 1;
  a.S = 1;
  a.c_in = c_in;
  a.c_out = c_out;
  a.xh_next = xh_next;
  a.h_out = h_out;
  a.x_next = x_next;
  a.y2 = nullptr;
  a.gates_out = gates_out;
  a.ht_in = ht_in;
  a.ht_out = ht_out;
  a.seq_stride = d->seq_stride;
  a.H = d->H;
  a.hoff = d->hoff;
  a.n = d->n;
  a.B = d->B;
  if ((ht_in || ht_out) && (!w_tiled || d->H % 32 || d->hoff % 32 || d->ldx != d->hoff + d->H || ht_in == ht_out)) return VUNET_ERR_ARG;
  const dim3 grid(a.M / 16, 1, 1);
  hipStream_t st = (hipStream_t)stream;
  const bool wide = a.K >= 16 * 32 && d->B <= 32;   // (3 - 4 batch tiles: four waves, as vunet_seq_linear; 50-step roll-out at 64 rows 1.03 -> 0.93 ms)
  if (ht_in && !wide && a.Bp >= 48) {   // the h part of the operand from tiles: the four-wave form at 3 - 4 batch tiles
    if (a.Bp == 48) VUNET_LAUNCH((seq_linear_kernel<3, 1, 4, true, 25>), grid, dim3(256), 0, st, a);
    else VUNET_LAUNCH((seq_linear_kernel<4, 1, 4, true, 25>), grid, dim3(256), 0, st, a);
    return vunet_check_launch();
  }
#define SEQ_LSTM_CASE(NB)                                                                                   \
  case NB:                                                                                                  \
    if (w_tiled) {                                                                                          \
      if (wide) VUNET_LAUNCH((seq_linear_kernel<NB, 1, 16, true, 9>), grid, dim3(1024), 0, st, a);          \
      else VUNET_LAUNCH((seq_linear_kernel<NB, 1, 4, true, 9>), grid, dim3(256), 0, st, a);                 \
    } else if (wide) VUNET_LAUNCH((seq_linear_kernel<NB, 1, 16, true>), grid, dim3(1024), 0, st, a);        \
    else VUNET_LAUNCH((seq_linear_kernel<NB, 1, 4, true>), grid, dim3(256), 0, st, a);                      \
    break;
  switch (a.Bp / 16) {
    SEQ_LSTM_CASE(1)
    SEQ_LSTM_CASE(2)
    SEQ_LSTM_CASE(3)
    SEQ_LSTM_CASE(4)
    default: return VUNET_ERR_ARG;
  }
#undef SEQ_LSTM_CASE
  return vunet_check_launch();
}

extern "C" int vunet_seq_lstm_gates(const vunet_seq_lstm_desc* d, const float* w_perm, const float* xh, const float* bias_perm,
                                    const float* c_in, float* c_out, float* xh_next, float* h_out, const float* x_next, void* stream) {
  return seq_lstm_launch(d, w_perm, xh, bias_perm, c_in, c_out, xh_next, h_out, x_next, nullptr, false, stream);
}

// the same step on a TILE-MAJOR gate image (vunet_seq_pack_tiles of w_perm; include/vunet_seq_tiled.h): gates_out as
// vunet_seq_lstm_gates_train (NULL: not kept)
extern "C" int vunet_seq_lstm_gates_tiled(const vunet_seq_lstm_desc* d, const float* w_tiles, const float* xh, const float* bias_perm,
                                          const float* c_in, float* c_out, float* xh_next, float* h_out, const float* x_next,
                                          float* gates_out, void* stream) {
  return seq_lstm_launch(d, w_tiles, xh, bias_perm, c_in, c_out, xh_next, h_out, x_next, gates_out, true, stream);
}

// ... with the h part of the operand rows read from / written as tiles as well (either may be NULL: the first step of a
// sequence reads the rows vunet_seq_start wrote; batches of <= 32 rows read the rows anyway)
extern "C" int vunet_seq_lstm_gates_tiled_h(const vunet_seq_lstm_desc* d, const float* w_tiles, const float* xh, const float* h_tiles_in,
                                            const float* bias_perm, const float* c_in, float* c_out, float* xh_next,
                                            float* h_tiles_out, float* h_out, const float* x_next, float* gates_out, void* stream) {
  return seq_lstm_launch(d, w_tiles, xh, bias_perm, c_in, c_out, xh_next, h_out, x_next, gates_out, true, stream, h_tiles_in,
                         h_tiles_out);
}

// the same step with the gate activations kept for the backward pass (include/vunet_seq_train.h)
extern "C" int vunet_seq_lstm_gates_train(const vunet_seq_lstm_desc* d, const float* w_perm, const float* xh, const float* bias_perm,
                                          const float* c_in, float* c_out, float* xh_next, float* h_out, const float* x_next,
                                          float* gates_out, void* stream) {
  if (!gates_out) return VUNET_ERR_ARG;
  return seq_lstm_launch(d, w_perm, xh, bias_perm, c_in, c_out, xh_next, h_out, x_next, gates_out, false, stream);
}

extern "C" int vunet_seq_decoder_out(const vunet_seq_lstm_desc* d, float* xh, const float* w_out, const float* b_out, float* xraw, float* xs,
                                     float* cs, void* stream) {
  if (!d || !xh || !w_out || !b_out || !xraw || !xs || !cs || d->B < 1 || d->H < 1 || d->n < 1) return VUNET_ERR_ARG;
  if (d->hoff < d->n || d->ldx < d->hoff + d->H || d->ldraw < d->n) return VUNET_ERR_ARG;
  const size_t lds = (size_t)d->H * sizeof(float);
  if (lds > 64 * 1024) return VUNET_ERR_UNSUPPORTED;
  SeqDecOutArgs a;
  a.xh = xh;
  a.xh_w = xh;
  a.w_out = w_out;
  a.b_out = b_out;
  a.xraw = xraw;
  a.xs = xs;
  a.cs = cs;
  a.seq_stride = d->seq_stride;
  a.B = d->B;
  a.H = d->H;
  a.ldx = d->ldx;
  a.hoff = d->hoff;
  a.n = d->n;
  a.ldraw = d->ldraw;
  VUNET_LAUNCH(seq_dec_out_kernel, dim3(d->B, (d->n + 7) / 8), dim3(256), lds, (hipStream_t)stream, a);
  return vunet_check_launch();
}

extern "C" int vunet_seq_lstm_bias(const float* b_ih, const float* b_hh, const float* fold, int32_t H, float* bias_perm, void* stream) {
  if (!b_ih || !b_hh || !bias_perm || H < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_lstm_bias_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, b_ih, b_hh, fold, H, bias_perm);
  return vunet_check_launch();
}

extern "C" int vunet_seq_fold_input(const float* w_ih, const float* w_in, const float* b_in, int32_t M, int32_t n, float* w_fold,
                                    float* bias_fold, void* stream) {
  if (!w_ih || !w_in || !b_in || !w_fold || !bias_fold || M < 1 || n < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_fold_input_kernel, dim3((M * (n + 1) + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ih, w_in, b_in, M, n,
               w_fold, bias_fold);
  return vunet_check_launch();
}

extern "C" int vunet_seq_bottleneck(const float* heads, int32_t Mp, const float* eps, float* mu, float* logstd, float* b_out, int32_t B,
                                    int32_t H, void* stream) {
  if (!heads || !mu || !logstd || B < 1 || H < 1 || Mp < H) return VUNET_ERR_ARG;
  const int Bp = (B + 15) / 16 * 16;
  VUNET_LAUNCH(seq_bottleneck_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, heads, Bp, Mp, eps, mu, logstd,
               b_out, B, H);
  return vunet_check_launch();
}

extern "C" int vunet_seq_pack_rows(const float* src, int32_t M, int32_t K, const float* row_scale, float* dst, int32_t ld_dst,
                                   int32_t col_off, int32_t row_off, int32_t row_mul, void* stream) {
  if (!src || !dst || M < 1 || K < 1 || col_off < 0 || ld_dst < col_off + K || row_off < 0 || row_mul < 1) return VUNET_ERR_ARG;
  const size_t n = (size_t)M * K;
  VUNET_LAUNCH(seq_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, M, K, row_scale, dst,
               ld_dst, col_off, row_off, row_mul);
  return vunet_check_launch();
}

extern "C" int vunet_seq_normlinear_rows(const float* v, const float* g, const float* bias, const float* gamma, const float* beta,
                                         int32_t M, int32_t K, float* row_scale, float* bias_eff, void* stream) {
  if (!v || !g || !bias || !gamma || !beta || !row_scale || !bias_eff || M < 1 || K < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_normlinear_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, g, bias, gamma, beta, M, K,
               row_scale, bias_eff);
  return vunet_check_launch();
}
