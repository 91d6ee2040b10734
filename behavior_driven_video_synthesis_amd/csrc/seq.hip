// Behaviour front half of BASELINE config 5: flow sample -> pose_behavior_rnn decode, on pose vectors [B, n_kps] and behaviour
// codes [B, C] (experiments/behavior_net.py:1086-1100, :1173-1184).
//
// Everything here is a chain of skinny matrix products  Y[B, M] = f(X)[B, K] . W[M, K]^T  with B <= 64 rows and weights of
// 0.2 - 17 MB per layer (the flow of config/behavior_net.yaml: 60 MLPs of 512-2048-2048-2048-512, 2.5 GB of fp32 weights per
// sample pass; the decoder: one [4096, 1075] gate matrix read 50 times).  Each weight is used once per pass, so the bound is
// the weight stream from HBM, not the matrix cores: the kernels are laid out for that.
//
//   * seq_linear_kernel: one wave owns 16 output rows; a K chunk of 32 is 128 contiguous bytes per weight row, two 16-byte
//     loads per lane, eight v_mfma_f32_16x16x4_f32 (exact fp32: the reference computes in fp32) against the batch tile.  The
//     four waves of a workgroup and the S_out workgroups of a row tile split K, so that even the 512-row layers put >= 256
//     workgroups on the chip; a workgroup adds its waves through LDS and stores ONE partial slab, in a fixed order.  Nothing is
//     reduced with atomics: the consumer adds the S partial slabs, the producing layer's bias and its activation while it loads
//     its operand ("X = act(sum_s P[s] + bias)"), so an MLP is one launch per layer and bit-reproducible.  The s and t nets of
//     a coupling (same input, same shapes) and the mu / logstd heads run as one launch (grid z).
//   * seq_coupling_kernel: everything between two MLP evaluations of the flow -- the affine coupling itself, the half swap,
//     ``Shuffle`` and ``ActNorm`` -- as "v = couple(in); out[c] = affine(v[map[c]])" with a host-composed index map.
//   * seq_lstm_kernel: the gate nonlinearities and state update of one LSTM step from the gate partials, the decoder's output
//     layer + residual (``ResidualRNNDecoder.forward``), the optional input layer, and the next step's operand row
//     [x | 0 | h] -- one workgroup per batch row, two launches per time step.
#include "common.h"

namespace {

constexpr int SEQ_MAX_NB = 4;   // batch tiles of 16: B <= 64

struct SeqLinearArgs {
  const float* w[2];        // [M][K] row-major, M % 16 == 0, K % 32 == 0 (zero padded)
  const float* xin;         // [nets_in][S_in][Bp][ldx]
  const float* bias_in[2];  // [K] bias of the layer that produced xin's partials (NULL: none)
  float* out;               // [nets][S_out][Bp][M]
  int M, K, Bp, ldx, S_in, S_out, act_in, shared_in;
};

__device__ __forceinline__ float4 seq_act4(float4 v, int act) {
  if (act == 1) {   // nn.LeakyReLU() default slope 0.01 (lib/modules.py:244)
    v.x = v.x > 0.f ? v.x : 0.01f * v.x;
    v.y = v.y > 0.f ? v.y : 0.01f * v.y;
    v.z = v.z > 0.f ? v.z : 0.01f * v.z;
    v.w = v.w > 0.f ? v.w : 0.01f * v.w;
  }
  return v;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// grid (M/16, S_out, nets), 256 threads.  Lane l of a wave: i = l & 15 (weight row / batch row), kq = l >> 4; in a K chunk of 32
// it holds k = 8 kq .. 8 kq + 7 of its row -- the same k for the A (weights) and B (batch) operand of MFMA step j = 0..7.
template <int NB>
__global__ __launch_bounds__(256) void seq_linear_kernel(SeqLinearArgs a) {
  __shared__ float4 red[3][NB][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 16, split = blockIdx.y, net = blockIdx.z;
  const int kb = a.K / a.S_out, nchunk = kb >> 5;
  const float* __restrict__ w = a.w[net] + (size_t)(m0 + i) * a.K + (size_t)split * kb + 8 * kq;
  const size_t slab_in = (size_t)a.Bp * a.ldx;
  const float* __restrict__ x = a.xin + (a.shared_in ? 0 : (size_t)net * a.S_in * slab_in) + (size_t)i * a.ldx + (size_t)split * kb + 8 * kq;
  const float* __restrict__ bias = a.bias_in[net] ? a.bias_in[net] + (size_t)split * kb + 8 * kq : nullptr;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = wave; c < nchunk; c += 4) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + 32 * c), w1 = *reinterpret_cast<const float4*>(w + 32 * c + 4);
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (bias) {
      b0 = *reinterpret_cast<const float4*>(bias + 32 * c);
      b1 = *reinterpret_cast<const float4*>(bias + 32 * c + 4);
    }
    const float wa[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float* xp = x + (size_t)nb * 16 * a.ldx + 32 * c;
      float4 x0 = b0, x1 = b1;
      for (int s = 0; s < a.S_in; ++s) {
        x0 = add4(x0, *reinterpret_cast<const float4*>(xp + s * slab_in));
        x1 = add4(x1, *reinterpret_cast<const float4*>(xp + s * slab_in + 4));
      }
      x0 = seq_act4(x0, a.act_in);
      x1 = seq_act4(x1, a.act_in);
      const float xb[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb[j], acc[nb], 0, 0, 0);
    }
  }
  // waves 1..3 -> LDS; wave 0 adds them in wave order and stores the slab.  D: lane holds batch row n = l & 15 of its tile,
  // output rows m0 + 4 (l >> 4) + r.
  if (wave) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) red[wave - 1][nb][lane] = make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]);
  }
  __syncthreads();
  if (wave == 0) {
    float* out = a.out + ((size_t)net * a.S_out + split) * a.Bp * a.M + m0 + 4 * kq;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float4 v = make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]);
#pragma unroll
      for (int q = 0; q < 3; ++q) v = add4(v, red[q][nb][lane]);
      *reinterpret_cast<float4*>(out + (size_t)(nb * 16 + i) * a.M) = v;
    }
  }
}

struct SeqCouplingArgs {
  const float* in;       // [B..][ld_in]: xa = in[b][0..c1), xk = in[b][c1..C)
  const float* st;       // [2][S][Bp][Mp] partials of the s net (0) and the t net (1); NULL: no coupling (v = in)
  const float* bias_s;   // [c2]
  const float* bias_t;
  const int* map;        // [C] out[c] = v[map[c]]; NULL: identity
  const float* scale;    // [C] ActNorm scale / loc (NULL: none)
  const float* loc;
  float* out;            // [B..][ld_out]
  float* logdet;         // [B] forward only: += sum(s) + sum log|scale|
  int B, Bp, C, c1, ld_in, ld_out, S, Mp, reverse, affine_on_src;
};

// grid B, 256 threads: one batch row per workgroup.
__global__ __launch_bounds__(256) void seq_coupling_kernel(SeqCouplingArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* in = a.in + (size_t)b * a.ld_in;
  const size_t slab = (size_t)a.Bp * a.Mp;
  float ld_sum = 0.f;
  for (int c = threadIdx.x; c < a.C; c += 256) {
    const int j = a.map ? a.map[c] : c;
    float v = in[j];
    if (a.st && j >= a.c1) {
      const int q = j - a.c1;
      float s = a.bias_s[q], t = a.bias_t[q];
      for (int p = 0; p < a.S; ++p) {
        s += a.st[p * slab + (size_t)b * a.Mp + q];
        t += a.st[(a.S + p) * slab + (size_t)b * a.Mp + q];
      }
      s = tanhf(s);   // BasicFullyConnectedNet(use_tanh=True) is the scale net (models/flow/blocks.py:283-287)
      if (a.reverse) v = (v - t) * expf(-s);   // :316
      else {
        v = v * expf(s) + t;                   // :304
        ld_sum += s;                           // :306
      }
    }
    if (a.scale) {
      const int pi = a.affine_on_src ? j : c;
      const float sc = a.scale[pi], lo = a.loc[pi];
      if (a.reverse) v = v / sc - lo;          // lib/modules.py:327
      else {
        v = sc * (v + lo);                     // :307
        ld_sum += logf(fabsf(sc));             // :313-314 (H = W = 1)
      }
    }
    a.out[(size_t)b * a.ld_out + c] = v;
  }
  if (a.logdet) {
    ld_sum = wave_sum(ld_sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ld_sum;
    __syncthreads();
    if (threadIdx.x == 0) a.logdet[b] += (red[0] + red[1]) + (red[2] + red[3]);
  }
}

struct SeqLstmArgs {
  const float* gates;   // [S][Bp][4 H] partials of [W_ih | 0 | W_hh] . [x | 0 | h]
  const float* bias;    // [4 H] b_ih
  const float* bias2;   // [4 H] b_hh
  float* c;             // [Bp][H] cell state, in place
  float* xh;            // [Bp][ldx] operand row of the next step: x at 0, h at hoff
  float* h_out;         // [Bp][H] copy of h (NULL: none) -- the encoder's ``pre``
  // decoder (w_out != NULL): x' = n_out(h) + x, xs[b] = x', cs[b] = x; the next operand is n_in(x') or x'
  const float* w_out;   // [n][H]
  const float* b_out;
  const float* w_in;    // [n][n] (NULL: ``linear_in_decoder`` off)
  const float* b_in;
  float* xraw;          // [Bp][64-padded n] the step's input pose (the residual), replaced by x'
  float* xs;            // row b at xs + b * seq_stride
  float* cs;
  // encoder (w_out == NULL): the next input comes from the sequence
  const float* x_next;  // row b at x_next + b * seq_stride (NULL at the last step)
  long long seq_stride;
  int B, Bp, H, S, ldx, hoff, n, ldraw;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// grid B, 256 threads, dynamic LDS: H + 64-padded n floats
__global__ __launch_bounds__(256) void seq_lstm_kernel(SeqLstmArgs a) {
  extern __shared__ float lds[];
  float* hs = lds;
  float* xn = lds + a.H;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slab = (size_t)a.Bp * 4 * a.H;
  for (int j = threadIdx.x; j < a.H; j += 256) {
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = a.bias[q * a.H + j] + a.bias2[q * a.H + j];
      for (int s = 0; s < a.S; ++s) v += a.gates[s * slab + (size_t)b * 4 * a.H + q * a.H + j];
      g[q] = v;
    }
    // gate order i, f, g, o (torch.nn.LSTMCell, models/pose_behavior_rnn.py:476, :498)
    const float c2 = sigmoid_f(g[1]) * a.c[(size_t)b * a.H + j] + sigmoid_f(g[0]) * tanhf(g[2]);
    const float h = sigmoid_f(g[3]) * tanhf(c2);
    a.c[(size_t)b * a.H + j] = c2;
    a.xh[(size_t)b * a.ldx + a.hoff + j] = h;
    if (a.h_out) a.h_out[(size_t)b * a.H + j] = h;
    hs[j] = h;
  }
  if (!a.w_out) {
    if (a.x_next)
      for (int r = threadIdx.x; r < a.n; r += 256) a.xh[(size_t)b * a.ldx + r] = a.x_next[b * a.seq_stride + r];
    return;
  }
  __syncthreads();
  for (int r = wave; r < a.n; r += 4) {   // out = n_out(h) + res (:504-506)
    const float* wr = a.w_out + (size_t)r * a.H;
    float s = 0.f;
    for (int j = lane; j < a.H; j += 64) s += wr[j] * hs[j];
    s = wave_sum(s);
    if (lane == 0) {
      const float res = a.xraw[(size_t)b * a.ldraw + r];
      const float x2 = (s + a.b_out[r]) + res;
      a.xs[b * a.seq_stride + r] = x2;
      a.cs[b * a.seq_stride + r] = res;
      a.xraw[(size_t)b * a.ldraw + r] = x2;
      xn[r] = x2;
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < a.n; r += 256) {
    float v = xn[r];
    if (a.w_in) {   // x = n_in(x) in front of the cell (:494-495)
      v = a.b_in[r];
      for (int q = 0; q < a.n; ++q) v += a.w_in[(size_t)r * a.n + q] * xn[q];
    }
    a.xh[(size_t)b * a.ldx + r] = v;
  }
}

// first operand row [x0 | 0 | h0] and the initial state; grid B, 256 threads, LDS: 64-padded n floats
__global__ __launch_bounds__(256) void seq_start_kernel(const float* x0, long long x0_stride, const float* h0, const float* c0,
                                                        const float* w_in, const float* b_in, float* xraw, int ldraw, float* xh,
                                                        int ldx, int hoff, float* c, int n, int H) {
  extern __shared__ float lds[];
  const int b = blockIdx.x;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float v = x0[b * x0_stride + r];
    lds[r] = v;
    if (xraw) xraw[(size_t)b * ldraw + r] = v;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += 256) {
    float v = lds[r];
    if (w_in) {
      v = b_in[r];
      for (int q = 0; q < n; ++q) v += w_in[(size_t)r * n + q] * lds[q];
    }
    xh[(size_t)b * ldx + r] = v;
  }
  for (int j = threadIdx.x; j < H; j += 256) {
    xh[(size_t)b * ldx + hoff + j] = h0 ? h0[(size_t)b * H + j] : 0.f;
    c[(size_t)b * H + j] = c0 ? c0[(size_t)b * H + j] : 0.f;
  }
}

// mu = sum_s P0[s] + bias0, logstd = sum_s P1[s] + bias1, b = eps * exp(logstd) + mu  (models/pose_behavior_rnn.py:180-201)
__global__ __launch_bounds__(256) void seq_bottleneck_kernel(const float* part, int S, int Bp, int Mp, const float* bias_mu,
                                                             const float* bias_std, const float* eps, float* mu, float* logstd,
                                                             float* bout, int B, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const size_t slab = (size_t)Bp * Mp;
  float m = bias_mu[j], l = bias_std[j];
  for (int s = 0; s < S; ++s) {
    m += part[s * slab + (size_t)b * Mp + j];
    l += part[(S + s) * slab + (size_t)b * Mp + j];
  }
  mu[idx] = m;
  logstd[idx] = l;
  if (bout) bout[idx] = eps ? eps[idx] * expf(l) + m : m;
}

// out[b][j] = act(sum_s part[s][b][j] + bias[j]): the value of a layer whose consumer is not one of the kernels above
__global__ __launch_bounds__(256) void seq_finish_kernel(const float* part, int S, int Bp, int Mp, const float* bias, int act, float* out,
                                                         int ld_out, int B, int M) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * M) return;
  const int b = idx / M, j = idx - b * M;
  float v = bias ? bias[j] : 0.f;
  for (int s = 0; s < S; ++s) v += part[(size_t)s * Bp * Mp + (size_t)b * Mp + j];
  if (act == 1) v = v > 0.f ? v : 0.01f * v;
  else if (act == 2) v = tanhf(v);
  out[(size_t)b * ld_out + j] = v;
}

// dst[m][col_off + k] = src[m][k] * (row_scale ? row_scale[m] : 1) for m < M, k < K; dst is a zero-filled padded image
__global__ __launch_bounds__(256) void seq_pack_rows_kernel(const float* src, int M, int K, const float* row_scale, float* dst,
                                                            int ld_dst, int col_off) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * K) return;
  const int m = (int)(idx / K), k = (int)(idx - (size_t)m * K);
  dst[(size_t)m * ld_dst + col_off + k] = src[idx] * (row_scale ? row_scale[m] : 1.f);
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

}  // namespace

extern "C" int vunet_seq_actnorm_init(const float* x, int32_t ld, int32_t B, int32_t C, float* loc, float* scale, void* stream) {
  if (!x || !loc || !scale || B < 2 || C < 1 || ld < C) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_actnorm_init_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ld, B, C, loc, scale);
  return vunet_check_launch();
}

extern "C" int vunet_seq_linear(const vunet_seq_linear_desc* d, const float* w0, const float* w1, const float* xin,
                                const float* bias_in0, const float* bias_in1, float* out, void* stream) {
  if (!d || !w0 || !xin || !out) return VUNET_ERR_ARG;
  if (d->nets < 1 || d->nets > 2 || (d->nets == 2 && !w1)) return VUNET_ERR_ARG;
  if (d->B < 1 || d->B > 16 * SEQ_MAX_NB || d->M < 16 || d->M % 16 || d->S_in < 1 || d->S_out < 1) return VUNET_ERR_ARG;
  if (d->K < 32 || d->K % (32 * d->S_out) || d->ldx < d->K || d->ldx % 4) return VUNET_ERR_ARG;
  if (d->act_in < 0 || d->act_in > 1) return VUNET_ERR_ARG;
  SeqLinearArgs a;
  a.w[0] = w0;
  a.w[1] = w1;
  a.xin = xin;
  a.bias_in[0] = bias_in0;
  a.bias_in[1] = bias_in1;
  a.out = out;
  a.M = d->M;
  a.K = d->K;
  a.Bp = (d->B + 15) / 16 * 16;
  a.ldx = d->ldx;
  a.S_in = d->S_in;
  a.S_out = d->S_out;
  a.act_in = d->act_in;
  a.shared_in = d->shared_in;
  const dim3 grid(d->M / 16, d->S_out, d->nets);
  hipStream_t st = (hipStream_t)stream;
  switch (a.Bp / 16) {
    case 1: VUNET_LAUNCH(seq_linear_kernel<1>, grid, dim3(256), 0, st, a); break;
    case 2: VUNET_LAUNCH(seq_linear_kernel<2>, grid, dim3(256), 0, st, a); break;
    case 3: VUNET_LAUNCH(seq_linear_kernel<3>, grid, dim3(256), 0, st, a); break;
    default: VUNET_LAUNCH(seq_linear_kernel<4>, grid, dim3(256), 0, st, a); break;
  }
  return vunet_check_launch();
}

extern "C" int vunet_seq_coupling(const vunet_seq_coupling_desc* d, const float* in, const float* st, const float* bias_s,
                                  const float* bias_t, const int32_t* map, const float* scale, const float* loc, float* out,
                                  float* logdet, void* stream) {
  if (!d || !in || !out || d->B < 1 || d->C < 1 || d->ld_in < d->C || d->ld_out < d->C) return VUNET_ERR_ARG;
  if (st && (!bias_s || !bias_t || d->S < 1 || d->c1 < 0 || d->c1 > d->C || d->Mp < d->C - d->c1)) return VUNET_ERR_ARG;
  if ((scale == nullptr) != (loc == nullptr)) return VUNET_ERR_ARG;
  if (in == out && map) return VUNET_ERR_ARG;   // a gather cannot run in place
  SeqCouplingArgs a;
  a.in = in;
  a.st = st;
  a.bias_s = bias_s;
  a.bias_t = bias_t;
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
  a.S = d->S;
  a.Mp = d->Mp;
  a.reverse = d->reverse;
  a.affine_on_src = d->affine_on_src;
  VUNET_LAUNCH(seq_coupling_kernel, dim3(d->B), dim3(256), 0, (hipStream_t)stream, a);
  return vunet_check_launch();
}

extern "C" int vunet_seq_start(const float* x0, int64_t x0_stride, const float* h0, const float* c0, const float* w_in,
                               const float* b_in, float* xraw, int32_t ldraw, float* xh, int32_t ldx, int32_t hoff, float* c,
                               int32_t B, int32_t n, int32_t H, void* stream) {
  if (!x0 || !xh || !c || B < 1 || n < 1 || H < 1 || hoff < n || ldx < hoff + H) return VUNET_ERR_ARG;
  if ((w_in == nullptr) != (b_in == nullptr) || (xraw && ldraw < n)) return VUNET_ERR_ARG;
  const size_t lds = (size_t)((n + 63) / 64 * 64) * sizeof(float);
  VUNET_LAUNCH(seq_start_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x0, (long long)x0_stride, h0, c0, w_in, b_in, xraw,
               ldraw, xh, ldx, hoff, c, n, H);
  return vunet_check_launch();
}

extern "C" int vunet_seq_lstm_step(const vunet_seq_lstm_desc* d, const float* gates, const float* bias, const float* bias2, float* c,
                                   float* xh, float* h_out, const float* w_out, const float* b_out, const float* w_in, const float* b_in,
                                   float* xraw, float* xs, float* cs, const float* x_next, void* stream) {
  if (!d || !gates || !bias || !bias2 || !c || !xh || d->B < 1 || d->H < 1 || d->S < 1 || d->n < 1) return VUNET_ERR_ARG;
  if (d->hoff < d->n || d->ldx < d->hoff + d->H) return VUNET_ERR_ARG;
  if (w_out && (!b_out || !xraw || !xs || !cs || d->ldraw < d->n)) return VUNET_ERR_ARG;
  if ((w_in == nullptr) != (b_in == nullptr)) return VUNET_ERR_ARG;
  const size_t lds = (size_t)(d->H + (d->n + 63) / 64 * 64) * sizeof(float);
  if (lds > 160 * 1024) return VUNET_ERR_UNSUPPORTED;
  SeqLstmArgs a;
  a.gates = gates;
  a.bias = bias;
  a.bias2 = bias2;
  a.c = c;
  a.xh = xh;
  a.h_out = h_out;
  a.w_out = w_out;
  a.b_out = b_out;
  a.w_in = w_in;
  a.b_in = b_in;
  a.xraw = xraw;
  a.xs = xs;
  a.cs = cs;
  a.x_next = x_next;
  a.seq_stride = d->seq_stride;
  a.B = d->B;
  a.Bp = (d->B + 15) / 16 * 16;
  a.H = d->H;
  a.S = d->S;
  a.ldx = d->ldx;
  a.hoff = d->hoff;
  a.n = d->n;
  a.ldraw = d->ldraw;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_lstm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  VUNET_LAUNCH(seq_lstm_kernel, dim3(d->B), dim3(256), lds, (hipStream_t)stream, a);
  return vunet_check_launch();
}

extern "C" int vunet_seq_bottleneck(const float* part, int32_t S, int32_t Mp, const float* bias_mu, const float* bias_std,
                                    const float* eps, float* mu, float* logstd, float* b_out, int32_t B, int32_t H, void* stream) {
  if (!part || !bias_mu || !bias_std || !mu || !logstd || S < 1 || B < 1 || H < 1 || Mp < H) return VUNET_ERR_ARG;
  const int Bp = (B + 15) / 16 * 16;
  VUNET_LAUNCH(seq_bottleneck_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, S, Bp, Mp, bias_mu,
               bias_std, eps, mu, logstd, b_out, B, H);
  return vunet_check_launch();
}

extern "C" int vunet_seq_finish(const float* part, int32_t S, int32_t Mp, const float* bias, int32_t act, float* out, int32_t ld_out,
                                int32_t B, int32_t M, void* stream) {
  if (!part || !out || S < 1 || B < 1 || M < 1 || Mp < M || ld_out < M || act < 0 || act > 2) return VUNET_ERR_ARG;
  const int Bp = (B + 15) / 16 * 16;
  VUNET_LAUNCH(seq_finish_kernel, dim3((B * M + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, S, Bp, Mp, bias, act, out, ld_out,
               B, M);
  return vunet_check_launch();
}

extern "C" int vunet_seq_pack_rows(const float* src, int32_t M, int32_t K, const float* row_scale, float* dst, int32_t ld_dst,
                                   int32_t col_off, void* stream) {
  if (!src || !dst || M < 1 || K < 1 || col_off < 0 || ld_dst < col_off + K) return VUNET_ERR_ARG;
  const size_t n = (size_t)M * K;
  VUNET_LAUNCH(seq_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, M, K, row_scale, dst,
               ld_dst, col_off);
  return vunet_check_launch();
}

extern "C" int vunet_seq_normlinear_rows(const float* v, const float* g, const float* bias, const float* gamma, const float* beta,
                                         int32_t M, int32_t K, float* row_scale, float* bias_eff, void* stream) {
  if (!v || !g || !bias || !gamma || !beta || !row_scale || !bias_eff || M < 1 || K < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_normlinear_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, g, bias, gamma, beta, M, K,
               row_scale, bias_eff);
  return vunet_check_launch();
}
