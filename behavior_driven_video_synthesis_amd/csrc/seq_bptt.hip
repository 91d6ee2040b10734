// The behaviour cVAE's training step (BASELINE config 4, first stage; experiments/behavior_net.py:591-660): the pointwise and
// reduction kernels of back-propagation through time over ``ResidualBehaviorNet`` (models/pose_behavior_rnn.py:125-209, :463-534,
// :574-626).  The matrix work of the pass runs on the kernels of csrc/seq_train.hip: per time step one vunet_seq_dx over the
// gate-interleaved image [W_ih | 0 | W_hh] (dgates . W: the gradient wrt the step's operand row [x | 0 | h] as raw slabs, added
// by the next cell kernel), and at the end one vunet_seq_dw per recurrent layer whose reduction runs over all time steps
// (dW = sum_t dgates_t^T [x_t | 0 | h_t]).  Interface: include/vunet_seq_train.h.
#include "common.h"
#include "../../include/vunet_seq_train.h"

namespace {

struct SeqCellBwdArgs {
  const float* hsl;       // [n_sl][Bp][ld_sl] raw slabs of the later step's dgates . W (NULL / first: none)
  const float* w_out;     // [n][H] decoder output layer (NULL: encoder)
  const float* gl;        // d loss / d x' of this step: row b at + b * gl_stride
  const float* gx_next;   // [Bp][64] the later step's total gradient wrt ITS x' (residual path)
  float* gx_out;          // [Bp][64] this step's total gradient wrt x' (the output layer's dZ)
  const float* gates;     // [Bp][H][4] sigmoid(i), sigmoid(f), tanh(g), sigmoid(o)
  const float* c_prev;    // [Bp][H] c, the state the step started from
  const float* c_new;     // [Bp][H] c'
  float* gc;              // [Bp][H] in: d / d c' from the later step; out: d / d c (in place)
  float* dgates;          // [Bp][H][4] = [Bp][4H] in the image's row order
  long long gl_stride;
  int B, Bp, H, n, n_sl, ld_sl, hoff_sl, first;
};

// grid (B, ceil(H / 256)), 256 threads: batch row b, hidden unit j.  The decoder's workgroups first assemble the row's gradient
// wrt x' (n <= 64 values) in LDS -- every workgroup of the row forms the same sums in the same order; the first writes them out.
// Everything a thread will need is REQUESTED before that: its slabs of W_hh^T dgates, its column of the output layer (n <= 64
// values), its gates and cell states -- the kernel was three dependent round trips (x' gradient, barrier, slabs, the 51-term
// product) of 12 us at one launch per time step; the sums keep their order.
__global__ __launch_bounds__(256) void seq_cell_bwd_kernel(SeqCellBwdArgs a) {
  __shared__ float gx[64];
  const int b = blockIdx.x, j = blockIdx.y * 256 + threadIdx.x;
  const int jj = j < a.H ? j : a.H - 1;   // (threads past H load a valid neighbour's values and drop them)
  const size_t slab = (size_t)a.Bp * a.ld_sl;
  const bool later = !(a.first & 1) && a.hsl;
  constexpr int SMAX = 16;                // slabs held in registers at once (vunet_seq_dx splits W's rows into <= 16 ranges)
  float hv[SMAX];
  const float* const hp = later ? a.hsl + (size_t)b * a.ld_sl + a.hoff_sl + jj : nullptr;
#pragma unroll
  for (int s = 0; s < SMAX; ++s) hv[s] = (later && s < a.n_sl) ? hp[s * slab] : 0.f;
  float wv[64];
  if (a.w_out) {
#pragma unroll
    for (int r = 0; r < 64; ++r) wv[r] = r < a.n ? a.w_out[(size_t)r * a.H + jj] : 0.f;
  }
  const size_t e = (size_t)b * a.H + jj;
  const float4 g = *reinterpret_cast<const float4*>(a.gates + e * 4);
  const float cn = a.c_new[e], cp = a.c_prev[e], gc_in = (a.first & 2) ? 0.f : a.gc[e];
  if (a.w_out) {
    if (threadIdx.x < 64) {
      const int r = threadIdx.x;
      float v = 0.f;
      if (r < a.n) {
        v = a.gl[b * a.gl_stride + r];
        if (!(a.first & 1)) {
          v += a.gx_next[(size_t)b * 64 + r];                                                   // x'' = n_out(h') + x'   (:506)
          const float* q = a.hsl + (size_t)b * a.ld_sl + r;                                     // W_ih^T dgates of the later step
          int s = 0;
          for (; s + 8 <= a.n_sl; s += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = q[(s + u) * slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
          }
          for (; s < a.n_sl; ++s) v += q[s * slab];
        }
      }
      gx[r] = v;
      if (blockIdx.y == 0) a.gx_out[(size_t)b * 64 + r] = v;
    }
    __syncthreads();
  }
  if (j >= a.H) return;
  float dh = 0.f;
  if (later) {   // W_hh^T dgates of the later step: slabs in slab order
#pragma unroll
    for (int s = 0; s < SMAX; ++s)
      if (s < a.n_sl) dh += hv[s];
    for (int s = SMAX; s < a.n_sl; ++s) dh += hp[s * slab];
  }
  if (a.w_out) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 64; ++r)
      if (r < a.n) acc += wv[r] * gx[r];   // out = n_out(h)   (:504)
    dh += acc;
  }
  // c' = f c + i g,  h = o tanh(c')   (torch.nn.LSTMCell, models/pose_behavior_rnn.py:476, :498)
  const float tc = tanhf(cn);
  const float dc = gc_in + dh * g.w * (1.f - tc * tc);
  a.gc[e] = dc * g.y;
  float4 d;
  d.x = dc * g.z * g.x * (1.f - g.x);
  d.y = dc * cp * g.y * (1.f - g.y);
  d.z = dc * g.x * (1.f - g.z * g.z);
  d.w = dh * tc * g.w * (1.f - g.w);
  *reinterpret_cast<float4*>(a.dgates + e * 4) = d;
}

// b = eps exp(logstd) + mu is both the decoder's first hidden and first cell state; one thread per (row, unit)
__global__ __launch_bounds__(256) void seq_bottleneck_bwd_kernel(const float* __restrict__ hsl, int n_sl, int ld_sl, int hoff_sl, int Bp,
                                                                 const float* __restrict__ gc, const float* __restrict__ eps,
                                                                 const float* __restrict__ logstd, const float* __restrict__ dmu,
                                                                 const float* __restrict__ dlogstd, float* __restrict__ dy, int B,
                                                                 int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const size_t slab = (size_t)Bp * ld_sl;
  float gsum = 0.f;
  const float* p = hsl + (size_t)b * ld_sl + hoff_sl + j;
  for (int s = 0; s < n_sl; ++s) gsum += p[s * slab];
  gsum += gc[(size_t)b * H + j];
  const size_t o = (size_t)b * H + j;
  dy[o] = (dmu ? dmu[idx] : 0.f) + gsum;
  dy[(size_t)Bp * H + o] = (dlogstd ? dlogstd[idx] : 0.f) + (eps ? gsum * eps[idx] * expf(logstd[idx]) : 0.f);
}

// blocks 0 .. T-1: time step t: sum over (b, d) of (xs - target)^2 -> part[t], and d loss / d xs
// blocks T .. T+B-1: batch row b of kl_loss -> part[T + b], and d loss / d mu, d loss / d logstd
__global__ __launch_bounds__(256) void seq_vae_loss_parts_kernel(const float* __restrict__ xs, const float* __restrict__ tgt,
                                                                 const float* __restrict__ mu, const float* __restrict__ logstd, int B,
                                                                 int T, int n, int H, float recon_weight, const float* __restrict__ gamma,
                                                                 float* __restrict__ part, float* __restrict__ dxs, float* __restrict__ dmu,
                                                                 float* __restrict__ dlogstd) {
  __shared__ float red[4];
  float s = 0.f;
  if ((int)blockIdx.x < T) {
    const int t = blockIdx.x;
    const float k = recon_weight * 2.f / ((float)B * (float)T * (float)n);
    for (int idx = threadIdx.x; idx < B * n; idx += 256) {
      const int b = idx / n, d = idx - b * n;
      const size_t o = ((size_t)b * T + t) * n + d;
      const float e = xs[o] - tgt[o];
      s += e * e;
      if (dxs) dxs[o] = k * e;
    }
  } else {
    const int b = blockIdx.x - T;
    const float gm = *gamma / (float)B;
    float sm = 0.f, sl = 0.f;
    for (int j = threadIdx.x; j < H; j += 256) {
      const size_t o = (size_t)b * H + j;
      const float l = logstd[o], m = mu[o], sd = expf(l);
      s += -l + 0.5f * (sd * sd + m * m);          // lib/losses.py:288-289
      sm += m;
      sl += l;
      if (dmu) dmu[o] = gm * m;
      if (dlogstd) dlogstd[o] = gm * (sd * sd - 1.f);
    }
    // the row's share of the logged means of mu and logstd (experiments/behavior_net.py:717-718)
    __shared__ float red2[2][4];
    sm = wave_sum(sm);
    sl = wave_sum(sl);
    if ((threadIdx.x & 63) == 0) {
      red2[0][threadIdx.x >> 6] = sm;
      red2[1][threadIdx.x >> 6] = sl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      part[T + B + b] = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
      part[T + 2 * B + b] = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one wave: the partial sums in index order -> the scalars, per-step errors, and gamma for the next step
__global__ __launch_bounds__(64) void seq_vae_loss_final_kernel(const float* __restrict__ part, int B, int T, int n, int H, float recon_weight,
                                                                float* __restrict__ gamma, const float* __restrict__ imax, float gamma_step,
                                                                float* __restrict__ scalars, float* __restrict__ per_seq) {
  const int lane = threadIdx.x;
  float r = 0.f, k = 0.f;
  for (int t = lane; t < T; t += 64) {
    r += part[t];
    if (per_seq) per_seq[t] = part[t] / ((float)B * (float)n);
  }
  float sm = 0.f, sl = 0.f;
  for (int b = lane; b < B; b += 64) {
    k += part[T + b] - 0.5f * (float)H;   // ... - 0.5 dim   (lib/losses.py:289)
    sm += part[T + B + b];
    sl += part[T + 2 * B + b];
  }
  r = wave_sum(r);
  k = wave_sum(k);
  sm = wave_sum(sm);
  sl = wave_sum(sl);
  if (lane == 0) {
    scalars[4] = sm / ((float)B * (float)H);
    scalars[5] = sl / ((float)B * (float)H);
    const float recon = r / ((float)B * (float)T * (float)n), kl = k / (float)B, g = *gamma;
    scalars[0] = recon_weight * recon + g * kl;
    scalars[1] = recon;
    scalars[2] = kl;
    scalars[3] = g;
    if (gamma_step > 0.f) *gamma = fmaxf(g - gamma_step * (*imax - kl), 0.f);   // experiments/behavior_net.py:111-116
  }
}

// one wave per row m of W_eff = gamma g v / ||v||
__global__ __launch_bounds__(256) void seq_normlinear_bwd_kernel(const float* __restrict__ dweff, const float* __restrict__ dbeff,
                                                                 const float* __restrict__ v, const float* __restrict__ g,
                                                                 const float* __restrict__ bias, const float* __restrict__ gamma, int M,
                                                                 int K, float* __restrict__ dv, float* __restrict__ dg,
                                                                 float* __restrict__ dbias, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  float nn = 0.f, dot = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float t = v[(size_t)m * K + k];
    nn += t * t;
    dot += t * dweff[(size_t)m * K + k];
  }
  nn = wave_sum(nn);
  dot = wave_sum(dot);
  const float inv = 1.f / sqrtf(nn);
  const float proj = dot * inv;                      // dW_eff . v / ||v||
  const float rs = gamma[m] * g[m] * inv;
  for (int k = lane; k < K; k += 64)
    dv[(size_t)m * K + k] = rs * (dweff[(size_t)m * K + k] - proj * inv * v[(size_t)m * K + k]);
  if (lane == 0) {
    dg[m] = gamma[m] * proj;
    dgamma[m] = g[m] * proj + bias[m] * dbeff[m];
    dbias[m] = gamma[m] * dbeff[m];
    dbeta[m] = dbeff[m];
  }
}

// image row 4 j + q -> torch row q H + j; one thread per element of [4H][n + H + 1] (the last column: the bias)
__global__ __launch_bounds__(256) void seq_lstm_grads_unpack_kernel(const float* __restrict__ gimg, const float* __restrict__ gbias,
                                                                    int ldx, int hoff, int n, int H, float* __restrict__ dwih,
                                                                    float* __restrict__ dwhh, float* __restrict__ dbih,
                                                                    float* __restrict__ dbhh) {
  const int w = n + H + 1;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)4 * H * w) return;
  const int row = (int)(idx / w), col = (int)(idx - (size_t)row * w);   // torch row
  const int q = row / H, j = row - q * H, irow = 4 * j + q;
  if (col < n) dwih[(size_t)row * n + col] = gimg[(size_t)irow * ldx + col];
  else if (col < n + H) dwhh[(size_t)row * H + (col - n)] = gimg[(size_t)irow * ldx + hoff + (col - n)];
  else {
    const float gb = gbias[irow];
    dbih[row] = gb;
    dbhh[row] = gb;
  }
}

}  // namespace

extern "C" int vunet_seq_cell_bwd(const vunet_seq_cell_bwd_desc* d, const float* hsl, const float* w_out, const float* gl,
                                  const float* gx_next, float* gx_out, const float* gates, const float* c_prev, const float* c_new,
                                  float* gc, float* dgates, void* stream) {
  if (!d || !gates || !c_prev || !c_new || !gc || !dgates || d->B < 1 || d->B > 64 || d->H < 1) return VUNET_ERR_ARG;
  if (w_out && (!gl || !gx_out || d->n < 1 || d->n > 64)) return VUNET_ERR_ARG;
  if (!(d->first & 1) && ((hsl && (d->n_sl < 1 || d->ld_sl < d->hoff_sl + d->H)) || (w_out && (!gx_next || !hsl)))) return VUNET_ERR_ARG;
  SeqCellBwdArgs a;
  a.hsl = hsl;
  a.w_out = w_out;
  a.gl = gl;
  a.gx_next = gx_next;
  a.gx_out = gx_out;
  a.gates = gates;
  a.c_prev = c_prev;
  a.c_new = c_new;
  a.gc = gc;
  a.dgates = dgates;
  a.gl_stride = d->gl_stride;
  a.B = d->B;
  a.Bp = (d->B + 15) / 16 * 16;
  a.H = d->H;
  a.n = d->n;
  a.n_sl = d->n_sl;
  a.ld_sl = d->ld_sl;
  a.hoff_sl = d->hoff_sl;
  a.first = d->first;
  VUNET_LAUNCH(seq_cell_bwd_kernel, dim3(d->B, (d->H + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return vunet_check_launch();
}

extern "C" int vunet_seq_bottleneck_bwd(const float* hsl, int32_t n_sl, int32_t ld_sl, int32_t hoff_sl, const float* gc, const float* eps,
                                        const float* logstd, const float* dmu, const float* dlogstd, float* dy, int32_t B, int32_t H,
                                        void* stream) {
  if (!hsl || !gc || !dy || n_sl < 1 || ld_sl < hoff_sl + H || B < 1 || B > 64 || H < 1 || (eps && !logstd)) return VUNET_ERR_ARG;
  const int Bp = (B + 15) / 16 * 16;
  VUNET_LAUNCH(seq_bottleneck_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, hsl, n_sl, ld_sl, hoff_sl, Bp, gc,
               eps, logstd, dmu, dlogstd, dy, B, H);
  return vunet_check_launch();
}

extern "C" int vunet_seq_vae_loss(const float* xs, const float* target, const float* mu, const float* logstd, int32_t B, int32_t T,
                                  int32_t n, int32_t H, float recon_weight, float* gamma_dev, const float* imax_dev, float gamma_step,
                                  float* part, float* scalars, float* per_seq, float* dxs, float* dmu, float* dlogstd, void* stream) {
  if (!xs || !target || !mu || !logstd || !gamma_dev || !part || !scalars || B < 1 || B > 64 || T < 1 || n < 1 || H < 1) return VUNET_ERR_ARG;
  if (gamma_step > 0.f && !imax_dev) return VUNET_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  VUNET_LAUNCH(seq_vae_loss_parts_kernel, dim3(T + B), dim3(256), 0, st, xs, target, mu, logstd, B, T, n, H, recon_weight, gamma_dev, part,
               dxs, dmu, dlogstd);
  VUNET_LAUNCH(seq_vae_loss_final_kernel, dim3(1), dim3(64), 0, st, part, B, T, n, H, recon_weight, gamma_dev, imax_dev, gamma_step, scalars,
               per_seq);
  return vunet_check_launch();
}

extern "C" int vunet_seq_normlinear_bwd(const float* dweff, const float* dbeff, const float* v, const float* g, const float* bias,
                                        const float* gamma, int32_t M, int32_t K, float* dv, float* dg, float* dbias, float* dgamma,
                                        float* dbeta, void* stream) {
  if (!dweff || !dbeff || !v || !g || !bias || !gamma || !dv || !dg || !dbias || !dgamma || !dbeta || M < 1 || K < 1) return VUNET_ERR_ARG;
  VUNET_LAUNCH(seq_normlinear_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dweff, dbeff, v, g, bias, gamma, M, K, dv,
               dg, dbias, dgamma, dbeta);
  return vunet_check_launch();
}

extern "C" int vunet_seq_lstm_grads_unpack(const float* gimg, const float* gbias, int32_t ldx, int32_t hoff, int32_t n, int32_t H,
                                           float* dwih, float* dwhh, float* dbih, float* dbhh, void* stream) {
  if (!gimg || !gbias || !dwih || !dwhh || !dbih || !dbhh || n < 1 || H < 1 || hoff < n || ldx < hoff + H) return VUNET_ERR_ARG;
  const size_t total = (size_t)4 * H * (n + H + 1);
  VUNET_LAUNCH(seq_lstm_grads_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gimg, gbias, ldx,
               hoff, n, H, dwih, dwhh, dbih, dbhh);
  return vunet_check_launch();
}
