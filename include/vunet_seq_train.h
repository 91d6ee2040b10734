/*
 * vunet_seq_train.h -- C ABI of the TRAINING half of the behaviour path (BASELINE config 4): the normalizing flow's
 * maximum-likelihood step and the behaviour cVAE's step, csrc/seq_train.hip.  Part of libvunet_hip.so; same conventions as
 * vunet_hip.h (plain pointers and sizes, caller-owned fp32 device buffers, `stream` = hipStream_t as void*, 0 or a negative
 * VUNET_ERR_* code).
 *
 * What it replaces in the reference (CompVis/behavior-driven-video-synthesis, paths relative to the upstream root):
 *   experiments/behavior_net.py:703-714   gauss, logdet = latent_flow(bs.detach()); f_loss = flow_loss(gauss, logdet);
 *                                         flow_optimizer.zero_grad(); f_loss.backward(); flow_optimizer.step()
 *   experiments/behavior_net.py:384-395   Adam(latent_flow.parameters(), lr = flow_lr * batch_size, betas = (0.5, 0.9), weight_decay)
 *   lib/losses.py:294-331                 FlowLoss / nll
 *   experiments/behavior_net.py:591-660   the cVAE step: net(seq, seq, len) -> recon (MSE) + gamma * kl_loss -> backward -> Adam
 *   lib/losses.py:283-291                 kl_loss
 * i.e. what autograd derives for models/flow/blocks.py:95-128, :276-319, :531-559, :692-704, lib/modules.py:236-331 and
 * models/pose_behavior_rnn.py:125-209, :463-534, :574-626, plus torch.optim.Adam's update.
 *
 * Layout of a flow training step (B <= 64 rows, Bp = B rounded up to 16; all matrices row-major fp32):
 *   forward     vunet_seq_linear / vunet_seq_coupling (vunet_hip.h) into per-layer buffers that stay alive (the saved activations)
 *   loss        vunet_seq_flow_loss          nll + (-logdet) means, d loss / d z, d loss / d logdet
 *   backward    vunet_seq_coupling_bwd       everything between two MLP evaluations, backwards: the incoming gradient (+ the raw
 *                                            input-gradient slabs of the MLP evaluated after it), un-shuffle, ActNorm, affine
 *                                            coupling -> gradient wrt the step's input rows and the head layers' dZ
 *               vunet_seq_dx                 dX = dZ . W of one MLP layer (both nets of a coupling in one launch) as RAW partial
 *                                            slabs over S row ranges of W -- a workgroup owns a 64-column stripe of W over M / S rows
 *               vunet_seq_dz_finish          dZ_prev = (sum of the slabs) * LeakyReLU'(Y_prev)
 *   update      vunet_seq_dw                 dW = dZ^T . X per 64 x 64 tile of W on the fp32 matrix cores, and -- in the same
 *                                            tile, while it is in registers -- torch.optim.Adam's update of W, exp_avg,
 *                                            exp_avg_sq (or, hp == NULL, the gradient written out for autograd); bias likewise
 *               vunet_seq_actnorm_bwd        ActNorm's loc / scale gradients (a column reduction over the batch) + Adam
 *               vunet_seq_adam_tick          ++step (device-resident, so a captured hipGraph replays the step unchanged)
 */
#ifndef VUNET_SEQ_TRAIN_H
#define VUNET_SEQ_TRAIN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* torch.optim.Adam's hyper-parameters; the learning rate and the step count live in DEVICE memory (nothing that changes
 * between steps is a launch argument).  The update follows torch/optim/adam.py (_single_tensor_adam, amsgrad False,
 * maximize False): g += weight_decay * p; m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g g;
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps). */
typedef struct vunet_seq_adam_hp {
  const double* lr_dev;
  const int64_t* step_dev;   /* t of the update being applied (>= 1) */
  float beta1, beta2, eps, weight_decay;
  const float* resolved_dev; /* NULL, or {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)} as vunet_seq_adam_tick left them for THIS t:
                                the kernels then read two floats instead of forming the powers themselves */
} vunet_seq_adam_hp;

/* dX = dZ . W: w_n [M][ldw] (nets <= 2; columns 0 .. K of each row are used: ldw = 0 means K), dz [nets][Bp][M],
 * raw [nets][S][Bp][K]: slab s is the share of W's 16-row groups [s G / S, (s + 1) G / S), G = M / 16 (equal ranges where S
 * divides G).  K % 64 == 0, M % 16 == 0, S <= M / 16, ldw % 4 == 0, B <= 64. */
typedef struct vunet_seq_dx_desc {
  int32_t B, M, K, nets, S, ldw;
} vunet_seq_dx_desc;
int vunet_seq_dx(const vunet_seq_dx_desc* d, const float* w0, const float* w1, const float* dz, float* raw, void* stream);

/* dz[n][b][k] = (sum_s raw[n][s][b][k]) * (y[n][b][k] > 0 ? 1 : slope): the derivative of nn.LeakyReLU (lib/modules.py:244)
 * read off the saved OUTPUT (slope > 0 keeps the sign).  All Bp rows. */
int vunet_seq_dz_finish(const float* raw, const float* y, float* dz, int32_t nets, int32_t S, int32_t Bp, int32_t K, float slope,
                        void* stream);

/* vunet_seq_dx and vunet_seq_dz_finish of its slabs in ONE launch (the hidden layers of an MLP: a dependent launch less per layer).
 * The S workgroups of a 64-column stripe count their arrivals in counters[n][K / 64] (int32, zero before the first use and left
 * zero); the one that arrives last adds the stripe's S slabs in slab order -- the sum does not depend on which one it is -- and
 * writes dz_prev[n][b][k] for the stripe.  y, dz_prev: [nets][Bp][K].  Same values as the two launches.  (Measured slower than
 * the two launches inside a training step -- the device-scope fences around the counter flush every L2: DESIGN.md 5.R6 -- and not
 * used by default.) */
int vunet_seq_dx_finish(const vunet_seq_dx_desc* d, const float* w0, const float* w1, const float* dz, float* raw, const float* y,
                        float* dz_prev, int32_t* counters, float slope, void* stream);

/* One step of the flow backwards (the mirror of vunet_seq_coupling, forward direction).  The forward step was
 *   v = couple(in);  out[c] = A(v[map[c]])   with couple = x_k exp(s) + t on columns >= c1, A = ActNorm of the NEXT block.
 * g[c]      = gbase[b][c] + (c < c1s ? sum over n_sl slabs gslabs[n][b][c] : 0)      the gradient wrt out (complete)
 * gfull     (optional) receives g: the ActNorm gradient kernel's input
 * dv[j]     = g[inv_map[j]] * scale[inv_map[j]]
 * gout[b][j] = dv[j] (j < c1)  |  dv[j] exp(s)  (j >= c1)
 * dzh       [2][Bp][Mp]: the head layers' dZ -- scale net: (dv x_k exp(s) + dld[b]) (1 - s^2), translation net: dv
 * st / bias_s / bias_t / S / Mp as vunet_seq_coupling took them in the forward pass (st == NULL: no coupling in this step). */
typedef struct vunet_seq_coupling_bwd_desc {
  int32_t B, C, c1, ld_g, ld_in, ld_out, ld_full, Mp, S, n_sl, ld_sl, c1s;
} vunet_seq_coupling_bwd_desc;
int vunet_seq_coupling_bwd(const vunet_seq_coupling_bwd_desc* d, const float* gbase, const float* gslabs, const int32_t* inv_map,
                           const float* scale, const float* in, const float* st, const float* bias_s, const float* bias_t,
                           const float* dld, float* gfull, float* gout, float* dzh, void* stream);

/* One Linear layer of the update sweep.  A launch covers a list of layers: `tile0` is the layer's first 64 x 64 tile in the
 * launch's flat tile list, `tiles_k` = K / 64.  M % 64 == 0, K % 64 == 0 (zero-padded images; a parameter whose shape fits is
 * used in place).  g / bg: gradient outputs of the write mode (hp == NULL); m, v / bm, bv: Adam's moments. */
typedef struct vunet_seq_dw_layer {
  float* w; float* m; float* v; float* g;
  float* bias; float* bm; float* bv; float* bg;
  const float* dz;   /* [Bp][ldz]: gradient wrt the layer's pre-activation output; rows >= B are zero */
  const float* x;    /* [Bp][ldx]: the layer's input rows */
  int32_t M, K, ldz, ldx, tile0, tiles_k;
  int32_t kv;        /* valid columns: the gradient of columns >= kv is forced to zero (an MLP's first layer reads its input from
                        rows that continue with other data beyond its c1 columns; its padding columns must stay zero) */
  int32_t nchunk;    /* the reduction runs over nchunk blocks of Bp rows (a recurrent layer: one block per time step):
                        block i at dz + i chunk_z, x + i chunk_x (floats); 0 or 1: a single block */
  int32_t chunk_z, chunk_x;
  float* dx_raw;     /* vunet_seq_dwx only: [ceil(M / (64 nchunk))][Bp][K] raw slabs of dX = dZ . W (NULL: not wanted) */
} vunet_seq_dw_layer;
int vunet_seq_dw(const vunet_seq_dw_layer* table_dev, int32_t n_layers, int32_t first_tile, int32_t n_tiles, int32_t B,
                 const vunet_seq_adam_hp* hp, void* stream);
/* The same update AND the layer's input gradient in ONE pass over W: a workgroup owns a (64 nchunk)-row x 64-column tile (nchunk
 * = 1, 2 or 4 here: the tile's rows / 64, not a row-block count; tile0 / tiles_k count THOSE tiles: ceil(M / (64 nchunk)) x K / 64
 * per layer; the last row tile may be short), takes it through its 64 x 64 sub-tiles as vunet_seq_dw does, and adds each sub-tile's share of dX = dZ . W_old (the values it loaded, before the update) for
 * its 64 columns: dx_raw[tm][b][k], raw slabs over the row tiles, added in slab order by whoever reads them (vunet_seq_dz_finish,
 * vunet_seq_coupling_bwd).
 * No separate vunet_seq_dx pass over W.  A single block of Bp rows (no reduction over time steps here). */
int vunet_seq_dwx(const vunet_seq_dw_layer* table_dev, int32_t n_layers, int32_t first_tile, int32_t n_tiles, int32_t B,
                  const vunet_seq_adam_hp* hp, void* stream);

/* ActNorm (lib/modules.py:260-331) of every block in one launch: out = scale (u + loc), logdet += sum log|scale|.
 * d scale[c] = (sum_b gfull[b][c] out[b][c] + sum_b dld[b]) / scale[c];  d loc[c] = scale[c] sum_b gfull[b][c]. */
typedef struct vunet_seq_actnorm_layer {
  float* scale; float* loc;
  float* sm; float* sv; float* lm; float* lv;   /* Adam moments of scale / loc */
  float* gs; float* gl;                         /* write mode: gradients */
  const float* gfull; const float* out;         /* [B][ld] */
  int32_t ld, pad;
} vunet_seq_actnorm_layer;
int vunet_seq_actnorm_bwd(const vunet_seq_actnorm_layer* table_dev, int32_t n_layers, int32_t C, int32_t B, const float* dld,
                          const vunet_seq_adam_hp* hp, void* stream);

/* FlowLoss (lib/losses.py:294-317): nll = mean_b 0.5 sum_c z^2, nlogdet = -mean_b logdet, loss = nll + nlogdet,
 * reference_nll = mean_b 0.5 sum_c noise^2 (noise NULL: 0).  scalars[4] = loss, reference_nll, nlogdet, nll.
 * dz[b][c] = z[b][c] / B (rows B..Bp zeroed), dld[b] = -1 / B. */
int vunet_seq_flow_loss(const float* z, int32_t ldz, const float* logdet, const float* noise, int32_t B, int32_t C, float* scalars,
                        float* dz, int32_t ld_dz, float* dld, void* stream);

/* ++*step_dev; resolved_dev != NULL: also the step's constants {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)} (double arithmetic) */
int vunet_seq_adam_tick(int64_t* step_dev, const double* lr_dev, float beta1, float beta2, float* resolved_dev, void* stream);

/* dst[m][k] = src[(row_off + row_mul m) ld_src + col_off + k] (the inverse of vunet_seq_pack_rows; `accumulate`: +=) */
int vunet_seq_unpack_rows(const float* src, int32_t ld_src, int32_t col_off, int32_t row_off, int32_t row_mul, float* dst, int32_t M,
                          int32_t K, int32_t accumulate, void* stream);

/* ---- the behaviour cVAE's step (experiments/behavior_net.py:591-660; models/pose_behavior_rnn.py:125-209, :463-534, :574-626):
 * back-propagation through time over the decoder's roll-out and the encoder's LSTM.
 *
 *   vunet_seq_lstm_gates_train   vunet_seq_lstm_gates (vunet_hip.h) that also keeps the step's gate activations:
 *                                gates_out [Bp][H][4] = sigmoid(i), sigmoid(f), tanh(g), sigmoid(o)
 *   vunet_seq_cell_bwd           one LSTM step backwards, pointwise part.  For batch row b, unit j:
 *                                  dh = sum_s hsl[s][b][hoff_sl + j]                        (the later step's W_hh^T dgates, raw slabs of vunet_seq_dx)
 *                                     + sum_r w_out[r][j] gx[b][r]                          (decoder: x' = W_out h + b_out + x)
 *                                  dc = gc[b][j] + dh o (1 - tanh(c')^2);  gc[b][j] <- dc f  (in place)
 *                                  `first`: bit 0 -- no later step: hsl / gx_next are not read; bit 1 -- gc is taken as 0
 *                                  dgates[b][j][0..4) = dc g i(1-i), dc c f(1-f), dc i (1-g^2), dh tanh(c') o(1-o)
 *                                  (c = c_prev: the state the step started from, c' = c_new; gates = the kept activations)
 *                                decoder (w_out != NULL): gx[b][r] = gl[b][r] (d loss / d x' of this step, row at + b gl_stride)
 *                                  + gx_next[b][r] + sum_s hsl[s][b][r] (the later step's residual and W_ih^T dgates; absent with bit 0 of `first`),
 *                                  also written to gx_out [Bp][64] (the output layer's dZ).
 *   vunet_seq_bottleneck_bwd     b = eps exp(logstd) + mu (models/pose_behavior_rnn.py:203-206) with h0 = c0 = b (:612-614):
 *                                  G = sum_s hsl[s][b][hoff_sl + j] + gc[b][j];  dy[0] = dmu + G;  dy[1] = dlogstd + G eps exp(logstd)
 *   vunet_seq_vae_loss           recon = mean (xs - target)^2 (nn.MSELoss(reduction none) + mean, :358, :134-149), kl = kl_loss(mu, logstd)
 *                                (lib/losses.py:283-291), loss = w recon + gamma kl (:606-611); gamma read from the device and -- `gamma_step`
 *                                > 0 -- advanced for the NEXT step as __update_gamma does after the optimiser step (:111-116, :655):
 *                                gamma <- max(gamma - gamma_step (imax - kl), 0).  scalars[6] = loss, recon, kl, gamma used, mean(mu),
 *                                mean(logstd) (the logged mu_s / logstd_s, :717-718); per_seq[t] =
 *                                mean over batch and dims of step t's squared error.  dxs = w 2 (xs - target) / (B T n);
 *                                dmu = gamma mu / B; dlogstd = gamma (exp(2 logstd) - 1) / B.   `part`: scratch of T + 3 B floats.
 *   vunet_seq_normlinear_bwd     NormConv2d 1x1 as a linear layer, W_eff = gamma g v / ||v||, b_eff = gamma bias + beta: from dW_eff [M][K]
 *                                and db_eff [M] the gradients of v [M][K], g, bias, gamma, beta [M]
 *   vunet_seq_lstm_grads_unpack  the gate-interleaved image gradient [4H][ldx] = [dW_ih | . | dW_hh] and bias gradient [4H] -> torch's
 *                                layout: dW_ih [4H][n], dW_hh [4H][H], db_ih = db_hh [4H] (row q H + j <- image row 4 j + q) */
typedef struct vunet_seq_cell_bwd_desc {
  int32_t B, H, n, n_sl, ld_sl, hoff_sl, first, pad;
  int64_t gl_stride;
} vunet_seq_cell_bwd_desc;
int vunet_seq_lstm_gates_train(const vunet_seq_lstm_desc* d, const float* w_perm, const float* xh, const float* bias_perm, const float* c_in,
                               float* c_out, float* xh_next, float* h_out, const float* x_next, float* gates_out, void* stream);
int vunet_seq_cell_bwd(const vunet_seq_cell_bwd_desc* d, const float* hsl, const float* w_out, const float* gl, const float* gx_next,
                       float* gx_out, const float* gates, const float* c_prev, const float* c_new, float* gc, float* dgates,
                       void* stream);
int vunet_seq_bottleneck_bwd(const float* hsl, int32_t n_sl, int32_t ld_sl, int32_t hoff_sl, const float* gc, const float* eps,
                             const float* logstd, const float* dmu, const float* dlogstd, float* dy, int32_t B, int32_t H, void* stream);
int vunet_seq_vae_loss(const float* xs, const float* target, const float* mu, const float* logstd, int32_t B, int32_t T, int32_t n,
                       int32_t H, float recon_weight, float* gamma_dev, const float* imax_dev, float gamma_step, float* part,
                       float* scalars, float* per_seq, float* dxs, float* dmu, float* dlogstd, void* stream);
int vunet_seq_normlinear_bwd(const float* dweff, const float* dbeff, const float* v, const float* g, const float* bias, const float* gamma,
                             int32_t M, int32_t K, float* dv, float* dg, float* dbias, float* dgamma, float* dbeta, void* stream);
int vunet_seq_lstm_grads_unpack(const float* gimg, const float* gbias, int32_t ldx, int32_t hoff, int32_t n, int32_t H, float* dwih,
                                float* dwhh, float* dbih, float* dbhh, void* stream);

#ifdef __cplusplus
}
#endif
#endif
