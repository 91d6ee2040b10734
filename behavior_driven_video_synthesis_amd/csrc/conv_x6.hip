// Host-side dispatch of the fp32-accurate split-bf16 3x3 / stride-1 convolution (kernel: conv_x6_kernel.h) and the
// unified convolution entry point vunet_conv2d, which picks between it and the fp32-MFMA / VALU kernels of
// vunet_conv2d_gather.
#include "conv_common.h"

int vunet_conv_x6_launch_mt1(const GatherArgs& ga, const void* wx, int mtiles_pad, int pro, int NT, hipStream_t st);
int vunet_conv_x6_launch_mt2(const GatherArgs& ga, const void* wx, int mtiles_pad, int pro, int NT, hipStream_t st);
int vunet_conv_h2_launch_mt1(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, int NT,
                             hipStream_t st);
int vunet_conv_h2_launch_mt2(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, int NT,
                             hipStream_t st);
// conv_h2_s2.hip: forward of the stride-2 layers (fp16 scheme)
bool vunet_conv_h2_s2_ok(const vunet_conv_desc* d, int pro);
int vunet_conv_h2_s2_name(const vunet_conv_desc* d, char* name, int len);
int vunet_conv_h2_s2_launch(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, hipStream_t st);
int vunet_conv_h2_launch_par(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, hipStream_t st);
int vunet_conv_h2_launch_w16(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, int MT,
                             hipStream_t st);

// conv_h2_small.hip: K-split workgroups for the small maps (fp16 scheme)
bool vunet_conv_h2_small_ok(const vunet_conv_desc* d, int pro);
bool vunet_conv_h2_small_wanted(const vunet_conv_desc* d, int pro);
int vunet_conv_h2_small_launch(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro,
                               hipStream_t st);
int vunet_conv_h2_small_name(const vunet_conv_desc* d, int pro, char* name, int len);

extern "C" int vunet_x6_mtiles(int32_t M);
int vunet_conv_thin_kind(const vunet_conv_desc* d, int pro, bool has_aux, bool has_res);

// VUNET_CONV_PRECISION=f32 keeps every layer on the fp32-input MFMA kernels (read once per process)
static bool x6_enabled() {
  static const bool on = [] {
    const char* e = getenv("VUNET_CONV_PRECISION");
    return !(e && (e[0] == 'f' || e[0] == 'F'));
  }();
  return on;
}

static int x6_prologue_code(const vunet_conv_desc* d, bool has_mask) {
  if (has_mask) return (d->in_act == ACT_NONE && d->drop_p <= 0.f) ? 4 : 3;
  if (d->in_act == ACT_NONE && d->drop_p <= 0.f) return 0;
  if (d->in_act == ACT_ELU && d->drop_p <= 0.f) return 1;
  if (d->in_act == ACT_ELU) return 2;
  return 3;
}

// 16-wide staged maps (fp16 scheme only): column tiles of two rows x 16 pixels, 8-row workgroup tiles, stride 1
static bool h2_w16(const vunet_conv_desc* d) { return d->Ws % 32 != 0 && d->Ws % 16 == 0 && d->Hs % 8 == 0 && d->stride == 1; }

// geometry the kernel family covers (nothing about whether it is the fastest choice)
static bool x6_geometry_ok(const vunet_conv_desc* d, int pro, bool h2 = false) {
  if (d->KH != 3 || d->KW != 3 || d->pad != 1) return false;
  if (d->C1 <= 0 || d->C1 % 16 || d->C2 % 16 || d->Hs % 4) return false;   // Hs, Ws: the gathered (staged) map
  if (d->Ws % 32 && !(h2 && h2_w16(d))) return false;
  if (d->M % 32 || d->m_off % 32) return false;
  if (d->stride == 2)   // data gradient of the stride-2 Downsample conv: four parity launches over the dy map
    return d->mode == 1 && pro == 0 && d->C2 == 0 && d->Ho == 2 * d->Hs && d->Wo == 2 * d->Ws;
  if (d->stride != 1 || d->Hs != d->Ho || d->Ws != d->Wo) return false;
  return d->mode == 0 ? (pro == 0 || pro == 1 || pro == 2) : (pro == 0 || pro == 4);
}

static long x6_blocks(const vunet_conv_desc* d, int MT, int NT) {
  if (d->Ws % 32) return NT == 1 ? (long)d->N * (d->Hs / 8) * (d->Ws / 16) * ((d->M + 32 * MT - 1) / (32 * MT)) : 0;  // h2_w16
  if (d->Hs % (4 * NT)) return 0;
  return (long)d->N * (d->Hs / (4 * NT)) * (d->Ws / 32) * ((d->M + 32 * MT - 1) / (32 * MT));
}

// Tile: M <= 32 -> one m-tile and up to 16 rows (the weight slab is re-staged per workgroup: tall tiles amortise it);
// wider layers -> two m-tiles and 8 rows (LDS: two workgroups per CU).  The tallest tile that still gives every CU its
// two resident workgroups wins; `min_blocks` is what the caller requires of the smallest tile.
static int x6_pick(const vunet_conv_desc* d, int* MT, long min_blocks, bool h2 = false) {
  *MT = d->M <= 32 ? 1 : 2;
  // bf16 scheme: up to 16 rows for one m-tile.  fp16 scheme (two accumulator sets): 8 rows for two m-tiles; one m-tile
  // runs best on 4-row tiles -- 113 VGPRs and 38 KB of LDS let FOUR workgroups share a CU, which hides more of the short-K
  // layers' load / store phases (32 channels at 256^2: 120 -> 112 us, r02)
  const int top = h2 ? (*MT == 1 ? 1 : 2) : (*MT == 1 ? 4 : 2);
  if (int NT = g_vunet_tune[VUNET_TUNE_SPLIT_FORCE_NT]) {  // tests / tuning (vunet_set_tuning)
    if (NT > top) NT = (h2 && *MT == 1 && NT == 2) ? 2 : top;   // (h2, one m-tile: the 8-row form is built, for A/B runs)
    if ((NT == 1 || NT == 2 || NT == 4) && x6_blocks(d, *MT, NT) > 0) return NT;
  }
  // (r03: one m-tile per workgroup for the 16-wide maps -- VGG19 conv5_x at bs 16 is 256 workgroups of two m-tiles, half of
  //  the residency slots empty -- measured no faster forward (92 us either way) and 6 % slower data gradient: a chunk's time
  //  there is its staging, the same for one m-tile as for two)
  for (int NT = top; NT >= 2; NT >>= 1)
    if (x6_blocks(d, *MT, NT) >= 512) return NT;
  // (r03: one m-tile per workgroup for the 128-channel 32^2 layers -- 512 workgroups instead of 256 -- measured no
  //  different: 27.7 / 26.2 us either way; these launches are latency-bound, not short of parallelism)
  return x6_blocks(d, *MT, 1) >= min_blocks ? 1 : 0;
}

static void fill_args(GatherArgs& ga, const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt,
                      const float* shift, const float* res, const float* aux, const float* mask, float* y) {
  ga.d = *d;
  ga.x1 = x1; ga.x2 = x2; ga.wt = wt; ga.shift = shift; ga.res = res; ga.aux = aux; ga.y = y;
  ga.NP = d->N * d->Ho * d->Wo;
  ga.HoWo = d->Ho * d->Wo;
  ga.HsWs = d->Hs * d->Ws;
  ga.in1 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed);
  ga.in2 = make_inact(d->in_act, d->in_slope, d->drop_p, d->drop_seed + 0x9E3779B9u);
  ga.auxa = make_inact(d->aux_act, d->aux_slope, d->aux_drop_p, d->aux_drop_seed);
  ga.ph = ga.pw = -1;
  ga.subW = ga.subHW = 0;
  ga.mask = mask;
  const uintptr_t al = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(aux);
  ga.wide = (al & 15) == 0 && !d->d2s && d->stride == 1 && d->Wo % 4 == 0;
  ga.amax_out = nullptr;
  ga.amax2 = nullptr;
  ga.res2 = nullptr;
}

static int x6_launch(const vunet_conv_desc* d, const float* x1, const float* x2, const void* wx, const float* shift,
                     const float* res, const float* aux, const float* mask, float* y, const float* amax, float* amax_out,
                     long min_blocks, void* stream, const float* amax2 = nullptr, const float* res2 = nullptr) {
  const int pro = x6_prologue_code(d, mask != nullptr);
  if (res2 && (d->mode != 1 || !amax)) return VUNET_ERR_UNSUPPORTED;   // (second residual: the fp16 scheme's data gradients)
  if (amax && !mask && !aux && !res && !g_vunet_tune[VUNET_TUNE_S2_FWD_F32] && vunet_conv_h2_s2_ok(d, pro)) {
    GatherArgs gs;   // stride-2 forward: its own kernel (parity-plane staging)
    fill_args(gs, d, x1, x2, nullptr, shift, res, aux, mask, y);
    gs.amax_out = amax_out;
    gs.amax2 = amax2;
    return vunet_conv_h2_s2_launch(gs, wx, vunet_x6_mtiles(d->Mpad), amax, (hipStream_t)stream);
  }
  int MT = 0;
  int NT = x6_geometry_ok(d, pro, amax != nullptr) ? x6_pick(d, &MT, min_blocks, amax != nullptr) : 0;
  if (g_vunet_tune[VUNET_TUNE_FORCE_SMALL] && amax && !mask && vunet_conv_h2_small_ok(d, pro)) NT = 0;   // tests
  GatherArgs ga;
  fill_args(ga, d, x1, x2, nullptr, shift, res, aux, mask, y);
  ga.amax2 = amax2;
  ga.res2 = res2;
  // the 16-byte epilogue reads res2 with float4 loads too: an offset view that is contiguous but not 16-byte aligned
  // takes the scalar epilogue
  ga.wide = ga.wide && (reinterpret_cast<uintptr_t>(res2) & 15) == 0;
  if (amax) ga.amax_out = amax_out;   // only the fp16 kernels publish |y| maxima
  if (NT == 0) {
    if (amax) ga.amax_out = amax_out;   // (the small-map kernel publishes through every store, depth-to-space included)   // the row-tiled kernels do not cover / cannot fill the chip with this problem: the small-map form (fp16 scheme)
    if (!amax || mask || !(min_blocks <= 1 ? vunet_conv_h2_small_ok(d, pro) : vunet_conv_h2_small_wanted(d, pro)))
      return VUNET_ERR_UNSUPPORTED;
    return vunet_conv_h2_small_launch(ga, wx, vunet_x6_mtiles(d->Mpad), amax, pro, (hipStream_t)stream);
  }
  // K dimension of the image = the gathered tensor's channels; M dimension = all columns of the weight matrix
  const int mtp = vunet_x6_mtiles(d->Mpad);
  if (amax && d->Ws % 32) return vunet_conv_h2_launch_w16(ga, wx, mtp, amax, pro, MT, (hipStream_t)stream);
  if (amax && d->mode == 1 && d->stride == 2 && !g_vunet_tune[VUNET_TUNE_PARITY_LAUNCHES])
    return vunet_conv_h2_launch_par(ga, wx, mtp, amax, (hipStream_t)stream);   // all four output parities in one launch
  if (amax)   // two-term fp16 image
    return MT == 1 ? vunet_conv_h2_launch_mt1(ga, wx, mtp, amax, pro, NT, (hipStream_t)stream)
                   : vunet_conv_h2_launch_mt2(ga, wx, mtp, amax, pro, NT, (hipStream_t)stream);
  return MT == 1 ? vunet_conv_x6_launch_mt1(ga, wx, mtp, pro, NT, (hipStream_t)stream)
                 : vunet_conv_x6_launch_mt2(ga, wx, mtp, pro, NT, (hipStream_t)stream);
}

// would vunet_conv2d route this problem to the split-bf16 kernel?  (big enough to fill the chip, not a 3-channel layer)
static bool x6_wanted(const vunet_conv_desc* d, bool has_wx, bool has_aux, bool has_res, bool has_mask, bool h2 = false) {
  if (!has_wx || !x6_enabled()) return false;
  const int pro = x6_prologue_code(d, has_mask);
  if (vunet_conv_thin_kind(d, pro == 4 ? 0 : pro, has_aux, has_res) != 0) return false;
  if (h2 && !has_mask && !has_aux && !has_res && !g_vunet_tune[VUNET_TUNE_S2_FWD_F32] && vunet_conv_h2_s2_ok(d, pro))
    return true;   // stride-2 forward (conv_h2_s2.hip)
  int MT;
  if (x6_geometry_ok(d, pro, h2) && x6_pick(d, &MT, 128, h2) > 0) return true;
  return h2 && !has_mask && vunet_conv_h2_small_wanted(d, pro);   // the small-map form (conv_h2_small.hip)
}

static bool x6_uses_small(const vunet_conv_desc* d, bool has_mask, bool h2) {
  const int pro = x6_prologue_code(d, has_mask);
  if (h2 && !has_mask && !g_vunet_tune[VUNET_TUNE_S2_FWD_F32] && vunet_conv_h2_s2_ok(d, pro)) return false;
  int MT;
  return h2 && !has_mask && !(x6_geometry_ok(d, pro, h2) && x6_pick(d, &MT, 128, h2) > 0);
}

extern "C" int vunet_conv2d_x6_supported(const vunet_conv_desc* d, int32_t has_mask) {
  return d && x6_geometry_ok(d, x6_prologue_code(d, has_mask != 0)) ? 1 : 0;
}

extern "C" int vunet_conv2d_wants_split(const vunet_conv_desc* d, int32_t has_aux, int32_t has_res, int32_t has_mask,
                                        int32_t split) {
  return d && x6_wanted(d, true, has_aux != 0, has_res != 0, has_mask != 0, split == 2) ? 1 : 0;
}

extern "C" int vunet_conv2d_x6(const vunet_conv_desc* d, const float* x1, const float* x2, const void* wx,
                               const float* shift, const float* res, const float* aux, const float* mask, float* y,
                               const float* amax, float* amax_out, void* stream) {
  if (!d || !x1 || !wx || !y || (d->C2 > 0 && !x2) || d->Mpad % 32 != 0) return VUNET_ERR_ARG;
  if (mask && (d->mode != 1 || d->C2 != 0)) return VUNET_ERR_ARG;
  if (d->d2s && (d->M % 4 != 0 || d->mode != 0)) return VUNET_ERR_ARG;
  return x6_launch(d, x1, x2, wx, shift, res, aux, mask, y, amax, amax_out, 1, stream);
}

extern "C" int vunet_conv2d_a2(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt, const void* wx,
                               const float* shift, const float* res, const float* res2, const float* aux, float* y,
                               const float* amax, const float* amax2, float* amax_out, void* stream) {
  if (!d) return VUNET_ERR_ARG;
  if (amax2 && (!amax || d->C2 <= 0)) return VUNET_ERR_ARG;
  if (res2 && (!res || d->mode != 1)) return VUNET_ERR_ARG;
  if (x6_wanted(d, wx != nullptr, aux != nullptr, res != nullptr, false, amax != nullptr)) {
    if (!x1 || !y || (d->C2 > 0 && !x2) || d->Mpad % 32 != 0) return VUNET_ERR_ARG;
    return x6_launch(d, x1, x2, wx, shift, res, aux, nullptr, y, amax, amax_out, 128, stream, amax2, res2);
  }
  if (res2) return VUNET_ERR_UNSUPPORTED;   // (the fp32 kernels take one residual: the caller adds the second itself)
  return vunet_conv2d_gather_amax(d, x1, x2, wt, shift, res, aux, y, amax_out, stream);
}

extern "C" int vunet_conv2d(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt, const void* wx,
                            const float* shift, const float* res, const float* aux, float* y, const float* amax,
                            float* amax_out, void* stream) {
  return vunet_conv2d_a2(d, x1, x2, wt, wx, shift, res, nullptr, aux, y, amax, nullptr, amax_out, stream);
}

extern "C" int vunet_conv2d_publishes_amax(const vunet_conv_desc* d, int32_t has_aux, int32_t has_res, int32_t split) {
  if (!d) return 0;
  if (d->d2s)   // through the sub-pixel store only the fp16 kernels publish
    return split == 2 && x6_wanted(d, true, has_aux != 0, has_res != 0, false, true) ? 1 : 0;
  if (split != 0 && x6_wanted(d, true, has_aux != 0, has_res != 0, false, split == 2)) return split == 2 ? 1 : 0;
  return vunet_conv2d_gather_publishes(d, has_aux != 0, has_res != 0) ? 1 : 0;
}

extern "C" int vunet_conv2d_dgrad_relu_x6(const vunet_conv_desc* d, const float* dy, const float* y, const void* wx,
                                          const float* res, float* dx, const float* amax, float* amax_out, void* stream) {
  if (!d || !dy || !y || !wx || !dx || d->mode != 1 || d->C2 != 0 || d->Mpad % 32 != 0) return VUNET_ERR_ARG;
  if (d->aux_act != ACT_NONE || d->aux_drop_p > 0.f || !x6_wanted(d, true, false, res != nullptr, true, amax != nullptr))
    return VUNET_ERR_UNSUPPORTED;
  return x6_launch(d, dy, nullptr, wx, nullptr, res, nullptr, y, dx, amax, amax_out, 128, stream);
}

// Name (rocprofv3 spelling) of the kernel vunet_conv2d / vunet_conv2d_dgrad_relu_x6 launches for this problem.
extern "C" int vunet_conv2d_variant(const vunet_conv_desc* d, int32_t has_aux, int32_t has_wx, int32_t has_mask,
                                    char* name, int32_t len) {
  if (!d || !name || len < 8) return VUNET_ERR_ARG;
  if (x6_wanted(d, has_wx != 0, has_aux != 0, false, has_mask != 0, has_wx == 2)) {
    if (has_wx == 2 && !has_mask && !has_aux && !g_vunet_tune[VUNET_TUNE_S2_FWD_F32] &&
        vunet_conv_h2_s2_ok(d, x6_prologue_code(d, false))) {
      vunet_conv_h2_s2_name(d, name, len);
      return VUNET_OK;
    }
    if (x6_uses_small(d, has_mask != 0, has_wx == 2))
      return vunet_conv_h2_small_name(d, x6_prologue_code(d, has_mask != 0), name, len);
    int MT;
    const int NT = x6_pick(d, &MT, 128, has_wx == 2);
    const char* fam = has_wx == 2 ? "conv_h2_kernel" : "conv_x6_kernel";
    if (has_wx == 2 && d->Ws % 32) {
      snprintf(name, len, "%s<%d, 1, %d, %d, -1, 4, 16>", fam, MT, d->mode, x6_prologue_code(d, has_mask != 0));
    } else if (has_wx == 2) {
      if (d->stride == 2 && !g_vunet_tune[VUNET_TUNE_PARITY_LAUNCHES]) snprintf(name, len, "%s<1, 1, 1, 0, 4, 4, 32>", fam);
      else if (d->stride == 2) snprintf(name, len, "%s<%d, %d, 1, 0, parity x4, 4, 32>", fam, MT, NT);
      else snprintf(name, len, "%s<%d, %d, %d, %d, -1, 4, 32>", fam, MT, NT, d->mode, x6_prologue_code(d, has_mask != 0));
    } else if (d->stride == 2) snprintf(name, len, "%s<%d, %d, 1, 0, parity x4>", fam, MT, NT);
    else snprintf(name, len, "%s<%d, %d, %d, %d, -1>", fam, MT, NT, d->mode, x6_prologue_code(d, has_mask != 0));
    return VUNET_OK;
  }
  return vunet_conv2d_gather_variant(d, has_aux, name, len);
}
