/*
 * vunet_hip.h -- C ABI of libvunet_hip.so: MI355X (gfx950) kernels for the VUnet
 * shape-and-posture synthesis hot path.
 *
 * The upstream reference (CompVis/behavior-driven-video-synthesis) has no native layer: the path
 * sits between its Python nn.Module API and ATen/cuDNN.  Each entry point below therefore cites
 * the reference *operation* (file:line, relative to the upstream root) that it replaces; the
 * Python host side (behavior_driven_video_synthesis_amd/) binds these with ctypes -- see
 * INTEGRATION.md for the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only, no torch types.  All tensors are contiguous fp32
 * NCHW device buffers owned by the caller; the library allocates nothing persistent, is stateless
 * and re-entrant per stream.  `stream` is a hipStream_t passed as void*.  Every function returns 0
 * on success or a negative VUNET_ERR_* code (no exceptions cross the ABI).
 */
#ifndef VUNET_HIP_H
#define VUNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VUNET_OK 0
#define VUNET_ERR_ARG (-1)
#define VUNET_ERR_LAUNCH (-2)
#define VUNET_ERR_UNSUPPORTED (-3)

/* activation codes */
#define VUNET_ACT_NONE 0
#define VUNET_ACT_ELU 1      /* nn.ELU(alpha=1), lib/modules.py:207 */
#define VUNET_ACT_RELU 2     /* VGG19 features ReLU, models/imagenet_pretrained.py:53-59 */
#define VUNET_ACT_SIGMOID 3  /* EncDownAlter.squash, models/vunets.py:556,575 */
#define VUNET_ACT_LRELU 4    /* PatchGAN LeakyReLU(0.2), models/synth_discriminator.py:34 */

int vunet_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *
 * Replaces F.conv2d + the pointwise ops around it in NormConv2d / VunetRNB / Upsample /
 * Downsample (lib/modules.py:140-145, 221-233, 160-161, 179-182) and the VGG19 conv+ReLU stack
 * (models/imagenet_pretrained.py:53-59):
 *   prologue  : activation (ELU ...) and dropout on the gathered input values (lib/modules.py:229-230)
 *   sources   : two input tensors read as one channel-concatenated tensor (torch.cat, lib/modules.py:227)
 *   epilogue  : + shift[c] (folded gamma*bias+beta), activation, + residual (lib/modules.py:233),
 *               block-major depth-to-space store (lib/modules.py:29-34)
 * mode 1 runs the transposed gather (data gradient) with epilogue  y = acc * act'(aux) + res.
 * ------------------------------------------------------------------------------------------ */
typedef struct vunet_conv_desc {
  int32_t N;
  int32_t C1, C2;       /* channels of source 1 / source 2 (C2 == 0: single source)            */
  int32_t Hs, Ws;       /* spatial size of the gathered (source) tensors                        */
  int32_t M;            /* output channels produced by this launch                              */
  int32_t m_off;        /* first column of wt used                                              */
  int32_t Mpad;         /* row pitch of wt in floats (multiple of 32)                           */
  int32_t Ho, Wo;       /* spatial size of the produced tensor (before depth-to-space)          */
  int32_t KH, KW, stride, pad;
  int32_t mode;         /* 0: ih = oh*stride - pad + kh ; 1: ih = (oh + pad - kh)/stride        */
  int32_t in_act;       /* prologue activation                                                  */
  float in_slope;
  float drop_p;         /* prologue dropout probability (0: off), keep-mask = hash(idx+seed)    */
  uint32_t drop_seed;
  int32_t out_act;      /* mode 0 epilogue activation                                           */
  int32_t d2s;          /* mode 0: store through DepthToSpace(2)                                */
  int32_t aux_act;      /* mode 1 epilogue: multiply by d act / d v evaluated at aux            */
  float aux_slope;
  float aux_drop_p;
  uint32_t aux_drop_seed;
} vunet_conv_desc;

/* wt: [T*(C1p+C2p)][Mpad] K-major effective weights (rows: source, tap, channel; CXp = CX rounded up
 * to 2), shift: [M] or NULL, res: tensor shaped like y or NULL, aux: tensor shaped like y or NULL. */
int vunet_conv2d_gather(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt,
                        const float* shift, const float* res, const float* aux, float* y, void* stream);

/* Data gradient through a layer whose forward epilogue was ReLU (the VGG19 stack, models/imagenet_pretrained.py:
 * autograd's relu backward + conv backward):  dx = dgrad(dy * [y > 0]) + res, the mask applied while dy is staged.
 * d as for vunet_conv2d_gather mode 1 (C1 = channels of dy / y, M = channels of dx, wt = wt_d).  Geometries the
 * LDS-tiled kernel does not cover return VUNET_ERR_UNSUPPORTED (use vunet_act_bwd_from_out + vunet_conv2d_gather). */
int vunet_conv2d_dgrad_relu(const vunet_conv_desc* d, const float* dy, const float* y, const float* wt,
                            const float* res, float* dx, void* stream);

/* Name (as rocprofv3 prints it) of the kernel vunet_conv2d_gather selects for this problem; has_aux: the
 * call passes an aux tensor.  For profiling / roofline bookkeeping only. */
int vunet_conv2d_gather_variant(const vunet_conv_desc* d, int32_t has_aux, char* name, int32_t len);

/* ------------------------------------------------------------------------------------------
 * fp32-accurate convolution on the bf16 matrix cores ("x6": exact 3-way bf16 split of both operands, the six
 * leading partial products accumulated in fp32 on v_mfma_f32_32x32x16_bf16 -- csrc/conv_x6_kernel.h).  Same
 * operation, prologue / sources / epilogue and fp32 NCHW tensors as vunet_conv2d_gather; the dropped partial
 * products are below fp32 rounding, so results match the fp32-MFMA kernels to fp32 accuracy at 2.67x their MFMA roof.
 * Covers 3x3 / stride 1 / pad 1, C1 and C2 multiples of 16, M and m_off multiples of 32, Ws % 32 == 0, Hs % 4 == 0
 * (the two-term fp16 form below also Ws % 16 == 0 with Hs % 8 == 0),
 * prologue none / ELU / ELU+dropout (mode 0), none (mode 1).
 *   wx   : split weight image written by vunet_weightnorm_fwd* (wx_f for mode 0, wx_d for mode 1);
 *   mask : mode 1 only, tensor shaped like x1 -- x1 is multiplied by [mask > 0] while staged (ReLU backward);
 *   vunet_conv2d          : THE convolution entry point of the host code: the split-bf16 kernel when wx != NULL, the
 *                           geometry is covered and the launch fills the chip, else vunet_conv2d_gather.
 *                           VUNET_CONV_PRECISION=f32 in the environment pins every layer to the fp32-MFMA kernels.
 *   vunet_conv2d_x6       : the split-bf16 kernel or VUNET_ERR_UNSUPPORTED (no size heuristics; tests).
 *
 * "h2": the same kernel structure on the fp16 matrix cores with HALF the matrix instructions (csrc/conv_h2_kernel.h):
 * operands scaled by a power of two (x: from its tensor maximum; w: per layer, inside the image), split into two fp16
 * terms, three partial products, fp32 accumulate; as accurate as x6 wherever |x| >= 2^-28 of its tensor's maximum.
 *   amax : NULL -> wx is a three-term bf16 image (vunet_wn_desc.split 0/1);  non-NULL -> wx is a two-term fp16 image
 *          (split 2) and amax holds the 1024 partial maxima of |x1|, |x2| written by vunet_absmax_partials.
 *   amax_out : optional, >= 512 floats ZEROED by the caller: when amax != NULL and vunet_conv2d_publishes_amax(d, ...) the
 *          kernel's epilogue leaves partial maxima of |y| there -- slots 0..511 of the `amax` of a convolution that reads
 *          y as its first source (512..1023: the second source's maxima, zeros if there is none).
 * ------------------------------------------------------------------------------------------ */
int vunet_conv2d(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt, const void* wx,
                 const float* shift, const float* res, const float* aux, float* y, const float* amax, float* amax_out,
                 void* stream);
/* The same with the two sources' maxima in buffers of their own (each the 512-slot |y| maxima its producer published through
 * amax_out): amax = source 1's, amax2 = source 2's (>= 512 floats each); amax2 NULL = vunet_conv2d.  Saves the caller the
 * launch that would concatenate them. */
/* res2 (data gradient, mode 1, with res): a second tensor shaped like y added in the epilogue -- the gradient that reaches the
 * layer's input through its other reader when `res` is taken by dy itself (VunetRNB: x is the convolution's source AND the
 * residual, lib/modules.py:221-233); the fp16-scheme kernels only, VUNET_ERR_UNSUPPORTED (nothing launched) elsewhere. */
int vunet_conv2d_a2(const vunet_conv_desc* d, const float* x1, const float* x2, const float* wt, const void* wx,
                    const float* shift, const float* res, const float* res2, const float* aux, float* y, const float* amax,
                    const float* amax2, float* amax_out, void* stream);
int vunet_conv2d_x6(const vunet_conv_desc* d, const float* x1, const float* x2, const void* wx, const float* shift,
                    const float* res, const float* aux, const float* mask, float* y, const float* amax, float* amax_out,
                    void* stream);
int vunet_conv2d_x6_supported(const vunet_conv_desc* d, int32_t has_mask);
/* 1: vunet_conv2d (has_mask: vunet_conv2d_dgrad_relu_x6) given a split image runs a split kernel for this problem --
 * the caller of the fp16 scheme then owes it the |x| maxima (vunet_absmax_partials); 0: it will not look at amax */
int vunet_conv2d_wants_split(const vunet_conv_desc* d, int32_t has_aux, int32_t has_res, int32_t has_mask, int32_t split);
/* (split: vunet_wn_desc.split of the image the caller holds -- the fp16 scheme also covers 16-wide maps) */
/* vunet_conv2d_dgrad_relu on the split kernels (wx = wx_d); VUNET_ERR_UNSUPPORTED -> use vunet_conv2d_dgrad_relu */
int vunet_conv2d_dgrad_relu_x6(const vunet_conv_desc* d, const float* dy, const float* y, const void* wx,
                               const float* res, float* dx, const float* amax, float* amax_out, void* stream);
/* 1: vunet_conv2d, given a split image of layout `split` (0: none) and (for split 2) the maxima, will fill amax_out for this problem
 * (the two-term fp16 kernels -- also through depth-to-space --, the streaming 1x1 kernel, the LDS-tiled kernel, the split-K
 * kernel, the 3-channel-input kernel); 0: it will leave amax_out untouched */
int vunet_conv2d_publishes_amax(const vunet_conv_desc* d, int32_t has_aux, int32_t has_res, int32_t split);
/* out[0..511] / out[512..1023]: partial maxima of |x1| / |x2| (x2 may be NULL: zeros); one launch, no atomics */
int vunet_absmax_partials(const float* x1, int64_t n1, const float* x2, int64_t n2, float* out, void* stream);
/* kernel name vunet_conv2d (has_mask: vunet_conv2d_dgrad_relu_x6) selects, rocprofv3 spelling */
int vunet_conv2d_variant(const vunet_conv_desc* d, int32_t has_aux, int32_t has_wx, int32_t has_mask, char* name,
                         int32_t len);   /* has_wx: 0 none, 1 three-term bf16 image, 2 two-term fp16 image */

/* bf16-operand forward convolution of the inference path (models/vunets.py:508-515 `transfer`, run per frame
 * by the render loop; BASELINE config 5): operands rounded to bf16 (RNE) on the way into LDS, fp32 accumulate
 * on v_mfma_f32_32x32x16_bf16, fp32 NCHW tensors in HBM, same prologue / sources / epilogue as
 * vunet_conv2d_gather mode 0.  Covers 3x3 / stride 1 / pad 1 layers with C1, C2 multiples of 16, Ws a
 * multiple of 32 and Hs of 4, prologue none or ELU, no dropout; vunet_conv2d_bf16_supported tells (1 / 0),
 * anything else goes through vunet_conv2d_gather in fp32.
 *   vunet_pack_bf16: wt_f (the fp32 K-major weights of vunet_weightnorm_fwd, row pitch Mpad) ->
 *                    wb [(C1+C2)/16][9][Mpad][16] bf16  ((C1+C2)*9*Mpad*2 bytes). */
int vunet_conv2d_bf16_supported(const vunet_conv_desc* d);
int vunet_pack_bf16(const float* wt_f, void* wb, int32_t C1, int32_t C2, int32_t Mpad, void* stream);
int vunet_conv2d_bf16(const vunet_conv_desc* d, const float* x1, const float* x2, const void* wb,
                      const float* shift, const float* res, float* y, void* stream);

/* The same inference path on CHANNEL-BLOCKED bf16 activations (csrc/conv_blk.hip): tensors are
 *   blk [N][C/8][H][W][8] bf16  -- 16-byte units of 8 consecutive channels of one pixel (C a multiple of 8),
 * half the bytes of fp32 NCHW per layer and one 16-byte load per matrix-core B fragment.  Every layer of
 * VunetAlter.transfer (models/vunets.py:508-515) is covered: 3x3 / 1x1, stride 1 / 2, two sources (the skip concat),
 * ELU prologue, residual, depth-to-space store; products accumulate in fp32, activations are rounded to bf16 (RNE)
 * once, when stored.  d as for vunet_conv2d_gather mode 0 with C1, C2 multiples of 16, M of 8 (32 with d2s), m_off 0,
 * no dropout; x1 / x2 / res / y blk tensors; y_fp32_nchw != 0: y is a fp32 NCHW tensor instead (any M; the network's
 * 3-channel output layer).
 *   vunet_pack_bf16_taps     wt_f (fp32 K-major weights of vunet_weightnorm_fwd, row pitch Mpad) ->
 *                            wb [(C1+C2)/16][2 k-halves][taps][Mpad][8] bf16, taps = 1 or 9
 *   vunet_nchw_to_blk / vunet_blk_to_nchw   fp32 NCHW <-> blk (RNE / exact)
 *   vunet_conv1x1_few_to_blk 1x1 convolution of a fp32 NCHW tensor with C <= 4 channels (the stickman planes; wt_f rows =
 *                            input channel) in fp32, stored as blk -- the first layer of the pose encoder */
int vunet_conv2d_blk(const vunet_conv_desc* d, const void* x1, const void* x2, const void* wb, const float* shift,
                     const void* res, void* y, int32_t y_fp32_nchw, void* stream);
/* A residual block with a skip input in one launch (lib/modules.py:221-233, eval mode): y = res + conv3x3(elu(cat(x, nin(elu(skip))))),
 * the 1x1 `nin` computed on the tile's halo in LDS instead of stored and re-loaded; bit-identical to vunet_conv2d_blk(nin) followed by
 * vunet_conv2d_blk(3x3).  d: the 3x3 (C1 = C2 = M in {32, 64}, ELU prologue, Ws % 32 = 0, Hs % 4 = 0); wb_nin / shift_nin: the nin
 * layer's vunet_pack_bf16_taps image (taps = 1, row pitch Mpad_nin) and shift.  _supported: 1 / 0. */
int vunet_conv2d_blk_rnb_supported(const vunet_conv_desc* d);
int vunet_conv2d_blk_rnb(const vunet_conv_desc* d, const void* x, const void* skip, const void* wb_nin, const float* shift_nin,
                         int32_t Mpad_nin, const void* wb, const float* shift, const void* res, void* y, void* stream);
/* 1: vunet_conv2d_blk runs the LDS-tiled kernel for this problem (conv_blk_tiled_kernel), 0: the direct one */
int vunet_conv2d_blk_tiled(const vunet_conv_desc* d);
int vunet_pack_bf16_taps(const float* wt_f, void* wb, int32_t C1, int32_t C2, int32_t Mpad, int32_t taps, void* stream);
int vunet_nchw_to_blk(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int vunet_blk_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int vunet_conv1x1_few_to_blk(const float* x, const float* wt_f, const float* shift, void* y, int32_t N, int32_t C,
                             int32_t H, int32_t W, int32_t M, int32_t Mpad, void* stream);

/* Weight gradient  dW[co][tap][ci] = sum_px dy[co][px] * f(x)[ci][px (+) tap]  with the same
 * prologue f as the forward; split over `nsplit` pixel ranges into partial slabs
 *   slabs[nsplit][Coutp][T*(C1+C2)]  (Coutp = Cout rounded up to 32)  and  dshift[nsplit][Coutp].
 * Replaces the weight-gradient half of autograd's conv backward for the layers above. */
typedef struct vunet_wgrad_desc {
  int32_t N, C1, C2, Hs, Ws;
  int32_t Cout, Ho, Wo;
  int32_t KH, KW, stride, pad;
  int32_t in_act;
  float in_slope;
  float drop_p;
  uint32_t drop_seed;
  int32_t nsplit;
  int32_t flags;        /* bit 0: keep this problem on the fp32-input MFMA kernels (default: an fp32-accurate split
                           kernel wherever one applies: 3x3 / stride 1 / pad 1, channel counts in 32s, Ws % 32 == 0,
                           Hs % 4 == 0);  bit 1: the split kernel is the two-term fp16 one (csrc/conv_wgrad_h2.hip: three
                           products, needs amax_x / amax_dy) instead of the three-term bf16 one (csrc/conv_wgrad_x6.hip) */
} vunet_wgrad_desc;

/* amax_x / amax_dy: partial maxima (vunet_absmax_partials) of |x1|, |x2| and of |dy|; read only when flags bit 1 is set
 * and vunet_conv2d_wgrad_wants_split(d) == 1, else may be NULL */
int vunet_conv2d_wgrad(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                       float* slabs, float* dshift, const float* amax_x, const float* amax_dy, void* stream);
/* (amax_x2: the second source's 512 partial maxima in a buffer of their own, amax_x then holding the first source's; NULL:
 * amax_x holds all 1024 -- as vunet_conv2d_a2) */
int vunet_conv2d_wgrad_a2(const vunet_wgrad_desc* d, const float* x1, const float* x2, const float* dy,
                          float* slabs, float* dshift, const float* amax_x, const float* amax_x2, const float* amax_dy,
                          void* stream);
/* The weight gradients of SEVERAL layers in as few launches as their kernel forms allow (items of one form share a launch,
 * up to 12 per launch; argument blocks travel by value).  For the layers vunet_conv2d_wgrad_batchable(d) == 1 accepts: the
 * small-map / 1x1 / stride-2 layers of the fp16 scheme whose own launch is latency, not work (N * Ho * Wo <= 16384) -- about 60
 * per training step of the reference's VunetAlter at 256^2, each the weight-gradient half of a NormConv2d backward
 * (lib/modules.py:120-145).  Every item as vunet_conv2d_wgrad_a2's arguments; results identical to one call per item. */
typedef struct vunet_wgrad_item {
  vunet_wgrad_desc d;
  const float* x1;
  const float* x2;
  const float* dy;
  float* slabs;
  float* dshift;
  const float* amax_x;
  const float* amax_x2;
  const float* amax_dy;
} vunet_wgrad_item;
int vunet_conv2d_wgrad_multi(const vunet_wgrad_item* items, int32_t n, void* stream);
int vunet_conv2d_wgrad_batchable(const vunet_wgrad_desc* d);
int vunet_conv2d_wgrad_wants_split(const vunet_wgrad_desc* d);
int vunet_conv2d_wgrad_variant(const vunet_wgrad_desc* d, char* name, int32_t len);
/* number of pixel splits the library wants for this problem (caller sizes the slabs from it) */
int vunet_conv2d_wgrad_nsplit(const vunet_wgrad_desc* d);

/* ------------------------------------------------------------------------------------------
 * Weight normalisation  (torch._weight_norm, lib/modules.py:135-138) folded with the learned
 * affine of NormConv2d (lib/modules.py:143-145):
 *   scale[co] = gamma[co] * g[co] / ||v[co]||      shift[co] = gamma[co]*bias[co] + beta[co]
 * and packed into the K-major layouts the conv kernels read.
 *   kind 0: NormConv2d (v, g, bias, gamma, beta)      kind 1: plain conv (weight=v, bias)
 *   kind 2: L2NormConv2d (weight=v normalised, gamma, beta; lib/modules.py:89-101)
 * ------------------------------------------------------------------------------------------ */
typedef struct vunet_wn_desc {
  int32_t Cout, C1, C2, KH, KW;
  int32_t kind;
  int32_t split;        /* layout of the split weight images wx_f / wx_d: 0 or 1 = three bf16 terms (conv_x6_kernel.h),
                           2 = two fp16 terms with a per-layer power-of-two scale (conv_h2_kernel.h) */
} vunet_wn_desc;

/* outputs: wt_f [T*(C1p+C2p)][Coutp32]  (forward),  wt_d [T*Coutp2][Cinp32] (dgrad; NULL to skip),
 * wx_f / wx_d: the split images (d->split) of the same two matrices for vunet_conv2d (vunet_x6_image_bytes bytes each;
 * NULL to skip), scale[Cout], shift[Cout], invnorm[Cout], wmax[Cout] = max |w_eff| per row (required when d->split == 2
 * and an image is wanted, else may be NULL).  Any of g/bias/gamma/beta may be NULL per kind.
 * (CXp = CX rounded up to 2, Coutp2 = Cout rounded up to 2, Coutp32/Cinp32 rounded up to 32.) */
int vunet_weightnorm_fwd(const vunet_wn_desc* d, const float* v, const float* g, const float* bias,
                         const float* gamma, const float* beta, float* wt_f, float* wt_d, void* wx_f, void* wx_d,
                         float* scale, float* shift, float* invnorm, float* wmax, void* stream);

/* Batched form: the weights of every layer of a model in two launches.  items_dev: DEVICE array of n_items
 * entries (the pointers are device pointers; entries with kind/NULL rules as above); max_cout = max over items.
 * PRECONDITION (ABI 8 on): wt_f, wt_d, wx_f and wx_d of every item must be ZERO-FILLED once by the caller before the first
 * call (hipMemset; repeated calls on the same buffers need no refill).  The batched pack is LDS-tiled and writes only what
 * its tiles own: the padding m-tiles of wx_f / wx_d, the lanes past Ctot and the columns past Ctot of wt_d are never
 * written, and the split kernels multiply whatever those positions hold (uninitialised memory -> garbage / NaN in the
 * padding rows).  The per-layer vunet_weightnorm_fwd writes every position itself and has no such precondition. */
typedef struct vunet_wn_item {
  const float *v, *g, *bias, *gamma, *beta;
  float *wt_f, *wt_d, *scale, *shift, *invnorm;
  void *wx_f, *wx_d;
  float* wmax;
  vunet_wn_desc d;
} vunet_wn_item;
int vunet_weightnorm_fwd_multi(const vunet_wn_item* items_dev, int32_t n_items, int32_t max_cout, void* stream);
/* size in bytes of the split weight image of a layer (0: geometry not covered); dgrad != 0: the wx_d image */
int vunet_x6_image_bytes(const vunet_wn_desc* d, int32_t dgrad);
int vunet_x6_mtiles(int32_t M);
/* (vunet_x6_mtiles: 32-channel tiles of the image's M dimension, padded so that any workgroup may read two) */

/* backward: reduces the wgrad slabs (fixed order) and produces the parameter gradients
 * dv[Cout][Cin][KH][KW], dg[Cout], dbias[Cout], dgamma[Cout], dbeta[Cout] (NULL to skip).
 * workspace: Cout*(KH*KW*(C1+C2) + 1) floats.  accumulate != 0: add into the outputs (they are then the
 * parameters' .grad buffers, written without an autograd accumulation pass). */
int vunet_weightnorm_bwd(const vunet_wn_desc* d, const float* slabs, const float* dshift, int32_t nsplit,
                         const float* v, const float* g, const float* bias, const float* gamma,
                         const float* invnorm, float* dv, float* dg, float* dbias, float* dgamma,
                         float* dbeta, float* workspace, int32_t accumulate, void* stream);

/* Batched form: the slab reduction and parameter gradients of many layers in two launches (one grid row per layer).
 * items_dev: DEVICE array of n_items entries, fields as the arguments of vunet_weightnorm_bwd; max_cout = max Cout over
 * the items, max_reduce_blocks = max over the items of Cout * ceil(KH*KW*(C1+C2) / 64).  Replaces, per training step,
 * the 2 launches per layer the autograd of torch._weight_norm + the gamma/beta affine issue (lib/modules.py:120-145). */
typedef struct vunet_wn_bwd_item {
  const float *slabs, *dshift, *v, *g, *bias, *gamma, *invnorm;
  float *dv, *dg, *dbias, *dgamma, *dbeta;
  float* workspace;
  vunet_wn_desc d;
  int32_t nsplit, accumulate;
} vunet_wn_bwd_item;
int vunet_weightnorm_bwd_multi(const vunet_wn_bwd_item* items_dev, int32_t n_items, int32_t max_cout,
                               int32_t max_reduce_blocks, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pointwise / index / reduction kernels (HBM-bound)
 * ------------------------------------------------------------------------------------------ */
/* DepthToSpace / SpaceToDepth, block-major channel order (lib/modules.py:11-34). x:[N,C,H,W] */
int vunet_depth_to_space(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int vunet_space_to_depth(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);

/* nn.Upsample(scale_factor=2, mode="bilinear"), align_corners False -- the non-sub-pixel branch of Upsample
 * (lib/modules.py:172-175).  x: [NC][H][W] -> y: [NC][2H][2W]; bwd: the adjoint in gather form (deterministic). */
int vunet_upsample_bilinear2x_fwd(const float* x, float* y, int64_t NC, int32_t H, int32_t W, void* stream);
int vunet_upsample_bilinear2x_bwd(const float* dy, float* dx, int64_t NC, int32_t H, int32_t W, void* stream);

/* y = a*x + b*y_in style helpers used by the autograd glue */
int vunet_axpby(const float* x, const float* y_in, float* y, float a, float b, int64_t n, void* stream);
/* dx = dy * act'(.) given the activation OUTPUT (sigmoid: y(1-y); relu: y>0; lrelu) */
int vunet_act_bwd_from_out(const float* y, const float* dy, float* dx, int32_t act, float slope, int64_t n,
                           void* stream);
int vunet_act_fwd(const float* x, float* y, int32_t act, float slope, int64_t n, void* stream);

/* reparametrisation z = eps*exp(logstd) + mu  (models/vunets.py:594-597) and its gradients */
int vunet_reparam_fwd(const float* mu, const float* logstd, const float* eps, float* z, int64_t n, void* stream);
int vunet_reparam_bwd(const float* dz, const float* logstd, const float* eps, float* dmu, float* dlogstd,
                      int64_t n, void* stream);

/* mean |a-b| (lib/losses.py:98-102): out[0] += weight * mean|a-b| ; backward wrt b:
 * db = (add ? add : 0) + gscale * gout[0] * sign(b-a)  with gscale = weight/n; gout: device scalar (upstream
 * gradient, NULL = 1) so that no host sync is needed */
int vunet_l1_mean_fwd(const float* a, const float* b, float* partial, float* out, float weight, int64_t n,
                      void* stream);
int vunet_l1_mean_bwd(const float* a, const float* b, const float* add, float* db, float gscale,
                      const float* gout, int64_t n, void* stream);
/* the same; amax_out (optional, >= 512 floats zeroed by the caller) receives partial maxima of |db|, in the form
 * vunet_conv2d's `amax` expects of the tensor it reads (the gradient enters the last VGG layer's data gradient);
 * relu_mask != 0: b is the output of a ReLU layer and db is zeroed where b <= 0 -- the mask that layer's backward
 * (autograd's relu backward, models/imagenet_pretrained.py) would apply next, applied here in passing */
int vunet_l1_mean_bwd_amax(const float* a, const float* b, const float* add, float* db, float gscale,
                           const float* gout, int64_t n, float* amax_out, int32_t relu_mask, void* stream);

/* KL(N(mu, exp(l)^2) || N(0,1)) per lib/losses.py:283-291: out[0] += weight * mean_n(sum_d(-l + .5(e^{2l}+mu^2)) - .5 D) */
int vunet_kl_fwd(const float* mu, const float* logstd, float* partial, float* out, float weight, int32_t N, int64_t D,
                 void* stream);  /* partial: >= 256 floats of workspace (two-stage deterministic sum) */
int vunet_kl_bwd(const float* mu, const float* logstd, float* dmu, float* dlogstd, float gscale,
                 const float* gout, int64_t n, void* stream);
/* 0.5*(p-q)^2 summed over CHW, batch mean (lib/losses.py:26-37) */
int vunet_sqdiff_fwd(const float* p, const float* q, float* partial, float* out, float weight, int32_t N, int64_t D,
                     void* stream);
int vunet_sqdiff_bwd(const float* p, const float* q, float* dp, float* dq, float gscale, const float* gout,
                     int64_t n, void* stream);

/* VGG input affine ((x+1)/2 - mean_c)/std_c  (models/imagenet_pretrained.py:43-44) */
int vunet_vgg_preprocess(const float* x, float* y, int32_t N, int32_t H, int32_t W, void* stream);
/* dx = (add ? add : 0) + dy * 0.5/std_c */
int vunet_vgg_preprocess_bwd(const float* dy, const float* add, float* dx, int32_t N, int32_t H, int32_t W,
                             void* stream);
/* MaxPool2d(2,2) forward / backward (recomputes the argmax from x) */
int vunet_maxpool2_fwd(const float* x, float* y, int32_t NC, int32_t H, int32_t W, void* stream);
/* Backward of an L1 tap (vunet_l1_mean_*) and of the 2x2 max-pool that reads the same tensor b [NC, H, W], in one pass
 * (models/imagenet_pretrained.py: relu1_2 / relu2_2 feed a loss term AND the next pool):
 *   db = route(dy_pool) + gscale * gout[0] * sign(b - a)   (dy_pool [NC, H/2, W/2] or NULL; routing as vunet_maxpool2_bwd),
 * zeroed where b <= 0 if relu_mask, |db| maxima to amax_out (optional) -- as vunet_l1_mean_bwd_amax */
/* forward of the same pair in one pass: out[0] += weight * mean|a - b| (partial: >= 1024 floats of workspace) and
 * y = maxpool2(b) [NC, H/2, W/2] */
int vunet_l1_pool_fwd(const float* a, const float* b, float* partial, float* out, float* y, float weight, int32_t NC,
                      int32_t H, int32_t W, void* stream);
int vunet_l1_pool_bwd(const float* a, const float* b, const float* dy_pool, float* db, float gscale, const float* gout,
                      int32_t NC, int32_t H, int32_t W, float* amax_out, int32_t relu_mask, void* stream);
/* (vunet_maxpool2_bwd_relu: x is a ReLU output -- dx additionally zeroed where x <= 0, as for vunet_l1_mean_bwd_amax) */
int vunet_maxpool2_bwd_relu(const float* x, const float* y, const float* dy, float* dx, int32_t NC, int32_t H, int32_t W,
                            void* stream);
int vunet_maxpool2_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t NC, int32_t H,
                       int32_t W, void* stream);

/* InstanceNorm2d(affine=False, eps) forward/backward (lib/modules.py:112; synth_discriminator.py:48,63)
 * stats: [NC][2] (mean, rstd) written by fwd and read by bwd */
int vunet_instnorm_fwd(const float* x, float* y, float* stats, int32_t NC, int32_t HW, float eps, void* stream);
int vunet_instnorm_bwd(const float* y, const float* dy, const float* stats, float* dx, int32_t NC, int32_t HW,
                       void* stream);

/* Fused multi-tensor Adam over one flat fp32 buffer (torch.optim.Adam semantics, eps outside sqrt;
 * experiments/shape_and_pose_net.py:237-246).  grad_scale multiplies the gradient first. */
int vunet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                    void* stream);

/* The same update with lr (float64) and the step count (int64) read from device memory: no per-step launch argument,
 * so the launch can be captured in a hipGraph and replayed (experiments.shape_and_pose_net graph mode). */
int vunet_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const double* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                        const int64_t* step_dev, float grad_scale, void* stream);

/* Optional, process-wide: a device-resident step counter for the dropout hash.  While set (non-NULL), every dropout
 * prologue / epilogue launched afterwards uses  seed + (*step_dev) * 0x9E3779B1  instead of seed: launches whose arguments
 * are frozen in a captured hipGraph then draw a fresh mask on every replay, identically in the forward, data-gradient and
 * weight-gradient kernels of a step.  NULL (the default) restores plain seeds.  The pointer must stay valid while set. */
int vunet_set_dropout_step(const uint32_t* step_dev);

/* Process-wide tuning knobs (tests / kernel tuning; the defaults are what production runs).  Replaces per-launch getenv()
 * lookups: read with a plain load on every launch.  key: VUNET_TUNE_*; value 0 restores the dispatcher's own choice.
 *   VUNET_TUNE_SPLIT_FORCE_NT  tile height (32-pixel rows per wave: 1, 2 or 4) of the split-fp16 / split-bf16 3x3 kernels
 *   VUNET_TUNE_TILED_FORCE_NT  the same for the LDS-tiled fp32 kernel
 *   VUNET_TUNE_FORCE_SMALL     1: vunet_conv2d_x6 takes the small-map K-split kernel wherever it covers the geometry
 *   VUNET_TUNE_BLK_FORCE_NT    tile height (rows per wave: 1 or 2) of the LDS-tiled blocked-bf16 kernel (vunet_conv2d_blk)
 *   VUNET_TUNE_BLK_WS          the LDS-tiled blocked-bf16 kernel: 1 = always its uniform form (every wave stages and multiplies),
 *                              2 = always the wave-specialised form (four staging + four matrix waves); 0 = by layer width;
 *                              3 = the direct kernel with its weights from global memory instead of through LDS
 *   VUNET_TUNE_S2_FWD_F32      1: the stride-2 forward layers on the fp32-input MFMA kernel instead of the fp16 scheme's
 *                              parity-plane kernel (A/B timing, tests)
 *   VUNET_TUNE_WGRAD_ROWSPLIT  the direct weight-gradient kernel's kernel-row split (one kernel row per wave): 0 / 2 = every 3x3 layer
 *                              on maps >= 8 wide, 1 = never, 4 = the stride-2 layers only; 3 = the large stride-2 layers on the
 *                              direct kernel instead of the LDS-staged conv_wgrad_s2_kernel
 *   VUNET_TUNE_PARITY_LAUNCHES 1: the stride-2 data gradient of the fp16 scheme as four launches, one per output parity,
 *                              instead of the fused kernel (A/B timing, tests)
 *   VUNET_TUNE_P2_FORM         vunet_p2_conv: 1 = always the four-wave / 64-channel workgroup, 2 = the eight-wave / 128-channel
 *                              one (wave groups in antiphase) wherever the channel count allows, 3 = that one with all waves in
 *                              lockstep (A/B); 0 = by how many workgroups the problem has
 * Returns VUNET_ERR_ARG for an unknown key. */
#define VUNET_TUNE_SPLIT_FORCE_NT 0
#define VUNET_TUNE_TILED_FORCE_NT 1
#define VUNET_TUNE_FORCE_SMALL 2
#define VUNET_TUNE_BLK_FORCE_NT 3
#define VUNET_TUNE_PARITY_LAUNCHES 4
#define VUNET_TUNE_BLK_WS 5
#define VUNET_TUNE_WGRAD_ROWSPLIT 6
#define VUNET_TUNE_S2_FWD_F32 7
#define VUNET_TUNE_P2_FORM 8
#define VUNET_TUNE_COUNT 9
int vunet_set_tuning(int32_t key, int32_t value);

/* Dropout keep-mask of the conv prologue, materialised (parity tests / debugging only) */
int vunet_dropout_mask(float* mask, int64_t n, float p, uint32_t seed, void* stream);

/* Pose "stickman" rasteriser: lib/utils.py:325-512 (make_joint_img) for thickness-1 LINE_8 cv2.line and
 * cv2.fillPoly of the body polygon, one launch for a batch of frames (replaces the per-frame CPU raster of
 * data/human36m.py:808-848 and the render loop data/data_conversions_3d.py:1130-1185).
 * kps: [B][J][2] float (x, y) in pixels (a joint is valid iff both >= 0; coordinates are truncated like np.int_);
 * body: [n_body] joint ids of the polygon; cmds: [n_cmds][6] int32 {kind, joint a, joint b, joint c, plane, colour},
 * executed in order (later commands overwrite); kind 0 polygon, 1 line a->b, 4 head line (a line whose length
 * feeds the throat length, :434-467), 2 neck line midpoint(a, b)->c of the models without head lines (:407-433),
 * 3 face line a->b drawn only if shorter than the throat length (:468-505); all device pointers.
 * out_u8: [B][3][H][W] uint8 and/or out_f32: the same planes as fp32 (u/255)*2-1; W % 4 == 0.  Integer work,
 * bit-exact. */
int vunet_stickman_raster(const float* kps, int32_t B, int32_t J, const int32_t* body, int32_t n_body,
                          const int32_t* cmds, int32_t n_cmds, uint8_t* out_u8, float* out_f32, int32_t H, int32_t W,
                          void* stream);
/* The same with cv2.line's ``thickness`` (lib/utils.py:334-339: img_shape[1] // scale_factor, the `stickman_scale` of
 * data/base_dataset.py:163-168) handed to every line of the frame: thickness > 1 is OpenCV's ThickLine -- a quad filled by
 * FillConvexPoly in 16.16 fixed point (outline by Line2) plus a filled Circle at both end points -- as per-pixel closed
 * forms; 1 <= thickness <= 79.  The body polygon (cv2.fillPoly) does not depend on it. */
int vunet_stickman_raster_thick(const float* kps, int32_t B, int32_t J, const int32_t* body, int32_t n_body,
                                const int32_t* cmds, int32_t n_cmds, uint8_t* out_u8, float* out_f32, int32_t H, int32_t W,
                                int32_t thickness, void* stream);
/* uint8 planes -> fp32 in [-1,1]  (ToTensor, *2-1; data/base_dataset.py:183-190) */
int vunet_u8_to_unit(const uint8_t* in, float* out, int64_t n, void* stream);

/* SSIM evaluation hook (lib/metrics.py:94-107: skimage structural_similarity with multichannel=True,
 * data_range, gaussian_weights=True, use_sample_covariance=False): x, y [planes][H][W] fp32 (planes = N*C),
 * window11 = the 11 normalised Gaussian taps (sigma 1.5) on the device; partial[planes * ceil(H/8) * ceil(W/32)]
 * receives per-tile sums of the SSIM map over the 5-pixel-cropped interior; the caller sums them in order and
 * divides by planes_per_image * (H-10) * (W-10).  H, W >= 11. */
int vunet_ssim_partial(const float* x, const float* y, int32_t planes, int32_t H, int32_t W, float data_range,
                       const float* window11, float* partial, void* stream);

/* ---- data parallelism over RCCL.  Replaces nn.DataParallel of experiments/shape_and_pose_net.py:213-214,
 * 223-224,230-233: one process per GPU holds a persistent replica; the only exchange of a step is the SUM of the flat
 * gradient buckets.  RCCL is bound at run time (dlopen of librccl.so.1 -- the instance PyTorch mapped, if any); without it
 * these return VUNET_ERR_UNSUPPORTED.  Bootstrap: rank 0 calls vunet_dp_unique_id and hands the 128 bytes to every rank
 * through any side channel (the host mirror uses the torch.distributed store it was launched with); every rank then calls
 * vunet_dp_init(world, rank, id) with its GPU current (collective: returns when all ranks have joined).
 *   vunet_dp_allreduce_bucket  in-place ncclAllReduce(ncclFloat, ncclSum or ncclAvg) of n floats on `stream`: asynchronous,
 *                              ordered like a kernel launch on that stream, capturable in a hipGraph; every rank must issue
 *                              its buckets in the same order.
 *   vunet_dp_world             ranks of the communicator, 0 before init / after finalize.
 *   vunet_dp_finalize          destroys the communicator (idempotent). */
int vunet_dp_available(void);   /* 1 / 0: librccl binds in this process (no GPU call, no communicator): exchange it first */
int vunet_dp_unique_id(void* id128);
int vunet_dp_init(int32_t world, int32_t rank, const void* id128);
int vunet_dp_world(void);
int vunet_dp_allreduce_bucket(float* buf, int64_t n, int32_t average, void* stream);
int vunet_dp_finalize(void);

/* Square window of a batch of maps with its corner read from DEVICE memory (the adversarial term's patch: one random
 * (P x P) window per step, experiments' gan option): y[n][c][i][j] = x[n][c][off[0] + i][off[1] + j], off = int32[2] on the
 * device -- the launch arguments do not change from step to step, so the crop can sit in a captured hipGraph.
 * vunet_crop_window_bwd scatters dy back into a ZEROED dx of x's shape (it writes the window only). */
int vunet_crop_window(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t P, const int32_t* off,
                      void* stream);
int vunet_crop_window_bwd(const float* dy, float* dx, int32_t planes, int32_t H, int32_t W, int32_t P, const int32_t* off,
                          void* stream);

/* The training step's scalar glue, one launch each (every operand a device scalar: nothing here changes between steps, so
 * the launches sit in the captured hipGraph).
 * vunet_total_loss: *ll = likelihood_loss = ll_weight * sum_i terms[i][0]  (n <= VUNET_LOSS_MAX_TERMS perceptual terms; terms is
 *   a HOST array of n device pointers), *loss = likelihood_loss + g * kl[0] when use_kl, else likelihood_loss; g = gamma[0] (device) or gamma_const when
 *   gamma is NULL  (reference experiments/shape_and_pose_net.py:391-405: torch.stack / sum / ll_weight * / + tuning * kl).
 * vunet_total_loss_bwd: d[i] = ll_weight * (g_loss + g_ll) for the n terms, d[n] = g_loss * g (0 without use_kl); g_loss /
 *   g_ll may be NULL (= 0).
 * vunet_gamma_update: gamma[0] <- max(gamma[0] - gamma_step * (imax[0] - avg_kl[0]), 0)   (:82-85, :442). */
#define VUNET_LOSS_MAX_TERMS 8
int vunet_total_loss(const float* const* terms, int32_t n, const float* kl, const float* gamma, float gamma_const,
                     float ll_weight, int32_t use_kl, float* loss, float* ll, void* stream);
int vunet_total_loss_bwd(const float* g_loss, const float* g_ll, int32_t n, const float* gamma, float gamma_const,
                         float ll_weight, int32_t use_kl, float* d, void* stream);
int vunet_gamma_update(float* gamma, const float* imax, const float* avg_kl, float gamma_step, void* stream);

/* z = mu + eps (logstd NULL; reference models/vunets.py:151-156, latent_sample: p + randn_like(p)) or z = eps * exp(logstd) + mu
 * (:594-597, reparametrize) with eps ~ N(0, 1) drawn inside the kernel: Box-Muller on two hashes of (element, seed, step), the
 * step taken from the device counter of vunet_set_dropout_step when one is set.  eps_out (may be NULL) receives the noise
 * (vunet_reparam_bwd needs it).  Without logstd the gradient w.r.t. mu is the identity. */
int vunet_unit_sample(const float* mu, const float* logstd, float* z, float* eps_out, int64_t n, uint32_t seed, void* stream);

/* The host-side schedule values of a step (learning rate as double, information_max, dropout step) written to their device
 * scalars by ONE launch; the values travel as kernel arguments, so the host may run any number of steps ahead.  NULL
 * pointers are skipped. */
int vunet_set_schedule(double* lr_dev, double lr, float* imax_dev, float imax, int32_t* step_dev, int32_t step, void* stream);

/* out = ((g0 + g1) + g2) + g3 over n = 2..4 tensors of numel floats each (srcs: HOST array of device pointers), in that order,
 * and -- amax_out non-NULL: 512 zero-initialised floats -- the partial maxima of |out| (as vunet_conv2d's amax_out).  One launch for
 * what autograd does with n - 1 aten::add_ launches plus the next layer's vunet_absmax_partials pass: the gradients of a tensor
 * with several readers (the bottleneck's hidden state feeds mu, log-sigma and the next block: reference models/vunets.py:560-590). */
int vunet_sum_amax(const float* const* srcs, int32_t n, float* out, float* amax_out, int64_t numel, void* stream);

/* ---- "p2": the frozen VGG19 stack of the perceptual loss on PRE-SPLIT activations (ABI 9; csrc/conv_p2.hip).  Replaces, for
 * models/imagenet_pretrained.py:42-61 run twice per step by lib/losses.py:81-119, the per-workgroup operand conversion of
 * vunet_conv2d's fp16 scheme by ONE conversion per tensor, done by the kernel that produces it.
 *
 * A planes tensor ("P2") of logical shape [N][C][H][W], C % 8 == 0:
 *     fp16  [2 planes][N][C / 8][H + 2][W + 2][8]     plane 0 = hi, plane 1 = lo:  2^e x = hi + lo / 2^11
 *     int32 meta[128]:  [0] e;  [16 .. 79] 64 slots, the largest hold an upper bound of max |x| (fp32 bit patterns)
 * PRECONDITIONS (the caller's): the buffer is zero-filled once -- no kernel writes the one-pixel border, every consumer
 * reads it as the convolution's zero padding; meta[16 .. 79] of an OUTPUT are zero before its producer is launched
 * (the producers atomicMax into them).  Buffers and metas may be reused from step to step.
 * The scale of an output is derived in its producer from a bound (max row sum of |w| x the input's maximum + max |shift|),
 * see csrc/conv_p2.hip for why a loose bound costs no accuracy. */
typedef struct vunet_p2_desc {
  int32_t N, C, H, W;   /* input planes: C % 32 == 0; H % 8 == 0; W == 16 or W % 32 == 0 */
  int32_t M;            /* output channels, M % 64 == 0; the output planes are [N][M][H][W] */
  int32_t relu;         /* 1: ReLU on the output (the forward layers of the stack) */
} vunet_p2_desc;
int vunet_p2_conv_supported(const vunet_p2_desc* d);   /* 1 / 0 */
int vunet_p2_conv_variant(const vunet_p2_desc* d, char* name, int32_t len);   /* kernel instantiation, rocprofv3 spelling */
/* y = [relu]( conv3x3_pad1(x; image) + shift ), or with `mask` (planes shaped like y: the forward activation that feeds the
 * layer below) the data-gradient form  y = conv3x3(x; dgrad image) * [mask != 0].  image / wk from vunet_p2_pack_weights. */
int vunet_p2_conv(const vunet_p2_desc* d, const void* x, const int32_t* xmeta, const void* w_image, const float* wk,
                  const float* shift, const void* mask, void* y, int32_t* ymeta, void* stream);
/* Weight image of a 3x3 layer, w [Cout][Cin][3][3] fp32.  dgrad = 0: rows Cout, sums over Cin (Cout % 16 == 0, Cin % 32 == 0);
 * dgrad = 1: rows Cin, sums over Cout, taps mirrored.  wk[4]: max row sum of |w| (x 1 + 1e-6), max |shift| (dgrad: 0), the
 * weights' scale exponent, max |w|.  workspace: 2 * rows floats.  vunet_p2_weight_image_bytes: bytes of the image, 0 if the
 * geometry is not covered. */
int vunet_p2_weight_image_bytes(int32_t Cout, int32_t Cin, int32_t dgrad);
int vunet_p2_pack_weights(const float* w, const float* shift, int32_t Cout, int32_t Cin, int32_t dgrad, void* image, float* wk,
                          float* workspace, void* stream);
/* The FIRST layer of the stack (3 input channels: no matrix-core work): conv3x3 + shift + ReLU on the fp32 image, written as
 * planes (csrc/conv_thin.hip).  wt_f / Mpad / shift: the layer's K-major weights from vunet_weightnorm_fwd; wk from
 * vunet_p2_weight_bound (max row sum of |w|, max |shift|); amax_x: n_amax partial maxima of |x|.  M % 8 == 0, M <= 128,
 * H % 32 == 0, W % 32 == 0. */
int vunet_p2_weight_bound(const float* w, const float* shift, int32_t Cout, int32_t Cin, float* wk, float* workspace, void* stream);
int vunet_p2_conv_first(const float* x, const float* amax_x, int32_t n_amax, const float* wt_f, int32_t Mpad, const float* shift,
                        const float* wk, void* y, int32_t* ymeta, int32_t N, int32_t H, int32_t W, int32_t M, void* stream);
/* fp32 NCHW <-> planes.  from_nchw: amax = n_amax partial maxima of |x| (vunet_absmax_partials / a producer's amax_out): the
 * scale comes from the true maximum; relu = 1 applies max(x, 0) on the way. */
int vunet_p2_from_nchw(const float* x, const float* amax, int32_t n_amax, int32_t relu, void* y, int32_t* ymeta, int32_t N,
                       int32_t C, int32_t H, int32_t W, void* stream);
int vunet_p2_to_nchw(const void* x, const int32_t* xmeta, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);

/* Pointwise kernels of the p2 pass (csrc/planes.hip), all on planes tensors [N][C][H][W] with their metas:
 *   vunet_p2_l1_fwd    out[0] += weight * mean |t - p|  (lib/losses.py:98-102); partial: 1024 floats of workspace
 *   vunet_p2_pool_fwd  y = maxpool2x2(p) (planes [N][C][H/2][W/2]; inherits p's scale and maximum); with t != NULL also the L1
 *                      term of the tap that feeds the pool (relu1_2, relu2_2), from the same pass over p
 *   vunet_p2_l1_bwd    g = [add] + c sign(p - t), zeroed where p == 0 (p is a ReLU output): the gradient w.r.t. the convolution
 *                      output under the tap; c = gscale * gout[0] (gout: the loss term's upstream gradient, a device scalar, or
 *                      NULL for 1); add (optional): the gradient arriving from the layer above, planes shaped like p
 *   vunet_p2_pool_bwd  g = route(dy) [+ c sign(p - t) with t != NULL], zeroed where p == 0: backward of the pool (first maximum in
 *                      scan order, as ATen) fused with the ReLU backward and, for relu1_2 / relu2_2, the tap's step
 * Output metas: maximum slots zeroed by the caller beforehand; the scale is derived from max|add| (max|dy|) + |c|. */
int vunet_p2_l1_fwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, float* partial, float* out,
                    float weight, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int vunet_p2_pool_fwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, float* partial, float* out,
                      float weight, void* y, int32_t* ymeta, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int vunet_p2_l1_bwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, const void* add,
                    const int32_t* addmeta, void* g, int32_t* gmeta, float gscale, const float* gout, int32_t N, int32_t C,
                    int32_t H, int32_t W, void* stream);
int vunet_p2_pool_bwd(const void* t, const int32_t* tmeta, const void* p, const int32_t* pmeta, const void* dy,
                      const int32_t* dymeta, void* g, int32_t* gmeta, float gscale, const float* gout, int32_t N, int32_t C,
                      int32_t H, int32_t W, void* stream);

/* ---- Behaviour front half of BASELINE config 5 (ABI 10; csrc/seq.hip): flow sample -> pose_behavior_rnn decode.
 * Replaces, per call, the ATen chain behind ``UnsupervisedTransformer2.reverse`` / ``forward``
 * (models/flow/simple_flow.py:136-176, models/flow/blocks.py:95-128, :276-319, :531-559, :692-704, lib/modules.py:236-331) and
 * ``ResidualBehaviorNet.infer_b`` / ``generate_seq`` (models/pose_behavior_rnn.py:125-209, :463-534, :587-626).
 * All tensors fp32.  "Bp" is B rounded up to 16; every [Bp][..] buffer is allocated zero-filled by the caller (rows >= B are
 * computed but never read back).
 *
 *   vunet_seq_linear      y[net][b][m] = act_net(sum_k w_net[m][k] * x_net[b][k] + bias_net[m])   (act 0: none, 1: LeakyReLU(0.01),
 *                         2: tanh).  w: [M][K] row-major, M % 16 == 0, K % 32 == 0 (zero-padded images, vunet_seq_pack_rows);
 *                         x: [nets_in][Bp][ldx] with nets_in = 1 (shared_in != 0: both nets read the same operand) or nets;
 *                         y: [nets][Bp][M].  S > 1 (the flow's 512-row head layers, which would otherwise run on 32 workgroups per
 *                         net): K split over S workgroups per row tile, y = [nets][S][Bp][M] RAW partial slabs -- no bias, no
 *                         activation; vunet_seq_coupling adds them.  K % (32 S) == 0, S <= 8.  B <= 64.  v_mfma_f32_16x16x4_f32: exact fp32 products and sums; a wave adds its K
 *                         chunks in ascending order, the four waves of a workgroup are added in wave order: bit-reproducible.
 *   vunet_seq_coupling    one step between two MLP evaluations of the flow: v = in with the half v[c1..C) replaced by
 *                         (v - t) exp(-s) (reverse) or v exp(s) + t (forward); st: [2][S][Bp][Mp] from vunet_seq_linear (NULL: no
 *                         coupling): S = 1 finished values (act0 = tanh), S > 1 raw slabs added here in slab order, then + bias_s /
 *                         bias_t and tanh for s; out[b][c] = A(v[map[c]]) with A = v / scale - loc (reverse) or
 *                         scale (v + loc) (forward), parameters indexed by map[c] (affine_on_src != 0) or by c (scale NULL: none);
 *                         forward adds sum(s) + sum(log|scale|) into logdet[b].  in != out when map != NULL; rows of in / out
 *                         are ld_in / ld_out floats apart.
 *   vunet_seq_start       first operand row of a recurrence: xh[b] = [x0[b] | 0 | h0[b]], c = c0 (NULL: zeros), xraw[b] = x0[b]
 *                         (NULL: skip); x0 row b at x0 + b * x0_stride.  hoff >= n, ldx >= hoff + H.
 *   vunet_seq_lstm_gates  one LSTM step's gate product AND cell update: w_perm = [W_ih | 0 | W_hh] with GATE-INTERLEAVED rows (row
 *                         4 j + q = gate q of hidden unit j, q = i, f, g, o; vunet_seq_pack_rows with row_mul = 4, row_off = q),
 *                         bias_perm likewise (vunet_seq_lstm_bias); xh: this step's operand rows [Bp][ldx] = [x | 0 | h];
 *                         c' = sig(f) c_in + sig(i) tanh(g) -> c_out, h = sig(o) tanh(c') -> xh_next[b][hoff..] (and h_out, optional):
 *                         xh_next and c_out are OTHER buffers than xh and c_in (every workgroup reads those), swapped by the caller
 *                         per step.  x_next != NULL (the encoder): xh_next[b][0..n) = x_next[b] (row at + b * seq_stride).
 *   vunet_seq_decoder_out the decoder's second launch of a step: x' = w_out h + b_out + xraw[b] with h = xh[b][hoff..]; xs[b] = x',
 *                         cs[b] = the old xraw[b] (rows at + b * seq_stride); xraw[b] = x'; xh[b][0..n) = x'.
 *   vunet_seq_lstm_bias   bias_perm[4 j + q] = b_ih[q H + j] + b_hh[q H + j] (+ fold[q H + j], the folded input layer's W_ih b_in).
 *   vunet_seq_fold_input  ``linear_in_decoder`` (models/pose_behavior_rnn.py:494-495) folded into the gate matrix:
 *                         w_fold [M][n] = w_ih [M][n] . w_in [n][n], bias_fold [M] = w_ih . b_in.
 *   vunet_seq_bottleneck  heads: [2][Bp][Mp] = (mu, logstd) from vunet_seq_linear with nets = 2; copies them out and forms
 *                         b = eps exp(logstd) + mu (eps NULL: mu; b_out NULL: skip)   (models/pose_behavior_rnn.py:180-201).
 *   vunet_seq_pack_rows   dst[row_off + row_mul m][col_off + k] = src[m][k] * (row_scale ? row_scale[m] : 1): builds the zero-padded
 *                         weight images (row_off 0, row_mul 1: plain).
 *   vunet_seq_normlinear_rows   NormConv2d 1x1 as a linear layer (lib/modules.py:135-145): row_scale[m] = gamma g / ||v_m||,
 *                         bias_eff[m] = gamma bias + beta.
 *   vunet_seq_pose_project      decoded pose vectors -> pixel keypoints [T][J][2]: unNormalizeData (data/data_conversions_3d.py:178-211;
 *                         dims_to_use ascending, the others hold mean), [R | t] (:588-605), pinhole projection (:892-912) and the
 *                         rescale to the synthesis resolution (:1139-1140), in float64 like the reference's numpy.  cam: 18 DEVICE
 *                         doubles = [R | t] row-major (12), fx, x0, fy, y0, scale_x, scale_y; mean / stdv: D device doubles, D >= 3 J;
 *                         f32_math != 0: x * std + mean in float32 (numpy's result when the statistics are float32 arrays).
 *   vunet_seq_actnorm_init      ActNorm's data-dependent initialisation (lib/modules.py:270-290): loc = -mean, scale = 1 / (std + 1e-6)
 *                         per channel over the B rows of x (unbiased std); B >= 2. */
typedef struct vunet_seq_linear_desc {
  int32_t B, M, K, ldx, act0, act1, nets, shared_in, S;
} vunet_seq_linear_desc;
typedef struct vunet_seq_coupling_desc {
  int32_t B, C, c1, ld_in, ld_out, Mp, reverse, affine_on_src, S;
} vunet_seq_coupling_desc;
typedef struct vunet_seq_lstm_desc {
  int32_t B, H, ldx, hoff, n, ldraw;
  int64_t seq_stride;
} vunet_seq_lstm_desc;
int vunet_seq_linear(const vunet_seq_linear_desc* d, const float* w0, const float* w1, const float* x, const float* bias0,
                     const float* bias1, float* y, void* stream);
int vunet_seq_coupling(const vunet_seq_coupling_desc* d, const float* in, const float* st, const float* bias_s, const float* bias_t,
                       const int32_t* map, const float* scale, const float* loc, float* out, float* logdet, void* stream);
int vunet_seq_start(const float* x0, int64_t x0_stride, const float* h0, const float* c0, float* xraw, int32_t ldraw, float* xh,
                    int32_t ldx, int32_t hoff, float* c, int32_t B, int32_t n, int32_t H, void* stream);
int vunet_seq_lstm_gates(const vunet_seq_lstm_desc* d, const float* w_perm, const float* xh, const float* bias_perm, const float* c_in,
                         float* c_out, float* xh_next, float* h_out, const float* x_next, void* stream);
int vunet_seq_decoder_out(const vunet_seq_lstm_desc* d, float* xh, const float* w_out, const float* b_out, float* xraw, float* xs,
                          float* cs, void* stream);
int vunet_seq_lstm_bias(const float* b_ih, const float* b_hh, const float* fold, int32_t H, float* bias_perm, void* stream);
int vunet_seq_fold_input(const float* w_ih, const float* w_in, const float* b_in, int32_t M, int32_t n, float* w_fold,
                         float* bias_fold, void* stream);
int vunet_seq_bottleneck(const float* heads, int32_t Mp, const float* eps, float* mu, float* logstd, float* b_out, int32_t B,
                         int32_t H, void* stream);
int vunet_seq_pack_rows(const float* src, int32_t M, int32_t K, const float* row_scale, float* dst, int32_t ld_dst, int32_t col_off,
                        int32_t row_off, int32_t row_mul, void* stream);
int vunet_seq_pose_project(const float* x, int32_t n_use, const int32_t* dims_to_use, const double* mean, const double* stdv, int32_t D,
                           int32_t f32_math, const double* cam, float* kps, int32_t T, int32_t J, void* stream);
int vunet_seq_actnorm_init(const float* x, int32_t ld, int32_t B, int32_t C, float* loc, float* scale, void* stream);
int vunet_seq_normlinear_rows(const float* v, const float* g, const float* bias, const float* gamma, const float* beta, int32_t M,
                              int32_t K, float* row_scale, float* bias_eff, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VUNET_HIP_H */
