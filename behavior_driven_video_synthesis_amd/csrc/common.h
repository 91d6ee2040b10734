// Shared device/host helpers for the gfx950 (CDNA4) VUnet kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "../../include/vunet_hip.h"

// ---- activation codes shared with include/vunet_hip.h
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_LRELU = 4 };

// Stateless dropout hash: keep(idx) <=> hash(idx + seed) >= thresh.  The CPU side of the parity
// tests reproduces this function bit for bit (tests/hip_parity_utils.py: dropout_keep_mask).
__host__ __device__ __forceinline__ uint32_t vunet_hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : (__expf(x) - 1.f); }
__device__ __forceinline__ float elu_grad_f(float x) { return x > 0.f ? 1.f : __expf(x); }

// input-side activation (+ optional dropout keep/scale) used as the conv prologue
struct InAct {
  int act;            // ACT_NONE / ACT_ELU / ACT_RELU / ACT_LRELU
  uint32_t thresh;    // dropout: keep iff hash >= thresh ; 0 = no dropout
  uint32_t seed;
  float keep_scale;   // 1/(1-p)
  float slope;        // leaky-relu slope
  // optional device-resident step counter (vunet_set_dropout_step): the mask of step t uses seed + t * DROP_STEP_MUL, so
  // a launch whose arguments are frozen in a captured hipGraph still draws a fresh mask on every replay
  const uint32_t* step;
};

#define VUNET_DROP_STEP_MUL 0x9E3779B1u

__device__ __forceinline__ uint32_t inact_seed(const InAct& a) {
  return a.step ? a.seed + (*a.step) * VUNET_DROP_STEP_MUL : a.seed;
}

// kernels resolve the step counter ONCE, at entry, into their own copy of the launch arguments (one scalar load before
// any store; afterwards `step` is null and the hash calls below touch no memory)
__device__ __forceinline__ void inact_resolve(InAct& a) {
  a.seed = inact_seed(a);
  a.step = nullptr;
}

__device__ __forceinline__ float apply_in_act(const InAct& a, float v, uint32_t idx) {
  if (a.act == ACT_ELU) v = elu_f(v);
  else if (a.act == ACT_RELU) v = v > 0.f ? v : 0.f;
  else if (a.act == ACT_LRELU) v = v > 0.f ? v : v * a.slope;
  if (a.thresh) v = (vunet_hash_u32(idx + inact_seed(a)) >= a.thresh) ? v * a.keep_scale : 0.f;
  return v;
}

// The same prologue as STRAIGHT-LINE code for the forms the models use (FORM: 0 none, 1 ELU, 2 ELU + dropout, 3 whatever the
// descriptor says = apply_in_act).  apply_in_act branches on the activation code and the dropout threshold per ELEMENT; a
// staging loop that calls it is dominated by scalar branches (conv_wgrad_h2.hip, round 4: 358 of them, 3.0 - 3.4 us per
// tile against 2.6 us for the tile's MFMAs).  Kernels pick the form once per launch (in_act_form_of, wave-uniform) and
// instantiate their staging code per form behind that one branch.  The argument must have been resolved (inact_resolve).
template <int FORM>
__device__ __forceinline__ float in_act_form(const InAct& a, float v, uint32_t idx) {
  if constexpr (FORM == 0) return v;
  else if constexpr (FORM == 1) return elu_f(v);
  else if constexpr (FORM == 2) {
    v = elu_f(v);
    return (vunet_hash_u32(idx + a.seed) >= a.thresh) ? v * a.keep_scale : 0.f;
  } else return apply_in_act(a, v, idx);
}
__device__ __forceinline__ int in_act_form_of(const InAct& a) {
  if (a.thresh) return a.act == ACT_ELU ? 2 : 3;
  return a.act == ACT_NONE ? 0 : (a.act == ACT_ELU ? 1 : 3);
}

// d/dv of apply_in_act evaluated at the pre-activation value v
__device__ __forceinline__ float in_act_grad(const InAct& a, float v, uint32_t idx) {
  float g = 1.f;
  if (a.act == ACT_ELU) g = elu_grad_f(v);
  else if (a.act == ACT_RELU) g = v > 0.f ? 1.f : 0.f;
  else if (a.act == ACT_LRELU) g = v > 0.f ? 1.f : a.slope;
  if (a.thresh) g = (vunet_hash_u32(idx + inact_seed(a)) >= a.thresh) ? g * a.keep_scale : 0.f;
  return g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective block remap (blocks b and b+8 share an XCD under round-robin dispatch):
// gives each XCD a contiguous chunk of the logical grid so neighbouring tiles hit one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, idx = bid >> 3;
  const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// process-wide: device pointer to the dropout step counter, or nullptr (pointwise.hip: vunet_set_dropout_step)
extern const uint32_t* g_vunet_drop_step;

static inline InAct make_inact(int act, float slope, float p, uint32_t seed) {
  InAct a;
  a.act = act;
  a.slope = slope;
  a.seed = seed;
  a.step = p > 0.f ? g_vunet_drop_step : nullptr;
  if (p > 0.f) {
    double t = (double)p * 4294967296.0;
    a.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    if (a.thresh == 0) a.thresh = 1;
    a.keep_scale = 1.f / (1.f - p);
  } else {
    a.thresh = 0;
    a.keep_scale = 1.f;
  }
  return a;
}

// Process-start switches (VUNET_NO_*): read once, not per launch.
#define VUNET_ENV_FLAG(fn, name)                              \
  static inline bool fn() {                                   \
    static const bool on = getenv(name) != nullptr;           \
    return on;                                                \
  }

// process-wide tuning knobs (vunet_set_tuning, pointwise.hip): plain loads on the launch path
extern int g_vunet_tune[16];

static inline int vunet_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VUNET_OK : VUNET_ERR_LAUNCH;
}

// hipGetLastError() also reports stale, unrelated errors left by other users of the runtime in this
// process (PyTorch does not always clear them): clear before every launch, then check.
#define VUNET_LAUNCH(...)          \
  do {                             \
    (void)hipGetLastError();       \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)
