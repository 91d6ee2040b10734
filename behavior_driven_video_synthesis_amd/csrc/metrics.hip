// SSIM of Wang et al. as the reference's evaluation hook computes it (lib/metrics.py:94-107:
// skimage.metrics.structural_similarity(multichannel=True, data_range=1.0, gaussian_weights=True,
// use_sample_covariance=False)): 11-tap Gaussian window (sigma 1.5, truncate 3.5), scipy "reflect" borders, the
// SSIM map averaged over the interior (5-pixel crop) of every channel plane.
//
// One workgroup owns an 8 x 32 tile of one plane: the two images' (8+10) x (32+10) neighbourhoods go to LDS once,
// the five windowed moments (x, y, x^2, y^2, xy) are filtered separably there (rows, then columns), the SSIM
// formula runs on registers, and the tile's sum over the cropped area leaves as ONE partial (summed by the caller in a
// fixed order -- deterministic, no atomics).  HBM-bound: each input element is read ~1.7 times (halo).
#include "common.h"

namespace {
constexpr int R = 5, TWS = 32, THS = 8, IWS = TWS + 2 * R, IHS = THS + 2 * R;

__device__ __forceinline__ int reflect(int i, int n) {   // scipy.ndimage mode="reflect": d c b a | a b c d | d c b a
  while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - i - 1;
  return i;
}

__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ xa, const float* __restrict__ ya,
                                                           int H, int W, float c1, float c2, const float* __restrict__ g,
                                                           float* __restrict__ partial) {
  __shared__ float xs[IHS][IWS], ys[IHS][IWS];
  __shared__ float hx[IHS][TWS], hy[IHS][TWS], hxx[IHS][TWS], hyy[IHS][TWS], hxy[IHS][TWS];
  __shared__ float red[4];
  const int tiles_w = (W + TWS - 1) / TWS, tiles_h = (H + THS - 1) / THS;
  const int tx = blockIdx.x % tiles_w, ty = (blockIdx.x / tiles_w) % tiles_h, plane = blockIdx.x / (tiles_w * tiles_h);
  const int row0 = ty * THS, col0 = tx * TWS, tid = threadIdx.x;
  const float* __restrict__ x = xa + (size_t)plane * H * W;
  const float* __restrict__ y = ya + (size_t)plane * H * W;
  float w[2 * R + 1];
#pragma unroll
  for (int k = 0; k <= 2 * R; ++k) w[k] = g[k];
  for (int e = tid; e < IHS * IWS; e += 256) {
    const int r = e / IWS, c = e - r * IWS;
    const int ih = reflect(row0 - R + r, H), iw = reflect(col0 - R + c, W);
    xs[r][c] = x[(size_t)ih * W + iw];
    ys[r][c] = y[(size_t)ih * W + iw];
  }
  __syncthreads();
  for (int e = tid; e < IHS * TWS; e += 256) {   // horizontal pass
    const int r = e / TWS, c = e - r * TWS;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) {
      const float a = xs[r][c + k], b = ys[r][c + k], wk = w[k];
      sx = fmaf(wk, a, sx);
      sy = fmaf(wk, b, sy);
      sxx = fmaf(wk, a * a, sxx);
      syy = fmaf(wk, b * b, syy);
      sxy = fmaf(wk, a * b, sxy);
    }
    hx[r][c] = sx; hy[r][c] = sy; hxx[r][c] = sxx; hyy[r][c] = syy; hxy[r][c] = sxy;
  }
  __syncthreads();
  const int r = tid >> 5, c = tid & 31;          // vertical pass: one output pixel per thread
  float ux = 0.f, uy = 0.f, uxx = 0.f, uyy = 0.f, uxy = 0.f;
#pragma unroll
  for (int k = 0; k <= 2 * R; ++k) {
    const float wk = w[k];
    ux = fmaf(wk, hx[r + k][c], ux);
    uy = fmaf(wk, hy[r + k][c], uy);
    uxx = fmaf(wk, hxx[r + k][c], uxx);
    uyy = fmaf(wk, hyy[r + k][c], uyy);
    uxy = fmaf(wk, hxy[r + k][c], uxy);
  }
  const int oh = row0 + r, ow = col0 + c;
  float s = 0.f;
  if (oh >= R && oh < H - R && ow >= R && ow < W - R) {   // the interior skimage averages over
    const float vx = uxx - ux * ux, vy = uyy - uy * uy, vxy = uxy - ux * uy;
    s = ((2.f * ux * uy + c1) * (2.f * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
}  // namespace

extern "C" int vunet_ssim_partial(const float* x, const float* y, int32_t planes, int32_t H, int32_t W, float data_range,
                                  const float* window11, float* partial, void* stream) {
  if (!x || !y || !window11 || !partial || planes <= 0 || H < 2 * R + 1 || W < 2 * R + 1) return VUNET_ERR_ARG;
  const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
  const long blocks = (long)planes * ((H + THS - 1) / THS) * ((W + TWS - 1) / TWS);
  if (blocks >= (1l << 31)) return VUNET_ERR_UNSUPPORTED;
  VUNET_LAUNCH(ssim_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, H, W, c1, c2, window11,
               partial);
  return vunet_check_launch();
}
