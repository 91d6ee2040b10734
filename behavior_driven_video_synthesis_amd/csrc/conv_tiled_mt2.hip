// Instantiations of the LDS-tiled 3x3 convolution for MT = 2 (split per MT to compile in parallel).
#include "conv_tiled_kernel.h"

int vunet_conv_tiled_launch_mt2(const GatherArgs& ga, int pro, int NT, int TW, hipStream_t st) {
  if (ga.d.stride == 2) return launch_tiled<2, 1, 8, 32, 2>(ga, pro, st);
  if (TW == 16) return launch_tiled<2, 1, 8, 16>(ga, pro, st);
  if (NT == 4) return launch_tiled<2, 4, 4>(ga, pro, st);
  if (NT == 2) return launch_tiled<2, 2, 8>(ga, pro, st);
  return launch_tiled<2, 1, 8>(ga, pro, st);
}
