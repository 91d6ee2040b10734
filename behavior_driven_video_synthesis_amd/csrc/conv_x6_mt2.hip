// Instantiations of the split-bf16 3x3 convolution for MT = 2 (split per MT to compile in parallel).
#include "conv_x6_kernel.h"

int vunet_conv_x6_launch_mt2(const GatherArgs& ga, const void* wx, int mtiles_pad, int pro, int NT, hipStream_t st) {
  if (NT == 2) return launch_x6<2, 2>(ga, wx, mtiles_pad, pro, st);
  return launch_x6<2, 1>(ga, wx, mtiles_pad, pro, st);
}
