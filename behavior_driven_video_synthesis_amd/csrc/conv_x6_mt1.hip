// Instantiations of the split-bf16 3x3 convolution for MT = 1 (split per MT to compile in parallel).
#include "conv_x6_kernel.h"

int vunet_conv_x6_launch_mt1(const GatherArgs& ga, const void* wx, int mtiles_pad, int pro, int NT, hipStream_t st) {
  if (NT == 4) return launch_x6<1, 4>(ga, wx, mtiles_pad, pro, st);
  if (NT == 2) return launch_x6<1, 2>(ga, wx, mtiles_pad, pro, st);
  return launch_x6<1, 1>(ga, wx, mtiles_pad, pro, st);
}
