// Instantiations of the split-fp16 3x3 convolution for MT = 2, four waves (split per MT / wave count to compile in parallel).
#include "conv_h2_kernel.h"

int vunet_conv_h2_launch_mt2(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, int NT,
                             hipStream_t st) {
  if (NT == 2) return launch_h2<2, 2>(ga, wx, mtiles_pad, amax, pro, st);
  return launch_h2<2, 1>(ga, wx, mtiles_pad, amax, pro, st);
}
