// Instantiation of the split-fp16 kernel for the data gradient of the stride-2 (Downsample) layers with all four output
// parities in one launch (conv_h2_kernel.h: PHW == 4); one m-tile (32 dx channels) per workgroup, any number of them.
#include "conv_h2_kernel.h"

int vunet_conv_h2_launch_par(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, hipStream_t st) {
  return launch_h2_one<1, 1, 1, 0, 4, 4>(ga, wx, mtiles_pad, amax, st);
}
