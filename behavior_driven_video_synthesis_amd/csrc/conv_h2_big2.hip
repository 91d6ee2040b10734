// Instantiations of the split-fp16 3x3 convolution with the big workgroup tile of 64 channels x 16 rows (MT 2, NT 4).
#include "conv_h2_kernel.h"

int vunet_conv_h2_launch_big2(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, hipStream_t st) {
  return launch_h2_big<2, 4>(ga, wx, mtiles_pad, amax, pro, st);
}
