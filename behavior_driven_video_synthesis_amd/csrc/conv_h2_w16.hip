// Instantiations of the split-fp16 3x3 convolution for 16-wide maps (a column tile = 16 pixels of two rows).
#include "conv_h2_kernel.h"

int vunet_conv_h2_launch_w16(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, int MT,
                             hipStream_t st) {
  return MT == 1 ? launch_h2_w16<1>(ga, wx, mtiles_pad, amax, pro, st) : launch_h2_w16<2>(ga, wx, mtiles_pad, amax, pro, st);
}
