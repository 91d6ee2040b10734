// Instantiations of the split-fp16 3x3 convolution with the big workgroup tile of 128 channels x 8 rows (MT 4, NT 2):
// one workgroup per CU, one wave per SIMD.  (The 64-channel x 16-row form is in conv_h2_big2.hip: compiled in parallel.)
#include "conv_h2_kernel.h"

int vunet_conv_h2_launch_big2(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, hipStream_t st);

int vunet_conv_h2_launch_big(const GatherArgs& ga, const void* wx, int mtiles_pad, const float* amax, int pro, int MT,
                             hipStream_t st) {
  if (MT == 4) return launch_h2_big<4, 2>(ga, wx, mtiles_pad, amax, pro, st);
  return vunet_conv_h2_launch_big2(ga, wx, mtiles_pad, amax, pro, st);
}
