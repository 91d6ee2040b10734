// Host-side dispatch of the LDS-tiled 3x3 / stride-1 convolution (kernel: conv_tiled_kernel.h).
#include "conv_common.h"

int vunet_conv_tiled_launch_mt1(const GatherArgs& ga, int pro, int NT, hipStream_t st);
int vunet_conv_tiled_launch_mt2(const GatherArgs& ga, int pro, int NT, hipStream_t st);

bool vunet_conv_tiled_applicable(const vunet_conv_desc* d) {
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Hs == d->Ho && d->Ws == d->Wo &&
         d->Ws % 32 == 0 && d->Hs % 4 == 0 && d->C1 % 8 == 0 && d->C2 % 8 == 0;
}

static long tiled_blocks(const vunet_conv_desc& d, int MT, int NT) {
  if (d.Hs % (4 * NT) != 0) return 0;
  return (long)d.N * (d.Hs / (4 * NT)) * (d.Ws / 32) * ((d.M + 32 * MT - 1) / (32 * MT));
}

// largest tile that still gives >= 2 workgroups per CU; 0 if even the smallest tile cannot fill half the chip
int vunet_conv_tiled_pick(const vunet_conv_desc* d, int* MT) {
  *MT = d->M <= 32 ? 1 : 2;
  if (const char* f = getenv("VUNET_TILED_FORCE_NT")) {  // tests: force a tile height on small tensors
    const int NT = atoi(f);
    if ((NT == 1 || NT == 2 || NT == 4) && tiled_blocks(*d, *MT, NT) > 0) return NT;
  }
  for (int NT = 4; NT >= 1; NT >>= 1)
    if (tiled_blocks(*d, *MT, NT) >= 512) return NT;
  return tiled_blocks(*d, *MT, 1) >= 128 ? 1 : 0;
}

int vunet_conv_tiled_launch(const GatherArgs& ga, int pro, hipStream_t st) {
  int MT;
  const int NT = vunet_conv_tiled_pick(&ga.d, &MT);
  if (pro == 3) return VUNET_ERR_UNSUPPORTED;
  return MT == 1 ? vunet_conv_tiled_launch_mt1(ga, pro, NT, st) : vunet_conv_tiled_launch_mt2(ga, pro, NT, st);
}
