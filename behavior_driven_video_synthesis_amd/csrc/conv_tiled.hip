// Host-side dispatch of the LDS-tiled 3x3 / stride-1 convolution (kernel: conv_tiled_kernel.h).
#include "conv_common.h"

int vunet_conv_tiled_launch_mt1(const GatherArgs& ga, int pro, int NT, int TW, hipStream_t st);
int vunet_conv_tiled_launch_mt2(const GatherArgs& ga, int pro, int NT, int TW, hipStream_t st);

bool vunet_conv_tiled_applicable(const vunet_conv_desc* d) {
  if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->C1 % 8 != 0 || d->C2 % 8 != 0) return false;
  if (d->stride == 2)  // Downsample conv, forward only: 4 x 32 output tiles
    return d->mode == 0 && d->Hs == 2 * d->Ho && d->Ws == 2 * d->Wo && d->Wo % 32 == 0 && d->Ho % 4 == 0;
  return d->stride == 1 && d->Hs == d->Ho && d->Ws == d->Wo && (d->Ws % 32 == 0 || d->Ws % 16 == 0) &&
         d->Hs % (d->Ws % 32 == 0 ? 4 : 8) == 0;
}

static int tile_width(const vunet_conv_desc& d) { return d.Wo % 32 == 0 ? 32 : 16; }

static long tiled_blocks(const vunet_conv_desc& d, int MT, int NT) {
  const int TW = tile_width(d), TH = 4 * NT * (32 / TW);
  if (d.Ho % TH != 0 || ((TW == 16 || d.stride == 2) && NT != 1)) return 0;
  return (long)d.N * (d.Ho / TH) * (d.Wo / TW) * ((d.M + 32 * MT - 1) / (32 * MT));
}

// Tile height, from measurements on MI355X (tools/profile_layers.py, tools/bench_conv.py):
//  * plain layers (the VGG19 stack) run best with the tallest tile that still leaves >= 2 workgroups per CU
//    (129 vs 113 TF/s on 256@64^2 and 512@32^2);
//  * layers that do per-element VALU work around the MFMA loop -- the ELU/dropout prologue at staging, or the
//    act'(aux) epilogue of the data gradient -- and the narrow (<= 32 output channel) layers choose between the
//    4-row and the 8-row tile by a cost model: a launch runs ceil(blocks / (CUs * resident workgroups)) rounds and a
//    partly filled last round costs a whole one (128@64^2, batch 16: 1024 four-row tiles on 768 slots = 2 rounds,
//    63 TF/s; 512 eight-row tiles on 512 slots = 1 round, 91 TF/s), times a measured per-tile efficiency.
// Returns 0 if even the smallest tile cannot give half the chip one workgroup each.
static int resident_workgroups(int MT, int NT) {  // per CU: min(VGPR, LDS) limits of the instantiations
  if (MT == 1) return NT == 1 ? 3 : NT == 2 ? 4 : 2;
  return NT == 1 ? 3 : 2;
}

int vunet_conv_tiled_pick(const vunet_conv_desc* d, int* MT, bool valu_heavy) {
  *MT = d->M <= 32 ? 1 : 2;
  if (const int NT = g_vunet_tune[VUNET_TUNE_TILED_FORCE_NT]) {  // tests / tuning: force a tile height
    if ((NT == 1 || NT == 2 || NT == 4) && tiled_blocks(*d, *MT, NT) > 0) return NT;
  }
  if (!valu_heavy && *MT == 2) {
    for (int NT = 4; NT >= 2; NT >>= 1)
      if (tiled_blocks(*d, *MT, NT) >= 512) return NT;
    return tiled_blocks(*d, *MT, 1) >= 128 ? 1 : 0;
  }
  const int nch = (d->C1 + d->C2) / 8;  // K chunks: the taller tile gains with a longer K loop
  int best = 0;
  double best_cost = 0.0;
  for (int NT = 1; NT <= 2; ++NT) {
    const long blocks = tiled_blocks(*d, *MT, NT);
    if (blocks < 128) continue;
    const int occ = resident_workgroups(*MT, NT);
    const long slots = 256L * occ;
    const long rounds = (blocks + slots - 1) / slots;
    double eff = 1.0;
    if (NT == 2) {
      eff = d->mode == 1 ? 0.88 : (*MT == 1 ? 1.12 : 0.93);
      if (nch >= 16) eff *= 1.08;
    }
    const double cost = (double)rounds * occ * NT / eff;
    if (best == 0 || cost < best_cost) {
      best = NT;
      best_cost = cost;
    }
  }
  return best;
}

int vunet_conv_tiled_launch(const GatherArgs& ga, int pro, hipStream_t st) {
  int MT;
  const int NT = vunet_conv_tiled_pick(&ga.d, &MT, (pro != 0 && pro != 4) || (ga.d.mode == 1 && ga.aux != nullptr));
  if (pro == 3) return VUNET_ERR_UNSUPPORTED;
  const int TW = tile_width(ga.d);
  return MT == 1 ? vunet_conv_tiled_launch_mt1(ga, pro, NT, TW, st) : vunet_conv_tiled_launch_mt2(ga, pro, NT, TW, st);
}

// kernel the tiled path would launch, in rocprofv3's spelling (bench.py matches it against the kernel trace)
int vunet_conv_tiled_name(const vunet_conv_desc* d, int pro, bool has_aux, char* name, int len) {
  int MT;
  const int NT = vunet_conv_tiled_pick(d, &MT, (pro != 0 && pro != 4) || (d->mode == 1 && has_aux));
  const int TW = tile_width(*d);
  const int nt = TW == 16 ? 1 : NT;
  const int CK = (MT == 2 && nt == 4) ? 4 : 8;
  return snprintf(name, len, "conv_tiled_kernel<%d, %d, %d, %d, %d, %d, %d>", MT, d->stride == 2 ? 1 : nt, CK, d->mode,
                  d->mode == 1 ? (pro == 4 ? 4 : 0) : pro, TW, d->stride);
}
