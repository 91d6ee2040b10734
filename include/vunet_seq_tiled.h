/*
 * vunet_seq_tiled.h -- C ABI of the tile-major form of the behaviour path's MLP layers (csrc/seq.hip; part of libvunet_hip.so,
 * conventions of vunet_hip.h).  It replaces the same reference operations as vunet_seq_linear: the Linear + LeakyReLU / Tanh
 * stacks of lib/modules.py:236-257 inside models/flow/blocks.py:276-319.
 *
 * vunet_seq_linear reads a [16 rows][32 k] chunk of W as sixteen 128-byte row pieces, 2048 floats apart, and a lane's two 16-byte
 * loads of a chunk sit 16 bytes apart with 16-byte gaps between lanes.  In TILE-MAJOR order a chunk is 2 KB contiguous and every
 * wave instruction loads 1 KB contiguous:
 *
 *   weights   wt[M / 16][K / 32][2][64][4]   wt[mt][c][h][l][e] = w[16 mt + (l & 15)][32 c + 8 (l >> 4) + 4 h + e]
 *   operand   xt[Bp / 16][K / 32][2][64][4]  xt[nb][c][h][l][e] = x[16 nb + (l & 15)][32 c + 8 (l >> 4) + 4 h + e]
 *
 * (lane l of a wave holds row l & 15 and the eight k of slot l >> 4 of a chunk: the operand registers of the kernel's eight
 * v_mfma_f32_16x16x4_f32 per chunk, loaded as they are used).  The arithmetic and its order are those of vunet_seq_linear: results
 * are bit-identical.
 *
 *   vunet_seq_pack_tiles     w [M][ld] row-major (M % 16 == 0, K % 32 == 0, ld % 4 == 0) -> wt
 *   vunet_seq_linear_tiled   vunet_seq_linear with `layout` bits: 1 = the weights are wt images, 2 = the operand is xt (one per
 *                            net or shared, as x; ldx is ignored), 4 = the output is written as the NEXT layer's xt
 *                            ([nets][Bp / 16][M / 32][2][64][4]; M % 32 == 0, S == 1).  Layer 0 of an MLP of the flow reads the
 *                            row-major state rows and writes tiles (1 | 4), hidden layers 1 | 2 | 4, the head layer 1 | 2.
 *                            Training reads the parameters themselves (row-major, updated every step: no bit 1) and still hands the
 *                            activations on tile-major (4, 2 | 4, 2) -- at 64 batch rows the operand is four times the weight
 *                            bytes of a workgroup; y_rowmajor != NULL (with bit 4): the output is ALSO written row-major
 *                            [nets][Bp][M], what the backward kernels of vunet_seq_train.h read.
 */
#ifndef VUNET_SEQ_TILED_H
#define VUNET_SEQ_TILED_H
#include "vunet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VUNET_SEQ_TILED_W 1
#define VUNET_SEQ_TILED_X 2
#define VUNET_SEQ_TILED_Y 4

int vunet_seq_pack_tiles(const float* w, int32_t ld, int32_t M, int32_t K, float* wt, void* stream);
/* vunet_seq_lstm_gates / _train (vunet_hip.h, vunet_seq_train.h) on a tile-major copy of the gate image [W_ih | 0 | W_hh]
 * (vunet_seq_pack_tiles of the gate-interleaved image; ldx % 32 == 0).  gates_out may be NULL. */
int vunet_seq_lstm_gates_tiled(const vunet_seq_lstm_desc* d, const float* w_tiles, const float* xh, const float* bias_perm,
                               const float* c_in, float* c_out, float* xh_next, float* h_out, const float* x_next, float* gates_out,
                               void* stream);
/* The same step with the h part of the operand rows ALSO handed from step to step as tiles, ht[Bp / 16][H / 32][2][64][4] (the
 * layout of xt above over the H columns of h; H % 32 == 0, hoff % 32 == 0, ldx == hoff + H): the step reads x (and the padding)
 * from the rows `xh` and h from `h_tiles_in` (NULL: from the rows as well -- the first step of a sequence, whose rows
 * vunet_seq_start wrote; batches of <= 32 rows always read the rows), and writes its h to `xh_next` AND to `h_tiles_out` (NULL: not
 * wanted).  At 64 rows the operand is four times the gate image's bytes per workgroup, and a row-major chunk is two half-used
 * 128-byte lines per row.  Bit-identical values. */
int vunet_seq_lstm_gates_tiled_h(const vunet_seq_lstm_desc* d, const float* w_tiles, const float* xh, const float* h_tiles_in,
                                 const float* bias_perm, const float* c_in, float* c_out, float* xh_next, float* h_tiles_out,
                                 float* h_out, const float* x_next, float* gates_out, void* stream);
int vunet_seq_linear_tiled(const vunet_seq_linear_desc* d, int32_t layout, const float* w0, const float* w1, const float* x,
                           const float* bias0, const float* bias1, float* y, float* y_rowmajor, void* stream);

#ifdef __cplusplus
}
#endif
#endif
