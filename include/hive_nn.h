/*
 * hive_nn.h -- C ABI of the hand-written MFMA kernels behind the batched leaf evaluator
 * (libhive_hip.so).  Replaces the 3x3 convolutions of the reference network's forward
 * (alpha_zero/alpha_net.py:25-54: ConvBlock.conv1 and ResBlock.conv1/conv2 with their
 * BatchNorm, ReLU and skip connection folded/fused) when it is evaluated on leaf batches
 * (woker/api_hive.py:47-74, alpha_zero/MCTS_chess.py:136-142).
 */
#ifndef HIVE_NN_H
#define HIVE_NN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y[b][pixel][k] = act( sum_{tap,c} x[b][pixel+tap][c] * w[tap][k][c] + bias[k] (+ residual[b][pixel][k]) )
 *   x        bf16 [batch][144][cin]      (12x12 board, channels-last; cin = 56 or 256)
 *   w        bf16 [9][cinp/32][16][64][8] fragment-major: tap (dy*3+dx), 32-deep k-step, 16-channel M tile,
 *                                        MFMA lane (= (c%32)/8 * 16 + k%16), 8 consecutive input channels;
 *                                        cinp = cin rounded up to 64, zero padded
 *                                        (hive-alphazero_amd/alpha_net.py::_frag_major builds it)
 *   bias     f32  [256]                  (BatchNorm folded)
 *   residual bf16 [batch][144][256] or NULL
 *   y        bf16 [batch][144][256]
 * fp32 accumulation on the MFMA units, one rounding to bf16 at the end.  Zero padding at the board edge. */
int hive_nn_conv3x3(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                    int batch, int relu, void *stream);

/* One residual block (alpha_net.py:36-54) in one launch: y = relu(conv2(relu(conv1(x) + b1)) + b2 + x), all
 * tensors bf16 [batch][144][256], weights fragment-major as above, y must not alias x.  The intermediate
 * activation stays in LDS. */
int hive_nn_resblock(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                     int batch, void *stream);

/* The same two entry points for either 16-bit format of the matrix cores: dtype = HIVE_BF16 or HIVE_F16
 * (include/hive_abi.h HiveDType); x, w, residual, y are then all of that type.  fp16 carries three more mantissa bits
 * than bf16 at the same MFMA rate (the planes are exactly representable in both; BatchNorm-folded weights and the
 * activations of this network stay far inside fp16's range). */
int hive_nn_conv3x3_dt(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                       int batch, int relu, int dtype, void *stream);
int hive_nn_resblock_dt(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                        int batch, int dtype, void *stream);

/* The same with a row selection: need = int8[batch] on the device (hive_search_leaf_need) or NULL; the workgroup of a
 * board with need[b] == 0 returns at once and that board's rows of y keep whatever they held.  What the reference does by
 * never calling its model for a finished or capped leaf (solo_play.py:169-197): the leaf batch keeps its shape, the
 * skipped boards cost a workgroup launch instead of 2 x 170 MFLOP x blocks. */
int hive_nn_conv3x3_sel(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                        int batch, int relu, int dtype, const int8_t *need, void *stream);
int hive_nn_resblock_sel(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                         int batch, int dtype, const int8_t *need, void *stream);

/* Rows of equal leaves: row i of y ([batch] rows of row_bytes bytes, a multiple of 16) takes the bytes of row rep[i];
 * rep[i] == i leaves the row alone.  rep comes from hive_leaf_dedup_launch (a representative always maps to itself).
 * Run on the tower's output before the heads, it gives a duplicate exactly the activations it would have computed. */
int hive_nn_copy_rows(void *y, const int32_t *rep, int batch, long long row_bytes, void *stream);

/* The whole residual tower (alpha_net.py:87-99, the loop over res_0 .. res_{nblocks-1}) in ONE launch:
 *   x, y   [batch][144][256] channels-last (dtype as above); y must not alias x
 *   w      [2 * nblocks][9][8][16][64][8]: the fragment-major weights of conv1, conv2 of block 0, conv1 of block 1, ...
 *          back to back (each as hive_nn_conv3x3 reads it);  bias f32 [2 * nblocks][256]
 * A workgroup keeps its board(s) in LDS across all blocks; per block only the skip operand is re-read from, and the
 * block's output written to, global memory (y holds every block's output in turn; at return, the tower's).
 * boards_per_group: 1 = one board per workgroup (two workgroups per CU); 2 = two boards per workgroup (one per CU)
 * whose wave pairs fetch the same weight fragments; 3 = one board per workgroup at ONE wave per SIMD (512 registers:
 * eight LDS fragment reads in flight, weights two k-steps ahead); 0 = choose by batch size (1 or 2).
 * The fp16 epilogues round to fp16 without saturating: activations beyond 65504 become inf.  fp16 is validated on the
 * reference's initialisation and on peaked-head weights (tests/golden/net_wide.npz) and guarded for any other checkpoint
 * by InferenceNet.range_probe (alpha_net.py), which refuses fp16 within 8x of that limit; bf16 has fp32's range.
 * Results are bit-identical to nblocks calls of hive_nn_resblock_dt. */
int hive_nn_tower(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                  int boards_per_group, void *stream);

/* The same tower with a 72-tile wave (hand-written gfx950 assembly, csrc/gen_tower_asm.py): TWO boards per 4-wave
 * workgroup, one wave per SIMD, every wave accumulating 64 output channels x both boards' 288 pixels in 288 accumulator
 * registers, so that every weight fragment fetched from L2 feeds two boards.  Same arguments, layouts and bits as
 * hive_nn_tower (bit-identical to nblocks calls of hive_nn_resblock_dt).
 * rows / nrows (both NULL, or both device pointers): the boards to evaluate -- rows[0 .. *nrows-1] ascending board indices
 * (hive_nn_compact_rows of the leaf batch's `need` flags); boards not listed keep whatever y held.  The launch covers
 * ceil(batch / 2) workgroups; those beyond *nrows return at once. */
int hive_nn_tower72(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                    const int32_t *rows, const int32_t *nrows, void *stream);

/* The same launch with its work BALANCED over the chip.  hive_nn_tower72 gives every pair of boards one workgroup for the
 * whole tower, so its time is ceil(pairs / CUs) whole towers: 460 pairs on 256 CUs cost as much as 512.  Here one workgroup
 * per CU runs; a plan kernel (same stream, reads *nrows on the device) lays the pairs' blocks end to end and gives every
 * workgroup an equal share: a pair cut by a share boundary is started by one workgroup and finished by its neighbour, handed
 * over through y (the blocks' outputs already live there) and an agent-scope release / acquire flag.  Time is proportional to
 * the boards evaluated; results are the same bits.  plan_workspace: hive_nn_tower72_plan_bytes(batch) bytes, 32-byte
 * aligned, contents undefined between calls.  Do not run two balanced launches concurrently on one device (each wants
 * every CU; a tail that waits for a head that is not scheduled falls back to computing the pair itself after a bounded spin). */
int hive_nn_tower72_balanced(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                             const int32_t *rows, const int32_t *nrows, void *plan_workspace, void *stream);
long long hive_nn_tower72_plan_bytes(int batch);

/* ONE 256 -> 256 3x3 convolution on the same 72-tile kernel: y = [relu](conv(x, w) + bias), arguments as hive_nn_conv3x3_dt
 * with cin = 256 and no residual (two boards per workgroup; bit-identical to it).  What the training step's forward and
 * data-gradient convolutions run on (alpha_net.py:117-162 as executed by alpha_net.Trainer). */
int hive_nn_conv72(const void *x, const void *w, const float *bias, void *y, int batch, int relu, int dtype, void *stream);
/* ... with a residual: y = [relu](conv(x, w) + bias + residual), the sum taken in fp32 before the one rounding (the
 * arithmetic of hive_nn_conv3x3_dt with a residual; bit-identical to it).  residual may be y itself (in place), not x.
 * The training step's backward uses it for the gradient that reaches a residual block's input on two ways: dx =
 * dgrad_conv1(dy) + dskip in one pass instead of a convolution and an elementwise add (alpha_net.py:36-54). */
/* ... with the statistics pass of a training-mode BatchNorm behind it: partial[batch / 2][2][256] floats = every
 * workgroup's per-channel sum and sum of squares of the values it stored (the 16-bit results, widened again: exactly what
 * a pass over y would add up).  batch must be even.  hive_nn_bn_act_fwd_partial consumes it. */
int hive_nn_conv72_stats(const void *x, const void *w, const float *bias, void *y, int batch, int relu, int dtype,
                         float *partial, void *stream);
int hive_nn_conv72_add(const void *x, const void *w, const float *bias, const void *residual, void *y, int batch, int relu,
                       int dtype, void *stream);

/* need int8[batch] (1 = evaluate) -> rows int32[batch] (indices of the flagged boards, ascending), *nrows = their number. */
int hive_nn_compact_rows(const int8_t *need, int batch, int32_t *rows, int32_t *nrows, void *stream);

/* The two heads of the network (alpha_net.py:56-80, OutBlock) on hand-written kernels -- no library GEMM in a forward:
 *   x      [batch][144][256] the tower's output (dtype HIVE_BF16 / HIVE_F16), channels-last
 *   wconv  [9][8][64][8]   both 1x1 convolutions (BatchNorm folded) as MFMA A fragments: rows 0..127 the policy convolution
 *                          (alpha_net.py:62), row 128 the value convolution (:58), rows 129..143 zero; bconv f32 [144]
 *   wfc    [99][576][64][8] the policy FC (:63) as MFMA B fragments: action tile, 32-deep k-step over k = pixel * 128 +
 *                          channel, lane = (k % 32) / 8 * 16 + action % 16; bfc f32 [1584]
 *   w1 f32 [144][64] (fc1's weight TRANSPOSED), b1 f32 [64], w2 f32 [64], b2 f32 [1]: the value MLP (:59-60)
 *   workspace  hive_nn_heads_workspace_bytes(batch) bytes, 16-byte aligned, contents undefined between calls
 *   p f32 [batch][1584] = softmax(policy logits), v f32 [batch] = tanh(value)
 * Rounding points: the 1x1 convolutions' outputs are rounded to the 16-bit type (as 16-bit modules do); the FC accumulates
 * in fp32 over K split into hive_nn_heads_splits(batch) ranges (a constant: a board's outputs do not depend on the batch it
 * is evaluated in, nor on its position) summed in ascending order, and the softmax reads those fp32 logits. */
int hive_nn_heads(const void *x, int batch, int dtype, const void *wconv, const float *bconv, const void *wfc, const float *bfc,
                  const float *w1, const float *b1, const float *w2, const float *b2, void *workspace, float *p, float *v,
                  void *stream);
long long hive_nn_heads_workspace_bytes(int batch);
int hive_nn_heads_splits(int batch);

/* Training-mode BatchNorm2d + optional skip connection + optional ReLU of the 256-channel tower, forward and
 * backward (alpha_net.py:25-54 as executed by the training step alpha_net.py:117-162), channels-last bf16:
 *   forward : y = act( (x - mean_c) / sqrt(var_c + eps) * gamma_c + beta_c (+ residual) ), batch statistics over all
 *             `rows` (= batch * 144) positions; running_mean / running_var (may both be NULL) are updated with
 *             `momentum` and the unbiased variance like torch.nn.BatchNorm2d; save_mean / save_invstd f32[256] out.
 *   backward: dx (bf16), dgamma / dbeta (f32[256]) and, when dresidual != NULL, dresidual = dy * (y > 0) (bf16) --
 *             the gradient that flows into the skip input; y is only read when relu != 0.
 *   workspace: device f32[hive_nn_bn_workspace_floats()], contents undefined between calls.
 *   channels: 256 (the tower), or any smaller power of two whose [rows][channels] matrix is a whole number of 256-wide
 *   rows (the heads' BatchNorm2d(128) and BatchNorm2d(1), alpha_net.py:56-80: the matrix is streamed as [rows *
 *   channels / 256][256] and the columns of a channel are added up); the per-channel arrays then hold `channels`
 *   entries. */
int hive_nn_bn_workspace_floats(void);
int hive_nn_bn_act_fwd(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, float momentum, float eps, void *y, float *save_mean, float *save_invstd,
                       float *workspace, long long rows, int channels, int relu, void *stream);
/* hive_nn_bn_act_fwd (256 channels) without its statistics pass: partial = parts x [2][256] per-channel sums / sums of
 * squares covering all `rows` positions (hive_nn_conv72_stats: parts = batch / 2). */
int hive_nn_bn_act_fwd_partial(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean,
                               float *running_var, float momentum, float eps, void *y, float *save_mean, float *save_invstd,
                               float *workspace, const float *partial, int parts, long long rows, int relu, void *stream);
int hive_nn_bn_act_bwd(const void *dy, const void *x, const void *y, const float *gamma, const float *save_mean,
                       const float *save_invstd, void *dx, void *dresidual, float *dgamma, float *dbeta,
                       float *workspace, long long rows, int channels, int relu, void *stream);

/* nn.Conv2d weight (f32, logical shape [256][cin][3][3]; channels_last = 0: stored in that order, 1: stored
 * [256][3][3][cin] as torch.channels_last keeps it) -> the bf16 fragment-major layout hive_nn_conv3x3 reads (cin = 56
 * or 256; out = bf16[9 * cinp * 256], cinp = cin rounded up to 64).  transpose != 0 (cin = 256 only) packs the weights
 * of the data-gradient convolution: hive_nn_conv3x3(dy, 256, packed_T, ...) then returns dx of y = conv3x3(x, w). */
int hive_nn_pack_conv3x3_weights(const float *w, int cin, int transpose, int channels_last, void *out, void *stream);
/* The same for n 256 -> 256 convolutions in ONE launch, both forms: weights = DEVICE array of n device pointers (f32
 * [256][256][3][3], all stored the same way), out_fwd / out_t = bf16[n][9 * 256 * 256] (forward form and data-gradient form of
 * weight i at offset i * 589824).  A training step packs its 38 tower convolutions once, up front, instead of 77 times. */
int hive_nn_pack_conv3x3_weights_multi(const float *const *weights, int n, int channels_last, void *out_fwd, void *out_t,
                                       void *stream);

/* Weight gradient of a 3x3 / stride 1 / zero-padded convolution with 256 input and 256 output channels (what autograd
 * computes for ResBlock.conv1/conv2.weight in the training step, alpha_net.py:36-54,117-162):
 *   dw[tap][k][c] = sum_{b, pixel} dy[b][pixel][k] * x[b][pixel + tap][c],   tap = (dy + 1) * 3 + (dx + 1)
 *   x, dy  bf16 [batch][144][256] channels-last;  dw  f32 [9][256][256], overwritten (torch's weight.grad layout is
 *   dw.view(3, 3, 256, 256).permute(2, 3, 0, 1)).  fp32 accumulation on the MFMA units over up to 32 board ranges, whose
 *   partial sums a second kernel adds in a fixed order: the result is deterministic.
 *   workspace: device f32[hive_nn_wgrad_workspace_floats()] (75 MB of partial sums), contents undefined between calls. */
int hive_nn_wgrad_workspace_floats(void);
int hive_nn_conv3x3_wgrad(const void *x, const void *dy, float *dw, int batch, float *workspace, void *stream);
/* The same with the output order chosen: layout 0 = dw[ty][tx][k][c] (as above), 1 = dw[k][ty][tx][c], the memory of a
 * torch.channels_last nn.Conv2d weight -- the gradient can then be handed to autograd with the parameter's own strides
 * (no strided copy when it is accumulated). */
int hive_nn_conv3x3_wgrad_layout(const void *x, const void *dy, float *dw, int batch, float *workspace, int layout,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HIVE_NN_H */
