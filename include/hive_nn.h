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

#ifdef __cplusplus
}
#endif
#endif /* HIVE_NN_H */
