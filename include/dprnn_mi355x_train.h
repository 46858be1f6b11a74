/*
 * dprnn_mi355x_train.h -- C ABI of the training (backward) path of libdprnn_mi355x.so.
 *
 * The reference trains with stock autograd: `epoch_loss.backward()` (src/trainers/trainer.py:106) differentiates
 * the nn.Module forward of src/models/dprnn.py.  The entry points below are the hand-written MI355X kernels that
 * produce the same gradients for the DPRNNTasNet forward of dprnn_mi355x.h; each cites the forward lines it
 * differentiates.  Conventions as in dprnn_mi355x.h: device pointers to contiguous fp32 unless stated, `stream` is a
 * hipStream_t passed as void*, asynchronous, 0 on success.  No torch types cross this boundary.
 */
#ifndef DPRNN_MI355X_TRAIN_H
#define DPRNN_MI355X_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- training forward of SingleRNN (dprnn.py:23-37), LSTM cell, exact fp32: dprnn_rnn_fwd on 32-sequence tiles that
 * also records the activated gates i,f,g,o and the cell state of every step in `tape`
 * (dprnn_lstm_tape_floats(nseq, T, ndir) floats, private layout shared with dprnn_lstm_bwd). */
int64_t dprnn_lstm_tape_floats(int nseq, int T, int ndir);
int dprnn_lstm_fwd_train(const float* X, const float* Z, const double* stats, const float* gamma,
                         const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                         const float* packed, const float* bsum, float* H, float* tape,
                         int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                         int64_t t_stride, int ndir, void* stream);

/* ---- transposed weight fragments for dprnn_lstm_bwd: packedT = ndir * 512*256 floats */
int dprnn_lstm_pack_bwd(const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r,
                        float* packedT, void* stream);

/* ---- backward through time of the same recurrence (autograd of nn.LSTM, dprnn.py:36): given dH [rows][ndir*128]
 * writes dX [ndir][rows_total][128] (one slab per direction: the caller adds them) and the gate pre-activation
 * gradients dG [rows][ndir*512] (gate order i,f,g,o) that dprnn_wgrad turns into dW_ih, dW_hh, db.  Sequence
 * addressing as dprnn_bilstm_fwd; must be called with the geometry of the dprnn_lstm_fwd_train that wrote `tape`. */
int dprnn_lstm_bwd(const float* dH, const float* tape, const float* packedT, float* dX, float* dG,
                   int64_t rows_total, int nseq, int T, int inner_n, int64_t outer_stride,
                   int64_t inner_stride, int64_t t_stride, int ndir, void* stream);

/* ---- weight gradient of every Linear / 1x1 conv / LSTM weight: dW[m*ldm + n*ldn] (+)= sum_rows P[row][m]*Q[row][n],
 * m < M (multiple of 4), n < n0 + n1; Q is given as up to two blocks of <= 128 columns (Q1 may be NULL; n0 must be
 * 128 when Q1 is used).  A block with dt != 0 reads Q row (row - dt*t_stride) and zero where t - dt leaves [0, T),
 * t = (row / t_stride) % T: this is h_{t-1} of the recurrence (dt = +1 forward direction, -1 reverse direction).
 * db0 / db1 (optional) (+)= column sums of P (bias gradients; the LSTM's b_ih and b_hh receive the same values).
 * accumulate: 0 overwrite, 1 add to the destination.  workspace: dprnn_wgrad_workspace_floats(M, n0+n1, rows) floats.
 * The reduction order is fixed (no floating-point atomics): results are reproducible run to run. */
int64_t dprnn_wgrad_workspace_floats(int M, int N, int64_t rows);
int dprnn_wgrad(const float* P, int ldp, int M, const float* Q0, int ldq0, int n0, int dt0,
                const float* Q1, int ldq1, int n1, int dt1, int64_t rows, int64_t t_stride, int T,
                float* dW, int64_t ldm, int64_t ldn, float* db0, float* db1, int accumulate,
                float* workspace, void* stream);

/* ---- data gradient of a Linear / 1x1 conv: dA[row][0:nd] = dOut[row][0:kd] @ W, W = [W0 (k0 rows); W1 (kd-k0 rows)]
 * row-major torch weights [kd][ldw] read transposed; (kd, nd) in {(128,256), (128,128), (64,128), (256,128), (128,64)}:
 * intra/inter Linear (dprnn.py:88,96), end_conv1x1 (:160), out + gate convs (:157-158), bottleneck conv (:135). */
int dprnn_dgrad(const float* dOut, int kd, const float* W0, int ldw0, int k0, const float* W1, int ldw1,
                float* dA, int nd, int64_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif
