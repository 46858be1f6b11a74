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

/* Arithmetic of the step GEMM of the LSTM backward (the calling thread's default for dprnn_rnn_pack_bwd / dprnn_rnn_bwd; their _ex forms
 * take it as an argument; the pack records the layout with the buffer and the launch checks it, DPRNN_EFORMAT): 0 = exact fp32 MFMA,
 * 3 = two f16 pieces / three products (train_recur_f16.hip: da scaled by a power of two taken from max|dH| of the launch, which
 * dprnn_rnn_bwd finds with one extra pass over dH unless dprnn_rnn_bwd_use_gradmax hands it over; weights by a per-LSTM 2^e).  GRU / Elman cells always use the fp32 kernel.
 * packedT must hold dprnn_rnn_pack_bwd_floats(ndir) floats in either mode (ndir * 131072 fragments + 128 of scales). */
void dprnn_set_bwd_gemm_mode(int mode);
int dprnn_get_bwd_gemm_mode(void);
int64_t dprnn_rnn_pack_bwd_floats(int ndir);
/* Weight gradients of an LSTM whose BPTT ran in mode 3: dprnn_wgrad_use_f16_pieces(packedT + dprnn_rnn_bwd_gradmax_offset(ndir))
 * makes the NEXT dprnn_wgrad call (only that one) run dG^T [x | h_prev] in f16 pieces (wgrad_f16_kernel: rows as the MFMA k
 * dimension through ds_read_b64_tr_b16), with P scaled by the power of two the BPTT launch took from max|dH| and Q by 2^4. */
void dprnn_wgrad_use_f16_pieces(const float* grad_max);
/* the same for the NEXT dprnn_dgrad call of shape 128 -> 256 (dH = dZ @ W of a bidirectional path's Linear); other shapes ignore it */
void dprnn_dgrad_use_f16_pieces(const float* grad_max);
/* max |x| over n floats (n % 4 == 0) into *out: the scale source for a P that no BPTT launch has measured (the Linears' dZ) */
int dprnn_absmax(const float* x, int64_t n, float* out, void* stream);
int64_t dprnn_rnn_bwd_gradmax_offset(int ndir);
/* The same maxima without the extra passes -- one-shot selectors like the two above (per THREAD: they apply to the next call the
 * same host thread makes; the *_ex forms of the entry points take them as arguments instead), each consumed by the NEXT call of its entry point:
 *   dprnn_gn_bwd_dz_store_absmax(slot)   dprnn_gn_bwd_dz also leaves max|dZ| in *slot (what dprnn_absmax(dZ) would write);
 *   dprnn_dgrad_store_absmax(slot)       dprnn_dgrad also leaves max|dA| in *slot (in the epilogue of its f16 form; its fp32 forms run
 *                                        dprnn_absmax on a dense dA afterwards);
 *   dprnn_rnn_bwd_use_gradmax(slot)      dprnn_rnn_bwd in mode 3 takes max|dH| from *slot instead of passing over dH (the fp32 kernel
 *                                        ignores it).
 * slot: one device float, written / read on the stream of the call it applies to. */
void dprnn_gn_bwd_dz_store_absmax(float* slot);
void dprnn_dgrad_store_absmax(float* slot);
void dprnn_rnn_bwd_use_gradmax(const float* slot);

/* ---- the same three entry points for every rnn_type the reference accepts (getattr(nn, rnn_type), dprnn.py:23):
 * cell 0 = LSTM, 1 = GRU, 2 = RNN (tanh), packing as dprnn_rnn_pack.  dG per row and direction: LSTM [i,f,g,o] (512),
 * GRU [r, z, n_x, n_h] (512: dW_ih rows (r,z,n) pair with (r,z,n_x), dW_hh rows with (r,z,n_h); b_ih / b_hh likewise),
 * RNN [a] (128). */
int64_t dprnn_rnn_tape_floats(int cell, int nseq, int T, int ndir);
int dprnn_rnn_fwd_train(int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                        const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                        const float* packed, const float* bsum, float* H, float* tape,
                        int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                        int64_t t_stride, int ndir, void* stream);
int dprnn_rnn_pack_bwd(int cell, const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r,
                       float* packedT, void* stream);
int dprnn_rnn_bwd(int cell, const float* dH, const float* tape, const float* packedT, float* dX, float* dG,
                  int64_t rows_total, int nseq, int T, int inner_n, int64_t outer_stride,
                  int64_t inner_stride, int64_t t_stride, int ndir, void* stream);
/* floats of dG for dprnn_rnn_bwd.  The fp32 kernel (any cell) writes dG [rows][ndir * G * 128] in slab row order.  Mode 3 (LSTM) writes
 * the QUAD layout  dG4[dir][t][seq / 4][gate * 128 + unit] = 16 bytes {first f16 pieces of S_g da for sequences 4 j .. 4 j + 3 | second
 * pieces}: the BPTT stores 16 bytes per lane instead of four times 4, and dprnn_wgrad_lstm_f16 -- the only reader, which multiplies in
 * the same pieces under the same scale S_g -- loads four consecutive rows of its reduction in one load without converting anything
 * (sequence count rounded up to 4; the bytes of an fp32 dG).  A gate gradient beyond the scale's range is clamped by the BPTT and
 * seen (|p1 + p2| = 65000) by dprnn_wgrad_lstm_f16, which raises saturation flag 4. */
int64_t dprnn_rnn_bwd_dg_floats(int cell, int bwd_gemm_mode, int nseq, int T, int ndir);
/* LSTM weight gradients of ONE direction from the quad-layout dG of a mode-3 BPTT: dWcat [512][256] = [dW_ih | dW_hh] (gate rows
 * i,f,g,o; pitch 256), db0 / db1 [512] (optional; b_ih and b_hh receive the same values).  dG4: that direction's block of dG
 * (dG + dir * T * ceil(nseq / 4) * 2048 floats); X: slab rows [.][128]; Hd: the h columns of the direction (H + dir * 128), pitch ldh
 * = ndir * 128; reverse != 0: the recurrent operand is h_{t+1}.  Sequence addressing as dprnn_rnn_fwd.  grad_max: the device float the
 * BPTT launch scaled by (packedT + dprnn_rnn_bwd_gradmax_offset(ndir)).  workspace: dprnn_wgrad_workspace_floats(512, 256, T *
 * roundup(nseq, 32)) floats.  f16 pieces, fixed-order reduction (reproducible), saturation flag 4. */
int dprnn_wgrad_lstm_f16(const float* grad_max, const float* dG4, const float* X, const float* Hd, int ldh, int reverse,
                         int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride,
                         float* dWcat, float* db0, float* db1, float* workspace, void* stream);
/* the same with every selector as an argument (no per-thread state involved): gemm modes as in dprnn_mi355x.h / above;
 * dh_absmax = device float holding max|dH| (NULL: dprnn_rnn_bwd measures it with a pass over dH; mode 3 only).
 * Packed gate tape: gemm_mode 3 | DPRNN_TAPE_PACKED on dprnn_rnn_fwd_train_ex (LSTM) writes the activated gates i, f, g, o of a
 * (sequence, unit, step) as 24-bit fixed point in three words (tape slots 0-2; slot 3 stays unwritten, c stays fp32 in slot 4)
 * instead of four floats -- a gate is held to 2^-25 (i, f, o) / 2^-24 (g) absolute, the spacing of fp32 near 1 -- and
 * dprnn_rnn_bwd_ex must then be called with bwd_gemm_mode 3 | DPRNN_TAPE_PACKED on that tape.  The tape is the largest stream of a
 * training step: 4 KB instead of 5 KB per slab row (both directions), written once and read once.  Same buffer size
 * (dprnn_rnn_tape_floats); any other mode / cell with the flag returns -1. */
#define DPRNN_TAPE_PACKED 0x100
int dprnn_rnn_fwd_train_ex(int gemm_mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                           const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                           const float* packed, const float* bsum, float* H, float* tape,
                           int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                           int64_t t_stride, int ndir, void* stream);
int dprnn_rnn_pack_bwd_ex(int bwd_gemm_mode, int cell, const float* w_ih, const float* w_hh, const float* w_ih_r,
                          const float* w_hh_r, float* packedT, void* stream);
int dprnn_rnn_bwd_ex(int bwd_gemm_mode, const float* dh_absmax, int cell, const float* dH, const float* tape,
                     const float* packedT, float* dX, float* dG, int64_t rows_total, int nseq, int T, int inner_n,
                     int64_t outer_stride, int64_t inner_stride, int64_t t_stride, int ndir, void* stream);

/* diagnostic: while a device buffer of [workgroups][8] u64 is registered (NULL clears it), dprnn_rnn_bwd with the LSTM
 * cell runs its stamped twin and wave 0 of every workgroup writes: cycles of the whole step loop, then the cycles spent
 * in gate derivatives / first barrier / GEMM / second barrier / hand-over, and the 100 MHz ticks of the loop
 * (tools/bench_recur.py --train 2).  Not used by the product path. */
void dprnn_debug_bwd_stamps(unsigned long long* stamps);

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
/* f16_grad_max != NULL: in f16 pieces, P scaled from *f16_grad_max (what dprnn_wgrad_use_f16_pieces selects for the next call) */
int dprnn_wgrad_ex(const float* f16_grad_max, const float* P, int ldp, int M, const float* Q0, int ldq0, int n0, int dt0,
                   const float* Q1, int ldq1, int n1, int dt1, int64_t rows, int64_t t_stride, int T,
                   float* dW, int64_t ldm, int64_t ldn, float* db0, float* db1, int accumulate,
                   float* workspace, void* stream);

/* ---- data gradient of a Linear / 1x1 conv: dA[row*lda + 0:nd] = dOut[row][0:kd] @ W, W = [W0 (k0 rows); W1 (kd-k0 rows)]
 * row-major torch weights [kd][ldw] read transposed; (kd, nd) in {(128,256), (128,128), (64,128), (256,128), (128,64)}:
 * intra/inter Linear (dprnn.py:88,96), end_conv1x1 (:160), out + gate convs (:157-158), bottleneck conv (:135), speaker
 * encoder convs (dprnn_spe.py:18-19,27,117-122; a 256 -> 256 conv is two calls on 128-column halves: W0 and dA offset by
 * the column, lda = 256). */
int dprnn_dgrad(const float* dOut, int kd, const float* W0, int ldw0, int k0, const float* W1, int ldw1,
                float* dA, int lda, int nd, int64_t rows, void* stream);
/* f16_grad_max / absmax_out as arguments (dprnn_dgrad_use_f16_pieces / dprnn_dgrad_store_absmax select them for the next call) */
int dprnn_dgrad_ex(const float* f16_grad_max, float* absmax_out, const float* dOut, int kd, const float* W0, int ldw0, int k0,
                   const float* W1, int ldw1, float* dA, int lda, int nd, int64_t rows, void* stream);

/* ---- training variants of the mask-head forward (dprnn_mi355x.h) that keep what the backward needs:
 * TG [rows][256] = [tanh(out conv) | sigmoid(gate conv)] (dprnn.py:157-158), mask [B*L*2][N] rows (b,l,spk) (:161-164);
 * masked_out (optional, [B*L][N]) = mask_0*enc, the d0 of DPRNN-Spe-IRA (dprnn_spe_ira.py:79-80) */
int dprnn_head_gate_fwd_train(const float* U, const float* Wo, const float* bo, const float* Wg, const float* bg,
                              float* V, float* TG, int64_t rows, void* stream);
int dprnn_head_mask_decode_fwd_train(const float* V, const float* Wend, const float* wav, const float* enc_w,
                                     const float* dec_w, float* taps, float* mask, float* masked_out, int relu_act,
                                     int B, int T, int L, int N, int ks, int stride, void* stream);

/* Small reductions (per-utterance sums, per-channel parameter gradients, scalar PReLU slope) are accumulated with fp64
 * atomics into caller-zeroed fp64 scratch; dprnn_scratch_to_grad converts n of them to the fp32 gradient
 * (accumulate: 0 overwrite, 1 add). */
int dprnn_scratch_to_grad(const double* scratch, float* grad, int n, int accumulate, void* stream);

/* ---- decoder + masking + mask activation backward (encoder_decoder.py:40-42, dprnn.py:277, :161-164):
 * dOut [B][2][T_out] -> dE [B*L*2][N] (gradient of the end_conv1x1 output, rows (b,l,spk)),
 * dEnc [B*L][N] (gradient reaching the encoder output through the masking), dWd[N*ks] fp64 scratch (decoder.weight).
 * dM0 (optional, [B*L][N]): an encoder-domain gradient of d0 = mask_0*enc added for speaker 0 (DPRNN-Spe-IRA feeds d0
 * to the speaker encoder, dprnn_spe_ira.py:79-84). */
int dprnn_mask_decode_bwd(const float* dOut, const float* wav, const float* enc_w, const float* dec_w,
                          const float* mask, int relu_act, float* dE, float* dEnc, double* dWd, const float* dM0,
                          int B, int T, int L, int N, int ks, int stride, void* stream);

/* ---- tanh * sigmoid gate backward (dprnn.py:180): dAG [rows][256] = [d out-conv pre-activation | d gate-conv ...] */
int dprnn_gate_bwd(const float* dV, const float* TG, float* dAG, int64_t rows, void* stream);

/* ---- Q operand of the conv2d weight gradient: out[(b,l)] = sum over chunks covering l of prelu(X[b][s][k])
 * (dprnn.py:173,203-217), and the conv2d bias gradient dbias[256] += cnt(l) * dU[(b,l)] (fp64 scratch) */
int dprnn_ola_prelu_sum(const float* X, const float* prelu_w, float* out, const float* dU, double* dbias,
                        int B, int L, int K, int P, int S, void* stream);

/* ---- conv2d + PReLU backward w.r.t. the slab (dprnn.py:173-178):
 * dX[(b,s,k)] = (dU[(b, s*P+k-K)] @ W2d) * (x > 0 ? 1 : alpha) (zero outside [0,L)); dalpha (fp64 scratch scalar) */
int dprnn_head_conv_bwd(const float* dU, const float* W2d, const float* X, const float* prelu_w, float* dX,
                        double* dalpha, int B, int L, int K, int P, int S, void* stream);

/* ---- GroupNorm(1,128)/GlobLN + residual backward on the slab (dprnn.py:90-92,98-99), two passes.
 * Pass 1: dy = dXn (+ dXa + dXb when given -- the two per-direction dX slabs of the later recurrence -- written to dXo,
 * which may alias dXn); red[b] += (sum gamma*dy, sum gamma*dy*xh), dgb[0:128] += dgamma, dgb[128:256] += dbeta (fp64
 * scratch).  Z == NULL: only the sum.  Pass 2: dZ = rstd_b*(gamma*dy - red0_b/n - xh*red1_b/n). */
int dprnn_gn_bwd_stats(const float* dXn, const float* dXa, const float* dXb, float* dXo, const float* Z,
                       const double* stats, const float* gamma, float eps, int B, int64_t rows_per_utt,
                       double* red, double* dgb, void* stream);
/* The same with absmax [B][2] (optional; zeroed by the call): per utterance max |gamma dy| and max |x_hat| as float bits.  With red
 * they bound the dZ that dprnn_lin_bwd_f16 forms on the fly: |dZ| <= rstd (max|gamma dy| + |red0 / n| + max|x_hat| |red1 / n|). */
int dprnn_gn_bwd_stats_ex(const float* dXn, const float* dXa, const float* dXb, float* dXo, const float* Z,
                          const double* stats, const float* gamma, float eps, int B, int64_t rows_per_utt,
                          double* red, double* dgb, unsigned* absmax, void* stream);
int dprnn_gn_bwd_dz(const float* dY, const float* Z, const double* stats, const double* red, const float* gamma,
                    float eps, int B, int64_t rows_per_utt, float* dZ, void* stream);
/* absmax_out as an argument (dprnn_gn_bwd_dz_store_absmax selects it for the next call) */
int dprnn_gn_bwd_dz_ex(float* absmax_out, const float* dY, const float* Z, const double* stats, const double* red,
                       const float* gamma, float eps, int B, int64_t rows_per_utt, float* dZ, void* stream);

/* ---- Pass 2 of the GroupNorm backward AND the backward of the Linear in front of it (dprnn.py:88-90, 96-98) in one pass over the rows,
 * in f16 pieces (bidirectional recurrences: H [rows][256], W = linear.weight [128][256]): dZ is formed while dY and Z are staged and never
 * stored; dH [rows][256] = dZ W, dW [128][256] = dZ^T H, db [128] = column sums of dZ; hmax (optional device float) = max |dH|, the f16
 * BPTT's gradient scale.  6 slab units of 512 B per row where dprnn_gn_bwd_dz_ex + dprnn_dgrad_ex + dprnn_wgrad_ex move 9.  red / absmax:
 * from dprnn_gn_bwd_stats_ex (red times built / real channels for a narrow model's twin).  B <= 1024 utterances.
 * workspace: dprnn_lin_bwd_workspace_floats(B * rows_per_utt) floats. */
int64_t dprnn_lin_bwd_workspace_floats(int64_t rows);
int dprnn_lin_bwd_f16(const float* dY, const float* Z, const double* stats, const double* red, const unsigned* absmax,
                      const float* gamma, float eps, int B, int64_t rows_per_utt, const float* H, const float* W,
                      float* dH, float* dW, float* db, float* hmax, float* workspace, void* stream);

/* ---- DPRNN._segmentation backward (dprnn.py:189-201): dY[(b,l)] = sum over chunks covering l of (dXn + dXa + dXb) */
int dprnn_segment_bwd(const float* dXn, const float* dXa, const float* dXb, float* dY,
                      int B, int L, int K, int P, int S, void* stream);

/* ---- encoder + bottleneck norm (encoder_decoder.py:25-33, dprnn.py:131-134); the encoder output is recomputed from
 * the waveform, never stored.  norm_rows: A[(b,l)][n] = enc*sc + sh (Q operand of the bottleneck conv weight gradient).
 * norm_bwd_stats: red[b], dgb[0:64 | 64:128] of the bottleneck norm for dy = dA.  encoder_bwd: gradient of the
 * encoder output (norm backward + dEnc from the masking; dEnc may be NULL), ReLU, and dWe[N*ks] += d_pre * wav (fp64 scratch). */
int dprnn_encoder_norm_rows(const float* wav, const float* enc_w, const float* sc, const float* sh, float* A,
                            int B, int T, int L, int N, int ks, int stride, void* stream);
int dprnn_encoder_norm_bwd_stats(const float* wav, const float* enc_w, const float* dA, const double* stats,
                                 const float* gamma, float eps, double* red, double* dgb,
                                 int B, int T, int L, int N, int ks, int stride, void* stream);
int dprnn_encoder_bwd(const float* wav, const float* enc_w, const float* dA, const double* stats,
                      const float* gamma, float eps, const double* red, const float* dEnc, double* dWe,
                      int B, int T, int L, int N, int ks, int stride, void* stream);

/* ---- GroupNorm(1,64) of a materialised [B][L][64] tensor (d0 in DPRNN-Spe-IRA, dprnn_spe_ira.py:84 -> dprnn_spe.py:116):
 * rows64_affine: A = X*sc[b] + sh[b] (the normalised rows, Q operand of the following conv's weight gradient);
 * rows64_norm_bwd: red[b] / dgb (dgamma | dbeta) as dprnn_encoder_norm_bwd_stats, then dX = gradient of the rows. */
int dprnn_rows64_affine(const float* X, const float* sc, const float* sh, float* A, int B, int L, void* stream);
int dprnn_rows64_norm_bwd(const float* X, const float* dA, const double* stats, const float* gamma, float eps,
                          double* red, double* dgb, float* dX, int B, int L, void* stream);
/* The same for the zero-padded twin of a narrower model (padded.py: real channels in the leading columns, zeros behind them, gamma = beta =
 * 0 there): count_scale = 64 / real channels turns the element count L x 64 of the two sums into L x real.  The other global norms take the
 * same correction from the host between their two calls (stats and red are plain (sum, sum of squares) / (sum, sum) pairs of doubles: multiply
 * them by built / real channels -- training._count_scale_red, engine._count_scale). */
int dprnn_rows64_norm_bwd_ex(const float* X, const float* dA, const double* stats, const float* gamma, float eps,
                             double* red, double* dgb, float* dX, int B, int L, double count_scale, void* stream);

/* ---- PIT SI-SDR loss = PITLossWrapper(pairwise_neg_sisdr, pit_from='pw_mtx') of asteroid==0.7.0 as called at
 * src/trainers/trainer.py:39,104.  fwd: moments [B][n][n][5], pw [B][n][n] (SI-SDR dB, estimate i vs target j),
 * loss_b [B] = -max over permutations of mean_i pw[i][perm[i]], perm [B][n] int32, coef [B][n][4] (all fp64 but perm);
 * n <= 3.  bwd: dEst = upstream * d(mean_b loss_b)/d est  (upstream: device float scalar or NULL = 1). */
int dprnn_pit_sisdr_fwd(const float* est, const float* tgt, double* moments, double* pw, double* loss_b,
                        int* perm, double* coef, int B, int n, int T, void* stream);
int dprnn_pit_sisdr_bwd(const float* est, const float* tgt, const int* perm, const double* coef,
                        const float* upstream, float* dEst, int B, int n, int T, void* stream);

/* =====================  DPRNN-Spe speaker branch and fusions (src/models/dprnn_spe.py), backward  ===================== */

/* ---- time mean (dprnn_spe.py:159-162): dE[b][t][c] = d_emb[b][c] / aux_T_b */
int dprnn_time_mean_bwd(const float* d_emb, const float* aux_len, int n_len, float* dE, int B, int T3, int C, int ks,
                        void* stream);

/* ---- tail of ResBlock.forward backward (dprnn_spe.py:36-42; MaxPool1d(3) + PReLU + residual add): given dOut [B][T_in/3][C]
 * writes dU [B][T_in][C], the gradient of u = (Y2 - mean)*scale + shift + res (it is the gradient of the BatchNorm2 output AND of
 * the residual branch), and adds the PReLU slope gradient to the fp64 scratch scalar dalpha */
int dprnn_resblock_pool_bwd(const float* dOut, const float* Y2, const float* scale, const float* shift, const float* mean,
                            const float* res, const float* prelu_w, float* dU, double* dalpha, int B, int T_in, int C,
                            void* stream);

/* ---- nn.BatchNorm1d backward (dprnn_spe.py:20-21,32,35).  dprnn_bn_moments: per-channel mean / rstd the forward used
 * (col_stats != NULL: batch statistics of train mode; NULL: running statistics of eval mode).  dprnn_bn_bwd: red[0:C] +=
 * sum dY (= dbeta), red[C:2C] += sum dY*xh (= dgamma) in fp64 scratch, then dX = gamma*rstd*(dY - red0/M - xh*red1/M)
 * (train) or dY*gamma*rstd (eval).  C = 128 or 256. */
int dprnn_bn_moments(const double* col_stats, const float* running_mean, const float* running_var, int64_t count,
                     float eps, float* mean, float* rstd, int C, void* stream);
int dprnn_bn_bwd(const float* dY, const float* Y, const float* mean, const float* rstd, const float* gamma,
                 int train, double* red, float* dX, int64_t M, int C, void* stream);

/* ---- PReLU(BatchNorm1(conv1)) between the two convs of a ResBlock (dprnn_spe.py:33-34): rows = prelu((Y-mean)*scale + shift)
 * (the Q operand of conv2's weight gradient) and its backward dB = dA*prelu'(b), dalpha (fp64 scratch scalar) */
int dprnn_affine_prelu_rows(const float* Y, const float* scale, const float* shift, const float* mean, const float* prelu_w,
                            float* out, int64_t M, int C, void* stream);
int dprnn_affine_prelu_bwd(const float* dA, const float* Y, const float* scale, const float* shift, const float* mean,
                           const float* prelu_w, float* dB, double* dalpha, int64_t M, int C, void* stream);

/* out = a + b (n floats, multiple of 4): the two branches of a ResBlock meeting at its input */
int dprnn_add_rows(const float* a, const float* b, float* out, int64_t n, void* stream);

/* out[b][c] += sum_l A[(b,l)][c] (fp64 scratch [B][C]): gradient of the per-utterance bias row of the 'cat' fusion */
int dprnn_utt_colsum(const float* A, double* out, int B, int L, int C, void* stream);

/* ---- gradients of dprnn_linear_small_fwd (pred_linear, fusion_linear*, aux_linear, the 'cat' half of bottleneck.1):
 * d_in[b][k] (+)= sum_o d_out[b][o]*W[o*ldw + k0 + k] (d_in may be NULL); dW[o*ldw + k0 + k] = sum_b d_out[b][o]*in[b][k],
 * db[o] = sum_b d_out[b][o] (dW / db may be NULL) */
int dprnn_linear_small_bwd(const float* d_out, int do_pitch, const float* in, int in_pitch, const float* W, int ldw,
                           int k0, int K, float* d_in, int di_pitch, int accumulate_in, float* dW, float* db,
                           int B, int O, void* stream);

/* ---- per-channel fusion operands (dprnn_spe.py:194-229: 'add', 'mul', 'film'; also the 'att' channel factor):
 * a' = an*mul + add with an = enc*sc0 + sh0 the plain bottleneck-norm output (sc0/sh0 from dprnn_bottleneck_prep without
 * mul/add).  Given dAp = gradient of a': dAn = dAp*mul (mul NULL: copy), dmul[b][c] += sum_l dAp*an, dadd[b][c] += sum_l dAp
 * (fp64 scratch [B][N], either may be NULL).
 * 'att' fusion (dprnn_spe.py:178-183,214-222; mul = a = fusion_linear(emb)): with d_scores [B][La] (dprnn_att_softmax_bwd)
 * and the averaging conv (avg_w [N][ks_a], avg_b [N], stride ks_a) the attention-score path is added:
 * dAn[b][l][c] += dS[b][t]*a[b][c]*avg_w[c][j], dmul[b][c] += dS[b][t]*(avg_w[c][j]*an + [j==0]*avg_b[c]), l = t*ks_a + j.
 * d_scores NULL: no attention path. */
int dprnn_fusion_bwd(const float* wav, const float* enc_w, const float* sc0, const float* sh0, const float* dAp,
                     const float* mul, float* dAn, double* dmul, double* dadd,
                     const float* d_scores, const float* avg_w, const float* avg_b, int ks_a,
                     int B, int T, int L, int N, int ks, int stride, void* stream);

/* ---- 'att' fusion, per-frame factor (dprnn_bottleneck_att_fwd: Y = rs[b][l]*(W a') + bias, rs = upsampled softmax + 1):
 * dG = dY*rs (gradient of W a'), d_rs[(b,l)] = sum_f dY*(Y - bias)/rs, dbias[f] += sum dY (fp64 scratch [128]).
 * dprnn_att_softmax_bwd: nearest-upsampling + softmax backward of dprnn_att_row_scale:
 * d_scores[b][t] = att*(d_att - sum_t att*d_att), att = softmax_t(scores), d_att[t] = sum of d_rs over the frames reading t. */
int dprnn_att_scale_bwd(const float* dY, const float* Y, const float* bias, const float* row_scale, float* dG,
                        float* d_rs, double* dbias, int64_t rows, void* stream);
int dprnn_att_softmax_bwd(const float* d_rs, const float* scores, float* d_scores, int B, int L, int ks_a, void* stream);

/* ---- torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step over ALL parameters in two launches
 * (src/trainers/trainer.py:115-118; scripts/train/config_bss.yaml:30-33).  table: n_tensors device records
 * {float* p; const float* g; float* m; float* v; int64 numel} (40 bytes each).  dprnn_multi_sumsq adds the sum of squares
 * of every g to out[0] (fp64, zeroed by the caller).  dprnn_adam_multi applies, with the clip coefficient
 * c = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (max_norm <= 0: no clipping): g' = c*g + weight_decay*p;
 * m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps) -- the update of
 * torch.optim.Adam (amsgrad off); .grad is not modified.  The betas are doubles and the bias corrections 1-b^step are
 * formed in double on the host, as torch forms them in Python floats (1-0.999^1 in fp32 is 6e-5 off). */
int dprnn_multi_sumsq(const void* table, int n_tensors, double* out, void* stream);
int dprnn_adam_multi(const void* table, int n_tensors, float lr, double beta1, double beta2, float eps,
                     float weight_decay, int step, float max_norm, const double* sumsq, void* stream);

#ifdef __cplusplus
}
#endif
#endif
