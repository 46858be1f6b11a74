/*
 * dprnn_mi355x_wide.h -- C ABI of the any-width DPRNN-TasNet forward of libdprnn_mi355x.so ("API completeness
 * of the kept surface").
 *
 * The kernels of dprnn_mi355x.h are built for input_size 64, feature_size = hidden_size 128 (every shipped YAML); the reference's
 * constructors take any width (src/models/dprnn.py:117-120, 234-239).  Narrower models run on those kernels as zero-padded twins;
 * WIDER ones (feature_size or hidden_size above 128, input_size above 64) run on the entry points below: the same channels-last
 * row layout ([rows][C] fp32, C a run-time argument), the norms materialised between the launches instead of folded into the next loader;
 * fp32 operands on the fp32 MFMA, and the f16-pieces arithmetic (dprnn_wide_lstm_f16_fwd, dprnn_wide_linear_f16) for LSTM recurrences and the
 * Linear after them.  Conventions as in dprnn_mi355x.h: device pointers to contiguous fp32 unless stated,
 * `stream` is a hipStream_t passed as void*, asynchronous, 0 on success, dprnn_last_error() says what failed.
 * No torch types cross this boundary.  One forward is (tss-with-dprnn_amd/wide.py):
 *
 *   E = encoder_fwd(wav)                                   encoder_decoder.py:25-33
 *   flat_stats(E) -> norm_prep -> gemm(E, sc, sh, W_bott)  dprnn.py:131-136,169      Y [B L][F]
 *   segment_fwd(Y)                                         dprnn.py:189-201          X [B S K][F]
 *   per block, per path: rnn_pack, rnn_fwd(X) -> H [rows][ndir Hd]; gemm(H, W_lin, stats) -> Z; norm_prep; norm_residual_fwd(X, Z)
 *                                                          dprnn.py:79-99
 *   gemm(X, prelu, W_conv2d) -> U [rows][2 F]; ola_fwd(U) -> frames [B 2 L][F]      dprnn.py:172-179,203-217
 *   gemm(frames, W_out, tanh) ; gemm(frames, W_gate, sigmoid, mul = the former)     dprnn.py:180
 *   gemm(., W_end, sigmoid | relu, mul = E)                dprnn.py:181-186,277     masked encoder rows [B 2 L][N]
 *   gemm(., decoder taps) ; dprnn_decoder_fwd              encoder_decoder.py:40-49
 */
#ifndef DPRNN_MI355X_WIDE_H
#define DPRNN_MI355X_WIDE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* E[b*L + l][n] = relu(sum_j enc_w[n][j] * wav[b][l*stride + j]): Encoder.forward (encoder_decoder.py:25-33) as rows [B*L][N]. */
int dprnn_wide_encoder_fwd(const float* wav, const float* enc_w, float* E, int B, int T, int L, int N, int ks, int stride, void* stream);

/* stats[b] += (sum, sum of squares) of A[b*per_utt .. (b+1)*per_utt) in fp64 (stats [B][2] double, zeroed by the caller): the sums
 * nn.GroupNorm(1, N) / norms.GlobLN take over an utterance (dprnn.py:131-134, norms.py:9-18). */
int dprnn_wide_flat_stats(const float* A, double* stats, int B, int64_t per_utt, void* stream);

/* (sc, sh)[b][c] with x*sc + sh = (x - mean_b) / sqrt(var_b + eps) * gamma[c] + beta[c]; mean, var over rows_per_utt * C elements
 * (biased variance, as both norm classes use: norms.py:11). */
int dprnn_wide_norm_prep(const double* stats, const float* gamma, const float* beta, float eps, float* sc, float* sh, int B,
                         int64_t rows_per_utt, int C, void* stream);

/* DPRNN._segmentation (dprnn.py:189-201): X[b][s][k][:] = Y[b][s*P + k - K][:] or 0; S = (L + K) / P + 1. */
int dprnn_wide_segment_fwd(const float* Y, float* X, int B, int L, int K, int P, int S, int C, void* stream);

/* Recurrence weights of one direction of a torch RNN module (dprnn.py:23-28) in the fragment order dprnn_wide_rnn_fwd streams.
 * cell: 0 LSTM, 1 GRU, 2 RNN (tanh).  Wp: dprnn_wide_rnn_pack_floats(cell, F, H) floats, bv: (cell == 2 ? 1 : 4) * roundup16(H). */
int64_t dprnn_wide_rnn_pack_floats(int cell, int F, int H);
int dprnn_wide_rnn_pack(int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int F, int H,
                        float* Wp, float* bv, void* stream);

/* SingleRNN.forward (dprnn.py:33-37) over nseq sequences of T steps, ndir directions (direction d's pack at Wp + d*pack_floats,
 * bv + d*gates*roundup16(H)); input row of (sequence q, step t) = (q / inner_n)*outer + (q % inner_n)*inner + t*t_stride in rows of
 * F floats, output rows of ndir*H floats at the same index (forward | backward halves, as torch concatenates them).  Intra-chunk:
 * inner_n = 1, outer = K, t_stride = 1; inter-chunk: inner_n = K, outer = S*K, inner = 1, t_stride = K (dprnn.py:84,94).
 * hidden_size <= 512 and 2*16*(roundup16(F) + roundup16(H) + 8)*4 bytes of LDS <= 160 KB. */
int dprnn_wide_rnn_fwd(int cell, const float* X, const float* Wp, const float* bv, float* Hout, int ndir, int F, int H, int nseq, int T,
                       int64_t inner_n, int64_t outer, int64_t inner, int64_t t_stride, void* stream);

/* The same recurrence over TILES of up to 16 sequences with their own step counts: utterances of different lengths in one launch (the
 * inter-chunk path of a ragged batch: a tile = 16 chunk positions of one utterance, S_u steps; src/inferencers/inferencer.py:54-58 runs
 * them one at a time).  desc[4*i .. 4*i+3] = {first row, steps, sequences (1..16), row step between sequences} (device int64); row of
 * (tile sequence j, step t) = first + j*step_between + t*t_stride.  List the longest tiles first: their workgroups run longest. */
int dprnn_wide_rnn_fwd_tiles(int cell, const float* X, const float* Wp, const float* bv, float* Hout, int ndir, int F, int H,
                             const int64_t* desc, int ntiles, int64_t t_stride, void* stream);

/* ---- LSTM recurrence in the f16-pieces arithmetic (csrc/wide_f16.hip): every fp32 operand as two f16 pieces of a power-of-two multiple of
 * itself, a product = three v_mfma_f32_16x16x32_f16 (fp32 accumulate, 22+ operand bits), 64 sequences per 8-wave workgroup.  Ranges are by
 * construction: dprnn_wide_absmax leaves max |.| over the finite elements (as float bits; zero the word first) of the slab the launch
 * reads in amax[0] and of W_ih / W_hh of direction d in amax[1 + 2 d] / amax[2 + 2 d]; dprnn_wide_lstm_pack_f16 derives the direction's
 * scales from them (scal[4 d .. 4 d + 3], device floats) and writes the weight fragments (dprnn_wide_lstm_pack_f16_words(F, H) 16-byte
 * words per direction at Wp + d * words) and the bias sums bv[d][4 * roundup16(H)]; dprnn_wide_lstm_f16_fwd is dprnn_wide_rnn_fwd /
 * dprnn_wide_rnn_fwd_tiles (desc != NULL: tiles of up to ns sequences) for LSTM cells; ns = 64 sequences per workgroup up to hidden_size 256,
 * 32 up to 512.  ns * (2 * (roundup32(F) + roundup32(H)) + 16) * 2 bytes of LDS <= 160 KB. */
int dprnn_wide_absmax(const float* A, int64_t n, void* out_bits, void* stream);
int64_t dprnn_wide_lstm_pack_f16_words(int F, int H);
int dprnn_wide_lstm_pack_f16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int F, int H, int dir,
                             const void* amax_bits, float* scal, void* Wp, float* bv, void* stream);
int dprnn_wide_lstm_f16_fwd(const float* X, const void* Wp, const float* bv, const float* scal, float* Hout, int ndir, int F, int H,
                            int nseq, int T, int64_t inner_n, int64_t outer, int64_t inner, int64_t t_stride, const int64_t* desc,
                            int ntiles, void* stream);

/* The Linear after a recurrence (dprnn.py:59,68) in the same arithmetic: W (O x K, ldw) packed once into f16-piece fragments with the scale from
 * its measured maximum (amax_bits: dprnn_wide_absmax of W), the rows A (|a| <= a_bound: 1 for an LSTM's h) split on their way into LDS;
 * out[m][o] = sum_k A[m][k] W[o][k] + bias[o], stats as dprnn_wide_gemm.  Wp: dprnn_wide_linear_pack_f16_words(K, O) 16-byte words, scal: 2 floats. */
int64_t dprnn_wide_linear_pack_f16_words(int K, int O);
int dprnn_wide_linear_pack_f16(const float* W, int ldw, int K, int O, const void* amax_bits, float a_bound, void* Wp, float* scal, void* stream);
int dprnn_wide_linear_f16(const float* A, int64_t lda, const void* Wp, const float* scal, const float* bias, float* out, int64_t ldo,
                          double* stats, int64_t rows_per_utt, int64_t M, int K, int O, void* stream);

/* out[m][o] = act(rscale[m] * sum_k a(m,k) * W[o*ldw + k] + bias[o] + ubias[u*O + o]) * mul[mrow(m)*ldm + o]
 *   a(m,k) = prelu(A[m*lda + k] * sc[u*K + k] + sh[u*K + k]), u = m / rows_per_utt (sc / sh NULL: no affine; prelu NULL: none, else
 *   one device float: nn.PReLU(), dprnn.py:150,172);  act: 0 none, 1 relu, 2 sigmoid, 3 tanh;  bias, ubias, rscale, mul, stats optional;
 *   ubias [U][O]: a bias per utterance (the embedding half of the 'cat' fusion's 1x1 conv, dprnn_spe.py:186-192); rscale [M]: a factor per
 *   row (the 'att' fusion's frame weights, dprnn_spe.py:178-183,214-222: conv(x * r) = r * (W x) + b);
 *   mrow(m) = m when mul_rep == 1, else (m / (mul_L*mul_rep))*mul_L + m % mul_L (the encoder rows shared by the mul_rep speakers,
 *   dprnn.py:277);  stats[u] += (sum, sum of squares) of the outputs of utterance u = m / rows_per_utt in fp64.
 * Every Linear / 1x1 convolution of the path: bottleneck (dprnn.py:135-136), intra / inter Linear (:59,68), conv2d (:151), out / gate
 * (:153-154), end_conv1x1 (:156) and the decoder's per-frame taps (encoder_decoder.py:36-42). */
int dprnn_wide_gemm(const float* A, int64_t lda, const float* sc, const float* sh, int64_t rows_per_utt, const float* prelu,
                    const float* W, int ldw, const float* bias, const float* ubias, const float* rscale, int act, const float* mul,
                    int64_t ldm, int64_t mul_L, int mul_rep, float* out, int64_t ldo, double* stats, int64_t M, int K, int O, void* stream);

/* X[m][c] += Z[m][c]*sc[u][c] + sh[u][c], u = m / rows_per_utt: the norm after a path's Linear and the residual (dprnn.py:90-92,98-99).
 * amax_bits (optional, as dprnn_wide_absmax: zero it first) receives max |X| of the result -- the next recurrence's activation scale. */
int dprnn_wide_norm_residual_fwd(float* X, const float* Z, const float* sc, const float* sh, int64_t M, int C, int64_t rows_per_utt,
                                 void* amax_bits, void* stream);

/* DPRNN._overlap_add (dprnn.py:203-217) of U [B*S*K][nspk*C] (conv2d output, speaker-major channels: the reshape of dprnn.py:176)
 * -> frames [(b*nspk + spk)*L + l][C]. */
int dprnn_wide_ola_fwd(const float* U, float* frames, int B, int L, int K, int P, int S, int C, int nspk, void* stream);

#ifdef __cplusplus
}
#endif
#endif
