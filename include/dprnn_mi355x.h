/*
 * dprnn_mi355x.h -- C ABI of libdprnn_mi355x.so: the MI355X (gfx950) forward path of the DPRNN separator.
 *
 * The reference (Aleksashka-i/tss-with-dprnn) has no native code; its hot path is the Python
 * nn.Module code in src/models/.  Each entry point below replaces the reference lines cited next to it
 * (paths relative to the reference checkout).  All pointers are DEVICE pointers to contiguous fp32
 * unless stated; `stream` is a hipStream_t passed as void*; every function is asynchronous on that
 * stream, returns 0 on success or a negative DPRNN_E* code, and records a message for
 * dprnn_last_error().  No torch types cross this boundary.
 *
 * Device-side data layout (channels-last, one layout for both DPRNN paths):
 *   frames  Y [B][L][F]            F = feature_size (128)
 *   slab    X [B][S][K][F]         S chunks of K frames, hop P; row (b,s,k) holds F contiguous floats.
 *                                  intra-chunk sequences walk k, inter-chunk sequences walk s -- both
 *                                  read whole 512-byte rows, so no transpose is ever materialised.
 *   rnn out H [B*S*K][ndir*Hd]     same row order as X
 *   stats     [B][2] double        (sum, sum of squares) per utterance
 */
#ifndef DPRNN_MI355X_H
#define DPRNN_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPRNN_ABI_VERSION   3   /* dprnn_abi_version(): 2 since dprnn_adam_multi takes its betas as double and the *_ex, saturation and
                                 * pack-format entry points exist; 3 (round 5) with the piece-quad H / packed gate tape entry points of
                                 * round 4, the folded Linear (dprnn_lin_pack_f16, dprnn_rnn_lin_fwd, dprnn_rnn_lin_split_fwd) and the forked
                                 * activation-scale record (4 x DPRNN_SCALE_SLOTS words).  Hosts compare it when they load the library */
#define DPRNN_OK            0
#define DPRNN_EINVAL       -1   /* unsupported size / null pointer */
#define DPRNN_ELAUNCH      -2   /* HIP launch error */
#define DPRNN_EFORMAT      -3   /* packed weights of another mode / cell / layout than the launch reads (or never packed) */

#define DPRNN_F   128           /* feature_size the kernels are built for (config_bss.yaml:19) */
#define DPRNN_HD  128           /* hidden_size  (config_bss.yaml:20) */

const char* dprnn_last_error(void);
int dprnn_abi_version(void);
/* number of compute units of the current device (grid sizing of the persistent kernels) */
int dprnn_num_cus(void);
/* numerics switches (process-wide): 1 = v_exp_f32/v_rcp_f32 forms of sigmoid/tanh (default; measured
 * 2e-7 rel-L2 against the reference output end to end, tests/test_gpu_parity.py), 0 = libm expf/tanhf */
void dprnn_set_fast_activations(int on);        /* LSTM gates (dprnn_bilstm_fwd) */
void dprnn_set_fast_head_activations(int on);   /* mask head tanh/sigmoid */
int dprnn_get_fast_activations(void);
int dprnn_get_fast_head_activations(void);
/* Arithmetic mode of the contractions (gate GEMM of the LSTM recurrence; in mode 3 also the Linear after it, the bottleneck and
 * the mask head):
 *   0 = exact fp32 operands on the fp32 MFMA: bit-for-bit an fma chain (the library's own default; what GRU / Elman cells
 *       always run);
 *   1 = operands split into three bf16 pieces, the six cross products of order <= 2^-16 on the bf16 MFMA with fp32 accumulation
 *       (product error ~2^-24, fp32's exponent range: NO assumption about magnitudes) -- the fallback when mode 3 saturates;
 *   3 = two f16 pieces of power-of-two-scaled operands (activations x 2^4, weights x a per-matrix 2^e), three products on the
 *       f16 MFMA: 22+ operand bits, product error ~2^-24 at half the MFMA work of mode 1 (eight-wave kernels, seq_tile 2016 ..
 *       2096 and the ragged entry).  Values with |2^4 x| > 65000 do not fit f16: the conversion clamps them AND raises a
 *       saturation flag (below) -- a host must check the flags and re-run in mode 1 or 0, as the Python host does
 *       (engine.py: the default mode of its forwards, what the bench headline is measured in).
 *   (Mode 2, two bf16 pieces / three products at ~2^-16, existed in ABI 1 and is gone.)
 * The mode is a property of the PACKED weights as much as of the launch.  dprnn_set_gemm_mode sets the calling THREAD's default,
 * read by the entry points without an explicit mode (dprnn_rnn_pack, dprnn_rnn_fwd, ..., and the row GEMMs, whose operands are
 * the plain torch parameters); the *_ex forms take the mode as an argument.  Either way every pack call records the format it
 * wrote with the buffer's device address and every recurrence launch checks it: a mismatch returns DPRNN_EFORMAT. */
void dprnn_set_gemm_mode(int mode);
int dprnn_get_gemm_mode(void);
/* Workgroups of the f16-pieces row GEMMs (projection, bottleneck, mask head; process-wide tuning knob).  They are bound by their
 * HBM streams, and a persistent workgroup holds a CU's LDS: the grid is also the number of CUs taken from whatever else runs (the
 * other utterance group's recurrence).  0 = one workgroup per CU; the default comes from DPRNN_ROWGEMM_GRID. */
void dprnn_set_rowgemm_grid(int workgroups);
int dprnn_get_rowgemm_grid(void);
/* forget the recorded format of a packed buffer (NULL: of all) -- for hosts that recycle device memory outside the library */
void dprnn_pack_forget(const void* packed);

/* ---- saturation flags of the f16-pieces arithmetic (mode 3, and the f16 forms of the backward).  A thread registers a
 * device buffer of DPRNN_SAT_SLOTS int32 (zeroed by the call, on `stream`); every f16-pieces kernel launched by that thread
 * afterwards ORs 1 into slot s when a value it converted exceeded the f16 range (Inf / NaN inputs included):
 *   0 recurrence input rows; 2 rows of the row GEMMs whose operand is not bounded by construction (bottleneck, mask-head conv and
 *   gate; NOT the Linear after a recurrence, whose rows are h = o tanh(c) in (-1, 1), the mask + decode kernel, whose rows are
 *   tanh * sigmoid products, or the data gradients, which scale from their own measured maximum: slot 1 stays 0); 4 gate gradients
 *   of the f16 BPTT (clamped there, seen by dprnn_wgrad_lstm_f16 when it reads them back; slot 3 stays 0) and the f16 weight-gradient
 *   GEMMs' own operands; 5 non-finite input: the statistics
 *   dprnn_bottleneck_prep reads are NaN / Inf, i.e. the utterance holds such a sample (the conversion sites track maxima, which
 *   see Inf but skip NaN; every forward passes dprnn_bottleneck_prep first).
 * dprnn_saturation_flags copies them to `host_out` (DPRNN_SAT_SLOTS int32 of host memory) after synchronising `stream`, and
 * returns 1 if any is set, 0 if none, < 0 on error.  flags == NULL unregisters (detection off; the clamp stays).  The Python host
 * brackets every forward / backward with the pair and re-runs a saturated call in mode 1 (forward) / the exact fp32 kernels
 * (backward): clamped results are never returned. */
#define DPRNN_SAT_SLOTS 8
int dprnn_saturation_begin(void* flags, void* stream);
int dprnn_saturation_flags(void* host_out, void* stream);

/* Adaptive activation scale of the f16-pieces arithmetic (mode 3).  Activations are multiplied by 2^s before they are split into
 * f16 pieces; without a record s = 4 (|x| <= 4062, beyond that the saturation flags above).  A host that wants s to follow the
 * model -- unbounded norm gains (reference: dprnn.py:76-77) can put the residual stream anywhere -- keeps one device record per
 * model, float32[2 * DPRNN_SCALE_SLOTS]: word [slot] is the largest |x| conversion site `slot` has met so far (dprnn_scale_init
 * sets 256: s = 4), word [DPRNN_SCALE_SLOTS + slot] the largest of the forward in progress.  Per forward, on the calling thread:
 * dprnn_scale_begin(record); before every dprnn_rnn_pack_ex / recurrence launch / bottleneck / mask-head launch
 * dprnn_scale_slot(k) with the k of that site (a pack call and the launches that read the pack share a slot: the pack folds
 * 2^(4 - s) into W_ih, h keeps 2^4); dprnn_scale_update(record, stream) once at the end (sticky maximum); dprnn_scale_begin(NULL).
 * s = clamp(13 - exponent(max), -8, 4): 8 x headroom, and never above 4, so a model that stays below 512 computes what it
 * computed without a record.  The forward that first meets large values still raises the saturation flags. */
#define DPRNN_SCALE_SLOTS 32
int dprnn_scale_init(void* record, void* stream);
int dprnn_scale_begin(void* record);
int dprnn_scale_slot(int slot);
int dprnn_scale_update(void* record, void* stream);
/* The same with a working copy per forward (what the Python host does): the record is float32[4 * DPRNN_SCALE_SLOTS] (sticky maxima,
 * unused, low-streak length, low-streak maximum; dprnn_scale_init fills all of it), `work` float32[2 * DPRNN_SCALE_SLOTS] owned by
 * the forward.  dprnn_scale_fork(record, work, stream) snapshots the sticky maxima into `work` and registers `work` for this thread's
 * launches (instead of dprnn_scale_begin(record)); dprnn_scale_join(record, work, stream) at the end folds the forward's maxima in
 * with atomics; then dprnn_scale_begin(NULL).  Forwards of one model on several threads / streams then never see the scale change
 * between their pack call and the launches that read the pack, and the record comes down again: a site whose forward maximum stays
 * 16 x below its sticky maximum for 8 consecutive forwards falls back to the largest value of that streak (at least 256). */
int dprnn_scale_fork(const void* record, void* work, void* stream);
int dprnn_scale_join(void* record, const void* work, void* stream);

/* packed weights of the split modes: ndir * 196608 u32 (three bf16 piece planes in B-fragment order; the f16 form uses
 * two planes and keeps its scale 2^e, and 1 / (2^4 2^e), in words 512 and 513 of the third) */
int dprnn_lstm_pack_split(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                          const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                          void* packed, float* bsum, void* stream);
int dprnn_lstm_pack_f16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                        const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                        void* packed, float* bsum, void* stream);

/* ---- device memory of one DPRNNTasNet forward (B utterances of T samples; encoder N filters of ks taps, stride; chunks
 * of K frames, hop P; n_blocks DPRNN blocks): returns the total number of bytes, or a negative DPRNN_E* code, and -- when
 * `parts` is not NULL -- the size of each buffer the entry points below take, indexed by DPRNN_WS_*.  The host allocates
 * them (tss-with-dprnn_amd/engine.py does with torch.empty; an utterance group on its own stream is one such set for its
 * share of B) and passes the pointers; the library itself never allocates.  split_gemm != 0 sizes the packed weights for
 * dprnn_set_gemm_mode(1|3).  DPRNN-Spe / IRA add B*(La+L)*128 floats of speaker-encoder rows at most. */
enum {
    DPRNN_WS_Y = 0,      /* frames      [B][L][F]                 (dprnn_bottleneck_fwd -> dprnn_segment_fwd) */
    DPRNN_WS_X,          /* slab        [B][S][K][F]              residual stream */
    DPRNN_WS_X2,         /* slab        second copy: x' of the folded norm lands here (X / X2 ping-pong) */
    DPRNN_WS_Z,          /* slab        Linear output before the norm */
    DPRNN_WS_H,          /* [B*S*K][2*Hd] recurrence output; the mask head's U [B*L][2F] and V [B*L*2][F] reuse it */
    DPRNN_WS_STATS,      /* [1 + 2*n_blocks][B][2] double, zeroed by the host */
    DPRNN_WS_PACKED,     /* [2*n_blocks] packed recurrence weights (dprnn_rnn_pack) */
    DPRNN_WS_BSUM,       /* [2*n_blocks][2][4*Hd] summed biases */
    DPRNN_WS_AFFINE,     /* sc, sh [B][N] each (dprnn_bottleneck_prep) */
    DPRNN_WS_TAPS,       /* [B][2][L][ks] decoder taps (dprnn_head_mask_decode_fwd -> dprnn_decoder_fwd) */
    DPRNN_WS_OUT,        /* [B][2][T_out] separated waveforms */
    DPRNN_WS_PARTS
};
int64_t dprnn_workspace_bytes(int B, int T, int N, int K, int P, int ks, int stride, int n_blocks, int split_gemm,
                              int64_t* parts);

/* ---- encoder statistics: Encoder.forward (src/models/encoder_decoder.py:25-33) feeding the
 * bottleneck norm (src/models/dprnn.py:131-136).  enc[b,n,l] = relu(sum_j We[n][j]*wav[b][l*stride+j])
 * is never stored; this accumulates sum / sum-of-squares of enc over (n,l) into stats[b][0..1]
 * (stats must be zeroed by the caller on the same stream). */
int dprnn_encoder_stats(const float* wav, const float* enc_w, double* stats,
                        int B, int T, int L, int N, int ks, int stride, void* stream);

/* ---- per-utterance affine of the bottleneck norm, folded with the 'add'/'mul'/'film' fusions
 * (dprnn.py:131-136 GroupNorm(1,N) eps 1e-5 / norms.py:6-31 GlobLN eps 1e-8; dprnn_spe.py:194-229):
 *   a[b,l,c] = enc[b,c,l]*sc[b][c] + sh[b][c],
 *   sc = gamma[c]*rstd_b*mul[b][c],  sh = (beta[c]-gamma[c]*mean_b*rstd_b)*mul[b][c] + add[b][c]
 * mul/add may be NULL (1 / 0). */
int dprnn_bottleneck_prep(const double* stats, const float* gamma, const float* beta, float eps,
                          const float* mul, const float* add, float* sc, float* sh,
                          int B, int L, int N, void* stream);

/* ---- bottleneck 1x1 conv on the normalised encoder output (dprnn.py:135-136,169; 'cat' fusion
 * dprnn_spe.py:186-192 enters as the per-utterance bias row_bias[B][F], may be NULL):
 *   Y[b][l][f] = sum_c Wb[f][c]*a[b,l,c] + bb[f] + row_bias[b][f]
 * Wb is [F][ldw] row-major (ldw >= N: the first N input channels are used).  In gate-GEMM mode 3 the product runs in f16
 * pieces like the Linears and the mask head (a in fp32, then split; |a| clamped to 4062). */
int dprnn_bottleneck_fwd(const float* wav, const float* enc_w, const float* sc, const float* sh,
                         const float* Wb, int ldw, const float* bb, const float* row_bias, float* Y,
                         int B, int T, int L, int N, int ks, int stride, void* stream);

/* same for the 'att' fusion (dprnn_spe.py:178-183): Y = row_scale[b][l] * (W a) + bias, where a already carries the
 * per-channel factor fusion_linear(emb) through sc/sh (mul) and row_scale = upsampled attention + 1 */
int dprnn_bottleneck_att_fwd(const float* wav, const float* enc_w, const float* sc, const float* sh,
                             const float* Wb, int ldw, const float* bb, const float* row_scale, float* Y,
                             int B, int T, int L, int N, int ks, int stride, void* stream);

/* ---- attention weights of the 'att' fusion (dprnn_spe.py:98-105,214-222): depth-wise averaging conv
 * (avg_w [N][ks_a], avg_b [N], stride ks_a) of the normalised encoder output (sc0/sh0 = plain norm affine from
 * dprnn_bottleneck_prep with mul = add = NULL), scores[b][t] = sum_c a[b][c]*avg[b][c][t], softmax over t,
 * nearest upsampling to L: row_scale[b][l] = att[b][min(floor(l*La/L), La-1)] + 1.  scores: [B][La] scratch. */
int dprnn_att_row_scale(const float* wav, const float* enc_w, const float* sc0, const float* sh0,
                        const float* avg_w, const float* avg_b, const float* a, float* scores, float* row_scale,
                        int B, int T, int L, int N, int ks, int stride, int ks_a, void* stream);

/* ---- DPRNN._segmentation (dprnn.py:189-201): X[b][s][k][:] = Y[b][s*P+k-K][:] or 0 outside [0,L) */
int dprnn_segment_fwd(const float* Y, float* X, int B, int L, int K, int P, int S, void* stream);

/* ---- pack one nn.LSTM direction pair for dprnn_bilstm_fwd (SingleRNN, dprnn.py:23-28):
 * w_ih [4H][F], w_hh [4H][H], b_ih/b_hh [4H] (gate order i,f,g,o), *_r = reverse direction or NULL.
 * packed: ndir*4*16*8*256 floats (ndir*196608 when a split gemm mode is active), bsum: ndir*4H floats. */
int dprnn_lstm_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                    const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                    float* packed, float* bsum, void* stream);

/* ---- SingleRNN.forward (dprnn.py:33-37) for one DPRNN path, input projection fused into the
 * recurrence: for every sequence q < nseq and step t < T
 *   row(q,t) = (q / inner_n)*outer_stride + (q % inner_n)*inner_stride + t*t_stride
 * reads X row(q,t) [F], writes H row(q,t) [ndir*Hd] (fwd | bwd).  Zero initial state (dprnn.py:36).
 * intra path (dprnn.py:84-86): nseq=B*S, T=K, inner_n=nseq, inner_stride=K, t_stride=1
 * inter path (dprnn.py:94-95): nseq=B*K, T=S, inner_n=K, outer_stride=S*K, inner_stride=1, t_stride=K
 * seq_tile: sequences per workgroup (16, 32, 48 or 64; 0 = choose).  16 runs on eight waves of 16 hidden units each
 * (the small-batch latency form); -16 selects the four-wave form of the same tile (for A/B measurements).  In the split
 * gate-GEMM modes the codes are: 32 (mode 1: two 32-sequence workgroups per CU), 2000 + n (the eight-wave kernel with n
 * sequences per workgroup, one per CU: n = 64, 80, 96 in mode 1; 16, 32, 48, 64, 80, 96 in mode 3 -- 16 keeps the first weight pieces in registers, the form for a few
 * utterances); anything else is refused. */
int dprnn_bilstm_fwd(const float* X, const float* packed, const float* bsum, float* H,
                     int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                     int64_t t_stride, int ndir, int seq_tile, void* stream);

/* ---- dprnn_bilstm_fwd with the previous path's norm + residual folded into the x loader
 * (DPRNNBlock.forward, dprnn.py:90-92 feeding :94-95, and :98-99 feeding the next block's :84-86):
 * the kernel consumes x' = X + ((Z - mean_b)*rstd_b*gamma + beta) (b = sequence / seqs_per_utt, statistics over
 * rows_per_utt*F elements, eps 1e-5|1e-8) and writes x' once to Xout (must differ from X: both directions read X).
 * Saves the separate dprnn_norm_residual_fwd pass over the slab. */
int dprnn_bilstm_norm_fwd(const float* X, const float* Z, const double* stats, const float* gamma,
                          const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                          const float* packed, const float* bsum, float* H,
                          int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                          int64_t t_stride, int ndir, int seq_tile, void* stream);

/* ---- the same two entry points for every rnn_type the reference accepts (getattr(nn, rnn_type), dprnn.py:23):
 * cell 0 = LSTM (gates i,f,g,o), 1 = GRU (r,z,n with n = tanh(W_in x + b_in + r*(W_hn h + b_hn))), 2 = RNN (tanh).
 * Weight shapes as torch: w_ih [G*H][F], w_hh [G*H][H], biases [G*H], G = 4 | 3 | 1.  dprnn_rnn_fwd: Z == NULL means
 * no pending norm (then stats/gamma/beta/Xout are ignored).  GRU / RNN use 16- or 32-sequence tiles. */
int dprnn_rnn_pack(int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                   const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                   float* packed, float* bsum, void* stream);
int dprnn_rnn_fwd(int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                  const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                  const float* packed, const float* bsum, float* H,
                  int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                  int64_t t_stride, int ndir, int seq_tile, void* stream);
/* the same with the gate-GEMM mode (0, 1 or 3) as an argument instead of the thread's default: what a host that runs several
 * models / threads should call (the Python host does) */
int dprnn_rnn_pack_ex(int gemm_mode, int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                      float* packed, float* bsum, void* stream);
int dprnn_rnn_fwd_ex(int gemm_mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                     const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                     const float* packed, const float* bsum, float* H,
                     int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                     int64_t t_stride, int ndir, int seq_tile, void* stream);

/* The same recurrence with H written as the NEXT kernel's operand (gate-GEMM mode 3, LSTM, inference): per row and four adjacent units
 * H holds the 16 bytes {first f16 pieces of 2^4 h x 4 | second pieces x 4} instead of four floats (same addresses, same bytes) -- exactly
 * the words the f16-pieces projection would form from fp32 h, so dprnn_proj_stats_hp_fwd multiplies with them as loaded and its result
 * is bit-identical to dprnn_rnn_fwd_ex + dprnn_proj_stats_fwd.  Replaces SingleRNN's output tensor between dprnn.py:86 and :88 (95 / 96),
 * which no one else reads. */
int dprnn_rnn_fwd_hp_ex(int gemm_mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                        const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                        const float* packed, const float* bsum, float* H,
                        int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                        int64_t t_stride, int ndir, int seq_tile, void* stream);
/* intra_linear / inter_linear + GroupNorm sums on such an H (utt_starts == NULL: rows_per_utt rows per utterance; otherwise the ragged
 * form with utterance u owning rows [utt_starts[u], utt_starts[u+1])) */
int dprnn_proj_stats_hp_fwd(const float* Hp, const float* W, const float* bias, float* Z, double* stats,
                            int64_t M, int64_t rows_per_utt, const int64_t* utt_starts, int n_utt, int kdim, void* stream);

/* Direction-serial BiLSTM with the Linear that follows it folded in (gate-GEMM mode 3, LSTM, inference; replaces dprnn.py:86 + :88 and
 * :95 + :96 -- SingleRNN and intra_linear / inter_linear of one path -- for callers that do not need h itself).  One launch runs ONE
 * direction and multiplies every h_t with that direction's half of the Linear while h_t is the next step's operand in LDS, so h never
 * reaches HBM and the projection kernel disappears: a path moves 7 slab rows of 512 B per row instead of 10.
 *   dprnn_lin_pack_f16: W [F][ndir H] row-major (intra_linear.weight / inter_linear.weight) -> lin_packed, dprnn_lin_pack_floats() floats.
 *   dprnn_rnn_lin_fwd, phase 1 (dir 0 of a bidirectional LSTM): X (+ the pending norm Z, stats, gamma, beta, eps -> x' to Xout, as
 *       dprnn_rnn_fwd_ex) -> P [rows][F] = W[:, :H] h_fwd, fp32.
 *   phase 2 (dir 1; or dir 0 of a unidirectional LSTM with P == NULL): X (x' of phase 1 when a norm was pending: pass Z = NULL) and
 *       P -> Zout [rows][F] = P + W[:, dir H:] h + bias, and stats_out [utterances][2] += (sum, sum of squares) of each utterance's
 *       rows in fp64.  P may be Zout, and may be the buffer phase 1 read its pending Z from (every row is consumed before it is
 *       rewritten).
 * packed / bsum: dprnn_rnn_pack_ex in mode 3; addressing arguments as dprnn_rnn_fwd_ex; seq_tile 0 (= 2064), 2032, 2048 or 2064;
 * seqs_per_utt / rows_per_utt are always required (phase 2 maps sequences to utterances with them). */
int64_t dprnn_lin_pack_floats(void);
int dprnn_lin_pack_f16(const float* W, int ndir, void* lin_packed, void* stream);
/* The same fold with BOTH directions in every launch: the path is cut in time at M = T / 2 instead of by direction, so a launch
 * has twice the workgroups (one per direction and sequence tile) and two pipelines pack the chip as the unfolded launches do.
 *   phase 1: the forward direction walks steps [0, M) (rows t < M), the reverse direction steps [0, T - M) (rows t >= M); each applies
 *       the pending norm to the rows it meets (Z, stats, gamma, beta, eps -> x' to Xout; Z == NULL: X is used as it is), leaves its
 *       partial product in P [rows][F] (P must not be Z) and its cell state + h image in `state`
 *       (dprnn_rnn_lin_state_floats(nseq, seq_tile) floats).
 *   phase 2 (pass X = the x' of phase 1, or the same X when nothing was pending; Z = NULL): both directions continue from `state`
 *       over the rows the other one met in phase 1 and finish them in place: P = P + own half + bias, stats_out += GroupNorm sums.
 * The kernel boundary between the two launches is the only synchronisation.  7 slab rows of HBM traffic per row and path. */
int64_t dprnn_rnn_lin_state_floats(int nseq, int seq_tile);
int dprnn_rnn_lin_split_fwd(int phase, const float* X, const float* Z, const double* stats, const float* gamma,
                            const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                            const float* packed, const float* bsum, const void* lin_packed, const float* lin_bias,
                            float* P, double* stats_out, float* state,
                            int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride,
                            int seq_tile, void* stream);
/* ... and for ragged batches (utterances of different lengths in one launch; addressing arguments as dprnn_rnn_fwd_ragged: utt_rows,
 * seq_utt (intra-chunk, T = K) or wg_desc (inter-chunk: every workgroup is cut at the middle of its own T = S_u), 64-sequence tiles;
 * state: dprnn_rnn_lin_state_floats(64 * tiles, 2064) floats with tiles = ceil(n_items / 64) or n_items). */
int dprnn_rnn_lin_split_ragged_fwd(int phase, const float* X, const float* Z, const double* stats, const float* gamma,
                                   const float* beta, float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc,
                                   int n_items, float* Xout, const float* packed, const float* bsum, const void* lin_packed,
                                   const float* lin_bias, float* P, double* stats_out, float* state, int T, int K, void* stream);
int dprnn_rnn_lin_fwd(int phase, int dir, const float* X, const float* Z, const double* stats, const float* gamma,
                      const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                      const float* packed, const float* bsum, int ndir, const void* lin_packed, const float* lin_bias,
                      float* P, float* Zout, double* stats_out,
                      int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride,
                      int seq_tile, void* stream);

/* diagnostic twin of dprnn_bilstm_fwd (seq_tile 32 or 64, fast activations): stamps[workgroup][8] u64 =
 * {shader cycles, 100 MHz ticks, k-loop cycles, cell-phase cycles, start tick, end tick, XCC id, HW id}; in the split modes (seq_tile 2064)
 * stamps[workgroup][2][8] for waves 0 and 4 = {shader cycles, ticks, h half (or whole MFMA phase), cell update, own x half, wait + x staging,
 * barrier wait, 0}; used by tools/bench_recur.py only. */
int dprnn_bilstm_fwd_stamped(const float* X, const float* packed, const float* bsum, float* H,
                             int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                             int64_t t_stride, int ndir, int seq_tile, unsigned long long* stamps, void* stream);

/* ---- intra_linear / inter_linear (dprnn.py:88,96) + GroupNorm statistics (dprnn.py:90,98):
 * Z[m][f] = sum_k W[f][k]*H[m][k] + bias[f]; stats[m / rows_per_utt] += (sum Z, sum Z^2).
 * kdim = 256 (bidirectional) or 128.  stats zeroed by the caller. */
int dprnn_proj_stats_fwd(const float* H, const float* W, const float* bias, float* Z, double* stats,
                         int64_t M, int64_t rows_per_utt, int kdim, void* stream);

/* ---- intra_norm / inter_norm + residual (dprnn.py:90-92,98-99):
 * Xout = X + ((Z-mean_b)*rstd_b*gamma[f] + beta[f]); count = rows_per_utt*F elements per utterance,
 * biased variance; eps 1e-5 (ln) or 1e-8 (gLN).  Xout may alias X. */
int dprnn_norm_residual_fwd(const float* X, const float* Z, const double* stats, const float* gamma,
                            const float* beta, float eps, float* Xout, int B, int64_t rows_per_utt,
                            void* stream);

/* ---- mask head part 1: PReLU -> Conv2d 1x1 (F -> 2F) -> overlap-add (dprnn.py:173-178,203-217),
 * computed as conv(sum of overlapping prelu rows) + count*bias (the conv is linear):
 * U[b][l][0:2F], channel block 0 = speaker 0 (dprnn.py:176 reshape). */
int dprnn_head_conv_ola_fwd(const float* X, const float* prelu_w, const float* W2d, const float* b2d,
                            float* U, int B, int L, int K, int P, int S, void* stream);

/* same with the last path's norm + residual still pending (see dprnn_bilstm_norm_fwd): rows are read as
 * X + ((Z - mean_b)*rstd_b*gamma + beta) before the PReLU */
int dprnn_head_conv_ola_norm_fwd(const float* X, const float* Z, const double* stats, const float* gamma,
                                 const float* beta, float eps, const float* prelu_w, const float* W2d,
                                 const float* b2d, float* U, int B, int L, int K, int P, int S, void* stream);
/* The same over a RAGGED batch in one launch (gate-GEMM mode 3): utterance u's slab is rows utt_starts[u] .. utt_starts[u + 1] of X (and Z),
 * its frames are l_off[u] .. l_off[u + 1] of U; both tables are int64 on the device, n + 1 entries; stats [n][2]; Z may be NULL (no
 * pending norm).  Replaces one dprnn_head_conv_ola(_norm)_fwd launch per utterance (ragged.py; inferencer_spe.py:20-54 is one at a time). */
int dprnn_head_conv_ola_ragged_fwd(const float* X, const float* Z, const double* stats, const float* gamma, const float* beta,
                                   float eps, const float* prelu_w, const float* W2d, const float* b2d, float* U,
                                   const int64_t* utt_starts, const int64_t* l_off, int n, int64_t total_frames, int K, int P,
                                   void* stream);

/* ---- ragged batches (utterances of DIFFERENT lengths through one launch per DPRNN path; the reference evaluates full-length
 * test utterances one at a time, src/inferencers/inferencer.py:54-58, inferencer_spe.py:27-32, because zero-padding would
 * change the per-utterance GroupNorm statistics).  The slab is the concatenation of the utterances' [S_u][K][F] slabs; per
 * utterance tables replace the uniform (B, S) arithmetic:
 *   dprnn_rnn_fwd_ragged: SingleRNN.forward (dprnn.py:33-37) for one path, split gate GEMM (gemm mode 1 or 3), LSTM.
 *     intra-chunk: wg_desc NULL, n_items = total chunks, T = K, seq_utt[chunk] = its utterance;
 *     inter-chunk: wg_desc[n_items][8] = {row0 lo, row0 hi, T = S_u, sequences <= 64, u, 0, 0, 0} per 64-sequence workgroup
 *       (tiles never straddle utterances), seq_utt NULL.
 *     Z != NULL folds the previous path's norm + residual into the loader as dprnn_bilstm_norm_fwd does (stats[u], count
 *     utt_rows[u] * F).
 *   dprnn_proj_stats_ragged_fwd: dprnn_proj_stats_fwd with utterance u owning rows [utt_starts[u], utt_starts[u+1]). */
int dprnn_rnn_fwd_ragged(const float* X, const float* Z, const double* stats, const float* gamma, const float* beta,
                         float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc, int n_items,
                         float* Xout, const float* packed, const float* bsum, float* H, int T, int K, int ndir, void* stream);
int dprnn_rnn_fwd_ragged_ex(int gemm_mode, const float* X, const float* Z, const double* stats, const float* gamma,
                            const float* beta, float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc,
                            int n_items, float* Xout, const float* packed, const float* bsum, float* H, int T, int K, int ndir,
                            void* stream);
int dprnn_rnn_fwd_ragged_hp_ex(int gemm_mode, const float* X, const float* Z, const double* stats, const float* gamma,
                               const float* beta, float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc,
                               int n_items, float* Xout, const float* packed, const float* bsum, float* H, int T, int K, int ndir,
                               void* stream);      /* H as f16 piece quads, for dprnn_proj_stats_hp_fwd (above) */
int dprnn_proj_stats_ragged_fwd(const float* H, const float* W, const float* bias, float* Z, double* stats,
                                int64_t M, const int64_t* utt_starts, int n_utt, int kdim, void* stream);

/* ---- mask head part 2: tanh(out conv) * sigmoid(gate conv) (dprnn.py:157-158,180) on rows
 * (b,l,spk) of U viewed as [B*L*2][F] -> V [B*L*2][F] */
int dprnn_head_gate_fwd(const float* U, const float* Wo, const float* bo, const float* Wg, const float* bg,
                        float* V, int64_t rows, void* stream);

/* ---- mask head part 3 + masking + decoder taps: end_conv1x1 (F->N, no bias) -> sigmoid|relu
 * (dprnn.py:160-164,181-184), mask*enc (dprnn.py:277) and the per-frame decoder dot products of
 * ConvTranspose1d (encoder_decoder.py:40-42): taps[b][spk][l][j] = sum_n Wd[n][j]*mask*enc,  j<ks.
 * relu_act: 0 sigmoid, 1 relu.  If masked_out != NULL also stores mask*enc of speaker 0 as
 * [B][L][N] (DPRNN-Spe-IRA d0/d1, dprnn_spe_ira.py:79-80,106-107). */
int dprnn_head_mask_decode_fwd(const float* V, const float* Wend, const float* wav, const float* enc_w,
                               const float* dec_w, float* taps, float* masked_out, int relu_act,
                               int B, int T, int L, int N, int ks, int stride, void* stream);

/* ---- decoder overlap of the transposed conv (encoder_decoder.py:40-42):
 * out[b][spk][t] = sum_{l*stride+j=t} taps[b][spk][l][j];  nspk speakers are written (2, or 1 for
 * DPRNN-Spe which decodes speaker 0 only, dprnn_spe.py:325). */
int dprnn_decoder_fwd(const float* taps, float* out, int B, int nspk, int T, int L, int ks, int stride,
                      void* stream);

/* ---- pairwise SI-SDR in fp64 (asteroid==0.7.0 pairwise_neg_sisdr, negated: reference call sites
 * src/trainers/trainer.py:39,104, src/inferencers/inferencer.py:30,60): out[b][i][j] in dB for estimate i vs
 * target j; est [B][n_est][T], tgt [B][n_tgt][T], out [B][n_est][n_tgt] double. */
int dprnn_pairwise_sisdr(const float* est, const float* tgt, double* out, int B, int n_est, int n_tgt, int T,
                         void* stream);
/* the evaluation-metric form asteroid.metrics.get_metrics(['si_sdr']) reports (src/inferencers/inferencer.py:64-70,
 * delegating to pb_bss_eval.evaluation.si_sdr): no mean removal, no eps */
int dprnn_pairwise_sisdr_metric(const float* est, const float* tgt, double* out, int B, int n_est, int n_tgt, int T,
                                void* stream);

/* =====================  DPRNN-Spe / DPRNN-Spe-IRA speaker branch (src/models/dprnn_spe.py)  ===================== */

/* ---- generic 1x1 conv on channels-last rows (spk_encoder convs, ResBlock convs: dprnn_spe.py:18-19,27,115-122):
 * out[m][col0 + o] = sum_k W[(col0+o)*ldw + k] * a(m,k) + bias[col0+o],  o < 128, k < kdim (64|128|256)
 * a(m,k) = prelu((A[m*a_pitch + k] - in_mean[..]) * in_scale[u*aff_stride + k] + in_shift[..]),  u = m / rows_per_utt
 * (in_scale/in_shift NULL: no affine; in_mean NULL: 0; aff_stride 0: shared over utterances; in_prelu NULL: no PReLU) --
 * this is how BatchNorm1d+PReLU of the previous conv (dprnn_spe.py:33-34) and GroupNorm of d0 are folded in.
 * col_stats (optional, [channels][2] fp64, zeroed by caller) += per-channel (sum, sumsq) of the outputs:
 * the batch statistics nn.BatchNorm1d needs in train mode. */
int dprnn_pointwise_fwd(const float* A, int a_pitch, int kdim, const float* in_scale, const float* in_shift,
                        const float* in_mean, int aff_stride, int64_t rows_per_utt, const float* in_prelu, const float* W, int ldw,
                        const float* bias, float* out, int out_pitch, int col0, double* col_stats, int64_t M,
                        void* stream);

/* ---- nn.BatchNorm1d (dprnn_spe.py:20-21) as (y - mean[c])*scale[c] + shift[c] with scale = gamma/sqrt(var + eps),
 * shift = beta, eps 1e-5 (the mean is subtracted before the scale, as torch does: folding it into the shift loses
 * channels whose mean dwarfs their spread to cancellation).
 * col_stats NULL: eval mode (running statistics).  Otherwise train mode: batch statistics over `count`
 * elements per channel, and running_mean / running_var / num_batches_tracked are updated in place as torch does. */
int dprnn_bn_prep(const float* gamma, const float* beta, float* running_mean, float* running_var,
                  int64_t* num_batches_tracked, const double* col_stats, int64_t count, float momentum,
                  float eps, float* scale, float* shift, float* mean, int C, void* stream);

/* ---- tail of ResBlock.forward (dprnn_spe.py:35-42): out[b][t][c] = max_{j<3} prelu2((Y2-mean)*scale+shift + res)[b][3t+j][c]
 * (BatchNorm2 + residual/downsample add + PReLU + MaxPool1d(3)); out is [B][T_in/3][C]. */
int dprnn_resblock_pool_fwd(const float* Y2, const float* scale, const float* shift, const float* mean, const float* res,
                            const float* prelu_w, float* out, int B, int T_in, int C, void* stream);

/* ---- DPRNNSpe._auxiliary tail (dprnn_spe.py:159-162): emb[b][c] = sum_t E[b][t][c] / aux_T,
 * aux_T = (((aux_len - ks) // (ks//2) + 1) // 3 // 3 // 3); aux_len: n_len (1 or B) floats on the device. */
int dprnn_time_mean_fwd(const float* E, const float* aux_len, int n_len, float* emb, int B, int T3, int C, int ks,
                        void* stream);

/* ---- small per-utterance Linear (pred_linear, aux_linear, fusion_linear*, the 'cat' half of bottleneck.1:
 * dprnn_spe.py:90,123,186-212; dprnn_spe_ira.py:51): out[b][o] = sum_{k<K} W[o*ldw + k0 + k]*in[b][k] + bias[o] */
int dprnn_linear_small_fwd(const float* in, int in_pitch, const float* W, int ldw, int k0, int K, const float* bias,
                           float* out, int out_pitch, int B, int O, void* stream);

/* ---- per-utterance (sum, sumsq) of a materialised [B][rows_per_utt][C] tensor: GroupNorm(1,N) of d0 in the
 * second IRA pass (dprnn_spe_ira.py:84 -> dprnn_spe.py:116).  stats zeroed by the caller. */
int dprnn_rows_stats(const float* A, double* stats, int B, int64_t rows_per_utt, int C, void* stream);

/* Per-frame glue of the separator-level surfaces (callers that hold encoder outputs [B, N, L] and drive the separator's pieces
 * themselves: reference src/models/dprnn_rawnet.py:79-105), on channels-last rows [M][C]:
 * dprnn_rows_affine_fwd: out[row][c] = (A[row][c] * sc[u][c] + sh[u][c]) * row_scale[row], u = row / rows_per_utt (sc, sh [U][C],
 *   row_scale [M]; each optional) -- nn.GroupNorm(1, C) / norms.GlobLN apply with (sc, sh) from dprnn_bottleneck_prep
 *   (dprnn.py:131-136, norms.py:27-31), the add / mul / film fusions (dprnn_spe.py:194-229), the 'att' frame scale (:214-222);
 * dprnn_att_row_scale_rows: dprnn_att_row_scale on materialised (optionally still un-normalised: sc0 / sh0) rows instead of the
 *   waveform (dprnn_spe.py:98-105,178-183). */
int dprnn_rows_affine_fwd(const float* A, const float* sc, const float* sh, const float* row_scale, float* out, int64_t M, int C,
                          int64_t rows_per_utt, void* stream);
int dprnn_att_row_scale_rows(const float* rows, int pitch, const float* sc0, const float* sh0, const float* avg_w, const float* avg_b,
                             const float* a, float* scores, float* row_scale, int B, int L, int N, int ks_a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
