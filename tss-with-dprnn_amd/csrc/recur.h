// Shared declarations of the recurrence kernels (recur.hip, recur_split.hip, train_recur.hip).
#pragma once
#include "common.h"

namespace recur {
constexpr int F = 128;      // input features of the LSTM
constexpr int HD = 128;     // hidden units
constexpr int XS = 257;     // LDS row stride in floats: [x(128) | h(128)] + 1 pad (odd => conflict-free)
constexpr int NGRP = 16;    // k-groups of 16 per step: 0..7 over x, 8..15 over h
constexpr int PACK_PER_DIR = 4 * NGRP * 8 * 256;   // floats

struct RecurArgs {
    const float* X;
    const float* Wp;
    const float* bsum;
    float* H;
    int nseq, T, inner_n, ndir;
    long long outer_stride, inner_stride, t_stride;   // in rows
    unsigned long long* stamps;                       // diagnostic build only: [wg][8] counters
    // optional fused "pending norm + residual" of the previous path (dprnn.py:90-92,98-99): when Z != nullptr the
    // kernel reads x' = X + (Z - mean_b)*rstd_b*gamma + beta, and direction 0 also writes x' to Xout (!= X)
    const float* Z;
    const double* stats;
    const float* gamma;
    const float* beta;
    float* Xout;
    float eps;
    int seqs_per_utt;
    long long rows_per_utt;
    int cell;                                         // 0 LSTM, 1 GRU, 2 Elman RNN (tanh)
    float* tape;                                      // training forward only: activated gates + cell state per step
    int tape_packed;                                  // f16-pieces training forward: gates as 24-bit fixed point in tape slots 0-2 (common.h: tape_pack), c in slot 4
    // ---- ragged batches (utterances of different lengths in one launch; bf16-split eight-wave kernel only)
    const int* wg_desc;                               // per workgroup of a direction: {row0 lo, row0 hi, T, sequences, utterance, 0, 0, 0}:
                                                      // its sequences are rows row0 + j * inner_stride, it runs T steps (inter-chunk path)
    const int* seq_utt;                               // utterance of every sequence (intra-chunk path; wg_desc == nullptr)
    const long long* utt_rows;                        // rows of every utterance (count of its pending norm); nullptr: rows_per_utt
    int h_pieces;                                     // f16-pieces kernels, inference: H receives, per four adjacent units, the 16 bytes {first f16 pieces x 4 | second
                                                      // pieces x 4} of 2^4 h instead of four floats -- the operand dprnn_proj_stats_hp_fwd multiplies with
    int* sat;                                         // f16-pieces kernels: saturation flags of the calling thread (common.h), may be null
    const float* xs_src;                              // f16-pieces kernels: the record word the x scale 2^s is chosen from (common.h: act_scale), null: 2^4
    unsigned* xmax_dst;                               // ... and where the launch leaves the largest |x| it converted, null: nowhere
    // ---- direction-serial form with the Linear folded in (dprnn_rnn_lin_fwd; f16-pieces eight-wave kernels, recur_split8.h: PJ).
    // One launch runs ONE direction (dir0; the grid has no direction axis) and multiplies every h_t it forms, while it is the
    // operand of the next step's gate GEMM in LDS, with this direction's half of the Linear that follows the BiLSTM
    // (dprnn.py:88,96): phase 1 leaves the partial products W[:, dir H : (dir + 1) H] h_t as fp32 rows in P, phase 2 adds its
    // own half and the bias to them, writes z to Zout (may be P: every element is read and rewritten by one lane) and adds the
    // GroupNorm sums of z to stats_out.  H is not written: h never leaves the CU.
    int proj;                                         // 0: none; 1: partial products -> P; 2: P (may be null) + own half + bias -> Zout, sums -> stats_out
    int dir0;                                         // direction this launch runs when proj != 0 (weights, bias, time order)
    const void* Lp;                                   // packed Linear (dprnn_lin_pack_f16): f16 piece fragments per direction + the scale words
    const float* lbias;                               // Linear bias [F] (phase 2)
    float* P;
    float* Zout;
    double* stats_out;                                // [utterances][2], accumulated (atomics)
    int proj_dirs;                                    // directions dir0 .. dir0 + proj_dirs - 1 in one launch (2: the time-split form below)
    // ---- the same fold with BOTH directions in every launch (dprnn_rnn_lin_split_fwd): the path is cut in time instead of by direction.
    // split_m >= 0 (the cut is M = T / 2 of the workgroup's own T): phase 1 runs the forward direction's steps [0, M) and the reverse direction's [0, T - M) (rows [0, M) and
    // [M, T): each row met by exactly one direction, which leaves x' and its partial product), phase 2 the remaining steps of both
    // (each row met by the other direction, which finishes z).  state: [dir][tile][512 x NT x 4 floats of c | the h image] carried
    // across the kernel boundary.  -1: no split.
    int split_m;
    float* state;
};

// packed Linear of the direction-serial form: [dir][wave 8][k-slab 4][piece 2][lane 64] 16-byte fragments, then the scale words
constexpr int LINP_FRAG_FLOATS = 2 * 8 * 4 * 2 * 64 * 4;      // both directions
constexpr int LINP_SCALE_WORD = LINP_FRAG_FLOATS;             // S_w, then 1 / (2^4 S_w)
constexpr int LINP_FLOATS = LINP_FRAG_FLOATS + 64;

// The recurrence kernels address every tensor as (scalar row of the step) + (32-bit per-lane byte offset of the lane's
// sequence inside its tile); offsets at or above 2^31 mean "no such sequence".  True when the widest tile (64 sequences)
// stays below that for a row pitch of pitch_floats: a tile of `tile` sequences (104 in the wide split kernel) spans at most
// ceil(tile / inner_n) outer strides.
inline bool tile_offsets_fit(int inner_n, long long outer_stride, long long inner_stride, long long t_stride, long long pitch_floats,
                             int tile = 104) {
    if (outer_stride < 0 || inner_stride < 0 || t_stride < 0) return false;
    const long long outer_steps = (tile - 1) / inner_n + 1, inner_steps = inner_n < tile ? inner_n - 1 : tile - 1;
    const long long span_rows = outer_steps * outer_stride + inner_steps * inner_stride + 1;
    return span_rows * pitch_floats * 4 < 0x7fffffffLL;
}

// split gate GEMMs (recur_split.hip): gemm_mode 1 = three bf16 pieces / six products, 3 = two f16 pieces / three products
int launch_bilstm_split(const RecurArgs& a, int gemm_mode, int seq_tile, bool fast, hipStream_t st);
int launch_split8_bf16(const RecurArgs& a, int nsr, bool fast, hipStream_t st);                            // recur_split8_bf16.hip
int launch_split8_f16(const RecurArgs& a, int nsr, bool fast, hipStream_t st);                             // recur_split8_f16.hip
int launch_split8_f16_train(const RecurArgs& a, bool fast, hipStream_t st);                                // ... with the gate tape (a.tape)
}  // namespace recur
