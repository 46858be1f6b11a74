// Eight-wave split recurrence kernel (template) and its launcher; instantiated by recur_split8_bf16.hip (three bf16 pieces,
// gate-GEMM mode 1) and recur_split8_f16.hip (two f16 pieces, mode 3) so that the two sets build in parallel.
#pragma once
#include "recur_split.h"

namespace rsplit {

// ------------------------------------------------------------------------------------------------------------------------
// EIGHT-WAVE form (one 512-thread workgroup per CU, NSR = 48 .. 96 sequences, wave w owns the 16 hidden units of weight
// fragment column tile (w >> 1, w & 1) for all four gates).  Why: profiles/r02_split_wide.txt -- with ONE wave per SIMD
// every weight request (buffer_load_dwordx4, ~33 cycles of issue) stops that SIMD's MFMA stream, 27 % of the MFMA phase
// at 64 sequences; with two waves per SIMD one wave's memory and vector instructions issue under the other's MFMAs.
//   per step: [gate GEMM (8 k-slabs: x half, then h half)] [cell update + staging of x_{t+1}] barrier
// (Round 3 also measured a skewed loop -- the x half of step t+1 before the barrier, so that one wave's cell update runs under its
// SIMD partner's MFMAs: correct, 2 % slower; the per-phase stamps and the reasons are in profiles/r03_recur_skew.txt.)
// (Round 3 also measured an ASYMMETRIC order -- waves 0-3 [h half][cell][x half of t+1], waves 4-7 [x half][h half][cell], x staged two
// steps ahead behind an LDS counter, so that one wave of every SIMD multiplies while the other does vector work: correct, 2 % slower;
// a wave alone on the matrix pipe needs 25 cycles per MFMA, not 16, its weight requests stall its own in-order stream:
// profiles/r03_recur_asym.txt.)
// (A staggered form -- cell update of one wave group beside the x half of the other's next step -- was measured in round 2
// and removed in round 3: two accumulator sets spill at 64 sequences, 15.1 against 11.6 us per step; profiles/r02_split_wide.txt.)
// Image: regions h (x 2) | x (x 2) when double-buffered, each [piece][NSR][256 B], 16-byte chunks XOR-swizzled by (row & 15).
#ifndef DPRNN_SPLIT8_DB
#define DPRNN_SPLIT8_DB 1
#endif
#ifndef DPRNN_SPLIT8_PRIO
#define DPRNN_SPLIT8_PRIO 1
#endif
#ifndef DPRNN_SPLIT8_DIAG
#define DPRNN_SPLIT8_DIAG 0                            // energy probes of tools/energy_budget.py (results are WRONG, timing and energy valid); a bit mask:
#endif                                                 //   1 no weight stream from L2 | 2 a quarter of the LDS fragment reads | 4 one MFMA product of the first k-slab only (1 / 24)
                                                       //   8 no cell arithmetic (h = clamp(scaled sum of the four gate accumulators)) | 16 no x staging at all (HBM loads, norm, split, LDS writes, x' store)
                                                       //   32 no h store to HBM | 64 no x' store | 128 the x / z loads of every step read step 0's rows: L2 hits instead of HBM reads
                                                       //   1024 (PJ) no store of the folded Linear's rows | 2048 (PJ = 2) no load of the partial products
constexpr bool diag(int bit) { return (DPRNN_SPLIT8_DIAG & bit) != 0; }
#ifndef DPRNN_SPLIT8_NT
#define DPRNN_SPLIT8_NT 1                              // experiment: non-temporal cache policy (aux bit 1) on the activation streams of the PJ forms:
#endif                                                 //   1 x / z loads | 2 x' stores | 4 stores of the folded Linear's rows | 8 loads of the partial products
constexpr int nt_aux(int bit) { return (DPRNN_SPLIT8_NT & bit) ? 2 : 0; }
#ifndef DPRNN_SPLIT8_DEFER_XS
#define DPRNN_SPLIT8_DEFER_XS 1                        // launches with a pending norm: the x' rows staged at the end of a step are STORED after the first k-slab of
#endif                                                 // the next step -- behind, not in front of, that step's first weight requests in the in-order vmcnt queue
                                                       // (round 5: +1.2 % on the forward; 1 = the folded (PJ) forms, 2 = also the unfolded inference forms)
#ifndef DPRNN_CELL_SHARED_RCP
#define DPRNN_CELL_SHARED_RCP 1                        // hardware-activation builds of the TRANS forms: two reciprocals per unit-step instead of five
#endif
#ifndef DPRNN_SPLIT8_RES32
#define DPRNN_SPLIT8_RES32 1
#endif
#ifndef DPRNN_SPLIT8_LRES
#define DPRNN_SPLIT8_LRES 3
#endif
constexpr bool split8_double_buffered(int nsr, bool f16) { return DPRNN_SPLIT8_DB && f16 && nsr <= 64 && !(nsr == 16 && DPRNN_SPLIT8_LRES > 3); }
// 16-sequence f16 tile: second weight pieces of the LAST `lres` k-slabs live in LDS (32 KB per slab and direction)
constexpr int split8_lds_resident_slabs(int nsr, bool f16) { return (f16 && nsr == 16) ? DPRNN_SPLIT8_LRES : 0; }

// SAVE (training forward, dprnn_rnn_fwd_train in gate-GEMM mode 3): every step also writes the activated gates and the cell
// state to a.tape in the layout of the fp32 training kernel (recur.hip: [dir][32-sequence workgroup][step][wave][nt][u][slot]
// [lane] float4 over four sequences), which is what the backward kernels read: a 64-sequence tile is two of those workgroups.
// PJ (direction-serial form, RecurArgs::proj; dprnn_rnn_lin_fwd): the launch runs one direction and carries a FIFTH weight set through
// the h half of the gate GEMM -- this direction's half of the Linear that follows the BiLSTM, 16 output columns per wave.  The A
// fragments are the ones the h half reads anyway (h_{t-1} as f16 pieces in the LDS image), so the fold costs 12 MFMAs per 16
// sequences, wave and step (+12.5 %), 8 KB of weight pieces per wave and step from L2, and no LDS traffic; the product for time
// t is complete after step t + 1's h half and leaves in that step's cell phase (one more h half after the loop for the last one).
// PJ = 1 writes the partial product (fp32, unscaled) to a.P; PJ = 2 starts its accumulators from a.P (x 2^4 S_w: exact), adds the
// bias, writes z to a.Zout and keeps the GroupNorm sums: fp32 over 8 steps x 4 columns x 4 lanes (128 elements, as in
// proj_f16_kernel), then fp64 per (wave, sequence) in LDS, one fp64 atomic pair per utterance of the tile at the end.
template <int NSR, int NP, bool FAST, bool FUSE, bool STAMP = false, bool F16 = false, bool SAVE = false, int PJ = 0>
__global__ __launch_bounds__(512, 2)
void bilstm_split8_kernel(RecurArgs a) {
    static_assert(!SAVE || NSR == 64, "the tape layout is written for 64-sequence tiles");
    static_assert(PJ == 0 || (F16 && F16_UNITS_CONTIGUOUS && !SAVE && !STAMP && NP == 3), "the folded Linear exists in the f16-pieces inference kernels");
    static_assert(NSR % 16 == 0 && NSR >= 16 && NSR <= 96, "whole 16-row tiles");
    static_assert(NP == 6 || (F16 && NP == 3), "three bf16 pieces / six products, or two f16 pieces / three products");
    constexpr int NT = NSR / 16;
    constexpr int NPC = NP == 6 ? 3 : 2;
    // UC: the f16 pack's unit order (recur_split.h).  TRANS: weights as the MFMA's ROW operand, so the accumulator quad of lane
    // (r, q) is units 4q .. 4q + 3 of sequence 16 nt + r instead of sequences 4q .. 4q + 3 of unit r: per tile the h pieces are two
    // 8-byte LDS stores (were eight 2-byte ones) and h one 16-byte global store (were four 4-byte ones, in the vmcnt queue in front
    // of the next step's weight loads).  The tape of the training forward is laid out over sequence quads, so SAVE keeps the
    // sequence-major accumulators.
    constexpr bool UC = F16 && F16_UNITS_CONTIGUOUS;
    constexpr bool TRANS = UC && !SAVE;
    constexpr int PL = NSR * 256;                      // bytes of one piece plane of one region
    constexpr int REG = NPC * PL;                      // bytes of one region
    // DB (f16 pieces, up to 64 sequences: 128 KB): h and x images double-buffered by step parity, which removes the barrier
    // between the gate GEMM and the cell update -- a wave that is through its MFMAs starts its cell update while its SIMD
    // partner still multiplies (they share the matrix pipe, so one of them always finishes first), instead of both waiting
    // and then both doing vector work with the matrix pipe idle.
    constexpr bool DB = split8_double_buffered(NSR, F16);
    constexpr int NHB = DB ? 2 : 1;                    // h buffers
    constexpr int NXB = DB ? 2 : 1;                    // x buffers
    extern __shared__ __attribute__((aligned(16))) char img[];
    char* const hreg = img;                            // + hb * REG
    char* const xreg = img + NHB * REG;                // + xb * REG
    unsigned* const rowoff_x = reinterpret_cast<unsigned*>(img + (NHB + NXB) * REG);
    unsigned* const rowoff_h = rowoff_x + NSR;
    float* const musig = reinterpret_cast<float*>(rowoff_h + NSR);
    float* const gb = musig + 2 * NSR;
    constexpr int LRES = split8_lds_resident_slabs(NSR, F16);       // second pieces of k-slabs NSLAB - LRES .. NSLAB - 1 in LDS
    char* const wlds = reinterpret_cast<char*>(gb + 256);                 // [slab][w4][gate*2 + wu][lane] 16-byte fragments
    double* const pstat = reinterpret_cast<double*>(wlds + LRES * 32768);     // PJ == 2: [wave][NSR][2] sums of z
    // PJ: the (pre-scaled) gate biases and the Linear's bias live in LDS instead of 20 registers -- the fifth accumulator set and its
    // weight fragments take their place (kernel_resources.sh: 11 spilled registers otherwise)
    constexpr bool BIAS_LDS = PJ != 0;
    float* const bias_l = reinterpret_cast<float*>(pstat + (PJ == 2 ? 8 * NSR * 2 : 0));      // [gate][HD], then [F] of the Linear

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int w4 = wave >> 1, wu = wave & 1;
#if DPRNN_SPLIT8_PRIO
    if (DB && wave < 4) __builtin_amdgcn_s_setprio(DPRNN_SPLIT8_PRIO);     // one wave of every SIMD pair runs ahead through the MFMAs
#endif
    const int dir = PJ ? a.dir0 + (int)blockIdx.y : (int)blockIdx.y;
    const bool split = PJ != 0 && a.split_m >= 0;      // both directions per launch, the path cut in time (below: sb, se)
    const int seq0 = blockIdx.x * NSR;
    constexpr unsigned OOB = 0x80000000u;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    // Ragged batch, inter-chunk path: this workgroup's sequences, step count and utterance come from its descriptor
    // (tiles never straddle utterances there, their lengths differ); otherwise the uniform addressing of the launch.
    const int* const desc = a.wg_desc ? a.wg_desc + 8 * blockIdx.x : nullptr;
    const int T = __builtin_amdgcn_readfirstlane(desc ? desc[2] : a.T);
    const int nhere = desc ? desc[3] : (a.nseq - seq0 < NSR ? a.nseq - seq0 : NSR);      // sequences that exist in this tile
    const long long row0 = desc ? (((long long)desc[1] << 32) | (unsigned)desc[0]) : rowbase(seq0);
    const int hpitch = PJ ? F : a.ndir * HD;           // PJ: rowoff_h addresses the rows of P / Zout
    if constexpr (PJ == 2) for (int i = tid; i < 8 * NSR * 2; i += 512) pstat[i] = 0.0;
    for (int j = tid; j < NSR; j += 512) {
        const bool valid = j < nhere;
        const int seq = valid ? seq0 + j : seq0;
        const long long ro = desc ? (long long)(valid ? j : 0) * a.inner_stride : rowbase(seq) - row0;
        rowoff_x[j] = (unsigned)(ro * F * 4);          // rows that do not exist alias the first row: loads stay in range
        rowoff_h[j] = valid ? (unsigned)(ro * hpitch * 4) : OOB;
        if (FUSE) {
            const int b = desc ? desc[4] : (a.seq_utt ? a.seq_utt[seq] : seq / a.seqs_per_utt);
            const double cnt = (double)(a.utt_rows ? a.utt_rows[b] : a.rows_per_utt) * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            musig[2 * j] = (float)mean;
            musig[2 * j + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
    }
    if (FUSE && tid < 128) { gb[tid] = a.gamma[tid]; gb[128 + tid] = a.beta[tid]; }
    const int c4 = tid & 31, xrow0 = tid >> 5;         // x quad: row xrow0 + 16 m, floats 4 c4 .. 4 c4 + 3
    const int xl = xrow0 * 256 + (((c4 >> 1) ^ xrow0) << 4) + ((c4 & 1) << 3);
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };
    const int unit0 = TRANS ? 32 * w4 + 16 * wu + 4 * q : UC ? 32 * w4 + 16 * wu + r : 32 * w4 + 2 * r + wu;   // this lane's (first) hidden unit

    float bias[4][(TRANS && !BIAS_LDS) ? 4 : 1];
    if constexpr (BIAS_LDS) {
        for (int i = tid; i < 4 * HD; i += 512) {
            const float b = a.bsum[dir * 4 * HD + i];
            bias_l[i] = FAST ? b * ((i >> 7) == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
        }
        if (PJ == 2 && tid < F) bias_l[4 * HD + tid] = a.lbias[tid];
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < (TRANS ? 4 : 1); ++e) {
                const float b = a.bsum[dir * 4 * HD + g * HD + unit0 + e];
                bias[g][e] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
            }
    }
    auto bget = [&](int g) __attribute__((always_inline)) {        // TRANS forms: the four units' bias of gate g
        if constexpr (BIAS_LDS) return *reinterpret_cast<const f32x4*>(bias_l + g * HD + unit0);
        else return f32x4{bias[g][0], bias[g][TRANS ? 1 : 0], bias[g][TRANS ? 2 : 0], bias[g][TRANS ? 3 : 0]};
    };
    // f16 form: the accumulators are S_x S_w times the gate pre-activations; inv = 1 / (S_x S_w) rides on the multiply the
    // nonlinearity starts with anyway
    const float inv = F16 ? a.Wp[F16_SCALE_WORD + 1] : 1.f;
    const float ksig = FAST ? -1.4426950408889634f * inv : inv, ktanh = FAST ? 2.8853900817779268f * inv : inv;
    const int w4_u = __builtin_amdgcn_readfirstlane(w4), wu_u = __builtin_amdgcn_readfirstlane(wu);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(reinterpret_cast<const u32x4*>(a.Wp) + (size_t)(dir * 4 + w4_u) * NSLAB * 8 * 3 * 64), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16u;
    auto wload = [&](int slab, int g, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)(((slab * 8 + g * 2 + wu_u) * 3 + pc) * 1024), 0);
    };
    auto rsrc_at = [&](const float* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffffu, 0x00020000);
    };

    if constexpr (LRES > 0) {                              // whole workgroup: copy the direction's fragments once
        const u32x4* const src = reinterpret_cast<const u32x4*>(a.Wp) + (size_t)dir * 4 * NSLAB * 8 * 3 * 64;
        for (int i = tid; i < LRES * 4 * 8 * 64; i += 512) {
            const int ln = i & 63, gu = (i >> 6) & 7, ww = (i >> 9) & 3, sl = i >> 11;
            reinterpret_cast<u32x4*>(wlds)[i] = src[(((size_t)ww * NSLAB + (NSLAB - LRES + sl)) * 8 + gu) * 3 * 64 + 1 * 64 + ln];
        }
    }
    auto wlds_load = [&](int slab, int g) __attribute__((always_inline)) {      // second piece of (slab, gate) of this wave
        return *reinterpret_cast<const u32x4*>(wlds + ((((slab - (NSLAB - LRES)) * 4 + w4) * 8 + g * 2 + wu) * 64 + lane) * 16);
    };
    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    unsigned satm = 0u;                                    // f16 pieces: bits of the largest |x| this lane converted (common.h: sat_track)
    // x is split as 2^s x with s from the launch's record word (common.h: adaptive activation scale; 2^4 without one).  h keeps
    // 2^4 -- |h| < 1, and its pieces are also the projection's operand -- so the pack kernel has multiplied W_ih by 2^(4 - s):
    // the accumulators come out scaled by 2^4 S_w whatever s is.
    const float xs = F16 ? act_scale(a.xs_src) : 1.f;
    // x of one time step -> bf16 pieces in x buffer xb, one quad (4 floats of one row) per thread and m; XG quads are in
    // flight together (registers: the partner wave on the SIMD covers the rest of the latency)
    auto x_stage = [&](long long t, int xb, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
        constexpr int XG = 2;
#pragma unroll
        for (int m0 = 0; m0 < NT; m0 += XG) {
            f32x4 xv[XG], zv[XG];
            unsigned xo[XG];
            fence();
#pragma unroll
            for (int i = 0; i < XG; ++i) {
                const int m = m0 + i < NT ? m0 + i : NT - 1;
                xo[i] = rowoff_x[xrow0 + 16 * m] + c4 * 16u;
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[i], 0, 0));
                if (FUSE) zv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo[i], 0, 0));
            }
#pragma unroll
            for (int i = 0; i < XG; ++i) {
                const int m = m0 + i;
                if (m >= NT) break;
                f32x4 v = xv[i];
                if (FUSE) {
                    const float mu = musig[2 * (xrow0 + 16 * m)], rs = musig[2 * (xrow0 + 16 * m) + 1];
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zv[i][e] - mu) * rs * gam[e] + bet[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && (dir == 0 || split)) ? xo[i] : OOB, 0, 0);
                }
                unsigned pa[3], pb[3];
                if constexpr (F16) {
                    sat_track(satm, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * xs, -F16_XMAX, F16_XMAX);
                    split2_pair_f16(v[0], v[1], pa[0], pa[1]);
                    split2_pair_f16(v[2], v[3], pb[0], pb[1]);
                } else {
                    split3_pair<NPC>(v[0], v[1], pa[0], pa[1], pa[2]);
                    split3_pair<NPC>(v[2], v[3], pb[0], pb[1], pb[2]);
                }
                char* d = xreg + xb * REG + xl + m * 4096;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x2*>(d + pc * PL) = u32x2{pa[pc], pb[pc]};
            }
        }
        fence();
    };

    // The same in two parts (XPF: plain form up to 64 sequences, where the registers allow it): the loads of x_{t+1} are
    // issued at the start of the LAST k-slab of step t and land under its MFMAs, barrier A and the cell update, instead of
    // being waited for right after they are issued.  (vmcnt retires in order: loads issued earlier in the step would sit in
    // front of every weight wait; from the last slab on, the only younger loads are next step's slab-0 weights.)
    constexpr bool XPF = NT <= 4;
    f32x4 xq[XPF ? NT : 1], zq[XPF ? NT : 1];
    auto x_issue = [&](long long t) __attribute__((always_inline)) {
        if (diag(128)) t = 0;                          // energy probe: every step loads the rows of step 0 -- L2 hits instead of HBM reads (results are wrong)
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
        fence();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const unsigned xo = rowoff_x[xrow0 + 16 * m] + c4 * 16u;
            xq[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, 0, PJ ? nt_aux(1) : 0));
            if (FUSE) zq[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo, 0, PJ ? nt_aux(1) : 0));
        }
        fence();
    };
    constexpr bool DEFER = FUSE && XPF && !SAVE && !STAMP && F16 && (DPRNN_SPLIT8_DEFER_XS >= 2 || (DPRNN_SPLIT8_DEFER_XS == 1 && PJ != 0));
    f32x4 xkeep[DEFER ? NT : 1];
    long long tkeep = 0;
    bool keep_live = false;
    auto x_flush = [&]() __attribute__((always_inline)) {
        if constexpr (DEFER) {
            if (keep_live) {
                const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(tkeep) * F);
#pragma unroll
                for (int m = 0; m < NT; ++m)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xkeep[m]), ors, rowoff_x[xrow0 + 16 * m] + c4 * 16u, 0, nt_aux(2));
                keep_live = false;
            }
        }
    };
    auto x_finish = [&](long long t, int xb, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
        if constexpr (DEFER) { tkeep = t; keep_live = live && (dir == 0 || split); }
        fence();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            f32x4 v = xq[m];
            if (FUSE) {
                const float mu = musig[2 * (xrow0 + 16 * m)], rs = musig[2 * (xrow0 + 16 * m) + 1];
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zq[m][e] - mu) * rs * gam[e] + bet[e]);
                const unsigned xo = rowoff_x[xrow0 + 16 * m] + c4 * 16u;
                if constexpr (DEFER) xkeep[m] = v;
                else if (!diag(64)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && (dir == 0 || split)) ? xo : OOB, 0, PJ ? nt_aux(2) : 0);
            }
            unsigned pa[3], pb[3];
            if constexpr (F16) {
                sat_track(satm, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * xs, -F16_XMAX, F16_XMAX);
                split2_pair_f16(v[0], v[1], pa[0], pa[1]);
                split2_pair_f16(v[2], v[3], pb[0], pb[1]);
            } else {
                split3_pair<NPC>(v[0], v[1], pa[0], pa[1], pa[2]);
                split3_pair<NPC>(v[2], v[3], pb[0], pb[1], pb[2]);
            }
            char* d = xreg + xb * REG + xl + m * 4096;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x2*>(d + pc * PL) = u32x2{pa[pc], pb[pc]};
        }
        fence();
    };

    __syncthreads();                                   // row tables are in place
    // Steps [sb, se) of this direction's T.  PJ with a split point M = a.split_m (dprnn_rnn_lin_split_fwd): the path runs as two launches of
    // BOTH directions -- phase 1 the first M (forward) / T - M (reverse) steps, phase 2 the rest -- so every row is met first by one
    // direction in phase 1 (partial product) and then by the other in phase 2 (finished z), with the kernel boundary between them as
    // the only synchronisation; the cell state and the h image cross it through a.state.
    const int nfirst = split ? (dir ? T - (T >> 1) : (T >> 1)) : 0;       // M = T / 2 of THIS workgroup's T (ragged inter-chunk tiles differ)
    const int sb = (split && PJ == 2) ? nfirst : 0, se = (split && PJ == 1) ? nfirst : T;
    const int xb0 = DB ? (sb & 1) : 0;                 // image buffers the first step reads
    const long long tfirst = dir ? (T - 1 - sb) : sb;
    if (se > sb) {                                     // (a direction may have no step at all in one of the phases: T = 1)
        if constexpr (XPF) { x_issue(tfirst); x_finish(tfirst, xb0, true); }
        else x_stage(tfirst, xb0, true);
    }
    f32x4 c[NT];
    float* const state = (PJ != 0 && split) ? a.state + ((size_t)dir * gridDim.x + blockIdx.x) * (size_t)(512 * NT * 4 + REG / 4) : nullptr;
    if (PJ == 2 && split) {                            // phase 2: where phase 1 stopped
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[nt] = *reinterpret_cast<const f32x4*>(state + ((size_t)nt * 512 + tid) * 4);
        const u32x4* src = reinterpret_cast<const u32x4*>(state + 512 * NT * 4);
        for (int i = tid; i < REG / 16; i += 512) reinterpret_cast<u32x4*>(hreg + xb0 * REG)[i] = src[i];
    } else {
        for (int i = tid; i < REG / 4; i += 512) reinterpret_cast<unsigned*>(hreg)[i] = 0u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // A fragments: lane (r, q) of tile nt reads row 16 nt + r, chunk (4 s4 + q) ^ r of the region
    const int abase = r * 256 + ((q ^ (r & 3)) << 4);
    int sl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[j] = (j ^ ((r >> 2) & 3)) << 6;
    auto aload = [&](u32x4 (&af)[NT], const char* reg, int pc, int s4) __attribute__((always_inline)) {
        if (diag(2) && s4 != 0) return;                // energy probe: a quarter of the fragment reads (results are wrong)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt] = *reinterpret_cast<const u32x4*>(reg + pc * PL + nt * 4096 + abase + sl[s4]);
    };
    // h element (row 16 nt + 4q + e, this lane's unit): + nt * 4096 + piece * PL
    int hb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        hb[e] = UC ? (4 * q + e) * 256 + ((((w4 ^ q) << 2) | ((2 * wu + (r >> 3)) ^ e)) << 4) + 2 * (r & 7)      // unit 32 w4 + 16 wu + r
                   : (4 * q + e) * 256 + ((((w4 ^ q) << 2) | ((r >> 2) ^ e)) << 4) + 4 * (r & 3) + 2 * wu;       // unit 32 w4 + 2 r + wu
    // TRANS: units 4q .. 4q + 3 of the wave's 16 (chunk 4 w4 + 2 wu + (q >> 1), its half q & 1) in row 16 nt + r: 8 bytes per piece
    const int hbt = r * 256 + (((4 * w4 + 2 * wu + (q >> 1)) ^ r) << 4) + 8 * (q & 1);
    const unsigned hcol = (unsigned)(dir * HD + unit0) * 4u;

    // WRES (f16 pieces, 16-sequence tile: the low-latency form for a few utterances): a step of a small tile is bound by the
    // L2 -> CU stream of the direction's 512 KB of weight pieces, not by its 96 MFMAs per wave, so the FIRST pieces of this
    // wave's weights (128 registers) stay resident for the whole kernel and only the second pieces are streamed: half the bytes
    // per step.
    // The 32-sequence tile has registers for half of them (k-slabs 0 .. 3).
    constexpr int RES = F16 ? (NSR == 16 ? NSLAB : (NSR == 32 && DPRNN_SPLIT8_RES32) ? NSLAB / 2 : 0) : 0;   // resident k-slabs
    constexpr bool WRES = RES > 0;
    u32x4 wres[WRES ? RES : 1][4];
    if constexpr (WRES) {
#pragma unroll
        for (int s = 0; s < RES; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) wres[s][g] = wload(s, g, 0);
    }
    u32x4 bw[4][NPC];                                  // B fragments of the k-slab in progress (single-buffered: a piece is
#pragma unroll                                         // re-requested for the next slab when its last product is done)
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pc = WRES ? 1 : 0; pc < NPC; ++pc) bw[g][pc] = wload(0, g, pc);       // slab 0 is resident whenever any is
    f32x4 acc[NT][4];
    u32x4 fa[NT], fb[NT];
    // ---- PJ: the Linear's fragments of this wave (16 output columns), one k-slab of two pieces in registers
    const float inv_l = PJ ? reinterpret_cast<const float*>(a.Lp)[LINP_SCALE_WORD + 1] : 1.f;      // 1 / (2^4 S_w)
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(reinterpret_cast<const u32x4*>(a.Lp) + (size_t)(PJ ? (dir * 8 + __builtin_amdgcn_readfirstlane(wave)) * 4 * 2 * 64 : 0)),
        0, PJ ? 0x7fffffff : 0, 0x00020000);
    auto pload = [&](int s4, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(prs, lane16, (unsigned)((s4 * 2 + pc) * 1024), 0);
    };
    u32x4 bp[2];
    f32x4 pacc[PJ ? NT : 1];
    float ps1[PJ == 2 ? NT : 1], ps2[PJ == 2 ? NT : 1];
    if constexpr (PJ != 0) {
        bp[0] = pload(0, 0); bp[1] = pload(0, 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (PJ == 2) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { ps1[nt] = 0.f; ps2[nt] = 0.f; }
    }
    auto ppass = [&](const u32x4 (&af)[NT], int wpc, bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pacc[nt] = mfma_f16(bp[wpc], af[nt], (first && PJ == 1) ? f32x4{0.f, 0.f, 0.f, 0.f} : pacc[nt]);
    };
    auto pass = [&](f32x4 (&ac)[NT][4], const u32x4 (&af)[NT], int wpc, bool first) __attribute__((always_inline)) {
        if (diag(4) && (wpc != 1 || !first)) return;   // energy probe: one product of the first k-slab only (results are wrong)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                ac[nt][g] = TRANS ? mfma_f16(bw[g][wpc], af[nt], first ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[nt][g])
                            : F16 ? mfma_f16(af[nt], bw[g][wpc], first ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[nt][g])
                                  : mfma_bf16(af[nt], bw[g][wpc], first ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[nt][g]);
    };
    auto wreload = [&](int sn, int wpc) __attribute__((always_inline)) {
        if (diag(1)) return;                           // energy probe: no weight stream (results are wrong)
        if (WRES && wpc == 0 && sn < RES) return;      // resident
        if (LRES > 0 && wpc == 1 && sn >= NSLAB - LRES) {      // from LDS
#pragma unroll
            for (int g = 0; g < 4; ++g) bw[g][1] = wlds_load(sn, g);
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[g][wpc] = wload(sn, g, wpc);
    };
    auto pass_res = [&](f32x4 (&ac)[NT][4], const u32x4 (&af)[NT], int sc) __attribute__((always_inline)) {     // first pieces of slab sc
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                ac[nt][g] = TRANS ? mfma_f16(wres[WRES && sc < RES ? sc : 0][g], af[nt], ac[nt][g])
                                  : mfma_f16(af[nt], wres[WRES && sc < RES ? sc : 0][g], ac[nt][g]);
    };
    // k-slab s4 of region reg (weights of global slab `s` are in bw; those of slab `sn` are requested as pieces die)
    auto slab = [&](f32x4 (&ac)[NT][4], const char* reg, u32x4 (&P)[NT], u32x4 (&Q)[NT], int s4, int sn, bool have_p, bool prefetch,
                    bool first, bool pj = false) __attribute__((always_inline)) {
        fence();
        if (!have_p) aload(P, reg, 0, s4);
        aload(Q, reg, 1, s4);
        fence();
        if constexpr (NP == 6) {
            pass(ac, P, 2, first); wreload(sn, 2); fence();
            pass(ac, P, 1, false); fence();
            pass(ac, Q, 1, false); wreload(sn, 1); fence();
            pass(ac, P, 0, false); fence();
            aload(P, reg, 2, s4);
            pass(ac, Q, 0, false); fence();
            if (prefetch) aload(Q, reg, 0, s4 + 1);
            pass(ac, P, 0, false); wreload(sn, 0); fence();
        } else {
            const int sc = (sn + NSLAB - 1) & (NSLAB - 1);      // the slab in progress
            pass(ac, P, 1, first);
            if (PJ && pj) { ppass(P, 1, s4 == 0); bp[1] = pload((s4 + 1) & 3, 1); }      // h half: k-slab s4 of the Linear rides along
            wreload(sn, 1); fence();
            if (WRES && sc < RES) pass_res(ac, P, sc); else pass(ac, P, 0, false);
            if (PJ && pj) ppass(P, 0, false);
            fence();
            if (prefetch) aload(P, reg, 0, s4 + 1);
            if (WRES && sc < RES) pass_res(ac, Q, sc); else pass(ac, Q, 0, false);
            if (PJ && pj) { ppass(Q, 0, false); bp[0] = pload((s4 + 1) & 3, 0); }
            wreload(sn, 0); fence();
        }
    };
    // four slabs of one region: global slabs s0 .. s0 + 3, the weights of slab (s0 + 4) & 7 are in flight at the end
    auto half = [&](f32x4 (&ac)[NT][4], const char* reg, int s0, bool first, bool xhook = false, long long tx = 0, bool pj = false, int flush = -1) __attribute__((always_inline)) {
        if constexpr (NP == 6) {
            slab(ac, reg, fa, fb, 0, s0 + 1, false, true, first);
            slab(ac, reg, fb, fa, 1, s0 + 2, true, true, false);
            slab(ac, reg, fa, fb, 2, s0 + 3, true, true, false);
            if (xhook) x_issue(tx);
            slab(ac, reg, fb, fa, 3, (s0 + 4) & 7, true, false, false);
        } else {
            slab(ac, reg, fa, fb, 0, s0 + 1, false, true, first, pj);
            if (flush == 0) x_flush();                 // (DEFER) last step's x' rows: behind this step's weight requests so far
            slab(ac, reg, fa, fb, 1, s0 + 2, true, true, false, pj);
            if (flush == 1) x_flush();
            slab(ac, reg, fa, fb, 2, s0 + 3, true, true, false, pj);
            if (flush == 2) x_flush();
            if (xhook) x_issue(tx);
            slab(ac, reg, fa, fb, 3, (s0 + 4) & 7, true, false, false, pj);
            if (flush == 3) x_flush();
        }
    };
    // PJ: the Linear alone over the h image `reg` (after the last step: its h has no gate GEMM to ride on)
    auto phalf = [&](const char* reg) __attribute__((always_inline)) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            fence();
            aload(fa, reg, 0, s4);
            aload(fb, reg, 1, s4);
            fence();
            ppass(fa, 1, s4 == 0);
            if (s4 < 3) bp[1] = pload(s4 + 1, 1);
            ppass(fa, 0, false);
            ppass(fb, 0, false);
            if (s4 < 3) bp[0] = pload(s4 + 1, 0);
            fence();
        }
    };
    // PJ: the finished products of time tp leave (phase 1: partial rows; phase 2: z = product + bias, sums), and phase 2 asks
    // for the partial rows of time tnext straight into the accumulators (scaled when the next h half begins)
    const unsigned zcol = (unsigned)(16 * wave + 4 * q) * 4u;
    auto pout = [&](long long tp, bool store, long long tnext, bool load) __attribute__((always_inline)) {
        fence();
        if (store) {
            const __amdgpu_buffer_rsrc_t ors = rsrc_at((PJ == 2 ? a.Zout : a.P) + tbase(tp) * F);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 v;
                if constexpr (PJ == 2) {
                    const f32x4 lbv = *reinterpret_cast<const f32x4*>(bias_l + 4 * HD + 16 * wave + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaf(pacc[nt][e], inv_l, lbv[e]);
                        ps1[nt] += v[e]; ps2[nt] = fmaf(v[e], v[e], ps2[nt]);
                    }
                } else {
                    v = pacc[nt] * f32x4{inv_l, inv_l, inv_l, inv_l};
                }
                if (!diag(1024)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, rowoff_h[16 * nt + r] + zcol, 0, nt_aux(4));
            }
        }
        if constexpr (PJ == 2) {
            if (load && a.P && !diag(2048)) {
                const __amdgpu_buffer_rsrc_t irs = rsrc_at(a.P + tbase(tnext) * F);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    pacc[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, rowoff_h[16 * nt + r] + zcol, 0, nt_aux(8)));
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) pacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        fence();
    };
    auto pscale = [&]() __attribute__((always_inline)) {      // phase 2: partial rows -> the accumulators' scale 2^4 S_w (a power of two)
        if constexpr (PJ == 2) {
            const float sl = reinterpret_cast<const float*>(a.Lp)[LINP_SCALE_WORD] * F16_XS;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) pacc[nt] = pacc[nt] * f32x4{sl, sl, sl, sl};
        }
    };
    auto pflush = [&]() __attribute__((always_inline)) {      // phase 2: fp32 sums of the last steps -> this wave's fp64 row of the table
        if constexpr (PJ == 2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float r1 = ps1[nt], r2 = ps2[nt];
                r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
                r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
                if (q == 0) {
                    double* d = pstat + ((size_t)wave * NSR + 16 * nt + r) * 2;
                    d[0] += (double)r1; d[1] += (double)r2;
                }
                ps1[nt] = 0.f; ps2[nt] = 0.f;
            }
        }
    };
    // LSTM cell of this wave's 16 units for every row, h -> image (3 pieces) and HBM
    // tape (SAVE): float index inside a 32-sequence workgroup's step = ((nt_b*2 + u)*5 + slot)*256 + wave_b*5120 + lane_b*4 + e with
    // wave_b = unit / 32, u = (unit % 32) / 16, lane_b = 16 q + unit % 16 -- per lane a constant plus (nt_b*10 + slot) KiB
    const int tv = UC ? 16 * wu + r : 2 * r + wu;       // unit index inside its 32-unit group
    const unsigned toff = (unsigned)(w4 * 5120 + (tv >> 4) * 5 * 256 + (q * 16 + (tv & 15)) * 4) * 4u;
    const int nwg_b = (a.nseq + 31) / 32;
    auto cell = [&](f32x4 (&ac)[NT][4], long long t, int step = 0, char* hw = nullptr) __attribute__((always_inline)) {
        if (!hw) hw = hreg;
        const __amdgpu_buffer_rsrc_t hrs = rsrc_at(a.H + tbase(t) * hpitch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            fence();
            float hv[4];
            f32x4 sv[SAVE ? 5 : 1];
            if constexpr (TRANS) {
                unsigned w1[2], w2[2];
                f32x2 cn2[2], hh[2];
                const f32x4 bl[4] = {bget(0), bget(1), bget(2), bget(3)};
                bool separate = true;                                  // the five-reciprocal forms (libm build, or the guard below)
                if constexpr (diag(8)) {                               // energy probe: no cell arithmetic (results are wrong)
                    separate = false;
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        cn2[p2] = f32x2{c[nt][2 * p2], c[nt][2 * p2 + 1]};
                        const f32x2 sg = (f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]} + f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]}) +
                                         (f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]} + f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]});      // every gate's MFMAs stay live
                        hh[p2] = f32x2{__builtin_amdgcn_fmed3f(sg[0] * ksig, -1.f, 1.f), __builtin_amdgcn_fmed3f(sg[1] * ksig, -1.f, 1.f)};
                    }
                } else if constexpr (FAST && DPRNN_CELL_SHARED_RCP) {
                    // shared reciprocals (common.h: lstm_cell_shared2): 5 v_exp_f32 + 2 v_rcp_f32 per unit-step instead of 5 + 5
                    float dmax = 0.f;
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const f32x2 kk = f32x2{ksig, ksig};
                        const f32x2 ai = f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]} * kk + f32x2{bl[0][2 * p2], bl[0][2 * p2 + 1]};
                        const f32x2 af = f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]} * kk + f32x2{bl[1][2 * p2], bl[1][2 * p2 + 1]};
                        const f32x2 ag = f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]} * f32x2{ktanh, ktanh} + f32x2{bl[2][2 * p2], bl[2][2 * p2 + 1]};
                        const f32x2 ao = f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]} * kk + f32x2{bl[3][2 * p2], bl[3][2 * p2 + 1]};
                        lstm_cell_shared2(ai, af, ag, ao, f32x2{c[nt][2 * p2], c[nt][2 * p2 + 1]}, cn2[p2], hh[p2], dmax);
                    }
                    // a denominator beyond 2^100 (a pre-activation below -60 or so; Inf included: the comparison fails) leaves the range in
                    // which the shared quotient is exact to fp32: that tile is recomputed with the separate forms (wave-uniform branch)
                    separate = __builtin_amdgcn_ballot_w64(!(dmax < 0x1p100f)) != 0ull;
                }
                if (separate) {
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const f32x2 ig = sigmoid_k2v<FAST>(f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]}, ksig, f32x2{bl[0][2 * p2], bl[0][2 * p2 + 1]});
                        const f32x2 fg = sigmoid_k2v<FAST>(f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]}, ksig, f32x2{bl[1][2 * p2], bl[1][2 * p2 + 1]});
                        const f32x2 gg = tanh_k2v<FAST>(f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]}, ktanh, f32x2{bl[2][2 * p2], bl[2][2 * p2 + 1]});
                        const f32x2 og = sigmoid_k2v<FAST>(f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]}, ksig, f32x2{bl[3][2 * p2], bl[3][2 * p2 + 1]});
                        cn2[p2] = fg * f32x2{c[nt][2 * p2], c[nt][2 * p2 + 1]} + ig * gg;
                        hh[p2] = og * tanh_f2<FAST>(cn2[p2]);
                    }
                }
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    c[nt][2 * p2] = cn2[p2][0]; c[nt][2 * p2 + 1] = cn2[p2][1];
                    hv[2 * p2] = hh[p2][0]; hv[2 * p2 + 1] = hh[p2][1];
                    split2_pair_f16(hh[p2][0] * F16_XS, hh[p2][1] * F16_XS, w1[p2], w2[p2]);      // pieces of units 4q + 2 p2, + 1
                }
                char* d = hw + hbt + nt * 4096;
                *reinterpret_cast<u32x2*>(d) = u32x2{w1[0], w1[1]};
                *reinterpret_cast<u32x2*>(d + PL) = u32x2{w2[0], w2[1]};
                // h_pieces: the next kernel is the f16-pieces projection, which would split 2^4 h into exactly these words again
                const u32x4 hout = a.h_pieces ? u32x4{w1[0], w1[1], w2[0], w2[1]}
                                              : u32x4{__float_as_uint(hv[0]), __float_as_uint(hv[1]), __float_as_uint(hv[2]), __float_as_uint(hv[3])};
                if (!diag(32) && PJ == 0) __builtin_amdgcn_raw_buffer_store_b128(hout, hrs, rowoff_h[16 * nt + r] + hcol, 0, 0);
                continue;
            }
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                f32x2 ig, fg, gg, og;
                if constexpr (F16) {
                    ig = sigmoid_k2<FAST>(f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]}, ksig, bias[0][0]);
                    fg = sigmoid_k2<FAST>(f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]}, ksig, bias[1][0]);
                    gg = tanh_k2<FAST>(f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]}, ktanh, bias[2][0]);
                    og = sigmoid_k2<FAST>(f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]}, ksig, bias[3][0]);
                } else {
                    ig = sigmoid_b2<FAST>(f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]}, bias[0][0]);
                    fg = sigmoid_b2<FAST>(f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]}, bias[1][0]);
                    gg = tanh_b2<FAST>(f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]}, bias[2][0]);
                    og = sigmoid_b2<FAST>(f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]}, bias[3][0]);
                }
                const f32x2 cn = fg * f32x2{c[nt][2 * p2], c[nt][2 * p2 + 1]} + ig * gg;
                const f32x2 h2 = og * tanh_f2<FAST>(cn);
                c[nt][2 * p2] = cn[0]; c[nt][2 * p2 + 1] = cn[1];
                hv[2 * p2] = h2[0]; hv[2 * p2 + 1] = h2[1];
                if constexpr (SAVE) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        sv[0][2 * p2 + k] = ig[k]; sv[1][2 * p2 + k] = fg[k]; sv[2][2 * p2 + k] = gg[k];
                        sv[3][2 * p2 + k] = og[k]; sv[4][2 * p2 + k] = cn[k];
                    }
                }
                unsigned w1, w2, w3 = 0u;              // pieces of rows e = 2 p2 (low halves) and 2 p2 + 1 (high halves)
                if constexpr (F16) split2_pair_f16(h2[0] * F16_XS, h2[1] * F16_XS, w1, w2);
                else split3_pair<NPC>(h2[0], h2[1], w1, w2, w3);
                char* d0 = hw + hb[2 * p2] + nt * 4096;
                char* d1 = hw + hb[2 * p2 + 1] + nt * 4096;
                *reinterpret_cast<unsigned short*>(d0) = (unsigned short)w1;
                *reinterpret_cast<unsigned short*>(d1) = (unsigned short)(w1 >> 16);
                *reinterpret_cast<unsigned short*>(d0 + PL) = (unsigned short)w2;
                *reinterpret_cast<unsigned short*>(d1 + PL) = (unsigned short)(w2 >> 16);
                if (NPC == 3) {
                    *reinterpret_cast<unsigned short*>(d0 + 2 * PL) = (unsigned short)w3;
                    *reinterpret_cast<unsigned short*>(d1 + 2 * PL) = (unsigned short)(w3 >> 16);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!diag(512)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv[e]), hrs, rowoff_h[16 * nt + 4 * q + e] + hcol, 0, 0);
            if constexpr (SAVE && !diag(256)) {                        // (probes 256 / 512: no tape / h stores -- what the stores cost)
                const int wg_b = 2 * blockIdx.x + (nt >> 1);           // the 32-sequence workgroup of the tape layout
                const __amdgpu_buffer_rsrc_t trs = rsrc_at(a.tape + (((size_t)dir * nwg_b + wg_b) * T + step) * 20480);
                const unsigned live = wg_b < nwg_b ? toff + (unsigned)((nt & 1) * 10 * 1024) : OOB;
                if (a.tape_packed) {                                   // wave-uniform: three words of fixed-point gates + c
                    u32x4 pk[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned w0, w1, w2;
                        tape_pack(sv[0][e], sv[1][e], sv[2][e], sv[3][e], w0, w1, w2);
                        pk[0][e] = w0; pk[1][e] = w1; pk[2][e] = w2;
                    }
#pragma unroll
                    for (int x = 0; x < 3; ++x) __builtin_amdgcn_raw_buffer_store_b128(pk[x], trs, live + (unsigned)(x * 1024), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv[4]), trs, live + 4096u, 0, 0);
                } else {
#pragma unroll
                    for (int x = 0; x < 5; ++x)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv[x]), trs, live + (unsigned)(x * 1024), 0, 0);   // soffset stays 0: see the store hazard note in recur.hip
                }
            }
        }
        fence();
    };

    unsigned long long st_t0 = 0, st_r0 = 0, st_p1 = 0, st_p2 = 0;
    {
        __syncthreads();
        if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
        for (int step = sb; step < se; ++step) {
            unsigned long long st_a = 0, st_b = 0;
            if constexpr (STAMP) { fence(); st_a = __builtin_amdgcn_s_memtime(); fence(); }
            const long long t = dir ? (T - 1 - step) : step;
            const bool more = step + 1 < se;
            const long long tn = more ? (dir ? (t - 1) : (t + 1)) : t;
            const int rb = DB ? (step & 1) : 0, wb = DB ? (rb ^ 1) : 0;      // image read in this step / written for the next
#ifndef DPRNN_SPLIT8_DEFER_AT
#define DPRNN_SPLIT8_DEFER_AT 1                        // k-slab (0 .. 7 over [x | h]) after which the deferred x' rows are stored
#endif
            half(acc, xreg + rb * REG, 0, true, false, 0, false, (DEFER && DPRNN_SPLIT8_DEFER_AT < 4) ? DPRNN_SPLIT8_DEFER_AT : -1);
            pscale();
            half(acc, hreg + rb * REG, 4, false, XPF && !diag(16), tn, PJ != 0, (DEFER && DPRNN_SPLIT8_DEFER_AT >= 4) ? DPRNN_SPLIT8_DEFER_AT - 4 : -1);
            if constexpr (STAMP) { fence(); st_b = __builtin_amdgcn_s_memtime(); fence(); st_p1 += st_b - st_a; }
            if constexpr (!DB) __syncthreads();        // single-buffered: every wave is done reading the image of this step
            if constexpr (PJ != 0) {                   // the Linear of h_{step-1} is complete: out; phase 2 asks for time t's partial rows
                pout(dir ? t + 1 : t - 1, step > sb, t, true);
                if (PJ == 2 && (step & 7) == 0) pflush();
            }
            cell(acc, t, step, hreg + wb * REG);
            if constexpr (diag(16)) { }                // energy probe: the image keeps the first step's x (results are wrong)
            else if constexpr (XPF) x_finish(tn, wb, more);
            else x_stage(tn, wb, more);
            if constexpr (STAMP) { fence(); st_p2 += __builtin_amdgcn_s_memtime() - st_b; fence(); }
            __syncthreads();
        }
    }
    x_flush();
    if constexpr (PJ != 0) {                           // the last step's h: the Linear alone, then the sums of the tile
        const char* const hlast = hreg + (DB ? (se & 1) : 0) * REG;
        if (se > sb) {
            pscale();
            phalf(hlast);
            pout(dir ? T - se : se - 1, true, 0, false);
        }
        if (PJ == 1 && split) {                        // phase 1: the state phase 2 continues from (ordered by the kernel boundary)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(state + ((size_t)nt * 512 + tid) * 4) = c[nt];
            u32x4* dst = reinterpret_cast<u32x4*>(state + 512 * NT * 4);
            for (int i = tid; i < REG / 16; i += 512) dst[i] = reinterpret_cast<const u32x4*>(hlast)[i];
        }
        if constexpr (PJ == 2) {
            pflush();
            __syncthreads();
            if (tid < ((NSR + 63) / 64) * 64) {        // whole waves: thread j = sequence j of the tile
                const int j = tid;
                double d1 = 0.0, d2 = 0.0;
                const bool valid = j < nhere;
                int b = -1;
                if (valid) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) { d1 += pstat[((size_t)w * NSR + j) * 2]; d2 += pstat[((size_t)w * NSR + j) * 2 + 1]; }
                    const int seq = seq0 + j;
                    b = desc ? desc[4] : (a.seq_utt ? a.seq_utt[seq] : seq / a.seqs_per_utt);
                }
                const int b0 = __builtin_amdgcn_readfirstlane(b);
                if (__builtin_amdgcn_ballot_w64(valid && b != b0) == 0ull && b0 >= 0) {      // one utterance (lane 0 is valid whenever any lane is)
                    const double s1 = wave_sum(d1), s2 = wave_sum(d2);
                    if (lane == 0) { atomicAdd(&a.stats_out[2 * b0], s1); atomicAdd(&a.stats_out[2 * b0 + 1], s2); }
                } else if (valid) {
                    atomicAdd(&a.stats_out[2 * b], d1); atomicAdd(&a.stats_out[2 * b + 1], d2);
                }
            }
        }
    }
    if constexpr (F16) sat_commit(a.sat, DPRNN_SAT_RECUR_X, satm, F16_XMAX / xs, a.xmax_dst);
    if constexpr (STAMP) {
        if ((tid == 0 || tid == 256) && a.stamps) {               // waves 0 and 4: a SIMD's pair; [workgroup][2][8] words
            unsigned long long* o = a.stamps + 16 * (blockIdx.y * gridDim.x + blockIdx.x) + (tid ? 8 : 0);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
            o[2] = st_p1;                                         // MFMA phase
            o[3] = st_p2;                                         // cell + staging
            o[4] = 0; o[5] = 0; o[6] = 0; o[7] = 0;
        }
    }
}

template <int NSR, int NP, bool FAST, bool FUSE, bool STAMP = false, bool F16 = false, bool SAVE = false, int PJ = 0>
int launch_split8(const RecurArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(split8_double_buffered(NSR, F16) ? 4 : 2) * (NP == 6 ? 3 : 2) * NSR * 256 +
                       (size_t)(NSR + NSR + 2 * NSR + 256) * 4 + (size_t)split8_lds_resident_slabs(NSR, F16) * 32768 +
                       (PJ == 2 ? (size_t)8 * NSR * 2 * 8 : 0) + (PJ ? (size_t)(4 * 128 + 128) * 4 : 0);
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split8_kernel<NSR, NP, FAST, FUSE, STAMP, F16, SAVE, PJ>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split8): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    dim3 grid(a.wg_desc ? a.nseq : (a.nseq + NSR - 1) / NSR, PJ ? (a.proj_dirs > 1 ? a.proj_dirs : 1) : a.ndir);          // ragged inter-chunk launches pass their tile count in nseq
    hipLaunchKernelGGL((bilstm_split8_kernel<NSR, NP, FAST, FUSE, STAMP, F16, SAVE, PJ>), grid, dim3(512), lds, st, a);
    return dprnn_check_launch("bilstm_split8_kernel");
}

// One tile size of one piece format: FAST / FUSE / STAMP variants.  The libm-activation variants (dprnn_set_fast_activations(0))
// are built for the 64-sequence tile only; other tile sizes fall back to it (any tile size serves any launch).
template <int NSR, bool F16>
int launch_split8_form(const RecurArgs& a, bool fast, hipStream_t st) {
    constexpr int NP = F16 ? 3 : 6;
    const bool fuse = a.Z != nullptr;
    if constexpr (NSR == 64) {
        if (a.stamps) return launch_split8<64, NP, true, false, true, F16>(a, st);    // in-kernel stamps (tools/bench_recur.py --stamps): fast, no pending norm
        if (!fast) return fuse ? launch_split8<64, NP, false, true, false, F16>(a, st) : launch_split8<64, NP, false, false, false, F16>(a, st);
    } else {
        if (!fast) return launch_split8_form<64, F16>(a, false, st);
    }
    return fuse ? launch_split8<NSR, NP, true, true, false, F16>(a, st) : launch_split8<NSR, NP, true, false, false, F16>(a, st);
}

}  // namespace rsplit
