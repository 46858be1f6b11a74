// Fused BiLSTM recurrence for one DPRNN path (gfx950).
//
// Replaces SingleRNN.forward = nn.LSTM(128,128,bidirectional) (reference src/models/dprnn.py:23-37)
// called at dprnn.py:86 (intra-chunk) and dprnn.py:95 (inter-chunk).
//
// One workgroup (4 waves) owns NS = 16*NT sequences of ONE direction for all T steps; sequences are
// independent so there is no inter-workgroup traffic.  Per step it evaluates
//     G[seq][4*128] = [x_t | h_{t-1}] (NS x 256)  @  [W_ih | W_hh]^T (256 x 512)  + (b_ih + b_hh)
// on v_mfma_f32_16x16x4_f32 (exact fp32), i.e. the input projection is fused into the recurrence and the
// [M x 1024] gate pre-activations never touch HBM.
//   * A operand = sequences: x_t and h_{t-1} live in LDS as one [seq][x(128)|h(128)|pad] image (odd
//     stride 257: conflict-free ds_read_b32 fragment reads with the k-permutation below),
//     double-buffered, one barrier per step.
//   * B operand = weights: wave w owns hidden units [32w, 32w+32) of all four gates, so i/f/g/o of one
//     (seq, unit) land in the same lane/register and the cell update is lane-local.  Its 128 KB weight
//     slice is streamed from L2 every step as pre-packed 16-byte fragments (1 KB per wave-instruction),
//     double-buffered in registers one k-group ahead.
//   * k-permutation inside each 128-wide source (x or h): k = 64*(g>>2) + 16*q + 4*(g&3) + i for k-group
//     g<8, lane quarter q=lane>>4 and MFMA sub-step i<4, so the two lane-quarters of a 32-lane LDS group
//     are 16 floats apart (bank = (row + 16*(q&1)) mod 32: a bijection).
//   * c stays in registers for the whole sequence; h goes registers -> LDS (next step's operand) and
//     registers -> HBM (64-byte segments, merged in L2).
#include "recur.h"
#ifndef DPRNN_CELL_SHARED_RCP
#define DPRNN_CELL_SHARED_RCP 1       // inference builds of the LSTM cell: shared reciprocals (common.h: lstm_cell_shared2)
#endif
#ifndef DPRNN_CELL_PRIO
#define DPRNN_CELL_PRIO 1
#endif
#include <stdlib.h>
#include <type_traits>
using namespace recur;

namespace {

typedef const f32x4 __attribute__((address_space(1)))* gw_ptr;   // global address space pointer to packed weights

// packed[dir][wave][grp][slot*2+u][lane][i] = Wslot[dir][slot][32*wave + 16*u + (lane&15)][k], k over [x(128) | h(128)].
// The four accumulator slots per hidden unit are
//   LSTM (cell 0): i, f, g, o            rows slot*128+unit of [W_ih | W_hh], bias b_ih + b_hh
//   GRU  (cell 1): r, z, n_x, n_h        r, z as above; n_x = [W_in | 0] + b_in; n_h = [0 | W_hn] + b_hn
//                                        (n = tanh(n_x + r * n_h) keeps the two halves apart, torch.nn.GRU semantics)
//   RNN  (cell 2): a, -, -, -            [W_ih | W_hh], bias sum; slots 1..3 are not computed
__global__ void rnn_pack_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r,
                                float* __restrict__ packed, float* __restrict__ bsum, int ndir, int cell) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ndir * PACK_PER_DIR) return;
    const int i = idx & 3, lane = (idx >> 2) & 63, gu = (idx >> 8) & 7, grp = (idx >> 11) & 15;
    const int wave = (idx >> 15) & 3, dir = idx >> 17;
    const int slot = gu >> 1, u = gu & 1, j = lane & 15, q = lane >> 4;
    const int unit = 32 * wave + 16 * u + j;
    const int g = grp & 7;
    const int k = 64 * (g >> 2) + 16 * q + 4 * (g & 3) + i;
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    const bool xpart = grp < 8;
    float v = 0.f;
    if (cell == 0) {
        v = xpart ? wi[(slot * HD + unit) * F + k] : wh[(slot * HD + unit) * HD + k];
    } else if (cell == 1) {
        const int gate = slot < 2 ? slot : 2;
        const bool use = slot < 2 || (slot == 2 && xpart) || (slot == 3 && !xpart);
        if (use) v = xpart ? wi[(gate * HD + unit) * F + k] : wh[(gate * HD + unit) * HD + k];
    } else {
        if (slot == 0) v = xpart ? wi[unit * F + k] : wh[unit * HD + k];
    }
    packed[idx] = v;
    if (idx < ndir * 4 * HD) {
        const int d = idx / (4 * HD), n = idx % (4 * HD), sl = n / HD, un = n % HD;
        const float* bi = d ? b_ih_r : b_ih;
        const float* bh = d ? b_hh_r : b_hh;
        float b = 0.f;
        if (cell == 0) b = bi[n] + bh[n];
        else if (cell == 1) b = sl < 2 ? bi[n] + bh[n] : (sl == 2 ? bi[2 * HD + un] : bh[2 * HD + un]);
        else b = sl == 0 ? bi[un] + bh[un] : 0.f;
        bsum[idx] = b;
    }
}

// DIAG (diagnostic builds only, wrong results on purpose): 1 = every k-group re-reads weight group 0 (L1-resident:
// no L2 weight stream), 2 = A fragments read once per step instead of per k-group (no LDS operand traffic),
// 3 = cell update skipped (h = gate pre-activation), 4 = no HBM store of h.
// SAVE (training forward): every step also writes what the backward needs to a.tape in this kernel's own fragment order
// -- [dir][workgroup][step][wave][nt][u][slot][lane] float4 over the lane's four sequences; slots: LSTM i,f,g,o,c;
// GRU r,z,n,(W_hn h + b_hn),h; Elman RNN h -- which the backward kernels (train_recur.hip) read back with the same thread
// mapping: 16-byte coalesced stores and loads, no transposition.
// NW = 8 (16-sequence tile only): the low-latency form for small batches -- eight waves, each owning 16 hidden units
// (NU = 1 column tile per gate instead of 2), so a step carries half the MFMAs per wave; same packed weights (wave w reads
// the fragments of wave w/2, column tile w&1), same LDS image.
template <int NT, bool FAST, bool STAMP = false, int DIAG = 0, int SCHED = 0, int CELL = 0, bool SAVE = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NT <= 2 ? 2 : 1))
void bilstm_kernel(RecurArgs a) {
    static_assert(NW == 4 || (NW == 8 && NT == 1 && SCHED == 0 && !SAVE && !STAMP), "8 waves: 16-sequence inference tile only");
    constexpr int NU = 8 / NW;                        // 16-unit column tiles per wave and gate
    constexpr bool PRIO = DPRNN_CELL_PRIO;
    constexpr int THREADS = 64 * NW;
    constexpr int NTAPE = CELL == 2 ? 1 : 5;          // tape slots per (nt, u)
    constexpr int NS = 16 * NT;
    constexpr int NGA = CELL == 2 ? 1 : 4;            // accumulator slots in use (see rnn_pack_kernel)
    constexpr int NXV = NS * 32 / THREADS;            // float4 x loads per thread per step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const XH = smem;                           // [2][NS][XS]: row = x_t (128) | h_{t-1} (128)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wq = NW == 4 ? wave : (wave >> 1);      // quarter of the hidden units (the packed weights' wave index)
    const int ub = NW == 4 ? 0 : (wave & 1);          // first column tile of this wave inside the quarter
    const int ucol0 = 32 * wq + 16 * ub;              // first hidden unit of the wave
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NS;
    const int T = a.T;

    // Addressing: every row this workgroup touches is (first row of the workgroup) + a small per-lane offset, and all
    // rows advance by the same amount per step, so per-lane offsets are 32-bit constants and the per-step part is a
    // scalar base (SGPR adds, saddr-form global_load/store): no vector ALU work for addresses inside the step loop
    // -- fp32 VALU instructions are not hidden by fp32 MFMAs on gfx950, every one of them costs matrix time.
    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const long long row0 = rowbase(seq0 < a.nseq ? seq0 : 0);          // uniform
    constexpr unsigned OOB = 0x80000000u;             // buffer offset beyond every resource's range: stores dropped, loads return 0
    const int hpitch = a.ndir * HD;
    // ---- x staging descriptors: float4 f = tid + 256*m covers row f>>5, floats 4*(f&31)..+3
    unsigned xoff[NXV];                               // byte offset from X + row0*F
    int xlds[NXV];
    bool xval[NXV];
    const int c4 = tid & 31;
#pragma unroll
    for (int m = 0; m < NXV; ++m) {
        const int f = tid + THREADS * m, row = f >> 5;
        const int seq = seq0 + row;
        xval[m] = seq < a.nseq;
        xoff[m] = (unsigned)((rowbase(xval[m] ? seq : seq0) - row0) * F + c4 * 4) * 4u;
        xlds[m] = row * XS + c4 * 4;
    }
    // ---- rows of the accumulator tile this lane owns: seq row 16*nt + 4*q + reg
    unsigned hoff[NT][4];                             // byte offset from H + row0*hpitch
    bool hval[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int seq = seq0 + 16 * nt + 4 * q + e;
            hval[nt][e] = seq < a.nseq;
            hoff[nt][e] = hval[nt][e] ? (unsigned)((rowbase(seq) - row0) * hpitch + dir * HD + ucol0 + r) * 4u : OOB;
        }
    // ---- pending norm of the previous path, applied while x is staged
    const bool fuse = a.Z != nullptr;
    f32x4 gam = {1.f, 1.f, 1.f, 1.f}, bet = {0.f, 0.f, 0.f, 0.f};
    float mu[NXV], rs[NXV];
#pragma unroll
    for (int m = 0; m < NXV; ++m) { mu[m] = 0.f; rs[m] = 0.f; }
    if (fuse) {
        gam = *reinterpret_cast<const f32x4*>(a.gamma + 4 * c4);
        bet = *reinterpret_cast<const f32x4*>(a.beta + 4 * c4);
#pragma unroll
        for (int m = 0; m < NXV; ++m) {
            const int seq = seq0 + (tid >> 5) + (THREADS / 32) * m;
            const int b = (seq < a.nseq ? seq : seq0) / a.seqs_per_utt;
            const double cnt = (double)a.rows_per_utt * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            mu[m] = (float)mean;
            rs[m] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
    }
    // x' of one staged float4 (identity when nothing is pending)
    auto fix = [&](f32x4 x, f32x4 z, int m) {        // whole-vector arithmetic: packed fp32 instructions, two lanes each
        const f32x4 mu4 = {mu[m], mu[m], mu[m], mu[m]}, rs4 = {rs[m], rs[m], rs[m], rs[m]};
        return x + ((z - mu4) * rs4 * gam + bet);
    };
    // time index -> scalar bases
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };

    // bias enters through the activation's own first instruction (x*k + b*k is one v_fma), so the accumulators
    // start from an inline-constant zero instead of 64 v_mov per step
    float bias[4][NU];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float b = a.bsum[dir * 4 * HD + g * HD + ucol0 + 16 * u + r];
            // pre-scaled for the activation that consumes the slot directly (sigmoid: -log2e, tanh: 2*log2e);
            // the GRU candidate slots (2, 3) enter a product first and keep the plain bias
            const bool is_tanh = (CELL == 0 && g == 2) || CELL == 2;
            const bool plain = CELL == 1 && g >= 2;
            bias[g][u] = (FAST && !plain) ? b * (is_tanh ? 2.8853900817779268f : -1.4426950408889634f) : b;
        }

    // wave-uniform weight slice (SGPR base, global address space) + lane index: saddr-form loads, scalar address math
    const int wave_u = __builtin_amdgcn_readfirstlane(wq);
    const int ub_u = __builtin_amdgcn_readfirstlane(ub);
    // The wave's weight slice as a buffer resource (4 SGPRs): fragment x of k-group g is
    // buffer_load_dwordx4 v, lane*16, rsrc, soffset = g*8 KB + x KB -- the group offset lives in an SGPR, so no vector ALU
    // instruction computes a weight address inside the step loop (with 64-bit global addresses the compiler spent ~15 of them
    // per pair of k-groups, and fp32 vector instructions cost matrix time here).
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4*>(reinterpret_cast<const f32x4*>(a.Wp) + (size_t)(dir * 4 + wave_u) * NGRP * 8 * 64), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16u;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto wfrag = [&](int grp, int x) __attribute__((always_inline)) {     // x = 2*gate + column tile of the quarter
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)(grp * 8 * 1024 + x * 1024), 0);
        return __builtin_bit_cast(f32x4, v);
    };
    // this wave's fragment (gate gt, own column tile u) of k-group grp
    auto wown = [&](int grp, int gt, int u) __attribute__((always_inline)) { return wfrag(grp, 2 * gt + ub_u + u); };
    // The same for the activations: one buffer resource per tensor whose base is the row of the step (scalar), per-lane
    // 32-bit offsets fixed for the whole kernel.  A lane whose sequence does not exist gets an offset beyond the range
    // (OOB): the hardware drops its stores and returns zeros for its loads -- no exec-mask branches in the step loop.
    constexpr unsigned RANGE = 0x7fffffffu;
    auto rsrc_at = [&](const float* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, RANGE, 0x00020000);
    };
    auto bload4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) __attribute__((always_inline)) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    };
    auto bstore4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 0);
    };

    // ---- prologue: x_{t0} -> Xs[0], zeros -> Hs[0]
    {
        const long long t0 = dir ? (T - 1) : 0;
        const char* xs = reinterpret_cast<const char*>(a.X + tbase(t0) * F);
        const char* zs = reinterpret_cast<const char*>(a.Z + tbase(t0) * F);
        char* os = reinterpret_cast<char*>(a.Xout + tbase(t0) * F);
#pragma unroll
        for (int m = 0; m < NXV; ++m) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xval[m]) {
                v = *reinterpret_cast<const f32x4*>(xs + xoff[m]);
                if (fuse) {
                    v = fix(v, *reinterpret_cast<const f32x4*>(zs + xoff[m]), m);
                    if (dir == 0) *reinterpret_cast<f32x4*>(os + xoff[m]) = v;
                }
            }
            float* d = XH + xlds[m];
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        for (int i = tid; i < NS * HD; i += THREADS) XH[(i >> 7) * XS + HD + (i & 127)] = 0.f;
    }
    f32x4 c[NT][NU];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int u = 0; u < NU; ++u) c[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 w[2][4 * NU];                               // [buffer][gate * NU + column tile]
#pragma unroll
    for (int gt = 0; gt < NGA; ++gt)
#pragma unroll
        for (int u = 0; u < NU; ++u) w[0][gt * NU + u] = wown(0, gt, u);
    __syncthreads();

    unsigned long long st_t0 = 0, st_r0 = 0, st_mfma = 0, st_act = 0;
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int step = 0; step < T; ++step) {
        unsigned long long st_a = 0, st_b = 0;
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        const int cur = step & 1, nxt = cur ^ 1;
        // The SIMD arbiter favours the older of two co-resident waves, so one workgroup of a pair would run at its
        // stand-alone pace and finish ~15 % before the other, which then has the CU to itself at half occupancy.
        // Alternating the wave priority in blocks of steps shares the matrix pipe evenly.
        if constexpr (PRIO) { if (((step >> 3) + blockIdx.y) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        // keep "offset + constant" inside the loop body so the constant lands in the instruction's immediate field
        // instead of a second, hoisted copy of every offset register
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(hoff[nt][e]));
        // prefetch x_{t+1} into registers
        f32x4 xr[NXV], zr[NXV];
        const long long tn = dir ? (t - 1) : (t + 1);
        if (more) {
            // rows past nseq alias the tile's first row (xoff), so every lane loads unconditionally: no exec-mask
            // branches, no zero fills -- their h is never stored and a duplicated x' store carries identical bytes
            const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(tn) * F);
#pragma unroll
            for (int m = 0; m < NXV; ++m) xr[m] = bload4(xrs, xoff[m]);
            if (fuse) {
                const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(tn) * F);
#pragma unroll
                for (int m = 0; m < NXV; ++m) zr[m] = bload4(zrs, xoff[m]);
            }
        }
        f32x4 acc[NT][4][NU];
        const float* const Ac = XH + cur * NS * XS + r * XS + 16 * q;
        if constexpr (SCHED == 0) {
        // 16 k-groups, two per trip; the weights of group g+1 are in flight while group g computes
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[nt][g][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int gp = 0; gp < NGRP / 2; ++gp) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int grp = 2 * gp + half;
                // weights of the next k-group (wraps to group 0 of the next step: same bytes every step)
                const int gnext = DIAG == 1 ? 0 : ((grp + 1) & 15);
#pragma unroll
                for (int gt = 0; gt < NGA; ++gt)
#pragma unroll
                    for (int u = 0; u < NU; ++u) w[half ^ 1][gt * NU + u] = wown(gnext, gt, u);
                const float* src = Ac + (DIAG == 2 ? 0 : 64 * (grp >> 2) + 4 * (grp & 3));
                float av[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[nt][i] = src[nt * 16 * XS + i];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int gt = 0; gt < NGA; ++gt)
#pragma unroll
                            for (int u = 0; u < NU; ++u)
                                acc[nt][gt][u] = mfma16(av[nt][i], w[half][gt * NU + u][i], acc[nt][gt][u]);
            }
        }
        } else if constexpr (NU == 2) {
        static_assert(SCHED == 0 || CELL == 0, "the interleaved schedule is instantiated for the LSTM cell only");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[nt][g][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Interleaved schedule: weights AND A fragments of group g+1 are requested while group g computes, and
        // sched_group_barrier spreads the 8 VMEM + 2*NT DS instructions one per MFMA cluster instead of leaving
        // them bunched at the group boundary (each MFMA gap hides only a few other instructions).
        float av[2][NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) av[0][nt][i] = Ac[nt * 16 * XS + i];
#pragma unroll 1
        for (int gp = 0; gp < NGRP / 2; ++gp) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int grp = 2 * gp + half;
#pragma unroll
                for (int x = 0; x < 8; ++x) w[half ^ 1][x] = wfrag((grp + 1) & 15, x);
                const int gn = (grp + 1) & 15;             // group 15 prefetches group 0 of the old image: unused
                const float* src = Ac + 64 * (gn >> 2) + 4 * (gn & 3);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[half ^ 1][nt][i] = src[nt * 16 * XS + i];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                acc[nt][gt][u] = mfma16(av[half][nt][i], w[half][gt * 2 + u][i], acc[nt][gt][u]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 // 1 VMEM read
                    if (k < 2 * NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);            // 4*NT MFMA
                }
            }
        }
        }

        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); st_mfma += st_b - st_a; }
        // ---- LSTM cell (gate order i,f,g,o), lane-local
        float* const Hn = XH + nxt * NS * XS + HD;
        const __amdgpu_buffer_rsrc_t hrs = rsrc_at(a.H + tbase(t) * hpitch);
        // tape: this workgroup's steps are contiguous; entry i of the step lives in 4 KiB window i>>2 (one resource per
        // window, scalar base) at immediate offset (i&3) KiB.  The scalar-offset operand stays the constant 0 on purpose:
        // a 128-bit buffer store whose data registers are rewritten by the next VALU instruction needs a wait state,
        // and the compiler's hazard recognizer inserts it only for the immediate form -- with an SGPR offset the
        // o-gate entry of one tile came back corrupted on gfx950 (profiles/r01_recur_microbench.txt).
        constexpr unsigned TSTEP = 4u * (NT * 2 * NTAPE * 64) * 16u;
        const float* const tstep = SAVE ? a.tape + (((size_t)dir * gridDim.x + blockIdx.x) * T + step) * (TSTEP / 4) : nullptr;
        unsigned tlane = (unsigned)(wave * (NT * 2 * NTAPE * 64) + lane) * 16u;
        if constexpr (SAVE) asm volatile("" : "+v"(tlane));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                f32x4 sv[5];
                if constexpr (CELL == 0 && DIAG != 3) {
                    // LSTM, gate order i,f,g,o (torch.nn.LSTM), two sequences per instruction where the arithmetic allows
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        f32x2 ig, fg, gg, og, cn, h2;
                        bool separate = true;
                        if constexpr (FAST && !SAVE && DPRNN_CELL_SHARED_RCP && NT < 4) {      // (the 64-sequence tile has no register left for it: 6 spills)
                            // inference, hardware activations: two reciprocals per unit-step instead of five (common.h: lstm_cell_shared2).
                            // On the fp32 MFMA every vector instruction costs matrix time (the shared datapath), transcendentals most:
                            // 48 fewer v_rcp_f32 per wave and step.  The tape of the training forward wants the gates themselves, so
                            // SAVE keeps the separate forms.  A pair out of the shared quotient's range is recomputed (wave-uniform).
                            float dmax = 0.f;
                            const f32x2 ai = f32x2{acc[nt][0][u][2 * p], acc[nt][0][u][2 * p + 1]} * f32x2{-1.4426950408889634f, -1.4426950408889634f} + f32x2{bias[0][u], bias[0][u]};
                            const f32x2 af = f32x2{acc[nt][1][u][2 * p], acc[nt][1][u][2 * p + 1]} * f32x2{-1.4426950408889634f, -1.4426950408889634f} + f32x2{bias[1][u], bias[1][u]};
                            const f32x2 ag = f32x2{acc[nt][2][u][2 * p], acc[nt][2][u][2 * p + 1]} * f32x2{2.8853900817779268f, 2.8853900817779268f} + f32x2{bias[2][u], bias[2][u]};
                            const f32x2 ao = f32x2{acc[nt][3][u][2 * p], acc[nt][3][u][2 * p + 1]} * f32x2{-1.4426950408889634f, -1.4426950408889634f} + f32x2{bias[3][u], bias[3][u]};
                            lstm_cell_shared2(ai, af, ag, ao, f32x2{c[nt][u][2 * p], c[nt][u][2 * p + 1]}, cn, h2, dmax);
                            separate = __builtin_amdgcn_ballot_w64(!(dmax < 0x1p100f)) != 0ull;
                        }
                        if (separate) {
                            ig = sigmoid_b2<FAST>(f32x2{acc[nt][0][u][2 * p], acc[nt][0][u][2 * p + 1]}, bias[0][u]);
                            fg = sigmoid_b2<FAST>(f32x2{acc[nt][1][u][2 * p], acc[nt][1][u][2 * p + 1]}, bias[1][u]);
                            gg = tanh_b2<FAST>(f32x2{acc[nt][2][u][2 * p], acc[nt][2][u][2 * p + 1]}, bias[2][u]);
                            og = sigmoid_b2<FAST>(f32x2{acc[nt][3][u][2 * p], acc[nt][3][u][2 * p + 1]}, bias[3][u]);
                            cn = fg * f32x2{c[nt][u][2 * p], c[nt][u][2 * p + 1]} + ig * gg;
                            h2 = og * tanh_f2<FAST>(cn);
                        }
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int e = 2 * p + k;
                            c[nt][u][e] = cn[k];
                            if constexpr (SAVE) { sv[0][e] = ig[k]; sv[1][e] = fg[k]; sv[2][e] = gg[k]; sv[3][e] = og[k]; sv[4][e] = cn[k]; }
                            Hn[(16 * nt + 4 * q + e) * XS + ucol0 + 16 * u + r] = h2[k];
                            if (DIAG != 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(float(h2[k])), hrs, hoff[nt][e] + 64 * u, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float h;
                    if constexpr (DIAG == 3) {
                        h = acc[nt][0][u][e] + acc[nt][1][u][e] + acc[nt][2][u][e] + acc[nt][3][u][e] + bias[0][u];
                    } else if constexpr (CELL == 0) {     // LSTM, gate order i,f,g,o (torch.nn.LSTM)
                        const float ig = sigmoid_b<FAST>(acc[nt][0][u][e], bias[0][u]);
                        const float fg = sigmoid_b<FAST>(acc[nt][1][u][e], bias[1][u]);
                        const float gg = tanh_b<FAST>(acc[nt][2][u][e], bias[2][u]);
                        const float og = sigmoid_b<FAST>(acc[nt][3][u][e], bias[3][u]);
                        const float cn = fg * c[nt][u][e] + ig * gg;
                        c[nt][u][e] = cn;
                        h = og * tanh_f<FAST>(cn);
                        if constexpr (SAVE) { sv[0][e] = ig; sv[1][e] = fg; sv[2][e] = gg; sv[3][e] = og; sv[4][e] = cn; }
                    } else if constexpr (CELL == 1) {     // GRU: n = tanh(n_x + r*n_h), h' = (1-z)*n + z*h (c holds h)
                        const float rg = sigmoid_b<FAST>(acc[nt][0][u][e], bias[0][u]);
                        const float zg = sigmoid_b<FAST>(acc[nt][1][u][e], bias[1][u]);
                        const float nh = acc[nt][3][u][e] + bias[3][u];
                        const float ng = tanh_f<FAST>((acc[nt][2][u][e] + bias[2][u]) + rg * nh);
                        h = (1.f - zg) * ng + zg * c[nt][u][e];
                        c[nt][u][e] = h;
                        if constexpr (SAVE) { sv[0][e] = rg; sv[1][e] = zg; sv[2][e] = ng; sv[3][e] = nh; sv[4][e] = h; }
                    } else {                              // Elman RNN, tanh
                        h = tanh_b<FAST>(acc[nt][0][u][e], bias[0][u]);
                        if constexpr (SAVE) sv[0][e] = h;
                    }
                    Hn[(16 * nt + 4 * q + e) * XS + ucol0 + 16 * u + r] = h;
                    if (DIAG != 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), hrs, hoff[nt][e] + 64 * u, 0, 0);
                }
                }
                if constexpr (SAVE) {
#pragma unroll
                    for (int x = 0; x < NTAPE; ++x) {
                        const int i = (nt * 2 + u) * NTAPE + x;
                        bstore4(rsrc_at(tstep + (i >> 2) * 1024), tlane + (unsigned)((i & 3) * 1024), sv[x]);
                    }
                }
            }
        if (more) {
            float* const Xn = XH + nxt * NS * XS;
            const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(tn) * F);
#pragma unroll
            for (int m = 0; m < NXV; ++m) {
                f32x4 v = xr[m];
                if (fuse) {
                    v = fix(v, zr[m], m);
                    if (dir == 0) bstore4(ors, xoff[m], v);
                }
                float* d = Xn + xlds[m];
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        }
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_act += __builtin_amdgcn_s_memtime() - st_b; __builtin_amdgcn_sched_barrier(0); }
        __syncthreads();
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;          // shader cycles, whole step loop
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;      // 100 MHz ticks
            o[2] = st_mfma;                                       // cycles inside the k-loop (wave 0)
            o[3] = st_act;                                        // cycles in cell update + stores, before the barrier
            o[4] = st_r0;                                         // absolute start (100 MHz)
            o[5] = __builtin_amdgcn_s_memrealtime();              // absolute end
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            o[6] = xcc; o[7] = hwid;
        }
    }
}

template <int NT, bool FAST, int CELL = 0, bool SAVE = false, int NW = 4>
int launch_bilstm(const RecurArgs& a, hipStream_t st) {
    constexpr int NS = 16 * NT;
    // one workgroup per CU (NT >= 3): prefetch A fragments a k-group ahead and spread the memory instructions
    // between the MFMAs (measured +6 % at NT = 4); with two co-resident workgroups (NT <= 2) the plain loop is as fast
    constexpr int SCHED = (NT >= 3 && CELL == 0) ? 1 : 0;
    const size_t lds = (size_t)2 * NS * XS * sizeof(float);
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_kernel<NT, FAST, false, 0, SCHED, CELL, SAVE, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    dim3 grid((a.nseq + NS - 1) / NS, a.ndir);
    hipLaunchKernelGGL((bilstm_kernel<NT, FAST, false, 0, SCHED, CELL, SAVE, NW>), grid, dim3(64 * NW), lds, st, a);
    return dprnn_check_launch("bilstm_kernel");
}

}  // namespace

// Arithmetic of the LSTM gate GEMM: 0 exact fp32 MFMA, 1 = three bf16 pieces / six products, 3 = two f16 pieces / three products
// (recur_split.h).  The *_ex entry points take it as an argument; the plain ones read this per-THREAD default (two host threads
// driving two models never see each other's setting).  Whatever selects it, the format a pack call wrote is recorded with the
// buffer (dprnn_pack_note) and checked by every launch that reads it.
static thread_local int g_gemm_mode = 0;
static bool mode_ok(int m) { return m == 0 || m == 1 || m == 3; }
extern "C" void dprnn_set_gemm_mode(int mode) { g_gemm_mode = mode_ok(mode) ? mode : 0; }
extern "C" int dprnn_get_gemm_mode(void) { return g_gemm_mode; }
extern "C" int dprnn_lstm_pack_split(const float*, const float*, const float*, const float*, const float*, const float*,
                                     const float*, const float*, void*, float*, void*);
extern "C" int dprnn_lstm_pack_f16(const float*, const float*, const float*, const float*, const float*, const float*,
                                   const float*, const float*, void*, float*, void*);

static int rnn_pack(int mode, int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                    const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                    float* packed, float* bsum, void* stream) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !bsum) { dprnn_set_error("dprnn_rnn_pack: null pointer"); return -1; }
    if (cell < 0 || cell > 2) { dprnn_set_error("dprnn_rnn_pack: cell must be 0 (LSTM), 1 (GRU) or 2 (RNN tanh)"); return -1; }
    if (!mode_ok(mode)) { dprnn_set_error("dprnn_rnn_pack: gemm mode must be 0, 1 or 3 (got %d)", mode); return -1; }
    if (cell == 0 && mode == 3)              // f16 form: two f16 piece planes of 2^e * W
        return dprnn_lstm_pack_f16(w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, stream);
    if (cell == 0 && mode == 1)              // three bf16 piece planes instead of fp32 fragments
        return dprnn_lstm_pack_split(w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, stream);
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && (!w_hh_r || !b_ih_r || !b_hh_r)) { dprnn_set_error("dprnn_rnn_pack: incomplete reverse direction"); return -1; }
    const int n = ndir * PACK_PER_DIR;
    hipLaunchKernelGGL(rnn_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, ndir, cell);
    dprnn_pack_note(packed, DPRNN_PACK_FWD, cell, 0, ndir);
    return dprnn_check_launch("rnn_pack_kernel");
}

extern "C" int dprnn_rnn_pack(int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                              const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                              float* packed, float* bsum, void* stream) {
    return rnn_pack(g_gemm_mode, cell, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, stream);
}
extern "C" int dprnn_rnn_pack_ex(int gemm_mode, int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                 const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                 float* packed, float* bsum, void* stream) {
    return rnn_pack(gemm_mode, cell, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, stream);
}

extern "C" int dprnn_lstm_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                               const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                               float* packed, float* bsum, void* stream) {
    return dprnn_rnn_pack(0, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, stream);
}

static int g_fast_act = 1;   // v_exp_f32/v_rcp_f32 gate activations: 2e-7 rel-L2 end to end, same as the libm forms
extern "C" void dprnn_set_fast_activations(int on) { g_fast_act = on; }
extern "C" int dprnn_get_fast_activations(void) { return g_fast_act; }
int dprnn_fast_activations_enabled(void) { return g_fast_act; }      // train_recur.hip

static int bilstm_dispatch(RecurArgs a, int mode, int seq_tile, hipStream_t st) {
    if (!a.X || !a.Wp || !a.bsum || !a.H) { dprnn_set_error("dprnn_bilstm_fwd: null pointer"); return -1; }
    if (a.nseq <= 0 || a.T <= 0 || a.inner_n <= 0 || (a.ndir != 1 && a.ndir != 2)) {
        dprnn_set_error("dprnn_bilstm_fwd: bad sizes nseq=%d T=%d inner_n=%d ndir=%d", a.nseq, a.T, a.inner_n, a.ndir);
        return -1;
    }
    if (a.cell < 0 || a.cell > 2) { dprnn_set_error("dprnn_rnn_fwd: cell must be 0 (LSTM), 1 (GRU) or 2 (RNN tanh)"); return -1; }
    if (!mode_ok(mode)) { dprnn_set_error("dprnn_rnn_fwd: gemm mode must be 0, 1 or 3 (got %d)", mode); return -1; }
    if (a.cell != 0) mode = 0;                // GRU / Elman cells have the exact fp32 kernels only (and dprnn_rnn_pack packs them so)
    if (a.Z && (!a.stats || !a.gamma || !a.beta || !a.Xout || a.Xout == a.X || a.seqs_per_utt <= 0 || a.rows_per_utt <= 0)) {
        dprnn_set_error("dprnn_bilstm_norm_fwd: the pending norm needs stats, gamma, beta, sizes and an Xout different from X");
        return -1;
    }
    if (!tile_offsets_fit(a.inner_n, a.outer_stride, a.inner_stride, a.t_stride, a.ndir * HD)) {
        dprnn_set_error("dprnn_bilstm_fwd: a sequence tile spans more than 2 GiB of H (outer_stride=%lld rows): split the batch",
                        a.outer_stride);
        return -1;
    }
    if (int rc = dprnn_pack_expect(a.Wp, DPRNN_PACK_FWD, a.cell, mode, a.ndir, "dprnn_rnn_fwd")) return rc;
    a.sat = dprnn_sat_ptr();
    { const ScaleRef sr = dprnn_scale_ref(); a.xs_src = sr.src; a.xmax_dst = sr.dst; }
    if (seq_tile == 0) seq_tile = mode != 0 ? 2064 : (a.nseq <= 16 ? 16 : 32);
    const bool fast = g_fast_act != 0;
    // 16-sequence tiles run on eight waves (half the MFMA chain per wave and step: the small-batch latency form);
    // -16 selects the four-wave form of the same tile (kept for A/B measurements: tools/latency.py)
    if (a.cell == 1 || a.cell == 2) {       // GRU / Elman: 16- or 32-sequence tiles
        const bool small = seq_tile == 16 || seq_tile == -16;
        if (a.cell == 1) {
            if (small) return fast ? launch_bilstm<1, true, 1, false, 8>(a, st) : launch_bilstm<1, false, 1, false, 8>(a, st);
            return fast ? launch_bilstm<2, true, 1>(a, st) : launch_bilstm<2, false, 1>(a, st);
        }
        if (small) return fast ? launch_bilstm<1, true, 2, false, 8>(a, st) : launch_bilstm<1, false, 2, false, 8>(a, st);
        return fast ? launch_bilstm<2, true, 2>(a, st) : launch_bilstm<2, false, 2>(a, st);
    }
    if (mode != 0) return launch_bilstm_split(a, mode, seq_tile, fast, st);
    switch (seq_tile) {
        case 16: return fast ? launch_bilstm<1, true, 0, false, 8>(a, st) : launch_bilstm<1, false, 0, false, 8>(a, st);
        case -16: return fast ? launch_bilstm<1, true>(a, st) : launch_bilstm<1, false>(a, st);
        case 32: return fast ? launch_bilstm<2, true>(a, st) : launch_bilstm<2, false>(a, st);
        case 48: return fast ? launch_bilstm<3, true>(a, st) : launch_bilstm<3, false>(a, st);
        case 64:
            return fast ? launch_bilstm<4, true>(a, st) : launch_bilstm<4, false>(a, st);
        default: dprnn_set_error("dprnn_bilstm_fwd: seq_tile must be 0,16,-16,32,48,64 in gemm mode 0 (got %d)", seq_tile); return -1;
    }
}

extern "C" int dprnn_bilstm_fwd(const float* X, const float* packed, const float* bsum, float* H,
                                int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                int64_t t_stride, int ndir, int seq_tile, void* stream) {
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = H; a.nseq = nseq; a.T = T; a.inner_n = inner_n; a.ndir = ndir;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride;
    return bilstm_dispatch(a, g_gemm_mode, seq_tile, (hipStream_t)stream);
}

static RecurArgs rnn_args(int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                          const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                          const float* packed, const float* bsum, float* H,
                          int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride, int ndir) {
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = H; a.nseq = nseq; a.T = T; a.inner_n = inner_n; a.ndir = ndir;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride; a.cell = cell;
    a.Z = Z; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.seqs_per_utt = seqs_per_utt;
    a.rows_per_utt = rows_per_utt; a.Xout = Xout;
    return a;
}

extern "C" int dprnn_rnn_fwd(int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                             const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                             const float* packed, const float* bsum, float* H,
                             int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                             int64_t t_stride, int ndir, int seq_tile, void* stream) {
    return bilstm_dispatch(rnn_args(cell, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, nseq, T,
                                    inner_n, outer_stride, inner_stride, t_stride, ndir), g_gemm_mode, seq_tile, (hipStream_t)stream);
}
extern "C" int dprnn_rnn_fwd_ex(int gemm_mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                                const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                const float* packed, const float* bsum, float* H,
                                int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                int64_t t_stride, int ndir, int seq_tile, void* stream) {
    return bilstm_dispatch(rnn_args(cell, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, nseq, T,
                                    inner_n, outer_stride, inner_stride, t_stride, ndir), gemm_mode, seq_tile, (hipStream_t)stream);
}

// H as f16 piece quads (RecurArgs::h_pieces) for dprnn_proj_stats_hp_fwd: mode 3, LSTM, the eight-wave f16 tiles
extern "C" int dprnn_rnn_fwd_hp_ex(int gemm_mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                                   const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                   const float* packed, const float* bsum, float* H,
                                   int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                   int64_t t_stride, int ndir, int seq_tile, void* stream) {
    if (gemm_mode != 3 || cell != 0) {
        dprnn_set_error("dprnn_rnn_fwd_hp_ex: piece output exists for the LSTM in gate-GEMM mode 3 (got mode %d, cell %d)", gemm_mode, cell);
        return -1;
    }
    RecurArgs a = rnn_args(cell, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, nseq, T,
                           inner_n, outer_stride, inner_stride, t_stride, ndir);
    a.h_pieces = 1;
    return bilstm_dispatch(a, gemm_mode, seq_tile, (hipStream_t)stream);
}

extern "C" int dprnn_bilstm_norm_fwd(const float* X, const float* Z, const double* stats, const float* gamma,
                                     const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                     const float* packed, const float* bsum, float* H,
                                     int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                     int64_t t_stride, int ndir, int seq_tile, void* stream) {
    if (!Z) { dprnn_set_error("dprnn_bilstm_norm_fwd: Z is null (use dprnn_bilstm_fwd)"); return -1; }
    return bilstm_dispatch(rnn_args(0, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, nseq, T,
                                    inner_n, outer_stride, inner_stride, t_stride, ndir), g_gemm_mode, seq_tile, (hipStream_t)stream);
}

// ---- ragged batches: one recurrence launch over utterances of DIFFERENT lengths (split gate-GEMM modes 1 / 3, LSTM cell,
// 64-sequence eight-wave kernel).  The slab is the concatenation of the utterances' [S_u][K][F] slabs.
//   intra-chunk (wg_desc == NULL): n_items = total number of chunks, T = K; sequence q is rows q*K .. q*K + K-1; seq_utt[q]
//       names its utterance (needed only with a pending norm).
//   inter-chunk (wg_desc != NULL): n_items = workgroups per direction; workgroup w runs the sequences row0 + j (j < count)
//       for T steps of stride K rows: wg_desc[w] = {row0 lo, row0 hi, T = S_u, count <= 64, u, 0, 0, 0}; tiles never
//       straddle utterances (sort them by T, longest first: they end at different times).
// utt_rows[u] = S_u * K (the count of the pending norm; stats[u] as everywhere).
static int rnn_fwd_ragged(int mode, const float* X, const float* Z, const double* stats, const float* gamma, const float* beta,
                          float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc, int n_items,
                          float* Xout, const float* packed, const float* bsum, float* H, int T, int K, int ndir,
                          void* stream, int h_pieces = 0) {
    if (!X || !packed || !bsum || !H) { dprnn_set_error("dprnn_rnn_fwd_ragged: null pointer"); return -1; }
    if (mode != 1 && mode != 3) {
        dprnn_set_error("dprnn_rnn_fwd_ragged: needs a split gate-GEMM mode, 1 (bf16 pieces) or 3 (f16 pieces); got %d", mode);
        return -1;
    }
    if (n_items <= 0 || K <= 0 || (ndir != 1 && ndir != 2) || (!wg_desc && T <= 0)) {
        dprnn_set_error("dprnn_rnn_fwd_ragged: bad sizes n_items=%d T=%d K=%d ndir=%d", n_items, T, K, ndir);
        return -1;
    }
    if (Z && (!stats || !gamma || !beta || !Xout || Xout == X || !utt_rows || (!wg_desc && !seq_utt))) {
        dprnn_set_error("dprnn_rnn_fwd_ragged: the pending norm needs stats, gamma, beta, utt_rows, (seq_utt) and an Xout different from X");
        return -1;
    }
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = H; a.nseq = n_items; a.T = T; a.ndir = ndir; a.cell = 0;
    a.Z = Z; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.Xout = Xout;
    a.seqs_per_utt = 1; a.rows_per_utt = 1; a.utt_rows = reinterpret_cast<const long long*>(utt_rows);
    a.seq_utt = seq_utt; a.wg_desc = wg_desc; a.h_pieces = h_pieces;
    if (wg_desc) { a.inner_n = 1 << 30; a.outer_stride = 0; a.inner_stride = 1; a.t_stride = K; }
    else { a.inner_n = n_items; a.outer_stride = 0; a.inner_stride = K; a.t_stride = 1; }
    if ((long long)64 * K * ndir * HD * 4 >= 0x7fffffffLL) { dprnn_set_error("dprnn_rnn_fwd_ragged: chunk_length too large"); return -1; }
    if (int rc = dprnn_pack_expect(packed, DPRNN_PACK_FWD, 0, mode, ndir, "dprnn_rnn_fwd_ragged")) return rc;
    a.sat = dprnn_sat_ptr();
    { const ScaleRef sr = dprnn_scale_ref(); a.xs_src = sr.src; a.xmax_dst = sr.dst; }
    return launch_bilstm_split(a, mode, 2064, g_fast_act != 0, (hipStream_t)stream);
}
extern "C" int dprnn_rnn_fwd_ragged_hp_ex(int gemm_mode, const float* X, const float* Z, const double* stats, const float* gamma,
                                          const float* beta, float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc,
                                          int n_items, float* Xout, const float* packed, const float* bsum, float* H, int T, int K,
                                          int ndir, void* stream) {
    if (gemm_mode != 3) { dprnn_set_error("dprnn_rnn_fwd_ragged_hp_ex: piece output exists in gate-GEMM mode 3 (got %d)", gemm_mode); return -1; }
    return rnn_fwd_ragged(gemm_mode, X, Z, stats, gamma, beta, eps, utt_rows, seq_utt, wg_desc, n_items, Xout, packed, bsum, H, T, K, ndir, stream, 1);
}
extern "C" int dprnn_rnn_fwd_ragged(const float* X, const float* Z, const double* stats, const float* gamma, const float* beta,
                                    float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc, int n_items,
                                    float* Xout, const float* packed, const float* bsum, float* H, int T, int K, int ndir,
                                    void* stream) {
    return rnn_fwd_ragged(g_gemm_mode, X, Z, stats, gamma, beta, eps, utt_rows, seq_utt, wg_desc, n_items, Xout, packed, bsum, H, T, K, ndir, stream);
}
extern "C" int dprnn_rnn_fwd_ragged_ex(int gemm_mode, const float* X, const float* Z, const double* stats, const float* gamma,
                                       const float* beta, float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc,
                                       int n_items, float* Xout, const float* packed, const float* bsum, float* H, int T, int K,
                                       int ndir, void* stream) {
    return rnn_fwd_ragged(gemm_mode, X, Z, stats, gamma, beta, eps, utt_rows, seq_utt, wg_desc, n_items, Xout, packed, bsum, H, T, K, ndir, stream);
}

// ---- training forward: dprnn_rnn_fwd for the LSTM cell on 32-sequence tiles, plus the gate tape for dprnn_lstm_bwd
extern "C" int64_t dprnn_rnn_tape_floats(int cell, int nseq, int T, int ndir) {
    return (int64_t)ndir * ((nseq + 31) / 32) * T * (32 * HD * (cell == 2 ? 1 : 5));
}
extern "C" int64_t dprnn_lstm_tape_floats(int nseq, int T, int ndir) { return dprnn_rnn_tape_floats(0, nseq, T, ndir); }

static int rnn_fwd_train(int mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                                   const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                   const float* packed, const float* bsum, float* H, float* tape,
                                   int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                   int64_t t_stride, int ndir, void* stream) {
    if (!tape) { dprnn_set_error("dprnn_lstm_fwd_train: tape is null"); return -1; }
    const bool tape_packed = (mode & DPRNN_TAPE_PACKED) != 0;
    mode &= ~DPRNN_TAPE_PACKED;
    if (tape_packed && !(mode == 3 && cell == 0)) {
        dprnn_set_error("dprnn_rnn_fwd_train: the packed gate tape (mode | 0x100) exists for the f16-pieces LSTM forward (mode 3, cell 0) only");
        return -1;
    }
    if (cell < 0 || cell > 2) { dprnn_set_error("dprnn_rnn_fwd_train: cell must be 0 (LSTM), 1 (GRU) or 2 (RNN tanh)"); return -1; }
    if (mode != 0 && !(mode == 3 && cell == 0)) {
        dprnn_set_error("dprnn_rnn_fwd_train: the training forward runs in gate-GEMM mode 0 (any cell) or 3 (LSTM), not %d", mode);
        return -1;
    }
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = H; a.nseq = nseq; a.T = T; a.inner_n = inner_n; a.ndir = ndir;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride; a.cell = cell;
    a.Z = Z; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.seqs_per_utt = seqs_per_utt;
    a.rows_per_utt = rows_per_utt; a.Xout = Xout; a.tape = tape; a.tape_packed = tape_packed ? 1 : 0;
    if (!a.X || !a.Wp || !a.bsum || !a.H) { dprnn_set_error("dprnn_lstm_fwd_train: null pointer"); return -1; }
    if (a.nseq <= 0 || a.T <= 0 || a.inner_n <= 0 || (a.ndir != 1 && a.ndir != 2)) {
        dprnn_set_error("dprnn_lstm_fwd_train: bad sizes nseq=%d T=%d inner_n=%d ndir=%d", a.nseq, a.T, a.inner_n, a.ndir);
        return -1;
    }
    if (a.Z && (!a.stats || !a.gamma || !a.beta || !a.Xout || a.Xout == a.X || a.seqs_per_utt <= 0 || a.rows_per_utt <= 0)) {
        dprnn_set_error("dprnn_lstm_fwd_train: the pending norm needs stats, gamma, beta, sizes and an Xout different from X");
        return -1;
    }
    if (!tile_offsets_fit(a.inner_n, a.outer_stride, a.inner_stride, a.t_stride, a.ndir * HD) ||
        (long long)a.T * 4 * (2 * 2 * 5 * 64) * 16 >= 0x7fffffffLL) {
        dprnn_set_error("dprnn_rnn_fwd_train: a sequence tile spans more than 2 GiB (outer_stride=%lld rows, T=%d)", a.outer_stride, a.T);
        return -1;
    }
    if (int rc = dprnn_pack_expect(a.Wp, DPRNN_PACK_FWD, cell, mode, ndir, "dprnn_rnn_fwd_train")) return rc;
    a.sat = dprnn_sat_ptr();
    { const ScaleRef sr = dprnn_scale_ref(); a.xs_src = sr.src; a.xmax_dst = sr.dst; }
    hipStream_t st = (hipStream_t)stream;
    if (mode == 3) return launch_split8_f16_train(a, g_fast_act != 0, st);      // f16 pieces, same tape
    if (cell == 1) return g_fast_act ? launch_bilstm<2, true, 1, true>(a, st) : launch_bilstm<2, false, 1, true>(a, st);
    if (cell == 2) return g_fast_act ? launch_bilstm<2, true, 2, true>(a, st) : launch_bilstm<2, false, 2, true>(a, st);
    return g_fast_act ? launch_bilstm<2, true, 0, true>(a, st) : launch_bilstm<2, false, 0, true>(a, st);
}

extern "C" int dprnn_rnn_fwd_train(int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                                   const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                   const float* packed, const float* bsum, float* H, float* tape,
                                   int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                   int64_t t_stride, int ndir, void* stream) {
    return rnn_fwd_train(g_gemm_mode, cell, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, tape,
                         nseq, T, inner_n, outer_stride, inner_stride, t_stride, ndir, stream);
}
extern "C" int dprnn_rnn_fwd_train_ex(int gemm_mode, int cell, const float* X, const float* Z, const double* stats, const float* gamma,
                                      const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                      const float* packed, const float* bsum, float* H, float* tape,
                                      int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                      int64_t t_stride, int ndir, void* stream) {
    return rnn_fwd_train(gemm_mode, cell, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, tape,
                         nseq, T, inner_n, outer_stride, inner_stride, t_stride, ndir, stream);
}

extern "C" int dprnn_lstm_fwd_train(const float* X, const float* Z, const double* stats, const float* gamma,
                                    const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                    const float* packed, const float* bsum, float* H, float* tape,
                                    int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                    int64_t t_stride, int ndir, void* stream) {
    return dprnn_rnn_fwd_train(0, X, Z, stats, gamma, beta, eps, seqs_per_utt, rows_per_utt, Xout, packed, bsum, H, tape,
                               nseq, T, inner_n, outer_stride, inner_stride, t_stride, ndir, stream);
}

// Diagnostic (tools/bench_recur.py --stamps): same kernel with s_memtime stamps; stamps = [workgroups][8] u64 (exact kernel), [workgroups][2][8] (split kernels: waves 0 and 4)
// {cycles, 100 MHz ticks, k-loop cycles, cell-phase cycles}.  Not part of the product path.
extern "C" int dprnn_bilstm_fwd_stamped(const float* X, const float* packed, const float* bsum, float* H,
                                        int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                        int64_t t_stride, int ndir, int seq_tile, unsigned long long* stamps, void* stream) {
    if (!X || !packed || !bsum || !H || !stamps) { dprnn_set_error("dprnn_bilstm_fwd_stamped: null pointer"); return -1; }
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = H; a.nseq = nseq; a.T = T; a.inner_n = inner_n; a.ndir = ndir;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride; a.stamps = stamps;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = dprnn_pack_expect(packed, DPRNN_PACK_FWD, 0, g_gemm_mode, ndir, "dprnn_bilstm_fwd_stamped")) return rc;
    if (g_gemm_mode == 1 || g_gemm_mode == 3) {   // eight-wave split kernel with stamps (packed by dprnn_rnn_pack in the same mode)
        if (seq_tile != 2064) { dprnn_set_error("dprnn_bilstm_fwd_stamped: split mode stamps exist for seq_tile 2064"); return -1; }
        return launch_bilstm_split(a, g_gemm_mode, seq_tile, true, st);
    }
    auto go = [&](auto kern, int NS) {
        const size_t lds = (size_t)2 * NS * XS * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
        hipLaunchKernelGGL(kern, dim3((nseq + NS - 1) / NS, ndir), dim3(256), lds, st, a);
        return dprnn_check_launch("bilstm_kernel(stamped)");
    };
    const char* dg = getenv("DPRNN_DIAG");
    const int diag = dg ? atoi(dg) : 0;
    const char* sc = getenv("DPRNN_SCHED");
    if (sc && atoi(sc) == 1) {
        if (seq_tile == 32) return go(&bilstm_kernel<2, true, true, 0, 1>, 32);
        if (seq_tile == 64) return go(&bilstm_kernel<4, true, true, 0, 1>, 64);
    }
    if (seq_tile == 32) {
        switch (diag) {
            case 1: return go(&bilstm_kernel<2, true, true, 1>, 32);
            case 2: return go(&bilstm_kernel<2, true, true, 2>, 32);
            case 3: return go(&bilstm_kernel<2, true, true, 3>, 32);
            case 4: return go(&bilstm_kernel<2, true, true, 4>, 32);
            default: return go(&bilstm_kernel<2, true, true, 0>, 32);
        }
    }
    if (seq_tile == 64) {
        switch (diag) {
            case 1: return go(&bilstm_kernel<4, true, true, 1>, 64);
            case 2: return go(&bilstm_kernel<4, true, true, 2>, 64);
            case 3: return go(&bilstm_kernel<4, true, true, 3>, 64);
            case 4: return go(&bilstm_kernel<4, true, true, 4>, 64);
            default: return go(&bilstm_kernel<4, true, true, 0>, 64);
        }
    }
    dprnn_set_error("dprnn_bilstm_fwd_stamped: seq_tile 32 or 64");
    return -1;
}
