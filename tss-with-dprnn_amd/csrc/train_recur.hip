// Backward of the fused BiLSTM recurrence (training path).
//
// Gradient of SingleRNN.forward = nn.LSTM(128,128,bidirectional) (reference src/models/dprnn.py:23-37) as autograd
// computes it for `epoch_loss.backward()` (src/trainers/trainer.py:106).  The forward (bilstm_kernel<.., SAVE>,
// recur.hip) left a tape of the activated gates i,f,g,o and the cell state c of every step.  One workgroup (4 waves)
// owns the same 32 sequences of one direction as in the forward and walks the steps in reverse order:
//
//   da_t = gate derivatives of (dH_t + dh_rec, dc_rec) from the tape                 lane-local, same lane mapping
//   [dx_t | dh_rec] (32 x 256) = da_t (32 x 512) @ [W_ih | W_hh] (512 x 256)         v_mfma_f32_16x16x4_f32
//
// i.e. the same FLOPs and the same structure as the forward step with K and N exchanged: da_t goes through LDS as
// the A operand ([seq][512 + 1 pad], conflict-free fragment reads with the forward's k-permutation), the transposed
// weights are streamed from L2 as pre-packed B fragments (dprnn_lstm_pack_bwd), wave w owns input columns and hidden
// units [32w, 32w+32) so dh_rec and dc_rec never leave registers.  da_t is also written to HBM ([row][dir][4*128]):
// the weight gradients dW_ih = da^T x, dW_hh = da^T h_{t-1}, db = sum da are one large GEMM over all rows
// (dprnn_wgrad, train_gemm.hip) outside the serial loop.  dx of the two directions goes to two slabs that the
// consumer adds (no atomics: the result does not depend on scheduling).
//
// The same kernel serves the other cells torch.nn offers behind rnn_type (dprnn.py:23), selected by CELL:
//   GRU  (1): tape r, z, n, nh = W_hn h + b_hn, h;  da slots (r, z, n_x, n_h) -- K = 512 like the LSTM, the transposed
//             weight pack carries zero blocks where n_x has no h part and n_h no x part; dh_{t-1} also gets dh_t * z directly
//   RNN  (2): tape h;  da = dh_t * (1 - h^2), K = 128.
#include "recur.h"
#include <stdlib.h>
using namespace recur;

namespace {

typedef const f32x4 __attribute__((address_space(1)))* gw_ptr;

constexpr int kdim_of(int cell) { return cell == 2 ? 128 : 512; }          // rows of the da image: gate slots * 128
constexpr int PACKT_PER_DIR = 4 * 32 * 4 * 256;   // floats: [wave][grp 32][col tile 4][lane][4] (RNN uses the first 8 groups)

// packedT[dir][wave][grp][j][lane][i] = Wcat[k][col],  k = 64*(grp>>2) + 16*q + 4*(grp&3) + i (gate row, 0..511),
// col: j < 2 -> input column 32*wave + 16*j + r (W_ih), j >= 2 -> hidden column 32*wave + 16*(j-2) + r (W_hh)
__global__ void lstm_pack_bwd_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                     const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                     float* __restrict__ packedT, int ndir, int cell) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ndir * PACKT_PER_DIR) return;
    const int i = idx & 3, lane = (idx >> 2) & 63, j = (idx >> 8) & 3, grp = (idx >> 10) & 31;
    const int wave = (idx >> 15) & 3, dir = idx >> 17;
    const int r = lane & 15, q = lane >> 4;
    const int k = 64 * (grp >> 2) + 16 * q + 4 * (grp & 3) + i;
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    const int xc = 32 * wave + 16 * j + r, hc = 32 * wave + 16 * (j - 2) + r;
    float v = 0.f;
    if (cell == 0) {
        v = j < 2 ? wi[k * F + xc] : wh[k * HD + hc];
    } else if (cell == 1) {            // da slots r, z, n_x, n_h; torch rows r, z, n
        const int slot = k >> 7, unit = k & 127;
        if (j < 2) { if (slot < 3) v = wi[(slot * HD + unit) * F + xc]; }
        else { if (slot < 2) v = wh[(slot * HD + unit) * HD + hc]; else if (slot == 3) v = wh[(2 * HD + unit) * HD + hc]; }
    } else if (k < HD) {
        v = j < 2 ? wi[k * F + xc] : wh[k * HD + hc];
    }
    packedT[idx] = v;
}

struct BwdArgs {
    const float* dH;          // [rows][ndir*128] gradient of the recurrence output
    const float* tape;        // forward tape
    const float* WpT;         // dprnn_lstm_pack_bwd
    float* dX;                // [ndir][rows_total][128]
    float* dG;                // [rows][ndir*512] gate pre-activation gradients (i,f,g,o)
    long long rows_total;
    int nseq, T, inner_n, ndir;
    long long outer_stride, inner_stride, t_stride;
    unsigned long long* stamps;   // diagnostic build only: [workgroup][8] cycle counters of wave 0
};

// NTS: the once-touched streams (tape and dH in, dG and dX out) carry the non-temporal cache policy, so that they do not push
// the 1 MB of transposed weight fragments -- re-read by every workgroup every step -- out of the XCD's L2 (PMC, round 1:
// 6.6 GB read per launch against 4.8 GB algorithmic, the difference being weights re-fetched from HBM).
template <int NT, bool FAST, int CELL = 0, bool STAMP = false, bool NTS = true>
__global__ __launch_bounds__(256, 2)
void lstm_bwd_kernel(BwdArgs a) {
    constexpr int SAUX = NTS ? 2 : 0;
    constexpr int NS = 16 * NT;
    constexpr int KG = kdim_of(CELL), GS = KG + 1, NGB = KG / 16;     // da image: KG columns + 1 pad (odd stride), k-groups of 16
    constexpr int NTAPE = CELL == 2 ? 1 : 5;
    extern __shared__ __attribute__((aligned(16))) float DA[];        // [NS][GS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NS;
    const int T = a.T;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const long long row0 = rowbase(seq0 < a.nseq ? seq0 : 0);
    const int hpitch = a.ndir * HD, gpitch = a.ndir * KG;
    // rows of the accumulator tile this lane owns: seq 16*nt + 4*q + e.
    // Every global access goes through a buffer resource whose base is the step's row (SGPRs) with a 32-bit per-lane
    // byte offset fixed for the whole kernel (+ immediate): no vector ALU work for addressing inside the step loop --
    // fp32 VALU instructions cost matrix time.  Rows past nseq carry an offset beyond the resource's range: the
    // hardware returns zeros for their loads and drops their stores, so the loop has no exec-mask branches either.
    constexpr unsigned OOB = 0x80000000u;
    unsigned hoff[NT][4], goff[NT][4], xoff[NT][4];      // byte offsets into dH / dG / dX rows of this lane's sequences
    const int ucol = 32 * wave + r;        // + 16*u
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int seq = seq0 + 16 * nt + 4 * q + e;
            const bool val = seq < a.nseq;
            const unsigned ro = (unsigned)(rowbase(val ? seq : seq0) - row0);
            hoff[nt][e] = val ? (ro * hpitch + dir * HD + ucol) * 4u : OOB;
            goff[nt][e] = val ? (ro * gpitch + dir * KG + ucol) * 4u : OOB;
            xoff[nt][e] = val ? (ro * F + ucol) * 4u : OOB;
        }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto rsrc_at = [&](const void* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
    };
    auto bload4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) __attribute__((always_inline)) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    const unsigned lane16 = lane * 16u;
    // weights: fragment (k-group g, column tile j) of this wave at scalar offset (4 g + j) KiB
    const __amdgpu_buffer_rsrc_t wrs = rsrc_at(reinterpret_cast<const f32x4*>(a.WpT) + (size_t)(dir * 4 + wave_u) * 32 * 4 * 64);
    auto wfrag = [&](int grp, int j) __attribute__((always_inline)) { return bload4(wrs, lane16, (unsigned)((grp * 4 + j) * 1024)); };
    // tape: this workgroup's steps are contiguous; entry x of (step fs, wave) at scalar offset
    constexpr unsigned TSTEP = 4u * (NT * 2 * NTAPE * 64) * 16u;      // bytes per step (4 waves)
    const __amdgpu_buffer_rsrc_t trs = rsrc_at(reinterpret_cast<const f32x4*>(a.tape) +
                                               ((size_t)dir * gridDim.x + blockIdx.x) * T * 4 * (NT * 2 * NTAPE * 64));
    const unsigned tlane = (unsigned)(wave * (NT * 2 * NTAPE * 64) + lane) * 16u;
    auto tape_ld = [&](int fs, int x) __attribute__((always_inline)) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, tlane, (unsigned)fs * TSTEP + (unsigned)x * 1024u, SAUX));
    };
    auto t_of = [&](int fs) { return (long long)(dir ? (T - 1 - fs) : fs); };      // forward step -> time index

    f32x4 gi[NT][2], gf[NT][2], gg[NT][2], go[NT][2], cc[NT][2], cp[NT][2], dho[NT][2];
    f32x4 dh[NT][2], dc[NT][2];
    auto load_step = [&](int fs) __attribute__((always_inline)) {     // gates of step fs, c of step fs-1, dH of step fs
        const int fq = fs > 0 ? fs - 1 : 0;
        const __amdgpu_buffer_rsrc_t dhs = rsrc_at(a.dH + (row0 + t_of(fs) * a.t_stride) * hpitch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int b = (nt * 2 + u) * NTAPE;
                // LSTM: i, f, g, o and c_{t-1};  GRU: r, z, n, nh and h_{t-1};  RNN: h_t
                gi[nt][u] = tape_ld(fs, b + 0);
                if constexpr (CELL != 2) {
                    gf[nt][u] = tape_ld(fs, b + 1);
                    gg[nt][u] = tape_ld(fs, b + 2); go[nt][u] = tape_ld(fs, b + 3);
                    const f32x4 c1 = tape_ld(fq, b + 4);
                    cp[nt][u] = fs > 0 ? c1 : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dho[nt][u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dhs, hoff[nt][e] + 64 * u, 0, SAUX));
            }
    };
    {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                cc[nt][u] = CELL == 0 ? tape_ld(T - 1, (nt * 2 + u) * NTAPE + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                dh[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                dc[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    load_step(T - 1);
    f32x4 w[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[0][j] = wfrag(0, j);

    unsigned long long sp[6] = {0, 0, 0, 0, 0, 0}, tk = 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            if (slot >= 0) sp[slot] += now - tk;
            tk = now;
        }
    };
    const unsigned long long st_begin = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long st_real = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;       // 100 MHz
    for (int fs = T - 1; fs >= 0; --fs) {
        stamp(-1);
        // share the matrix pipe evenly between the two co-resident workgroups (see bilstm_kernel)
        if (((fs >> 3) + blockIdx.y) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        const long long trow = row0 + t_of(fs) * a.t_stride;
        // keep "offset + constant" inside the loop body so the constant lands in the instruction's immediate field
        // instead of a second, hoisted copy of every offset register
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(hoff[nt][e]), "+v"(goff[nt][e]), "+v"(xoff[nt][e]));
        // ---- gate derivatives (torch.nn.LSTM backward), lane-local; da -> LDS image and HBM
        const __amdgpu_buffer_rsrc_t dgs = rsrc_at(a.dG + trow * gpitch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // whole-vector arithmetic over the lane's four sequences: packed fp32 instructions (two results per issue slot)
                const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                const f32x4 dht = dho[nt][u] + dh[nt][u];
                f32x4 da0, da1, da2, da3;                  // gradient of the pre-activations, in the image's slot order
                if constexpr (CELL == 0) {
                    const f32x4 i_ = gi[nt][u], f_ = gf[nt][u], g_ = gg[nt][u], o_ = go[nt][u];
                    const f32x2 t01 = tanh_f2<FAST>(f32x2{cc[nt][u][0], cc[nt][u][1]}), t23 = tanh_f2<FAST>(f32x2{cc[nt][u][2], cc[nt][u][3]});
                    const f32x4 tc = {t01[0], t01[1], t23[0], t23[1]};
                    const f32x4 dct = dc[nt][u] + dht * o_ * (one - tc * tc);
                    da3 = dht * tc * o_ * (one - o_);
                    da0 = dct * g_ * i_ * (one - i_);
                    da1 = dct * cp[nt][u] * f_ * (one - f_);
                    da2 = dct * i_ * (one - g_ * g_);
                    dc[nt][u] = dct * f_;
                } else if constexpr (CELL == 1) {     // h' = (1-z) n + z h,  n = tanh(n_x + r*n_h); slots r, z, n_x, n_h
                    const f32x4 r_ = gi[nt][u], z_ = gf[nt][u], n_ = gg[nt][u], nh = go[nt][u];
                    da2 = dht * (one - z_) * (one - n_ * n_);
                    da1 = dht * (cp[nt][u] - n_) * z_ * (one - z_);
                    da0 = da2 * nh * r_ * (one - r_);
                    da3 = da2 * r_;
                    dc[nt][u] = dht * z_;                 // the direct dh_{t-1} term, added after the GEMM
                } else {                              // h = tanh(a)
                    da0 = dht * (one - gi[nt][u] * gi[nt][u]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* d = DA + (16 * nt + 4 * q + e) * GS + ucol + 16 * u;
                    const unsigned go_ = goff[nt][e] + 64 * u;
                    const float v0 = da0[e];
                    d[0] = v0;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), dgs, go_, 0, SAUX);
                    if constexpr (CELL != 2) {
                        const float v1 = da1[e], v2 = da2[e], v3 = da3[e];
                        d[HD] = v1; d[2 * HD] = v2; d[3 * HD] = v3;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), dgs, go_ + 4 * HD, 0, SAUX);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v2), dgs, go_ + 8 * HD, 0, SAUX);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v3), dgs, go_ + 12 * HD, 0, SAUX);
                    }
                }
            }
        if constexpr (CELL == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int u = 0; u < 2; ++u) cc[nt][u] = cp[nt][u];
        }
        if (fs > 0) load_step(fs - 1);          // in flight during the GEMM below
        stamp(0);                               // gate derivatives, image and dG stores, prefetch issue
        __syncthreads();
        stamp(1);                               // wait at the first barrier

        // ---- [dx | dh] = da @ [W_ih | W_hh]: 32 k-groups, weights of group g+1 in flight while group g computes
        f32x4 acc[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[nt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* const Ac = DA + r * GS + 16 * q;
#pragma unroll 1
        for (int gp = 0; gp < NGB / 2; ++gp) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int grp = 2 * gp + half;
                const int gn = (grp + 1) & (NGB - 1);                            // wraps to group 0 of the next step
#pragma unroll
                for (int j = 0; j < 4; ++j) w[half ^ 1][j] = wfrag(gn, j);
                const float* src = Ac + 64 * (grp >> 2) + 4 * (grp & 3);
                float av[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[nt][i] = src[nt * 16 * GS + i];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[nt][j] = mfma16(av[nt][i], w[half][j][i], acc[nt][j]);
            }
        }
        stamp(2);                               // GEMM
        __syncthreads();                        // every wave is done reading the image
        stamp(3);                               // wait at the second barrier

        const __amdgpu_buffer_rsrc_t dxs = rsrc_at(a.dX + ((long long)dir * a.rows_total + trow) * F);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dh[nt][u] = acc[nt][2 + u];
                if constexpr (CELL == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[nt][u][e] += dc[nt][u][e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[nt][u][e];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dxs, xoff[nt][e] + 64 * u, 0, SAUX);
                }
            }
        }
        stamp(4);                               // dh hand-over and dX stores
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = __builtin_amdgcn_s_memtime() - st_begin;
            for (int i = 0; i < 5; ++i) o[1 + i] = sp[i];
            o[6] = __builtin_amdgcn_s_memrealtime() - st_real;
            o[7] = st_real;                                        // absolute start (100 MHz)
        }
    }
}

}  // namespace

// Arithmetic of the step GEMM of the LSTM backward: 0 = exact fp32 MFMA (lstm_bwd_kernel), 3 = two f16 pieces / three products
// (train_recur_f16.hip).  The *_ex entry points take it as an argument, the plain ones read this per-thread default; either way
// dprnn_rnn_pack_bwd records the layout it wrote with the buffer and dprnn_rnn_bwd refuses a buffer of the other layout.
// GRU / Elman cells always use the fp32 kernel.
static thread_local int g_bwd_gemm_mode = 0;
extern "C" void dprnn_set_bwd_gemm_mode(int mode) { g_bwd_gemm_mode = mode == 3 ? 3 : 0; }
extern "C" int dprnn_get_bwd_gemm_mode(void) { return g_bwd_gemm_mode; }
int recur_bwd_f16_pack(const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r, float* packedT, hipStream_t st);
int recur_bwd_f16_launch(const float* dH, const float* tape, const float* packedT, float* dX, float* dG, long long rows_total,
                         int nseq, int T, int inner_n, long long outer_stride, long long inner_stride, long long t_stride, int ndir,
                         bool fast, const float* dhmax, bool tape_packed, hipStream_t st);
// One-shot selector: the NEXT dprnn_rnn_bwd launch in mode 3 takes max|dH| from *slot (device float, e.g. written by the dprnn_dgrad
// call that produced dH: dprnn_dgrad_store_absmax) instead of measuring it with a pass over dH.  Ignored by the fp32 kernel.
// (dprnn_rnn_bwd_ex takes it as an argument.)
static thread_local const float* g_bwd_dhmax = nullptr;
extern "C" void dprnn_rnn_bwd_use_gradmax(const float* slot) { g_bwd_dhmax = slot; }

static int rnn_pack_bwd(int mode, int cell, const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r,
                        float* packedT, void* stream) {
    if (!w_ih || !w_hh || !packedT) { dprnn_set_error("dprnn_rnn_pack_bwd: null pointer"); return -1; }
    if (cell < 0 || cell > 2) { dprnn_set_error("dprnn_rnn_pack_bwd: cell must be 0 (LSTM), 1 (GRU) or 2 (RNN tanh)"); return -1; }
    if (mode != 0 && mode != 3) { dprnn_set_error("dprnn_rnn_pack_bwd: backward gemm mode must be 0 or 3 (got %d)", mode); return -1; }
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && !w_hh_r) { dprnn_set_error("dprnn_rnn_pack_bwd: incomplete reverse direction"); return -1; }
    if (cell == 0 && mode == 3) {
        dprnn_pack_note(packedT, DPRNN_PACK_BWD, 0, 3, ndir);
        return recur_bwd_f16_pack(w_ih, w_hh, w_ih_r, w_hh_r, packedT, (hipStream_t)stream);
    }
    const int n = ndir * PACKT_PER_DIR;
    hipLaunchKernelGGL(lstm_pack_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, w_ih_r, w_hh_r, packedT, ndir, cell);
    dprnn_pack_note(packedT, DPRNN_PACK_BWD, cell, 0, ndir);
    return dprnn_check_launch("lstm_pack_bwd_kernel");
}
extern "C" int dprnn_rnn_pack_bwd(int cell, const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r,
                                  float* packedT, void* stream) {
    return rnn_pack_bwd(g_bwd_gemm_mode, cell, w_ih, w_hh, w_ih_r, w_hh_r, packedT, stream);
}
extern "C" int dprnn_rnn_pack_bwd_ex(int bwd_gemm_mode, int cell, const float* w_ih, const float* w_hh, const float* w_ih_r,
                                     const float* w_hh_r, float* packedT, void* stream) {
    return rnn_pack_bwd(bwd_gemm_mode, cell, w_ih, w_hh, w_ih_r, w_hh_r, packedT, stream);
}

extern "C" int dprnn_lstm_pack_bwd(const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r,
                                   float* packedT, void* stream) {
    return dprnn_rnn_pack_bwd(0, w_ih, w_hh, w_ih_r, w_hh_r, packedT, stream);
}

extern int dprnn_fast_activations_enabled(void);

namespace {
template <int CELL>
int launch_bwd(const BwdArgs& a, hipStream_t st) {
    constexpr int NT = 2;
    const size_t lds = (size_t)16 * NT * (kdim_of(CELL) + 1) * sizeof(float);
    // the LSTM's tanh(c_t) is recomputed from the tape with the same form (v_exp_f32 / libm) the forward used
    const bool fast = dprnn_fast_activations_enabled() != 0;
    static const bool nts = !(getenv("DPRNN_BWD_NT") && atoi(getenv("DPRNN_BWD_NT")) == 0);      // A/B knob: DPRNN_BWD_NT=0 = default policy
    auto kern = fast ? (nts ? &lstm_bwd_kernel<NT, true, CELL> : &lstm_bwd_kernel<NT, true, CELL, false, false>)
                     : (nts ? &lstm_bwd_kernel<NT, false, CELL> : &lstm_bwd_kernel<NT, false, CELL, false, false>);
    static std::atomic<unsigned long long> attr_mask[2];
    int attr_dev;
    if (dprnn_attr_needed(attr_mask[fast], &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(rnn_bwd): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask[fast], attr_dev);
    }
    hipLaunchKernelGGL(kern, dim3((a.nseq + 16 * NT - 1) / (16 * NT), a.ndir), dim3(256), lds, st, a);
    return dprnn_check_launch("lstm_bwd_kernel");
}
}  // namespace

// diagnostic: when set, dprnn_rnn_bwd (LSTM) runs the stamped twin and writes [workgroup][8] u64: total cycles of the
// step loop, then the cycles wave 0 spent in gates / barrier 1 / GEMM / barrier 2 / tail (tools/bench_recur.py --train 2)
static thread_local unsigned long long* g_bwd_stamps = nullptr;
extern "C" void dprnn_debug_bwd_stamps(unsigned long long* buf) { g_bwd_stamps = buf; }

static int rnn_bwd(int mode, const float* dhmax, int cell, const float* dH, const float* tape, const float* packedT, float* dX, float* dG,
                   int64_t rows_total, int nseq, int T, int inner_n, int64_t outer_stride,
                   int64_t inner_stride, int64_t t_stride, int ndir, void* stream) {
    if (cell < 0 || cell > 2) { dprnn_set_error("dprnn_rnn_bwd: cell must be 0 (LSTM), 1 (GRU) or 2 (RNN tanh)"); return -1; }
    const bool tape_packed = (mode & DPRNN_TAPE_PACKED) != 0;           // the tape dprnn_rnn_fwd_train_ex(3 | 0x100, ...) wrote
    mode &= ~DPRNN_TAPE_PACKED;
    if (mode != 0 && mode != 3) { dprnn_set_error("dprnn_rnn_bwd: backward gemm mode must be 0 or 3 (got %d)", mode); return -1; }
    if (tape_packed && !(cell == 0 && mode == 3)) {
        dprnn_set_error("dprnn_rnn_bwd: a packed gate tape (mode | 0x100) is read by the f16-pieces LSTM backward (mode 3, cell 0) only");
        return -1;
    }
    if (cell != 0) mode = 0;
    if (!dH || !tape || !packedT || !dX || !dG) { dprnn_set_error("dprnn_rnn_bwd: null pointer"); return -1; }
    if (nseq <= 0 || T <= 0 || inner_n <= 0 || (ndir != 1 && ndir != 2) || rows_total < (int64_t)nseq * T) {
        dprnn_set_error("dprnn_rnn_bwd: bad sizes nseq=%d T=%d inner_n=%d ndir=%d rows_total=%lld", nseq, T, inner_n, ndir,
                        (long long)rows_total);
        return -1;
    }
    if (!recur::tile_offsets_fit(inner_n, outer_stride, inner_stride, t_stride, (long long)ndir * 512) ||
        (long long)T * 4 * (2 * 2 * 5 * 64) * 16 >= 0x7fffffffLL) {
        dprnn_set_error("dprnn_rnn_bwd: a sequence tile spans more than 2 GiB of dG (outer_stride=%lld rows, T=%d): split the batch",
                        (long long)outer_stride, T);
        return -1;
    }
    if (int rc = dprnn_pack_expect(packedT, DPRNN_PACK_BWD, cell, mode, ndir, "dprnn_rnn_bwd")) return rc;
    if (cell == 0 && mode == 3 && (!g_bwd_stamps || tape_packed))
        return recur_bwd_f16_launch(dH, tape, packedT, dX, dG, (long long)rows_total, nseq, T, inner_n, (long long)outer_stride,
                                    (long long)inner_stride, (long long)t_stride, ndir, dprnn_fast_activations_enabled() != 0,
                                    dhmax, tape_packed, (hipStream_t)stream);
    BwdArgs a{dH, tape, packedT, dX, dG, (long long)rows_total, nseq, T, inner_n, ndir,
              (long long)outer_stride, (long long)inner_stride, (long long)t_stride, g_bwd_stamps};
    if (g_bwd_stamps && cell == 0) {            // diagnostic twin: same kernel with cycle stamps of wave 0 per workgroup
        const size_t lds = (size_t)32 * (kdim_of(0) + 1) * sizeof(float);
        auto kern = &lstm_bwd_kernel<2, true, 0, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((a.nseq + 31) / 32, a.ndir), dim3(256), lds, (hipStream_t)stream, a);
        return dprnn_check_launch("lstm_bwd_kernel(stamped)");
    }
    switch (cell) {
        case 0: return launch_bwd<0>(a, (hipStream_t)stream);
        case 1: return launch_bwd<1>(a, (hipStream_t)stream);
        case 2: return launch_bwd<2>(a, (hipStream_t)stream);
        default: dprnn_set_error("dprnn_rnn_bwd: cell must be 0 (LSTM), 1 (GRU) or 2 (RNN tanh)"); return -1;
    }
}

extern "C" int dprnn_rnn_bwd(int cell, const float* dH, const float* tape, const float* packedT, float* dX, float* dG,
                             int64_t rows_total, int nseq, int T, int inner_n, int64_t outer_stride,
                             int64_t inner_stride, int64_t t_stride, int ndir, void* stream) {
    const float* const dhmax = g_bwd_dhmax;              // one-shot: consumed by this call whatever it returns
    g_bwd_dhmax = nullptr;
    return rnn_bwd(g_bwd_gemm_mode, dhmax, cell, dH, tape, packedT, dX, dG, rows_total, nseq, T, inner_n, outer_stride, inner_stride, t_stride, ndir, stream);
}
// the same with the mode and max|dH| (device float written by dH's producer; NULL = measure it with a pass over dH) as arguments
extern "C" int dprnn_rnn_bwd_ex(int bwd_gemm_mode, const float* dh_absmax, int cell, const float* dH, const float* tape,
                                const float* packedT, float* dX, float* dG, int64_t rows_total, int nseq, int T, int inner_n,
                                int64_t outer_stride, int64_t inner_stride, int64_t t_stride, int ndir, void* stream) {
    return rnn_bwd(bwd_gemm_mode, dh_absmax, cell, dH, tape, packedT, dX, dG, rows_total, nseq, T, inner_n, outer_stride, inner_stride, t_stride, ndir, stream);
}

extern "C" int dprnn_lstm_bwd(const float* dH, const float* tape, const float* packedT, float* dX, float* dG,
                              int64_t rows_total, int nseq, int T, int inner_n, int64_t outer_stride,
                              int64_t inner_stride, int64_t t_stride, int ndir, void* stream) {
    return dprnn_rnn_bwd(0, dH, tape, packedT, dX, dG, rows_total, nseq, T, inner_n, outer_stride, inner_stride, t_stride, ndir, stream);
}
