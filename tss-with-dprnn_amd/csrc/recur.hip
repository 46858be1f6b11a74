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
#include "common.h"

namespace {

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
    unsigned long long* stamps;                       // diagnostic build only: [wg][4] cycle counters
};

// packed[dir][wave][grp][gate*2+u][lane][i] = Wcat[dir][gate*128 + 32*wave + 16*u + (lane&15)][k]
__global__ void lstm_pack_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                 const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                 const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                 const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r,
                                 float* __restrict__ packed, float* __restrict__ bsum, int ndir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ndir * PACK_PER_DIR) return;
    const int i = idx & 3, lane = (idx >> 2) & 63, gu = (idx >> 8) & 7, grp = (idx >> 11) & 15;
    const int wave = (idx >> 15) & 3, dir = idx >> 17;
    const int gate = gu >> 1, u = gu & 1, j = lane & 15, q = lane >> 4;
    const int row = gate * HD + 32 * wave + 16 * u + j;
    const int g = grp & 7;
    const int k = 64 * (g >> 2) + 16 * q + 4 * (g & 3) + i;
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    packed[idx] = (grp < 8) ? wi[row * F + k] : wh[row * HD + k];
    if (idx < ndir * 4 * HD) {
        const int d = idx / (4 * HD), n = idx % (4 * HD);
        bsum[idx] = (d ? b_ih_r : b_ih)[n] + (d ? b_hh_r : b_hh)[n];
    }
}

template <int NT, bool FAST, bool STAMP = false>
__global__ __launch_bounds__(256, (NT <= 2 ? 2 : 1))
void bilstm_kernel(RecurArgs a) {
    constexpr int NS = 16 * NT;
    constexpr int NXV = NS / 8;                       // float4 x loads per thread per step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const XH = smem;                           // [2][NS][XS]: row = x_t (128) | h_{t-1} (128)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NS;
    const int T = a.T;

    // ---- x staging descriptors: float4 f = tid + 256*m covers row f>>5, floats 4*(f&31)..+3
    const float* xptr[NXV];
    int xlds[NXV];
    bool xval[NXV];
#pragma unroll
    for (int m = 0; m < NXV; ++m) {
        const int f = tid + 256 * m, row = f >> 5, c4 = f & 31;
        const int seq = seq0 + row;
        xval[m] = seq < a.nseq;
        const int sq = xval[m] ? seq : 0;
        const long long base = (long long)(sq / a.inner_n) * a.outer_stride + (long long)(sq % a.inner_n) * a.inner_stride;
        xptr[m] = a.X + base * F + c4 * 4;
        xlds[m] = row * XS + c4 * 4;
    }
    // ---- rows of the accumulator tile this lane owns: seq row 16*nt + 4*q + reg
    long long hbase[NT][4];
    bool hval[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int seq = seq0 + 16 * nt + 4 * q + e;
            hval[nt][e] = seq < a.nseq;
            const int sq = hval[nt][e] ? seq : 0;
            hbase[nt][e] = (long long)(sq / a.inner_n) * a.outer_stride + (long long)(sq % a.inner_n) * a.inner_stride;
        }
    const int hpitch = a.ndir * HD;
    float* const hout = a.H + dir * HD + 32 * wave + r;

    float bias[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) bias[g][u] = a.bsum[dir * 4 * HD + g * HD + 32 * wave + 16 * u + r];

    const f32x4* const wbase = reinterpret_cast<const f32x4*>(a.Wp) + (size_t)(dir * 4 + wave) * NGRP * 8 * 64 + lane;

    // ---- prologue: x_{t0} -> Xs[0], zeros -> Hs[0]
    {
        const long long t0 = dir ? (T - 1) : 0;
#pragma unroll
        for (int m = 0; m < NXV; ++m) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xval[m]) v = *reinterpret_cast<const f32x4*>(xptr[m] + t0 * a.t_stride * F);
            float* d = XH + xlds[m];
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        for (int i = tid; i < NS * HD; i += 256) XH[(i >> 7) * XS + HD + (i & 127)] = 0.f;
    }
    f32x4 c[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int u = 0; u < 2; ++u) c[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 w[2][8];
#pragma unroll
    for (int x = 0; x < 8; ++x) w[0][x] = wbase[x * 64];
    __syncthreads();

    unsigned long long st_t0 = 0, st_r0 = 0, st_mfma = 0, st_act = 0;
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int step = 0; step < T; ++step) {
        unsigned long long st_a = 0, st_b = 0;
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        const int cur = step & 1, nxt = cur ^ 1;
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        // prefetch x_{t+1} into registers
        f32x4 xr[NXV];
        if (more) {
            const long long tn = dir ? (t - 1) : (t + 1);
#pragma unroll
            for (int m = 0; m < NXV; ++m) {
                xr[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (xval[m]) xr[m] = *reinterpret_cast<const f32x4*>(xptr[m] + tn * a.t_stride * F);
            }
        }
        f32x4 acc[NT][4][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[nt][g][u] = f32x4{bias[g][u], bias[g][u], bias[g][u], bias[g][u]};

        const float* const Ac = XH + cur * NS * XS + r * XS + 16 * q;
        // 16 k-groups, two per trip; the weights of group g+1 are in flight while group g computes
#pragma unroll 1
        for (int gp = 0; gp < NGRP / 2; ++gp) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int grp = 2 * gp + half;
                // weights of the next k-group (wraps to group 0 of the next step: same bytes every step)
                const f32x4* wn = wbase + (((grp + 1) & 15) * 8) * 64;
#pragma unroll
                for (int x = 0; x < 8; ++x) w[half ^ 1][x] = wn[x * 64];
                const float* src = Ac + 64 * (grp >> 2) + 4 * (grp & 3);
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
                        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                acc[nt][gt][u] = mfma16(av[nt][i], w[half][gt * 2 + u][i], acc[nt][gt][u]);
            }
        }

        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); st_mfma += st_b - st_a; }
        // ---- LSTM cell (gate order i,f,g,o), lane-local
        float* const Hn = XH + nxt * NS * XS + HD;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ig = sigmoid_f<FAST>(acc[nt][0][u][e]);
                    const float fg = sigmoid_f<FAST>(acc[nt][1][u][e]);
                    const float gg = tanh_f<FAST>(acc[nt][2][u][e]);
                    const float og = sigmoid_f<FAST>(acc[nt][3][u][e]);
                    const float cn = fg * c[nt][u][e] + ig * gg;
                    c[nt][u][e] = cn;
                    const float h = og * tanh_f<FAST>(cn);
                    Hn[(16 * nt + 4 * q + e) * XS + 32 * wave + 16 * u + r] = h;
                    if (hval[nt][e]) hout[(hbase[nt][e] + t * a.t_stride) * hpitch + 16 * u] = h;
                }
        if (more) {
            float* const Xn = XH + nxt * NS * XS;
#pragma unroll
            for (int m = 0; m < NXV; ++m) {
                float* d = Xn + xlds[m];
                d[0] = xr[m][0]; d[1] = xr[m][1]; d[2] = xr[m][2]; d[3] = xr[m][3];
            }
        }
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_act += __builtin_amdgcn_s_memtime() - st_b; __builtin_amdgcn_sched_barrier(0); }
        __syncthreads();
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + 4 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;          // shader cycles, whole step loop
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;      // 100 MHz ticks
            o[2] = st_mfma;                                       // cycles inside the k-loop (wave 0)
            o[3] = st_act;                                        // cycles in cell update + stores, before the barrier
        }
    }
}

template <int NT, bool FAST>
int launch_bilstm(const RecurArgs& a, hipStream_t st) {
    constexpr int NS = 16 * NT;
    const size_t lds = (size_t)2 * NS * XS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_kernel<NT, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    dim3 grid((a.nseq + NS - 1) / NS, a.ndir);
    hipLaunchKernelGGL((bilstm_kernel<NT, FAST>), grid, dim3(256), lds, st, a);
    return dprnn_check_launch("bilstm_kernel");
}

}  // namespace

extern "C" int dprnn_lstm_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                               const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                               float* packed, float* bsum, void* stream) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !bsum) { dprnn_set_error("dprnn_lstm_pack: null pointer"); return -1; }
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && (!w_hh_r || !b_ih_r || !b_hh_r)) { dprnn_set_error("dprnn_lstm_pack: incomplete reverse direction"); return -1; }
    const int n = ndir * PACK_PER_DIR;
    hipLaunchKernelGGL(lstm_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, packed, bsum, ndir);
    return dprnn_check_launch("lstm_pack_kernel");
}

static int g_fast_act = 1;   // v_exp_f32/v_rcp_f32 gate activations: 2e-7 rel-L2 end to end, same as the libm forms
extern "C" void dprnn_set_fast_activations(int on) { g_fast_act = on; }

extern "C" int dprnn_bilstm_fwd(const float* X, const float* packed, const float* bsum, float* H,
                                int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                int64_t t_stride, int ndir, int seq_tile, void* stream) {
    if (!X || !packed || !bsum || !H) { dprnn_set_error("dprnn_bilstm_fwd: null pointer"); return -1; }
    if (nseq <= 0 || T <= 0 || inner_n <= 0 || (ndir != 1 && ndir != 2)) {
        dprnn_set_error("dprnn_bilstm_fwd: bad sizes nseq=%d T=%d inner_n=%d ndir=%d", nseq, T, inner_n, ndir);
        return -1;
    }
    if (seq_tile == 0) seq_tile = nseq <= 16 ? 16 : 32;
    RecurArgs a{X, packed, bsum, H, nseq, T, inner_n, ndir, outer_stride, inner_stride, t_stride, nullptr};
    hipStream_t st = (hipStream_t)stream;
    const bool fast = g_fast_act != 0;
    switch (seq_tile) {
        case 16: return fast ? launch_bilstm<1, true>(a, st) : launch_bilstm<1, false>(a, st);
        case 32: return fast ? launch_bilstm<2, true>(a, st) : launch_bilstm<2, false>(a, st);
        case 48: return fast ? launch_bilstm<3, true>(a, st) : launch_bilstm<3, false>(a, st);
        case 64: return fast ? launch_bilstm<4, true>(a, st) : launch_bilstm<4, false>(a, st);
        default: dprnn_set_error("dprnn_bilstm_fwd: seq_tile must be 0,16,32,48,64 (got %d)", seq_tile); return -1;
    }
}

// Diagnostic (tools/bench_recur.py --stamps): same kernel with s_memtime stamps; stamps = [workgroups][4] u64
// {cycles, 100 MHz ticks, k-loop cycles, cell-phase cycles}.  Not part of the product path.
extern "C" int dprnn_bilstm_fwd_stamped(const float* X, const float* packed, const float* bsum, float* H,
                                        int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride,
                                        int64_t t_stride, int ndir, int seq_tile, unsigned long long* stamps, void* stream) {
    if (!X || !packed || !bsum || !H || !stamps) { dprnn_set_error("dprnn_bilstm_fwd_stamped: null pointer"); return -1; }
    RecurArgs a{X, packed, bsum, H, nseq, T, inner_n, ndir, outer_stride, inner_stride, t_stride, stamps};
    hipStream_t st = (hipStream_t)stream;
    auto go = [&](auto kern, int NS) {
        const size_t lds = (size_t)2 * NS * XS * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
        hipLaunchKernelGGL(kern, dim3((nseq + NS - 1) / NS, ndir), dim3(256), lds, st, a);
        return dprnn_check_launch("bilstm_kernel(stamped)");
    };
    switch (seq_tile) {
        case 32: return go(&bilstm_kernel<2, true, true>, 32);
        case 64: return go(&bilstm_kernel<4, true, true>, 64);
        default: dprnn_set_error("dprnn_bilstm_fwd_stamped: seq_tile 32 or 64"); return -1;
    }
}
