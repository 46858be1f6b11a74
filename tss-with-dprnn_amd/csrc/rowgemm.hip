// Main-path instances of the row GEMM (rowgemm_core.h): bottleneck, Linear+GroupNorm sums, mask head.
#include "rowgemm_core.h"
#include <stdlib.h>
using namespace rg;

extern "C" int dprnn_num_cus(void) { return num_cus(); }

#ifndef BOTTLENECK_T_RT
#define BOTTLENECK_T_RT 2               // bottleneck on transposed f16 tiles: 16-row tiles per wave, lookahead across tiles
#endif
#ifndef BOTTLENECK_T_XC
#define BOTTLENECK_T_XC true
#endif
extern "C" int dprnn_get_gemm_mode(void);
static std::atomic<int> g_rowgemm_grid{-1};
int rg::rowgemm_grid_cap() {
    int v = g_rowgemm_grid.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("DPRNN_ROWGEMM_GRID");
        v = e ? atoi(e) : 0;
        if (v < 0) v = 0;
        g_rowgemm_grid.store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" void dprnn_set_rowgemm_grid(int workgroups) { g_rowgemm_grid.store(workgroups > 0 ? workgroups : 0, std::memory_order_relaxed); }
extern "C" int dprnn_get_rowgemm_grid(void) { return rg::rowgemm_grid_cap(); }
static int g_fast_head = 1;
extern "C" void dprnn_set_fast_head_activations(int on) { g_fast_head = on; }
extern "C" int dprnn_get_fast_head_activations(void) { return g_fast_head; }

static int bottleneck(const float* wav, const float* enc_w, const float* sc, const float* sh,
                      const float* Wb, int ldw, const float* bb, const float* row_bias, const float* row_scale, float* Y,
                      int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !sc || !sh || !Wb || !bb || !Y) { dprnn_set_error("dprnn_bottleneck_fwd: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || ldw < N || (ldw % 4) != 0 || L != (T - ks) / stride + 1) {
        dprnn_set_error("dprnn_bottleneck_fwd: unsupported N=%d ks=%d ldw=%d L=%d T=%d stride=%d (built for input_size 64, ks<=8)", N, ks, ldw, L, T, stride);
        return -1;
    }
    if (dprnn_get_gemm_mode() == 3) {          // f16 pieces on transposed tiles (16-byte stores); A = the normalised encoder output
        GenBottleneck<BOTTLENECK_T_RT> alt{wav, enc_w, sc, sh, Geo{B, T, L, N, ks, stride}};
        EpTBiasStore<BOTTLENECK_T_RT, 8> ept{Y, bb, row_bias, (long long)L, row_scale};
        return launch_rowgemm_f16t<64, 128, BOTTLENECK_T_RT, BOTTLENECK_T_XC>("bottleneck_f16", Wb, ldw, 128, nullptr, 0, (long long)B * L, alt, ept, (hipStream_t)stream);
    }
    constexpr int RT = 4;
    GenBottleneck<RT> al{wav, enc_w, sc, sh, Geo{B, T, L, N, ks, stride}};
    EpBiasStore<RT, 8> ep{Y, bb, row_bias, (long long)L, row_scale};
    return launch_rowgemm<64, 128, RT>("bottleneck", Wb, ldw, 128, nullptr, 0, (long long)B * L, al, ep, (hipStream_t)stream);
}

extern "C" int dprnn_bottleneck_fwd(const float* wav, const float* enc_w, const float* sc, const float* sh,
                                    const float* Wb, int ldw, const float* bb, const float* row_bias, float* Y,
                                    int B, int T, int L, int N, int ks, int stride, void* stream) {
    return bottleneck(wav, enc_w, sc, sh, Wb, ldw, bb, row_bias, nullptr, Y, B, T, L, N, ks, stride, stream);
}

extern "C" int dprnn_bottleneck_att_fwd(const float* wav, const float* enc_w, const float* sc, const float* sh,
                                        const float* Wb, int ldw, const float* bb, const float* row_scale, float* Y,
                                        int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!row_scale) { dprnn_set_error("dprnn_bottleneck_att_fwd: row_scale is null"); return -1; }
    return bottleneck(wav, enc_w, sc, sh, Wb, ldw, bb, nullptr, row_scale, Y, B, T, L, N, ks, stride, stream);
}

#ifndef PROJ_NW
#define PROJ_NW 8
#endif


// ------------------------------------------------------------------------------------------------------------------------
// Linear 256|128 -> 128 + GroupNorm sums in f16 pieces (gate-GEMM mode 3): the one row GEMM that runs 24 times per forward,
// written for the HBM stream it is bound by (1 KB in, 512 B out per row):
//  * the MFMA operands are SWAPPED against rowgemm_kernel -- weights as the A operand (16 output columns), 16 rows of H
//    as the B operand -- so a lane ends up with FOUR ADJACENT output columns of ONE row: 16-byte stores (a wave's store
//    instruction writes 1 KB instead of 256 B), and one row, hence one utterance, per lane in the GroupNorm sums;
//  * the rows of the NEXT 32-row tile are requested while the current one is still in the MFMAs (the rotating slab buffers
//    run on across tiles), so a wave always has three k-slabs (12 KB) of loads in flight, epilogue included.
namespace {
typedef rg_u32x4 u32x4_t;

// PIECES: H holds, per four adjacent units, {first f16 pieces x 4 | second pieces x 4} of 2^4 h as written by the recurrence
// (dprnn_rnn_fwd_hp_ex) -- the same 16 bytes per quad and the very words this kernel would form, so the loads ARE the operands:
// no conversion (40 vector instructions per 24 MFMAs otherwise), bit-identical results.
template <int KD, bool PIECES>
__global__ __launch_bounds__(512, 1)
void proj_f16_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ H,
                     float* __restrict__ Z, double* __restrict__ stats, long long M, long long rows_per_b,
                     const long long* __restrict__ starts, int n_utt) {
    constexpr int ND = 128, NCT = 8, NS = KD / 32, K8 = KD / 8, RT = 2, NW = 8, PF = 3;
    static_assert(NS % (PF + 1) == 0, "the slab buffers rotate across tiles");
    extern __shared__ __attribute__((aligned(16))) u32x4_t wl16[];      // [NS][NCT][2][64] + bias
    __shared__ float red[NW];
    float* const bias_l = reinterpret_cast<float*>(wl16 + NS * NCT * 2 * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;

    float m = 0.f;
    for (int idx = tid; idx < ND * KD / 4; idx += 64 * NW) {
        const f32x4 v = reinterpret_cast<const f32x4*>(W)[idx];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    if (tid < ND) bias_l[tid] = bias[tid];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NW; ++i) m = fmaxf(m, red[i]);
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
    ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
    const float sw = ldexpf(1.f, 14 - ex);
    for (int idx = tid; idx < ND * K8; idx += 64 * NW) {
        const int col = idx / K8, k8 = idx % K8;                        // lane (col % 16, k8 % 4): k = 8 k8 .. 8 k8 + 7
        const f32x4* src = reinterpret_cast<const f32x4*>(W + (long long)col * KD + 8 * k8);
        f32x4 lo = src[0], hi = src[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] *= sw; hi[e] *= sw; }
        u32x4_t w1, w2;
        f16_pieces8(lo, hi, w1, w2);
        const int s = k8 >> 2, ln = (k8 & 3) * 16 + (col & 15);
        wl16[((s * NCT + (col >> 4)) * 2 + 0) * 64 + ln] = w1;
        wl16[((s * NCT + (col >> 4)) * 2 + 1) * 64 + ln] = w2;
    }
    __syncthreads();
    const float inv = 1.f / (F16P_XS * sw);

    constexpr int ROWS_IT = NW * RT * 16;
    const long long nchunk = (M + ROWS_IT - 1) / ROWS_IT;
    const long long per = (nchunk + gridDim.x - 1) / gridDim.x;
    const long long c0 = per * blockIdx.x;
    long long c1 = c0 + per;
    if (c1 > nchunk) c1 = nchunk;

    auto rowptr = [&](long long ch, int rt) {
        long long row = ch * ROWS_IT + wave * RT * 16 + 16 * rt + c;
        if (row >= M) row = M - 1;
        return H + row * KD + 8 * q;
    };
    // per-utterance sums of the current utterance (wave-uniform cur_b), flushed when a tile moves on to another one
    double d1 = 0.0, d2 = 0.0;
    long long cur_b = -1;
    auto flush = [&]() {
        if (cur_b >= 0) {
            const double s1 = wave_sum(d1), s2 = wave_sum(d2);
            if (lane == 0) { atomicAdd(&stats[2 * cur_b], s1); atomicAdd(&stats[2 * cur_b + 1], s2); }
        }
        d1 = 0.0; d2 = 0.0;
    };
    auto utt_of = [&](long long row) {
        if (!starts) return row / rows_per_b;
        long long b = cur_b >= 0 && starts[cur_b] <= row ? cur_b : 0;
        while (b + 1 < n_utt && starts[b + 1] <= row) ++b;
        return b;
    };
    auto first_row = [&](long long b) { return starts ? starts[b] : b * rows_per_b; };

    f32x4 a[PF + 1][RT][2];
    if (c0 < c1) {
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float* p = rowptr(c0, rt) + 32 * s;
                a[s][rt][0] = *reinterpret_cast<const f32x4*>(p);
                a[s][rt][1] = *reinterpret_cast<const f32x4*>(p + 4);
            }
    }
    for (long long ch = c0; ch < c1; ++ch) {
        const long long row0 = ch * ROWS_IT + wave * RT * 16;
        const long long chn = ch + 1 < c1 ? ch + 1 : ch;               // last tile: re-request its own rows (never used)
        f32x4 acc[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cb = s % (PF + 1);
            __builtin_amdgcn_sched_barrier(0);                          // keep the slabs apart: without it the scheduler hoists every
            {                                                           // weight fragment of the tile (512 registers) above the first MFMA
                const int sn = s + PF;                                  // slab to request: this tile's, or the next tile's
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float* p = (sn < NS ? rowptr(ch, rt) + 32 * sn : rowptr(chn, rt) + 32 * (sn - NS));
                    a[sn % (PF + 1)][rt][0] = *reinterpret_cast<const f32x4*>(p);
                    a[sn % (PF + 1)][rt][1] = *reinterpret_cast<const f32x4*>(p + 4);
                }
            }
            u32x4_t h1[RT], h2[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (PIECES) {
                    const u32x4_t ql = __builtin_bit_cast(u32x4_t, a[cb][rt][0]), qh = __builtin_bit_cast(u32x4_t, a[cb][rt][1]);
                    h1[rt] = u32x4_t{ql[0], ql[1], qh[0], qh[1]};
                    h2[rt] = u32x4_t{ql[2], ql[3], qh[2], qh[3]};
                    continue;
                }
                f32x4 lo = a[cb][rt][0], hi = a[cb][rt][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                               // rows are RNN outputs, |h| < 1 by construction (o * tanh(c)):
                    lo[e] = __builtin_amdgcn_fmed3f(lo[e] * F16P_XS, -F16P_XMAX, F16P_XMAX);     // 2^4 h cannot leave the f16 range, nothing
                    hi[e] = __builtin_amdgcn_fmed3f(hi[e] * F16P_XS, -F16P_XMAX, F16P_XMAX);     // to track (the clamp only tames NaN garbage)
                }
                f16_pieces8(lo, hi, h1[rt], h2[rt]);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if ((ct & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                const u32x4_t w1 = wl16[((s * NCT + ct) * 2 + 0) * 64 + lane], w2 = wl16[((s * NCT + ct) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {                       // D[col 4q+e of tile ct][row c] (weights are the A operand)
                    acc[rt][ct] = mfma_f16p(w1, h2[rt], acc[rt][ct]);
                    acc[rt][ct] = mfma_f16p(w2, h1[rt], acc[rt][ct]);
                    acc[rt][ct] = mfma_f16p(w1, h1[rt], acc[rt][ct]);
                }
            }
        }
        if (row0 >= M) continue;
        // epilogue: lane (c, q) holds columns 16 ct + 4 q .. + 3 of row row0 + 16 rt + c
        const long long last = (row0 + 16 * RT - 1 < M ? row0 + 16 * RT - 1 : M - 1);
        const long long b_first = utt_of(row0);
        long long b_last = b_first;
        if (starts) { while (b_last + 1 < n_utt && starts[b_last + 1] <= last) ++b_last; } else b_last = last / rows_per_b;
        const bool whole = row0 + 16 * RT <= M && b_last == b_first;
        if (b_first != cur_b) { flush(); cur_b = b_first; }
        float p1 = 0.f, p2 = 0.f, o1 = 0.f, o2 = 0.f;                  // o*: rows of later utterances in a straddling tile
        const long long hi_first = first_row(b_first + 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            const bool ok = whole || row < M;
            const bool mine = whole || row < hi_first;
            float r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 16 * ct + 4 * q);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = fmaf(acc[rt][ct][e], inv, bv[e]); r1 += v[e]; r2 = fmaf(v[e], v[e], r2); }
                if (ok) *reinterpret_cast<f32x4*>(Z + row * ND + 16 * ct + 4 * q) = v;
            }
            if (ok && mine) { p1 += r1; p2 += r2; }
            else if (ok) { o1 += r1; o2 += r2; }
        }
        d1 += (double)p1; d2 += (double)p2;
        if (!whole && b_last != b_first) {
            // rows of the later utterances of this tile (rare: one tile per utterance boundary): one utterance at a time
            for (long long b = b_first + 1; b <= b_last; ++b) {
                flush(); cur_b = b;
                const long long lo = first_row(b), hi = first_row(b + 1);
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const long long row = row0 + 16 * rt + c;
                    if (row < M && row >= lo && row < hi) {
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 16 * ct + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float v = fmaf(acc[rt][ct][e], inv, bv[e]); t1 += v; t2 = fmaf(v, v, t2); }
                        }
                    }
                }
                d1 += (double)t1; d2 += (double)t2;
            }
        }
        (void)o1; (void)o2;
    }
    flush();
}

template <int KD, bool PIECES = false>
int launch_proj_f16(const char* what, const float* W, const float* bias, const float* H, float* Z, double* stats, long long M,
                    long long rows_per_b, const long long* starts, int n_utt, hipStream_t st) {
    const size_t lds = (size_t)KD * 128 * 4 + 128 * 4;
    auto kern = &proj_f16_kernel<KD, PIECES>;
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    const long long grid = rowgemm_f16_grid((M + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, W, bias, H, Z, stats, M, rows_per_b, starts, n_utt);
    return dprnn_check_launch(what);
}
}  // namespace

// gate-GEMM mode 3 (f16 pieces) covers the Linear that follows the recurrence too: its rows are RNN outputs, |h| < 1
#ifndef HEAD_MD_RT
#define HEAD_MD_RT 2                    // mask + decoder on transposed tiles: 16-row tiles per wave, and lookahead across tiles
#endif
#ifndef HEAD_MD_XC
#define HEAD_MD_XC true
#endif
#ifndef HEAD_F16T_RT
#define HEAD_F16T_RT 1                  // 16-row tiles per wave in the transposed-tile head kernels (256 output columns: 2 spill)
#endif

extern "C" int dprnn_proj_stats_fwd(const float* H, const float* W, const float* bias, float* Z, double* stats,
                                    int64_t M, int64_t rows_per_utt, int kdim, void* stream) {
    if (!H || !W || !bias || !Z || !stats) { dprnn_set_error("dprnn_proj_stats_fwd: null pointer"); return -1; }
    if (M <= 0 || rows_per_utt <= 0 || (kdim != 256 && kdim != 128)) {
        dprnn_set_error("dprnn_proj_stats_fwd: bad sizes M=%lld rows_per_utt=%lld kdim=%d", (long long)M, (long long)rows_per_utt, kdim);
        return -1;
    }
    if (dprnn_get_gemm_mode() == 3) {
        if (kdim == 256) return launch_proj_f16<256>("proj_f16_256", W, bias, H, Z, stats, M, rows_per_utt, nullptr, 0, (hipStream_t)stream);
        return launch_proj_f16<128>("proj_f16_128", W, bias, H, Z, stats, M, rows_per_utt, nullptr, 0, (hipStream_t)stream);
    }
    constexpr int RT = 4;
    PlainRows<RT> al{H, kdim};
    EpProjStats<RT, 8> ep{Z, bias, stats, (long long)rows_per_utt, nullptr, 0};
    if (kdim == 256) return launch_rowgemm<256, 128, RT, false, PROJ_NW>("proj_stats256", W, 256, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
    return launch_rowgemm<128, 128, RT, false, PROJ_NW>("proj_stats128", W, 128, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
}

// H as f16 piece quads (written by dprnn_rnn_fwd_hp_ex / dprnn_rnn_fwd_ragged_hp_ex): gate-GEMM mode 3 only; utt_starts != NULL: ragged
extern "C" int dprnn_proj_stats_hp_fwd(const float* Hp, const float* W, const float* bias, float* Z, double* stats,
                                       int64_t M, int64_t rows_per_utt, const int64_t* utt_starts, int n_utt, int kdim, void* stream) {
    if (!Hp || !W || !bias || !Z || !stats) { dprnn_set_error("dprnn_proj_stats_hp_fwd: null pointer"); return -1; }
    if (M <= 0 || (!utt_starts && rows_per_utt <= 0) || (utt_starts && n_utt < 1) || (kdim != 256 && kdim != 128)) {
        dprnn_set_error("dprnn_proj_stats_hp_fwd: bad sizes M=%lld rows_per_utt=%lld n_utt=%d kdim=%d", (long long)M, (long long)rows_per_utt, n_utt, kdim);
        return -1;
    }
    const long long* st_ = reinterpret_cast<const long long*>(utt_starts);
    const long long rpu = utt_starts ? 1 : rows_per_utt;
    if (kdim == 256) return launch_proj_f16<256, true>("proj_f16_256p", W, bias, Hp, Z, stats, M, rpu, st_, utt_starts ? n_utt : 0, (hipStream_t)stream);
    return launch_proj_f16<128, true>("proj_f16_128p", W, bias, Hp, Z, stats, M, rpu, st_, utt_starts ? n_utt : 0, (hipStream_t)stream);
}

// ragged batch: utterance u owns rows [utt_starts[u], utt_starts[u + 1]) of the concatenated slab
extern "C" int dprnn_proj_stats_ragged_fwd(const float* H, const float* W, const float* bias, float* Z, double* stats,
                                           int64_t M, const int64_t* utt_starts, int n_utt, int kdim, void* stream) {
    if (!H || !W || !bias || !Z || !stats || !utt_starts) { dprnn_set_error("dprnn_proj_stats_ragged_fwd: null pointer"); return -1; }
    if (M <= 0 || n_utt < 1 || (kdim != 256 && kdim != 128)) {
        dprnn_set_error("dprnn_proj_stats_ragged_fwd: bad sizes M=%lld n_utt=%d kdim=%d", (long long)M, n_utt, kdim);
        return -1;
    }
    if (dprnn_get_gemm_mode() == 3) {
        const long long* st_ = reinterpret_cast<const long long*>(utt_starts);
        if (kdim == 256) return launch_proj_f16<256>("proj_f16_256r", W, bias, H, Z, stats, M, 1, st_, n_utt, (hipStream_t)stream);
        return launch_proj_f16<128>("proj_f16_128r", W, bias, H, Z, stats, M, 1, st_, n_utt, (hipStream_t)stream);
    }
    constexpr int RT = 4;
    PlainRows<RT> al{H, kdim};
    EpProjStats<RT, 8> ep{Z, bias, stats, 1, reinterpret_cast<const long long*>(utt_starts), n_utt};
    if (kdim == 256) return launch_rowgemm<256, 128, RT, false, PROJ_NW>("proj_stats256r", W, 256, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
    return launch_rowgemm<128, 128, RT, false, PROJ_NW>("proj_stats128r", W, 128, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
}

// The PReLU slope is a device scalar (nn.PReLU weight): the loader reads it through its pointer.
namespace {
template <int RT>
struct OlaGatherP {             // OlaGather with the PReLU slope behind a device pointer
    OlaGather<RT> g; const float* alpha_p;
    __device__ void init(long long row0, int r, int q, long long M) { g.alpha = *alpha_p; g.init(row0, r, q, M); }
    __device__ f32x4 load(int rt, int gi) const { return g.load(rt, gi); }
};
}

static int head_conv_ola(const float* X, const float* Z, const double* stats, const float* gamma, const float* beta, float eps,
                         const float* prelu_w, const float* W2d, const float* b2d, float* U, int B, int L, int K, int P, int S,
                         void* stream) {
    if (!X || !prelu_w || !W2d || !b2d || !U) { dprnn_set_error("dprnn_head_conv_ola_fwd: null pointer"); return -1; }
    if (Z && (!stats || !gamma || !beta)) { dprnn_set_error("dprnn_head_conv_ola_norm_fwd: pending norm needs stats, gamma, beta"); return -1; }
    if (K < 1 || P < 1 || P > K || S != (L + K) / P + 1) {
        dprnn_set_error("dprnn_head_conv_ola_fwd: bad geometry L=%d K=%d P=%d S=%d", L, K, P, S);
        return -1;
    }
    constexpr int RT = 2;
    OlaGatherP<RT> al{OlaGather<RT>{X, 0.f, L, K, P, S, Z, stats, gamma, beta, eps}, prelu_w};
    EpHeadConv<RT> ep{U, b2d, L, K, P, S};
    if (dprnn_get_gemm_mode() == 3) {      // f16 pieces, transposed tiles: the rows are the residual stream the recurrences read too
        OlaGatherP<HEAD_F16T_RT> alt{OlaGather<HEAD_F16T_RT>{X, 0.f, L, K, P, S, Z, stats, gamma, beta, eps}, prelu_w};
        EpTHeadConv<HEAD_F16T_RT> ept{U, b2d, L, K, P, S};
        return launch_rowgemm_f16t<128, 256, HEAD_F16T_RT, false>("head_conv_ola_f16", W2d, 128, 256, nullptr, 0, (long long)B * L, alt, ept, (hipStream_t)stream);
    }
    return launch_rowgemm<128, 256, RT>("head_conv_ola", W2d, 128, 256, nullptr, 0, (long long)B * L, al, ep, (hipStream_t)stream);
}

// ---- the same for a RAGGED batch (round 5): utterances of different lengths, their slabs back to back in X (utterance u: rows utt_starts[u] ..
// utt_starts[u + 1] of 128 floats, S_u K of them) and their frames back to back in U (l_off[u] .. l_off[u + 1]).  One launch over all frames: a row
// finds its utterance by bisection of l_off and then gathers as OlaGather does.  (A launch per utterance stages the 128 KB of conv2d weights in
// every workgroup for a few hundred rows each: 309 us per utterance where the batched launch needs 37, profiles/r05_ragged_ira_kernel_stats.csv.)
namespace {
__device__ __forceinline__ int ragged_utt(const long long* l_off, int n, long long row) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (l_off[mid] <= row) lo = mid; else hi = mid;
    }
    return lo;
}
template <int RT>
struct OlaGatherRagged {
    OlaGather<RT> g; const float* alpha_p; const long long* utt_starts; const long long* l_off; int n;
    __device__ void init(long long row0, int r, int q, long long M) {
        g.alpha = *alpha_p;
        g.q4 = 4 * q;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = row0 + 16 * rt + r;
            if (row >= M) row = M - 1;
            const int u = ragged_utt(l_off, n, row);
            const int l = (int)(row - l_off[u]);
            const long long r0 = utt_starts[u];
            const int S = (int)((utt_starts[u + 1] - r0) / g.K);
            const int lo = OlaGather<RT>::s_lo(l, g.P), hi = OlaGather<RT>::s_hi(l, g.K, g.P, S);
            g.nterm[rt] = hi - lo + 1;
            g.p0[rt] = g.X + (r0 + (long long)lo * g.K + (l + g.K - lo * g.P)) * F + 4 * q;
            g.mu[rt] = 0.f; g.rs[rt] = 0.f;
            if (g.Z) {
                const double cnt = (double)S * (double)g.K * (double)F;
                const double mean = g.stats[2 * u] / cnt;
                double var = g.stats[2 * u + 1] / cnt - mean * mean;
                if (var < 0.0) var = 0.0;
                g.mu[rt] = (float)mean;
                g.rs[rt] = (float)(1.0 / sqrt(var + (double)g.eps));
            }
        }
    }
    __device__ f32x4 load(int rt, int gi) const { return g.load(rt, gi); }
};
template <int RT>
struct EpTHeadConvRagged {      // U[frame][col] = acc + cnt(l) * bias[col], 256 columns; cnt from the frame's own utterance
    float* U; const float* bias; const long long* utt_starts; const long long* l_off; int n, K, P;
    __device__ void begin(int tid) {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][16], long long row0, int c, int q, long long M, float inv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            if (row >= M) continue;
            const int u = ragged_utt(l_off, n, row);
            const int l = (int)(row - l_off[u]);
            const int S = (int)((utt_starts[u + 1] - utt_starts[u]) / K);
            const float cnt = (float)(OlaGather<1>::s_hi(l, K, P, S) - OlaGather<1>::s_lo(l, P) + 1);
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 16 * ct + 4 * q);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(cnt, b[e], acc[rt][ct][e] * inv);
                *reinterpret_cast<f32x4*>(U + row * 256 + 16 * ct + 4 * q) = v;
            }
        }
    }
};
}

extern "C" int dprnn_head_conv_ola_ragged_fwd(const float* X, const float* Z, const double* stats, const float* gamma, const float* beta,
                                              float eps, const float* prelu_w, const float* W2d, const float* b2d, float* U,
                                              const int64_t* utt_starts, const int64_t* l_off, int n, int64_t total_frames, int K, int P,
                                              void* stream) {
    if (!X || !prelu_w || !W2d || !b2d || !U || !utt_starts || !l_off) { dprnn_set_error("dprnn_head_conv_ola_ragged_fwd: null pointer"); return -1; }
    if (Z && (!stats || !gamma || !beta)) { dprnn_set_error("dprnn_head_conv_ola_ragged_fwd: pending norm needs stats, gamma, beta"); return -1; }
    if (n < 1 || total_frames < 1 || K < 1 || P < 1 || P > K) {
        dprnn_set_error("dprnn_head_conv_ola_ragged_fwd: bad sizes n=%d frames=%lld K=%d P=%d", n, (long long)total_frames, K, P);
        return -1;
    }
    if (dprnn_get_gemm_mode() != 3) { dprnn_set_error("dprnn_head_conv_ola_ragged_fwd: gate-GEMM mode 3 (f16 pieces) only"); return -1; }
    static_assert(sizeof(long long) == sizeof(int64_t), "table element size");
    const long long* us = reinterpret_cast<const long long*>(utt_starts);
    const long long* lo = reinterpret_cast<const long long*>(l_off);
    OlaGatherRagged<HEAD_F16T_RT> al{OlaGather<HEAD_F16T_RT>{X, 0.f, 0, K, P, 0, Z, stats, gamma, beta, eps}, prelu_w, us, lo, n};
    EpTHeadConvRagged<HEAD_F16T_RT> ep{U, b2d, us, lo, n, K, P};
    return launch_rowgemm_f16t<128, 256, HEAD_F16T_RT, false>("head_conv_ola_ragged_f16", W2d, 128, 256, nullptr, 0, (long long)total_frames, al, ep,
                                                              (hipStream_t)stream);
}

extern "C" int dprnn_head_conv_ola_fwd(const float* X, const float* prelu_w, const float* W2d, const float* b2d,
                                       float* U, int B, int L, int K, int P, int S, void* stream) {
    return head_conv_ola(X, nullptr, nullptr, nullptr, nullptr, 0.f, prelu_w, W2d, b2d, U, B, L, K, P, S, stream);
}

extern "C" int dprnn_head_conv_ola_norm_fwd(const float* X, const float* Z, const double* stats, const float* gamma,
                                            const float* beta, float eps, const float* prelu_w, const float* W2d,
                                            const float* b2d, float* U, int B, int L, int K, int P, int S, void* stream) {
    if (!Z) { dprnn_set_error("dprnn_head_conv_ola_norm_fwd: Z is null (use dprnn_head_conv_ola_fwd)"); return -1; }
    return head_conv_ola(X, Z, stats, gamma, beta, eps, prelu_w, W2d, b2d, U, B, L, K, P, S, stream);
}

static int head_gate(const float* U, const float* Wo, const float* bo, const float* Wg, const float* bg,
                     float* V, float* TG, int64_t rows, void* stream) {
    if (!U || !Wo || !bo || !Wg || !bg || !V || rows <= 0) { dprnn_set_error("dprnn_head_gate_fwd: null pointer / rows"); return -1; }
    constexpr int RT = 2;
    PlainRows<RT> al{U, 128};
    if (g_fast_head) {
        EpGate<RT, true> ep{V, bo, bg, TG};
        if (dprnn_get_gemm_mode() == 3) {
            PlainRows<HEAD_F16T_RT> alt{U, 128};
            EpTGate<HEAD_F16T_RT, true> ept{V, bo, bg, TG};
            return launch_rowgemm_f16t<128, 256, HEAD_F16T_RT, true>("head_gate_f16", Wo, 128, 128, Wg, 128, rows, alt, ept, (hipStream_t)stream);
        }
        return launch_rowgemm<128, 256, RT>("head_gate", Wo, 128, 128, Wg, 128, rows, al, ep, (hipStream_t)stream);
    }
    EpGate<RT, false> ep{V, bo, bg, TG};
    return launch_rowgemm<128, 256, RT>("head_gate", Wo, 128, 128, Wg, 128, rows, al, ep, (hipStream_t)stream);
}

extern "C" int dprnn_head_gate_fwd(const float* U, const float* Wo, const float* bo, const float* Wg, const float* bg,
                                   float* V, int64_t rows, void* stream) {
    return head_gate(U, Wo, bo, Wg, bg, V, nullptr, rows, stream);
}

extern "C" int dprnn_head_gate_fwd_train(const float* U, const float* Wo, const float* bo, const float* Wg, const float* bg,
                                         float* V, float* TG, int64_t rows, void* stream) {
    if (!TG) { dprnn_set_error("dprnn_head_gate_fwd_train: TG is null"); return -1; }
    return head_gate(U, Wo, bo, Wg, bg, V, TG, rows, stream);
}

static int head_mask_decode(const float* V, const float* Wend, const float* wav, const float* enc_w,
                            const float* dec_w, float* taps, float* masked_out, float* mask_all, int relu_act,
                            int B, int T, int L, int N, int ks, int stride, void* stream);

extern "C" int dprnn_head_mask_decode_fwd(const float* V, const float* Wend, const float* wav, const float* enc_w,
                                          const float* dec_w, float* taps, float* masked_out, int relu_act,
                                          int B, int T, int L, int N, int ks, int stride, void* stream) {
    return head_mask_decode(V, Wend, wav, enc_w, dec_w, taps, masked_out, nullptr, relu_act, B, T, L, N, ks, stride, stream);
}

extern "C" int dprnn_head_mask_decode_fwd_train(const float* V, const float* Wend, const float* wav, const float* enc_w,
                                                const float* dec_w, float* taps, float* mask, float* masked_out, int relu_act,
                                                int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!mask) { dprnn_set_error("dprnn_head_mask_decode_fwd_train: mask is null"); return -1; }
    return head_mask_decode(V, Wend, wav, enc_w, dec_w, taps, masked_out, mask, relu_act, B, T, L, N, ks, stride, stream);
}

static int head_mask_decode(const float* V, const float* Wend, const float* wav, const float* enc_w,
                            const float* dec_w, float* taps, float* masked_out, float* mask_all, int relu_act,
                            int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!V || !Wend || !wav || !enc_w || !dec_w || !taps) { dprnn_set_error("dprnn_head_mask_decode_fwd: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || L != (T - ks) / stride + 1) {
        dprnn_set_error("dprnn_head_mask_decode_fwd: unsupported N=%d ks=%d (built for input_size 64, ks<=8)", N, ks);
        return -1;
    }
    if ((long long)B * L * 2 >= (1ll << 31)) { dprnn_set_error("dprnn_head_mask_decode_fwd: B*L*2 = %lld rows do not fit 31 bits", (long long)B * L * 2); return -1; }
    constexpr int RT = 4;
    PlainRows<RT> al{V, 128};
    const Geo geo{B, T, L, N, ks, stride};
    if (g_fast_head) {
        EpMaskDecode<RT, 4, true> ep{wav, enc_w, dec_w, taps, masked_out, relu_act, geo, mask_all};
        if (dprnn_get_gemm_mode() == 3) {      // rows are tanh * sigmoid products, |v| < 1
            PlainRows<HEAD_MD_RT> alt{V, 128};
            if (ks <= 2) {                                   // the reference's configurations use kernel_size 2
                EpTMaskDecode<HEAD_MD_RT, true, 2> ept{wav, enc_w, dec_w, taps, masked_out, relu_act, geo, mask_all};
                return launch_rowgemm_f16t<128, 64, HEAD_MD_RT, HEAD_MD_XC, false, false>("head_mask_decode_f16", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, alt, ept, (hipStream_t)stream);
            }
            EpTMaskDecode<HEAD_MD_RT, true, KSMAX> ept{wav, enc_w, dec_w, taps, masked_out, relu_act, geo, mask_all};
            return launch_rowgemm_f16t<128, 64, HEAD_MD_RT, false, false, false>("head_mask_decode_f16", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, alt, ept, (hipStream_t)stream);
        }
        return launch_rowgemm<128, 64, RT>("head_mask_decode", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, al, ep, (hipStream_t)stream);
    }
    EpMaskDecode<RT, 4, false> ep{wav, enc_w, dec_w, taps, masked_out, relu_act, geo, mask_all};
    return launch_rowgemm<128, 64, RT>("head_mask_decode", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, al, ep, (hipStream_t)stream);
}
