// Row GEMMs of the DPRNN path: every 1x1 conv / Linear is out[row][0:ND] = A[row][0:KD] @ W^T with a
// fused prologue (how A rows are produced) and epilogue (what happens to the ND outputs of a row).
//
// Structure (gfx950): persistent workgroups (one per CU, 4 waves).  The whole fp32 weight matrix
// (KD*ND <= 32768 floats = 128 KB) is staged ONCE per workgroup into LDS, already in
// v_mfma_f32_16x16x4_f32 B-fragment order ([k-group][col-tile][lane] float4 -> one conflict-free
// ds_read_b128 per fragment), straight from the row-major torch parameter: no packed weight copies.
// Each wave then walks its own tiles of 16*RT rows: A fragments come straight from HBM as 16-byte
// loads in fragment order (lane (r,q) reads floats 16g+4q..+3 of row r: k = 16g + 4q + i for MFMA
// sub-step i), one k-group prefetched ahead; accumulators stay in registers; no barrier in the loop.
//
// Instances (reference lines):
//   bottleneck   KD=64  ND=128  A generated from the waveform (encoder + norm affine)   dprnn.py:131-136,169
//   proj+stats   KD=256/128 ND=128  intra/inter Linear + GroupNorm sums                 dprnn.py:88,96 (+90,98)
//   head conv    KD=128 ND=256  A = overlap-add of PReLU rows                           dprnn.py:173-178
//   head gate    KD=128 ND=256  tanh(out) * sigmoid(gate)                               dprnn.py:157-158,180
//   mask+decode  KD=128 ND=64   end conv, activation, mask*enc, decoder taps            dprnn.py:181-184,277
#pragma once
#include "common.h"

namespace rg {

constexpr int F = 128;
constexpr int KSMAX = 8;     // max encoder kernel_size supported by the generated-A / decode paths

struct Geo {                 // shared row -> (b, l) geometry
    int B, T, L, N, ks, stride;
};

// ------------------------------------------------------------------------------------------------ A loaders
// Each loader is initialised per wave-tile with the global row of this lane for every row tile rt and
// returns the lane's float4 of k-group g.

template <int RT>
struct PlainRows {
    const float* A; int pitch;
    const float* p[RT];
    __device__ void init(long long row0, int r, int q, long long M) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = row0 + 16 * rt + r;
            if (row >= M) row = M - 1;
            p[rt] = A + row * pitch + 4 * q;
        }
    }
    __device__ f32x4 load(int rt, int g) const { return *reinterpret_cast<const f32x4*>(p[rt] + 16 * g); }
};

// a[b,l,c] = relu(sum_j We[c][j]*wav[b][l*stride+j]) * sc[b][c] + sh[b][c]
template <int RT>
struct GenBottleneck {
    const float* wav; const float* We; const float* sc; const float* sh; Geo geo;
    float xv[RT][KSMAX];
    int boff[RT];
    int q4;
    __device__ void init(long long row0, int r, int q, long long M) {
        q4 = 4 * q;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = row0 + 16 * rt + r;
            if (row >= M) row = M - 1;
            const int b = (int)(row / geo.L), l = (int)(row % geo.L);
            boff[rt] = b * geo.N;
            const float* x = wav + (long long)b * geo.T + (long long)l * geo.stride;
#pragma unroll
            for (int j = 0; j < KSMAX; ++j) xv[rt][j] = (j < geo.ks) ? x[j] : 0.f;
        }
    }
    __device__ f32x4 load(int rt, int g) const {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cch = 16 * g + q4 + e;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KSMAX; ++j)
                if (j < geo.ks) s = fmaf(We[cch * geo.ks + j], xv[rt][j], s);
            s = relu_f(s);
            v[e] = fmaf(s, sc[boff[rt] + cch], sh[boff[rt] + cch]);
        }
        return v;
    }
};

// a[(b,l)][f] = sum over chunks s covering frame l of prelu(x'[b][s][l+K-s*P][f])   (dprnn.py:173,203-217)
// x' = X, or X + ((Z - mean_b)*rstd_b*gamma + beta) when the last path's norm + residual is still pending (Z != null)
template <int RT>
struct OlaGather {
    const float* X; float alpha; int L, K, P, S;
    const float* Z; const double* stats; const float* gamma; const float* beta; float eps;
    const float* p0[RT];       // row of the first term
    int nterm[RT];
    float mu[RT], rs[RT];
    int q4;
    __device__ static int s_lo(int l, int P) { return (l + P) / P; }                       // ceil((l+1)/P)
    __device__ static int s_hi(int l, int K, int P, int S) { int s = (l + K) / P; return s < S - 1 ? s : S - 1; }
    __device__ void init(long long row0, int r, int q, long long M) {
        q4 = 4 * q;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = row0 + 16 * rt + r;
            if (row >= M) row = M - 1;
            const int b = (int)(row / L), l = (int)(row % L);
            const int lo = s_lo(l, P), hi = s_hi(l, K, P, S);
            nterm[rt] = hi - lo + 1;
            // term s: row ((b*S + s)*K + (l + K - s*P)); successive s differ by (K - P) rows
            p0[rt] = X + (((long long)b * S + lo) * K + (l + K - lo * P)) * F + 4 * q;
            mu[rt] = 0.f; rs[rt] = 0.f;
            if (Z) {
                const double cnt = (double)S * (double)K * (double)F;
                const double mean = stats[2 * b] / cnt;
                double var = stats[2 * b + 1] / cnt - mean * mean;
                if (var < 0.0) var = 0.0;
                mu[rt] = (float)mean;
                rs[rt] = (float)(1.0 / sqrt(var + (double)eps));
            }
        }
    }
    __device__ f32x4 load(int rt, int g) const {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* p = p0[rt] + 16 * g;
        f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f};
        if (Z) {
            gm = *reinterpret_cast<const f32x4*>(gamma + 16 * g + q4);
            bt = *reinterpret_cast<const f32x4*>(beta + 16 * g + q4);
        }
        for (int s = 0; s < nterm[rt]; ++s, p += (long long)(K - P) * F) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p);
            if (Z) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(Z + (p - X));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((z[e] - mu[rt]) * rs[rt] * gm[e] + bt[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e] >= 0.f ? v[e] : alpha * v[e];
        }
        return acc;
    }
};

// ------------------------------------------------------------------------------------------------ epilogues
// acc[rt][ct][e]: row = row0 + 16*rt + 4*q + e, col = 16*ct + r.

template <int RT, int NCT>
struct EpBiasStore {            // out[row][col] = acc*row_scale[row] + bias[col] (+ row_bias[row / rows_per_b][col])
    float* out; const float* bias; const float* row_bias; long long rows_per_b; const float* row_scale;
    __device__ void begin() {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int r, int q, int lane, long long M) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = 16 * ct + r;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long row = row0 + 16 * rt + 4 * q + e;
                    if (row < M) {
                        float v = acc[rt][ct][e];
                        if (row_scale) v *= row_scale[row];
                        v += bv;
                        if (row_bias) v += row_bias[(row / rows_per_b) * (16 * NCT) + col];
                        out[row * (16 * NCT) + col] = v;
                    }
                }
        }
    }
};

template <int RT, int NCT>
struct EpStore {                // out[row*pitch + col] = acc: plain store, one base pointer per row on full tiles
    float* out; long long pitch;
    __device__ void begin() {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int r, int q, int lane, long long M) {
        const bool full = row0 + 16 * RT <= M;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long row = row0 + 16 * rt + 4 * q + e;
                if (full || row < M) {
                    float* o = out + row * pitch + r;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) o[16 * ct] = acc[rt][ct][e];
                }
            }
    }
};

template <int RT, int NCT>
struct EpProjStats {            // Z = acc + bias; per-utterance (sum, sumsq) in double
    float* Z; const float* bias; double* stats; long long rows_per_b;
    const long long* starts; int n_utt;        // ragged batches: first row of every utterance ([n_utt + 1]); nullptr = uniform
    double d1, d2; long long cur_b;
    __device__ void begin() { d1 = 0.0; d2 = 0.0; cur_b = -1; }
    __device__ void flush(int lane) {
        if (cur_b >= 0) {
            const double s1 = wave_sum(d1), s2 = wave_sum(d2);
            if (lane == 0) { atomicAdd(&stats[2 * cur_b], s1); atomicAdd(&stats[2 * cur_b + 1], s2); }
        }
        d1 = 0.0; d2 = 0.0;
    }
    __device__ void end(int lane) { flush(lane); cur_b = -1; }
    // utterance of a row; a wave walks its row tiles in increasing order, so the scan resumes from the current utterance
    __device__ long long utt_of(long long row) const {
        if (!starts) return row / rows_per_b;
        long long b = cur_b >= 0 && starts[cur_b] <= row ? cur_b : 0;
        while (b + 1 < n_utt && starts[b + 1] <= row) ++b;
        return b;
    }
    __device__ long long first_row(long long b) const { return starts ? starts[b] : b * rows_per_b; }
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int r, int q, int lane, long long M) {
        float bv[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) bv[ct] = bias[16 * ct + r];
        const long long last = (row0 + 16 * RT - 1 < M ? row0 + 16 * RT - 1 : M - 1);
        const long long b_first = utt_of(row0);
        const long long next_start = first_row(b_first + 1);
        if (row0 + 16 * RT <= M && last < next_start) {
            // fast path (almost every tile): all 16*RT rows exist and belong to one utterance -- no per-row tests,
            // one base pointer per row, immediate column offsets (fp32 VALU work costs matrix time on gfx950)
            if (b_first != cur_b) { flush(lane); cur_b = b_first; }
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* zr = Z + (row0 + 16 * rt + 4 * q + e) * (16 * NCT) + r;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const float v = acc[rt][ct][e] + bv[ct];
                        zr[16 * ct] = v;
                        p1 += v; p2 = fmaf(v, v, p2);
                    }
                }
            d1 += (double)p1; d2 += (double)p2;
            return;
        }
        // general path: ragged last tile and tiles that straddle utterances (one utterance at a time)
        long long b_last = b_first;
        if (starts) { while (b_last + 1 < n_utt && starts[b_last + 1] <= last) ++b_last; } else b_last = last / rows_per_b;
        for (long long b = b_first; b <= b_last; ++b) {
            if (b != cur_b) { flush(lane); cur_b = b; }
            const long long lo = first_row(b), hi = first_row(b + 1);
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long row = row0 + 16 * rt + 4 * q + e;
                    if (row < M && row >= lo && row < hi) {
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            const float v = acc[rt][ct][e] + bv[ct];
                            Z[row * (16 * NCT) + 16 * ct + r] = v;
                            p1 += v; p2 = fmaf(v, v, p2);
                        }
                    }
                }
            d1 += (double)p1; d2 += (double)p2;
        }
    }
};

template <int RT>
struct EpHeadConv {             // U[(b,l)][col] = acc + cnt(l)*bias[col], 256 columns
    float* U; const float* bias; int L, K, P, S;
    __device__ void begin() {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][16], long long row0, int r, int q, int lane, long long M) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long row = row0 + 16 * rt + 4 * q + e;
                if (row < M) {
                    const int l = (int)(row % L);
                    const float cnt = (float)(OlaGather<1>::s_hi(l, K, P, S) - OlaGather<1>::s_lo(l, P) + 1);
#pragma unroll
                    for (int ct = 0; ct < 16; ++ct)
                        U[row * 256 + 16 * ct + r] = fmaf(cnt, bias[16 * ct + r], acc[rt][ct][e]);
                }
            }
    }
};

template <int RT, bool FAST>
struct EpGate {                 // V[row][col] = tanh(acc[col]+bo[col]) * sigmoid(acc[128+col]+bg[col])
    float* V; const float* bo; const float* bg;
    float* TG = nullptr;        // training forward: also keeps [tanh | sigmoid] as [row][256]
    __device__ void begin() {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][16], long long row0, int r, int q, int lane, long long M) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const float b1 = bo[16 * ct + r], b2 = bg[16 * ct + r];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long row = row0 + 16 * rt + 4 * q + e;
                    if (row < M) {
                        const float ta = tanh_f<FAST>(acc[rt][ct][e] + b1), sg = sigmoid_f<FAST>(acc[rt][ct + 8][e] + b2);
                        V[row * F + 16 * ct + r] = ta * sg;
                        if (TG) { TG[row * 256 + 16 * ct + r] = ta; TG[row * 256 + 128 + 16 * ct + r] = sg; }
                    }
                }
        }
    }
};

// rows are (b, l, spk): mask = act(acc[n]); me = mask*enc[b,n,l]; taps[b][spk][l][j] = sum_n Wd[n][j]*me
template <int RT, int NCT, bool FAST>
struct EpMaskDecode {
    const float* wav; const float* We; const float* Wd; float* taps; float* masked; int relu_act; Geo geo;
    float* mask_all = nullptr;  // training forward: also keeps the mask as [row][N]
    __device__ void begin() {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int r, int q, int lane, long long M) {
        const int ks = geo.ks;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                long long row = row0 + 16 * rt + 4 * q + e;
                const bool ok = row < M;
                if (!ok) row = M - 1;
                const int spk = (int)(row & 1);
                const long long bl = row >> 1;
                const int b = (int)(bl / geo.L), l = (int)(bl % geo.L);
                const float* x = wav + (long long)b * geo.T + (long long)l * geo.stride;
                float xv[KSMAX];
#pragma unroll
                for (int j = 0; j < KSMAX; ++j) xv[j] = j < ks ? x[j] : 0.f;
                float tp[KSMAX];
#pragma unroll
                for (int j = 0; j < KSMAX; ++j) tp[j] = 0.f;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int n = 16 * ct + r;
                    float en = 0.f;
#pragma unroll
                    for (int j = 0; j < KSMAX; ++j)
                        if (j < ks) en = fmaf(We[n * ks + j], xv[j], en);
                    en = relu_f(en);
                    const float a = acc[rt][ct][e];
                    const float m = relu_act ? relu_f(a) : sigmoid_f<FAST>(a);
                    const float me = m * en;
                    if (masked && ok && spk == 0) masked[bl * geo.N + n] = me;
                    if (mask_all && ok) mask_all[row * geo.N + n] = m;
#pragma unroll
                    for (int j = 0; j < KSMAX; ++j)
                        if (j < ks) tp[j] = fmaf(Wd[n * ks + j], me, tp[j]);
                }
#pragma unroll
                for (int j = 0; j < KSMAX; ++j)
                    if (j < ks) {
                        float v = tp[j];
                        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
                        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                        if (ok && r == 0) taps[(((long long)b * 2 + spk) * geo.L + l) * ks + j] = v;
                    }
            }
    }
};

// ------------------------------------------------------------------------------------------------ kernel
// WT = false: out = A @ W^T with W [ND][ldw] row-major (W0: columns < nd0, W1: the rest) -- the forward products.
// WT = true (backward, data gradients): out = A @ W with W [KD][ldw] row-major (W0: k < nd0, W1: the rest), i.e. the
// same torch parameter read transposed while it is staged.
// NW: waves per workgroup (4 or 8).  The workgroup count is one per CU (the weights fill most of the LDS), so NW = 8 puts
// two waves on every SIMD: one wave's HBM row loads and epilogue stores overlap the other's MFMAs.
template <int KD, int ND, int RT, class AL, class EP, bool WT = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1)
void rowgemm_kernel(const float* __restrict__ W0, int ldw0, int nd0, const float* __restrict__ W1, int ldw1,
                    long long M, AL al, EP ep) {
    constexpr int NCT = ND / 16, NG = KD / 16, K4 = KD / 4;
    extern __shared__ __attribute__((aligned(16))) f32x4 wl[];          // [NG][NCT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;

    if constexpr (!WT) {
        for (int idx = tid; idx < ND * K4; idx += 64 * NW) {
            const int col = idx / K4, k4 = idx % K4;
            const float* src = col < nd0 ? W0 + (long long)col * ldw0 : W1 + (long long)(col - nd0) * ldw1;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * k4);
            wl[((k4 >> 2) * NCT + (col >> 4)) * 64 + (k4 & 3) * 16 + (col & 15)] = v;
        }
    } else {
        for (int idx = tid; idx < ND * K4; idx += 64 * NW) {
            const int col = idx % ND, k4 = idx / ND;            // consecutive threads read consecutive columns
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * k4 + e;
                v[e] = k < nd0 ? W0[(long long)k * ldw0 + col] : W1[(long long)(k - nd0) * ldw1 + col];
            }
            wl[((k4 >> 2) * NCT + (col >> 4)) * 64 + (k4 & 3) * 16 + (col & 15)] = v;
        }
    }
    __syncthreads();

    constexpr int ROWS_IT = NW * RT * 16;
    const long long nchunk = (M + ROWS_IT - 1) / ROWS_IT;
    const long long per = (nchunk + gridDim.x - 1) / gridDim.x;
    long long c0 = per * blockIdx.x, c1 = c0 + per;
    if (c1 > nchunk) c1 = nchunk;
    ep.begin();
    for (long long ch = c0; ch < c1; ++ch) {
        const long long row0 = ch * ROWS_IT + wave * RT * 16;
        if (row0 >= M) break;
        al.init(row0, r, q, M);
        f32x4 acc[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A fragments come straight from HBM: keep PF k-groups in flight (one group of MFMAs is ~4096 cycles, about one
        // loaded HBM latency, so a single group of lookahead leaves the wave waiting)
        constexpr int PF = (NG >= 4) ? 2 : 1;
        f32x4 a[PF + 1][RT];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[p][rt] = al.load(rt, p);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int cb = g % (PF + 1);
            if (g + PF < NG) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[(g + PF) % (PF + 1)][rt] = al.load(rt, g + PF);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 b = wl[(g * NCT + ct) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma16(a[cb][rt][i], b[i], acc[rt][ct]);
            }
        }
        ep(acc, row0, r, q, lane, M);
    }
    ep.end(lane);
}


// ------------------------------------------------------------------------------------------------ f16-pieces form
// The same persistent row GEMM with every fp32 operand written as TWO f16 pieces of a power-of-two multiple of itself
// (recur_split.hip explains the arithmetic: 22+ operand bits, three products on v_mfma_f32_16x16x32_f16, the dropped term
// ~2^-24 |a||w|): the fp32 form above is bound by the fp32 MFMA (2048 matrix cycles per 16x16x256 tile; this one 384), which
// leaves the kernel to the HBM stream.  Rows are scaled by 2^4 and clamped to +-4062 like the recurrence's activations.  The weight scale 2^e
// (max|W| into [2^13, 2^14)) is found by every workgroup while it stages W, so there is no packed copy and no extra launch.
// LDS: [slab KD/32][col tile][piece 2][lane] 16-byte B fragments (lane (c, q): column 16 ct + c, k = 32 s + 8 q .. + 7).
typedef _Float16 rg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rg_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned rg_u32x4 __attribute__((ext_vector_type(4)));
constexpr float F16P_XS = 16.f, F16P_XMAX = 65000.f;

__device__ __forceinline__ void f16_pieces2(float s0, float s1, unsigned& w1, unsigned& w2) {
    const rg_f16x2 h = __builtin_convertvector(f32x2{s0, s1}, rg_f16x2);
    w1 = __builtin_bit_cast(unsigned, h);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0 - (float)h[0], s1 - (float)h[1]}, rg_f16x2));
}
__device__ __forceinline__ void f16_pieces8(const f32x4& lo, const f32x4& hi, rg_u32x4& w1, rg_u32x4& w2) {
    unsigned x1[4], x2[4];
    f16_pieces2(lo[0], lo[1], x1[0], x2[0]); f16_pieces2(lo[2], lo[3], x1[1], x2[1]);
    f16_pieces2(hi[0], hi[1], x1[2], x2[2]); f16_pieces2(hi[2], hi[3], x1[3], x2[3]);
    w1 = rg_u32x4{x1[0], x1[1], x1[2], x1[3]}; w2 = rg_u32x4{x2[0], x2[1], x2[2], x2[3]};
}
__device__ __forceinline__ f32x4 mfma_f16p(rg_u32x4 a, rg_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, a), __builtin_bit_cast(rg_f16x8, b), c, 0, 0, 0);
}

// A loaders and f16 slabs: two of a loader's 16-byte loads make one lane's eight k values of a 32-wide slab (k = 32 s + 4 q + i and
// 32 s + 16 + 4 q + i) -- the MFMA does not care which k a slot holds as long as the weight fragments agree, and this keeps every
// loader's fp32 fragment addressing.

inline int num_cus() {
    static int g_num_cus = 0;
    if (!g_num_cus) {
        int dev = 0; (void)hipGetDevice(&dev);
        hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, dev);
        g_num_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return g_num_cus;
}

// Workgroups of the f16-pieces row GEMMs (projection, bottleneck, mask head: bound by their HBM streams, not by the CUs): a
// persistent workgroup holds a CU's LDS, so the grid is also the number of CUs the kernel takes away from whatever else is on the
// chip -- the other utterance group's recurrence.  dprnn_set_rowgemm_grid(n) caps it (0 = one workgroup per CU); the default
// comes from DPRNN_ROWGEMM_GRID.  (rowgemm.hip)
int rowgemm_grid_cap();
inline long long rowgemm_f16_grid(long long nchunk) {
    long long cap = rowgemm_grid_cap();
    if (cap <= 0 || cap > num_cus()) cap = num_cus();
    long long grid = nchunk < cap ? nchunk : cap;
    return grid < 1 ? 1 : grid;
}

template <int KD, int ND, int RT, bool WT = false, int NW = 4, class AL, class EP>
int launch_rowgemm(const char* what, const float* W0, int ldw0, int nd0, const float* W1, int ldw1, long long M,
                   const AL& al, const EP& ep, hipStream_t st) {
    const size_t lds = (size_t)KD * ND * sizeof(float);
    auto kern = &rowgemm_kernel<KD, ND, RT, AL, EP, WT, NW>;
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    constexpr int ROWS_IT = NW * RT * 16;
    long long nchunk = (M + ROWS_IT - 1) / ROWS_IT;
    long long grid = nchunk < num_cus() ? nchunk : num_cus();
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, st, W0, ldw0, nd0, W1, ldw1, M, al, ep);
    return dprnn_check_launch(what);
}


// ------------------------------------------------------------------------------------------------ f16 pieces, transposed tiles
// The f16-pieces row GEMM: MFMA operands swapped against rowgemm_kernel (weights as the A operand, 16 rows of the input as the B operand), as in
// proj_f16_kernel (rowgemm.hip): acc[rt][ct][e] is row row0 + 16 rt + c, column 16 ct + 4 q + e (c = lane & 15, q = lane >> 4), so
// an epilogue stores 16 bytes per lane and a wave's store instruction writes 1 KB instead of 256 B -- the 4-byte stores of the
// plain layout, not HBM, bound the mask-head kernels (1.2-1.5 TB/s).  XC: the loader is cheap enough to keep a second copy
// for the next row tile, whose k-slabs are requested while the current tile is in the MFMAs (PlainRows).
template <int RT, bool FAST>
struct EpTGate {                // V[row][col] = tanh(acc[col] + bo[col]) * sigmoid(acc[128 + col] + bg[col])
    float* V; const float* bo; const float* bg;
    float* TG = nullptr;        // training forward: also keeps [tanh | sigmoid] as [row][256]
    __device__ void begin(int tid) {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][16], long long row0, int c, int q, long long M, float inv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            if (row >= M) continue;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bo + 16 * ct + 4 * q), b2 = *reinterpret_cast<const f32x4*>(bg + 16 * ct + 4 * q);
                f32x4 v, ta, sg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ta[e] = tanh_f<FAST>(fmaf(acc[rt][ct][e], inv, b1[e]));
                    sg[e] = sigmoid_f<FAST>(fmaf(acc[rt][ct + 8][e], inv, b2[e]));
                    v[e] = ta[e] * sg[e];
                }
                *reinterpret_cast<f32x4*>(V + row * F + 16 * ct + 4 * q) = v;
                if (TG) {
                    *reinterpret_cast<f32x4*>(TG + row * 256 + 16 * ct + 4 * q) = ta;
                    *reinterpret_cast<f32x4*>(TG + row * 256 + 128 + 16 * ct + 4 * q) = sg;
                }
            }
        }
    }
};

template <int RT>
struct EpTHeadConv {            // U[(b,l)][col] = acc + cnt(l)*bias[col], 256 columns
    float* U; const float* bias; int L, K, P, S;
    __device__ void begin(int tid) {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][16], long long row0, int c, int q, long long M, float inv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            if (row >= M) continue;
            const int l = (int)(row % L);
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

// WT: out = A @ W with W [KD][ldw] row-major (W0: k < nd0, W1: the rest) -- the data gradients; amax (may be null): device pointer
// to max|A|, which replaces the fixed 2^4 activation scale by the power of two that puts it at 2^6 (gradients have no natural scale).
template <int RT, int NCT>
struct EpTStore {               // out[row*pitch + col] = acc; amax_out (may be null): max |out| of the launch, as float bits (zeroed by the caller)
    float* out; int pitch;
    unsigned* amax_out = nullptr;
    float mx = 0.f;
    __device__ void begin(int tid) {}
    __device__ void end(int lane) {
        if (!amax_out) return;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0 && mx > 0.f) atomicMax(amax_out, __float_as_uint(mx));
    }
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int c, int q, long long M, float inv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            if (row >= M) continue;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 v = acc[rt][ct] * f32x4{inv, inv, inv, inv};
                *reinterpret_cast<f32x4*>(out + row * pitch + 16 * ct + 4 * q) = v;
                if (amax_out) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
        }
    }
};

template <int RT, int NCT>
struct EpTBiasStore {           // out[row][col] = acc*row_scale[row] + bias[col] (+ row_bias[row / rows_per_b][col]), 16 NCT columns
    float* out; const float* bias; const float* row_bias; long long rows_per_b; const float* row_scale;
    __device__ void begin(int tid) {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int c, int q, long long M, float inv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            if (row >= M) continue;
            const float rs = row_scale ? inv * row_scale[row] : inv;
            const float* rb = row_bias ? row_bias + (row / rows_per_b) * (16 * NCT) + 4 * q : nullptr;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 v = *reinterpret_cast<const f32x4*>(bias + 16 * ct + 4 * q);
                if (rb) v = v + *reinterpret_cast<const f32x4*>(rb + 16 * ct);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[rt][ct][e], rs, v[e]);
                *reinterpret_cast<f32x4*>(out + row * (16 * NCT) + 16 * ct + 4 * q) = v;
            }
        }
    }
};

// Mask 1x1 conv + activation, times the encoder output (recomputed from the waveform), through the decoder basis: rows are
// (b, l, spk), taps[b][spk][l][j] = sum_n Wd[n][j] * act(acc[n]) * relu(sum_i We[n][i] * wav[b][l*stride + i]).  In the transposed
// layout a lane owns 16 channels of ONE row per tile, so the frame's samples are loaded once per row, the sum over channels is
// lane-local down to the four q lanes (two shuffles per tap instead of four, one row per lane instead of sixteen) and the row's
// (b, l) takes one 32-bit division.  We / Wd sit in LDS as [n][KS] (zero-padded to KS >= ks taps, KS = 2 or KSMAX): the lanes of
// a q group read one address.
template <int RT, bool FAST, int KS>
struct EpTMaskDecode {
    const float* wav; const float* We; const float* Wd; float* taps; float* masked; int relu_act; Geo geo;
    float* mask_all = nullptr;  // training forward: also keeps the mask as [row][N]
    __device__ static float* tab() {
        __shared__ __attribute__((aligned(16))) float t[2 * 64 * KS];
        return t;
    }
    __device__ void begin(int tid) {
        float* t = tab();
        for (int i = tid; i < 64 * KS; i += 512) {
            const int n = i / KS, j = i % KS;
            t[i] = j < geo.ks ? We[n * geo.ks + j] : 0.f;
            t[64 * KS + i] = j < geo.ks ? Wd[n * geo.ks + j] : 0.f;
        }
    }
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][4], long long row0, int c, int q, long long M, float inv) {
        const int ks = geo.ks;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int qo = 4 * q * KS;
            asm volatile("" : "+v"(qo));                   // the table words are re-read per row, not kept in registers across tiles
            const float* t = tab() + qo;
            long long row = row0 + 16 * rt + c;
            const bool ok = row < M;                       // every lane takes part in the shuffles below
            if (!ok) row = M - 1;
            const unsigned ur = (unsigned)row, spk = ur & 1u, bl = ur >> 1;        // M < 2^31 (host check)
            const unsigned b = bl / (unsigned)geo.L, l = bl - b * (unsigned)geo.L;
            const float* x = wav + (long long)b * geo.T + (long long)l * geo.stride;
            float xv[KS], tp[KS];
#pragma unroll
            for (int j = 0; j < KS; ++j) { xv[j] = j < ks ? x[j] : 0.f; tp[j] = 0.f; }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 mev, mkv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* te = t + (16 * ct + e) * KS;           // channel n = 16 ct + 4 q + e
                    float en = 0.f;
#pragma unroll
                    for (int j = 0; j < KS; ++j) en = fmaf(te[j], xv[j], en);
                    en = relu_f(en);
                    const float a = acc[rt][ct][e] * inv;
                    const float mk = relu_act ? relu_f(a) : sigmoid_f<FAST>(a);
                    const float me = mk * en;
                    mev[e] = me; mkv[e] = mk;
#pragma unroll
                    for (int j = 0; j < KS; ++j) tp[j] = fmaf(te[64 * KS + j], me, tp[j]);
                }
                if (masked && ok && spk == 0) *reinterpret_cast<f32x4*>(masked + (long long)bl * geo.N + 16 * ct + 4 * q) = mev;
                if (mask_all && ok) *reinterpret_cast<f32x4*>(mask_all + row * geo.N + 16 * ct + 4 * q) = mkv;
            }
            float* dst = taps + (((long long)b * 2 + spk) * geo.L + l) * ks;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                float v = tp[j];
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (ok && q == (j & 3) && j < ks) dst[j] = v;          // the four lanes of a row share its taps' stores
            }
        }
    }
};

// TRACK: follow the largest |row element| for the saturation flag (common.h).  Off where the rows are bounded by construction:
// tanh * sigmoid products (mask + decode), gradients scaled from their own measured maximum (data gradients).
template <int KD, int ND, int RT, bool XC, class AL, class EPT, bool WT = false, bool TRACK = true>
__global__ __launch_bounds__(512, 1)
void rowgemm_f16t_kernel(const float* __restrict__ W0, int ldw0, int nd0, const float* __restrict__ W1, int ldw1,
                         long long M, AL al, EPT ep, const float* __restrict__ amax = nullptr, int* sat = nullptr,
                         unsigned* __restrict__ amax_dst = nullptr) {
    constexpr int NCT = ND / 16, NS = KD / 32, K8 = KD / 8, NW = 8;
    unsigned satm = 0u;                                                // bits of the largest |row element| converted to f16 pieces
    constexpr int PF = XC ? (NS >= 4 ? 3 : NS - 1) : (NS >= 4 ? 2 : 1);
    static_assert(!XC || NS % (PF + 1) == 0, "the slab buffers rotate across tiles");
    extern __shared__ __attribute__((aligned(16))) rg_u32x4 wl16[];     // [NS][NCT][2][64]
    __shared__ float red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;

    auto wrow = [&](int col) { return col < nd0 ? W0 + (long long)col * ldw0 : W1 + (long long)(col - nd0) * ldw1; };
    auto wt = [&](int k, int col) { return k < nd0 ? W0[(long long)k * ldw0 + col] : W1[(long long)(k - nd0) * ldw1 + col]; };   // WT
    float m = 0.f;
    if constexpr (WT) {
        for (int idx = tid; idx < ND * KD; idx += 64 * NW) m = fmaxf(m, fabsf(wt(idx / ND, idx % ND)));
    } else {
        for (int idx = tid; idx < ND * KD / 4; idx += 64 * NW) {
            const int col = idx / (KD / 4), k4 = idx % (KD / 4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow(col) + 4 * k4);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NW; ++i) m = fmaxf(m, red[i]);
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
    ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
    const float sw = ldexpf(1.f, 14 - ex);
    for (int idx = tid; idx < ND * K8; idx += 64 * NW) {               // slot (s, qq, i): k = 32 s + 4 qq + i (i < 4), 32 s + 16 + 4 qq + i - 4
        const int col = WT ? idx % ND : idx / K8, k8 = WT ? idx / ND : idx % K8;      // WT: consecutive threads read consecutive columns
        const int s = k8 >> 2, qq = k8 & 3;
        f32x4 lo, hi;
        if constexpr (WT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = wt(32 * s + 4 * qq + e, col); hi[e] = wt(32 * s + 16 + 4 * qq + e, col); }
        } else {
            const float* src = wrow(col) + 32 * s + 4 * qq;
            lo = *reinterpret_cast<const f32x4*>(src); hi = *reinterpret_cast<const f32x4*>(src + 16);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] *= sw; hi[e] *= sw; }
        rg_u32x4 w1, w2;
        f16_pieces8(lo, hi, w1, w2);
        const int ln = qq * 16 + (col & 15);
        wl16[((s * NCT + (col >> 4)) * 2 + 0) * 64 + ln] = w1;
        wl16[((s * NCT + (col >> 4)) * 2 + 1) * 64 + ln] = w2;
    }
    ep.begin(tid);
    __syncthreads();
    float xs = F16P_XS;
    if (amax && amax_dst) xs = act_scale(amax);        // forward rows: the adaptive activation scale of the launch's record word (common.h)
    else if (amax) {                                   // gradients: scaled from their own measured maximum
        const float am = *amax;
        int ea = 0;
        if (am > 0.f && am < 3.0e38f) frexpf(am, &ea);
        ea = ea < -100 ? -100 : ea > 100 ? 100 : ea;
        xs = ldexpf(1.f, 6 - ea);
    }
    const float inv = 1.f / (xs * sw);

    constexpr int ROWS_IT = NW * RT * 16;
    const long long nchunk = (M + ROWS_IT - 1) / ROWS_IT;
    const long long per = (nchunk + gridDim.x - 1) / gridDim.x;
    const long long c0 = per * blockIdx.x;
    long long c1 = c0 + per;
    if (c1 > nchunk) c1 = nchunk;

    f32x4 a[PF + 1][RT][2];
    AL aln = al;                                                       // XC: loader of the next tile
    if (c0 < c1) {
        const long long row0 = c0 * ROWS_IT + wave * RT * 16;
        al.init(row0 < M ? row0 : M - 1, c, q, M);
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { a[s][rt][0] = al.load(rt, 2 * s); a[s][rt][1] = al.load(rt, 2 * s + 1); }
    }
    for (long long ch = c0; ch < c1; ++ch) {
        const long long row0 = ch * ROWS_IT + wave * RT * 16;
        if constexpr (XC) {
            const long long rown = (ch + 1 < c1 ? ch + 1 : ch) * ROWS_IT + wave * RT * 16;
            aln.init(rown < M ? rown : M - 1, c, q, M);
        }
        f32x4 acc[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cb = s % (PF + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int sn = s + PF;
            if (sn < NS) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) { a[sn % (PF + 1)][rt][0] = al.load(rt, 2 * sn); a[sn % (PF + 1)][rt][1] = al.load(rt, 2 * sn + 1); }
            } else if constexpr (XC) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    a[sn % (PF + 1)][rt][0] = aln.load(rt, 2 * (sn - NS)); a[sn % (PF + 1)][rt][1] = aln.load(rt, 2 * (sn - NS) + 1);
                }
            }
            rg_u32x4 h1[RT], h2[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 lo = a[cb][rt][0], hi = a[cb][rt][1];
                if constexpr (TRACK) { sat_track(satm, lo); sat_track(satm, hi); }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __builtin_amdgcn_fmed3f(lo[e] * xs, -F16P_XMAX, F16P_XMAX);
                    hi[e] = __builtin_amdgcn_fmed3f(hi[e] * xs, -F16P_XMAX, F16P_XMAX);
                }
                f16_pieces8(lo, hi, h1[rt], h2[rt]);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if ((ct & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                const rg_u32x4 w1 = wl16[((s * NCT + ct) * 2 + 0) * 64 + lane], w2 = wl16[((s * NCT + ct) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][ct] = mfma_f16p(w1, h2[rt], acc[rt][ct]);
                    acc[rt][ct] = mfma_f16p(w2, h1[rt], acc[rt][ct]);
                    acc[rt][ct] = mfma_f16p(w1, h1[rt], acc[rt][ct]);
                }
            }
        }
        if (row0 < M) ep(acc, row0, c, q, M, inv);
        if constexpr (XC) al = aln;
        else if (ch + 1 < c1) {                                         // no lookahead across tiles: start the next tile's first slabs now
            const long long rown = (ch + 1) * ROWS_IT + wave * RT * 16;
            al.init(rown < M ? rown : M - 1, c, q, M);
#pragma unroll
            for (int s = 0; s < PF; ++s)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) { a[s][rt][0] = al.load(rt, 2 * s); a[s][rt][1] = al.load(rt, 2 * s + 1); }
        }
    }
    if constexpr (TRACK) sat_commit(sat, DPRNN_SAT_GEMM_ROWS, satm, F16P_XMAX / xs, amax_dst);
    ep.end(lane);
}

template <int KD, int ND, int RT, bool XC, bool WT = false, bool TRACK = true, class AL, class EPT>
int launch_rowgemm_f16t(const char* what, const float* W0, int ldw0, int nd0, const float* W1, int ldw1, long long M,
                        const AL& al, const EPT& ep, hipStream_t st, const float* amax = nullptr) {
    const size_t lds = (size_t)KD * ND * 4;
    auto kern = &rowgemm_f16t_kernel<KD, ND, RT, XC, AL, EPT, WT, TRACK>;
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    constexpr int ROWS_IT = 8 * RT * 16;
    const long long grid = rowgemm_f16_grid((M + ROWS_IT - 1) / ROWS_IT);
    unsigned* amax_dst = nullptr;
    if (TRACK && !amax) {                              // forward rows: scale from / report to the calling thread's record word (common.h)
        const ScaleRef sr = dprnn_scale_ref();
        amax = sr.src;
        amax_dst = sr.dst;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, W0, ldw0, nd0, W1, ldw1, M, al, ep, amax, dprnn_sat_ptr(), amax_dst);
    return dprnn_check_launch(what);
}

}  // namespace rg
