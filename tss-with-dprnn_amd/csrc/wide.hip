// Any-width DPRNN-TasNet forward for gfx950 (round 5): the kernels behind configurations WIDER than the 64 / 128 / 128 build
// (feature_size, hidden_size, input_size of any size; dprnn.py:117-120 takes any).  Same channels-last rows as the built path
// ([rows][C], C runtime), fp32 operands on the fp32 MFMA (v_mfma_f32_16x16x4_f32), norms materialised instead of folded into
// the next loader.  See include/dprnn_mi355x_wide.h for the contract of every entry point.
#include "common.h"
#include "dprnn_mi355x.h"
#include "dprnn_mi355x_wide.h"
#include <stdlib.h>

namespace {

inline unsigned grid_cap(long long n, int per_block, long long cap) {
    long long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---- E[b*L + l][n] = relu(sum_j We[n][j] wav[b][l*stride + j])      (encoder_decoder.py:25-33)
__global__ __launch_bounds__(256) void wide_encoder_kernel(const float* __restrict__ wav, const float* __restrict__ We,
                                                           float* __restrict__ E, int T, int L, int N, int ks, int stride,
                                                           long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % N); const long long row = i / N;
        const int l = (int)(row % L); const long long b = row / L;
        const float* x = wav + b * T + (long long)l * stride;
        float s = 0.f;
        for (int j = 0; j < ks; ++j) s = fmaf(We[n * ks + j], x[j], s);
        E[i] = relu_f(s);
    }
}

// ---- per-utterance (sum, sum of squares) of per_utt contiguous floats, fp64 (GroupNorm(1, N) / gLN of the encoder output)
__global__ __launch_bounds__(256) void wide_flat_stats_kernel(const float* __restrict__ A, double* __restrict__ stats, long long per_utt) {
    const int b = blockIdx.y;
    const float* p = A + (long long)b * per_utt;
    double d1 = 0.0, d2 = 0.0;
    float p1 = 0.f, p2 = 0.f;
    int n = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_utt; i += (long long)gridDim.x * 256) {
        const float v = p[i];
        p1 += v; p2 = fmaf(v, v, p2);
        if (++n == 64) { d1 += p1; d2 += p2; p1 = p2 = 0.f; n = 0; }
    }
    d1 += p1; d2 += p2;
    d1 = wave_sum(d1); d2 = wave_sum(d2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[2 * b], d1); atomicAdd(&stats[2 * b + 1], d2); }
}

// ---- (sc, sh)[b][c] of a global norm: y = x sc + sh = (x - mean) rstd gamma + beta     (dprnn.py:131-136, norms.py:27-31)
__global__ void wide_norm_prep_kernel(const double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float eps, float* __restrict__ sc, float* __restrict__ sh, int B, double count, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const double mean = stats[2 * b] / count;
    double var = stats[2 * b + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma[c] * rstd;
    sc[idx] = g;
    sh[idx] = beta[c] - g * (float)mean;
}

// ---- X[b][s][k][:] = Y[b][s*P + k - K][:] or 0     (dprnn.py:189-201)
__global__ __launch_bounds__(256) void wide_segment_kernel(const float* __restrict__ Y, float* __restrict__ X, int L, int K, int P, int S,
                                                           int C, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C); long long r = i / C;
        const int k = (int)(r % K); r /= K;
        const int s = (int)(r % S); const long long b = r / S;
        const int l = s * P + k - K;
        X[i] = (l >= 0 && l < L) ? Y[(b * L + l) * C + c] : 0.f;
    }
}

// ---- X += Z sc[u] + sh[u]     (dprnn.py:90-92, 98-99)
__global__ __launch_bounds__(256) void wide_norm_residual_kernel(float* __restrict__ X, const float* __restrict__ Z, const float* __restrict__ sc,
                                                                 const float* __restrict__ sh, long long total, int C, long long rows_per_utt,
                                                                 unsigned* __restrict__ amax) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C); const long long row = i / C;
        const long long u = row / rows_per_utt;
        const float v = X[i] + fmaf(Z[i], sc[u * C + c], sh[u * C + c]);
        X[i] = v;
        const float a = fabsf(v);
        if (a < 3.0e38f) m = fmaxf(m, a);
    }
    if (amax) {                                               // max |x| over the finite elements: the next recurrence's activation scale (wide_f16.hip)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax, __float_as_uint(m));
    }
}

// ---- frames[(b*nspk + spk)*L + l][c] = sum over chunks s with 0 <= l + K - s*P < K of U[(b*S + s)*K + k][spk*C + c]    (dprnn.py:203-217)
__global__ __launch_bounds__(256) void wide_ola_kernel(const float* __restrict__ U, float* __restrict__ frames, int L, int K, int P, int S,
                                                       int C, int nspk, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C); long long r = i / C;
        const int l = (int)(r % L); r /= L;
        const int spk = (int)(r % nspk); const long long b = r / nspk;
        const int pos = l + K;                                    // position in the padded axis: chunk s covers [s*P, s*P + K)
        int s_hi = pos / P; if (s_hi > S - 1) s_hi = S - 1;
        const int s_lo = l / P + 1;                                // s P + K > pos  <=>  s P > l
        float acc = 0.f;
        for (int s = s_lo; s <= s_hi; ++s) {
            const int k = pos - s * P;
            if (k >= 0 && k < K) acc += U[((b * S + s) * K + k) * ((long long)nspk * C) + (long long)spk * C + c];
        }
        frames[i] = acc;
    }
}

// ---- out[m][o] = act(sum_k a(m,k) W[o][k] + bias[o]) * mul[mrow(m)][o],  a(m,k) = prelu(A[m][k] sc[u][k] + sh[u][k]), u = m / rows_per_utt
// 64 x (64 NB) output tile per workgroup, four waves of 16 rows x 64 NB columns; k in blocks of 16 through LDS; the MFMA's k index inside
// a block is permuted (step s of lane group kk takes k = 4 kk + s) so that every lane reads its operands as one 16-byte LDS word.
// NB = 4 (256 columns per workgroup) for the wide outputs: a 64-column tile does 16 FLOPs per byte it stages and would re-read the A rows
// from HBM once per column tile (9.8 TB/s at the fp32 MFMA peak); at 256 columns the rows are read once and W (L2-resident) paces nothing.
constexpr int GP = 24;          // LDS pitch of a 16-float k block (floats): 16-byte aligned; pitch / 4 = 2 (mod 4) puts the 16 lanes of every
                                // ds_read_b128 lane group on 16 different 4-bank slots (pitch 20: one 2-way conflict per group)
template <int NB>
__global__ __launch_bounds__(256) void wide_gemm_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, long long rows_per_utt, const float* __restrict__ prelu,
                                                        const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                        const float* __restrict__ ubias, const float* __restrict__ rscale, int act,
                                                        const float* __restrict__ mul, long long ldm, long long mul_L, int mul_rep,
                                                        float* __restrict__ out, long long ldo, double* __restrict__ stats,
                                                        long long M, int K, int O, int vec) {
    __shared__ __attribute__((aligned(16))) float As[64 * GP];
    __shared__ __attribute__((aligned(16))) float Ws[64 * NB * GP];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 15, kk = lane >> 4;
    const long long m0 = (long long)blockIdx.x * 64;
    const int o0 = blockIdx.y * (64 * NB);
    const int lr = tid >> 2, lk = (tid & 3) * 4;                  // this thread stages (row lr [+ 64 i], k lk .. lk+3) of the tiles
    const long long am = m0 + lr;
    const bool a_ok = am < M;
    const long long au = a_ok ? am / rows_per_utt : 0;
    const float pw = prelu ? prelu[0] : 0.f;
    f32x4 acc[4 * NB];
#pragma unroll
    for (int t = 0; t < 4 * NB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the next k block's A and W words are requested right after this block's went to LDS, so their latency (HBM for A) runs under
    // this block's 16 NB MFMAs per wave instead of in front of them
    f32x4 av, wv[NB];
    auto stage = [&](int k0) {
        av = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kq = k0 + lk;
        if (vec) {                                                // K, lda, ldw multiples of 4 and 16-byte aligned bases: whole 16-byte words
            if (a_ok && kq < K) {
                av = *reinterpret_cast<const f32x4*>(A + am * lda + kq);
                if (sc) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + au * K + kq), h4 = *reinterpret_cast<const f32x4*>(sh + au * K + kq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = fmaf(av[e], s4[e], h4[e]);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int wo = o0 + lr + 64 * i;
                wv[i] = (wo < O && kq < K) ? *reinterpret_cast<const f32x4*>(W + (long long)wo * ldw + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) wv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kq + e;
                if (k < K) {
                    if (a_ok) {
                        const float v = A[am * lda + k];
                        av[e] = sc ? fmaf(v, sc[au * K + k], sh[au * K + k]) : v;
                    }
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const int wo = o0 + lr + 64 * i;
                        if (wo < O) wv[i][e] = W[(long long)wo * ldw + k];
                    }
                }
            }
        }
        if (prelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = av[e] >= 0.f ? av[e] : pw * av[e];
        }
    };
    stage(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();                                          // the previous block's reads are done
        *reinterpret_cast<f32x4*>(&As[lr * GP + lk]) = av;
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Ws[(lr + 64 * i) * GP + lk]) = wv[i];
        __syncthreads();
        if (k0 + 16 < K) stage(k0 + 16);
        const f32x4 a = *reinterpret_cast<const f32x4*>(&As[(16 * w + j) * GP + 4 * kk]);
#pragma unroll
        for (int t = 0; t < 4 * NB; ++t) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(&Ws[(16 * t + j) * GP + 4 * kk]);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[t] = mfma16(a[s], b[s], acc[t]);
        }
    }
    // epilogue: lane (j, kk), register r <-> row 16 w + 4 kk + r, column 16 t + j
    float rs1[4] = {0.f, 0.f, 0.f, 0.f}, rs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long m = m0 + 16 * w + 4 * kk + r;
        if (m >= M) continue;
        long long mrow = m;
        if (mul && mul_rep > 1) mrow = (m / (mul_L * mul_rep)) * mul_L + m % mul_L;
        const float rs = rscale ? rscale[m] : 1.f;
        const float* ub = ubias ? ubias + (m / rows_per_utt) * O : nullptr;
#pragma unroll
        for (int t = 0; t < 4 * NB; ++t) {
            const int o = o0 + 16 * t + j;
            if (o >= O) continue;
            float v = acc[t][r];
            if (rscale) v *= rs;
            v += (bias ? bias[o] : 0.f) + (ub ? ub[o] : 0.f);
            if (act == 1) v = relu_f(v);
            else if (act == 2) v = sigmoid_acc(v);
            else if (act == 3) v = tanh_acc(v);
            if (mul) v *= mul[mrow * ldm + o];
            out[m * ldo + o] = v;
            rs1[r] += v; rs2[r] = fmaf(v, v, rs2[r]);
        }
    }
    if (stats) {
        // per row: the 16 lanes that share kk hold its columns; rows of one wave's strip usually share an utterance
        const long long mf = m0 + 16 * w, ml = (mf + 15 < M ? mf + 15 : M - 1);
        if (mf < M) {
            if (mf / rows_per_utt == ml / rows_per_utt) {
                const double s1 = wave_sum((double)(rs1[0] + rs1[1] + rs1[2] + rs1[3]));
                const double s2 = wave_sum((double)(rs2[0] + rs2[1] + rs2[2] + rs2[3]));
                if (lane == 0) { const long long u = mf / rows_per_utt; atomicAdd(&stats[2 * u], s1); atomicAdd(&stats[2 * u + 1], s2); }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a1 = rs1[r], a2 = rs2[r];
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) { a1 += __shfl_xor(a1, d, 64); a2 += __shfl_xor(a2, d, 64); }
                    const long long m = mf + 4 * kk + r;
                    if (j == 0 && m < M) { const long long u = m / rows_per_utt; atomicAdd(&stats[2 * u], (double)a1); atomicAdd(&stats[2 * u + 1], (double)a2); }
                }
            }
        }
    }
}

// ---- recurrence weights in MFMA fragment order.  Virtual gates (G): LSTM i f g o (rows of [W_ih | W_hh], bias b_ih + b_hh);
// GRU r z n_x n_h with n_x = [W_in | 0] + b_in, n_h = [0 | W_hn] + b_hn (torch: n = tanh(n_x + r n_h)); Elman: one, [W_ih | W_hh].
// k axis: x part 0 .. Fp-1 (Fp = F rounded up to 16, zeros beyond F), h part Fp .. Fp + Hp - 1 (Hp = H rounded up to 16).
// Wp[((ut*nkb + kb)*G + g)*64 + lane][s] = Wv[g][ut*16 + (lane & 15)][kb*16 + 4 (lane >> 4) + s];  bv[g*Hp + unit].
__global__ __launch_bounds__(256) void wide_rnn_pack_kernel(int cell, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                            const float* __restrict__ b_ih, const float* __restrict__ b_hh, int F, int H,
                                                            float* __restrict__ Wp, float* __restrict__ bv, long long total) {
    const int Fp = (F + 15) & ~15, Hp = (H + 15) & ~15, nkb = (Fp + Hp) / 16, G = cell == 2 ? 1 : 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int s = (int)(i & 3); long long r = i >> 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int g = (int)(r % G); r /= G;
        const int kb = (int)(r % nkb); const int ut = (int)(r / nkb);
        const int unit = ut * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + s;
        float v = 0.f;
        if (unit < H) {
            const bool xpart = k < Fp;
            const int kc = xpart ? k : k - Fp;
            if (xpart ? kc < F : kc < H) {
                int gate = g; bool take = true;
                if (cell == 1) { gate = g < 2 ? g : 2; take = g < 2 || (g == 2) == xpart; }        // GRU: n_x takes x only, n_h takes h only
                if (take) v = xpart ? w_ih[((long long)gate * H + unit) * F + kc] : w_hh[((long long)gate * H + unit) * H + kc];
            }
        }
        Wp[i] = v;
        if (kb == 0 && s == 0 && (lane >> 4) == 0) {
            float b = 0.f;
            if (unit < H) {
                if (cell == 1) b = g < 2 ? b_ih[g * H + unit] + b_hh[g * H + unit] : (g == 2 ? b_ih[2 * H + unit] : b_hh[2 * H + unit]);
                else b = b_ih[g * H + unit] + b_hh[g * H + unit];
            }
            bv[g * Hp + unit] = b;
        }
    }
}

// ---- one direction of SingleRNN.forward (dprnn.py:33-37) over 16 NT sequences per workgroup, all T steps.
// LDS: two images [16 NT][x_t (Fp) | h_{t-1} (Hp) | pad 4]; step t reads image cur with the MFMA (A operand 16 bytes per lane),
// writes h_t into image nxt, x_{t+1} is copied into nxt at the start of the step; one barrier per step.  Wave w owns unit tiles
// w, w+4, ...: the four gates of one (sequence, unit) meet in one lane, c / h state stays in registers.  Weights stream from L2
// every step as 16-byte fragments (1 KB per wave instruction).
constexpr int MAXUT_WIDE = 8;   // unit tiles per wave: hidden_size <= 4 * 8 * 16 = 512; the 2-tile build (hidden_size <= 128) keeps the
                                // registers of a narrow model's recurrence low enough for four workgroups per CU
template <int NT, int G, bool FAST, int MAXUT>
__global__ __launch_bounds__(256) void wide_rnn_kernel(int cell, const float* __restrict__ X, const float* __restrict__ WpAll,
                                                       const float* __restrict__ bvAll, float* __restrict__ Hout, int F, int H, int nseq, int T,
                                                       long long inner_n, long long outer, long long inner, long long t_stride, int ldh,
                                                       long long wp_dir_stride, const long long* __restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Fp = (F + 15) & ~15, Hp = (H + 15) & ~15, Kp = Fp + Hp, pitch = Kp + 8, nkb = Kp / 16, nut = Hp / 16;    // pitch / 4 = 2 (mod 4): see GP
    constexpr int NS = 16 * NT;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 15, kk = lane >> 4;
    const int dir = blockIdx.y;
    const float* Wp = WpAll + (long long)dir * wp_dir_stride;
    const float* bv = bvAll + (long long)dir * G * Hp;
    const long long q0 = (long long)blockIdx.x * NS;
    float* img0 = lds;
    float* img1 = lds + NS * pitch;
    for (int i = tid; i < 2 * NS * pitch; i += 256) lds[i] = 0.f;
    __syncthreads();
    // row of (sequence q, time t): an intra-chunk sequence is K consecutive rows, an inter-chunk one strides over the chunks.
    // Tile descriptors (ragged batches: utterances of different lengths in one launch): {first row, steps, sequences, row step between
    // sequences} per workgroup instead of the regular formula -- every tile has its own step count.
    long long d_base = 0, d_inner = 0;
    int d_valid = 0;
    if (desc) {
        const long long* d = desc + 4 * (long long)blockIdx.x;
        d_base = d[0]; T = (int)d[1]; d_valid = (int)d[2]; d_inner = d[3];
    }
    auto seq_row0 = [&](int sr) -> long long {                  // first row of the tile's sequence sr, or -1 beyond the tile's last one
        if (desc) return sr < d_valid ? d_base + sr * d_inner : -1;
        const long long q = q0 + sr;
        return q < nseq ? (q / inner_n) * outer + (q % inner_n) * inner : -1;
    };
    auto load_x = [&](float* img, int t) {
        for (int sr = w; sr < NS; sr += 4) {
            const long long r0 = seq_row0(sr);
            if (r0 < 0) continue;
            const float* src = X + (r0 + (long long)t * t_stride) * F;
            for (int c = lane; c < F; c += 64) img[sr * pitch + c] = src[c];
        }
    };
    const int tfirst = dir ? T - 1 : 0, tstep = dir ? -1 : 1;
    load_x(img0, tfirst);
    float state[MAXUT][NT][4];
    float bq[MAXUT][G];
#pragma unroll
    for (int u = 0; u < MAXUT; ++u) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) state[u][nt][r] = 0.f;
        const int ut = w + 4 * u;
#pragma unroll
        for (int g = 0; g < G; ++g) bq[u][g] = ut < nut ? bv[g * Hp + ut * 16 + j] : 0.f;
    }
    __syncthreads();
    float* cur = img0;
    float* nxt = img1;
    for (int step = 0; step < T; ++step) {
        const int t = tfirst + step * tstep;
        if (step + 1 < T) load_x(nxt, t + tstep);
#pragma unroll
        for (int u = 0; u < MAXUT; ++u) {
            const int ut = w + 4 * u;
            if (ut < nut) {
                f32x4 acc[NT][G];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((long long)ut * nkb * G) * 64 + lane;
                const float* ap = cur + j * pitch + 4 * kk;
                // two k blocks in flight: the next block's weight fragments (L2) and A words (LDS) are requested before this block's MFMAs
                f32x4 a0[NT], b0[G], a1[NT], b1[G];
                auto fetch = [&](f32x4 (&a)[NT], f32x4 (&b)[G], int kb) {
#pragma unroll
                    for (int g = 0; g < G; ++g) b[g] = wp[((long long)kb * G + g) * 64];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const f32x4*>(ap + nt * 16 * pitch + kb * 16);
                };
                auto mma = [&](const f32x4 (&a)[NT], const f32x4 (&b)[G]) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)                    // consecutive MFMAs on different accumulators: NT x G between dependent ones
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][g] = mfma16(a[nt][s], b[g][s], acc[nt][g]);
                };
                fetch(a0, b0, 0);
                for (int kb = 0; kb < nkb; kb += 2) {
                    if (kb + 1 < nkb) fetch(a1, b1, kb + 1);
                    mma(a0, b0);
                    if (kb + 1 < nkb) {
                        if (kb + 2 < nkb) fetch(a0, b0, kb + 2);
                        mma(a1, b1);
                    }
                }
                const int unit = ut * 16 + j;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int sr = nt * 16 + 4 * kk + r;
                        float h;
                        if constexpr (G == 4) {
                            if (cell == 0) {                                       // LSTM (torch gate order i f g o)
                                const float ig = sigmoid_f<FAST>(acc[nt][0][r] + bq[u][0]), fg = sigmoid_f<FAST>(acc[nt][1][r] + bq[u][1]);
                                const float gg = tanh_f<FAST>(acc[nt][2][r] + bq[u][2]), og = sigmoid_f<FAST>(acc[nt][3][r] + bq[u][3]);
                                const float c = fmaf(fg, state[u][nt][r], ig * gg);
                                state[u][nt][r] = c;
                                h = og * tanh_f<FAST>(c);
                            } else {                                               // GRU (r z n)
                                const float rg = sigmoid_f<FAST>(acc[nt][0][r] + bq[u][0]), zg = sigmoid_f<FAST>(acc[nt][1][r] + bq[u][1]);
                                const float ng = tanh_f<FAST>(acc[nt][2][r] + bq[u][2] + rg * (acc[nt][3][r] + bq[u][3]));
                                h = (1.f - zg) * ng + zg * state[u][nt][r];
                                state[u][nt][r] = h;
                            }
                        } else {
                            h = tanh_f<FAST>(acc[nt][0][r] + bq[u][0]);            // Elman, tanh
                        }
                        nxt[sr * pitch + Fp + unit] = h;
                        const long long r0 = seq_row0(sr);
                        if (r0 >= 0 && unit < H) Hout[(r0 + (long long)t * t_stride) * ldh + (long long)dir * H + unit] = h;
                    }
            }
        }
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
}

}  // namespace

#define WIDE_NULL(cond, name) if (cond) { dprnn_set_error(name ": null pointer"); return -1; }

extern "C" int dprnn_wide_encoder_fwd(const float* wav, const float* enc_w, float* E, int B, int T, int L, int N, int ks, int stride,
                                      void* stream) {
    WIDE_NULL(!wav || !enc_w || !E, "dprnn_wide_encoder_fwd");
    if (B < 1 || N < 1 || ks < 1 || stride < 1 || T < ks || L != (T - ks) / stride + 1) {
        dprnn_set_error("dprnn_wide_encoder_fwd: bad geometry B=%d T=%d L=%d N=%d ks=%d stride=%d", B, T, L, N, ks, stride);
        return -1;
    }
    const long long total = (long long)B * L * N;
    hipLaunchKernelGGL(wide_encoder_kernel, dim3(grid_cap(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, wav, enc_w, E, T, L, N, ks,
                       stride, total);
    return dprnn_check_launch("wide_encoder_kernel");
}

extern "C" int dprnn_wide_flat_stats(const float* A, double* stats, int B, int64_t per_utt, void* stream) {
    WIDE_NULL(!A || !stats, "dprnn_wide_flat_stats");
    if (B < 1 || per_utt < 1) { dprnn_set_error("dprnn_wide_flat_stats: bad sizes"); return -1; }
    hipLaunchKernelGGL(wide_flat_stats_kernel, dim3(grid_cap(per_utt, 256 * 16, 256), B), dim3(256), 0, (hipStream_t)stream, A, stats,
                       (long long)per_utt);
    return dprnn_check_launch("wide_flat_stats_kernel");
}

extern "C" int dprnn_wide_norm_prep(const double* stats, const float* gamma, const float* beta, float eps, float* sc, float* sh, int B,
                                    int64_t rows_per_utt, int C, void* stream) {
    WIDE_NULL(!stats || !gamma || !beta || !sc || !sh, "dprnn_wide_norm_prep");
    if (B < 1 || rows_per_utt < 1 || C < 1) { dprnn_set_error("dprnn_wide_norm_prep: bad sizes"); return -1; }
    hipLaunchKernelGGL(wide_norm_prep_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, eps, sc, sh, B,
                       (double)rows_per_utt * (double)C, C);
    return dprnn_check_launch("wide_norm_prep_kernel");
}

extern "C" int dprnn_wide_segment_fwd(const float* Y, float* X, int B, int L, int K, int P, int S, int C, void* stream) {
    WIDE_NULL(!Y || !X, "dprnn_wide_segment_fwd");
    if (B < 1 || L < 1 || K < 1 || P < 1 || P > K || C < 1 || S != (L + K) / P + 1) {
        dprnn_set_error("dprnn_wide_segment_fwd: bad geometry L=%d K=%d P=%d S=%d C=%d", L, K, P, S, C);
        return -1;
    }
    const long long total = (long long)B * S * K * C;
    hipLaunchKernelGGL(wide_segment_kernel, dim3(grid_cap(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, Y, X, L, K, P, S, C, total);
    return dprnn_check_launch("wide_segment_kernel");
}

extern "C" int dprnn_wide_norm_residual_fwd(float* X, const float* Z, const float* sc, const float* sh, int64_t M, int C, int64_t rows_per_utt,
                                            void* amax_bits, void* stream) {
    WIDE_NULL(!X || !Z || !sc || !sh, "dprnn_wide_norm_residual_fwd");
    if (M < 1 || C < 1 || rows_per_utt < 1) { dprnn_set_error("dprnn_wide_norm_residual_fwd: bad sizes"); return -1; }
    const long long total = (long long)M * C;
    hipLaunchKernelGGL(wide_norm_residual_kernel, dim3(grid_cap(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, X, Z, sc, sh, total, C,
                       (long long)rows_per_utt, static_cast<unsigned*>(amax_bits));
    return dprnn_check_launch("wide_norm_residual_kernel");
}

extern "C" int dprnn_wide_ola_fwd(const float* U, float* frames, int B, int L, int K, int P, int S, int C, int nspk, void* stream) {
    WIDE_NULL(!U || !frames, "dprnn_wide_ola_fwd");
    if (B < 1 || L < 1 || K < 1 || P < 1 || P > K || C < 1 || nspk < 1 || S != (L + K) / P + 1) {
        dprnn_set_error("dprnn_wide_ola_fwd: bad geometry L=%d K=%d P=%d S=%d C=%d nspk=%d", L, K, P, S, C, nspk);
        return -1;
    }
    const long long total = (long long)B * nspk * L * C;
    hipLaunchKernelGGL(wide_ola_kernel, dim3(grid_cap(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, U, frames, L, K, P, S, C, nspk, total);
    return dprnn_check_launch("wide_ola_kernel");
}

extern "C" int dprnn_wide_gemm(const float* A, int64_t lda, const float* sc, const float* sh, int64_t rows_per_utt, const float* prelu,
                               const float* W, int ldw, const float* bias, const float* ubias, const float* rscale, int act, const float* mul,
                               int64_t ldm, int64_t mul_L, int mul_rep, float* out, int64_t ldo, double* stats, int64_t M, int K, int O,
                               void* stream) {
    WIDE_NULL(!A || !W || !out || ((sc == nullptr) != (sh == nullptr)), "dprnn_wide_gemm");
    if (M < 1 || K < 1 || O < 1 || lda < K || ldw < K || ldo < O || act < 0 || act > 3 || rows_per_utt < 1 || (mul && (ldm < O || mul_rep < 1 || mul_L < 1))) {
        dprnn_set_error("dprnn_wide_gemm: bad sizes M=%lld K=%d O=%d lda=%lld ldw=%d ldo=%lld act=%d", (long long)M, K, O, (long long)lda, ldw,
                        (long long)ldo, act);
        return -1;
    }
    const long long gm = (M + 63) / 64;
    if (gm > 0x7fffffffLL) { dprnn_set_error("dprnn_wide_gemm: M too large"); return -1; }
    const int vec = (K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && (((uintptr_t)A | (uintptr_t)W | (uintptr_t)sc | (uintptr_t)sh) & 15) == 0) ? 1 : 0;
    if (O > 128)
        hipLaunchKernelGGL(wide_gemm_kernel<4>, dim3((unsigned)gm, (O + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, (long long)lda, sc, sh,
                           (long long)rows_per_utt, prelu, W, ldw, bias, ubias, rscale, act, mul, (long long)ldm, (long long)mul_L, mul_rep, out, (long long)ldo,
                           stats, (long long)M, K, O, vec);
    else
        hipLaunchKernelGGL(wide_gemm_kernel<1>, dim3((unsigned)gm, (O + 63) / 64), dim3(256), 0, (hipStream_t)stream, A, (long long)lda, sc, sh,
                           (long long)rows_per_utt, prelu, W, ldw, bias, ubias, rscale, act, mul, (long long)ldm, (long long)mul_L, mul_rep, out, (long long)ldo,
                           stats, (long long)M, K, O, vec);
    return dprnn_check_launch("wide_gemm_kernel");
}

extern "C" int64_t dprnn_wide_rnn_pack_floats(int cell, int F, int H) {
    if (cell < 0 || cell > 2 || F < 1 || H < 1) return -1;
    const long long Fp = (F + 15) & ~15, Hp = (H + 15) & ~15;
    return (cell == 2 ? 1 : 4) * Hp * (Fp + Hp);
}

extern "C" int dprnn_wide_rnn_pack(int cell, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int F, int H,
                                   float* Wp, float* bv, void* stream) {
    WIDE_NULL(!w_ih || !w_hh || !b_ih || !b_hh || !Wp || !bv, "dprnn_wide_rnn_pack");
    const long long total = dprnn_wide_rnn_pack_floats(cell, F, H);
    if (total < 0) { dprnn_set_error("dprnn_wide_rnn_pack: cell %d (0 LSTM, 1 GRU, 2 RNN), F=%d, H=%d", cell, F, H); return -1; }
    hipLaunchKernelGGL(wide_rnn_pack_kernel, dim3(grid_cap(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, cell, w_ih, w_hh, b_ih, b_hh, F, H,
                       Wp, bv, total);
    return dprnn_check_launch("wide_rnn_pack_kernel");
}

static int wide_rnn_launch(int cell, const float* X, const float* Wp, const float* bv, float* Hout, int ndir, int F, int H, int nseq, int T,
                           int64_t inner_n, int64_t outer, int64_t inner, int64_t t_stride, const long long* desc, int ntiles, void* stream) {
    WIDE_NULL(!X || !Wp || !bv || !Hout, "dprnn_wide_rnn_fwd");
    if (cell < 0 || cell > 2 || ndir < 1 || ndir > 2 || F < 1 || H < 1 || nseq < 1 || T < 1 || inner_n < 1) {
        dprnn_set_error("dprnn_wide_rnn_fwd: bad sizes cell=%d ndir=%d F=%d H=%d nseq=%d T=%d", cell, ndir, F, H, nseq, T);
        return -1;
    }
    const int Fp = (F + 15) & ~15, Hp = (H + 15) & ~15, pitch = Fp + Hp + 8;
    if (Hp / 16 > 4 * MAXUT_WIDE) { dprnn_set_error("dprnn_wide_rnn_fwd: hidden_size %d > %d", H, 4 * MAXUT_WIDE * 16); return -1; }
    // 32 sequences per workgroup when they fit the LDS and still fill the chip, else 16
    // 16 sequences per workgroup: two workgroups per CU (two waves per SIMD) hide the weight-fragment and LDS latencies better than one
    // 32-sequence workgroup halves the weight stream (F = H = 256, 64 x 3 s: 0.52 / 0.60 of the MFMA peak against 0.41 / 0.51);
    // DPRNN_WIDE_NT=2 selects the 32-sequence form (diagnostics, tests)
    int nt = 1;
    if (const char* e = getenv("DPRNN_WIDE_NT")) { if (atoi(e) == 2) nt = 2; }
    if ((size_t)2 * 32 * pitch * sizeof(float) > 160 * 1024 || desc) nt = 1;
    const size_t lds = (size_t)2 * 16 * nt * pitch * sizeof(float);
    if (lds > 160 * 1024) { dprnn_set_error("dprnn_wide_rnn_fwd: feature_size %d + hidden_size %d need %zu bytes of LDS (160 KB per CU)", F, H, lds); return -1; }
    const long long wp_dir = dprnn_wide_rnn_pack_floats(cell, F, H);
    const dim3 grid(desc ? ntiles : (nseq + 16 * nt - 1) / (16 * nt), ndir), block(256);
    hipStream_t st = (hipStream_t)stream;
#define WIDE_RNN3(NT_, G_, FAST_, UT_)                                                                                                                 \
    do {                                                                                                                                    \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_rnn_kernel<NT_, G_, FAST_, UT_>), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                160 * 1024) != hipSuccess) { dprnn_set_error("dprnn_wide_rnn_fwd: hipFuncSetAttribute"); return -2; }        \
        hipLaunchKernelGGL((wide_rnn_kernel<NT_, G_, FAST_, UT_>), grid, block, lds, st, cell, X, Wp, bv, Hout, F, H, nseq, T, (long long)inner_n,      \
                           (long long)outer, (long long)inner, (long long)t_stride, ndir * H, wp_dir, desc);                                \
    } while (0)
#define WIDE_RNN2(NT_, G_, FAST_) do { if (Hp / 16 <= 8) WIDE_RNN3(NT_, G_, FAST_, 2); else WIDE_RNN3(NT_, G_, FAST_, MAXUT_WIDE); } while (0)
#define WIDE_RNN(NT_, G_) do { if (fast) WIDE_RNN2(NT_, G_, true); else WIDE_RNN2(NT_, G_, false); } while (0)
    const bool fast = dprnn_get_fast_activations() != 0;        // v_exp_f32 / v_rcp_f32 forms (about 1 ulp each), as the built kernels' default
    if (cell == 2) { if (nt == 2) WIDE_RNN(2, 1); else WIDE_RNN(1, 1); }
    else { if (nt == 2) WIDE_RNN(2, 4); else WIDE_RNN(1, 4); }
#undef WIDE_RNN
#undef WIDE_RNN2
#undef WIDE_RNN3
    return dprnn_check_launch("wide_rnn_kernel");
}

extern "C" int dprnn_wide_rnn_fwd(int cell, const float* X, const float* Wp, const float* bv, float* Hout, int ndir, int F, int H, int nseq, int T,
                                  int64_t inner_n, int64_t outer, int64_t inner, int64_t t_stride, void* stream) {
    return wide_rnn_launch(cell, X, Wp, bv, Hout, ndir, F, H, nseq, T, inner_n, outer, inner, t_stride, nullptr, 0, stream);
}

extern "C" int dprnn_wide_rnn_fwd_tiles(int cell, const float* X, const float* Wp, const float* bv, float* Hout, int ndir, int F, int H,
                                        const int64_t* desc, int ntiles, int64_t t_stride, void* stream) {
    if (!desc || ntiles < 1) { dprnn_set_error("dprnn_wide_rnn_fwd_tiles: no tiles"); return -1; }
    return wide_rnn_launch(cell, X, Wp, bv, Hout, ndir, F, H, ntiles, 1, 1, 0, 0, t_stride, reinterpret_cast<const long long*>(desc), ntiles, stream);
}
