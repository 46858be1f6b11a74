// Bandwidth-bound stages of the DPRNN path (gfx950): everything that is a pure gather / affine / reduction.
// All loads and stores are 16 bytes per lane over the channels-last rows (512 B per row).
#include "common.h"

namespace {

constexpr int F = 128;
constexpr int KSMAX = 8;

// ---- sum / sum-of-squares of enc[b,n,l] = relu(sum_j We[n][j] * wav[b][l*stride + j]) over (n, l)
// (encoder_decoder.py:25-33 feeding dprnn.py:134).  grid (chunks of 256 frames, B).
__global__ __launch_bounds__(256) void encoder_stats_kernel(const float* __restrict__ wav, const float* __restrict__ We,
                                                           double* __restrict__ stats, int T, int L, int N, int ks, int stride) {
    extern __shared__ float wsh[];                    // We [N][ks]
    for (int i = threadIdx.x; i < N * ks; i += 256) wsh[i] = We[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int l = blockIdx.x * 256 + threadIdx.x;
    float p1 = 0.f, p2 = 0.f;
    if (l < L) {
        const float* x = wav + (long long)b * T + (long long)l * stride;
        float xv[KSMAX];
#pragma unroll
        for (int j = 0; j < KSMAX; ++j) xv[j] = j < ks ? x[j] : 0.f;
        for (int n = 0; n < N; ++n) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KSMAX; ++j)
                if (j < ks) s = fmaf(wsh[n * ks + j], xv[j], s);
            s = relu_f(s);
            p1 += s; p2 = fmaf(s, s, p2);
        }
    }
    const double s1 = wave_sum((double)p1), s2 = wave_sum((double)p2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[2 * b], s1); atomicAdd(&stats[2 * b + 1], s2); }
}

__global__ void bottleneck_prep_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, const float* __restrict__ mul,
                                       const float* __restrict__ add, float* __restrict__ sc, float* __restrict__ sh,
                                       int B, int L, int N, int* __restrict__ sat) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int b = idx / N, c = idx % N;
    const double cnt = (double)L * (double)N;
    const double mean = stats[2 * b] / cnt;
    double var = stats[2 * b + 1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float m = mul ? mul[idx] : 1.f, a = add ? add[idx] : 0.f;
    const float g = gamma[c] * rstd;
    sc[idx] = g * m;
    sh[idx] = (beta[c] - g * (float)mean) * m + a;
    // statistics that are not finite numbers = a NaN / Inf sample in this utterance's input: every forward passes here first, so this
    // is where non-finite inputs raise their saturation flag (the f16-pieces conversions downstream track maxima, which skip NaN)
    if (sat && c == 0) {
        const double s0 = stats[2 * b], s1 = stats[2 * b + 1];
        if (!(fabs(s0) <= 1.7e308) || !(fabs(s1) <= 1.7e308)) atomicOr(sat + DPRNN_SAT_NONFINITE, 1);
    }
}

// ---- X[b][s][k][:] = Y[b][s*P + k - K][:] or 0 (dprnn.py:189-201): a chunk is K consecutive frames of one utterance, so a workgroup
// copies whole chunks (K x 512 B, contiguous on both sides, zero outside [0, L)) and nothing in the loop divides
__global__ __launch_bounds__(256) void segment_kernel(const f32x4* __restrict__ Y, f32x4* __restrict__ X,
                                                      int nchunk, int L, int K, int P, int S) {
    const int c4 = threadIdx.x & 31, kr = threadIdx.x >> 5;
    for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const int b = ch / S, s = ch - b * S;
        const int l0 = s * P - K;
        const f32x4* src = Y + (long long)b * L * 32 + c4;
        f32x4* dst = X + (long long)ch * K * 32 + c4;
        for (int k = kr; k < K; k += 8) {
            const int l = l0 + k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (l >= 0 && l < L) v = src[(long long)l * 32];
            dst[(long long)k * 32] = v;
        }
    }
}

// ---- Xout = X + ((Z - mean_b) * rstd_b * gamma + beta)   (dprnn.py:90-92, 98-99)
__global__ __launch_bounds__(256) void norm_residual_kernel(const f32x4* __restrict__ X, const f32x4* __restrict__ Z,
                                                            const double* __restrict__ stats, const f32x4* __restrict__ gamma,
                                                            const f32x4* __restrict__ beta, float eps, f32x4* __restrict__ Xout,
                                                            long long total4, long long rows_per_utt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long row = i >> 5; const int c4 = (int)(i & 31);
        const long long b = row / rows_per_utt;
        const double cnt = (double)rows_per_utt * (double)F;
        const double mean = stats[2 * b] / cnt;
        double var = stats[2 * b + 1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float mu = (float)mean;
        const f32x4 x = X[i], z = Z[i], g = gamma[c4], be = beta[c4];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = x[e] + ((z[e] - mu) * rstd * g[e] + be[e]);
        Xout[i] = o;
    }
}

// ---- out[b][spk][t] = sum_{l*stride + j = t} taps[b][spk][l][j]   (ConvTranspose1d, encoder_decoder.py:40-42)
__global__ __launch_bounds__(256) void decoder_kernel(const float* __restrict__ taps, float* __restrict__ out,
                                                      int B, int nspk, int T, int L, int ks, int stride) {
    const long long total = (long long)B * nspk * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = (int)(i % T); const long long bs = i / T;
        const int spk = (int)(bs % nspk); const long long b = bs / nspk;
        const float* tp = taps + ((b * 2 + spk) * (long long)L) * ks;
        float s = 0.f;
        for (int j = 0; j < ks; ++j) {
            const int d = t - j;
            if (d >= 0 && d % stride == 0) {
                const int l = d / stride;
                if (l < L) s += tp[(long long)l * ks + j];
            }
        }
        out[i] = s;
    }
}

inline unsigned grid_for(long long n, int per_block = 256, int cap = 256 * 8) {
    long long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int dprnn_encoder_stats(const float* wav, const float* enc_w, double* stats,
                                   int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !stats) { dprnn_set_error("dprnn_encoder_stats: null pointer"); return -1; }
    if (B < 1 || ks < 1 || ks > KSMAX || stride < 1 || L != (T - ks) / stride + 1 || L < 1) {
        dprnn_set_error("dprnn_encoder_stats: bad geometry B=%d T=%d L=%d ks=%d stride=%d", B, T, L, ks, stride);
        return -1;
    }
    hipLaunchKernelGGL(encoder_stats_kernel, dim3((L + 255) / 256, B), dim3(256), (size_t)N * ks * sizeof(float),
                       (hipStream_t)stream, wav, enc_w, stats, T, L, N, ks, stride);
    return dprnn_check_launch("encoder_stats_kernel");
}

extern "C" int dprnn_bottleneck_prep(const double* stats, const float* gamma, const float* beta, float eps,
                                     const float* mul, const float* add, float* sc, float* sh,
                                     int B, int L, int N, void* stream) {
    if (!stats || !gamma || !beta || !sc || !sh) { dprnn_set_error("dprnn_bottleneck_prep: null pointer"); return -1; }
    hipLaunchKernelGGL(bottleneck_prep_kernel, dim3((B * N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       stats, gamma, beta, eps, mul, add, sc, sh, B, L, N, dprnn_sat_ptr());
    return dprnn_check_launch("bottleneck_prep_kernel");
}

extern "C" int dprnn_segment_fwd(const float* Y, float* X, int B, int L, int K, int P, int S, void* stream) {
    if (!Y || !X) { dprnn_set_error("dprnn_segment_fwd: null pointer"); return -1; }
    if (K < 1 || P < 1 || S != (L + K) / P + 1) { dprnn_set_error("dprnn_segment_fwd: bad geometry L=%d K=%d P=%d S=%d", L, K, P, S); return -1; }
    const long long nchunk = (long long)B * S;
    if (B < 1 || nchunk > 0x7fffffffLL) { dprnn_set_error("dprnn_segment_fwd: B=%d S=%d out of range", B, S); return -1; }
    const long long grid = nchunk < 16 * 256 ? nchunk : 16 * 256;
    hipLaunchKernelGGL(segment_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(Y), reinterpret_cast<f32x4*>(X), (int)nchunk, L, K, P, S);
    return dprnn_check_launch("segment_kernel");
}

extern "C" int dprnn_norm_residual_fwd(const float* X, const float* Z, const double* stats, const float* gamma,
                                       const float* beta, float eps, float* Xout, int B, int64_t rows_per_utt, void* stream) {
    if (!X || !Z || !stats || !gamma || !beta || !Xout) { dprnn_set_error("dprnn_norm_residual_fwd: null pointer"); return -1; }
    const long long total4 = (long long)B * rows_per_utt * 32;
    hipLaunchKernelGGL(norm_residual_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(X), reinterpret_cast<const f32x4*>(Z), stats,
                       reinterpret_cast<const f32x4*>(gamma), reinterpret_cast<const f32x4*>(beta), eps,
                       reinterpret_cast<f32x4*>(Xout), total4, (long long)rows_per_utt);
    return dprnn_check_launch("norm_residual_kernel");
}

extern "C" int dprnn_decoder_fwd(const float* taps, float* out, int B, int nspk, int T, int L, int ks, int stride, void* stream) {
    if (!taps || !out) { dprnn_set_error("dprnn_decoder_fwd: null pointer"); return -1; }
    if (nspk < 1 || nspk > 2 || T != (L - 1) * stride + ks) { dprnn_set_error("dprnn_decoder_fwd: bad geometry nspk=%d T=%d L=%d ks=%d stride=%d", nspk, T, L, ks, stride); return -1; }
    hipLaunchKernelGGL(decoder_kernel, dim3(grid_for((long long)B * nspk * T)), dim3(256), 0, (hipStream_t)stream,
                       taps, out, B, nspk, T, L, ks, stride);
    return dprnn_check_launch("decoder_kernel");
}

// ---- pairwise SI-SDR (the metric/loss core of asteroid's pairwise_neg_sisdr that the reference calls at
// src/trainers/trainer.py:39,104 and src/inferencers/inferencer.py:30,60), fp64 accumulation:
// out[b][i][j] = 10 log10( |s|^2 / (|e - s|^2 + eps) + eps ),  e = est[b][i] - mean, t = tgt[b][j] - mean,
// s = <e,t> t / (|t|^2 + eps), eps = 1e-8.  One workgroup per (b, i, j).
namespace {
// variant 0: the loss form above.  variant 1: the evaluation metric asteroid.metrics.get_metrics(['si_sdr']) reports
// (src/inferencers/inferencer.py:64; it delegates to pb_bss_eval.evaluation.si_sdr): no mean removal, no eps --
// 10 log10(|a t|^2 / |e - a t|^2), a = <e,t>/|t|^2.
__global__ __launch_bounds__(256) void pairwise_sisdr_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                             double* __restrict__ out, int n_est, int n_tgt, int T, int variant) {
    const int b = blockIdx.z, i = blockIdx.y, j = blockIdx.x;
    const float* e = est + ((long long)b * n_est + i) * T;
    const float* t = tgt + ((long long)b * n_tgt + j) * T;
    double se = 0, st = 0, see = 0, stt = 0, set = 0;
    for (int k = threadIdx.x; k < T; k += 256) {
        const double x = e[k], y = t[k];
        se += x; st += y; see += x * x; stt += y * y; set += x * y;
    }
    __shared__ double red[5][4];
    se = wave_sum(se); st = wave_sum(st); see = wave_sum(see); stt = wave_sum(stt); set = wave_sum(set);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = se; red[1][w] = st; red[2][w] = see; red[3][w] = stt; red[4][w] = set; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double v[5];
        for (int k = 0; k < 5; ++k) v[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        const double n = (double)T, eps = variant == 0 ? 1e-8 : 0.0;
        if (variant != 0) { v[0] = 0.0; v[1] = 0.0; }
        const double ee = v[2] - v[0] * v[0] / n, tt = v[3] - v[1] * v[1] / n, et = v[4] - v[0] * v[1] / n;   // zero-mean moments
        const double alpha = et / (tt + eps);
        const double s2 = alpha * alpha * tt;                       // |s|^2
        const double n2 = ee - 2.0 * alpha * et + alpha * alpha * tt;   // |e - s|^2
        out[((long long)b * n_est + i) * n_tgt + j] = 10.0 * log10(s2 / (n2 + eps) + eps);
    }
}
}  // namespace

static int pairwise_sisdr(const float* est, const float* tgt, double* out, int B, int n_est, int n_tgt, int T, int variant,
                          void* stream) {
    if (!est || !tgt || !out) { dprnn_set_error("dprnn_pairwise_sisdr: null pointer"); return -1; }
    if (B < 1 || n_est < 1 || n_tgt < 1 || T < 2 || variant < 0 || variant > 1) { dprnn_set_error("dprnn_pairwise_sisdr: bad sizes"); return -1; }
    hipLaunchKernelGGL(pairwise_sisdr_kernel, dim3(n_tgt, n_est, B), dim3(256), 0, (hipStream_t)stream, est, tgt, out, n_est, n_tgt, T, variant);
    return dprnn_check_launch("pairwise_sisdr_kernel");
}

extern "C" int dprnn_pairwise_sisdr(const float* est, const float* tgt, double* out, int B, int n_est, int n_tgt, int T,
                                    void* stream) {
    return pairwise_sisdr(est, tgt, out, B, n_est, n_tgt, T, 0, stream);
}

extern "C" int dprnn_pairwise_sisdr_metric(const float* est, const float* tgt, double* out, int B, int n_est, int n_tgt, int T,
                                           void* stream) {
    return pairwise_sisdr(est, tgt, out, B, n_est, n_tgt, T, 1, stream);
}
