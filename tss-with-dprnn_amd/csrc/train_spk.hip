// Backward of the DPRNN-Spe speaker branch and fusions (training path; reference src/models/dprnn_spe.py:8-42 ResBlock,
// :115-123 spk_encoder, :156-163 _auxiliary, :165-229 fusions) as autograd forms it for trainer_spe.py:44-45
// (`epoch_loss = l + ce_gamma * ce_loss; epoch_loss.backward()`).  The 1x1 convs reuse the generic data / weight
// gradient GEMMs (train_gemm.hip); this file holds the streaming pieces: MaxPool1d(3) + PReLU + residual split,
// BatchNorm1d backward (batch statistics in train mode, running statistics in eval mode), the PReLU between the convs,
// the time mean, the small per-utterance Linears and the per-channel fusion operands.  Small reductions go to fp64
// scratch with fp64 atomics (dprnn_scratch_to_grad converts).
#include "rowgemm_core.h"
using namespace rg;

namespace {

inline int grid_rows(long long items, int per_block) {
    long long g = (items + per_block - 1) / per_block;
    const long long cap = (long long)num_cus() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

__device__ __forceinline__ float aux_frames(float len, int ks) {      // dprnn_spe.py:159
    const float a = floorf((len - (float)ks) / (float)(ks / 2)) + 1.f;
    return floorf(floorf(floorf(a / 3.f) / 3.f) / 3.f);
}

// dE[b][t][c] = d_emb[b][c] / aux_T_b                                   (emb = sum_t E / aux_T, dprnn_spe.py:160-162)
__global__ __launch_bounds__(256)
void time_mean_bwd_kernel(const float* __restrict__ d_emb, const float* __restrict__ aux_len, int n_len,
                          float* __restrict__ dE, int T3, int C, int ks, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long b = i / ((long long)T3 * C);
        dE[i] = d_emb[b * C + c] / aux_frames(aux_len[n_len > 1 ? b : 0], ks);
    }
}

// MaxPool1d(3) + PReLU + residual add backward (dprnn_spe.py:36-42): u = Y2*scale + shift + res, v = prelu(u),
// out[t3] = max_j v[3*t3+j] (first maximum wins, as ATen's max_pool1d);  dU[3*t3+j] = dOut[t3]*prelu'(u) at the argmax,
// 0 elsewhere (and on the up to two tail rows the pooling drops); dalpha += dOut*u where u <= 0 at the argmax.
__global__ __launch_bounds__(256)
void resblock_pool_bwd_kernel(const f32x4* __restrict__ dOut, const f32x4* __restrict__ Y, const f32x4* __restrict__ scale,
                              const f32x4* __restrict__ shift, const f32x4* __restrict__ mean, const f32x4* __restrict__ res,
                              const float* __restrict__ prelu_w,
                              f32x4* __restrict__ dU, double* __restrict__ dalpha, long long total4, int T_in, int T_out, int C4) {
    const float alpha = *prelu_w;
    float part = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4); const long long bt = i / C4;
        const int t3 = (int)(bt % T_out); const long long b = bt / T_out;
        const f32x4 s = scale[c4], t = shift[c4], mu = mean[c4], g = dOut[i];
        f32x4 u[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const long long src = (b * T_in + 3 * t3 + j) * C4 + c4;
            const f32x4 y = Y[src], rr = res[src];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[j][e] = fmaf(y[e] - mu[e], s[e], t[e]) + rr[e];
        }
        f32x4 o[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int am = 0;
            float best = u[0][e] >= 0.f ? u[0][e] : alpha * u[0][e];
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                const float v = u[j][e] >= 0.f ? u[j][e] : alpha * u[j][e];
                if (v > best) { best = v; am = j; }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float d = 0.f;
                if (j == am) {
                    d = u[j][e] > 0.f ? g[e] : alpha * g[e];
                    if (!(u[j][e] > 0.f)) part = fmaf(g[e], u[j][e], part);
                }
                o[j][e] = d;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) dU[(b * T_in + 3 * t3 + j) * C4 + c4] = o[j];
        if (t3 == T_out - 1)
            for (int tt = 3 * T_out; tt < T_in; ++tt) dU[(b * T_in + tt) * C4 + c4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const double sum = wave_sum((double)part);
    if ((threadIdx.x & 63) == 0) atomicAdd(dalpha, sum);
}

// BatchNorm1d moments used by the backward: batch statistics (col_stats = per-channel (sum, sumsq) of the forward) in
// train mode, running statistics in eval mode; rstd = 1/sqrt(var + eps)
__global__ void bn_moments_kernel(const double* __restrict__ col_stats, const float* __restrict__ running_mean,
                                  const float* __restrict__ running_var, long long count, float eps,
                                  float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m, v;
    if (col_stats) {
        const double mm = col_stats[2 * c] / (double)count;
        double vv = col_stats[2 * c + 1] / (double)count - mm * mm;
        if (vv < 0.0) vv = 0.0;
        m = (float)mm; v = (float)vv;
    } else {
        m = running_mean[c]; v = running_var[c];
    }
    mean[c] = m;
    rstd[c] = 1.f / sqrtf(v + eps);
}

// red[0][c] += sum_rows dY (dbeta), red[1][c] += sum_rows dY * xh (dgamma), xh = (Y - mean)*rstd; C = 128 or 256
template <int C>
__global__ __launch_bounds__(256)
void bn_bwd_stats_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ mean,
                         const float* __restrict__ rstd, double* __restrict__ red, long long M, long long rows_per_block) {
    constexpr int C4 = C / 4, RL = 256 / C4;
    const int c4 = threadIdx.x % C4, rr = threadIdx.x / C4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + 4 * c4), rs = *reinterpret_cast<const f32x4*>(rstd + 4 * c4);
    const long long lo = (long long)blockIdx.x * rows_per_block;
    long long hi = lo + rows_per_block;
    if (hi > M) hi = M;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (long long row = lo + rr; row < hi; row += RL) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dY + row * C + 4 * c4);
        const f32x4 y = *reinterpret_cast<const f32x4*>(Y + row * C + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += d[e]; s2[e] = fmaf(d[e], (y[e] - mu[e]) * rs[e], s2[e]); }
    }
    __shared__ float sm[RL][2][C];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[rr][0][4 * c4 + e] = s1[e]; sm[rr][1][4 * c4 + e] = s2[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < RL; ++k) s += (double)sm[k][which][c];
        atomicAdd(&red[which * C + c], s);
    }
}

// train: dX = gamma*rstd*(dY - red0/M - xh*red1/M);  eval (running statistics are constants): dX = dY*gamma*rstd
__global__ __launch_bounds__(256)
void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ mean,
                         const float* __restrict__ rstd, const float* __restrict__ gamma, const double* __restrict__ red,
                         int train, float* __restrict__ dX, long long M, int C) {
    const long long total4 = M * (C / 4);
    const int C4 = C / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const f32x4 d = reinterpret_cast<const f32x4*>(dY)[i], y = reinterpret_cast<const f32x4*>(Y)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * c4 + e;
            const float gr = gamma[c] * rstd[c];
            if (train) {
                const float xh = (y[e] - mean[c]) * rstd[c];
                o[e] = gr * (d[e] - (float)(red[c] / (double)M) - xh * (float)(red[C + c] / (double)M));
            } else {
                o[e] = gr * d[e];
            }
        }
        reinterpret_cast<f32x4*>(dX)[i] = o;
    }
}

// out = prelu(Y*scale + shift): the input of the ResBlock's second conv (dprnn_spe.py:33-35), Q operand of its weight gradient
__global__ __launch_bounds__(256)
void affine_prelu_rows_kernel(const float* __restrict__ Y, const float* __restrict__ scale, const float* __restrict__ shift,
                              const float* __restrict__ mean, const float* __restrict__ prelu_w, float* __restrict__ out,
                              long long M, int C) {
    const float alpha = *prelu_w;
    const int C4 = C / 4;
    const long long total4 = M * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const f32x4 y = reinterpret_cast<const f32x4*>(Y)[i];
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + 4 * c4), t = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + 4 * c4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float b = fmaf(y[e] - mu[e], s[e], t[e]); o[e] = b >= 0.f ? b : alpha * b; }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// dB = dA * prelu'(b), b = Y*scale + shift; dalpha += dA*b where b <= 0
__global__ __launch_bounds__(256)
void affine_prelu_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ Y, const float* __restrict__ scale,
                             const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ prelu_w,
                             float* __restrict__ dB,
                             double* __restrict__ dalpha, long long M, int C) {
    const float alpha = *prelu_w;
    const int C4 = C / 4;
    const long long total4 = M * C4;
    float part = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const f32x4 y = reinterpret_cast<const f32x4*>(Y)[i], g = reinterpret_cast<const f32x4*>(dA)[i];
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + 4 * c4), t = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + 4 * c4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = fmaf(y[e] - mu[e], s[e], t[e]);
            o[e] = b > 0.f ? g[e] : alpha * g[e];
            if (!(b > 0.f)) part = fmaf(g[e], b, part);
        }
        reinterpret_cast<f32x4*>(dB)[i] = o;
    }
    const double sum = wave_sum((double)part);
    if ((threadIdx.x & 63) == 0) atomicAdd(dalpha, sum);
}

__global__ __launch_bounds__(256)
void add_rows_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 x = a[i], y = b[i];
        out[i] = f32x4{x[0] + y[0], x[1] + y[1], x[2] + y[2], x[3] + y[3]};
    }
}

// out[b][c] += sum_l A[(b,l)][c]   (fp64 scratch): gradient of a per-utterance bias row ('cat' fusion, dprnn_spe.py:186-192)
__global__ __launch_bounds__(256)
void utt_colsum_kernel(const float* __restrict__ A, double* __restrict__ out, int L, int C, int chunks) {
    const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int per = (L + chunks - 1) / chunks;
    const int l0 = per * ch, l1 = l0 + per < L ? l0 + per : L;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int l = l0; l < l1; ++l) s += A[((long long)b * L + l) * C + c];
        atomicAdd(&out[(long long)b * C + c], (double)s);
    }
}

// small per-utterance Linear out[b][o] = sum_k W[o][k0+k]*in[b][k] + bias[o] (dprnn_spe.py:90,123,186-212): gradients
__global__ void linear_small_bwd_in_kernel(const float* __restrict__ d_out, int do_pitch, const float* __restrict__ W, int ldw,
                                           int k0, float* __restrict__ d_in, int di_pitch, int B, int O, int K, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * K) return;
    const int b = idx / K, k = idx % K;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(d_out[(long long)b * do_pitch + o], W[(long long)o * ldw + k0 + k], s);
    float* d = d_in + (long long)b * di_pitch + k;
    *d = accumulate ? *d + s : s;
}
__global__ void linear_small_bwd_w_kernel(const float* __restrict__ d_out, int do_pitch, const float* __restrict__ in, int in_pitch,
                                          float* __restrict__ dW, int ldw, int k0, float* __restrict__ db, int B, int O, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < O * K) {
        const int o = idx / K, k = idx % K;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(d_out[(long long)b * do_pitch + o], in[(long long)b * in_pitch + k], s);
        dW[(long long)o * ldw + k0 + k] = s;
    }
    if (db && idx < O) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += d_out[(long long)b * do_pitch + idx];
        db[idx] = s;
    }
}

// per-channel fusion operands (dprnn_spe.py:194-229): a' = an*mul + add with an = enc*sc0 + sh0 the plain norm output.
// d_an = da'*mul;  dmul[b][c] += sum_l da'*an;  dadd[b][c] += sum_l da'   (fp64 scratch; mul == null: d_an = da')
// 'att' fusion (dprnn_spe.py:178-183,214-222; mul = a = fusion_linear(emb)): the attention scores
// scores[b][t] = sum_c a[b][c]*avg[b][c][t], avg = depth-wise conv (avg_w [N][ks_a], avg_b, stride ks_a) of an, add
// d_an[b][l][c] += dS[b][t]*a[b][c]*avg_w[c][j] and dmul[b][c] += dS[b][t]*(avg_w[c][j]*an + [j == 0]*avg_b[c]),
// l = t*ks_a + j, for the score gradient dS (dprnn_att_softmax_bwd).
__global__ __launch_bounds__(256)
void fusion_bwd_kernel(const float* __restrict__ wav, const float* __restrict__ We, Geo geo, const float* __restrict__ sc0,
                       const float* __restrict__ sh0, const float* __restrict__ dAp, const float* __restrict__ mul,
                       float* __restrict__ dAn, double* __restrict__ dmul, double* __restrict__ dadd, int chunks,
                       const float* __restrict__ dS, const float* __restrict__ avg_w, const float* __restrict__ avg_b,
                       int ks_a, int La) {
    const int n = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int ks = geo.ks;
    const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int per = (geo.L + chunks - 1) / chunks;
    const int l0 = per * ch, l1 = (l0 + per < geo.L) ? l0 + per : geo.L;
    float we[KSMAX];
#pragma unroll
    for (int j = 0; j < KSMAX; ++j) we[j] = j < ks ? We[n * ks + j] : 0.f;
    const float s0 = sc0[b * geo.N + n], t0 = sh0[b * geo.N + n];
    const float m = mul ? mul[b * geo.N + n] : 1.f;
    float am = 0.f, aa = 0.f;
    for (int l = l0 + slot; l < l1; l += 4) {
        const float* x = wav + (long long)b * geo.T + (long long)l * geo.stride;
        float en = 0.f;
#pragma unroll
        for (int j = 0; j < KSMAX; ++j)
            if (j < ks) en = fmaf(we[j], x[j], en);
        en = relu_f(en);
        const long long row = (long long)b * geo.L + l;
        const float g = dAp[row * 64 + n];
        const float an = fmaf(en, s0, t0);
        am = fmaf(g, an, am);
        aa += g;
        float dan = g * m;
        if (dS) {
            const int t = l / ks_a, j = l - t * ks_a;
            if (t < La) {
                const float ds = dS[(long long)b * La + t], w = avg_w[n * ks_a + j];
                dan = fmaf(ds * m, w, dan);
                am = fmaf(ds, fmaf(w, an, j == 0 ? avg_b[n] : 0.f), am);
            }
        }
        dAn[row * 64 + n] = dan;
    }
    __shared__ float red[4][2][64];
    red[slot][0][n] = am; red[slot][1][n] = aa;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6;
        const double s = (double)red[0][which][n] + (double)red[1][which][n] + (double)red[2][which][n] + (double)red[3][which][n];
        double* dst = which == 0 ? dmul : dadd;
        if (dst) atomicAdd(&dst[b * geo.N + n], s);
    }
}

}  // namespace

extern "C" int dprnn_time_mean_bwd(const float* d_emb, const float* aux_len, int n_len, float* dE, int B, int T3, int C, int ks,
                                   void* stream) {
    if (!d_emb || !aux_len || !dE || B < 1 || T3 < 1 || C < 1 || ks < 2 || (n_len != 1 && n_len != B)) {
        dprnn_set_error("dprnn_time_mean_bwd: null pointer / sizes"); return -1;
    }
    const long long total = (long long)B * T3 * C;
    hipLaunchKernelGGL(time_mean_bwd_kernel, dim3(grid_rows(total, 256)), dim3(256), 0, (hipStream_t)stream, d_emb, aux_len, n_len,
                       dE, T3, C, ks, total);
    return dprnn_check_launch("time_mean_bwd_kernel");
}

extern "C" int dprnn_resblock_pool_bwd(const float* dOut, const float* Y2, const float* scale, const float* shift, const float* mean,
                                       const float* res, const float* prelu_w, float* dU, double* dalpha, int B, int T_in, int C,
                                       void* stream) {
    if (!dOut || !Y2 || !scale || !shift || !mean || !res || !prelu_w || !dU || !dalpha) { dprnn_set_error("dprnn_resblock_pool_bwd: null pointer"); return -1; }
    if (C % 4 || T_in < 3) { dprnn_set_error("dprnn_resblock_pool_bwd: bad sizes T_in=%d C=%d", T_in, C); return -1; }
    const int T_out = T_in / 3;
    const long long total4 = (long long)B * T_out * (C / 4);
    hipLaunchKernelGGL(resblock_pool_bwd_kernel, dim3(grid_rows(total4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(dOut), reinterpret_cast<const f32x4*>(Y2), reinterpret_cast<const f32x4*>(scale),
                       reinterpret_cast<const f32x4*>(shift), reinterpret_cast<const f32x4*>(mean), reinterpret_cast<const f32x4*>(res),
                       prelu_w, reinterpret_cast<f32x4*>(dU), dalpha, total4, T_in, T_out, C / 4);
    return dprnn_check_launch("resblock_pool_bwd_kernel");
}

extern "C" int dprnn_bn_moments(const double* col_stats, const float* running_mean, const float* running_var, int64_t count,
                                float eps, float* mean, float* rstd, int C, void* stream) {
    if ((!col_stats && (!running_mean || !running_var)) || !mean || !rstd || C < 1 || count < 1) {
        dprnn_set_error("dprnn_bn_moments: null pointer / sizes"); return -1;
    }
    hipLaunchKernelGGL(bn_moments_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, col_stats, running_mean,
                       running_var, (long long)count, eps, mean, rstd, C);
    return dprnn_check_launch("bn_moments_kernel");
}

extern "C" int dprnn_bn_bwd(const float* dY, const float* Y, const float* mean, const float* rstd, const float* gamma,
                            int train, double* red, float* dX, int64_t M, int C, void* stream) {
    if (!dY || !Y || !mean || !rstd || !gamma || !red || !dX || M < 1) { dprnn_set_error("dprnn_bn_bwd: null pointer / rows"); return -1; }
    if (C != 128 && C != 256) { dprnn_set_error("dprnn_bn_bwd: C must be 128 or 256 (got %d)", C); return -1; }
    long long blocks = (M + 255) / 256;
    const long long cap = 4LL * num_cus();
    if (blocks > cap) blocks = cap;
    const long long rpb = (M + blocks - 1) / blocks;
    blocks = (M + rpb - 1) / rpb;
    if (C == 128)
        hipLaunchKernelGGL(bn_bwd_stats_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dY, Y, mean, rstd, red, (long long)M, rpb);
    else
        hipLaunchKernelGGL(bn_bwd_stats_kernel<256>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dY, Y, mean, rstd, red, (long long)M, rpb);
    int rc = dprnn_check_launch("bn_bwd_stats_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_rows((long long)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       dY, Y, mean, rstd, gamma, red, train, dX, (long long)M, C);
    return dprnn_check_launch("bn_bwd_apply_kernel");
}

extern "C" int dprnn_affine_prelu_rows(const float* Y, const float* scale, const float* shift, const float* mean, const float* prelu_w,
                                       float* out, int64_t M, int C, void* stream) {
    if (!Y || !scale || !shift || !mean || !prelu_w || !out || M < 1 || (C % 4)) { dprnn_set_error("dprnn_affine_prelu_rows: null pointer / sizes"); return -1; }
    hipLaunchKernelGGL(affine_prelu_rows_kernel, dim3(grid_rows((long long)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       Y, scale, shift, mean, prelu_w, out, (long long)M, C);
    return dprnn_check_launch("affine_prelu_rows_kernel");
}

extern "C" int dprnn_affine_prelu_bwd(const float* dA, const float* Y, const float* scale, const float* shift, const float* mean,
                                      const float* prelu_w, float* dB, double* dalpha, int64_t M, int C, void* stream) {
    if (!dA || !Y || !scale || !shift || !mean || !prelu_w || !dB || !dalpha || M < 1 || (C % 4)) { dprnn_set_error("dprnn_affine_prelu_bwd: null pointer / sizes"); return -1; }
    hipLaunchKernelGGL(affine_prelu_bwd_kernel, dim3(grid_rows((long long)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       dA, Y, scale, shift, mean, prelu_w, dB, dalpha, (long long)M, C);
    return dprnn_check_launch("affine_prelu_bwd_kernel");
}

extern "C" int dprnn_add_rows(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 1 || (n % 4)) { dprnn_set_error("dprnn_add_rows: null pointer / n"); return -1; }
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_rows(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(b), reinterpret_cast<f32x4*>(out), (long long)(n / 4));
    return dprnn_check_launch("add_rows_kernel");
}

extern "C" int dprnn_utt_colsum(const float* A, double* out, int B, int L, int C, void* stream) {
    if (!A || !out || B < 1 || L < 1 || C < 1) { dprnn_set_error("dprnn_utt_colsum: null pointer / sizes"); return -1; }
    int chunks = (2 * num_cus() + B - 1) / B;
    const int maxc = (L + 63) / 64;
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(utt_colsum_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, A, out, L, C, chunks);
    return dprnn_check_launch("utt_colsum_kernel");
}

extern "C" int dprnn_linear_small_bwd(const float* d_out, int do_pitch, const float* in, int in_pitch, const float* W, int ldw,
                                      int k0, int K, float* d_in, int di_pitch, int accumulate_in, float* dW, float* db,
                                      int B, int O, void* stream) {
    if (!d_out || !in || !W || B < 1 || O < 1 || K < 1) { dprnn_set_error("dprnn_linear_small_bwd: null pointer / sizes"); return -1; }
    if (d_in) {
        hipLaunchKernelGGL(linear_small_bwd_in_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           d_out, do_pitch, W, ldw, k0, d_in, di_pitch, B, O, K, accumulate_in);
        int rc = dprnn_check_launch("linear_small_bwd_in_kernel");
        if (rc) return rc;
    }
    if (dW) {
        const int n = O * K;
        hipLaunchKernelGGL(linear_small_bwd_w_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           d_out, do_pitch, in, in_pitch, dW, ldw, k0, db, B, O, K);
        return dprnn_check_launch("linear_small_bwd_w_kernel");
    }
    return 0;
}

extern "C" int dprnn_fusion_bwd(const float* wav, const float* enc_w, const float* sc0, const float* sh0, const float* dAp,
                                const float* mul, float* dAn, double* dmul, double* dadd,
                                const float* d_scores, const float* avg_w, const float* avg_b, int ks_a,
                                int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !sc0 || !sh0 || !dAp || !dAn) { dprnn_set_error("dprnn_fusion_bwd: null pointer"); return -1; }
    if (d_scores && (!mul || !avg_w || !avg_b || ks_a < 1 || L < ks_a)) { dprnn_set_error("dprnn_fusion_bwd: incomplete 'att' operands"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || L != (T - ks) / stride + 1) { dprnn_set_error("dprnn_fusion_bwd: unsupported N=%d ks=%d", N, ks); return -1; }
    const Geo geo{B, T, L, N, ks, stride};
    int chunks = (4 * num_cus() + B - 1) / B;
    const int maxc = (L + 63) / 64;
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    const int La = d_scores ? (L - ks_a) / ks_a + 1 : 0;
    hipLaunchKernelGGL(fusion_bwd_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, wav, enc_w, geo, sc0, sh0, dAp, mul,
                       dAn, dmul, dadd, chunks, d_scores, avg_w, avg_b, ks_a, La);
    return dprnn_check_launch("fusion_bwd_kernel");
}

// ---- 'att' fusion, frame factor: Y = rs[b][l]*G + bias with G = W a' (dprnn_bottleneck_att_fwd).  Given dY:
// dG = dY*rs, d_rs[(b,l)] = sum_f dY*(Y - bias)/rs (rs = softmax + 1 >= 1), dbias[f] += sum dY (fp64 scratch)
namespace {
__global__ __launch_bounds__(256)
void att_scale_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ bias,
                          const float* __restrict__ rs, float* __restrict__ dG, float* __restrict__ d_rs,
                          double* __restrict__ dbias, long long rows) {
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * c4);
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    for (long long row = (long long)blockIdx.x * 8 + rr; row < rows; row += (long long)gridDim.x * 8) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dY + row * 128 + 4 * c4);
        const f32x4 y = *reinterpret_cast<const f32x4*>(Y + row * 128 + 4 * c4);
        const float r = rs[row], ir = 1.f / r;
        float dot = 0.f;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dot = fmaf(d[e], (y[e] - bv[e]) * ir, dot); o[e] = d[e] * r; bs[e] += d[e]; }
        *reinterpret_cast<f32x4*>(dG + row * 128 + 4 * c4) = o;
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) dot += __shfl_xor(dot, m, 64);       // the 32 lanes of a row are one half-wave
        if (c4 == 0) d_rs[row] = dot;
    }
    __shared__ float red[8][128];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rr][4 * c4 + e] = bs[e];
    __syncthreads();
    if (threadIdx.x < 128) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)red[k][threadIdx.x];
        atomicAdd(&dbias[threadIdx.x], s);
    }
}

// softmax over time + nearest upsampling backward (dprnn_att_row_scale): att = softmax(scores),
// d_att[t] = sum over the frames l that read att[t] of d_rs[l], dS = att*(d_att - sum_t att*d_att); one workgroup per utterance
__global__ __launch_bounds__(256)
void att_softmax_bwd_kernel(const float* __restrict__ d_rs, const float* __restrict__ scores, float* __restrict__ dS, int La, int L) {
    const int b = blockIdx.x;
    const float* s = scores + (long long)b * La;
    const float* g = d_rs + (long long)b * L;
    float* o = dS + (long long)b * La;
    __shared__ float red[4];
    __shared__ float stat[3];
    float m = -INFINITY;
    for (int t = threadIdx.x; t < La; t += 256) m = fmaxf(m, s[t]);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) stat[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = stat[0];
    float z = 0.f;
    for (int t = threadIdx.x; t < La; t += 256) z += expf(s[t] - m);
    z = wave_sum(z);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
    __syncthreads();
    if (threadIdx.x == 0) stat[1] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const float inv = 1.f / stat[1];
    const float scale = (float)La / (float)L;
    float dot = 0.f;
    for (int t = threadIdx.x; t < La; t += 256) {
        // frames l with min(floor(l*scale), La-1) == t: search a window around t/scale (same float arithmetic as the forward)
        int lo = (int)floorf((float)t / scale) - 2, hi = (int)ceilf((float)(t + 1) / scale) + 2;
        if (lo < 0) lo = 0;
        if (t == La - 1 || hi > L) hi = L;
        float da = 0.f;
        for (int l = lo; l < hi; ++l) {
            int src = (int)floorf((float)l * scale);
            if (src > La - 1) src = La - 1;
            if (src == t) da += g[l];
        }
        o[t] = da;
        dot = fmaf(expf(s[t] - m) * inv, da, dot);
    }
    dot = wave_sum(dot);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) stat[2] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    dot = stat[2];
    for (int t = threadIdx.x; t < La; t += 256) o[t] = expf(s[t] - m) * inv * (o[t] - dot);
}
}  // namespace

extern "C" int dprnn_att_scale_bwd(const float* dY, const float* Y, const float* bias, const float* row_scale, float* dG,
                                   float* d_rs, double* dbias, int64_t rows, void* stream) {
    if (!dY || !Y || !bias || !row_scale || !dG || !d_rs || !dbias || rows < 1) { dprnn_set_error("dprnn_att_scale_bwd: null pointer / rows"); return -1; }
    hipLaunchKernelGGL(att_scale_bwd_kernel, dim3(grid_rows((rows + 7) / 8, 1)), dim3(256), 0, (hipStream_t)stream,
                       dY, Y, bias, row_scale, dG, d_rs, dbias, (long long)rows);
    return dprnn_check_launch("att_scale_bwd_kernel");
}

extern "C" int dprnn_att_softmax_bwd(const float* d_rs, const float* scores, float* d_scores, int B, int L, int ks_a, void* stream) {
    if (!d_rs || !scores || !d_scores || B < 1 || ks_a < 1 || L < ks_a) { dprnn_set_error("dprnn_att_softmax_bwd: null pointer / sizes"); return -1; }
    const int La = (L - ks_a) / ks_a + 1;
    hipLaunchKernelGGL(att_softmax_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, d_rs, scores, d_scores, La, L);
    return dprnn_check_launch("att_softmax_bwd_kernel");
}
