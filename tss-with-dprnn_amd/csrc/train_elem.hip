// HBM-bound pieces of the training backward pass: GroupNorm / PReLU / gate / mask /
// encoder / decoder gradients and the SI-SDR PIT loss.  Every kernel is a streaming pass over rows of 64 or 128
// channels (16-byte accesses, one row segment per 16 or 32 lanes); reductions to small outputs (per-utterance sums,
// per-channel parameter gradients) are accumulated in fp64 and added with fp64 atomics to a scratch buffer that
// dprnn_scratch_to_grad converts to the fp32 .grad layout.
#include "rowgemm_core.h"
using namespace rg;

namespace {

__device__ __forceinline__ void mean_rstd(const double* stats, int b, double cnt, float eps, float& mu, float& rs) {
    const double mean = stats[2 * b] / cnt;
    double var = stats[2 * b + 1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    mu = (float)mean;
    rs = (float)(1.0 / sqrt(var + (double)eps));
}

// ------------------------------------------------------------------------------------------------ slab GroupNorm backward
// y = gamma*xh + beta, xh = (Z - mu_b)*rs_b over an utterance's rows_per_utt*128 elements (dprnn.py:90,98; norms.py:6-31).
// Pass 1: dy = dXn (+ dXa + dXb, written to dXo when given); red[b] += (sum gamma*dy, sum gamma*dy*xh);
// dgb[0][c] += sum dy*xh (dgamma), dgb[1][c] += sum dy (dbeta).  Z == null: only the sum dXo = dXn + dXa + dXb.
__global__ __launch_bounds__(256)
void gn_bwd_stats_kernel(const float* __restrict__ dXn, const float* __restrict__ dXa, const float* __restrict__ dXb,
                         float* __restrict__ dXo, const float* __restrict__ Z, const double* __restrict__ stats,
                         const float* __restrict__ gamma, float eps, long long rows_per_utt, int chunks_per_utt,
                         double* __restrict__ red, double* __restrict__ dgb, unsigned* __restrict__ absmax) {
    const int b = blockIdx.x / chunks_per_utt, ch = blockIdx.x % chunks_per_utt;
    const long long per = (rows_per_utt + chunks_per_utt - 1) / chunks_per_utt;
    const long long lo = (long long)b * rows_per_utt + per * ch;
    long long hi = lo + per;
    if (hi > (long long)(b + 1) * rows_per_utt) hi = (long long)(b + 1) * rows_per_utt;
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    float mu = 0.f, rs = 0.f;
    f32x4 gm = {0.f, 0.f, 0.f, 0.f};
    if (Z) {
        mean_rstd(stats, b, (double)rows_per_utt * 128.0, eps, mu, rs);
        gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
    }
    f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    float mgd = 0.f, mxh = 0.f;              // max |gamma dy| and max |x_hat| of this thread's elements (absmax)
    double S1 = 0.0, S2 = 0.0;
    int n = 0;
    for (long long row = lo + rr; row < hi; row += 8) {
        f32x4 dy = *reinterpret_cast<const f32x4*>(dXn + row * 128 + 4 * c4);
        if (dXa) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(dXa + row * 128 + 4 * c4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(dXb + row * 128 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dy[e] = dy[e] + a[e] + c[e];
            *reinterpret_cast<f32x4*>(dXo + row * 128 + 4 * c4) = dy;
        }
        if (Z) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(Z + row * 128 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (z[e] - mu) * rs;
                dg[e] = fmaf(dy[e], xh, dg[e]);
                db[e] += dy[e];
                const float gd = gm[e] * dy[e];
                s1 += gd;
                s2 = fmaf(gd, xh, s2);
                mgd = fmaxf(mgd, fabsf(gd));
                mxh = fmaxf(mxh, fabsf(xh));
            }
            if (++n == 64) { S1 += (double)s1; S2 += (double)s2; s1 = 0.f; s2 = 0.f; n = 0; }
        }
    }
    if (!Z) return;
    if (absmax) {                            // per-utterance maxima as float bits: what bounds |dZ| before dZ exists (dprnn_lin_bwd_f16)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mgd = fmaxf(mgd, __shfl_xor(mgd, o, 64)); mxh = fmaxf(mxh, __shfl_xor(mxh, o, 64)); }
        if ((threadIdx.x & 63) == 0) {
            if (mgd > 0.f) atomicMax(absmax + 2 * b, __float_as_uint(mgd));
            if (mxh > 0.f) atomicMax(absmax + 2 * b + 1, __float_as_uint(mxh));
        }
    }
    S1 += (double)s1; S2 += (double)s2;
    __shared__ double sred[2][4];
    __shared__ float cred[8][2][128];
    S1 = wave_sum(S1); S2 = wave_sum(S2);
    if ((threadIdx.x & 63) == 0) { sred[0][threadIdx.x >> 6] = S1; sred[1][threadIdx.x >> 6] = S2; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { cred[rr][0][4 * c4 + e] = dg[e]; cred[rr][1][4 * c4 + e] = db[e]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&red[2 * b], sred[0][0] + sred[0][1] + sred[0][2] + sred[0][3]);
        atomicAdd(&red[2 * b + 1], sred[1][0] + sred[1][1] + sred[1][2] + sred[1][3]);
    }
    {
        const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)cred[k][which][c];
        atomicAdd(&dgb[which * 128 + c], s);
    }
}

// Pass 2: dZ = rs_b * (gamma*dy - s1_b/n - xh*s2_b/n)
__global__ __launch_bounds__(256)
void gn_bwd_dz_kernel(const float* __restrict__ dY, const float* __restrict__ Z, const double* __restrict__ stats,
                      const double* __restrict__ red, const float* __restrict__ gamma, float eps, long long rows_per_utt,
                      long long rows, float* __restrict__ dZ, unsigned* __restrict__ amax) {
    const int c4 = threadIdx.x & 31;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
    float mx = 0.f;
    for (long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); row < rows; row += (long long)gridDim.x * 8) {
        const int b = (int)(row / rows_per_utt);
        const double cnt = (double)rows_per_utt * 128.0;
        float mu, rs;
        mean_rstd(stats, b, cnt, eps, mu, rs);
        const float m1 = (float)(red[2 * b] / cnt), m2 = (float)(red[2 * b + 1] / cnt);
        const f32x4 dy = *reinterpret_cast<const f32x4*>(dY + row * 128 + 4 * c4);
        const f32x4 z = *reinterpret_cast<const f32x4*>(Z + row * 128 + 4 * c4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (z[e] - mu) * rs;
            o[e] = rs * (gm[e] * dy[e] - m1 - xh * m2);
        }
        *reinterpret_cast<f32x4*>(dZ + row * 128 + 4 * c4) = o;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (amax) {                                          // max |dZ| of the launch as float bits (dprnn_gn_bwd_dz_store_absmax)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(amax, __float_as_uint(mx));
    }
}

// ------------------------------------------------------------------------------------------------ mask head
// dAG[row][c] = dV*sG*(1 - tA^2), dAG[row][128+c] = dV*tA*sG*(1 - sG);  TG = [tanh(out conv) | sigmoid(gate conv)]
// (dprnn.py:157-158,180)
__global__ __launch_bounds__(256)
void gate_bwd_kernel(const float* __restrict__ dV, const float* __restrict__ TG, float* __restrict__ dAG, long long rows) {
    const int c4 = threadIdx.x & 31;
    for (long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); row < rows; row += (long long)gridDim.x * 8) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dV + row * 128 + 4 * c4);
        const f32x4 ta = *reinterpret_cast<const f32x4*>(TG + row * 256 + 4 * c4);
        const f32x4 sg = *reinterpret_cast<const f32x4*>(TG + row * 256 + 128 + 4 * c4);
        f32x4 a, g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = dv[e] * sg[e] * (1.f - ta[e] * ta[e]);
            g[e] = dv[e] * ta[e] * sg[e] * (1.f - sg[e]);
        }
        *reinterpret_cast<f32x4*>(dAG + row * 256 + 4 * c4) = a;
        *reinterpret_cast<f32x4*>(dAG + row * 256 + 128 + 4 * c4) = g;
    }
}

// decoder + masking + mask activation backward (encoder_decoder.py:40-42, dprnn.py:277, :161-164):
//   dm[b,spk,l,n] = sum_j Wd[n][j]*dOut[b][spk][l*stride+j];  enc = relu(sum_j We[n][j]*wav[b][l*stride+j])
//   dE[(b,l,spk)][n] = dm*enc*act'(mask);  dEnc[(b,l)][n] = sum_spk dm*mask;  dWd[n][j] += mask*enc*dOut[..]
__global__ __launch_bounds__(256)
void mask_decode_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ wav, const float* __restrict__ We,
                            const float* __restrict__ Wd, const float* __restrict__ Mk, int relu_act, Geo geo, int T_out,
                            float* __restrict__ dE, float* __restrict__ dEnc, double* __restrict__ dWd,
                            const float* __restrict__ dM0) {
    const int n = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int ks = geo.ks;
    float we[KSMAX], wd[KSMAX], acc[KSMAX];
#pragma unroll
    for (int j = 0; j < KSMAX; ++j) {
        we[j] = j < ks ? We[n * ks + j] : 0.f;
        wd[j] = j < ks ? Wd[n * ks + j] : 0.f;
        acc[j] = 0.f;
    }
    const long long rows = (long long)geo.B * geo.L;
    for (long long row = (long long)blockIdx.x * 4 + slot; row < rows; row += (long long)gridDim.x * 4) {
        const int b = (int)(row / geo.L), l = (int)(row % geo.L);
        const float* x = wav + (long long)b * geo.T + (long long)l * geo.stride;
        float en = 0.f;
#pragma unroll
        for (int j = 0; j < KSMAX; ++j)
            if (j < ks) en = fmaf(we[j], x[j], en);
        en = relu_f(en);
        float denc = 0.f;
#pragma unroll
        for (int spk = 0; spk < 2; ++spk) {
            const float* g = dOut + ((long long)b * 2 + spk) * T_out + (long long)l * geo.stride;
            const float m = Mk[(row * 2 + spk) * 64 + n];
            float dm = 0.f;
#pragma unroll
            for (int j = 0; j < KSMAX; ++j)
                if (j < ks) { dm = fmaf(wd[j], g[j], dm); acc[j] = fmaf(m * en, g[j], acc[j]); }
            if (dM0 && spk == 0) dm += dM0[row * 64 + n];        // encoder-domain gradient of d0 = mask_0*enc (DPRNN-Spe-IRA)
            const float dact = relu_act ? (m > 0.f ? 1.f : 0.f) : m * (1.f - m);
            dE[(row * 2 + spk) * 64 + n] = dm * en * dact;
            denc = fmaf(dm, m, denc);
        }
        dEnc[row * 64 + n] = denc;
    }
    __shared__ float red[4][64][KSMAX];
#pragma unroll
    for (int j = 0; j < KSMAX; ++j) red[slot][n][j] = acc[j];
    __syncthreads();
    if (slot == 0) {
#pragma unroll
        for (int j = 0; j < KSMAX; ++j)
            if (j < ks) atomicAdd(&dWd[n * ks + j], (double)red[0][n][j] + (double)red[1][n][j] + (double)red[2][n][j] + (double)red[3][n][j]);
    }
}

// PqSum[(b,l)][c] = sum over chunks covering frame l of prelu(X[b][s][k][c])   (dprnn.py:173,203-217) -- the Q
// operand of the conv2d weight gradient; dbias[c] += cnt(l)*dU[(b,l)][c] for the 256 conv2d outputs
__global__ __launch_bounds__(256)
void ola_prelu_sum_kernel(const float* __restrict__ X, const float* __restrict__ alpha_p, int B, int L, int K, int P, int S,
                          float* __restrict__ out, const float* __restrict__ dU, double* __restrict__ dbias) {
    const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;
    const float alpha = *alpha_p;
    const long long rows = (long long)B * L;
    f32x4 bs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (long long row = (long long)blockIdx.x * 8 + rr; row < rows; row += (long long)gridDim.x * 8) {
        const int b = (int)(row / L), l = (int)(row % L);
        const int lo = OlaGather<1>::s_lo(l, P), hi = OlaGather<1>::s_hi(l, K, P, S);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = lo; s <= hi; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(X + (((long long)b * S + s) * K + (l + K - s * P)) * 128 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e] >= 0.f ? v[e] : alpha * v[e];
        }
        *reinterpret_cast<f32x4*>(out + row * 128 + 4 * c4) = acc;
        const float cnt = (float)(hi - lo + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dU + row * 256 + 128 * h + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[h][e] = fmaf(cnt, g[e], bs[h][e]);
        }
    }
    __shared__ float red[8][256];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[rr][128 * h + 4 * c4 + e] = bs[h][e];
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (double)red[k][threadIdx.x];
    atomicAdd(&dbias[threadIdx.x], s);
}

// gradient of DPRNN._segmentation (dprnn.py:189-201): dY[(b,l)] = sum over chunks covering l of (dXn + dXa + dXb)[b][s][k]
__global__ __launch_bounds__(256)
void segment_bwd_kernel(const float* __restrict__ dXn, const float* __restrict__ dXa, const float* __restrict__ dXb,
                        int B, int L, int K, int P, int S, float* __restrict__ dY) {
    const int c4 = threadIdx.x & 31;
    const long long rows = (long long)B * L;
    for (long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); row < rows; row += (long long)gridDim.x * 8) {
        const int b = (int)(row / L), l = (int)(row % L);
        const int lo = OlaGather<1>::s_lo(l, P), hi = OlaGather<1>::s_hi(l, K, P, S);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = lo; s <= hi; ++s) {
            const long long o = (((long long)b * S + s) * K + (l + K - s * P)) * 128 + 4 * c4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(dXn + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
            if (dXa) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(dXa + o);
                const f32x4 c = *reinterpret_cast<const f32x4*>(dXb + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += a[e] + c[e];
            }
        }
        *reinterpret_cast<f32x4*>(dY + row * 128 + 4 * c4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------ encoder / bottleneck norm
// mode 0: A[(b,l)][n] = enc*sc[b][n] + sh[b][n]                         (the normalised encoder output, Q of the bottleneck wgrad)
// mode 1: dy = dA[(b,l)][n]: red[b] += (sum g*dy, sum g*dy*xh), dgb[0][n] += dy*xh, dgb[1][n] += dy        (norm backward, pass 1)
// mode 2: denc = rs*(g*dy - s1/n - xh*s2/n) + dEnc;  dpre = denc*[enc > 0];  dWe[n][j] += dpre*wav[b][l*stride+j]   (pass 2)
// MAT: the rows are a materialised [B*L][64] tensor Xmat instead of the encoder output recomputed from the waveform
// (GroupNorm of d0 = mask_0*enc in DPRNN-Spe-IRA, dprnn_spe_ira.py:84 -> dprnn_spe.py:116): no ReLU, and mode 2 writes
// the gradient of the rows to dXout instead of accumulating the encoder weight gradient.
template <int MODE, bool MAT = false>
__global__ __launch_bounds__(256)
void encoder_rows_kernel(const float* __restrict__ wav, const float* __restrict__ We, Geo geo,
                         const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ A,
                         const float* __restrict__ dA, const double* __restrict__ stats, const float* __restrict__ gamma,
                         float eps, double* __restrict__ red, double* __restrict__ dgb,
                         const float* __restrict__ dEnc, double* __restrict__ dWe, int chunks_per_utt,
                         const float* __restrict__ Xmat = nullptr, float* __restrict__ dXout = nullptr) {
    const int n = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int ks = geo.ks;
    const int b = blockIdx.x / chunks_per_utt, ch = blockIdx.x % chunks_per_utt;
    const int per = (geo.L + chunks_per_utt - 1) / chunks_per_utt;
    const int l0 = per * ch, l1 = (l0 + per < geo.L) ? l0 + per : geo.L;
    float we[KSMAX], acc[KSMAX];
#pragma unroll
    for (int j = 0; j < KSMAX; ++j) { we[j] = (!MAT && j < ks) ? We[n * ks + j] : 0.f; acc[j] = 0.f; }
    float mu = 0.f, rs = 0.f, g = 0.f, m1 = 0.f, m2 = 0.f, scn = 0.f, shn = 0.f;
    const double cnt = (double)geo.L * (double)geo.N;
    if (MODE == 0) { scn = sc[b * geo.N + n]; shn = sh[b * geo.N + n]; }
    if (MODE >= 1) { mean_rstd(stats, b, cnt, eps, mu, rs); g = gamma[n]; }
    if (MODE == 2) { m1 = (float)(red[2 * b] / cnt); m2 = (float)(red[2 * b + 1] / cnt); }
    float dgam = 0.f, dbet = 0.f, s1 = 0.f, s2 = 0.f;
    for (int l = l0 + slot; l < l1; l += 4) {
        const long long row = (long long)b * geo.L + l;
        const float* x = MAT ? nullptr : wav + (long long)b * geo.T + (long long)l * geo.stride;
        float en = 0.f;
        if constexpr (MAT) {
            en = Xmat[row * 64 + n];
        } else {
#pragma unroll
            for (int j = 0; j < KSMAX; ++j)
                if (j < ks) en = fmaf(we[j], x[j], en);
            en = relu_f(en);
        }
        if (MODE == 0) {
            A[row * 64 + n] = fmaf(en, scn, shn);
        } else {
            const float dy = dA[row * 64 + n];
            const float xh = (en - mu) * rs;
            if (MODE == 1) {
                dgam = fmaf(dy, xh, dgam); dbet += dy;
                s1 = fmaf(g, dy, s1); s2 = fmaf(g * dy, xh, s2);
            } else {
                const float denc = rs * (g * dy - m1 - xh * m2) + (dEnc ? dEnc[row * 64 + n] : 0.f);
                if constexpr (MAT) {
                    dXout[row * 64 + n] = denc;
                } else {
                    const float dpre = en > 0.f ? denc : 0.f;
#pragma unroll
                    for (int j = 0; j < KSMAX; ++j)
                        if (j < ks) acc[j] = fmaf(dpre, x[j], acc[j]);
                }
            }
        }
    }
    if (MODE == 1) {
        __shared__ float cred[4][2][64];
        __shared__ double sred[2][4];
        cred[slot][0][n] = dgam; cred[slot][1][n] = dbet;
        const double S1 = wave_sum((double)s1), S2 = wave_sum((double)s2);
        if (n == 0) { sred[0][slot] = S1; sred[1][slot] = S2; }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6;
            atomicAdd(&dgb[which * 64 + n], (double)cred[0][which][n] + (double)cred[1][which][n] + (double)cred[2][which][n] + (double)cred[3][which][n]);
        }
        if (threadIdx.x == 0) {
            atomicAdd(&red[2 * b], sred[0][0] + sred[0][1] + sred[0][2] + sred[0][3]);
            atomicAdd(&red[2 * b + 1], sred[1][0] + sred[1][1] + sred[1][2] + sred[1][3]);
        }
    }
    if (MODE == 2 && !MAT) {
        __shared__ float wred[4][64][KSMAX];
#pragma unroll
        for (int j = 0; j < KSMAX; ++j) wred[slot][n][j] = acc[j];
        __syncthreads();
        if (slot == 0) {
#pragma unroll
            for (int j = 0; j < KSMAX; ++j)
                if (j < ks) atomicAdd(&dWe[n * ks + j], (double)wred[0][n][j] + (double)wred[1][n][j] + (double)wred[2][n][j] + (double)wred[3][n][j]);
        }
    }
}

__global__ void scale_f64_kernel(double* __restrict__ x, int n, double f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= f;
}

__global__ void scratch_to_grad_kernel(const double* __restrict__ s, float* __restrict__ g, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = accumulate ? g[i] + (float)s[i] : (float)s[i];
}

// ------------------------------------------------------------------------------------------------ SI-SDR PIT loss
// moments[b][i][j][5] = (sum e, sum t, sum e^2, sum t^2, sum e*t) in fp64, one workgroup per (b, i, j)
__global__ __launch_bounds__(256)
void sisdr_moments_kernel(const float* __restrict__ est, const float* __restrict__ tgt, double* __restrict__ mom, int n, int T) {
    const int b = blockIdx.z, i = blockIdx.y, j = blockIdx.x;
    const float* e = est + ((long long)b * n + i) * T;
    const float* t = tgt + ((long long)b * n + j) * T;
    double v[5] = {0, 0, 0, 0, 0};
    for (int k = threadIdx.x; k < T; k += 256) {
        const double x = e[k], y = t[k];
        v[0] += x; v[1] += y; v[2] += x * x; v[3] += y * y; v[4] += x * y;
    }
    __shared__ double red[5][4];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        v[k] = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5)
        mom[(((long long)b * n + i) * n + j) * 5 + threadIdx.x] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// per item: pairwise SI-SDR, best permutation (maximum mean SI-SDR = minimum PIT loss, first one wins ties in
// itertools.permutations order), loss_b = -mean_i sisdr(i, perm[i]); coef[b][i] = (c1, c2, mean e, mean t) with
// d sisdr / d est_i[k] = c1*(e[k] - mean e) + c2*(t[k] - mean t), t = tgt[perm[i]]
__global__ void pit_select_kernel(const double* __restrict__ mom, int B, int n, int T, double* __restrict__ pw,
                                  double* __restrict__ loss_b, int* __restrict__ perm, double* __restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double eps = 1e-8, N = (double)T;
    double sd[3][3];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double* v = mom + (((long long)b * n + i) * n + j) * 5;
            const double ee = v[2] - v[0] * v[0] / N, tt = v[3] - v[1] * v[1] / N, et = v[4] - v[0] * v[1] / N;
            const double alpha = et / (tt + eps);
            const double s2 = alpha * alpha * tt, n2 = ee - 2.0 * alpha * et + alpha * alpha * tt;
            sd[i][j] = 10.0 * log10(s2 / (n2 + eps) + eps);
            pw[((long long)b * n + i) * n + j] = sd[i][j];
        }
    int best[3] = {0, 1, 2};
    double bs = -1e300;
    const int P3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int P2[2][3] = {{0, 1, 2}, {1, 0, 2}};
    const int np = n == 1 ? 1 : (n == 2 ? 2 : 6);
    for (int p = 0; p < np; ++p) {
        const int* pp = n == 3 ? P3[p] : P2[p];
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += sd[i][pp[i]];
        s /= n;
        if (s > bs) { bs = s; for (int i = 0; i < n; ++i) best[i] = pp[i]; }
    }
    loss_b[b] = -bs;
    for (int i = 0; i < n; ++i) {
        const int j = best[i];
        perm[b * n + i] = j;
        const double* v = mom + (((long long)b * n + i) * n + j) * 5;
        const double ee = v[2] - v[0] * v[0] / N, tt = v[3] - v[1] * v[1] / N, et = v[4] - v[0] * v[1] / N;
        const double ia = 1.0 / (tt + eps), alpha = et * ia;
        const double s2 = alpha * alpha * tt, n2 = ee - 2.0 * alpha * et + alpha * alpha * tt;
        const double R = s2 / (n2 + eps) + eps;
        const double Kc = 10.0 / log(10.0) / R;
        const double q = s2 / ((n2 + eps) * (n2 + eps));
        double* c = coef + ((long long)b * n + i) * 4;
        c[0] = -Kc * 2.0 * q;
        c[1] = Kc * (2.0 * alpha * tt * ia / (n2 + eps) - q * (-4.0 * alpha + 2.0 * alpha * tt * ia));
        c[2] = v[0] / N;
        c[3] = v[1] / N;
    }
}

// dEst[b][i][k] = scale * (c1*(e - me) + c2*(t - mt)), scale = upstream * (-1 / (B*n)) read from a device scalar
__global__ __launch_bounds__(256)
void pit_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt, const int* __restrict__ perm,
                    const double* __restrict__ coef, const float* __restrict__ upstream, int B, int n, int T,
                    float* __restrict__ dEst) {
    const int bi = blockIdx.y, b = bi / n;
    const double* c = coef + (long long)bi * 4;
    const float g = (upstream ? *upstream : 1.f) * (-1.f / (float)(B * n));
    const float c1 = (float)c[0] * g, c2 = (float)c[1] * g, me = (float)c[2], mt = (float)c[3];
    const float* e = est + (long long)bi * T;
    const float* t = tgt + ((long long)b * n + perm[bi]) * T;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < T; k += gridDim.x * 256)
        dEst[(long long)bi * T + k] = c1 * (e[k] - me) + c2 * (t[k] - mt);
}

inline int grid_for(long long items, int cap_mult = 8) {
    long long g = items;
    const long long cap = (long long)num_cus() * cap_mult;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

// absmax (optional, [B][2] words, zeroed here): per utterance max |gamma dy| and max |x_hat| as float bits -- with red they bound |dZ|:
// |dZ| <= rstd (max|gamma dy| + |red0 / n| + max|x_hat| |red1 / n|), the scale source of dprnn_lin_bwd_f16, which forms dZ on the fly
extern "C" int dprnn_gn_bwd_stats_ex(const float* dXn, const float* dXa, const float* dXb, float* dXo, const float* Z,
                                     const double* stats, const float* gamma, float eps, int B, int64_t rows_per_utt,
                                     double* red, double* dgb, unsigned* absmax, void* stream) {
    if (!dXn || (dXa && (!dXb || !dXo)) || (Z && (!stats || !gamma || !red || !dgb)) || B < 1 || rows_per_utt < 1 || (absmax && !Z)) {
        dprnn_set_error("dprnn_gn_bwd_stats: null pointer / sizes"); return -1;
    }
    if (absmax) {
        hipError_t e = hipMemsetAsync(absmax, 0, sizeof(unsigned) * 2 * B, (hipStream_t)stream);
        if (e != hipSuccess) { dprnn_set_error("dprnn_gn_bwd_stats: %s", hipGetErrorString(e)); return -2; }
    }
    int chunks = (4 * num_cus() + B - 1) / B;
    const long long maxc = (rows_per_utt + 63) / 64;
    if (chunks > maxc) chunks = (int)maxc;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream,
                       dXn, dXa, dXb, dXo, Z, stats, gamma, eps, (long long)rows_per_utt, chunks, red, dgb, absmax);
    return dprnn_check_launch("gn_bwd_stats_kernel");
}
extern "C" int dprnn_gn_bwd_stats(const float* dXn, const float* dXa, const float* dXb, float* dXo, const float* Z,
                                  const double* stats, const float* gamma, float eps, int B, int64_t rows_per_utt,
                                  double* red, double* dgb, void* stream) {
    return dprnn_gn_bwd_stats_ex(dXn, dXa, dXb, dXo, Z, stats, gamma, eps, B, rows_per_utt, red, dgb, nullptr, stream);
}

// One-shot selector: the NEXT dprnn_gn_bwd_dz call also leaves max|dZ| in *slot (device float) -- the scale source of the Linear's
// f16-pieces gradients (dprnn_dgrad_use_f16_pieces, dprnn_wgrad_use_f16_pieces), measured while dZ is written instead of by dprnn_absmax.
static thread_local float* g_gn_dz_absmax_out = nullptr;
extern "C" void dprnn_gn_bwd_dz_store_absmax(float* slot) { g_gn_dz_absmax_out = slot; }

static int gn_bwd_dz_impl(float* max_out, const float* dY, const float* Z, const double* stats, const double* red, const float* gamma,
                          float eps, int B, int64_t rows_per_utt, float* dZ, void* stream);
extern "C" int dprnn_gn_bwd_dz(const float* dY, const float* Z, const double* stats, const double* red, const float* gamma,
                               float eps, int B, int64_t rows_per_utt, float* dZ, void* stream) {
    float* const max_out = g_gn_dz_absmax_out;           // one-shot: consumed by this call whatever it returns
    g_gn_dz_absmax_out = nullptr;
    return gn_bwd_dz_impl(max_out, dY, Z, stats, red, gamma, eps, B, rows_per_utt, dZ, stream);
}
// the same with the selector as an argument: absmax_out != NULL = also leave max|dZ| there (device float)
extern "C" int dprnn_gn_bwd_dz_ex(float* absmax_out, const float* dY, const float* Z, const double* stats, const double* red,
                                  const float* gamma, float eps, int B, int64_t rows_per_utt, float* dZ, void* stream) {
    return gn_bwd_dz_impl(absmax_out, dY, Z, stats, red, gamma, eps, B, rows_per_utt, dZ, stream);
}
static int gn_bwd_dz_impl(float* max_out, const float* dY, const float* Z, const double* stats, const double* red, const float* gamma,
                          float eps, int B, int64_t rows_per_utt, float* dZ, void* stream) {
    if (!dY || !Z || !stats || !red || !gamma || !dZ || B < 1 || rows_per_utt < 1) { dprnn_set_error("dprnn_gn_bwd_dz: null pointer / sizes"); return -1; }
    const long long rows = (long long)B * rows_per_utt;
    if (max_out) {
        hipError_t e = hipMemsetAsync(max_out, 0, sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) { dprnn_set_error("dprnn_gn_bwd_dz: %s", hipGetErrorString(e)); return -2; }
    }
    hipLaunchKernelGGL(gn_bwd_dz_kernel, dim3(grid_for((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                       dY, Z, stats, red, gamma, eps, (long long)rows_per_utt, rows, dZ, reinterpret_cast<unsigned*>(max_out));
    return dprnn_check_launch("gn_bwd_dz_kernel");
}

extern "C" int dprnn_gate_bwd(const float* dV, const float* TG, float* dAG, int64_t rows, void* stream) {
    if (!dV || !TG || !dAG || rows < 1) { dprnn_set_error("dprnn_gate_bwd: null pointer / rows"); return -1; }
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, dV, TG, dAG, (long long)rows);
    return dprnn_check_launch("gate_bwd_kernel");
}

extern "C" int dprnn_mask_decode_bwd(const float* dOut, const float* wav, const float* enc_w, const float* dec_w,
                                     const float* mask, int relu_act, float* dE, float* dEnc, double* dWd, const float* dM0,
                                     int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!dOut || !wav || !enc_w || !dec_w || !mask || !dE || !dEnc || !dWd) { dprnn_set_error("dprnn_mask_decode_bwd: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || L != (T - ks) / stride + 1) { dprnn_set_error("dprnn_mask_decode_bwd: unsupported N=%d ks=%d", N, ks); return -1; }
    const Geo geo{B, T, L, N, ks, stride};
    const int T_out = (L - 1) * stride + ks;
    hipLaunchKernelGGL(mask_decode_bwd_kernel, dim3(grid_for(((long long)B * L + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       dOut, wav, enc_w, dec_w, mask, relu_act, geo, T_out, dE, dEnc, dWd, dM0);
    return dprnn_check_launch("mask_decode_bwd_kernel");
}

extern "C" int dprnn_ola_prelu_sum(const float* X, const float* prelu_w, float* out, const float* dU, double* dbias,
                                   int B, int L, int K, int P, int S, void* stream) {
    if (!X || !prelu_w || !out || !dU || !dbias) { dprnn_set_error("dprnn_ola_prelu_sum: null pointer"); return -1; }
    if (K < 1 || P < 1 || P > K || S != (L + K) / P + 1) { dprnn_set_error("dprnn_ola_prelu_sum: bad geometry"); return -1; }
    hipLaunchKernelGGL(ola_prelu_sum_kernel, dim3(grid_for(((long long)B * L + 7) / 8, 4)), dim3(256), 0, (hipStream_t)stream,
                       X, prelu_w, B, L, K, P, S, out, dU, dbias);
    return dprnn_check_launch("ola_prelu_sum_kernel");
}

extern "C" int dprnn_segment_bwd(const float* dXn, const float* dXa, const float* dXb, float* dY,
                                 int B, int L, int K, int P, int S, void* stream) {
    if (!dXn || (dXa && !dXb) || !dY) { dprnn_set_error("dprnn_segment_bwd: null pointer"); return -1; }
    if (K < 1 || P < 1 || P > K || S != (L + K) / P + 1) { dprnn_set_error("dprnn_segment_bwd: bad geometry"); return -1; }
    hipLaunchKernelGGL(segment_bwd_kernel, dim3(grid_for(((long long)B * L + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                       dXn, dXa, dXb, B, L, K, P, S, dY);
    return dprnn_check_launch("segment_bwd_kernel");
}

static int enc_chunks(int B, int L) {
    int chunks = (4 * num_cus() + B - 1) / B;
    const int maxc = (L + 63) / 64;
    if (chunks > maxc) chunks = maxc;
    return chunks < 1 ? 1 : chunks;
}

extern "C" int dprnn_encoder_norm_rows(const float* wav, const float* enc_w, const float* sc, const float* sh, float* A,
                                       int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !sc || !sh || !A) { dprnn_set_error("dprnn_encoder_norm_rows: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || L != (T - ks) / stride + 1) { dprnn_set_error("dprnn_encoder_norm_rows: unsupported N=%d ks=%d", N, ks); return -1; }
    const Geo geo{B, T, L, N, ks, stride};
    const int ch = enc_chunks(B, L);
    hipLaunchKernelGGL(encoder_rows_kernel<0>, dim3(B * ch), dim3(256), 0, (hipStream_t)stream, wav, enc_w, geo, sc, sh, A,
                       nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, ch);
    return dprnn_check_launch("encoder_rows_kernel<0>");
}

extern "C" int dprnn_encoder_norm_bwd_stats(const float* wav, const float* enc_w, const float* dA, const double* stats,
                                            const float* gamma, float eps, double* red, double* dgb,
                                            int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !dA || !stats || !gamma || !red || !dgb) { dprnn_set_error("dprnn_encoder_norm_bwd_stats: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || L != (T - ks) / stride + 1) { dprnn_set_error("dprnn_encoder_norm_bwd_stats: unsupported N=%d ks=%d", N, ks); return -1; }
    const Geo geo{B, T, L, N, ks, stride};
    const int ch = enc_chunks(B, L);
    hipLaunchKernelGGL(encoder_rows_kernel<1>, dim3(B * ch), dim3(256), 0, (hipStream_t)stream, wav, enc_w, geo, nullptr, nullptr,
                       nullptr, dA, stats, gamma, eps, red, dgb, nullptr, nullptr, ch);
    return dprnn_check_launch("encoder_rows_kernel<1>");
}

extern "C" int dprnn_encoder_bwd(const float* wav, const float* enc_w, const float* dA, const double* stats,
                                 const float* gamma, float eps, const double* red, const float* dEnc, double* dWe,
                                 int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !dA || !stats || !gamma || !red || !dWe) { dprnn_set_error("dprnn_encoder_bwd: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || L != (T - ks) / stride + 1) { dprnn_set_error("dprnn_encoder_bwd: unsupported N=%d ks=%d", N, ks); return -1; }
    const Geo geo{B, T, L, N, ks, stride};
    const int ch = enc_chunks(B, L);
    hipLaunchKernelGGL(encoder_rows_kernel<2>, dim3(B * ch), dim3(256), 0, (hipStream_t)stream, wav, enc_w, geo, nullptr, nullptr,
                       nullptr, dA, stats, gamma, eps, const_cast<double*>(red), nullptr, dEnc, dWe, ch);
    return dprnn_check_launch("encoder_rows_kernel<2>");
}

// ---- the same three passes on a materialised [B][L][64] tensor (GroupNorm(1,64) of d0 in DPRNN-Spe-IRA)
extern "C" int dprnn_rows64_affine(const float* X, const float* sc, const float* sh, float* A, int B, int L, void* stream) {
    if (!X || !sc || !sh || !A || B < 1 || L < 1) { dprnn_set_error("dprnn_rows64_affine: null pointer / sizes"); return -1; }
    const Geo geo{B, 0, L, 64, 1, 1};
    const int ch = enc_chunks(B, L);
    hipLaunchKernelGGL((encoder_rows_kernel<0, true>), dim3(B * ch), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, geo, sc, sh, A,
                       nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, ch, X, nullptr);
    return dprnn_check_launch("encoder_rows_kernel<0,mat>");
}

// count_scale: built / real channels when the 64 columns are the zero-padded twin of a narrower model (the two sums are divided by
// L x 64 in the second pass; times count_scale that is L x real channels -- the padded columns add nothing to either sum)
extern "C" int dprnn_rows64_norm_bwd_ex(const float* X, const float* dA, const double* stats, const float* gamma, float eps,
                                        double* red, double* dgb, float* dX, int B, int L, double count_scale, void* stream) {
    if (!X || !dA || !stats || !gamma || !red || !dgb || !dX || B < 1 || L < 1) { dprnn_set_error("dprnn_rows64_norm_bwd: null pointer / sizes"); return -1; }
    if (!(count_scale >= 1.0)) { dprnn_set_error("dprnn_rows64_norm_bwd: count_scale must be >= 1 (built / real channels)"); return -1; }
    const Geo geo{B, 0, L, 64, 1, 1};
    const int ch = enc_chunks(B, L);
    hipLaunchKernelGGL((encoder_rows_kernel<1, true>), dim3(B * ch), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, geo, nullptr,
                       nullptr, nullptr, dA, stats, gamma, eps, red, dgb, nullptr, nullptr, ch, X, nullptr);
    int rc = dprnn_check_launch("encoder_rows_kernel<1,mat>");
    if (rc) return rc;
    if (count_scale != 1.0) {
        hipLaunchKernelGGL(scale_f64_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, red, 2 * B, count_scale);
        rc = dprnn_check_launch("scale_f64_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL((encoder_rows_kernel<2, true>), dim3(B * ch), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, geo, nullptr,
                       nullptr, nullptr, dA, stats, gamma, eps, red, nullptr, nullptr, nullptr, ch, X, dX);
    return dprnn_check_launch("encoder_rows_kernel<2,mat>");
}
extern "C" int dprnn_rows64_norm_bwd(const float* X, const float* dA, const double* stats, const float* gamma, float eps,
                                     double* red, double* dgb, float* dX, int B, int L, void* stream) {
    return dprnn_rows64_norm_bwd_ex(X, dA, stats, gamma, eps, red, dgb, dX, B, L, 1.0, stream);
}

extern "C" int dprnn_scratch_to_grad(const double* scratch, float* grad, int n, int accumulate, void* stream) {
    if (!scratch || !grad || n < 1) { dprnn_set_error("dprnn_scratch_to_grad: null pointer / n"); return -1; }
    hipLaunchKernelGGL(scratch_to_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, grad, n, accumulate);
    return dprnn_check_launch("scratch_to_grad_kernel");
}

extern "C" int dprnn_pit_sisdr_fwd(const float* est, const float* tgt, double* moments, double* pw, double* loss_b,
                                   int* perm, double* coef, int B, int n, int T, void* stream) {
    if (!est || !tgt || !moments || !pw || !loss_b || !perm || !coef) { dprnn_set_error("dprnn_pit_sisdr_fwd: null pointer"); return -1; }
    if (B < 1 || n < 1 || n > 3 || T < 2) { dprnn_set_error("dprnn_pit_sisdr_fwd: n_src must be 1..3, T >= 2"); return -1; }
    hipLaunchKernelGGL(sisdr_moments_kernel, dim3(n, n, B), dim3(256), 0, (hipStream_t)stream, est, tgt, moments, n, T);
    int rc = dprnn_check_launch("sisdr_moments_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(pit_select_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, moments, B, n, T, pw, loss_b, perm, coef);
    return dprnn_check_launch("pit_select_kernel");
}

extern "C" int dprnn_pit_sisdr_bwd(const float* est, const float* tgt, const int* perm, const double* coef,
                                   const float* upstream, float* dEst, int B, int n, int T, void* stream) {
    if (!est || !tgt || !perm || !coef || !dEst) { dprnn_set_error("dprnn_pit_sisdr_bwd: null pointer"); return -1; }
    int gx = (T + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pit_bwd_kernel, dim3(gx, B * n), dim3(256), 0, (hipStream_t)stream, est, tgt, perm, coef, upstream, B, n, T, dEst);
    return dprnn_check_launch("pit_bwd_kernel");
}

// ------------------------------------------------------------------------------------------------ clip + Adam over all parameters
// torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (src/trainers/trainer.py:115-118; optimizer configured in
// scripts/train/config_bss.yaml:30-33) as two launches over a device table of tensors instead of ~10 small launches per
// parameter: (1) sum of squares of every gradient in fp64, (2) the Adam update with the clip coefficient
// min(1, max_norm / (norm + 1e-6)) applied to the gradient on the fly (the update is identical to clipping .grad first;
// .grad itself is left untouched).  weight_decay is the L2 form of torch.optim.Adam (g += wd * p).
namespace {
struct AdamEntry { float* p; const float* g; float* m; float* v; long long n; };

__global__ __launch_bounds__(256)
void multi_sumsq_kernel(const AdamEntry* __restrict__ tab, double* __restrict__ out) {
    const AdamEntry e = tab[blockIdx.y];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e.n; i += (long long)gridDim.x * 256) {
        const double x = e.g[i];
        s += x * x;
    }
    s = wave_sum(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256)
void adam_multi_kernel(const AdamEntry* __restrict__ tab, float lr, float b1, float b2, float eps, float wd,
                       float bc1, float bc2_sqrt, float max_norm, const double* __restrict__ sumsq) {
    const AdamEntry e = tab[blockIdx.y];
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float nrm = (float)sqrt(*sumsq);
        coef = fminf(1.f, max_norm / (nrm + 1e-6f));
    }
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e.n; i += (long long)gridDim.x * 256) {
        float g = e.g[i] * coef;
        const float p = e.p[i];
        if (wd != 0.f) g = fmaf(wd, p, g);
        const float m = b1 * e.m[i] + (1.f - b1) * g;
        const float v = b2 * e.v[i] + (1.f - b2) * g * g;
        e.m[i] = m; e.v[i] = v;
        e.p[i] = p - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}
}  // namespace

extern "C" int dprnn_multi_sumsq(const void* table, int n_tensors, double* out, void* stream) {
    if (!table || !out || n_tensors < 1) { dprnn_set_error("dprnn_multi_sumsq: null pointer / count"); return -1; }
    hipLaunchKernelGGL(multi_sumsq_kernel, dim3(32, n_tensors), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const AdamEntry*>(table), out);
    return dprnn_check_launch("multi_sumsq_kernel");
}

extern "C" int dprnn_adam_multi(const void* table, int n_tensors, float lr, double beta1, double beta2, float eps,
                                float weight_decay, int step, float max_norm, const double* sumsq, void* stream) {
    if (!table || n_tensors < 1 || step < 1 || (max_norm > 0.f && !sumsq)) { dprnn_set_error("dprnn_adam_multi: bad arguments"); return -1; }
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);      // in double, as torch does
    hipLaunchKernelGGL(adam_multi_kernel, dim3(32, n_tensors), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const AdamEntry*>(table), lr, (float)beta1, (float)beta2, eps, weight_decay, (float)bc1,
                       (float)sqrt(bc2), max_norm, sumsq);
    return dprnn_check_launch("adam_multi_kernel");
}
