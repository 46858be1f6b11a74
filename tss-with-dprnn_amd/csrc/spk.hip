// Speaker-encoder branch of DPRNN-Spe / DPRNN-Spe-IRA (reference src/models/dprnn_spe.py:8-42,115-123,156-163):
// 1x1 convs as row GEMMs (rowgemm_core.h) with BatchNorm/PReLU folded into the next GEMM's A loader, MaxPool1d(3)
// fused with the residual add, the time mean and the small per-utterance Linears.
#include "rowgemm_core.h"
using namespace rg;
extern "C" int dprnn_get_gemm_mode(void);

namespace {

// A[row][k] -> prelu((a - m[k]) * s[u*aff_stride + k] + t[...]), u = row / rows_per_utt; every piece optional.
// The mean is subtracted BEFORE the scale (as torch's batch_norm does): folding it into the shift (t - m*s) loses the
// result to cancellation in channels whose mean dwarfs their spread.
template <int RT>
struct AffineRows {
    const float* A; int pitch;
    const float* sc; const float* sh; const float* mn; int aff_stride; long long rows_per_utt;
    const float* prelu_p;
    const float* p[RT];
    const float* ps[RT];
    const float* pt[RT];
    float alpha;
    __device__ void init(long long row0, int r, int q, long long M) {
        alpha = prelu_p ? *prelu_p : 1.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = row0 + 16 * rt + r;
            if (row >= M) row = M - 1;
            p[rt] = A + row * pitch + 4 * q;
            const long long u = aff_stride ? row / rows_per_utt : 0;
            ps[rt] = sc ? sc + u * aff_stride + 4 * q : nullptr;
            pt[rt] = sc ? sh + u * aff_stride + 4 * q : nullptr;
        }
    }
    __device__ f32x4 load(int rt, int g) const {
        f32x4 v = *reinterpret_cast<const f32x4*>(p[rt] + 16 * g);
        if (sc) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(ps[rt] + 16 * g);
            const f32x4 t = *reinterpret_cast<const f32x4*>(pt[rt] + 16 * g);
            if (mn) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(mn + (ps[rt] - sc) + 16 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] -= m[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], s[e], t[e]);
        }
        if (prelu_p) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : alpha * v[e];
        }
        return v;
    }
};

// out[row*pitch + col0 + col] = acc + bias[col0+col]; optional per-channel (sum, sumsq) in fp64 (BatchNorm batch stats)
template <int RT>
struct EpStoreColStats {
    float* out; int pitch; int col0; const float* bias; double* col_stats;
    double d1[8], d2[8];
    __device__ void begin() {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) { d1[ct] = 0.0; d2[ct] = 0.0; }
    }
    __device__ void end(int lane) {
        if (!col_stats) return;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {      // reduce the 4 lane quarters that share a column
            double a = d1[ct], b = d2[ct];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (lane < 16) {
                atomicAdd(&col_stats[2 * (col0 + 16 * ct + lane)], a);
                atomicAdd(&col_stats[2 * (col0 + 16 * ct + lane) + 1], b);
            }
        }
    }
    __device__ void operator()(f32x4 (&acc)[RT][8], long long row0, int r, int q, int lane, long long M) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const int col = col0 + 16 * ct + r;
            const float bv = bias ? bias[col] : 0.f;
            // sums in fp64 element by element: the batch variance is E[y^2] - mean^2, and channels whose mean dwarfs their
            // spread (after PReLU + MaxPool) lose it to fp32 rounding of the squares otherwise
            double p1 = 0.0, p2 = 0.0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long row = row0 + 16 * rt + 4 * q + e;
                    if (row < M) {
                        const float v = acc[rt][ct][e] + bv;
                        out[row * pitch + col] = v;
                        p1 += (double)v; p2 = fma((double)v, (double)v, p2);
                    }
                }
            d1[ct] += p1; d2[ct] += p2;
        }
    }
};

// BatchNorm1d as (y - mean)*scale + shift with scale = gamma/sqrt(var + eps), shift = beta (eps 1e-5).
// col_stats == NULL: running statistics (eval).
// Otherwise batch statistics (biased variance) and the running buffers are updated in place exactly as
// nn.BatchNorm1d does in train mode (momentum, unbiased variance, num_batches_tracked += 1).
__global__ void bn_prep_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                               float* running_var, long long* nbt, const double* __restrict__ col_stats, long long count,
                               float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                               float* __restrict__ mean_out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (col_stats) {
        const double m = col_stats[2 * c] / (double)count;
        double v = col_stats[2 * c + 1] / (double)count - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m; var = (float)v;
        const double unb = count > 1 ? v * (double)count / (double)(count - 1) : v;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        if (c == 0 && nbt) *nbt += 1;
    } else {
        mean = running_mean[c]; var = running_var[c];
    }
    scale[c] = gamma[c] / sqrtf(var + eps);
    shift[c] = beta[c];
    mean_out[c] = mean;
}

// out[b][t3][c] = max_{j<3} prelu((y[b][3*t3+j][c] - mean[c])*scale[c] + shift[c] + res[b][3*t3+j][c])   (dprnn_spe.py:35-42)
__global__ __launch_bounds__(256) void resblock_pool_kernel(const f32x4* __restrict__ Y, const f32x4* __restrict__ scale,
                                                            const f32x4* __restrict__ shift, const f32x4* __restrict__ mean,
                                                            const f32x4* __restrict__ res,
                                                            const float* __restrict__ prelu_w, f32x4* __restrict__ out,
                                                            long long total4, int T_in, int T_out, int C4) {
    const float alpha = *prelu_w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4); const long long bt = i / C4;
        const int t3 = (int)(bt % T_out); const long long b = bt / T_out;
        const f32x4 s = scale[c4], t = shift[c4], mu = mean[c4];
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const long long src = (b * T_in + 3 * t3 + j) * C4 + c4;
            const f32x4 y = Y[src], rr = res[src];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = fmaf(y[e] - mu[e], s[e], t[e]) + rr[e];
                v = v >= 0.f ? v : alpha * v;
                m[e] = fmaxf(m[e], v);
            }
        }
        out[i] = m;
    }
}

// emb[b][c] = sum_t E[b][t][c] / aux_T,  aux_T = ((((len - ks) // (ks//2)) + 1) // 3 // 3 // 3)   (dprnn_spe.py:156-163)
// One block per utterance: TSL time slices x C channels (C <= 128), fixed-order combination (run-to-run deterministic).
constexpr int TSL = 8;
__global__ __launch_bounds__(1024)
void time_mean_kernel(const float* __restrict__ E, const float* __restrict__ aux_len, int n_len,
                      float* __restrict__ emb, int T3, int C, int ks) {
    __shared__ float part[TSL][128];
    const int b = blockIdx.x, c = threadIdx.x % 128, sl = threadIdx.x / 128, nsl = blockDim.x / 128;
    float s = 0.f;
    if (c < C) {
        const int per = (T3 + nsl - 1) / nsl, t0 = sl * per, t1 = t0 + per < T3 ? t0 + per : T3;
        const float* p = E + (long long)b * T3 * C + c;
        for (int t = t0; t < t1; ++t) s += p[(long long)t * C];
    }
    part[sl][c] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < TSL; ++i) tot += i < nsl ? part[i][c] : 0.f;
        const float len = aux_len[n_len > 1 ? b : 0];
        float aT = floorf((len - (float)ks) / (float)(ks / 2)) + 1.f;
        aT = floorf(floorf(floorf(aT / 3.f) / 3.f) / 3.f);
        emb[b * C + c] = tot / aT;
    }
}

// out[b][o] = sum_k W[o*ldw + k0 + k] * in[b*in_pitch + k] + bias[o]
__global__ void linear_small_kernel(const float* __restrict__ in, int in_pitch, const float* __restrict__ W, int ldw,
                                    int k0, int K, const float* __restrict__ bias, float* __restrict__ out, int out_pitch,
                                    int B, int O) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * O) return;
    const int b = idx / O, o = idx % O;
    const float* w = W + (long long)o * ldw + k0;
    const float* x = in + (long long)b * in_pitch;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(w[k], x[k], s);
    out[(long long)b * out_pitch + o] = s + (bias ? bias[o] : 0.f);
}

// per-utterance (sum, sumsq) over a materialised [B][rows_per_utt][C] tensor (GroupNorm(1,C) of d0, dprnn_spe_ira.py:84)
__global__ __launch_bounds__(256) void rows_stats_kernel(const f32x4* __restrict__ A, double* __restrict__ stats,
                                                         long long per_utt4) {
    const int b = blockIdx.y;
    const f32x4* p = A + (long long)b * per_utt4;
    float p1 = 0.f, p2 = 0.f;
    double d1 = 0.0, d2 = 0.0;
    int n = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_utt4; i += (long long)gridDim.x * 256) {
        const f32x4 v = p[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { p1 += v[e]; p2 = fmaf(v[e], v[e], p2); }
        if (++n == 32) { d1 += p1; d2 += p2; p1 = p2 = 0.f; n = 0; }
    }
    d1 += p1; d2 += p2;
    d1 = wave_sum(d1); d2 = wave_sum(d2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[2 * b], d1); atomicAdd(&stats[2 * b + 1], d2); }
}

// ---- 'att' fusion (dprnn_spe.py:98-105,178-183,214-222).  scores[b][t] = sum_c a[b][c] * avg[b][c][t] with
// avg = depth-wise Conv1d(k=ks_a, stride=ks_a) of the normalised encoder output xn[b,c,l] = enc*sc0[b][c] + sh0[b][c].
__global__ __launch_bounds__(256) void att_scores_kernel(const float* __restrict__ wav, const float* __restrict__ We,
                                                         const float* __restrict__ sc0, const float* __restrict__ sh0,
                                                         const float* __restrict__ avg_w, const float* __restrict__ avg_b,
                                                         const float* __restrict__ a, float* __restrict__ scores,
                                                         int T, int La, int N, int ks, int stride, int ks_a) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= La) return;
    float s = 0.f;
    for (int c = 0; c < N; ++c) {
        float av = avg_b[c];
        for (int j = 0; j < ks_a; ++j) {
            const float* x = wav + (long long)b * T + (long long)(t * ks_a + j) * stride;
            float e = 0.f;
            for (int jj = 0; jj < ks; ++jj) e = fmaf(We[c * ks + jj], x[jj], e);
            e = relu_f(e);
            av = fmaf(avg_w[c * ks_a + j], fmaf(e, sc0[b * N + c], sh0[b * N + c]), av);
        }
        s = fmaf(av, a[b * N + c], s);
    }
    scores[(long long)b * La + t] = s;
}

// softmax over time per utterance, then nearest-neighbour upsampling to L frames (nn.Upsample(size=L), float index
// arithmetic as torch: src = min(floor(l * (La / L)), La - 1)); row_scale[b][l] = att[b][src] + 1  (att*a + a = a*(att+1))
__global__ __launch_bounds__(256) void att_softmax_upsample_kernel(const float* __restrict__ scores, float* __restrict__ row_scale,
                                                                   int La, int L) {
    const int b = blockIdx.x;
    const float* s = scores + (long long)b * La;
    __shared__ float red[4];
    __shared__ float stat[2];
    float m = -INFINITY;
    for (int t = threadIdx.x; t < La; t += 256) m = fmaxf(m, s[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
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
    for (int l = threadIdx.x; l < L; l += 256) {
        int src = (int)floorf((float)l * scale);
        if (src > La - 1) src = La - 1;
        row_scale[(long long)b * L + l] = expf(s[src] - m) * inv + 1.f;
    }
}

// 'att' scores from MATERIALISED rows xn[b][l][c] (separator-level callers hold encoder outputs, not the waveform): the same
// sums as att_scores_kernel with the rows in place of the recomputed encoder output.
__global__ __launch_bounds__(256) void att_scores_rows_kernel(const float* __restrict__ rows, int pitch, const float* __restrict__ sc0,
                                                              const float* __restrict__ sh0, const float* __restrict__ avg_w,
                                                              const float* __restrict__ avg_b, const float* __restrict__ a,
                                                              float* __restrict__ scores, int L, int La, int N, int ks_a) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= La) return;
    float s = 0.f;
    for (int c = 0; c < N; ++c) {
        float av = avg_b[c];
        for (int j = 0; j < ks_a; ++j) {
            const float e = rows[((long long)b * L + (long long)t * ks_a + j) * pitch + c];
            av = fmaf(avg_w[c * ks_a + j], sc0 ? fmaf(e, sc0[b * N + c], sh0[b * N + c]) : e, av);
        }
        s = fmaf(av, a[b * N + c], s);
    }
    scores[(long long)b * La + t] = s;
}

// out[row][c] = (A[row][c] * sc[u][c] + sh[u][c]) * row_scale[row], u = row / rows_per_utt; sc / sh / row_scale optional (1, 0, 1).
// The per-frame glue of the separator-level surfaces: GroupNorm / gLN apply, add / mul / film fusion, the 'att' frame scale.
__global__ __launch_bounds__(256) void rows_affine_kernel(const f32x4* __restrict__ A, const float* __restrict__ sc, const float* __restrict__ sh,
                                                          const float* __restrict__ row_scale, f32x4* __restrict__ out,
                                                          long long M, int C4, long long rows_per_utt) {
    const long long n = M * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long row = i / C4;
        const int c4 = (int)(i - row * C4);
        const long long u = row / rows_per_utt;
        f32x4 v = A[i];
        if (sc) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(sc + (u * C4 + c4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= s[e];
        }
        if (sh) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(sh + (u * C4 + c4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
        if (row_scale) {
            const float r = row_scale[row];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= r;
        }
        out[i] = v;
    }
}

// the same store on the transposed f16-pieces tiles (rowgemm_f16t_kernel: a lane owns four adjacent output columns of one row)
template <int RT, int NCT>
struct EpTStoreBias {
    float* out; int pitch; int col0; const float* bias;
    __device__ void begin(int tid) {}
    __device__ void end(int lane) {}
    __device__ void operator()(f32x4 (&acc)[RT][NCT], long long row0, int c, int q, long long M, float inv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt + c;
            if (row >= M) continue;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 v = bias ? *reinterpret_cast<const f32x4*>(bias + col0 + 16 * ct + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[rt][ct][e], inv, v[e]);
                *reinterpret_cast<f32x4*>(out + row * pitch + col0 + 16 * ct + 4 * q) = v;
            }
        }
    }
};

// 1x1 convs of the speaker encoder in gate-GEMM mode 3: two f16 pieces per operand, three products (fp32-class, as everywhere in that
// mode), HBM-bound instead of fp32-MFMA-bound -- 6.3 -> ms of the DPRNN-Spe forward's serial prologue at 64 x 3 s (13 launches).  The rows
// are BatchNorm / PReLU outputs of no fixed scale: tracked, scaled by the calling thread's activation-scale record.  Batch statistics
// (train-mode BatchNorm: fp64 column sums in the epilogue) stay on the fp32 kernel.
template <int KD, int RT>
int launch_pointwise_f16(const AffineRows<RT>& al, const EpTStoreBias<RT, 8>& ep, const float* W, int ldw, long long M, hipStream_t st) {
    return launch_rowgemm_f16t<KD, 128, RT, true>("pointwise_f16", W, ldw, 128, nullptr, 0, M, al, ep, st);
}

template <int KD>
int launch_pointwise(const AffineRows<4>& al, const EpStoreColStats<4>& ep, const float* W, int ldw, long long M, hipStream_t st) {
    return launch_rowgemm<KD, 128, 4>("pointwise", W, ldw, 128, nullptr, 0, M, al, ep, st);
}

}  // namespace

extern "C" int dprnn_pointwise_fwd(const float* A, int a_pitch, int kdim, const float* in_scale, const float* in_shift,
                                   const float* in_mean, int aff_stride, int64_t rows_per_utt, const float* in_prelu, const float* W, int ldw,
                                   const float* bias, float* out, int out_pitch, int col0, double* col_stats, int64_t M,
                                   void* stream) {
    if (!A || !W || !out || M <= 0) { dprnn_set_error("dprnn_pointwise_fwd: null pointer / M"); return -1; }
    if ((kdim != 64 && kdim != 128 && kdim != 256) || a_pitch < kdim || (a_pitch % 4) || ldw < kdim || (ldw % 4) ||
        (col0 % 16) || out_pitch < col0 + 128 || (aff_stride && rows_per_utt <= 0) || ((in_scale == nullptr) != (in_shift == nullptr)) ||
        (in_mean && !in_scale)) {
        dprnn_set_error("dprnn_pointwise_fwd: unsupported kdim=%d a_pitch=%d ldw=%d col0=%d out_pitch=%d", kdim, a_pitch, ldw, col0, out_pitch);
        return -1;
    }
    const float* Wc = W + (long long)col0 * ldw;
    if (dprnn_get_gemm_mode() == 3 && !col_stats) {
        const long long rpu = rows_per_utt > 0 ? rows_per_utt : 1;
        hipStream_t st = (hipStream_t)stream;
        if (kdim == 256) {                             // two 16-row tiles per wave would spill at K = 256 (73 registers)
            AffineRows<1> alt{A, a_pitch, in_scale, in_shift, in_mean, aff_stride, rpu, in_prelu};
            EpTStoreBias<1, 8> ept{out, out_pitch, col0, bias};
            return launch_pointwise_f16<256, 1>(alt, ept, Wc, ldw, M, st);
        }
        AffineRows<2> alt{A, a_pitch, in_scale, in_shift, in_mean, aff_stride, rpu, in_prelu};
        EpTStoreBias<2, 8> ept{out, out_pitch, col0, bias};
        if (kdim == 64) return launch_pointwise_f16<64, 2>(alt, ept, Wc, ldw, M, st);
        return launch_pointwise_f16<128, 2>(alt, ept, Wc, ldw, M, st);
    }
    AffineRows<4> al{A, a_pitch, in_scale, in_shift, in_mean, aff_stride, (long long)(rows_per_utt > 0 ? rows_per_utt : 1), in_prelu};
    EpStoreColStats<4> ep{out, out_pitch, col0, bias, col_stats};
    // the epilogue indexes bias/out/stats with col0 + col, the weight rows are pre-offset
    ep.col0 = col0;
    hipStream_t st = (hipStream_t)stream;
    if (kdim == 64) return launch_pointwise<64>(al, ep, Wc, ldw, M, st);
    if (kdim == 128) return launch_pointwise<128>(al, ep, Wc, ldw, M, st);
    return launch_pointwise<256>(al, ep, Wc, ldw, M, st);
}

extern "C" int dprnn_bn_prep(const float* gamma, const float* beta, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, const double* col_stats, int64_t count, float momentum,
                             float eps, float* scale, float* shift, float* mean, int C, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || !mean) { dprnn_set_error("dprnn_bn_prep: null pointer"); return -1; }
    hipLaunchKernelGGL(bn_prep_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, (long long*)num_batches_tracked, col_stats, (long long)count, momentum, eps, scale, shift, mean, C);
    return dprnn_check_launch("bn_prep_kernel");
}

extern "C" int dprnn_resblock_pool_fwd(const float* Y2, const float* scale, const float* shift, const float* mean, const float* res,
                                       const float* prelu_w, float* out, int B, int T_in, int C, void* stream) {
    if (!Y2 || !scale || !shift || !mean || !res || !prelu_w || !out) { dprnn_set_error("dprnn_resblock_pool_fwd: null pointer"); return -1; }
    if (C % 4 || T_in < 3) { dprnn_set_error("dprnn_resblock_pool_fwd: bad sizes T_in=%d C=%d", T_in, C); return -1; }
    const int T_out = T_in / 3;
    const long long total4 = (long long)B * T_out * (C / 4);
    long long g = (total4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(resblock_pool_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(Y2), reinterpret_cast<const f32x4*>(scale), reinterpret_cast<const f32x4*>(shift),
                       reinterpret_cast<const f32x4*>(mean), reinterpret_cast<const f32x4*>(res), prelu_w, reinterpret_cast<f32x4*>(out),
                       total4, T_in, T_out, C / 4);
    return dprnn_check_launch("resblock_pool_kernel");
}

extern "C" int dprnn_time_mean_fwd(const float* E, const float* aux_len, int n_len, float* emb, int B, int T3, int C, int ks,
                                   void* stream) {
    if (!E || !aux_len || !emb) { dprnn_set_error("dprnn_time_mean_fwd: null pointer"); return -1; }
    if (C > 128 || ks < 2 || (n_len != 1 && n_len != B)) { dprnn_set_error("dprnn_time_mean_fwd: bad sizes C=%d ks=%d n_len=%d", C, ks, n_len); return -1; }
    static const int nsl = getenv("DPRNN_TIME_SLICES") ? atoi(getenv("DPRNN_TIME_SLICES")) : TSL;   // A/B knob (1 .. 8)
    hipLaunchKernelGGL(time_mean_kernel, dim3(B), dim3(128 * (nsl < 1 ? 1 : nsl > TSL ? TSL : nsl)), 0, (hipStream_t)stream, E, aux_len, n_len, emb, T3, C, ks);
    return dprnn_check_launch("time_mean_kernel");
}

extern "C" int dprnn_linear_small_fwd(const float* in, int in_pitch, const float* W, int ldw, int k0, int K, const float* bias,
                                      float* out, int out_pitch, int B, int O, void* stream) {
    if (!in || !W || !out) { dprnn_set_error("dprnn_linear_small_fwd: null pointer"); return -1; }
    hipLaunchKernelGGL(linear_small_kernel, dim3((B * O + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, in_pitch, W, ldw,
                       k0, K, bias, out, out_pitch, B, O);
    return dprnn_check_launch("linear_small_kernel");
}

extern "C" int dprnn_rows_stats(const float* A, double* stats, int B, int64_t rows_per_utt, int C, void* stream) {
    if (!A || !stats) { dprnn_set_error("dprnn_rows_stats: null pointer"); return -1; }
    if (C % 4) { dprnn_set_error("dprnn_rows_stats: C must be a multiple of 4"); return -1; }
    const long long per4 = (long long)rows_per_utt * C / 4;
    long long g = (per4 + 256 * 16 - 1) / (256 * 16); if (g > 256) g = 256; if (g < 1) g = 1;
    hipLaunchKernelGGL(rows_stats_kernel, dim3((unsigned)g, B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(A), stats, per4);
    return dprnn_check_launch("rows_stats_kernel");
}

extern "C" int dprnn_att_row_scale_rows(const float* rows, int pitch, const float* sc0, const float* sh0, const float* avg_w, const float* avg_b,
                                        const float* a, float* scores, float* row_scale, int B, int L, int N, int ks_a, void* stream) {
    if (!rows || !avg_w || !avg_b || !a || !scores || !row_scale || ((sc0 == nullptr) != (sh0 == nullptr))) { dprnn_set_error("dprnn_att_row_scale_rows: null pointer"); return -1; }
    if (ks_a < 1 || L < ks_a || pitch < N || B < 1) { dprnn_set_error("dprnn_att_row_scale_rows: bad sizes L=%d ks_a=%d pitch=%d N=%d", L, ks_a, pitch, N); return -1; }
    const int La = (L - ks_a) / ks_a + 1;
    hipLaunchKernelGGL(att_scores_rows_kernel, dim3((La + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, rows, pitch, sc0, sh0, avg_w, avg_b, a,
                       scores, L, La, N, ks_a);
    int rc = dprnn_check_launch("att_scores_rows_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(att_softmax_upsample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, row_scale, La, L);
    return dprnn_check_launch("att_softmax_upsample_kernel");
}

extern "C" int dprnn_rows_affine_fwd(const float* A, const float* sc, const float* sh, const float* row_scale, float* out, int64_t M, int C,
                                     int64_t rows_per_utt, void* stream) {
    if (!A || !out || M < 1) { dprnn_set_error("dprnn_rows_affine_fwd: null pointer / M"); return -1; }
    if (C < 4 || (C % 4) || rows_per_utt < 1) { dprnn_set_error("dprnn_rows_affine_fwd: C=%d must be a multiple of 4, rows_per_utt >= 1", C); return -1; }
    long long g = (M * (C / 4) + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rows_affine_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(A), sc, sh, row_scale,
                       reinterpret_cast<f32x4*>(out), (long long)M, C / 4, (long long)rows_per_utt);
    return dprnn_check_launch("rows_affine_kernel");
}

extern "C" int dprnn_att_row_scale(const float* wav, const float* enc_w, const float* sc0, const float* sh0,
                                   const float* avg_w, const float* avg_b, const float* a, float* scores, float* row_scale,
                                   int B, int T, int L, int N, int ks, int stride, int ks_a, void* stream) {
    if (!wav || !enc_w || !sc0 || !sh0 || !avg_w || !avg_b || !a || !scores || !row_scale) { dprnn_set_error("dprnn_att_row_scale: null pointer"); return -1; }
    if (ks_a < 1 || L < ks_a || ks < 1) { dprnn_set_error("dprnn_att_row_scale: bad sizes L=%d ks_a=%d", L, ks_a); return -1; }
    const int La = (L - ks_a) / ks_a + 1;
    hipLaunchKernelGGL(att_scores_kernel, dim3((La + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, wav, enc_w, sc0, sh0,
                       avg_w, avg_b, a, scores, T, La, N, ks, stride, ks_a);
    int rc = dprnn_check_launch("att_scores_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(att_softmax_upsample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, row_scale, La, L);
    return dprnn_check_launch("att_softmax_upsample_kernel");
}
