// Main-path instances of the row GEMM (rowgemm_core.h): bottleneck, Linear+GroupNorm sums, mask head.
#include "rowgemm_core.h"
using namespace rg;

extern "C" int dprnn_num_cus(void) { return num_cus(); }

static int g_fast_head = 1;
extern "C" void dprnn_set_fast_head_activations(int on) { g_fast_head = on; }

static int bottleneck(const float* wav, const float* enc_w, const float* sc, const float* sh,
                      const float* Wb, int ldw, const float* bb, const float* row_bias, const float* row_scale, float* Y,
                      int B, int T, int L, int N, int ks, int stride, void* stream) {
    if (!wav || !enc_w || !sc || !sh || !Wb || !bb || !Y) { dprnn_set_error("dprnn_bottleneck_fwd: null pointer"); return -1; }
    if (N != 64 || ks < 1 || ks > KSMAX || ldw < N || (ldw % 4) != 0 || L != (T - ks) / stride + 1) {
        dprnn_set_error("dprnn_bottleneck_fwd: unsupported N=%d ks=%d ldw=%d L=%d T=%d stride=%d (built for input_size 64, ks<=8)", N, ks, ldw, L, T, stride);
        return -1;
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

extern "C" int dprnn_get_gemm_mode(void);
// gate-GEMM mode 3 (f16 pieces) covers the Linear that follows the recurrence too: its rows are RNN outputs, |h| < 1
#ifndef PROJ_F16_RT
#define PROJ_F16_RT 2
#endif
#ifndef HEAD_F16_NW
#define HEAD_F16_NW 8
#endif

extern "C" int dprnn_proj_stats_fwd(const float* H, const float* W, const float* bias, float* Z, double* stats,
                                    int64_t M, int64_t rows_per_utt, int kdim, void* stream) {
    if (!H || !W || !bias || !Z || !stats) { dprnn_set_error("dprnn_proj_stats_fwd: null pointer"); return -1; }
    if (M <= 0 || rows_per_utt <= 0 || (kdim != 256 && kdim != 128)) {
        dprnn_set_error("dprnn_proj_stats_fwd: bad sizes M=%lld rows_per_utt=%lld kdim=%d", (long long)M, (long long)rows_per_utt, kdim);
        return -1;
    }
    if (dprnn_get_gemm_mode() == 3) {
        PlainRows<PROJ_F16_RT> al{H, kdim};
        EpProjStats<PROJ_F16_RT, 8> ep{Z, bias, stats, (long long)rows_per_utt, nullptr, 0};
        if (kdim == 256) return launch_rowgemm_f16<256, 128, PROJ_F16_RT, 8, false>("proj_stats256_f16", W, 256, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
        return launch_rowgemm_f16<128, 128, PROJ_F16_RT, 8, false>("proj_stats128_f16", W, 128, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
    }
    constexpr int RT = 4;
    PlainRows<RT> al{H, kdim};
    EpProjStats<RT, 8> ep{Z, bias, stats, (long long)rows_per_utt, nullptr, 0};
    if (kdim == 256) return launch_rowgemm<256, 128, RT, false, PROJ_NW>("proj_stats256", W, 256, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
    return launch_rowgemm<128, 128, RT, false, PROJ_NW>("proj_stats128", W, 128, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
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
        PlainRows<PROJ_F16_RT> al{H, kdim};
        EpProjStats<PROJ_F16_RT, 8> ep{Z, bias, stats, 1, reinterpret_cast<const long long*>(utt_starts), n_utt};
        if (kdim == 256) return launch_rowgemm_f16<256, 128, PROJ_F16_RT, 8, false>("proj_stats256r_f16", W, 256, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
        return launch_rowgemm_f16<128, 128, PROJ_F16_RT, 8, false>("proj_stats128r_f16", W, 128, 128, nullptr, 0, M, al, ep, (hipStream_t)stream);
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
    if (dprnn_get_gemm_mode() == 3)        // f16 pieces (rowgemm_f16_kernel): the rows are the residual stream the recurrences read too
        return launch_rowgemm_f16<128, 256, RT, HEAD_F16_NW, true>("head_conv_ola_f16", W2d, 128, 256, nullptr, 0, (long long)B * L, al, ep, (hipStream_t)stream);
    return launch_rowgemm<128, 256, RT>("head_conv_ola", W2d, 128, 256, nullptr, 0, (long long)B * L, al, ep, (hipStream_t)stream);
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
        if (dprnn_get_gemm_mode() == 3 && !TG)
            return launch_rowgemm_f16<128, 256, RT, HEAD_F16_NW, true>("head_gate_f16", Wo, 128, 128, Wg, 128, rows, al, ep, (hipStream_t)stream);
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
    constexpr int RT = 4;
    PlainRows<RT> al{V, 128};
    const Geo geo{B, T, L, N, ks, stride};
    if (g_fast_head) {
        EpMaskDecode<RT, 4, true> ep{wav, enc_w, dec_w, taps, masked_out, relu_act, geo, mask_all};
        if (dprnn_get_gemm_mode() == 3 && !mask_all)       // rows are tanh * sigmoid products, |v| < 1
            return launch_rowgemm_f16<128, 64, RT, HEAD_F16_NW, true>("head_mask_decode_f16", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, al, ep, (hipStream_t)stream);
        return launch_rowgemm<128, 64, RT>("head_mask_decode", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, al, ep, (hipStream_t)stream);
    }
    EpMaskDecode<RT, 4, false> ep{wav, enc_w, dec_w, taps, masked_out, relu_act, geo, mask_all};
    return launch_rowgemm<128, 64, RT>("head_mask_decode", Wend, 128, 64, nullptr, 0, (long long)B * L * 2, al, ep, (hipStream_t)stream);
}
