// Error plumbing and ABI bookkeeping of libdprnn_mi355x.so (see include/dprnn_mi355x.h).
#include "common.h"
#include "dprnn_mi355x.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void dprnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int dprnn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dprnn_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" const char* dprnn_last_error(void) { return g_err; }
extern "C" int dprnn_abi_version(void) { return 1; }

// ---- buffer sizes of one forward (what a C host must allocate; tss-with-dprnn_amd/engine.py sizes its arenas with it)
extern "C" int64_t dprnn_workspace_bytes(int B, int T, int N, int K, int P, int ks, int stride, int n_blocks,
                                         int split_gemm, int64_t* parts) {
    if (B < 1 || N < 1 || K < 1 || P < 1 || P > K || ks < 1 || stride < 1 || n_blocks < 0 || T < ks) {
        dprnn_set_error("dprnn_workspace_bytes: bad geometry (B=%d T=%d N=%d K=%d P=%d ks=%d stride=%d blocks=%d)",
                        B, T, N, K, P, ks, stride, n_blocks);
        return -1;
    }
    const int64_t F = DPRNN_F, f4 = sizeof(float);
    const int64_t L = (T - ks) / stride + 1;          // nn.Conv1d output length (encoder_decoder.py:25-33)
    const int64_t S = (L + K) / P + 1;                // unfold with padding K (dprnn.py:192-198)
    const int64_t rows = (int64_t)B * S * K;
    const int64_t head = 2 * (int64_t)B * L * 2 * F;  // U [B*L][2F] + V [B*L*2][F] reuse the H region
    int64_t p[DPRNN_WS_PARTS];
    p[DPRNN_WS_Y] = (int64_t)B * L * F * f4;
    p[DPRNN_WS_X] = rows * F * f4;
    p[DPRNN_WS_X2] = rows * F * f4;
    p[DPRNN_WS_Z] = rows * F * f4;
    p[DPRNN_WS_H] = (rows * 2 * F > head ? rows * 2 * F : head) * f4;
    p[DPRNN_WS_STATS] = (int64_t)(1 + 2 * n_blocks) * B * 2 * sizeof(double);
    p[DPRNN_WS_PACKED] = (int64_t)2 * n_blocks * 2 * (split_gemm ? 196608 : 4 * 16 * 8 * 256) * f4;
    p[DPRNN_WS_BSUM] = (int64_t)2 * n_blocks * 2 * 4 * DPRNN_HD * f4;
    p[DPRNN_WS_AFFINE] = (int64_t)2 * B * N * f4;
    p[DPRNN_WS_TAPS] = (int64_t)B * 2 * L * ks * f4;
    p[DPRNN_WS_OUT] = (int64_t)B * 2 * ((L - 1) * stride + ks) * f4;
    int64_t total = 0;
    for (int i = 0; i < DPRNN_WS_PARTS; ++i) {
        if (parts) parts[i] = p[i];
        total += p[i];
    }
    return total;
}
