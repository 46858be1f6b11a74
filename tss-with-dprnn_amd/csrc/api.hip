// Error plumbing and ABI bookkeeping of libdprnn_mi355x.so (see include/dprnn_mi355x.h).
#include "common.h"
#include "dprnn_mi355x.h"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <unordered_map>

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
extern "C" int dprnn_abi_version(void) { return DPRNN_ABI_VERSION; }

// ---- saturation flags: the calling thread's device buffer (common.h: sat_track / sat_commit)
static thread_local int* g_sat_flags = nullptr;
int* dprnn_sat_ptr() { return g_sat_flags; }

extern "C" int dprnn_saturation_begin(void* flags, void* stream) {
    g_sat_flags = static_cast<int*>(flags);
    if (!flags) return 0;
    hipError_t e = hipMemsetAsync(flags, 0, DPRNN_SAT_SLOTS * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) { dprnn_set_error("dprnn_saturation_begin: %s", hipGetErrorString(e)); g_sat_flags = nullptr; return -2; }
    return 0;
}

extern "C" int dprnn_saturation_flags(void* host_out, void* stream) {
    if (!host_out) { dprnn_set_error("dprnn_saturation_flags: null pointer"); return -1; }
    int* out = static_cast<int*>(host_out);
    for (int i = 0; i < DPRNN_SAT_SLOTS; ++i) out[i] = 0;
    if (!g_sat_flags) return 0;
    hipError_t e = hipMemcpyAsync(out, g_sat_flags, DPRNN_SAT_SLOTS * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { dprnn_set_error("dprnn_saturation_flags: %s", hipGetErrorString(e)); return -2; }
    int any = 0;
    for (int i = 0; i < DPRNN_SAT_SLOTS; ++i) any |= out[i] != 0;
    return any;
}

// ---- adaptive activation scale (common.h): the calling thread's record and the slot of the next tracked launch
static thread_local float* g_scale_state = nullptr;
static thread_local int g_scale_slot = 0;
ScaleRef dprnn_scale_ref() {
    if (!g_scale_state) return ScaleRef{nullptr, nullptr};
    return ScaleRef{g_scale_state + g_scale_slot, reinterpret_cast<unsigned*>(g_scale_state) + DPRNN_SCALE_SLOTS + g_scale_slot};
}
extern "C" int dprnn_scale_begin(void* state) { g_scale_state = static_cast<float*>(state); g_scale_slot = 0; return 0; }
extern "C" int dprnn_scale_slot(int slot) {
    if (slot < 0 || slot >= DPRNN_SCALE_SLOTS) { dprnn_set_error("dprnn_scale_slot: slot %d of %d", slot, DPRNN_SCALE_SLOTS); return -1; }
    g_scale_slot = slot;
    return 0;
}
__global__ void scale_update_kernel(float* state, int init) {
    const int i = threadIdx.x;
    if (i >= DPRNN_SCALE_SLOTS) return;
    unsigned* track = reinterpret_cast<unsigned*>(state) + DPRNN_SCALE_SLOTS;
    if (init) { state[i] = 256.f; track[DPRNN_SCALE_SLOTS + i] = 0u; track[2 * DPRNN_SCALE_SLOTS + i] = 0u; }     // 4 x SLOTS words (dprnn_scale_join)
    else {
        const float seen = __uint_as_float(track[i]);                  // NaN / Inf patterns (a non-finite input) leave the record alone
        if (seen < 3.0e38f && seen > state[i]) state[i] = seen;
    }
    track[i] = 0u;
}
extern "C" int dprnn_scale_init(void* state, void* stream) {
    if (!state) { dprnn_set_error("dprnn_scale_init: null pointer"); return -1; }
    hipLaunchKernelGGL(scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<float*>(state), 1);
    return dprnn_check_launch("scale_update_kernel");
}
extern "C" int dprnn_scale_update(void* state, void* stream) {
    if (!state) { dprnn_set_error("dprnn_scale_update: null pointer"); return -1; }
    hipLaunchKernelGGL(scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<float*>(state), 0);
    return dprnn_check_launch("scale_update_kernel");
}

// Per-forward working copy (round 5).  The sticky maxima of a model are shared by every forward of it, and a forward reads its
// scale more than once (the pack kernel folds 2^(4 - s) into W_ih, the recurrence splits x by 2^s): a second thread's update
// between the two would give W_ih and x different scales, silently.  So a forward FORKS the record -- `work` (2 x SLOTS words, its
// own) receives a snapshot of the sticky maxima, which is what its launches read, and collects its maxima -- and JOINS at its end:
// sticky = max(sticky, seen) with an atomic, and the record can come DOWN again: a site whose forward maximum stays 16 x (four
// exponents) below the sticky one for DPRNN_SCALE_DECAY_FORWARDS consecutive forwards falls back to the largest value of that
// streak (never below 256, the default range) -- one outlier forward no longer costs every later forward four bits of its
// second f16 piece.  state: 4 x SLOTS words {sticky, (single-record tracking), low-streak length, low-streak maximum}.
constexpr int DPRNN_SCALE_DECAY_FORWARDS = 8;
__global__ void scale_fork_kernel(const float* state, float* work) {
    const int i = threadIdx.x;
    if (i >= DPRNN_SCALE_SLOTS) return;
    work[i] = state[i];
    reinterpret_cast<unsigned*>(work)[DPRNN_SCALE_SLOTS + i] = 0u;
}
__global__ void scale_join_kernel(float* state, const float* work) {
    const int i = threadIdx.x;
    if (i >= DPRNN_SCALE_SLOTS) return;
    const float seen = __uint_as_float(reinterpret_cast<const unsigned*>(work)[DPRNN_SCALE_SLOTS + i]);
    if (!(seen > 0.f && seen < 3.0e38f)) return;                          // site unused in this forward, or a non-finite input
    unsigned* sticky = reinterpret_cast<unsigned*>(state) + i;            // positive floats order like their bit patterns
    unsigned* streak = reinterpret_cast<unsigned*>(state) + 2 * DPRNN_SCALE_SLOTS + i;
    unsigned* smax = reinterpret_cast<unsigned*>(state) + 3 * DPRNN_SCALE_SLOTS + i;
    const float cur = __uint_as_float(atomicMax(sticky, __float_as_uint(seen)));
    if (seen * 16.f < cur) {
        atomicMax(smax, __float_as_uint(seen));
        if (atomicAdd(streak, 1u) + 1u >= (unsigned)DPRNN_SCALE_DECAY_FORWARDS) {
            const float m = fmaxf(__uint_as_float(atomicExch(smax, 0u)), 256.f);
            atomicExch(sticky, __float_as_uint(m));
            atomicExch(streak, 0u);
        }
    } else {
        atomicExch(streak, 0u);
        atomicExch(smax, 0u);
    }
}
extern "C" int dprnn_scale_fork(const void* state, void* work, void* stream) {
    if (!state || !work) { dprnn_set_error("dprnn_scale_fork: null pointer"); return -1; }
    hipLaunchKernelGGL(scale_fork_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<const float*>(state), static_cast<float*>(work));
    g_scale_state = static_cast<float*>(work); g_scale_slot = 0;
    return dprnn_check_launch("scale_fork_kernel");
}
extern "C" int dprnn_scale_join(void* state, const void* work, void* stream) {
    if (!state || !work) { dprnn_set_error("dprnn_scale_join: null pointer"); return -1; }
    hipLaunchKernelGGL(scale_join_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<float*>(state), static_cast<const float*>(work));
    return dprnn_check_launch("scale_join_kernel");
}

// ---- packed-weight formats, keyed by device address (common.h)
static std::mutex g_pack_mu;
static std::unordered_map<const void*, unsigned> g_pack_fmt;
static unsigned pack_word(int kind, int cell, int mode, int ndir) {
    return ((unsigned)DPRNN_ABI_VERSION << 24) | ((unsigned)kind << 16) | ((unsigned)cell << 12) | ((unsigned)mode << 8) | (unsigned)ndir;
}
static const char* pack_mode_name(int kind, int mode) {
    if (kind == DPRNN_PACK_BWD) return mode == 3 ? "bwd mode 3 (f16 pieces)" : "bwd mode 0 (fp32 fragments)";
    switch (mode) {
        case 0: return "mode 0 (fp32 fragments)";
        case 1: return "mode 1 (bf16 pieces)";
        case 3: return "mode 3 (f16 pieces)";
        default: return "an unknown mode";
    }
}
void dprnn_pack_note(const void* packed, int kind, int cell, int mode, int ndir) {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    g_pack_fmt[packed] = pack_word(kind, cell, mode, ndir);
}
int dprnn_pack_expect(const void* packed, int kind, int cell, int mode, int ndir, const char* what) {
    unsigned have = 0;
    {
        std::lock_guard<std::mutex> lk(g_pack_mu);
        auto it = g_pack_fmt.find(packed);
        if (it != g_pack_fmt.end()) have = it->second;
    }
    if (!have) {
        dprnn_set_error("%s: the weight buffer at %p was not filled by a dprnn_*_pack* call of this library", what, packed);
        return DPRNN_EFORMAT;
    }
    const int h_kind = (have >> 16) & 0xff, h_cell = (have >> 12) & 0xf, h_mode = (have >> 8) & 0xf, h_ndir = have & 0xff;
    if ((have >> 24) != (unsigned)DPRNN_ABI_VERSION || h_kind != kind || h_cell != cell || h_mode != mode || h_ndir < ndir) {
        dprnn_set_error("%s: packed weights at %p hold %s of cell %d, %d direction(s), %s; this launch reads cell %d, %d direction(s), %s "
                        "-- pack and launch under the same mode", what, packed, h_kind == DPRNN_PACK_BWD ? "the backward layout" : "the forward layout",
                        h_cell, h_ndir, pack_mode_name(h_kind, h_mode), cell, ndir, pack_mode_name(kind, mode));
        return DPRNN_EFORMAT;
    }
    return 0;
}
extern "C" void dprnn_pack_forget(const void* packed) {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    if (packed) g_pack_fmt.erase(packed); else g_pack_fmt.clear();
}

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
