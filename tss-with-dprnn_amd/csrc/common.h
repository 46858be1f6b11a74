// Shared device helpers for the gfx950 DPRNN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k]*B[k][j], k<4.
// lane l: a = A[l&15][l>>4], b = B[l>>4][l&15], acc reg r = D[4*(l>>4)+r][l&15]  (exact f32 fma chain).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ReLU that keeps NaN (torch.relu(nan) = nan; fmaxf(nan, 0) would be 0): a NaN sample must come out as NaN, as the reference returns it
__device__ __forceinline__ float relu_f(float x) { return x < 0.f ? 0.f : x; }

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_acc(float x) { return tanhf(x); }

// Fast forms: v_exp_f32 + v_rcp_f32 (about 1 ulp each).  Selected by the FAST template flag.
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    // 1 - 2/(e^{2x}+1): saturates correctly at +-inf, absolute error ~1e-7 near 0
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.0f);
}

template <bool FAST> __device__ __forceinline__ float sigmoid_f(float x) {
    if constexpr (FAST) return sigmoid_fast(x); else return sigmoid_acc(x);
}
template <bool FAST> __device__ __forceinline__ float tanh_f(float x) {
    if constexpr (FAST) return tanh_fast(x); else return tanh_acc(x);
}

// activation of (x + b) with the bias pre-scaled for the FAST forms: bk = -log2e*b (sigmoid), 2*log2e*b (tanh);
// for the libm forms bk is the plain bias.
template <bool FAST> __device__ __forceinline__ float sigmoid_b(float x, float bk) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(x, -1.4426950408889634f, bk)));
    else return sigmoid_acc(x + bk);
}
template <bool FAST> __device__ __forceinline__ float tanh_b(float x, float bk) {
    if constexpr (FAST) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(x, 2.8853900817779268f, bk)) + 1.0f);
    else return tanh_acc(x + bk);
}

// Two-element forms: the regular arithmetic becomes packed fp32 instructions (v_pk_fma_f32 / v_pk_add_f32, two results per
// issue slot); v_exp_f32 / v_rcp_f32 stay per element.  Same operations as the scalar forms above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool FAST> __device__ __forceinline__ f32x2 sigmoid_b2(f32x2 x, float bk) {
    if constexpr (FAST) {
        const f32x2 a = x * f32x2{-1.4426950408889634f, -1.4426950408889634f} + f32x2{bk, bk};
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2{1.f, 1.f};
        return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    } else {
        return f32x2{sigmoid_acc(x[0] + bk), sigmoid_acc(x[1] + bk)};
    }
}
template <bool FAST> __device__ __forceinline__ f32x2 tanh_b2(f32x2 x, float bk) {
    if constexpr (FAST) {
        const f32x2 a = x * f32x2{2.8853900817779268f, 2.8853900817779268f} + f32x2{bk, bk};
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2{1.f, 1.f};
        const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return f32x2{1.f, 1.f} - f32x2{2.f, 2.f} * r;
    } else {
        return f32x2{tanh_acc(x[0] + bk), tanh_acc(x[1] + bk)};
    }
}
template <bool FAST> __device__ __forceinline__ f32x2 tanh_f2(f32x2 x) { return tanh_b2<FAST>(x, 0.f); }
// the same with the pre-activation scaled: sigmoid(k' x + b), tanh(k' x + b); FAST: k = k' * (-log2 e | 2 log2 e), bk as above
template <bool FAST> __device__ __forceinline__ f32x2 sigmoid_k2(f32x2 x, float k, float bk) {
    const f32x2 a = x * f32x2{k, k} + f32x2{bk, bk};
    if constexpr (FAST) {
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2{1.f, 1.f};
        return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    } else {
        return f32x2{sigmoid_acc(a[0]), sigmoid_acc(a[1])};
    }
}
template <bool FAST> __device__ __forceinline__ f32x2 tanh_k2(f32x2 x, float k, float bk) {
    const f32x2 a = x * f32x2{k, k} + f32x2{bk, bk};
    if constexpr (FAST) {
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2{1.f, 1.f};
        const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return f32x2{1.f, 1.f} - f32x2{2.f, 2.f} * r;
    } else {
        return f32x2{tanh_acc(a[0]), tanh_acc(a[1])};
    }
}

// ... and with a bias per element
template <bool FAST> __device__ __forceinline__ f32x2 sigmoid_k2v(f32x2 x, float k, f32x2 bk) {
    const f32x2 a = x * f32x2{k, k} + bk;
    if constexpr (FAST) {
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2{1.f, 1.f};
        return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    } else {
        return f32x2{sigmoid_acc(a[0]), sigmoid_acc(a[1])};
    }
}
template <bool FAST> __device__ __forceinline__ f32x2 tanh_k2v(f32x2 x, float k, f32x2 bk) {
    const f32x2 a = x * f32x2{k, k} + bk;
    if constexpr (FAST) {
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2{1.f, 1.f};
        const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return f32x2{1.f, 1.f} - f32x2{2.f, 2.f} * r;
    } else {
        return f32x2{tanh_acc(a[0]), tanh_acc(a[1])};
    }
}

// LSTM cell with SHARED reciprocals (hardware-activation builds).  Arguments are the pre-scaled exponents the separate forms use
// (ai = -log2e (i + b_i), af, ao likewise; ag = 2 log2e (g + b_g)); with Ei = 2^ai = e^-i, Ef = e^-f, G = 2^ag = e^2g:
//     c' = sigmoid(f) c + sigmoid(i) tanh(g) = [c (1 + Ei)(G + 1) + (G - 1)(1 + Ef)] / [(1 + Ef)(1 + Ei)(G + 1)]      ONE v_rcp_f32
//     h  = sigmoid(o) tanh(c')               = (Gc - 1) / [(1 + Eo)(Gc + 1)],   Gc = e^2c'                           ONE v_rcp_f32
// 5 v_exp_f32 + 2 v_rcp_f32 per unit-step where the separate forms (dprnn.py:23-28 = torch's LSTM cell, one reciprocal per
// sigmoid / tanh) take 5 + 5; every factor is >= 1, so nothing cancels that the separate forms do not cancel too (G - 1 is
// tanh's own numerator).  Range: an exponential near 2^127 would overflow a denominator (Inf -> quotient NaN) or push its
// reciprocal into the flushed denormals (quotient 0): `dmax` follows the largest denominator and the caller recomputes a tile
// whose dmax is not below 2^100 with the separate forms -- pre-activations beyond about -60, never seen on the goldens.
__device__ __forceinline__ void lstm_cell_shared2(f32x2 ai, f32x2 af, f32x2 ag, f32x2 ao, f32x2 c, f32x2& cn, f32x2& h, float& dmax) {
    const f32x2 one = f32x2{1.f, 1.f};
    const f32x2 Ei = f32x2{__builtin_amdgcn_exp2f(ai[0]), __builtin_amdgcn_exp2f(ai[1])};
    const f32x2 Ef = f32x2{__builtin_amdgcn_exp2f(af[0]), __builtin_amdgcn_exp2f(af[1])};
    const f32x2 G = f32x2{__builtin_amdgcn_exp2f(ag[0]), __builtin_amdgcn_exp2f(ag[1])};
    const f32x2 Eo = f32x2{__builtin_amdgcn_exp2f(ao[0]), __builtin_amdgcn_exp2f(ao[1])};
    const f32x2 C = Ef + one;
    const f32x2 P = (Ei + one) * (G + one);
    const f32x2 den = P * C;
    const f32x2 num = c * P + (G - one) * C;
    cn = num * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2 acn = cn * f32x2{2.8853900817779268f, 2.8853900817779268f};
    const f32x2 Gc = f32x2{__builtin_amdgcn_exp2f(acn[0]), __builtin_amdgcn_exp2f(acn[1])};
    const f32x2 D = (Eo + one) * (Gc + one);
    h = (Gc - one) * f32x2{__builtin_amdgcn_rcpf(D[0]), __builtin_amdgcn_rcpf(D[1])};
    dmax = __builtin_fmaxf(dmax, __builtin_fmaxf(__builtin_fmaxf(den[0], den[1]), __builtin_fmaxf(D[0], D[1])));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// host-side error plumbing (api.hip)
void dprnn_set_error(const char* fmt, ...);
int dprnn_check_launch(const char* what);

// ---- saturation flags (api.hip: dprnn_saturation_begin / dprnn_saturation_flags).  The f16-pieces kernels multiply their fp32
// operands by a power of two and convert to f16; a value beyond the f16 range would be clamped.  Every such site tracks the
// largest |scaled value| it converts (as float bits: Inf and NaN patterns order above every finite value) and, at kernel end,
// raises flag `slot` of the calling thread's registered device buffer when the limit was crossed -- one ballot per wave, one
// atomic per wave that saw it.  The host checks the flags once per forward / backward and re-runs in a range-free arithmetic.
enum { DPRNN_SAT_RECUR_X = 0, DPRNN_SAT_PROJ_ROWS = 1 /* unused: |h| < 1 */, DPRNN_SAT_GEMM_ROWS = 2, DPRNN_SAT_BPTT_DA = 3 /* reported through 4 */, DPRNN_SAT_WGRAD = 4,
       DPRNN_SAT_NONFINITE = 5, DPRNN_SAT_SLOTS = 8 };
int* dprnn_sat_ptr();        // device int32[DPRNN_SAT_SLOTS] registered by the calling thread, or nullptr (detection off)
// Tracking costs one v_max3_f32 per two converted values (|.| is an operand modifier).  A float maximum skips NaN operands, so a NaN
// is not seen here: NaN / Inf INPUTS are flagged where every forward begins instead (bottleneck_prep_kernel: non-finite statistics
// of the encoder output raise DPRNN_SAT_NONFINITE), Inf is an ordinary "beyond the limit" value.
// DPRNN_SAT_TRACK (A/B builds only): 0 = no tracking, 1 = integer compare of the bit patterns (sees NaN, three times the work).
#ifndef DPRNN_SAT_TRACK
#define DPRNN_SAT_TRACK 2
#endif
__device__ __forceinline__ void sat_track(unsigned& m, float a, float b) {
#if DPRNN_SAT_TRACK == 1
    const unsigned ua = __float_as_uint(a) & 0x7fffffffu, ub = __float_as_uint(b) & 0x7fffffffu;
    m = ua > m ? ua : m;
    m = ub > m ? ub : m;
#elif DPRNN_SAT_TRACK == 2
    m = __float_as_uint(__builtin_fmaxf(__uint_as_float(m), __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))));
#endif
}
__device__ __forceinline__ void sat_track(unsigned& m, const f32x4& v) { sat_track(m, v[0], v[1]); sat_track(m, v[2], v[3]); }
// amax_dst (optional): the launch's largest tracked magnitude, as float bits, folded into a device word with one atomic per
// wave (adaptive activation scale below).
__device__ __forceinline__ void sat_commit(int* flags, int slot, unsigned m, float limit, unsigned* amax_dst = nullptr) {
    if (flags) {
        const bool over = m > __float_as_uint(limit);    // non-negative floats order like their bit patterns
        if (__builtin_amdgcn_ballot_w64(over) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags + slot, 1);
    }
    if (amax_dst) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
        if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(amax_dst, m);
    }
}

// ---- packed LSTM gate tape (training, f16-pieces forward + f16-pieces BPTT; flag bit on the gemm_mode argument of
// dprnn_rnn_fwd_train_ex / dprnn_rnn_bwd_ex).  The tape is the largest stream of a training step (5 floats per row, direction
// and hidden unit, written once and read once).  The activated gates live in [0, 1] (i, f, o) and [-1, 1] (g), so 24-bit fixed
// point holds them to 2^-25 (i, f, o: exact above 1/2) / 2^-24 (g) ABSOLUTE -- what fp32 gives near 1, which is where the gate
// derivatives i(1-i), 1-g^2 lose their bits anyway -- and the four of a (sequence, unit) fit three words instead of four:
//   w0 = qi | o[7:0] << 24,  w1 = qf | o[15:8] << 24,  w2 = qg | o[23:16] << 24;  qg = (g + 1) 2^23.   c stays fp32.
constexpr int DPRNN_TAPE_PACKED = 0x100;
__device__ __forceinline__ void tape_pack(float i, float f, float g, float o, unsigned& w0, unsigned& w1, unsigned& w2) {
    const unsigned qi = min((unsigned)__builtin_rintf(i * 16777216.f), 0xffffffu);
    const unsigned qf = min((unsigned)__builtin_rintf(f * 16777216.f), 0xffffffu);
    const unsigned qo = min((unsigned)__builtin_rintf(o * 16777216.f), 0xffffffu);
    const unsigned qg = min((unsigned)__builtin_rintf(__builtin_fmaf(g, 8388608.f, 8388608.f)), 0xffffffu);
    w0 = qi | (qo << 24);
    w1 = qf | ((qo << 16) & 0xff000000u);
    w2 = qg | ((qo << 8) & 0xff000000u);
}
__device__ __forceinline__ void tape_unpack(unsigned w0, unsigned w1, unsigned w2, float& i, float& f, float& g, float& o) {
    i = (float)(w0 & 0xffffffu) * 0x1p-24f;
    f = (float)(w1 & 0xffffffu) * 0x1p-24f;
    g = __builtin_fmaf((float)(w2 & 0xffffffu), 0x1p-23f, -1.f);
    o = (float)((w0 >> 24) | ((w1 >> 16) & 0xff00u) | ((w2 >> 8) & 0xff0000u)) * 0x1p-24f;
}

// ---- adaptive activation scale of the f16-pieces arithmetic (round 4; api.hip: dprnn_scale_begin / _slot / _update).
// Activations are multiplied by 2^s before they are split into f16 pieces; s was the constant 4 (|x| <= 4062).  A model whose
// residual stream runs higher (norm gains are unbounded, dprnn.py:76-77) would saturate on EVERY forward and pay the range-free
// repeat each time.  Now s follows the data: the calling thread registers a device record `state` of 2 x DPRNN_SCALE_SLOTS words
// -- [slot] = the largest |x| the conversion site `slot` has ever met (sticky, float; 256 at first: s = 4), [SLOTS + slot] = the
// largest it met in the forward in progress (float bits, atomicMax) -- and every tracked launch reads its scale from the first
// and leaves its maximum in the second; dprnn_scale_update folds the second into the first at the end of the forward.  So the
// forward that first meets large values still saturates and is repeated (engine._guarded), the next ones run under a scale that
// fits, with 8 x headroom: s = clamp(13 - exponent(amax), -8, 4), 2^s amax in [2^12, 2^13).  s never exceeds 4: models that stay
// below 512 compute exactly what they computed with the constant, run to run.  No record registered: s = 4, nothing tracked.
constexpr int DPRNN_SCALE_SLOTS = 32;
struct ScaleRef { const float* src; unsigned* dst; };
ScaleRef dprnn_scale_ref();                              // the calling thread's (record, current slot), or {nullptr, nullptr}
__host__ __device__ __forceinline__ int act_scale_exp(float amax) {
    int ex = 9;
    if (amax > 0.f && amax < 3.0e38f) frexpf(amax, &ex);               // amax = f 2^ex, f in [0.5, 1)
    const int s = 13 - ex;
    return s < -8 ? -8 : s > 4 ? 4 : s;
}
__device__ __forceinline__ float act_scale(const float* src) { return src ? ldexpf(1.f, act_scale_exp(*src)) : 16.f; }

// ---- packed-weight formats (api.hip).  The recurrence kernels read weights in a layout that depends on the cell and on the
// arithmetic mode the pack call ran under; every pack entry point records the format of the buffer it filled (keyed by the
// device address) and every launcher checks it: a mismatch is an error (rc DPRNN_EFORMAT), never numbers.
enum { DPRNN_PACK_FWD = 1, DPRNN_PACK_BWD = 2 };
void dprnn_pack_note(const void* packed, int kind, int cell, int mode, int ndir);
int dprnn_pack_expect(const void* packed, int kind, int cell, int mode, int ndir, const char* what);   // 0 or -3

// hipFuncSetAttribute is per device: launchers keep one bit per device ordinal
#include <atomic>
inline bool dprnn_attr_needed(std::atomic<unsigned long long>& mask, int* dev_out) {
    int dev = 0; (void)hipGetDevice(&dev);
    *dev_out = dev;
    return !(mask.load(std::memory_order_relaxed) & (1ull << (dev & 63)));
}
inline void dprnn_attr_done(std::atomic<unsigned long long>& mask, int dev) { mask.fetch_or(1ull << (dev & 63), std::memory_order_relaxed); }
