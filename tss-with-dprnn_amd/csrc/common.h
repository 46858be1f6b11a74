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
