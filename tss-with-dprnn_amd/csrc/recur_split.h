// Split gate-GEMM forms of the BiLSTM recurrence (dprnn_set_gemm_mode 1 / 3): what the kernels share -- operand splitting,
// MFMA wrappers, constants.
//
// The exact path (recur.hip) multiplies fp32 operands on v_mfma_f32_16x16x4_f32, which runs at the fp32 *vector* rate and --
// measured, profiles/r01_mfma_valu_coissue.txt -- shares that datapath with every VALU instruction.  The 16-bit MFMAs are 16x
// faster per FLOP and leave the vector ALU free, so these kernels write every fp32 operand as a sum of 16-bit pieces and sum
// cross products with fp32 accumulation inside the MFMA:
//     mode 1: three bf16 pieces, 6 products (v1w1, v1w2, v2w1, v1w3, v2w2, v3w1): product error ~2^-24, fp32's exponent range;
//     (mode 2, two bf16 pieces / 3 products with product error ~2^-16, was opt-in only and left the tree in round 3;)
//     mode 3: two f16 pieces of power-of-two-scaled operands, 3 products: product error ~2^-24 at half the work of mode 1 (below).
// Everything else (cell arithmetic, state, addressing, pending-norm loader) is the fp32 code of recur.hip.
//
// Weights are pre-split by dprnn_lstm_pack_split / dprnn_lstm_pack_f16 into [dir][wave 4][k-slab 8][gate 4][u 2][piece 3][lane 64]
// 16-byte fragments (lane (r, q) = unit row r, k = 32*slab + 8*q .. +7: the B operand of v_mfma_f32_16x16x32_{bf16,f16}).
// Files: recur_split.hip (pack kernels, the two-workgroups-per-CU 32-sequence kernel, the dispatcher), recur_split8.h (the
// eight-wave one-workgroup-per-CU kernel), recur_split8_bf16.hip / recur_split8_f16.hip (its instantiations, built in parallel).
#pragma once
#include "recur.h"
#include <stdlib.h>
#include <type_traits>

namespace rsplit {
using namespace recur;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(1)))* gfrag_ptr;

// f16 pieces: column j of weight tile (wave, u) is hidden unit 32 wave + 16 u + j (CONTIGUOUS units), so that with the weight
// fragment as the MFMA's row operand a lane's four accumulator elements are four adjacent units of one sequence: 8-byte LDS and
// 16-byte global stores of h (recur_split8.h, TRANS).  bf16 pieces (mode 1) keep the interleaved order 32 wave + 2 j + u.
#ifndef DPRNN_SPLIT8_TRANS
#define DPRNN_SPLIT8_TRANS 1
#endif
constexpr bool F16_UNITS_CONTIGUOUS = DPRNN_SPLIT8_TRANS;
constexpr int NSLAB = 8;                       // k-slabs of 32 over [x | h]
constexpr int PITCH = 528;                     // bytes per LDS row (256 bf16 + 16 B pad)

__device__ __forceinline__ unsigned bf16_bits(float v) {            // round-to-nearest-even bf16 of v, in the low 16 bits
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = __builtin_convertvector(f32x2{v, 0.f}, bf16x2);
    return __builtin_bit_cast(unsigned, p) & 0xFFFFu;
}
__device__ __forceinline__ float bf16_to_f32(unsigned bits16) { return __builtin_bit_cast(float, bits16 << 16); }

// v -> three bf16 pieces (bits), v ~= p1 + p2 + p3 with |p2| <= 2^-8 |v|, |p3| <= 2^-16 |v|
__device__ __forceinline__ void split3(float v, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = bf16_bits(v);
    const float r1 = v - bf16_to_f32(p1);
    p2 = bf16_bits(r1);
    const float r2 = r1 - bf16_to_f32(p2);
    p3 = bf16_bits(r2);
}

// (v0, v1) -> three words of packed bf16 pieces (low half = v0's piece, high half = v1's)
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
template <int NPC>
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = cvt_pk(v0, v1);
    const float a0 = v0 - __builtin_bit_cast(float, w1 << 16), a1 = v1 - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = cvt_pk(a0, a1);
    w3 = 0u;
    if constexpr (NPC == 3) {
        const float b0 = a0 - __builtin_bit_cast(float, w2 << 16), b1 = a1 - __builtin_bit_cast(float, w2 & 0xffff0000u);
        w3 = cvt_pk(b0, b1);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- f16 form (gemm mode 3): v -> two f16 pieces of S * v (11 significant bits each, so v1 + v2 carries 22+ bits and the
// three products v1w1 + v1w2 + v2w1 leave v2w2 ~ 2^-24 |v||w| behind -- the size of one fp32 product rounding), at HALF the
// MFMA work of the six bf16 products.  f16 has 5 exponent bits, so operands are scaled by powers of two into its range:
// activations by F16_XS = 2^4 (x clamped to +-4062; |h| < 1), weights by a per-LSTM 2^e that puts max|W| in [2^13, 2^14)
// (lstm_absmax_kernel); second pieces may be f16 subnormals, which v_cvt_pk_f16_f32 produces and the MFMA consumes
// without flushing (tools/probes/mfma_f16_denorm.hip).  The accumulators come out scaled by S_x * S_w, a power of two that the
// gate nonlinearity's first multiply-add removes (inv, read from the packed weights).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_XS = 16.f;
constexpr float F16_XMAX = 65000.f;                // clamp of S_x * x (f16 max 65504): finite results for any input
constexpr int F16_SCALE_WORD = 2 * 256;            // packed word of S_w (then 1 / (S_x S_w)): plane 2 of fragment 0 is unused in this form
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
// (s0, s1), already scaled -> two words of packed f16 pieces (low half = s0's piece, high half = s1's)
__device__ __forceinline__ void split2_pair_f16(float s0, float s1, unsigned& w1, unsigned& w2) {
    w1 = cvt_pk_f16(s0, s1);
    const f16x2 h = __builtin_bit_cast(f16x2, w1);
    w2 = cvt_pk_f16(s0 - (float)h[0], s1 - (float)h[1]);       // the differences are exact in fp32
}

}  // namespace rsplit
