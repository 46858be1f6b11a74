// OPTIONAL gate-GEMM mode of the BiLSTM recurrence: bf16-split operands on the bf16 MFMA, fp32 accumulate.
//
// Off by default (dprnn_set_gemm_mode).  The default path (recur.hip) multiplies exact fp32 operands on
// v_mfma_f32_16x16x4_f32, which runs at the fp32 *vector* rate and -- measured, profiles/r01_mfma_valu_coissue.txt --
// shares that datapath with every VALU instruction.  v_mfma_f32_16x16x32_bf16 is 16x faster per FLOP and leaves the
// vector ALU free, so this kernel writes every fp32 operand v as a sum of bf16 pieces v = v1 + v2 (+ v3) and sums
// the cross products of order <= 2^-16 with fp32 accumulation inside the MFMA:
//     mode 1 (6 products: v1w1, v1w2, v2w1, v1w3, v2w2, v3w1): product error ~2^-24 -- end to end the same
//            1.4e-7 rel-L2 against fp64 as the exact fp32 path (tools/split_precision_study.py),
//     mode 2 (3 products: v1w1, v1w2, v2w1):                   product error ~2^-16 -- 1.2e-6 rel-L2.
// Everything else (cell arithmetic, state, layout, addressing, pending-norm loader) is the fp32 code of recur.hip.
//
// Layout differences: x/h live in LDS as three bf16 planes [plane][seq][x(128)|h(128)] (row pitch 528 B so that the
// 16-byte fragment reads of 16 rows spread over the banks), single-buffered with two barriers per step; weights are
// pre-split by dprnn_lstm_pack_split into [dir][wave][k-slab 8][gate 4][u 2][piece 3][lane 64] 16-byte fragments
// (lane (r, q) = unit row r, k = 32*slab + 8*q .. +7: the B operand of v_mfma_f32_16x16x32_bf16).
#include "recur.h"
#include <stdlib.h>
#include <type_traits>
using namespace recur;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(1)))* gfrag_ptr;

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

// packed[dir][wave][slab][gate*2+u][piece][lane][4 u32] : bf16 piece `piece` of Wcat[gate*128 + 32*wave + 2*r + u][32*slab + 8*q + j]
// (a lane's two column tiles u = 0, 1 are the adjacent hidden units 2r, 2r+1: their h pieces share one 32-bit LDS store
// and their fp32 h one 64-bit global store)
__global__ void lstm_pack_split_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                       const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                       const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r,
                                       unsigned* __restrict__ packed, float* __restrict__ bsum, int ndir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (.., lane, dword) of piece 0
    const int per_dir = 4 * NSLAB * 8 * 64 * 4;
    if (idx >= ndir * per_dir) return;
    const int dw = idx & 3, lane = (idx >> 2) & 63, gu = (idx >> 8) & 7, slab = (idx >> 11) & 7;
    const int wave = (idx >> 14) & 3, dir = idx >> 16;
    const int gate = gu >> 1, u = gu & 1, r = lane & 15, q = lane >> 4;
    const int row = gate * HD + 32 * wave + 2 * r + u;      // lane r of a wave owns the ADJACENT units 2r, 2r+1 (u = 0, 1)
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 32 * slab + 8 * q + 2 * dw + half;            // 0..255 over [x | h]
        const float v = k < F ? wi[row * F + k] : wh[row * HD + (k - F)];
        unsigned p1, p2, p3;
        split3(v, p1, p2, p3);
        out[0] |= p1 << (16 * half); out[1] |= p2 << (16 * half); out[2] |= p3 << (16 * half);
    }
    const size_t base = ((((size_t)(dir * 4 + wave) * NSLAB + slab) * 8 + gu) * 3) * 256 + lane * 4 + dw;
#pragma unroll
    for (int p = 0; p < 3; ++p) packed[base + (size_t)p * 256] = out[p];
    if (idx < ndir * 4 * HD) {
        const int d = idx / (4 * HD), n = idx % (4 * HD);
        bsum[idx] = (d ? b_ih_r : b_ih)[n] + (d ? b_hh_r : b_hh)[n];
    }
}

// S_w of one LSTM: the power of two that puts max|W_ih, W_hh| (all directions) in [2^13, 2^14).  ABSMAX_BLOCKS blocks leave
// their partial maxima in the unused third plane of the packed buffer; every thread of the pack kernel folds them (uniform
// loads), so there is no atomic, no memset and no one-block pass on the critical path of a forward.
constexpr int ABSMAX_BLOCKS = 32;
__global__ __launch_bounds__(256)
void lstm_absmax_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, const float* __restrict__ w_ih_r,
                        const float* __restrict__ w_hh_r, float* __restrict__ packed_f) {
    __shared__ float part[4];
    const float* src[4] = {w_ih, w_hh, w_ih_r, w_hh_r};
    float m = 0.f;
    for (int k = 0; k < 4; ++k) {
        if (!src[k]) continue;
        const f32x4* p = reinterpret_cast<const f32x4*>(src[k]);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 4 * HD * F / 4; i += ABSMAX_BLOCKS * 256) {   // W_ih [4H][F], W_hh [4H][H]: same size (F == H)
            const f32x4 v = p[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) packed_f[F16_SCALE_WORD + 2 + blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}
__device__ __forceinline__ float f16_weight_scale(const float* packed_f) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < ABSMAX_BLOCKS; ++i) m = fmaxf(m, packed_f[F16_SCALE_WORD + 2 + i]);
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);                               // m = f * 2^ex, f in [0.5, 1)
    ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
    return ldexpf(1.f, 14 - ex);
}

// same fragment layout as lstm_pack_split_kernel, planes 0 and 1 = the two f16 pieces of S_w * W (plane 2 stays unused)
__global__ void lstm_pack_f16_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                     const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                     const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                     const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r,
                                     unsigned* __restrict__ packed, float* __restrict__ bsum, int ndir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_dir = 4 * NSLAB * 8 * 64 * 4;
    if (idx >= ndir * per_dir) return;
    const float sw = f16_weight_scale(reinterpret_cast<const float*>(packed));
    if (idx == 0) {                                     // what the recurrence reads: S_w and 1 / (S_x S_w)
        reinterpret_cast<float*>(packed)[F16_SCALE_WORD] = sw;
        reinterpret_cast<float*>(packed)[F16_SCALE_WORD + 1] = 1.f / (F16_XS * sw);
    }
    const int dw = idx & 3, lane = (idx >> 2) & 63, gu = (idx >> 8) & 7, slab = (idx >> 11) & 7;
    const int wave = (idx >> 14) & 3, dir = idx >> 16;
    const int gate = gu >> 1, u = gu & 1, r = lane & 15, q = lane >> 4;
    const int row = gate * HD + 32 * wave + 2 * r + u;
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    float v[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 32 * slab + 8 * q + 2 * dw + half;
        v[half] = sw * (k < F ? wi[row * F + k] : wh[row * HD + (k - F)]);
    }
    unsigned w1, w2;
    split2_pair_f16(v[0], v[1], w1, w2);
    const size_t base = ((((size_t)(dir * 4 + wave) * NSLAB + slab) * 8 + gu) * 3) * 256 + lane * 4 + dw;
    packed[base] = w1;
    packed[base + 256] = w2;
    if (idx < ndir * 4 * HD) {
        const int d = idx / (4 * HD), n = idx % (4 * HD);
        bsum[idx] = (d ? b_ih_r : b_ih)[n] + (d ? b_hh_r : b_hh)[n];
    }
}

template <int NT, int NP, bool FAST>
__global__ __launch_bounds__(256, (NT <= 3 ? 2 : 1))
void bilstm_split_kernel(RecurArgs a) {
    constexpr int NS = 16 * NT, NXV = NS / 8;
    constexpr int NPC = NP == 6 ? 3 : 2;               // pieces actually needed per operand
    constexpr int PLANE = NS * PITCH;                  // bytes
    extern __shared__ __attribute__((aligned(16))) char img[];       // [3][NS][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NS;
    const int T = a.T;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const long long row0 = rowbase(seq0 < a.nseq ? seq0 : 0);
    const int hpitch = a.ndir * HD;
    unsigned xoff[NXV];
    int xlds[NXV];
    const int c4 = tid & 31;
#pragma unroll
    for (int m = 0; m < NXV; ++m) {
        const int row = (tid >> 5) + 8 * m;
        const int seq = seq0 + row;
        xoff[m] = (unsigned)((rowbase(seq < a.nseq ? seq : seq0) - row0) * F + c4 * 4) * 4u;
        xlds[m] = row * PITCH + c4 * 8;                // 4 bf16 = 8 bytes
    }
    unsigned hoff[NT][4];
    bool hval[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int seq = seq0 + 16 * nt + 4 * q + e;
            hval[nt][e] = seq < a.nseq;
            hoff[nt][e] = (unsigned)((rowbase(hval[nt][e] ? seq : seq0) - row0) * hpitch + dir * HD + 32 * wave + 2 * r) * 4u;
        }
    const bool fuse = a.Z != nullptr;
    f32x4 gam = {1.f, 1.f, 1.f, 1.f}, bet = {0.f, 0.f, 0.f, 0.f};
    float mu[NXV], rs[NXV];
#pragma unroll
    for (int m = 0; m < NXV; ++m) { mu[m] = 0.f; rs[m] = 0.f; }
    if (fuse) {
        gam = *reinterpret_cast<const f32x4*>(a.gamma + 4 * c4);
        bet = *reinterpret_cast<const f32x4*>(a.beta + 4 * c4);
#pragma unroll
        for (int m = 0; m < NXV; ++m) {
            const int seq = seq0 + (tid >> 5) + 8 * m;
            const int b = (seq < a.nseq ? seq : seq0) / a.seqs_per_utt;
            const double cnt = (double)a.rows_per_utt * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            mu[m] = (float)mean;
            rs[m] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
    }
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };

    float bias[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float b = a.bsum[dir * 4 * HD + g * HD + 32 * wave + 2 * r + u];
            bias[g][u] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
        }

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const gfrag_ptr wsl = (gfrag_ptr)(reinterpret_cast<const u32x4*>(a.Wp) + (size_t)(dir * 4 + wave_u) * NSLAB * 8 * 3 * 64) + lane;

    // stage one time step of x (with the pending norm applied) as bf16 pieces into the image
    auto stage_x = [&](long long t) __attribute__((always_inline)) {
        const char* xs = reinterpret_cast<const char*>(a.X + tbase(t) * F);
        const char* zs = reinterpret_cast<const char*>(a.Z + tbase(t) * F);
        char* os = reinterpret_cast<char*>(a.Xout + tbase(t) * F);
#pragma unroll
        for (int m = 0; m < NXV; ++m) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xs + xoff[m]);
            if (fuse) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(zs + xoff[m]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((z[e] - mu[m]) * rs[m] * gam[e] + bet[e]);
                if (dir == 0) *reinterpret_cast<f32x4*>(os + xoff[m]) = v;
            }
            unsigned p[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(v[e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc)
                *reinterpret_cast<u32x2*>(img + pc * PLANE + xlds[m]) = u32x2{p[pc][0] | (p[pc][1] << 16), p[pc][2] | (p[pc][3] << 16)};
        }
    };

    // ---- prologue: x(t0) pieces, h = 0
    stage_x(dir ? (T - 1) : 0);
    for (int i = tid; i < 3 * NS * 64; i += 256) {     // h region of every plane: 128 bf16 = 64 dwords per row (NS rows)
        const int pl = i / (NS * 64), rem = i % (NS * 64);
        *reinterpret_cast<unsigned*>(img + pl * PLANE + (rem >> 6) * PITCH + 256 + 4 * (rem & 63)) = 0u;
    }
    f32x4 c[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int u = 0; u < 2; ++u) c[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // B fragments of one k-slab: [gate][u][piece]; slab s+1 of a gate is requested as soon as slab s of it is consumed
    u32x4 bfr[4][2][NPC];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bfr[g][u][pc] = wsl[((0 * 8 + g * 2 + u) * 3 + pc) * 64];
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        f32x4 acc[NT][4][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[nt][g][u] = f32x4{0.f, 0.f, 0.f, 0.f};

        const char* const Aim = img + r * PITCH + 16 * q;
#pragma unroll 1
        for (int s = 0; s < NSLAB; ++s) {
            u32x4 afr[NT][NPC];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)
                    afr[nt][pc] = *reinterpret_cast<const u32x4*>(Aim + pc * PLANE + nt * 16 * PITCH + 64 * s);
            const int sn = (s + 1) & 7;               // wraps to slab 0 of the next step (same bytes every step)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // products outermost: consecutive MFMAs go to different accumulators (no dependent-issue stalls),
                // smallest cross terms first
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int pr = (NP == 6 ? 0 : 3); pr < 6; ++pr)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt][g][u] = mfma_bf16(afr[nt][PA[pr]], bfr[g][u][PB[pr]], acc[nt][g][u]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) bfr[g][u][pc] = wsl[((sn * 8 + g * 2 + u) * 3 + pc) * 64];
            }
        }
        __syncthreads();                              // every wave is done reading the image of this step

        // ---- LSTM cell, lane-local; h -> bf16 pieces in the image (next step's operand) and fp32 to HBM
        char* const hs = reinterpret_cast<char*>(a.H + tbase(t) * hpitch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ig = sigmoid_b<FAST>(acc[nt][0][u][e], bias[0][u]);
                    const float fg = sigmoid_b<FAST>(acc[nt][1][u][e], bias[1][u]);
                    const float gg = tanh_b<FAST>(acc[nt][2][u][e], bias[2][u]);
                    const float og = sigmoid_b<FAST>(acc[nt][3][u][e], bias[3][u]);
                    const float cn = fg * c[nt][u][e] + ig * gg;
                    c[nt][u][e] = cn;
                    const float h = og * tanh_f<FAST>(cn);
                    unsigned p1, p2, p3;
                    split3(h, p1, p2, p3);
                    char* d = img + (16 * nt + 4 * q + e) * PITCH + 2 * (HD + 32 * wave + 2 * r + u);
                    *reinterpret_cast<unsigned short*>(d) = (unsigned short)p1;
                    *reinterpret_cast<unsigned short*>(d + PLANE) = (unsigned short)p2;
                    if (NPC == 3) *reinterpret_cast<unsigned short*>(d + 2 * PLANE) = (unsigned short)p3;
                    if (hval[nt][e]) *reinterpret_cast<float*>(hs + hoff[nt][e] + 4 * u) = h;
                }
        if (more) stage_x(dir ? (t - 1) : (t + 1));
        __syncthreads();
    }
}

template <int NT, int NP, bool FAST>
int launch_split(const RecurArgs& a, hipStream_t st) {
    constexpr int NS = 16 * NT;
    const size_t lds = (size_t)3 * NS * PITCH;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split_kernel<NT, NP, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    dim3 grid((a.nseq + NS - 1) / NS, a.ndir);
    hipLaunchKernelGGL((bilstm_split_kernel<NT, NP, FAST>), grid, dim3(256), lds, st, a);
    return dprnn_check_launch("bilstm_split_kernel");
}


// ------------------------------------------------------------------------------------------------------------------------
// WIDE form: ONE 4-wave workgroup per CU owning NSR = 48 .. 104 sequences of one direction.
//
// The two-per-CU form above pulls a direction's 786 KB of split weights from L2 once per 32 sequences and step and is
// bound by that stream (~65 GB/s per CU, profiles/r01_split_modes.txt) with the bf16 MFMAs busy 40 % of the time.  Here
// a CU makes ONE pass over the weights per step for up to 104 sequences, which puts the stream below the matrix time
// (786 KB per 15-18 us at 96-104 sequences).  What that takes:
//   * image: x/h as three bf16 planes [plane][seq][256] WITHOUT row padding (104 rows x 3 planes x 512 B = 156 KB of the
//     160 KB); bank conflicts are avoided by XOR-ing the 16-byte chunk index of a row with (row & 15) instead -- the 16
//     lanes of every ds_read_b128 service group then touch 16 different bank quads (MI355X guide, LDS table);
//   * registers: one wave per SIMD (512 VGPRs): 16*NT x 128 gate accumulators and the cell state of the wave's 32 hidden
//     units, the B fragments of one k-slab (double-buffered by slab while NT <= 5, else re-requested piece by piece as
//     they die) and two A-fragment sets (one piece of all NT tiles each);
//   * per-row global offsets live in a small LDS table instead of 41 VGPRs; rows that do not exist carry an offset
//     beyond the buffer range (stores dropped by the hardware) and are never written to the image;
//   * a last tile of 8 rows (NSR = 104) re-reads existing rows for its missing ones; their results are never stored.
// Product order per k-slab (A piece, W piece): (1,3) (1,2) (2,2) (1,1) (2,1) (3,1) -- every term of order <= 2^-16.
// DIAG (stamped diagnostic builds only, wrong results on purpose): 1 = weight fragments loaded once (no L2 stream),
// 2 = A fragments loaded once per step (no LDS operand traffic), 3 = both: what the MFMA phase costs without its loads.
template <int NSR, int NP, bool FAST, bool FUSE, bool STAMP = false, int DIAG = 0>
__global__ __launch_bounds__(256, 1)
void bilstm_split_wide_kernel(RecurArgs a) {
    constexpr int NT = (NSR + 15) / 16;
    constexpr int LASTROWS = NSR - 16 * (NT - 1);      // rows of the last 16-row tile that exist (16 or 8)
    static_assert(NSR % 8 == 0 && (LASTROWS == 16 || LASTROWS == 8), "NSR must be a multiple of 8");
    constexpr int NXV = NSR / 8;                       // float4 of x per thread and step
    constexpr int NPC = NP == 6 ? 3 : 2;
    constexpr int PLANE = NSR * 512;                   // bytes
    constexpr bool BDB = NT <= 4;
    // once-touched activation streams (x, z in; h, x' out) are marked non-temporal so that they do not displace the weight
    // slices, which every CU re-reads each step, from the XCD's L2 (DIAG & 4: default policy, for A/B timing)
    constexpr int STREAM_AUX = (DIAG & 4) ? 0 : 2;
    constexpr int NBUF = BDB ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char img[];       // [NPC][NSR][512 B] + row tables
    unsigned* const rowoff_x = reinterpret_cast<unsigned*>(img + NPC * PLANE);     // [NSR]  byte offset of the row in X
    unsigned* const rowoff_h = rowoff_x + NSR;                                     // [16*NT] byte offset in H, or OOB
    float* const musig = reinterpret_cast<float*>(rowoff_h + 16 * NT);             // [NSR][2] (mean, rstd) of the pending norm
    float* const gb = musig + 2 * NSR;                                             // [2][128] gamma, beta of the pending norm

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NSR;
    const int T = a.T;
    constexpr unsigned OOB = 0x80000000u;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const long long row0 = rowbase(seq0 < a.nseq ? seq0 : 0);
    const int hpitch = a.ndir * HD;
    for (int j = tid; j < 16 * NT; j += 256) {
        const int seq = seq0 + j;
        const bool valid = j < NSR && seq < a.nseq;
        const long long ro = rowbase(valid ? seq : (seq0 < a.nseq ? seq0 : 0)) - row0;
        if (j < NSR) rowoff_x[j] = (unsigned)(ro * F * 4);          // rows past nseq alias the first row: loads stay in range
        rowoff_h[j] = valid ? (unsigned)(ro * hpitch * 4) : OOB;
    }
    const int c4 = tid & 31, xrow0 = tid >> 5;
    // image addresses of this thread's x quads: row xrow0 + 8m, chunk (c4>>1) ^ (row & 15), half (c4 & 1)
    const int xl_even = xrow0 * 512 + (((c4 >> 1) ^ xrow0) << 4) + ((c4 & 1) << 3);
    const int xl_odd = xrow0 * 512 + (((c4 >> 1) ^ xrow0 ^ 8) << 4) + ((c4 & 1) << 3);

    constexpr bool fuse = FUSE;                        // pending norm of the previous path folded into the x loader
    // pending norm: per-row (mean, rstd) -- the row's utterance is fixed for the whole kernel -- and gamma / beta live in LDS
    // tables (a few ds_reads per staged quad instead of 34 registers held through the whole kernel)
    if (fuse) {
        for (int j = tid; j < NSR; j += 256) {
            const int seq = seq0 + j;
            const int b = (seq < a.nseq ? seq : (seq0 < a.nseq ? seq0 : 0)) / a.seqs_per_utt;
            const double cnt = (double)a.rows_per_utt * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            musig[2 * j] = (float)mean;
            musig[2 * j + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
        if (tid < 128) { gb[tid] = a.gamma[tid]; gb[128 + tid] = a.beta[tid]; }
    }
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };

    float bias[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float b = a.bsum[dir * 4 * HD + g * HD + 32 * wave + 2 * r + u];
            bias[g][u] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
        }

    // weights: the wave's slice as a buffer resource; fragment (slab, gate*2+u, piece) at scalar offset ((slab*8+gu)*3+piece) KB
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(reinterpret_cast<const u32x4*>(a.Wp) + (size_t)(dir * 4 + wave_u) * NSLAB * 8 * 3 * 64), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16u;
    auto wload = [&](int slab, int gu, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)(((slab * 8 + gu) * 3 + pc) * 1024), 0);
    };
    constexpr unsigned RANGE = 0x7fffffffu;
    auto rsrc_at = [&](const float* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, RANGE, 0x00020000);
    };

    // x of one time step goes into the image as bf16 pieces (pending norm applied), in chunks of XCH quads per thread so
    // that the cell phase can keep a few chunks in flight instead of all NXV (registers): x_issue requests chunk k,
    // x_commit splits and stores it
    constexpr int XCH = (NXV + NT - 1) / NT;           // quads per chunk: one chunk per 16-row cell tile
    constexpr int NCHUNK = (NXV + XCH - 1) / XCH;
    f32x4 xv[NXV], zv[NXV];
    unsigned xo[NXV];
    auto x_issue = [&](long long t, int k) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
#pragma unroll
        for (int m = k * XCH; m < (k + 1) * XCH && m < NXV; ++m) {
            xo[m] = rowoff_x[xrow0 + 8 * m] + c4 * 16u;
            xv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[m], 0, STREAM_AUX));
            if (fuse) zv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo[m], 0, STREAM_AUX));
        }
    };
    // `live` false (after the last step): the same code runs on the current step's rows again -- no branch inside the step
    // loop, the register allocator does badly across them -- and only the x' store is suppressed, by an out-of-range offset
    auto x_commit = [&](long long t, int k, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
#pragma unroll
        for (int m = k * XCH; m < (k + 1) * XCH && m < NXV; ++m) {
            f32x4 v = xv[m];
            if (fuse) {
                const float mu = musig[2 * (xrow0 + 8 * m)], rs = musig[2 * (xrow0 + 8 * m) + 1];
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zv[m][e] - mu) * rs * gam[e] + bet[e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && dir == 0) ? xo[m] : OOB, 0, STREAM_AUX);
            }
            unsigned pa[3], pb[3];
            split3_pair<NPC>(v[0], v[1], pa[0], pa[1], pa[2]);
            split3_pair<NPC>(v[2], v[3], pb[0], pb[1], pb[2]);
            char* d = img + ((m & 1) ? xl_odd : xl_even) + m * 4096;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x2*>(d + pc * PLANE) = u32x2{pa[pc], pb[pc]};
        }
    };

    __syncthreads();                                   // row tables are in place
    // ---- prologue: x(t0) pieces, h = 0
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) { x_issue(dir ? (T - 1) : 0, k); x_commit(dir ? (T - 1) : 0, k, true); }
    for (int i = tid; i < NPC * NSR * 64; i += 256) {  // h half of every row (chunks 16..31 map onto themselves under the swizzle)
        const int pl = i / (NSR * 64), rem = i % (NSR * 64);
        *reinterpret_cast<unsigned*>(img + pl * PLANE + (rem >> 6) * 512 + 256 + 4 * (rem & 63)) = 0u;
    }
    f32x4 c[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int u = 0; u < 2; ++u) c[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragment addresses: lane (r, q) of tile nt reads row 16 nt + r, chunk (4 slab + q) ^ r
    const int rl = (LASTROWS == 16 || r < LASTROWS) ? r : (r - LASTROWS);      // last tile: rows that exist
    const int abase = r * 512 + ((q ^ (r & 3)) << 4);
    const int abase_last = (16 * (NT - 1) + rl) * 512 + ((q ^ (rl & 3)) << 4);
    int sl[4], sl_last[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sl[j] = (j ^ ((r >> 2) & 3)) << 6; sl_last[j] = (j ^ ((rl >> 2) & 3)) << 6; }
    auto aload = [&](u32x4 (&af)[NT], int pc, int s4, int hs) __attribute__((always_inline)) {
        if ((DIAG & 2) != 0 && (s4 != 0 || hs != 0 || pc > 1)) return;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int off = (nt == NT - 1 && LASTROWS != 16) ? (abase_last + sl_last[s4]) : (nt * 8192 + abase + sl[s4]);
            af[nt] = *reinterpret_cast<const u32x4*>(img + pc * PLANE + off + hs * 256);
        }
    };
    // elements (row 16 nt + 4q + e, units 32 wave + 2r, + 1) of h in the image: one 32-bit word per piece, + nt*8192 + piece*PLANE.
    // A last tile of 8 rows: its lanes q >= 2 hold exact copies of rows 0..7 (they read those rows as A operands and start
    // from the same zero state), so they write the same values to the same addresses as lanes q - 2 -- no predication.
    int hb[4], hb_last[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hb[e] = (4 * q + e) * 512 + ((16 | ((wave ^ q) << 2) | ((r >> 2) ^ e)) << 4) + 4 * (r & 3);
        const int ql = LASTROWS == 16 ? q : (q & 1);
        hb_last[e] = (16 * (NT - 1) + 4 * ql + e) * 512 + ((16 | ((wave ^ ql) << 2) | ((r >> 2) ^ e)) << 4) + 4 * (r & 3);
    }
    const unsigned hcol = (unsigned)(dir * HD + 32 * wave + 2 * r) * 4u;

    u32x4 bw[NBUF][4][2][NPC];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bw[0][g][u][pc] = wload(0, g * 2 + u, pc);
    __syncthreads();

    unsigned long long st_t0 = 0, st_r0 = 0, st_mfma = 0, st_cell = 0;
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int step = 0; step < T; ++step) {
        unsigned long long st_a = 0, st_b = 0;
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        f32x4 acc[NT][4][2];                           // the first product of a step starts every accumulator from an inline zero
        u32x4 fa[NT], fb[NT];                          // two A-fragment sets (one piece of every tile each)
        auto pass = [&](const u32x4 (&af)[NT], int buf, int wpc, bool first = false) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][g][u] = mfma_bf16(af[nt], bw[buf][g][u][wpc], first ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt][g][u]);
        };
        auto wreload = [&](int sn, int buf, int wpc) __attribute__((always_inline)) {     // W piece wpc of slab sn
            if constexpr ((DIAG & 1) != 0) return;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) bw[buf][g][u][wpc] = wload(sn, g * 2 + u, wpc);
        };
        // one k-slab; P holds piece 0 of this slab on entry when `have_p`, else it is loaded here
        // The weights of slab 0 are NOT requested from the last slab of a step: they would occupy 96 registers through the
        // whole cell phase, whose own working set then spills; the cell phase requests them before its last tile instead.
        auto slab = [&](u32x4 (&P)[NT], u32x4 (&Q)[NT], int s4, int hs, bool have_p, bool prefetch) __attribute__((always_inline)) {
            const int s = 4 * hs + s4, sn = (s + 1) & 7;
            const int cur = BDB ? (s4 & 1) : 0, nxt = BDB ? (cur ^ 1) : 0;
            const bool wnext = !(s4 == 3 && hs == 1);  // uniform
            if (!have_p) aload(P, 0, s4, hs);
            aload(Q, 1, s4, hs);
            if constexpr (BDB) {                       // all of the next slab's weights, in the order they are used
                if (wnext) {
#pragma unroll
                    for (int pc = NPC - 1; pc >= 0; --pc) wreload(sn, nxt, pc);
                }
            }
            // sched_barrier: the machine scheduler would otherwise hoist every LDS / weight load of the unrolled slabs to the
            // top (it sees 512 registers) and spill hundreds of them; the loads stay where the pipeline wants them
            auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
            fence();
            if constexpr (NP == 6) {
                pass(P, cur, 2, s4 == 0 && hs == 0); if constexpr (!BDB) { if (wnext) wreload(sn, 0, 2); }
                fence();
                pass(P, cur, 1);
                fence();
                pass(Q, cur, 1); if constexpr (!BDB) { if (wnext) wreload(sn, 0, 1); }
                fence();
                pass(P, cur, 0);
                fence();
                aload(P, 2, s4, hs);                   // piece 3 replaces piece 1 (done)
                pass(Q, cur, 0);
                fence();
                if (prefetch) aload(Q, 0, s4 + 1, hs); // piece 1 of the next slab replaces piece 2 (done)
                pass(P, cur, 0); if constexpr (!BDB) { if (wnext) wreload(sn, 0, 0); }
                fence();
            } else {
                pass(P, cur, 1, s4 == 0 && hs == 0); if constexpr (!BDB) { if (wnext) wreload(sn, 0, 1); }
                fence();
                pass(P, cur, 0);
                fence();
                if (prefetch) aload(P, 0, s4 + 1, hs);
                pass(Q, cur, 0); if constexpr (!BDB) { if (wnext) wreload(sn, 0, 0); }
                fence();
            }
        };
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {               // x half (slabs 0..3), h half (slabs 4..7); straight-line code: no branches
            if constexpr (NP == 6) {                   // the prefetched piece 1 lands in Q: roles alternate
                slab(fa, fb, 0, hs, false, true);
                slab(fb, fa, 1, hs, true, true);
                slab(fa, fb, 2, hs, true, true);
                slab(fb, fa, 3, hs, true, false);
            } else {                                   // the prefetched piece 1 lands in P
                slab(fa, fb, 0, hs, false, true);
                slab(fa, fb, 1, hs, true, true);
                slab(fa, fb, 2, hs, true, true);
                slab(fa, fb, 3, hs, true, false);
            }
        }
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); st_mfma += st_b - st_a; }
        __syncthreads();                               // every wave is done reading the image of this step

        // ---- LSTM cell, lane-local; h -> bf16 pieces in the image (next step's operand) and fp32 to HBM.  The next step's x
        // is requested XDEPTH cell tiles ahead of the point where it is split into the image (its latency hides behind the
        // cell arithmetic; only XDEPTH chunks of it are in registers at a time).
        constexpr int XDEPTH = 2;
        const __amdgpu_buffer_rsrc_t hrs = rsrc_at(a.H + tbase(t) * hpitch);
        const long long tn = more ? (dir ? (t - 1) : (t + 1)) : t;       // after the last step: harmless re-staging of x_t
#pragma unroll
        for (int k = 0; k < XDEPTH - 1 && k < NCHUNK; ++k) x_issue(tn, k);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt + XDEPTH - 1 < NCHUNK) x_issue(tn, nt + XDEPTH - 1);
            // next step's first k-slab of weights (buffer 0), pieces in the order of use: at once when the weights are
            // double-buffered (buffer 0 is free through the whole cell phase), else before the last tile (registers)
            if (nt == (BDB ? 0 : NT - 1)) {
#pragma unroll
                for (int pc = NPC - 1; pc >= 0; --pc) wreload(0, 0, pc);
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned ho[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ho[e] = rowoff_h[16 * nt + 4 * q + e] + hcol;
            float hv[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {       // two rows per instruction where the arithmetic allows (v_pk_*)
                    const f32x2 ig = sigmoid_b2<FAST>(f32x2{acc[nt][0][u][2 * p2], acc[nt][0][u][2 * p2 + 1]}, bias[0][u]);
                    const f32x2 fg = sigmoid_b2<FAST>(f32x2{acc[nt][1][u][2 * p2], acc[nt][1][u][2 * p2 + 1]}, bias[1][u]);
                    const f32x2 gg = tanh_b2<FAST>(f32x2{acc[nt][2][u][2 * p2], acc[nt][2][u][2 * p2 + 1]}, bias[2][u]);
                    const f32x2 og = sigmoid_b2<FAST>(f32x2{acc[nt][3][u][2 * p2], acc[nt][3][u][2 * p2 + 1]}, bias[3][u]);
                    const f32x2 cn = fg * f32x2{c[nt][u][2 * p2], c[nt][u][2 * p2 + 1]} + ig * gg;
                    const f32x2 h2 = og * tanh_f2<FAST>(cn);
                    c[nt][u][2 * p2] = cn[0]; c[nt][u][2 * p2 + 1] = cn[1];
                    hv[u][2 * p2] = h2[0]; hv[u][2 * p2 + 1] = h2[1];
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned w1, w2, w3;
                split3_pair<NPC>(hv[0][e], hv[1][e], w1, w2, w3);
                char* d = img + (nt == NT - 1 ? hb_last[e] : hb[e] + nt * 8192);
                *reinterpret_cast<unsigned*>(d) = w1;
                *reinterpret_cast<unsigned*>(d + PLANE) = w2;
                if (NPC == 3) *reinterpret_cast<unsigned*>(d + 2 * PLANE) = w3;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(hv[0][e]), __float_as_uint(hv[1][e])}, hrs, ho[e], 0, STREAM_AUX);
            }
            if (nt < NCHUNK) x_commit(tn, nt, more);
            __builtin_amdgcn_sched_barrier(0);         // keep the x chunks (and their registers) in this order
        }
#pragma unroll
        for (int k = NT; k < NCHUNK; ++k) x_commit(tn, k, more);
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_cell += __builtin_amdgcn_s_memtime() - st_b; __builtin_amdgcn_sched_barrier(0); }
        __syncthreads();
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;          // shader cycles, whole step loop
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;      // 100 MHz ticks
            o[2] = st_mfma;                                       // cycles in the MFMA phase (wave 0)
            o[3] = st_cell;                                       // cycles in cell update + x staging, before the barrier
            o[4] = st_r0;
            o[5] = __builtin_amdgcn_s_memrealtime();
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            o[6] = xcc; o[7] = hwid;
        }
    }
}

template <int NSR, int NP, bool FAST, bool FUSE, bool STAMP = false, int DIAG = 0>
int launch_split_wide(const RecurArgs& a, hipStream_t st) {
    constexpr int NT = (NSR + 15) / 16;
    const size_t lds = (size_t)(NP == 6 ? 3 : 2) * NSR * 512 + (size_t)(NSR + 16 * NT + 2 * NSR + 256) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split_wide_kernel<NSR, NP, FAST, FUSE, STAMP, DIAG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split_wide): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    dim3 grid((a.nseq + NSR - 1) / NSR, a.ndir);
    hipLaunchKernelGGL((bilstm_split_wide_kernel<NSR, NP, FAST, FUSE, STAMP, DIAG>), grid, dim3(256), lds, st, a);
    return dprnn_check_launch("bilstm_split_wide_kernel");
}

// ------------------------------------------------------------------------------------------------------------------------
// OVERLAPPED form (NSR = 48 or 64, one 4-wave workgroup per CU): the cell update is vector work -- ~45 instructions per
// (sequence, unit), half as long as the step's bf16 MFMAs when it runs on its own (profiles/r02_split_wide.txt) -- and a bf16
// MFMA leaves half of the SIMD's issue slots free.  The x half of the gate GEMM does not depend on h, so a step is cut
// differently here, with TWO accumulator sets (2 x 16 NT x 128 values, all 256 AGPRs at NT = 4):
//     phase 1   acc[cur] += h_{t-1} W_hh   (MFMA)   beside   x_{t+1} -> bf16 pieces -> image        (vector)
//     barrier   (h_{t-1} no longer read; x_{t+1} in place)
//     phase 2   acc[nxt]  = x_{t+1} W_ih   (MFMA)   beside   cell(t) from acc[cur]: c, h_t -> image, HBM   (vector)
//     barrier   (h_t in place; x_{t+1} no longer read)
// sched_group_barrier pins the instruction mix (one MFMA, then two vector instructions, ...): the hardware issues a wave's
// instructions in order, so only an interleaved stream overlaps the two pipes.  Weight fragments (single-buffered, a piece
// is re-requested for the next k-slab when its last product of this slab is done) stream continuously across both phases.
#define DPRNN_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
template <int NT, int NP, bool FAST, bool FUSE, bool STAMP = false>
__global__ __launch_bounds__(256, 1)
void bilstm_split_ov_kernel(RecurArgs a) {
    static_assert(NT == 3 || NT == 4, "two accumulator sets fit the register file for 48 or 64 sequences");
    constexpr int NSR = 16 * NT;
    constexpr int NXV = NSR / 8;                       // float4 of x per thread and step
    constexpr int NPC = NP == 6 ? 3 : 2;
    constexpr int NPASS = NP == 6 ? 6 : 3;
    constexpr int PLANE = NSR * 512;                   // bytes
    extern __shared__ __attribute__((aligned(16))) char img[];       // [NPC][NSR][512 B] + row tables
    unsigned* const rowoff_x = reinterpret_cast<unsigned*>(img + NPC * PLANE);
    unsigned* const rowoff_h = rowoff_x + NSR;
    float* const musig = reinterpret_cast<float*>(rowoff_h + NSR);
    float* const gb = musig + 2 * NSR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NSR;
    const int T = a.T;
    constexpr unsigned OOB = 0x80000000u;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const long long row0 = rowbase(seq0);
    const int hpitch = a.ndir * HD;
    for (int j = tid; j < NSR; j += 256) {
        const int seq = seq0 + j;
        const bool valid = seq < a.nseq;
        const long long ro = rowbase(valid ? seq : seq0) - row0;
        rowoff_x[j] = (unsigned)(ro * F * 4);
        rowoff_h[j] = valid ? (unsigned)(ro * hpitch * 4) : OOB;
        if (FUSE) {
            const int b = (valid ? seq : seq0) / a.seqs_per_utt;
            const double cnt = (double)a.rows_per_utt * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            musig[2 * j] = (float)mean;
            musig[2 * j + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
    }
    if (FUSE && tid < 128) { gb[tid] = a.gamma[tid]; gb[128 + tid] = a.beta[tid]; }
    const int c4 = tid & 31, xrow0 = tid >> 5;
    const int xl_even = xrow0 * 512 + (((c4 >> 1) ^ xrow0) << 4) + ((c4 & 1) << 3);
    const int xl_odd = xrow0 * 512 + (((c4 >> 1) ^ xrow0 ^ 8) << 4) + ((c4 & 1) << 3);
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };

    float bias[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float b = a.bsum[dir * 4 * HD + g * HD + 32 * wave + 2 * r + u];
            bias[g][u] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
        }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(reinterpret_cast<const u32x4*>(a.Wp) + (size_t)(dir * 4 + wave_u) * NSLAB * 8 * 3 * 64), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16u;
    auto wload = [&](int slab, int gu, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)(((slab * 8 + gu) * 3 + pc) * 1024), 0);
    };
    auto rsrc_at = [&](const float* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffffu, 0x00020000);
    };

    f32x4 xv[NXV], zv[NXV];
    unsigned xo[NXV];
    auto x_issue = [&](long long t, int m0, int m1) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
#pragma unroll
        for (int m = m0; m < m1; ++m) {
            xo[m] = rowoff_x[xrow0 + 8 * m] + c4 * 16u;
            xv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[m], 0, 0));
            if (FUSE) zv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo[m], 0, 0));
        }
    };
    auto x_commit = [&](long long t, int m0, int m1, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
#pragma unroll
        for (int m = m0; m < m1; ++m) {
            f32x4 v = xv[m];
            if (FUSE) {
                const float mu = musig[2 * (xrow0 + 8 * m)], rs = musig[2 * (xrow0 + 8 * m) + 1];
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zv[m][e] - mu) * rs * gam[e] + bet[e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && dir == 0) ? xo[m] : OOB, 0, 0);
            }
            unsigned pa[3], pb[3];
            split3_pair<NPC>(v[0], v[1], pa[0], pa[1], pa[2]);
            split3_pair<NPC>(v[2], v[3], pb[0], pb[1], pb[2]);
            char* d = img + ((m & 1) ? xl_odd : xl_even) + m * 4096;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x2*>(d + pc * PLANE) = u32x2{pa[pc], pb[pc]};
        }
    };

    __syncthreads();                                   // row tables are in place
    x_issue(dir ? (T - 1) : 0, 0, NXV);
    x_commit(dir ? (T - 1) : 0, 0, NXV, true);
    for (int i = tid; i < NPC * NSR * 64; i += 256) {  // h half of every row
        const int pl = i / (NSR * 64), rem = i % (NSR * 64);
        *reinterpret_cast<unsigned*>(img + pl * PLANE + (rem >> 6) * 512 + 256 + 4 * (rem & 63)) = 0u;
    }
    f32x4 c[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int u = 0; u < 2; ++u) c[nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int abase = r * 512 + ((q ^ (r & 3)) << 4);
    int sl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[j] = (j ^ ((r >> 2) & 3)) << 6;
    auto aload = [&](u32x4 (&af)[NT], int pc, int s4, int hs) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            af[nt] = *reinterpret_cast<const u32x4*>(img + pc * PLANE + nt * 8192 + abase + sl[s4] + hs * 256);
    };
    int hb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hb[e] = (4 * q + e) * 512 + ((16 | ((wave ^ q) << 2) | ((r >> 2) ^ e)) << 4) + 4 * (r & 3);
    const unsigned hcol = (unsigned)(dir * HD + 32 * wave + 2 * r) * 4u;

    u32x4 bw[4][2][NPC];                               // B fragments of the k-slab in progress
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bw[g][u][pc] = wload(0, g * 2 + u, pc);
    f32x4 acc[2][NT][4][2];
    u32x4 fa[NT], fb[NT];
    __syncthreads();

    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    // One pass = the 8 NT MFMAs of weight piece `wpc` against the A fragments in af, in a scheduling region of its own,
    // together with `work()` (vector instructions that do not depend on these MFMAs) and `loads()` (LDS / weight requests
    // for later passes).  The sched_group_barrier sequence asks for MFMA, 2 vector, MFMA, 2 vector, ... with the memory
    // instructions sprinkled in: a wave issues in order, so only an interleaved stream keeps both pipes busy.
    auto pass = [&](f32x4 (&ac)[NT][4][2], const u32x4 (&af)[NT], int wpc, bool first, auto&& work, auto&& loads)
                    __attribute__((always_inline)) {
        fence();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    ac[nt][g][u] = mfma_bf16(af[nt], bw[g][u][wpc], first ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[nt][g][u]);
        loads();
        work();
#pragma unroll
        for (int i = 0; i < 8 * NT; ++i) {
            DPRNN_SGB(0x008, 1);                       // MFMA
            DPRNN_SGB(0x406, 2);                       // VALU | SALU | TRANS
            if (i % 4 == 1) DPRNN_SGB(0x100, 1);       // DS read
            if (i % 4 == 3) DPRNN_SGB(0x020, 1);       // VMEM read
            if (i % 8 == 2) DPRNN_SGB(0x200, 1);       // DS write
            if (i % 16 == 6) DPRNN_SGB(0x040, 1);      // VMEM write
        }
        fence();
    };
    auto wreload = [&](int sn, int wpc) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u) bw[g][u][wpc] = wload(sn, g * 2 + u, wpc);
    };
    auto nothing = [&]() __attribute__((always_inline)) {};
    // the MFMAs of one k-slab (k-slab s = 4 hs + s4; weights of slab s are in bw, those of s + 1 are requested as pieces
    // die); work(p) = the vector work placed beside pass p
    auto slab = [&](f32x4 (&ac)[NT][4][2], u32x4 (&P)[NT], u32x4 (&Q)[NT], int s4, int hs, bool have_p, bool prefetch, bool first,
                    auto&& work) __attribute__((always_inline)) {
        const int sn = (4 * hs + s4 + 1) & 7;
        fence();
        if (!have_p) aload(P, 0, s4, hs);
        aload(Q, 1, s4, hs);
        if constexpr (NP == 6) {
            pass(ac, P, 2, first, [&]() __attribute__((always_inline)) { work(0); }, [&]() __attribute__((always_inline)) { wreload(sn, 2); });
            pass(ac, P, 1, false, [&]() __attribute__((always_inline)) { work(1); }, nothing);
            pass(ac, Q, 1, false, [&]() __attribute__((always_inline)) { work(2); }, [&]() __attribute__((always_inline)) { wreload(sn, 1); });
            pass(ac, P, 0, false, [&]() __attribute__((always_inline)) { work(3); }, nothing);
            pass(ac, Q, 0, false, [&]() __attribute__((always_inline)) { work(4); }, [&]() __attribute__((always_inline)) { aload(P, 2, s4, hs); });
            pass(ac, P, 0, false, [&]() __attribute__((always_inline)) { work(5); },
                 [&]() __attribute__((always_inline)) { if (prefetch) aload(Q, 0, s4 + 1, hs); wreload(sn, 0); });
        } else {
            pass(ac, P, 1, first, [&]() __attribute__((always_inline)) { work(0); work(1); }, [&]() __attribute__((always_inline)) { wreload(sn, 1); });
            pass(ac, P, 0, false, [&]() __attribute__((always_inline)) { work(2); work(3); }, nothing);
            pass(ac, Q, 0, false, [&]() __attribute__((always_inline)) { work(4); work(5); },
                 [&]() __attribute__((always_inline)) { if (prefetch) aload(P, 0, s4 + 1, hs); wreload(sn, 0); });
        }
    };
    // cell update of tile nt from ac (lane-local), h -> image + HBM, cut into 12 units (two per pass of a 6-pass slab):
    // unit 3e + u (u < 2) = gates and state of (row e, unit u); unit 3e + 2 = split h(e, 0..1) into pieces, store
    float hh[2];
    auto cell_unit = [&](f32x4 (&ac)[NT][4][2], int nt, int k, __amdgpu_buffer_rsrc_t hrs) __attribute__((always_inline)) {
        const int e = k / 3, u = k % 3;
        if (u < 2) {
            const float ig = sigmoid_b<FAST>(ac[nt][0][u][e], bias[0][u]);
            const float fg = sigmoid_b<FAST>(ac[nt][1][u][e], bias[1][u]);
            const float gg = tanh_b<FAST>(ac[nt][2][u][e], bias[2][u]);
            const float og = sigmoid_b<FAST>(ac[nt][3][u][e], bias[3][u]);
            const float cn = fg * c[nt][u][e] + ig * gg;
            c[nt][u][e] = cn;
            hh[u] = og * tanh_f<FAST>(cn);
        } else {
            unsigned w1, w2, w3;
            split3_pair<NPC>(hh[0], hh[1], w1, w2, w3);
            char* d = img + hb[e] + nt * 8192;
            *reinterpret_cast<unsigned*>(d) = w1;
            *reinterpret_cast<unsigned*>(d + PLANE) = w2;
            if (NPC == 3) *reinterpret_cast<unsigned*>(d + 2 * PLANE) = w3;
            const unsigned ho = rowoff_h[16 * nt + 4 * q + e] + hcol;
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(hh[0]), __float_as_uint(hh[1])}, hrs, ho, 0, 0);
        }
    };
    auto no_work = [&](int) __attribute__((always_inline)) {};

    // ---- prologue: acc[0] = x_{t0} W_ih (plain MFMA phase), then everybody is done reading x_{t0}
    slab(acc[0], fa, fb, 0, 0, false, true, true, no_work);
    if constexpr (NP == 6) {
        slab(acc[0], fb, fa, 1, 0, true, true, false, no_work);
        slab(acc[0], fa, fb, 2, 0, true, true, false, no_work);
        slab(acc[0], fb, fa, 3, 0, true, false, false, no_work);
    } else {
        slab(acc[0], fa, fb, 1, 0, true, true, false, no_work);
        slab(acc[0], fa, fb, 2, 0, true, true, false, no_work);
        slab(acc[0], fa, fb, 3, 0, true, false, false, no_work);
    }
    __syncthreads();

    unsigned long long st_t0 = 0, st_r0 = 0, st_p1 = 0, st_p2 = 0;
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    // one step; acur = accumulator set holding x_t W_ih on entry, anxt receives x_{t+1} W_ih
    auto one_step = [&](f32x4 (&acur)[NT][4][2], f32x4 (&anxt)[NT][4][2], int step) __attribute__((always_inline)) {
        unsigned long long st_a = 0, st_b = 0;
        if constexpr (STAMP) { fence(); st_a = __builtin_amdgcn_s_memtime(); fence(); }
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        const long long tn = more ? (dir ? (t - 1) : (t + 1)) : t;       // after the last step: harmless re-staging of x_t
        // ---- phase 1: h half of the gates beside the staging of x_{t+1}: quads requested during slab 0 (first half) and
        // slab 1 (second half), split into the image one quad per second pass of slabs 1..3
        constexpr int XH = (NXV + 1) / 2;
        constexpr int QPS = NP == 6 ? 3 : 3;            // quads committed per slab (passes 0, 2, 4; NP = 3: units 0, 2, 4)
        auto xwork = [&](int slab_i, int p) __attribute__((always_inline)) {       // slab_i = 1..3
            if (p % 2 == 0) {
                const int m = (slab_i - 1) * QPS + p / 2;
                if (m < NXV) x_commit(tn, m, m + 1, more);
            }
        };
        fence();
        x_issue(tn, 0, XH);
        slab(acur, fa, fb, 0, 1, false, true, false, no_work);
        fence();
        x_issue(tn, XH, NXV);
        if constexpr (NP == 6) slab(acur, fb, fa, 1, 1, true, true, false, [&](int p) __attribute__((always_inline)) { xwork(1, p); });
        else slab(acur, fa, fb, 1, 1, true, true, false, [&](int p) __attribute__((always_inline)) { xwork(1, p); });
        slab(acur, fa, fb, 2, 1, true, true, false, [&](int p) __attribute__((always_inline)) { xwork(2, p); });
        if constexpr (NP == 6) slab(acur, fb, fa, 3, 1, true, false, false, [&](int p) __attribute__((always_inline)) { xwork(3, p); });
        else slab(acur, fa, fb, 3, 1, true, false, false, [&](int p) __attribute__((always_inline)) { xwork(3, p); });
        fence();
        if constexpr (STAMP) { st_b = __builtin_amdgcn_s_memtime(); fence(); st_p1 += st_b - st_a; }
        __syncthreads();                               // h_{t-1} is no longer read; x_{t+1} is in place
        // ---- phase 2: x half of the next step's gates beside the cell update of this step (one tile per k-slab)
        const __amdgpu_buffer_rsrc_t hrs = rsrc_at(a.H + tbase(t) * hpitch);
        auto cwork = [&](int nt, int p) __attribute__((always_inline)) {
            if (nt < NT) { cell_unit(acur, nt, 2 * p, hrs); cell_unit(acur, nt, 2 * p + 1, hrs); }
        };
        slab(anxt, fa, fb, 0, 0, false, true, true, [&](int p) __attribute__((always_inline)) { cwork(0, p); });
        if constexpr (NP == 6) slab(anxt, fb, fa, 1, 0, true, true, false, [&](int p) __attribute__((always_inline)) { cwork(1, p); });
        else slab(anxt, fa, fb, 1, 0, true, true, false, [&](int p) __attribute__((always_inline)) { cwork(1, p); });
        slab(anxt, fa, fb, 2, 0, true, true, false, [&](int p) __attribute__((always_inline)) { cwork(2, p); });
        if constexpr (NP == 6) slab(anxt, fb, fa, 3, 0, true, false, false, [&](int p) __attribute__((always_inline)) { cwork(3, p); });
        else slab(anxt, fa, fb, 3, 0, true, false, false, [&](int p) __attribute__((always_inline)) { cwork(3, p); });
        fence();
        if constexpr (STAMP) { st_p2 += __builtin_amdgcn_s_memtime() - st_b; fence(); }
        __syncthreads();                               // h_t is in place; x_{t+1} is no longer read
    };
    for (int step = 0; step < T; step += 2) {
        one_step(acc[0], acc[1], step);
        if (step + 1 < T) one_step(acc[1], acc[0], step + 1);
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
            o[2] = st_p1;                                         // cycles in phase 1 (wave 0): h half + x staging
            o[3] = st_p2;                                         // cycles in phase 2: x half + cell update
            o[4] = st_r0;
            o[5] = __builtin_amdgcn_s_memrealtime();
            o[6] = 0; o[7] = 0;
        }
    }
}

template <int NT, int NP, bool FAST, bool FUSE, bool STAMP = false>
int launch_split_ov(const RecurArgs& a, hipStream_t st) {
    constexpr int NSR = 16 * NT;
    const size_t lds = (size_t)(NP == 6 ? 3 : 2) * NSR * 512 + (size_t)(NSR + NSR + 2 * NSR + 256) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split_ov_kernel<NT, NP, FAST, FUSE, STAMP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split_ov): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    dim3 grid((a.nseq + NSR - 1) / NSR, a.ndir);
    hipLaunchKernelGGL((bilstm_split_ov_kernel<NT, NP, FAST, FUSE, STAMP>), grid, dim3(256), lds, st, a);
    return dprnn_check_launch("bilstm_split_ov_kernel");
}

template <int NT>
int launch_split_ov_any(const RecurArgs& a, int products, bool fast, hipStream_t st) {
    if constexpr (NT == 4) {
        if (a.stamps && products == 6) return launch_split_ov<NT, 6, true, false, true>(a, st);
    }
    const bool fuse = a.Z != nullptr;
    if (products == 6) {
        if (fast) return fuse ? launch_split_ov<NT, 6, true, true>(a, st) : launch_split_ov<NT, 6, true, false>(a, st);
        return fuse ? launch_split_ov<NT, 6, false, true>(a, st) : launch_split_ov<NT, 6, false, false>(a, st);
    }
    if (fast) return fuse ? launch_split_ov<NT, 3, true, true>(a, st) : launch_split_ov<NT, 3, true, false>(a, st);
    return fuse ? launch_split_ov<NT, 3, false, true>(a, st) : launch_split_ov<NT, 3, false, false>(a, st);
}

// ------------------------------------------------------------------------------------------------------------------------
// EIGHT-WAVE form (one 512-thread workgroup per CU, NSR = 48 .. 96 sequences, wave w owns the 16 hidden units of weight
// fragment column tile (w >> 1, w & 1) for all four gates).  Why: profiles/r02_split_wide.txt -- with ONE wave per SIMD
// every weight request (buffer_load_dwordx4, ~33 cycles of issue) stops that SIMD's MFMA stream, 27 % of the MFMA phase
// at 64 sequences; with two waves per SIMD one wave's memory and vector instructions issue under the other's MFMAs.
//   STAG = false: [gate GEMM (8 k-slabs)] barrier [cell update + staging of x_{t+1}] barrier        (as the wide form)
//   STAG = true : the x half of the gate GEMM does not depend on h, so between the two barriers that the recurrence needs
//       all waves     : h half of step t                                  (MFMA)
//       barrier A       (h_{t-1} no longer read)
//       waves 0..3    : cell(t), stage x_{t+2}   then   x half of step t+1
//       waves 4..7    : x half of step t+1       then   cell(t), stage x_{t+2}
//       barrier B       (h_t in place)
//   each SIMD holds one wave of either group (waves w and w + 4 share a SIMD), so its vector pipe works on one group's
//   cell update while its matrix pipe works on the other's x half.  Costs: a second accumulator set, x double-buffered
//   in LDS (NSR <= 64).
// Image: regions h | x0 (| x1), each [piece][NSR][256 B], 16-byte chunks XOR-swizzled by (row & 15).
template <int NSR, int NP, bool FAST, bool FUSE, bool STAG, bool STAMP = false, bool F16 = false>
__global__ __launch_bounds__(512, 2)
void bilstm_split8_kernel(RecurArgs a) {
    static_assert(NSR % 16 == 0 && NSR >= 32 && NSR <= 96, "whole 16-row tiles");
    static_assert(!F16 || NP == 3, "the f16 form has two pieces and three products");
    constexpr int NT = NSR / 16;
    constexpr int NPC = NP == 6 ? 3 : 2;
    constexpr int PL = NSR * 256;                      // bytes of one piece plane of one region
    constexpr int REG = NPC * PL;                      // bytes of one region
    constexpr int NXB = STAG ? 2 : 1;                  // x buffers
    extern __shared__ __attribute__((aligned(16))) char img[];
    char* const hreg = img;
    char* const xreg = img + REG;                      // + xb * REG
    unsigned* const rowoff_x = reinterpret_cast<unsigned*>(img + (1 + NXB) * REG);
    unsigned* const rowoff_h = rowoff_x + NSR;
    float* const musig = reinterpret_cast<float*>(rowoff_h + NSR);
    float* const gb = musig + 2 * NSR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int w4 = wave >> 1, wu = wave & 1;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NSR;
    constexpr unsigned OOB = 0x80000000u;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    // Ragged batch, inter-chunk path: this workgroup's sequences, step count and utterance come from its descriptor
    // (tiles never straddle utterances there, their lengths differ); otherwise the uniform addressing of the launch.
    const int* const desc = a.wg_desc ? a.wg_desc + 8 * blockIdx.x : nullptr;
    const int T = __builtin_amdgcn_readfirstlane(desc ? desc[2] : a.T);
    const int nhere = desc ? desc[3] : (a.nseq - seq0 < NSR ? a.nseq - seq0 : NSR);      // sequences that exist in this tile
    const long long row0 = desc ? (((long long)desc[1] << 32) | (unsigned)desc[0]) : rowbase(seq0);
    const int hpitch = a.ndir * HD;
    for (int j = tid; j < NSR; j += 512) {
        const bool valid = j < nhere;
        const int seq = valid ? seq0 + j : seq0;
        const long long ro = desc ? (long long)(valid ? j : 0) * a.inner_stride : rowbase(seq) - row0;
        rowoff_x[j] = (unsigned)(ro * F * 4);          // rows that do not exist alias the first row: loads stay in range
        rowoff_h[j] = valid ? (unsigned)(ro * hpitch * 4) : OOB;
        if (FUSE) {
            const int b = desc ? desc[4] : (a.seq_utt ? a.seq_utt[seq] : seq / a.seqs_per_utt);
            const double cnt = (double)(a.utt_rows ? a.utt_rows[b] : a.rows_per_utt) * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            musig[2 * j] = (float)mean;
            musig[2 * j + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
    }
    if (FUSE && tid < 128) { gb[tid] = a.gamma[tid]; gb[128 + tid] = a.beta[tid]; }
    const int c4 = tid & 31, xrow0 = tid >> 5;         // x quad: row xrow0 + 16 m, floats 4 c4 .. 4 c4 + 3
    const int xl = xrow0 * 256 + (((c4 >> 1) ^ xrow0) << 4) + ((c4 & 1) << 3);
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };
    const int unit0 = 32 * w4 + 2 * r + wu;             // this lane's hidden unit

    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float b = a.bsum[dir * 4 * HD + g * HD + unit0];
        bias[g] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
    }
    // f16 form: the accumulators are S_x S_w times the gate pre-activations; inv = 1 / (S_x S_w) rides on the multiply the
    // nonlinearity starts with anyway
    const float inv = F16 ? a.Wp[F16_SCALE_WORD + 1] : 1.f;
    const float ksig = FAST ? -1.4426950408889634f * inv : inv, ktanh = FAST ? 2.8853900817779268f * inv : inv;
    const int w4_u = __builtin_amdgcn_readfirstlane(w4), wu_u = __builtin_amdgcn_readfirstlane(wu);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(reinterpret_cast<const u32x4*>(a.Wp) + (size_t)(dir * 4 + w4_u) * NSLAB * 8 * 3 * 64), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16u;
    auto wload = [&](int slab, int g, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)(((slab * 8 + g * 2 + wu_u) * 3 + pc) * 1024), 0);
    };
    auto rsrc_at = [&](const float* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffffu, 0x00020000);
    };

    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    // x of one time step -> bf16 pieces in x buffer xb, one quad (4 floats of one row) per thread and m; XG quads are in
    // flight together (registers: the partner wave on the SIMD covers the rest of the latency)
    auto x_stage = [&](long long t, int xb, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
        constexpr int XG = STAG ? 1 : 2;
#pragma unroll
        for (int m0 = 0; m0 < NT; m0 += XG) {
            f32x4 xv[XG], zv[XG];
            unsigned xo[XG];
            fence();
#pragma unroll
            for (int i = 0; i < XG; ++i) {
                const int m = m0 + i < NT ? m0 + i : NT - 1;
                xo[i] = rowoff_x[xrow0 + 16 * m] + c4 * 16u;
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[i], 0, 0));
                if (FUSE) zv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo[i], 0, 0));
            }
#pragma unroll
            for (int i = 0; i < XG; ++i) {
                const int m = m0 + i;
                if (m >= NT) break;
                f32x4 v = xv[i];
                if (FUSE) {
                    const float mu = musig[2 * (xrow0 + 16 * m)], rs = musig[2 * (xrow0 + 16 * m) + 1];
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zv[i][e] - mu) * rs * gam[e] + bet[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && dir == 0) ? xo[i] : OOB, 0, 0);
                }
                unsigned pa[3], pb[3];
                if constexpr (F16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * F16_XS, -F16_XMAX, F16_XMAX);
                    split2_pair_f16(v[0], v[1], pa[0], pa[1]);
                    split2_pair_f16(v[2], v[3], pb[0], pb[1]);
                } else {
                    split3_pair<NPC>(v[0], v[1], pa[0], pa[1], pa[2]);
                    split3_pair<NPC>(v[2], v[3], pb[0], pb[1], pb[2]);
                }
                char* d = xreg + xb * REG + xl + m * 4096;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x2*>(d + pc * PL) = u32x2{pa[pc], pb[pc]};
            }
        }
        fence();
    };

    // The same in two parts (XPF: plain form up to 64 sequences, where the registers allow it): the loads of x_{t+1} are
    // issued at the start of the LAST k-slab of step t and land under its MFMAs, barrier A and the cell update, instead of
    // being waited for right after they are issued.  (vmcnt retires in order: loads issued earlier in the step would sit in
    // front of every weight wait; from the last slab on, the only younger loads are next step's slab-0 weights.)
    constexpr bool XPF = !STAG && NT <= 4;
    f32x4 xq[XPF ? NT : 1], zq[XPF ? NT : 1];
    auto x_issue = [&](long long t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
        fence();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const unsigned xo = rowoff_x[xrow0 + 16 * m] + c4 * 16u;
            xq[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, 0, 0));
            if (FUSE) zq[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo, 0, 0));
        }
        fence();
    };
    auto x_finish = [&](long long t, int xb, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
        fence();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            f32x4 v = xq[m];
            if (FUSE) {
                const float mu = musig[2 * (xrow0 + 16 * m)], rs = musig[2 * (xrow0 + 16 * m) + 1];
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zq[m][e] - mu) * rs * gam[e] + bet[e]);
                const unsigned xo = rowoff_x[xrow0 + 16 * m] + c4 * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && dir == 0) ? xo : OOB, 0, 0);
            }
            unsigned pa[3], pb[3];
            if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * F16_XS, -F16_XMAX, F16_XMAX);
                split2_pair_f16(v[0], v[1], pa[0], pa[1]);
                split2_pair_f16(v[2], v[3], pb[0], pb[1]);
            } else {
                split3_pair<NPC>(v[0], v[1], pa[0], pa[1], pa[2]);
                split3_pair<NPC>(v[2], v[3], pb[0], pb[1], pb[2]);
            }
            char* d = xreg + xb * REG + xl + m * 4096;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x2*>(d + pc * PL) = u32x2{pa[pc], pb[pc]};
        }
        fence();
    };

    __syncthreads();                                   // row tables are in place
    const long long tfirst = dir ? (T - 1) : 0, tsecond = T > 1 ? (dir ? (T - 2) : 1) : tfirst;
    if constexpr (XPF) { x_issue(tfirst); x_finish(tfirst, 0, true); }
    else x_stage(tfirst, 0, true);
    if (STAG) x_stage(tsecond, 1, T > 1);
    for (int i = tid; i < REG / 4; i += 512) reinterpret_cast<unsigned*>(hreg)[i] = 0u;
    f32x4 c[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragments: lane (r, q) of tile nt reads row 16 nt + r, chunk (4 s4 + q) ^ r of the region
    const int abase = r * 256 + ((q ^ (r & 3)) << 4);
    int sl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[j] = (j ^ ((r >> 2) & 3)) << 6;
    auto aload = [&](u32x4 (&af)[NT], const char* reg, int pc, int s4) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt] = *reinterpret_cast<const u32x4*>(reg + pc * PL + nt * 4096 + abase + sl[s4]);
    };
    // h element (row 16 nt + 4q + e, this lane's unit): + nt * 4096 + piece * PL
    int hb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hb[e] = (4 * q + e) * 256 + ((((w4 ^ q) << 2) | ((r >> 2) ^ e)) << 4) + 4 * (r & 3) + 2 * wu;
    const unsigned hcol = (unsigned)(dir * HD + unit0) * 4u;

    u32x4 bw[4][NPC];                                  // B fragments of the k-slab in progress (single-buffered: a piece is
#pragma unroll                                         // re-requested for the next slab when its last product is done)
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) bw[g][pc] = wload(STAG ? 0 : 0, g, pc);
    f32x4 acc[STAG ? 2 : 1][NT][4];
    u32x4 fa[NT], fb[NT];
    auto pass = [&](f32x4 (&ac)[NT][4], const u32x4 (&af)[NT], int wpc, bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                ac[nt][g] = F16 ? mfma_f16(af[nt], bw[g][wpc], first ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[nt][g])
                                : mfma_bf16(af[nt], bw[g][wpc], first ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[nt][g]);
    };
    auto wreload = [&](int sn, int wpc) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[g][wpc] = wload(sn, g, wpc);
    };
    // k-slab s4 of region reg (weights of global slab `s` are in bw; those of slab `sn` are requested as pieces die)
    auto slab = [&](f32x4 (&ac)[NT][4], const char* reg, u32x4 (&P)[NT], u32x4 (&Q)[NT], int s4, int sn, bool have_p, bool prefetch,
                    bool first) __attribute__((always_inline)) {
        fence();
        if (!have_p) aload(P, reg, 0, s4);
        aload(Q, reg, 1, s4);
        fence();
        if constexpr (NP == 6) {
            pass(ac, P, 2, first); wreload(sn, 2); fence();
            pass(ac, P, 1, false); fence();
            pass(ac, Q, 1, false); wreload(sn, 1); fence();
            pass(ac, P, 0, false); fence();
            aload(P, reg, 2, s4);
            pass(ac, Q, 0, false); fence();
            if (prefetch) aload(Q, reg, 0, s4 + 1);
            pass(ac, P, 0, false); wreload(sn, 0); fence();
        } else {
            pass(ac, P, 1, first); wreload(sn, 1); fence();
            pass(ac, P, 0, false); fence();
            if (prefetch) aload(P, reg, 0, s4 + 1);
            pass(ac, Q, 0, false); wreload(sn, 0); fence();
        }
    };
    // four slabs of one region: global slabs s0 .. s0 + 3, the weights of slab (s0 + 4) & 7 are in flight at the end
    auto half = [&](f32x4 (&ac)[NT][4], const char* reg, int s0, bool first, bool xhook = false, long long tx = 0) __attribute__((always_inline)) {
        if constexpr (NP == 6) {
            slab(ac, reg, fa, fb, 0, s0 + 1, false, true, first);
            slab(ac, reg, fb, fa, 1, s0 + 2, true, true, false);
            slab(ac, reg, fa, fb, 2, s0 + 3, true, true, false);
            if (xhook) x_issue(tx);
            slab(ac, reg, fb, fa, 3, (s0 + 4) & 7, true, false, false);
        } else {
            slab(ac, reg, fa, fb, 0, s0 + 1, false, true, first);
            slab(ac, reg, fa, fb, 1, s0 + 2, true, true, false);
            slab(ac, reg, fa, fb, 2, s0 + 3, true, true, false);
            if (xhook) x_issue(tx);
            slab(ac, reg, fa, fb, 3, (s0 + 4) & 7, true, false, false);
        }
    };
    // LSTM cell of this wave's 16 units for every row, h -> image (3 pieces) and HBM
    auto cell = [&](f32x4 (&ac)[NT][4], long long t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t hrs = rsrc_at(a.H + tbase(t) * hpitch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            fence();
            float hv[4];
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                f32x2 ig, fg, gg, og;
                if constexpr (F16) {
                    ig = sigmoid_k2<FAST>(f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]}, ksig, bias[0]);
                    fg = sigmoid_k2<FAST>(f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]}, ksig, bias[1]);
                    gg = tanh_k2<FAST>(f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]}, ktanh, bias[2]);
                    og = sigmoid_k2<FAST>(f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]}, ksig, bias[3]);
                } else {
                    ig = sigmoid_b2<FAST>(f32x2{ac[nt][0][2 * p2], ac[nt][0][2 * p2 + 1]}, bias[0]);
                    fg = sigmoid_b2<FAST>(f32x2{ac[nt][1][2 * p2], ac[nt][1][2 * p2 + 1]}, bias[1]);
                    gg = tanh_b2<FAST>(f32x2{ac[nt][2][2 * p2], ac[nt][2][2 * p2 + 1]}, bias[2]);
                    og = sigmoid_b2<FAST>(f32x2{ac[nt][3][2 * p2], ac[nt][3][2 * p2 + 1]}, bias[3]);
                }
                const f32x2 cn = fg * f32x2{c[nt][2 * p2], c[nt][2 * p2 + 1]} + ig * gg;
                const f32x2 h2 = og * tanh_f2<FAST>(cn);
                c[nt][2 * p2] = cn[0]; c[nt][2 * p2 + 1] = cn[1];
                hv[2 * p2] = h2[0]; hv[2 * p2 + 1] = h2[1];
                unsigned w1, w2, w3 = 0u;              // pieces of rows e = 2 p2 (low halves) and 2 p2 + 1 (high halves)
                if constexpr (F16) split2_pair_f16(h2[0] * F16_XS, h2[1] * F16_XS, w1, w2);
                else split3_pair<NPC>(h2[0], h2[1], w1, w2, w3);
                char* d0 = hreg + hb[2 * p2] + nt * 4096;
                char* d1 = hreg + hb[2 * p2 + 1] + nt * 4096;
                *reinterpret_cast<unsigned short*>(d0) = (unsigned short)w1;
                *reinterpret_cast<unsigned short*>(d1) = (unsigned short)(w1 >> 16);
                *reinterpret_cast<unsigned short*>(d0 + PL) = (unsigned short)w2;
                *reinterpret_cast<unsigned short*>(d1 + PL) = (unsigned short)(w2 >> 16);
                if (NPC == 3) {
                    *reinterpret_cast<unsigned short*>(d0 + 2 * PL) = (unsigned short)w3;
                    *reinterpret_cast<unsigned short*>(d1 + 2 * PL) = (unsigned short)(w3 >> 16);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv[e]), hrs, rowoff_h[16 * nt + 4 * q + e] + hcol, 0, 0);
        }
        fence();
    };

    unsigned long long st_t0 = 0, st_r0 = 0, st_p1 = 0, st_p2 = 0;
    if constexpr (!STAG) {
        __syncthreads();
        if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
        for (int step = 0; step < T; ++step) {
            unsigned long long st_a = 0, st_b = 0;
            if constexpr (STAMP) { fence(); st_a = __builtin_amdgcn_s_memtime(); fence(); }
            const long long t = dir ? (T - 1 - step) : step;
            const bool more = step + 1 < T;
            const long long tn = more ? (dir ? (t - 1) : (t + 1)) : t;
            half(acc[0], xreg, 0, true);
            half(acc[0], hreg, 4, false, XPF, tn);
            if constexpr (STAMP) { fence(); st_b = __builtin_amdgcn_s_memtime(); fence(); st_p1 += st_b - st_a; }
            __syncthreads();                           // every wave is done reading the image of this step
            cell(acc[0], t);
            if constexpr (XPF) x_finish(tn, 0, more);
            else x_stage(tn, 0, more);
            if constexpr (STAMP) { fence(); st_p2 += __builtin_amdgcn_s_memtime() - st_b; fence(); }
            __syncthreads();
        }
    } else {
        __syncthreads();
        half(acc[0], xreg, 0, true);                   // x half of the first step
        if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
        auto one_step = [&](f32x4 (&acur)[NT][4], f32x4 (&anxt)[NT][4], int step) __attribute__((always_inline)) {
            unsigned long long st_a = 0, st_b = 0;
            if constexpr (STAMP) { fence(); st_a = __builtin_amdgcn_s_memtime(); fence(); }
            const long long t = dir ? (T - 1 - step) : step;
            // x_{t+1} sits in buffer (step + 1) & 1; x_{t+2} goes to buffer step & 1 (last read by the x half of step t)
            const long long t2 = step + 2 < T ? (dir ? (t - 2) : (t + 2)) : t;
            const char* const xnext = xreg + ((step + 1) & 1) * REG;
            half(acur, hreg, 4, false);
            if constexpr (STAMP) { fence(); st_b = __builtin_amdgcn_s_memtime(); fence(); st_p1 += st_b - st_a; }
            __syncthreads();                           // barrier A: h_{t-1} is no longer read
            // one copy of the MFMA code, the (much shorter) vector code before it for waves 0..3 and after it for waves 4..7
            if (grp == 0) { cell(acur, t); x_stage(t2, step & 1, step + 2 < T); }
            half(anxt, xnext, 0, true);
            if (grp != 0) { cell(acur, t); x_stage(t2, step & 1, step + 2 < T); }
            if constexpr (STAMP) { fence(); st_p2 += __builtin_amdgcn_s_memtime() - st_b; fence(); }
            __syncthreads();                           // barrier B: h_t and x_{t+2} are in place
        };
        for (int step = 0; step < T; step += 2) {
            one_step(acc[0], acc[STAG ? 1 : 0], step);
            if (step + 1 < T) one_step(acc[STAG ? 1 : 0], acc[0], step + 1);
        }
    }
    if constexpr (STAMP) {
        if (tid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
            o[2] = st_p1;                                         // MFMA phase (STAG: h half)
            o[3] = st_p2;                                         // cell + staging (STAG: cell + staging + x half)
            o[4] = st_r0; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = 0; o[7] = 0;
        }
    }
}

template <int NSR, int NP, bool FAST, bool FUSE, bool STAG, bool STAMP = false, bool F16 = false>
int launch_split8(const RecurArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(STAG ? 3 : 2) * (NP == 6 ? 3 : 2) * NSR * 256 + (size_t)(NSR + NSR + 2 * NSR + 256) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split8_kernel<NSR, NP, FAST, FUSE, STAG, STAMP, F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split8): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    dim3 grid(a.wg_desc ? a.nseq : (a.nseq + NSR - 1) / NSR, a.ndir);          // ragged inter-chunk launches pass their tile count in nseq
    hipLaunchKernelGGL((bilstm_split8_kernel<NSR, NP, FAST, FUSE, STAG, STAMP, F16>), grid, dim3(512), lds, st, a);
    return dprnn_check_launch("bilstm_split8_kernel");
}

template <int NSR, bool STAG>
int launch_split8_any(const RecurArgs& a, int products, bool fast, hipStream_t st) {
    if constexpr (NSR == 64) {
        if (a.stamps && products == 6) return launch_split8<NSR, 6, true, false, STAG, true>(a, st);
        if (a.stamps && products == 2) return launch_split8<NSR, 3, true, false, STAG, true, true>(a, st);
    }
    const bool fuse = a.Z != nullptr;
    if (products == 2) {                                   // f16 form: two pieces, three products
        if (fast) return fuse ? launch_split8<NSR, 3, true, true, STAG, false, true>(a, st)
                              : launch_split8<NSR, 3, true, false, STAG, false, true>(a, st);
        return fuse ? launch_split8<NSR, 3, false, true, STAG, false, true>(a, st)
                    : launch_split8<NSR, 3, false, false, STAG, false, true>(a, st);
    }
    if (products == 6) {
        if (fast) return fuse ? launch_split8<NSR, 6, true, true, STAG>(a, st) : launch_split8<NSR, 6, true, false, STAG>(a, st);
        return fuse ? launch_split8<NSR, 6, false, true, STAG>(a, st) : launch_split8<NSR, 6, false, false, STAG>(a, st);
    }
    if (fast) return fuse ? launch_split8<NSR, 3, true, true, STAG>(a, st) : launch_split8<NSR, 3, true, false, STAG>(a, st);
    return fuse ? launch_split8<NSR, 3, false, true, STAG>(a, st) : launch_split8<NSR, 3, false, false, STAG>(a, st);
}

template <int NSR>
int launch_split_wide_any(const RecurArgs& a, int products, bool fast, hipStream_t st) {
    if constexpr (NSR == 64) {                         // diagnostic builds with in-kernel stamps (tools/bench_recur.py --stamps)
        if (a.stamps && products == 6) {
            const char* dg = getenv("DPRNN_DIAG");
            switch (dg ? atoi(dg) : 0) {
                case 1: return launch_split_wide<NSR, 6, true, false, true, 1>(a, st);
                case 2: return launch_split_wide<NSR, 6, true, false, true, 2>(a, st);
                case 3: return launch_split_wide<NSR, 6, true, false, true, 3>(a, st);
                case 4: return launch_split_wide<NSR, 6, true, false, true, 4>(a, st);
                default: return launch_split_wide<NSR, 6, true, false, true>(a, st);
            }
        }
    }
    const bool fuse = a.Z != nullptr;
    if (products == 6) {
        if (fast) return fuse ? launch_split_wide<NSR, 6, true, true>(a, st) : launch_split_wide<NSR, 6, true, false>(a, st);
        return fuse ? launch_split_wide<NSR, 6, false, true>(a, st) : launch_split_wide<NSR, 6, false, false>(a, st);
    }
    if (fast) return fuse ? launch_split_wide<NSR, 3, true, true>(a, st) : launch_split_wide<NSR, 3, true, false>(a, st);
    return fuse ? launch_split_wide<NSR, 3, false, true>(a, st) : launch_split_wide<NSR, 3, false, false>(a, st);
}

}  // namespace

int recur::launch_bilstm_split(const RecurArgs& a, int products, int seq_tile, bool fast, hipStream_t st) {
    if (products == 2) {                                   // f16 form (gemm mode 3): the eight-wave kernels only
        if (seq_tile == 2080) return launch_split8_any<80, false>(a, 2, fast, st);
        if (seq_tile == 2096) return launch_split8_any<96, false>(a, 2, fast, st);
        if (seq_tile == 2032) return launch_split8_any<32, false>(a, 2, fast, st);
        if (seq_tile == 2048) return launch_split8_any<48, false>(a, 2, fast, st);
        if (seq_tile == 3048) return launch_split8_any<48, true>(a, 2, fast, st);
        if (seq_tile == 3064) return launch_split8_any<64, true>(a, 2, fast, st);
        return launch_split8_any<64, false>(a, 2, fast, st);
    }
    // wide form: one workgroup per CU, one pass over the weights per step for 48 .. 104 sequences
    switch (seq_tile) {
        case 2064: return launch_split8_any<64, false>(a, products, fast, st);  // eight-wave forms (two waves per SIMD)
        case 2080: return launch_split8_any<80, false>(a, products, fast, st);
        case 2096: return launch_split8_any<96, false>(a, products, fast, st);
        case 3048: return launch_split8_any<48, true>(a, products, fast, st);   // ... staggered: cell update of one wave group
        case 3064: return launch_split8_any<64, true>(a, products, fast, st);   //     beside the x-half MFMAs of the other
        case 1048: return launch_split_ov_any<3>(a, products, fast, st);      // overlapped forms (cell update beside the x-half MFMAs)
        case 1064: return launch_split_ov_any<4>(a, products, fast, st);
        case 64: return launch_split_wide_any<64>(a, products, fast, st);
        case 80: return launch_split_wide_any<80>(a, products, fast, st);
        case 96: return launch_split_wide_any<96>(a, products, fast, st);
        case 104: return launch_split_wide_any<104>(a, products, fast, st);
        default: break;
    }
    // two workgroups per CU (L2-weight-stream bound, ~65 GB/s per CU): small launches; -48 / -64 keep the round-1 forms for A/B
    if (seq_tile == -64) {
        if (products == 6) return fast ? launch_split<4, 6, true>(a, st) : launch_split<4, 6, false>(a, st);
        return fast ? launch_split<4, 3, true>(a, st) : launch_split<4, 3, false>(a, st);
    }
    if (seq_tile == -48) {
        if (products == 6) return fast ? launch_split<3, 6, true>(a, st) : launch_split<3, 6, false>(a, st);
        return fast ? launch_split<3, 3, true>(a, st) : launch_split<3, 3, false>(a, st);
    }
    if (products == 6) return fast ? launch_split<2, 6, true>(a, st) : launch_split<2, 6, false>(a, st);
    return fast ? launch_split<2, 3, true>(a, st) : launch_split<2, 3, false>(a, st);
}

extern "C" int dprnn_lstm_pack_f16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                   const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                   void* packed, float* bsum, void* stream) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !bsum) { dprnn_set_error("dprnn_lstm_pack_f16: null pointer"); return -1; }
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && (!w_hh_r || !b_ih_r || !b_hh_r)) { dprnn_set_error("dprnn_lstm_pack_f16: incomplete reverse direction"); return -1; }
    hipLaunchKernelGGL(lstm_absmax_kernel, dim3(ABSMAX_BLOCKS), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh, w_ih_r, w_hh_r, (float*)packed);
    const int n = ndir * 4 * NSLAB * 8 * 64 * 4;
    hipLaunchKernelGGL(lstm_pack_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, (unsigned*)packed, bsum, ndir);
    return dprnn_check_launch("lstm_pack_f16_kernel");
}

extern "C" int dprnn_lstm_pack_split(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                     const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                     void* packed, float* bsum, void* stream) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !bsum) { dprnn_set_error("dprnn_lstm_pack_split: null pointer"); return -1; }
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && (!w_hh_r || !b_ih_r || !b_hh_r)) { dprnn_set_error("dprnn_lstm_pack_split: incomplete reverse direction"); return -1; }
    const int n = ndir * 4 * NSLAB * 8 * 64 * 4;
    hipLaunchKernelGGL(lstm_pack_split_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, (unsigned*)packed, bsum, ndir);
    return dprnn_check_launch("lstm_pack_split_kernel");
}
