// Any-width LSTM recurrence in the f16-pieces arithmetic (round 5): the fp32-equivalent form of csrc/wide.hip's wide_rnn_kernel for LSTM cells.
// Every fp32 operand is written as TWO f16 pieces of a power-of-two multiple of itself (hi = f16(s v), lo = f16(s v - hi): 22+ operand bits)
// and a product is three v_mfma_f32_16x16x32_f16 (hi hi + hi lo + lo hi, fp32 accumulate; the dropped lo lo term is ~2^-22 of the
// product) -- 48 matrix cycles per 16 x 16 x 32 block where the fp32 MFMA takes 256.  What then paces the step is the weight stream
// from L2, so a workgroup owns 64 sequences (8 waves, four 16-row tiles per weight fragment): a quarter of wide_rnn_kernel's bytes per FLOP.
//
// Range is by construction, not by detection: the launch's activation scale 2^ex comes from the measured max |x| of the slab it reads
// (wide_absmax), h lives in (-1, 1) and takes 2^4, and the weight scales are chosen from the measured max |W_ih|, max |W_hh| so that both
// halves of [x | h] [W_ih | W_hh]^T land on ONE common scale with every piece inside the f16 range (wide_f16_prep_kernel).  Non-finite
// inputs do not enter the maxima and come out as non-finite results of their own utterance only.
#include "common.h"
#include "dprnn_mi355x.h"
#include "dprnn_mi355x_wide.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void pieces(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// ---- max |A[i]| over the finite elements, as the bit pattern of a non-negative float (atomicMax on unsigned orders them); out zeroed by the caller
__global__ __launch_bounds__(256) void wide_absmax_kernel(const float* __restrict__ A, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = fabsf(A[i]);
        if (v < 3.0e38f) m = fmaxf(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// ---- scales of one direction: scal[0] = 2^ex (x), scal[1] = W_ih factor, scal[2] = W_hh factor, scal[3] = 1 / common product scale.
// x 2^ex in [2^13, 2^14), h 2^4; W_hh 2^ew, W_ih 2^(4 + ew - ex) so that both products carry 2^(4 + ew); ew the largest exponent that keeps
// max |W_hh| 2^ew and max |W_ih| 2^(4 + ew - ex) below 2^14.  One workgroup; amax words: [0] x, then per direction [1 + 2 d] W_ih, [2 + 2 d] W_hh.
__global__ void wide_f16_prep_kernel(const unsigned* __restrict__ amax, int dir, float* __restrict__ scal) {
    if (threadIdx.x != 0) return;
    auto expo = [](float m) { int e = 0; if (m > 0.f && m < 3.0e38f) frexpf(m, &e); return e < -100 ? -100 : e > 100 ? 100 : e; };   // m in [2^(e-1), 2^e)
    const int ex = 14 - expo(__uint_as_float(amax[0]));
    const int e_ih = expo(__uint_as_float(amax[1 + 2 * dir])), e_hh = expo(__uint_as_float(amax[2 + 2 * dir]));
    int ew = 14 - e_hh;
    if (14 - e_ih - 4 + ex < ew) ew = 14 - e_ih - 4 + ex;
    float* s = scal + 4 * dir;
    s[0] = ldexpf(1.f, ex);
    s[1] = ldexpf(1.f, 4 + ew - ex);
    s[2] = ldexpf(1.f, ew);
    s[3] = ldexpf(1.f, -(4 + ew));
}

// ---- LSTM weights of one direction as f16-piece MFMA fragments.  k axis: x part 0 .. Fp-1 (Fp = F rounded up to 32), h part Fp .. Fp + Hp - 1
// (Hp = H rounded up to 32); units padded to 16.  Fragment (ut, kb, g, piece) = 64 lanes x 8 halves: lane (j, kk) holds
// W[g][ut*16 + j][kb*32 + 8 kk + i], i < 8, times its part's factor; Wp[((((ut*nkb + kb)*4 + g)*2 + piece)*64 + lane)] as 16-byte words.
__global__ __launch_bounds__(256) void wide_lstm_pack_f16_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                                 const float* __restrict__ b_ih, const float* __restrict__ b_hh, int F, int H,
                                                                 const float* __restrict__ scal, u32x4* __restrict__ Wp, float* __restrict__ bv,
                                                                 long long nfrag) {
    const int Fp = (F + 31) & ~31, Hp = (H + 31) & ~31, nkb = (Fp + Hp) / 32, Hu = (H + 15) & ~15;
    const float s_ih = scal[1], s_hh = scal[2];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nfrag; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63); long long r = i >> 6;
        const int g = (int)(r & 3); r >>= 2;
        const int kb = (int)(r % nkb); const int ut = (int)(r / nkb);
        const int unit = ut * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
        _Float16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float v = 0.f;
            if (unit < H) {
                if (k < Fp) { if (k < F) v = w_ih[((long long)g * H + unit) * F + k] * s_ih; }
                else if (k - Fp < H) v = w_hh[((long long)g * H + unit) * H + (k - Fp)] * s_hh;
            }
            pieces(v, hi[e], lo[e]);
        }
        u32x4 whi, wlo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            whi[e] = (unsigned)__builtin_bit_cast(unsigned short, hi[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, hi[2 * e + 1]) << 16);
            wlo[e] = (unsigned)__builtin_bit_cast(unsigned short, lo[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, lo[2 * e + 1]) << 16);
        }
        const long long base = ((((long long)ut * nkb + kb) * 4 + g) * 2) * 64 + lane;
        Wp[base] = whi;
        Wp[base + 64] = wlo;
        if (kb == 0 && (lane >> 4) == 0) bv[g * Hu + unit] = unit < H ? b_ih[g * H + unit] + b_hh[g * H + unit] : 0.f;
    }
}

// ---- one direction of a torch LSTM (dprnn.py:23-37) over 64 sequences per 8-wave workgroup, all T steps.  LDS: ONE image
// [64][hi pieces of (x_t | h_t-1) | lo pieces | pad] in halves; a step is: gate GEMM of every owned unit tile from the image (A operand: two
// 16-byte words per lane and 32-k block) + cell update with h_t kept in registers -> barrier -> h_t pieces and x_t+1 pieces into the
// image -> barrier.  Wave w owns unit tiles w, w + 8, ...: the four gates of one (sequence, unit) meet in one lane, c in registers.
template <int MAXUT, bool FAST, int NT>
__global__ __launch_bounds__(512) void wide_lstm_f16_kernel(const float* __restrict__ X, const u32x4* __restrict__ WpAll, const float* __restrict__ bvAll,
                                                            const float* __restrict__ scalAll, float* __restrict__ Hout, int F, int H, int nseq, int T,
                                                            long long inner_n, long long outer, long long inner, long long t_stride, int ldh,
                                                            long long wp_dir_words, const long long* __restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) _Float16 img[];
    const int Fp = (F + 31) & ~31, Hp = (H + 31) & ~31, Kp = Fp + Hp, nkb = Kp / 32, Hu = (H + 15) & ~15, nut = Hu / 16;
    const int pitch = 2 * Kp + 16;                            // halves per row: hi | lo | 8 dwords of pad (dword pitch / 4 = 2 mod 4)
    constexpr int NS = 16 * NT, NW = 8;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 15, kk = lane >> 4;
    const int dir = blockIdx.y;
    const u32x4* Wp = WpAll + (long long)dir * wp_dir_words;
    const float* bv = bvAll + (long long)dir * 4 * Hu;
    const float sx = scalAll[4 * dir], inv = scalAll[4 * dir + 3];
    const long long q0 = (long long)blockIdx.x * NS;
    for (int i = tid; i < NS * pitch / 2; i += 64 * NW) reinterpret_cast<unsigned*>(img)[i] = 0u;
    long long d_base = 0, d_inner = 0;
    int d_valid = 0;
    if (desc) {
        const long long* d = desc + 4 * (long long)blockIdx.x;
        d_base = d[0]; T = (int)d[1]; d_valid = (int)d[2]; d_inner = d[3];
    }
    // first row of every sequence of the tile (or -1), once, in LDS: kept in registers the 16 (tile row, register) pairs of a lane cost 32 VGPRs
    __shared__ long long row0s[NS];
    if (tid < NS) {
        long long r0;
        if (desc) r0 = tid < d_valid ? d_base + tid * d_inner : -1;
        else { const long long q = q0 + tid; r0 = q < nseq ? (q / inner_n) * outer + (q % inner_n) * inner : -1; }
        row0s[tid] = r0;
    }
    auto seq_row0 = [&](int sr) -> long long { return row0s[sr]; };
    auto load_x = [&](int t) {                                // x_t of the 64 sequences -> pieces in the image's x part
        for (int sr = w; sr < NS; sr += NW) {
            const long long r0 = seq_row0(sr);
            if (r0 < 0) continue;
            const float* src = X + (r0 + (long long)t * t_stride) * F;
            _Float16* row = img + sr * pitch;
            for (int c = lane; c < F; c += 64) {
                _Float16 hi, lo;
                pieces(src[c] * sx, hi, lo);
                row[c] = hi; row[Kp + c] = lo;
            }
        }
    };
    float cst[MAXUT][NT][4], bq[MAXUT][4];
#pragma unroll
    for (int u = 0; u < MAXUT; ++u) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cst[u][nt][r] = 0.f;
        const int ut = w + NW * u;
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[u][g] = ut < nut ? bv[g * Hu + ut * 16 + j] : 0.f;
    }
    __syncthreads();
    const int tfirst = dir ? T - 1 : 0, tstep = dir ? -1 : 1;
    load_x(tfirst);
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = tfirst + step * tstep;
        float hnew[MAXUT][NT][4];
#pragma unroll
        for (int u = 0; u < MAXUT; ++u) {
            const int ut = w + NW * u;
            if (ut < nut) {
                f32x4 acc[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                const u32x4* wp = Wp + ((long long)ut * nkb * 8) * 64 + lane;
                const _Float16* ap = img + j * pitch + 8 * kk;
                for (int kb = 0; kb < nkb; ++kb) {
                    u32x4 ahi[NT], alo[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        ahi[nt] = *reinterpret_cast<const u32x4*>(ap + nt * 16 * pitch + kb * 32);
                        alo[nt] = *reinterpret_cast<const u32x4*>(ap + nt * 16 * pitch + Kp + kb * 32);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {             // one gate's fragment pair at a time: 12 MFMAs per 32 bytes of weights and lane
                        const u32x4 bhi = wp[((long long)(kb * 4 + g) * 2) * 64], blo = wp[((long long)(kb * 4 + g) * 2 + 1) * 64];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt][g] = mfma_f16(ahi[nt], bhi, acc[nt][g]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt][g] = mfma_f16(ahi[nt], blo, acc[nt][g]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt][g] = mfma_f16(alo[nt], bhi, acc[nt][g]);
                    }
                }
                const int unit = ut * 16 + j;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ig = sigmoid_f<FAST>(fmaf(acc[nt][0][r], inv, bq[u][0])), fg = sigmoid_f<FAST>(fmaf(acc[nt][1][r], inv, bq[u][1]));
                        const float gg = tanh_f<FAST>(fmaf(acc[nt][2][r], inv, bq[u][2])), og = sigmoid_f<FAST>(fmaf(acc[nt][3][r], inv, bq[u][3]));
                        const float c = fmaf(fg, cst[u][nt][r], ig * gg);
                        cst[u][nt][r] = c;
                        const float h = og * tanh_f<FAST>(c);
                        hnew[u][nt][r] = h;
                        const long long r0 = seq_row0(nt * 16 + 4 * kk + r);
                        if (r0 >= 0 && unit < H) Hout[(r0 + (long long)t * t_stride) * ldh + (long long)dir * H + unit] = h;
                    }
            }
        }
        __syncthreads();                                      // every wave has read the image of this step
#pragma unroll
        for (int u = 0; u < MAXUT; ++u) {
            const int ut = w + NW * u;
            if (ut < nut) {
                const int unit = ut * 16 + j;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        _Float16 hi, lo;
                        pieces(hnew[u][nt][r] * 16.f, hi, lo);
                        _Float16* row = img + (nt * 16 + 4 * kk + r) * pitch;
                        row[Fp + unit] = hi; row[Kp + Fp + unit] = lo;
                    }
            }
        }
        if (step + 1 < T) load_x(t + tstep);
        __syncthreads();
    }
}

// ---- the Linear after a recurrence (dprnn.py:59,68,88,97) in the same arithmetic: out[m][o] = sum_k A[m][k] W[o][k] + bias[o], A = recurrence
// output rows (|a| <= a_bound, 1 for an LSTM's h: scale 2^13 / a_bound's exponent, no measurement needed), W as pre-packed f16-piece
// fragments.  Weight fragments of one 32-k block and one group of up to 256 output columns: Wp[((cg*nkb + kb)*16 + ct)*2 + piece][lane],
// lane (j, kk) = W[cg*256 + ct*16 + j][kb*32 + 8 kk + i] 2^ew; scal[0] = 2^ea, scal[1] = 1 / (2^ea 2^ew).
__global__ __launch_bounds__(256) void wide_linear_pack_f16_kernel(const float* __restrict__ W, int ldw, int K, int O, const unsigned* __restrict__ amax,
                                                                   float a_bound, u32x4* __restrict__ Wp, float* __restrict__ scal, long long nfrag) {
    const int nkb = (K + 31) / 32;
    int ew = 0, ea = 0;
    { const float m = __uint_as_float(amax[0]); if (m > 0.f && m < 3.0e38f) frexpf(m, &ew); ew = 14 - (ew < -100 ? -100 : ew > 100 ? 100 : ew); }
    frexpf(a_bound, &ea);                                       // a_bound < 2^ea: |a| 2^(14 - ea) stays below 2^14
    ea = 14 - ea;
    const float sw = ldexpf(1.f, ew);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scal[0] = ldexpf(1.f, ea); scal[1] = ldexpf(1.f, -ea - ew); }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nfrag; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63); long long r = i >> 6;
        const int ct = (int)(r & 15); r >>= 4;
        const int kb = (int)(r % nkb); const int cg = (int)(r / nkb);
        const int o = cg * 256 + ct * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
        _Float16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            pieces((o < O && k < K) ? W[(long long)o * ldw + k] * sw : 0.f, hi[e], lo[e]);
        }
        u32x4 whi, wlo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            whi[e] = (unsigned)__builtin_bit_cast(unsigned short, hi[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, hi[2 * e + 1]) << 16);
            wlo[e] = (unsigned)__builtin_bit_cast(unsigned short, lo[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, lo[2 * e + 1]) << 16);
        }
        const long long base = ((((long long)cg * nkb + kb) * 16 + ct) * 2) * 64 + lane;
        Wp[base] = whi;
        Wp[base + 64] = wlo;
    }
}

// 16 NW rows x 256 columns per workgroup of NW waves (wave w: rows 16 w .. 16 w + 15, all column tiles); per 32-k block the A rows are split
// into pieces on their way into LDS (row pitch 80 halves: hi 32 | lo 32 | pad) and the block's weight fragments are copied into LDS as they
// are; the next block's global words are requested before this block's 48 MFMAs per wave.
constexpr int LP = 80;
template <int NW>
__global__ __launch_bounds__(64 * NW) void wide_linear_f16_kernel(const float* __restrict__ A, long long lda, const u32x4* __restrict__ Wp,
                                                                  const float* __restrict__ scal, const float* __restrict__ bias,
                                                                  float* __restrict__ out, long long ldo, double* __restrict__ stats,
                                                                  long long rows_per_utt, long long M, int K, int O) {
    constexpr int BM = 16 * NW, NTH = 64 * NW, WPT = 2048 / NTH;            // 2048 16-byte words of weight fragments per block
    __shared__ __attribute__((aligned(16))) _Float16 As[BM * LP];
    __shared__ __attribute__((aligned(16))) u32x4 Wl[2048];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 15, kk = lane >> 4;
    const long long m0 = (long long)blockIdx.x * BM;
    const int cg = blockIdx.y, o0 = cg * 256;
    const int nkb = (K + 31) / 32;
    const float sa = scal[0], inv = scal[1];
    const int lr = tid >> 2, lk = (tid & 3) * 8;                 // this thread stages (row lr, k lk .. lk + 7) of the A tile
    const long long am = m0 + lr;
    const bool a_ok = am < M;
    const bool vec = (K % 4 == 0) && (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const u32x4* wsrc = Wp + ((long long)cg * nkb) * 2048 + tid;
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a0, a1;
    u32x4 wv[WPT];
    auto stage = [&](int kb) {
        a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0;
        const int k = kb * 32 + lk;
        if (a_ok) {
            const float* src = A + am * lda + k;
            if (vec) {
                if (k < K) a0 = *reinterpret_cast<const f32x4*>(src);
                if (k + 4 < K) a1 = *reinterpret_cast<const f32x4*>(src + 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { if (k + e < K) a0[e] = src[e]; if (k + 4 + e < K) a1[e] = src[4 + e]; }
            }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) wv[i] = wsrc[(long long)kb * 2048 + i * NTH];
    };
    stage(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        {
            _Float16 hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { pieces(a0[e] * sa, hi[e], lo[e]); pieces(a1[e] * sa, hi[4 + e], lo[4 + e]); }
            u32x4 whi, wlo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                whi[e] = (unsigned)__builtin_bit_cast(unsigned short, hi[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, hi[2 * e + 1]) << 16);
                wlo[e] = (unsigned)__builtin_bit_cast(unsigned short, lo[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, lo[2 * e + 1]) << 16);
            }
            *reinterpret_cast<u32x4*>(&As[lr * LP + lk]) = whi;
            *reinterpret_cast<u32x4*>(&As[lr * LP + 32 + lk]) = wlo;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) Wl[tid + i * NTH] = wv[i];
        __syncthreads();
        if (kb + 1 < nkb) stage(kb + 1);
        const u32x4 ahi = *reinterpret_cast<const u32x4*>(&As[(16 * w + j) * LP + 8 * kk]);
        const u32x4 alo = *reinterpret_cast<const u32x4*>(&As[(16 * w + j) * LP + 32 + 8 * kk]);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const u32x4 bhi = Wl[(t * 2) * 64 + lane], blo = Wl[(t * 2 + 1) * 64 + lane];
            acc[t] = mfma_f16(ahi, bhi, acc[t]);
            acc[t] = mfma_f16(ahi, blo, acc[t]);
            acc[t] = mfma_f16(alo, bhi, acc[t]);
        }
    }
    float rs1[4] = {0.f, 0.f, 0.f, 0.f}, rs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long m = m0 + 16 * w + 4 * kk + r;
        if (m >= M) continue;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int o = o0 + 16 * t + j;
            if (o >= O) continue;
            const float v = fmaf(acc[t][r], inv, bias ? bias[o] : 0.f);
            out[m * ldo + o] = v;
            rs1[r] += v; rs2[r] = fmaf(v, v, rs2[r]);
        }
    }
    if (stats) {
        const long long mf = m0 + 16 * w, ml = (mf + 15 < M ? mf + 15 : M - 1);
        if (mf < M) {
            if (mf / rows_per_utt == ml / rows_per_utt) {
                const double s1 = wave_sum((double)(rs1[0] + rs1[1] + rs1[2] + rs1[3]));
                const double s2 = wave_sum((double)(rs2[0] + rs2[1] + rs2[2] + rs2[3]));
                if (lane == 0) { const long long u = mf / rows_per_utt; atomicAdd(&stats[2 * u], s1); atomicAdd(&stats[2 * u + 1], s2); }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a1s = rs1[r], a2s = rs2[r];
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) { a1s += __shfl_xor(a1s, d, 64); a2s += __shfl_xor(a2s, d, 64); }
                    const long long m = mf + 4 * kk + r;
                    if (j == 0 && m < M) { const long long u = m / rows_per_utt; atomicAdd(&stats[2 * u], (double)a1s); atomicAdd(&stats[2 * u + 1], (double)a2s); }
                }
            }
        }
    }
}

inline unsigned grid_cap(long long n, int per_block, long long cap) {
    long long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

#define WF_NULL(cond, name) if (cond) { dprnn_set_error(name ": null pointer"); return -1; }

extern "C" int dprnn_wide_absmax(const float* A, int64_t n, void* out_bits, void* stream) {
    WF_NULL(!A || !out_bits, "dprnn_wide_absmax");
    if (n < 1) { dprnn_set_error("dprnn_wide_absmax: n < 1"); return -1; }
    hipLaunchKernelGGL(wide_absmax_kernel, dim3(grid_cap(n, 256 * 16, 2048)), dim3(256), 0, (hipStream_t)stream, A, (long long)n,
                       static_cast<unsigned*>(out_bits));
    return dprnn_check_launch("wide_absmax_kernel");
}

extern "C" int64_t dprnn_wide_lstm_pack_f16_words(int F, int H) {
    if (F < 1 || H < 1) return -1;
    const long long Fp = (F + 31) & ~31, Hp = (H + 31) & ~31, Hu = (H + 15) & ~15;
    return (Hu / 16) * ((Fp + Hp) / 32) * 4 * 2 * 64;          // 16-byte words of one direction
}

extern "C" int dprnn_wide_lstm_pack_f16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int F, int H, int dir,
                                        const void* amax_bits, float* scal, void* Wp, float* bv, void* stream) {
    WF_NULL(!w_ih || !w_hh || !b_ih || !b_hh || !amax_bits || !scal || !Wp || !bv, "dprnn_wide_lstm_pack_f16");
    const long long words = dprnn_wide_lstm_pack_f16_words(F, H);
    if (words < 0 || dir < 0 || dir > 1) { dprnn_set_error("dprnn_wide_lstm_pack_f16: F=%d H=%d dir=%d", F, H, dir); return -1; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wide_f16_prep_kernel, dim3(1), dim3(64), 0, st, static_cast<const unsigned*>(amax_bits), dir, scal);
    int rc = dprnn_check_launch("wide_f16_prep_kernel");
    if (rc) return rc;
    const long long nfrag = words / 2;
    hipLaunchKernelGGL(wide_lstm_pack_f16_kernel, dim3(grid_cap(nfrag, 256, 4096)), dim3(256), 0, st, w_ih, w_hh, b_ih, b_hh, F, H, scal + 4 * dir,
                       static_cast<u32x4*>(Wp), bv, nfrag);
    return dprnn_check_launch("wide_lstm_pack_f16_kernel");
}

extern "C" int dprnn_wide_lstm_f16_fwd(const float* X, const void* Wp, const float* bv, const float* scal, float* Hout, int ndir, int F, int H,
                                       int nseq, int T, int64_t inner_n, int64_t outer, int64_t inner, int64_t t_stride, const int64_t* desc,
                                       int ntiles, void* stream) {
    WF_NULL(!X || !Wp || !bv || !scal || !Hout, "dprnn_wide_lstm_f16_fwd");
    if (ndir < 1 || ndir > 2 || F < 1 || H < 1 || T < 1 || (desc ? ntiles < 1 : (nseq < 1 || inner_n < 1))) {
        dprnn_set_error("dprnn_wide_lstm_f16_fwd: bad sizes ndir=%d F=%d H=%d nseq=%d T=%d", ndir, F, H, nseq, T);
        return -1;
    }
    const int Fp = (F + 31) & ~31, Hp = (H + 31) & ~31, Hu = (H + 15) & ~15;
    // 64 sequences per workgroup with two unit tiles per wave (hidden_size <= 256); 32 with four (<= 512): c, h_t and the accumulators of a wave
    // have to fit its 256 registers
    const int ns = Hu / 16 <= 16 ? 64 : 32;
    const size_t lds = (size_t)ns * (2 * (Fp + Hp) + 16) * sizeof(_Float16);
    if (lds + 512 > 160 * 1024) {        // + the row bases
        dprnn_set_error("dprnn_wide_lstm_f16_fwd: feature_size %d + hidden_size %d need %zu bytes of LDS (160 KB per CU)", F, H, lds);
        return -1;
    }
    if (Hu / 16 > 8 * 4) { dprnn_set_error("dprnn_wide_lstm_f16_fwd: hidden_size %d > 512", H); return -1; }
    const long long words = dprnn_wide_lstm_pack_f16_words(F, H);
    const dim3 grid(desc ? ntiles : (nseq + ns - 1) / ns, ndir), block(512);
    const bool fast = dprnn_get_fast_activations() != 0;
    hipStream_t st = (hipStream_t)stream;
#define WF_LAUNCH(UT_, FAST_, NT_)                                                                                                              \
    do {                                                                                                                                    \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_lstm_f16_kernel<UT_, FAST_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess) { dprnn_set_error("dprnn_wide_lstm_f16_fwd: hipFuncSetAttribute"); return -2; }     \
        hipLaunchKernelGGL((wide_lstm_f16_kernel<UT_, FAST_, NT_>), grid, block, lds, st, X, static_cast<const u32x4*>(Wp), bv, scal, Hout, F, H, \
                           nseq, T, (long long)inner_n, (long long)outer, (long long)inner, (long long)t_stride, ndir * H, words,            \
                           reinterpret_cast<const long long*>(desc));                                                                       \
    } while (0)
    if (ns == 64) { if (fast) WF_LAUNCH(2, true, 4); else WF_LAUNCH(2, false, 4); }
    else { if (fast) WF_LAUNCH(4, true, 2); else WF_LAUNCH(4, false, 2); }
#undef WF_LAUNCH
    return dprnn_check_launch("wide_lstm_f16_kernel");
}

extern "C" int64_t dprnn_wide_linear_pack_f16_words(int K, int O) {
    if (K < 1 || O < 1) return -1;
    return (long long)((O + 255) / 256) * ((K + 31) / 32) * 2048;
}

extern "C" int dprnn_wide_linear_pack_f16(const float* W, int ldw, int K, int O, const void* amax_bits, float a_bound, void* Wp, float* scal,
                                          void* stream) {
    WF_NULL(!W || !amax_bits || !Wp || !scal, "dprnn_wide_linear_pack_f16");
    const long long words = dprnn_wide_linear_pack_f16_words(K, O);
    if (words < 0 || ldw < K || !(a_bound > 0.f)) { dprnn_set_error("dprnn_wide_linear_pack_f16: K=%d O=%d ldw=%d a_bound=%g", K, O, ldw, (double)a_bound); return -1; }
    const long long nfrag = words / 2;
    hipLaunchKernelGGL(wide_linear_pack_f16_kernel, dim3(grid_cap(nfrag, 256, 4096)), dim3(256), 0, (hipStream_t)stream, W, ldw, K, O,
                       static_cast<const unsigned*>(amax_bits), a_bound, static_cast<u32x4*>(Wp), scal, nfrag);
    return dprnn_check_launch("wide_linear_pack_f16_kernel");
}

extern "C" int dprnn_wide_linear_f16(const float* A, int64_t lda, const void* Wp, const float* scal, const float* bias, float* out, int64_t ldo,
                                     double* stats, int64_t rows_per_utt, int64_t M, int K, int O, void* stream) {
    WF_NULL(!A || !Wp || !scal || !out, "dprnn_wide_linear_f16");
    if (M < 1 || K < 1 || O < 1 || lda < K || ldo < O || rows_per_utt < 1) { dprnn_set_error("dprnn_wide_linear_f16: bad sizes M=%lld K=%d O=%d", (long long)M, K, O); return -1; }
    if (M >= 128LL * dprnn_num_cus()) {                      // 128 rows per 8-wave workgroup: half the weight-fragment copies per row (40.1 against 45.8 ms at F = H = 256)
        const long long gm8 = (M + 127) / 128;
        hipLaunchKernelGGL(wide_linear_f16_kernel<8>, dim3((unsigned)gm8, (O + 255) / 256), dim3(512), 0, (hipStream_t)stream, A, (long long)lda,
                           static_cast<const u32x4*>(Wp), scal, bias, out, (long long)ldo, stats, (long long)rows_per_utt, (long long)M, K, O);
        return dprnn_check_launch("wide_linear_f16_kernel");
    }
    const long long gm = (M + 63) / 64;
    if (gm > 0x7fffffffLL) { dprnn_set_error("dprnn_wide_linear_f16: M too large"); return -1; }
    hipLaunchKernelGGL(wide_linear_f16_kernel<4>, dim3((unsigned)gm, (O + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, (long long)lda,
                       static_cast<const u32x4*>(Wp), scal, bias, out, (long long)ldo, stats, (long long)rows_per_utt, (long long)M, K, O);
    return dprnn_check_launch("wide_linear_f16_kernel");
}
