// Backward of the fused BiLSTM recurrence with the step GEMM in f16 pieces (dprnn_set_bwd_gemm_mode(3); LSTM cells).
//
// Same mathematics and the same interfaces as lstm_bwd_kernel (train_recur.hip): per step, in reverse time order,
//     da_t = gate derivatives of (dH_t + dh_rec, dc_rec) from the forward tape            lane-local, fp32
//     [dx_t | dh_rec] (64 x 256) = da_t (64 x 512) @ [W_ih | W_hh] (512 x 256)             v_mfma_f32_16x16x32_f16
// but the GEMM runs like the forward's default arithmetic (recur_split.h): da_t and the weights as two f16 pieces of
// power-of-two multiples, three products, fp32 accumulate -- 384 matrix cycles per 16x16x256 tile instead of 2048 on the
// fp32 MFMA.  Structure of the forward's eight-wave kernel: ONE 512-thread workgroup per CU owns 64 sequences of one
// direction; wave (w4, wu) owns hidden units j = 32 w4 + 16 wu + r (r = lane & 15) for the gate derivatives AND output columns
// j of both halves of the GEMM (dx column j, dh column j), so dh_rec and dc_rec never leave registers; that unit mapping is
// the fp32 training kernel's, so the tape ([dir][32-sequence workgroup][step][wave][nt][u][slot][lane] float4) is read with
// 16-byte coalesced loads: a 64-sequence tile is two of those workgroups.  da_t goes to LDS as two piece planes
// [piece][seq][512] (1 KB rows, 16-byte chunks XOR-ed with seq & 15) and to HBM for the weight-gradient GEMM -- as the SAME two
// f16 pieces of S_g da, in the QUAD layout  dG4[dir][t][seq / 4][gate * 128 + unit] = 16 bytes {first pieces of sequences 4 j .. 4 j + 3
// | second pieces}  (dprnn_rnn_bwd_dg_floats: the bytes of an fp32 dG): a lane holds one unit of FOUR sequences (the MFMA's
// accumulator rows 4 q + e), so a quad is one 16-byte store (16 per lane and step instead of 64 four-byte ones on the in-order
// vector-memory queue), and dprnn_wgrad_lstm_f16 -- whose MFMA k dimension is the row and which multiplies in the same pieces under
// the same scale -- loads its A fragments without converting anything.
//
// Saturation: da S_g beyond the f16 range is clamped in the piece conversion.  The kernel itself does not track it (it sits at the
// register limit: following the maximum cost 84 bytes of scratch per lane and 14 % of its time); the gate gradients it writes to dG
// are read back, as the very pieces of S_g da, by the weight-gradient GEMM (wgrad_lstm_f16_kernel: flag DPRNN_SAT_WGRAD);
// the weight-gradient GEMM is the only reader of this dG (dprnn_wgrad_lstm_f16 sees a clamped piece pair as |p1 + p2| = 65000).
// Range: gradients have no natural scale, so da is multiplied by S_g = 2^(6 - e), e = exponent of max|dH| over the launch
// (dh_absmax_kernel, one pass over dH before the launch): the largest incoming gradient sits at 2^6, a factor 2^10 below the
// f16 limit for what the recurrence adds (beyond that the piece conversion clamps), and everything above 2^-20 of it keeps a
// normal first piece (below: f16 subnormals, absolute error 2^-31 of the largest gradient).  Weights: per-LSTM 2^e as in the
// forward.  The scales are powers of two and leave with one multiply of the accumulators.
#include "recur_split.h"
using namespace recur;
using namespace rsplit;

namespace {

constexpr int BWD_PACK_WORDS_PER_DIR = 4 * 16 * 4 * 2 * 64 * 4;     // u32 words of piece fragments per direction (= the fp32 pack's floats)
constexpr int BWD_EXTRA = 128;                                        // floats behind the fragments: scales and reduction slots
constexpr int EX_SW = 0, EX_DHMAX = 1, EX_WPART = 8;                  // S_w | bits of max|dH| | 32 partial maxima of |W|
constexpr int BWD_ABSMAX_BLOCKS = 32;

__global__ __launch_bounds__(256)
void bwd_w_absmax_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, const float* __restrict__ w_ih_r,
                         const float* __restrict__ w_hh_r, float* __restrict__ extra) {
    __shared__ float part[4];
    const float* src[4] = {w_ih, w_hh, w_ih_r, w_hh_r};
    float m = 0.f;
    for (int k = 0; k < 4; ++k) {
        if (!src[k]) continue;
        const f32x4* p = reinterpret_cast<const f32x4*>(src[k]);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 4 * HD * F / 4; i += BWD_ABSMAX_BLOCKS * 256) {
            const f32x4 v = p[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        extra[EX_WPART + blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (blockIdx.x == 0) extra[EX_DHMAX] = 0.f;                   // the dH pass of this LSTM's backward starts from zero
    }
}

__device__ __forceinline__ float pow2_scale_to(float m, int target_exp) {      // 2^(target_exp - ex), m = f * 2^ex, f in [0.5, 1)
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
    ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
    return ldexpf(1.f, target_exp - ex);
}

// packedT16[dir][w4][slab 16][ct*2 + wu][piece 2][lane][4 u32]: piece of S_w * Wcat[k][col], k = 32 slab + 8 q + i (gate row),
// col = 32 w4 + 16 wu + r of W_ih (ct = 0: dx columns) or W_hh (ct = 1: dh columns) -- the B operand of the step GEMM
__global__ void pack_bwd_f16_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                    const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                    unsigned* __restrict__ packed, float* __restrict__ extra, int ndir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;           // one thread per dword of piece 0
    if (idx >= ndir * (BWD_PACK_WORDS_PER_DIR / 2)) return;
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < BWD_ABSMAX_BLOCKS; ++i) m = fmaxf(m, extra[EX_WPART + i]);
    const float sw = pow2_scale_to(m, 14);
    if (idx == 0) extra[EX_SW] = sw;
    const int dw = idx & 3, lane = (idx >> 2) & 63, ctwu = (idx >> 8) & 3, s = (idx >> 10) & 15, w4 = (idx >> 14) & 3, dir = idx >> 16;
    const int ct = ctwu >> 1, wu = ctwu & 1, r = lane & 15, q = lane >> 4;
    const int col = 32 * w4 + 16 * wu + r;
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    float v[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 32 * s + 8 * q + 2 * dw + half;
        v[half] = sw * (ct == 0 ? wi[k * F + col] : wh[k * HD + col]);
    }
    unsigned w1, w2;
    split2_pair_f16(v[0], v[1], w1, w2);
    const size_t base = (((((size_t)dir * 4 + w4) * 16 + s) * 4 + ctwu) * 2) * 256 + lane * 4 + dw;
    packed[base] = w1;
    packed[base + 256] = w2;
}

// max |dH| of the launch, as float bits (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256)
void dh_absmax_kernel(const float* __restrict__ dH, long long n4, float* __restrict__ slot) {
    float m = 0.f;
    const f32x4* p = reinterpret_cast<const f32x4*>(dH);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = p[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
}

// Probe builds (tools/build_diag_libs.sh with SRC=train_recur_f16; results are wrong by construction, time and energy are valid):
// 1 no LDS image writes, 2 no dG stores, 4 the first k-slab only (1/16 of the MFMAs, weight stream and image reads), 8 no tape / dH loads
// after the first step, 16 no dX stores, 32 no gate-derivative arithmetic (da = dh)
#ifndef DPRNN_BWD16_DIAG
#define DPRNN_BWD16_DIAG 0
#endif
constexpr bool bdiag(int bit) { return (DPRNN_BWD16_DIAG & bit) != 0; }

struct Bwd16Args {
    const float* dH; const float* tape; const unsigned* Wp; const float* extra;
    float* dX; float* dG;
    long long rows_total;
    int nseq, T, inner_n, ndir;
    long long outer_stride, inner_stride, t_stride;
};

// PACKED: the forward left the gates as three words of 24-bit fixed point per (sequence, unit) in tape slots 0-2 (common.h:
// tape_pack) -- 4 KB instead of 5 KB of tape per row and direction, unpacked where the gate derivatives need them.
template <bool FAST, bool PACKED = false>
__global__ __launch_bounds__(512, 2)
void lstm_bwd8_f16_kernel(Bwd16Args a) {
    constexpr int NSR = 64, NT = 4, KG = 512, NSL = 16;
    constexpr int PL = NSR * KG * 2;                     // bytes of one piece plane: rows of 1 KB
    constexpr int SAUX = 2;                              // once-touched streams: non-temporal
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char img[];
    unsigned* const off_h = reinterpret_cast<unsigned*>(img + 2 * PL);     // byte offsets of a sequence's row in dH / dG / dX
    unsigned* const off_g = off_h + NSR;
    unsigned* const off_x = off_g + NSR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int w4 = wave >> 1, wu = wave & 1;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NSR;
    const int T = a.T;
    const int unit = 32 * w4 + 16 * wu + r;              // this lane's hidden unit = its dx / dh output column
    unsigned gcol = (unsigned)unit * 16u;                // this lane's unit inside a quad row of dG4 (+ gate * 2048 bytes; 16 bytes = 4 sequences)
    unsigned hcol = (unsigned)(dir * HD + unit) * 4u, xcol = (unsigned)unit * 4u;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const long long row0 = rowbase(seq0);
    const int hpitch = a.ndir * HD;
    const int nsq = (a.nseq + 3) >> 2;                   // quads of sequences per time step in dG4
    for (int j = tid; j < NSR; j += 512) {
        const bool valid = seq0 + j < a.nseq;
        const unsigned ro = (unsigned)(rowbase(valid ? seq0 + j : seq0) - row0);
        off_h[j] = valid ? ro * (unsigned)hpitch * 4u : OOB;
        off_g[j] = valid ? (unsigned)(((seq0 + j) >> 2) * KG) * 16u : OOB;        // quad (seq / 4) of this time step: [quad][512 columns][4]
        off_x[j] = valid ? ro * (unsigned)F * 4u : OOB;
    }
    auto rsrc_at = [&](const void* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
    };
    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    auto t_of = [&](int fs) { return (long long)(dir ? (T - 1 - fs) : fs); };          // forward step -> time index

    // scales: S_g from the launch's largest incoming gradient, S_w from the pack
    const float sw = a.extra[EX_SW];
    const float sg = pow2_scale_to(a.extra[EX_DHMAX], 6);
    const float inv = 1.f / (sg * sw);

    // weights: fragment (slab, ct, piece) of this wave
    const int w4_u = __builtin_amdgcn_readfirstlane(w4), wu_u = __builtin_amdgcn_readfirstlane(wu);
    const __amdgpu_buffer_rsrc_t wrs = rsrc_at(reinterpret_cast<const u32x4*>(a.Wp) + (size_t)(dir * 4 + w4_u) * NSL * 4 * 2 * 64);
    const unsigned lane16 = lane * 16u;
    auto wload = [&](int slab, int ct, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)((((slab * 4) + ct * 2 + wu_u) * 2 + pc) * 1024), 0);
    };

    // tape: 64 sequences = two 32-sequence workgroups of the forward's layout; this lane's float4 (four sequences) of tile nt:
    //   ((dir*nwg_b + wg_b)*T + fs)*20480 + ((nt_b*2 + wu)*5 + slot)*256 + w4*5120 + lane*4     floats
    const int nwg_b = (a.nseq + 31) / 32;
    const unsigned tlane = (unsigned)(w4 * 5120 + wu * 5 * 256 + lane * 4) * 4u;
    auto tape_rs = [&](int half, int fs) __attribute__((always_inline)) {
        const int wg_b = 2 * blockIdx.x + half;
        return rsrc_at(a.tape + (((size_t)dir * nwg_b + (wg_b < nwg_b ? wg_b : 0)) * T + fs) * 20480);
    };
    const unsigned tl0 = tlane, tl1 = (2 * (int)blockIdx.x + 1 < nwg_b) ? tlane : OOB;     // second half may not exist
    auto tape_ld = [&](const __amdgpu_buffer_rsrc_t (&trs)[2], int nt, int slot) __attribute__((always_inline)) {
        const unsigned base = (nt >> 1) ? tl1 : tl0;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs[nt >> 1], base + (unsigned)(((nt & 1) * 10 + slot) * 1024), 0, SAUX));
    };

    f32x4 gi[NT], gf[NT], gg[NT], go[PACKED ? 1 : NT], cc[NT], cp[NT], dho[NT];     // PACKED: gi / gf / gg hold the packed words
    f32x4 dh[NT], dc[NT];
    auto load_step = [&](int fs) __attribute__((always_inline)) {      // gates of step fs, c of step fs-1, dH of step fs
        const int fq = fs > 0 ? fs - 1 : 0;
        const __amdgpu_buffer_rsrc_t trs[2] = {tape_rs(0, fs), tape_rs(1, fs)};
        const __amdgpu_buffer_rsrc_t tqs[2] = {tape_rs(0, fq), tape_rs(1, fq)};
        const __amdgpu_buffer_rsrc_t dhs = rsrc_at(a.dH + (row0 + t_of(fs) * a.t_stride) * hpitch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            gi[nt] = tape_ld(trs, nt, 0); gf[nt] = tape_ld(trs, nt, 1); gg[nt] = tape_ld(trs, nt, 2);
            if constexpr (!PACKED) go[nt] = tape_ld(trs, nt, 3);
            const f32x4 c1 = tape_ld(tqs, nt, 4);
            cp[nt] = fs > 0 ? c1 : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dho[nt][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dhs, off_h[16 * nt + 4 * q + e] + hcol, 0, SAUX));
        }
    };
    __syncthreads();                                     // offset tables
    {
        const __amdgpu_buffer_rsrc_t trs[2] = {tape_rs(0, T - 1), tape_rs(1, T - 1)};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            cc[nt] = tape_ld(trs, nt, 4);
            dh[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    load_step(T - 1);

    // da image: element (seq row, k) of piece pc at pc*PL + row*1024 + (((k >> 3) ^ (row & 15)) << 4) + (k & 7)*2.
    // Every address below is a per-lane value that is fixed for the kernel plus an IMMEDIATE (tile, gate, slab group): the
    // empty asm statements in the step loop keep the compiler from hoisting the sums into 64 loop-invariant registers.
    // A fragment of tile nt, slab s: lane (r, q) reads row 16 nt + r, chunk (4 s + q) ^ r
    //   = r*1024 + ((q ^ (r & 3)) << 4) + (((s & 3) ^ (r >> 2)) << 6) + (s >> 2)*256 + nt*16384
    unsigned ab[2][4];                                   // [piece][s & 3]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ab[0][j] = (unsigned)(r * 1024 + ((q ^ (r & 3)) << 4) + ((j ^ (r >> 2)) << 6));
        ab[1][j] = ab[0][j] + (unsigned)PL;
    }
    // this lane's da elements: rows 16 nt + 4 q + e, k = 128 gate + unit: chunk (16 gate + (unit >> 3)) ^ (4 q + e)
    //   = (4 q + e)*1024 + (((unit >> 3) ^ (4 q + e)) << 4) + (unit & 7)*2 + gate*256 + nt*16384
    unsigned wb[2][4];                                   // [piece][e]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wb[0][e] = (unsigned)((4 * q + e) * 1024 + (((unit >> 3) ^ (4 * q + e)) << 4) + (unit & 7) * 2);
        wb[1][e] = wb[0][e] + (unsigned)PL;
    }
    u32x4 bw[2][2];                                      // B fragments of the slab in progress [ct][piece]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) bw[ct][pc] = wload(0, ct, pc);

    for (int fs = T - 1; fs >= 0; --fs) {
        const long long trow = row0 + t_of(fs) * a.t_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ab[0][j]), "+v"(ab[1][j]), "+v"(wb[0][j]), "+v"(wb[1][j]));
        asm volatile("" : "+v"(gcol), "+v"(hcol), "+v"(xcol));
        // ---- gate derivatives (torch.nn.LSTM backward), lane-local; da -> LDS pieces (scaled) and HBM (unscaled)
        const __amdgpu_buffer_rsrc_t dgs = rsrc_at(a.dG + ((size_t)dir * T + (size_t)t_of(fs)) * nsq * (KG * 4));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            fence();
            const f32x4 one = {1.f, 1.f, 1.f, 1.f};
            const f32x4 dht = dho[nt] + dh[nt];
            f32x4 i_, f_, g_, o_;
            if constexpr (PACKED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float vi, vf, vg, vo;
                    tape_unpack(__float_as_uint(gi[nt][e]), __float_as_uint(gf[nt][e]), __float_as_uint(gg[nt][e]), vi, vf, vg, vo);
                    i_[e] = vi; f_[e] = vf; g_[e] = vg; o_[e] = vo;
                }
            } else {
                i_ = gi[nt]; f_ = gf[nt]; g_ = gg[nt]; o_ = go[nt];
            }
            const f32x2 t01 = tanh_f2<FAST>(f32x2{cc[nt][0], cc[nt][1]}), t23 = tanh_f2<FAST>(f32x2{cc[nt][2], cc[nt][3]});
            const f32x4 tc = {t01[0], t01[1], t23[0], t23[1]};
            const f32x4 dct = dc[nt] + dht * o_ * (one - tc * tc);
            f32x4 da[4];
            if constexpr (bdiag(32)) {
                da[0] = dht; da[1] = dht + i_; da[2] = dht + f_; da[3] = dht + g_ + o_ + cp[nt];
                dc[nt] = dht;
            } else {
            da[3] = dht * tc * o_ * (one - o_);
            da[0] = dct * g_ * i_ * (one - i_);
            da[1] = dct * cp[nt] * f_ * (one - f_);
            da[2] = dct * i_ * (one - g_ * g_);
            dc[nt] = dct * f_;
            }
            cc[nt] = cp[nt];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned qw[4];                          // the quad's pieces as they go to HBM: {p1(e0,e1), p1(e2,e3), p2(e0,e1), p2(e2,e3)}
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const float s0 = __builtin_amdgcn_fmed3f(da[g][2 * p2] * sg, -F16_XMAX, F16_XMAX);
                    const float s1 = __builtin_amdgcn_fmed3f(da[g][2 * p2 + 1] * sg, -F16_XMAX, F16_XMAX);
                    unsigned w1, w2;
                    split2_pair_f16(s0, s1, w1, w2);
                    qw[p2] = w1; qw[2 + p2] = w2;
                    char* const d0 = img + nt * 16384 + g * 256, *const d1 = d0;
                    if constexpr (!bdiag(1)) {
                        *reinterpret_cast<unsigned short*>(d0 + wb[0][2 * p2]) = (unsigned short)w1;
                        *reinterpret_cast<unsigned short*>(d1 + wb[0][2 * p2 + 1]) = (unsigned short)(w1 >> 16);
                        *reinterpret_cast<unsigned short*>(d0 + wb[1][2 * p2]) = (unsigned short)w2;
                        *reinterpret_cast<unsigned short*>(d1 + wb[1][2 * p2 + 1]) = (unsigned short)(w2 >> 16);
                    }
                }
                // rows 4 q .. 4 q + 3 of the tile are one quad of sequences (a quad whose first sequence exists is stored whole: the
                // gate gradients of sequences beyond nseq are zero, their tape and dH lanes read as zero)
                if constexpr (!bdiag(2))
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{qw[0], qw[1], qw[2], qw[3]}, dgs, off_g[16 * nt + 4 * q] + gcol + (unsigned)(g * 2048), 0, SAUX);
                else asm volatile("" :: "v"(qw[0]), "v"(qw[1]), "v"(qw[2]), "v"(qw[3]));
            }
        }
        fence();
        if (fs > 0 && !(bdiag(8) && fs < T - 1)) load_step(fs - 1);                   // in flight during the GEMM below
        fence();
        __syncthreads();                                 // the image of this step is complete

        // ---- [dx | dh] = da @ [W_ih | W_hh]: 16 k-slabs of 32, three products, weights of slab s+1 requested as pieces die
        f32x4 acc[NT][2];
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            if (bdiag(4) && s > 0) break;
            fence();
            u32x4 P[NT], Q[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const char* src = img + nt * 16384 + (s >> 2) * 256;
                P[nt] = *reinterpret_cast<const u32x4*>(src + ab[0][s & 3]);
                Q[nt] = *reinterpret_cast<const u32x4*>(src + ab[1][s & 3]);
            }
            fence();
            const int sn = (s + 1) & (NSL - 1);          // wraps to slab 0 of the next step
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][ct] = mfma_f16(P[nt], bw[ct][1], s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt][ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) bw[ct][1] = wload(sn, ct, 1);
            fence();
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][ct] = mfma_f16(Q[nt], bw[ct][0], acc[nt][ct]);
            fence();
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][ct] = mfma_f16(P[nt], bw[ct][0], acc[nt][ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) bw[ct][0] = wload(sn, ct, 0);
        }
        fence();
        __syncthreads();                                 // every wave is done reading the image

        const __amdgpu_buffer_rsrc_t dxs = rsrc_at(a.dX + ((long long)dir * a.rows_total + trow) * F);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dh[nt][e] = acc[nt][1][e] * inv;
                if constexpr (!bdiag(16))
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[nt][0][e] * inv), dxs, off_x[16 * nt + 4 * q + e] + xcol, 0, SAUX);
                else asm volatile("" :: "v"(acc[nt][0][e]));
            }
        }
    }
}

}  // namespace

// ---- host side: called by dprnn_rnn_pack_bwd / dprnn_rnn_bwd (train_recur.hip) when dprnn_set_bwd_gemm_mode(3) is in effect
int recur_bwd_f16_pack(const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r, float* packedT, hipStream_t st) {
    const int ndir = w_ih_r ? 2 : 1;
    float* extra = packedT + (size_t)ndir * BWD_PACK_WORDS_PER_DIR;
    hipLaunchKernelGGL(bwd_w_absmax_kernel, dim3(BWD_ABSMAX_BLOCKS), dim3(256), 0, st, w_ih, w_hh, w_ih_r, w_hh_r, extra);
    const int n = ndir * (BWD_PACK_WORDS_PER_DIR / 2);
    hipLaunchKernelGGL(pack_bwd_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w_ih, w_hh, w_ih_r, w_hh_r,
                       reinterpret_cast<unsigned*>(packedT), extra, ndir);
    return dprnn_check_launch("pack_bwd_f16_kernel");
}

int recur_bwd_f16_launch(const float* dH, const float* tape, const float* packedT, float* dX, float* dG, long long rows_total,
                         int nseq, int T, int inner_n, long long outer_stride, long long inner_stride, long long t_stride, int ndir,
                         bool fast, const float* dhmax, bool tape_packed, hipStream_t st) {
    float* extra = const_cast<float*>(packedT) + (size_t)ndir * BWD_PACK_WORDS_PER_DIR;
    const long long n4 = rows_total * ndir * HD / 4;
    if (dhmax) {                                         // measured by dH's producer (dprnn_rnn_bwd_use_gradmax): 4 bytes instead of a pass
        hipError_t e = hipMemcpyAsync(extra + EX_DHMAX, dhmax, sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { dprnn_set_error("dprnn_rnn_bwd: %s", hipGetErrorString(e)); return -2; }
    } else {
        hipLaunchKernelGGL(dh_absmax_kernel, dim3(1024), dim3(256), 0, st, dH, n4, extra + EX_DHMAX);
    }
    Bwd16Args a{dH, tape, reinterpret_cast<const unsigned*>(packedT), extra, dX, dG, rows_total, nseq, T, inner_n, ndir,
                outer_stride, inner_stride, t_stride};
    const size_t lds = (size_t)2 * 64 * 512 * 2 + 3 * 64 * 4;
    auto kern = tape_packed ? (fast ? &lstm_bwd8_f16_kernel<true, true> : &lstm_bwd8_f16_kernel<false, true>)
                            : (fast ? &lstm_bwd8_f16_kernel<true> : &lstm_bwd8_f16_kernel<false>);
    static std::atomic<unsigned long long> attr_mask[4];
    int attr_dev;
    const int ki = (tape_packed ? 2 : 0) + (fast ? 1 : 0);
    if (dprnn_attr_needed(attr_mask[ki], &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(lstm_bwd8_f16): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask[ki], attr_dev);
    }
    hipLaunchKernelGGL(kern, dim3((nseq + 63) / 64, ndir), dim3(512), lds, st, a);
    return dprnn_check_launch("lstm_bwd8_f16_kernel");
}

// max |x| over n floats (n a multiple of 4) into *out, for dprnn_wgrad_use_f16_pieces with a P that no BPTT launch has measured
extern "C" int dprnn_absmax(const float* x, int64_t n, float* out, void* stream) {
    if (!x || !out || n <= 0 || (n % 4)) { dprnn_set_error("dprnn_absmax: null pointer or n not a multiple of 4"); return -1; }
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) { dprnn_set_error("dprnn_absmax: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(dh_absmax_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, x, (long long)(n / 4), out);
    return dprnn_check_launch("dh_absmax_kernel");
}

// floats of dG for dprnn_rnn_bwd: mode 3 with the LSTM cell writes the quad layout (sequence count rounded up to 4)
extern "C" int64_t dprnn_rnn_bwd_dg_floats(int cell, int bwd_gemm_mode, int nseq, int T, int ndir) {
    const int64_t kg = cell == 2 ? 128 : 512;
    if (cell == 0 && bwd_gemm_mode == 3) return (int64_t)ndir * T * ((nseq + 3) / 4) * 4 * kg;
    return (int64_t)ndir * kg * nseq * T;
}

extern "C" int64_t dprnn_rnn_pack_bwd_floats(int ndir) { return (int64_t)ndir * BWD_PACK_WORDS_PER_DIR + BWD_EXTRA; }
// float offset inside packedT of max|dH| of the last dprnn_rnn_bwd launch in mode 3 (for dprnn_wgrad_use_f16_pieces)
extern "C" int64_t dprnn_rnn_bwd_gradmax_offset(int ndir) { return (int64_t)ndir * BWD_PACK_WORDS_PER_DIR + EX_DHMAX; }
