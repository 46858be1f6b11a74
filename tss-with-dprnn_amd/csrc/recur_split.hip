// Split gate-GEMM forms of the BiLSTM recurrence (recur_split.h): weight packing, the two-workgroups-per-CU 32-sequence kernel
// (bf16 pieces, small launches of mode 1) and the dispatcher over the tile codes of dprnn_rnn_fwd's seq_tile argument.
#include "recur_split.h"
using namespace recur;
using namespace rsplit;

namespace {

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
                        const float* __restrict__ w_hh_r, float* __restrict__ packed_f, const float* __restrict__ xs_src) {
    __shared__ float part[4];
    const float* src[4] = {w_ih, w_hh, w_ih_r, w_hh_r};
    // W_ih is packed as 2^(4 - s) W_ih (s = the x scale exponent of the launches that will read this pack, 4 without a record):
    // the common weight scale is found over what is actually packed
    const float fih = ldexpf(1.f, 4 - (xs_src ? act_scale_exp(*xs_src) : 4));
    float m = 0.f;
    for (int k = 0; k < 4; ++k) {
        if (!src[k]) continue;
        const f32x4* p = reinterpret_cast<const f32x4*>(src[k]);
        const float f = (k & 1) ? 1.f : fih;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 4 * HD * F / 4; i += ABSMAX_BLOCKS * 256) {   // W_ih [4H][F], W_hh [4H][H]: same size (F == H)
            const f32x4 v = p[i];
            m = fmaxf(m, f * fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
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
                                     unsigned* __restrict__ packed, float* __restrict__ bsum, int ndir, const float* __restrict__ xs_src) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_dir = 4 * NSLAB * 8 * 64 * 4;
    if (idx >= ndir * per_dir) return;
    const float sw = f16_weight_scale(reinterpret_cast<const float*>(packed));
    const float swih = sw * ldexpf(1.f, 4 - (xs_src ? act_scale_exp(*xs_src) : 4));      // x arrives as 2^s x, h as 2^4 h
    if (idx == 0) {                                     // what the recurrence reads: S_w and 1 / (S_x S_w)
        reinterpret_cast<float*>(packed)[F16_SCALE_WORD] = sw;
        reinterpret_cast<float*>(packed)[F16_SCALE_WORD + 1] = 1.f / (F16_XS * sw);
    }
    const int dw = idx & 3, lane = (idx >> 2) & 63, gu = (idx >> 8) & 7, slab = (idx >> 11) & 7;
    const int wave = (idx >> 14) & 3, dir = idx >> 16;
    const int gate = gu >> 1, u = gu & 1, r = lane & 15, q = lane >> 4;
    const int row = gate * HD + 32 * wave + (F16_UNITS_CONTIGUOUS ? 16 * u + r : 2 * r + u);
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    float v[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 32 * slab + 8 * q + 2 * dw + half;
        v[half] = k < F ? swih * wi[row * F + k] : sw * wh[row * HD + (k - F)];
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
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split_kernel<NT, NP, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    dim3 grid((a.nseq + NS - 1) / NS, a.ndir);
    hipLaunchKernelGGL((bilstm_split_kernel<NT, NP, FAST>), grid, dim3(256), lds, st, a);
    return dprnn_check_launch("bilstm_split_kernel");
}


}  // namespace

// seq_tile codes of the split modes: 32 = two 32-sequence workgroups per CU (bf16 pieces only; the L2 weight stream bounds it);
// 2000 + n = the eight-wave kernel with n sequences per workgroup, one workgroup per CU (bf16: 64 / 80 / 96; f16: 16 .. 96).
int recur::launch_bilstm_split(const RecurArgs& a, int gemm_mode, int seq_tile, bool fast, hipStream_t st) {
    const int nsr = seq_tile % 1000;
    if (a.h_pieces && (gemm_mode != 3 || !F16_UNITS_CONTIGUOUS || a.tape)) {
        dprnn_set_error("dprnn_rnn_fwd: H as f16 pieces exists in the inference kernels of gate-GEMM mode 3 only");
        return -1;
    }
    if (gemm_mode == 3) {                                  // f16 pieces: the eight-wave kernels only
        if (seq_tile < 2000 || seq_tile >= 3000 || (nsr != 16 && nsr != 32 && nsr != 48 && nsr != 64 && nsr != 80 && nsr != 96)) {
            dprnn_set_error("dprnn_rnn_fwd: gemm mode 3 has seq_tile 2016, 2032, 2048, 2064, 2080, 2096 (got %d)", seq_tile);
            return -1;
        }
        return launch_split8_f16(a, nsr, fast, st);
    }
    if (gemm_mode != 1) { dprnn_set_error("dprnn_rnn_fwd: no split kernels for gemm mode %d", gemm_mode); return -1; }
    if (seq_tile >= 2000) {
        if (seq_tile >= 3000 || (nsr != 64 && nsr != 80 && nsr != 96)) { dprnn_set_error("dprnn_rnn_fwd: gemm mode 1 has seq_tile 32, 2064, 2080, 2096 (got %d)", seq_tile); return -1; }
        return launch_split8_bf16(a, nsr, fast, st);
    }
    if (seq_tile != 32) { dprnn_set_error("dprnn_rnn_fwd: gemm mode 1 has seq_tile 32, 2064, 2080, 2096 (got %d)", seq_tile); return -1; }
    if (a.wg_desc || a.seq_utt) { dprnn_set_error("dprnn_rnn_fwd: ragged batches need an eight-wave tile (seq_tile 2064)"); return -1; }
    return fast ? launch_split<2, 6, true>(a, st) : launch_split<2, 6, false>(a, st);
}

extern "C" int dprnn_lstm_pack_f16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                   const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                   void* packed, float* bsum, void* stream) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !bsum) { dprnn_set_error("dprnn_lstm_pack_f16: null pointer"); return -1; }
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && (!w_hh_r || !b_ih_r || !b_hh_r)) { dprnn_set_error("dprnn_lstm_pack_f16: incomplete reverse direction"); return -1; }
    const float* xs_src = dprnn_scale_ref().src;           // the record word of the launches that will read this pack (dprnn_scale_slot)
    hipLaunchKernelGGL(lstm_absmax_kernel, dim3(ABSMAX_BLOCKS), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh, w_ih_r, w_hh_r, (float*)packed, xs_src);
    const int n = ndir * 4 * NSLAB * 8 * 64 * 4;
    hipLaunchKernelGGL(lstm_pack_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, (unsigned*)packed, bsum, ndir, xs_src);
    dprnn_pack_note(packed, DPRNN_PACK_FWD, 0, 3, ndir);
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
    dprnn_pack_note(packed, DPRNN_PACK_FWD, 0, 1, ndir);
    return dprnn_check_launch("lstm_pack_split_kernel");
}
