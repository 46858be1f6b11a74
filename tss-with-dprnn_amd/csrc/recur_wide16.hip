// Sixteen-wave form of the f16-pieces BiLSTM recurrence (gate-GEMM mode 3, 64-sequence tiles, inference): one 1024-thread workgroup
// per CU, FOUR waves per SIMD.
//
// Why (profiles/r03_recur_asym.txt): in the eight-wave kernel (recur_split8.h) a wave owns 64 gate columns x 64 sequences = 64 accumulator
// registers, so only two waves fit a SIMD, and two are too few for the phases of a step to hide each other -- a wave alone on the matrix pipe
// needs 19-25 cycles per MFMA (its own weight requests and fragment reads stall its in-order stream), a cell update beside the partner's
// MFMAs takes twice its time alone, and the later wave's cell update is exposed: 18.8k cycles per step for 12.3k of matrix-pipe work.
// Here a wave owns the FOUR gates of EIGHT hidden units = 32 weight rows, as the row operand of v_mfma_f32_32x32x16_f16 (rows ordered
// gate * 8 + unit, so the 16 accumulator registers of a lane are gates i, f, g, o of four adjacent units of ONE sequence -- the cell update
// needs no lane exchange, h pieces leave as 8-byte LDS stores and h as 16-byte global stores).  32 accumulator registers, 128 in all: four
// waves per SIMD, each with half the cell update of an eight-wave kernel's wave behind half the MFMAs.  The weight stream per step and CU
// is the same 512 KB (every weight fragment is loaded by exactly one wave); the x / h fragments are read from LDS by sixteen waves instead
// of eight (1 MB per step, 4k cycles of the LDS array, under the MFMAs).
//
// Packed weights: appended to the mode-3 buffer after the eight-wave planes (dprnn_lstm_pack_f16 writes both):
//     [dir][wave 16][k-step 16][piece 2][lane 64] 16-byte fragments; lane (rho = lane & 31, kh = lane >> 5) holds
//     S_w W[gate = rho >> 3][unit = 8 wave + (rho & 7)][k = 16 kstep + 8 kh .. + 7] over k in [x | h].
// LDS images, x staging, pending norm, ragged descriptors, saturation flag: as in recur_split8.h (double-buffered h and x images,
// [piece][64 rows][256 B] with 16-byte chunks XOR-swizzled by row & 15; one barrier per step).
#include "recur_split.h"
#include <atomic>
using namespace recur;

namespace rsplit {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int W16_PACK_PER_DIR = 16 * 16 * 2 * 256;          // u32 words of one direction
constexpr int SPLIT8_PACK_PER_DIR = 4 * NSLAB * 8 * 3 * 256;  // u32 words of one direction in the eight-wave planes

__global__ void lstm_pack_w16_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                     const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                     const float* __restrict__ packed_f, unsigned* __restrict__ out, int ndir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (dir, wave, k-step, lane, dword)
    const int per_dir = 16 * 16 * 64 * 4;
    if (idx >= ndir * per_dir) return;
    const float sw = packed_f[F16_SCALE_WORD];                      // written by lstm_pack_f16_kernel, earlier on this stream
    const int dw = idx & 3, lane = (idx >> 2) & 63, ks = (idx >> 8) & 15, wave = (idx >> 12) & 15, dir = idx >> 16;
    const int rho = lane & 31, kh = lane >> 5;
    const int row = (rho >> 3) * HD + 8 * wave + (rho & 7);
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    float v[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 16 * ks + 8 * kh + 2 * dw + half;
        v[half] = sw * (k < F ? wi[row * F + k] : wh[row * HD + (k - F)]);
    }
    unsigned w1, w2;
    split2_pair_f16(v[0], v[1], w1, w2);
    const size_t base = ((((size_t)(dir * 16 + wave) * 16 + ks) * 2) * 64 + lane) * 4 + dw;
    out[base] = w1;
    out[base + 256] = w2;
}

template <bool FAST, bool FUSE, bool STAMP>
__global__ __launch_bounds__(1024)
void bilstm_wide16_kernel(RecurArgs a) {
    constexpr int NSR = 64;
    constexpr int PL = NSR * 256;                      // bytes of one piece plane of one region
    constexpr int REG = 2 * PL;                        // bytes of one region (two pieces)
    extern __shared__ __attribute__((aligned(16))) char img[];
    char* const hreg = img;                            // + buffer * REG
    char* const xreg = img + 2 * REG;
    unsigned* const rowoff_x = reinterpret_cast<unsigned*>(img + 4 * REG);
    unsigned* const rowoff_h = rowoff_x + NSR;
    float* const musig = reinterpret_cast<float*>(rowoff_h + NSR);
    float* const gb = musig + 2 * NSR;
    float* const biasl = gb + 256;                     // [gate][unit] bias of this direction (FAST: times the activation's exponent factor)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NSR;
    constexpr unsigned OOB = 0x80000000u;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const int* const desc = a.wg_desc ? a.wg_desc + 8 * blockIdx.x : nullptr;
    const int T = __builtin_amdgcn_readfirstlane(desc ? desc[2] : a.T);
    const int nhere = desc ? desc[3] : (a.nseq - seq0 < NSR ? a.nseq - seq0 : NSR);
    const long long row0 = desc ? (((long long)desc[1] << 32) | (unsigned)desc[0]) : rowbase(seq0);
    const int hpitch = a.ndir * HD;
    if (tid < NSR) {
        const int j = tid;
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
    if (FUSE && tid >= 128 && tid < 256) { gb[tid - 128] = a.gamma[tid - 128]; gb[tid] = a.beta[tid - 128]; }
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };
    auto rsrc_at = [&](const float* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffffu, 0x00020000);
    };
    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

    // ---- x staging: thread (xrow0 = tid >> 5, c4 = tid & 31) converts floats 4 c4 .. + 3 of rows xrow0 and xrow0 + 32
    const int c4 = tid & 31, xrow0 = tid >> 5;
    const int xl = xrow0 * 256 + (((c4 >> 1) ^ (xrow0 & 15)) << 4) + ((c4 & 1) << 3);      // + m * 8192 (32 rows: same row & 15)
    unsigned satm = 0u;
    f32x4 xq[2], zq[2];
    auto x_issue = [&](long long t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xrs = rsrc_at(a.X + tbase(t) * F);
        const __amdgpu_buffer_rsrc_t zrs = rsrc_at(a.Z + tbase(t) * F);
        fence();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned xo = rowoff_x[xrow0 + 32 * m] + c4 * 16u;
            xq[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, 0, 0));
            if (FUSE) zq[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, xo, 0, 0));
        }
        fence();
    };
    auto x_finish = [&](long long t, int xb, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ors = rsrc_at(a.Xout + tbase(t) * F);
        fence();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 v = xq[m];
            if (FUSE) {
                const float mu = musig[2 * (xrow0 + 32 * m)], rs = musig[2 * (xrow0 + 32 * m) + 1];
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gb + 4 * c4), bet = *reinterpret_cast<const f32x4*>(gb + 128 + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((zq[m][e] - mu) * rs * gam[e] + bet[e]);
                const unsigned xo = rowoff_x[xrow0 + 32 * m] + c4 * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (live && dir == 0) ? xo : OOB, 0, 0);
            }
            sat_track(satm, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * F16_XS, -F16_XMAX, F16_XMAX);
            unsigned pa[2], pb[2];
            split2_pair_f16(v[0], v[1], pa[0], pa[1]);
            split2_pair_f16(v[2], v[3], pb[0], pb[1]);
            char* d = xreg + xb * REG + xl + m * 8192;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) *reinterpret_cast<u32x2*>(d + pc * PL) = u32x2{pa[pc], pb[pc]};
        }
        fence();
    };

    // ---- this lane's part of the cell: gates of units 8 wave + 4 kh .. + 3, sequences n and 32 + n
    const int unit0 = 8 * wave + 4 * kh;
    if (tid >= 512) {                                  // registers: 16 bias values per lane do not fit beside the accumulators
        const int i = tid - 512, g = i >> 7;
        const float b = a.bsum[dir * 4 * HD + i];
        biasl[i] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
    }
    const float inv = a.Wp[F16_SCALE_WORD + 1];        // 1 / (S_x S_w): the accumulators are S_x S_w times the pre-activations
    const float ksig = FAST ? -1.4426950408889634f * inv : inv, ktanh = FAST ? 2.8853900817779268f * inv : inv;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned*>(a.Wp) + (size_t)a.ndir * SPLIT8_PACK_PER_DIR)
                           + (size_t)(dir * 16 + wave) * 16 * 2 * 64), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16u;
    auto wload = [&](int ks, int pc) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (unsigned)((ks * 2 + pc) * 1024), 0);
    };
    // x / h fragment of sequences 32 ct + n, k-step k8 of the region: chunk 2 k8 + kh of the row
    const int fbase = n * 256;
    auto fload = [&](const char* reg, int pc, int k8, int ct) __attribute__((always_inline)) {
        const int chunk = (2 * k8 + kh) ^ (n & 15);
        return *reinterpret_cast<const u32x4*>(reg + pc * PL + ct * 8192 + fbase + (chunk << 4));
    };
    const int hbt = n * 256 + ((wave ^ (n & 15)) << 4) + 8 * kh;     // + ct * 8192 (+ PL for the second piece)
    const unsigned hcol = (unsigned)(dir * HD + unit0) * 4u;

    __syncthreads();                                   // row tables are in place
    const long long tfirst = dir ? (T - 1) : 0;
    x_issue(tfirst);
    x_finish(tfirst, 0, true);
    for (int i = tid; i < REG / 4; i += 1024) reinterpret_cast<unsigned*>(hreg)[i] = 0u;
    f32x4 c[2];
    c[0] = f32x4{0.f, 0.f, 0.f, 0.f}; c[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 w[4][2];                                     // weight fragments of k-steps ks, ks + 1, ks + 2 (ks & 3: two steps ahead)
    w[0][0] = wload(0, 0); w[0][1] = wload(0, 1);
    w[1][0] = wload(1, 0); w[1][1] = wload(1, 1);
    f32x16 acc[2];

    auto mf = [&](u32x4 wa, u32x4 xb, f32x16 cc) __attribute__((always_inline)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa), __builtin_bit_cast(f16x8, xb), cc, 0, 0, 0);
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    unsigned long long st_t0 = 0, st_r0 = 0, st_p1 = 0, st_p2 = 0;
    __syncthreads();
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int step = 0; step < T; ++step) {
        unsigned long long st_a = 0, st_b = 0;
        if constexpr (STAMP) { fence(); st_a = __builtin_amdgcn_s_memtime(); fence(); }
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        const long long tn = more ? (dir ? (t - 1) : (t + 1)) : t;
        const int rb = step & 1, wb = rb ^ 1;
        const char* const xr = xreg + rb * REG;
        const char* const hr = hreg + rb * REG;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            fence();
            if (ks == 13) x_issue(tn);                 // x_{t+1}: in flight under the last k-steps and the cell update
            const int kn = (ks + 2) & 15;              // weights two k-steps ahead (the next step's first two at the end)
            w[(ks + 2) & 3][0] = wload(kn, 0);
            w[(ks + 2) & 3][1] = wload(kn, 1);
            const char* const reg = ks < 8 ? xr : hr;
            u32x4 f1[2], f2[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { f1[ct] = fload(reg, 0, ks & 7, ct); f2[ct] = fload(reg, 1, ks & 7, ct); }
            fence();
            const u32x4 w1 = w[ks & 3][0], w2 = w[ks & 3][1];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = mf(w2, f1[ct], ks == 0 ? zero16 : acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = mf(w1, f2[ct], acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = mf(w1, f1[ct], acc[ct]);
        }
        if constexpr (STAMP) { fence(); st_b = __builtin_amdgcn_s_memtime(); fence(); st_p1 += st_b - st_a; }
        // ---- cell update of this lane's four units, two sequences; h -> image wb (pieces) and HBM
        {
            char* const hw = hreg + wb * REG;
            const __amdgpu_buffer_rsrc_t hrs = rsrc_at(a.H + tbase(t) * hpitch);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                fence();
                float hv[4];
                unsigned p1[2], p2w[2];
                f32x4 bias[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(biasl + g * HD + unit0);
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int e = 2 * p2;
                    const f32x2 ig = sigmoid_k2v<FAST>(f32x2{acc[ct][0 + e], acc[ct][1 + e]}, ksig, f32x2{bias[0][e], bias[0][e + 1]});
                    const f32x2 fg = sigmoid_k2v<FAST>(f32x2{acc[ct][4 + e], acc[ct][5 + e]}, ksig, f32x2{bias[1][e], bias[1][e + 1]});
                    const f32x2 gg = tanh_k2v<FAST>(f32x2{acc[ct][8 + e], acc[ct][9 + e]}, ktanh, f32x2{bias[2][e], bias[2][e + 1]});
                    const f32x2 og = sigmoid_k2v<FAST>(f32x2{acc[ct][12 + e], acc[ct][13 + e]}, ksig, f32x2{bias[3][e], bias[3][e + 1]});
                    const f32x2 cn = fg * f32x2{c[ct][e], c[ct][e + 1]} + ig * gg;
                    const f32x2 h2 = og * tanh_f2<FAST>(cn);
                    c[ct][e] = cn[0]; c[ct][e + 1] = cn[1];
                    hv[e] = h2[0]; hv[e + 1] = h2[1];
                    split2_pair_f16(h2[0] * F16_XS, h2[1] * F16_XS, p1[p2], p2w[p2]);
                }
                char* d = hw + hbt + ct * 8192;
                *reinterpret_cast<u32x2*>(d) = u32x2{p1[0], p1[1]};
                *reinterpret_cast<u32x2*>(d + PL) = u32x2{p2w[0], p2w[1]};
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(hv[0]), __float_as_uint(hv[1]), __float_as_uint(hv[2]), __float_as_uint(hv[3])},
                                                       hrs, rowoff_h[32 * ct + n] + hcol, 0, 0);
            }
            fence();
        }
        x_finish(tn, wb, more);
        if constexpr (STAMP) { fence(); st_p2 += __builtin_amdgcn_s_memtime() - st_b; fence(); }
        __syncthreads();
    }
    sat_commit(a.sat, DPRNN_SAT_RECUR_X, satm, F16_XMAX / F16_XS);
    if constexpr (STAMP) {
        if ((tid == 0 || tid == 256) && a.stamps) {               // waves 0 and 4: two of one SIMD's four
            unsigned long long* o = a.stamps + 16 * (blockIdx.y * gridDim.x + blockIdx.x) + (tid ? 8 : 0);
            o[0] = __builtin_amdgcn_s_memtime() - st_t0;
            o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
            o[2] = st_p1; o[3] = st_p2; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = 0;
        }
    }
}

template <bool FAST, bool FUSE, bool STAMP>
int launch_wide16_t(const RecurArgs& a, hipStream_t st) {
    const size_t lds = (size_t)4 * 2 * 64 * 256 + (size_t)(64 + 64 + 2 * 64 + 256 + 512) * 4;
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_wide16_kernel<FAST, FUSE, STAMP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_wide16): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    dim3 grid(a.wg_desc ? a.nseq : (a.nseq + 63) / 64, a.ndir);
    hipLaunchKernelGGL((bilstm_wide16_kernel<FAST, FUSE, STAMP>), grid, dim3(1024), lds, st, a);
    return dprnn_check_launch("bilstm_wide16_kernel");
}

}  // namespace rsplit

int recur::launch_wide16_f16(const RecurArgs& a, bool fast, hipStream_t st) {
    using namespace rsplit;
    const bool fuse = a.Z != nullptr;
    if (a.stamps) return launch_wide16_t<true, false, true>(a, st);
    if (fast) return fuse ? launch_wide16_t<true, true, false>(a, st) : launch_wide16_t<true, false, false>(a, st);
    return fuse ? launch_wide16_t<false, true, false>(a, st) : launch_wide16_t<false, false, false>(a, st);
}

int recur::pack_wide16_f16(const float* w_ih, const float* w_hh, const float* w_ih_r, const float* w_hh_r, void* packed, int ndir,
                           hipStream_t st) {
    using namespace rsplit;
    const int n = ndir * 16 * 16 * 64 * 4;
    hipLaunchKernelGGL(lstm_pack_w16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w_ih, w_hh, w_ih_r, w_hh_r,
                       reinterpret_cast<const float*>(packed), reinterpret_cast<unsigned*>(packed) + (size_t)ndir * SPLIT8_PACK_PER_DIR, ndir);
    return dprnn_check_launch("lstm_pack_w16_kernel");
}
