// Generic GEMMs of the training backward pass: the gradients autograd forms for the
// Linear / 1x1-conv / LSTM weights of the reference model when src/trainers/trainer.py:106 calls backward().
//
// dprnn_wgrad    dW[m][n] (+)= sum_rows P[row][m] * Q[row][n]      weight gradients: a reduction over ALL rows of the slab
// dprnn_dgrad_*  dA[row][:] = dOut[row][:] @ W                     data gradients: the forward row GEMM with W read transposed
//
// wgrad structure (gfx950): the row index is the MFMA k dimension.  A workgroup of 8 waves owns a 256 x 128 block of
// dW and a contiguous range of rows; chunks of 16 rows of P (256 columns) and Q (128 columns) go HBM -> registers ->
// LDS (double-buffered, one barrier per chunk) and every wave multiplies its 64 x 64 sub-block out of LDS with
// v_mfma_f32_16x16x4_f32: a lane's 16-byte LDS read of four consecutive columns feeds four different 16x16 tiles (tile j
// owns columns 4c + j), so one ds_read_b128 of P and one of Q feed 16 MFMAs.  Row ranges are split over the grid (wgrad_block: XCD-aware order);
// the partial blocks are written to a workspace and summed by wgrad_reduce_kernel in a fixed order, so the result
// does not depend on scheduling (no floating-point atomics).  The column sums of P (bias gradients) ride along in the
// staging registers of the workgroups with nb == 0.
#include "rowgemm_core.h"
using namespace rg;

namespace {

constexpr int WG_MB = 256, WG_NB = 128, WG_R = 16, WG_THREADS = 512;

struct QDesc {                 // one 128-column block of Q
    const float* p; int ld; int cols;      // cols valid in this block (<= 128), the rest reads as zero
    int dt;                                // row shift in time steps: Q row = row - dt*t_stride, zero where t - dt is outside [0, T)
};
struct WgradArgs {
    const float* P; int ldp; int M;        // P [rows][ldp], M valid columns
    QDesc q[2];
    long long rows;
    long long t_stride; int T;             // time structure of the rows for shifted Q blocks: t = (row / t_stride) % T
    long long rows_per_split;              // multiple of WG_R
    float* part;                           // [splits][Mpad][Npad]
    float* part_b;                         // [splits][Mpad] or null
    int Mpad, Npad;
    int mblocks, nblocks, splits;          // grid: mblocks * nblocks * (splits rounded up to 8) workgroups in one dimension
};

// XCD-aware block order.  The (mb, nb) blocks of one row range read the same rows of P (the same mb) or of Q (the same nb); workgroups go to
// the 8 XCDs round-robin by linear id, so with (mb, nb) fastest the siblings of a row range sat on different XCDs, behind different L2s, and
// every row came from HBM once per sibling (PMC: 7.4 GB per launch against 3.9 GB of P and Q).  Here workgroup id = 8 j + xcd takes sibling
// j % nblk of row range (j / nblk) * 8 + xcd: the siblings share an XCD and leave the dispatcher back to back.
__device__ __forceinline__ bool wgrad_block(const WgradArgs& a, int& mb, int& nb, int& ks) {
    const int nblk = a.mblocks * a.nblocks;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int sib = j % nblk;
    ks = (j / nblk) * 8 + xcd;
    mb = sib % a.mblocks;
    nb = sib / a.mblocks;
    return ks < a.splits;
}

__global__ __launch_bounds__(WG_THREADS)
void wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float Ps[2][WG_R][WG_MB];
    __shared__ __attribute__((aligned(16))) float Qs[2][WG_R][WG_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    int mb, nb, ks;
    if (!wgrad_block(a, mb, nb, ks)) return;        // padding of the last group of 8 row ranges (uniform per workgroup)
    const QDesc qd = a.q[nb];
    const long long r0 = (long long)ks * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > a.rows) r1 = a.rows;

    // staging assignment: P chunk = 16 rows x 64 float4 -> 2 per thread (fixed column chunk), Q chunk = 16 x 32 -> 1.
    // Loads go through buffer resources whose base is the chunk's first row (scalar, advanced per chunk) and whose
    // range ends at row r1: the per-lane byte offsets are fixed for the whole kernel, rows past the range and masked
    // lanes (offset OOB) read as zero -- no 64-bit vector address arithmetic and no division inside the chunk loop.
    constexpr unsigned OOB = 0x80000000u;
    const int pc4 = tid & 63, prow = tid >> 6;          // rows prow, prow + 8
    const int qc4 = tid & 31, qrow = tid >> 5;
    const int pcol = mb * WG_MB + 4 * pc4;
    const bool pok = pcol < a.M;                         // M is a multiple of 4
    const bool qok = 4 * qc4 < qd.cols;
    unsigned poff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) poff[j] = pok ? (unsigned)(((prow + 8 * j) * a.ldp + pcol) * 4) : OOB;
    const unsigned qoff = qok ? (unsigned)((qrow * qd.ld + 4 * qc4) * 4) : OOB;
    // time index of this lane's Q row (shifted blocks only), advanced by 16 rows per chunk without dividing
    const int ts = (int)a.t_stride;                      // t_stride and T fit 32 bits (checked by the launcher)
    int tq = 0, remq = 0;
    if (qd.dt != 0) {
        const long long row = r0 + qrow;
        tq = (int)((row / a.t_stride) % a.T);
        remq = (int)(row % a.t_stride);
    }
    auto rsrc_rows = [&](const float* base, long long row, int ld) __attribute__((always_inline)) {
        // rows [row, r1) of a [rows][ld] matrix; the shifted Q block may start before the matrix (those lanes are masked)
        const long long left = (r1 - row) * (long long)ld * 4;
        const unsigned range = left > 0x7fffffffLL ? 0x7fffffffu : (unsigned)left;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + row * ld), 0, range, 0x00020000);
    };
    f32x4 pv[2], qv, psum = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](long long row0) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t prs = rsrc_rows(a.P, row0, a.ldp);
#pragma unroll
        for (int j = 0; j < 2; ++j) pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, poff[j], 0, 0));
        unsigned qo = qoff;
        if (qd.dt != 0) {
            const int t = tq - qd.dt;
            if (t < 0 || t >= a.T) qo = OOB;
        }
        // Q row = row - dt*t_stride: the shift moves the scalar base, the range still ends at (unshifted) row r1
        const long long shift = (long long)qd.dt * a.t_stride;
        const long long left = (r1 - row0) * (long long)qd.ld * 4;
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(qd.p + (row0 - shift) * qd.ld), 0, left > 0x7fffffffLL ? 0x7fffffffu : (unsigned)left, 0x00020000);
        qv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, qo, 0, 0));
        if (qd.dt != 0) {                                // advance the time index to the next chunk's row
            if (ts == 1) { tq += WG_R; while (tq >= a.T) tq -= a.T; }
            else { remq += WG_R; while (remq >= ts) { remq -= ts; if (++tq == a.T) tq = 0; } }
        }
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<f32x4*>(&Ps[buf][prow + 8 * j][4 * pc4]) = pv[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) psum[e] += pv[j][e];
        }
        *reinterpret_cast<f32x4*>(&Qs[buf][qrow][4 * qc4]) = qv;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (r0 < r1) {
        load(r0);
        store(0);
        __syncthreads();
        int buf = 0;
        for (long long row0 = r0; row0 < r1; row0 += WG_R, buf ^= 1) {
            const bool more = row0 + WG_R < r1;
            if (more) load(row0 + WG_R);
#pragma unroll
            for (int kk = 0; kk < WG_R / 4; ++kk) {
                const f32x4 pa = *reinterpret_cast<const f32x4*>(&Ps[buf][4 * kk + q][64 * wm + 4 * r]);
                const f32x4 qb = *reinterpret_cast<const f32x4*>(&Qs[buf][4 * kk + q][64 * wn + 4 * r]);
#pragma unroll
                for (int jm = 0; jm < 4; ++jm)
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn) acc[jm][jn] = mfma16(pa[jm], qb[jn], acc[jm][jn]);
            }
            if (more) store(buf ^ 1);
            __syncthreads();
        }
    }
    // acc[jm][jn][e] = dW[m][n], m = 256*mb + 64*wm + 4*(4q+e) + jm, n = 128*nb + 64*wn + 4*r + jn
    float* const out = a.part + (long long)ks * a.Mpad * a.Npad;
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = WG_MB * mb + 64 * wm + 4 * (4 * q + e) + jm;
            const int n = WG_NB * nb + 64 * wn + 4 * r;
            *reinterpret_cast<f32x4*>(out + (long long)m * a.Npad + n) =
                f32x4{acc[jm][0][e], acc[jm][1][e], acc[jm][2][e], acc[jm][3][e]};
        }
    if (a.part_b && nb == 0) {
        // column sums of P: threads tid and tid + 64*j hold the same column chunk
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(&Ps[0][0][0]);          // 512 float4 <= 2*16*256/4
        red[tid] = psum;
        __syncthreads();
        if (tid < 64) {
            f32x4 s = red[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const f32x4 v = red[tid + 64 * j];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += v[e];
            }
            *reinterpret_cast<f32x4*>(a.part_b + (long long)ks * a.Mpad + WG_MB * mb + 4 * tid) = s;
        }
    }
}

// ---- the same block GEMM in f16 pieces (LSTM weight gradients when the BPTT ran in f16 pieces, dprnn_wgrad_use_f16_pieces):
// chunks of 32 rows; P (x S_g, the gradient scale of the BPTT launch that produced dG) and Q (x 2^4) are split into two f16
// pieces while they are staged, row-major as they arrive: [piece][k row][column] with 8-byte units XOR-swizzled by the row.  The
// row index is the MFMA k dimension, so a lane needs eight consecutive ROWS of one column: ds_read_b64_tr_b16 reads a 4-row x
// 16-column block per 16 lanes and hands lane i column i (tools/probes/ds_read_tr16.hip), two reads per operand fragment.
// Three products on v_mfma_f32_16x16x32_f16: 48 matrix instructions per 32 rows and wave against 256 on the fp32 MFMA; the
// kernel is left to the HBM streams of P and Q.  Same block / split / workspace plan and the same reduce kernel as above.
constexpr int WF_R = 32;
constexpr int WF_PPL = WF_R * WG_MB * 2, WF_QPL = WF_R * WG_NB * 2;          // bytes of one piece plane of a P / Q chunk
constexpr int WF_BUF = 2 * WF_PPL + 2 * WF_QPL;                              // one buffer: P pieces, Q pieces (48 KB)
typedef short wf_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned wf_swz(int row) { return (unsigned)(((row & 3) << 2) ^ (((row >> 3) & 1) << 4)); }

__global__ __launch_bounds__(WG_THREADS, 2)
void wgrad_f16_kernel(WgradArgs a, const float* __restrict__ pmax, int* sat) {
    extern __shared__ __attribute__((aligned(16))) char wimg[];           // [buf 2][P piece 0 | P piece 1 | Q piece 0 | Q piece 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int mb, nb, ks;
    if (!wgrad_block(a, mb, nb, ks)) return;        // padding of the last group of 8 row ranges (uniform per workgroup)
    const QDesc qd = a.q[nb];
    const long long r0 = (long long)ks * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > a.rows) r1 = a.rows;

    // gradient scale: the largest |dH| of the BPTT launch sits at 2^6 (train_recur_f16.hip)
    float sg;
    {
        const float m = *pmax;
        int ex = 0;
        if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
        ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
        sg = ldexpf(1.f, 6 - ex);
    }
    const float inv = 1.f / (sg * F16P_XS);

    constexpr unsigned OOB = 0x80000000u;
    const int pc4 = tid & 63, prow = tid >> 6;          // P: rows prow + 8 j (j < 4), float4 column chunk pc4
    const int qc4 = tid & 31, qrow = tid >> 5;          // Q: rows qrow + 16 j (j < 2), chunk qc4
    const int pcol = mb * WG_MB + 4 * pc4;
    const bool pok = pcol < a.M;
    const bool qok = 4 * qc4 < qd.cols;
    unsigned poff[4], qoff[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) poff[j] = pok ? (unsigned)(((prow + 8 * j) * a.ldp + pcol) * 4) : OOB;
#pragma unroll
    for (int j = 0; j < 2; ++j) qoff[j] = qok ? (unsigned)(((qrow + 16 * j) * qd.ld + 4 * qc4) * 4) : OOB;
    const int ts = (int)a.t_stride;
    int tq[2] = {0, 0}, remq[2] = {0, 0};
    if (qd.dt != 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long row = r0 + qrow + 16 * j;
            tq[j] = (int)((row / a.t_stride) % a.T);
            remq[j] = (int)(row % a.t_stride);
        }
    }
    f32x4 pv[4], qv[2], psum = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](long long row0) __attribute__((always_inline)) {
        const long long leftp = (r1 - row0) * (long long)a.ldp * 4;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.P + row0 * a.ldp), 0, leftp > 0x7fffffffLL ? 0x7fffffffu : (unsigned)leftp, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, poff[j], 0, 0));
        const long long shift = (long long)qd.dt * a.t_stride;
        const long long leftq = (r1 - row0) * (long long)qd.ld * 4;
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(qd.p + (row0 - shift) * qd.ld), 0, leftq > 0x7fffffffLL ? 0x7fffffffu : (unsigned)leftq, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned qo = qoff[j];
            if (qd.dt != 0) {
                const int t = tq[j] - qd.dt;
                if (t < 0 || t >= a.T) qo = OOB;
            }
            qv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, qo, 0, 0));
            if (qd.dt != 0) {                            // advance the time index to the next chunk's row
                if (ts == 1) { tq[j] += WF_R; while (tq[j] >= a.T) tq[j] -= a.T; }
                else { remq[j] += WF_R; while (remq[j] >= ts) { remq[j] -= ts; if (++tq[j] == a.T) tq[j] = 0; } }
            }
        }
    };
    unsigned satm = 0u;                                  // bits of the largest |scaled dG / x element| converted (common.h)
    // element unit (row, 8-byte unit u) of a plane with `pitch` bytes per row: row*pitch + ((u ^ swz(row)) << 3)
    auto pieces4 = [&](const f32x4& v, float s, unsigned long long& lo, unsigned long long& hi) __attribute__((always_inline)) {
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(v[e] * s, -F16P_XMAX, F16P_XMAX);
        unsigned a1, a2, b1, b2;
        f16_pieces2(c[0], c[1], a1, a2);
        f16_pieces2(c[2], c[3], b1, b2);
        lo = (unsigned long long)a1 | ((unsigned long long)b1 << 32);
        hi = (unsigned long long)a2 | ((unsigned long long)b2 << 32);
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
        char* const pb = wimg + buf * WF_BUF;
        char* const qb = pb + 2 * WF_PPL;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = prow + 8 * j;
            unsigned long long lo, hi;
            sat_track(satm, pv[j]);                  // dG under the BPTT's gradient scale; x / h rows (Q) are tracked where the forward reads them
            pieces4(pv[j], sg, lo, hi);
            char* d = pb + row * (WG_MB * 2) + (((unsigned)pc4 ^ wf_swz(row)) << 3);
            *reinterpret_cast<unsigned long long*>(d) = lo;
            *reinterpret_cast<unsigned long long*>(d + WF_PPL) = hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) psum[e] += pv[j][e];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = qrow + 16 * j;
            unsigned long long lo, hi;
            pieces4(qv[j], F16P_XS, lo, hi);
            char* d = qb + row * (WG_NB * 2) + (((unsigned)qc4 ^ wf_swz(row)) << 3);
            *reinterpret_cast<unsigned long long*>(d) = lo;
            *reinterpret_cast<unsigned long long*>(d + WF_QPL) = hi;
        }
    };

    // transposed fragment reads: lane (g = lane >> 4, l = lane & 15, q' = l >> 2, p = l & 3) supplies row 8 g + 4 h + q',
    // unit (U + p) ^ swz(row) = U ^ (p | q' << 2 | (g & 1) << 4) of a 16-column tile whose first unit U is a multiple of 4
    const int g = lane >> 4, l = lane & 15, qq = l >> 2, pp = l & 3;
    const unsigned ub = (unsigned)(pp | (qq << 2) | ((g & 1) << 4));
    unsigned pa[4], qa[4];                               // byte offsets of the h = 0 read of tile jm / jn inside a piece plane
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pa[j] = (unsigned)((8 * g + qq) * (WG_MB * 2)) + ((((unsigned)(16 * wm + 4 * j)) ^ ub) << 3);
        qa[j] = (unsigned)((8 * g + qq) * (WG_NB * 2)) + ((((unsigned)(16 * wn + 4 * j)) ^ ub) << 3);
    }
    auto tr8 = [&](const char* plane, unsigned off, int pitch) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) wf_s4* lptr;
        const wf_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off));
        const wf_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off + 4 * pitch));
        rg_u32x4 out;
        const unsigned long long a0 = __builtin_bit_cast(unsigned long long, lo), a1 = __builtin_bit_cast(unsigned long long, hi);
        out[0] = (unsigned)a0; out[1] = (unsigned)(a0 >> 32); out[2] = (unsigned)a1; out[3] = (unsigned)(a1 >> 32);
        return out;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A wave whose 64 gradient columns lie beyond M (the Linears' dZ has 128 of the block's 256) multiplies nothing: it stages and
    // synchronises with the others, and the matrix pipe is left to the waves with work -- without this the kernel was bound by MFMAs on
    // zero padding (0.67 ms at 3.5 TB/s where its 2.38 GB take 0.48 ms).
    const bool live = WG_MB * mb + 64 * wm < a.M;

    if (r0 < r1) {
        load(r0);
        store(0);
        __syncthreads();
        int buf = 0;
        for (long long row0 = r0; row0 < r1; row0 += WF_R, buf ^= 1) {
            const bool more = row0 + WF_R < r1;
            if (more) load(row0 + WF_R);
            const char* const pb = wimg + buf * WF_BUF;
            const char* const qb = pb + 2 * WF_PPL;
            if (live) {
                rg_u32x4 q1[4], q2[4];
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) { q1[jn] = tr8(qb, qa[jn], WG_NB * 2); q2[jn] = tr8(qb + WF_QPL, qa[jn], WG_NB * 2); }
#pragma unroll
                for (int jm = 0; jm < 4; ++jm) {
                    const rg_u32x4 p1 = tr8(pb, pa[jm], WG_MB * 2), p2 = tr8(pb + WF_PPL, pa[jm], WG_MB * 2);
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn) {
                        acc[jm][jn] = mfma_f16p(p1, q2[jn], acc[jm][jn]);
                        acc[jm][jn] = mfma_f16p(p2, q1[jn], acc[jm][jn]);
                        acc[jm][jn] = mfma_f16p(p1, q1[jn], acc[jm][jn]);
                    }
                }
            }
            if (more) store(buf ^ 1);
            __syncthreads();
        }
    }
    sat_commit(sat, DPRNN_SAT_WGRAD, satm, F16P_XMAX / sg);
    // acc[jm][jn][e] = dW[m][n], m = 256 mb + 64 wm + 16 jm + 4 g + e, n = 128 nb + 64 wn + 16 jn + l
    float* const out = a.part + (long long)ks * a.Mpad * a.Npad;
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = WG_MB * mb + 64 * wm + 16 * jm + 4 * g + e;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) out[(long long)m * a.Npad + WG_NB * nb + 64 * wn + 16 * jn + l] = acc[jm][jn][e] * inv;
        }
    if (a.part_b && nb == 0) {
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(wimg);
        red[tid] = psum;
        __syncthreads();
        if (tid < 64) {
            f32x4 s = red[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const f32x4 v = red[tid + 64 * j];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += v[e];
            }
            *reinterpret_cast<f32x4*>(a.part_b + (long long)ks * a.Mpad + WG_MB * mb + 4 * tid) = s;
        }
    }
}

// ---- LSTM weight gradients from the QUAD layout of dG (train_recur_f16.hip: dG4[t][seq / 4][512 columns] = 16 bytes {first f16 pieces
// of S_g da for sequences 4 j .. 4 j + 3 | second pieces}, one direction):
//     dWcat[m][n] = sum over (t, seq) of dG(t, seq, m) * [x(t, seq) | h(t -+ 1, seq)][n]        m < 512 gate rows, n < 256
// The rows of this GEMM are the MFMA k dimension.  In the quad layout four consecutive k of one column m sit in one 16-byte load,
// already as the pieces the BPTT multiplied with: a thread loads two quads (8 rows of one column) and writes ONE 16-byte A fragment
// per piece ([piece][8-row group][column][16 B]) -- no conversion, no transposition; a wave's fragment read is a plain,
// conflict-free ds_read_b128.  The bias gradient (column sums of dG) is rebuilt as p1 + p2 by the x blocks, which also see a
// clamped pair there (|p1 + p2| = 65000: saturation flag).  Q (x and
// h rows, row-major in the slab) keeps the swizzled row-major image + ds_read_b64_tr_b16 of wgrad_f16_kernel, but its rows are now
// addressed as the recurrence addresses them -- row(t, seq) = (seq / inner_n) outer + (seq % inner_n) inner + t t_stride -- because the
// k order is (t, seq), not slab order.  k runs over t * nseq32 + seq with the sequence count rounded up to 32 (a chunk never
// straddles two time steps); sequences beyond nseq read as zero on both sides.  Same 256 x 128 blocks, row-range splits, workspace
// and reduce kernel as dprnn_wgrad.
struct WgLstmArgs {
    const float* P4; const float* X; const float* Hd;      // dG4 of this direction; x rows [.][128]; h columns of this direction, pitch ldh
    int ldh, dt;                                           // dt: +1 forward direction (h_{t-1}), -1 reverse (h_{t+1})
    int nseq, nseq32, nsq, T, inner_n;
    long long outer_stride, inner_stride, t_stride;
    long long rows_per_split;                              // in k = t * nseq32 + seq, a multiple of 32
    float* part; float* part_b;
    int splits;
};

__global__ __launch_bounds__(WG_THREADS, 2)
void wgrad_lstm_f16_kernel(WgLstmArgs a, const float* __restrict__ pmax, int* sat) {
    extern __shared__ __attribute__((aligned(16))) char wimg[];           // [buf 2][P piece 0 | P piece 1 | Q piece 0 | Q piece 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int Mpad = 512, Npad = 256;
    // blocks: mb in {0, 1} (256 gate rows each), nb in {0: x, 1: h}; the four siblings of a row range share an XCD (wgrad_block)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int sib = jj & 3, ks = (jj >> 2) * 8 + xcd;
    if (ks >= a.splits) return;
    const int mb = sib & 1, nb = sib >> 1;
    const long long total = (long long)a.T * a.nseq32;
    const long long r0 = (long long)ks * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > total) r1 = total;

    float sg;
    {
        const float m = *pmax;
        int ex = 0;
        if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
        ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
        sg = ldexpf(1.f, 6 - ex);
    }
    const float inv = 1.f / (sg * F16P_XS);

    // P staging: thread = (column pcol of the block, half kgh); 8-row groups kg = kgh + 2 j, j < 2; quads 2 kg, 2 kg + 1
    const int pcol = tid & 255, kgh = tid >> 8;
    // Q staging (as wgrad_f16_kernel): rows qrow + 16 j (j < 2), float4 chunk qc4
    const int qc4 = tid & 31, qrow = tid >> 5;
    const float* const qbase = nb == 0 ? a.X : a.Hd;
    const int qld = nb == 0 ? 128 : a.ldh;
    const int qdt = nb == 0 ? 0 : a.dt;
    // (t, first sequence) of the current chunk and, per Q row of this thread, the sequence's (quotient, remainder) by inner_n
    int t = (int)(r0 / a.nseq32), s0 = (int)(r0 % a.nseq32);
    int qq[2], qr[2];
    auto seq_split = [&](int seq, int& quo, int& rem) { quo = seq / a.inner_n; rem = seq - quo * a.inner_n; };
#pragma unroll
    for (int j = 0; j < 2; ++j) seq_split(s0 + qrow + 16 * j, qq[j], qr[j]);

    rg_u32x4 pv[2][2];
    f32x4 qv[2];
    float psum = 0.f;
    unsigned satm = 0u;
    const bool bias_block = a.part_b && nb == 0;
    auto load = [&]() __attribute__((always_inline)) {      // the chunk at (t, s0)
        const float* pt = a.P4 + ((size_t)t * a.nsq + (size_t)(s0 >> 2)) * (Mpad * 4) + (size_t)(mb * WG_MB + pcol) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qd = 2 * (kgh + 2 * j) + h;            // quad of the chunk: sequences s0 + 4 qd ..
                pv[j][h] = (s0 + 4 * qd < a.nseq) ? *reinterpret_cast<const rg_u32x4*>(pt + (size_t)qd * (Mpad * 4)) : rg_u32x4{0u, 0u, 0u, 0u};
            }
        const int tq = t - qdt;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int seq = s0 + qrow + 16 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (seq < a.nseq && tq >= 0 && tq < a.T) {
                const long long row = (long long)qq[j] * a.outer_stride + (long long)qr[j] * a.inner_stride + (long long)tq * a.t_stride;
                v = *reinterpret_cast<const f32x4*>(qbase + row * qld + 4 * qc4);
            }
            qv[j] = v;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {   // to the next chunk of 32 rows
        s0 += WF_R;
        if (s0 >= a.nseq32) {
            s0 = 0; ++t;
#pragma unroll
            for (int j = 0; j < 2; ++j) seq_split(qrow + 16 * j, qq[j], qr[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) { qr[j] += WF_R; while (qr[j] >= a.inner_n) { qr[j] -= a.inner_n; ++qq[j]; } }
        }
    };
    auto pieces4 = [&](const f32x4& v, float s, unsigned long long& lo, unsigned long long& hi) __attribute__((always_inline)) {
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(v[e] * s, -F16P_XMAX, F16P_XMAX);
        unsigned a1, a2, b1, b2;
        f16_pieces2(c[0], c[1], a1, a2);
        f16_pieces2(c[2], c[3], b1, b2);
        lo = (unsigned long long)a1 | ((unsigned long long)b1 << 32);
        hi = (unsigned long long)a2 | ((unsigned long long)b2 << 32);
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
        char* const pb = wimg + buf * WF_BUF;
        char* const qb = pb + 2 * WF_PPL;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* d = pb + (((kgh + 2 * j) * WG_MB + pcol) << 4);       // A fragment of (8-row group, column): rows 8 kg .. 8 kg + 7
            *reinterpret_cast<rg_u32x4*>(d) = rg_u32x4{pv[j][0][0], pv[j][0][1], pv[j][1][0], pv[j][1][1]};                // first pieces
            *reinterpret_cast<rg_u32x4*>(d + WF_PPL) = rg_u32x4{pv[j][0][2], pv[j][0][3], pv[j][1][2], pv[j][1][3]};      // second pieces
            if (bias_block) {                                           // S_g da = p1 + p2 (22+ bits): column sum and range
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const unsigned u1 = pv[j][h][w], u2 = pv[j][h][2 + w];
                        const float v0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u1 & 0xffffu)) +
                                         (float)__builtin_bit_cast(_Float16, (unsigned short)(u2 & 0xffffu));
                        const float v1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u1 >> 16)) +
                                         (float)__builtin_bit_cast(_Float16, (unsigned short)(u2 >> 16));
                        sat_track(satm, v0, v1);
                        psum += v0 + v1;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = qrow + 16 * j;
            unsigned long long lo, hi;
            pieces4(qv[j], F16P_XS, lo, hi);
            char* d = qb + row * (WG_NB * 2) + (((unsigned)qc4 ^ wf_swz(row)) << 3);
            *reinterpret_cast<unsigned long long*>(d) = lo;
            *reinterpret_cast<unsigned long long*>(d + WF_QPL) = hi;
        }
    };

    const int g = lane >> 4, l = lane & 15, qqq = l >> 2, pp = l & 3;
    const unsigned ub = (unsigned)(pp | (qqq << 2) | ((g & 1) << 4));
    unsigned pa[4], qa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pa[j] = (unsigned)((g * WG_MB + 64 * wm + 16 * j + l) << 4);                    // lane (l, g): column 64 wm + 16 j + l, rows 8 g .. 8 g + 7
        qa[j] = (unsigned)((8 * g + qqq) * (WG_NB * 2)) + ((((unsigned)(16 * wn + 4 * j)) ^ ub) << 3);
    }
    auto tr8 = [&](const char* plane, unsigned off, int pitch) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) wf_s4* lptr;
        const wf_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off));
        const wf_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off + 4 * pitch));
        rg_u32x4 out;
        const unsigned long long a0 = __builtin_bit_cast(unsigned long long, lo), a1 = __builtin_bit_cast(unsigned long long, hi);
        out[0] = (unsigned)a0; out[1] = (unsigned)(a0 >> 32); out[2] = (unsigned)a1; out[3] = (unsigned)(a1 >> 32);
        return out;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (r0 < r1) {
        load();
        store(0);
        __syncthreads();
        int buf = 0;
        for (long long k0 = r0; k0 < r1; k0 += WF_R, buf ^= 1) {
            const bool more = k0 + WF_R < r1;
            if (more) { advance(); load(); }
            const char* const pb = wimg + buf * WF_BUF;
            const char* const qb = pb + 2 * WF_PPL;
            rg_u32x4 q1[4], q2[4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) { q1[jn] = tr8(qb, qa[jn], WG_NB * 2); q2[jn] = tr8(qb + WF_QPL, qa[jn], WG_NB * 2); }
#pragma unroll
            for (int jm = 0; jm < 4; ++jm) {
                const rg_u32x4 p1 = *reinterpret_cast<const rg_u32x4*>(pb + pa[jm]), p2 = *reinterpret_cast<const rg_u32x4*>(pb + WF_PPL + pa[jm]);
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) {
                    acc[jm][jn] = mfma_f16p(p1, q2[jn], acc[jm][jn]);
                    acc[jm][jn] = mfma_f16p(p2, q1[jn], acc[jm][jn]);
                    acc[jm][jn] = mfma_f16p(p1, q1[jn], acc[jm][jn]);
                }
            }
            if (more) store(buf ^ 1);
            __syncthreads();
        }
    }
    sat_commit(sat, DPRNN_SAT_WGRAD, satm, F16P_XMAX * 0.99999f);          // a clamped pair is exactly +-65000
    // acc[jm][jn][e] = dW[m][n], m = 256 mb + 64 wm + 16 jm + 4 g + e, n = 128 nb + 64 wn + 16 jn + l
    float* const out = a.part + (long long)ks * Mpad * Npad;
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = WG_MB * mb + 64 * wm + 16 * jm + 4 * g + e;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) out[(long long)m * Npad + WG_NB * nb + 64 * wn + 16 * jn + l] = acc[jm][jn][e] * inv;
        }
    if (a.part_b && nb == 0) {                              // bias gradient: column sums of dG (threads tid and tid + 256 hold the same column)
        __syncthreads();
        float* red = reinterpret_cast<float*>(wimg);
        red[tid] = psum;
        __syncthreads();
        if (tid < 256) a.part_b[(long long)ks * Mpad + WG_MB * mb + tid] = (red[tid] + red[tid + 256]) * (1.f / sg);
    }
}

// ---- The same GEMM on 256 x 256 blocks (round 4).  The 256 x 128 form above gives every row range to four sibling workgroups
// (two gate-row halves x {x, h}), so each dG chunk and each [x | h] chunk is read by TWO workgroups; the XCD's L2 serves only part of
// the second reads -- PMC: 7.52 GB per launch against 4.77 GB of operands (profiles/r04_pmc_traffic_train_f16.json), in a kernel that
// runs at the HBM roofline (5.1 TB/s).  Here a workgroup owns 256 gate rows x ALL 256 columns of [x | h]: dG is read once, only the
// (three times smaller) [x | h] rows by the two siblings.  Waves (wm, wn): 64 gate rows x 128 columns -- wn = 0 multiplies with x,
// wn = 1 with h_prev -- 128 accumulator registers per lane; the Q image holds both sources side by side (512 B per row and piece),
// 64 KB per buffer, one workgroup per CU as before.  Staging, fragment formats, split-k plan and the reduce kernel are unchanged.
constexpr int WFW_QPL = WF_R * 256 * 2;                                       // bytes of one piece plane of a wide Q chunk (16 KB)
constexpr int WFW_BUF = 2 * WF_PPL + 2 * WFW_QPL;                            // 64 KB

__global__ __launch_bounds__(WG_THREADS, 2)
void wgrad_lstm_f16_wide_kernel(WgLstmArgs a, const float* __restrict__ pmax, int* sat) {
    extern __shared__ __attribute__((aligned(16))) char wimg[];           // [buf 2][P piece 0 | P piece 1 | Q piece 0 | Q piece 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int Mpad = 512, Npad = 256, QP = 512;                       // QP: bytes of a Q row of one piece ([x 128 | h 128] f16)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int mb = jj & 1, ks = (jj >> 1) * 8 + xcd;                      // the two siblings of a row range share an XCD
    if (ks >= a.splits) return;
    const long long total = (long long)a.T * a.nseq32;
    const long long r0 = (long long)ks * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > total) r1 = total;

    float sg;
    {
        const float m = *pmax;
        int ex = 0;
        if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
        ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
        sg = ldexpf(1.f, 6 - ex);
    }
    const float inv = 1.f / (sg * F16P_XS);

    const int pcol = tid & 255, kgh = tid >> 8;
    const int qc4 = tid & 31, qrow = tid >> 5;
    int t = (int)(r0 / a.nseq32), s0 = (int)(r0 % a.nseq32);
    int qq[2], qr[2];
    auto seq_split = [&](int seq, int& quo, int& rem) { quo = seq / a.inner_n; rem = seq - quo * a.inner_n; };
#pragma unroll
    for (int j = 0; j < 2; ++j) seq_split(s0 + qrow + 16 * j, qq[j], qr[j]);

    rg_u32x4 pv[2][2];
    f32x4 qv[2][2];                                                       // [row j][0: x(t), 1: h(t - dt)]
    float psum = 0.f;
    unsigned satm = 0u;
    auto load = [&]() __attribute__((always_inline)) {
        const float* pt = a.P4 + ((size_t)t * a.nsq + (size_t)(s0 >> 2)) * (Mpad * 4) + (size_t)(mb * WG_MB + pcol) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qd = 2 * (kgh + 2 * j) + h;
                pv[j][h] = (s0 + 4 * qd < a.nseq) ? *reinterpret_cast<const rg_u32x4*>(pt + (size_t)qd * (Mpad * 4)) : rg_u32x4{0u, 0u, 0u, 0u};
            }
        const int th = t - a.dt;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int seq = s0 + qrow + 16 * j;
            f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vh = {0.f, 0.f, 0.f, 0.f};
            if (seq < a.nseq) {
                const long long base = (long long)qq[j] * a.outer_stride + (long long)qr[j] * a.inner_stride;
                vx = *reinterpret_cast<const f32x4*>(a.X + (base + (long long)t * a.t_stride) * 128 + 4 * qc4);
                if (th >= 0 && th < a.T) vh = *reinterpret_cast<const f32x4*>(a.Hd + (base + (long long)th * a.t_stride) * a.ldh + 4 * qc4);
            }
            qv[j][0] = vx; qv[j][1] = vh;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        s0 += WF_R;
        if (s0 >= a.nseq32) {
            s0 = 0; ++t;
#pragma unroll
            for (int j = 0; j < 2; ++j) seq_split(qrow + 16 * j, qq[j], qr[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) { qr[j] += WF_R; while (qr[j] >= a.inner_n) { qr[j] -= a.inner_n; ++qq[j]; } }
        }
    };
    auto pieces4 = [&](const f32x4& v, float s, unsigned long long& lo, unsigned long long& hi) __attribute__((always_inline)) {
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(v[e] * s, -F16P_XMAX, F16P_XMAX);
        unsigned a1, a2, b1, b2;
        f16_pieces2(c[0], c[1], a1, a2);
        f16_pieces2(c[2], c[3], b1, b2);
        lo = (unsigned long long)a1 | ((unsigned long long)b1 << 32);
        hi = (unsigned long long)a2 | ((unsigned long long)b2 << 32);
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
        char* const pb = wimg + buf * WFW_BUF;
        char* const qb = pb + 2 * WF_PPL;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* d = pb + (((kgh + 2 * j) * WG_MB + pcol) << 4);
            *reinterpret_cast<rg_u32x4*>(d) = rg_u32x4{pv[j][0][0], pv[j][0][1], pv[j][1][0], pv[j][1][1]};
            *reinterpret_cast<rg_u32x4*>(d + WF_PPL) = rg_u32x4{pv[j][0][2], pv[j][0][3], pv[j][1][2], pv[j][1][3]};
            if (a.part_b) {                                             // S_g da = p1 + p2: column sum and range (every block holds x)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const unsigned u1 = pv[j][h][w], u2 = pv[j][h][2 + w];
                        const float v0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u1 & 0xffffu)) +
                                         (float)__builtin_bit_cast(_Float16, (unsigned short)(u2 & 0xffffu));
                        const float v1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u1 >> 16)) +
                                         (float)__builtin_bit_cast(_Float16, (unsigned short)(u2 >> 16));
                        sat_track(satm, v0, v1);
                        psum += v0 + v1;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = qrow + 16 * j;
#pragma unroll
            for (int src = 0; src < 2; ++src) {
                unsigned long long lo, hi;
                pieces4(qv[j][src], F16P_XS, lo, hi);
                char* d = qb + row * QP + src * 256 + (((unsigned)qc4 ^ wf_swz(row)) << 3);
                *reinterpret_cast<unsigned long long*>(d) = lo;
                *reinterpret_cast<unsigned long long*>(d + WFW_QPL) = hi;
            }
        }
    };

    const int g = lane >> 4, l = lane & 15, qqq = l >> 2, pp = l & 3;
    const unsigned ub = (unsigned)(pp | (qqq << 2) | ((g & 1) << 4));
    const unsigned pa0 = (unsigned)((g * WG_MB + 64 * wm + l) << 4);     // tile jm: + 256 jm bytes (an immediate)
    unsigned qa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qa[j] = (unsigned)((8 * g + qqq) * QP + wn * 256) + ((((unsigned)(4 * j)) ^ ub) << 3);
    auto tr8 = [&](const char* plane, unsigned off) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) wf_s4* lptr;
        const wf_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off));
        const wf_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off + 4 * QP));
        rg_u32x4 out;
        const unsigned long long a0 = __builtin_bit_cast(unsigned long long, lo), a1 = __builtin_bit_cast(unsigned long long, hi);
        out[0] = (unsigned)a0; out[1] = (unsigned)(a0 >> 32); out[2] = (unsigned)a1; out[3] = (unsigned)(a1 >> 32);
        return out;
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int jn = 0; jn < 8; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (r0 < r1) {
        load();
        store(0);
        __syncthreads();
        int buf = 0;
        for (long long k0 = r0; k0 < r1; k0 += WF_R, buf ^= 1) {
            const bool more = k0 + WF_R < r1;
            if (more) { advance(); load(); }
            const char* const pb = wimg + buf * WFW_BUF;
            const char* const qb = pb + 2 * WF_PPL;
#pragma unroll
            for (int hf = 0; hf < 4; ++hf) {                            // the wave's 128 columns two 16-column tiles at a time: 16 fragment registers
                rg_u32x4 q1[2], q2[2];
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) { q1[jn] = tr8(qb, qa[2 * hf + jn]); q2[jn] = tr8(qb + WFW_QPL, qa[2 * hf + jn]); }
#pragma unroll
                for (int jm = 0; jm < 4; ++jm) {
                    const rg_u32x4 p1 = *reinterpret_cast<const rg_u32x4*>(pb + pa0 + 256 * jm), p2 = *reinterpret_cast<const rg_u32x4*>(pb + WF_PPL + pa0 + 256 * jm);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        acc[jm][2 * hf + jn] = mfma_f16p(p1, q2[jn], acc[jm][2 * hf + jn]);
                        acc[jm][2 * hf + jn] = mfma_f16p(p2, q1[jn], acc[jm][2 * hf + jn]);
                        acc[jm][2 * hf + jn] = mfma_f16p(p1, q1[jn], acc[jm][2 * hf + jn]);
                    }
                }
            }
            if (more) store(buf ^ 1);
            __syncthreads();
        }
    }
    sat_commit(sat, DPRNN_SAT_WGRAD, satm, F16P_XMAX * 0.99999f);
    // acc[jm][jn][e] = dW[m][n], m = 256 mb + 64 wm + 16 jm + 4 g + e, n = 128 wn + 16 jn + l
    float* const out = a.part + (long long)ks * Mpad * Npad;
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = WG_MB * mb + 64 * wm + 16 * jm + 4 * g + e;
#pragma unroll
            for (int jn = 0; jn < 8; ++jn) out[(long long)m * Npad + 128 * wn + 16 * jn + l] = acc[jm][jn][e] * inv;
        }
    if (a.part_b) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(wimg);
        red[tid] = psum;
        __syncthreads();
        if (tid < 256) a.part_b[(long long)ks * Mpad + WG_MB * mb + tid] = (red[tid] + red[tid + 256]) * (1.f / sg);
    }
}

// row ranges of the wide form: two sibling workgroups (gate-row halves) per range, two ranges per CU over the launch
static void wgrad_lstm_wide_plan(long long rows, long long& splits, long long& per) {
    splits = num_cus();
    const long long max_splits = (rows + 63) / 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    per = (rows + splits - 1) / splits;
    per = (per + 31) / 32 * 32;
    splits = (rows + per - 1) / per;
}
static bool wgrad_lstm_wide() {
    static const bool on = [] { const char* e = getenv("DPRNN_WGRAD_LSTM_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}

// dW[m*ldm + n*ldn] = (accumulate ? dW : 0) + sum_s part[s][m][n];  db likewise (two optional destinations: the
// LSTM's b_ih and b_hh receive the same gradient).  A workgroup of 4 waves sums 64 float4 column groups: wave w adds the
// partials s = w, w+4, ... (eight independent loads in flight), the four wave sums are added in wave order -- a fixed
// order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256)
void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_b, int splits,
                         int Mpad, int Npad, int M, int N, float* __restrict__ dW, long long ldm,
                         long long ldn, float* __restrict__ db0, float* __restrict__ db1, int accumulate) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n4 = N / 4;                                   // N is a multiple of 4
    const int nmat = M * n4, nb4 = part_b ? M / 4 : 0;      // float4 items: the matrix, then the bias row
    const int item = blockIdx.x * 64 + lane;
    const bool mat = item < nmat, bias = !mat && item < nmat + nb4;
    const float* src = nullptr;
    long long stride = 0;
    int m = 0, n = 0;
    if (mat) { m = item / n4; n = 4 * (item % n4); src = part + (long long)m * Npad + n; stride = (long long)Mpad * Npad; }
    else if (bias) { m = 4 * (item - nmat); src = part_b + m; stride = Mpad; }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (src) {
        int k = w;
        for (; k + 28 < splits; k += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (k + 4 * u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < splits; k += 4) s += *reinterpret_cast<const f32x4*>(src + k * stride);
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && src) {
        s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (mat) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* d = dW + m * ldm + (n + e) * ldn;
                *d = accumulate ? *d + s[e] : s[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (db0) db0[m + e] = accumulate ? db0[m + e] + s[e] : s[e];
                if (db1) db1[m + e] = accumulate ? db1[m + e] + s[e] : s[e];
            }
        }
    }
}


// ---- The Linear after a recurrence (dprnn.py:88,96): its whole backward, and the GroupNorm backward's second pass in front of it, in ONE
// pass over the rows (round 5).  Before: gn_bwd_dz (dY, Z -> dZ), the data gradient (dZ -> dH) and the weight gradient (dZ, H -> dW, db)
// were three launches that moved 9 slab units of 512 B per row (2 R + 1 W, 1 R + 2 W, 1 R + 2 R), each paced by HBM at ~5 TB/s.  Here a
// workgroup owns a contiguous range of rows and streams dY, Z and H through it once (4 R), forms
//     dZ = rstd_b (gamma dy - m1_b - x_hat m2_b),   x_hat = (Z - mu_b) rstd_b                        (the norm's second pass; never stored)
// while it stages the chunk, as two f16 pieces of S_z dZ in a row-major LDS image, and multiplies that image twice:
//     dW[128][256] += dZ^T H      rows = the MFMA k dimension: ds_read_b64_tr_b16 fragments of the dZ and H images (wgrad_f16_kernel's scheme)
//     dH[32][256]   = dZ W        rows = the MFMA m dimension: plain 16-byte A fragments of the same dZ image; this wave's 32 columns of W
//                                 stay in registers as piece fragments for the whole kernel
// and writes dH (2 W): 6 units.  db = column sums of dZ ride in the staging registers; max|dH| (the BPTT's gradient scale) is followed in
// the epilogue that stores dH.  S_z: dZ does not exist before this kernel, so its scale comes from a BOUND the first norm pass leaves
// (dprnn_gn_bwd_stats_ex: per utterance max|gamma dy| and max|x_hat|): |dZ| <= rstd (max|gamma dy| + |m1| + max|x_hat| |m2|), placed at
// 2^6 like every other gradient -- a bound a few times above the true maximum costs as many of the ~2^-20 relative floor bits.
constexpr int LB_R = 32;
constexpr int LB_PPL = LB_R * 128 * 2, LB_QPL = LB_R * 256 * 2;              // bytes of one piece plane of a dZ / H chunk
constexpr int LB_BUF = 2 * LB_PPL + 2 * LB_QPL;                              // 48 KB
constexpr int LB_MAXB = 1024;                                                // utterances in the (mu, rstd, m1, m2) table

struct LinBwdArgs {
    const float* dY; const float* Z; const double* stats; const double* red; const unsigned* absmax; const float* gamma;
    float eps; int B; long long rows_per_utt;
    const float* H; const float* W; float* dH;
    long long rows, rows_per_split; int splits;
    float* part; float* part_b;                   // [splits][128][256], [splits][128]
    unsigned* hmax;                               // max |dH| as float bits, or null
};

__device__ __forceinline__ float lb_pow2_to(float m, int target) {
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
    ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
    return ldexpf(1.f, target - ex);
}

__global__ __launch_bounds__(WG_THREADS, 1)
void lin_bwd_f16_kernel(LinBwdArgs a, int* sat) {
    extern __shared__ __attribute__((aligned(16))) char limg[];             // [buf 2][dZ piece 0 | dZ piece 1 | H piece 0 | H piece 1] | table
    f32x4* const utab = reinterpret_cast<f32x4*>(limg + 2 * LB_BUF);
    __shared__ float redm[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = blockIdx.x;                                                // one workgroup per row range (grid = splits)
    const long long r0 = (long long)ks * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > a.rows) r1 = a.rows;

    // ---- per-utterance constants of the norm's backward, the bound on |dZ|, max |W|
    const double cnt = (double)a.rows_per_utt * 128.0;
    float zb = 0.f, wmx = 0.f;
    for (int b = tid; b < a.B; b += WG_THREADS) {
        const double mean = a.stats[2 * b] / cnt;
        double var = a.stats[2 * b + 1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)a.eps));
        const float m1 = (float)(a.red[2 * b] / cnt), m2 = (float)(a.red[2 * b + 1] / cnt);
        utab[b] = f32x4{mu, rs, m1, m2};
        zb = fmaxf(zb, rs * (__uint_as_float(a.absmax[2 * b]) + fabsf(m1) + __uint_as_float(a.absmax[2 * b + 1]) * fabsf(m2)));
    }
    for (int idx = tid; idx < 128 * 256 / 4; idx += WG_THREADS) {
        const f32x4 v = reinterpret_cast<const f32x4*>(a.W)[idx];
        wmx = fmaxf(wmx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { zb = fmaxf(zb, __shfl_xor(zb, o, 64)); wmx = fmaxf(wmx, __shfl_xor(wmx, o, 64)); }
    if (lane == 0) { redm[0][wave] = zb; redm[1][wave] = wmx; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { zb = fmaxf(zb, redm[0][i]); wmx = fmaxf(wmx, redm[1][i]); }
    const float sz = lb_pow2_to(zb, 6), sw = lb_pow2_to(wmx, 14);
    const float inv_w = 1.f / (sz * F16P_XS), inv_h = 1.f / (sz * sw);

    // ---- this wave's 32 columns of W as B fragments: lane (l, g) holds W[k = 32 s + 8 g .. + 7][n = 32 wave + 16 nt + l]
    const int g = lane >> 4, l = lane & 15, qq = l >> 2, pp = l & 3;
    rg_u32x4 wf[4][2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float* src = a.W + (long long)(32 * s + 8 * g) * 256 + 32 * wave + 16 * nt + l;
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = src[e * 256] * sw; hi[e] = src[(4 + e) * 256] * sw; }
            f16_pieces8(lo, hi, wf[s][nt][0], wf[s][nt][1]);
        }

    // ---- staging: dY / Z rows prow + 16 j (j < 2), float4 column chunk pc4; H rows qrow + 8 j (j < 4), chunk qc4
    const int pc4 = tid & 31, prow = tid >> 5;
    const int qc4 = tid & 63, qrow = tid >> 6;
    unsigned poff[2], qoff[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) poff[j] = (unsigned)(((prow + 16 * j) * 128 + 4 * pc4) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) qoff[j] = (unsigned)(((qrow + 8 * j) * 256 + 4 * qc4) * 4);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + 4 * pc4);
    // utterance of this thread's two dY / Z rows, advanced by 32 rows per chunk without dividing
    int ub_[2]; long long rem_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long row = r0 + prow + 16 * j;
        ub_[j] = (int)(row / a.rows_per_utt);
        rem_[j] = row % a.rows_per_utt;
    }
    f32x4 dyv[2], zv[2], qv[4], psum = {0.f, 0.f, 0.f, 0.f};
    int bcur[2] = {0, 0};
    bool vcur[2] = {false, false};
    auto load = [&](long long row0) __attribute__((always_inline)) {
        const long long leftp = (r1 - row0) * 128LL * 4, leftq = (r1 - row0) * 256LL * 4;
        const unsigned rp = leftp > 0x7fffffffLL ? 0x7fffffffu : (unsigned)leftp, rq = leftq > 0x7fffffffLL ? 0x7fffffffu : (unsigned)leftq;
        const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dY + row0 * 128), 0, rp, 0x00020000);
        const __amdgpu_buffer_rsrc_t zs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Z + row0 * 128), 0, rp, 0x00020000);
        const __amdgpu_buffer_rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.H + row0 * 256), 0, rq, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dyv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ys, poff[j], 0, 0));
            zv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zs, poff[j], 0, 0));
            vcur[j] = row0 + prow + 16 * j < r1;
            bcur[j] = ub_[j] < a.B ? ub_[j] : a.B - 1;
            rem_[j] += LB_R;                                 // the next chunk's row
            while (rem_[j] >= a.rows_per_utt) { rem_[j] -= a.rows_per_utt; ++ub_[j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) qv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hs, qoff[j], 0, 0));
    };
    unsigned satm = 0u;
    auto pieces4 = [&](const f32x4& v, float s, unsigned long long& lo, unsigned long long& hi) __attribute__((always_inline)) {
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(v[e] * s, -F16P_XMAX, F16P_XMAX);
        unsigned a1, a2, b1, b2;
        f16_pieces2(c[0], c[1], a1, a2);
        f16_pieces2(c[2], c[3], b1, b2);
        lo = (unsigned long long)a1 | ((unsigned long long)b1 << 32);
        hi = (unsigned long long)a2 | ((unsigned long long)b2 << 32);
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
        char* const pb = limg + buf * LB_BUF;
        char* const qb = pb + 2 * LB_PPL;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = prow + 16 * j;
            const f32x4 t = utab[bcur[j]];                   // (mu, rstd, m1, m2)
            f32x4 dz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (zv[j][e] - t[0]) * t[1];
                const float v = t[1] * (gm[e] * dyv[j][e] - t[2] - xh * t[3]);
                dz[e] = vcur[j] ? v : 0.f;                   // rows beyond the range read as zero and must stay zero
                psum[e] += dz[e];
            }
            sat_track(satm, dz);
            unsigned long long lo, hi;
            pieces4(dz, sz, lo, hi);
            char* d = pb + row * 256 + (((unsigned)pc4 ^ wf_swz(row)) << 3);
            *reinterpret_cast<unsigned long long*>(d) = lo;
            *reinterpret_cast<unsigned long long*>(d + LB_PPL) = hi;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = qrow + 8 * j;
            unsigned long long lo, hi;
            pieces4(qv[j], F16P_XS, lo, hi);
            char* d = qb + row * 512 + (((unsigned)qc4 ^ wf_swz(row)) << 3);
            *reinterpret_cast<unsigned long long*>(d) = lo;
            *reinterpret_cast<unsigned long long*>(d + LB_QPL) = hi;
        }
    };

    // dW: wave (wm, wn) owns dZ columns 64 wm .. + 63 (M) x H columns 64 wn .. + 63 (N); transposed fragment reads as in wgrad_f16_kernel
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned ubx = (unsigned)(pp | (qq << 2) | ((g & 1) << 4));
    unsigned pa[4], qa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pa[j] = (unsigned)((8 * g + qq) * 256) + ((((unsigned)(16 * wm + 4 * j)) ^ ubx) << 3);
        qa[j] = (unsigned)((8 * g + qq) * 512) + ((((unsigned)(16 * wn + 4 * j)) ^ ubx) << 3);
    }
    auto tr8 = [&](const char* plane, unsigned off, int pitch) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) wf_s4* lptr;
        const wf_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off));
        const wf_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(__attribute__((address_space(3))) char*)(plane + off + 4 * pitch));
        rg_u32x4 out;
        const unsigned long long a0 = __builtin_bit_cast(unsigned long long, lo), a1 = __builtin_bit_cast(unsigned long long, hi);
        out[0] = (unsigned)a0; out[1] = (unsigned)(a0 >> 32); out[2] = (unsigned)a1; out[3] = (unsigned)(a1 >> 32);
        return out;
    };
    // dH: A fragment of row tile mt, slab s: lane (l, g) reads the 16 bytes of dZ row 16 mt + l, columns 32 s + 8 g .. + 7 (units 8 s + 2 g, + 1:
    // the swizzle flips bits 2-4 of the unit index, the pair stays together)
    auto ha_off = [&](int mt, int s) __attribute__((always_inline)) {
        return (unsigned)((16 * mt + l) * 256) + ((((unsigned)(8 * s + 2 * g)) ^ wf_swz(16 * mt + l)) << 3);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    float hmx = 0.f;

    if (r0 < r1) {
        load(r0);
        store(0);
        __syncthreads();
        int buf = 0;
        for (long long row0 = r0; row0 < r1; row0 += LB_R, buf ^= 1) {
            const bool more = row0 + LB_R < r1;
            if (more) load(row0 + LB_R);
            const char* const pb = limg + buf * LB_BUF;
            const char* const qb = pb + 2 * LB_PPL;
            {   // dW += dZ^T H
                rg_u32x4 q1[4], q2[4];
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) { q1[jn] = tr8(qb, qa[jn], 512); q2[jn] = tr8(qb + LB_QPL, qa[jn], 512); }
#pragma unroll
                for (int jm = 0; jm < 4; ++jm) {
                    const rg_u32x4 p1 = tr8(pb, pa[jm], 256), p2 = tr8(pb + LB_PPL, pa[jm], 256);
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn) {
                        acc[jm][jn] = mfma_f16p(p1, q2[jn], acc[jm][jn]);
                        acc[jm][jn] = mfma_f16p(p2, q1[jn], acc[jm][jn]);
                        acc[jm][jn] = mfma_f16p(p1, q1[jn], acc[jm][jn]);
                    }
                }
            }
            {   // dH = dZ W: rows 16 mt + 4 g + e, columns 32 wave + 16 nt + l
                f32x4 ah[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) ah[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const unsigned off = ha_off(mt, s);
                        const rg_u32x4 a1 = *reinterpret_cast<const rg_u32x4*>(pb + off), a2 = *reinterpret_cast<const rg_u32x4*>(pb + LB_PPL + off);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            ah[mt][nt] = mfma_f16p(a1, wf[s][nt][1], ah[mt][nt]);
                            ah[mt][nt] = mfma_f16p(a2, wf[s][nt][0], ah[mt][nt]);
                            ah[mt][nt] = mfma_f16p(a1, wf[s][nt][0], ah[mt][nt]);
                        }
                    }
                const long long lefth = (r1 - row0) * 256LL * 4;
                const __amdgpu_buffer_rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(a.dH + row0 * 256, 0,
                                                                                    lefth > 0x7fffffffLL ? 0x7fffffffu : (unsigned)lefth, 0x00020000);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const float v = ah[mt][nt][e] * inv_h;
                            hmx = fmaxf(hmx, fabsf(v));
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ds,
                                                                  (unsigned)(((16 * mt + 4 * g + e) * 256 + 32 * wave + 16 * nt + l) * 4), 0, 0);
                        }
            }
            if (more) store(buf ^ 1);
            __syncthreads();
        }
    }
    sat_commit(sat, DPRNN_SAT_WGRAD, satm, F16P_XMAX / sz);
    if (a.hmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hmx = fmaxf(hmx, __shfl_xor(hmx, o, 64));
        if (lane == 0 && hmx > 0.f) atomicMax(a.hmax, __float_as_uint(hmx));
    }
    // acc[jm][jn][e] = dW[m][n], m = 64 wm + 16 jm + 4 g + e, n = 64 wn + 16 jn + l
    float* const out = a.part + (long long)ks * 128 * 256;
#pragma unroll
    for (int jm = 0; jm < 4; ++jm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 64 * wm + 16 * jm + 4 * g + e;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) out[m * 256 + 64 * wn + 16 * jn + l] = acc[jm][jn][e] * inv_w;
        }
    if (a.part_b) {                                      // db: threads tid, tid + 32 j hold the same column chunk
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(limg);
        red[tid] = psum;
        __syncthreads();
        if (tid < 32) {
            f32x4 s = red[tid];
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                const f32x4 v = red[tid + 32 * j];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += v[e];
            }
            *reinterpret_cast<f32x4*>(a.part_b + (long long)ks * 128 + 4 * tid) = s;
        }
    }
}

}  // namespace

// row ranges: about two workgroups per CU, at least 64 rows each
static void wgrad_plan(int M, int N, long long rows, int& mblocks, int& nblocks, long long& splits, long long& per) {
    mblocks = (M + WG_MB - 1) / WG_MB;
    nblocks = (N + WG_NB - 1) / WG_NB;
    splits = (2LL * num_cus() + mblocks * nblocks - 1) / (mblocks * nblocks);
    const long long max_splits = (rows + 63) / 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    per = (rows + splits - 1) / splits;
    per = (per + 31) / 32 * 32;                        // whole chunks of either kernel (16 rows fp32, 32 rows f16 pieces)
    splits = (rows + per - 1) / per;
}

extern "C" int64_t dprnn_wgrad_workspace_floats(int M, int N, int64_t rows) {
    int mblocks, nblocks; long long splits, per;
    wgrad_plan(M, N, rows > 0 ? rows : 1, mblocks, nblocks, splits, per);
    if (M == 512 && N == 256) {                        // dprnn_wgrad_lstm_f16's 256 x 256 form splits the rows more finely
        long long ws, wp;
        wgrad_lstm_wide_plan(rows > 0 ? rows : 1, ws, wp);
        if (ws > splits) splits = ws;
    }
    return splits * ((int64_t)mblocks * WG_MB * nblocks * WG_NB + mblocks * WG_MB);
}

// One-shot selector: the NEXT dprnn_wgrad call on this thread of control multiplies in f16 pieces, scaling P by the power of two
// that the BPTT launch derived from *grad_max (device pointer to max|dH|, dprnn_rnn_bwd_gradmax_offset); cleared by that call.
static thread_local const float* g_wgrad_f16_pmax = nullptr;
extern "C" void dprnn_wgrad_use_f16_pieces(const float* grad_max) { g_wgrad_f16_pmax = grad_max; }

static int wgrad_impl(const float* pmax, const float* P, int ldp, int M, const float* Q0, int ldq0, int n0, int dt0,
                      const float* Q1, int ldq1, int n1, int dt1, int64_t rows, int64_t t_stride, int T,
                      float* dW, int64_t ldm, int64_t ldn, float* db0, float* db1, int accumulate,
                      float* workspace, void* stream) {
    if (!P || !Q0 || !dW || !workspace) { dprnn_set_error("dprnn_wgrad: null pointer"); return -1; }
    const int N = n0 + (Q1 ? n1 : 0);
    if (rows <= 0 || M <= 0 || (M % 4) || n0 <= 0 || n0 > WG_NB || (n0 % 4) || (Q1 && (n1 <= 0 || n1 > WG_NB || (n1 % 4) || n0 != WG_NB)) ||
        (ldp % 4) || (ldq0 % 4) || (Q1 && (ldq1 % 4)) || ((dt0 || dt1) && (t_stride <= 0 || T <= 0)) || t_stride > 0x7fffffffLL ||
        ldp > (1 << 20) || ldq0 > (1 << 20) || ldq1 > (1 << 20)) {
        dprnn_set_error("dprnn_wgrad: unsupported shapes M=%d n0=%d n1=%d ldp=%d ldq0=%d rows=%lld", M, n0, n1, ldp, ldq0, (long long)rows);
        return -1;
    }
    WgradArgs a{};
    a.P = P; a.ldp = ldp; a.M = M;
    a.q[0] = QDesc{Q0, ldq0, n0, dt0};
    a.q[1] = QDesc{Q1, ldq1, Q1 ? n1 : 0, dt1};
    a.rows = rows; a.t_stride = t_stride > 0 ? t_stride : 1; a.T = T > 0 ? T : 1;
    int mblocks, nblocks; long long splits, per;
    wgrad_plan(M, N, rows, mblocks, nblocks, splits, per);
    a.Mpad = mblocks * WG_MB; a.Npad = nblocks * WG_NB;
    a.mblocks = mblocks; a.nblocks = nblocks; a.splits = (int)splits;
    const unsigned grid = (unsigned)(mblocks * nblocks * ((splits + 7) / 8 * 8));
    a.rows_per_split = per;
    a.part = workspace;
    a.part_b = (db0 || db1) ? workspace + splits * (long long)a.Mpad * a.Npad : nullptr;
    int rc;
    if (pmax) {
        static std::atomic<unsigned long long> attr_mask{0};
        int attr_dev;
        if (dprnn_attr_needed(attr_mask, &attr_dev)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WF_BUF);
            if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(wgrad_f16): %s", hipGetErrorString(e)); return -2; }
            dprnn_attr_done(attr_mask, attr_dev);
        }
        hipLaunchKernelGGL(wgrad_f16_kernel, dim3(grid), dim3(WG_THREADS), 2 * WF_BUF, (hipStream_t)stream, a, pmax, dprnn_sat_ptr());
        rc = dprnn_check_launch("wgrad_f16_kernel");
    } else {
        hipLaunchKernelGGL(wgrad_kernel, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, a);
        rc = dprnn_check_launch("wgrad_kernel");
    }
    if (rc) return rc;
    const int items = M * (N / 4) + (a.part_b ? M / 4 : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((items + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       a.part, a.part_b, (int)splits, a.Mpad, a.Npad, M, N, dW, (long long)ldm, (long long)ldn, db0, db1, accumulate);
    return dprnn_check_launch("wgrad_reduce_kernel");
}

extern "C" int dprnn_wgrad(const float* P, int ldp, int M, const float* Q0, int ldq0, int n0, int dt0,
                           const float* Q1, int ldq1, int n1, int dt1, int64_t rows, int64_t t_stride, int T,
                           float* dW, int64_t ldm, int64_t ldn, float* db0, float* db1, int accumulate,
                           float* workspace, void* stream) {
    const float* const pmax = g_wgrad_f16_pmax;          // one-shot selector: consumed by this call whatever it returns
    g_wgrad_f16_pmax = nullptr;
    return wgrad_impl(pmax, P, ldp, M, Q0, ldq0, n0, dt0, Q1, ldq1, n1, dt1, rows, t_stride, T, dW, ldm, ldn, db0, db1, accumulate, workspace, stream);
}
// the same with the selector as an argument: f16_grad_max != NULL = f16 pieces with P scaled from *f16_grad_max (device float)
extern "C" int dprnn_wgrad_ex(const float* f16_grad_max, const float* P, int ldp, int M, const float* Q0, int ldq0, int n0, int dt0,
                              const float* Q1, int ldq1, int n1, int dt1, int64_t rows, int64_t t_stride, int T,
                              float* dW, int64_t ldm, int64_t ldn, float* db0, float* db1, int accumulate,
                              float* workspace, void* stream) {
    return wgrad_impl(f16_grad_max, P, ldp, M, Q0, ldq0, n0, dt0, Q1, ldq1, n1, dt1, rows, t_stride, T, dW, ldm, ldn, db0, db1, accumulate, workspace, stream);
}

// LSTM weight gradients of one direction from the quad-layout dG of a mode-3 BPTT (wgrad_lstm_f16_kernel): dWcat [512][256] =
// [dW_ih | dW_hh] (row-major, pitch 256), db0 / db1 [512] (optional, the same values: b_ih and b_hh).  dG4: this direction's block of
// dprnn_rnn_bwd's dG (dir * T * ceil(nseq / 4) * 2048 floats in); X: slab rows [.][128]; Hd: the h columns of this direction
// (H + dir * 128, pitch ldh = ndir * 128); reverse != 0: h_prev is h_{t+1}.  Sequence addressing as dprnn_rnn_fwd.  grad_max: the
// device float the BPTT launch scaled by (packedT + dprnn_rnn_bwd_gradmax_offset).  workspace: dprnn_wgrad_workspace_floats(512, 256,
// T * roundup(nseq, 32)) floats.
extern "C" int dprnn_wgrad_lstm_f16(const float* grad_max, const float* dG4, const float* X, const float* Hd, int ldh, int reverse,
                                    int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride,
                                    float* dWcat, float* db0, float* db1, float* workspace, void* stream) {
    if (!grad_max || !dG4 || !X || !Hd || !dWcat || !workspace) { dprnn_set_error("dprnn_wgrad_lstm_f16: null pointer"); return -1; }
    if (nseq <= 0 || T <= 0 || inner_n <= 0 || (ldh != 128 && ldh != 256) || outer_stride < 0 || inner_stride < 0 || t_stride < 0) {
        dprnn_set_error("dprnn_wgrad_lstm_f16: bad sizes nseq=%d T=%d inner_n=%d ldh=%d", nseq, T, inner_n, ldh);
        return -1;
    }
    WgLstmArgs a{};
    a.P4 = dG4; a.X = X; a.Hd = Hd; a.ldh = ldh; a.dt = reverse ? -1 : 1;
    a.nseq = nseq; a.nseq32 = (nseq + 31) / 32 * 32; a.nsq = (nseq + 3) / 4; a.T = T; a.inner_n = inner_n;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride;
    const long long rows = (long long)T * a.nseq32;
    int mblocks, nblocks; long long splits, per;
    const bool wide = wgrad_lstm_wide();
    if (wide) wgrad_lstm_wide_plan(rows, splits, per);
    else wgrad_plan(512, 256, rows, mblocks, nblocks, splits, per);
    a.rows_per_split = per; a.splits = (int)splits;
    a.part = workspace;
    a.part_b = (db0 || db1) ? workspace + splits * 512LL * 256 : nullptr;
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lstm_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WF_BUF);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lstm_f16_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WFW_BUF);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(wgrad_lstm_f16): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    int rc;
    if (wide) {
        const unsigned grid = (unsigned)(2 * ((splits + 7) / 8 * 8));
        hipLaunchKernelGGL(wgrad_lstm_f16_wide_kernel, dim3(grid), dim3(WG_THREADS), 2 * WFW_BUF, (hipStream_t)stream, a, grad_max, dprnn_sat_ptr());
        rc = dprnn_check_launch("wgrad_lstm_f16_wide_kernel");
    } else {
        const unsigned grid = (unsigned)(4 * ((splits + 7) / 8 * 8));
        hipLaunchKernelGGL(wgrad_lstm_f16_kernel, dim3(grid), dim3(WG_THREADS), 2 * WF_BUF, (hipStream_t)stream, a, grad_max, dprnn_sat_ptr());
        rc = dprnn_check_launch("wgrad_lstm_f16_kernel");
    }
    if (rc) return rc;
    const int items = 512 * (256 / 4) + (a.part_b ? 512 / 4 : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((items + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       a.part, a.part_b, (int)splits, 512, 256, 512, 256, dWcat, 256LL, 1LL, db0, db1, 0);
    return dprnn_check_launch("wgrad_reduce_kernel");
}


// ---- dprnn_lin_bwd_f16: GroupNorm backward (second pass) + Linear backward of one DPRNN path in one launch (lin_bwd_f16_kernel above).
// dY [rows][128] = the gradient of the norm's output (the summed residual gradient), Z the Linear's output as the forward left it, stats
// its (sum, sum of squares) per utterance, red / absmax what dprnn_gn_bwd_stats_ex left (red possibly scaled by the narrow twin's element
// count correction), H [rows][256] the BiLSTM output, W [128][256] the Linear's weight.  Writes dH [rows][256], dW [128][256], db [128],
// and (hmax != NULL) max |dH| as a float for the f16 BPTT's gradient scale.  workspace: dprnn_lin_bwd_workspace_floats(rows) floats.
static void lin_bwd_plan(long long rows, long long& splits, long long& per) {
    splits = num_cus();
    const long long max_splits = (rows + 63) / 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    per = (rows + splits - 1) / splits;
    per = (per + LB_R - 1) / LB_R * LB_R;
    splits = (rows + per - 1) / per;
}
extern "C" int64_t dprnn_lin_bwd_workspace_floats(int64_t rows) {
    long long splits, per;
    lin_bwd_plan(rows > 0 ? rows : 1, splits, per);
    return splits * (128LL * 256 + 128);
}
extern "C" int dprnn_lin_bwd_f16(const float* dY, const float* Z, const double* stats, const double* red, const unsigned* absmax,
                                 const float* gamma, float eps, int B, int64_t rows_per_utt, const float* H, const float* W,
                                 float* dH, float* dW, float* db, float* hmax, float* workspace, void* stream) {
    if (!dY || !Z || !stats || !red || !absmax || !gamma || !H || !W || !dH || !dW || !db || !workspace) {
        dprnn_set_error("dprnn_lin_bwd_f16: null pointer"); return -1;
    }
    if (B < 1 || B > LB_MAXB || rows_per_utt < 1) {
        dprnn_set_error("dprnn_lin_bwd_f16: 1 <= B <= %d utterances, rows_per_utt >= 1 (got B=%d rows_per_utt=%lld)", LB_MAXB, B, (long long)rows_per_utt);
        return -1;
    }
    const long long rows = (long long)B * rows_per_utt;
    hipStream_t st = (hipStream_t)stream;
    LinBwdArgs a{};
    a.dY = dY; a.Z = Z; a.stats = stats; a.red = red; a.absmax = absmax; a.gamma = gamma; a.eps = eps; a.B = B; a.rows_per_utt = rows_per_utt;
    a.H = H; a.W = W; a.dH = dH; a.rows = rows;
    long long splits, per;
    lin_bwd_plan(rows, splits, per);
    a.rows_per_split = per; a.splits = (int)splits;
    a.part = workspace; a.part_b = workspace + splits * 128LL * 256;
    a.hmax = reinterpret_cast<unsigned*>(hmax);
    if (hmax) {
        hipError_t e = hipMemsetAsync(hmax, 0, sizeof(float), st);
        if (e != hipSuccess) { dprnn_set_error("dprnn_lin_bwd_f16: %s", hipGetErrorString(e)); return -2; }
    }
    const int lds = 2 * LB_BUF + LB_MAXB * 16;
    static std::atomic<unsigned long long> attr_mask{0};
    int attr_dev;
    if (dprnn_attr_needed(attr_mask, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_bwd_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(lin_bwd_f16): %s", hipGetErrorString(e)); return -2; }
        dprnn_attr_done(attr_mask, attr_dev);
    }
    hipLaunchKernelGGL(lin_bwd_f16_kernel, dim3((unsigned)splits), dim3(WG_THREADS), lds, st, a, dprnn_sat_ptr());
    int rc = dprnn_check_launch("lin_bwd_f16_kernel");
    if (rc) return rc;
    const int items = 128 * (256 / 4) + 128 / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((items + 63) / 64), dim3(256), 0, st,
                       a.part, a.part_b, (int)splits, 128, 256, 128, 256, dW, 256LL, 1LL, db, static_cast<float*>(nullptr), 0);
    return dprnn_check_launch("wgrad_reduce_kernel");
}

// ------------------------------------------------------------------------------------------------ data gradients
// dA[row*lda + 0:nd] = dOut[row][0:kd] @ W, W = [W0 (k0 rows); W1] row-major [kd][ldw]: (kd, nd) in {(128,256), (128,128),
// (64,128), (256,128), (128,64)} -- intra/inter Linear (dprnn.py:88,96), end_conv1x1 (:160), out+gate convs (:157-158),
// bottleneck conv (:135), speaker-encoder convs (dprnn_spe.py:18-19,27,117-122; 256 -> 256 as two 128-column halves:
// offset W0 and dA by the column, lda = 256).
// One-shot selector, like dprnn_wgrad_use_f16_pieces: the NEXT dprnn_dgrad call of shape (128 -> 256) multiplies in f16 pieces
// with dOut scaled by the power of two that puts *grad_max (device pointer to max|dOut|, dprnn_absmax) at 2^6.
static thread_local const float* g_dgrad_f16_amax = nullptr;
extern "C" void dprnn_dgrad_use_f16_pieces(const float* grad_max) { g_dgrad_f16_amax = grad_max; }
// One-shot as well: the NEXT dprnn_dgrad call also leaves max|dA| in *slot (device float) -- what dprnn_rnn_bwd's f16 form needs of
// its dH (dprnn_rnn_bwd_use_gradmax), measured in the epilogue that writes dH instead of by a pass over it.  The f16 form does it
// in its epilogue; the fp32 forms run dprnn_absmax afterwards (dA must then be dense: lda == nd).
static thread_local float* g_dgrad_absmax_out = nullptr;
extern "C" void dprnn_dgrad_store_absmax(float* slot) { g_dgrad_absmax_out = slot; }
extern "C" int dprnn_absmax(const float* x, int64_t n, float* out, void* stream);

static int dgrad_impl(const float* amax, float* max_out, const float* dOut, int kd, const float* W0, int ldw0, int k0, const float* W1, int ldw1,
                      float* dA, int lda, int nd, int64_t rows, void* stream);
extern "C" int dprnn_dgrad(const float* dOut, int kd, const float* W0, int ldw0, int k0, const float* W1, int ldw1,
                           float* dA, int lda, int nd, int64_t rows, void* stream) {
    const float* const amax = g_dgrad_f16_amax;         // one-shot selectors: consumed by this call whatever it returns
    g_dgrad_f16_amax = nullptr;
    float* const max_out = g_dgrad_absmax_out;
    g_dgrad_absmax_out = nullptr;
    return dgrad_impl(amax, max_out, dOut, kd, W0, ldw0, k0, W1, ldw1, dA, lda, nd, rows, stream);
}
// the same with the selectors as arguments: f16_grad_max != NULL = the (128 -> 256) product in f16 pieces with dOut scaled from
// *f16_grad_max; absmax_out != NULL = also leave max|dA| there
extern "C" int dprnn_dgrad_ex(const float* f16_grad_max, float* absmax_out, const float* dOut, int kd, const float* W0, int ldw0, int k0,
                              const float* W1, int ldw1, float* dA, int lda, int nd, int64_t rows, void* stream) {
    return dgrad_impl(f16_grad_max, absmax_out, dOut, kd, W0, ldw0, k0, W1, ldw1, dA, lda, nd, rows, stream);
}
static int dgrad_impl(const float* amax, float* max_out, const float* dOut, int kd, const float* W0, int ldw0, int k0, const float* W1, int ldw1,
                      float* dA, int lda, int nd, int64_t rows, void* stream) {
    if (!dOut || !W0 || !dA || rows <= 0 || (k0 < kd && !W1) || lda < nd) { dprnn_set_error("dprnn_dgrad: null pointer / rows / lda"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    if (amax && kd == 128 && nd == 256 && (lda % 4) == 0) {
        if (max_out) {
            hipError_t e = hipMemsetAsync(max_out, 0, sizeof(float), st);
            if (e != hipSuccess) { dprnn_set_error("dprnn_dgrad: %s", hipGetErrorString(e)); return -2; }
        }
        PlainRows<1> al{dOut, kd};
        EpTStore<1, 16> ep{dA, lda, reinterpret_cast<unsigned*>(max_out)};
        return launch_rowgemm_f16t<128, 256, 1, true, true, false>("dgrad128x256_f16", W0, ldw0, k0, W1, ldw1, (long long)rows, al, ep, st, amax);
    }
    if (max_out) {
        if (lda != nd) { dprnn_set_error("dprnn_dgrad: dprnn_dgrad_store_absmax needs a dense dA (lda == nd) outside the f16 form"); return -1; }
        const int rc = dgrad_impl(nullptr, nullptr, dOut, kd, W0, ldw0, k0, W1, ldw1, dA, lda, nd, rows, stream);
        return rc ? rc : dprnn_absmax(dA, rows * nd, max_out, stream);
    }
    // row tiles per wave: 4 x 16 rows while the accumulators fit (ND <= 128), 2 x 16 for the 256-wide outputs
#define DG_CASE(KD, ND, RT)                                                                                          \
    if (kd == KD && nd == ND) {                                                                                      \
        PlainRows<RT> al{dOut, kd};                                                                                  \
        EpStore<RT, ND / 16> ep{dA, lda};                                                                               \
        return launch_rowgemm<KD, ND, RT, true>("dgrad" #KD "x" #ND, W0, ldw0, k0, W1, ldw1, (long long)rows, al, ep, st); \
    }
    DG_CASE(128, 256, 2)
    DG_CASE(128, 128, 4)
    DG_CASE(64, 128, 4)
    DG_CASE(256, 128, 4)
    DG_CASE(128, 64, 4)
#undef DG_CASE
    dprnn_set_error("dprnn_dgrad: unsupported (kd, nd) = (%d, %d)", kd, nd);
    return -1;
}

// ------------------------------------------------------------------------------------------------ mask head: conv2d + PReLU backward
// Gradient of  U[(b,l)] = sum_{(s,k) -> l} (W2d prelu(X[b][s][k]) + b2d)  (dprnn.py:173-178,203-217) wrt X:
//   dX[(b,s,k)][c] = (dU[(b, l = s*P + k - K)][0:256] @ W2d)[c] * (x > 0 ? 1 : alpha),   zero where l is outside [0, L)
// and dalpha += sum (dU @ W2d) * x * [x <= 0]  (fp64 atomics into a scratch scalar).
namespace {
template <int RT>
struct SegGatherRows {         // A row (b,s,k) = dU row (b, s*P + k - K) or zeros
    const float* dU; int L, K, P, S;
    const float* p[RT];
    __device__ void init(long long row0, int r, int q, long long M) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = row0 + 16 * rt + r;
            if (row >= M) row = M - 1;
            const int k = (int)(row % K);
            const long long bs = row / K;
            const int s = (int)(bs % S), b = (int)(bs / S);
            const int l = s * P + k - K;
            p[rt] = (l >= 0 && l < L) ? dU + ((long long)b * L + l) * 256 + 4 * q : nullptr;
        }
    }
    __device__ f32x4 load(int rt, int g) const {
        return p[rt] ? *reinterpret_cast<const f32x4*>(p[rt] + 16 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

template <int RT>
struct EpPreluBwd {
    const float* X; const float* alpha_p; float* dX; double* dalpha;
    float part;
    __device__ void begin() { part = 0.f; }
    __device__ void end(int lane) {
        const double s = wave_sum((double)part);
        if (lane == 0) atomicAdd(dalpha, s);
    }
    __device__ void operator()(f32x4 (&acc)[RT][8], long long row0, int r, int q, int lane, long long M) {
        const float alpha = *alpha_p;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long row = row0 + 16 * rt + 4 * q + e;
                if (row < M) {
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) {
                        const float x = X[row * 128 + 16 * ct + r], g = acc[rt][ct][e];
                        dX[row * 128 + 16 * ct + r] = x > 0.f ? g : alpha * g;
                        if (!(x > 0.f)) part = fmaf(g, x, part);
                    }
                }
            }
    }
};
}  // namespace

extern "C" int dprnn_head_conv_bwd(const float* dU, const float* W2d, const float* X, const float* prelu_w, float* dX,
                                   double* dalpha, int B, int L, int K, int P, int S, void* stream) {
    if (!dU || !W2d || !X || !prelu_w || !dX || !dalpha) { dprnn_set_error("dprnn_head_conv_bwd: null pointer"); return -1; }
    if (K < 1 || P < 1 || P > K || S != (L + K) / P + 1) { dprnn_set_error("dprnn_head_conv_bwd: bad geometry"); return -1; }
    constexpr int RT = 2;
    SegGatherRows<RT> al{dU, L, K, P, S};
    EpPreluBwd<RT> ep{X, prelu_w, dX, dalpha, 0.f};
    return launch_rowgemm<256, 128, RT, true>("head_conv_bwd", W2d, 128, 256, nullptr, 0, (long long)B * S * K, al, ep, (hipStream_t)stream);
}
