// Software-pipelined BiLSTM recurrence, 64 sequences per workgroup (gfx950).
//
// Same contraction, operand layouts and packed weights as recur.hip (see there), restructured so that the matrix
// pipe never waits for the LSTM cell arithmetic:
//   * the 64 sequences of a workgroup are two halves A, B of 32.  Phase 1 of step t runs the 16 k-groups of A's
//     gate GEMM while the *same instruction stream* evaluates the cell update of B's previous step, one
//     (seq, unit) element per k-group, scheduled into the MFMA shadows with sched_group_barrier; phase 2 swaps
//     the roles.  Each wave therefore issues MFMAs back to back for the whole step; sigmoid/tanh, the LDS write
//     of h, the HBM store of h, the x prefetch and the weight stream ride along.
//   * because a half's rows are idle while the other half computes, the LDS image needs no double buffering:
//     66 KB per workgroup (one [64][x|h|pad] image), one barrier per phase.
//   * 64 sequences per weight pass halve the L2 weight bytes per FLOP relative to the 32-sequence kernel.
#include "recur.h"
#include <type_traits>
using namespace recur;

namespace {

typedef const f32x4 __attribute__((address_space(1)))* gw_ptr;   // global address space: global_load with SGPR base, never flat_load

template <bool FAST>
__global__ __launch_bounds__(256, 1)
void bilstm64_kernel(RecurArgs a) {
    constexpr int NS = 64, NT = 2;                    // NT row tiles of 16 per half
    extern __shared__ __attribute__((aligned(16))) float XH[];   // [NS][XS]: x_t (128) | h (128) | pad

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * NS;
    const int T = a.T;
    const int hpitch = a.ndir * HD;
    const long long t_first = dir ? (long long)(T - 1) * a.t_stride : 0;
    const long long xstep = (dir ? -a.t_stride : a.t_stride) * F;
    const long long hstep = (dir ? -a.t_stride : a.t_stride) * hpitch;

    // ---- x staging: per half 32 rows x 32 float4 = 4 per thread
    const float* xptr[2][4];
    int xlds[2][4];
    bool xval[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int f = tid + 256 * m, row = 32 * hh + (f >> 5), c4 = f & 31;
            const int seq = seq0 + row;
            xval[hh][m] = seq < a.nseq;
            const int sq = xval[hh][m] ? seq : 0;
            const long long base = (long long)(sq / a.inner_n) * a.outer_stride + (long long)(sq % a.inner_n) * a.inner_stride;
            xptr[hh][m] = a.X + (base + t_first) * F + c4 * 4;
            xlds[hh][m] = row * XS + c4 * 4;
        }
    // ---- accumulator rows of this lane: seq0 + 32*hh + 16*nt + 4*q + e
    float* hptr[2][NT][4];
    bool hval[2][NT][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int seq = seq0 + 32 * hh + 16 * nt + 4 * q + e;
                hval[hh][nt][e] = seq < a.nseq;
                const int sq = hval[hh][nt][e] ? seq : 0;
                const long long hb = (long long)(sq / a.inner_n) * a.outer_stride + (long long)(sq % a.inner_n) * a.inner_stride;
                // rows past nseq (last workgroup only) store into a private dump row instead of being masked off:
                // the k-loop stays one branch-free scheduling region
                hptr[hh][nt][e] = hval[hh][nt][e]
                    ? a.H + (hb + t_first) * hpitch + dir * HD + 32 * wave + r
                    : a.dump + (32 * hh + 16 * nt + 4 * q + e) * 2 * HD + dir * HD + 32 * wave + r;
            }

    float bias[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) bias[g][u] = a.bsum[dir * 4 * HD + g * HD + 32 * wave + 16 * u + r];

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const f32x4* const wslice = reinterpret_cast<const f32x4*>(a.Wp) + (size_t)(dir * 4 + wave_u) * NGRP * 8 * 64;

    f32x4 c[2][NT][2], acc[2][NT][4][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c[hh][nt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[hh][nt][g][u] = f32x4{0.f, 0.f, 0.f, 0.f};   // zero => first fake cell gives h = c = 0
            }

    // ---- prologue: x_A(t0) -> LDS, x_B(t0) -> registers (written by the first phase), h = 0
    f32x4 xr[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (xval[0][m]) v = *reinterpret_cast<const f32x4*>(xptr[0][m]);
        float* d = XH + xlds[0][m];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        xr[1][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xval[1][m]) xr[1][m] = *reinterpret_cast<const f32x4*>(xptr[1][m]);
        xr[0][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < NS * HD; i += 256) XH[(i >> 7) * XS + HD + (i & 127)] = 0.f;

    f32x4 w[2][8];
#pragma unroll
    for (int x = 0; x < 8; ++x) w[0][x] = wslice[x * 64 + lane];
    __syncthreads();

    // One phase: gate GEMM of half HK for this step, interleaved with the cell update of half HC (its previous GEMM).
    //   write_x : xr[HC] holds the x rows HC needs for its next GEMM -> store them to LDS first
    //   load_x  : HK has a following step -> start fetching its next x rows into xr[HK]
    //   adv_h   : the cell of HC is real (not the zero-state priming of the very first phase) -> advance its output pointers
    auto phase = [&](auto hk_tag, bool write_x, bool load_x, bool adv_h) __attribute__((always_inline)) {
        constexpr int HK = decltype(hk_tag)::value, HC = 1 - HK;
        if (write_x) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float* d = XH + xlds[HC][m];
                d[0] = xr[HC][m][0]; d[1] = xr[HC][m][1]; d[2] = xr[HC][m][2]; d[3] = xr[HC][m][3];
            }
        }
        if (load_x) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                xptr[HK][m] += xstep;
                xr[HK][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (xval[HK][m]) xr[HK][m] = *reinterpret_cast<const f32x4*>(xptr[HK][m]);
            }
        }
        // GEMM accumulators of HK start from the bias; HC's accumulators hold its finished pre-activations
        f32x4 gk[NT][4][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) gk[nt][g][u] = f32x4{bias[g][u], bias[g][u], bias[g][u], bias[g][u]};

        const float* const Ak = XH + (32 * HK + r) * XS + 16 * q;
        float* const Hc = XH + (32 * HC) * XS + HD + 32 * wave + r;
        gw_ptr ws = (gw_ptr)wslice;
        asm volatile("" : "+s"(ws));               // keep the 128 load addresses as SGPR adds, not hoisted VGPR pairs
        float av[2][NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) av[0][nt][i] = Ak[nt * 16 * XS + i];

#pragma unroll
        for (int grp = 0; grp < NGRP; ++grp) {
            const int cb = grp & 1;
            gw_ptr wn = ws + ((grp + 1) & 15) * 8 * 64;
#pragma unroll
            for (int x = 0; x < 8; ++x) w[cb ^ 1][x] = wn[x * 64 + lane];
            if (grp + 1 < NGRP) {
                const float* src = Ak + 64 * ((grp + 1) >> 2) + 4 * ((grp + 1) & 3);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[cb ^ 1][nt][i] = src[nt * 16 * XS + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            gk[nt][gt][u] = mfma16(av[cb][nt][i], w[cb][gt * 2 + u][i], gk[nt][gt][u]);
            // cell update of element `grp` of half HC: (nt, u, e) = (grp>>3, (grp>>2)&1, grp&3)
            {
                const int nt = grp >> 3, u = (grp >> 2) & 1, e = grp & 3;
                const float ig = sigmoid_f<FAST>(acc[HC][nt][0][u][e]);
                const float fg = sigmoid_f<FAST>(acc[HC][nt][1][u][e]);
                const float gg = tanh_f<FAST>(acc[HC][nt][2][u][e]);
                const float og = sigmoid_f<FAST>(acc[HC][nt][3][u][e]);
                const float cn = fg * c[HC][nt][u][e] + ig * gg;
                c[HC][nt][u][e] = cn;
                const float h = og * tanh_f<FAST>(cn);
                Hc[(16 * nt + 4 * q + e) * XS + 16 * u] = h;
                hptr[HC][nt][e][16 * u] = h;
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) {          // 1 MFMA : 1 VALU, a VMEM read every 8th and a DS read every 16th slot
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                if ((k & 7) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((k & 15) == 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the pipeline distance: nothing crosses a group boundary
        }
        // publish: HK's finished pre-activations become "acc" for the next phase; advance HC's output pointers
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[HK][nt][g][u] = gk[nt][g][u];
        if (adv_h) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) hptr[HC][nt][e] += hval[HC][nt][e] ? hstep : 0;
        }
        __syncthreads();
    };

    for (int step = 0; step < T; ++step) {
        const bool more = step + 1 < T;
        // phase 1: GEMM A(step) || cell B(step-1); the x rows of B for this step were fetched in the previous phase
        phase(std::integral_constant<int, 0>{}, true, more, step > 0);
        // phase 2: GEMM B(step) || cell A(step); x rows of A for step+1 were fetched in phase 1
        phase(std::integral_constant<int, 1>{}, more, more, true);
    }
    // ---- epilogue: cell of B for the last step
    {
        float* const Hc = XH + 32 * XS + HD + 32 * wave + r;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ig = sigmoid_f<FAST>(acc[1][nt][0][u][e]);
                    const float fg = sigmoid_f<FAST>(acc[1][nt][1][u][e]);
                    const float gg = tanh_f<FAST>(acc[1][nt][2][u][e]);
                    const float og = sigmoid_f<FAST>(acc[1][nt][3][u][e]);
                    const float cn = fg * c[1][nt][u][e] + ig * gg;
                    const float h = og * tanh_f<FAST>(cn);
                    (void)Hc;
                    hptr[1][nt][e][16 * u] = h;
                }
    }
}

template <bool FAST>
int launch64(const RecurArgs& a, hipStream_t st) {
    const size_t lds = (size_t)64 * XS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm64_kernel<FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm64): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    static float* dump = nullptr;          // 64 rows x 256 floats: landing zone of the padded rows of the last tile
    if (!dump) {
        hipError_t e = hipMalloc(&dump, 64 * 2 * HD * sizeof(float));
        if (e != hipSuccess) { dprnn_set_error("hipMalloc(dump): %s", hipGetErrorString(e)); return -2; }
    }
    RecurArgs b = a;
    b.dump = dump;
    dim3 grid((a.nseq + 63) / 64, a.ndir);
    hipLaunchKernelGGL((bilstm64_kernel<FAST>), grid, dim3(256), lds, st, b);
    return dprnn_check_launch("bilstm64_kernel");
}

}  // namespace

int recur::launch_bilstm64(const RecurArgs& a, bool fast, hipStream_t st) {
    return fast ? launch64<true>(a, st) : launch64<false>(a, st);
}
