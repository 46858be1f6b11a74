// OPTIONAL bf16-split gate GEMM, second structure (see recur_split.hip for the arithmetic): the first split kernel
// is bound by the L2 weight stream (~65 GB/s per CU, every wave pulling its private 192 KB per step).  Here the
// workgroup is organised like a GEMM instead:
//   * the four waves split the 64 SEQUENCES (16 each) and every wave computes all 4 x 128 gate columns of its rows,
//     so h of a row is produced and consumed by the same wave: the x|h image is wave-private (fp32, split into bf16
//     pieces when the A fragment is read) and needs no workgroup barrier;
//   * the weights are shared: one stage = (k-slab of 32, gate) = 8 unit tiles x NPC pieces x 1 KB goes through a
//     3-deep LDS ring (each wave copies a quarter of it: global_load_dwordx4 one stage ahead, ds_write_b128 after the
//     barrier; already in B-fragment order), each byte once per workgroup-step instead of once per wave -- half the
//     L2 bytes per FLOP of the 32-sequence kernel;
//   * one barrier per stage orders the ring: barrier, publish stage i+1, request stage i+2, compute stage i.
#include "recur.h"
using namespace recur;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int NSLAB = 8;
constexpr int IMG_PITCH = 260;                  // floats per image row: x(128) | h(128) | 4 pad (16-byte aligned rows)
constexpr int IMG_BYTES = 64 * IMG_PITCH * 4;   // 66 560

__device__ __forceinline__ unsigned bf16_bits(float v) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = __builtin_convertvector(f32x2{v, 0.f}, bf16x2);
    return __builtin_bit_cast(unsigned, p) & 0xFFFFu;
}
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// packed[dir][slab][gate][unit tile j][piece][lane][4 u32]: piece of Wcat[gate*128 + 16*j + r][32*slab + 8*q + jj]
__global__ void lstm_pack_split3_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                        const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                        const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                        const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r,
                                        unsigned* __restrict__ packed, float* __restrict__ bsum, int ndir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_dir = NSLAB * 4 * 8 * 64 * 4;
    if (idx >= ndir * per_dir) return;
    const int dw = idx & 3, lane = (idx >> 2) & 63, j = (idx >> 8) & 7, gate = (idx >> 11) & 3, slab = (idx >> 13) & 7;
    const int dir = idx >> 16;
    const int r = lane & 15, q = lane >> 4;
    const int row = gate * HD + 16 * j + r;
    const float* wi = dir ? w_ih_r : w_ih;
    const float* wh = dir ? w_hh_r : w_hh;
    unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 32 * slab + 8 * q + 2 * dw + half;
        const float v = k < F ? wi[row * F + k] : wh[row * HD + (k - F)];
        const unsigned p1 = bf16_bits(v);
        const float r1 = v - __builtin_bit_cast(float, p1 << 16);
        const unsigned p2 = bf16_bits(r1);
        const float r2 = r1 - __builtin_bit_cast(float, p2 << 16);
        const unsigned p3 = bf16_bits(r2);
        out[0] |= p1 << (16 * half); out[1] |= p2 << (16 * half); out[2] |= p3 << (16 * half);
    }
    const size_t base = (((((size_t)dir * NSLAB + slab) * 4 + gate) * 8 + j) * 3) * 256 + lane * 4 + dw;
#pragma unroll
    for (int p = 0; p < 3; ++p) packed[base + (size_t)p * 256] = out[p];
    if (idx < ndir * 4 * HD) {
        const int d = idx / (4 * HD), n = idx % (4 * HD);
        bsum[idx] = (d ? b_ih_r : b_ih)[n] + (d ? b_hh_r : b_hh)[n];
    }
}

template <int NP, bool FAST>
__global__ __launch_bounds__(256, 1)
void bilstm_split3_kernel(RecurArgs a) {
    constexpr int NPC = NP == 6 ? 3 : 2;               // bf16 pieces per operand
    constexpr int STAGE_KB = 8 * NPC;                  // 1-KB DMA pieces per stage
    constexpr int DPW = STAGE_KB / 4;                  // ... per wave
    constexpr int RING_BYTES = STAGE_KB * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // image [64][260] f32 | ring [3][STAGE_KB KB]
    float* const img = reinterpret_cast<float*>(smem);
    char* const ring = smem + IMG_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * 64 + 16 * wave;      // first sequence of this wave
    const int T = a.T;

    auto rowbase = [&](int seq) {
        return (long long)(seq / a.inner_n) * a.outer_stride + (long long)(seq % a.inner_n) * a.inner_stride;
    };
    const int seqc = seq0 < a.nseq ? seq0 : 0;         // clamp for address arithmetic of empty waves
    const long long row0 = rowbase(seqc);
    const int hpitch = a.ndir * HD;
    // x staging of this wave's 16 rows: float4 f = lane + 64*m -> row f>>5 (two rows per m), floats 4*(f&31)
    unsigned xoff[8];
    const int c4 = lane & 31;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int row = (lane >> 5) + 2 * m;
        const int seq = seq0 + row;
        xoff[m] = (unsigned)((rowbase(seq < a.nseq ? seq : seqc) - row0) * F + c4 * 4) * 4u;
    }
    unsigned hoff[4];
    bool hval[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int seq = seq0 + 4 * q + e;
        hval[e] = seq < a.nseq;
        hoff[e] = (unsigned)((rowbase(hval[e] ? seq : seqc) - row0) * hpitch + dir * HD + r) * 4u;
    }
    const bool fuse = a.Z != nullptr;
    f32x4 gam = {1.f, 1.f, 1.f, 1.f}, bet = {0.f, 0.f, 0.f, 0.f};
    float mu[8], rs[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { mu[m] = 0.f; rs[m] = 0.f; }
    if (fuse) {
        gam = *reinterpret_cast<const f32x4*>(a.gamma + 4 * c4);
        bet = *reinterpret_cast<const f32x4*>(a.beta + 4 * c4);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int seq = seq0 + (lane >> 5) + 2 * m;
            const int b = (seq < a.nseq ? seq : seqc) / a.seqs_per_utt;
            const double cnt = (double)a.rows_per_utt * (double)F;
            const double mean = a.stats[2 * b] / cnt;
            double var = a.stats[2 * b + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            mu[m] = (float)mean;
            rs[m] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
    }
    auto tbase = [&](long long t) { return (row0 + t * a.t_stride); };

    float bias[4][8];                                  // pre-scaled for the activation (see recur.hip)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float b = a.bsum[dir * 4 * HD + g * HD + 16 * j + r];
            bias[g][j] = FAST ? b * (g == 2 ? 2.8853900817779268f : -1.4426950408889634f) : b;
        }

    float* const myimg = img + (16 * wave) * IMG_PITCH;           // this wave's 16 rows
    // stage x(t) (pending norm applied) into the wave-private image rows, fp32
    auto stage_x = [&](long long t) __attribute__((always_inline)) {
        const char* xs = reinterpret_cast<const char*>(a.X + tbase(t) * F);
        const char* zs = reinterpret_cast<const char*>(a.Z + tbase(t) * F);
        char* os = reinterpret_cast<char*>(a.Xout + tbase(t) * F);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xs + xoff[m]);
            if (fuse) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(zs + xoff[m]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + ((z[e] - mu[m]) * rs[m] * gam[e] + bet[e]);
                if (dir == 0) *reinterpret_cast<f32x4*>(os + xoff[m]) = v;
            }
            *reinterpret_cast<f32x4*>(myimg + ((lane >> 5) + 2 * m) * IMG_PITCH + 4 * c4) = v;
        }
    };

    // weight stream: stage st (0..31 within a step, same bytes every step).  Register-staged: this wave's DPW 1-KB
    // pieces of a stage are loaded to VGPRs one stage ahead and written to the ring right after the barrier
    // (an LDS-DMA version of the same ring cost ~100 issue cycles per piece and was slower).
    const char* const wsrc = reinterpret_cast<const char*>(a.Wp) + (size_t)dir * NSLAB * 4 * 8 * 3 * 1024;
    u32x4 wreg[DPW];
    auto load_stage = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int piece_idx = wave * DPW + i;                  // (tile j, piece p) = divmod by NPC
            const int j = piece_idx / NPC, p = piece_idx % NPC;
            wreg[i] = *reinterpret_cast<const u32x4*>(wsrc + ((size_t)(st * 8 + j) * 3 + p) * 1024 + lane * 16);
        }
    };
    auto store_stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            *reinterpret_cast<u32x4*>(ring + buf * RING_BYTES + (wave * DPW + i) * 1024 + lane * 16) = wreg[i];
    };

    // ---- prologue
    stage_x(dir ? (T - 1) : 0);
    for (int i = lane; i < 16 * 32; i += 64)            // h region of this wave's rows := 0
        *reinterpret_cast<f32x4*>(myimg + (i >> 5) * IMG_PITCH + HD + 4 * (i & 31)) = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_stage(0);
    store_stage(0);
    load_stage(1);                                       // written to the ring after the first barrier
    int gst = 0;                                         // global stage counter (ring position), wave-uniform

    for (int step = 0; step < T; ++step) {
        const long long t = dir ? (T - 1 - step) : step;
        const bool more = step + 1 < T;
        f32x4 acc[4][8];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int s = 0; s < NSLAB; ++s) {
            // A fragment of this wave's rows for k-slab s: row r, k = 32s + 8q .. +7 (fp32 -> bf16 pieces)
            const float* ap = myimg + r * IMG_PITCH + 32 * s + 8 * q;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
            u32x4 afr[NPC];
            {
                float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    u32x4 w;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned pk = pack2_bf16(v[2 * d], v[2 * d + 1]);
                        w[d] = pk;
                        v[2 * d] -= __builtin_bit_cast(float, pk << 16);
                        v[2 * d + 1] -= __builtin_bit_cast(float, pk & 0xFFFF0000u);
                    }
                    afr[pc] = w;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int buf = gst % 3;
                // barrier: every wave has finished compute(gst-1) and has written its pieces of stage gst (previous trip);
                // then publish stage gst+1 (its buffer was last read in compute(gst-2)) and request stage gst+2
                __syncthreads();
                store_stage((gst + 1) % 3);
                load_stage((s * 4 + g + 2) & 31);
                const u32x4* bp = reinterpret_cast<const u32x4*>(ring + buf * RING_BYTES) + lane;
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
                // four unit tiles at a time: the products of one accumulator are spaced by three other accumulators
                // (back-to-back MFMAs into the same accumulator would wait for each other)
#pragma unroll
                for (int jq = 0; jq < 8; jq += 4) {
                    u32x4 b[4][NPC];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc) b[jj][pc] = bp[((jq + jj) * NPC + pc) * 64];
#pragma unroll
                    for (int pr = (NP == 6 ? 0 : 3); pr < 6; ++pr)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            acc[g][jq + jj] = mfma_bf16(afr[PA[pr]], b[jj][PB[pr]], acc[g][jq + jj]);
                }
                ++gst;
            }
        }

        // ---- LSTM cell for this wave's 16 rows x 128 units, lane-local; h -> own image rows (fp32) and HBM
        char* const hs = reinterpret_cast<char*>(a.H + tbase(t) * hpitch);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ig = sigmoid_b<FAST>(acc[0][j][e], bias[0][j]);
                const float fg = sigmoid_b<FAST>(acc[1][j][e], bias[1][j]);
                const float gg = tanh_b<FAST>(acc[2][j][e], bias[2][j]);
                const float og = sigmoid_b<FAST>(acc[3][j][e], bias[3][j]);
                const float cn = fg * c[j][e] + ig * gg;
                c[j][e] = cn;
                const float h = og * tanh_f<FAST>(cn);
                myimg[(4 * q + e) * IMG_PITCH + HD + 16 * j + r] = h;
                if (hval[e]) *reinterpret_cast<float*>(hs + hoff[e] + 64 * j) = h;
            }
        if (more) stage_x(dir ? (t - 1) : (t + 1));
    }
}

template <int NP, bool FAST>
int launch_split3(const RecurArgs& a, hipStream_t st) {
    constexpr int NPC = NP == 6 ? 3 : 2;
    const size_t lds = (size_t)IMG_BYTES + 3 * 8 * NPC * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_split3_kernel<NP, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dprnn_set_error("hipFuncSetAttribute(bilstm_split3): %s", hipGetErrorString(e)); return -2; }
        attr_set = true;
    }
    dim3 grid((a.nseq + 63) / 64, a.ndir);
    hipLaunchKernelGGL((bilstm_split3_kernel<NP, FAST>), grid, dim3(256), lds, st, a);
    return dprnn_check_launch("bilstm_split3_kernel");
}

}  // namespace

int recur::launch_bilstm_split3(const RecurArgs& a, int products, bool fast, hipStream_t st) {
    if (products == 6) return fast ? launch_split3<6, true>(a, st) : launch_split3<6, false>(a, st);
    return fast ? launch_split3<3, true>(a, st) : launch_split3<3, false>(a, st);
}

extern "C" int dprnn_lstm_pack_split3(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                      const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                      void* packed, float* bsum, void* stream) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !bsum) { dprnn_set_error("dprnn_lstm_pack_split3: null pointer"); return -1; }
    const int ndir = w_ih_r ? 2 : 1;
    if (ndir == 2 && (!w_hh_r || !b_ih_r || !b_hh_r)) { dprnn_set_error("dprnn_lstm_pack_split3: incomplete reverse direction"); return -1; }
    const int n = ndir * NSLAB * 4 * 8 * 64 * 4;
    hipLaunchKernelGGL(lstm_pack_split3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, (unsigned*)packed, bsum, ndir);
    return dprnn_check_launch("lstm_pack_split3_kernel");
}
