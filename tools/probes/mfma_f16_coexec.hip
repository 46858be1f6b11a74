// Can the vector ALU work under the f16 MFMAs of ANOTHER wave on the same SIMD?  (The recurrence kernel's step is MFMA time +
// cell-update time: 12.3k + 7.4k cycles per SIMD; this probe says whether that sum is a law or a scheduling artefact.)
// One 512-thread workgroup per CU = two waves per SIMD: waves 0..3 loop v_mfma_f32_16x16x32_f16 (16 accumulators, register
// operands), waves 4..7 loop a vector kernel: mode 0 nothing, 1 v_fma_f32, 2 v_exp_f32 + v_rcp_f32 (quarter rate), 3 the same
// MFMA loop (two MFMA waves per SIMD).  Prints MFMA TFLOP/s and vector instructions per ns and CU, alone and together.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f16_coexec.hip -o /tmp/coexec ; run: /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool MFMA_ON>
__global__ __launch_bounds__(512, 2) void probe(const float* in, float* out, int iters, int prio) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4 || MODE == 3) {
        if (!MFMA_ON && MODE != 3) { out[tid] = 0.f; return; }
        if (prio && wave < 4) __builtin_amdgcn_s_setprio(1);
        f16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)in[(tid * 64 + i * 8 + j) & 0xFFFF]; b[i][j] = (_Float16)in[(tid * 64 + 32 + i * 8 + j) & 0xFFFF]; }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b[(k + i) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (MODE == 1) {
        float v[8], x = in[tid & 0xFFFF], y = in[(tid + 7) & 0xFFFF];
        for (int i = 0; i < 8; ++i) v[i] = x * (float)(i + 1);
        for (int it = 0; it < iters * 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(y));
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if (MODE == 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = in[(tid + i) & 0xFFFF];
        for (int it = 0; it < iters * 2; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])); }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[tid] = s;
}

template <int MODE, bool MFMA_ON>
static float run(const float* in, float* out, int blocks, int iters, int prio) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, MFMA_ON>), dim3(blocks), dim3(512), 0, 0, in, out, iters, prio);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 1 && ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    float *in, *out;
    (void)hipMalloc(&in, 65536 * 4); (void)hipMalloc(&out, (size_t)cus * 512 * 4);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = rand() / (float)RAND_MAX - 0.5f;
    (void)hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    const double mfma_flop = (double)cus * 4 * iters * 4 * 16 * (2.0 * 16 * 16 * 32);      // 4 MFMA waves per CU
    const double fma_ops = (double)cus * 4 * iters * 8.0 * 8, trans_ops = (double)cus * 4 * iters * 2.0 * 16;
    const float m0 = run<0, true>(in, out, cus, iters, 0);
    printf("MFMA waves alone (one per SIMD):                 %.2f ms  %.0f TFLOP/s f16 dense\n", m0, mfma_flop / m0 / 1e9);
    const float m3 = run<3, true>(in, out, cus, iters, 0);
    printf("two MFMA waves per SIMD:                         %.2f ms  %.0f TFLOP/s\n", m3, 2 * mfma_flop / m3 / 1e9);
    const float f0 = run<1, false>(in, out, cus, iters, 0), t0 = run<2, false>(in, out, cus, iters, 0);
    printf("v_fma_f32 waves alone:                           %.2f ms  %.1f wave-instructions per ns and CU\n", f0, fma_ops / cus / f0 / 1e6);
    printf("v_exp_f32 + v_rcp_f32 waves alone:               %.2f ms  %.1f wave-instructions per ns and CU\n", t0, trans_ops / cus / t0 / 1e6);
    for (int prio = 0; prio < 2; ++prio) {
        const float m1 = run<1, true>(in, out, cus, iters, prio), m2 = run<2, true>(in, out, cus, iters, prio);
        printf("MFMA wave + v_fma_f32 wave per SIMD (prio %d):    %.2f ms  (sum of the two alone %.2f, longer one %.2f)\n", prio, m1, m0 + f0, m0 > f0 ? m0 : f0);
        printf("MFMA wave + exp/rcp wave per SIMD (prio %d):      %.2f ms  (sum of the two alone %.2f, longer one %.2f)\n", prio, m2, m0 + t0, m0 > t0 ? m0 : t0);
    }
    return 0;
}
