// Sustained fp32-MFMA ceiling of this MI355X: register-only v_mfma_f32_16x16x4_f32 loop on random data,
// every CU busy, long enough for the clock to settle.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak
// usage: mfma_peak [waves_per_simd=1] [valu_fma_per_mfma=0] [zero_data=0] [accumulators=16 (4, 8 or 16: distance between dependent MFMAs)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NV plain v_fma_f32 (independent registers) follow every MFMA: do fp32 VALU ops ride in the MFMA shadow or cost MFMA issue time?
template <int NACC, int NV>
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xFFFF]; b[i] = in[(tid * 16 + 8 + i) & 0xFFFF]; }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a[i] * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[(k + i) & 7], acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + j) & 7]) : "v"(b[j & 7]), "v"(a[(j + 1) & 7]));
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int zero = argc > 3 ? atoi(argv[3]) : 0;
    const int nv = argc > 2 ? atoi(argv[2]) : 0;      // VALU fma per MFMA (0,1,2,4,6)
    const int nacc = argc > 4 ? atoi(argv[4]) : 16;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * wps;                 // 256 threads = 4 waves = 1 wave per SIMD per block
    float *in, *out;
    (void)hipMalloc(&in, 65536 * 4); (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = zero ? 0.f : (rand() / (float)RAND_MAX - 0.5f);
    (void)hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;                      // 40000 * 8 * 16 MFMAs * 32 cycles ~ 68 ms at 2.4 GHz per wave
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        if (nacc == 4) hipLaunchKernelGGL((mfma_loop<4, 0>), dim3(blocks), dim3(256), 0, 0, in, out, iters * 4);
        else if (nacc == 8) hipLaunchKernelGGL((mfma_loop<8, 0>), dim3(blocks), dim3(256), 0, 0, in, out, iters * 2);
        else switch (nv) {
            case 0: hipLaunchKernelGGL((mfma_loop<16, 0>), dim3(blocks), dim3(256), 0, 0, in, out, iters); break;
            case 1: hipLaunchKernelGGL((mfma_loop<16, 1>), dim3(blocks), dim3(256), 0, 0, in, out, iters); break;
            case 2: hipLaunchKernelGGL((mfma_loop<16, 2>), dim3(blocks), dim3(256), 0, 0, in, out, iters); break;
            case 4: hipLaunchKernelGGL((mfma_loop<16, 4>), dim3(blocks), dim3(256), 0, 0, in, out, iters); break;
            default: hipLaunchKernelGGL((mfma_loop<16, 6>), dim3(blocks), dim3(256), 0, 0, in, out, iters); break;
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 8 * 16 * 2048.0;
        printf("rep %d: %.2f ms  %.1f MFMA-TFLOP/s  (waves/SIMD %d, %d v_fma per MFMA, %s data, %d accumulators)\n", rep, ms,
               flop / ms / 1e9, wps, nv, zero ? "zero" : "random", nacc);
    }
    return 0;
}
