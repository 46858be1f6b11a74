// Joules per matrix FLOP by MFMA shape on a power-capped MI355X: the recurrence step's energy is 46-56 % f16 MFMAs
// (profiles/r04_energy_budget.txt), issued as v_mfma_f32_16x16x32_f16.  The 32x32x16 form does twice the FLOPs per instruction with
// the same operand registers (half the operand fetch per FLOP, a quarter of the accumulator tiles): does it cost fewer joules per FLOP?
// Every CU runs one 512-thread workgroup (two waves per SIMD, as the recurrence kernel does) looping register-operand MFMAs on random
// f16 data with 64 accumulator registers per lane in both shapes; launches are queued for ~4 s and the socket's energy counter
// (rocm-smi --showenergycounter) is read before and after.  Also the bf16 16x16x32 form and the 16x16x16 f16 form for reference.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_shape_energy.hip -o tools/probes/mfma_shape_energy.bin ; run: the binary
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 16x16x32 f16 (16 tiles of 4 registers, 4 A x 4 B fragments)     16 MFMAs x 16384 FLOP per k-step of 32
// SHAPE 1: 32x32x16 f16 (4 tiles of 16 registers, 2 A x 2 B fragments x 2 k-halves)  8 MFMAs x 32768 FLOP per k-step of 32
// SHAPE 2: 16x16x32 bf16 (as 0)
// SHAPE 3: 16x16x16 f16 (legacy form: 32 MFMAs x 8192 FLOP)
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (_Float16)in[(tid * 64 + i * 8 + j) & 0xFFFF];
            b[i][j] = (_Float16)in[(tid * 64 + 32 + i * 8 + j) & 0xFFFF];
        }
    if constexpr (SHAPE == 0 || SHAPE == 2) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i >> 2]), __builtin_bit_cast(bf16x8, b[i & 3]), acc[i], 0, 0, 0);
            }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (SHAPE == 1) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * kh + (i >> 1)], b[2 * kh + (i & 1)], acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f16x8 av = a[i >> 2], bv = b[i & 3];
                    const f16x4 ah = kh ? f16x4{av[4], av[5], av[6], av[7]} : f16x4{av[0], av[1], av[2], av[3]};
                    const f16x4 bh = kh ? f16x4{bv[4], bv[5], bv[6], bv[7]} : f16x4{bv[0], bv[1], bv[2], bv[3]};
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[i], 0, 0, 0);
                }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[tid] = s;
}

static double energy_uj() {
    FILE* p = popen("rocm-smi --showenergycounter 2>/dev/null", "r");
    if (!p) return -1.0;
    char line[512];
    double v = -1.0;
    while (fgets(line, sizeof line, p)) {
        const char* k = strstr(line, "Accumulated Energy (uJ):");
        if (k) v = atof(k + strlen("Accumulated Energy (uJ):"));
    }
    pclose(p);
    return v;
}
static void power_clock(char* buf, size_t n) {
    FILE* p = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power \\(W\\)|sclk' | tr -s ' ' | tr '\\n' ';'", "r");
    buf[0] = 0;
    if (!p) return;
    if (!fgets(buf, (int)n, p)) buf[0] = 0;
    pclose(p);
}

template <int SHAPE>
static void run(const char* name, const float* in, float* out, int blocks, double seconds) {
    const int iters = 40000;                                    // one k-step of 32 over the wave's 64 x 64 outputs per iteration
    const double flop_per_launch = (double)blocks * 8 * iters * 16 * 16384.0;
    hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(512), 0, 0, in, out, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(512), 0, 0, in, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms1; (void)hipEventElapsedTime(&ms1, e0, e1);
    const int launches = (int)(seconds * 1e3 / ms1) + 1;
    const double j0 = energy_uj();
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(512), 0, 0, in, out, iters);
    (void)hipEventRecord(e1);
    char pc[256];
    std::this_thread::sleep_for(std::chrono::milliseconds((int)(seconds * 500)));
    power_clock(pc, sizeof pc);                                 // mid-run power and clock
    (void)hipEventSynchronize(e1);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double j1 = energy_uj();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = flop_per_launch * launches;
    printf("%-22s %4d launches %8.1f ms  %7.1f TFLOP/s  energy %8.1f J over %.2f s wall (%.0f W)  %6.3f pJ/FLOP   mid-run: %s\n", name, launches, ms,
           flops / (ms * 1e-3) / 1e12, (j1 - j0) * 1e-6, wall, (j1 - j0) * 1e-6 / wall, (j1 - j0) * 1e-6 / flops * 1e12, pc);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount;
    float *in, *out;
    (void)hipMalloc(&in, 65536 * 4); (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = ((rand() % 2001) - 1000) * 1e-5f;      // |v| <= 0.01, random mantissas
    (void)hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    printf("%d CUs, one 512-thread workgroup per CU (two waves per SIMD), register operands, ~%.0f s per shape\n", blocks, seconds);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("f16 16x16x32", in, out, blocks, seconds);
        run<1>("f16 32x32x16", in, out, blocks, seconds);
        run<2>("bf16 16x16x32", in, out, blocks, seconds);
        run<3>("f16 16x16x16 (legacy)", in, out, blocks, seconds);
    }
    return 0;
}
