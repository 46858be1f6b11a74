// Probe: which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8])
//   hipcc --offload-arch=gfx950 -O2 wave_simd_map.hip -o wave_simd_map.bin && ./wave_simd_map.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    const unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    const int nb = 512;
    unsigned* d; unsigned h[nb * 8];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 4; ++b) { printf("wg %d simd ids:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u\n", (h[b * 8] >> 8) & 15); }
    return 0;
}
