// Probe: ds_read_b64_tr_b16 (the __builtin_amdgcn_ds_read_tr16_b64_v4i16 builtin) on a plain [row][col] image of 16-bit values:
// value(row, col) = 100 * row + col.  Lane 4q+p of each 16-lane group supplies the address of row R+q, columns C+4p .. C+4p+3;
// expected: lane i of the group receives (R+0..R+3, C+i).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(short* out) {
    __shared__ __attribute__((aligned(16))) short img[32][64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) img[i / 64][i % 64] = (short)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
    const int R = 8 * g, C = 16;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s4*>(
        (__attribute__((address_space(3))) char*)(&img[R + q][C + 4 * p])));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int want = 100 * (8 * (lane >> 4) + e) + 16 + (lane & 15);
            if (h[lane * 4 + e] != want) ++bad;
        }
    printf("lane 0: %d %d %d %d   lane 5: %d %d %d %d   lane 21: %d %d %d %d   mismatches vs (row R+e, col C+i): %d\n",
           h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23], h[84], h[85], h[86], h[87], bad);
    return 0;
}
