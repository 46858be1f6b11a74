// Probe: does v_mfma_f32_16x16x32_f16 keep f16 subnormal inputs, and what does the float -> f16 conversion the compiler
// emits for a vector convert round to?   hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, float* out) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row = lane % 16][k = 8 * (lane / 16) + i]; put one subnormal in k = 0 of every row, B[k = 0][col] = 2^10
    if (lane < 16) { a[0] = (_Float16)in[0]; b[0] = (_Float16)in[1]; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
    // conversions
    f2 v = {in[2], in[3]};
    h2 hv = __builtin_convertvector(v, h2);
    if (lane == 0) { out[1] = (float)hv[0]; out[2] = (float)hv[1]; out[3] = (float)(_Float16)in[4]; }
}

int main() {
    float h_in[8] = {ldexpf(1.f, -20), 1024.f, 1.f + ldexpf(3.f, -12), -(1.f + ldexpf(3.f, -12)), ldexpf(1.f, -16) * 1.0001f, 0, 0, 0};
    float *d_in, *d_out, h_out[4];
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, sizeof h_out);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("mfma f16: 2^-20 (subnormal) * 2^10 = %g (kept: %g, flushed: 0)\n", h_out[0], ldexp(1.0, -10));
    printf("convert 1+3*2^-12 -> %.10g / %.10g  (RNE: 1.0009765625 [ties-to-even up], RTZ: 1.0)\n", h_out[1], h_out[2]);
    printf("convert 2^-16*1.0001 -> %.10g (subnormal kept: ~1.5259e-05)\n", h_out[3]);
    return 0;
}
