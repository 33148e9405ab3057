// Calibration: sustained v_mfma_f32_32x32x2_f32 rate of THIS box (clock under load), no memory traffic.
// hipcc --offload-arch=gfx950 -O3 tools/csrc/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *d;
    hipMalloc(&d, 256 * 2048 * sizeof(float));
    for (int blocks_per_cu = 1; blocks_per_cu <= 3; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu, iters = 20000;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 100, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEvent_t s, e;
        hipEventCreate(&s);
        hipEventCreate(&e);
        hipEventRecord(s);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        const double flops = (double)grid * 4 /*waves*/ * iters * 32.0 * (2.0 * 32 * 32 * 2);
        printf("waves/SIMD=%d  %.2f ms  %.1f TFLOP/s  (=> %.0f MHz at 64 FLOP/clk/SIMD)\n", blocks_per_cu, ms, flops / ms / 1e9,
               flops / ms / 1e3 / (1024.0 * 64.0));
    }
    return 0;
}
