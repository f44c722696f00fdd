// Probe: sustained v_mfma_f32_32x32x2_f32 rate on this box (the ceiling every conv number is judged against).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 3; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu, iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double flop = 2.0 * 32 * 32 * 2 * 4.0 * iters * 4 /*waves*/ * grid;
            printf("blocks/CU %d rep %d: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, rep, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
