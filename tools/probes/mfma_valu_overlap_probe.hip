// Probe: do VALU instructions overlap with v_mfma_f32_32x32x2_f32 on one SIMD?  NV independent VALU ops per MFMA, 1..3 waves
// per SIMD (blocks per CU).  If time ~ MFMA time + NV * 4 cycles at every occupancy, the fp32 matrix op and the vector ALU
// are one issue resource and every VALU instruction in an MFMA loop is paid in matrix throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = a0 * j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, b0);     // independent of the MFMAs
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV> void run(float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 3; ++bpc) {
        const int grid = 256 * bpc, iters = 20000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double mfma = 4.0 * iters * bpc;            // MFMAs per SIMD
        printf("NV=%d waves/SIMD=%d: %.3f ms  %.1f cycles per MFMA at 2.4 GHz  (%.1f TFLOP/s)\n", NV, bpc, ms, ms * 1e-3 * 2.4e9 / mfma,
               2.0 * 32 * 32 * 2 * 4.0 * iters * 4 * grid / ms / 1e9);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 3 * 4 * 4);
    run<0>(d); run<2>(d); run<4>(d); run<8>(d);
    return 0;
}
