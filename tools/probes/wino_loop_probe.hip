// Probe: which part of the Winograd K-loop (one wave per SIMD, 16 accumulator sets) keeps the matrix pipe from peak?
// MODE 0: bare MFMAs, register operands.  1: + A operand from LDS (ds_read_b128).  2: + raw patch from LDS (no transform, B = raw).
// 3: + input transform (48 VALU per 16 MFMAs) = the kernel's compute().  4: 3 + one __syncthreads per chunk.
// 5: transform from registers (no LDS at all).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PLANE = 10 * 34, IW = 34, BM = 64, CKh = 4, NRAW = 8 * PLANE, NU4 = 8 * 4 * BM;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    for (int i = tid; i < NRAW + NU4 * 4 + 16; i += 256) smem[i] = a0 + 1e-6f * i;
    __syncthreads();
    const int wch = wave & 1, wt = wave >> 1, txl = j & 15, tyl = j >> 4;
    f32x16 acc[16];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, zero, 0, 0, 0);
    const float* rb = smem + half * CKh * PLANE + (4 * wt + 2 * tyl) * IW + 2 * txl;
    const float4* ub = reinterpret_cast<const float4*>(smem + NRAW) + half * CKh * 4 * BM + wch * 32 + j;
    const float* sc = smem + NRAW + NU4 * 4 + half * CKh;
    float ra = a0 + lane * 1e-6f, rbv = b0;
    for (int it = 0; it < iters; ++it) {
        float2 dn[4][2]; float4 an[4]; float sn = 1.f;
        auto fetch = [&](int cc) {
            if constexpr (MODE >= 2 && MODE != 5) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dn[r][0] = *reinterpret_cast<const float2*>(rb + cc * PLANE + r * IW);
                    dn[r][1] = *reinterpret_cast<const float2*>(rb + cc * PLANE + r * IW + 2);
                }
                sn = sc[cc];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { dn[r][0] = make_float2(rbv + r, rbv - r); dn[r][1] = make_float2(rbv * r, rbv + 2 * r); }
            }
            if constexpr (MODE >= 1 && MODE != 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = ub[(cc * 4 + i) * BM];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = make_float4(ra, ra + i, ra - i, ra * i);
            }
        };
        if constexpr (MODE >= 6) {
            // v for k-pair cc+1 is computed while the MFMAs of k-pair cc run (no VALU -> MFMA dependency inside a group)
            float vn[4][4]; float4 an2[4];
            auto xform = [&]() {
                float t[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float d0 = c < 2 ? (c == 0 ? dn[0][0].x : dn[0][0].y) : (c == 2 ? dn[0][1].x : dn[0][1].y);
                    const float d1 = c < 2 ? (c == 0 ? dn[1][0].x : dn[1][0].y) : (c == 2 ? dn[1][1].x : dn[1][1].y);
                    const float d2 = c < 2 ? (c == 0 ? dn[2][0].x : dn[2][0].y) : (c == 2 ? dn[2][1].x : dn[2][1].y);
                    const float d3 = c < 2 ? (c == 0 ? dn[3][0].x : dn[3][0].y) : (c == 2 ? dn[3][1].x : dn[3][1].y);
                    t[0][c] = d0 - d2; t[1][c] = d1 + d2; t[2][c] = d2 - d1; t[3][c] = d1 - d3;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { vn[i][0] = (t[i][0] - t[i][2]) * sn; vn[i][1] = (t[i][1] + t[i][2]) * sn; vn[i][2] = (t[i][2] - t[i][1]) * sn; vn[i][3] = (t[i][1] - t[i][3]) * sn; }
#pragma unroll
                for (int i = 0; i < 4; ++i) an2[i] = an[i];
            };
            fetch(0); xform(); fetch(1);
#pragma unroll
            for (int cc = 0; cc < CKh; ++cc) {
                float v[4][4]; float4 a4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a4[i] = an2[i]; for (int c = 0; c < 4; ++c) v[i][c] = vn[i][c]; }
                __builtin_amdgcn_sched_barrier(0);
                if (cc + 1 < CKh) xform();
                if (cc + 2 < CKh) fetch(cc + 2);
                if constexpr (MODE == 6) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, v[i][0], acc[i * 4 + 0], 0, 0, 0);
                    acc[i * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, v[i][1], acc[i * 4 + 1], 0, 0, 0);
                    acc[i * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, v[i][2], acc[i * 4 + 2], 0, 0, 0);
                    acc[i * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, v[i][3], acc[i * 4 + 3], 0, 0, 0);
                }
                if constexpr (MODE == 7) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // 3 VALU
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
                    }
                }
            }
            continue;
        }
        fetch(0);
#pragma unroll
        for (int cc = 0; cc < CKh; ++cc) {
            float d[4][4]; float4 a4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { d[r][0] = dn[r][0].x; d[r][1] = dn[r][0].y; d[r][2] = dn[r][1].x; d[r][3] = dn[r][1].y; }
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = an[i];
            const float s = sn;
            if (cc + 1 < CKh) fetch(cc + 1);
            __builtin_amdgcn_sched_barrier(0);
            float v[4][4];
            if constexpr (MODE >= 3) {
                float t[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { t[0][c] = d[0][c] - d[2][c]; t[1][c] = d[1][c] + d[2][c]; t[2][c] = d[2][c] - d[1][c]; t[3][c] = d[1][c] - d[3][c]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i][0] = (t[i][0] - t[i][2]) * s; v[i][1] = (t[i][1] + t[i][2]) * s; v[i][2] = (t[i][2] - t[i][1]) * s; v[i][3] = (t[i][1] - t[i][3]) * s; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[i][c] = d[i][c];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, v[i][0], acc[i * 4 + 0], 0, 0, 0);
                acc[i * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, v[i][1], acc[i * 4 + 1], 0, 0, 0);
                acc[i * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, v[i][2], acc[i * 4 + 2], 0, 0, 0);
                acc[i * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, v[i][3], acc[i * 4 + 3], 0, 0, 0);
            }
        }
        if constexpr (MODE == 4) __syncthreads();
        if constexpr (MODE == 5 || MODE == 0) { rbv += 1e-3f; ra -= 1e-3f; }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256, iters = 4000;
    const size_t lds = (NRAW + NU4 * 4 + 16) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, iters, 0.5f, 0.25f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * 32 * 32 * 2 * 64.0 * iters * 4 * grid;
        if (rep == 2) printf("mode %d: %.3f ms  %.1f TFLOP/s (MFMA flops)  err=%s\n", MODE, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 4);
    run<0>(d); run<2>(d); run<3>(d); run<6>(d); run<7>(d);
    return 0;
}
