// Packed-fp32 VALU beside matrix-core waves: does v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 stay bit-equal to its scalar twin when ANOTHER
// kernel (another stream of the same process) keeps the matrix cores busy?  Found while chasing run-to-run differences of the 16-bit path
// (tools/probes/h8_two_streams.py): the streaming kernels that hipcc had SLP-packed into v_pk_* gave sporadically different results beside
// conv_h8_kernel; built with -fno-slp-vectorize they did not.
// Result (MI355X, ROCm 7.2): only the forms with op_sel set on SRC1 (src1's high half for the low result) go wrong, and only beside an aggressor whose
// bf16 MFMAs are fed from LDS (KIND 2 below; the register-only spinners KIND 0 / 1 leave every form intact): ~2e4 wrong results of 8.4e10.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/pk_beside_mfma tools/probes/pk_beside_mfma.hip && tools/probes/bin/pk_beside_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND> __global__ __launch_bounds__(256) void mfma_spin(float* out, int iters) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    if (KIND == 0) {                                     // bf16 32x32x16
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    } else if (KIND == 2) {                              // bf16 32x32x16 with both operands re-read from LDS (ds_read_b128) every step, as conv_h8's K loop does
        extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
        for (int i = threadIdx.x; i < 9 * 1024; i += 256) lds_u[i] = 0x3c003c00u + (unsigned)i * 7u;
        __syncthreads();
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* l4 = reinterpret_cast<const u32x4*>(lds_u);
        const int lane = threadIdx.x & 63;
        for (int i = 0; i < iters; ++i) {
            const int o = (i & 7) * 256;
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, l4[o + lane]), a1 = __builtin_bit_cast(bf16x8, l4[o + 64 + lane]);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, l4[o + 128 + lane]), b1 = __builtin_bit_cast(bf16x8, l4[o + 192 + lane]);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
            if ((i & 15) == 15) __syncthreads();
        }
    } else {                                             // fp32 32x32x2
        const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// OP 0: fma, 1: mul, 2: add.  Every step evaluates the packed instruction and its two scalar twins on the same operands and compares bits.
template <int OP> __global__ __launch_bounds__(256) void pk_check(unsigned long long* bad, int iters, const float* __restrict__ seed) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {seed[t & 1023], seed[(t + 17) & 1023]}, y = {seed[(t + 5) & 1023], seed[(t + 9) & 1023]}, z = {seed[(t + 3) & 1023], seed[(t + 11) & 1023]};
    const f32x2 sc = {seed[blockIdx.x & 1023], seed[(blockIdx.x + 1) & 1023]};      // block-uniform: lives in an SGPR pair
    unsigned long long n = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 p;
        float s0, s1;
        if (OP == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(y.x), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(y.y), "v"(z.y));
        } else if (OP == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(y.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(y.y));
        } else if (OP == 3) {                               // src1's LOW half for both results (what hipcc's SLP packing of  acc += v[e] * w  emits)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(p) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(y.x), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(y.x), "v"(z.y));
        } else if (OP == 4) {                               // src1's HIGH half for both results
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(y.y), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(y.y), "v"(z.y));
        } else if (OP == 5) {                               // a scalar-register pair as src0
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "s"(sc), "v"(y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "s"(sc.x), "v"(y.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "s"(sc.y), "v"(y.y));
        } else if (OP == 6) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "s"(sc), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "s"(sc.x), "v"(y.x), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "s"(sc.y), "v"(y.y), "v"(z.y));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(y.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(y.y));
        }
        n += (__float_as_uint(p.x) != __float_as_uint(s0)) + (__float_as_uint(p.y) != __float_as_uint(s1));
        // next operands: bounded, data dependent
        x = f32x2{s0 * 0.5f + 0.25f, s1 * 0.5f - 0.125f};
        y = f32x2{y.y * 0.999f + 0.001f, y.x * 1.001f - 0.001f};
    }
    if (n) atomicAdd(bad, n);
}

template <int OP> static unsigned long long run(int aggressor, const float* seed, float* sink, unsigned long long* bad, hipStream_t sa, hipStream_t sb) {
    hipMemsetAsync(bad, 0, 8, sb);
    for (int r = 0; r < 40; ++r) {
        // 36 KiB of (unused) LDS per block: four blocks per CU, as conv_h8_kernel runs — half of every SIMD's wave slots stay free for the other stream
        if (aggressor == 0) hipLaunchKernelGGL(mfma_spin<0>, dim3(2048), dim3(256), 36 * 1024, sa, sink, 2000);
        if (aggressor == 1) hipLaunchKernelGGL(mfma_spin<1>, dim3(2048), dim3(256), 36 * 1024, sa, sink, 1000);
        if (aggressor == 2) hipLaunchKernelGGL(mfma_spin<2>, dim3(2048), dim3(256), 36 * 1024, sa, sink, 2000);
        hipLaunchKernelGGL(pk_check<OP>, dim3(1024), dim3(256), 0, sb, bad, 4000, seed);
    }
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    return h;
}

int main() {
    float hs[1024];
    for (int i = 0; i < 1024; ++i) hs[i] = 0.37f + 0.001f * i;
    float *seed, *sink; unsigned long long* bad;
    hipMalloc(&seed, sizeof(hs)); hipMalloc(&sink, 2048 * 256 * 4); hipMalloc(&bad, 8);
    hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    const char* agg[4] = {"bf16 MFMA 32x32x16 on the other stream", "fp32 MFMA 32x32x2 on the other stream", "bf16 MFMA fed from LDS on the other stream", "nothing on the other stream"};
    const double total = 40.0 * 1024 * 256 * 4000 * 2;
    for (int a = 0; a < 4; ++a) {
        const int which = a == 3 ? -1 : a;
        const unsigned long long r0 = run<0>(which, seed, sink, bad, sa, sb), r1 = run<1>(which, seed, sink, bad, sa, sb), r2 = run<2>(which, seed, sink, bad, sa, sb);
        const unsigned long long r3 = run<3>(which, seed, sink, bad, sa, sb), r4 = run<4>(which, seed, sink, bad, sa, sb), r5 = run<5>(which, seed, sink, bad, sa, sb);
        const unsigned long long r6 = run<6>(which, seed, sink, bad, sa, sb);
        printf("%-42s mismatching results of %.2e: v_pk_fma_f32 %llu   v_pk_mul_f32 %llu   v_pk_add_f32 %llu   pk_fma op_sel_hi:[1,0,1] %llu   pk_fma op_sel:[0,1,0] %llu   "
               "pk_mul SGPR src %llu   pk_fma SGPR src %llu\n", agg[a], total, r0, r1, r2, r3, r4, r5, r6);
    }
    return 0;
}
