// Victim side of tools/probes/pk_beside_conv_h8.py: packed-fp32 instruction forms checked bit-for-bit against their scalar twins, launched on a
// caller-supplied stream (the aggressor — the library's real conv_h8_kernel — runs on another stream of the same process).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/bin/libpkcheck.so tools/probes/pk_check_lib.hip
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void pk_check(unsigned long long* bad, int iters, const float* __restrict__ seed) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {seed[t & 1023], seed[(t + 17) & 1023]}, y = {seed[(t + 5) & 1023], seed[(t + 9) & 1023]}, z = {seed[(t + 3) & 1023], seed[(t + 11) & 1023]};
    const f32x2 sc = {seed[blockIdx.x & 1023], seed[(blockIdx.x + 1) & 1023]};      // block-uniform: an SGPR pair
    unsigned long long n = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 p;
        float s0, s1;
        if (OP == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(y.x), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(y.y), "v"(z.y));
        } else if (OP == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(p) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(y.x), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(y.x), "v"(z.y));
        } else if (OP == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(y.y), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(y.y), "v"(z.y));
        } else if (OP == 3) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "s"(sc), "v"(y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "s"(sc.x), "v"(y.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "s"(sc.y), "v"(y.y));
        } else if (OP == 4) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "s"(sc), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "s"(sc.x), "v"(y.x), "v"(z.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "s"(sc.y), "v"(y.y), "v"(z.y));
        } else if (OP == 6) {                               // src0's LOW half for both results (what `pair += scalar` compiles to in conv_h8_kernel's lean epilogue)
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(x), "v"(y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(y.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(x.x), "v"(y.y));
        } else if (OP == 7) {                               // the butterfly of the fp32 Winograd transform: (a.x + b.y, a.x - b.y)
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(x), "v"(y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(y.y));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s1) : "v"(x.x), "v"(y.y));
        } else if (OP == 8) {                               // src0's HIGH half for the low result
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(p) : "v"(x), "v"(y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(x.y), "v"(y.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(y.y));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(y.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(y.y));
        }
        n += (__float_as_uint(p.x) != __float_as_uint(s0)) + (__float_as_uint(p.y) != __float_as_uint(s1));
        x = f32x2{s0 * 0.5f + 0.25f, s1 * 0.5f - 0.125f};
        y = f32x2{y.y * 0.999f + 0.001f, y.x * 1.001f - 0.001f};
    }
    if (n) atomicAdd(bad, n);
}
extern "C" int pk_check_launch(int op, unsigned long long* bad, int iters, const float* seed, int blocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
        case 0: hipLaunchKernelGGL(pk_check<0>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 1: hipLaunchKernelGGL(pk_check<1>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 2: hipLaunchKernelGGL(pk_check<2>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 3: hipLaunchKernelGGL(pk_check<3>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 4: hipLaunchKernelGGL(pk_check<4>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 6: hipLaunchKernelGGL(pk_check<6>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 7: hipLaunchKernelGGL(pk_check<7>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        case 8: hipLaunchKernelGGL(pk_check<8>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
        default: hipLaunchKernelGGL(pk_check<5>, dim3(blocks), dim3(256), 0, st, bad, iters, seed); break;
    }
    return (int)hipGetLastError();
}
