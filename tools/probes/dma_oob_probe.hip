// Probe: does buffer_load_dword ... lds write ZERO to LDS for lanes whose offset is out of the descriptor's range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const float* x, float* out, int n) {
    __shared__ float lds[256];
    lds[threadIdx.x] = -123.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)n * 4u, 0x00020000);
    const unsigned voff = (threadIdx.x & 1) ? 0xFFFFFF00u : threadIdx.x * 4u;      // odd lanes: out of range
    const unsigned ldsa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 64);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(ldsa)) : "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
    float *x, *o; (void)hipMalloc(&x, 1024); (void)hipMalloc(&o, 1024);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
    (void)hipMemcpy(x, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, o, 256);
    (void)hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%g ", h[i]);
    printf("... %g %g\n", h[254], h[255]);
    return 0;
}
