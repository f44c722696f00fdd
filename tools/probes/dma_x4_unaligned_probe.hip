// Probe: buffer_load_dwordx4 ... lds (16-byte LDS-DMA) from a source that is only 4-byte aligned (a halo tile starting at column
// ix0 - 1 of a float map), and what a 16-byte piece returns when the descriptor's range ends in its middle (per-dword range check?).
// Build: hipcc --offload-arch=gfx950 -O2 dma_x4_unaligned_probe.hip -o dma_x4_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const float* x, float* out, int shift, unsigned nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = -123.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const unsigned voff = (unsigned)(threadIdx.x * 16 + shift * 4);
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + wave * 256);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(ldsa)) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
    const int n = 1024 + 16;
    std::vector<float> h(n), o(1024);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *x, *d; (void)hipMalloc(&x, n * 4); (void)hipMalloc(&d, 4096);
    (void)hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, d, shift, (unsigned)(n * 4));
        hipError_t e = hipDeviceSynchronize();
        (void)hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) if (o[i] != h[i + shift]) ++bad;
        printf("x4 LDS-DMA, source shifted by %d floats: %s, mismatches %d (first words %g %g %g %g %g)\n", shift, hipGetErrorString(e), bad, o[0], o[1], o[2], o[3], o[4]);
    }
    // range ends in the middle of lane 10's piece (after 2 of its 4 dwords), source shift 1
    for (int cut = 1; cut < 4; ++cut) {
        const unsigned nbytes = (unsigned)((10 * 4 + 1 + cut) * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, d, 1, nbytes);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
        printf("range ends after %d dword(s) of lane 10's piece: lane 9 -> %g %g %g %g | lane 10 -> %g %g %g %g | lane 11 -> %g %g %g %g\n", cut,
               o[36], o[37], o[38], o[39], o[40], o[41], o[42], o[43], o[44], o[45], o[46], o[47]);
    }
    return 0;
}
