// Probe: are 16-byte global loads/stores legal at 4-byte-aligned addresses on gfx950 (needed for odd-width maps)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float* y, const float* x, int off, int n4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        float4 v = *reinterpret_cast<const float4*>(x + off + 4 * i);
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        *reinterpret_cast<float4*>(y + off + 4 * i) = v;
    }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n + 8), o(n + 8);
    for (int i = 0; i < n + 8; ++i) h[i] = (float)(i % 1000);
    float *dx, *dy;
    hipMalloc(&dx, (n + 8) * 4); hipMalloc(&dy, (n + 8) * 4);
    hipMemcpy(dx, h.data(), (n + 8) * 4, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        hipMemset(dy, 0, (n + 8) * 4);
        hipLaunchKernelGGL(k, dim3(n / 4 / 256), dim3(256), 0, 0, dy, dx, off, n / 4);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), dy, (n + 8) * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) if (o[off + i] != h[off + i] + 1.f) ++bad;
        printf("offset %d floats: %s, mismatches %d\n", off, hipGetErrorString(e), bad);
    }
    return 0;
}
