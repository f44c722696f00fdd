// l2i_half.hip — the half-precision entry points of the two reference ops.  The reference dispatches both CUDA ops with
// AT_DISPATCH_FLOATING_TYPES_AND_HALF (op/fused_bias_act_kernel.cu:79, op/upfirdn2d_kernel.cu:225); the walk-training path itself
// runs them in float32 (l2i_stream.hip), these exist so that the C ABI covers the dtypes an FFI for the ops would bind.
//
// Arithmetic follows the reference instantiation for scalar_t = c10::Half operation by operation, so the results are bit-identical
// to it: a c10::Half binary op is the float op rounded to half (round to nearest even), hence
//   fused_bias_act: x = h(x + b); y = (cond) ? x : h(x * h(alpha)); out = h(y * h(scale))     (alpha / scale are scalar_t kernel arguments)
//   upfirdn2d:      the taps sit in float shared arrays, products are float, the accumulator is scalar_t: v = h(float(v) + sx * sk) per tap,
//                   taps walked over ascending input row, then ascending input column (upfirdn2d_kernel.cu:118-123)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

__device__ __forceinline__ float hr(float v) { return __half2float(__float2half_rn(v)); }      // round a float result to half precision

__global__ __launch_bounds__(256) void fba_f16_kernel(__half* __restrict__ y, const __half* __restrict__ x, const __half* __restrict__ b,
                                                      const __half* __restrict__ ref, long long n, long long step_b, long long size_b,
                                                      int code, float alpha, float scale) {
    const float ah = hr(alpha), sh = hr(scale);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = __half2float(x[i]);
        if (b) v = hr(v + __half2float(b[(i / step_b) % size_b]));
        const float r = ref ? __half2float(ref[i]) : 0.f;
        float o;
        switch (code) {
            default:
            case 10: case 11: o = v; break;
            case 12: case 32: o = 0.f; break;
            case 30: o = (v > 0.f) ? v : hr(v * ah); break;
            case 31: o = (r > 0.f) ? v : hr(v * ah); break;
        }
        y[i] = __float2half_rn(o * sh);
    }
}

extern "C" int l2i_fused_bias_act_f16(void* y, const void* x, const void* b, const void* ref, int64_t n, int64_t step_b, int64_t size_b,
                                      int act, int grad, float alpha, float scale, void* stream) {
    if (n == 0) return L2I_OK;
    if (!y || !x || n < 0) return l2i_set_error(L2I_E_ARG, "fused_bias_act_f16: null tensor");
    if (b && (step_b <= 0 || size_b <= 0)) return l2i_set_error(L2I_E_ARG, "fused_bias_act_f16: bias needs step_b,size_b > 0");
    hipLaunchKernelGGL(fba_f16_kernel, dim3(l2i_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (__half*)y, (const __half*)x, (const __half*)b,
                       (const __half*)ref, (long long)n, (long long)step_b, (long long)size_b, act * 10 + grad, alpha, scale);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

struct UfdHalfParams {
    __half* y; const __half* x; const __half* k;
    long long major; int in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

// one thread = one output pixel of one [in_h, in_w] map (minor_dim == 1, op/upfirdn2d.py:98)
__global__ __launch_bounds__(256) void upfirdn2d_f16_kernel(const UfdHalfParams p) {
    __shared__ float sk[64];
    const int ntap = p.kh * p.kw;
    for (int t = threadIdx.x; t < ntap; t += 256) {
        const int ky = t / p.kw, kx = t - ky * p.kw;
        sk[t] = __half2float(p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)]);      // flipped: true convolution (.cu:79)
    }
    __syncthreads();
    const long long total = p.major * p.out_h * p.out_w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % p.out_w), oy = (int)((i / p.out_w) % p.out_h);
        const long long mj = i / ((long long)p.out_w * p.out_h);
        const __half* xin = p.x + mj * (long long)p.in_h * p.in_w;
        const int my = oy * p.down_y - p.pad_y0, mx = ox * p.down_x - p.pad_x0;
        float v = 0.f;                                                             // holds a half-representable value between taps
        for (int ky = 0; ky < p.kh; ++ky) {
            const int uy = my + ky;
            const int iy = uy >= 0 ? uy / p.up_y : -1;
            if (uy < 0 || iy * p.up_y != uy) continue;                             // not a tap of this output phase
            for (int kx = 0; kx < p.kw; ++kx) {
                const int ux = mx + kx;
                const int ix = ux >= 0 ? ux / p.up_x : -1;
                if (ux < 0 || ix * p.up_x != ux) continue;
                const float xv = (iy < p.in_h && ix < p.in_w) ? __half2float(xin[(long long)iy * p.in_w + ix]) : 0.f;     // zero padding enters the sum
                v = hr(v + xv * sk[ky * p.kw + kx]);
            }
        }
        p.y[i] = __float2half_rn(v);
    }
}

extern "C" int l2i_upfirdn2d_f16(void* y, const void* x, const void* k, int64_t major, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                                 int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
    if (!y || !x || !k) return l2i_set_error(L2I_E_ARG, "upfirdn2d_f16: null tensor");
    if (major <= 0 || in_h <= 0 || in_w <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d_f16: empty input");
    if (kh <= 0 || kw <= 0 || kh * kw > 64) return l2i_set_error(L2I_E_ARG, "upfirdn2d_f16: FIR must have 1..64 taps");
    if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d_f16: up/down must be positive");
    UfdHalfParams p;
    p.y = (__half*)y; p.x = (const __half*)x; p.k = (const __half*)k; p.major = major; p.in_h = in_h; p.in_w = in_w; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;       // op/upfirdn2d.py:102-103
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (p.out_h <= 0 || p.out_w <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d_f16: empty output");
    hipLaunchKernelGGL(upfirdn2d_f16_kernel, dim3(l2i_grid_for(major * p.out_h * p.out_w, 256)), dim3(256), 0, (hipStream_t)stream, p);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
