// l2i_pggan.hip — streaming kernels of the PGGAN-256 generator (BASELINE config 1; reference graphs/pggan/model_256.py): PixelNorm fused
// with the LeakyReLU that always follows it (model_256.py:78-84,128-150), its backward, nearest-neighbour 2x upsampling
// (F.upsample(scale_factor=2), model_256.py:240) and the 2x2 box filter that is both the upsample's adjoint and the graph's bilinear
// halving of the generator output (graphs/pggan/transform_base.py:320: bilinear with align_corners=False at an exact factor 2 is the
// mean of each 2x2 window).  All HBM-bound: one pass over the tensor each (PixelNorm: the channel column of a pixel is read twice, the
// second time from L2 — a 512-channel column of 4 pixels is 8 KiB per lane group).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

// y[b,c,p] = lrelu(x[b,c,p] * rsqrt(mean_c x[b,c,p]^2 + eps), slope)          (slope = 1: plain PixelNorm)
// V = pixels per thread (4: 16-byte accesses, HW % 4 == 0; 1: any HW, e.g. the [B, 511] latent code with HW = 1)
template <int V>
__global__ __launch_bounds__(256) void pixelnorm_act_kernel(float* __restrict__ y, const float* __restrict__ x, int C, long long HW, long long nvec,
                                                            float eps, float slope) {
    const long long per_b = HW / V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long b = i / per_b, p = (i - b * per_b) * V;
        const float* xp = x + b * C * HW + p;
        float* yp = y + b * C * HW + p;
        float ss[V];
#pragma unroll
        for (int v = 0; v < V; ++v) ss[v] = 0.f;
        for (int c = 0; c < C; ++c) {
            if constexpr (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xp + (long long)c * HW);
                ss[0] += t.x * t.x; ss[1] += t.y * t.y; ss[2] += t.z * t.z; ss[3] += t.w * t.w;
            } else {
                const float t = xp[(long long)c * HW];
                ss[0] += t * t;
            }
        }
        float r[V];
#pragma unroll
        for (int v = 0; v < V; ++v) r[v] = sqrtf(ss[v] / (float)C + eps);                 // the reference DIVIDES by sqrt(mean + 1e-8): so does this
        for (int c = 0; c < C; ++c) {
            if constexpr (V == 4) {
                float4 t = *reinterpret_cast<const float4*>(xp + (long long)c * HW);
                t.x /= r[0]; t.y /= r[1]; t.z /= r[2]; t.w /= r[3];
                t.x = t.x > 0.f ? t.x : t.x * slope; t.y = t.y > 0.f ? t.y : t.y * slope;
                t.z = t.z > 0.f ? t.z : t.z * slope; t.w = t.w > 0.f ? t.w : t.w * slope;
                *reinterpret_cast<float4*>(yp + (long long)c * HW) = t;
            } else {
                float t = xp[(long long)c * HW] / r[0];
                yp[(long long)c * HW] = t > 0.f ? t : t * slope;
            }
        }
    }
}

// backward of the above: with n = x r, g' = gy * (n > 0 ? 1 : slope):   dx = r * (g' - n * mean_c(g' n)) = r g' - x r^3 sum_c(g' x) / C
template <int V>
__global__ __launch_bounds__(256) void pixelnorm_act_bwd_kernel(float* __restrict__ dx, const float* __restrict__ gy, const float* __restrict__ x, int C,
                                                                long long HW, long long nvec, float eps, float slope) {
    const long long per_b = HW / V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long b = i / per_b, p = (i - b * per_b) * V;
        const float* xp = x + b * C * HW + p;
        const float* gp = gy + b * C * HW + p;
        float* dp = dx + b * C * HW + p;
        float ss[V], sg[V];
#pragma unroll
        for (int v = 0; v < V; ++v) ss[v] = sg[v] = 0.f;
        for (int c = 0; c < C; ++c) {
            float xv[V], gv[V];
            if constexpr (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xp + (long long)c * HW);
                const float4 g = *reinterpret_cast<const float4*>(gp + (long long)c * HW);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
            } else {
                xv[0] = xp[(long long)c * HW]; gv[0] = gp[(long long)c * HW];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                ss[v] += xv[v] * xv[v];
                sg[v] += (xv[v] > 0.f ? gv[v] : gv[v] * slope) * xv[v];
            }
        }
        float r[V], k[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            r[v] = 1.0f / sqrtf(ss[v] / (float)C + eps);
            k[v] = r[v] * r[v] * r[v] * sg[v] / (float)C;
        }
        for (int c = 0; c < C; ++c) {
            float xv[V], gv[V], o[V];
            if constexpr (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xp + (long long)c * HW);
                const float4 g = *reinterpret_cast<const float4*>(gp + (long long)c * HW);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
            } else {
                xv[0] = xp[(long long)c * HW]; gv[0] = gp[(long long)c * HW];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) o[v] = r[v] * (xv[v] > 0.f ? gv[v] : gv[v] * slope) - xv[v] * k[v];
            if constexpr (V == 4) *reinterpret_cast<float4*>(dp + (long long)c * HW) = make_float4(o[0], o[1], o[2], o[3]);
            else dp[(long long)c * HW] = o[0];
        }
    }
}

// y[pl, 2i + a, 2j + b] = scale * x[pl, i, j]: one thread = two input pixels -> two rows of one float4
__global__ __launch_bounds__(256) void upsample2x_nearest_kernel(float* __restrict__ y, const float* __restrict__ x, int H, int W, long long n2, float scale) {
    const int W2 = W >> 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
        const long long row = i / W2;                      // (plane, input row)
        const int j = (int)(i - row * W2) * 2;
        const long long pl = row / H;
        const int iy = (int)(row - pl * H);
        const float2 v = *reinterpret_cast<const float2*>(x + row * W + j);
        const float4 o = make_float4(v.x * scale, v.x * scale, v.y * scale, v.y * scale);
        float* yp = y + (pl * 2 * H + 2 * iy) * (2LL * W) + 2 * j;
        *reinterpret_cast<float4*>(yp) = o;
        *reinterpret_cast<float4*>(yp + 2 * W) = o;
    }
}

// y[pl, i, j] = scale * (x[2i,2j] + x[2i,2j+1] + x[2i+1,2j] + x[2i+1,2j+1]): one thread = two outputs from two rows of one float4
__global__ __launch_bounds__(256) void pool2x2_kernel(float* __restrict__ y, const float* __restrict__ x, int OH, int OW, long long n2, float scale) {
    const int W2 = OW >> 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
        const long long row = i / W2;                      // (plane, output row)
        const int j = (int)(i - row * W2) * 2;
        const long long pl = row / OH;
        const int oy = (int)(row - pl * OH);
        const float* xp = x + (pl * 2 * OH + 2 * oy) * (2LL * OW) + 2 * j;
        const float4 a = *reinterpret_cast<const float4*>(xp);
        const float4 b = *reinterpret_cast<const float4*>(xp + 2 * OW);
        *reinterpret_cast<float2*>(y + row * OW + j) = make_float2(scale * ((a.x + a.y) + (b.x + b.y)), scale * ((a.z + a.w) + (b.z + b.w)));
    }
}

extern "C" int l2i_pixelnorm_act_f32(float* y, const float* x, int B, int C, int64_t HW, float eps, float slope, void* stream) {
    if (!y || !x || B <= 0 || C <= 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "pixelnorm_act: null/empty tensor");
    const bool v4 = (HW % 4) == 0 && (((uintptr_t)y | (uintptr_t)x) % 16) == 0;
    const long long nvec = (long long)B * (HW / (v4 ? 4 : 1));
    if (v4) hipLaunchKernelGGL((pixelnorm_act_kernel<4>), dim3(l2i_grid_for(nvec, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, x, C, (long long)HW, nvec, eps, slope);
    else hipLaunchKernelGGL((pixelnorm_act_kernel<1>), dim3(l2i_grid_for(nvec, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, x, C, (long long)HW, nvec, eps, slope);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_pixelnorm_act_bwd_f32(float* dx, const float* gy, const float* x, int B, int C, int64_t HW, float eps, float slope, void* stream) {
    if (!dx || !gy || !x || B <= 0 || C <= 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "pixelnorm_act_bwd: null/empty tensor");
    const bool v4 = (HW % 4) == 0 && (((uintptr_t)dx | (uintptr_t)gy | (uintptr_t)x) % 16) == 0;
    const long long nvec = (long long)B * (HW / (v4 ? 4 : 1));
    if (v4) hipLaunchKernelGGL((pixelnorm_act_bwd_kernel<4>), dim3(l2i_grid_for(nvec, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, dx, gy, x, C, (long long)HW, nvec, eps, slope);
    else hipLaunchKernelGGL((pixelnorm_act_bwd_kernel<1>), dim3(l2i_grid_for(nvec, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, dx, gy, x, C, (long long)HW, nvec, eps, slope);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_upsample2x_nearest_f32(float* y, const float* x, int64_t planes, int H, int W, float scale, void* stream) {
    if (!y || !x || planes <= 0 || H <= 0 || W <= 0) return l2i_set_error(L2I_E_ARG, "upsample2x_nearest: null/empty tensor");
    if ((W % 2) != 0 || (((uintptr_t)y) % 16) != 0 || (((uintptr_t)x) % 8) != 0)
        return l2i_set_error(L2I_E_UNSUPPORTED, "upsample2x_nearest: even widths and aligned tensors only");
    const long long n2 = (long long)planes * H * (W / 2);
    hipLaunchKernelGGL(upsample2x_nearest_kernel, dim3(l2i_grid_for(n2, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, x, H, W, n2, scale);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_pool2x2_f32(float* y, const float* x, int64_t planes, int OH, int OW, float scale, void* stream) {
    if (!y || !x || planes <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "pool2x2: null/empty tensor");
    if ((OW % 2) != 0 || (((uintptr_t)x) % 16) != 0 || (((uintptr_t)y) % 8) != 0)
        return l2i_set_error(L2I_E_UNSUPPORTED, "pool2x2: even output widths and aligned tensors only");
    const long long n2 = (long long)planes * OH * (OW / 2);
    hipLaunchKernelGGL(pool2x2_kernel, dim3(l2i_grid_for(n2, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, x, OH, OW, n2, scale);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
