// l2i_train.hip — what TRAINING a conv net needs on top of the frozen-weight kernels: the weight-gradient of a convolution and
// BatchNorm in training mode (batch statistics).  Used by the regressor-training row of the scope table: the reference's
// scene_regressor_256.py:118-171 trains torchvision's ResNet-50 (fc -> 40, BatchNorm in train mode, MSE loss, Adam over every
// parameter); the walk-training hot path itself never computes a weight gradient.
//
//   l2i_conv2d_wgrad_f32   dw[co,ci,ky,kx] += sum_{b,oy,ox} gy[b,co,oy,ox] * x[b,ci,oy*s+ky-p,ox*s+kx-p]
//                          GEMM with M = Cout, N = Cin (per tap), K = pixels, on v_mfma_f32_32x32x2_f32 (exact fp32 products): a wave
//                          owns a 32 x 32 (co, ci) tile for all taps of one kernel row (<= 7 accumulator tiles) and walks output rows two
//                          pixels per MFMA; operands come straight from global memory (a lane reads its channel's row: every 64/128-byte
//                          line is reused by the next 16..32 pixel steps out of L1), partial sums of the row groups meet in dw by atomics.
//   l2i_bn_stats_f32       per-channel sum / sum of squares in float64 (batch mean and biased variance of BatchNorm2d.forward)
//   l2i_bn_apply_f32       y = relu?(x * scale[c] + shift[c] (+ residual))
//   l2i_bn_bwd_reduce_f32  sum_dy[c], sum_dy_xhat[c] with dy = gy * (out > 0 ?), xhat = (x - mean[c]) * invstd[c]      (float64 sums)
//   l2i_bn_bwd_apply_f32   dx = gamma*invstd * (dy - sum_dy/N - xhat * sum_dy_xhat/N); optionally also writes the masked dy (the gradient
//                          of a residual branch)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
    const float* x; const float* gy; float* dw;
    int B, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad_y, pad_x;
    int mtiles, ntiles, rows_total, row_groups;
};

// block = 4 waves; block (tile, row group g): wave w walks the (sample, output row) pairs r = (g * 4 + w) + k * 4 * row_groups
template <int KW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p, int ky) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x, nt = tile % p.ntiles, mt = tile / p.ntiles;
    const int co = mt * 32 + j, ci = nt * 32 + j;                  // this lane's A row (output channel) / B column (input channel)
    const bool co_ok = co < p.Cout, ci_ok = ci < p.Cin;
    const size_t plane_x = (size_t)p.H * p.W, plane_y = (size_t)p.OH * p.OW;
    f32x16 acc[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int r = blockIdx.y * 4 + wave; r < p.rows_total; r += 4 * p.row_groups) {
        const int b = r / p.OH, oy = r - b * p.OH;
        const int iy = oy * p.stride - p.pad_y + ky;
        if (iy < 0 || iy >= p.H) continue;                                      // (wave-uniform)
        const float* gy_row = p.gy + ((size_t)b * p.Cout + (co_ok ? co : 0)) * plane_y + (size_t)oy * p.OW;
        const float* x_row = p.x + ((size_t)b * p.Cin + (ci_ok ? ci : 0)) * plane_x + (size_t)iy * p.W;
        for (int ox0 = 0; ox0 < p.OW; ox0 += 2) {
            const int ox = ox0 + half;                                          // the two lane halves take consecutive pixels (MFMA K = 2)
            const float a = (co_ok && ox < p.OW) ? gy_row[ox] : 0.f;
            const int ixb = ox * p.stride - p.pad_x;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int ix = ixb + kx;
                const float bv = (ci_ok && ox < p.OW && ix >= 0 && ix < p.W) ? x_row[ix] : 0.f;
                acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[kx], 0, 0, 0);
            }
        }
    }
    // D[i][j]: lane j = input channel, register r = output channel (r & 3) + 8 (r >> 2) + 4 half
    if (ci_ok) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o < p.Cout) unsafeAtomicAdd(p.dw + (((size_t)o * p.Cin + ci) * p.KH + ky) * p.KW + kx, acc[kx][r]);
            }
    }
}

extern "C" int l2i_conv2d_wgrad_f32(float* dw, const float* x, const float* gy, int B, int Cin, int H, int W, int Cout, int OH, int OW,
                                    int KH, int KW, int stride, int pad_y, int pad_x, void* stream) {
    if (!dw || !x || !gy) return l2i_set_error(L2I_E_ARG, "conv2d_wgrad: null tensor");
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d_wgrad: non-positive dimension");
    WgradParams p;
    p.x = x; p.gy = gy; p.dw = dw; p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
    p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.mtiles = (Cout + 31) / 32; p.ntiles = (Cin + 31) / 32;
    p.rows_total = B * OH;
    const long tiles = (long)p.mtiles * p.ntiles;
    long groups = (4096 + tiles - 1) / tiles;                      // ~4k blocks in flight; at least one row per wave where there are rows
    const long max_groups = (p.rows_total + 3) / 4;
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
    if (groups > 65535) groups = 65535;
    p.row_groups = (int)groups;
    hipStream_t st = (hipStream_t)stream;
    for (int ky = 0; ky < KH; ++ky) {                              // one kernel row per launch: <= 7 accumulator tiles per wave
        const dim3 grid((unsigned)tiles, (unsigned)groups);
        if (KW == 1) hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, st, p, ky);
        else if (KW == 3) hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), 0, st, p, ky);
        else if (KW == 7) hipLaunchKernelGGL(conv_wgrad_kernel<7>, grid, dim3(256), 0, st, p, ky);
        else return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wgrad: built for kernel widths 1, 3 and 7 (ResNet-50)");
        L2I_CHECK_LAUNCH();
    }
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm2d, training mode
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// grid (C, chunks): block sums one slice of channel c over all samples
__global__ __launch_bounds__(256) void bn_stats_kernel(double* __restrict__ sum, double* __restrict__ sumsq, const float* __restrict__ x,
                                                       int B, int C, long long HW) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)gridDim.y * 256) {
        const long long b = i / HW, pix = i - b * HW;
        const double v = x[((size_t)b * C + c) * HW + pix];
        s += v; q += v * v;
    }
    __shared__ double sh[2][4];
    s = wave_sum_d(s); q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(sum + c, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        unsafeAtomicAdd(sumsq + c, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

extern "C" int l2i_bn_stats_f32(double* sum, double* sumsq, const float* x, int B, int C, int64_t HW, void* stream) {
    if (!sum || !sumsq || !x || B <= 0 || C <= 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "bn_stats: bad argument");
    long long chunks = ((long long)B * HW + 256 * 16 - 1) / (256 * 16);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)C, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, sum, sumsq, x, B, C, (long long)HW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ residual, int relu, int C,
                                                       long long HW, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        float v = x[i] * scale[c] + shift[c];
        if (residual) v += residual[i];
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

extern "C" int l2i_bn_apply_f32(float* y, const float* x, const float* scale, const float* shift, const float* residual, int relu, int B, int C,
                                int64_t HW, void* stream) {
    if (!y || !x || !scale || !shift || B <= 0 || C <= 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "bn_apply: bad argument");
    const long long n = (long long)B * C * HW;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(l2i_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, scale, shift, residual, relu, C, (long long)HW, n);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(double* __restrict__ sum_dy, double* __restrict__ sum_dyxh, const float* __restrict__ gy,
                                                            const float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int B, int C, long long HW) {
    const int c = blockIdx.x;
    const float mu = mean[c], is = invstd[c];
    double s = 0.0, q = 0.0;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)gridDim.y * 256) {
        const long long b = i / HW, pix = i - b * HW;
        const size_t idx = ((size_t)b * C + c) * HW + pix;
        float dy = gy[idx];
        if (out && !(out[idx] > 0.f)) dy = 0.f;
        s += dy;
        q += (double)dy * (double)((x[idx] - mu) * is);
    }
    __shared__ double sh[2][4];
    s = wave_sum_d(s); q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(sum_dy + c, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        unsafeAtomicAdd(sum_dyxh + c, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

extern "C" int l2i_bn_bwd_reduce_f32(double* sum_dy, double* sum_dyxh, const float* gy, const float* out_mask, const float* x, const float* mean,
                                     const float* invstd, int B, int C, int64_t HW, void* stream) {
    if (!sum_dy || !sum_dyxh || !gy || !x || !mean || !invstd || B <= 0 || C <= 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "bn_bwd_reduce: bad argument");
    long long chunks = ((long long)B * HW + 256 * 16 - 1) / (256 * 16);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)C, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, sum_dy, sum_dyxh, gy, out_mask, x, mean,
                       invstd, B, C, (long long)HW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dx, float* __restrict__ dy_masked, const float* __restrict__ gy,
                                                           const float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ m_dy, const float* __restrict__ m_dyxh, int C, long long HW, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        float dy = gy[i];
        if (out && !(out[i] > 0.f)) dy = 0.f;
        if (dy_masked) dy_masked[i] = dy;
        const float xh = (x[i] - mean[c]) * invstd[c];
        dx[i] = gamma[c] * invstd[c] * (dy - m_dy[c] - xh * m_dyxh[c]);
    }
}

extern "C" int l2i_bn_bwd_apply_f32(float* dx, float* dy_masked, const float* gy, const float* out_mask, const float* x, const float* mean,
                                    const float* invstd, const float* gamma, const float* mean_dy, const float* mean_dyxh, int B, int C, int64_t HW,
                                    void* stream) {
    if (!dx || !gy || !x || !mean || !invstd || !gamma || !mean_dy || !mean_dyxh || B <= 0 || C <= 0 || HW <= 0)
        return l2i_set_error(L2I_E_ARG, "bn_bwd_apply: bad argument");
    const long long n = (long long)B * C * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(l2i_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dx, dy_masked, gy, out_mask, x, mean, invstd, gamma,
                       mean_dy, mean_dyxh, C, (long long)HW, n);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
