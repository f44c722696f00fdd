// l2i_stream.hip — the HBM-bound kernels of the walk-training path (gfx950): fused_bias_act, upfirdn2d (+fused
// StyledConv / ToRGB epilogues), ToRGB, the fused StyledConv backward (leaky-ReLU' + style-gradient reductions),
// row reductions, max-pool and the content-loss difference.  All are one-pass streaming kernels: 16-byte
// per-lane accesses where the layout allows, wave64 shuffles + one atomic per wave for the reductions, grids capped
// at 8 blocks/CU with grid-stride loops.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include "l2i.h"
#include "l2i_internal.h"

// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
int l2i_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}
extern "C" const char* l2i_last_error(void) { return g_err; }
extern "C" int l2i_abi_version(void) { return L2I_ABI_VERSION; }
extern "C" int l2i_sizeof_conv_params(void) { return (int)sizeof(l2i_conv_params); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// fused_bias_act: reference op/fused_bias_act_kernel.cu:18-49 (all act*10+grad cases)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fba_one(float x, float r, int code, float alpha, float scale) {
    float y;
    switch (code) {
        default:
        case 10: case 11: y = x; break;
        case 12: case 32: y = 0.f; break;
        case 30: y = (x > 0.f) ? x : x * alpha; break;
        case 31: y = (r > 0.f) ? x : x * alpha; break;
    }
    return y * scale;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void fba_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ b,
                                                  const float* __restrict__ ref, long long n, long long step_b, long long size_b,
                                                  int code, float alpha, float scale) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (VEC4) {
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ref) r = reinterpret_cast<const float4*>(ref)[i];
            float bv = 0.f;
            if (b) bv = b[((i << 2) / step_b) % size_b];      // step_b % 4 == 0: one channel per float4
            float4 o;
            o.x = fba_one(v.x + bv, r.x, code, alpha, scale);
            o.y = fba_one(v.y + bv, r.y, code, alpha, scale);
            o.z = fba_one(v.z + bv, r.z, code, alpha, scale);
            o.w = fba_one(v.w + bv, r.w, code, alpha, scale);
            reinterpret_cast<float4*>(y)[i] = o;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float v = x[i];
            if (b) v += b[(i / step_b) % size_b];
            y[i] = fba_one(v, ref ? ref[i] : 0.f, code, alpha, scale);
        }
    }
}

extern "C" int l2i_fused_bias_act_f32(float* y, const float* x, const float* b, const float* ref, int64_t n, int64_t step_b,
                                      int64_t size_b, int act, int grad, float alpha, float scale, void* stream) {
    if (n == 0) return L2I_OK;
    if (!y || !x || n < 0) return l2i_set_error(L2I_E_ARG, "fused_bias_act: null tensor");
    if (b && (step_b <= 0 || size_b <= 0)) return l2i_set_error(L2I_E_ARG, "fused_bias_act: bias needs step_b,size_b > 0");
    const int code = act * 10 + grad;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (n % 4 == 0) && (!b || step_b % 4 == 0) && (((uintptr_t)y | (uintptr_t)x | (uintptr_t)ref) % 16 == 0);
    if (vec) hipLaunchKernelGGL(fba_kernel<true>, dim3(l2i_grid_for(n / 4, 256)), dim3(256), 0, st, y, x, b, ref, (long long)n, (long long)step_b, (long long)size_b, code, alpha, scale);
    else hipLaunchKernelGGL(fba_kernel<false>, dim3(l2i_grid_for(n, 256)), dim3(256), 0, st, y, x, b, ref, (long long)n, (long long)step_b, (long long)size_b, code, alpha, scale);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// upfirdn2d: reference op/upfirdn2d_kernel.cu:52-137 on [major, H, W] maps, + optional fused epilogue
//   out[o] = sum_k u[o*down + k - pad0] * K[kh-1-k],  u[i*up] = x[i]
// ---------------------------------------------------------------------------------------------------------------
struct UfdParams {
    float* y; const float* x; const float* k;
    long long major; int in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
    int channels; const float* noise; float noise_w; const float* bias; const float* addend; int act; float slope, gain;
    const float* mask; float mask_pos, mask_neg;       // [r5] l2i_upfirdn2d_masked_f32: y *= (mask > 0 ? mask_pos : mask_neg) after the activation (a gradient through a (leaky) ReLU)
};

// one thread = one output pixel; a block covers a 16x64 output tile of one map so that the <=19x67 input footprint is
// served from L1/L2 (the FIR taps sit in LDS).
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UfdParams p) {
    __shared__ float sk[64];
    const int ntap = p.kh * p.kw;
    for (int t = threadIdx.x; t < ntap; t += 256) {
        const int ky = t / p.kw, kx = t - ky * p.kw;
        sk[t] = p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];      // flipped: true convolution (.cu:79)
    }
    __syncthreads();
    const int tiles_x = (p.out_w + 63) >> 6, tiles_y = (p.out_h + 15) >> 4;
    const long long ntiles = p.major * tiles_x * tiles_y;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x);
        const int ty = (int)((t / tiles_x) % tiles_y);
        const long long mj = t / ((long long)tiles_x * tiles_y);
        const int ox = tx * 64 + (threadIdx.x & 63);
        const float* xin = p.x + mj * (long long)p.in_h * p.in_w;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int oy = ty * 16 + (threadIdx.x >> 6) * 4 + rr;
            if (ox >= p.out_w || oy >= p.out_h) continue;
            const int my = oy * p.down_y - p.pad_y0, mx = ox * p.down_x - p.pad_x0;
            float v = 0.f;
            for (int ky = 0; ky < p.kh; ++ky) {
                const int uy = my + ky;
                if (uy < 0) continue;
                const int iy = uy / p.up_y;
                if (iy * p.up_y != uy || iy >= p.in_h) continue;
                for (int kx = 0; kx < p.kw; ++kx) {
                    const int ux = mx + kx;
                    if (ux < 0) continue;
                    const int ix = ux / p.up_x;
                    if (ix * p.up_x != ux || ix >= p.in_w) continue;
                    v += xin[(long long)iy * p.in_w + ix] * sk[ky * p.kw + kx];
                }
            }
            const long long oidx = (mj * p.out_h + oy) * p.out_w + ox;
            if (p.noise) v += p.noise[((mj / p.channels) * p.out_h + oy) * p.out_w + ox] * p.noise_w;
            if (p.bias) v += p.bias[mj % p.channels];
            if (p.addend) v += p.addend[oidx];
            if (p.act == L2I_ACT_LRELU) v = (v > 0.f ? v : v * p.slope) * p.gain;
            else if (p.act == L2I_ACT_RELU) v = v > 0.f ? v : 0.f;
            if (p.mask) v *= p.mask[oidx] > 0.f ? p.mask_pos : p.mask_neg;
            p.y[oidx] = v;
        }
    }
}

// Fast path for the hot case up = down = 1 with a 4x4 FIR (every Blur of the generator and discriminator, forward and
// backward): one block = 32x64 outputs of one map; the 35x67 input footprint is staged once in LDS (coalesced rows),
// each thread produces a 2x4 patch from a 5x7 register window (35 LDS dwords per 8 outputs), 16-byte stores.
__global__ __launch_bounds__(256) void upfirdn2d_k4_kernel(const UfdParams p) {
    constexpr int TH = 32, TW = 64, IH = TH + 3, PITCH = 72, NQ = PITCH / 4;   // 18 float4 per staged row
    __shared__ __attribute__((aligned(16))) float tile[IH * PITCH];
    float kf[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.k[15 - t];                 // flipped taps (true convolution), wave-uniform
    const int tiles_x = (p.out_w + TW - 1) / TW, tiles_y = (p.out_h + TH - 1) / TH;
    const long long ntiles = p.major * tiles_x * tiles_y;
    const int tx4 = (threadIdx.x & 15) * 4, ty2 = (threadIdx.x >> 4) * 2;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x);
        const int ty = (int)((t / tiles_x) % tiles_y);
        const long long mj = t / ((long long)tiles_x * tiles_y);
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 - p.pad_y0, ix0 = ox0 - p.pad_x0;
        const float* xin = p.x + mj * (long long)p.in_h * p.in_w;
        __syncthreads();
        {   // stage the footprint with 16-byte loads (4-byte alignment suffices on gfx950), all of a thread's loads
            // issued back-to-back; edge vectors fall back to guarded scalar loads
            constexpr int NS = (IH * NQ + 255) / 256;
            float4 v[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int e = threadIdx.x + u * 256;
                const int r = e / NQ, q = e - r * NQ;
                const int gy = iy0 + r, gx = ix0 + 4 * q;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < IH * NQ && gy >= 0 && gy < p.in_h) {
                    const float* row = xin + (long long)gy * p.in_w;
                    if (gx >= 0 && gx + 3 < p.in_w) {
                        v[u] = *reinterpret_cast<const float4*>(row + gx);
                    } else {
                        if (gx >= 0 && gx < p.in_w) v[u].x = row[gx];
                        if (gx + 1 >= 0 && gx + 1 < p.in_w) v[u].y = row[gx + 1];
                        if (gx + 2 >= 0 && gx + 2 < p.in_w) v[u].z = row[gx + 2];
                        if (gx + 3 >= 0 && gx + 3 < p.in_w) v[u].w = row[gx + 3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int e = threadIdx.x + u * 256;
                if (e < IH * NQ) *reinterpret_cast<float4*>(&tile[e * 4]) = v[u];     // e*4 == r*PITCH + 4q
            }
        }
        __syncthreads();
        float win[5][7];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float4 a = *reinterpret_cast<const float4*>(&tile[(ty2 + r) * PITCH + tx4]);
            const float4 b = *reinterpret_cast<const float4*>(&tile[(ty2 + r) * PITCH + tx4 + 4]);
            win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w; win[r][4] = b.x; win[r][5] = b.y; win[r][6] = b.z;
        }
        const int c = (int)(mj % p.channels);
        const long long bsel = mj / p.channels;
        const float bia = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = oy0 + ty2 + r;
            if (oy >= p.out_h) continue;
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) v += win[r + ky][q + kx] * kf[ky * 4 + kx];
                o[q] = v + bia;
            }
            const int ox = ox0 + tx4;
            const long long obase = (mj * p.out_h + oy) * p.out_w + ox;
            const long long nbase = (bsel * p.out_h + oy) * p.out_w + ox;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ox + q < p.out_w) {
                    if (p.noise) o[q] += p.noise[nbase + q] * p.noise_w;
                    if (p.addend) o[q] += p.addend[obase + q];
                    if (p.act == L2I_ACT_LRELU) o[q] = (o[q] > 0.f ? o[q] : o[q] * p.slope) * p.gain;
                    else if (p.act == L2I_ACT_RELU) o[q] = o[q] > 0.f ? o[q] : 0.f;
                    if (p.mask) o[q] *= p.mask[obase + q] > 0.f ? p.mask_pos : p.mask_neg;
                }
            }
            if (ox + 3 < p.out_w && ((obase & 3) == 0) && (((uintptr_t)p.y & 15) == 0)) {
                *reinterpret_cast<float4*>(p.y + obase) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ox + q < p.out_w) p.y[obase + q] = o[q];
            }
        }
    }
}

// 4x4 FIR with down = 2, pad0 = 1 on wide maps: the Blur in front of the discriminator's stride-2 1x1 skip conv (networks.py:530-536, :586-590),
// evaluated only at the pixels that conv samples.  Register-streaming like upfirdn2d_k4_stream_kernel: a lane owns two output columns
// (input columns 4l - 1 .. 4l + 4: one aligned 16-byte load + one value from each neighbour lane) and walks RS output rows, two new input rows
// per output row.  Taps in (ky, kx) order: the same sums as the generic kernel.
__global__ __launch_bounds__(256) void upfirdn2d_k4_down2_stream_kernel(const UfdParams p, int bands, int strips) {
    constexpr int RS = 8;
    float kf[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.k[15 - t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long bid = blockIdx.x;
    const int strip = (int)(bid % strips); bid /= strips;
    const int band = (int)(bid % bands);
    const long long mj = bid / bands;
    const int ox = strip * 128 + lane * 2;                             // output columns ox, ox + 1; input vector at column 2 ox
    const int oy0 = (band * 4 + wave) * RS;
    if (oy0 >= p.out_h) return;
    const float* xin = p.x + mj * (long long)p.in_h * p.in_w;
    float win[4][6];
    auto load_row = [&](int iy, float (&w)[6]) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool rok = iy >= 0 && iy < p.in_h;
        const float* row = xin + (long long)iy * p.in_w;
        const int ix = 2 * ox;
        if (rok && ix < p.in_w) v = *reinterpret_cast<const float4*>(row + ix);
        float l1 = __shfl_up(v.w, 1), r1 = __shfl_down(v.x, 1);
        if (lane == 0) l1 = (rok && ix - 1 >= 0 && ix - 1 < p.in_w) ? row[ix - 1] : 0.f;
        if (lane == 63) r1 = (rok && ix + 4 < p.in_w) ? row[ix + 4] : 0.f;
        w[0] = l1; w[1] = v.x; w[2] = v.y; w[3] = v.z; w[4] = v.w; w[5] = r1;
    };
    const int iy0 = 2 * oy0 - 1;                                       // pad_y0 = 1
    load_row(iy0, win[0]);
    load_row(iy0 + 1, win[1]);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        // window rows of output row oy0 + r: input rows iy0 + 2r .. iy0 + 2r + 3 in slots (2r + ky) & 3
        load_row(iy0 + 2 * r + 2, win[(2 * r + 2) & 3]);
        load_row(iy0 + 2 * r + 3, win[(2 * r + 3) & 3]);
        const int oy = oy0 + r;
        if (oy >= p.out_h || ox >= p.out_w) continue;                  // out_w % 2 == 0: both columns are inside together
        float o[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) v += win[(2 * r + ky) & 3][2 * q + kx] * kf[ky * 4 + kx];
            o[q] = v;
        }
        *reinterpret_cast<float2*>(p.y + (mj * p.out_h + oy) * p.out_w + ox) = make_float2(o[0], o[1]);
    }
}

// up = 2, down = 1, 4x4 FIR, pad0 = 2 (the ToRGB skip upsample, networks.py:35-43 / :353-356, on 3-channel images): a thread owns a 2 x 4
// patch of outputs (rows 2i, 2i + 1; columns 4j .. 4j + 3) = input rows i - 1 .. i + 1, columns 2j - 1 .. 2j + 2.  Every output sums its
// 2 x 2 live taps in the generic kernel's (ky, kx) order (bit-identical results); 16-byte stores, addend read as 16-byte vectors.
__global__ __launch_bounds__(256) void upfirdn2d_up2k4_kernel(const UfdParams p) {
    float kf[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.k[15 - t];                 // flipped taps (true convolution)
    const int W4 = p.out_w >> 2, H2 = p.out_h >> 1;
    const long long n = p.major * H2 * W4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int j = (int)(i % W4);
        const int r = (int)((i / W4) % H2);
        const long long mj = i / ((long long)W4 * H2);
        const float* xin = p.x + mj * (long long)p.in_h * p.in_w;
        float win[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int iy = r - 1 + a, ix = 2 * j - 1 + b;
                win[a][b] = (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) ? xin[(long long)iy * p.in_w + ix] : 0.f;
            }
        const float bia = p.bias ? p.bias[mj % p.channels] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {                                  // output row 2r + a: taps ky = a, a + 2 on input rows r - 1 + a, r + a
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                              // output column 4j + q: taps kx = (q & 1), (q & 1) + 2 on input columns 2j + (q >> 1) - 1 + (q & 1) ...
                const int c = q & 1, cb = (q >> 1) + c;                //   ... = window columns cb, cb + 1
                float v = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int iy = r - 1 + a + t, ix = 2 * j - 1 + cb + s2;
                        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v += win[a + t][cb + s2] * kf[(a + 2 * t) * 4 + c + 2 * s2];
                    }
                o[q] = v;
            }
            const int oy = 2 * r + a, ox = 4 * j;
            const long long obase = (mj * p.out_h + oy) * p.out_w + ox;
            if (p.noise) {
                const long long nbase = ((mj / p.channels) * p.out_h + oy) * p.out_w + ox;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] += p.noise[nbase + q] * p.noise_w;
            }
            if (p.bias) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] += bia;
            }
            if (p.addend) {
                const float4 ad = *reinterpret_cast<const float4*>(p.addend + obase);
                o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (p.act == L2I_ACT_LRELU) o[q] = (o[q] > 0.f ? o[q] : o[q] * p.slope) * p.gain;
                else if (p.act == L2I_ACT_RELU) o[q] = o[q] > 0.f ? o[q] : 0.f;
            }
            *reinterpret_cast<float4*>(p.y + obase) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// 4x4 FIR, up = down = 1, wide maps (the blurs at >= 256 columns are 90 % of the FIR time of a step): no LDS, no barriers.  A lane owns four
// output columns and walks RS output rows downwards with the last four input rows in registers; per input row it issues ONE aligned
// 16-byte load and takes the three halo columns from its neighbour lanes (wave shuffles; the two edge lanes of a wave read theirs from
// memory), so a wave reads 1 KiB of contiguous row per step and every input element is fetched once per 16-row band (+ 3 halo rows).
// Taps are summed in the same (ky, kx) order as upfirdn2d_k4_kernel: bit-identical results.  (Measured, not kept: prefetching the next
// LDS tile of the staged kernel into registers, 3.4 -> 3.0 TB/s.)
template <int PX>       // pad_x0: 1 or 2 halo columns on the left, 3 - PX on the right
__global__ __launch_bounds__(256) void upfirdn2d_k4_stream_kernel(const UfdParams p, int bands, int strips) {
    constexpr int RS = 16;                                           // output rows per wave
    float kf[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.k[15 - t];                 // flipped taps (true convolution), wave-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long bid = blockIdx.x;
    const int strip = (int)(bid % strips); bid /= strips;
    const int band = (int)(bid % bands);
    const long long mj = bid / bands;
    const int ox = strip * 256 + lane * 4;
    const int oy0 = (band * 4 + wave) * RS;
    if (oy0 >= p.out_h) return;
    const float* xin = p.x + mj * (long long)p.in_h * p.in_w;
    const int c = (int)(mj % p.channels);
    const long long bsel = mj / p.channels;
    const float bia = p.bias ? p.bias[c] : 0.f;
    float win[4][7];
    // one input row -> the lane's 7-column window (columns ox - PX .. ox - PX + 6)
    auto load_row = [&](int iy, float (&w)[7]) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool rok = iy >= 0 && iy < p.in_h;
        const float* row = xin + (long long)iy * p.in_w;
        if (rok && ox < p.in_w) v = *reinterpret_cast<const float4*>(row + ox);          // in_w % 4 == 0: a vector is inside or outside
        float l1 = __shfl_up(v.w, 1), l2 = __shfl_up(v.z, 1), r1 = __shfl_down(v.x, 1), r2 = __shfl_down(v.y, 1);
        if (lane == 0) {
            l1 = (rok && ox - 1 >= 0 && ox - 1 < p.in_w) ? row[ox - 1] : 0.f;
            l2 = (PX == 2 && rok && ox - 2 >= 0 && ox - 2 < p.in_w) ? row[ox - 2] : 0.f;
        }
        if (lane == 63) {
            r1 = (rok && ox + 4 < p.in_w) ? row[ox + 4] : 0.f;
            r2 = (PX == 1 && rok && ox + 5 < p.in_w) ? row[ox + 5] : 0.f;
        }
        if (PX == 1) { w[0] = l1; w[1] = v.x; w[2] = v.y; w[3] = v.z; w[4] = v.w; w[5] = r1; w[6] = r2; }
        else { w[0] = l2; w[1] = l1; w[2] = v.x; w[3] = v.y; w[4] = v.z; w[5] = v.w; w[6] = r1; }
    };
    const int iy0 = oy0 - p.pad_y0;
    load_row(iy0, win[0]);
    load_row(iy0 + 1, win[1]);
    load_row(iy0 + 2, win[2]);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        load_row(iy0 + r + 3, win[(r + 3) & 3]);
        const int oy = oy0 + r;
        if (oy >= p.out_h || ox >= p.out_w) continue;                  // out_w % 4 == 0: the four columns are inside together
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) v += win[(r + ky) & 3][q + kx] * kf[ky * 4 + kx];
            o[q] = v + bia;
        }
        const long long obase = (mj * p.out_h + oy) * p.out_w + ox;
        if (p.noise) {
            const float4 nz = *reinterpret_cast<const float4*>(p.noise + (bsel * p.out_h + oy) * p.out_w + ox);
            o[0] += nz.x * p.noise_w; o[1] += nz.y * p.noise_w; o[2] += nz.z * p.noise_w; o[3] += nz.w * p.noise_w;
        }
        if (p.addend) {
            const float4 ad = *reinterpret_cast<const float4*>(p.addend + obase);
            o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (p.act == L2I_ACT_LRELU) o[q] = (o[q] > 0.f ? o[q] : o[q] * p.slope) * p.gain;
            else if (p.act == L2I_ACT_RELU) o[q] = o[q] > 0.f ? o[q] : 0.f;
        }
        if (p.mask) {
            const float4 mk = *reinterpret_cast<const float4*>(p.mask + obase);
            o[0] *= mk.x > 0.f ? p.mask_pos : p.mask_neg; o[1] *= mk.y > 0.f ? p.mask_pos : p.mask_neg;
            o[2] *= mk.z > 0.f ? p.mask_pos : p.mask_neg; o[3] *= mk.w > 0.f ? p.mask_pos : p.mask_neg;
        }
        *reinterpret_cast<float4*>(p.y + obase) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

static int upfirdn2d_launch(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw,
                            int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                            int channels, const float* noise, float noise_w, const float* bias, const float* addend,
                            int act, float act_slope, float act_gain, const float* mask, float mask_pos, float mask_neg, void* stream);

extern "C" int l2i_upfirdn2d_f32(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw,
                                 int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                 int channels, const float* noise, float noise_w, const float* bias, const float* addend,
                                 int act, float act_slope, float act_gain, void* stream) {
    return upfirdn2d_launch(y, x, k, major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, channels, noise, noise_w, bias, addend,
                            act, act_slope, act_gain, nullptr, 1.f, 0.f, stream);
}

extern "C" int l2i_upfirdn2d_masked_f32(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw,
                                        int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                        int channels, const float* noise, float noise_w, const float* bias, const float* addend,
                                        int act, float act_slope, float act_gain, const float* mask, float mask_pos, float mask_neg, void* stream) {
    return upfirdn2d_launch(y, x, k, major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, channels, noise, noise_w, bias, addend,
                            act, act_slope, act_gain, mask, mask_pos, mask_neg, stream);
}

static int upfirdn2d_launch(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw,
                            int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                            int channels, const float* noise, float noise_w, const float* bias, const float* addend,
                            int act, float act_slope, float act_gain, const float* mask, float mask_pos, float mask_neg, void* stream) {
    if (!y || !x || !k) return l2i_set_error(L2I_E_ARG, "upfirdn2d: null tensor");
    if (major <= 0 || in_h <= 0 || in_w <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d: empty input");
    if (kh <= 0 || kw <= 0 || kh * kw > 64) return l2i_set_error(L2I_E_ARG, "upfirdn2d: FIR must have 1..64 taps");
    if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d: up/down must be positive");
    UfdParams p;
    p.y = y; p.x = x; p.k = k; p.major = major; p.in_h = in_h; p.in_w = in_w; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;       // op/upfirdn2d.py:102-103
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (p.out_h <= 0 || p.out_w <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d: empty output");
    p.channels = channels > 0 ? channels : 1;
    p.noise = noise; p.noise_w = noise_w; p.bias = bias; p.addend = addend; p.act = act; p.slope = act_slope; p.gain = act_gain;
    p.mask = mask; p.mask_pos = mask_pos; p.mask_neg = mask_neg;
    if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4 && (pad_x0 == 1 || pad_x0 == 2) && p.out_w >= 192 && (in_w % 4) == 0 &&
        (p.out_w % 4) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)noise | (uintptr_t)addend | (uintptr_t)mask) % 16) == 0) {
        const int strips = (p.out_w + 255) / 256, bands = (p.out_h + 63) / 64;
        const long long grid = major * bands * strips;
        if (grid > 0 && grid <= 0x7fffffffLL) {
            if (pad_x0 == 1) hipLaunchKernelGGL((upfirdn2d_k4_stream_kernel<1>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, bands, strips);
            else hipLaunchKernelGGL((upfirdn2d_k4_stream_kernel<2>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, bands, strips);
            L2I_CHECK_LAUNCH();
            return L2I_OK;
        }
    }
    if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4) {
        const long long nt = major * ((p.out_w + 63) / 64) * ((p.out_h + 31) / 32);
        hipLaunchKernelGGL(upfirdn2d_k4_kernel, dim3(l2i_grid_for(nt, 1, 256 * 16)), dim3(256), 0, (hipStream_t)stream, p);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    if (up_x == 1 && up_y == 1 && down_x == 2 && down_y == 2 && kh == 4 && kw == 4 && pad_x0 == 1 && pad_y0 == 1 && p.out_w >= 96 && (in_w % 4) == 0 &&
        (p.out_w % 2) == 0 && !noise && !bias && !addend && !mask && act == L2I_ACT_NONE && (((uintptr_t)x) % 16) == 0 && (((uintptr_t)y) % 8) == 0) {
        const int strips = (p.out_w + 127) / 128, bands = (p.out_h + 31) / 32;
        const long long grid = major * bands * strips;
        if (grid > 0 && grid <= 0x7fffffffLL) {
            hipLaunchKernelGGL(upfirdn2d_k4_down2_stream_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, bands, strips);
            L2I_CHECK_LAUNCH();
            return L2I_OK;
        }
    }
    if (up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4 && pad_x0 == 2 && pad_y0 == 2 && p.out_h == 2 * in_h && p.out_w == 2 * in_w &&
        (p.out_w % 4) == 0 && !mask && (((uintptr_t)y | (uintptr_t)addend) % 16) == 0) {
        const long long n = major * (p.out_h / 2) * (p.out_w / 4);
        hipLaunchKernelGGL(upfirdn2d_up2k4_kernel, dim3(l2i_grid_for(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, p);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    const long long ntiles = major * ((p.out_w + 63) / 64) * ((p.out_h + 15) / 16);
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3(l2i_grid_for(ntiles, 1, 256 * 16)), dim3(256), 0, (hipStream_t)stream, p);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ToRGB forward: rgb[b,o,p] = sum_c x[b,c,p]*wmod[b,o,c] + bias[o]         (networks.py:346-351)
// one thread = 4 consecutive pixels; the 3xC modulated weights of the sample sit in LDS; x is read exactly once.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void torgb_fwd_kernel(float* __restrict__ rgb, const float* __restrict__ x, const float* __restrict__ wmod,
                                                        const float* __restrict__ bias, int C, long long HW, int blocks_per_b) {
    extern __shared__ float sw[];       // [3][C]
    const int b = blockIdx.x / blocks_per_b;
    const int blk = blockIdx.x - b * blocks_per_b;
    for (int i = threadIdx.x; i < 3 * C; i += 256) sw[i] = wmod[(long long)b * 3 * C + i];
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
    const long long hw4 = HW >> 2;
    const float* xb = x + (long long)b * C * HW;
    float* ob = rgb + (long long)b * 3 * HW;
    // [r4] eight channel rows in flight per thread: the plain loop left one 16-byte load outstanding per wave between dependent FMA groups
    // (2.4 TB/s where the FIR kernels of the same maps stream at 5.2); channels in the order 0 .. C-1 as before (same sums, bit-identical)
    for (long long i = (long long)blk * 256 + threadIdx.x; i < hw4; i += (long long)blocks_per_b * 256) {
        float4 a0 = make_float4(b0, b0, b0, b0), a1 = make_float4(b1, b1, b1, b1), a2 = make_float4(b2, b2, b2, b2);
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(xb + (long long)(c + u) * HW)[i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float w0 = sw[c + u], w1 = sw[C + c + u], w2 = sw[2 * C + c + u];
                a0.x += v[u].x * w0; a0.y += v[u].y * w0; a0.z += v[u].z * w0; a0.w += v[u].w * w0;
                a1.x += v[u].x * w1; a1.y += v[u].y * w1; a1.z += v[u].z * w1; a1.w += v[u].w * w1;
                a2.x += v[u].x * w2; a2.y += v[u].y * w2; a2.z += v[u].z * w2; a2.w += v[u].w * w2;
            }
        }
        for (; c < C; ++c) {
            const float4 v = reinterpret_cast<const float4*>(xb + (long long)c * HW)[i];
            const float w0 = sw[c], w1 = sw[C + c], w2 = sw[2 * C + c];
            a0.x += v.x * w0; a0.y += v.y * w0; a0.z += v.z * w0; a0.w += v.w * w0;
            a1.x += v.x * w1; a1.y += v.y * w1; a1.z += v.z * w1; a1.w += v.w * w1;
            a2.x += v.x * w2; a2.y += v.y * w2; a2.z += v.z * w2; a2.w += v.w * w2;
        }
        reinterpret_cast<float4*>(ob)[i] = a0;
        reinterpret_cast<float4*>(ob + HW)[i] = a1;
        reinterpret_cast<float4*>(ob + 2 * HW)[i] = a2;
    }
}

extern "C" int l2i_torgb_fwd_f32(float* rgb, const float* x, const float* wmod, const float* bias, int B, int C, int64_t HW, void* stream) {
    if (!rgb || !x || !wmod) return l2i_set_error(L2I_E_ARG, "torgb_fwd: null tensor");
    if (B <= 0 || C <= 0 || HW <= 0 || (HW % 4) != 0) return l2i_set_error(L2I_E_ARG, "torgb_fwd: HW must be a positive multiple of 4");
    int bpb = (int)((HW / 4 + 255) / 256);
    const int cap = (256 * 8 + B - 1) / B;
    if (bpb > cap) bpb = cap;
    if (bpb < 1) bpb = 1;
    hipLaunchKernelGGL(torgb_fwd_kernel, dim3(B * bpb), dim3(256), 3 * C * sizeof(float), (hipStream_t)stream, rgb, x, wmod, bias, C, (long long)HW, bpb);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused StyledConv backward elementwise pass (see l2i.h).  grid = (B*C rows) x (chunks of the HW axis).
// ---------------------------------------------------------------------------------------------------------------
struct ActBwdParams {
    float* dz; const float* gin; const float* gin_scale; const float* grgb; const float* wmod_rgb; const float* y;
    const float* bias; const float* noise; float noise_w, slope, gain; float* red_dz_z; float* red_x_grgb; float* red_gin_y;
    int B, C; long long HW; int chunks;
};

__global__ __launch_bounds__(256) void sg2_act_bwd_kernel(const ActBwdParams p) {
    const int row = blockIdx.x / p.chunks;             // b*C + c
    const int chunk = blockIdx.x - row * p.chunks;
    const int b = row / p.C, c = row - b * p.C;
    const long long hw4 = p.HW >> 2;
    const float gs = p.gin_scale ? p.gin_scale[row] : 1.f;
    const float bia = p.bias ? p.bias[c] : 0.f;
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
    if (p.grgb) {
        const float* wm = p.wmod_rgb + (long long)b * 3 * p.C;
        w0 = wm[c]; w1 = wm[p.C + c]; w2 = wm[2 * p.C + c];
    }
    const float gpos = p.gain, gneg = p.gain * p.slope;
    const float ipos = 1.f / gpos, ineg = 1.f / gneg;
    const float4* y4 = reinterpret_cast<const float4*>(p.y + (long long)row * p.HW);
    const float4* g4 = p.gin ? reinterpret_cast<const float4*>(p.gin + (long long)row * p.HW) : nullptr;
    const float4* n4 = p.noise ? reinterpret_cast<const float4*>(p.noise + (long long)b * p.HW) : nullptr;
    const float4* r0 = p.grgb ? reinterpret_cast<const float4*>(p.grgb + (long long)b * 3 * p.HW) : nullptr;
    const float4* r1 = r0 ? r0 + hw4 : nullptr;
    const float4* r2 = r0 ? r1 + hw4 : nullptr;
    float4* d4 = reinterpret_cast<float4*>(p.dz + (long long)row * p.HW);
    float s_dz = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s_gy = 0.f;
    for (long long i = (long long)chunk * 256 + threadIdx.x; i < hw4; i += (long long)p.chunks * 256) {
        const float4 yv = y4[i];
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g4) {
            const float4 t = g4[i];
            g.x = t.x * gs; g.y = t.y * gs; g.z = t.z * gs; g.w = t.w * gs;
            s_gy += t.x * yv.x + t.y * yv.y + t.z * yv.z + t.w * yv.w;         // [r5] red_gin_y: both maps are in registers anyway
        }
        if (r0) {
            const float4 a = r0[i], bq = r1[i], cq = r2[i];
            g.x += a.x * w0 + bq.x * w1 + cq.x * w2; g.y += a.y * w0 + bq.y * w1 + cq.y * w2;
            g.z += a.z * w0 + bq.z * w1 + cq.z * w2; g.w += a.w * w0 + bq.w * w1 + cq.w * w2;
            s0 += yv.x * a.x + yv.y * a.y + yv.z * a.z + yv.w * a.w;
            s1 += yv.x * bq.x + yv.y * bq.y + yv.z * bq.z + yv.w * bq.w;
            s2 += yv.x * cq.x + yv.y * cq.y + yv.z * cq.z + yv.w * cq.w;
        }
        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n4) { nz = n4[i]; nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w; }
        float4 d;
        d.x = g.x * (yv.x > 0.f ? gpos : gneg); d.y = g.y * (yv.y > 0.f ? gpos : gneg);
        d.z = g.z * (yv.z > 0.f ? gpos : gneg); d.w = g.w * (yv.w > 0.f ? gpos : gneg);
        s_dz += d.x * (yv.x * (yv.x > 0.f ? ipos : ineg) - bia - nz.x) + d.y * (yv.y * (yv.y > 0.f ? ipos : ineg) - bia - nz.y) +
                d.z * (yv.z * (yv.z > 0.f ? ipos : ineg) - bia - nz.z) + d.w * (yv.w * (yv.w > 0.f ? ipos : ineg) - bia - nz.w);
        d4[i] = d;
    }
    __shared__ float red[4][5];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    s_dz = wave_sum(s_dz);
    if (r0) { s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); }
    if (g4 && p.red_gin_y) s_gy = wave_sum(s_gy);
    if (lane == 0) { red[wv][0] = s_dz; red[wv][1] = s0; red[wv][2] = s1; red[wv][3] = s2; red[wv][4] = s_gy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (p.red_dz_z) atomicAdd(p.red_dz_z + row, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
        if (g4 && p.red_gin_y) atomicAdd(p.red_gin_y + row, red[0][4] + red[1][4] + red[2][4] + red[3][4]);
        if (r0 && p.red_x_grgb) {
            atomicAdd(p.red_x_grgb + (long long)row * 3 + 0, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
            atomicAdd(p.red_x_grgb + (long long)row * 3 + 1, red[0][2] + red[1][2] + red[2][2] + red[3][2]);
            atomicAdd(p.red_x_grgb + (long long)row * 3 + 2, red[0][3] + red[1][3] + red[2][3] + red[3][3]);
        }
    }
}

extern "C" int l2i_sg2_act_bwd_f32(float* dz, const float* gin, const float* gin_scale, const float* grgb, const float* wmod_rgb,
                                   const float* y, const float* bias, const float* noise, float noise_w, float slope, float gain,
                                   float* red_dz_z, float* red_x_grgb, float* red_gin_y, int B, int C, int64_t HW, void* stream) {
    if (!dz || !y) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd: null tensor");
    if (!gin && !grgb) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd: need gin and/or grgb");
    if (grgb && !wmod_rgb) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd: grgb needs wmod_rgb");
    if (B <= 0 || C <= 0 || HW <= 0 || (HW % 4) != 0) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd: HW must be a positive multiple of 4");
    if (gain == 0.f || slope == 0.f) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd: gain and slope must be non-zero");
    ActBwdParams p;
    p.dz = dz; p.gin = gin; p.gin_scale = gin_scale; p.grgb = grgb; p.wmod_rgb = wmod_rgb; p.y = y; p.bias = bias; p.noise = noise;
    p.noise_w = noise_w; p.slope = slope; p.gain = gain; p.red_dz_z = red_dz_z; p.red_x_grgb = red_x_grgb; p.red_gin_y = red_gin_y; p.B = B; p.C = C; p.HW = HW;
    const long long rows = (long long)B * C;
    long long chunks = (HW / 4 + 1023) / 1024;          // >= 4 float4 per thread
    const long long cap = (256 * 16 + rows - 1) / rows;
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    p.chunks = (int)chunks;
    hipLaunchKernelGGL(sg2_act_bwd_kernel, dim3((unsigned)(rows * chunks)), dim3(256), 0, (hipStream_t)stream, p);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// row dot / sum reduction
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dot_reduce_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ bq,
                                                         long long cols, int chunks) {
    const long long row = blockIdx.x / chunks;
    const int chunk = blockIdx.x - (int)(row * chunks);
    const float* ar = a + row * cols;
    const float* br = bq ? bq + row * cols : nullptr;
    float s = 0.f;
    if ((cols & 3) == 0 && ((uintptr_t)ar % 16 == 0) && (!br || (uintptr_t)br % 16 == 0)) {
        const long long c4 = cols >> 2;
        for (long long i = (long long)chunk * 256 + threadIdx.x; i < c4; i += (long long)chunks * 256) {
            const float4 v = reinterpret_cast<const float4*>(ar)[i];
            if (br) { const float4 w = reinterpret_cast<const float4*>(br)[i]; s += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w; }
            else s += v.x + v.y + v.z + v.w;
        }
    } else {
        for (long long i = (long long)chunk * 256 + threadIdx.x; i < cols; i += (long long)chunks * 256)
            s += br ? ar[i] * br[i] : ar[i];
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + row, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int l2i_dot_reduce_f32(float* out, const float* a, const float* b, int64_t rows, int64_t cols, void* stream) {
    if (!out || !a) return l2i_set_error(L2I_E_ARG, "dot_reduce: null tensor");
    if (rows <= 0 || cols <= 0) return l2i_set_error(L2I_E_ARG, "dot_reduce: empty input");
    long long chunks = (cols / 4 + 1023) / 1024;
    const long long cap = (256 * 16 + rows - 1) / rows;
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    if (rows * chunks > 0x7fffffffLL) return l2i_set_error(L2I_E_ARG, "dot_reduce: too many rows");
    hipLaunchKernelGGL(dot_reduce_kernel, dim3((unsigned)(rows * chunks)), dim3(256), 0, (hipStream_t)stream, out, a, b, (long long)cols, (int)chunks);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// MaxPool2d forward / backward on [planes, H, W]
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(float* __restrict__ y, uint8_t* __restrict__ idx, const float* __restrict__ x,
                                                          long long planes, int H, int W, int k, int s, int pad, int OH, int OW) {
    const long long total = planes * OH * OW;
    const unsigned OHW = (unsigned)OH * (unsigned)OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pl = i / OHW;                          // one 64-bit division; the rest is 32-bit
        const unsigned rem = (unsigned)(i - pl * OHW);
        const int oy = (int)(rem / (unsigned)OW);
        const int ox = (int)(rem - (unsigned)oy * (unsigned)OW);
        const float* xp = x + pl * H * W;
        float best = -INFINITY;
        int bi = 0;
        bool found = false;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - pad + kx;
                if (ix < 0 || ix >= W) continue;
                const float v = xp[(long long)iy * W + ix];
                if (!found || v > best || (v != v)) { best = v; bi = ky * k + kx; found = true; }   // first max; NaN propagates like ATen
            }
        }
        y[i] = best;
        idx[i] = (uint8_t)bi;
    }
}

// KT/ST/PT > 0: compile-time geometry (3,2,1 = the ResNet stem pool: the window arithmetic becomes shifts); 0: run-time k, s, pad
template <int KT, int ST, int PT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(float* __restrict__ gx, const float* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                          long long planes, int H, int W, int k_, int s_, int pad_, int OH, int OW) {
    const int k = KT ? KT : k_, s = KT ? ST : s_, pad = KT ? PT : pad_;
    const long long total = planes * H * W;
    const unsigned HW = (unsigned)H * (unsigned)W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pl = i / HW;
        const unsigned rem = (unsigned)(i - pl * HW);
        const int iy = (int)(rem / (unsigned)W);
        const int ix = (int)(rem - (unsigned)iy * (unsigned)W);
        float g = 0.f;
        // windows that contain (iy, ix): oy in [ceil((iy+pad-k+1)/s), floor((iy+pad)/s)]
        int oy_lo = iy + pad - k + 1; oy_lo = oy_lo <= 0 ? 0 : (oy_lo + s - 1) / s;
        int oy_hi = (iy + pad) / s; if (oy_hi > OH - 1) oy_hi = OH - 1;
        int ox_lo = ix + pad - k + 1; ox_lo = ox_lo <= 0 ? 0 : (ox_lo + s - 1) / s;
        int ox_hi = (ix + pad) / s; if (ox_hi > OW - 1) ox_hi = OW - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const long long o = (pl * OH + oy) * OW + ox;
                const int ky = iy - (oy * s - pad), kx = ix - (ox * s - pad);
                if ((int)idx[o] == ky * k + kx) g += gy[o];
            }
        gx[i] = g;
    }
}

// The two geometries on the walk-training path, four outputs of one row per thread from 16-byte loads (the generic kernel reads every
// window element as a stride-2 scalar: 3.3 TB/s).  Same arg-max rule as above: taps in (ky, kx) order, first maximum, NaN propagates.
// K = 2: 2x2 / stride 2 / pad 0 (VGG-19 pool1), W == 2 OW.   K = 3: 3x3 / stride 2 / pad 1 (ResNet-50 stem pool), W == 2 OW.
template <int K>
__global__ __launch_bounds__(256) void maxpool_fwd_vec_kernel(float* __restrict__ y, uint8_t* __restrict__ idx, const float* __restrict__ x,
                                                              long long n4, int H, int W, int OH, int OW) {
    const int W4 = OW >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long row = i / W4;                          // (plane, output row)
        const int ox0 = (int)(i - row * W4) * 4;
        const long long pl = row / OH;
        const int oy = (int)(row - pl * OH);
        const float* xp = x + pl * H * (long long)W;
        float best[4];
        int bi[4];
        bool found[4] = {false, false, false, false};
#pragma unroll
        for (int q = 0; q < 4; ++q) { best[q] = -INFINITY; bi[q] = 0; }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = 2 * oy - (K == 3 ? 1 : 0) + ky;
            if (iy < 0 || iy >= H) continue;
            const float* rp = xp + (long long)iy * W + 2 * ox0;
            const float4 a = *reinterpret_cast<const float4*>(rp);
            const float4 b = *reinterpret_cast<const float4*>(rp + 4);
            float v[9];                                        // columns 2 ox0 - 1 .. 2 ox0 + 7
            v[0] = (K == 3 && ox0 > 0) ? rp[-1] : 0.f;
            v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w; v[5] = b.x; v[6] = b.y; v[7] = b.z; v[8] = b.w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    if (K == 3 && kx == 0 && ox0 + q == 0) continue;          // column -1: padding
                    const float t = v[2 * q + kx + (K == 3 ? 0 : 1)];
                    if (!found[q] || t > best[q] || (t != t)) { best[q] = t; bi[q] = ky * K + kx; found[q] = true; }
                }
            }
        }
        *reinterpret_cast<float4*>(y + row * OW + ox0) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<uint32_t*>(idx + row * OW + ox0) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}

extern "C" int l2i_maxpool2d_fwd_f32(float* y, uint8_t* idx, const float* x, int64_t planes, int H, int W, int k, int s, int pad, int OH, int OW, void* stream) {
    if (!y || !idx || !x) return l2i_set_error(L2I_E_ARG, "maxpool_fwd: null tensor");
    if (planes <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || s <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "maxpool_fwd: bad geometry");
    const bool vec_ok = s == 2 && W == 2 * OW && (OW % 4) == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0 && (((uintptr_t)idx) % 4) == 0;
    if (vec_ok && ((k == 2 && pad == 0 && H == 2 * OH) || (k == 3 && pad == 1 && (H + 1) / 2 == OH))) {
        const long long n4 = (long long)planes * OH * (OW / 4);
        if (k == 2) hipLaunchKernelGGL((maxpool_fwd_vec_kernel<2>), dim3(l2i_grid_for(n4, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, idx, x, n4, H, W, OH, OW);
        else hipLaunchKernelGGL((maxpool_fwd_vec_kernel<3>), dim3(l2i_grid_for(n4, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, idx, x, n4, H, W, OH, OW);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(l2i_grid_for(planes * OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, y, idx, x, (long long)planes, H, W, k, s, pad, OH, OW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// k = 2, s = 2, pad = 0 (VGG): every input pixel belongs to exactly one window -> one thread per window writes its 2x2
// inputs (gradient at the arg-max, zeros elsewhere) as two 8-byte stores.
__global__ __launch_bounds__(256) void maxpool_bwd_k2s2_kernel(float* __restrict__ gx, const float* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                               long long planes, int H, int W, int OH, int OW) {
    const long long total = planes * OH * OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        const int oy = (int)((i / OW) % OH);
        const long long pl = i / ((long long)OW * OH);
        const float g = gy[i];
        const int a = idx[i];
        float* dst = gx + (pl * H + 2 * oy) * (long long)W + 2 * ox;
        *reinterpret_cast<float2*>(dst) = make_float2(a == 0 ? g : 0.f, a == 1 ? g : 0.f);
        *reinterpret_cast<float2*>(dst + W) = make_float2(a == 2 ? g : 0.f, a == 3 ? g : 0.f);
    }
}

// k = 3, s = 2, pad = 1 with W == 2 OW (ResNet-50 stem pool): four consecutive input pixels of one row per thread.  Column 2m belongs to
// window m only (kx = 1), column 2m + 1 to windows m (kx = 2) and m + 1 (kx = 0); rows alike: the thread reads gy / idx of windows
// 2t .. 2t + 2 of one or two window rows (8-byte + 4-byte loads) instead of up to 16 scalar pairs.
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2p1_vec_kernel(float* __restrict__ gx, const float* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                                     long long n4, int H, int W, int OH, int OW) {
    const int W4 = W >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long row = i / W4;                          // (plane, input row)
        const int t = (int)(i - row * W4);                     // columns 4t .. 4t + 3; windows 2t, 2t + 1, 2t + 2
        const long long pl = row / H;
        const int iy = (int)(row - pl * H);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        // window rows: iy even -> iy / 2 (ky = 1); iy odd -> (iy - 1) / 2 (ky = 2) and (iy + 1) / 2 (ky = 0)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            int oy, ky;
            if ((iy & 1) == 0) { if (c) continue; oy = iy >> 1; ky = 1; }
            else { oy = (iy >> 1) + c; ky = c ? 0 : 2; }
            if (oy >= OH) continue;
            const long long o = (pl * OH + oy) * OW + 2 * t;
            const float2 ga = *reinterpret_cast<const float2*>(gy + o);
            const unsigned ia = *reinterpret_cast<const uint16_t*>(idx + o);
            const bool third = 2 * t + 2 < OW;
            const float gc = third ? gy[o + 2] : 0.f;
            const int ic = third ? (int)idx[o + 2] : -1;
            const int i0 = (int)(ia & 0xff), i1 = (int)(ia >> 8);
            // column 4t = 2 (2t): window 2t, kx = 1.   4t + 1: windows 2t (kx = 2), 2t + 1 (kx = 0).   4t + 2: window 2t + 1 (kx = 1).
            // 4t + 3: windows 2t + 1 (kx = 2), 2t + 2 (kx = 0)
            if (i0 == ky * 3 + 1) g[0] += ga.x;
            if (i0 == ky * 3 + 2) g[1] += ga.x;
            if (i1 == ky * 3 + 0) g[1] += ga.y;
            if (i1 == ky * 3 + 1) g[2] += ga.y;
            if (i1 == ky * 3 + 2) g[3] += ga.y;
            if (ic == ky * 3 + 0) g[3] += gc;
        }
        *reinterpret_cast<float4*>(gx + row * W + 4 * t) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

extern "C" int l2i_maxpool2d_bwd_f32(float* gx, const float* gy, const uint8_t* idx, int64_t planes, int H, int W, int k, int s, int pad, int OH, int OW, void* stream) {
    if (!gx || !gy || !idx) return l2i_set_error(L2I_E_ARG, "maxpool_bwd: null tensor");
    if (planes <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || s <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "maxpool_bwd: bad geometry");
    if (k == 2 && s == 2 && pad == 0 && H == 2 * OH && W == 2 * OW && (((uintptr_t)gx) % 8) == 0) {
        hipLaunchKernelGGL(maxpool_bwd_k2s2_kernel, dim3(l2i_grid_for(planes * OH * OW, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, gx, gy, idx, (long long)planes, H, W, OH, OW);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    if ((long long)H * W >= 0x7fffffffLL) return l2i_set_error(L2I_E_ARG, "maxpool_bwd: plane too large");
    if (k == 3 && s == 2 && pad == 1 && W == 2 * OW && (H + 1) / 2 == OH && (W % 4) == 0 && (((uintptr_t)gx) % 16) == 0 && (((uintptr_t)gy) % 8) == 0 &&
        (((uintptr_t)idx) % 2) == 0) {
        const long long n4 = (long long)planes * H * (W / 4);
        hipLaunchKernelGGL(maxpool_bwd_k3s2p1_vec_kernel, dim3(l2i_grid_for(n4, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, gx, gy, idx, n4, H, W, OH, OW);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    if (k == 3 && s == 2 && pad == 1)
        hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2, 1>), dim3(l2i_grid_for(planes * H * W, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, gx, gy, idx, (long long)planes, H, W, k, s, pad, OH, OW);
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<0, 0, 0>), dim3(l2i_grid_for(planes * H * W, 256)), dim3(256), 0, (hipStream_t)stream, gx, gy, idx, (long long)planes, H, W, k, s, pad, OH, OW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// 2x2/2 pool backward + content-loss gradient of the pool's input in one pass: a thread owns two horizontally adjacent windows
// = 4 consecutive pixels of two rows: 16-byte loads of a and b, 16-byte stores of gx.
__global__ __launch_bounds__(256) void maxpool2x2_bwd_add_diff_kernel(float* __restrict__ gx, const float* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                                      const float* __restrict__ a, const float* __restrict__ b, float coef,
                                                                      const float* __restrict__ coef_dev, long long planes, int OH, int OW) {
    if (coef_dev) coef *= coef_dev[0];
    const int OW2 = OW >> 1, W = 2 * OW;
    const long long total = planes * OH * OW2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox2 = (int)(i % OW2);
        const long long row = i / OW2;                         // plane * OH + oy
        const float2 g = *reinterpret_cast<const float2*>(gy + row * OW + 2 * ox2);
        const int i0 = idx[row * OW + 2 * ox2], i1 = idx[row * OW + 2 * ox2 + 1];
        const long long base = (2 * row) * (long long)W + 4 * ox2;       // rows 2*oy of plane: (plane*OH + oy) * 2 rows of width W
        const float4 a0 = *reinterpret_cast<const float4*>(a + base), a1 = *reinterpret_cast<const float4*>(a + base + W);
        const float4 b0 = *reinterpret_cast<const float4*>(b + base), b1 = *reinterpret_cast<const float4*>(b + base + W);
        float4 r0, r1;
        r0.x = coef * (b0.x - a0.x) + (i0 == 0 ? g.x : 0.f); r0.y = coef * (b0.y - a0.y) + (i0 == 1 ? g.x : 0.f);
        r0.z = coef * (b0.z - a0.z) + (i1 == 0 ? g.y : 0.f); r0.w = coef * (b0.w - a0.w) + (i1 == 1 ? g.y : 0.f);
        r1.x = coef * (b1.x - a1.x) + (i0 == 2 ? g.x : 0.f); r1.y = coef * (b1.y - a1.y) + (i0 == 3 ? g.x : 0.f);
        r1.z = coef * (b1.z - a1.z) + (i1 == 2 ? g.y : 0.f); r1.w = coef * (b1.w - a1.w) + (i1 == 3 ? g.y : 0.f);
        *reinterpret_cast<float4*>(gx + base) = r0;
        *reinterpret_cast<float4*>(gx + base + W) = r1;
    }
}

extern "C" int l2i_maxpool2x2_bwd_add_diff_f32(float* gx, const float* gy, const uint8_t* idx, const float* a, const float* b, float coef,
                                               const float* coef_dev, int64_t planes, int OH, int OW, void* stream) {
    if (!gx || !gy || !idx || !a || !b) return l2i_set_error(L2I_E_ARG, "maxpool2x2_bwd_add_diff: null tensor");
    if (planes <= 0 || OH <= 0 || OW <= 0 || (OW & 1)) return l2i_set_error(L2I_E_ARG, "maxpool2x2_bwd_add_diff: OW must be even and positive");
    if (((uintptr_t)gx | (uintptr_t)a | (uintptr_t)b) % 16 || ((uintptr_t)gy) % 8) return l2i_set_error(L2I_E_ARG, "maxpool2x2_bwd_add_diff: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(maxpool2x2_bwd_add_diff_kernel, dim3(l2i_grid_for(planes * OH * (OW / 2), 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                       gx, gy, idx, a, b, coef, coef_dev, (long long)planes, OH, OW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ContentLoss pieces and small utilities
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqdiff_kernel(float* __restrict__ sum_out, float* __restrict__ grad, const float* __restrict__ a,
                                                     const float* __restrict__ b, long long n, float coef,
                                                     const float* __restrict__ coef_dev) {
    float s = 0.f;
    if (coef_dev) coef *= coef_dev[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = b[i] - a[i];
        s += d * d;
        if (grad) grad[i] = coef * d;
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && sum_out) atomicAdd(sum_out, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int l2i_sqdiff_f32(float* sum_out, float* grad, const float* a, const float* b, int64_t n, float coef,
                              const float* coef_dev, void* stream) {
    if (!a || !b || n <= 0) return l2i_set_error(L2I_E_ARG, "sqdiff: null/empty tensor");
    hipLaunchKernelGGL(sqdiff_kernel, dim3(l2i_grid_for(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, sum_out, grad, a, b, (long long)n, coef, coef_dev);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b,
                                                    float alpha, float beta, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

extern "C" int l2i_axpby_f32(float* y, const float* a, const float* b, float alpha, float beta, int64_t n, void* stream) {
    if (!y || !a || n <= 0) return l2i_set_error(L2I_E_ARG, "axpby: null/empty tensor");
    hipLaunchKernelGGL(axpby_kernel, dim3(l2i_grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, y, a, b, alpha, beta, (long long)n);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ ref, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = ref[i] > 0.f ? g[i] : 0.f;
}

extern "C" int l2i_relu_mask_f32(float* y, const float* g, const float* ref, int64_t n, void* stream) {
    if (!y || !g || !ref || n <= 0) return l2i_set_error(L2I_E_ARG, "relu_mask: null/empty tensor");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(l2i_grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, y, g, ref, (long long)n);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
