// l2i_stream_h8.hip — the streaming (HBM-bound) companions of l2i_conv_h8.hip on bf16 tensors in the channel-blocked h8 layout
// [B][C/8][H][W][8] (gfx950): layout casts, upfirdn2d (the 4x4 blurs of the generator's up layers and of the discriminator, with the
// generator's noise + bias + leaky-ReLU epilogue), ToRGB, the fused activation backward of a styled conv with its two style reductions,
// pixel reductions, max-pools, the ContentLoss difference, zero insertion.  Same functions as their fp32 NCHW counterparts in
// l2i_stream.hip (reference call sites are cited there and in include/l2i.h); arithmetic in fp32 registers, 16-bit only in HBM.
// A thread works on one 16-byte pixel slot (the 8 channels of a pixel): consecutive lanes = consecutive pixels = consecutive 16 bytes, so
// every global access is a full-width coalesced one whatever the channel count, and a pixel's channel group is lane-local.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ float blo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void unpack8(const u32x4& u, float (&v)[8]) {
    v[0] = blo(u.x); v[1] = bhi(u.x); v[2] = blo(u.y); v[3] = bhi(u.y); v[4] = blo(u.z); v[5] = bhi(u.z); v[6] = blo(u.w); v[7] = bhi(u.w);
}
__device__ __forceinline__ unsigned pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) { return u32x4{pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7])}; }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
}

// ---- layout casts ---------------------------------------------------------------------------------------------------------------------
// y[b][g][p][e] = bf16(x[b][8 g + e][p]) (zero for channels >= C); lane = pixel: eight strided 4-byte reads (each a coalesced 256-byte row
// segment across the wave), one 16-byte store
__global__ __launch_bounds__(256) void cast_f32_to_h8_kernel(u32x4* __restrict__ y, const float* __restrict__ x, int C, int G8, long long HW, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long bg = i / HW, pix = i - bg * HW;
        const int g = (int)(bg % G8);
        const long long b = bg / G8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (8 * g + e < C) ? x[(b * C + 8 * g + e) * HW + pix] : 0.f;
        y[i] = pack8(v);
    }
}
__global__ __launch_bounds__(256) void cast_h8_to_f32_kernel(float* __restrict__ y, const u32x4* __restrict__ x, int C, int G8, long long HW, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long bg = i / HW, pix = i - bg * HW;
        const int g = (int)(bg % G8);
        const long long b = bg / G8;
        float v[8];
        unpack8(x[i], v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * g + e < C) y[(b * C + 8 * g + e) * HW + pix] = v[e];
    }
}
extern "C" int l2i_cast_f32_to_h8(void* y, const float* x, int B, int C, int Cpad, int64_t HW, void* stream) {
    if (!y || !x || B <= 0 || C <= 0 || Cpad < C || (Cpad % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "cast_f32_to_h8: bad arguments");
    const long long total = (long long)B * (Cpad / 8) * HW;
    hipLaunchKernelGGL(cast_f32_to_h8_kernel, dim3(l2i_grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, x, C, Cpad / 8, (long long)HW, total);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
extern "C" int l2i_cast_h8_to_f32(float* y, const void* x, int B, int C, int Cpad, int64_t HW, void* stream) {
    if (!y || !x || B <= 0 || C <= 0 || Cpad < C || (Cpad % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "cast_h8_to_f32: bad arguments");
    const long long total = (long long)B * (Cpad / 8) * HW;
    hipLaunchKernelGGL(cast_h8_to_f32_kernel, dim3(l2i_grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, y, (const u32x4*)x, C, Cpad / 8, (long long)HW, total);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- upfirdn2d ------------------------------------------------------------------------------------------------------------------------
// The reference op (upfirdn2d_kernel.cu:52-137 semantics: zero-insert by `up`, pad, correlate with the FLIPPED kernel, keep every `down`-th
// sample) on h8 planes, same taps for the 8 channels of a slot, with the fused epilogue of l2i_upfirdn2d_f32:
//   y = act( fir(x) + noise[b,oy,ox] * noise_w + bias[c] ) * act_gain.        KH, KW <= 4, up / down in {1, 2} (the path's geometries).
__global__ __launch_bounds__(256) void upfirdn2d_h8_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ x, const float* __restrict__ k, long long planes, int G8,
                                                           int in_h, int in_w, int out_h, int out_w, int kh, int kw, int up, int down, int pad_x0, int pad_y0,
                                                           const float* __restrict__ noise, float noise_w, const float* __restrict__ bias, int act, float slope, float gain,
                                                           const u32x4* __restrict__ mask, float mpos, float mneg, const u32x4* __restrict__ addend) {
    __shared__ float taps[16];
    if (threadIdx.x < kh * kw) taps[threadIdx.x] = k[threadIdx.x];
    __syncthreads();
    const long long OHW = (long long)out_h * out_w, total = planes * OHW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pl = i / OHW;
        const unsigned rem = (unsigned)(i - pl * OHW);
        const int oy = (int)(rem / (unsigned)out_w), ox = (int)(rem - (unsigned)oy * (unsigned)out_w);
        const u32x4* xp = x + pl * in_h * in_w;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // out[oy, ox] = sum_{ky, kx} k[kh-1-ky][kw-1-kx] * xup[oy*down + ky - pad_y0][ox*down + kx - pad_x0],  xup[u] = x[u / up] if u % up == 0
        for (int ky = 0; ky < kh; ++ky) {
            const int uy = oy * down + ky - pad_y0;
            if (uy < 0 || (up == 2 && (uy & 1))) continue;
            const int iy = up == 2 ? uy >> 1 : uy;
            if (iy >= in_h) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int ux = ox * down + kx - pad_x0;
                if (ux < 0 || (up == 2 && (ux & 1))) continue;
                const int ix = up == 2 ? ux >> 1 : ux;
                if (ix >= in_w) continue;
                const float t = taps[(kh - 1 - ky) * kw + (kw - 1 - kx)];
                float v[8];
                unpack8(xp[(long long)iy * in_w + ix], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += t * v[e];
            }
        }
        if (noise || bias || act != L2I_ACT_NONE || gain != 1.f) {
            const int g = (int)(pl % G8);
            const long long b = pl / G8;
            const float nz = noise ? noise[b * OHW + rem] * noise_w : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = acc[e] + nz + (bias ? bias[8 * g + e] : 0.f);
                if (act == L2I_ACT_LRELU) v = v > 0.f ? v : v * slope;
                else if (act == L2I_ACT_RELU) v = v > 0.f ? v : 0.f;
                acc[e] = v * gain;
            }
        }
        if (mask) {                                            // a gradient through a (leaky) ReLU: * (mask > 0 ? mpos : mneg)
            float m[8];
            unpack8(mask[i], m);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] *= m[e] > 0.f ? mpos : mneg;
        }
        if (addend) {
            float a[8];
            unpack8(addend[i], a);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += a[e];
        }
        y[i] = pack8(acc);
    }
}
// Separable 4x4 FIR without resampling (up = down = 1: the blurs of the generator's up layers and of the discriminator, forward and backward — 90 %
// of the FIR bytes), register-streaming: a lane owns one output column and walks RB output rows downwards; per INPUT row it loads the four
// slots of its horizontal window (neighbouring lanes overlap: L1 hits), forms the horizontal sum once and keeps the last four of them in
// registers for the vertical sum — 4 loads and 8 multiply-adds per output instead of 16 and 16.  Same fused epilogue as the generic kernel.
template <int RB>
__global__ __launch_bounds__(256) void upfirdn2d_h8_sep4_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ x, float4 ty, float4 tx, long long planes, int G8, int in_h, int in_w,
                                                                int out_h, int out_w, int pad_x0, int pad_y0, const float* __restrict__ noise, float noise_w,
                                                                const float* __restrict__ bias, int act, float slope, float gain, const u32x4* __restrict__ mask, float mpos,
                                                                float mneg, const u32x4* __restrict__ addend) {
    const int bands = (out_h + RB - 1) / RB;
    const long long total = planes * bands * out_w;
    const float txa[4] = {tx.x, tx.y, tx.z, tx.w}, tya[4] = {ty.x, ty.y, ty.z, ty.w};
    for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long long)gridDim.x * 256) {
        const int ox = (int)(u % out_w);
        const long long pb = u / out_w;
        const int band = (int)(pb % bands);
        const long long pl = pb / bands;
        const int r0 = band * RB;
        const u32x4* xp = x + pl * in_h * in_w;
        const int ix0 = ox - pad_x0;
        const int g = (int)(pl % G8);
        const long long b = pl / G8;
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = bias ? bias[8 * g + e] : 0.f;
        float ring[4][8];
#pragma unroll
        for (int r = 0; r < RB + 3; ++r) {
            const int iy = r0 - pad_y0 + r;
            float hrow[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) hrow[e] = 0.f;
            if (iy >= 0 && iy < in_h && r0 + r - 3 < out_h + 3) {
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ix = ix0 + kx;
                    if (ix >= 0 && ix < in_w) {
                        float v[8];
                        unpack8(xp[(long long)iy * in_w + ix], v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) hrow[e] += txa[kx] * v[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ring[r & 3][e] = hrow[e];
            if (r >= 3) {
                const int oy = r0 + r - 3;
                if (oy < out_h) {
                    float acc[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[e] = tya[0] * ring[(r - 3) & 3][e] + tya[1] * ring[(r - 2) & 3][e] + tya[2] * ring[(r - 1) & 3][e] + tya[3] * ring[r & 3][e];
                    const long long o = (pl * out_h + oy) * out_w + ox;
                    if (noise || bias || act != L2I_ACT_NONE || gain != 1.f) {
                        const float nz = noise ? noise[(b * out_h + oy) * out_w + ox] * noise_w : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = acc[e] + nz + bs[e];
                            if (act == L2I_ACT_LRELU) v = v > 0.f ? v : v * slope;
                            else if (act == L2I_ACT_RELU) v = v > 0.f ? v : 0.f;
                            acc[e] = v * gain;
                        }
                    }
                    if (mask) {
                        float m[8];
                        unpack8(mask[o], m);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] *= m[e] > 0.f ? mpos : mneg;
                    }
                    if (addend) {
                        float a[8];
                        unpack8(addend[o], a);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] += a[e];
                    }
                    y[o] = pack8(acc);
                }
            }
        }
    }
}

extern "C" int l2i_upfirdn2d_h8(void* y, const void* x, const float* k, int64_t planes, int channels, int in_h, int in_w, int kh, int kw, int up, int down,
                                int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise, float noise_w, const float* bias, int act, float act_slope,
                                float act_gain, const void* mask, float mask_pos, float mask_neg, const void* addend, const float* k1y, const float* k1x, void* stream) {
    if (!y || !x || !k) return l2i_set_error(L2I_E_ARG, "upfirdn2d_h8: null tensor");
    if (planes <= 0 || channels <= 0 || (channels % 8) != 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh > 4 || kw > 4 || (up != 1 && up != 2) || (down != 1 && down != 2))
        return l2i_set_error(L2I_E_ARG, "upfirdn2d_h8: kernels up to 4x4, up / down in {1, 2}, channels % 8 == 0");
    const int out_h = (in_h * up + pad_y0 + pad_y1 - kh) / down + 1, out_w = (in_w * up + pad_x0 + pad_x1 - kw) / down + 1;
    if (out_h <= 0 || out_w <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d_h8: empty output");
    const long long total = (long long)planes * out_h * out_w;
    if (k1y && k1x && kh == 4 && kw == 4 && up == 1 && down == 1) {
        // the caller vouches that k = outer(k1y, k1x) (the path's blurs: [1,3,3,1] x [1,3,3,1] * gain); taps of the flipped kernel
        constexpr int RB = 16;
        const float4 ty = make_float4(k1y[3], k1y[2], k1y[1], k1y[0]), tx = make_float4(k1x[3], k1x[2], k1x[1], k1x[0]);
        const long long units = (long long)planes * ((out_h + RB - 1) / RB) * out_w;
        hipLaunchKernelGGL((upfirdn2d_h8_sep4_kernel<RB>), dim3(l2i_grid_for(units, 256, 256 * 64)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)x, ty, tx,
                           (long long)planes, channels / 8, in_h, in_w, out_h, out_w, pad_x0, pad_y0, noise, noise_w, bias, act, act_slope, act_gain, (const u32x4*)mask,
                           mask_pos, mask_neg, (const u32x4*)addend);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    hipLaunchKernelGGL(upfirdn2d_h8_kernel, dim3(l2i_grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)x, k, (long long)planes,
                       channels / 8, in_h, in_w, out_h, out_w, kh, kw, up, down, pad_x0, pad_y0, noise, noise_w, bias, act, act_slope, act_gain, (const u32x4*)mask, mask_pos,
                       mask_neg, (const u32x4*)addend);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- ToRGB ----------------------------------------------------------------------------------------------------------------------------
// rgb[b,o,p] = sum_c x[b,c,p] * wmod[b,o,c] + bias[o]  (networks.py:346-351; wmod = scale * W * s_rgb from the modulation launch): x h8, rgb fp32
// NCHW (3-channel images stay fp32: they are the interface to the losses).  Lane = pixel, wmod of the sample through LDS.
__global__ __launch_bounds__(256) void torgb_fwd_h8_kernel(float* __restrict__ rgb, const u32x4* __restrict__ x, const float* __restrict__ wmod, const float* __restrict__ bias,
                                                           int C, long long HW, int blocks_per_sample) {
    extern __shared__ float wl[];                          // [3][C]
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x - b * blocks_per_sample;
    for (int i = threadIdx.x; i < 3 * C; i += 256) wl[i] = wmod[(size_t)b * 3 * C + i];
    __syncthreads();
    const int G8 = C / 8;
    for (long long pix = (long long)blk * 256 + threadIdx.x; pix < HW; pix += (long long)blocks_per_sample * 256) {
        float a0 = bias[0], a1 = bias[1], a2 = bias[2];
        for (int g = 0; g < G8; ++g) {
            float v[8];
            unpack8(x[((size_t)b * G8 + g) * HW + pix], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0 += v[e] * wl[8 * g + e]; a1 += v[e] * wl[C + 8 * g + e]; a2 += v[e] * wl[2 * C + 8 * g + e]; }
        }
        rgb[((size_t)b * 3 + 0) * HW + pix] = a0; rgb[((size_t)b * 3 + 1) * HW + pix] = a1; rgb[((size_t)b * 3 + 2) * HW + pix] = a2;
    }
}
extern "C" int l2i_torgb_fwd_h8(float* rgb, const void* x, const float* wmod, const float* bias, int B, int C, int64_t HW, void* stream) {
    if (!rgb || !x || !wmod || !bias || B <= 0 || C <= 0 || (C % 8) != 0 || C > 4096 || HW <= 0) return l2i_set_error(L2I_E_ARG, "torgb_fwd_h8: bad arguments");
    int bps = (int)((HW + 255) / 256);
    if (bps > 512) bps = 512;
    hipLaunchKernelGGL(torgb_fwd_h8_kernel, dim3((unsigned)(B * bps)), dim3(256), (size_t)3 * C * sizeof(float), (hipStream_t)stream, rgb, (const u32x4*)x, wmod, bias, C, (long long)HW, bps);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- fused elementwise backward of a styled conv output (l2i_sg2_act_bwd_f32 on h8 maps) -------------------------------------------------
//   g    = gin[idx] * gin_scale[b,c] + sum_o wmod_rgb[b,o,c] * grgb[b,o,p]              (gin h8 or NULL; grgb fp32 [B,3,HW] or NULL)
//   dz   = g * (y > 0 ? gain : gain * slope)                                              -> dz h8
//   zpre = (y > 0 ? y / gain : y / (gain * slope)) - bias[c] - noise[b,p] * noise_w
//   red_dz_z[b,c] += sum_p dz * zpre        red_x_grgb[b,c,o] += sum_p y * grgb[b,o,p]    (fp32 atomics, one per wave and channel)
// A block walks a strip of pixels of one (sample, channel group): per lane eight running sums (+ 24 for the ToRGB term), reduced over the
// wave at the end.
__global__ __launch_bounds__(256) void sg2_act_bwd_h8_kernel(u32x4* __restrict__ dz, const u32x4* __restrict__ gin, const float* __restrict__ gin_scale,
                                                             const float* __restrict__ grgb, const float* __restrict__ wmod_rgb, const u32x4* __restrict__ y,
                                                             const float* __restrict__ bias, const float* __restrict__ noise, float noise_w, float slope, float gain,
                                                             float* __restrict__ red_dz_z, float* __restrict__ red_x_grgb, int C, long long HW, int strips) {
    const int G8 = C / 8;
    int bid = blockIdx.x;
    const int strip = bid % strips; bid /= strips;
    const int g = bid % G8, b = bid / G8;
    const size_t base = ((size_t)b * G8 + g) * HW;
    float sc[8], bs[8], wr[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = gin_scale ? gin_scale[(size_t)b * C + 8 * g + e] : 1.f;
        bs[e] = bias ? bias[8 * g + e] : 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) wr[o][e] = wmod_rgb ? wmod_rgb[((size_t)b * 3 + o) * C + 8 * g + e] : 0.f;
    }
    float r1[8], r2[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { r1[e] = 0.f; r2[0][e] = r2[1][e] = r2[2][e] = 0.f; }
    const float gp = gain, gn = gain * slope, ip = 1.f / gain, in_ = 1.f / (gain * slope);
    for (long long pix = (long long)strip * 256 + threadIdx.x; pix < HW; pix += (long long)strips * 256) {
        float yv[8], gv[8];
        unpack8(y[base + pix], yv);
        if (gin) unpack8(gin[base + pix], gv);
        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
        if (grgb) { q0 = grgb[((size_t)b * 3 + 0) * HW + pix]; q1 = grgb[((size_t)b * 3 + 1) * HW + pix]; q2 = grgb[((size_t)b * 3 + 2) * HW + pix]; }
        const float nz = noise ? noise[(size_t)b * HW + pix] * noise_w : 0.f;
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float gg = gin ? gv[e] * sc[e] : 0.f;
            gg += wr[0][e] * q0 + wr[1][e] * q1 + wr[2][e] * q2;
            const bool pos = yv[e] > 0.f;
            d[e] = gg * (pos ? gp : gn);
            const float zpre = yv[e] * (pos ? ip : in_) - bs[e] - nz;
            r1[e] += d[e] * zpre;
            r2[0][e] += yv[e] * q0; r2[1][e] += yv[e] * q1; r2[2][e] += yv[e] * q2;
        }
        dz[base + pix] = pack8(d);
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = wave_sum(r1[e]);
        if (lane == 0) atomicAdd(red_dz_z + (size_t)b * C + 8 * g + e, s);
        if (red_x_grgb) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float t = wave_sum(r2[o][e]);
                if (lane == 0) atomicAdd(red_x_grgb + ((size_t)b * C + 8 * g + e) * 3 + o, t);
            }
        }
    }
}
extern "C" int l2i_sg2_act_bwd_h8(void* dz, const void* gin, const float* gin_scale, const float* grgb, const float* wmod_rgb, const void* y, const float* bias,
                                  const float* noise, float noise_w, float slope, float gain, float* red_dz_z, float* red_x_grgb, int B, int C, int64_t HW, void* stream) {
    if (!dz || !y || !red_dz_z || B <= 0 || C <= 0 || (C % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd_h8: bad arguments");
    if ((grgb != nullptr) != (wmod_rgb != nullptr)) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd_h8: grgb and wmod_rgb go together");
    int strips = (int)((HW + 256 * 8 - 1) / (256 * 8));
    if (strips < 1) strips = 1;
    if (strips > 64) strips = 64;
    hipLaunchKernelGGL(sg2_act_bwd_h8_kernel, dim3((unsigned)(B * (C / 8) * strips)), dim3(256), 0, (hipStream_t)stream, (u32x4*)dz, (const u32x4*)gin, gin_scale, grgb, wmod_rgb,
                       (const u32x4*)y, bias, noise, noise_w, slope, gain, red_dz_z, grgb ? red_x_grgb : nullptr, C, (long long)HW, strips);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- out[b,c] += sum_p a[b,c,p] * (b ? b[b,c,p] : 1) on h8 maps ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dot_reduce_h8_kernel(float* __restrict__ out, const u32x4* __restrict__ a, const u32x4* __restrict__ bb, int C, long long HW, int strips) {
    const int G8 = C / 8;
    int bid = blockIdx.x;
    const int strip = bid % strips; bid /= strips;
    const int g = bid % G8, b = bid / G8;
    const size_t base = ((size_t)b * G8 + g) * HW;
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = 0.f;
    for (long long pix = (long long)strip * 256 + threadIdx.x; pix < HW; pix += (long long)strips * 256) {
        float av[8], bv[8];
        unpack8(a[base + pix], av);
        if (bb) {
            unpack8(bb[base + pix], bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] += av[e] * bv[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] += av[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = wave_sum(r[e]);
        if ((threadIdx.x & 63) == 0) atomicAdd(out + (size_t)b * C + 8 * g + e, s);
    }
}
extern "C" int l2i_dot_reduce_h8(float* out, const void* a, const void* b, int B, int C, int64_t HW, void* stream) {
    if (!out || !a || B <= 0 || C <= 0 || (C % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "dot_reduce_h8: bad arguments");
    int strips = (int)((HW + 256 * 8 - 1) / (256 * 8));
    if (strips < 1) strips = 1;
    if (strips > 64) strips = 64;
    hipLaunchKernelGGL(dot_reduce_h8_kernel, dim3((unsigned)(B * (C / 8) * strips)), dim3(256), 0, (hipStream_t)stream, out, (const u32x4*)a, (const u32x4*)b, C, (long long)HW, strips);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- max-pool ---------------------------------------------------------------------------------------------------------------------------
// MaxPool2d(k, s, pad) per channel on h8 planes; idx = window-local arg-max per element (first maximum in row-major order, NaN propagates,
// like l2i_maxpool2d_fwd_f32) as [planes][OH][OW][8] bytes.  relu != 0: y = max(pool, 0) (the ReLU that follows the pool commutes with it;
// the next conv then needs no ReLU-on-load and the backward mask y > 0 is unchanged).
__global__ __launch_bounds__(256) void maxpool_fwd_h8_kernel(u32x4* __restrict__ y, uint2* __restrict__ idx, const u32x4* __restrict__ x, long long planes, int H, int W,
                                                             int k, int s, int pad, int OH, int OW, int relu) {
    const long long OHW = (long long)OH * OW, total = planes * OHW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pl = i / OHW;
        const unsigned rem = (unsigned)(i - pl * OHW);
        const int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * (unsigned)OW);
        const u32x4* xp = x + pl * H * W;
        float best[8];
        unsigned bi[8];
        bool found = false;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - pad + kx;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                unpack8(xp[(long long)iy * W + ix], v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!found || v[e] > best[e] || (v[e] != v[e])) { best[e] = v[e]; bi[e] = (unsigned)(ky * k + kx); }
                found = true;
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) best[e] = best[e] > 0.f ? best[e] : (best[e] != best[e] ? best[e] : 0.f);
        }
        y[i] = pack8(best);
        idx[i] = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
    }
}
// backward: gx[iy, ix] = sum over the windows that contain (iy, ix) and whose arg-max it is of gy (+ coef * coef_dev * (b - a) when a / b are given:
// the ContentLoss direct term of the pooled tap, l2i_maxpool2x2_bwd_add_diff_f32)
__global__ __launch_bounds__(256) void maxpool_bwd_h8_kernel(u32x4* __restrict__ gx, const u32x4* __restrict__ gy, const uint2* __restrict__ idx, const u32x4* __restrict__ a,
                                                             const u32x4* __restrict__ bq, float coef, const float* __restrict__ coef_dev, long long planes, int H, int W,
                                                             int k, int s, int pad, int OH, int OW) {
    const long long HW = (long long)H * W, total = planes * HW;
    const float cf = a ? coef * (coef_dev ? coef_dev[0] : 1.f) : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pl = i / HW;
        const unsigned rem = (unsigned)(i - pl * HW);
        const int iy = (int)(rem / (unsigned)W), ix = (int)(rem - (unsigned)iy * (unsigned)W);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // windows oy with oy*s - pad <= iy <= oy*s - pad + k - 1
        const int ny = iy + pad - k + 1, nx = ix + pad - k + 1;
        const int oy_lo = ny > 0 ? (ny + s - 1) / s : 0, ox_lo = nx > 0 ? (nx + s - 1) / s : 0;
        const int oy_hi = (iy + pad) / s < OH - 1 ? (iy + pad) / s : OH - 1, ox_hi = (ix + pad) / s < OW - 1 ? (ix + pad) / s : OW - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const int ky = iy - (oy * s - pad);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int kx = ix - (ox * s - pad);
                const unsigned me = (unsigned)(ky * k + kx);
                const long long o = pl * OH * OW + (long long)oy * OW + ox;
                const uint2 id = idx[o];
                float g[8];
                unpack8(gy[o], g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w8 = ((e < 4 ? id.x : id.y) >> (8 * (e & 3))) & 0xffu;
                    if (w8 == me) acc[e] += g[e];
                }
            }
        }
        if (a) {
            float av[8], bv[8];
            unpack8(a[i], av);
            unpack8(bq[i], bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += cf * (bv[e] - av[e]);
        }
        gx[i] = pack8(acc);
    }
}
extern "C" int l2i_maxpool2d_fwd_h8(void* y, void* idx, const void* x, int64_t planes, int H, int W, int k, int s, int pad, int OH, int OW, int relu, void* stream) {
    if (!y || !idx || !x || planes <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || s <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "maxpool_fwd_h8: bad arguments");
    hipLaunchKernelGGL(maxpool_fwd_h8_kernel, dim3(l2i_grid_for((long long)planes * OH * OW, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (uint2*)idx, (const u32x4*)x,
                       (long long)planes, H, W, k, s, pad, OH, OW, relu);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
extern "C" int l2i_maxpool2d_bwd_h8(void* gx, const void* gy, const void* idx, const void* a, const void* b, float coef, const float* coef_dev, int64_t planes, int H, int W,
                                    int k, int s, int pad, int OH, int OW, void* stream) {
    if (!gx || !gy || !idx || planes <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || s <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "maxpool_bwd_h8: bad arguments");
    if ((a != nullptr) != (b != nullptr)) return l2i_set_error(L2I_E_ARG, "maxpool_bwd_h8: a and b go together");
    hipLaunchKernelGGL(maxpool_bwd_h8_kernel, dim3(l2i_grid_for((long long)planes * H * W, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)gx, (const u32x4*)gy,
                       (const uint2*)idx, (const u32x4*)a, (const u32x4*)b, coef, coef_dev, (long long)planes, H, W, k, s, pad, OH, OW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- ContentLoss difference on h8 maps: sum_out[0] += sum (a-b)^2 ; grad = coef * coef_dev * (b - a)  (l2i_sqdiff_f32) -------------------------
__global__ __launch_bounds__(256) void sqdiff_h8_kernel(float* __restrict__ sum_out, u32x4* __restrict__ grad, const u32x4* __restrict__ a, const u32x4* __restrict__ b, long long n,
                                                        float coef, const float* __restrict__ coef_dev) {
    __shared__ float red[4];
    const float cf = coef * (coef_dev ? coef_dev[0] : 1.f);
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float av[8], bv[8], g[8];
        unpack8(a[i], av);
        unpack8(b[i], bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = bv[e] - av[e]; s += d * d; g[e] = cf * d; }
        if (grad) grad[i] = pack8(g);
    }
    if (sum_out) {
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sum_out, (red[0] + red[1]) + (red[2] + red[3]));
    }
}
extern "C" int l2i_sqdiff_h8(float* sum_out, void* grad, const void* a, const void* b, int64_t slots, float coef, const float* coef_dev, void* stream) {
    if (!a || !b || slots <= 0) return l2i_set_error(L2I_E_ARG, "sqdiff_h8: bad arguments");
    hipLaunchKernelGGL(sqdiff_h8_kernel, dim3(l2i_grid_for(slots, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, sum_out, (u32x4*)grad, (const u32x4*)a, (const u32x4*)b,
                       (long long)slots, coef, coef_dev);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- y[2 oy, 2 ox] += c[oy, ox] (zero insertion: the input-gradient of a strided 1x1 conv added to the gradient of the other branch) ------------
__global__ __launch_bounds__(256) void add_zero_insert_h8_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ c, const u32x4* __restrict__ mask, long long planes, int H, int W,
                                                                 int OH, int OW) {
    const long long OHW = (long long)OH * OW, total = planes * OHW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pl = i / OHW;
        const unsigned rem = (unsigned)(i - pl * OHW);
        const int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * (unsigned)OW);
        if (2 * oy >= H || 2 * ox >= W) continue;
        const long long o = pl * H * W + (long long)(2 * oy) * W + 2 * ox;
        float a[8], b[8];
        unpack8(y[o], a);
        unpack8(c[i], b);
        if (mask) {                                            // ReLU mask of the map the sum is the gradient of
            float m[8];
            unpack8(mask[o], m);
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = m[e] > 0.f ? b[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        y[o] = pack8(a);
    }
}
extern "C" int l2i_add_zero_insert_h8(void* y, const void* c, const void* mask, int64_t planes, int H, int W, int OH, int OW, void* stream) {
    if (!y || !c || planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "add_zero_insert_h8: bad arguments");
    hipLaunchKernelGGL(add_zero_insert_h8_kernel, dim3(l2i_grid_for((long long)planes * OH * OW, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)c,
                       (const u32x4*)mask, (long long)planes, H, W, OH, OW);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- per-sample weight planes of a modulated conv ------------------------------------------------------------------------------------------
// planes[b][...][co][e] = bf16( w32[...][co][e] * s[b, channel of (.., e)] )   (networks.py:234-235: weight * style, evaluated per sample; the
// demodulation factor stays an out_scale of the conv's epilogue).  w32: the fp32 weights in the SAME plane order [Cin/16][KK][2][CoutP][8]
// (channel of an element = 16 * (index / (KK*2*CoutP*8)) + 8 * half + e); one launch per layer and pass.
__global__ __launch_bounds__(256) void modulate_planes_kernel(u32x4* __restrict__ planes, const float* __restrict__ w32, const float* __restrict__ s, int Cs, long long slots_per_sample,
                                                              int KK, int CoutP, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / slots_per_sample, sl = i - b * slots_per_sample;
        const long long r = sl / CoutP;                        // ((c16 * KK + tap) * 2 + half)
        const int half = (int)(r & 1), c16 = (int)((r >> 1) / KK);
        const float4 w0 = *reinterpret_cast<const float4*>(w32 + sl * 8), w1 = *reinterpret_cast<const float4*>(w32 + sl * 8 + 4);
        const float* sp = s + b * Cs + 16 * c16 + 8 * half;
        float v[8] = {w0.x * sp[0], w0.y * sp[1], w0.z * sp[2], w0.w * sp[3], w1.x * sp[4], w1.y * sp[5], w1.z * sp[6], w1.w * sp[7]};
        planes[i] = pack8(v);
    }
}
extern "C" int l2i_modulate_planes_h8(void* planes, const float* w32, const float* s, int B, int Cs, int CinP, int KK, int CoutP, void* stream) {
    if (!planes || !w32 || !s || B <= 0 || CinP <= 0 || (CinP % 16) != 0 || Cs > CinP || KK <= 0 || CoutP <= 0) return l2i_set_error(L2I_E_ARG, "modulate_planes_h8: bad arguments");
    if (Cs != CinP) return l2i_set_error(L2I_E_ARG, "modulate_planes_h8: the scale vector must cover the padded channel count");
    const long long sps = (long long)(CinP / 16) * KK * 2 * CoutP, total = sps * B;
    hipLaunchKernelGGL(modulate_planes_kernel, dim3(l2i_grid_for(total, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (u32x4*)planes, w32, s, Cs, sps, KK, CoutP, total);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ---- y = g * (ref > 0 ? pos : neg): a gradient through a (leaky) ReLU whose output `ref` was saved (the conv kernels of this path have no
// prologue, so a mask that cannot ride on the producing epilogue — the gradient also feeds an unmasked branch — is one pass) ----------------------
__global__ __launch_bounds__(256) void mask_mul_h8_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ g, const u32x4* __restrict__ ref, float pos, float neg, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float a[8], m[8];
        unpack8(g[i], a);
        unpack8(ref[i], m);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= m[e] > 0.f ? pos : neg;
        y[i] = pack8(a);
    }
}
extern "C" int l2i_mask_mul_h8(void* y, const void* g, const void* ref, float pos, float neg, int64_t slots, void* stream) {
    if (!y || !g || !ref || slots <= 0) return l2i_set_error(L2I_E_ARG, "mask_mul_h8: bad arguments");
    hipLaunchKernelGGL(mask_mul_h8_kernel, dim3(l2i_grid_for(slots, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)g, (const u32x4*)ref, pos, neg, (long long)slots);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
