// l2i_stream_h8.hip — the streaming (HBM-bound) companions of l2i_conv_h8.hip on bf16 tensors in the channel-blocked h8 layout
// [B][C/8][H][W][8] (gfx950): layout casts, upfirdn2d (the 4x4 blurs of the generator's up layers and of the discriminator, with the
// generator's noise + bias + leaky-ReLU epilogue), ToRGB, the fused activation backward of a styled conv with its two style reductions,
// pixel reductions, max-pools, the ContentLoss difference, zero insertion.  Same functions as their fp32 NCHW counterparts in
// l2i_stream.hip (reference call sites are cited there and in include/l2i.h); arithmetic in fp32 registers, 16-bit only in HBM.
// A thread works on one 16-byte pixel slot (the 8 channels of a pixel): consecutive lanes = consecutive pixels = consecutive 16 bytes, so
// every global access is a full-width coalesced one whatever the channel count, and a pixel's channel group is lane-local.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// [r5] Compiled twice like l2i_conv_h8.hip: bf16 elements as is, IEEE fp16 elements with -DL2I_H8_F16 (entry points with the suffix _f16).
// Only the three converts below know the element type.
#ifdef L2I_H8_F16
#define H8_NS l2i_h8s_f16
#define H8_NAME(n) n##_f16
#else
#define H8_NS l2i_h8s_bf16
#define H8_NAME(n) n
#endif

namespace H8_NS {
namespace {
#ifdef L2I_H8_F16
__device__ __forceinline__ float blo(unsigned u) { float r; asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(u)); return r; }
__device__ __forceinline__ float bhi(unsigned u) { float r; asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(u)); return r; }
#else
__device__ __forceinline__ float blo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#endif
__device__ __forceinline__ void unpack8(const u32x4& u, float (&v)[8]) {
    v[0] = blo(u.x); v[1] = bhi(u.x); v[2] = blo(u.y); v[3] = bhi(u.y); v[4] = blo(u.z); v[5] = bhi(u.z); v[6] = blo(u.w); v[7] = bhi(u.w);
}
__device__ __forceinline__ unsigned pk(float lo, float hi) {
    unsigned r;
#ifdef L2I_H8_F16
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#else
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
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
extern "C" int H8_NAME(l2i_cast_f32_to_h8)(void* y, const float* x, int B, int C, int Cpad, int64_t HW, void* stream) {
    if (!y || !x || B <= 0 || C <= 0 || Cpad < C || (Cpad % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "cast_f32_to_h8: bad arguments");
    const long long total = (long long)B * (Cpad / 8) * HW;
    hipLaunchKernelGGL(cast_f32_to_h8_kernel, dim3(l2i_grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, x, C, Cpad / 8, (long long)HW, total);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
extern "C" int H8_NAME(l2i_cast_h8_to_f32)(float* y, const void* x, int B, int C, int Cpad, int64_t HW, void* stream) {
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
// of the FIR bytes), register-streaming.  A WAVE owns 61 output columns of a band of RB output rows: lane l loads input column c0 + l of every
// input row ONCE (one 16-byte buffer load per lane and row; out-of-image rows and columns read zeros through the descriptor's range check) and
// the horizontal sum of output column c0 + l - 3 + pad is built by passing partial sums one lane to the right three times (DPP wave_shr:1, no LDS,
// no second load):  a = t0 v;  b = t1 v + a(l-1);  c = t2 v + b(l-1);  h = t3 v + c(l-1)  =  sum_k t_k v(l - 3 + k).
// Lanes 0-2 of a wave only feed their neighbours (61 of 64 lanes store).  The last four horizontal sums stay in registers for the vertical sum;
// input rows are requested PF rows ahead of their use (the first version loaded its four window slots per row with nothing in flight behind
// them: 2.8 TB/s, latency bound).  Same fused epilogue as the generic kernel.
__device__ __forceinline__ float lane_shr1(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x138, 0xf, 0xf, false));   // wave_shr:1 (lane 0 gets 0)
}
// MBITS [r6]: `mask` is a sign plane (one byte per pixel slot, bit e = element e > 0: l2i.h, l2i_conv_params::mask_out) instead of an h8 map — the
// discriminator's conv1 outputs are read by this kernel for their signs only (1/16 of the bytes)
template <int RB, bool MBITS = false>
__global__ __launch_bounds__(256) void upfirdn2d_h8_sep4_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ x, float4 ty, float4 tx, long long planes, int G8, int in_h, int in_w,
                                                                int out_h, int out_w, int pad_x0, int pad_y0, const float* __restrict__ noise, float noise_w,
                                                                const float* __restrict__ bias, int act, float slope, float gain, const u32x4* __restrict__ mask, float mpos,
                                                                float mneg, const u32x4* __restrict__ addend) {
    constexpr int PF = 4, EP = 4;                            // input rows / epilogue-operand rows in flight per lane
    constexpr int NR = RB + 3;
    const int lane = threadIdx.x & 63;
    const int chunks = (out_w + 60) / 61, bands = (out_h + RB - 1) / RB;
    const long long total = planes * bands * chunks;
    const unsigned plane_bytes = (unsigned)in_h * (unsigned)in_w * 16u;
    const bool epi = noise || bias || act != L2I_ACT_NONE || gain != 1.f;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform unit index: the buffer descriptor lives in SGPRs
    for (long long u = (long long)blockIdx.x * 4 + wv; u < total; u += (long long)gridDim.x * 4) {
        const int chunk = (int)(u % chunks);
        const long long pb = u / chunks;
        const int band = (int)(pb % bands);
        const long long pl = pb / bands;
        const int r0 = band * RB;
        const int ox = chunk * 61 + lane - 3;                  // this lane's output column (lanes 0-2: none)
        const int ix = ox - pad_x0 + 3;                        // its input column: the LAST tap of its window
        const bool colok = ix >= 0 && ix < in_w;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + pl * (long long)in_h * in_w), 0, plane_bytes, 0x00020000);
        const int g = (int)(pl % G8);
        const long long b = pl / G8;
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = bias ? bias[8 * g + e] : 0.f;
        const int iy0 = r0 - pad_y0;
        // byte offset of (row, this lane's column); rows above / below the image and columns outside it land outside the descriptor's range: zeros
        auto off = [&](int r) -> int { return colok ? ((iy0 + r) * in_w + ix) * 16 : -1; };
        // epilogue operands of an output row (noise, mask, addend) are requested EP rows ahead as well: loads return in order, so an operand
        // fetched at its point of use would wait for every input row in flight behind it
        const bool lane_out = lane >= 3 && ox < out_w;
        float nzq[EP];
        u32x4 mq[EP], aq[EP];
        auto issue_ops = [&](int j, int slot) {                     // slot = j % EP, a compile-time index at every call site
            const int oy = r0 + j;
            const bool ok = lane_out && oy < out_h;
            const long long o = ok ? (pl * out_h + oy) * out_w + ox : 0;
            if (noise) nzq[slot] = noise[ok ? (b * out_h + oy) * out_w + ox : 0];
            if (mask) { if (MBITS) mq[slot].x = reinterpret_cast<const uint8_t*>(mask)[o]; else mq[slot] = mask[o]; }
            if (addend) aq[slot] = addend[o];
        };
        u32x4 q[PF];
#pragma unroll
        for (int r = 0; r < PF; ++r) q[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off(r), 0, 0));
#pragma unroll
        for (int j = 0; j < EP; ++j) issue_ops(j, j);
        float ring[4][8];
        // rows in groups of four: ring / prefetch slots are compile-time indices inside a group, the group loop is a real loop (a fully unrolled band
        // with its runtime epilogue switches is 60 KB of code: past the instruction cache)
        for (int rb = 0; rb < NR; rb += 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = rb + i;
            if (r >= NR) break;
            float v[8];
            unpack8(q[i], v);
            q[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, r + PF < NR ? off(r + PF) : -1, 0, 0));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = tx.x * v[e];
                const float bq = fmaf(tx.y, v[e], lane_shr1(a));
                const float c = fmaf(tx.z, v[e], lane_shr1(bq));
                ring[i][e] = fmaf(tx.w, v[e], lane_shr1(c));
            }
            if (r >= 3) {
                const int j = r - 3, oy = r0 + j;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = ty.x * ring[(i + 1) & 3][e] + ty.y * ring[(i + 2) & 3][e] + ty.z * ring[(i + 3) & 3][e] + ty.w * ring[i][e];
                if (epi) {
                    const float nz = noise ? nzq[(i + 1) & 3] * noise_w : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float w = acc[e] + nz + bs[e];
                        if (act == L2I_ACT_LRELU) w = w > 0.f ? w : w * slope;
                        else if (act == L2I_ACT_RELU) w = w > 0.f ? w : 0.f;
                        acc[e] = w * gain;
                    }
                }
                if (mask) {
                    if (MBITS) {
                        const unsigned mb = mq[(i + 1) & 3].x;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] *= ((mb >> e) & 1u) ? mpos : mneg;
                    } else {
                        float m[8];
                        unpack8(mq[(i + 1) & 3], m);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] *= m[e] > 0.f ? mpos : mneg;
                    }
                }
                if (addend) {
                    float a2[8];
                    unpack8(aq[(i + 1) & 3], a2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += a2[e];
                }
                if (j + EP < RB) issue_ops(j + EP, (i + 1) & 3);
                if (lane_out && oy < out_h) y[(pl * out_h + oy) * out_w + ox] = pack8(acc);
            }
          }
        }
    }
}

// The discriminator's skip path (networks.py:586-590) in the same register-streaming form.  Down: blur with pad (1, 1) evaluated at every second
// pixel — a lane owns one OUTPUT column, loads its two input columns 2 ox, 2 ox + 1 of every input row once and takes column 2 ox - 1 from its
// left neighbour and 2 ox + 2 from its right one (DPP wave_shr:1 / wave_shl:1); 62 of 64 lanes store.  Up: its adjoint, zero insertion + blur with
// pad (2, 1) — a lane owns one INPUT column j and produces output columns 2 j (t0 in[j-1] + t2 in[j]) and 2 j + 1 (t1 in[j] + t3 in[j+1]) of
// output rows 2 i (t0 H[i-1] + t2 H[i]) and 2 i + 1 (t1 H[i] + t3 H[i+1]): every input slot is loaded once, every output slot (and addend slot)
// touched once, no tap is multiplied by an inserted zero.
__device__ __forceinline__ float lane_shl1(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x130, 0xf, 0xf, false));   // wave_shl:1 (lane 63 gets 0)
}
template <int RB>
__global__ __launch_bounds__(256) void upfirdn2d_h8_sep4_down2_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ x, float4 ty, float4 tx, long long planes, int in_h, int in_w,
                                                                      int out_h, int out_w) {
    constexpr int PF = 4, NR = 2 * RB + 2;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunks = (out_w + 61) / 62, bands = (out_h + RB - 1) / RB;
    const long long total = planes * bands * chunks;
    const unsigned plane_bytes = (unsigned)in_h * (unsigned)in_w * 16u;
    for (long long u = (long long)blockIdx.x * 4 + wv; u < total; u += (long long)gridDim.x * 4) {
        const int chunk = (int)(u % chunks);
        const long long pb = u / chunks;
        const int band = (int)(pb % bands);
        const long long pl = pb / bands;
        const int r0 = band * RB;
        const int ox = chunk * 62 + lane - 1;                  // lanes 1 .. 62 store
        const int ixa = 2 * ox, ixb = 2 * ox + 1;
        const bool oka = ixa >= 0 && ixa < in_w, okb = ixb >= 0 && ixb < in_w;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + pl * (long long)in_h * in_w), 0, plane_bytes, 0x00020000);
        const int iy0 = 2 * r0 - 1;
        auto offa = [&](int r) -> int { return oka ? ((iy0 + r) * in_w + ixa) * 16 : -1; };
        auto offb = [&](int r) -> int { return okb ? ((iy0 + r) * in_w + ixb) * 16 : -1; };
        u32x4 qa[PF], qb[PF];
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            qa[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offa(r), 0, 0));
            qb[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offb(r), 0, 0));
        }
        float ring[4][8];
        const bool lane_out = lane >= 1 && lane <= 62 && ox < out_w;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float va[8], vb[8];
            unpack8(qa[r % PF], va);
            unpack8(qb[r % PF], vb);
            if (r + PF < NR) {
                qa[r % PF] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offa(r + PF), 0, 0));
                qb[r % PF] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offb(r + PF), 0, 0));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                ring[r & 3][e] = fmaf(tx.x, lane_shr1(vb[e]), fmaf(tx.y, va[e], fmaf(tx.z, vb[e], tx.w * lane_shl1(va[e]))));
            if (r >= 3 && ((r - 3) & 1) == 0) {
                const int oy = r0 + (r - 3) / 2;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = ty.x * ring[(r - 3) & 3][e] + ty.y * ring[(r - 2) & 3][e] + ty.z * ring[(r - 1) & 3][e] + ty.w * ring[r & 3][e];
                if (lane_out && oy < out_h) y[(pl * out_h + oy) * out_w + ox] = pack8(acc);
            }
        }
    }
}
template <int RB>
__global__ __launch_bounds__(256) void upfirdn2d_h8_sep4_up2_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ x, float4 ty, float4 tx, long long planes, int in_h, int in_w,
                                                                    int out_h, int out_w, const u32x4* __restrict__ addend) {
    constexpr int PF = 4, NR = RB + 2;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunks = (in_w + 61) / 62, bands = (in_h + RB - 1) / RB;
    const long long total = planes * bands * chunks;
    const unsigned plane_bytes = (unsigned)in_h * (unsigned)in_w * 16u;
    for (long long u = (long long)blockIdx.x * 4 + wv; u < total; u += (long long)gridDim.x * 4) {
        const int chunk = (int)(u % chunks);
        const long long pb = u / chunks;
        const int band = (int)(pb % bands);
        const long long pl = pb / bands;
        const int r0 = band * RB;
        const int jx = chunk * 62 + lane - 1;                  // this lane's input column; lanes 1 .. 62 store output columns 2 jx, 2 jx + 1
        const bool ok = jx >= 0 && jx < in_w;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + pl * (long long)in_h * in_w), 0, plane_bytes, 0x00020000);
        auto off = [&](int r) -> int { return ok ? ((r0 - 1 + r) * in_w + jx) * 16 : -1; };      // rows above / below the map: out of range, zeros
        const bool lane_out = lane >= 1 && lane <= 62 && ok;
        u32x4* const yp = y + pl * (long long)out_h * out_w + 2 * jx;
        const u32x4* const ap = addend ? addend + pl * (long long)out_h * out_w + 2 * jx : nullptr;
        u32x4 q[PF];
#pragma unroll
        for (int r = 0; r < PF; ++r) q[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off(r), 0, 0));
        // addend slots of the rows a step emits (odd row of input row i - 1, even row of input row i), requested one step ahead
        u32x4 ad[2][4];
        auto issue_add = [&](int r, int slot) {                 // step r: input row i = r0 - 1 + r; rows 2 i - 1 (r >= 2) and 2 i (1 <= r <= RB)
            const int i = r0 - 1 + r;
            const int oyo = 2 * i - 1, oye = 2 * i;
            const bool vo = lane_out && r >= 2 && r <= RB + 1 && oyo < out_h, ve = lane_out && r >= 1 && r <= RB && oye < out_h;
            const long long bo = vo ? (long long)oyo * out_w : -2LL * jx, be = ve ? (long long)oye * out_w : -2LL * jx;      // invalid: slot 0 of the plane
            ad[slot][0] = ap[bo]; ad[slot][1] = ap[bo + (vo ? 1 : 0)]; ad[slot][2] = ap[be]; ad[slot][3] = ap[be + (ve ? 1 : 0)];
        };
        float pe[8], po[8];                                     // horizontal sums of the previous input row: even / odd output columns
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = po[e] = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float v[8], he[8], ho[8];
            unpack8(q[r % PF], v);
            if (r + PF < NR) q[r % PF] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off(r + PF), 0, 0));
            if (ap && r + 1 < NR) issue_add(r + 1, (r + 1) & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                he[e] = fmaf(tx.x, lane_shr1(v[e]), tx.z * v[e]);
                ho[e] = fmaf(tx.y, v[e], tx.w * lane_shl1(v[e]));
            }
            const int i = r0 - 1 + r;
            if (r >= 2) {                                       // odd output row of input row i - 1: t1 H[i-1] + t3 H[i]
                const int oy = 2 * i - 1;
                float a0[8], a1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] = fmaf(ty.y, pe[e], ty.w * he[e]); a1[e] = fmaf(ty.y, po[e], ty.w * ho[e]); }
                if (ap) {
                    float t0[8], t1[8];
                    unpack8(ad[r & 1][0], t0); unpack8(ad[r & 1][1], t1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a0[e] += t0[e]; a1[e] += t1[e]; }
                }
                if (lane_out && oy < out_h) { yp[(long long)oy * out_w] = pack8(a0); yp[(long long)oy * out_w + 1] = pack8(a1); }
            }
            if (r >= 1 && r <= RB) {                            // even output row of input row i: t0 H[i-1] + t2 H[i]
                const int oy = 2 * i;
                float a0[8], a1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] = fmaf(ty.x, pe[e], ty.z * he[e]); a1[e] = fmaf(ty.x, po[e], ty.z * ho[e]); }
                if (ap) {
                    float t0[8], t1[8];
                    unpack8(ad[r & 1][2], t0); unpack8(ad[r & 1][3], t1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a0[e] += t0[e]; a1[e] += t1[e]; }
                }
                if (lane_out && oy < out_h) { yp[(long long)oy * out_w] = pack8(a0); yp[(long long)oy * out_w + 1] = pack8(a1); }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { pe[e] = he[e]; po[e] = ho[e]; }
        }
    }
}

extern "C" int H8_NAME(l2i_upfirdn2d_h8)(void* y, const void* x, const float* k, int64_t planes, int channels, int in_h, int in_w, int kh, int kw, int up, int down,
                                int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise, float noise_w, const float* bias, int act, float act_slope,
                                float act_gain, const void* mask, float mask_pos, float mask_neg, const void* addend, const float* k1y, const float* k1x, int mask_bits,
                                void* stream) {
    if (!y || !x || !k) return l2i_set_error(L2I_E_ARG, "upfirdn2d_h8: null tensor");
    if (mask_bits && !(mask && k1y && k1x && kh == 4 && kw == 4 && up == 1 && down == 1))
        return l2i_set_error(L2I_E_UNSUPPORTED, "upfirdn2d_h8: a sign-plane mask rides on the separable 4x4 blur without resampling only");
    if (planes <= 0 || channels <= 0 || (channels % 8) != 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh > 4 || kw > 4 || (up != 1 && up != 2) || (down != 1 && down != 2))
        return l2i_set_error(L2I_E_ARG, "upfirdn2d_h8: kernels up to 4x4, up / down in {1, 2}, channels % 8 == 0");
    const int out_h = (in_h * up + pad_y0 + pad_y1 - kh) / down + 1, out_w = (in_w * up + pad_x0 + pad_x1 - kw) / down + 1;
    if (out_h <= 0 || out_w <= 0) return l2i_set_error(L2I_E_ARG, "upfirdn2d_h8: empty output");
    const long long total = (long long)planes * out_h * out_w;
    if (k1y && k1x && kh == 4 && kw == 4 && up == 1 && down == 1) {
        // the caller vouches that k = outer(k1y, k1x) (the path's blurs: [1,3,3,1] x [1,3,3,1] * gain); taps of the flipped kernel
        constexpr int RB = 16;
        const float4 ty = make_float4(k1y[3], k1y[2], k1y[1], k1y[0]), tx = make_float4(k1x[3], k1x[2], k1x[1], k1x[0]);
        const long long waves = (long long)planes * ((out_h + RB - 1) / RB) * ((out_w + 60) / 61);
        if (mask_bits)
            hipLaunchKernelGGL((upfirdn2d_h8_sep4_kernel<RB, true>), dim3(l2i_grid_for(waves, 4, 256 * 64)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)x, ty, tx,
                               (long long)planes, channels / 8, in_h, in_w, out_h, out_w, pad_x0, pad_y0, noise, noise_w, bias, act, act_slope, act_gain, (const u32x4*)mask,
                               mask_pos, mask_neg, (const u32x4*)addend);
        else
        hipLaunchKernelGGL((upfirdn2d_h8_sep4_kernel<RB>), dim3(l2i_grid_for(waves, 4, 256 * 64)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)x, ty, tx,
                           (long long)planes, channels / 8, in_h, in_w, out_h, out_w, pad_x0, pad_y0, noise, noise_w, bias, act, act_slope, act_gain, (const u32x4*)mask,
                           mask_pos, mask_neg, (const u32x4*)addend);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    const bool plain = !noise && !bias && act == L2I_ACT_NONE && act_gain == 1.f && !mask;
    if (k1y && k1x && kh == 4 && kw == 4 && plain && up == 1 && down == 2 && !addend && pad_x0 == 1 && pad_y0 == 1 && (long long)in_h * in_w * 16 < 0x7fffffffLL) {
        constexpr int RB = 8;                              // the discriminator's skip blur, evaluated where the stride-2 1x1 samples it
        const float4 ty = make_float4(k1y[3], k1y[2], k1y[1], k1y[0]), tx = make_float4(k1x[3], k1x[2], k1x[1], k1x[0]);
        const long long waves = (long long)planes * ((out_h + RB - 1) / RB) * ((out_w + 61) / 62);
        hipLaunchKernelGGL((upfirdn2d_h8_sep4_down2_kernel<RB>), dim3(l2i_grid_for(waves, 4, 256 * 64)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)x, ty, tx,
                           (long long)planes, in_h, in_w, out_h, out_w);
        L2I_CHECK_LAUNCH();
        return L2I_OK;
    }
    if (k1y && k1x && kh == 4 && kw == 4 && plain && up == 2 && down == 1 && pad_x0 == 2 && pad_y0 == 2 && out_h == 2 * in_h && out_w == 2 * in_w &&
        (long long)in_h * in_w * 16 < 0x7fffffffLL) {
        constexpr int RB = 8;                              // its adjoint (zero insertion + blur), with the skip sum as addend
        const float4 ty = make_float4(k1y[3], k1y[2], k1y[1], k1y[0]), tx = make_float4(k1x[3], k1x[2], k1x[1], k1x[0]);
        const long long waves = (long long)planes * ((in_h + RB - 1) / RB) * ((in_w + 61) / 62);
        hipLaunchKernelGGL((upfirdn2d_h8_sep4_up2_kernel<RB>), dim3(l2i_grid_for(waves, 4, 256 * 64)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)x, ty, tx,
                           (long long)planes, in_h, in_w, out_h, out_w, (const u32x4*)addend);
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
    extern __shared__ __attribute__((aligned(16))) float wl[];                          // [3][C]
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x - b * blocks_per_sample;
    for (int i = threadIdx.x; i < 3 * C; i += 256) wl[i] = wmod[(size_t)b * 3 * C + i];
    __syncthreads();
    const int G8 = C / 8;
    // [r4] up to four pixel slots (32 channels) in flight per lane, the weights of a group as 16-byte LDS reads (the first version issued one
    // 16-byte load and 24 ds_read_b32 per group: 3.2 TB/s); groups in the order 0 .. G8-1 as before (same sums)
    const float4* wl4 = reinterpret_cast<const float4*>(wl);
    const int C4 = C / 4;
    for (long long pix = (long long)blk * 256 + threadIdx.x; pix < HW; pix += (long long)blocks_per_sample * 256) {
        float a0 = bias[0], a1 = bias[1], a2 = bias[2];
        const u32x4* xp = x + (size_t)b * G8 * HW + pix;
        int g = 0;
        for (; g + 4 <= G8; g += 4) {
            u32x4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = xp[(size_t)(g + u) * HW];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                unpack8(q[u], v);
                const float4 w0a = wl4[2 * (g + u)], w0b = wl4[2 * (g + u) + 1], w1a = wl4[C4 + 2 * (g + u)], w1b = wl4[C4 + 2 * (g + u) + 1];
                const float4 w2a = wl4[2 * C4 + 2 * (g + u)], w2b = wl4[2 * C4 + 2 * (g + u) + 1];
                const float w0[8] = {w0a.x, w0a.y, w0a.z, w0a.w, w0b.x, w0b.y, w0b.z, w0b.w}, w1[8] = {w1a.x, w1a.y, w1a.z, w1a.w, w1b.x, w1b.y, w1b.z, w1b.w};
                const float w2[8] = {w2a.x, w2a.y, w2a.z, w2a.w, w2b.x, w2b.y, w2b.z, w2b.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0 += v[e] * w0[e]; a1 += v[e] * w1[e]; a2 += v[e] * w2[e]; }
            }
        }
        for (; g < G8; ++g) {
            float v[8];
            unpack8(xp[(size_t)g * HW], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0 += v[e] * wl[8 * g + e]; a1 += v[e] * wl[C + 8 * g + e]; a2 += v[e] * wl[2 * C + 8 * g + e]; }
        }
        rgb[((size_t)b * 3 + 0) * HW + pix] = a0; rgb[((size_t)b * 3 + 1) * HW + pix] = a1; rgb[((size_t)b * 3 + 2) * HW + pix] = a2;
    }
}
extern "C" int H8_NAME(l2i_torgb_fwd_h8)(float* rgb, const void* x, const float* wmod, const float* bias, int B, int C, int64_t HW, void* stream) {
    if (!rgb || !x || !wmod || !bias || B <= 0 || C <= 0 || (C % 8) != 0 || C > 4096 || HW <= 0) return l2i_set_error(L2I_E_ARG, "torgb_fwd_h8: bad arguments");
    int bps = (int)((HW + 255) / 256);
    if (bps > 512) bps = 512;
    hipLaunchKernelGGL(torgb_fwd_h8_kernel, dim3((unsigned)(B * bps)), dim3(256), (size_t)3 * C * sizeof(float), (hipStream_t)stream, rgb, (const u32x4*)x, wmod, bias, C, (long long)HW, bps);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// strips per (sample, channel group) of the reducing kernels: about 4096 blocks in flight in total (16 per CU), never fewer than 2048 pixel slots per
// strip (one atomic per block and sum: the fewer strips, the fewer same-address atomics)
static int h8_strips(long long groups, long long HW) {
    static const int det = getenv("L2I_H8_DET") ? atoi(getenv("L2I_H8_DET")) : 0;
    if (det) return 1;
    long long s = (4096 + groups - 1) / groups;
    const long long cap = (HW + 2047) / 2048;
    if (s > cap) s = cap;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int)s;
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
                                                             float* __restrict__ red_dz_z, float* __restrict__ red_x_grgb, float* __restrict__ red_gin_y, int C, long long HW, int strips) {
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
    float r1[8], r2[3][8], r3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { r1[e] = 0.f; r2[0][e] = r2[1][e] = r2[2][e] = 0.f; r3[e] = 0.f; }
    const float gp = gain, gn = gain * slope, ip = 1.f / gain, in_ = 1.f / (gain * slope);
    const long long step = (long long)strips * 256;
    // two pixel slots per iteration, all loads of both issued before the arithmetic (a lane's second slot is `step` further: both coalesced)
    for (long long pix = (long long)strip * 256 + threadIdx.x; pix < HW; pix += 2 * step) {
        const long long pix2 = pix + step;
        const bool two = pix2 < HW;
        const long long p2 = two ? pix2 : pix;
        const u32x4 yq0 = y[base + pix], yq1 = y[base + p2];
        u32x4 gq0 = u32x4{0, 0, 0, 0}, gq1 = u32x4{0, 0, 0, 0};
        if (gin) { gq0 = gin[base + pix]; gq1 = gin[base + p2]; }
        float q[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (grgb) {
#pragma unroll
            for (int o = 0; o < 3; ++o) { q[0][o] = grgb[((size_t)b * 3 + o) * HW + pix]; q[1][o] = grgb[((size_t)b * 3 + o) * HW + p2]; }
        }
        float nz[2] = {0.f, 0.f};
        if (noise) { nz[0] = noise[(size_t)b * HW + pix] * noise_w; nz[1] = noise[(size_t)b * HW + p2] * noise_w; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            float yv[8], gv[8], d[8];
            unpack8(h ? yq1 : yq0, yv);
            unpack8(h ? gq1 : gq0, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float gg = gv[e] * sc[e];
                r3[e] += gv[e] * yv[e];                               // [r5] red_gin_y: both maps are in registers anyway
                gg += wr[0][e] * q[h][0] + wr[1][e] * q[h][1] + wr[2][e] * q[h][2];
                const bool pos = yv[e] > 0.f;
                d[e] = gg * (pos ? gp : gn);
                const float zpre = yv[e] * (pos ? ip : in_) - bs[e] - nz[h];
                r1[e] += d[e] * zpre;
                r2[0][e] += yv[e] * q[h][0]; r2[1][e] += yv[e] * q[h][1]; r2[2][e] += yv[e] * q[h][2];
            }
            dz[base + (h ? pix2 : pix)] = pack8(d);
        }
    }
    // block-level reduction: wave sums meet in LDS, ONE atomic per block and sum (the per-wave atomics of the first version were the
    // kernel's bound on the mid-resolution layers: 0.5 M same-address atomics per launch)
    __shared__ float part[4][40];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s0 = wave_sum(r1[e]);
        if (lane == 0) part[wv][e] = s0;
        if (gin && red_gin_y) {
            const float t = wave_sum(r3[e]);
            if (lane == 0) part[wv][32 + e] = t;
        }
        if (red_x_grgb) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float t = wave_sum(r2[o][e]);
                if (lane == 0) part[wv][8 + 3 * e + o] = t;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int e = threadIdx.x;
        atomicAdd(red_dz_z + (size_t)b * C + 8 * g + e, part[0][e] + part[1][e] + part[2][e] + part[3][e]);
    } else if (threadIdx.x < 32 && red_x_grgb) {
        const int i = threadIdx.x;                              // 8 + 3 e + o
        atomicAdd(red_x_grgb + ((size_t)b * C + 8 * g) * 3 + (i - 8), part[0][i] + part[1][i] + part[2][i] + part[3][i]);
    } else if (threadIdx.x >= 32 && threadIdx.x < 40 && gin && red_gin_y) {
        const int i = threadIdx.x;                              // 32 + e
        atomicAdd(red_gin_y + (size_t)b * C + 8 * g + (i - 32), part[0][i] + part[1][i] + part[2][i] + part[3][i]);
    }
}
extern "C" int H8_NAME(l2i_sg2_act_bwd_h8)(void* dz, const void* gin, const float* gin_scale, const float* grgb, const float* wmod_rgb, const void* y, const float* bias,
                                  const float* noise, float noise_w, float slope, float gain, float* red_dz_z, float* red_x_grgb, float* red_gin_y, int B, int C, int64_t HW, void* stream) {
    if (!dz || !y || !red_dz_z || B <= 0 || C <= 0 || (C % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd_h8: bad arguments");
    if ((grgb != nullptr) != (wmod_rgb != nullptr)) return l2i_set_error(L2I_E_ARG, "sg2_act_bwd_h8: grgb and wmod_rgb go together");
    const int strips = h8_strips(B * (C / 8), HW);
    hipLaunchKernelGGL(sg2_act_bwd_h8_kernel, dim3((unsigned)(B * (C / 8) * strips)), dim3(256), 0, (hipStream_t)stream, (u32x4*)dz, (const u32x4*)gin, gin_scale, grgb, wmod_rgb,
                       (const u32x4*)y, bias, noise, noise_w, slope, gain, red_dz_z, grgb ? red_x_grgb : nullptr, gin ? red_gin_y : nullptr, C, (long long)HW, strips);
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
    const long long step = (long long)strips * 256;
    for (long long pix = (long long)strip * 256 + threadIdx.x; pix < HW; pix += 2 * step) {
        const long long pix2 = pix + step;
        const bool two = pix2 < HW;
        const long long p2 = two ? pix2 : pix;
        const u32x4 a0 = a[base + pix], a1 = a[base + p2];
        float av[8], bv[8], cv[8], dv[8];
        unpack8(a0, av);
        unpack8(a1, cv);
        const float w2 = two ? 1.f : 0.f;
        if (bb) {
            const u32x4 b0 = bb[base + pix], b1 = bb[base + p2];
            unpack8(b0, bv);
            unpack8(b1, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] += av[e] * bv[e] + w2 * (cv[e] * dv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] += av[e] + w2 * cv[e];
        }
    }
    __shared__ float part[4][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = wave_sum(r[e]);
        if (lane == 0) part[wv][e] = t;
    }
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(out + (size_t)b * C + 8 * g + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
extern "C" int H8_NAME(l2i_dot_reduce_h8)(float* out, const void* a, const void* b, int B, int C, int64_t HW, void* stream) {
    if (!out || !a || B <= 0 || C <= 0 || (C % 8) != 0 || HW <= 0) return l2i_set_error(L2I_E_ARG, "dot_reduce_h8: bad arguments");
    const int strips = h8_strips(B * (C / 8), HW);
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
extern "C" int H8_NAME(l2i_maxpool2d_fwd_h8)(void* y, void* idx, const void* x, int64_t planes, int H, int W, int k, int s, int pad, int OH, int OW, int relu, void* stream) {
    if (!y || !idx || !x || planes <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || s <= 0 || OH <= 0 || OW <= 0) return l2i_set_error(L2I_E_ARG, "maxpool_fwd_h8: bad arguments");
    hipLaunchKernelGGL(maxpool_fwd_h8_kernel, dim3(l2i_grid_for((long long)planes * OH * OW, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (uint2*)idx, (const u32x4*)x,
                       (long long)planes, H, W, k, s, pad, OH, OW, relu);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
extern "C" int H8_NAME(l2i_maxpool2d_bwd_h8)(void* gx, const void* gy, const void* idx, const void* a, const void* b, float coef, const float* coef_dev, int64_t planes, int H, int W,
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
extern "C" int H8_NAME(l2i_sqdiff_h8)(float* sum_out, void* grad, const void* a, const void* b, int64_t slots, float coef, const float* coef_dev, void* stream) {
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
extern "C" int H8_NAME(l2i_add_zero_insert_h8)(void* y, const void* c, const void* mask, int64_t planes, int H, int W, int OH, int OW, void* stream) {
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
extern "C" int H8_NAME(l2i_modulate_planes_h8)(void* planes, const float* w32, const float* s, int B, int Cs, int CinP, int KK, int CoutP, void* stream) {
    if (!planes || !w32 || !s || B <= 0 || CinP <= 0 || (CinP % 16) != 0 || Cs > CinP || KK <= 0 || CoutP <= 0) return l2i_set_error(L2I_E_ARG, "modulate_planes_h8: bad arguments");
    if (Cs != CinP) return l2i_set_error(L2I_E_ARG, "modulate_planes_h8: the scale vector must cover the padded channel count");
    const long long sps = (long long)(CinP / 16) * KK * 2 * CoutP, total = sps * B;
    hipLaunchKernelGGL(modulate_planes_kernel, dim3(l2i_grid_for(total, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (u32x4*)planes, w32, s, Cs, sps, KK, CoutP, total);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// [r5] Every layer of a pass in ONE launch (the generator's forward folds the style into 17 plane sets, its backward the demodulation factor into 17
// more: 51 launches of 15 us per step).  `table`: nseg rows of 8 int64 on the device — {w32 offset (floats), s offset (floats: the layer's [B, Cs]
// block), output offset (16-byte slots: the layer's [B][slots per sample] block), slots per sample, KK, CoutP, Cs, first block of the segment} —
// blocks [first, next first) walk their segment grid-stride.
struct ModSeg { long long w_off, s_off, out_off, sps, KK, CoutP, Cs, first; };
__global__ __launch_bounds__(256) void modulate_planes_multi_kernel(u32x4* __restrict__ planes, const float* __restrict__ w32, const float* __restrict__ s, const ModSeg* __restrict__ table,
                                                                    int nseg, int B, int nblocks) {
    int sg = 0;
    while (sg + 1 < nseg && (long long)blockIdx.x >= table[sg + 1].first) ++sg;
    const ModSeg t = table[sg];
    const long long nb = (sg + 1 < nseg ? table[sg + 1].first : (long long)nblocks) - t.first;
    const long long total = t.sps * B;
    const float* wl = w32 + t.w_off;
    const float* sl = s + t.s_off;
    u32x4* out = planes + t.out_off;
    const int KK = (int)t.KK, CoutP = (int)t.CoutP, Cs = (int)t.Cs;
    for (long long i = ((long long)blockIdx.x - t.first) * 256 + threadIdx.x; i < total; i += nb * 256) {
        const long long b = i / t.sps, slot = i - b * t.sps;
        const long long r = slot / CoutP;                      // ((c16 * KK + tap) * 2 + half)
        const int half = (int)(r & 1), c16 = (int)((r >> 1) / KK);
        const float4 w0 = *reinterpret_cast<const float4*>(wl + slot * 8), w1 = *reinterpret_cast<const float4*>(wl + slot * 8 + 4);
        const float* sp = sl + b * Cs + 16 * c16 + 8 * half;
        float v[8] = {w0.x * sp[0], w0.y * sp[1], w0.z * sp[2], w0.w * sp[3], w1.x * sp[4], w1.y * sp[5], w1.z * sp[6], w1.w * sp[7]};
        out[i] = pack8(v);
    }
}
extern "C" int H8_NAME(l2i_modulate_planes_multi_h8)(void* planes, const float* w32, const float* s, const void* table, int nseg, int B, int nblocks, void* stream) {
    if (!planes || !w32 || !s || !table || nseg <= 0 || B <= 0 || nblocks < nseg) return l2i_set_error(L2I_E_ARG, "modulate_planes_multi_h8: bad arguments");
    hipLaunchKernelGGL(modulate_planes_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (u32x4*)planes, w32, s, (const ModSeg*)table, nseg, B, nblocks);
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
// [r6] the same with `ref` given as its sign plane (one byte per slot)
__global__ __launch_bounds__(256) void mask_mul_bits_h8_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ g, const uint8_t* __restrict__ bits, float pos, float neg, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float a[8];
        unpack8(g[i], a);
        const unsigned mb = bits[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= ((mb >> e) & 1u) ? pos : neg;
        y[i] = pack8(a);
    }
}
extern "C" int H8_NAME(l2i_mask_mul_bits_h8)(void* y, const void* g, const void* bits, float pos, float neg, int64_t slots, void* stream) {
    if (!y || !g || !bits || slots <= 0) return l2i_set_error(L2I_E_ARG, "mask_mul_bits_h8: bad arguments");
    hipLaunchKernelGGL(mask_mul_bits_h8_kernel, dim3(l2i_grid_for(slots, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)g, (const uint8_t*)bits, pos, neg,
                       (long long)slots);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
extern "C" int H8_NAME(l2i_mask_mul_h8)(void* y, const void* g, const void* ref, float pos, float neg, int64_t slots, void* stream) {
    if (!y || !g || !ref || slots <= 0) return l2i_set_error(L2I_E_ARG, "mask_mul_h8: bad arguments");
    hipLaunchKernelGGL(mask_mul_h8_kernel, dim3(l2i_grid_for(slots, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (u32x4*)y, (const u32x4*)g, (const u32x4*)ref, pos, neg, (long long)slots);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
}  // namespace H8_NS
