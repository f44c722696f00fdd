// l2i_epilogue.h — the fused epilogue of the 32x32-accumulator conv kernels (shared by the bf16-split kernels; the fp32 kernels carry
// the same code inline).  Accumulator layout: wave w of the block owns output rows oy0 + w*WN + n (n < WN), 32 columns from ox0;
// acc[m][n][r] = channel m0 + 32 m + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), pixel lane & 31.
//   epi(a) = act( a*out_scale[b,co] * (out_mask > 0) + noise*noise_w + bias[co] + R * (res_mask > 0) ) * out_gain (+ y if accumulate)
// (include/l2i.h).  vec = true: per-wave LDS transpose (8 KiB strip per wave at `smemf`) -> 16-byte global accesses.
#ifndef L2I_EPILOGUE_H
#define L2I_EPILOGUE_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline bool l2i_epilogue_vec_ok(const l2i_conv_params& p) {
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    return p.ox_step == 1 && p.oy_step == 1 && (p.OWf % 4) == 0 && (p.OW % 4) == 0 && (p.ox_off % 4) == 0 && al16(p.y) && al16(p.residual) &&
           al16(p.res_mask) && al16(p.res_sub) && al16(p.out_mask) && al16(p.noise);
}

// ---- [r5] 16-bit elements of the h8 layout inside fp32 translation units (`f16`: IEEE fp16, else bf16): l2i_convt_small.hip (in_h8), l2i_img_h8.hip ----
__device__ __forceinline__ unsigned l2i_cvt_pk_h8(float lo, float hi, bool f16) {
    unsigned r;
    if (f16) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float l2i_h8_lo(unsigned u, bool f16) {
    float r;
    if (f16) asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(u));
    else r = __uint_as_float(u << 16);
    return r;
}
__device__ __forceinline__ float l2i_h8_hi(unsigned u, bool f16) {
    float r;
    if (f16) asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(u));
    else r = __uint_as_float(u & 0xffff0000u);
    return r;
}
typedef unsigned int l2i_u32x4 __attribute__((ext_vector_type(4)));
template <int WM, int WN>
__device__ __forceinline__ void l2i_epilogue_32x32(const l2i_conv_params& p, f32x16 (&acc)[WM][WN], float* smemf, int b, int m0, int oy0, int ox0, bool vec) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    if (vec) {
        float sq = 0.f;                                        // sq_ref: ContentLoss value of a VGG tap, summed while y is in registers (l2i.h)
        float* reg = smemf + wave * (32 * 64);
        const int ch_l = lane >> 4;
        const int px = (lane & 15) * 4;
#pragma unroll
        for (int n0 = 0; n0 < WN; n0 += 2) {
            const int nn = px >> 5;
            const int oy = oy0 + wave * WN + n0 + nn;
            const int ox = ox0 + (px & 31);
            const bool pok = (n0 + nn < WN) && (oy < p.OH) && (ox < p.OW);
            const size_t poff = (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pok && p.noise) {
                nz = *reinterpret_cast<const float4*>(p.noise + (size_t)b * plane_o + poff);
                nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
            }
            const float* osc = (pok && p.out_scale) ? p.out_scale + (size_t)b * p.Cout : nullptr;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (n0 + q < WN) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            reg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + q * 32 + j] = acc[m][(n0 + q) < WN ? (n0 + q) : 0][r];
                    }
                }
#pragma unroll 2
                for (int i = 0; i < 8; ++i) {
                    const int ch = i * 4 + ch_l;
                    const int co = m0 + m * 32 + ch;
                    const float4 t = *reinterpret_cast<const float4*>(&reg[ch * 64 + px]);
                    if (pok && co < p.Cout) {
                        float4 v = t;
                        if (osc) { const float sc = osc[co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                        const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + poff;
                        if (p.out_mask) {
                            const float4 mk = *reinterpret_cast<const float4*>(p.out_mask + oidx);
                            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                        }
                        if (p.noise) { v.x += nz.x; v.y += nz.y; v.z += nz.z; v.w += nz.w; }
                        if (p.bias) { const float bv = p.bias[co]; v.x += bv; v.y += bv; v.z += bv; v.w += bv; }
                        if (p.residual) {
                            float4 rv = *reinterpret_cast<const float4*>(p.residual + oidx);
                            if (p.res_sub) {                                   // residual term = res_coef * (residual - res_sub)
                                const float4 sb = *reinterpret_cast<const float4*>(p.res_sub + oidx);
                                const float rc = p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f);
                                rv.x = rc * (rv.x - sb.x); rv.y = rc * (rv.y - sb.y); rv.z = rc * (rv.z - sb.z); rv.w = rc * (rv.w - sb.w);
                            }
                            if (p.res_mask) {
                                const float4 mk = *reinterpret_cast<const float4*>(p.res_mask + oidx);
                                rv.x = mk.x > 0.f ? rv.x : 0.f; rv.y = mk.y > 0.f ? rv.y : 0.f; rv.z = mk.z > 0.f ? rv.z : 0.f; rv.w = mk.w > 0.f ? rv.w : 0.f;
                            }
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                        if (p.act == L2I_ACT_LRELU) {
                            v.x = (v.x > 0.f ? v.x : v.x * p.act_slope) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.act_slope) * p.act_gain;
                            v.z = (v.z > 0.f ? v.z : v.z * p.act_slope) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.act_slope) * p.act_gain;
                        } else if (p.act == L2I_ACT_RELU) {
                            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                        }
                        if (p.out_gain != 1.f) { v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain; }
                        if (p.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(p.y + oidx) = v;
                        if (p.sq_ref) {
                            const float4 rf = *reinterpret_cast<const float4*>(p.sq_ref + oidx);
                            const float d0 = v.x - rf.x, d1 = v.y - rf.y, d2 = v.z - rf.z, d3 = v.w - rf.w;
                            sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                        }
                    }
                }
            }
        }
        if (p.sq_ref) {                                        // one atomic per block into L2I_SQ_SLOTS slots
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
            __syncthreads();                                   // every wave is done with its transpose strip
            if (lane == 0) smemf[wave] = sq;
            __syncthreads();
            if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (smemf[0] + smemf[1]) + (smemf[2] + smemf[3]));
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int oy = oy0 + wave * WN + n, ox = ox0 + j;
        const bool pok = (oy < p.OH) && (ox < p.OW);
        const size_t poff = (size_t)(oy * p.oy_step + p.oy_off) * p.OWf + ox * p.ox_step + p.ox_off;
        float nz = 0.f;
        if (pok && p.noise) nz = p.noise[(size_t)b * plane_o + poff] * p.noise_w;
        const int co_lane = m0 + 4 * half;
        const size_t lane_base = ((size_t)b * p.Cout + co_lane) * plane_o + poff;
        const float* osc = p.out_scale ? p.out_scale + (size_t)b * p.Cout : nullptr;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cofs = m * 32 + (r & 3) + 8 * (r >> 2);
                const int co = co_lane + cofs;
                if (pok && co < p.Cout) {
                    float v = acc[m][n][r];
                    if (osc) v *= osc[co];
                    const size_t oidx = lane_base + (size_t)cofs * plane_o;
                    if (p.out_mask) v = (p.out_mask[oidx] > 0.f) ? v : 0.f;
                    v += nz;
                    if (p.bias) v += p.bias[co];
                    if (p.residual) {
                        float rv = p.residual[oidx];
                        if (p.res_sub) rv = p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) * (rv - p.res_sub[oidx]);
                        if (p.res_mask) rv = (p.res_mask[oidx] > 0.f) ? rv : 0.f;
                        v += rv;
                    }
                    if (p.act == L2I_ACT_LRELU) v = (v > 0.f ? v : v * p.act_slope) * p.act_gain;
                    else if (p.act == L2I_ACT_RELU) v = v > 0.f ? v : 0.f;
                    v *= p.out_gain;
                    if (p.accumulate) v += p.y[oidx];
                    p.y[oidx] = v;
                }
            }
        }
    }
}
#endif
