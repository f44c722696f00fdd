// l2i_conv16.hip — stride-1 2-D correlation on the bf16 matrix cores with a 3-term operand split (gfx950).
//
// OPT-IN path (conv.PRECISION = 'bf16x3'); the default everywhere is the exact-fp32 kernel of l2i_conv.hip.
// Every fp32 operand x is split as x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product a*b is evaluated as
// ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped al*bl term is <= 2^-18 |ab|, the
// split itself is accurate to 2^-17 |x|): fp32-class results (far inside the path's rtol 1e-3 parity bar) at 3/16 of
// the matrix-pipe time of v_mfma_f32_32x32x2_f32.  fp16 is not used: gradients of the path span ~1e-9..1e2 and would
// flush; bf16 keeps fp32's exponent.
//
// Same implicit-GEMM mapping and epilogue as l2i_conv.hip (M = out-channels, N = 32-pixel row segments, accumulators in
// the 32x32 C/D layout).  K runs over 16-channel groups: a lane's MFMA operand is 8 consecutive channels of one pixel /
// one output channel, so the LDS images are channel-contiguous: input [pixel][2 x 8 ch] and weights [tap][cout][2 x 8 ch],
// each as a hi plane and a lo plane; fragments are single ds_read_b128.  The fp32 -> (hi, lo) split of the activations
// happens on the register -> LDS commit of the staging pipeline (with the style/demod scale and the activation-gradient
// mask); weights are split once on the host.  Large maps only (OW >= 32, Cin % 16 == 0, stride 1); everything else
// stays on the fp32 kernel.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Conv16Launch {
    int tiles_x, tiles_y, mblocks;
    int IH, IW, nitems;                // staged pixels, items = pixels * 2 (8-channel halves)
    unsigned magic_iw;
    int w_vec;                         // uint4 per weight plane and chunk: KK*BM*2
    int off_in_lo, off_w_hi, off_w_lo; // uint4 offsets of the planes inside LDS
    int vec_epi;
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 8 fp32 -> 8 bf16 hi (round to nearest even) and 8 bf16 lo = bf16(x - hi)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = cvt_pk_bf16(v[2 * q], v[2 * q + 1]);
        const float h0 = __uint_as_float(h[q] << 16), h1 = __uint_as_float(h[q] & 0xffff0000u);
        l[q] = cvt_pk_bf16(v[2 * q] - h0, v[2 * q + 1] - h1);
    }
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}

template <int WM, int WN, bool MASK>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_kernel(const l2i_conv_params p, const Conv16Launch L) {
    constexpr int BM = WM * 32;
    constexpr int NS = (WN <= 2) ? 3 : 5;                 // input item slots per thread per chunk
    constexpr int NWS = (BM == 64) ? 5 : 3;               // weight uint4 slots per thread per plane per chunk (3x3: KK*BM*2/256)
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4* in_hi = smem4;
    u32x4* in_lo = smem4 + L.off_in_lo;
    u32x4* w_hi = smem4 + L.off_w_hi;
    u32x4* w_lo = smem4 + L.off_w_lo;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int KK = p.KH * p.KW;
    constexpr int TH = 4 * WN;
    int bid = blockIdx.x;
    const int mblk = bid % L.mblocks; bid /= L.mblocks;
    const int tx = bid % L.tiles_x; bid /= L.tiles_x;
    const int ty = bid % L.tiles_y; bid /= L.tiles_y;
    const int b = bid;
    const int m0 = mblk * BM;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;

    int pixbase[WN];                                       // staged-pixel index of this lane's pixel, tap (0,0)
#pragma unroll
    for (int n = 0; n < WN; ++n) pixbase[n] = (wave * WN + n) * L.IW + j;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- staging state: thread = (pixel, 8-channel half) items; the 8 channels of an item share one VGPR offset and
    //      differ by an SGPR offset, so issuing a chunk costs no VALU at all ----
    const size_t plane_x = (size_t)p.H * p.W;
    const unsigned in_bytes = (unsigned)((size_t)p.Cin * plane_x * sizeof(float));
    const size_t smp_off = (size_t)b * p.Cin * plane_x;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + smp_off), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + smp_off), 0, in_bytes, 0x00020000);
    const unsigned wpl_bytes = (unsigned)((size_t)(p.Cin / 16) * KK * p.CoutP * 32);           // one weight plane
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, wpl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, wpl_bytes, 0x00020000);
    const unsigned sc_bytes = (unsigned)((size_t)p.Cin * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.in_scale ? p.in_scale : p.x) + (size_t)b * p.Cin), 0, p.in_scale ? sc_bytes : 0u, 0x00020000);

    const int kg = tid & 1;                                // this thread always stages the same 8-channel half
    unsigned voff[NS];
    float xin[NS][8];
    float xmk[MASK ? NS : 1][MASK ? 8 : 1];
    float scl[8];
    u32x4 rwh[NWS], rwl[NWS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int it = tid + u * 256;
        voff[u] = in_bytes;
        if (it < L.nitems) {
            const unsigned pix = (unsigned)it >> 1;
            const unsigned iy = __umulhi(pix, L.magic_iw), ix = pix - iy * L.IW;
            const int gy = iy0 + (int)iy, gx = ix0 + (int)ix;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                voff[u] = (unsigned)(((size_t)(kg * 8) * plane_x + (size_t)gy * p.W + gx) * sizeof(float));
        }
    }
    // weights: plane layout [chunk][tap][CoutP][2] uint4; the block's rows (tap, m0..m0+BM) are 2*BM consecutive uint4
    const int wrow = tid / (2 * BM), wcol = tid - wrow * (2 * BM);            // tap row / uint4 inside the row for slot 0
    constexpr int ROWS_PER_SLOT = 256 / (2 * BM);                             // BM=64: 2 taps per slot, BM=32: 4
    const unsigned wvoff = (unsigned)((((size_t)wrow * p.CoutP + m0) * 2 + wcol) * 16);
    const unsigned wstep = (unsigned)((size_t)ROWS_PER_SLOT * p.CoutP * 32);
    const unsigned wchunk = (unsigned)((size_t)KK * p.CoutP * 32);

    auto issue = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned so = (unsigned)((size_t)(c0 + e) * plane_x * sizeof(float));
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                xin[u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, voff[u], so, 0));
                if constexpr (MASK) xmk[u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_m, voff[u], so, 0));
            }
            scl[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_s, (unsigned)((kg * 8 + e) * sizeof(float)), (unsigned)(c0 * sizeof(float)), 0));
        }
        const unsigned sw = (unsigned)(c0 / 16) * wchunk;
#pragma unroll
        for (int u = 0; u < NWS; ++u) {
            rwh[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_wh, wvoff, sw + u * wstep, 0);
            rwl[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_wl, wvoff, sw + u * wstep, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int it = tid + u * 256;
            if (it < L.nitems) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = xin[u][e];
                    if constexpr (MASK) t *= (xmk[u][e] > 0.f) ? p.mask_pos : p.mask_neg;
                    if (p.in_scale) t *= scl[e];
                    v[e] = t;
                }
                u32x4 hi, lo;
                split8(v, hi, lo);
                in_hi[it] = hi;
                in_lo[it] = lo;
            }
        }
#pragma unroll
        for (int u = 0; u < NWS; ++u) {
            const int idx = tid + u * 256;
            if (idx < L.w_vec) { w_hi[idx] = rwh[u]; w_lo[idx] = rwl[u]; }
        }
    };

    issue(0);
    for (int c0 = 0; c0 < p.Cin; c0 += 16) {
        commit();
        __syncthreads();
        if (c0 + 16 < p.Cin) issue(c0 + 16);
        for (int ky = 0; ky < p.KH; ++ky) {
            for (int kx = 0; kx < p.KW; ++kx) {
                const int tap = ky * p.KW + kx;
                bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    const int idx = ((tap * BM + m * 32 + j) << 1) + half;
                    ah[m] = __builtin_bit_cast(bf16x8, w_hi[idx]);
                    al[m] = __builtin_bit_cast(bf16x8, w_lo[idx]);
                }
                const int toff = ky * L.IW + kx;
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    const int idx = ((pixbase[n] + toff) << 1) + half;
                    bh[n] = __builtin_bit_cast(bf16x8, in_hi[idx]);
                    bl[n] = __builtin_bit_cast(bf16x8, in_lo[idx]);
                }
                // three passes over the WM x WN tiles so that dependent accumulations are WM*WN MFMAs apart
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (same fusions as l2i_conv.hip): per-wave LDS transpose -> 16-byte global accesses, or scalar ----
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    float* smemf = reinterpret_cast<float*>(smem4);
    if (L.vec_epi) {
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
                        v.x += nz.x; v.y += nz.y; v.z += nz.z; v.w += nz.w;
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
                        v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain;
                        if (p.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(p.y + oidx) = v;
                    }
                }
            }
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

static unsigned magic16(unsigned d) { return d == 1 ? 0u : (unsigned)((0x100000000ULL + d - 1) / d); }

template <int WM, int WN>
static int launch16(const l2i_conv_params& p, hipStream_t st) {
    constexpr int BM = WM * 32, TH = 4 * WN;
    constexpr int NS = (WN <= 2) ? 3 : 5, NWS = (BM == 64) ? 5 : 3;
    const int KK = p.KH * p.KW;
    Conv16Launch L;
    L.tiles_x = (p.OW + 31) / 32;
    L.tiles_y = (p.OH + TH - 1) / TH;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    L.IH = TH + p.KH - 1;
    L.IW = 32 + p.KW - 1;
    L.nitems = L.IH * L.IW * 2;
    L.magic_iw = magic16((unsigned)L.IW);
    L.w_vec = KK * BM * 2;
    if (L.nitems > NS * 256 || L.w_vec > NWS * 256 || L.IW < 2) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_bf16x3: tile does not fit the staging slots");
    L.off_in_lo = L.nitems;
    L.off_w_hi = 2 * L.nitems;
    L.off_w_lo = 2 * L.nitems + L.w_vec;
    size_t lds = (size_t)(2 * L.nitems + 2 * L.w_vec) * 16;
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    L.vec_epi = (p.ox_step == 1 && p.oy_step == 1 && (p.OWf % 4) == 0 && (p.OW % 4) == 0 && (p.ox_off % 4) == 0 &&
                 al16(p.y) && al16(p.residual) && al16(p.res_mask) && al16(p.res_sub) && al16(p.out_mask) && al16(p.noise)) ? 1 : 0;
    if (L.vec_epi && lds < 4 * 32 * 64 * sizeof(float)) lds = 4 * 32 * 64 * sizeof(float);
    if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_kernel<WM, WN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_kernel<WM, WN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
    }
    const long grid = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (grid <= 0 || grid > 0x7fffffffL) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: grid too large");
    if (p.in_mask) hipLaunchKernelGGL((conv_bf16x3_kernel<WM, WN, true>), dim3((unsigned)grid), dim3(256), lds, st, p, L);
    else hipLaunchKernelGGL((conv_bf16x3_kernel<WM, WN, false>), dim3((unsigned)grid), dim3(256), lds, st, p, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_conv2d_bf16x3_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w_hi || !p.w_lo || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: null tensor");
    if (p.stride != 1 || (p.Cin % 16) != 0 || p.OW < 32 || p.KH < 1 || p.KW < 1 || p.KH > 3 || p.KW > 3)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_bf16x3: needs stride 1, Cin % 16 == 0, OW >= 32, kernel <= 3x3");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: CoutP must be Cout rounded up to 32");
    if ((size_t)p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_bf16x3: sample >= 4 GiB");
    if ((((uintptr_t)p.w_hi) | ((uintptr_t)p.w_lo)) % 16) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: weight planes must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (p.CoutP % 64 == 0) return launch16<2, 2>(p, st);
    return launch16<1, 2>(p, st);
}
