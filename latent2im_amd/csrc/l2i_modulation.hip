// l2i_modulation.hip — every style-dependent vector of a StyleGAN2 generator pass in a handful of launches.
//
// Per styled conv the reference evaluates  s = EqualLinear(w)  (networks.py:148-156, 231-233), the demodulation factor
// rsqrt(sum (scale*W*s)^2 + 1e-8) (networks.py:236-239) and, on the way back, the gradients of both with respect to the latent.
// These are [B,512] x [512,C] products and [B,C] vector ops: < 0.1 % of the step's FLOPs but, issued layer by layer through
// rocBLAS / torch, ~400 launches per step (35 modulations x (addmm, square, mm, add, rsqrt, ...) x forward / backward).  They depend
// on the latent (and, backward, on per-layer reductions) only, so one "segmented mat-vec" kernel evaluates all layers of a kind in ONE
// launch:
//     out[b, r] = epi( sum_parts sum_k pre(in)[b, k] * W[k, r] )            r < rows of the segment, b < B
// A segment = one layer (or, for the latent gradient, one latent index fed by up to two layers).  A block = 64 rows of one segment
// (lane = row: W[k, r] is read as coalesced 256-byte rows; four waves split the contraction), the B x K input slice staged in LDS (read back
// as broadcasts), B accumulators per lane.  Segment tables are built once per network on the host (latent2im_amd/generator.py:_ModPlan) and live on
// the device; offsets that scale with the batch are stored as (constant, per-sample) pairs so the tables do not depend on B.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

namespace {
constexpr int BT = 8;          // samples per accumulator pass
constexpr int KMAX = 512;      // longest contraction (style_dim, or the widest layer)
}

__global__ __launch_bounds__(256) void segmv_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ in2,
                                                    const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ e1,
                                                    const float* __restrict__ e2, float* __restrict__ wmod, const float* __restrict__ wrgb,
                                                    const l2i_segmv_seg* __restrict__ segs, const int32_t* __restrict__ block_seg, int B) {
    // Four waves per block split the contraction (wave v takes a quarter of k) and meet in LDS; each wave keeps 16 weight
    // rows in flight per step.  The first version (one wave, four rows per step) was a chain of 128 dependent L2 round trips: 95 us per launch.
    __shared__ __attribute__((aligned(16))) float xin[BT][KMAX];
    __shared__ float red[3][BT][64];
    const l2i_segmv_seg sg = segs[block_seg[2 * blockIdx.x]];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = block_seg[2 * blockIdx.x + 1] * 64 + lane;
    const bool rok = r < sg.rows;
    for (int b0 = 0; b0 < B; b0 += BT) {
        const int nb = (B - b0) < BT ? (B - b0) : BT;
        float acc[BT];
#pragma unroll
        for (int q = 0; q < BT; ++q) acc[q] = 0.f;
        for (int pi = 0; pi < sg.nparts; ++pi) {
            const l2i_segmv_part pt = sg.part[pi];
            const long in_base = (long)pt.in_off_c + (long)B * pt.in_off_b;
            __syncthreads();                                    // the previous part's / pass's reads of xin are done
            for (int i = tid; i < nb * pt.K; i += 256) {
                const int bq = i / pt.K, k = i - bq * pt.K;
                const long idx = in_base + (long)(b0 + bq) * pt.in_bstride + k;
                float v;
                if (pt.pre == 3) {                              // sum_o red[b, k, o] * Wrgb[o, k]  (ToRGB: d s_rgb)
                    const float* rp = in2 + in_base + ((long)(b0 + bq) * pt.K + k) * 3;      // (the [B, C, 3] reductions travel in the in2 slot)
                    const float* wp = wrgb + pt.aux_off + k;
                    v = rp[0] * wp[0] + rp[1] * wp[pt.K] + rp[2] * wp[2 * pt.K];
                } else {
                    v = in[idx];
                    if (pt.pre == 1) v = v * v;
                    else if (pt.pre == 2) { const float d = in2[idx]; v = v * d * d; }
                }
                xin[bq][k] = v;
            }
            __syncthreads();
            if (rok) {
                const float* wp = w + pt.w_off + r;
                const int kq = (((pt.K >> 2) + 3) >> 2) << 2;                 // a quarter of K, rounded up to whole 4-steps (K % 4 == 0)
                const int k0 = wv * kq < pt.K ? wv * kq : pt.K, k1 = k0 + kq < pt.K ? k0 + kq : pt.K;
                int k = k0;
                for (; k + 16 <= k1; k += 16) {
                    float wr[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) wr[i] = wp[(long)(k + i) * pt.w_pitch];
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
#pragma unroll
                        for (int q = 0; q < BT; ++q) {
                            const float4 x = *reinterpret_cast<const float4*>(&xin[q][k + i]);      // same address in every lane: LDS broadcast
                            acc[q] = __builtin_fmaf(x.w, wr[i + 3], __builtin_fmaf(x.z, wr[i + 2], __builtin_fmaf(x.y, wr[i + 1], __builtin_fmaf(x.x, wr[i], acc[q]))));
                        }
                    }
                }
                for (; k < k1; k += 4) {
                    const float w0 = wp[(long)k * pt.w_pitch], w1 = wp[(long)(k + 1) * pt.w_pitch];
                    const float w2 = wp[(long)(k + 2) * pt.w_pitch], w3 = wp[(long)(k + 3) * pt.w_pitch];
#pragma unroll
                    for (int q = 0; q < BT; ++q) {
                        const float4 x = *reinterpret_cast<const float4*>(&xin[q][k]);
                        acc[q] = __builtin_fmaf(x.w, w3, __builtin_fmaf(x.z, w2, __builtin_fmaf(x.y, w1, __builtin_fmaf(x.x, w0, acc[q]))));
                    }
                }
            }
        }
        // waves 1-3 hand their partial sums to wave 0
        __syncthreads();
        if (wv > 0) {
#pragma unroll
            for (int q = 0; q < BT; ++q) red[wv - 1][q][lane] = acc[q];
        }
        __syncthreads();
        if (wv == 0 && rok) {
            const long out_base = (long)sg.out_off_c + (long)B * sg.out_off_b + r;
#pragma unroll
            for (int q = 0; q < BT; ++q) {
                if (q < nb) {
                    const int b = b0 + q;
                    float v = (acc[q] + red[0][q][lane]) + (red[1][q][lane] + red[2][q][lane]);
                    if (sg.epi == 0) v += bias[sg.bias_off + r];
                    else if (sg.epi == 1) v = rsqrtf(v + 1e-8f);
                    else if (sg.epi == 2) {
                        const long ei = (long)sg.e_off_c + (long)B * sg.e_off_b + (long)b * sg.e_bstride + r;
                        v = e1[ei] - e2[ei] * v;
                    }
                    out[out_base + (long)b * sg.out_bstride] = v;
                    if (sg.epi == 0 && sg.rgb_off_b >= 0) {     // ToRGB: wmod[b, o, c] = (scale * W[o, c]) * s[b, c]   (networks.py:346-351, no demodulation)
                        float* wm = wmod + (long)B * sg.rgb_off_b + (long)b * 3 * sg.rows + r;
                        const float* wr2 = wrgb + sg.rgb_w_off + r;
                        wm[0] = wr2[0] * v; wm[sg.rows] = wr2[sg.rows] * v; wm[2 * sg.rows] = wr2[2 * sg.rows] * v;
                    }
                }
            }
        }
    }
}

extern "C" int l2i_segmented_matvec_f32(float* out, const float* in, const float* in2, const float* w, const float* bias, const float* e1, const float* e2,
                                        float* wmod, const float* wrgb, const l2i_segmv_seg* segs, const int32_t* block_seg, int nblocks, int B, void* stream) {
    if (!out || !in || !w || !segs || !block_seg) return l2i_set_error(L2I_E_ARG, "segmented_matvec: null pointer");
    if (nblocks <= 0 || B <= 0) return l2i_set_error(L2I_E_ARG, "segmented_matvec: non-positive size");
    hipLaunchKernelGGL(segmv_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, out, in, in2, w, bias, e1, e2, wmod, wrgb, segs, block_seg, B);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
