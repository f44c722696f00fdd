// l2i_cin3.hip — 3x3 stride-1 pad-1 convolution of an image with <= 3 channels (VGG-19 conv1_1 on the [-1,1] RGB image,
// transform_base.py:426-454) as ONE short contraction over the 27 (channel, tap) pairs (gfx950, fp32 MFMA).
//
// The generic implicit-GEMM kernel walks (channel pair, tap): with 3 input channels that is 2 chunks x 9 taps = 18 MFMA
// steps of which a quarter multiply zeros, each with its own LDS reads and address arithmetic, and the layer ends up
// 2.7x above its output-write time.  Here K = 27 (padded to 28) is walked in 14 `v_mfma_f32_32x32x2_f32` steps: the lane
// half picks k = 2s or 2s + 1, the B operand is the im2col value x[c(k)][y + ky(k) - 1][x + kx(k) - 1] read from an LDS
// halo tile at a per-lane offset computed once, and the A operands (28 x 64 weights) stay in registers for the whole
// block.  Output through a per-wave LDS transpose so that global stores are 16 bytes per lane (the layer is bound by its
// 64-channel fp32 output write).  Inside l2i_conv2d_f32; fuses bias, activation, out_gain only.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace c3 {
constexpr int TH = 8, TW = 64, IH = TH + 2, IW = TW + 2, IWp = 67, PLANE = IH * IWp;      // block tile: 8 rows x 64 columns
constexpr int NK = 14;                                                                     // K steps of 2
constexpr int TILE_F = (3 * PLANE + 3) & ~3;
constexpr int STRIP_F = 64 * 32;                                                           // per wave: [64 channels][32 pixels]
constexpr int LDS_FLOATS = TILE_F + 4 * STRIP_F;
}

__global__ __launch_bounds__(256, 3) void conv_cin3_kernel(const l2i_conv_params p, int tiles_x, int tiles_y, int mblocks) {
    using namespace c3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    float* strip = smem + TILE_F + wave * STRIP_F;
    int bid = blockIdx.x;
    const int mblk = bid % mblocks; bid /= mblocks;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid;
    const int m0 = mblk * 64;
    const int oy0 = ty * TH, ox0 = tx * TW;

    // A operands: w[k][m0 + blk*32 + j] for k = 2s + half (packed [Cin][9][CoutP] is linear in k); rows past 9*Cin are zero
    float wa[NK][2];
    const int kmax = 9 * p.Cin;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int k = 2 * s + half;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) wa[s][blk] = (k < kmax && m0 + blk * 32 + j < p.CoutP) ? p.w[(size_t)k * p.CoutP + m0 + blk * 32 + j] : 0.f;
    }
    // B operand offsets inside the halo tile
    int koff[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int k = 2 * s + half;
        const int c = k / 9, t = k - 9 * c;
        koff[s] = (k < kmax) ? c * PLANE + (t / 3) * IWp + (t % 3) : 0;
    }

    // halo tile: 3 channels x 10 rows x 66 columns, zero outside the image (= the padding)
    const size_t plane_x = (size_t)p.H * p.W;
    // [r5] every load of the thread is issued before the first LDS write: the loop form (load, store, next element) exposed one memory round trip
    // per element — eight per block, three blocks per CU
    constexpr int NE = (3 * IH * IW + 255) / 256;
    float stg[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        const int c = e / (IH * IW), r2 = e - c * (IH * IW);
        const int iy = r2 / IW, ix = r2 - iy * IW;
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        stg[i] = (e < 3 * IH * IW && c < p.Cin && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? p.x[((size_t)b * p.Cin + c) * plane_x + (size_t)gy * p.W + gx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        const int c = e / (IH * IW), r2 = e - c * (IH * IW);
        const int iy = r2 / IW, ix = r2 - iy * IW;
        if (e < 3 * IH * IW) tile[c * PLANE + iy * IWp + ix] = stg[i];
    }
    __syncthreads();

    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const int q8 = lane & 7, chl = lane >> 3;                  // wide pass: lane = (channel within 8, 4-pixel group)
    // a wave walks 4 strips of 32 pixels: rows 2*wave, 2*wave+1, column halves 0 / 1
    float sq = 0.f;
#pragma unroll 1
    for (int st = 0; st < 4; ++st) {
        const int row = 2 * wave + (st >> 1), cx = (st & 1) * 32;
        const float* bp = tile + row * IWp + cx + j;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const float bv = bp[koff[s]];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s][0], bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s][1], bv, acc1, 0, 0, 0);
        }
        // accumulators (lane = pixel, register = channel) -> strip [64 channels][32 pixels]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
            strip[ch * 32 + j] = acc0[r];
            strip[(32 + ch) * 32 + j] = acc1[r];
        }
        const int oy = oy0 + row, ox = ox0 + cx + q8 * 4;
        const bool pok = oy < p.OH && ox < p.OW;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ch = it * 8 + chl, co = m0 + ch;
            float4 v = *reinterpret_cast<const float4*>(&strip[ch * 32 + q8 * 4]);
            if (pok && co < p.Cout) {
                const float bv = p.bias ? p.bias[co] : 0.f;
                v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                if (p.act == L2I_ACT_LRELU) {
                    v.x = (v.x > 0.f ? v.x : v.x * p.act_slope) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.act_slope) * p.act_gain;
                    v.z = (v.z > 0.f ? v.z : v.z * p.act_slope) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.act_slope) * p.act_gain;
                } else if (p.act == L2I_ACT_RELU) {
                    v.x = __builtin_fmaxf(v.x, 0.f); v.y = __builtin_fmaxf(v.y, 0.f); v.z = __builtin_fmaxf(v.z, 0.f); v.w = __builtin_fmaxf(v.w, 0.f);
                }
                v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain;
                const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + (size_t)oy * p.OWf + ox;
                *reinterpret_cast<float4*>(p.y + oidx) = v;
                if (p.sq_ref) {                                    // ContentLoss value of VGG conv_1 (see l2i.h: sq_ref / sq_out)
                    const float4 rf = *reinterpret_cast<const float4*>(p.sq_ref + oidx);
                    const float d0 = v.x - rf.x, d1 = v.y - rf.y, d2 = v.z - rf.z, d3 = v.w - rf.w;
                    sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
        }
    }
    if (p.sq_ref) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        __syncthreads();                                           // the staged tile is no longer read
        if (lane == 0) tile[wave] = sq;
        __syncthreads();
        if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (tile[0] + tile[1]) + (tile[2] + tile[3]));
    }
}

bool l2i_cin3_eligible(const l2i_conv_params& p) {
    return p.Cin <= 3 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_y == 1 && p.pad_x == 1 && p.oy_step == 1 && p.ox_step == 1 &&
           p.oy_off == 0 && p.ox_off == 0 && p.OH == p.H && p.OW == p.W && p.OHf == p.OH && p.OWf == p.OW && (p.OW % 4) == 0 &&
           !p.in_scale && !p.in_mask && !p.out_scale && !p.noise && !p.residual && !p.res_mask && !p.out_mask && !p.accumulate &&
           p.ksplit <= 1 && p.Cout > 4 && (p.CoutP % 32) == 0 && (((uintptr_t)p.y) % 16) == 0;
}

int l2i_launch_cin3(const l2i_conv_params& p, hipStream_t st) {
    const int tiles_x = (p.OW + c3::TW - 1) / c3::TW, tiles_y = (p.OH + c3::TH - 1) / c3::TH, mblocks = (p.CoutP + 63) / 64;
    const long total = (long)p.B * tiles_y * tiles_x * mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d(cin<=3): grid too large");
    hipLaunchKernelGGL(conv_cin3_kernel, dim3((unsigned)total), dim3(256), c3::LDS_FLOATS * sizeof(float), st, p, tiles_x, tiles_y, mblocks);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
