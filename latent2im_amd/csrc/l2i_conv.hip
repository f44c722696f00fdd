// l2i_conv.hip — implicit-GEMM 2-D correlation on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// 157 TFLOP/s dense peak on MI355X).  One kernel serves every dense contraction of the walk-training path:
// StyleGAN2 modulated 3x3 convs in the activation-modulated form (in_scale = style, out_scale = demod; shared
// weights, no per-sample weight materialisation — reference networks.py:231-272), the four phases of the stride-2
// transposed conv of the up layers, discriminator / ResNet-50 / VGG-19 convs with folded BN + activation epilogue,
// and all their input-gradients (same kernel, transposed/flipped weight packs).
//
// Mapping (CDNA4, wave64):  GEMM-M = output channels (MFMA A operand, from the packed weights),
//                           GEMM-N = output pixels  (MFMA B operand, from an LDS-staged input halo tile),
//                           GEMM-K = (input channel, tap).
// D[i][j]: lane l holds pixel j = l&31 and 16 output channels -> each accumulator register is 32 consecutive
// pixels of one channel = a 128-byte coalesced NCHW row segment.  The two lane halves of the K=2 MFMA take
// channels c and c+CK/2 of the staged chunk, so all LDS fragment addresses are "lane base + wave-uniform offset".
//
// Block = 256 threads = 4 waves, all along N: block tile = (WM*32 channels) x (4*WN*32 pixels); the pixel tile is
// TH rows x TW columns with TW = min(32, pow2(OW)).  Per K chunk: input tile [CK][IH][IWp] + weights [CK][KH*KW][BM]
// staged in LDS (<= 48 KiB so 2-3 blocks share a CU and hide each other's staging).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvLaunch {
    int tw_log2;      // log2(TW)
    int tiles_x, tiles_y, mblocks;
    int CK;           // channels per LDS chunk (even)
    int IH, IW, IWp;  // staged input tile
    int plane;        // IH*IWp
};

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const l2i_conv_params p, const ConvLaunch L) {
    constexpr int BM = WM * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_in = smem;                               // [CK][IH][IWp]
    float* lds_w = smem + L.CK * L.plane;               // [CK][KK][BM]   (offset kept 16-B aligned by the host)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int j = lane & 31;
    const int KK = p.KH * p.KW;
    const int TW = 1 << L.tw_log2;
    const int RPT = 32 >> L.tw_log2;                    // rows per 32-pixel N tile
    const int TH = 4 * WN * RPT;

    // ---- block -> (batch, tile, channel block) ----
    int bid = blockIdx.x;
    const int mblk = bid % L.mblocks; bid /= L.mblocks;
    const int tx = bid % L.tiles_x; bid /= L.tiles_x;
    const int ty = bid % L.tiles_y; bid /= L.tiles_y;
    const int b = bid;
    const int m0 = mblk * BM;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * p.stride - p.pad_y, ix0 = ox0 * p.stride - p.pad_x;

    // ---- per-lane fragment bases ----
    const int CKh = L.CK >> 1;
    int pixoff[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int q = wave * WN + n;
        const int r = q * RPT + (j >> L.tw_log2);
        const int c = j & (TW - 1);
        pixoff[n] = (r * p.stride) * L.IWp + c * p.stride + half * CKh * L.plane;
    }
    const int wlane = half * CKh * KK * BM + j;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const size_t x_b = (size_t)b * p.Cin * p.H * p.W;
    const int scol = tid & 63, srow = tid >> 6;

    for (int c0 = 0; c0 < p.Cin; c0 += L.CK) {
        __syncthreads();        // previous chunk's fragments fully consumed
        // ---- stage input tile (prologue fused: style/demod scale, activation-gradient mask) ----
        {
            int c = 0, iy = srow;
            while (iy >= L.IH) { iy -= L.IH; ++c; }
            while (c < L.CK) {
                const int ci = c0 + c;
                const int gy = iy0 + iy;
                const bool rowok = (ci < p.Cin) && (gy >= 0) && (gy < p.H);
                float sc = 1.f;
                if (rowok && p.in_scale) sc = p.in_scale[(size_t)b * p.Cin + ci];
                const size_t rbase = x_b + ((size_t)ci * p.H + gy) * p.W;
                float* dst = lds_in + c * L.plane + iy * L.IWp;
                for (int ix = scol; ix < L.IW; ix += 64) {
                    const int gx = ix0 + ix;
                    float v = 0.f;
                    if (rowok && gx >= 0 && gx < p.W) {
                        v = p.x[rbase + gx] * sc;
                        if (p.in_mask) v *= (p.in_mask[rbase + gx] > 0.f) ? p.mask_pos : p.mask_neg;
                    }
                    dst[ix] = v;
                }
                iy += 4;
                while (iy >= L.IH) { iy -= L.IH; ++c; }
            }
        }
        // ---- stage weights: rows (c*KK+tap) of BM contiguous channels, float4 coalesced ----
        {
            const int rows = L.CK * KK;
            const int rows_valid = (p.Cin - c0) * KK;       // rows beyond Cin are zero
            constexpr int V = BM / 4;
            const float* wsrc = p.w + (size_t)c0 * KK * p.CoutP + m0;
            for (int idx = tid; idx < rows * V; idx += 256) {
                const int row = idx / V, c4 = idx - row * V;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < rows_valid && m0 + c4 * 4 < p.CoutP)
                    v = *reinterpret_cast<const float4*>(wsrc + (size_t)row * p.CoutP + c4 * 4);
                *reinterpret_cast<float4*>(lds_w + row * BM + c4 * 4) = v;
            }
        }
        __syncthreads();
        // ---- MFMA over the chunk: lanes 0-31 take channel cc, lanes 32-63 channel cc + CK/2 ----
        for (int cc = 0; cc < CKh; ++cc) {
            const float* wr = lds_w + wlane + cc * KK * BM;
            const float* ir = lds_in + cc * L.plane;
            for (int ky = 0; ky < p.KH; ++ky) {
                for (int kx = 0; kx < p.KW; ++kx) {
                    float a[WM], bb[WN];
                    const int toff = ky * L.IWp + kx;
#pragma unroll
                    for (int m = 0; m < WM; ++m) a[m] = wr[m * 32];
#pragma unroll
                    for (int n = 0; n < WN; ++n) bb[n] = ir[pixoff[n] + toff];
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
                    wr += BM;
                }
            }
        }
    }

    // ---- epilogue: demod / noise / bias / residual / activation, 128-B row segments per register ----
    const size_t plane_o = (size_t)p.OHf * p.OWf;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int q = wave * WN + n;
        const int oy = oy0 + q * RPT + (j >> L.tw_log2);
        const int ox = ox0 + (j & (TW - 1));
        const bool pok = (oy < p.OH) && (ox < p.OW);
        const int oyf = oy * p.oy_step + p.oy_off, oxf = ox * p.ox_step + p.ox_off;
        const size_t poff = (size_t)oyf * p.OWf + oxf;
        float nz = 0.f;
        if (pok && p.noise) nz = p.noise[(size_t)b * plane_o + poff] * p.noise_w;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pok && co < p.Cout) {
                    float v = acc[m][n][r];
                    if (p.out_scale) v *= p.out_scale[(size_t)b * p.Cout + co];
                    const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + poff;
                    if (p.out_mask) v = (p.out_mask[oidx] > 0.f) ? v : 0.f;
                    v += nz;
                    if (p.bias) v += p.bias[co];
                    if (p.residual) {
                        float rv = p.residual[oidx];
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

// ------------------------------------------------------------------------------------------------------------
static int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct TileCfg { int wm, wn; };
static const TileCfg kTiles[] = {{4, 2}, {2, 2}, {1, 4}, {2, 1}, {1, 1}, {1, 2}, {4, 1}};
static const int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int WM, int WN>
static hipError_t launch_cfg(const l2i_conv_params& p, const ConvLaunch& L, int grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;          // allow > 64 KiB dynamic LDS if ever requested
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN>), dim3(grid), dim3(256), lds, st, p, L);
    return hipGetLastError();
}

extern "C" int l2i_conv2d_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d: null tensor");
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d: non-positive dimension");
    if (p.KH <= 0 || p.KW <= 0 || p.KH > 16 || p.KW > 16 || (p.stride != 1 && p.stride != 2))
        return l2i_set_error(L2I_E_ARG, "conv2d: kernel size must be 1..16 and stride 1 or 2");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv2d: CoutP must be Cout rounded up to 32");
    if (p.oy_step <= 0 || p.ox_step <= 0 || (p.OH - 1) * p.oy_step + p.oy_off >= p.OHf || (p.OW - 1) * p.ox_step + p.ox_off >= p.OWf ||
        p.oy_off < 0 || p.ox_off < 0)
        return l2i_set_error(L2I_E_ARG, "conv2d: output window exceeds the output tensor");

    // ---- tile selection: largest channel tile that divides the work into enough blocks for 256 CUs ----
    int sel = -1;
    if (p.tile_hint > 0) {
        if (p.tile_hint > kNumTiles) return l2i_set_error(L2I_E_ARG, "conv2d: tile_hint out of range");
        sel = p.tile_hint - 1;
    } else {
        const int twl = ilog2_ceil(p.OW < 32 ? p.OW : 32);
        const int TW = 1 << twl, RPT = 32 >> twl;
        long best_blocks = -1;
        for (int i = 0; i < 5; ++i) {       // preference order of kTiles[0..4]
            const int BM = kTiles[i].wm * 32, TH = 4 * kTiles[i].wn * RPT;
            if (BM > p.CoutP) continue;
            const long blocks = (long)p.B * ((p.OH + TH - 1) / TH) * ((p.OW + TW - 1) / TW) * (p.CoutP / BM + (p.CoutP % BM ? 1 : 0));
            if (blocks >= 512) { sel = i; break; }
            if (blocks > best_blocks) { best_blocks = blocks; sel = i; }
        }
        if (sel < 0) sel = 4;
    }
    const int WM = kTiles[sel].wm, WN = kTiles[sel].wn, BM = WM * 32;

    ConvLaunch L;
    L.tw_log2 = ilog2_ceil(p.OW < 32 ? p.OW : 32);
    const int TW = 1 << L.tw_log2, RPT = 32 >> L.tw_log2, TH = 4 * WN * RPT;
    L.tiles_x = (p.OW + TW - 1) / TW;
    L.tiles_y = (p.OH + TH - 1) / TH;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    L.IH = (TH - 1) * p.stride + p.KH;
    L.IW = (TW - 1) * p.stride + p.KW;
    L.IWp = L.IW | 1;
    L.plane = L.IH * L.IWp;
    L.plane = (L.plane + 3) & ~3;           // keeps the weight region 16-B aligned for any even CK
    const int KK = p.KH * p.KW;
    const size_t per_c = (size_t)(L.plane + KK * BM) * sizeof(float);
    int ck = (int)((48 * 1024) / per_c) & ~1;
    if (ck < 2) ck = 2;
    int cin_even = (p.Cin + 1) & ~1;
    if (ck > cin_even) ck = cin_even;
    if (ck > 64) ck = 64;
    L.CK = ck;
    const size_t lds = per_c * ck;
    if (lds > 160 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d: tile does not fit in LDS");
    const long grid = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (grid <= 0 || grid > 0x7fffffffL) return l2i_set_error(L2I_E_ARG, "conv2d: grid too large");

    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    switch (sel) {
        case 0: e = launch_cfg<4, 2>(p, L, (int)grid, lds, st); break;
        case 1: e = launch_cfg<2, 2>(p, L, (int)grid, lds, st); break;
        case 2: e = launch_cfg<1, 4>(p, L, (int)grid, lds, st); break;
        case 3: e = launch_cfg<2, 1>(p, L, (int)grid, lds, st); break;
        case 4: e = launch_cfg<1, 1>(p, L, (int)grid, lds, st); break;
        case 5: e = launch_cfg<1, 2>(p, L, (int)grid, lds, st); break;
        default: e = launch_cfg<4, 1>(p, L, (int)grid, lds, st); break;
    }
    if (e != hipSuccess) return l2i_set_error(L2I_E_LAUNCH, hipGetErrorString(e));
    return L2I_OK;
}
