// l2i_gemm.hip — 1x1 stride-1 convolution without a style scale (the bottleneck 1x1s of ResNet-50, forward and input-gradient:
// half of its launches) as a plain fp32 GEMM  y[co, px] = sum_ci w[ci, co] x[ci, px]  per sample.
//
// Same MFMA mapping as l2i_conv.hip (M = out-channels from the [Cin][CoutP] pack, N = pixels, v_mfma_f32_32x32x2_f32; lane
// halves take channels p and p + CK/2 of a chunk), but built on what the Winograd kernel taught about this chip: a VALU
// instruction costs ~4 cycles of fp32-MFMA time at any occupancy, while LDS and memory instructions ride free beside MFMAs.
// Both operand tiles (and the activation-gradient mask tile of the backward launches, applied to the B fragments: 3 VALU per
// 2*WM MFMAs) go global -> LDS by DMA (buffer_load_dwordx4 ... lds: an x row of 256
// pixels is exactly one 1 KiB wave instruction) — no staging registers, no commit pass, no per-tap pointer arithmetic: the K
// loop is ds_read + MFMA only (fragment addresses are lane base + compile-time offsets).  Double-buffered stages, one
// barrier per 16-channel chunk, three blocks per CU.  Epilogue = the wide LDS-transposed one of l2i_conv.hip (bias,
// residual (+mask), output mask, activation, gains, accumulate).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace gm {
constexpr int CK = 16, CKh = 8, BN = 256;              // channels per chunk, pixels per block (4 waves x 64)
}

template <int WM, bool MASK, int NOPS>
__global__ __launch_bounds__(256, (MASK || NOPS > 0) ? 2 : 3) void gemm1x1_kernel(const l2i_conv_params p, int tiles, int mblocks, int total) {
    using namespace gm;
    constexpr int BM = WM * 32;
    constexpr int XS = CK * BN, MS = MASK ? CK * BN : 0, WS = CK * BM, STAGE = XS + MS + WS;     // floats per stage: x | gradient mask | w
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    // XCD-aware order: the channel blocks of one pixel tile run on the same XCD and share the x tile in its L2
    const int G = gridDim.x;
    int w = (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    if (w >= total) return;
    const int mblk = w % mblocks; w /= mblocks;
    const int tile = w % tiles;
    const int b = w / tiles;
    const int m0 = mblk * BM, px0 = tile * BN;
    const size_t HW = (size_t)p.H * p.W;

    const unsigned x_bytes = (unsigned)((size_t)p.Cin * HW * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.Cin * HW), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + (size_t)b * p.Cin * HW), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((size_t)p.Cin * p.CoutP * sizeof(float)), 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // x: wave k-th piece = row (k * 4 + wave) of the chunk, lane l = pixels px0 + 4 l .. + 3   (1 KiB contiguous in global and in LDS)
    const unsigned xvoff = (unsigned)((px0 + lane * 4) * sizeof(float));
    const unsigned xrow_b = (unsigned)(HW * sizeof(float));
    // w: piece = 64 float4 = (256 / BM) rows of BM channels; float4 index q = piece * 64 + lane -> row q / (BM/4), column q % (BM/4)
    constexpr int WV = BM / 4;                         // float4 per weight row
    constexpr int WPIECES = CK * WV / 64;              // pieces per chunk (8 for BM = 128, 4 for BM = 64), spread over the 4 waves
    constexpr int WPW = (WPIECES + 3) / 4;

    // Staging of the next chunk in DMA slots (inline asm, not the builtin: hipcc cannot tell the DMA target from the stage being read
    // and would drain vmcnt before the next ds_read, see l2i_wino.hip): slots 0..3 = x rows (k*4 + wave) (+ the mask rows), slots
    // 4..4+WPW-1 = weight pieces.  The K loop issues one slot per k-pair, between the MFMAs; `on` = false swaps in null descriptors.
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);
    auto issue_slot = [&](int sl, int c0, float* stage, bool on) {
        const unsigned lds_x = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)stage;
        unsigned keep;
        if (sl < CK / 4) {
            const int row = sl * 4 + wave_u;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(xvoff), "s"(on ? rs_x : rs_null), "s"(__builtin_amdgcn_readfirstlane(lds_x + row * BN * 4)), "s"((unsigned)(c0 + row) * xrow_b));
            if constexpr (MASK)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(xvoff), "s"(on ? rs_m : rs_null), "s"(__builtin_amdgcn_readfirstlane(lds_x + (XS + row * BN) * 4)), "s"((unsigned)(c0 + row) * xrow_b));
        } else if (sl < CK / 4 + WPW) {
            const int piece = (sl - CK / 4) * 4 + wave_u;
            const int q = (piece < WPIECES ? piece : 0) * 64 + lane;
            const unsigned wv = (unsigned)((((q / WV)) * p.CoutP + m0 + (q % WV) * 4) * sizeof(float));
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wv), "s"((on && piece < WPIECES) ? rs_w : rs_null),
                           "s"(__builtin_amdgcn_readfirstlane(lds_x + (XS + MS + (piece < WPIECES ? piece : 0) * 256) * 4)),
                           "s"((unsigned)((size_t)c0 * p.CoutP * sizeof(float))));
        }
    };

    f32x16 acc[WM][2];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int nchunks = p.Cin / CK;
#pragma unroll
    for (int sl = 0; sl < CK / 4 + WPW; ++sl) issue_slot(sl, 0, smem, true);
    for (int ch = 0; ch < nchunks; ++ch) {
        float* st = smem + (ch & 1) * STAGE;
        float* nx = smem + ((ch + 1) & 1) * STAGE;
        const bool on = ch + 1 < nchunks;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's tiles have landed
        __syncthreads();                                   // ... for every wave; the other stage is free to refill
        const float* xb = st + half * CKh * BN + wave * 64 + j;
        const float* wb = st + XS + MS + half * CKh * BM + j;
#pragma unroll
        for (int pp = 0; pp < CKh; ++pp) {
            float a[WM], bb[2];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = wb[pp * BM + m * 32];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bb[n] = xb[pp * BN + n * 32];
                if constexpr (MASK) bb[n] *= (xb[XS + pp * BN + n * 32] > 0.f) ? p.mask_pos : p.mask_neg;   // activation-gradient mask: 3 VALU per fragment
            }
            issue_slot(pp, (ch + 1) * CK, nx, on);         // next chunk, one DMA slot per k-pair, beside the MFMAs
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (null-descriptor DMA of the last chunk)
    __syncthreads();                                       // the stages become the transpose strips

    // ---- epilogue (as l2i_conv.hip, epilogue A): per-wave LDS transpose -> 16-byte global accesses.  NOPS > 0: the global operands of
    // the fused terms (up to NOPS of: output mask, residual, res_sub, residual mask, the old y of an accumulating launch; the host picks
    // the instantiation by counting them) are fetched one group of four channel rows AHEAD of the group being finished.  With Cin =
    // 64 .. 256 this kernel is a streaming kernel whose K loop is shorter than its epilogue, and the load -> use -> store chain of the
    // plain loop (NOPS = 0: launches without such operands, or with more than two) exposes one HBM round trip per row pair ----
    float* reg = smem + wave * (32 * 64);                  // [32 channels][64 pixels] of this wave
    const int ch_l = lane >> 4, px = (lane & 15) * 4;
    const size_t pix = (size_t)px0 + wave * 64 + px;
    const bool pok = pix < HW;
    const float rc = p.res_sub ? p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) : 1.f;
    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pok && p.noise) {
        nz = *reinterpret_cast<const float4*>(p.noise + (size_t)b * HW + pix);
        nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
    }
    // one output vector: v = 4 pixels of channel co straight from the accumulators; om / rv / sb / rm / yo = the operand vectors (only read
    // when the operand exists)
    auto apply = [&](float4 v, int co, size_t oidx, const float4& om, float4 rv, const float4& sb, const float4& rm, const float4& yo) {
        if (p.out_scale) { const float sc = p.out_scale[(size_t)b * p.Cout + co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        if (p.out_mask) { v.x = om.x > 0.f ? v.x : 0.f; v.y = om.y > 0.f ? v.y : 0.f; v.z = om.z > 0.f ? v.z : 0.f; v.w = om.w > 0.f ? v.w : 0.f; }
        if (p.noise) { v.x += nz.x; v.y += nz.y; v.z += nz.z; v.w += nz.w; }
        if (p.bias) { const float bv = p.bias[co]; v.x += bv; v.y += bv; v.z += bv; v.w += bv; }
        if (p.residual) {
            if (p.res_sub) { rv.x = rc * (rv.x - sb.x); rv.y = rc * (rv.y - sb.y); rv.z = rc * (rv.z - sb.z); rv.w = rc * (rv.w - sb.w); }   // res_coef * (residual - res_sub)
            if (p.res_mask) { rv.x = rm.x > 0.f ? rv.x : 0.f; rv.y = rm.y > 0.f ? rv.y : 0.f; rv.z = rm.z > 0.f ? rv.z : 0.f; rv.w = rm.w > 0.f ? rv.w : 0.f; }
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (p.act == L2I_ACT_LRELU) {
            v.x = (v.x > 0.f ? v.x : v.x * p.act_slope) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.act_slope) * p.act_gain;
            v.z = (v.z > 0.f ? v.z : v.z * p.act_slope) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.act_slope) * p.act_gain;
        } else if (p.act == L2I_ACT_RELU) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        }
        if (p.out_gain != 1.f) { v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain; }
        if (p.accumulate) { v.x += yo.x; v.y += yo.y; v.z += yo.z; v.w += yo.w; }
        *reinterpret_cast<float4*>(p.y + oidx) = v;
    };
    auto to_strip = [&](int m) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) reg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + q * 32 + j] = acc[m][q][r];
    };
    if constexpr (NOPS == 0) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            to_strip(m);
#pragma unroll 2
            for (int i = 0; i < 8; ++i) {
                const int chn = i * 4 + ch_l;
                const int co = m0 + m * 32 + chn;
                const float4 v = *reinterpret_cast<const float4*>(&reg[chn * 64 + px]);
                if (pok && co < p.Cout) {
                    const size_t oidx = ((size_t)b * p.Cout + co) * HW + pix;
                    apply(v, co, oidx, p.out_mask ? *reinterpret_cast<const float4*>(p.out_mask + oidx) : z4,
                          p.residual ? *reinterpret_cast<const float4*>(p.residual + oidx) : z4,
                          p.res_sub ? *reinterpret_cast<const float4*>(p.res_sub + oidx) : z4,
                          p.res_mask ? *reinterpret_cast<const float4*>(p.res_mask + oidx) : z4,
                          p.accumulate ? *reinterpret_cast<const float4*>(p.y + oidx) : z4);
                }
            }
        }
    } else {
        // operand slots in canonical order; the host guarantees that at most NOPS exist
        // ([r6] res_mask == out_mask — the ReLU mask of a bottleneck's input on both terms of its input gradient, regressor.py — is ONE operand: one fetch)
        const bool rm_is_om = p.res_mask && p.res_mask == p.out_mask;
        const float* cand[5] = {p.out_mask, p.residual, p.res_sub, rm_is_om ? nullptr : p.res_mask, p.accumulate ? p.y : nullptr};
        int slot[5];
        const float* sp0 = nullptr;
        const float* sp1 = nullptr;
        int k = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            slot[c] = cand[c] ? k++ : -1;
            if (slot[c] == 0) sp0 = cand[c];
            if (slot[c] == 1) sp1 = cand[c];
        }
        const int s_om = slot[0], s_rv = slot[1], s_sb = slot[2], s_rm = rm_is_om ? slot[0] : slot[3], s_yo = slot[4];
        struct Ops { float4 s[NOPS][4]; };
        auto fetch = [&](int g, Ops& o) {                  // group g = channel rows ((g & 1) * 4 + i) * 4 + ch_l of channel block g >> 1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = m0 + (g >> 1) * 32 + ((g & 1) * 4 + i) * 4 + ch_l;
                const size_t oidx = ((size_t)b * p.Cout + co) * HW + pix;
#pragma unroll
                for (int t = 0; t < NOPS; ++t) {
                    const float* sp = t == 0 ? sp0 : sp1;
                    o.s[t][i] = (pok && co < p.Cout && sp) ? *reinterpret_cast<const float4*>(sp + oidx) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto finish = [&](int g, const Ops& o) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chn = ((g & 1) * 4 + i) * 4 + ch_l;
                const int co = m0 + (g >> 1) * 32 + chn;
                const float4 v = *reinterpret_cast<const float4*>(&reg[chn * 64 + px]);
                if (pok && co < p.Cout) {
                    const size_t oidx = ((size_t)b * p.Cout + co) * HW + pix;
                    auto pick = [&](int sl) -> float4 { return (NOPS > 1 && sl == 1) ? o.s[NOPS - 1][i] : o.s[0][i]; };
                    apply(v, co, oidx, pick(s_om), pick(s_rv), pick(s_sb), pick(s_rm), pick(s_yo));
                }
            }
        };
        Ops oa, ob;
        fetch(0, oa);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            to_strip(m);
            fetch(2 * m + 1, ob);
            finish(2 * m, oa);
            if (m + 1 < WM) fetch(2 * m + 2, oa);
            finish(2 * m + 1, ob);
        }
    }
}

// eligibility: 1x1, stride 1, no padding, dense output window, no style scale, whole 256-pixel tiles inside a sample
bool l2i_gemm1x1_eligible(const l2i_conv_params& p) {
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    const size_t HW = (size_t)p.H * p.W;
    return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_y == 0 && p.pad_x == 0 && !p.in_scale && p.oy_step == 1 && p.ox_step == 1 &&
           p.oy_off == 0 && p.ox_off == 0 && p.OH == p.H && p.OW == p.W && p.OHf == p.H && p.OWf == p.W && (HW % gm::BN) == 0 && (p.Cin % gm::CK) == 0 &&
           (p.CoutP % 64) == 0 && al16(p.x) && al16(p.in_mask) && al16(p.y) && al16(p.w) && al16(p.residual) && al16(p.res_mask) && al16(p.res_sub) && al16(p.out_mask) && al16(p.noise) &&
           (size_t)p.Cin * HW * sizeof(float) < 0xFFFFFFF0ull;
}

int l2i_launch_gemm1x1(const l2i_conv_params& p, hipStream_t st) {
    const int tiles = (int)(((size_t)p.H * p.W) / gm::BN);
    bool wide = (p.CoutP % 128) == 0 && (long)p.B * tiles * (p.CoutP / 128) >= 512;     // small grids: 64-channel blocks fill more CUs
    // expanding layers of the forward pass (ResNet-50 conv3: Cin = Cout / 4, residual + ReLU epilogue): the K loop is shorter than the
    // epilogue, and four 64-channel blocks per CU overlap one block's stores with three K loops better than three 128-channel blocks
    // (measured per shape: 0.185 / 0.210 / 0.304 ms against 0.209 / 0.236 / 0.317 at 256->1024 @64^2, 128->512 @128^2, 64->256 @256^2)
    if (!p.in_mask && p.Cin * 4 <= p.Cout) wide = false;
    if (const char* e = getenv("L2I_GEMM_BM")) wide = (p.CoutP % 128) == 0 && atoi(e) == 128;          // tuning override (tools/bench_layers.py)
    const int BM = wide ? 128 : 64;
    const int mblocks = p.CoutP / BM;
    const long total = (long)p.B * tiles * mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d(1x1 gemm): grid too large");
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    const bool mask = p.in_mask != nullptr;
    const size_t lds = (size_t)2 * (gm::CK * gm::BN * (mask ? 2 : 1) + gm::CK * BM) * sizeof(float);
    // epilogue operands that are read from global memory: up to two are prefetched (NOPS), otherwise the plain loop
    int nops = (p.out_mask ? 1 : 0) + (p.residual ? 1 : 0) + (p.res_sub ? 1 : 0) + ((p.res_mask && p.res_mask != p.out_mask) ? 1 : 0) + (p.accumulate ? 1 : 0);
    if (nops > 2) nops = 0;
    const dim3 g(grid), t(256);
#define L2I_GEMM_LAUNCH(WM_, MASK_, NOPS_)                                                                                              \
    do {                                                                                                                                \
        if (lds > 64 * 1024)                                                                                                            \
            L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm1x1_kernel<WM_, MASK_, NOPS_>),             \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                     \
        hipLaunchKernelGGL((gemm1x1_kernel<WM_, MASK_, NOPS_>), g, t, lds, st, p, tiles, mblocks, (int)total);                           \
    } while (0)
    if (wide) {
        if (mask) { if (nops == 2) L2I_GEMM_LAUNCH(4, true, 2); else if (nops == 1) L2I_GEMM_LAUNCH(4, true, 1); else L2I_GEMM_LAUNCH(4, true, 0); }
        else { if (nops == 2) L2I_GEMM_LAUNCH(4, false, 2); else if (nops == 1) L2I_GEMM_LAUNCH(4, false, 1); else L2I_GEMM_LAUNCH(4, false, 0); }
    } else {
        if (mask) { if (nops == 2) L2I_GEMM_LAUNCH(2, true, 2); else if (nops == 1) L2I_GEMM_LAUNCH(2, true, 1); else L2I_GEMM_LAUNCH(2, true, 0); }
        else { if (nops == 2) L2I_GEMM_LAUNCH(2, false, 2); else if (nops == 1) L2I_GEMM_LAUNCH(2, false, 1); else L2I_GEMM_LAUNCH(2, false, 0); }
    }
#undef L2I_GEMM_LAUNCH
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
