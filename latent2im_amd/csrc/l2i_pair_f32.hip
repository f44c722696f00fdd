// l2i_pair_f32.hip — [r6] two chained 1x1 convolutions on fp32 NCHW maps in ONE launch (gfx950): the fp32 twin of l2i_pair_h8.hip for the headline
// configuration's regressor.  Entry point l2i_conv1x1_pair_f32.
//
// ResNet-50's trunk (torchvision Bottleneck as called at transform_base.py:396-403, 416-424): conv3 (C -> 4C) + identity + ReLU of a block, then conv1
// (4C -> C) + ReLU of the next block.  As two l2i_conv2d_f32 launches the 4C-wide fp32 map is written and read straight back (layer1 at batch 8: 0.54 GB each
// way), and each launch pays its own MFMA time plus most of its byte time (an fp32 MFMA shares its issue slot with everything else a wave does: 327 + 203 us
// for 109 + 109 us of matrix time, tools/probes/f32_pair_room.py).  Here a wave keeps its 32 WN pixels for both convs:
//   * the accumulator layout of v_mfma_f32_32x32x2_f32 meets its B operand layout: register r of lane (half, j) is channel (r & 3) + 8 (r >> 2) + 4 half of
//     pixel j, and a B operand is "lane (half, j) holds K row `half` of column j" — so register r of the finished first conv IS the B operand of a K step that
//     contracts channels {c_r, c_r + 4}; the second conv's A operand for that step is W2[.][32 c + c_r + 4 half], one ds_read_b32 from the natural [k][Cout2]
//     weight rows.  No LDS round trip, no shuffle between the convs;
//   * per 32-channel chunk of the first conv a wave issues K1 / 2 * WN + 16 * MC * WN MFMAs (128 at layer1: 8 192 cycles) against ~100 VALU instructions of
//     epilogue: the matrix pipe is what the wave waits for, and that much matrix time per chunk covers an HBM round trip with one chunk of prefetch;
//   * weights (the chunk's 32 rows of W1 and 32 K rows of W2) go through two shared LDS stages by 16-byte DMA; the identity map of the chunk (32 channels x the
//     wave's pixels) by 16-byte DMA into a per-wave ring and is read back per lane; the wide map is stored once, 128 contiguous bytes per half wave and channel.
// Two tilings: 128-pixel blocks, two per CU (default), or 256-pixel blocks, one per CU with the accumulators in AGPRs.  Where the time goes (layer1, batch 8:
// 400 us against 530 for the two launches; floors: 218 us of fp32 MFMA at the nominal clock, 268 us of HBM at 5 TB/s): the kernel is matrix-bound at the
// same ~0.65 of the nominal peak as the GEMM kernel (clock under fp32 MFMA load, the epilogue's VALU on the shared issue slot, a block's prologue); what the
// fusion removes is the byte time the two launches add on top.  Inside the training step other streams fill those stalls anyway: c3 +0.2 .. 0.5 %
// (tools/ab/r06_pair_f32.sh), against +23 % / +16 % in isolation.
// The second conv accumulates channel pairs in another order than l2i_conv2d_f32's GEMM kernel ({c, c + 4} instead of {c, c + 1}): results agree to fp32
// rounding (tests: 1e-5 of the map's magnitude), not bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));

namespace l2i_pair_f32 {

struct PairF32Launch {
    int total, tiles_per_sample, npix, nch;
};

#define L2I_PF_DMA16(voff, rsrc, ldsaddr, soff)                                                                                            \
    do {                                                                                                                                   \
        unsigned keep_;                                                                                                                    \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"     \
                     : "=&s"(keep_) : "v"(voff), "s"(rsrc), "s"(ldsaddr), "s"(soff) : "memory");                                           \
    } while (0)

template <int K1, int MC, int WN, int OCC>
__global__ __launch_bounds__(256, OCC) void pair_f32_kernel(const l2i_conv_params p1, const l2i_conv_params p2, const PairF32Launch L) {
    constexpr int C3 = 32 * MC;
    constexpr int W1F = K1 * 32, W2F = 32 * C3, WSTAGE = W1F + W2F;       // floats per stage
    constexpr int W1P = K1 / 8, W2P = C3 / 8, NPIECES = W1P + W2P;         // 1 KiB DMA pieces per chunk
    static_assert(NPIECES % 4 == 0, "pieces are dealt to four waves");
    constexpr int NPW = NPIECES / 4;
    constexpr int PXW = 32 * WN;                                           // pixels of a wave
    constexpr int RF = 32 * PXW;                                           // floats of a wave's operand tile [32 channels][PXW]
    constexpr int RPW = RF / 256;                                          // its DMA pieces
    constexpr int CPP = 32 / RPW;                                          // channels per piece
    constexpr int RS = 2;
    extern __shared__ __attribute__((aligned(16))) float smemf[];
    float* const w_st = smemf;
    float* const r_st = smemf + 2 * WSTAGE;
    float* const bias_s = r_st + 4 * RS * RF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w = (int)blockIdx.x;
    if (w >= L.total) return;
    const int b = __builtin_amdgcn_readfirstlane(w / L.tiles_per_sample);
    const int tile = __builtin_amdgcn_readfirstlane(w - b * L.tiles_per_sample);
    const unsigned npix = (unsigned)L.npix;
    const unsigned pix0 = (unsigned)tile * (4u * PXW) + (unsigned)wave_u * PXW;
    const int nch = L.nch, c1n = p1.Cout, c2n = p2.Cout;

    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1.w, 0, (unsigned)((size_t)p1.Cin * p1.CoutP * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p2.w, 0, (unsigned)((size_t)p2.Cin * p2.CoutP * 4), 0x00020000);
    const unsigned res_bytes = (unsigned)c1n * npix * 4u;
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p1.residual) + (size_t)b * res_bytes), 0, res_bytes, 0x00020000);

    // ---- weight DMA: piece q = wave + 4 t.  q < W1P: rows 8 q .. 8 q + 7 of W1[k][32 c ..] (8 lanes = 32 channels a row); else rows of W2[32 c + .][0 .. C3) ----
    unsigned wvoff[NPW];
#pragma unroll
    for (int t = 0; t < NPW; ++t) {
        const int q = wave + 4 * t;
        if (q < W1P) wvoff[t] = (unsigned)(((8 * q + (lane >> 3)) * p1.CoutP + 4 * (lane & 7)) * 4);
        else {
            constexpr int LPR = C3 / 4, RPP = 64 / LPR;                    // lanes per row, rows per piece
            const int q2 = q - W1P;
            wvoff[t] = (unsigned)(((q2 * RPP + lane / LPR) * p2.CoutP + 4 * (lane % LPR)) * 4);
        }
    }
    auto dma_w = [&](int c, int stage) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(w_st + stage * WSTAGE);
        const unsigned so1 = (unsigned)c * 32u * 4u, so2 = (unsigned)c * 32u * (unsigned)p2.CoutP * 4u;
#pragma unroll
        for (int t = 0; t < NPW; ++t) {
            const int q = wave_u + 4 * t;
            if (q < W1P) L2I_PF_DMA16(wvoff[t], rs_w1, __builtin_amdgcn_readfirstlane(lds0 + q * 1024), so1);
            else L2I_PF_DMA16(wvoff[t], rs_w2, __builtin_amdgcn_readfirstlane(lds0 + (W1F + (q - W1P) * 256) * 4), so2);
        }
    };
    // ---- identity-map DMA: the wave's [32 channels][PXW pixels] of chunk c into its ring stage: piece p = channels p CPP .. , 16 bytes = 4 pixels per lane ----
    constexpr int LPC = PXW / 4;                                           // lanes per channel row
    const unsigned rvoff = ((unsigned)(lane / LPC) * npix + pix0 + 4u * (unsigned)(lane % LPC)) * 4u;
    float* const r_mine = r_st + wave * (RS * RF);
    auto dma_r = [&](int c) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(r_mine + (c % RS) * RF);
#pragma unroll
        for (int p = 0; p < RPW; ++p)
            L2I_PF_DMA16(rvoff, rs_r, __builtin_amdgcn_readfirstlane(lds0 + p * 1024), (unsigned)(32 * c + p * CPP) * npix * 4u);
    };

    // ---- prologue ----
    dma_w(0, 0);
    dma_r(0);
    float xf[K1 / 2][WN];                                                  // B operands of the first conv: K step ks = channels 2 ks, 2 ks + 1 (lane half)
    {
        const float* xb = p1.x + ((size_t)b * p1.Cin + half) * npix + pix0 + j;
#pragma unroll
        for (int ks = 0; ks < K1 / 2; ++ks)
#pragma unroll
            for (int n = 0; n < WN; ++n) xf[ks][n] = xb[(size_t)(2 * ks) * npix + 32 * n];
    }
    for (int i = tid; i < c1n + C3; i += 256) bias_s[i] = i < c1n ? (p1.bias ? p1.bias[i] : 0.f) : (p2.bias ? p2.bias[i - c1n] : 0.f);
    const bool relu1 = p1.act == L2I_ACT_RELU, relu2 = p2.act == L2I_ACT_RELU;

    pf32x16 accC[MC][WN];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accC[m][n][r] = 0.f;
    float* const y1b = p1.y + (size_t)b * c1n * npix + pix0 + j;

    for (int c = 0; c < nch; ++c) {
        // W(c) and the identity tile of c (issued in iteration c - 1, followed by its 16 WN stores) have landed
        if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * WN) : "memory");
        __syncthreads();
        if (c + 1 < nch) {
            dma_w(c + 1, (c + 1) & 1);
            dma_r(c + 1);
        }
        const float* w1s = w_st + (c & 1) * WSTAGE + half * 32 + j;       // A of K step ks: W1[32 c + j][2 ks + half] = stage[(2 ks + half) * 32 + j]
        const float* w2s = w_st + (c & 1) * WSTAGE + W1F + j;
        const float* rq = r_mine + (c % RS) * RF + j;
        pf32x16 accB[WN];
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[n][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K1 / 2; ++ks) {
            const float a = w1s[ks * 64];
#pragma unroll
            for (int n = 0; n < WN; ++n) accB[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xf[ks][n], accB[n], 0, 0, 0);
        }
        // epilogue of the first conv on the accumulator registers, then each register is the B operand of one K step of the second conv
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;              // channel of register r inside the chunk
            const float bv = bias_s[32 * c + ch];
            float v[WN];
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float t = accB[n][r] + bv;
                t += rq[ch * PXW + 32 * n];
                if (relu1) t = t > 0.f ? t : 0.f;
                y1b[(size_t)(32 * c + ch) * npix + 32 * n] = t;
                v[n] = t;
            }
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                const float a = w2s[ch * C3 + 32 * m];                     // W2[32 m + j][32 c + ch]
#pragma unroll
                for (int n = 0; n < WN; ++n) accC[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[n], accC[m][n], 0, 0, 0);
            }
        }
    }
    float* const y2b = p2.y + (size_t)b * c2n * npix + pix0 + j;
    const float* const bias2_s = bias_s + c1n;
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float bv = bias2_s[ch];
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float t = accC[m][n][r] + bv;
                if (relu2) t = t > 0.f ? t : 0.f;
                y2b[(size_t)ch * npix + 32 * n] = t;
            }
        }
}

template <int K1, int MC, int WN, int OCC>
static int launch_pair_f32(const l2i_conv_params& p1, const l2i_conv_params& p2, hipStream_t st) {
    PairF32Launch L;
    L.npix = p1.H * p1.W;
    L.tiles_per_sample = L.npix / (128 * WN);
    L.total = p1.B * L.tiles_per_sample;
    L.nch = p1.Cout / 32;
    constexpr int C3 = 32 * MC, WSTAGE = K1 * 32 + 32 * C3, RF = 32 * 32 * WN;
    const size_t lds = (size_t)(2 * WSTAGE + 4 * 2 * RF + p1.Cout + C3) * 4;
    if (lds > 160 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_f32: stages do not fit the LDS");
    L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_f32_kernel<K1, MC, WN, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((pair_f32_kernel<K1, MC, WN, OCC>), dim3((unsigned)L.total), dim3(256), lds, st, p1, p2, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

static const char* pair_f32_unsupported(const l2i_conv_params& p) {
    if (!p.w || !p.y) return "null tensor";
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_y != 0 || p.pad_x != 0 || p.oy_step != 1 || p.ox_step != 1 || p.oy_off || p.ox_off) return "both convs must be 1x1, stride 1, pad 0, dense output";
    if (p.OH != p.H || p.OW != p.W || p.OHf != p.H || p.OWf != p.W) return "output maps have the input's size";
    if (p.in_scale || p.in_mask || p.out_scale || p.noise || p.accumulate || p.res_sub || p.res_mask || p.out_mask || p.sq_ref || p.sq_out || p.ksplit > 1)
        return "only bias, residual (first conv) and ReLU are fused";
    if (l2i_unsupported_v5_fields(p, false, false, false)) return "ABI-5/6 fields do not apply";
    if (p.act != L2I_ACT_NONE && p.act != L2I_ACT_RELU) return "activation: none or ReLU";
    if (p.out_gain != 1.f) return "out_gain must be 1";
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0 || (p.Cout % 32) != 0 || (p.Cin % 32) != 0) return "channel counts must be multiples of 32";
    return nullptr;
}
}  // namespace l2i_pair_f32
using namespace l2i_pair_f32;

extern "C" int l2i_conv1x1_pair_f32(const l2i_conv_params* first, const l2i_conv_params* second, void* stream) {
    if (!first || !second) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_f32: null params");
    const l2i_conv_params &p1 = *first, &p2 = *second;
    if (const char* m = pair_f32_unsupported(p1)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (const char* m = pair_f32_unsupported(p2)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (!p1.x || !p1.residual) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_f32: the first conv needs its input and a residual operand (the trunk)");
    if (p2.x && p2.x != p1.y) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_f32: the second conv reads the first conv's output (second->x must be first->y or NULL)");
    if (p2.residual) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_f32: no residual on the second conv");
    if (p2.Cin != p1.Cout || p2.B != p1.B || p2.H != p1.H || p2.W != p1.W) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_f32: the two convs do not chain");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if (!al16(p1.x) || !al16(p1.w) || !al16(p1.y) || !al16(p1.residual) || !al16(p2.w) || !al16(p2.y)) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_f32: tensors must be 16-byte aligned");
    const long npix = (long)p1.H * p1.W;
    if ((size_t)p1.Cout * npix * 4 >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_f32: one sample must stay below 4 GiB");
    hipStream_t st = (hipStream_t)stream;
    const int K1 = p1.Cin, C3 = p2.Cout;
    if ((npix % 256) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_f32: H * W must be a multiple of 256");
    // (measured, tools/ab/r06_pair_f32_variant.sh: isolated the two tilings are within 3 % of each other — 396 / 407 us at layer1, 345 / 337 at layer2 against 530 / 394
    //  for the two launches — inside the step, where other streams' kernels share the CUs, the smaller blocks are ahead: 80.6 - 80.8 against 80.0 - 80.6 images/s)
    static const int var_env = getenv("L2I_PAIR_F32_VARIANT") ? atoi(getenv("L2I_PAIR_F32_VARIANT")) : 1;      // 1: 128-pixel blocks, two per CU (default); 0: 256-pixel blocks, one per CU
    if (var_env == 1) {
        if (K1 == 64 && C3 == 64) return launch_pair_f32<64, 2, 1, 2>(p1, p2, st);
        if (K1 == 64 && C3 == 128) return launch_pair_f32<64, 4, 1, 2>(p1, p2, st);
        if (K1 == 128 && C3 == 128) return launch_pair_f32<128, 4, 1, 2>(p1, p2, st);
    }
    if (K1 == 64 && C3 == 64) return launch_pair_f32<64, 2, 2, 1>(p1, p2, st);
    if (K1 == 64 && C3 == 128) return launch_pair_f32<64, 4, 2, 1>(p1, p2, st);
    if (K1 == 128 && C3 == 128) return launch_pair_f32<128, 4, 2, 1>(p1, p2, st);
    return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_f32: built for (Cin1, Cout2) = (64, 64), (64, 128), (128, 128)");
}
