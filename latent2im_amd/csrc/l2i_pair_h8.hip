// l2i_pair_h8.hip — [r6] two chained 1x1 convolutions on h8 maps in ONE launch, the wide map between them never re-read (gfx950).
// Entry points l2i_conv1x1_pair_h8 / l2i_conv1x1_pair_h8_f16.
//
// What it fuses.  ResNet-50's residual trunk alternates a narrow and a 4x wider map (torchvision Bottleneck as called at transform_base.py:396-403,
// 416-424): ... -> 3x3 (C -> C) -> 1x1 expand (C -> 4C) + identity + ReLU -> [next block] 1x1 reduce (4C -> C) + ReLU -> 3x3 ...  As separate
// launches the wide map is written by the expand conv and read straight back by the next block's reduce conv; both launches are HBM-bound
// (profiles/r06_c5_by_shape.txt: 3.7 TB/s, 11 % matrix-pipe busy).  Here a wave keeps its pixels for the whole chain:
//     first  conv:  mid = epi1(W1 . x + ...)          (K = Cin1, M = Cout1: 32 output channels at a time)
//     second conv:  y   = epi2(W2 . round16(mid))     (K = Cout1, M = Cout2 <= 256, accumulated over the chunks of the first)
// `mid` is stored (the next block's identity needs it) but not read again.  The same chain is the BACKWARD of the trunk read the other way:
// G_prev = (W1^T-conv of g_y1 + G) * [block input > 0], then g_y2 of the block below = (W3^T-conv of G_prev) * [y2 > 0].
//
// Why no LDS round trip between the convs.  The 32x32x16 MFMA's accumulator layout and its B operand layout meet: after the lane-half
// exchange of the h8 epilogue (v_permlane32_swap, l2i_h8_common.h: h8_gather) lane (half, j) holds the 8 channels of group 2 pr + half of pixel j
// as one packed 16-byte slot — which IS lane (half, j)'s B fragment of the 16-channel K step {groups 2 pr, 2 pr + 1}.  So the packed output slot
// of the first conv is stored to HBM and fed to the second conv's MFMA from the same registers.  Results are bit-identical to the two-launch
// path: the second conv sees the rounded 16-bit values in both, and its K steps accumulate in the same ascending order.
//
// Pipeline.  Only weights go through shared LDS: chunk c = 32 output channels of the first conv needs W1[32 c .. 32 c + 31][all Cin1] and the
// K slice W2[all Cout2][32 c .. 32 c + 31] — two stages, DMA'd one chunk ahead, one barrier per chunk.  The epilogue operand of the first conv
// (the identity / incoming trunk gradient: the dominant HBM read) is DMA'd RDIST chunks ahead into a per-WAVE ring (a lane fetches exactly the
// slot it will finish: no barrier, no registers held across the latency); counted vmcnt waits (gfx9: loads and stores retire in issue order on
// one counter; every count below is a LOWER bound of the operations issued after the awaited one, so extra compiler-issued stores only make a
// wait longer).  Per-channel vectors (biases) and, for the masked form, the sign bytes of the first epilogue are staged to LDS once in the
// prologue, so the loop holds no compiler-scheduled global load (its waits would not see the DMAs and would drain them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_h8_common.h"

namespace H8_NS {

struct PairLaunch {
    int total, tiles_per_sample, npix, nch;     // blocks, pixel tiles per sample, H * W, 32-channel chunks of the first conv's output
};

template <int KB, int MC, int WN, bool MASKED, int RDIST, int OCC>
__global__ __launch_bounds__(256, OCC) void pair_h8_kernel(const l2i_conv_params p1, const l2i_conv_params p2, const PairLaunch L) {
    constexpr int C3 = 32 * MC;                       // output channels of the second conv
    constexpr int W1S = KB * 64, W2S = 4 * C3;        // slots of a chunk's W1 rows ([kstep][half][32]) and W2 slice ([2 ksteps][half][C3])
    constexpr int WSTAGE = W1S + W2S;
    constexpr int NPIECES = KB + 2 * MC;              // 1 KiB DMA pieces per chunk
    static_assert(NPIECES % 4 == 0 && MC % 2 == 0, "pieces are dealt to four waves; a W2 piece is 64 channels");
    constexpr int NPW = NPIECES / 4;
    constexpr int RS = RDIST + 1;                     // ring stages of the epilogue operand
    constexpr int RPW = 2 * WN;                       // operand pieces per wave and chunk: (pr, n)
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4* const w_st = smem4;
    u32x4* const r_st = smem4 + 2 * WSTAGE;
    float* const bias_s = reinterpret_cast<float*>(r_st + 4 * RS * RPW * 64);
    uint8_t* const mb_s = reinterpret_cast<uint8_t*>(bias_s + (L.nch * 32 + C3));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w = (int)blockIdx.x;
    if (w >= L.total) return;
    const int b = __builtin_amdgcn_readfirstlane(w / L.tiles_per_sample);
    const int tile = __builtin_amdgcn_readfirstlane(w - b * L.tiles_per_sample);
    const unsigned npix = (unsigned)L.npix;
    const unsigned pix0 = (unsigned)tile * (128u * WN) + (unsigned)wave_u * (32u * WN);      // this wave's first pixel
    const int cg_in = p1.Cin / 8, cg1 = p1.Cout / 8, cg2 = p2.Cout / 8;
    const int nch = L.nch;

    // ---- descriptors ----
    const unsigned w1_bytes = (unsigned)((size_t)(p1.Cin / 16) * 2 * p1.CoutP * 16), w2_bytes = (unsigned)((size_t)(p2.Cin / 16) * 2 * p2.CoutP * 16);
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1.w_hi, 0, w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p2.w_hi, 0, w2_bytes, 0x00020000);
    const unsigned res_bytes = (unsigned)cg1 * npix * 16u;
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p1.residual) + (size_t)b * res_bytes), 0, res_bytes, 0x00020000);

    // ---- weight DMA: piece q = wave + 4 t; q < KB: K step q of W1 (both lane halves' 32 rows), else a 64-channel run of one (kstep, half) row of W2's slice ----
    unsigned wvoff[NPW];
#pragma unroll
    for (int t = 0; t < NPW; ++t) {
        const int q = wave + 4 * t;
        if (q < KB) wvoff[t] = (unsigned)(((q * 2 + half) * p1.CoutP + j) * 16);
        else {
            const int q2 = q - KB, r = q2 / (MC / 2), i64 = q2 - r * (MC / 2);
            wvoff[t] = (unsigned)((r * p2.CoutP + i64 * 64 + lane) * 16);
        }
    }
    auto dma_w = [&](int c, int stage) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(w_st + stage * WSTAGE);
        const unsigned so1 = (unsigned)c * 32u * 16u, so2 = (unsigned)c * 64u * (unsigned)p2.CoutP;       // W2 slice: K steps 2 c, 2 c + 1 = rows (2 c) * 2 .. of [kstep][half][CoutP] slots
#pragma unroll
        for (int t = 0; t < NPW; ++t) {
            const int q = wave_u + 4 * t;
            unsigned keep;
            if (q < KB) {
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(wvoff[t]), "s"(rs_w1), "s"(__builtin_amdgcn_readfirstlane(lds0 + q * 1024)), "s"(so1) : "memory");
            } else {
                const int q2 = q - KB, r = q2 / (MC / 2), i64 = q2 - r * (MC / 2);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(wvoff[t]), "s"(rs_w2), "s"(__builtin_amdgcn_readfirstlane(lds0 + (W1S + r * C3 + i64 * 64) * 16)), "s"(so2) : "memory");
            }
        }
    };
    // ---- epilogue-operand DMA: piece (pr, n) of chunk c: lane (half, j) fetches group 4 c + 2 pr + half of pixel pix0 + 32 n + j into its own ring slot ----
    unsigned rvoff[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) rvoff[n] = ((unsigned)half * npix + pix0 + 32u * n + (unsigned)j) * 16u;
    u32x4* const r_mine = r_st + wave * (RS * RPW * 64);
    auto dma_r = [&](int c) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(r_mine + (c % RS) * (RPW * 64));
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(rvoff[n]), "s"(rs_r), "s"(__builtin_amdgcn_readfirstlane(lds0 + (pr * WN + n) * 1024)),
                               "s"((unsigned)(4 * c + 2 * pr) * npix * 16u) : "memory");
            }
    };

    // ---- prologue ----
#pragma unroll
    for (int d = 0; d < RDIST; ++d)
        if (d < nch) dma_r(d);
    dma_w(0, 0);
    bf16x8 xf[KB][WN];                                // the wave's pixels of the first conv's input: K step ks = groups 2 ks, 2 ks + 1
    {
        const u32x4* xb = reinterpret_cast<const u32x4*>(p1.x) + ((size_t)b * cg_in + half) * npix + pix0 + j;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks)
#pragma unroll
            for (int n = 0; n < WN; ++n) xf[ks][n] = __builtin_bit_cast(bf16x8, xb[(size_t)(2 * ks) * npix + 32 * n]);
    }
    for (int i = tid; i < nch * 32 + C3; i += 256)
        bias_s[i] = i < nch * 32 ? (p1.bias ? p1.bias[i] : 0.f) : (p2.bias ? p2.bias[i - nch * 32] : 0.f);
    uint8_t* const mb_mine = mb_s + wave * (nch * RPW * 64);
    if constexpr (MASKED) {
        const uint8_t* mp = reinterpret_cast<const uint8_t*>(p1.out_mask) + ((size_t)b * cg1 + half) * npix + pix0 + j;
        for (int c = 0; c < nch; ++c)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int n = 0; n < WN; ++n) mb_mine[(c * RPW + pr * WN + n) * 64 + lane] = mp[(size_t)(4 * c + 2 * pr) * npix + 32 * n];
    }
    const bool has_b1 = p1.bias != nullptr, relu1 = p1.act == L2I_ACT_RELU, sign1 = p1.mask_out != nullptr, rmask1 = MASKED && p1.res_mask != nullptr;

    f32x16 accC[MC][WN];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accC[m][n][r] = 0.f;

    u32x4* const y1b = reinterpret_cast<u32x4*>(p1.y) + ((size_t)b * cg1 + half) * npix + pix0 + j;
    uint8_t* const s1b = sign1 ? p1.mask_out + ((size_t)b * cg1 + half) * npix + pix0 + j : nullptr;

    for (int c = 0; c < nch; ++c) {
        // W(c) — and the operand pieces of chunk c, issued before it (RDIST = 2) or right behind it (RDIST = 1) — have landed.  Issued after them in
        // iteration c - 1: [RDIST = 2: the operand pieces of chunk c + 1,] then at least RPW stores of `mid`.
        if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (RDIST == 2 && c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW) : "memory");
        __syncthreads();
        if (c + 1 < nch) dma_w(c + 1, (c + 1) & 1);
        if (c + RDIST < nch) dma_r(c + RDIST);

        // ---- first conv, channels 32 c .. 32 c + 31 ----
        const u32x4* w1s = w_st + (c & 1) * WSTAGE + lane;
        f32x16 accB[WN];
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[n][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, w1s[ks * 64]);
#pragma unroll
            for (int n = 0; n < WN; ++n) accB[n] = H8_MFMA(af, xf[ks][n], accB[n], 0, 0, 0);
        }
        // ---- its epilogue: (* mask) + bias + operand (* mask), ReLU, store, sign byte; the packed slots are the second conv's B fragments ----
        const u32x4* rq = r_mine + (c % RS) * (RPW * 64) + lane;
        const u32x4* w2s = w_st + (c & 1) * WSTAGE + W1S + half * C3 + j;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float bv[8];
            if (has_b1) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias_s + 32 * c + (2 * pr + half) * 8), b1 = *reinterpret_cast<const float4*>(bias_s + 32 * c + (2 * pr + half) * 8 + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            }
            bf16x8 frag[WN];
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float g[8], r[8];
                h8_gather(accB[n], pr, half, g);
                h8_unpack(rq[(pr * WN + n) * 64], r);
                if constexpr (MASKED) {
                    const unsigned mb = mb_mine[(c * RPW + pr * WN + n) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] *= ((mb >> e) & 1u) ? p1.mask_pos : p1.mask_neg;
                    if (rmask1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) r[e] = ((mb >> e) & 1u) ? r[e] : 0.f;
                    }
                }
                if (has_b1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] += bv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += r[e];
                if (relu1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = g[e] > 0.f ? g[e] : 0.f;
                }
                const u32x4 out = {cvt_pk_bf16_h8(g[0], g[1]), cvt_pk_bf16_h8(g[2], g[3]), cvt_pk_bf16_h8(g[4], g[5]), cvt_pk_bf16_h8(g[6], g[7])};
                const size_t off = (size_t)(4 * c + 2 * pr) * npix + 32 * n;
                y1b[off] = out;
                if (sign1) s1b[off] = (uint8_t)h8_sign_byte(out);
                frag[n] = __builtin_bit_cast(bf16x8, out);
            }
            // ---- second conv: K step 2 c + pr ----
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, w2s[pr * 2 * C3 + m * 32]);
#pragma unroll
                for (int n = 0; n < WN; ++n) accC[m][n] = H8_MFMA(af, frag[n], accC[m][n], 0, 0, 0);
            }
        }
    }

    // ---- the second conv's epilogue ----
    const bool has_b2 = p2.bias != nullptr, relu2 = p2.act == L2I_ACT_RELU;
    u32x4* const y2b = reinterpret_cast<u32x4*>(p2.y) + ((size_t)b * cg2 + half) * npix + pix0 + j;
    const uint8_t* const m2b = p2.out_mask ? reinterpret_cast<const uint8_t*>(p2.out_mask) + ((size_t)b * cg2 + half) * npix + pix0 + j : nullptr;
    uint8_t* const s2b = p2.mask_out ? p2.mask_out + ((size_t)b * cg2 + half) * npix + pix0 + j : nullptr;
    const float* const bias2_s = bias_s + nch * 32;
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float bv[8];
            if (has_b2) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias2_s + 32 * m + (2 * pr + half) * 8), b1 = *reinterpret_cast<const float4*>(bias2_s + 32 * m + (2 * pr + half) * 8 + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float g[8];
                h8_gather(accC[m][n], pr, half, g);
                const size_t off = (size_t)(4 * m + 2 * pr) * npix + 32 * n;
                if (m2b) {
                    const unsigned mb = m2b[off];
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] *= ((mb >> e) & 1u) ? p2.mask_pos : p2.mask_neg;
                }
                if (has_b2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] += bv[e];
                }
                if (relu2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = g[e] > 0.f ? g[e] : 0.f;
                }
                const u32x4 out = {cvt_pk_bf16_h8(g[0], g[1]), cvt_pk_bf16_h8(g[2], g[3]), cvt_pk_bf16_h8(g[4], g[5]), cvt_pk_bf16_h8(g[6], g[7])};
                y2b[off] = out;
                if (s2b) s2b[off] = (uint8_t)h8_sign_byte(out);
            }
        }
}

template <int KB, int MC, int WN, bool MASKED, int RDIST, int OCC>
static int launch_pair(const l2i_conv_params& p1, const l2i_conv_params& p2, hipStream_t st) {
    PairLaunch L;
    L.npix = p1.H * p1.W;
    L.tiles_per_sample = L.npix / (128 * WN);
    L.total = p1.B * L.tiles_per_sample;
    L.nch = p1.Cout / 32;
    constexpr int C3 = 32 * MC, WSTAGE = KB * 64 + 4 * C3;
    size_t lds = (size_t)(2 * WSTAGE + 4 * (RDIST + 1) * 2 * WN * 64) * 16 + (size_t)(L.nch * 32 + C3) * 4 + (MASKED ? (size_t)4 * L.nch * 2 * WN * 64 : 0);
    if (lds > 160 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: stages do not fit the LDS");
    L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_h8_kernel<KB, MC, WN, MASKED, RDIST, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((pair_h8_kernel<KB, MC, WN, MASKED, RDIST, OCC>), dim3((unsigned)L.total), dim3(256), lds, st, p1, p2, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

static const char* pair_conv_unsupported(const l2i_conv_params& p, bool first) {
    if (!p.w_hi || !p.y || (first && !p.x)) return "null tensor";
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_y != 0 || p.pad_x != 0 || p.oy_step != 1 || p.ox_step != 1 || p.oy_off || p.ox_off) return "both convs must be 1x1, stride 1, pad 0, dense output";
    if (p.OH != p.H || p.OW != p.W || p.OHf != p.H || p.OWf != p.W) return "output maps have the input's size";
    if (p.in_scale || p.in_mask || p.out_scale || p.noise || p.accumulate || p.res_sub || p.sq_ref || p.sq_out || p.rgb_w || p.rgb_bias || p.rgb_out || p.pool_out || p.pool_idx || p.in_h8 || p.out_f32 || p.w_bstride)
        return "only bias, residual (first conv), sign-plane masks, ReLU and the sign-plane output are fused";
    if (p.act != L2I_ACT_NONE && p.act != L2I_ACT_RELU) return "activation: none or ReLU";
    if (p.out_gain != 1.f) return "out_gain must be 1";
    if ((p.out_mask || p.res_mask) && !p.mask_bits) return "masks must be sign planes (mask_bits)";
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0 || (p.Cout % 32) != 0 || (p.Cin % 32) != 0) return "channel counts must be multiples of 32";
    return nullptr;
}

extern "C" int H8_NAME(l2i_conv1x1_pair_h8)(const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream) {
    if (!first || !second) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_h8: null params");
    const l2i_conv_params &p1 = *first, &p2 = *second;
    if (const char* m = pair_conv_unsupported(p1, true)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (const char* m = pair_conv_unsupported(p2, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p2.x && p2.x != p1.y) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_h8: the second conv reads the first conv's output (second->x must be first->y or NULL)");
    if (p2.Cin != p1.Cout || p2.B != p1.B || p2.H != p1.H || p2.W != p1.W) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_h8: the two convs do not chain");
    if (!p1.residual) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: the first conv carries a residual operand (the trunk)");
    if (p2.residual || p2.res_mask) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: no residual on the second conv");
    if (p1.res_mask && p1.res_mask != p1.out_mask) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: res_mask must be the out_mask plane");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if (!al16(p1.x) || !al16(p1.w_hi) || !al16(p1.y) || !al16(p1.residual) || !al16(p2.w_hi) || !al16(p2.y) || !al16(p1.bias) || !al16(p2.bias))
        return l2i_set_error(L2I_E_ARG, "conv1x1_pair_h8: tensors must be 16-byte aligned");
    const long npix = (long)p1.H * p1.W;
    if ((size_t)(p1.Cout / 8) * npix * 16 >= 0xFFFFFFF0ull || (size_t)(p1.Cin / 8) * npix * 16 >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: one sample must stay below 4 GiB");
    const int KB = p1.Cin / 16, MC = p2.Cout / 32;
    const bool masked = p1.out_mask != nullptr;
    if (p1.res_mask && !masked) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: res_mask without out_mask");
    hipStream_t st = (hipStream_t)stream;
    static const int var_env = getenv("L2I_PAIR_VARIANT") ? atoi(getenv("L2I_PAIR_VARIANT")) : -1;
    const int v = var_env >= 0 ? var_env : variant;      // 0: two 32-pixel rows per wave (256-pixel tiles); 1: one (128-pixel tiles, three blocks per CU)
    const int wn = (v == 1 || (npix % 256) != 0) ? 1 : 2;
    if ((npix % (128 * wn)) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: H * W must be a multiple of 128");
#define L2I_PAIR_CASE(kb, mc)                                                                                                              \
    if (KB == kb && MC == mc) {                                                                                                            \
        if (wn == 2) return masked ? launch_pair<kb, mc, 2, true, (kb >= 8 ? 1 : 2), 2>(p1, p2, st) : launch_pair<kb, mc, 2, false, (kb >= 8 ? 1 : 2), 2>(p1, p2, st); \
        return masked ? launch_pair<kb, mc, 1, true, 2, (mc <= 4 ? 3 : 2)>(p1, p2, st) : launch_pair<kb, mc, 1, false, 2, (mc <= 4 ? 3 : 2)>(p1, p2, st);             \
    }
    L2I_PAIR_CASE(4, 2)
    L2I_PAIR_CASE(8, 4)
    L2I_PAIR_CASE(4, 4)
#undef L2I_PAIR_CASE
    // 256 output channels of the second conv: 128 accumulator registers per 32-pixel row — 128-pixel tiles only
#define L2I_PAIR_CASE1(kb, mc)                                                                                                             \
    if (KB == kb && MC == mc) return masked ? launch_pair<kb, mc, 1, true, 1, 2>(p1, p2, st) : launch_pair<kb, mc, 1, false, 1, 2>(p1, p2, st);
    L2I_PAIR_CASE1(16, 8)
    L2I_PAIR_CASE1(8, 8)
#undef L2I_PAIR_CASE1
    return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: built for (Cin1, Cout2) = (64, 64), (128, 128), (256, 256), (64, 128), (128, 256)");
}
}  // namespace H8_NS
