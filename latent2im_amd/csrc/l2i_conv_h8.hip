// l2i_conv_h8.hip — 1x1 / 3x3 correlations (stride 1 / 2) and the stride-2 transposed 3x3 conv on bf16 tensors in the channel-blocked
// "h8" layout, one v_mfma_f32_32x32x16_bf16 product per MAC, fp32 accumulation (gfx950).  Entry points l2i_conv2d_h8 /
// l2i_conv_transpose2d_h8: the 16-bit path of BASELINE config 5.
//
// Layout.  An activation tensor is [B][C/8][H][W][8] bf16: the 8 channels of a pixel are 16 contiguous bytes.  That is exactly one B
// fragment of the 32x32x16 MFMA (lane (n, half) holds k = 8 half .. 8 half + 7 of pixel n), so
//   * a halo tile goes global -> LDS by DMA (buffer_load_dwordx4 ... lds) one 16-byte pixel slot per lane, ALWAYS 16-byte aligned whatever
//     the tile origin (the fp32 NCHW kernels need aligned 4-pixel row vectors, register staging and a split pass: what bounds
//     conv_bf16x3_pipe_kernel), stride-2 layers with even / odd columns in separate planes of a row (the DMA's source
//     address is per lane, its LDS target lane-linear) so that fragment reads stay unit-stride;
//   * a fragment read is one ds_read_b128 at lane base + immediate, consecutive lanes on consecutive slots (conflict free);
//   * the epilogue exchanges the two lane halves' channel quads with v_permlane32_swap and stores one 16-byte pixel slot per lane and
//     8-channel group: 512 contiguous bytes per half wave.
// Weights: bf16 planes in the LDS image order [Cin/16][tap][half][CoutP][8] (latent2im_amd/conv.py:pack_weight_bf16x3, hi plane), DMA'd one
// kernel ROW (K taps x 2 steps) per phase, double buffered; `w_bstride` != 0: one plane set per sample (the generator's modulated
// convs: the host folds style and demodulation into the weights per sample, like the reference's own formulation, networks.py:234-243).
// K chunk = 32 channels; phase = (chunk, kernel row): wait for the phase's weights, ONE barrier, start the next phase's weight DMA and
// (first phase of a chunk) the next chunk's tile DMA, then K x 2 x WM x WN MFMAs.
// Mapping as everywhere in this library: M = out-channels (A operand), N = 32-pixel row segments (B operand), 4 waves along N.
// No prologue fusions (in_scale / in_mask): masks are applied by the producing epilogues, scales live in the weights.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_epilogue.h"

#include "l2i_h8_common.h"

namespace H8_NS {

namespace g8 {
template <int WN, int K, int S, int TR, int KS_ = 2> struct Geo {
    static constexpr int TH = 4 * WN;
    static constexpr bool GATHER = (K == 1 && S == 2);                    // 1x1 stride 2: only even rows / columns are staged
    static constexpr int ROWS = TR ? TH + 1 : (GATHER ? TH : (TH - 1) * S + K);
    static constexpr int RSTEP = GATHER ? 2 : 1;                          // image rows between staged rows
    static constexpr int COLS = TR ? 33 : (GATHER ? 32 : 31 * S + K);     // staged pixel slots per row
    static constexpr int CSTEP = GATHER ? 2 : 1;
    static constexpr bool SPLIT = (S == 2 && !GATHER);                    // even columns first, odd columns from RPH on
    static constexpr int RPH = (COLS + 1) / 2;
    static constexpr int RP = COLS;
    static constexpr int KS = KS_;                                        // 16-channel MFMA steps per chunk (1: low-Cin layers, half the LDS, four blocks per CU)
    static constexpr int NH = 2 * KS;                                     // 8-channel groups per chunk
    static constexpr int CK = 16 * KS;
    static constexpr int HSTRIDE = ROWS * RP;                             // slots of one 8-channel group
    static constexpr int IN_SLOTS = NH * HSTRIDE;
    static constexpr int NPW = ((IN_SLOTS + 63) / 64 + 3) / 4;            // 1 KiB DMA pieces per wave and chunk (every wave issues NPW: the
    static constexpr int IN_STAGE = 4 * NPW * 64;                         // pieces past the tile land in padding behind it)
};
}

struct H8Launch {
    int tiles_x, tiles_y, mblocks, total, nchunks;
    int vec_epi;                       // fp32 NCHW output: the LDS-transposed epilogue with 16-byte accesses applies
    int lean_epi;                      // h8 output without per-pixel operand maps (no out_mask / residual / accumulate / sq_ref): the lean epilogue
    int rgb;                           // [r5] lean epilogue + ToRGB: the block holds every output channel of its pixels and also writes their 3-channel image (l2i.h: rgb_w)
};

// ---- epilogue, h8 output ----------------------------------------------------------------------------------------------------------
// acc[m][n][r]: channel m0 + 32 m + (r & 3) + 8 (r >> 2) + 4 half, pixel (oy0 + wave WN + n, ox0 + j).  Register quad q = r >> 2 of a lane
// half holds channels 4 half .. 4 half + 3 of 8-channel group q; v_permlane32_swap on the quads of groups (2 pr, 2 pr + 1) gives lanes
// 0-31 all eight channels of group 2 pr and lanes 32-63 those of group 2 pr + 1 (guide T21), i.e. lane (half, j) finishes pixel j of group
// 2 pr + half: fused terms are applied on whole 16-byte slots, then one 16-byte store.
//   epi(a) = act( a * out_scale[b,co] * (out_mask > 0) + noise * noise_w + bias[co] + R * (res_mask > 0) ) * out_gain,
//   R = res_sub ? res_coef * res_coef_dev[0] * (residual - res_sub) : residual         (include/l2i.h; all operand maps in h8 bf16)
struct H8Out {
    const l2i_conv_params& p;
    int b, cg_out;                       // sample, 8-channel groups of the output tensor
    float rc;                            // res_coef * res_coef_dev[0]
    float sq;                            // running sum (y - sq_ref)^2
    __device__ __forceinline__ void finish(float (&v)[8], int grp, int oy, int ox, float nz) {
        // v: eight channels 8 grp .. 8 grp + 7 of pixel (oy, ox) (output-tensor coordinates), raw accumulators
        const int co0 = grp * 8;
        const size_t slot = (((size_t)b * cg_out + grp) * p.OHf + oy) * p.OWf + ox;
        const u32x4* y4 = reinterpret_cast<const u32x4*>(p.y);
        // every operand of the slot is requested before the first one is used: the first version loaded and consumed them one by one, up to five
        // dependent memory round trips per slot and eight slots per wave
        u32x4 q_mask, q_res, q_sub, q_rmask, q_old;
        float4 s0, s1, b0, b1;
        unsigned mb = 0u, rmb = 0u;                                // [r6] mask_bits: the masks are sign planes, one byte per slot
        if (p.out_mask) { if (p.mask_bits) mb = reinterpret_cast<const uint8_t*>(p.out_mask)[slot]; else q_mask = reinterpret_cast<const u32x4*>(p.out_mask)[slot]; }
        if (p.residual) q_res = reinterpret_cast<const u32x4*>(p.residual)[slot];
        if (p.res_sub) q_sub = reinterpret_cast<const u32x4*>(p.res_sub)[slot];
        if (p.res_mask) {
            if (p.mask_bits) rmb = (p.res_mask == p.out_mask) ? mb : (unsigned)reinterpret_cast<const uint8_t*>(p.res_mask)[slot];
            else q_rmask = reinterpret_cast<const u32x4*>(p.res_mask)[slot];
        }
        if (p.accumulate) q_old = y4[slot];
        if (p.out_scale) { s0 = *reinterpret_cast<const float4*>(p.out_scale + (size_t)b * p.Cout + co0); s1 = *reinterpret_cast<const float4*>(p.out_scale + (size_t)b * p.Cout + co0 + 4); }
        if (p.bias) { b0 = *reinterpret_cast<const float4*>(p.bias + co0); b1 = *reinterpret_cast<const float4*>(p.bias + co0 + 4); }
        if (p.out_scale) { v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w; v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w; }
        if (p.out_mask) {
            if (p.mask_bits) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= ((mb >> e) & 1u) ? p.mask_pos : p.mask_neg;
            } else {
                float m[8];
                h8_unpack(q_mask, m);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= m[e] > 0.f ? p.mask_pos : p.mask_neg;   // (1, 0): a ReLU mask; (1, 0.2) / (sqrt2, 0.2 sqrt2): leaky ReLU'
            }
        }
        if (p.bias) { v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w; }
        if (p.noise) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += nz;
        }
        if (p.residual) {
            float r[8];
            h8_unpack(q_res, r);
            if (p.res_sub) {
                float s[8];
                h8_unpack(q_sub, s);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = rc * (r[e] - s[e]);
            }
            if (p.res_mask) {
                if (p.mask_bits) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = ((rmb >> e) & 1u) ? r[e] : 0.f;
                } else {
                    float m[8];
                    h8_unpack(q_rmask, m);
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = m[e] > 0.f ? r[e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (p.act == L2I_ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * p.act_slope) * p.act_gain;
        } else if (p.act == L2I_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (p.out_gain != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.out_gain;
        }
        if (p.accumulate) {
            float o[8];
            h8_unpack(q_old, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += o[e];
        }
        const u32x4 out = {cvt_pk_bf16_h8(v[0], v[1]), cvt_pk_bf16_h8(v[2], v[3]), cvt_pk_bf16_h8(v[4], v[5]), cvt_pk_bf16_h8(v[6], v[7])};
#ifdef L2I_H8_ABLATE_STORE                                 // timing ablation: the epilogue's arithmetic without its global store
        if (out.x == 0x12345678u && out.y == 0x9abcdef0u)
#endif
        reinterpret_cast<u32x4*>(p.y)[slot] = out;
        if (p.mask_out) p.mask_out[slot] = (uint8_t)h8_sign_byte(out);
        if (p.sq_ref) {                                            // ContentLoss value of a VGG tap on the ROUNDED output (what the next layer reads)
            float rf[8], w[8];
            h8_unpack(reinterpret_cast<const u32x4*>(p.sq_ref)[slot], rf);
            h8_unpack(out, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = w[e] - rf[e]; sq += d * d; }
        }
    }
};

// RELU_IN: pro(x) = max(x, 0) on the B fragments (VGG-19: a conv reads the PRE-ReLU tap of the layer below, which is what the ContentLoss
// and the backward masks need in HBM): four v_pk_max_i16 per fragment — a negative bf16 is a negative int16, so the integer max with 0 is
// the ReLU (and -0 -> +0) — beside the bf16 MFMAs, whose pipe the VALU does not share
// Measured, not kept (round 3, tools/probes/h8_ablate*.sh): a "deep" variant for the low-Cin 3x3 layers — three tile stages requested two chunks ahead,
// a chunk's whole 3x3 weight slice per stage with ONE barrier per chunk, tile DMA issued by waves 0-1 and weight DMA by waves 2-3 so that a wave's
// in-order vmcnt never makes an L2-hit weight slice wait behind an HBM tile.  72 KB of LDS = two blocks per CU instead of four: 9-15 % SLOWER
// on every layer (64 -> 64 @1024^2 0.94 against 0.83 ms).  The timing ablations say why nothing memory-side helps: with no DMA at all the launch
// still takes 0.64 ms, with no DMA and no MFMA 0.37, with the stores gone too 0.30 — the parts add instead of overlapping, and four co-resident
// blocks hide more of that than a deeper pipeline in two.  Also measured, no effect (inside +-1 %): the same wave roles at four blocks per CU (two tile
// stages: a tile wave waits for its tile only at the chunk's first phase, a weight wave never behind a tile), starting the four first-generation
// blocks of a CU a quarter of a block time apart (s_sleep sweep 0 ... 12 x 2048 cycles per slot), and removing every barrier (timing only).
// [r6] Measured, not kept (profiles/r06_h8_deep_ring_ab.txt): a DEEP ring for the mid-size launches (<= 2048 blocks: ResNet-50's layers at batch 8, the <= 64^2
// layers of G / D: 35 - 80 us per launch whatever their work) — four weight stages requested three phases ahead, three tile stages two chunks ahead, one counted
// vmcnt per head, null-descriptor tail, two blocks per CU.  Element-exact, and 0 - 15 % SLOWER on the 3x3 shapes, up to 1.8x slower on the >= 1024-block 1x1s; only the
// 256-block 1x1s gain (38 -> 31 us).  The ablations of the same shapes say why: with no DMA at all 256 -> 256 @64^2 still takes 31 of 38 us (15 us of matrix time),
// with no MFMA 28 — at one to two waves per SIMD the parts of a block add, and prefetch depth removes none of them.
// Also measured on those shapes, no gain: 4-row tiles (WN = 1: twice the blocks; 128 -> 128 @128^2 41.0 -> 44.6 us, 256 -> 256 @64^2 42.5 -> 42.9) and 32-channel chunks.
// Finer ablations of the no-DMA / no-MFMA / no-store skeleton (profiles/r03_h8_conv3x3_timing_ablations.txt): the epilogue is 0.25 ms of the
// 0.83 ms launch (0.12 of it the stores), the fragment reads + barriers of the K loop another 0.2.
// What all of these have in common (tools/probes/h8_clocks.sh, profiles/r03_h8_power_clocks.txt): while this kernel runs the package sits AT its
// 1400 W power limit and the shader clock is throttled from 2.40 to 1.80-1.97 GHz (the fp32 kernels run at 1.19-1.34 kW and 2.40 GHz, the 1x1 layers
// at 1.21 kW).  Under a power cap time follows energy: every part that is removed gives its share back, nothing overlaps "for free", and
// re-arranging the same work (deeper pipeline, roles, stagger, a 16-row register tile with half the weight traffic: 0.790 against 0.795 ms) changes
// nothing.  What moves it is less energy per output: fewer bytes (this path), fewer VALU instructions (the lean epilogue), fewer MFMA passes.
// [r5] Measured again with a 128-channel block (WM = 4, two blocks per CU: half the B-fragment reads and half the L2 -> LDS input traffic per MFMA): 256 -> 256
// @128^2 0.1301 against 0.1306 ms, 512 -> 512 @64^2 0.1286 / 0.1291, 128 -> 128 @512^2 0.636 / 0.617, 512 -> 512 @32^2 0.0735 / 0.0461 (too few blocks): not kept.
// EXTRA [r5]: the lean epilogue's optional outputs (ToRGB image, ContentLoss sum) live in their own instantiations of the 3x3 stride-1
// kernel (a fused 2x2 max-pool of the output was built here too, bit-identical to the pool kernel, and measured: c5 223.4 - 223.9 images/s with, 225.2 - 225.8
// without — the pooled values' registers slow the two big VGG launches by more than the pool pass costs: removed): their registers (24 weights + 6 sums) otherwise cost the plain launches — 58 of the c5 step's — spills at four blocks per CU
template <int WM, int WN, int K, int S, int TR, bool OUT32, bool RELU_IN = false, int KS = 2, bool EXTRA = false>
__global__ __launch_bounds__(256, ((KS == 1 || K == 1) && TR == 0) ? (EXTRA ? 3 : 4) : 2) void conv_h8_kernel(const l2i_conv_params p, const H8Launch L) {
    using G = g8::Geo<WN, K, S, TR, KS>;
    constexpr int NACC = TR ? 4 : 1;
    constexpr int DMIN = (TR == 1) ? -1 : 0;
    constexpr int BM = WM * 32;
    constexpr int WSLOTS = K * G::KS * 2 * BM;             // slots per phase: K taps x KS steps x 2 halves x BM channels
    constexpr int WPIECES = WSLOTS / 64;
    constexpr int TW = 4;                                  // waves that issue tile pieces / weight pieces
    constexpr int WPW = (WPIECES + TW - 1) / TW;
    constexpr int NPT = 4 * G::NPW / TW;                   // tile pieces per issuing wave and chunk
    constexpr int IN_STAGE = G::IN_STAGE;
    constexpr int NST = 2;                                 // tile stages
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4* const in_st = smem4;                            // NST stages x [group][row][slot]
    u32x4* const w_st = smem4 + NST * IN_STAGE;            // 2 stages x [tap][step][half][channel]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // XCD-aware block order: the channel blocks of a pixel tile share an XCD (and its L2)
    int w = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (w >= L.total) return;
    const int mblk = __builtin_amdgcn_readfirstlane(w % L.mblocks); w /= L.mblocks;
    const int tx = __builtin_amdgcn_readfirstlane(w % L.tiles_x); w /= L.tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(w % L.tiles_y); w /= L.tiles_y;
    const int b = __builtin_amdgcn_readfirstlane(w), m0 = mblk * BM, oy0 = ty * G::TH, ox0 = tx * 32;
    const int iy0 = TR ? oy0 + DMIN : oy0 * S - p.pad_y;
    const int ix0 = TR ? ox0 + DMIN : ox0 * S - p.pad_x;

    // ---- descriptors (one sample's tensor: 32-bit offsets) ----
    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * 16);                       // bytes of one 8-channel group
    const unsigned in_bytes = (unsigned)(p.Cin / 8) * plane_b;
    const char* xb = reinterpret_cast<const char*>(p.x) + (size_t)b * in_bytes;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, in_bytes, 0x00020000);
    const unsigned wpl_bytes = (unsigned)((size_t)(p.Cin / 16) * K * K * 2 * p.CoutP * 16);
    const char* wb = reinterpret_cast<const char*>(p.w_hi) + (size_t)b * (size_t)p.w_bstride;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, wpl_bytes, 0x00020000);

    // ---- tile DMA: piece q = wave + 4 t covers LDS slots [64 q, 64 q + 64) of the stage; slot -> (group, row, column) ----
    const int tw = wave, tw_u = wave_u;
    unsigned ivoff[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int sl = (tw + TW * t) * 64 + lane;
        ivoff[t] = in_bytes;                                                           // out of range: the DMA writes zeros (= the padding)
        if (sl < G::IN_SLOTS) {
            const int nh = sl / G::HSTRIDE, r2 = sl - nh * G::HSTRIDE;
            const int row = r2 / G::RP, rc = r2 - row * G::RP;
            const int col = G::SPLIT ? (rc < G::RPH ? 2 * rc : 2 * (rc - G::RPH) + 1) : rc * G::CSTEP;
            const int gy = iy0 + row * G::RSTEP, gx = ix0 + col;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ivoff[t] = (unsigned)nh * plane_b + (unsigned)(gy * p.W + gx) * 16u;
        }
    }
    auto dma_in = [&](int chunk, int stage) {
#ifdef L2I_H8_ABLATE_TILE                                  // timing ablation: no tile traffic (the MFMAs run on whatever the LDS holds)
        return;
#endif
        const unsigned soff = (unsigned)chunk * G::NH * plane_b;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(in_st + stage * IN_STAGE);
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(ivoff[t]), "s"(rs_x), "s"(__builtin_amdgcn_readfirstlane(lds0 + (tw_u + TW * t) * 1024)), "s"(soff)
                         : "memory");
        }
    };
    // ---- weight DMA: piece q covers slots [64 q, 64 q + 64) of the phase ([tap][step][half][channel]) ----
    unsigned wvoff[WPW];
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int sl = ((tw + TW * t) * 64 + lane) % WSLOTS;
        const int r = sl / BM, i = sl - r * BM;                            // r = (tap * KS + step) * 2 + half
        const int hf = r & 1, st = (r >> 1) % G::KS, tap = (r >> 1) / G::KS;
        wvoff[t] = (unsigned)((((st * K * K + tap) * 2 + hf) * p.CoutP + m0 + i) * 16);
    }
    auto dma_w = [&](int chunk, int ky, int stage_slot) {       // stage_slot: first LDS slot of the phase's weights (relative to w_st)
#ifdef L2I_H8_ABLATE_W                                     // timing ablation: no weight traffic
        return;
#endif
        const unsigned soff = (unsigned)((((size_t)chunk * G::KS * K * K + ky * K) * 2) * p.CoutP * 16);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(w_st + stage_slot);
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int q = tw_u + TW * t;                   // wave-uniform: a scalar branch
            if (q < WPIECES) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(wvoff[t]), "s"(rs_w), "s"(__builtin_amdgcn_readfirstlane(lds0 + q * 1024)), "s"(soff)
                             : "memory");
            }
        }
    };

    f32x16 acc[NACC][WM][WN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][m][n][r] = 0.f;

    // fragment bases (slots): lane (half, j) + compile-time / wave-uniform offsets
    constexpr int rstep_out = G::GATHER ? 1 : S;           // staged rows between consecutive output rows
    const int bbase = half * G::HSTRIDE + wave * WN * rstep_out * G::RP + j;
    const int abase = half * BM + j;

    auto mfma_phase = [&](int in_stage, int w_slot, auto ky_t) {     // w_slot: first LDS slot of the phase's weights (relative to w_st)
        constexpr int ky = decltype(ky_t)::value;
        constexpr int PADT = (TR == 2) ? 1 : 0;
        constexpr int py = (ky + PADT) & 1;
        constexpr int rowoff = TR ? (py + PADT - ky) / 2 - DMIN : (G::GATHER ? 0 : ky);
        const u32x4* ih = in_st + in_stage * IN_STAGE + bbase + rowoff * G::RP;
        const u32x4* wh = w_st + w_slot + abase;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            // transposed: tap kx feeds output parity px = (kx + pad) & 1 from input column s + dx, dx = (px + pad - kx) / 2; staged column dx - DMIN
            const int c = TR ? (((kx + PADT) & 1) + PADT - kx) / 2 - DMIN : kx;
            const int coloff = G::GATHER ? 0 : (G::SPLIT ? (c & 1) * G::RPH + (c >> 1) : c);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                bf16x8 af[WM], bf[WN];
#pragma unroll
#ifdef L2I_H8_ABLATE_LDSREAD                               // timing ablation: no fragment reads
                for (int m = 0; m < WM; ++m) af[m] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kx, (unsigned)ks, (unsigned)m, (unsigned)lane});
#else
                for (int m = 0; m < WM; ++m) af[m] = __builtin_bit_cast(bf16x8, wh[((kx * G::KS + ks) * 2) * BM + m * 32]);
#endif
#pragma unroll
                for (int n = 0; n < WN; ++n) {
#ifdef L2I_H8_ABLATE_LDSREAD
                    u32x4 raw = u32x4{(unsigned)n, (unsigned)kx, (unsigned)j, 7u};
#else
                    u32x4 raw = ih[ks * 2 * G::HSTRIDE + n * rstep_out * G::RP + coloff];
#endif
                    if constexpr (RELU_IN) {
                        asm("v_pk_max_i16 %0, %0, 0" : "+v"(raw.x)); asm("v_pk_max_i16 %0, %0, 0" : "+v"(raw.y));
                        asm("v_pk_max_i16 %0, %0, 0" : "+v"(raw.z)); asm("v_pk_max_i16 %0, %0, 0" : "+v"(raw.w));
                    }
                    bf[n] = __builtin_bit_cast(bf16x8, raw);
                }
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
#ifdef L2I_H8_ABLATE_MFMA                                  // timing ablation (tools/probes/h8_ablate.sh): fragments are read, no matrix work
                        acc[TR ? py * 2 + ((kx + PADT) & 1) : 0][m][n][0] += __builtin_bit_cast(f32x4_, af[m])[0] * __builtin_bit_cast(f32x4_, bf[n])[0];
#else
                        acc[TR ? py * 2 + ((kx + PADT) & 1) : 0][m][n] = H8_MFMA(af[m], bf[n], acc[TR ? py * 2 + ((kx + PADT) & 1) : 0][m][n], 0, 0, 0);
#endif
            }
        }
    };

    // ---- pipeline ----
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    {
    const int nphases = L.nchunks * K;
    dma_w(0, 0, 0);
    dma_in(0, 0);
    auto phase_head = [&](int ch, int ky, bool more) {
        const int ph = ch * K + ky;
        // this phase's weights (and, first phase of a chunk, the chunk's tile) have landed; the next chunk's tile, issued AFTER the
        // weights of phase (ch, 1), stays in flight across that phase's barrier
        if (K > 1 && ky == 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef L2I_H8_ABLATE_BARRIER
        __syncthreads();
#endif
        if (ph + 1 < nphases) dma_w(ky + 1 < K ? ch : ch + 1, ky + 1 < K ? ky + 1 : 0, ((ph + 1) & 1) * WSLOTS);
        if (ky == 0 && more) dma_in(ch + 1, (ch + 1) & 1);
    };
    for (int ch = 0; ch < L.nchunks; ++ch) {
        const bool more = ch + 1 < L.nchunks;
        phase_head(ch, 0, more);
        mfma_phase(ch & 1, ((ch * K) & 1) * WSLOTS, K0());
        if constexpr (K == 3) {
            phase_head(ch, 1, more);
            mfma_phase(ch & 1, ((ch * K + 1) & 1) * WSLOTS, K1());
            phase_head(ch, 2, more);
            mfma_phase(ch & 1, ((ch * K + 2) & 1) * WSLOTS, K2());
        }
    }
    }

    if constexpr (OUT32) {
        static_assert(!OUT32 || TR == 0, "fp32 NCHW output: correlations only");
        __syncthreads();                                   // the stages become the epilogue's transpose strips
        l2i_epilogue_32x32<WM, WN>(p, acc[0], reinterpret_cast<float*>(smem4), b, m0, oy0, ox0, L.vec_epi != 0);
#ifdef L2I_H8_ABLATE_EPI                                   // timing ablation: no epilogue (one dword per lane keeps the accumulators alive)
    } else if (true) {
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[a][m][n][r];
        if (t == 12345.678f) reinterpret_cast<float*>(p.y)[tid] = t;
#endif
    } else if (L.lean_epi) {
        // Lean epilogue (forward layers and plain gradient convs: per-channel vectors and the per-pixel noise only).  On the low-Cin layers the
        // general epilogue below was as many VALU instructions as the K loop was MFMA cycles (7 VALU per MFMA on 64 -> 64 @1024^2, SQ counters:
        // eight `finish` calls per wave, each with its own 64-bit slot arithmetic, operand switches and dependent loads) and the two do not
        // overlap.  Here: one base pointer per wave, the channel vectors of a group fetched once for its WN rows, noise fetched
        // up front, no per-slot branches beyond the store predicate.
        const int cg_out = p.Cout / 8;
        const size_t plane = (size_t)p.OHf * p.OWf;
        const int sx = ox0 + j;
        const int g0 = (m0 >> 3) + half;                               // this lane's group for (m, pr) = (0, 0); (m, pr) adds 4 m + 2 pr
        u32x4* const yb = reinterpret_cast<u32x4*>(p.y) + ((size_t)b * cg_out + g0) * plane;
        float nz[WN][NACC];
        auto pix = [&](int n, int a, unsigned& off) -> bool {          // store predicate and slot offset of the pixel inside a group plane
            const int t = oy0 + wave * WN + n;
            const int oy = TR ? 2 * t + (a >> 1) : t + p.oy_off, ox = TR ? 2 * sx + (a & 1) : sx + p.ox_off;
            const bool ok = TR ? (oy < p.OHf && ox < p.OWf) : (t < p.OH && sx < p.OW);
            off = ok ? (unsigned)oy * (unsigned)p.OWf + (unsigned)ox : 0u;
            return ok;
        };
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                unsigned off;
                pix(n, a, off);
                nz[n][a] = p.noise ? p.noise[(size_t)b * plane + off] * p.noise_w : 0.f;
            }
        constexpr int NQ = WM * 2;
        // [r5] ToRGB on the way out (networks.py:349-358 on the 512^2 / 1024^2 StyledConv outputs): the block holds all Cout <= 32 WM channels of its
        // pixels, so rgb[o] = rgb_bias[o] + sum_c rgb_w[b, o, c] * y[c] is 24 FMAs per stored slot and one lane-half exchange per pixel instead of a
        // pass that reads the whole feature map again (fp32 values before the 16-bit rounding of the store)
        float racc[WN][NACC][3];
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int a = 0; a < NACC; ++a) racc[n][a][0] = racc[n][a][1] = racc[n][a][2] = 0.f;
        float sq_lean = 0.f;
        const float gpos = (p.act == L2I_ACT_LRELU ? p.act_gain : 1.f) * p.out_gain;
        const float gneg = p.act == L2I_ACT_LRELU ? p.act_slope * p.act_gain * p.out_gain : (p.act == L2I_ACT_RELU ? 0.f : p.out_gain);
        const bool plain_relu = p.act == L2I_ACT_RELU && p.out_gain == 1.f, identity = p.act == L2I_ACT_NONE && p.out_gain == 1.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = q >> 1, pr = q & 1;
            const int co0 = (g0 + 4 * m + 2 * pr) * 8;
            const bool gok = co0 < p.Cout;
            const int cc = gok ? co0 : 0;
            // scale (1 when absent) and bias (0 when absent) as four channel pairs: g * s + b is one packed FMA per pair
            f32x2_ sc2[4] = {{1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}}, bs2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            if (p.out_scale) {
                const float* sp = p.out_scale + (size_t)b * p.Cout + cc;
                const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                sc2[0] = f32x2_{s0.x, s0.y}; sc2[1] = f32x2_{s0.z, s0.w}; sc2[2] = f32x2_{s1.x, s1.y}; sc2[3] = f32x2_{s1.z, s1.w};
            }
            if (p.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cc), b1 = *reinterpret_cast<const float4*>(p.bias + cc + 4);
                bs2[0] = f32x2_{b0.x, b0.y}; bs2[1] = f32x2_{b0.z, b0.w}; bs2[2] = f32x2_{b1.x, b1.y}; bs2[3] = f32x2_{b1.z, b1.w};
            }
            u32x4* const yq = yb + (size_t)(4 * m + 2 * pr) * plane;
            float rw[3][8];
            if (EXTRA && L.rgb) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float* wp = p.rgb_w + ((size_t)b * 3 + o) * p.Cout + cc;
                    const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
                    rw[o][0] = gok ? w0.x : 0.f; rw[o][1] = gok ? w0.y : 0.f; rw[o][2] = gok ? w0.z : 0.f; rw[o][3] = gok ? w0.w : 0.f;
                    rw[o][4] = gok ? w1.x : 0.f; rw[o][5] = gok ? w1.y : 0.f; rw[o][6] = gok ? w1.z : 0.f; rw[o][7] = gok ? w1.w : 0.f;
                }
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    float g[8];
                    h8_gather(acc[a][m][n], pr, half, g);
                    f32x2_ v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = f32x2_{g[2 * i], g[2 * i + 1]} * sc2[i] + bs2[i];
                    if (p.noise) {
                        const f32x2_ z = {nz[n][a], nz[n][a]};
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += z;
                    }
                    if (plain_relu) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f); }
                    } else if (!identity) {                     // leaky ReLU / gains: max(v * gpos, v * gneg), gpos >= gneg >= 0
                        const f32x2_ gp2 = {gpos, gpos}, gn2 = {gneg, gneg};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x2_ hi = v[i] * gp2, lo = v[i] * gn2;
                            v[i].x = fmaxf(hi.x, lo.x); v[i].y = fmaxf(hi.y, lo.y);
                        }
                    }
                    g[0] = v[0].x; g[1] = v[0].y; g[2] = v[1].x; g[3] = v[1].y; g[4] = v[2].x; g[5] = v[2].y; g[6] = v[3].x; g[7] = v[3].y;
                    if (EXTRA && L.rgb) {
#pragma unroll
                        for (int o = 0; o < 3; ++o)
#pragma unroll
                            for (int e = 0; e < 8; ++e) racc[n][a][o] += g[e] * rw[o][e];
                    }
                    const u32x4 out = {cvt_pk_bf16_h8(g[0], g[1]), cvt_pk_bf16_h8(g[2], g[3]), cvt_pk_bf16_h8(g[4], g[5]), cvt_pk_bf16_h8(g[6], g[7])};
                    unsigned off;
                    const bool ok = pix(n, a, off);
#ifdef L2I_H8_ABLATE_STORE
                    if (out.x == 0x12345678u && out.y == 0x9abcdef0u)
#endif
                    if (ok && gok) yq[off] = out;
                    if (!EXTRA && p.mask_out && ok && gok) p.mask_out[((size_t)b * cg_out + g0 + 4 * m + 2 * pr) * plane + off] = (uint8_t)h8_sign_byte(out);
                    if (EXTRA && p.sq_ref && ok && gok) {          // [r5] ContentLoss value of a VGG tap on the ROUNDED output, as in the general epilogue
                        float rf[8], w[8];
                        h8_unpack(reinterpret_cast<const u32x4*>(p.sq_ref)[((size_t)b * cg_out + g0 + 4 * m + 2 * pr) * plane + off], rf);
                        h8_unpack(out, w);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = w[e] - rf[e]; sq_lean += d * d; }
                    }
                }
            }
        }
        if (EXTRA && p.sq_ref) {                                   // one atomic per block into L2I_SQ_SLOTS slots (as below)
            float sq = sq_lean;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem4);
            if (lane == 0) red[wave] = sq;
            __syncthreads();
            if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (red[0] + red[1]) + (red[2] + red[3]));
        }
        if (EXTRA && L.rgb) {
            // the two lane halves hold the partial sums of different channel groups of the SAME pixel: exchange, add, lanes 0-31 store three fp32 rows
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    unsigned off;
                    const bool ok = pix(n, a, off);
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(racc[n][a][o]), __float_as_uint(racc[n][a][o]), false, false);
                        const float sum = __uint_as_float(r[0]) + __uint_as_float(r[1]) + p.rgb_bias[o];
                        if (ok && half == 0) p.rgb_out[((size_t)b * 3 + o) * plane + off] = sum;
                    }
                }
        }
    } else {
        H8Out o{p, b, p.Cout / 8, p.res_sub ? p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) : 1.f, 0.f};
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int t = oy0 + wave * WN + n, sx = ox0 + j;
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int oy = TR ? 2 * t + (a >> 1) : t + p.oy_off, ox = TR ? 2 * sx + (a & 1) : sx + p.ox_off;
                const bool pok = TR ? (oy < p.OHf && ox < p.OWf) : (t < p.OH && sx < p.OW);
                float nz = 0.f;
                if (pok && p.noise) nz = p.noise[((size_t)b * p.OHf + oy) * p.OWf + ox] * p.noise_w;
#pragma unroll
                for (int m = 0; m < WM; ++m) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float g[8];
                        h8_gather(acc[a][m][n], pr, half, g);
                        const int grp = (m0 >> 3) + 4 * m + 2 * pr + half;
                        if (pok && grp * 8 < p.Cout) o.finish(g, grp, oy, ox, nz);
                    }
                }
            }
        }
        if (p.sq_ref) {                                            // one atomic per block into L2I_SQ_SLOTS slots
            float sq = o.sq;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem4);
            if (lane == 0) red[wave] = sq;
            __syncthreads();
            if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (red[0] + red[1]) + (red[2] + red[3]));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int K, int S, int TR, bool OUT32, bool RELU_IN = false, int KS = 2>
static int launch_h8(const l2i_conv_params& p, hipStream_t st) {
    using G = g8::Geo<WN, K, S, TR, KS>;
    constexpr int BM = WM * 32;
    constexpr int WSLOTS = K * G::KS * 2 * BM;
    constexpr bool CAN_EXTRA = (K == 3 && S == 1 && TR == 0 && !OUT32);       // the instantiations that exist with the lean epilogue's optional outputs
    H8Launch L;
    L.tiles_x = (p.OW + 31) / 32;
    L.tiles_y = (p.OH + G::TH - 1) / G::TH;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d_h8: grid too large");
    L.total = (int)total;
    L.nchunks = p.Cin / G::CK;
    L.vec_epi = (OUT32 && l2i_epilogue_vec_ok(p)) ? 1 : 0;
    static const int lean_env = getenv("L2I_H8_LEAN") ? atoi(getenv("L2I_H8_LEAN")) : 1;
    // identity / ReLU / leaky ReLU as max(g * gpos, g * gneg) needs 0 <= gneg <= gpos
    const bool gains_ok = p.out_gain > 0.f && (p.act != L2I_ACT_LRELU || (p.act_gain > 0.f && p.act_slope >= 0.f && p.act_slope <= 1.f));
    bool lean = !OUT32 && lean_env && !p.out_mask && !p.residual && !p.accumulate && gains_ok && (size_t)p.OHf * p.OWf < 0xFFFFFFFFull;
    if (p.sq_ref && !CAN_EXTRA) lean = false;              // the ContentLoss sum rides on the lean epilogue of the EXTRA instantiations only; elsewhere: the general epilogue
    L.lean_epi = lean ? 1 : 0;
    const bool extra = CAN_EXTRA && lean && (p.rgb_w || p.sq_ref);
    L.rgb = p.rgb_w ? 1 : 0;
    // (Cout % 32: the epilogue fetches the eight rgb_w values of EVERY channel group of the block's 32- / 64-channel tile before it masks the groups
    //  past Cout — with Cout = 40 or 48 those loads would run past the sample's row, and past the buffer for the last sample; the generator's layers
    //  have 32 and 64 channels)
    if (L.rgb && !(extra && L.mblocks == 1 && p.rgb_bias && p.rgb_out && (p.Cout % 32) == 0 && (((uintptr_t)p.rgb_w) % 16) == 0))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: rgb_w needs a 3x3 stride-1 launch on the lean h8 epilogue (no per-pixel operand maps), every output channel in one block (Cout = 32 or 64), rgb_bias / rgb_out, a 16-byte aligned rgb_w");
    size_t lds = (size_t)(2 * G::IN_STAGE + 2 * WSLOTS) * 16;
    if (OUT32 && lds < 4 * 32 * 64 * sizeof(float)) lds = 4 * 32 * 64 * sizeof(float);
    if (lds > 160 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: tile does not fit the LDS");
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    if constexpr (CAN_EXTRA) {
        if (extra) {
            L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h8_kernel<WM, WN, K, S, TR, OUT32, RELU_IN, KS, true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((conv_h8_kernel<WM, WN, K, S, TR, OUT32, RELU_IN, KS, true>), dim3(grid), dim3(256), lds, st, p, L);
            L2I_CHECK_LAUNCH();
            return L2I_OK;
        }
    }
    L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h8_kernel<WM, WN, K, S, TR, OUT32, RELU_IN, KS>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((conv_h8_kernel<WM, WN, K, S, TR, OUT32, RELU_IN, KS>), dim3(grid), dim3(256), lds, st, p, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

static int h8_common_checks(const l2i_conv_params& p, const char* who) {
    if (!p.x || !p.w_hi || !p.y) return l2i_set_error(L2I_E_ARG, "conv h8: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, true, false, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0) return l2i_set_error(L2I_E_ARG, "conv h8: non-positive dimension");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv h8: CoutP must be Cout rounded up to 32");
    const bool relu_in = p.in_mask && (const void*)p.in_mask == (const void*)p.x && p.mask_pos == 1.f && p.mask_neg == 0.f && !p.out_mask;
    if (p.in_scale || (p.in_mask && !relu_in))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv h8: no prologue fusions (scales live in the weights, masks in the producing epilogue) except ReLU-on-load (in_mask == x)");
    if ((p.Cin % 16) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv h8: Cin must be a multiple of 16");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if (!al16(p.x) || !al16(p.w_hi) || !al16(p.y) || !al16(p.residual) || (!p.mask_bits && (!al16(p.res_mask) || !al16(p.out_mask))) || !al16(p.res_sub) || !al16(p.sq_ref) ||
        !al16(p.bias) || !al16(p.out_scale) || (p.w_bstride % 16) != 0)
        return l2i_set_error(L2I_E_ARG, "conv h8: tensors must be 16-byte aligned");
    if ((size_t)(p.Cin / 8) * p.H * p.W * 16 >= 0xFFFFFFF0ull || (size_t)(p.Cin / 16) * p.KH * p.KW * 2 * p.CoutP * 16 >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv h8: one sample / the weight planes must stay below 4 GiB (32-bit buffer offsets)");
    if (p.res_sub && !p.residual) return l2i_set_error(L2I_E_ARG, "conv h8: res_sub needs residual");
    if (p.mask_out && (p.out_f32 || p.rgb_w || p.sq_ref)) return l2i_set_error(L2I_E_UNSUPPORTED, "conv h8: mask_out rides on the plain h8 epilogues (no fp32 output, ToRGB or ContentLoss sum in the same launch)");
    if (p.mask_bits && (p.out_f32 || !(p.out_mask || p.res_mask))) return l2i_set_error(L2I_E_ARG, "conv h8: mask_bits describes out_mask / res_mask of an h8-output launch");
    if ((p.sq_ref != nullptr) != (p.sq_out != nullptr)) return l2i_set_error(L2I_E_ARG, "conv h8: sq_ref and sq_out go together");
    (void)who;
    return L2I_OK;
}

extern "C" int H8_NAME(l2i_conv2d_h8)(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d_h8: null params");
    const l2i_conv_params& p = *pp;
    if (int rc = h8_common_checks(p, "conv2d_h8")) return rc;
    const bool k1 = (p.KH == 1 && p.KW == 1 && p.pad_y == 0 && p.pad_x == 0);
    const bool k3 = (p.KH == 3 && p.KW == 3 && p.pad_y == p.pad_x && p.pad_x >= 0 && p.pad_x <= 1);
    if (!(k1 || k3) || (p.stride != 1 && p.stride != 2) || p.oy_step != 1 || p.ox_step != 1)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: 1x1 (pad 0) or 3x3 (pad 0 / 1) layers, stride 1 or 2, dense output window");
    if (p.oy_off < 0 || p.ox_off < 0 || p.OH + p.oy_off > p.OHf || p.OW + p.ox_off > p.OWf) return l2i_set_error(L2I_E_ARG, "conv2d_h8: output window exceeds the output tensor");
    hipStream_t st = (hipStream_t)stream;
    bool wide = (p.CoutP % 64) == 0;
    {   // [r5] small maps: with 64-channel blocks a launch of a <= 32^2 map has fewer blocks than the chip has CUs and every block walks its 48 - 96
        // phases alone on its CU (1 wave per SIMD: every barrier and DMA wait exposed); 32-channel blocks double the blocks (L2I_H8_SMALL_WM1=0: off)
        static const int small_env = getenv("L2I_H8_SMALL_WM1") ? atoi(getenv("L2I_H8_SMALL_WM1")) : 1;
        const int th = (p.KH == 3 && p.stride == 2) ? 4 : 8;
        const long blocks64 = (long)p.B * ((p.OW + 31) / 32) * ((p.OH + th - 1) / th) * ((p.CoutP + 63) / 64);
        if (small_env && wide && !p.out_f32 && !p.rgb_w && blocks64 < 256) wide = false;      // (measured, batch 8, 512 -> 512 3x3: 4^2 34.8 -> 26.5 us, 8^2 35.7 -> 27.5, 16^2 37.8 -> 30.4; 32^2 = 256 blocks: no difference)
    }
    if (p.out_f32) {                                       // fp32 NCHW output (gradients landing on images, the last layer in front of an fp32 consumer)
        if (p.in_mask) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: ReLU-on-load needs the h8 output");
        if (p.sq_ref) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: sq_ref needs the h8 output");
        if ((p.Cin % 32) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: fp32 output needs Cin % 32 == 0");
        if (p.KH == 3 && p.stride == 1) return wide ? launch_h8<2, 2, 3, 1, 0, true>(p, st) : launch_h8<1, 2, 3, 1, 0, true>(p, st);
        if (p.KH == 1 && p.stride == 1) return wide ? launch_h8<2, 2, 1, 1, 0, true>(p, st) : launch_h8<1, 2, 1, 1, 0, true>(p, st);
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: fp32 output is built for stride-1 layers");
    }
    if ((p.Cout % 8) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: h8 output needs Cout % 8 == 0");
    // K chunk of the 3x3 stride-1 layers: 16 channels (one MFMA step per tap) — half the LDS per block puts FOUR blocks on a CU, and it is the number
    // of tiles in flight that fills the memory pipeline on the HBM-bound high-resolution layers (64 -> 64 @1024^2: 738 against 663 TFLOP/s) and
    // hides the per-phase barriers on the others (512 -> 512 @64^2: 1188 against 1135); 32 channels only for 512 channels on <= 32^2 maps
    static const int ks_env = getenv("L2I_H8_KS") ? atoi(getenv("L2I_H8_KS")) : 0;
    const bool ks1 = (p.Cin % 32) != 0 || (ks_env ? ks_env == 1 : (p.Cin <= 256 || p.OW > 32));    // measured per shape (tools/probes/h8_bench.py): 16 wins everywhere but 512 channels on <= 32^2 maps
    if (p.in_mask) {                                       // ReLU-on-load: the 3x3 stride-1 layers of VGG-19
        if (p.KH == 3 && p.stride == 1) {
            if (ks1) return wide ? launch_h8<2, 2, 3, 1, 0, false, true, 1>(p, st) : launch_h8<1, 2, 3, 1, 0, false, true, 1>(p, st);
            return wide ? launch_h8<2, 2, 3, 1, 0, false, true>(p, st) : launch_h8<1, 2, 3, 1, 0, false, true>(p, st);
        }
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: ReLU-on-load is built for 3x3 stride-1 layers");
    }
    if (p.KH == 3 && p.stride == 1 && ks1) return wide ? launch_h8<2, 2, 3, 1, 0, false, false, 1>(p, st) : launch_h8<1, 2, 3, 1, 0, false, false, 1>(p, st);
    if ((p.Cin % 32) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_h8: Cin % 32 == 0 except for 3x3 stride-1 layers (Cin % 16 == 0)");
    if (p.KH == 3 && p.stride == 1) return wide ? launch_h8<2, 2, 3, 1, 0, false>(p, st) : launch_h8<1, 2, 3, 1, 0, false>(p, st);
    if (p.KH == 1 && p.stride == 1) return wide ? launch_h8<2, 2, 1, 1, 0, false>(p, st) : launch_h8<1, 2, 1, 1, 0, false>(p, st);
    if (p.KH == 3 && p.stride == 2) {                      // 16-channel chunks: the 9 x 65-slot tile of a 32-channel chunk leaves one block per CU
        static const int s2_env = getenv("L2I_H8_S2KS") ? atoi(getenv("L2I_H8_S2KS")) : 1;
        if (s2_env == 1) return wide ? launch_h8<2, 1, 3, 2, 0, false, false, 1>(p, st) : launch_h8<1, 1, 3, 2, 0, false, false, 1>(p, st);
        return wide ? launch_h8<2, 1, 3, 2, 0, false>(p, st) : launch_h8<1, 1, 3, 2, 0, false>(p, st);
    }
    return wide ? launch_h8<2, 2, 1, 2, 0, false>(p, st) : launch_h8<1, 2, 1, 2, 0, false>(p, st);
}

extern "C" int H8_NAME(l2i_conv_transpose2d_h8)(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv_transpose2d_h8: null params");
    const l2i_conv_params& p = *pp;
    if (int rc = h8_common_checks(p, "conv_transpose2d_h8")) return rc;
    const int nat = (p.H - 1) * 2 - 2 * p.pad_y + 3, natw = (p.W - 1) * 2 - 2 * p.pad_x + 3;
    if ((p.Cin % 32) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d_h8: Cin must be a multiple of 32");
    if (p.KH != 3 || p.KW != 3 || p.stride != 2 || p.pad_y != p.pad_x || p.pad_x < 0 || p.pad_x > 1 || p.OHf < nat || p.OHf > nat + 8 || p.OWf < natw || p.OWf > natw + 8 ||
        p.OH != (p.OHf + 1) / 2 || p.OW != (p.OWf + 1) / 2 || p.out_f32 || (p.Cout % 8) != 0 || p.sq_ref || p.in_mask)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d_h8: 3x3 stride-2 layer (pad 0 or 1), natural output size (or up to 8 larger), h8 output");
    hipStream_t st = (hipStream_t)stream;
    const bool wide = (p.CoutP % 64) == 0;
    if (p.pad_x == 0) return wide ? launch_h8<2, 1, 3, 1, 1, false>(p, st) : launch_h8<1, 2, 3, 1, 1, false>(p, st);
    return wide ? launch_h8<2, 1, 3, 1, 2, false>(p, st) : launch_h8<1, 2, 3, 1, 2, false>(p, st);
}
}  // namespace H8_NS
