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
//
// HEAD3 (l2i_conv_chain3_h8): the 3x3 stride-1 conv in FRONT of the pair joins the launch — bottleneck conv2 -> conv3 -> next block's conv1 (and, backwards,
// conv2's input gradient -> conv1's -> conv3's of the block below): everything between two 3x3 convs is pointwise, so a spatial tile (4 WN rows x 32 columns)
// runs the 3x3 on its halo tile exactly as conv_h8_kernel does (same chunking, same accumulation order), turns its accumulators into the first 1x1 conv's
// B fragments in registers, and continues as above with one 32-pixel row segment per (wave, n).  The 3x3 conv's output map never exists in HBM (its sign
// plane does, when asked for).  One launch per bottleneck.
// Measured, not kept (profiles/r06_pair_ab.txt): weights two chunks ahead in three stages and sched_group_barrier-pipelined A-fragment reads for the
// 256 -> 1024 -> 256 shape (one block per CU, one wave per SIMD): 71 us either way — at that occupancy the parts of a chunk add (timing ablations: no
// single part is worth more than 10 us of the 72), like the mid-size launches of conv_h8_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_h8_common.h"

namespace H8_NS {

struct PairLaunch {
    int total, tiles_per_sample, npix, nch;     // blocks, pixel tiles per sample, H * W, 32-channel chunks of the first conv's output
    int tiles_x, W;                             // HEAD3: 32-column tiles per row, map width
};

// ---- HEAD3: a 3x3 stride-1 pad-1 conv on the halo tile of (4 WN rows x 32 columns), all Cout = 32 WM channels in the block; its epilogue (sign-plane mask,
// bias, ReLU, optional sign plane of the result) leaves the packed 16-bit slots in `xf` as the B fragments of the 1x1 conv that follows.  The K loop is
// conv_h8_kernel<WM, WN, 3, 1, 0, false, false, KS = 1> (l2i_conv_h8.hip: 16-channel chunks, one kernel row per phase, tile and weights by DMA, double
// buffered, one barrier per phase): same MFMAs in the same order, so the result equals that launch's bit for bit.
template <int WM, int WN>
__device__ __forceinline__ void head3x3_to_frags(const l2i_conv_params& p, u32x4* smem4, int b, int oy0, int ox0, const unsigned (&pixn)[WN], bf16x8 (&xf)[2 * WM][WN]) {
    constexpr int TH = 4 * WN, ROWS = TH + 2, RP = 34, HSTRIDE = ROWS * RP, IN_SLOTS = 2 * HSTRIDE;
    constexpr int NPWA = ((IN_SLOTS + 63) / 64 + 3) / 4, IN_STAGE = 4 * NPWA * 64;
    constexpr int BM = 32 * WM, WSLOTS = 3 * 2 * BM, WPIECES = WSLOTS / 64, WPW = (WPIECES + 3) / 4;
    u32x4* const in_st = smem4;
    u32x4* const w_st = smem4 + 2 * IN_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * 16);
    const unsigned in_bytes = (unsigned)(p.Cin / 8) * plane_b;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.x) + (size_t)b * in_bytes), 0, in_bytes, 0x00020000);
    const unsigned wpl_bytes = (unsigned)((size_t)(p.Cin / 16) * 9 * 2 * p.CoutP * 16);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, wpl_bytes, 0x00020000);
    unsigned ivoff[NPWA];
#pragma unroll
    for (int t = 0; t < NPWA; ++t) {
        const int sl = (wave + 4 * t) * 64 + lane;
        ivoff[t] = in_bytes;                                                           // out of range: the DMA writes zeros (= the padding)
        if (sl < IN_SLOTS) {
            const int nh = sl / HSTRIDE, r2 = sl - nh * HSTRIDE, row = r2 / RP, col = r2 - row * RP;
            const int gy = iy0 + row, gx = ix0 + col;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ivoff[t] = (unsigned)nh * plane_b + (unsigned)(gy * p.W + gx) * 16u;
        }
    }
    auto dma_in = [&](int chunk, int stage) {
        const unsigned soff = (unsigned)chunk * 2u * plane_b;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(in_st + stage * IN_STAGE);
#pragma unroll
        for (int t = 0; t < NPWA; ++t) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ivoff[t]), "s"(rs_x), "s"(__builtin_amdgcn_readfirstlane(lds0 + (wave_u + 4 * t) * 1024)), "s"(soff) : "memory");
        }
    };
    unsigned wvoff[WPW];
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int sl = ((wave + 4 * t) * 64 + lane) % WSLOTS;
        const int r = sl / BM, i = sl - r * BM;                                        // r = tap * 2 + half
        wvoff[t] = (unsigned)((r * p.CoutP + i) * 16);
    }
    auto dma_w = [&](int chunk, int ky, int stage_slot) {
        const unsigned soff = (unsigned)((((size_t)chunk * 9 + ky * 3) * 2) * p.CoutP * 16);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(w_st + stage_slot);
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int q = wave_u + 4 * t;
            if (q < WPIECES) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(wvoff[t]), "s"(rs_w), "s"(__builtin_amdgcn_readfirstlane(lds0 + q * 1024)), "s"(soff) : "memory");
            }
        }
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int bbase = half * HSTRIDE + wave * WN * RP + j, abase = half * BM + j;
    auto mfma_phase = [&](int in_stage, int w_slot, int ky) {
        const u32x4* ih = in_st + in_stage * IN_STAGE + bbase + ky * RP;
        const u32x4* wh = w_st + w_slot + abase;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 af[WM], bf[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) af[m] = __builtin_bit_cast(bf16x8, wh[(kx * 2) * BM + m * 32]);
#pragma unroll
            for (int n = 0; n < WN; ++n) bf[n] = __builtin_bit_cast(bf16x8, ih[n * RP + kx]);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n) acc[m][n] = H8_MFMA(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
    };
    const int nchunks = p.Cin / 16, nphases = nchunks * 3;
    dma_w(0, 0, 0);
    dma_in(0, 0);
    auto phase_head = [&](int ch, int ky, bool more) {
        const int ph = ch * 3 + ky;
        // this phase's weights (and, first phase of a chunk, the chunk's tile) have landed; the next chunk's tile, issued AFTER the weights of phase (ch, 1),
        // stays in flight across that phase's barrier
        if (ky == 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPWA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ph + 1 < nphases) dma_w(ky + 1 < 3 ? ch : ch + 1, ky + 1 < 3 ? ky + 1 : 0, ((ph + 1) & 1) * WSLOTS);
        if (ky == 0 && more) dma_in(ch + 1, (ch + 1) & 1);
    };
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        phase_head(ch, 0, more);
        mfma_phase(ch & 1, ((ch * 3) & 1) * WSLOTS, 0);
        phase_head(ch, 1, more);
        mfma_phase(ch & 1, ((ch * 3 + 1) & 1) * WSLOTS, 1);
        phase_head(ch, 2, more);
        mfma_phase(ch & 1, ((ch * 3 + 2) & 1) * WSLOTS, 2);
    }
    // ---- epilogue: * [out_mask bit] + bias, ReLU; the packed slot of (m, pr) is K step 2 m + pr of the next conv; sign plane of the result when asked for ----
    const int cg = p.Cout / 8;
    const size_t npix = (size_t)p.H * p.W;
    const uint8_t* const mq = p.out_mask ? reinterpret_cast<const uint8_t*>(p.out_mask) + ((size_t)b * cg + half) * npix + j : nullptr;
    uint8_t* const sq = p.mask_out ? p.mask_out + ((size_t)b * cg + half) * npix + j : nullptr;
    u32x4* const yq = p.y ? reinterpret_cast<u32x4*>(p.y) + ((size_t)b * cg + half) * npix + j : nullptr;
    const bool relu = p.act == L2I_ACT_RELU;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + 32 * m + (2 * pr + half) * 8), b1 = *reinterpret_cast<const float4*>(p.bias + 32 * m + (2 * pr + half) * 8 + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float g[8];
                h8_gather(acc[m][n], pr, half, g);
                const size_t off = (size_t)(4 * m + 2 * pr) * npix + pixn[n];
                if (mq) {
                    const unsigned mb = mq[off];
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] *= ((mb >> e) & 1u) ? p.mask_pos : p.mask_neg;
                }
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] += bv[e];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = g[e] > 0.f ? g[e] : 0.f;
                }
                const u32x4 out = {cvt_pk_bf16_h8(g[0], g[1]), cvt_pk_bf16_h8(g[2], g[3]), cvt_pk_bf16_h8(g[4], g[5]), cvt_pk_bf16_h8(g[6], g[7])};
                if (yq) yq[off] = out;
                if (sq) sq[off] = (uint8_t)h8_sign_byte(out);
                xf[2 * m + pr][n] = __builtin_bit_cast(bf16x8, out);
            }
        }
}


template <int KB, int MC, int WN, bool MASKED, int RDIST, int OCC, bool HEAD3>
__global__ __launch_bounds__(256, OCC) void pair_h8_kernel(const l2i_conv_params p0, const l2i_conv_params p1, const l2i_conv_params p2, const PairLaunch L) {
    constexpr int C3 = 32 * MC;                       // output channels of the second conv
    constexpr int W1S = KB * 64, W2S = 4 * C3;        // slots of a chunk's W1 rows ([kstep][half][32]) and W2 slice ([2 ksteps][half][C3])
    constexpr int WSTAGE = W1S + W2S;
    constexpr int NPIECES = KB + 2 * MC;              // 1 KiB DMA pieces per chunk
    static_assert(NPIECES % 4 == 0 && MC % 2 == 0, "pieces are dealt to four waves; a W2 piece is 64 channels");
    constexpr int NPW = NPIECES / 4;
    constexpr int WST = 2;                            // weight stages
    constexpr int RS = RDIST + 1;                     // ring stages of the epilogue operand
    constexpr int RPW = 2 * WN;                       // operand pieces per wave and chunk: (pr, n)
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4* const w_st = smem4;
    u32x4* const r_st = smem4 + WST * WSTAGE;
    float* const bias_s = reinterpret_cast<float*>(r_st + 4 * RS * RPW * 64);
    uint8_t* const mb_s = reinterpret_cast<uint8_t*>(bias_s + (L.nch * 32 + C3));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w = (int)blockIdx.x;
    if (w >= L.total) return;
    const int b = __builtin_amdgcn_readfirstlane(w / L.tiles_per_sample);
    const int tile = __builtin_amdgcn_readfirstlane(w - b * L.tiles_per_sample);
    const unsigned npix = (unsigned)L.npix;
    // the wave's WN 32-pixel row segments: consecutive runs of the flat map, or (HEAD3) rows oy0 + wave WN + n, columns ox0 .. ox0 + 31 of a spatial tile
    const int ty = HEAD3 ? tile / L.tiles_x : 0, oy0 = ty * (4 * WN), ox0 = HEAD3 ? (tile - ty * L.tiles_x) * 32 : 0;
    unsigned pixn[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n)
        pixn[n] = HEAD3 ? (unsigned)((oy0 + wave_u * WN + n) * L.W + ox0) : (unsigned)tile * (128u * WN) + (unsigned)wave_u * (32u * WN) + 32u * n;
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
#ifdef L2I_PAIR_ABL_W                                      // timing ablations (tools/probes/pair_ablate.sh): results are wrong by construction
        if (c > 0) return;
#endif
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
    for (int n = 0; n < WN; ++n) rvoff[n] = ((unsigned)half * npix + pixn[n] + (unsigned)j) * 16u;
    u32x4* const r_mine = r_st + wave * (RS * RPW * 64);
    auto dma_r = [&](int c) {
#ifdef L2I_PAIR_ABL_RES
        return;
#endif
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

    bf16x8 xf[KB][WN];                                // the wave's pixels of the first 1x1 conv's input: K step ks = groups 2 ks, 2 ks + 1
    if constexpr (HEAD3) {
        head3x3_to_frags<KB / 2, WN>(p0, smem4, b, oy0, ox0, pixn, xf);
        __syncthreads();                              // every wave is through with the 3x3's stages: the pair's stages overlay them
    }
    // ---- prologue ----
#pragma unroll
    for (int d = 0; d < RDIST; ++d)
        if (d < nch) dma_r(d);
    dma_w(0, 0);
    if constexpr (!HEAD3) {
        const u32x4* xb = reinterpret_cast<const u32x4*>(p1.x) + ((size_t)b * cg_in + half) * npix + j;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks)
#pragma unroll
            for (int n = 0; n < WN; ++n) xf[ks][n] = __builtin_bit_cast(bf16x8, xb[(size_t)(2 * ks) * npix + pixn[n]]);
    }
    for (int i = tid; i < nch * 32 + C3; i += 256)
        bias_s[i] = i < nch * 32 ? (p1.bias ? p1.bias[i] : 0.f) : (p2.bias ? p2.bias[i - nch * 32] : 0.f);
    uint8_t* const mb_mine = mb_s + wave * (nch * RPW * 64);
    if constexpr (MASKED) {
        const uint8_t* mp = reinterpret_cast<const uint8_t*>(p1.out_mask) + ((size_t)b * cg1 + half) * npix + j;
        for (int c = 0; c < nch; ++c)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int n = 0; n < WN; ++n) mb_mine[(c * RPW + pr * WN + n) * 64 + lane] = mp[(size_t)(4 * c + 2 * pr) * npix + pixn[n]];
    }
    const bool has_b1 = p1.bias != nullptr, relu1 = p1.act == L2I_ACT_RELU, sign1 = p1.mask_out != nullptr, rmask1 = MASKED && p1.res_mask != nullptr;

    f32x16 accC[MC][WN];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accC[m][n][r] = 0.f;

    u32x4* const y1b = reinterpret_cast<u32x4*>(p1.y) + ((size_t)b * cg1 + half) * npix + j;
    uint8_t* const s1b = sign1 ? p1.mask_out + ((size_t)b * cg1 + half) * npix + j : nullptr;

    for (int c = 0; c < nch; ++c) {
        // W(c) — and the operand pieces of chunk c, issued before it (RDIST = 2) or right behind it (RDIST = 1) — have landed.  Issued after them in
        // iteration c - 1: [RDIST = 2: the operand pieces of chunk c + 1,] then at least RPW stores of `mid`.
        if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (RDIST == 2 && c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW) : "memory");
#ifndef L2I_PAIR_ABL_BAR
        __syncthreads();
#endif
        if (c + 1 < nch) dma_w(c + 1, (c + 1) % WST);
        if (c + RDIST < nch) dma_r(c + RDIST);

        // ---- first conv, channels 32 c .. 32 c + 31 ----
        const u32x4* w1s = w_st + (c % WST) * WSTAGE + lane;
        f32x16 accB[WN];
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[n][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, w1s[ks * 64]);
#pragma unroll
#ifdef L2I_PAIR_ABL_MFMA1
            for (int n = 0; n < WN; ++n) accB[n][0] += __builtin_bit_cast(f32x4_, af)[0] * __builtin_bit_cast(f32x4_, xf[ks][n])[0];
#else
            for (int n = 0; n < WN; ++n) accB[n] = H8_MFMA(af, xf[ks][n], accB[n], 0, 0, 0);
#endif
        }
        // ---- its epilogue: (* mask) + bias + operand (* mask), ReLU, store, sign byte; the packed slots are the second conv's B fragments ----
        const u32x4* rq = r_mine + (c % RS) * (RPW * 64) + lane;
        const u32x4* w2s = w_st + (c % WST) * WSTAGE + W1S + half * C3 + j;
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
                const size_t off = (size_t)(4 * c + 2 * pr) * npix + pixn[n];
#ifdef L2I_PAIR_ABL_STORE
                if (out.x == 0x12345678u && out.y == 0x9abcdef0u)
#endif
                y1b[off] = out;
                if (sign1) s1b[off] = (uint8_t)h8_sign_byte(out);
                frag[n] = __builtin_bit_cast(bf16x8, out);
            }
            // ---- second conv: K step 2 c + pr ----
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, w2s[pr * 2 * C3 + m * 32]);
#pragma unroll
#ifdef L2I_PAIR_ABL_MFMA2
                for (int n = 0; n < WN; ++n) accC[m][n][0] += __builtin_bit_cast(f32x4_, af)[0] * __builtin_bit_cast(f32x4_, frag[n])[0];
#else
                for (int n = 0; n < WN; ++n) accC[m][n] = H8_MFMA(af, frag[n], accC[m][n], 0, 0, 0);
#endif
            }
        }
    }

    // ---- the second conv's epilogue ----
    const bool has_b2 = p2.bias != nullptr, relu2 = p2.act == L2I_ACT_RELU;
    u32x4* const y2b = reinterpret_cast<u32x4*>(p2.y) + ((size_t)b * cg2 + half) * npix + j;
    const uint8_t* const m2b = p2.out_mask ? reinterpret_cast<const uint8_t*>(p2.out_mask) + ((size_t)b * cg2 + half) * npix + j : nullptr;
    uint8_t* const s2b = p2.mask_out ? p2.mask_out + ((size_t)b * cg2 + half) * npix + j : nullptr;
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
                const size_t off = (size_t)(4 * m + 2 * pr) * npix + pixn[n];
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

template <int KB, int MC, int WN, bool MASKED, int RDIST, int OCC, bool HEAD3>
static int launch_pair(const l2i_conv_params* p0, const l2i_conv_params& p1, const l2i_conv_params& p2, hipStream_t st) {
    PairLaunch L;
    L.npix = p1.H * p1.W;
    L.W = p1.W;
    L.tiles_x = p1.W / 32;
    L.tiles_per_sample = HEAD3 ? L.tiles_x * (p1.H / (4 * WN)) : L.npix / (128 * WN);
    L.total = p1.B * L.tiles_per_sample;
    L.nch = p1.Cout / 32;
    constexpr int C3 = 32 * MC, WSTAGE = KB * 64 + 4 * C3;
    size_t lds = (size_t)(2 * WSTAGE + 4 * (RDIST + 1) * 2 * WN * 64) * 16 + (size_t)(L.nch * 32 + C3) * 4 + (MASKED ? (size_t)4 * L.nch * 2 * WN * 64 : 0);
    if (HEAD3) {                                           // the 3x3's stages (two halo tiles, two weight rows) are overlaid by the pair's
        constexpr int IN_SLOTS = 2 * (4 * WN + 2) * 34, IN_STAGE = 4 * (((IN_SLOTS + 63) / 64 + 3) / 4) * 64, WSLOTS = 3 * 2 * 16 * KB;
        const size_t lds3 = (size_t)(2 * IN_STAGE + 2 * WSLOTS) * 16;
        if (lds3 > lds) lds = lds3;
    }
    if (lds > 160 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: stages do not fit the LDS");
    L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_h8_kernel<KB, MC, WN, MASKED, RDIST, OCC, HEAD3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((pair_h8_kernel<KB, MC, WN, MASKED, RDIST, OCC, HEAD3>), dim3((unsigned)L.total), dim3(256), lds, st, p0 ? *p0 : p1, p1, p2, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

static const char* pair_conv_unsupported(const l2i_conv_params& p, int k, bool need_x, bool need_y) {
    if (!p.w_hi || (need_y && !p.y) || (need_x && !p.x)) return "null tensor";
    if (p.KH != k || p.KW != k || p.stride != 1 || p.pad_y != k / 2 || p.pad_x != k / 2 || p.oy_step != 1 || p.ox_step != 1 || p.oy_off || p.ox_off) return "1x1 (pad 0) / 3x3 (pad 1) stride-1 convs with a dense output";
    if (p.OH != p.H || p.OW != p.W || p.OHf != p.H || p.OWf != p.W) return "output maps have the input's size";
    if (p.in_scale || p.in_mask || p.out_scale || p.noise || p.accumulate || p.res_sub || p.sq_ref || p.sq_out || p.rgb_w || p.rgb_bias || p.rgb_out || p.pool_out || p.pool_idx || p.in_h8 || p.out_f32 || p.w_bstride)
        return "only bias, residual (first 1x1 conv), sign-plane masks, ReLU and the sign-plane output are fused";
    if (p.act != L2I_ACT_NONE && p.act != L2I_ACT_RELU) return "activation: none or ReLU";
    if (p.out_gain != 1.f) return "out_gain must be 1";
    if ((p.out_mask || p.res_mask) && !p.mask_bits) return "masks must be sign planes (mask_bits)";
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0 || (p.Cout % 32) != 0 || (p.Cin % (k == 3 ? 16 : 32)) != 0) return "channel counts must be multiples of 32 (16 for the 3x3's input)";
    return nullptr;
}

static int pair_dispatch(const l2i_conv_params* head, const l2i_conv_params& p1, const l2i_conv_params& p2, int variant, hipStream_t st, const char* who) {
    if (const char* m = pair_conv_unsupported(p1, 1, head == nullptr, true)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (const char* m = pair_conv_unsupported(p2, 1, false, true)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p2.x && p2.x != p1.y) return l2i_set_error(L2I_E_ARG, "conv pair h8: the second conv reads the first conv's output (second->x must be first->y or NULL)");
    if (p2.Cin != p1.Cout || p2.B != p1.B || p2.H != p1.H || p2.W != p1.W) return l2i_set_error(L2I_E_ARG, "conv pair h8: the two 1x1 convs do not chain");
    if (!p1.residual) return l2i_set_error(L2I_E_UNSUPPORTED, "conv pair h8: the first 1x1 conv carries a residual operand (the trunk)");
    if (p2.residual || p2.res_mask) return l2i_set_error(L2I_E_UNSUPPORTED, "conv pair h8: no residual on the second conv");
    if (p1.res_mask && p1.res_mask != p1.out_mask) return l2i_set_error(L2I_E_UNSUPPORTED, "conv pair h8: res_mask must be the out_mask plane");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if (!al16(p1.x) || !al16(p1.w_hi) || !al16(p1.y) || !al16(p1.residual) || !al16(p2.w_hi) || !al16(p2.y) || !al16(p1.bias) || !al16(p2.bias))
        return l2i_set_error(L2I_E_ARG, "conv pair h8: tensors must be 16-byte aligned");
    const long npix = (long)p1.H * p1.W;
    if ((size_t)(p1.Cout / 8) * npix * 16 >= 0xFFFFFFF0ull || (size_t)(p1.Cin / 8) * npix * 16 >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv pair h8: one sample must stay below 4 GiB");
    const int KB = p1.Cin / 16, MC = p2.Cout / 32;
    const bool masked = p1.out_mask != nullptr;
    if (p1.res_mask && !masked) return l2i_set_error(L2I_E_UNSUPPORTED, "conv pair h8: res_mask without out_mask");
    static const int var_env = getenv("L2I_PAIR_VARIANT") ? atoi(getenv("L2I_PAIR_VARIANT")) : -1;
    const int v = var_env >= 0 ? var_env : variant;      // 0: two 32-pixel rows per wave (256-pixel tiles); 1: one (128-pixel tiles, three blocks per CU)
    int wn;
    if (head) {
        const l2i_conv_params& p0 = *head;
        if (const char* m = pair_conv_unsupported(p0, 3, true, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
        if (p0.Cout != p1.Cin || p0.B != p1.B || p0.H != p1.H || p0.W != p1.W) return l2i_set_error(L2I_E_ARG, "conv_chain3_h8: the 3x3 conv does not feed the first 1x1 conv");
        if (p0.residual || p0.res_mask || !al16(p0.x) || !al16(p0.w_hi) || !al16(p0.y) || !al16(p0.bias)) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_chain3_h8: no residual on the 3x3 conv; 16-byte aligned tensors");
        if (p0.Cin > 256 || (size_t)(p0.Cin / 8) * npix * 16 >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_chain3_h8: the 3x3 conv runs on 16-channel chunks (Cin <= 256)");
        wn = (v == 1 || (p1.H % 8) != 0) ? 1 : 2;
        if ((p1.W % 32) != 0 || (p1.H % (4 * wn)) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_chain3_h8: W must be a multiple of 32, H of 4");
    } else {
        wn = (v == 1 || (npix % 256) != 0) ? 1 : 2;
        if ((npix % (128 * wn)) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv1x1_pair_h8: H * W must be a multiple of 128");
    }
#define L2I_PAIR_GO(kb, mc, wnv, rd, occ)                                                                                                   \
    return head ? (masked ? launch_pair<kb, mc, wnv, true, rd, occ, true>(head, p1, p2, st) : launch_pair<kb, mc, wnv, false, rd, occ, true>(head, p1, p2, st)) \
                : (masked ? launch_pair<kb, mc, wnv, true, rd, occ, false>(nullptr, p1, p2, st) : launch_pair<kb, mc, wnv, false, rd, occ, false>(nullptr, p1, p2, st));
#define L2I_PAIR_CASE(kb, mc)                                                                                                              \
    if (KB == kb && MC == mc) {                                                                                                            \
        if (wn == 2) { L2I_PAIR_GO(kb, mc, 2, 2, 2) }                                                                                      \
        L2I_PAIR_GO(kb, mc, 1, 2, 3)                                                                                                       \
    }
    L2I_PAIR_CASE(4, 2)
    L2I_PAIR_CASE(4, 4)
#undef L2I_PAIR_CASE
    if (KB == 8 && MC == 4) {                              // 128 -> 512 -> 128: two rows per wave spill (64 fragment + 64 + 128 accumulator registers): 128-pixel tiles only
        if (wn == 2 && head && (p1.H % 4) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_chain3_h8: H must be a multiple of 4");
        L2I_PAIR_GO(8, 4, 1, 2, 3)
    }
    // 256 output channels of the second conv: 128 accumulator registers per 32-pixel row — 128-pixel tiles only, no 3x3 head
    if (!head) {
#define L2I_PAIR_CASE1(kb, mc, occ)                                                                                                        \
        if (KB == kb && MC == mc) return masked ? launch_pair<kb, mc, 1, true, 1, occ, false>(nullptr, p1, p2, st) : launch_pair<kb, mc, 1, false, 1, occ, false>(nullptr, p1, p2, st);
        L2I_PAIR_CASE1(16, 8, 1)                      // one block per CU by its LDS: the whole register file (accumulators in AGPRs) to one wave per SIMD
        L2I_PAIR_CASE1(8, 8, 2)
#undef L2I_PAIR_CASE1
    }
#undef L2I_PAIR_GO
    (void)who;
    return l2i_set_error(L2I_E_UNSUPPORTED, head ? "conv_chain3_h8: built for (C, Cout2) = (64, 64), (128, 128), (64, 128)"
                                                 : "conv1x1_pair_h8: built for (Cin1, Cout2) = (64, 64), (128, 128), (256, 256), (64, 128), (128, 256)");
}

extern "C" int H8_NAME(l2i_conv1x1_pair_h8)(const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream) {
    if (!first || !second) return l2i_set_error(L2I_E_ARG, "conv1x1_pair_h8: null params");
    return pair_dispatch(nullptr, *first, *second, variant, (hipStream_t)stream, "conv1x1_pair_h8");
}

extern "C" int H8_NAME(l2i_conv_chain3_h8)(const l2i_conv_params* head3x3, const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream) {
    if (!head3x3 || !first || !second) return l2i_set_error(L2I_E_ARG, "conv_chain3_h8: null params");
    return pair_dispatch(head3x3, *first, *second, variant, (hipStream_t)stream, "conv_chain3_h8");
}
}  // namespace H8_NS
