// l2i_conv_s2.hip — [r5] the 3x3 STRIDE-2 correlation of the walk-training path (discriminator ResBlock conv2 after its blur, networks.py:530-536 /
// 574-583; ResNet-50's three strided 3x3s; the input-gradient of a generator up layer, networks.py:246-255) on the fp32 matrix cores with BOTH
// operands staged global -> LDS by DMA.  The generic kernel of l2i_conv.hip carries this shape through registers (12 dword loads per thread and chunk,
// a commit pass, two barriers per chunk; 62 % matrix-pipe busy, 20-25 % LDS bank conflicts: profiles/r03_conv_convt_counters.txt); the reviews of
// rounds 2-4 asked for the register-free path and an A/B on the step's shapes (tools/probes/conv_s2_ab.py, profiles/r05_conv_s2_ab.txt).
//
// Same mapping as conv_mfma_kernel<2,2>: v_mfma_f32_32x32x2_f32, block = 64 output channels x (8 rows x 32 columns) of output pixels, wave w owns
// output rows 2 w, 2 w + 1; the two lane halves of the K = 2 MFMA take the two channels of a chunk.  What is different:
//   * input halo tile [2 channels][17 rows][68 columns] by 16-BYTE DMA (buffer_load_dwordx4 ... lds), one channel plane per wave pair; the window
//     starts at the 4-aligned column at or below the tile's first tap (pad 1: four columns left of the first output's centre), so its groups are
//     whole inside or outside the image (outside: out-of-range offset = zeros) — no edge fix-up, no per-element address VALU in the loop;
//   * the plane pitch is ODD (1281 floats): lane j reads window column 2 j + kx — 32 distinct banks of one parity — and the other lane half, one
//     plane further, lands on the other parity: the stride-2 fragment reads are conflict-free;
//   * weights [2][9][64] by the same DMA straight from the [Cin][9][CoutP] pack;
//   * a ring of THREE stages, chunk c + 2 requested while chunk c multiplies, counted vmcnt, ONE barrier per chunk, three blocks per CU.
// Epilogue: the shared fused one (l2i_epilogue.h).  Unmasked launches on maps >= 32 wide; everything else stays on l2i_conv.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_epilogue.h"

namespace cs2 {
constexpr int BM = 64, CK = 2;
constexpr int TH = 8, TW = 32;
constexpr int IH = 2 * TH + 1, IWG = 17, IWP = 4 * IWG;      // 17 rows x 68 window columns
constexpr int NGRP = IH * IWG;                              // 289 16-byte groups per plane: slots 0..4 of 64 lanes
constexpr int PLANE = 1281;                                 // floats: >= 5 * 64 * 4 (the idle lanes of slot 4 write zeros inside their own plane), odd
constexpr int WST = 5 * 64 * 4;                             // 1280 floats: the [2][9][64] weights of a chunk (1152) + the zeros the idle lanes of its fifth DMA slot write
constexpr int STAGE = WST + CK * PLANE + 2;                 // 3844 floats
constexpr int NST = 3;
constexpr int DUMP = 256;
constexpr int NDMA = 5;                                     // DMA instructions per wave and chunk (3 input + 2 weight)
constexpr int LDS_FLOATS = NST * STAGE + DUMP;
static_assert(PLANE % 2 == 1 && PLANE >= 5 * 64 * 4 && WST >= CK * 9 * BM && NST * STAGE >= 4 * 32 * 64, "plane pitch / weight area / epilogue strips");
}

struct ConvS2Launch {
    int tiles_x, tiles_y, mblocks, total, nchunks;
    int vec_epi;                       // 1: LDS-transposed epilogue with 16-byte global accesses (l2i_epilogue_vec_ok)
};

template <bool SCALE>
__global__ __launch_bounds__(256, 3) void conv3x3s2_dma_kernel(const l2i_conv_params p, const ConvS2Launch L) {
    using namespace cs2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dump = smem + NST * STAGE;
    float* stab = dump + DUMP;                              // SCALE: [Cin] input scales of this sample

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // XCD-aware order: the channel blocks of a pixel tile and its x-neighbours share an XCD's L2
    int w = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (w >= L.total) return;
    // (the divisions run on the VALU: without readfirstlane hipcc treats everything derived from them — the buffer descriptors — as divergent)
    const int mblk = __builtin_amdgcn_readfirstlane(w % L.mblocks); w /= L.mblocks;
    const int tx = __builtin_amdgcn_readfirstlane(w % L.tiles_x); w /= L.tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(w % L.tiles_y); w /= L.tiles_y;
    const int b = __builtin_amdgcn_readfirstlane(w), m0 = mblk * BM, oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = 2 * oy0 - p.pad_y;
    const int gx0 = (2 * ox0 - p.pad_x) & ~3;               // window origin: 4-aligned (pad 0: the first tap itself; pad 1: three columns left of it)
    const int off0 = (2 * ox0 - p.pad_x) - gx0;             // 0 or 3

    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * sizeof(float));
    const size_t smp = (size_t)b * p.Cin * ((size_t)p.H * p.W);
    const unsigned wrow_b = (unsigned)p.CoutP * 4u;
    // [r6] The second input slot rides on the first one's M0 with an instruction immediate of 2048, which moves the LDS target AND the global address:
    // its register offset must carry -2048.  Round 5 subtracted it from the byte offset itself, so a pixel in the first 2 KB of a sample had a
    // "negative" (wrapped) voffset and relied on the hardware adding voffset + immediate modulo 2^32 before its range check.  Now the descriptor
    // starts 2048 bytes BEFORE the sample (never dereferenced there: every offset is built from an in-image pixel) and is 2048 bytes longer; every
    // register offset is byte + 2048 - immediate >= 0.
    constexpr unsigned XSHIFT = 2048u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.x + smp) - XSHIFT), 0, (unsigned)p.Cin * plane_b + XSHIFT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.Cin * 9u * wrow_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // input: wave pair (wave & 1) = channel plane of the chunk; wave >> 1 = 0 takes slots 0, 2, 4 of its 289 groups, wave >> 1 = 1 slots 1, 3 (+ a null one)
    const int pl = wave_u & 1, s0 = wave_u >> 1;
    unsigned voff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = (s0 + 2 * u) * 64 + lane;
        const int iy = e / IWG, ig = e - iy * IWG;
        const int gy = iy0 + iy, gx = gx0 + 4 * ig;
        const bool ok = (e < NGRP) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
        voff[u] = ok ? (unsigned)(gy * p.W + gx) * 4u + (u == 1 ? 0u : XSHIFT) : OOB;        // (slot u = 1 rides on slot 0's M0 with an immediate of 2048)
    }
    // weights: 288 16-byte pieces [c][tap][16] of the chunk: wave w takes pieces 64 w .., wave 0 also 256 .. 287
    unsigned wvoff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = (u ? 256 : wave * 64) + lane;
        const int c = e / 144, r = e - c * 144;             // r = tap * 16 + piece
        const bool ok = (e < CK * 9 * 16) & (u == 0 || wave == 0);
        wvoff[u] = ok ? (unsigned)(c * 9 + (r >> 4)) * wrow_b + (unsigned)(m0 + 4 * (r & 15)) * 4u : OOB;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned lds_dump = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dump;

    auto issue = [&](int ch, int st, bool on) {
        const unsigned sbase = lds0 + (unsigned)(st * STAGE * 4);
        const unsigned pbase = sbase + (unsigned)((WST + 1 + pl * PLANE) * 4) + (unsigned)s0 * 1024u;       // (+ 1: planes 4 bytes off 16-byte alignment — harmless to the DMA,
                                                                                                              //  and with the odd pitch every plane base has its own parity pattern)
        const unsigned xso = (unsigned)(ch * CK + pl) * plane_b;
        const unsigned wso = (unsigned)(ch * CK) * 9u * wrow_b;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %5, %7 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %5, %7 offen offset:2048 lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %6, %7 offen lds"
                     :: "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(pbase), "s"(s0 == 0 ? pbase + 4096u : lds_dump),
                        "s"(on ? rs_x : rs_null), "s"((on && s0 == 0) ? rs_x : rs_null), "s"(xso) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %4, %6 offen lds\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %5, %6 offen lds"
                     :: "v"(wvoff[0]), "v"(wvoff[1]), "s"(sbase + (unsigned)wave_u * 1024u), "s"(wave_u == 0 ? sbase + 4096u : lds_dump),
                        "s"(on ? rs_w : rs_null), "s"((on && wave_u == 0) ? rs_w : rs_null), "s"(wso) : "memory");
    };

    if constexpr (SCALE) {
        for (int i = tid; i < p.Cin; i += 256) stab[i] = p.in_scale[(size_t)b * p.Cin + i];
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // fragment bases inside a stage (floats): weights [half][tap][64] at + j; input plane `half`, halo row 2 (2 wave + n) + ky, window column off0 + 2 j + kx
    const int wl = half * 9 * BM + j;
    const int il = WST + 1 + half * PLANE + (4 * wave) * IWP + off0 + 2 * j;

    issue(0, 0, true);
    issue(1, 1, 1 < L.nchunks);
    int st = 0;
    for (int ch = 0; ch < L.nchunks; ++ch) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");           // chunk ch landed; chunk ch + 1 stays in flight
        __syncthreads();                                                      // ... for every wave; every wave is past chunk ch - 1: its stage may be refilled
        const int stn = st == 0 ? NST - 1 : st - 1;                           // stage of chunk ch + 2 = stage of chunk ch - 1
        issue(ch + 2, stn, ch + 2 < L.nchunks);
        const float* sw = smem + st * STAGE + wl;
        const float* si = smem + st * STAGE + il;
        float sv = 1.f;
        if constexpr (SCALE) sv = stab[ch * CK + half];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float a0 = sw[(ky * 3 + kx) * BM], a1 = sw[(ky * 3 + kx) * BM + 32];
                float b0 = si[ky * IWP + kx], b1 = si[(2 + ky) * IWP + kx];
                if constexpr (SCALE) { b0 *= sv; b1 *= sv; }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        st = st + 1 == NST ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // (the null tail requests) no DMA may outlive the block's LDS
    __syncthreads();                                                          // the stages become the epilogue's transpose strips
    l2i_epilogue_32x32<2, 2>(p, acc, smem, b, m0, oy0, ox0, L.vec_epi != 0);
}

bool l2i_conv3x3s2_eligible(const l2i_conv_params& p) {
    static const bool off = getenv("L2I_CONV_S2_DMA") && atoi(getenv("L2I_CONV_S2_DMA")) == 0;      // A/B switch (tools/probes/conv_s2_ab.py)
    if (off) return false;
    return p.KH == 3 && p.KW == 3 && p.stride == 2 && p.oy_step == 1 && p.ox_step == 1 && !p.in_mask && !p.ws && p.ksplit <= 1 && p.pad_x == p.pad_y &&
           (p.pad_x == 0 || (p.pad_x == 1 && (p.W % 4) == 0)) && (p.Cin % cs2::CK) == 0 && p.Cin >= 8 && ((p.CoutP % cs2::BM) == 0 || p.CoutP >= 160) && p.OW >= 32 && !p.sq_ref &&      // (a block is 64 channels wide: 32-channel layers keep the generic kernel's BM = 32 tile)
          
           (size_t)p.Cin * p.H * p.W * sizeof(float) < 0x7FFF0000ull && (!p.in_scale || p.Cin <= 1024);
}

int l2i_launch_conv3x3s2(const l2i_conv_params& p, hipStream_t st) {
    ConvS2Launch L;
    L.tiles_x = (p.OW + cs2::TW - 1) / cs2::TW;
    L.tiles_y = (p.OH + cs2::TH - 1) / cs2::TH;
    L.mblocks = (p.CoutP + cs2::BM - 1) / cs2::BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d (3x3 stride 2): too many tiles");
    L.total = (int)total;
    L.nchunks = p.Cin / cs2::CK;
    L.vec_epi = l2i_epilogue_vec_ok(p) ? 1 : 0;
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    const size_t lds = (size_t)(cs2::LDS_FLOATS + (p.in_scale ? ((p.Cin + 3) & ~3) : 0)) * sizeof(float);
    if (p.in_scale) {
        L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_dma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        hipLaunchKernelGGL((conv3x3s2_dma_kernel<true>), dim3(grid), dim3(256), lds, st, p, L);
    } else {
        L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        hipLaunchKernelGGL((conv3x3s2_dma_kernel<false>), dim3(grid), dim3(256), lds, st, p, L);
    }
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
