// l2i_conv16.hip — stride-1 / stride-2 1x1 and 3x3 correlations on the bf16 matrix cores with the 3-term operand split,
// pipelined (gfx950).  Entry point l2i_conv2d_bf16x3_f32.
//
// Arithmetic: every fp32 operand is x = hi + lo, hi = bf16(x), lo = bf16(x - hi); a*b = ah*bh + ah*bl + al*bh on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (relative product error <= ~2^-17: fp32-class results at 3/16 of the
// matrix-pipe time of v_mfma_f32_32x32x2_f32).  fp32 tensors in HBM on both sides: the streaming kernels around the convs are
// unchanged and the parity tests run on both matrix paths.
//
// What the first generation left on the table (254-352 TFLOP/s of the 833 the split allows): two barriers per chunk with
// nothing in flight across them, 4-byte staging loads, weights through registers, 2-way conflicts on every fragment read.
// This kernel:
//   * K chunk = 16 channels (3x3) or 32 (1x1).  The fp32 halo tile of the NEXT chunk travels global -> registers as aligned
//     16-byte row vectors (8 channels x 4 pixels per thread item: the 8 channels of a pixel are then lane-local, so the
//     hi/lo split and the bf16x8 pack need no cross-lane traffic) while the current chunk is on the matrix cores, and is
//     committed (activation-gradient mask, style scale, split) to the other LDS stage after them: ONE barrier per phase.
//   * Weights are split and packed on the host in exactly the LDS image order ([Cin/16][tap][half][CoutP][8 bf16]) and go
//     global -> LDS by DMA (buffer_load_dwordx4 ... lds), one kernel ROW (K taps) per phase, double buffered: no registers,
//     no VALU, and a 3x3 layer keeps two blocks per CU (76 KiB of LDS each).
//   * LDS images are [8-channel half][row][column] x 16 B: a fragment read is one ds_read_b128 at lane base + immediate,
//     consecutive lanes on consecutive 16-byte slots (conflict free); stride-2 layers store even and odd columns in separate
//     planes of the row so that their reads are unit-stride too.
//   * Blocks are renumbered so that the channel blocks of one pixel tile run back to back on one XCD (the tile is fetched
//     from HBM once, then from that XCD's L2).
// Mapping as everywhere in this library: M = out-channels (A operand, weights), N = 32-pixel row segments (B operand),
// 4 waves along N, accumulators in the 32x32 C/D layout (one register = 32 consecutive pixels of one channel).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef L2I_C16_AHEAD
#define L2I_C16_AHEAD 0
#endif

namespace s16 {
// K x K window, stride S, WN 32-pixel rows per wave (TH = 4 WN output rows per block, 32 columns).
// TR != 0: stride-2 TRANSPOSED 3x3 conv (TR = 1: pad 0, TR = 2: pad 1), all four output parities from one staged tile: the tile is
// TH x 32 input POSITIONS t, output (2t + py, 2s + px) = sum over the taps of that parity of x[t + dy][s + dx] w[ky][kx], dy, dx in
// {dmin, dmin + 1} (dmin = -1 for pad 0, 0 for pad 1): a 2x2-tap correlation per parity on a tile with one halo row / column.
template <int WN, int K, int S, int TR = 0> struct Geo {
    static constexpr int TH = 4 * WN;
    static constexpr bool GATHER = (K == 1 && S == 2);                    // 1x1 stride 2: only even rows / columns are staged
    static constexpr int ROWS = TR ? TH + 1 : (GATHER ? TH : (TH - 1) * S + K);   // staged input rows
    static constexpr int RSTEP = GATHER ? 2 : 1;                          // global row step between staged rows
    static constexpr int NV = (S == 1) ? (K == 1 ? 8 : (TR == 2 ? 9 : 10)) : (K == 1 ? 16 : 17);   // aligned 4-pixel vectors per staged row
    // 16-byte slots per staged row and 8-channel half (dense: measured, a padded pitch that makes the staging stores conflict-free
    // buys nothing — loads, split VALU and stores each cost ~10-15 % of the kernel, see DESIGN.md — and row-major items coalesce better)
    static constexpr int RP = GATHER ? 32 : NV * 4;
    static constexpr int RPH = RP / 2;                                    // stride 2: odd columns start here
    static constexpr int KS = (K == 1) ? 2 : 1;                           // 16-channel MFMA steps per chunk
    static constexpr int NH = 2 * KS;                                     // 8-channel halves per chunk
    static constexpr int CK = 16 * KS;
    static constexpr int HSTRIDE = ROWS * RP;                             // slots of an 8-channel half
    static constexpr int IN_SLOTS = NH * HSTRIDE;                         // slots per plane (hi or lo) per stage
    static constexpr int NITEM = ROWS * NV;                               // (row, vector) items per half; a thread keeps one half
    static constexpr int TPH = 256 / NH;                                  // threads per half
    static constexpr int NS = (NITEM + TPH - 1) / TPH;                    // items per thread per chunk
    static constexpr int DUMP = (NS * TPH == NITEM) ? 0 : (NS * TPH - NITEM) * NH + (S == 2 ? 36 : 4);   // slots behind a plane for the threads without an item
    static constexpr int IN_PLANE = IN_SLOTS + DUMP;
};
}

struct S16Launch {
    int tiles_x, tiles_y, mblocks, total, nchunks;
    int shift;                         // columns between the first staged column (aligned to 4) and the tile's first tap column
    int vec_epi;
};

__device__ __forceinline__ unsigned cvt_pk_bf16_b(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 8 fp32 -> 8 bf16 hi (round to nearest even) and 8 bf16 lo = bf16(x - hi)
__device__ __forceinline__ void split8b(const float (&v)[8], u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = cvt_pk_bf16_b(v[2 * q], v[2 * q + 1]);
        const float h0 = __uint_as_float(h[q] << 16), h1 = __uint_as_float(h[q] & 0xffff0000u);
        l[q] = cvt_pk_bf16_b(v[2 * q] - h0, v[2 * q + 1] - h1);
    }
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}

template <int WM, int WN, int K, int S, bool MASK, int TR = 0>
__global__ __launch_bounds__(256, (S == 1 && WM * WN <= 4) ? 2 : 1) void conv_bf16x3_pipe_kernel(const l2i_conv_params p, const S16Launch L) {
    using G = s16::Geo<WN, K, S, TR>;
    constexpr int NACC = TR ? 4 : 1;                       // transposed: one accumulator set per output parity (py, px)
    constexpr int DMIN = (TR == 1) ? -1 : 0;               // first input offset of the transposed taps
    constexpr int BM = WM * 32;
    constexpr int WSLOTS = K * G::KS * 2 * BM;             // slots per weight plane per phase: K taps x KS steps x 2 halves x BM channels
    constexpr int WPIECES = 2 * WSLOTS / 64;               // 1 KiB DMA pieces per phase, both planes
    constexpr int WPW = (WPIECES + 3) / 4;                 // per wave
    constexpr int IN_PLANE = G::IN_PLANE;                  // tile slots + the dump slots of the threads without an item (see lslot)
    constexpr int IN_STAGE = 2 * IN_PLANE, W_STAGE = 2 * WSLOTS;
    constexpr int NLOADS = G::NS * 8 * (MASK ? 2 : 1);     // register loads issued AFTER the weight DMA of a phase
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4* const in_st = smem4;                            // 2 stages x [hi | lo] x [half][row][slot]
    u32x4* const w_st = smem4 + 2 * IN_STAGE;              // 2 stages x [hi | lo] x [tap][step][half][channel]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // blocks are dealt round-robin to the 8 XCDs: renumber so that the channel blocks of a pixel tile share an XCD (and its L2)
    int w = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (w >= L.total) return;
    // (the divisions run on the VALU: readfirstlane makes the block coordinates provably wave-uniform again, otherwise every buffer
    // descriptor built from them is wrapped in a waterfall loop with a vmcnt(0) inside)
    const int mblk = __builtin_amdgcn_readfirstlane(w % L.mblocks); w /= L.mblocks;
    const int tx = __builtin_amdgcn_readfirstlane(w % L.tiles_x); w /= L.tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(w % L.tiles_y); w /= L.tiles_y;
    const int b = __builtin_amdgcn_readfirstlane(w), m0 = mblk * BM, oy0 = ty * G::TH, ox0 = tx * 32;
    const int iy0 = TR ? oy0 + DMIN : oy0 * S - p.pad_y;
    const int xs = TR ? ox0 + DMIN - L.shift : ox0 * S - p.pad_x - L.shift;     // first staged column: multiple of 4 (may be negative)

    // ---- descriptors ----
    const size_t plane_x = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane_x * sizeof(float));
    const unsigned in_bytes = (unsigned)p.Cin * plane_b;
    const size_t smp = (size_t)b * p.Cin * plane_x;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + smp), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + smp), 0, in_bytes, 0x00020000);
    const unsigned wpl_bytes = (unsigned)((size_t)(p.Cin / 16) * K * K * 2 * p.CoutP * 16);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, wpl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, wpl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)((p.in_scale ? p.in_scale : p.x) + (size_t)b * p.Cin), 0,
                                                                            p.in_scale ? (unsigned)(p.Cin * sizeof(float)) : 0u, 0x00020000);

    // ---- input staging: a thread owns one 8-channel half and NS (row, 4-pixel vector) items of the tile ----
    const int nh = tid % G::NH, tq = tid / G::NH;
    unsigned voff[G::NS];
    int lslot[G::NS];
#pragma unroll
    for (int u = 0; u < G::NS; ++u) {
        const int it = tq + u * G::TPH;
        voff[u] = in_bytes;
        lslot[u] = G::IN_SLOTS + (it - G::NITEM) * G::NH + nh;   // threads without an item store their zeros to a dump slot behind the plane:
        if (it < G::NITEM) {                               // the commit stays branch-free and can be scheduled between the MFMAs
            const int row = it / G::NV, v = it - row * G::NV;
            const int gy = iy0 + row * G::RSTEP, gx = xs + 4 * v;
            lslot[u] = nh * G::HSTRIDE + row * G::RP + (G::GATHER ? 2 * v : (S == 2 ? 2 * v : 4 * v));
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)        // W % 4 == 0: a vector is inside the row or outside it, never across
                voff[u] = (unsigned)nh * 8u * plane_b + (unsigned)(gy * p.W + gx) * 4u;
        }
    }
    u32x4 xin[G::NS][8];
    u32x4 xmk[MASK ? G::NS : 1][MASK ? 8 : 1];
    float scl[8];

    // weight DMA pieces of this wave: piece q = wave + 4 t covers LDS slots [64 q, 64 q + 64) of the phase ([plane][tap][step][half][channel])
    unsigned wvoff[WPW];
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int q = wave + 4 * t;
        const int sl = (q * 64 + lane) % WSLOTS;                           // slot inside the plane
        const int r = sl / BM, i = sl - r * BM;                            // r = (tap * KS + step) * 2 + half
        const int hf = r & 1, st = (r >> 1) % G::KS, tap = (r >> 1) / G::KS;
        wvoff[t] = (unsigned)((((st * K * K + tap) * 2 + hf) * p.CoutP + m0 + i) * 16);
    }

    auto dma_w = [&](int chunk, int ky, int stage) {
        const unsigned soff = (unsigned)((((size_t)chunk * G::KS * K * K + ky * K) * 2) * p.CoutP * 16);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(w_st + stage * W_STAGE);
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int q = wave_u + 4 * t;                  // wave-uniform: a scalar branch (a DMA through a null descriptor would WRITE zeros)
            if (q < WPIECES) {
                const bool lo = q >= WPIECES / 2;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(wvoff[t]), "s"(lo ? rs_wl : rs_wh), "s"(__builtin_amdgcn_readfirstlane(lds0 + q * 1024)), "s"(soff)
                             : "memory");
            }
        }
    };
    auto issue_scales = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            scl[e] = p.in_scale ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_s, (unsigned)((nh * 8 + e) * sizeof(float)), (unsigned)(c0 * sizeof(float)), 0)) : 1.f;
    };
    auto issue_loads = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned so = (unsigned)(c0 + e) * plane_b;
#pragma unroll
            for (int u = 0; u < G::NS; ++u) {
                xin[u][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff[u], so, 0);
                if constexpr (MASK) xmk[u][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, voff[u], so, 0);
            }
        }
    };
    // one pixel column (px of the 4-pixel vectors) of every item of the prefetched tile -> LDS stage: mask, style scale, split, two 16-byte stores.
    // The K > 1 pipeline spreads the four columns over the MFMA steps of the chunk's later phases (VALU in the shadow of the matrix pipe).
    auto commit_px = [&](int stage, int px) {
#ifdef L2I_ABL_NOCOMMIT
        return;
#endif
#ifdef L2I_ABL_LOADSONLY                                   // keep the loads alive, no VALU, no LDS stores
#pragma unroll
        for (int u = 0; u < G::NS; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(xin[u][e]));
        return;
#endif
        if (G::GATHER && (px & 1)) return;                 // 1x1 stride 2 reads even columns only
        u32x4* ih = in_st + stage * IN_STAGE;
        u32x4* il = ih + IN_PLANE;
#pragma unroll
        for (int u = 0; u < G::NS; ++u) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = __uint_as_float(xin[u][e][px]);
                if constexpr (MASK) t *= (__uint_as_float(xmk[u][e][px]) > 0.f) ? p.mask_pos : p.mask_neg;
                v[e] = t * scl[e];
            }
            u32x4 hi, lo;
            split8b(v, hi, lo);
            const int s = lslot[u] + ((S == 2) ? ((G::GATHER ? 0 : (px & 1) * G::RPH) + (px >> 1)) : px);
#ifdef L2I_ABL_NOSTORE
            asm volatile("" ::"v"(hi), "v"(lo), "v"(s));
#else
            ih[s] = hi;
            il[s] = lo;
#endif
        }
    };
    auto commit = [&](int stage) {
#pragma unroll
        for (int px = 0; px < 4; ++px) commit_px(stage, px);
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

    // fragment bases (in slots): lane (half, j); everything else is a compile-time or wave-uniform offset
    const int rstep_out = G::GATHER ? 1 : S;               // staged rows between consecutive output rows
    const int bbase = half * G::HSTRIDE + wave * WN * rstep_out * G::RP + j;
    int coloff[K];                                         // slot offset of tap column kx for output pixel 0
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        // transposed: tap kx feeds output parity (kx + pad) & 1 from input column s + dx, dx = (px + pad - kx) / 2; staged column dx - DMIN
        const int c = TR ? (((kx + (TR == 2)) & 1) + (TR == 2) - kx) / 2 - DMIN + L.shift : kx + L.shift;
        coloff[kx] = G::GATHER ? 0 : (S == 2 ? (c & 1) * G::RPH + (c >> 1) : c);
    }
    const int abase = half * BM + j;

    // fragments of one MFMA step (tap kx, 16-channel step ks): WM + WN pairs of 16-byte LDS reads
    struct Frag { bf16x8 ah[WM], al[WM], bh[WN], bl[WN]; };
    constexpr int NSTEP = K * G::KS;
    auto load_frags = [&](Frag& f, const u32x4* ih, const u32x4* il, const u32x4* wh, const u32x4* wl, int stp) {
        const int kx = stp / G::KS, ks = stp % G::KS;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int idx = ((kx * G::KS + ks) * 2) * BM + m * 32;
            f.ah[m] = __builtin_bit_cast(bf16x8, wh[idx]);
            f.al[m] = __builtin_bit_cast(bf16x8, wl[idx]);
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int idx = ks * 2 * G::HSTRIDE + n * rstep_out * G::RP + coloff[kx];
            f.bh[n] = __builtin_bit_cast(bf16x8, ih[idx]);
            f.bl[n] = __builtin_bit_cast(bf16x8, il[idx]);
        }
    };
    auto mma = [&](const Frag& f, f32x16 (&ac)[WM][WN]) {  // three passes over the WM x WN tiles: dependent accumulations are WM*WN MFMAs apart
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[m], f.bh[n], ac[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[m], f.bl[n], ac[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[m], f.bh[n], ac[m][n], 0, 0, 0);
    };
    // One phase: NSTEP MFMA steps; the fragments of step s+1 are read while step s is on the matrix pipe (two register sets), and
    // `px_of(step)` >= 0 names the pixel column of the NEXT chunk's tile that is committed to `commit_stage` beside that step's MFMAs.
    auto mfma_phase = [&](int in_stage, int w_stage, auto ky_t, int commit_stage, auto px0_t, auto px1_t, auto px2_t) {
        constexpr int px0 = decltype(px0_t)::value, px1 = decltype(px1_t)::value, px2 = decltype(px2_t)::value;   // compile time: no branch in the MFMA stream
        constexpr int ky = decltype(ky_t)::value;
        // transposed: kernel row ky feeds output parity py = (ky + pad) & 1 from input row t + dy, dy = (py + pad - ky) / 2
        constexpr int PADT = (TR == 2) ? 1 : 0;
        constexpr int py = (ky + PADT) & 1;
        constexpr int rowoff = TR ? (py + PADT - ky) / 2 - DMIN : (G::GATHER ? 0 : ky);
        const u32x4* ih = in_st + in_stage * IN_STAGE + bbase + rowoff * G::RP;
        const u32x4* il = ih + IN_PLANE;
        const u32x4* wh = w_st + w_stage * W_STAGE + abase;
        const u32x4* wl = wh + WSLOTS;
        constexpr bool AHEAD = (L2I_C16_AHEAD != 0);
        Frag f[2];
        load_frags(f[0], ih, il, wh, wl, 0);
#pragma unroll
        for (int stp = 0; stp < NSTEP; ++stp) {
            if (AHEAD && stp + 1 < NSTEP) load_frags(f[(stp + 1) & 1], ih, il, wh, wl, stp + 1);
            constexpr int pxs[3] = {px0, px1, px2};
            if constexpr (NSTEP == 3) { if (pxs[stp] >= 0) commit_px(commit_stage, pxs[stp]); }
            mma(f[AHEAD ? (stp & 1) : 0], acc[TR ? py * 2 + ((stp + PADT) & 1) : 0]);
            if (!AHEAD && stp + 1 < NSTEP) load_frags(f[0], ih, il, wh, wl, stp + 1);
        }
    };

    // ---- pipeline: phase = (chunk, kernel row).  Per phase: wait for this phase's weights, ONE barrier, start the next phase's
    //      weight DMA (and, in the first phase of a chunk, the next chunk's tile loads), MFMAs.  3x3: the next chunk's tile (loaded
    //      during phase 0) is committed to the other input stage column by column beside the MFMAs of phases 1 and 2; 1x1: after the
    //      chunk's only phase. ----
    const int nphases = L.nchunks * K;
    issue_scales(0);
    dma_w(0, 0, 0);
    issue_loads(0);
    commit(0);
    auto phase_head = [&](int ch, int ky, bool more) {
        const int ph = ch * K + ky;
        // the DMA of this phase's weights has landed; register loads issued after it (first phase of a chunk, K > 1) stay in flight
        if (K > 1 && ky == 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef L2I_ABL_NOBARRIER
        __syncthreads();
#endif
#ifndef L2I_ABL_NOLOADS
        if (ky == 0 && more) issue_scales((ch + 1) * G::CK);
#endif
#ifndef L2I_ABL_NODMA
        if (ph + 1 < nphases) dma_w(ky + 1 < K ? ch : ch + 1, ky + 1 < K ? ky + 1 : 0, (ph + 1) & 1);
#endif
#ifndef L2I_ABL_NOLOADS
        if (ky == 0 && more) issue_loads((ch + 1) * G::CK);
#endif
    };
    for (int ch = 0; ch < L.nchunks; ++ch) {
        const bool more = ch + 1 < L.nchunks;
        using N_ = std::integral_constant<int, -1>;
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        using K3 = std::integral_constant<int, 3>;
        if constexpr (K == 3 && (MASK || TR)) {            // mask variants: 32 more staging registers (transposed: 4 accumulator sets);
            const int cs = (ch + 1) & 1;                   // interleaving the commit spills them
            phase_head(ch, 0, more);
            mfma_phase(ch & 1, (ch * 3) & 1, K0(), cs, N_(), N_(), N_());
            phase_head(ch, 1, more);
            mfma_phase(ch & 1, (ch * 3 + 1) & 1, K1(), cs, N_(), N_(), N_());
            phase_head(ch, 2, more);
            mfma_phase(ch & 1, (ch * 3 + 2) & 1, K2(), cs, N_(), N_(), N_());
            if (more) commit(cs);
        } else if constexpr (K == 3) {
            const int cs = (ch + 1) & 1;
            phase_head(ch, 0, more);
            mfma_phase(ch & 1, (ch * 3) & 1, K0(), cs, N_(), N_(), N_());
            phase_head(ch, 1, more);
            if (more) mfma_phase(ch & 1, (ch * 3 + 1) & 1, K1(), cs, N_(), N_(), K0());
            else mfma_phase(ch & 1, (ch * 3 + 1) & 1, K1(), cs, N_(), N_(), N_());
            phase_head(ch, 2, more);
            if (more) mfma_phase(ch & 1, (ch * 3 + 2) & 1, K2(), cs, K1(), K2(), K3());
            else mfma_phase(ch & 1, (ch * 3 + 2) & 1, K2(), cs, N_(), N_(), N_());
        } else {
            phase_head(ch, 0, more);
            mfma_phase(ch & 1, ch & 1, K0(), 0, N_(), N_(), N_());
            if (more) commit((ch + 1) & 1);
        }
    }
    __syncthreads();                                       // the stages become the epilogue's transpose strips

    if constexpr (TR == 0) {
        l2i_epilogue_32x32<WM, WN>(p, acc[0], reinterpret_cast<float*>(smem4), b, m0, oy0, ox0, L.vec_epi != 0);
    } else {
        // transposed epilogue: y[b, co, 2t + py, 2s + px] = acc[py][px] * out_scale[b, co] * out_gain; the two x parities of a lane are adjacent
        // outputs: 8-byte stores, 256 contiguous bytes per channel row and lane half (the (2W+1)-wide rows are only 4-byte aligned)
        const size_t plane_o = (size_t)p.OHf * p.OWf;
        const float* osc = p.out_scale ? p.out_scale + (size_t)b * p.Cout : nullptr;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int t = oy0 + wave * WN + n, sx = ox0 + j;
#pragma unroll
            for (int py2 = 0; py2 < 2; ++py2) {
                const int oy = 2 * t + py2, ox = 2 * sx;
                if (oy < p.OHf && ox < p.OWf) {
                    const bool two = ox + 1 < p.OWf;
#pragma unroll
                    for (int m = 0; m < WM; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (co < p.Cout) {
                                const float sc = (osc ? osc[co] : 1.f) * p.out_gain;
                                float* dst = p.y + ((size_t)b * p.Cout + co) * plane_o + (size_t)oy * p.OWf + ox;
                                const float v0 = acc[py2 * 2 + 0][m][n][r] * sc, v1 = acc[py2 * 2 + 1][m][n][r] * sc;
                                if (two) *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);      // 4-byte aligned: fine on gfx950
                                else dst[0] = v0;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int K, int S>
static int launch_pipe(const l2i_conv_params& p, hipStream_t st) {
    using G = s16::Geo<WN, K, S>;
    constexpr int BM = WM * 32;
    constexpr int WSLOTS = K * G::KS * 2 * BM;
    S16Launch L;
    L.tiles_x = (p.OW + 31) / 32;
    L.tiles_y = (p.OH + G::TH - 1) / G::TH;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: grid too large");
    L.total = (int)total;
    L.nchunks = p.Cin / G::CK;
    L.shift = (4 - (p.pad_x & 3)) & 3;                      // ox0 * S is a multiple of 4
    L.vec_epi = l2i_epilogue_vec_ok(p) ? 1 : 0;
    size_t lds = (size_t)(2 * 2 * G::IN_PLANE + 2 * 2 * WSLOTS) * 16;
    if (lds < 4 * 32 * 64 * sizeof(float)) lds = 4 * 32 * 64 * sizeof(float);
    if (lds > 160 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_bf16x3: tile does not fit the LDS");
    L2I_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_pipe_kernel<WM, WN, K, S, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_pipe_kernel<WM, WN, K, S, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    if (p.in_mask) hipLaunchKernelGGL((conv_bf16x3_pipe_kernel<WM, WN, K, S, true>), dim3(grid), dim3(256), lds, st, p, L);
    else hipLaunchKernelGGL((conv_bf16x3_pipe_kernel<WM, WN, K, S, false>), dim3(grid), dim3(256), lds, st, p, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// stride-2 transposed 3x3 conv (pad 0 or 1), all four output parities per launch
template <int WM, int WN, int TR>
static int launch_pipe_tr(const l2i_conv_params& p, hipStream_t st) {
    using G = s16::Geo<WN, 3, 1, TR>;
    constexpr int BM = WM * 32;
    constexpr int WSLOTS = 3 * 2 * BM;
    S16Launch L;
    L.tiles_x = (p.OW + 31) / 32;                           // OH x OW = input positions t with an output: (OHf + 1) / 2
    L.tiles_y = (p.OH + G::TH - 1) / G::TH;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv_transpose2d_bf16x3: grid too large");
    L.total = (int)total;
    L.nchunks = p.Cin / 16;
    L.shift = (TR == 1) ? 3 : 0;
    L.vec_epi = 0;
    size_t lds = (size_t)(2 * 2 * G::IN_PLANE + 2 * 2 * WSLOTS) * 16;
    L2I_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_pipe_kernel<WM, WN, 3, 1, false, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_pipe_kernel<WM, WN, 3, 1, true, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    if (p.in_mask) hipLaunchKernelGGL((conv_bf16x3_pipe_kernel<WM, WN, 3, 1, true, TR>), dim3(grid), dim3(256), lds, st, p, L);
    else hipLaunchKernelGGL((conv_bf16x3_pipe_kernel<WM, WN, 3, 1, false, TR>), dim3(grid), dim3(256), lds, st, p, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_conv_transpose2d_bf16x3_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv_transpose2d_bf16x3: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w_hi || !p.w_lo || !p.y) return l2i_set_error(L2I_E_ARG, "conv_transpose2d_bf16x3: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, false, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OHf <= 0 || p.OWf <= 0)
        return l2i_set_error(L2I_E_ARG, "conv_transpose2d_bf16x3: non-positive dimension");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv_transpose2d_bf16x3: CoutP must be Cout rounded up to 32");
    if (p.sq_ref || p.sq_out) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d_bf16x3: sq_ref / sq_out are fused in l2i_conv2d_wino_f32 only");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    const int nat = (p.H - 1) * 2 - 2 * p.pad_y + 3, natw = (p.W - 1) * 2 - 2 * p.pad_x + 3;
    if (p.KH != 3 || p.KW != 3 || p.stride != 2 || p.pad_y != p.pad_x || p.pad_x < 0 || p.pad_x > 1 || (p.Cin % 16) != 0 || (p.W % 4) != 0 || p.W < 32 ||
        !al16(p.x) || !al16(p.in_mask) || !al16(p.w_hi) || !al16(p.w_lo) || p.ksplit > 1 || p.OHf < nat || p.OHf > nat + 8 || p.OWf < natw || p.OWf > natw + 8 ||
        p.OH != (p.OHf + 1) / 2 || p.OW != (p.OWf + 1) / 2 || p.noise || p.bias || p.residual || p.out_mask || p.accumulate || p.act != L2I_ACT_NONE ||
        (size_t)p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d_bf16x3: needs a 3x3 stride-2 layer (pad 0 or 1), Cin % 16 == 0, W % 4 == 0, W >= 32, natural output size, "
                                                "in_scale / in_mask / out_scale / out_gain fusions only");
    hipStream_t st = (hipStream_t)stream;
    if (p.pad_x == 0) return (p.CoutP % 64) == 0 ? launch_pipe_tr<2, 1, 1>(p, st) : launch_pipe_tr<1, 2, 1>(p, st);
    return (p.CoutP % 64) == 0 ? launch_pipe_tr<2, 1, 2>(p, st) : launch_pipe_tr<1, 2, 2>(p, st);
}

static bool l2i_bf16x3_pipe_eligible(const l2i_conv_params& p) {
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    const bool k1 = (p.KH == 1 && p.KW == 1 && p.pad_y == 0 && p.pad_x == 0);
    const bool k3 = (p.KH == 3 && p.KW == 3 && p.pad_y == p.pad_x && (p.pad_x == 1 || (p.pad_x == 0 && p.stride == 2)));
    return (k1 || k3) && (p.stride == 1 || p.stride == 2) && p.oy_step == 1 && p.ox_step == 1 && (p.Cin % (k1 ? 32 : 16)) == 0 && p.OW >= 32 &&
           (p.W % 4) == 0 && al16(p.x) && al16(p.in_mask) && p.w_hi && p.w_lo && al16(p.w_hi) && al16(p.w_lo) && p.ksplit <= 1 &&
           (size_t)p.Cin * p.H * p.W * sizeof(float) < 0xFFFFFFF0ull && (size_t)(p.Cin / 16) * p.KH * p.KW * 2 * p.CoutP * 16 < 0xFFFFFFF0ull;
}

static int l2i_launch_bf16x3_pipe(const l2i_conv_params& p, hipStream_t st) {
    const bool wide = (p.CoutP % 64) == 0;
    if (p.KH == 3 && p.stride == 1) return wide ? launch_pipe<2, 2, 3, 1>(p, st) : launch_pipe<1, 2, 3, 1>(p, st);
    if (p.KH == 1 && p.stride == 1) return wide ? launch_pipe<2, 2, 1, 1>(p, st) : launch_pipe<1, 2, 1, 1>(p, st);
    if (p.KH == 3 && p.stride == 2) return wide ? launch_pipe<2, 1, 3, 2>(p, st) : launch_pipe<1, 1, 3, 2>(p, st);
    return wide ? launch_pipe<2, 2, 1, 2>(p, st) : launch_pipe<1, 2, 1, 2>(p, st);
}

extern "C" int l2i_conv2d_bf16x3_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w_hi || !p.w_lo || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, false, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: non-positive dimension");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: CoutP must be Cout rounded up to 32");
    if ((p.sq_ref || p.sq_out) && !(p.sq_ref && p.sq_out && (((uintptr_t)p.sq_ref) % 16) == 0 && l2i_epilogue_vec_ok(p)))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_bf16x3: sq_ref (16-byte aligned) / sq_out need the vectorised epilogue (dense 16-byte output rows)");
    if (p.res_sub && !p.residual) return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: res_sub needs residual");
    if (p.oy_off < 0 || p.ox_off < 0 || p.OH + p.oy_off > p.OHf || p.OW + p.ox_off > p.OWf)
        return l2i_set_error(L2I_E_ARG, "conv2d_bf16x3: output window exceeds the output tensor");
    if (!l2i_bf16x3_pipe_eligible(p))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_bf16x3: needs a 1x1 (pad 0) or 3x3 (pad 1; stride 2 also pad 0) layer, stride 1 or 2, dense output, "
                                                "Cin % 16 == 0 (1x1: % 32), OW >= 32, W % 4 == 0, 16-byte aligned tensors");
    return l2i_launch_bf16x3_pipe(p, (hipStream_t)stream);
}
