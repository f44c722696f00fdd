// l2i_convt.hip — stride-2 TRANSPOSED convolution with all four output parities in one launch (gfx950, fp32 MFMA).
//
//   y[b, co, 2*iy + ky - pad, 2*ix + kx - pad] += x[b, ci, iy, ix] * w[co, ci, ky, kx]
//
// Used for the forward of the StyleGAN2 up layers (conv_transpose2d(stride 2), reference networks.py:246-255) and for
// the input-gradient of every stride-2 convolution on the path (discriminator conv2 of each ResBlock, ResNet-50
// layer{2,3,4}.0.conv2, the 7x7 stem).  A transposed conv is four interleaved stride-1 correlations (one per output
// parity) over the SAME input tile; issuing them as separate launches (l2i_conv2d_f32 with oy_step = 2) stages the tile
// four times, gives each launch only 1-4 taps of matrix work per staged chunk, and writes every other output pixel.
// Here one block keeps 4 accumulator sets (one per parity), walks all K*K taps per staged chunk (each tap feeds
// exactly one parity: same MAC count as the dense form), and writes whole contiguous output rows: the two x-parities
// are interleaved through a per-wave LDS transpose and stored 16 bytes per lane.
//
// Block = 256 threads = 4 waves along N; block tile = 32 output channels x (4*WN*32) INPUT-resolution positions t;
// staging: buffer loads issued two chunks ahead into registers, committed into the other of two LDS stages one chunk ahead,
// one barrier per chunk.
// [r5] DMAIN = true (unmasked 3x3 launches with one sample per tile: the generator's up layers and the stride-2 convs' input-gradients on maps >= 32
// positions wide): the input tile goes global -> LDS by dword DMA as well — a dense [CK][IH][IW] image, one channel plane per wave in turn, 64
// elements per instruction, the plane pitch padded so that the two lane halves of a fragment read land on disjoint bank halves — into the other
// stage while the current chunk multiplies: no staging registers (36 fewer per lane), no commit pass.  The rounds 2-4 reviews asked for this path
// and an A/B on the step's shapes: tools/probes/convt_ab.py, profiles/r05_convt_ab.txt; L2I_CONVT_DMA=0 restores the register path.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvTLaunch {
    int TW, TH, tb_log2;               // tile = 2^tb samples x TH rows x TW columns of positions (TW even; TW*TH*2^tb <= 128*WN slots)
    unsigned magic_tw, magic_th;
    int ksplit, cin_per;               // split-K (small maps): channel ranges of cin_per (multiple of CK) channels, partial sums into p.ws
    int tiles_x, tiles_y, bgroups, mblocks;
    int CK, IH, IW, IWp, planeS, plane, rows_c;
    unsigned magic_iw, magic_rc, magic_ih;
    int in_elems, w_vec;
    int total;                         // blocks with work (grid padded to a multiple of 8)
    int nslots;                        // DMAIN: 64-element DMA slots per channel plane (IH * IW elements, dense)
};

__device__ __forceinline__ unsigned fast_div_t(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

template <int K, int PAD> struct TrGeom {
    static constexpr int k0(int pi) { return (pi + PAD) % 2; }
    static constexpr int A(int pi) { return (K - k0(pi) + 1) / 2; }              // taps of parity pi
    static constexpr int d(int pi) { return (pi + PAD - k0(pi)) / 2; }
    static constexpr int pad(int pi) { return A(pi) - 1 - d(pi); }                // correlation padding of parity pi
    static constexpr int P = pad(0) > pad(1) ? pad(0) : pad(1);                   // halo before the tile
    static constexpr int q_(int pi) { return A(pi) - 1 - pad(pi); }
    static constexpr int Q = q_(0) > q_(1) ? q_(0) : q_(1);                       // halo after the tile
};

template <int K, int PAD, int WN, bool MASK, bool DMAIN = false>
__global__ __launch_bounds__(256, WN == 1 ? 3 : 2) void convt_mfma_kernel(const l2i_conv_params p, const ConvTLaunch L) {
    using G = TrGeom<K, PAD>;
    constexpr int BM = 32, KK = K * K, NIN = DMAIN ? 1 : 12, P = G::P, Q = G::Q;
    static_assert(!(DMAIN && MASK), "the DMA path has no VALU pass for a gradient mask");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // two stages of [input tile | weights | input scales]: the chunk after the one being multiplied is committed into the other
    // stage at the top of the iteration, so one barrier per chunk suffices
    const int stage_f = L.CK * L.plane + L.CK * KK * BM + ((L.CK << L.tb_log2) + 3 & ~3);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int TW = L.TW, TH = L.TH;
    // XCD-aware block order (see l2i_conv.hip): the channel blocks of one position tile share an XCD and its L2
    int bid = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (bid >= L.total) return;
    const int mblk = bid % L.mblocks; bid /= L.mblocks;
    const int tx = bid % L.tiles_x; bid /= L.tiles_x;
    const int ty = bid % L.tiles_y; bid /= L.tiles_y;
    const int bgrp = bid % L.bgroups;
    const int split = bid / L.bgroups;                           // 0 unless split-K
    const int b0 = bgrp << L.tb_log2;
    const int m0 = mblk * BM;
    const int c_begin = split * L.cin_per;
    const int c_end = (c_begin + L.cin_per < p.Cin) ? c_begin + L.cin_per : p.Cin;
    const int ty0 = ty * TH, tx0 = tx * TW;                        // tile origin in input-resolution positions t
    const int iy0 = ty0 - P, ix0 = tx0 - P;                        // origin of the staged tile

    const int CKh = L.CK >> 1;
    int pixoff[WN], sbase[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int pi = (wave * WN + n) * 32 + j;
        const int rowi = (int)fast_div_t((unsigned)pi, L.magic_tw), c = pi - rowi * TW;
        int tb = (int)fast_div_t((unsigned)rowi, L.magic_th);
        int r = rowi - tb * TH;
        const bool used = tb < (1 << L.tb_log2);                   // slots past TB*TH*TW (tiles that do not fill 128*WN positions) idle on pixel 0
        if (!used) { tb = 0; r = 0; }
        pixoff[n] = tb * L.planeS + r * L.IWp + (used ? c : 0) + half * CKh * L.plane;
        sbase[n] = tb * L.CK + half * CKh;
    }
    const int wlane = half * CKh * KK * BM + j;

    f32x16 acc[4][WN];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][n][r] = 0.f;

    // ---- staging state (see l2i_conv.hip) ----
    const size_t plane_x = (size_t)p.H * p.W;
    const int nb = (p.B - b0) < (1 << L.tb_log2) ? (p.B - b0) : (1 << L.tb_log2);
    const unsigned in_bytes = (unsigned)((size_t)nb * p.Cin * plane_x * sizeof(float));
    const size_t grp_off = (size_t)b0 * p.Cin * plane_x;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + grp_off), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + grp_off), 0, in_bytes, 0x00020000);
    const unsigned w_bytes = (unsigned)((size_t)p.Cin * KK * p.CoutP * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
    const unsigned sc_bytes = (unsigned)((size_t)nb * p.Cin * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.in_scale ? p.in_scale : p.x) + (size_t)b0 * p.Cin), 0, p.in_scale ? sc_bytes : 0u, 0x00020000);

    unsigned rin[NIN], rmk[MASK ? NIN : 1], voff[NIN];
    int loff[NIN];
    unsigned rsc = 0;
    // DMAIN: element e = 64 s + lane of a dense [IH][IW] channel plane -> byte offset inside the plane (minus the slot's immediate), or out of range
    constexpr int NSL = 6;
    unsigned dvoff[DMAIN ? NSL : 1];
    // [r6] slot s of a plane rides on the plane's M0 with an immediate of 256 s (LDS target and global address alike), so its register offset carries
    // -256 s.  To keep every register offset non-negative (round 5 let the first pixels of a sample wrap below zero and relied on voffset + immediate
    // being added modulo 2^32 before the range check) the DMA's descriptor starts DSHIFT bytes before the sample group and is DSHIFT bytes longer.
    constexpr unsigned DSHIFT = (NSL - 1) * 256u;
    const __amdgpu_buffer_rsrc_t rs_xd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.x + grp_off) - DSHIFT), 0, in_bytes + DSHIFT, 0x00020000);
    if constexpr (DMAIN) {
#pragma unroll
        for (int s_ = 0; s_ < NSL; ++s_) {
            const unsigned e = (unsigned)(s_ * 64 + lane);
            const unsigned row = fast_div_t(e, L.magic_iw), ixu = e - row * L.IW;
            const int gy = iy0 + (int)row, gx = ix0 + (int)ixu;
            const bool ok = (s_ < L.nslots) & ((int)row < L.IH) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
            dvoff[s_] = ok ? (unsigned)(gy * p.W + gx) * 4u + (DSHIFT - (unsigned)s_ * 256u) : 0x80000000u;      // >= 0: rs_xd starts DSHIFT bytes early
        }
    }
#pragma unroll
    for (int u = 0; u < (DMAIN ? 0 : NIN); ++u) {
        const unsigned e = tid + u * 256;
        voff[u] = in_bytes;
        loff[u] = -1;
        if ((int)e < L.in_elems) {
            const unsigned row = fast_div_t(e, L.magic_iw), ixu = e - row * L.IW;
            const unsigned c = fast_div_t(row, L.magic_rc);
            const unsigned r2 = row - c * L.rows_c;
            const unsigned tb = fast_div_t(r2, L.magic_ih);
            const int iy = (int)(r2 - tb * L.IH);
            const int gy = iy0 + iy, gx = ix0 + (int)ixu;
            loff[u] = (int)(c * L.plane + tb * L.planeS + iy * L.IWp + ixu);
            if ((int)tb < nb && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                voff[u] = (unsigned)((((size_t)tb * p.Cin + c) * plane_x + (size_t)gy * p.W + gx) * sizeof(float));
        }
    }
    constexpr int V = BM / 4;
    const int wrow0 = tid / V, wc4 = tid - wrow0 * V;
    const unsigned wvoff = (m0 + wc4 * 4 < p.CoutP) ? (unsigned)(((size_t)wrow0 * p.CoutP + m0 + wc4 * 4) * sizeof(float)) : w_bytes;
    const unsigned wstep = (unsigned)((256 / V) * p.CoutP * sizeof(float));
    const int TBCK = L.CK << L.tb_log2;
    const unsigned scoff = (tid < TBCK) ? (unsigned)((((tid / L.CK) * p.Cin) + (tid % L.CK)) * sizeof(float)) : sc_bytes;

    auto issue = [&](int c0) {
        const unsigned so = (unsigned)((size_t)c0 * plane_x * sizeof(float));
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            rin[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff[u], so, 0);
            if constexpr (MASK) rmk[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, voff[u], so, 0);
        }
        rsc = __builtin_amdgcn_raw_buffer_load_b32(rs_s, scoff, (unsigned)(c0 * sizeof(float)), 0);
    };
    // weights of a chunk: global -> LDS by DMA, 16 B per lane, a wave instruction fills 1 KiB of the [CK][KK][BM] tile in lane order
    // (inline asm: through the builtin hipcc drains every outstanding load before the next ds_read); no registers, no commit
    auto dma_w = [&](int c0, int st) {
        const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + st * stage_f + L.CK * L.plane);
        const unsigned sw = (unsigned)((size_t)c0 * KK * p.CoutP * sizeof(float));
        for (int u = 0; u * 256 < L.w_vec; ++u) {
            if (tid + u * 256 < L.w_vec) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(wvoff), "s"(rs_w), "s"(__builtin_amdgcn_readfirstlane(lds_w + (u * 256 + wave * 64) * 16)),
                               "s"(sw + u * wstep) : "memory");
            }
        }
    };
    // DMAIN: the input tile of a chunk: wave w takes channel planes w, w + 4, ...; one M0 per plane, its slots by immediate offsets (the immediate
    // moves the LDS target and the global address alike: dvoff carries DSHIFT - 256 s against a descriptor that starts DSHIFT bytes early).  Idle lanes / slots past the plane write zeros into its padding.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_in = [&](int c0, int st) {
        if constexpr (DMAIN) {
            const unsigned lds_in0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + st * stage_f);
            for (int c = wave_u; c < L.CK; c += 4) {
                const unsigned base = lds_in0 + (unsigned)(c * L.plane * 4);
                const unsigned so = (unsigned)((size_t)(c0 + c) * plane_x * sizeof(float));
                asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                             "buffer_load_dword %0, %7, %8 offen lds\n\t"
                             "buffer_load_dword %1, %7, %8 offen offset:256 lds\n\t"
                             "buffer_load_dword %2, %7, %8 offen offset:512 lds\n\t"
                             "buffer_load_dword %3, %7, %8 offen offset:768 lds\n\t"
                             "buffer_load_dword %4, %7, %8 offen offset:1024 lds\n\t"
                             "buffer_load_dword %5, %7, %8 offen offset:1280 lds"
                             :: "v"(dvoff[0]), "v"(dvoff[1]), "v"(dvoff[2]), "v"(dvoff[3]), "v"(dvoff[4]), "v"(dvoff[DMAIN ? 5 : 0]), "s"(base), "s"(rs_xd), "s"(so) : "memory");
            }
        }
    };
    constexpr int NREG = NIN * (MASK ? 2 : 1) + 1;            // register loads of one issue(): younger than the DMA they follow
    auto commit = [&](int st) {
        float* lds_in = smem + st * stage_f;
        float* lds_sc = lds_in + L.CK * L.plane + L.CK * KK * BM;
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            if (loff[u] >= 0) {
                float v = __uint_as_float(rin[u]);
                if constexpr (MASK) v *= (__uint_as_float(rmk[u]) > 0.f) ? p.mask_pos : p.mask_neg;
                lds_in[loff[u]] = v;
            }
        }
        if (p.in_scale) { if (tid < TBCK) lds_sc[tid] = __uint_as_float(rsc); }       // host: TB * CK <= 256 when a style scale is present
        else for (int i = tid; i < TBCK; i += 256) lds_sc[i] = 1.f;
    };

    // DMAIN: the scale-table entries of a chunk travel through one register per thread and are written after the wait at the end of the chunk before
    auto load_sc = [&](int c0) { rsc = __builtin_amdgcn_raw_buffer_load_b32(rs_s, scoff, (unsigned)(c0 * sizeof(float)), 0); };
    auto store_sc = [&](int st) {
        float* lds_sc = smem + st * stage_f + L.CK * L.plane + L.CK * KK * BM;
        if (p.in_scale) { if (tid < TBCK) lds_sc[tid] = __uint_as_float(rsc); }
        else for (int i = tid; i < TBCK; i += 256) lds_sc[i] = 1.f;
    };
    dma_w(c_begin, 0);
    if constexpr (DMAIN) {
        dma_in(c_begin, 0);
        load_sc(c_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_sc(0);
    } else {
        issue(c_begin);
        commit(0);
        if (c_begin + L.CK < c_end) issue(c_begin + L.CK);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREG) : "memory");      // the DMA (older than that issue) has landed
    }
    __syncthreads();
    int st = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += L.CK, st ^= 1) {
        const bool more = c0 + 2 * L.CK < c_end;
        if (c0 + L.CK < c_end) {
            if constexpr (DMAIN) {
                dma_w(c0 + L.CK, st ^ 1);                    // (every wave is past the barrier that ended chunk c0 - CK: the other stage is free)
                dma_in(c0 + L.CK, st ^ 1);
                load_sc(c0 + L.CK);
            } else {
                commit(st ^ 1);                              // loads issued one whole chunk ago
                dma_w(c0 + L.CK, st ^ 1);
                if (more) issue(c0 + 2 * L.CK);              // in flight during the MFMAs below and the next chunk's
            }
        }
        const float* lds_in = smem + st * stage_f;
        const float* lds_w = lds_in + L.CK * L.plane;
        const float* lds_sc = lds_w + L.CK * KK * BM;
        if constexpr (K == 3) {
            // 3x3: the fragments of channel pair cc + 1 are read from LDS while the 18 MFMAs of pair cc run (two register sets,
            // loop unrolled by two) — left to itself the compiler issues each ds_read right before its MFMA and exposes the LDS latency
            constexpr int NB = (P + Q + 1) * (P + Q + 1);
            struct Frag { float a[KK]; float b[NB][WN]; float sv[WN]; };
            auto load = [&](Frag& f, int cc) {
                const float* wr = lds_w + wlane + cc * KK * BM;
                const float* ir = lds_in + cc * L.plane;
#pragma unroll
                for (int t = 0; t < KK; ++t) f.a[t] = wr[t * BM];
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    f.sv[n] = lds_sc[sbase[n] + cc];
#pragma unroll
                    for (int t = 0; t < NB; ++t) f.b[t][n] = ir[pixoff[n] + (t / (P + Q + 1)) * L.IWp + (t % (P + Q + 1))];
                }
            };
            auto mfma = [&](const Frag& f) {
                int slot = 0;
#pragma unroll
                for (int dy = -P; dy <= Q; ++dy) {
#pragma unroll
                    for (int dx = -P; dx <= Q; ++dx) {
                        float bfr[WN];
#pragma unroll
                        for (int n = 0; n < WN; ++n) bfr[n] = f.b[(dy + P) * (P + Q + 1) + (dx + P)][n] * f.sv[n];
#pragma unroll
                        for (int py = 0; py < 2; ++py) {
#pragma unroll
                            for (int px = 0; px < 2; ++px) {
                                const int ay = dy + G::pad(py), ax = dx + G::pad(px);
                                if (ay >= 0 && ay < G::A(py) && ax >= 0 && ax < G::A(px)) {
#pragma unroll
                                    for (int n = 0; n < WN; ++n)
                                        acc[py * 2 + px][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[slot], bfr[n], acc[py * 2 + px][n], 0, 0, 0);
                                    ++slot;
                                }
                            }
                        }
                    }
                }
            };
            Frag f0, f1;
            load(f0, 0);
            for (int cc = 0; cc < CKh; cc += 2) {               // CK is a multiple of 4 for K == 3 (host): always whole pairs
                load(f1, cc + 1);
                __builtin_amdgcn_sched_barrier(0);              // keep the reads ahead of the MFMAs that do not need them
                mfma(f0);
                __builtin_amdgcn_sched_barrier(0);
                load(f0, cc + 2 < CKh ? cc + 2 : 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        for (int cc = 0; cc < CKh; ++cc) {
                const float* wr = lds_w + wlane + cc * KK * BM;
                const float* ir = lds_in + cc * L.plane;
                float sv[WN];
#pragma unroll
                for (int n = 0; n < WN; ++n) sv[n] = lds_sc[sbase[n] + cc];
                int slot = 0;                                   // compile-time after full unrolling
#pragma unroll
                for (int dy = -P; dy <= Q; ++dy) {
#pragma unroll
                    for (int dx = -P; dx <= Q; ++dx) {
                        float bfr[WN];
                        const int toff = (dy + P) * L.IWp + (dx + P);
#pragma unroll
                        for (int n = 0; n < WN; ++n) bfr[n] = ir[pixoff[n] + toff] * sv[n];
#pragma unroll
                        for (int py = 0; py < 2; ++py) {
#pragma unroll
                            for (int px = 0; px < 2; ++px) {
                                const int ay = dy + G::pad(py), ax = dx + G::pad(px);
                                if (ay >= 0 && ay < G::A(py) && ax >= 0 && ax < G::A(px)) {
                                    const float a = wr[slot * BM];
#pragma unroll
                                    for (int n = 0; n < WN; ++n)
                                        acc[py * 2 + px][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bfr[n], acc[py * 2 + px][n], 0, 0, 0);
                                    ++slot;
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (DMAIN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the next chunk's tile, weights and scales have landed (a chunk of MFMAs went by)
            if (c0 + L.CK < c_end) store_sc(st ^ 1);
        } else {
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREG) : "memory");     // the next chunk's weights are in LDS; the register loads stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    // ---- epilogue: interleave the two x-parities through a per-wave LDS strip, write whole output rows ----
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    if (L.ksplit > 1) {
        // split-K: raw partial sums of this channel range, laid out like y; splitk_epilogue_kernel (l2i_conv.hip) reduces and scales
        float* wsp = p.ws + (size_t)split * p.B * p.Cout * plane_o;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int pi = (wave * WN + n) * 32 + j;
            const int rowi = (int)fast_div_t((unsigned)pi, L.magic_tw);
            const int tbi = (int)fast_div_t((unsigned)rowi, L.magic_th);
            const int tcol = tx0 + (pi - rowi * TW), trow = ty0 + (rowi - tbi * TH);
            const int bb = tbi < (1 << L.tb_log2) ? b0 + tbi : p.B;
            if (bb < p.B) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oy = 2 * trow + (q >> 1), ox = 2 * tcol + (q & 1);
                    if (oy < p.OHf && ox < p.OWf) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (co < p.Cout) wsp[((size_t)bb * p.Cout + co) * plane_o + (size_t)oy * p.OWf + ox] = acc[q][n][r];
                        }
                    }
                }
            }
        }
        return;
    }
    float* strip = smem + wave * (32 * 64);                 // [32 channels][64 consecutive output pixels]
    const int ch_l = lane >> 4, q4 = lane & 15;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int pi = (wave * WN + n) * 32 + 2 * q4;      // input-resolution position of this lane's 4 outputs
        const int rowi = (int)fast_div_t((unsigned)pi, L.magic_tw);
        const int tbi = (int)fast_div_t((unsigned)rowi, L.magic_th);
        const int tcol = tx0 + (pi - rowi * TW);
        const int trow = ty0 + (rowi - tbi * TH);
        const int bb = tbi < (1 << L.tb_log2) ? b0 + tbi : p.B;       // unused slots: no sample
        const int ox = 2 * tcol;
        const float* osc = (bb < p.B && p.out_scale) ? p.out_scale + (size_t)bb * p.Cout : nullptr;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
                strip[ch * 64 + 2 * j] = acc[py * 2][n][r];
                strip[ch * 64 + 2 * j + 1] = acc[py * 2 + 1][n][r];
            }
            const int oy = 2 * trow + py;
            const bool rowok = (bb < p.B) && (oy < p.OHf);
            const size_t poff = (size_t)oy * p.OWf + ox;
#pragma unroll 2
            for (int i = 0; i < 8; ++i) {
                const int ch = i * 4 + ch_l;
                const int co = m0 + ch;
                float4 v = *reinterpret_cast<const float4*>(&strip[ch * 64 + 4 * q4]);
                if (rowok && co < p.Cout && ox < p.OWf) {
                    if (osc || p.out_gain != 1.f) {
                        float sc = p.out_gain;
                        if (osc) sc *= osc[co];
                        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                    }
                    float* dst = p.y + ((size_t)bb * p.Cout + co) * plane_o + poff;
                    if (ox + 3 < p.OWf) {
                        *reinterpret_cast<float4*>(dst) = v;                 // 4-byte aligned is enough on gfx950 (probed)
                    } else {
                        dst[0] = v.x;
                        if (ox + 1 < p.OWf) dst[1] = v.y;
                        if (ox + 2 < p.OWf) dst[2] = v.z;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
static int ilog2c(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static unsigned magic_of(unsigned d) { return (unsigned)((0x100000000ULL + d - 1) / d); }

template <int K, int PAD, int WN>
static int launch_convt(const l2i_conv_params& p, hipStream_t st) {
    using G = TrGeom<K, PAD>;
    constexpr int BM = 32, BN = 128 * WN, KK = K * K;
    ConvTLaunch L;
    const int Ty = (p.OHf + 1) / 2, Tx = (p.OWf + 1) / 2;       // input-resolution positions that own an output
    // pixel tile: small maps pack samples into power-of-two tiles; maps >= 32 positions wide (one sample per tile) take the
    // even-width tile that wastes the fewest of the 128*WN slots on the (2H+1)-wide maps of pad 0 (65 positions: 22 x 11 tiles
    // cover 66 x 66, the 32 x 8 grid 96 x 72).  Measured alternative, not kept: restricting the launch to H x W positions and
    // computing the one-pixel edge row / column separately (thin l2i_conv2d_f32 launches or a dedicated VALU kernel) — the edge
    // work walks all of Cin in a handful of blocks and cost more than the tiles it saved.
    int TW, TH;
    if (Tx < 32) {
        TW = 1 << ilog2c(Tx);
        int th = BN / TW;
        const int ty_p2 = 1 << ilog2c(Ty);
        if (th > ty_p2) th = ty_p2;
        TH = 1 << ilog2c(th);
        L.tb_log2 = ilog2c(BN / (TH * TW));
    } else {
        long best = -1;
        TW = 32; TH = BN / 32;
        for (int tw = 64; tw >= 8; tw -= 2) {
            const int th = BN / tw;
            const long tiles = (long)((Tx + tw - 1) / tw) * ((Ty + th - 1) / th);
            const long cost = tiles * 4096 + (long)(th + G::P + G::Q) * (tw + G::P + G::Q) + (tw == 32 ? 0 : 1);   // ties: smaller halo, then 32 wide
            if (best < 0 || cost < best) { best = cost; TW = tw; TH = th; }
        }
        L.tb_log2 = 0;
    }
    L.TW = TW; L.TH = TH;
    L.magic_tw = TW == 1 ? 0u : magic_of((unsigned)TW);
    L.magic_th = TH == 1 ? 0u : magic_of((unsigned)TH);
    const int TB = 1 << L.tb_log2;
    if ((size_t)TB * p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: sample group >= 4 GiB");
    L.tiles_x = (Tx + TW - 1) / TW;
    L.tiles_y = (Ty + TH - 1) / TH;
    L.bgroups = (p.B + TB - 1) / TB;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    L.IH = TH + G::P + G::Q;
    L.IW = TW + G::P + G::Q;
    L.IWp = L.IW | 1;
    L.planeS = L.IH * L.IWp;
    L.plane = (TB * L.planeS + 3) & ~3;
    L.rows_c = TB * L.IH;
    L.nslots = 0;
    static const bool dma_env = !(getenv("L2I_CONVT_DMA") && atoi(getenv("L2I_CONVT_DMA")) == 0);
    const bool dma_in = dma_env && K == 3 && !p.in_mask && TB == 1 && Tx >= 32 && !(p.ksplit > 1 && p.ws) && L.IH * L.IW <= 6 * 64 &&
                        (size_t)p.Cin * p.H * p.W * sizeof(float) < 0x7FFF0000ull;
    if (dma_in) {                                         // dense planes, one per group of DMA slots; the pitch is fixed below once CK is known
        L.IWp = L.IW;
        L.planeS = L.IH * L.IW;
        L.nslots = (L.planeS + 63) / 64;
        L.plane = 6 * 64 + 32;                            // (upper bound for the CK computation: six slots of zeros / data, plus the bank padding)
    }
    const size_t per_c = (size_t)(L.plane + KK * BM) * sizeof(float);
    int ck = (int)(((dma_in && WN == 1 ? 24 : 36) * 1024) / per_c);      // per stage; two stages, two blocks per CU (three of the 128-position tile)
    const int ck_in = dma_in ? 16 : (12 * 256) / (L.rows_c * L.IW);
    if (ck > ck_in) ck = ck_in;
    if (p.in_scale && ck > (256 >> L.tb_log2)) ck = 256 >> L.tb_log2;      // one scale-table entry per thread
    // channels per chunk: pairs (the two lane halves of an MFMA take one channel each); K == 3 walks two pairs per iteration.
    // Chunks must tile Cin exactly: the per-chunk channel offset travels in the scalar offset of the buffer loads, which the
    // hardware does not range-check.
    constexpr int CKQ = (K == 3) ? 4 : 2;
    if ((p.Cin % CKQ) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, K == 3 ? "conv_transpose2d: Cin must be a multiple of 4" : "conv_transpose2d: Cin must be even");
    ck &= ~(CKQ - 1);
    if (ck > p.Cin) ck = p.Cin;
    while (ck > CKQ && (p.Cin % ck) != 0) ck -= CKQ;
    if (ck < CKQ) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: tile cannot be staged");
    L.ksplit = 1;
    L.cin_per = p.Cin;
    if (p.ksplit > 1 && p.ws) {
        const int per = p.Cin / p.ksplit;
        if (per >= CKQ && per * p.ksplit == p.Cin && (per % CKQ) == 0) {
            while (ck > CKQ && (per % ck) != 0) ck -= CKQ;
            if ((per % ck) == 0) { L.ksplit = p.ksplit; L.cin_per = per; }
        }
    }
    if (dma_in) {
        // every slot of a plane is written whole (six instructions per plane: idle lanes write zeros), so the pitch is >= 384; the two lane halves of a
        // fragment read are CK / 2 planes apart: pad until that distance is = 32 (mod 64) banks
        int plane = 6 * 64;
        while (((ck / 2) * plane) % 64 != 32 && plane < 6 * 64 + 64) ++plane;
        if (((ck / 2) * plane) % 64 != 32) plane = 6 * 64 + 16;
        L.plane = plane;
    }
    L.CK = ck;
    L.in_elems = ck * L.rows_c * L.IW;
    L.w_vec = ck * KK * BM / 4;
    L.magic_iw = magic_of((unsigned)L.IW);
    L.magic_rc = magic_of((unsigned)L.rows_c);
    L.magic_ih = magic_of((unsigned)L.IH);
    size_t lds = 2 * ((size_t)(L.plane + KK * BM) * sizeof(float) * ck + (size_t)(((ck << L.tb_log2) + 3) & ~3) * sizeof(float));
    if (lds < 4 * 32 * 64 * sizeof(float)) lds = 4 * 32 * 64 * sizeof(float);
    long grid = (long)L.bgroups * L.tiles_y * L.tiles_x * L.mblocks * L.ksplit;
    if (grid <= 0 || grid > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: grid too large");
    L.total = (int)grid;
    grid = (grid + 7) & ~7L;
    if (lds > (dma_in ? 80 : 64) * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: tile cannot be staged");
    if constexpr (K == 3) {
        if (dma_in) L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_mfma_kernel<K, PAD, WN, false, true>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    }
    if (p.in_mask) hipLaunchKernelGGL((convt_mfma_kernel<K, PAD, WN, true>), dim3((unsigned)grid), dim3(256), lds, st, p, L);
    else if constexpr (K == 3) {
        if (dma_in) hipLaunchKernelGGL((convt_mfma_kernel<K, PAD, WN, false, true>), dim3((unsigned)grid), dim3(256), lds, st, p, L);
        else hipLaunchKernelGGL((convt_mfma_kernel<K, PAD, WN, false>), dim3((unsigned)grid), dim3(256), lds, st, p, L);
    } else hipLaunchKernelGGL((convt_mfma_kernel<K, PAD, WN, false>), dim3((unsigned)grid), dim3(256), lds, st, p, L);
    L2I_CHECK_LAUNCH();
    if (L.ksplit > 1) {                                 // second pass: sum the partials, out_scale / out_gain (l2i_conv.hip)
        l2i_conv_params q = p;
        q.ksplit = L.ksplit;
        q.OH = p.OHf; q.OW = p.OWf; q.oy_step = q.ox_step = 1; q.oy_off = q.ox_off = 0;
        return l2i_launch_splitk_epilogue(q, st);
    }
    return L2I_OK;
}

extern "C" int l2i_conv_transpose2d_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, false, true)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: non-positive dimension");
    if (p.KH != p.KW || p.pad_y != p.pad_x) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: square kernels / symmetric padding only");
    if (p.sq_ref || p.sq_out) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: sq_ref / sq_out are fused in l2i_conv2d_wino_f32 only");
    if (p.CoutP == 4 && p.Cout <= 3) {                   // <= 3 output channels, [Cin][K*K][4] weight pack: the fused VALU kernel (l2i_convt_small.hip)
        if (p.bias || p.noise || p.residual || p.res_mask || p.res_sub || p.out_mask || p.act != L2I_ACT_NONE)
            return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: only in_mask / out_gain / accumulate are fused in the small-output kernel");
        if (!l2i_convt_small_eligible(p)) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d(small): 7x7 / pad 3, W % 4 == 0, OWf % 4 == 0, aligned tensors, no scales");
        if (p.OHf > 2 * p.H + 8 || p.OWf > 2 * p.W + 8) return l2i_set_error(L2I_E_ARG, "conv_transpose2d(small): output window too large for the input");
        return l2i_launch_convt_small(p, (hipStream_t)stream);
    }
    if (p.in_h8) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: in_h8 is built for the 7x7 / pad 3 kernel onto <= 3 channels only");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: CoutP must be Cout rounded up to 32");
    const int full_h = (p.H - 1) * 2 - 2 * p.pad_y + p.KH, full_w = (p.W - 1) * 2 - 2 * p.pad_x + p.KW;
    // larger than natural: the extra rows / columns (no input reaches them) are written as zeros — output_padding, and the discriminator's
    // blur maps padded to whole 16-byte rows (latent2im_amd/discriminator.py)
    if (p.OHf < full_h || p.OWf < full_w || p.OHf > full_h + 8 || p.OWf > full_w + 8)
        return l2i_set_error(L2I_E_ARG, "conv_transpose2d: output must be the natural size (or up to 8 larger: zero rows / columns)");
    if (p.bias || p.noise || p.residual || p.res_mask || p.res_sub || p.out_mask || p.act != L2I_ACT_NONE || p.accumulate)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: only in_scale / in_mask / out_scale / out_gain are fused here");
    if ((((uintptr_t)p.w) % 16) != 0) return l2i_set_error(L2I_E_ARG, "conv_transpose2d: packed weights must be 16-byte aligned");
    if ((size_t)p.Cin * p.KH * p.KW * p.CoutP * sizeof(float) >= 0xFFFFFFF0ull) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: weight pack >= 4 GiB");
    hipStream_t st = (hipStream_t)stream;
    // 128 positions per block (three blocks per CU) on maps up to 33 positions wide, where 256-position tiles leave a CU with one
    // or two blocks for most of the launch; 256 positions (two per CU, half the weight traffic per MFMA) above.  tile_hint 1 / 2 force.
    const bool narrow = p.tile_hint == 1 || (p.tile_hint != 2 && ((p.OWf + 1) / 2 <= 33 || p.W <= 33));     // (outputs may be up to 8 larger than natural)
    if (p.KH == 3 && p.pad_y == 0) return narrow ? launch_convt<3, 0, 1>(p, st) : launch_convt<3, 0, 2>(p, st);
    if (p.KH == 3 && p.pad_y == 1) return narrow ? launch_convt<3, 1, 1>(p, st) : launch_convt<3, 1, 2>(p, st);
    if (p.KH == 7 && p.pad_y == 3) return launch_convt<7, 3, 2>(p, st);
    return l2i_set_error(L2I_E_UNSUPPORTED, "conv_transpose2d: fused kernel built for (K,pad) in {(3,0),(3,1),(7,3)}");
}

