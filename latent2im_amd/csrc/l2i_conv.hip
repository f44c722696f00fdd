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
// Block = 256 threads = 4 waves, all along N: block tile = (WM*32 channels) x (4*WN*32 pixels).  The pixel tile is
// TB samples x TH rows x TW columns (all powers of two, TB*TH*TW = 128*WN): large maps use TB = 1, TW = 32; maps
// smaller than the tile (4x4 .. 16x16 layers, and their odd-sized phase maps) pack several samples into one tile.
//
// Pipeline per K chunk (CK input channels): the NEXT chunk's global loads are issued into registers before the
// MFMA loop of the current chunk (independent loads, nothing waits on them), and written to LDS after it — HBM/L2
// latency hides under ~10^4 cycles of matrix work; 2 blocks per CU cover the two barriers per chunk.  Prologue
// fusions (style/demod scale, activation-gradient mask) are applied on the register->LDS write.  1x1 stride-1
// layers (half of ResNet-50) stage 16-byte vectors.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvLaunch {
    int tw_log2, th_log2, tb_log2;     // pixel tile = 2^tb samples x 2^th rows x 2^tw columns
    int tiles_x, tiles_y, bgroups, mblocks;
    int CK;                            // channels per LDS chunk (even)
    int IH, IW, IWp;                   // staged input rows / columns / padded pitch per sample
    int planeS, plane;                 // floats per (channel, sample) and per channel (>= TB*planeS, multiple of 4)
    int rows_c;                        // TB*IH: staged rows per channel
    unsigned magic_iw, magic_rc, magic_ih;   // ceil(2^32/d) for d = IW, rows_c, IH
    int in_elems;                      // CK*rows_c*IW   (VEC: in float4 units)
    int w_vec;                         // CK*KK*BM/4
    int vec_epi;                       // 1: LDS-transposed epilogue with 16-byte global accesses
    int lstride, gstep;                // LDS-coordinate stride of the fragment reads / global step between staged elements
    int total;                         // blocks with work (the grid is padded to a multiple of 8 for the XCD renumbering)
    int ksplit, cin_per;               // split-K: ksplit channel ranges of cin_per (multiple of CK) channels, raw partial sums into p.ws
                                       // (stride-2 1x1 convs gather only the pixels they use: lstride 1, gstep 2)
};

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }   // magic 0 encodes d == 1

template <int BM> struct WSlots { static constexpr int value = (BM == 128) ? 10 : (BM == 64 ? 8 : 4); };

// NIN: input slots per thread per chunk (floats, or float4 when VEC); NWV: weight float4 slots per thread per chunk
// KT > 0: square KT x KT kernel known at compile time (the 3x3 stride-2 / small-map layers and the strided 1x1s on the hot tile): the tap loop
// unrolls into immediate LDS offsets, its address arithmetic and loop control leave the MFMA stream and hipcc fetches the next taps'
// fragments ahead of the MFMAs that use them (round-1 finding: 3.3 VALU per MFMA and 55 % matrix-pipe busy with run-time tap loops)
template <int WM, int WN, bool VEC, bool MASK, int KT = 0>
__global__ __launch_bounds__(256, (WM * WN <= 4) ? 3 : ((WM == 2 && WN == 4) ? 2 : 1)) void conv_mfma_kernel(const l2i_conv_params p, const ConvLaunch L) {
    constexpr int BM = WM * 32;
    constexpr int NIN = VEC ? (MASK ? 4 : 8) : 12;
    constexpr int NWV = WSlots<BM>::value;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_in = smem;                               // [CK][TB][IH][IWp]
    float* lds_w = smem + L.CK * L.plane;               // [CK][KK][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int j = lane & 31;
    const int KK = KT ? KT * KT : p.KH * p.KW;
    const int TW = 1 << L.tw_log2, TH = 1 << L.th_log2;

    // ---- block -> (sample group, tile, channel block).  Blocks are dealt round-robin to the 8 XCDs: renumber them so that the channel
    //      blocks of one pixel tile (and its neighbours) run back to back on ONE XCD and find the input tile in its L2 (round 1 measured
    //      1.41x the algorithmic HBM bytes on the conv kernels: every XCD fetched every tile) ----
    int bid = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (bid >= L.total) return;
    const int mblk = bid % L.mblocks; bid /= L.mblocks;
    const int tx = bid % L.tiles_x; bid /= L.tiles_x;
    const int ty = bid % L.tiles_y; bid /= L.tiles_y;
    const int bgrp = bid % L.bgroups;
    const int split = bid / L.bgroups;                  // 0 unless split-K
    const int b0 = bgrp << L.tb_log2;
    const int m0 = mblk * BM;
    const int oy0 = ty << L.th_log2, ox0 = tx << L.tw_log2;
    const int iy0 = oy0 * p.stride - p.pad_y, ix0 = ox0 * p.stride - p.pad_x;
    const int c_begin = split * L.cin_per;
    const int c_end = (c_begin + L.cin_per < p.Cin) ? c_begin + L.cin_per : p.Cin;

    // ---- per-lane fragment bases: pixel index within the tile = q*32 + j -> (tb, row, col) ----
    const int CKh = L.CK >> 1;
    int pixoff[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int pi = (wave * WN + n) * 32 + j;
        const int c = pi & (TW - 1);
        const int r = (pi >> L.tw_log2) & (TH - 1);
        const int tb = pi >> (L.tw_log2 + L.th_log2);
        pixoff[n] = tb * L.planeS + (r * L.lstride) * L.IWp + c * L.lstride + half * CKh * L.plane;
    }
    const int wlane = half * CKh * KK * BM + j;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- staging state: every global access of the K loop is a buffer load "descriptor + per-slot VGPR offset +
    //      per-chunk SGPR offset": no address arithmetic and no bounds branches inside the loop (out-of-range = 0) ----
    const size_t plane_x = (size_t)p.H * p.W;
    const int nb = (p.B - b0) < (1 << L.tb_log2) ? (p.B - b0) : (1 << L.tb_log2);
    const unsigned in_bytes = (unsigned)((size_t)nb * p.Cin * plane_x * sizeof(float));   // host guarantees < 4 GiB
    const size_t grp_off = (size_t)b0 * p.Cin * plane_x;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + grp_off), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + grp_off), 0, in_bytes, 0x00020000);
    const unsigned w_bytes = (unsigned)((size_t)p.Cin * KK * p.CoutP * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
    const unsigned sc_bytes = (unsigned)((size_t)nb * p.Cin * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.in_scale ? p.in_scale : p.x) + (size_t)b0 * p.Cin), 0, p.in_scale ? sc_bytes : 0u, 0x00020000);

    typedef typename std::conditional<VEC, u32x4, unsigned>::type in_t;
    in_t rin[NIN];
    in_t rmk[MASK ? NIN : 1];
    u32x4 rw[NWV];
    unsigned rsc = 0;
    unsigned voff[NIN];          // byte offset of the slot inside the sample group (channel 0 of the chunk), or out of range
    int loff[NIN];               // float offset of the slot inside lds_in, or -1
    const int iw_units = VEC ? (L.IW >> 2) : L.IW;
#pragma unroll
    for (int u = 0; u < NIN; ++u) {
        const unsigned e = tid + u * 256;
        voff[u] = in_bytes;
        loff[u] = -1;
        if ((int)e < L.in_elems) {
            unsigned row, ixu;
            if constexpr (VEC) { row = e >> (L.tw_log2 - 2); ixu = e & (iw_units - 1); }
            else { row = fast_div(e, L.magic_iw); ixu = e - row * L.IW; }
            const unsigned c = fast_div(row, L.magic_rc);
            const unsigned r2 = row - c * L.rows_c;
            const unsigned tb = fast_div(r2, L.magic_ih);
            const int iy = (int)(r2 - tb * L.IH);
            const int gy = iy0 + iy * L.gstep;
            const int gx = ix0 + (int)(VEC ? ixu * 4 : ixu) * L.gstep;
            loff[u] = (int)(c * L.plane + tb * L.planeS + iy * L.IWp + (VEC ? ixu * 4 : ixu));
            if ((int)tb < nb && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                voff[u] = (unsigned)((((size_t)tb * p.Cin + c) * plane_x + (size_t)gy * p.W + gx) * sizeof(float));
        }
    }
    constexpr int V = BM / 4;
    const int wrow0 = tid / V, wc4 = tid - wrow0 * V;
    const unsigned wvoff = (m0 + wc4 * 4 < p.CoutP) ? (unsigned)(((size_t)wrow0 * p.CoutP + m0 + wc4 * 4) * sizeof(float)) : w_bytes;
    const unsigned wstep = (unsigned)((256 / V) * p.CoutP * sizeof(float));                // slot u = rows wrow0 + u*256/V
    const int TBCK = L.CK << L.tb_log2;
    const unsigned scoff = (tid < TBCK) ? (unsigned)((((tid / L.CK) * p.Cin) + (tid % L.CK)) * sizeof(float)) : sc_bytes;
    float* lds_sc = lds_w + L.CK * KK * BM;              // [TB][CK] input scales of the chunk (ones when absent)

    // issue the global loads of chunk c0 into registers (no waits, no VALU)
    auto issue = [&](int c0) {
        const unsigned so = (unsigned)((size_t)c0 * plane_x * sizeof(float));
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            if constexpr (VEC) {
                rin[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff[u], so, 0);
                if constexpr (MASK) rmk[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, voff[u], so, 0);
            } else {
                rin[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff[u], so, 0);
                if constexpr (MASK) rmk[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, voff[u], so, 0);
            }
        }
        const unsigned sw = (unsigned)((size_t)c0 * KK * p.CoutP * sizeof(float));
#pragma unroll
        for (int u = 0; u < NWV; ++u) rw[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff, sw + u * wstep, 0);
        rsc = __builtin_amdgcn_raw_buffer_load_b32(rs_s, scoff, (unsigned)(c0 * sizeof(float)), 0);
    };

    // write the prefetched registers to LDS (activation-gradient mask applied here; the style/demod scale goes to
    // a small LDS table that the MFMA loop multiplies into the B fragments)
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            if (loff[u] >= 0) {
                if constexpr (VEC) {
                    float4 v = make_float4(__uint_as_float(rin[u].x), __uint_as_float(rin[u].y), __uint_as_float(rin[u].z), __uint_as_float(rin[u].w));
                    if constexpr (MASK) {
                        v.x *= (__uint_as_float(rmk[u].x) > 0.f) ? p.mask_pos : p.mask_neg;
                        v.y *= (__uint_as_float(rmk[u].y) > 0.f) ? p.mask_pos : p.mask_neg;
                        v.z *= (__uint_as_float(rmk[u].z) > 0.f) ? p.mask_pos : p.mask_neg;
                        v.w *= (__uint_as_float(rmk[u].w) > 0.f) ? p.mask_pos : p.mask_neg;
                    }
                    *reinterpret_cast<float4*>(lds_in + loff[u]) = v;
                } else {
                    float v = __uint_as_float(rin[u]);
                    if constexpr (MASK) v *= (__uint_as_float(rmk[u]) > 0.f) ? p.mask_pos : p.mask_neg;
                    lds_in[loff[u]] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int idx = tid + u * 256;
            if (idx < L.w_vec) reinterpret_cast<u32x4*>(lds_w)[idx] = rw[u];
        }
        // one scale per (sample of the tile, channel of the chunk).  With a style scale the host caps CK so that the table fits one
        // entry per thread (TB * CK <= 256); without one the table is all ones and may be longer than the block (tiny maps pack up to
        // 128 samples into a tile: ResNet-50's tail at small inputs)
        if (p.in_scale) { if (tid < TBCK) lds_sc[tid] = __uint_as_float(rsc); }
        else for (int i = tid; i < TBCK; i += 256) lds_sc[i] = 1.f;
    };

    int sbase[WN];               // per N tile: index of this lane's (sample, first channel) in the scale table
#pragma unroll
    for (int n = 0; n < WN; ++n) sbase[n] = ((((wave * WN + n) * 32 + j) >> (L.tw_log2 + L.th_log2)) * L.CK) + half * CKh;

    issue(c_begin);
    for (int c0 = c_begin; c0 < c_end; c0 += L.CK) {
        commit();
        __syncthreads();
        if (c0 + L.CK < c_end) issue(c0 + L.CK);        // in flight during the MFMA loop below
        // ---- MFMA over the chunk: lanes 0-31 take channel cc, lanes 32-63 channel cc + CK/2 ----
        for (int cc = 0; cc < CKh; ++cc) {
            const float* wr = lds_w + wlane + cc * KK * BM;
            const float* ir = lds_in + cc * L.plane;
            float sv[WN];
#pragma unroll
            for (int n = 0; n < WN; ++n) sv[n] = lds_sc[sbase[n] + cc];
            if constexpr (KT > 0) {
                const float* ib[WN];
#pragma unroll
                for (int n = 0; n < WN; ++n) ib[n] = ir + pixoff[n];
#pragma unroll
                for (int ky = 0; ky < KT; ++ky) {
                    const int roff = ky * L.IWp;
#pragma unroll
                    for (int kx = 0; kx < KT; ++kx) {
                        float a[WM], bb[WN];
#pragma unroll
                        for (int m = 0; m < WM; ++m) a[m] = wr[(ky * KT + kx) * BM + m * 32];
#pragma unroll
                        for (int n = 0; n < WN; ++n) bb[n] = ib[n][roff + kx] * sv[n];
#pragma unroll
                        for (int m = 0; m < WM; ++m)
#pragma unroll
                            for (int n = 0; n < WN; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
                    }
                }
            } else {
                for (int ky = 0; ky < p.KH; ++ky) {
                    for (int kx = 0; kx < p.KW; ++kx) {
                        float a[WM], bb[WN];
                        const int toff = ky * L.IWp + kx;
#pragma unroll
                        for (int m = 0; m < WM; ++m) a[m] = wr[m * 32];
#pragma unroll
                        for (int n = 0; n < WN; ++n) bb[n] = ir[pixoff[n] + toff] * sv[n];
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
        __syncthreads();        // every wave is done reading this chunk's fragments
    }

    // ---- epilogue A (wide): accumulators -> per-wave LDS transpose -> 16-byte loads/stores.  In the MFMA layout a
    //      lane owns ONE pixel of 16 channels, so direct stores are 4 B per lane and the store path (not HBM) bounds
    //      low-K layers; after the transpose a lane owns 4 consecutive pixels of one channel: 4x fewer, 4x wider
    //      global instructions for y, residual, masks and noise alike.
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    if (L.ksplit > 1) {
        // split-K: raw partial sums of this channel range, laid out like y; splitk_epilogue_kernel reduces them and applies the fusions
        float* wsp = p.ws + (size_t)split * p.B * p.Cout * plane_o;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int pi = (wave * WN + n) * 32 + j;
            const int ox = ox0 + (pi & (TW - 1));
            const int oy = oy0 + ((pi >> L.tw_log2) & (TH - 1));
            const int bb = b0 + (pi >> (L.tw_log2 + L.th_log2));
            if (oy < p.OH && ox < p.OW && bb < p.B) {
                const size_t poff = (size_t)(oy * p.oy_step + p.oy_off) * p.OWf + ox * p.ox_step + p.ox_off;
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m0 + 4 * half + m * 32 + (r & 3) + 8 * (r >> 2);
                        if (co < p.Cout) wsp[((size_t)bb * p.Cout + co) * plane_o + poff] = acc[m][n][r];
                    }
            }
        }
        return;
    }
    if (L.vec_epi) {
        float* reg = smem + wave * (32 * 64);           // [32 channels][64 pixels] of this wave; the K loop is done
        const int ch_l = lane >> 4;                     // channel row within a group of 4
        const int px = (lane & 15) * 4;                 // first of this lane's 4 pixels inside the 64-pixel strip
#pragma unroll
        for (int n0 = 0; n0 < WN; n0 += 2) {
            const int nn = px >> 5;
            const int pi = (wave * WN + n0 + nn) * 32 + (px & 31);
            const int ox = ox0 + (pi & (TW - 1));
            const int oy = oy0 + ((pi >> L.tw_log2) & (TH - 1));
            const int bb = b0 + (pi >> (L.tw_log2 + L.th_log2));
            const bool pok = (n0 + nn < WN) && (oy < p.OH) && (ox < p.OW) && (bb < p.B);
            const size_t poff = (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pok && p.noise) {
                nz = *reinterpret_cast<const float4*>(p.noise + (size_t)bb * plane_o + poff);
                nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
            }
            const float* osc = (pok && p.out_scale) ? p.out_scale + (size_t)bb * p.Cout : nullptr;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (n0 + q < WN) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            reg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + q * 32 + j] = acc[m][(n0 + q) < WN ? (n0 + q) : 0][r];
                    }
                }
#pragma unroll 2
                for (int i = 0; i < 8; ++i) {
                    const int ch = i * 4 + ch_l;
                    const int co = m0 + m * 32 + ch;
                    const float4 t = *reinterpret_cast<const float4*>(&reg[ch * 64 + px]);
                    if (pok && co < p.Cout) {
                        float4 v = t;
                        if (osc) { const float sc = osc[co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                        const size_t oidx = ((size_t)bb * p.Cout + co) * plane_o + poff;
                        if (p.out_mask) {
                            const float4 mk = *reinterpret_cast<const float4*>(p.out_mask + oidx);
                            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                        }
                        if (p.noise) { v.x += nz.x; v.y += nz.y; v.z += nz.z; v.w += nz.w; }
                        if (p.bias) { const float bv = p.bias[co]; v.x += bv; v.y += bv; v.z += bv; v.w += bv; }
                        if (p.residual) {
                            float4 rv = *reinterpret_cast<const float4*>(p.residual + oidx);
                            if (p.res_sub) {                                   // residual term = res_coef * (residual - res_sub)
                                const float4 sb = *reinterpret_cast<const float4*>(p.res_sub + oidx);
                                const float rc = p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f);
                                rv.x = rc * (rv.x - sb.x); rv.y = rc * (rv.y - sb.y); rv.z = rc * (rv.z - sb.z); rv.w = rc * (rv.w - sb.w);
                            }
                            if (p.res_mask) {
                                const float4 mk = *reinterpret_cast<const float4*>(p.res_mask + oidx);
                                rv.x = mk.x > 0.f ? rv.x : 0.f; rv.y = mk.y > 0.f ? rv.y : 0.f; rv.z = mk.z > 0.f ? rv.z : 0.f; rv.w = mk.w > 0.f ? rv.w : 0.f;
                            }
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                        if (p.act == L2I_ACT_LRELU) {
                            v.x = (v.x > 0.f ? v.x : v.x * p.act_slope) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.act_slope) * p.act_gain;
                            v.z = (v.z > 0.f ? v.z : v.z * p.act_slope) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.act_slope) * p.act_gain;
                        } else if (p.act == L2I_ACT_RELU) {
                            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                        }
                        if (p.out_gain != 1.f) { v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain; }
                        if (p.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(p.y + oidx) = v;
                    }
                }
            }
        }
        return;
    }

    // ---- epilogue B (scalar): strided phase outputs, odd widths, unaligned tensors ----
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int pi = (wave * WN + n) * 32 + j;
        const int ox = ox0 + (pi & (TW - 1));
        const int oy = oy0 + ((pi >> L.tw_log2) & (TH - 1));
        const int bb = b0 + (pi >> (L.tw_log2 + L.th_log2));
        const bool pok = (oy < p.OH) && (ox < p.OW) && (bb < p.B);
        const int oyf = oy * p.oy_step + p.oy_off, oxf = ox * p.ox_step + p.ox_off;
        const size_t poff = (size_t)oyf * p.OWf + oxf;
        float nz = 0.f;
        if (pok && p.noise) nz = p.noise[(size_t)bb * plane_o + poff] * p.noise_w;
        const int co_lane = m0 + 4 * half;
        const size_t lane_base = ((size_t)bb * p.Cout + co_lane) * plane_o + poff;
        const float* osc = p.out_scale ? p.out_scale + (size_t)bb * p.Cout : nullptr;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cofs = m * 32 + (r & 3) + 8 * (r >> 2);          // compile-time constant
                const int co = co_lane + cofs;
                if (pok && co < p.Cout) {
                    float v = acc[m][n][r];
                    if (osc) v *= osc[co];
                    const size_t oidx = lane_base + (size_t)cofs * plane_o;
                    if (p.out_mask) v = (p.out_mask[oidx] > 0.f) ? v : 0.f;
                    v += nz;
                    if (p.bias) v += p.bias[co];
                    if (p.residual) {
                        float rv = p.residual[oidx];
                        if (p.res_sub) rv = p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) * (rv - p.res_sub[oidx]);
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
// Direct (VALU) path for layers with <= 4 output channels: the input-gradients that land on an RGB image (VGG conv1_1,
// discriminator from_rgb, the per-parity pieces of the ResNet stem).  On the matrix path these fill 3 of the 32 rows of
// an MFMA tile (11 TFLOP/s measured); here a thread owns a 4x1 column of pixels x 4 channels, reads its inputs from an
// LDS tile and each tap's 4 weights as one broadcast float4.
// TKH x TKW: compile-time window (0 = run-time p.KH x p.KW): the tap loop unrolls, the thread's (TKH+3) x TKW input window is read
// once per channel and every LDS read is in flight before the first FMA (the rolled loop waited out one LDS latency per tap)
template <bool MASK, int TKH, int TKW>
__global__ __launch_bounds__(256) void conv_direct_small_kernel(const l2i_conv_params p, int tiles_x, int tiles_y, int CK, int IH, int IW, int IWp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = p.KH * p.KW;
    float* tile = smem;                                  // [CK][IH][IWp]
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid;
    const int oy0 = ty * 32, ox0 = tx * 32;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
    const int col = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    const size_t plane_x = (size_t)p.H * p.W;
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[r][o] = 0.f;

    for (int c0 = 0; c0 < p.Cin; c0 += CK) {
        __syncthreads();
        // staging: thread = (row group, 4-column slot); 16-byte global loads (4-byte alignment suffices on gfx950), rows of the
        // [CK][IH] stack walked 16 apart with an incremental (channel, iy) pair — no integer divisions, 5 loads in flight per thread
        {
            const int sq = threadIdx.x & 15, sgrp = threadIdx.x >> 4;      // 16 slots x 4 columns cover IW <= 64
            const int lx = sq * 4, gx = ix0 + lx;
            const bool slot_on = lx < IW;
            const bool full = slot_on && lx + 3 < IW && gx >= 0 && gx + 3 < p.W;
            const int nrows = CK * IH;
            int c = 0, iy = sgrp;
            while (iy >= IH) { iy -= IH; ++c; }
            for (int r0s = sgrp; r0s < nrows; r0s += 16 * 5) {
                float4 v[5], mk[5];
                int cs[5], iys[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    cs[u] = c; iys[u] = iy;
                    const int gy = iy0 + iy, ci = c0 + c;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f); mk[u] = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (r0s + 16 * u < nrows && slot_on && ci < p.Cin && gy >= 0 && gy < p.H) {
                        const size_t off = ((size_t)b * p.Cin + ci) * plane_x + (size_t)gy * p.W;
                        if (full) {
                            v[u] = *reinterpret_cast<const float4*>(p.x + off + gx);
                            if (MASK) mk[u] = *reinterpret_cast<const float4*>(p.in_mask + off + gx);
                        } else {
                            float* vv = reinterpret_cast<float*>(&v[u]);
                            float* mm = reinterpret_cast<float*>(&mk[u]);
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (lx + q < IW && gx + q >= 0 && gx + q < p.W) {
                                    vv[q] = p.x[off + gx + q];
                                    if (MASK) mm[q] = p.in_mask[off + gx + q];
                                }
                        }
                    }
                    iy += 16;
                    while (iy >= IH) { iy -= IH; ++c; }
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    if (r0s + 16 * u < nrows && slot_on) {
                        float sc = 1.f;
                        if (p.in_scale && c0 + cs[u] < p.Cin) sc = p.in_scale[(size_t)b * p.Cin + c0 + cs[u]];
                        const float* vv = reinterpret_cast<const float*>(&v[u]);
                        const float* mm = reinterpret_cast<const float*>(&mk[u]);
                        float* dst = tile + (cs[u] * IH + iys[u]) * IWp + lx;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (lx + q < IW) {
                                float t = vv[q] * sc;
                                if (MASK) t *= (mm[q] > 0.f) ? p.mask_pos : p.mask_neg;
                                dst[q] = t;
                            }
                    }
                }
            }
        }
        __syncthreads();
        // the taps are the same for every lane: they come through the scalar cache into SGPRs (constant address space -> s_load; one SGPR
        // operand per FMA) instead of LDS broadcasts, which cost 8 LDS cycles per 16 FMAs and bounded this kernel (see l2i_convt_small.hip)
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(4))) const f32x4v cfloat4;
        const int cn = p.Cin - c0 < CK ? p.Cin - c0 : CK;
        for (int c = 0; c < cn; ++c) {
            const float* tc = tile + (c * IH + r0) * IWp + col;
            const float* wbase = p.w + (size_t)(c0 + c) * KK * p.CoutP;
            auto tap = [&](int t) -> f32x4v { return *(cfloat4*)(uintptr_t)(wbase + (size_t)t * p.CoutP); };
            if constexpr (TKH > 0) {
                float win[TKH + 3][TKW];
#pragma unroll
                for (int y = 0; y < TKH + 3; ++y)
#pragma unroll
                    for (int xk = 0; xk < TKW; ++xk) win[y][xk] = tc[y * IWp + xk];
#pragma unroll
                for (int ky = 0; ky < TKH; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < TKW; ++kx) {
                        const f32x4v w4 = tap(ky * TKW + kx);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x = win[r + ky][kx];
                            acc[r][0] += x * w4.x; acc[r][1] += x * w4.y; acc[r][2] += x * w4.z; acc[r][3] += x * w4.w;
                        }
                    }
                }
            } else {
                for (int ky = 0; ky < p.KH; ++ky) {
                    for (int kx = 0; kx < p.KW; ++kx) {
                        const f32x4v w4 = tap(ky * p.KW + kx);
                        const float* tr = tc + ky * IWp + kx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                   // static register indexing only (no scratch)
                            const float x = tr[r * IWp];
                            acc[r][0] += x * w4.x; acc[r][1] += x * w4.y; acc[r][2] += x * w4.z; acc[r][3] += x * w4.w;
                        }
                    }
                }
            }
        }
    }
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const int ox = ox0 + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = oy0 + r0 + r;
        if (oy < p.OH && ox < p.OW) {
            const size_t poff = (size_t)(oy * p.oy_step + p.oy_off) * p.OWf + ox * p.ox_step + p.ox_off;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < p.Cout) {
                    const size_t oidx = ((size_t)b * p.Cout + o) * plane_o + poff;
                    float v = acc[r][o] * p.out_gain;
                    if (p.accumulate) v += p.y[oidx];
                    p.y[oidx] = v;
                }
            }
        }
    }
}

// 3x3 / stride 1 / pad 1 onto <= 3 channels on wide maps (the input-gradient of VGG-19's conv_1 at 1024^2: 64 -> 3), register-streaming like the
// FIR kernels of l2i_stream.hip: a lane owns four output columns of a 4-row band (48 accumulators); per input channel it reads the band's six
// input rows with ONE aligned 16-byte load each and takes the two halo columns from its neighbour lanes by wave shuffles (edge lanes: from
// memory); the 27 taps of the channel come through the scalar cache ([Cin][9][4] pack) — no LDS, no barriers, 432 FMAs per 6 loads.
template <bool MASK>
__global__ __launch_bounds__(256) void conv3x3_small_stream_kernel(const l2i_conv_params p, int bands, int strips) {
    constexpr int RS = 4;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(4))) const f32x4v cfloat4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long bid = blockIdx.x;
    const int strip = (int)(bid % strips); bid /= strips;
    const int band = (int)(bid % bands);
    const int b = (int)(bid / bands);
    // a wave covers 248 output columns: lanes 1 .. 62 own four columns each, lanes 0 and 63 only fetch the halo vectors of their neighbours (a
    // divergent scalar load for the two edge lanes stalled the whole wave on its latency: twelve times per channel)
    const int ox = strip * 248 + (lane - 1) * 4;
    const int oy0 = (band * 4 + wave) * RS;
    if (oy0 >= p.OH) return;
    const size_t plane_x = (size_t)p.H * p.W;
    float acc[RS][4][3];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < 3; ++o) acc[r][q][o] = 0.f;
    // the six row vectors of channel c + 1 are in flight while channel c is on the VALU (0.91 -> 0.78 ms at 1024^2; the kernel is then bound by
    // VALU issue: hipcc packs the FMAs into v_pk_fma_f32 but moves every scalar-cache tap into a VGPR pair first, as many v_mov as FMAs)
    float4 raw[RS + 2], rawm[MASK ? RS + 2 : 1];
    auto fetch = [&](int c) {
        const float* xc = p.x + ((size_t)b * p.Cin + c) * plane_x;
#pragma unroll
        for (int r = 0; r < RS + 2; ++r) {
            const int iy = oy0 - 1 + r;
            raw[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (MASK) rawm[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ox >= 0 && ox < p.W) {
                raw[r] = *reinterpret_cast<const float4*>(xc + (size_t)iy * p.W + ox);
                if constexpr (MASK) rawm[r] = *reinterpret_cast<const float4*>(p.in_mask + ((size_t)b * p.Cin + c) * plane_x + (size_t)iy * p.W + ox);
            }
        }
    };
    fetch(0);
    for (int c = 0; c < p.Cin; ++c) {
        float win[RS + 2][6];                                          // rows oy0 - 1 .. oy0 + RS, columns ox - 1 .. ox + 4
#pragma unroll
        for (int r = 0; r < RS + 2; ++r) {
            float4 v = raw[r];
            if constexpr (MASK) {
                const float4 m = rawm[r];
                v.x *= m.x > 0.f ? p.mask_pos : p.mask_neg; v.y *= m.y > 0.f ? p.mask_pos : p.mask_neg;
                v.z *= m.z > 0.f ? p.mask_pos : p.mask_neg; v.w *= m.w > 0.f ? p.mask_pos : p.mask_neg;
            }
            const float l1 = __shfl_up(v.w, 1), r1 = __shfl_down(v.x, 1);
            win[r][0] = l1; win[r][1] = v.x; win[r][2] = v.y; win[r][3] = v.z; win[r][4] = v.w; win[r][5] = r1;
        }
        if (c + 1 < p.Cin) fetch(c + 1);
        cfloat4* wc = (cfloat4*)(uintptr_t)(p.w + (size_t)c * 9 * 4);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4v w4 = wc[ky * 3 + kx];
#pragma unroll
                for (int r = 0; r < RS; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = win[r + ky][q + kx];
                        acc[r][q][0] += x * w4.x; acc[r][q][1] += x * w4.y; acc[r][q][2] += x * w4.z;
                    }
            }
    }
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    if (lane == 0 || lane == 63 || ox >= p.OW) return;                 // OW % 4 == 0: the four columns are inside together
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int oy = oy0 + r;
        if (oy >= p.OH) continue;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (o < p.Cout) {
                float* yp = p.y + ((size_t)b * p.Cout + o) * plane_o + (size_t)oy * p.OWf + ox;
                float4 v = make_float4(acc[r][0][o] * p.out_gain, acc[r][1][o] * p.out_gain, acc[r][2][o] * p.out_gain, acc[r][3][o] * p.out_gain);
                if (p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(yp); v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
                *reinterpret_cast<float4*>(yp) = v;
            }
        }
    }
}

static int launch_direct_small(const l2i_conv_params& p, hipStream_t st) {
    {
        auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
        if (p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_y == 1 && p.pad_x == 1 && p.Cout <= 3 && p.CoutP == 4 && !p.in_scale && p.oy_step == 1 && p.ox_step == 1 &&
            p.oy_off == 0 && p.ox_off == 0 && p.OH == p.H && p.OW == p.W && p.OHf == p.OH && p.OWf == p.OW && p.OW >= 192 && (p.W % 4) == 0 && al16(p.x) && al16(p.in_mask) &&
            al16(p.y) && al16(p.w)) {
            const int strips = (p.OW + 247) / 248, bands = (p.OH + 15) / 16;
            const long grid = (long)p.B * bands * strips;
            if (grid > 0 && grid <= 0x7fffffffL) {
                if (p.in_mask) hipLaunchKernelGGL((conv3x3_small_stream_kernel<true>), dim3((unsigned)grid), dim3(256), 0, st, p, bands, strips);
                else hipLaunchKernelGGL((conv3x3_small_stream_kernel<false>), dim3((unsigned)grid), dim3(256), 0, st, p, bands, strips);
                L2I_CHECK_LAUNCH();
                return L2I_OK;
            }
        }
    }
    const int IH = 32 + p.KH - 1, IW = 32 + p.KW - 1, IWp = IW | 1, KK = p.KH * p.KW;
    const size_t per_c = (size_t)IH * IWp * sizeof(float) + (size_t)KK * 16;
    int ck = (int)((40 * 1024) / per_c);
    if (ck > 16) ck = 16;
    if (ck > p.Cin) ck = p.Cin;
    if (ck < 1) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d(direct): kernel window too large");
    const size_t lds = (((size_t)ck * IH * IWp + 3) & ~(size_t)3) * sizeof(float) + (size_t)ck * KK * 16;
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 31) / 32;
    const long grid = (long)p.B * tiles_x * tiles_y;
    if (grid <= 0 || grid > 0x7fffffffL) return l2i_set_error(L2I_E_ARG, "conv2d(direct): grid too large");
#define L2I_SMALL(TKH_, TKW_)                                                                                                              \
    do {                                                                                                                              \
        if (p.in_mask) hipLaunchKernelGGL((conv_direct_small_kernel<true, TKH_, TKW_>), dim3((unsigned)grid), dim3(256), lds, st, p, tiles_x, tiles_y, ck, IH, IW, IWp); \
        else hipLaunchKernelGGL((conv_direct_small_kernel<false, TKH_, TKW_>), dim3((unsigned)grid), dim3(256), lds, st, p, tiles_x, tiles_y, ck, IH, IW, IWp);          \
    } while (0)
    // windows on the path: 3x3 (VGG / discriminator gradients onto RGB), 1x1 (from_rgb), and the 4x4 / 4x3 / 3x4 / 3x3 parity pieces of
    // the ResNet stem's 7x7 stride-2 gradient
    if (p.KH == 3 && p.KW == 3) L2I_SMALL(3, 3);
    else if (p.KH == 4 && p.KW == 4) L2I_SMALL(4, 4);
    else if (p.KH == 4 && p.KW == 3) L2I_SMALL(4, 3);
    else if (p.KH == 3 && p.KW == 4) L2I_SMALL(3, 4);
    else if (p.KH == 1 && p.KW == 1) L2I_SMALL(1, 1);
    else L2I_SMALL(0, 0);
#undef L2I_SMALL
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Split-K second pass: y = epilogue(sum_s ws[s]) over the output window of the launch, same fusion order as the conv epilogues.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const l2i_conv_params p, long long total) {
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const size_t per_split = (size_t)p.B * p.Cout * plane_o;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % p.OW);
        const int oy = (int)((i / p.OW) % p.OH);
        const long long bc = i / ((long long)p.OW * p.OH);
        const int co = (int)(bc % p.Cout), bb = (int)(bc / p.Cout);
        const size_t poff = (size_t)(oy * p.oy_step + p.oy_off) * p.OWf + ox * p.ox_step + p.ox_off;
        const size_t oidx = ((size_t)bb * p.Cout + co) * plane_o + poff;
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.ws[(size_t)s * per_split + oidx];
        if (p.out_scale) v *= p.out_scale[(size_t)bb * p.Cout + co];
        if (p.out_mask) v = (p.out_mask[oidx] > 0.f) ? v : 0.f;
        if (p.noise) v += p.noise[(size_t)bb * plane_o + poff] * p.noise_w;
        if (p.bias) v += p.bias[co];
        if (p.residual) {
            float rv = p.residual[oidx];
            if (p.res_sub) rv = p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) * (rv - p.res_sub[oidx]);
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

int l2i_launch_splitk_epilogue(const l2i_conv_params& q, hipStream_t st) {
    const long long total = (long long)q.B * q.Cout * q.OH * q.OW;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(l2i_grid_for(total, 256)), dim3(256), 0, st, q, total);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

// ------------------------------------------------------------------------------------------------------------
static int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static unsigned magic_for(unsigned d) { return (unsigned)((0x100000000ULL + d - 1) / d); }    // exact while n*d < 2^32

struct TileCfg { int wm, wn; };
static const TileCfg kTiles[] = {{4, 2}, {2, 2}, {1, 4}, {2, 1}, {1, 1}, {1, 2}, {4, 1}, {2, 4}};
static const int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int WM, int WN, bool VEC, bool MASK>
static hipError_t launch_one(const l2i_conv_params& p, const ConvLaunch& L, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, VEC, MASK>), dim3(grid), dim3(256), lds, st, p, L);
    return hipGetLastError();
}

template <int WM, int WN, bool MASK, int KT>
static hipError_t launch_one_kt(const l2i_conv_params& p, const ConvLaunch& L, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, false, MASK, KT>), dim3(grid), dim3(256), lds, st, p, L);
    return hipGetLastError();
}

template <int WM, int WN>
static hipError_t launch_cfg(const l2i_conv_params& p, const ConvLaunch& L, int grid, size_t lds, bool vec, hipStream_t st) {
    const bool mask = p.in_mask != nullptr;
    if constexpr (WM == 2 && WN == 2) {                 // the hot tile of the stride-2 / small-map layers: compile-time tap loops
        if (!vec && p.KH == p.KW && (p.KH == 3 || p.KH == 1)) {
            hipError_t (*fn)(const l2i_conv_params&, const ConvLaunch&, int, size_t, hipStream_t) = nullptr;
            if (p.KH == 3) fn = mask ? &launch_one_kt<2, 2, true, 3> : &launch_one_kt<2, 2, false, 3>;
            else fn = mask ? &launch_one_kt<2, 2, true, 1> : &launch_one_kt<2, 2, false, 1>;
            return fn(p, L, grid, lds, st);
        }
    }
    if (vec) return mask ? launch_one<WM, WN, true, true>(p, L, grid, lds, st) : launch_one<WM, WN, true, false>(p, L, grid, lds, st);
    return mask ? launch_one<WM, WN, false, true>(p, L, grid, lds, st) : launch_one<WM, WN, false, false>(p, L, grid, lds, st);
}

// geometry of one tile configuration for this problem; returns false if it cannot be staged
static bool plan_tile(const l2i_conv_params& p, int wm, int wn, ConvLaunch& L, bool& vec, size_t& lds, long& grid) {
    const int BM = wm * 32, BN = 128 * wn;
    const int nwv = (BM == 128) ? 10 : (BM == 64 ? 8 : 4);
    L.tw_log2 = ilog2_ceil(p.OW < 32 ? p.OW : 32);
    const int TW = 1 << L.tw_log2;
    int th = BN / TW;
    const int oh_p2 = 1 << ilog2_ceil(p.OH);
    if (th > oh_p2) th = oh_p2;
    L.th_log2 = ilog2_ceil(th);
    const int TH = 1 << L.th_log2;
    L.tb_log2 = ilog2_ceil(BN / (TH * TW));
    const int TB = 1 << L.tb_log2;
    if ((size_t)TB * p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull) return false;     // 32-bit buffer offsets per sample group
    L.tiles_x = (p.OW + TW - 1) / TW;
    L.tiles_y = (p.OH + TH - 1) / TH;
    L.bgroups = (p.B + TB - 1) / TB;
    L.mblocks = (p.CoutP + BM - 1) / BM;
    const bool gather = (p.KH == 1 && p.KW == 1 && p.stride == 2);
    L.lstride = gather ? 1 : p.stride;
    L.gstep = gather ? 2 : 1;
    L.IH = (TH - 1) * L.lstride + p.KH;
    L.IW = (TW - 1) * L.lstride + p.KW;
    vec = (p.KW == 1 && p.KH == 1 && p.stride == 1 && p.pad_x == 0 && p.pad_y == 0 && (p.W % 4) == 0 && TW >= 4 &&
           (((uintptr_t)p.x) % 16) == 0 && (!p.in_mask || (((uintptr_t)p.in_mask) % 16) == 0));
    L.IWp = vec ? L.IW : (L.IW | 1);
    L.planeS = L.IH * L.IWp;
    L.plane = (TB * L.planeS + 3) & ~3;
    L.rows_c = TB * L.IH;
    const int KK = p.KH * p.KW;
    // channels per chunk: bounded by LDS (48 KiB -> 2-3 blocks per CU), by the register prefetch slots and by Cin
    const size_t per_c = (size_t)(L.plane + KK * BM) * sizeof(float);
    int ck = (int)((48 * 1024) / per_c);
    const int in_per_c = vec ? (L.rows_c * (L.IW / 4)) : (L.rows_c * L.IW);
    const int ck_in = ((vec ? (p.in_mask ? 4 : 8) : 12) * 256) / in_per_c;
    const int ck_w = (nwv * 256) / (KK * BM / 4);
    if (ck > ck_in) ck = ck_in;
    if (ck > ck_w) ck = ck_w;
    if (p.in_scale && ck > (256 >> L.tb_log2)) ck = 256 >> L.tb_log2;      // the per-(sample, channel) scale table is filled one entry per thread
    ck &= ~1;
    if (ck < 2) return false;
    const int cin_even = (p.Cin + 1) & ~1;
    if (ck > cin_even) ck = cin_even;
    L.ksplit = 1;
    L.cin_per = p.Cin;
    if (p.ksplit > 1 && p.ws) {                       // channel ranges must be whole chunks: the largest even CK that divides the range
        const int per = p.Cin / p.ksplit;
        if (per >= 2 && per * p.ksplit == p.Cin && (per % 2) == 0) {
            while (ck > 2 && (per % ck) != 0) ck -= 2;
            if ((per % ck) == 0) { L.ksplit = p.ksplit; L.cin_per = per; }
        }
    }
    L.CK = ck;
    L.in_elems = ck * in_per_c;
    L.w_vec = ck * KK * BM / 4;
    L.magic_iw = magic_for((unsigned)L.IW);
    L.magic_rc = magic_for((unsigned)L.rows_c);
    L.magic_ih = magic_for((unsigned)L.IH);
    lds = per_c * ck + (size_t)(ck << L.tb_log2) * sizeof(float);
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    L.vec_epi = (p.ox_step == 1 && p.oy_step == 1 && (p.OWf % 4) == 0 && (p.OW % 4) == 0 && (p.ox_off % 4) == 0 && L.tw_log2 >= 2 &&
                 al16(p.y) && al16(p.residual) && al16(p.res_mask) && al16(p.res_sub) && al16(p.out_mask) && al16(p.noise)) ? 1 : 0;
    if (L.vec_epi && lds < 4 * 32 * 64 * sizeof(float)) lds = 4 * 32 * 64 * sizeof(float);      // 8 KiB transpose strip per wave
    grid = (long)L.bgroups * L.tiles_y * L.tiles_x * L.mblocks * L.ksplit;
    if (!(lds <= 64 * 1024 && grid > 0 && grid <= 0x7ffffff0L)) return false;
    L.total = (int)grid;
    grid = (grid + 7) & ~7L;
    return true;
}

extern "C" int l2i_conv2d_family(const l2i_conv_params* pp) {
    if (!pp) return L2I_E_ARG;
    const l2i_conv_params& p = *pp;
    if (p.Cout <= 4 && p.stride == 1 && p.KH <= 16 && p.tile_hint == 0 && !p.out_scale && !p.noise && !p.bias && !p.residual && !p.out_mask &&
        p.act == L2I_ACT_NONE)
        return L2I_FAMILY_DIRECT_SMALL;
    if (p.tile_hint == 0 && l2i_gemm1x1_eligible(p)) return L2I_FAMILY_GEMM1X1;
    if (p.tile_hint == 0 && l2i_cin3_eligible(p)) return L2I_FAMILY_CIN3;
    return L2I_FAMILY_IMPLICIT_GEMM;
}

extern "C" int l2i_conv2d_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, false, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d: non-positive dimension");
    if (p.KH <= 0 || p.KW <= 0 || p.KH > 16 || p.KW > 16 || (p.stride != 1 && p.stride != 2))
        return l2i_set_error(L2I_E_ARG, "conv2d: kernel size must be 1..16 and stride 1 or 2");
    const bool dense4 = p.CoutP == 4 && p.Cout <= 4;      // [Cin][KH*KW][4] pack of a <= 4-channel layer: the direct VALU kernel only (its taps travel through the
                                                           // 16 KiB scalar cache: a 32-float pitch would spend a cache line per tap)
    if (!dense4 && (p.CoutP < p.Cout || (p.CoutP % 32) != 0)) return l2i_set_error(L2I_E_ARG, "conv2d: CoutP must be Cout rounded up to 32 (or 4 for the direct kernel of <= 4 channels)");
    if (dense4 && l2i_conv2d_family(pp) != L2I_FAMILY_DIRECT_SMALL) return l2i_set_error(L2I_E_ARG, "conv2d: the [Cin][K*K][4] pack is for launches of the direct <= 4-channel kernel only");
    if (p.res_sub && !p.residual) return l2i_set_error(L2I_E_ARG, "conv2d: res_sub needs residual");
    if ((p.sq_ref || p.sq_out) && !(p.sq_ref && p.sq_out && (((uintptr_t)p.sq_ref) % 16) == 0 && p.tile_hint == 0 && l2i_cin3_eligible(p)))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d: sq_ref / sq_out are fused in l2i_conv2d_wino_f32 and in the <= 3-input-channel kernel only");
    if (p.oy_step <= 0 || p.ox_step <= 0 || (p.OH - 1) * p.oy_step + p.oy_off >= p.OHf || (p.OW - 1) * p.ox_step + p.ox_off >= p.OWf ||
        p.oy_off < 0 || p.ox_off < 0)
        return l2i_set_error(L2I_E_ARG, "conv2d: output window exceeds the output tensor");
    if ((((uintptr_t)p.w) % 16) != 0) return l2i_set_error(L2I_E_ARG, "conv2d: packed weights must be 16-byte aligned");
    if ((size_t)p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull || (size_t)p.Cin * p.KH * p.KW * p.CoutP * sizeof(float) >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d: one sample / the weight pack must stay below 4 GiB (32-bit buffer offsets)");

    const int family = l2i_conv2d_family(pp);
    if (family == L2I_FAMILY_DIRECT_SMALL) return launch_direct_small(p, (hipStream_t)stream);
    if (family == L2I_FAMILY_GEMM1X1) return l2i_launch_gemm1x1(p, (hipStream_t)stream);
    if (family == L2I_FAMILY_CIN3) return l2i_launch_cin3(p, (hipStream_t)stream);
    if (p.tile_hint == 0 && l2i_conv3x3s2_eligible(p)) return l2i_launch_conv3x3s2(p, (hipStream_t)stream);      // [r5] same family, DMA-staged operands

    // ---- tile selection: minimise a simple time model  waves(grid / resident blocks) x cycles per block  ----
    //      cycles per block = MFMA issue (64 cycles each) + per-chunk barrier/commit cost + epilogue stores (hidden by co-resident blocks);
    //      resident blocks per CU from the register footprint of each instantiation and its LDS request.
    ConvLaunch L, Lbest;
    bool vec = false, vbest = false;
    size_t lds = 0, lbest = 0;
    long grid = 0, gbest = -1;
    int sel = -1;
    if (p.tile_hint > 0) {
        if (p.tile_hint > kNumTiles) return l2i_set_error(L2I_E_ARG, "conv2d: tile_hint out of range");
        sel = p.tile_hint - 1;
        if (!plan_tile(p, kTiles[sel].wm, kTiles[sel].wn, Lbest, vbest, lbest, gbest))
            return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d: requested tile cannot be staged for this problem");
    } else {
        // candidates: index into kTiles, blocks per CU allowed by VGPR+AGPR, measured MFMA-loop efficiency relative to (2,2)
        // ((2,4) at 2 blocks/CU wins isolated micro-benchmarks by 3-10 % but loses in the full step — masked/scaled variants spill)
        static const int kCand[6] = {1, 2, 3, 4, 0, 0};
        static const int kOcc[6] = {3, 3, 3, 4, 1, 1};
        static const double kEff[6] = {1.0, 1.0, 1.04, 1.08, 1.0, 1.0};
        double best_cost = 0.0;
        for (int ci = 0; ci < 5; ++ci) {
            const int i = kCand[ci];
            if (kTiles[i].wm * 32 > p.CoutP) continue;
            if (!plan_tile(p, kTiles[i].wm, kTiles[i].wn, L, vec, lds, grid)) continue;
            int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
            if (per_cu > kOcc[ci]) per_cu = kOcc[ci];
            if (per_cu < 1) per_cu = 1;
            const long rounds = (grid + 255) / 256;                   // blocks each CU works through (all share its 4 matrix pipes)
            const int nchunks = (p.Cin + L.CK - 1) / L.CK;
            const double mfma = 64.0 * kTiles[i].wm * kTiles[i].wn * (double)nchunks * (L.CK / 2) * p.KH * p.KW * kEff[ci];
            const double ovh = 1500.0 * nchunks + 400.0 * kTiles[i].wm * kTiles[i].wn;     // barriers + commit, epilogue
            const double cost = (double)rounds * (mfma + ovh / per_cu);                     // co-resident blocks hide each other's overhead
            if (sel < 0 || cost < best_cost) { sel = i; best_cost = cost; Lbest = L; vbest = vec; lbest = lds; gbest = grid; }
        }
        if (sel < 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d: no tile configuration fits this problem");
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    switch (sel) {
        case 0: e = launch_cfg<4, 2>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        case 1: e = launch_cfg<2, 2>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        case 2: e = launch_cfg<1, 4>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        case 3: e = launch_cfg<2, 1>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        case 4: e = launch_cfg<1, 1>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        case 5: e = launch_cfg<1, 2>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        case 6: e = launch_cfg<4, 1>(p, Lbest, (int)gbest, lbest, vbest, st); break;
        default: e = launch_cfg<2, 4>(p, Lbest, (int)gbest, lbest, vbest, st); break;
    }
    if (e != hipSuccess) return l2i_set_error(L2I_E_LAUNCH, hipGetErrorString(e));
    if (Lbest.ksplit > 1) {
        l2i_conv_params q = p;
        q.ksplit = Lbest.ksplit;
        return l2i_launch_splitk_epilogue(q, st);
    }
    return L2I_OK;
}
