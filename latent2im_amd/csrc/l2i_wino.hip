// l2i_wino.hip — 3x3 stride-1 correlation as Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950).
//
// The 3x3 stride-1 layers (StyleGAN2 styled convs, VGG-19, the ResNet-50 bottleneck 3x3s, discriminator convs and all of
// their input-gradients) are half of the step's matrix time, and the direct kernel of l2i_conv.hip already runs them at
// ~84 % of the clock-limited v_mfma_f32_32x32x2_f32 rate.  The remaining lever that keeps fp32 arithmetic is fewer
// multiplies: Y = A^T [ (G g G^T) . (B^T d B) ] A produces a 2x2 output tile from a 4x4 input patch with 16 multiplies
// per (cin, cout) instead of 36 — 2.25x less matrix work (the transform the vendor libraries pick for fp32 3x3 too).
// The 16 transform positions are 16 independent GEMMs  M[pos] = U[pos] (Cout x Cin) * V[pos] (Cin x tiles), and the inverse
// transform needs all 16 of a (channel, tile) in one lane.
//
// First design (history, measured, removed): 32x32x2 MFMAs with all 16 position accumulators of a 32x32 block per wave =
// 256 accumulator registers, ONE wave per SIMD, persistent blocks.  tools/probes/wino_loop_probe.hip shows why it stalls at
// ~50 % matrix-pipe busy: with one wave per SIMD, LDS reads and global loads issue for free beside the MFMAs (152 TFLOP/s
// with 8 ds_read_b128 per 16 MFMAs) but the wave's own VALU does not (48 VALU of input transform per 16 MFMAs: 105
// TFLOP/s wherever placed), and nothing overlaps a block's transform phase, barriers, tile setup or store-issue-bound
// epilogue.  This design:
//
// A wave owns 32 output channels x 16 tiles on v_mfma_f32_16x16x4_f32 (same 64 FLOP/clk/SIMD): 16 positions x 2 channel
// blocks x 4 registers = 128 accumulators, the kernel fits 256 registers and TWO blocks share a CU — while one block
// transforms, synchronises or stores, the other block's wave on the same SIMD keeps the matrix pipe busy (what the direct
// kernel gets from 3 blocks per CU).  So the input transform is back in the MFMA stream, done by the lane that consumes
// it: lane (k = lane/16, n = lane%16) reads the 4x4 raw patch of (channel 4s+k, tile n) from the LDS halo tile, forms
// V = B^T d B (48 VALU) and feeds 16 positions x 2 channel blocks = 32 MFMAs (1024 matrix-pipe cycles).
//
// Block = 256 threads = 4 waves = 32 channels x 64 tiles (32 x 8 pixels, one tile row per wave); K chunks of 8 channels:
// raw halo tile global -> registers -> LDS (gradient mask fused), U (pre-transformed weights, [Cin][4][CoutP][4]) global
// -> LDS by DMA; both double buffered: one barrier per chunk, the next chunk's loads are in flight during the MFMAs.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed fp32 VALU (two independent lanes of work per issue slot).  Inline asm because hipcc scalarises most f32x2
// arithmetic (and cannot see hazards inside asm: see pk_mul_op).  [r3] A/B against the same arithmetic as two single-lane
// v_add / v_sub / v_mul per packed instruction (one asm statement each, so that the SLP vectoriser cannot re-pack them), interleaved in one
// process on fifteen launch shapes of the step: the single-lane build is 1.2 - 6.4 % SLOWER on every shape.  Packed stays.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// pk_mul_op: the product feeds an MFMA next, the 2 wait states of "VALU write -> MFMA read" ride in the same asm statement
__device__ __forceinline__ f32x2 pk_mul_op(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (a.lo + b.hi, a.lo - b.hi).  The half selection sits on SRC0 (b first: the sum commutes, the bits are the same): packed fp32 with op_sel set on
// src1 (the first form of this helper, and what hipcc's SLP vectorizer emits) returns sporadically wrong results while a bf16-MFMA kernel is resident
// on the same CUs; op_sel on src0, op_sel_hi and plain operands do not (tools/probes/pk_beside_conv_h8.py).  No configuration
// of the step runs this kernel beside a bf16-MFMA one; tools/probes/wino_beside_conv_h8.py does: 170-197 distinct results in 200 before, 1 now.
__device__ __forceinline__ f32x2 pk_lo_pm_hi(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a)); return r;
}
// the same two with the "VALU write -> MFMA read" wait states attached (layers without a style scale feed them to the MFMAs directly)
__device__ __forceinline__ f32x2 pk_sub_op(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ f32x2 pk_lo_pm_hi_op(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\ts_nop 1" : "=v"(r) : "v"(b), "v"(a)); return r;
}

struct WinoLaunch {
    int tiles_x, tiles_y, mblocks;
    int total;                         // work items = B * tiles_y * tiles_x * mblocks (grid is padded to a multiple of 8)
    int nchunks;                       // Cin / 8
    int relu_in;                       // the gradient mask IS the input with a ReLU mask (in_mask == x, mask (1, 0): VGG-19's convs read the pre-ReLU tap of
                                       // the layer below): pro(x) = max(x, 0) — the unmasked instantiation with one v_max at commit, no second tile stream
};

namespace wg {
constexpr int BM = 32, CK = 8;               // block = 32 channels x 64 tiles (32 x 8 pixels)
constexpr int IH = 10, IW = 34, PLANE = IH * IW, NRAW = CK * PLANE, NIN = (NRAW + 255) / 256;
constexpr int NU4 = CK * 4 * BM;                       // float4 of U per chunk: [c][i][ch] x (4 j) = 16 KiB
constexpr int NWV = NU4 / 256;
constexpr int RAWBUF = NIN * 256;                      // a DMA slot writes all 256 lanes (out-of-tile lanes: zeros behind the tile)
constexpr int TAB = 96;                                // [2 x 8 style scales of the masked path, pad to 32][32 demod][32 bias]
constexpr int NRS = 3;                                 // raw stages: the unmasked (DMA) path fetches the raw tile TWO chunks ahead
constexpr int LDS_FLOATS = NRS * RAWBUF + 2 * NU4 * 4 + TAB;   // one barrier per chunk; the 32 KiB of epilogue transpose strips alias the two
                                                               // U stages; SCALE && !MASK: + Cin style scales
}

// SCALE: the launch has a style scale (in_scale: the generator's modulated convs).  Without one (VGG-19, ResNet-50, discriminator and every
// gradient conv that is not the generator's) the per-channel multiply leaves the MFMA stream: a VALU instruction costs its issue cycles of
// matrix time (ablation builds: the kernel's time is the MFMA time PLUS its other instructions' issue time, almost without overlap —
// 0.63 ms = 0.44 (MFMA-bound) + 0.26 (everything else, MFMAs replaced by one FMA each) at 512->512 @64^2), and 16 of the 48 packed VALU
// instructions of a chunk were that multiply
// [r3] MASK = false (87 % of the family's time): the raw halo tile goes global -> LDS by DMA too (buffer_load_dword ... lds: one element per
// lane, element e = slot * 256 + tid of the [CK][IH][IW] tile exactly as before, out-of-image elements through an out-of-range offset =
// zeros) — no staging registers, no commit pass (11 LDS stores + 34 VALU per chunk and wave: the mask / ReLU selects ran even when the launch
// had neither), no wait on register loads at the top of a chunk; the style scales of the whole layer are read into LDS once per block.
// RELU (the gradient mask is the input itself, VGG-19's convs on pre-ReLU taps): max(x, 0) on the fragment reads, only in that instantiation.
// MASK = true keeps the register path (the mask tile would otherwise be a second DMA stream multiplied into every fragment).
template <bool MASK, bool SCALE, bool RELU>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const l2i_conv_params p, const WinoLaunch L) {
    using namespace wg;
    static_assert(!(MASK && RELU), "relu_in launches take the unmasked path");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rawbuf = smem;                              // NRS x [CK][IH][IW] (the masked path uses two of them)
    float* ubuf = smem + NRS * RAWBUF;                 // 2 x [CK][4][BM] float4
    float* tab = smem + NRS * RAWBUF + 2 * NU4 * 4;
    float* stab = tab + TAB;                           // SCALE && !MASK: [Cin] style scales of this sample

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, n = lane & 15;           // MFMA K index (channel within a group of 4) / tile column

    // blocks are dealt round-robin to the 8 XCDs: renumber so that the channel blocks of one pixel tile (and its
    // x-neighbours) run on the same XCD and share the input tile in its L2
    const int G = gridDim.x;
    int w = (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    if (w >= L.total) return;
    const int mblk = w % L.mblocks; w /= L.mblocks;
    const int tx = w % L.tiles_x; w /= L.tiles_x;
    const int ty = w % L.tiles_y; w /= L.tiles_y;
    const int b = w, m0 = mblk * BM, oy0 = ty * 8, ox0 = tx * 32;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;

    // ---- staging ----
    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * sizeof(float));
    const unsigned in_bytes = (unsigned)p.Cin * plane_b;
    const size_t smp = (size_t)b * p.Cin * ((size_t)p.H * p.W);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + smp), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + smp), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((size_t)p.Cin * 16 * p.CoutP * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)((p.in_scale ? p.in_scale : p.x) + (size_t)b * p.Cin), 0,
                                                                            p.in_scale ? (unsigned)(p.Cin * sizeof(float)) : 0u, 0x00020000);
    unsigned voff[NIN];
    {   // element e = tid + 256 u of the [CK][IH][IW] tile -> (c, iy, ix), walked incrementally (256 = 7 * 34 + 18: one division for u = 0 instead of
        // two per slot: every VALU instruction of this kernel is matrix time lost, and at Cin = 32 .. 64 a block has only 4 .. 8 chunks to amortise it)
        static_assert(IW == 34 && IH == 10, "the increments below are for a 10 x 34 tile");
        int c = 0, iy = tid / IW, ix = tid - iy * IW;
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const int gy = iy0 + iy, gx = ix0 + ix;
            const bool ok = (tid + u * 256 < NRAW) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
            voff[u] = ok ? (unsigned)c * plane_b + (unsigned)(gy * p.W + gx) * 4u : in_bytes;
            ix += 18; iy += 7;
            if (ix >= IW) { ix -= IW; ++iy; }
            if (iy >= IH) { iy -= IH; ++c; }
        }
    }
    const unsigned wvoff = (unsigned)(((tid / BM) * p.CoutP + m0 + (tid % BM)) * 16);
    const unsigned wstep = (unsigned)((256 / BM) * p.CoutP * 16);
    const unsigned svoff = (tid < CK) ? (unsigned)(tid * sizeof(float)) : 0xFFFFFFF0u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned rin[NIN];
    unsigned rmk[MASK ? NIN : 1];
    unsigned rsc = 0;

    // Staging of the next chunk, one slot at a time: slots 0..NWV-1 = U pieces, global -> LDS without registers (buffer_load_dwordx4
    // ... lds: lane l of the wave lands at M0 base + 16 l; inline asm on purpose: through the builtin, hipcc cannot tell the DMA's
    // LDS target from the stage being read and drains vmcnt before the next ds_read; the explicit vmcnt(0) before the publishing
    // barrier is in the chunk loop), slots NWV.. = raw halo elements (+ the style scale) into registers.  `on` = false (no next
    // chunk) swaps in null descriptors: no traffic, zeros.  compute() issues two slots per MFMA group, between the MFMAs.
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);
    constexpr int NSLOT = NWV + NIN + (MASK ? 1 : 0);
    // [r3] Unmasked path: the raw tile is fetched TWO chunks ahead into a ring of three LDS stages (c0r / rstage), U one chunk ahead (c0 /
    // ustage): a chunk is ~1.2 us of matrix work, and on the high-resolution layers (2 GB inputs streamed from HBM while the chip moves
    // 2 TB/s) a one-chunk distance left the top-of-chunk wait exposed: 64 -> 64 @1024^2 ran at 185 TFLOP/s against 230 on the L2-resident
    // 512-channel layers.  LDS-DMA needs no registers, so the distance costs one 11 KiB stage.  U slots are issued first, raw slots last:
    // the top-of-chunk wait is vmcnt(NIN) = everything but the youngest raw tile.
    auto issue_slot = [&](int sl, int c0, float* ustage, int c0r, float* rstage, bool on) {
        if (sl < NWV) {
            const unsigned sw = (unsigned)((size_t)c0 * 4 * p.CoutP * 16);
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ustage + wave_u * 256);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wvoff), "s"(on ? rs_w : rs_null), "s"(__builtin_amdgcn_readfirstlane(lds0 + sl * 4096)), "s"(sw + sl * wstep));
        } else if (sl < NWV + NIN) {
            const int u = sl - NWV;
            const unsigned so = (unsigned)(MASK ? c0 : c0r) * plane_b;
            if constexpr (MASK) {
                rin[u] = __builtin_amdgcn_raw_buffer_load_b32(on ? rs_x : rs_null, voff[u], so, 0);
                rmk[u] = __builtin_amdgcn_raw_buffer_load_b32(on ? rs_m : rs_null, voff[u], so, 0);
            } else {                                       // element e = u * 256 + tid lands at rstage[e]: lane l of the wave at M0 base + 4 l
                const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(rstage + wave_u * 64);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff[u]), "s"(on ? rs_x : rs_null), "s"(__builtin_amdgcn_readfirstlane(lds0 + u * 1024)), "s"(so));
            }
        } else if (MASK && sl == NWV + NIN) {
            rsc = __builtin_amdgcn_raw_buffer_load_b32(on ? rs_s : rs_null, svoff, (unsigned)(c0 * sizeof(float)), 0);
        }
    };
    auto issue = [&](int c0, float* ustage, float* rstage) {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) issue_slot(sl, c0, ustage, c0, rstage, true);
    };
    const bool relu_mask = p.mask_pos == 1.f && p.mask_neg == 0.f;      // kernel arguments: wave-uniform
    auto commit = [&](int par) {                                         // masked launches only: registers -> LDS with the gradient mask applied
        if constexpr (MASK) {
            float* raw = rawbuf + par * RAWBUF;
#pragma unroll
            for (int u = 0; u < NIN; ++u) {
                float v = __uint_as_float(rin[u]);
                if (relu_mask) v = (__uint_as_float(rmk[u]) > 0.f) ? v : 0.f;      // ReLU masks (ResNet-50 gradients): select, no multiply
                else v *= (__uint_as_float(rmk[u]) > 0.f) ? p.mask_pos : p.mask_neg;
                raw[tid + u * 256] = v;
            }
            if (tid < CK) tab[par * CK + tid] = p.in_scale ? __uint_as_float(rsc) : 1.f;
        }
    };
    if constexpr (SCALE && !MASK) {                                      // the layer's style scales of this sample, once per block
        for (int i = tid; i < p.Cin; i += 256) stab[i] = p.in_scale[(size_t)b * p.Cin + i];
    }

    f32x4 acc[16][2];                                  // first defined by the MFMAs of the peeled first chunk (C = 0 constant)

    // ---- one chunk: 2 K-steps of 4 channels; lane (kq, n) transforms the patch of (channel 4 s + kq, tile (wave, n)) ----
    auto compute = [&](int par, int rpar, auto first_tag, int c0, int c0n, float* un, int c0r, float* rn, bool on) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const float4* u4 = reinterpret_cast<const float4*>(ubuf + par * NU4 * 4) + n;
        const float* rp0 = rawbuf + rpar * RAWBUF + kq * PLANE + (2 * wave) * IW + 2 * n;
        // fragments are fetched one step ahead of the MFMAs that use them (raw patch: one K-step ahead; U rows: one position
        // row ahead), so a wave does not depend on its SIMD partner to cover its own LDS latency
        float2 dn[4][2];
        auto fetch_d = [&](int s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dn[r][0] = *reinterpret_cast<const float2*>(rp0 + 4 * s * PLANE + r * IW);
                dn[r][1] = *reinterpret_cast<const float2*>(rp0 + 4 * s * PLANE + r * IW + 2);
            }
        };
        auto relu_d = [&]() {                                  // RELU: pro(x) = max(x, 0) (one plain v_max each: no canonicalising second max)
            if constexpr (RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        asm("v_max_f32 %0, 0, %0" : "+v"(dn[r][q].x));
                        asm("v_max_f32 %0, 0, %0" : "+v"(dn[r][q].y));
                    }
            }
        };
        float4 a0n, a1n;
        auto fetch_a = [&](int s, int i) {
            a0n = u4[((4 * s + kq) * 4 + i) * BM];             // U[c][i][channel n][j = 0..3]
            a1n = u4[((4 * s + kq) * 4 + i) * BM + 16];
        };
        fetch_d(0);
        fetch_a(0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // Measured (tools/probes/mfma_valu_overlap_probe.hip): a VALU instruction costs ~4 cycles of fp32 matrix time at any
            // occupancy, so the transform is written on register pairs (v_pk_add_f32 / v_pk_mul_f32: two lanes of work per issue
            // slot).  Pairs are the column pairs (0,1) and (2,3) of the patch, exactly what ds_read2_b64 delivers.
            f32x2 t01[4], t23[4];                                  // B^T d, rows i = 0..3
            relu_d();
            {
                const f32x2 d0a = {dn[0][0].x, dn[0][0].y}, d0b = {dn[0][1].x, dn[0][1].y};
                const f32x2 d1a = {dn[1][0].x, dn[1][0].y}, d1b = {dn[1][1].x, dn[1][1].y};
                const f32x2 d2a = {dn[2][0].x, dn[2][0].y}, d2b = {dn[2][1].x, dn[2][1].y};
                const f32x2 d3a = {dn[3][0].x, dn[3][0].y}, d3b = {dn[3][1].x, dn[3][1].y};
                t01[0] = pk_sub(d0a, d2a); t23[0] = pk_sub(d0b, d2b);
                t01[1] = pk_add(d1a, d2a); t23[1] = pk_add(d1b, d2b);
                t01[2] = pk_sub(d2a, d1a); t23[2] = pk_sub(d2b, d1b);
                t01[3] = pk_sub(d1a, d3a); t23[3] = pk_sub(d1b, d3b);
            }
            f32x2 scp = {1.f, 1.f};
            if constexpr (SCALE) { const float sc = MASK ? tab[par * CK + 4 * s + kq] : stab[c0 + 4 * s + kq]; scp = f32x2{sc, sc}; }
            if (s == 0) fetch_d(1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a0 = a0n, a1 = a1n;
                if (i < 3) fetch_a(s, i + 1);
                else if (s == 0) fetch_a(1, 0);
                issue_slot(2 * (4 * s + i), c0n, un, c0r, rn, on);   // later chunks: two staging slots per MFMA group, issued beside the MFMAs
                issue_slot(2 * (4 * s + i) + 1, c0n, un, c0r, rn, on);
                __builtin_amdgcn_sched_barrier(0);
                // (B^T d) B: (v0, v3) = (t0 - t2, t1 - t3);  (v1, v2) = (t2 + t1, t2 - t1); then the style scale of the channel
                const f32x2 v03 = SCALE ? pk_mul_op(pk_sub(t01[i], t23[i]), scp) : pk_sub_op(t01[i], t23[i]);
                const f32x2 v12 = SCALE ? pk_mul_op(pk_lo_pm_hi(t23[i], t01[i]), scp) : pk_lo_pm_hi_op(t23[i], t01[i]);
                acc[i * 4 + 0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, v03.x, (FIRST && s == 0) ? zero : acc[i * 4 + 0][0], 0, 0, 0);
                acc[i * 4 + 0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, v03.x, (FIRST && s == 0) ? zero : acc[i * 4 + 0][1], 0, 0, 0);
                acc[i * 4 + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, v12.x, (FIRST && s == 0) ? zero : acc[i * 4 + 1][0], 0, 0, 0);
                acc[i * 4 + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, v12.x, (FIRST && s == 0) ? zero : acc[i * 4 + 1][1], 0, 0, 0);
                acc[i * 4 + 2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, v12.y, (FIRST && s == 0) ? zero : acc[i * 4 + 2][0], 0, 0, 0);
                acc[i * 4 + 2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, v12.y, (FIRST && s == 0) ? zero : acc[i * 4 + 2][1], 0, 0, 0);
                acc[i * 4 + 3][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, v03.y, (FIRST && s == 0) ? zero : acc[i * 4 + 3][0], 0, 0, 0);
                acc[i * 4 + 3][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, v03.y, (FIRST && s == 0) ? zero : acc[i * 4 + 3][1], 0, 0, 0);
            }
        }
    };

    static_assert(NSLOT <= 16, "two staging slots per MFMA group, 8 groups per chunk");
    if constexpr (MASK) {
        issue(0, ubuf, rawbuf);
        commit(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the chunk's U DMA has landed
        __syncthreads();
        compute(0, 0, std::true_type(), 0, 1 < L.nchunks ? CK : 0, ubuf + NU4 * 4, 0, rawbuf, true);
        for (int ch = 1; ch < L.nchunks; ++ch) {
            const int par = ch & 1;
            commit(par);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's U DMA (issued one chunk ago) has landed
            __syncthreads();                                   // raw tile of this chunk written; every wave is past the MFMAs of the previous chunk,
                                                               // so the other raw / U stage may be refilled
            // (after the last chunk the staging slots re-fetch chunk 0 — valid addresses, L2 hits, never read — instead of switching every
            //  slot's descriptor to a null one: 32 scalar selects per chunk less in the MFMA stream)
            compute(par, par, std::false_type(), ch * CK, ch + 1 < L.nchunks ? (ch + 1) * CK : 0, ubuf + (par ^ 1) * NU4 * 4, 0, rawbuf, true);
        }
    } else {
        // U(0), raw(0), then raw(1): the wait leaves the youngest raw tile in flight
        issue(0, ubuf, rawbuf);
#pragma unroll
        for (int sl = NWV; sl < NWV + NIN; ++sl) issue_slot(sl, 0, ubuf, 1 < L.nchunks ? CK : 0, rawbuf + RAWBUF, true);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIN) : "memory");
        __syncthreads();
        compute(0, 0, std::true_type(), 0, 1 < L.nchunks ? CK : 0, ubuf + NU4 * 4, 2 < L.nchunks ? 2 * CK : 0, rawbuf + 2 * RAWBUF, true);
        int rpar = 1;
        for (int ch = 1; ch < L.nchunks; ++ch) {
            const int par = ch & 1;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIN) : "memory");   // U(ch) and raw(ch) have landed; raw(ch + 1) stays in flight
            __syncthreads();                                   // ... for every wave, and every wave is past the MFMAs of chunk ch - 1: its stages may be refilled
            const int rnext = rpar == 0 ? 2 : rpar - 1;        // (ch + 2) % 3 == (ch - 1) % 3
            compute(par, rpar, std::false_type(), ch * CK, ch + 1 < L.nchunks ? (ch + 1) * CK : 0, ubuf + (par ^ 1) * NU4 * 4,
                    ch + 2 < L.nchunks ? (ch + 2) * CK : 0, rawbuf + rnext * RAWBUF, true);
            rpar = rpar == 2 ? 0 : rpar + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (null-descriptor DMA of the last chunk)
    __syncthreads();                                       // the U stages become the transpose strips

    // ---- epilogue: lane-local inverse transform -> per-wave LDS transpose (aliases the U stages) -> 16-byte accesses ----
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    if (tid < BM) {
        const int co = m0 + tid;
        tab[32 + tid] = (p.out_scale && co < p.Cout) ? p.out_scale[(size_t)b * p.Cout + co] : 1.f;
        tab[64 + tid] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }
    float* strip = ubuf + wave * 2048;                 // [32 channels][2 rows][32 px]
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {                           // accumulator rows (r, r+1) = channels (chn, chn+1) as one register pair
            f32x2 a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = hp ? f32x2{acc[q][blk][2], acc[q][blk][3]} : f32x2{acc[q][blk][0], acc[q][blk][1]};
            f32x2 t0[4], t1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                          // A^T M
                t0[c] = pk_add(pk_add(a[c], a[4 + c]), a[8 + c]);
                t1[c] = pk_sub(pk_sub(a[4 + c], a[8 + c]), a[12 + c]);
            }
            const f32x2 y00 = pk_add(pk_add(t0[0], t0[1]), t0[2]), y01 = pk_sub(pk_sub(t0[1], t0[2]), t0[3]);
            const f32x2 y10 = pk_add(pk_add(t1[0], t1[1]), t1[2]), y11 = pk_sub(pk_sub(t1[1], t1[2]), t1[3]);
            const int chn = blk * 16 + 4 * kq + 2 * hp;
            float* s0 = &strip[chn * 64 + 2 * n];
            s0[0] = y00.x; s0[1] = y01.x; s0[32] = y10.x; s0[33] = y11.x;
            s0[64] = y00.y; s0[65] = y01.y; s0[96] = y10.y; s0[97] = y11.y;
        }
    }
    __syncthreads();                                   // epilogue tables visible (the strips themselves are per wave)
    const int q16 = lane & 15, chl = lane >> 4;
    const int row = q16 >> 3, col = (q16 & 7) * 4;
    const int oy = oy0 + 2 * wave + row, ox = ox0 + col;
    const bool pok = (oy < p.OH) && (ox < p.OW);
    const size_t poff = (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pok && p.noise) {
        nz = *reinterpret_cast<const float4*>(p.noise + (size_t)b * plane_o + poff);
        nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
    }
    float sq = 0.f;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int chn = 4 * it + chl;
        const int co = m0 + chn;
        float4 v = *reinterpret_cast<const float4*>(&strip[chn * 64 + row * 32 + col]);
        if (pok && co < p.Cout) {
            const float sc = tab[32 + chn], bv = tab[64 + chn];
            f32x2 va = {v.x, v.y}, vb = {v.z, v.w};
            if (p.out_scale) { const f32x2 scp = {sc, sc}; va = pk_mul(va, scp); vb = pk_mul(vb, scp); }
            const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + poff;
            if (p.out_mask) {
                const float4 mk = *reinterpret_cast<const float4*>(p.out_mask + oidx);
                va.x = mk.x > 0.f ? va.x : 0.f; va.y = mk.y > 0.f ? va.y : 0.f; vb.x = mk.z > 0.f ? vb.x : 0.f; vb.y = mk.w > 0.f ? vb.y : 0.f;
            }
            const f32x2 bvp = {bv, bv};
            va = pk_add(va, pk_add(f32x2{nz.x, nz.y}, bvp)); vb = pk_add(vb, pk_add(f32x2{nz.z, nz.w}, bvp));
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
                va = pk_add(va, f32x2{rv.x, rv.y}); vb = pk_add(vb, f32x2{rv.z, rv.w});
            }
            if (p.act == L2I_ACT_LRELU) {                          // max(v, slope v) == (v > 0 ? v : slope v) for 0 <= slope <= 1
                const f32x2 slp = {p.act_slope, p.act_slope}, gnp = {p.act_gain, p.act_gain};
                const f32x2 ta = pk_mul(va, slp), tb = pk_mul(vb, slp);
                va = pk_mul(f32x2{__builtin_fmaxf(va.x, ta.x), __builtin_fmaxf(va.y, ta.y)}, gnp);
                vb = pk_mul(f32x2{__builtin_fmaxf(vb.x, tb.x), __builtin_fmaxf(vb.y, tb.y)}, gnp);
            } else if (p.act == L2I_ACT_RELU) {
                va.x = __builtin_fmaxf(va.x, 0.f); va.y = __builtin_fmaxf(va.y, 0.f); vb.x = __builtin_fmaxf(vb.x, 0.f); vb.y = __builtin_fmaxf(vb.y, 0.f);
            }
            if (p.out_gain != 1.f) { const f32x2 ogp = {p.out_gain, p.out_gain}; va = pk_mul(va, ogp); vb = pk_mul(vb, ogp); }
            v = make_float4(va.x, va.y, vb.x, vb.y);
            if (p.accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(p.y + oidx) = v;
            if (p.sq_ref) {                                        // ContentLoss value of a VGG tap: sum (y - reference)^2 while y is in registers
                const float4 rf = *reinterpret_cast<const float4*>(p.sq_ref + oidx);
                const float d0 = v.x - rf.x, d1 = v.y - rf.y, d2 = v.z - rf.z, d3 = v.w - rf.w;
                sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
    }
    if (p.sq_ref) {                                                // (kernel argument: uniform branch) one atomic per block, 1024 slots
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        __syncthreads();                                           // every lane is past its reads of the epilogue tables
        if (lane == 0) tab[wave] = sq;
        __syncthreads();
        if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (tab[0] + tab[1]) + (tab[2] + tab[3]));
    }
}

static int launch_wino(const l2i_conv_params& p, hipStream_t st) {
    WinoLaunch L;
    L.tiles_x = (p.OW + 31) / 32;
    L.tiles_y = (p.OH + 7) / 8;
    L.mblocks = p.CoutP / wg::BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d_wino: too many tiles");
    L.total = (int)total;
    L.nchunks = p.Cin / wg::CK;
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    L.relu_in = (p.in_mask == p.x && p.mask_pos == 1.f && p.mask_neg == 0.f) ? 1 : 0;
    const bool mask = p.in_mask != nullptr && !L.relu_in;
    const bool scale = p.in_scale != nullptr;
    const size_t lds = (size_t)(wg::LDS_FLOATS + ((scale && !mask) ? ((p.Cin + 3) & ~3) : 0)) * sizeof(float);
    if (lds > 80 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: too many input channels for the style-scale table");
    // > 64 KiB of dynamic LDS (three raw stages): the limit is raised once per instantiation
#define L2I_WINO(M_, S_, R_)                                                                                                            \
    do {                                                                                                                                \
        L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<M_, S_, R_>),                      \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));                         \
        hipLaunchKernelGGL((conv_wino_kernel<M_, S_, R_>), dim3(grid), dim3(256), lds, st, p, L);                                        \
    } while (0)
    if (mask) { if (scale) L2I_WINO(true, true, false); else L2I_WINO(true, false, false); }
    else if (L.relu_in) { if (scale) L2I_WINO(false, true, true); else L2I_WINO(false, false, true); }
    else { if (scale) L2I_WINO(false, true, false); else L2I_WINO(false, false, false); }
#undef L2I_WINO
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_conv2d_wino_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d_wino: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d_wino: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, false, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d_wino: non-positive dimension");
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.oy_step != 1 || p.ox_step != 1 || (p.Cin % 8) != 0)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: needs a 3x3 stride-1 dense-output layer with Cin % 8 == 0");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_wino: CoutP must be Cout rounded up to 32");
    if (p.oy_off < 0 || p.ox_off < 0 || p.OH + p.oy_off > p.OHf || p.OW + p.ox_off > p.OWf)
        return l2i_set_error(L2I_E_ARG, "conv2d_wino: output window exceeds the output tensor");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if ((p.OWf % 4) != 0 || (p.OW % 4) != 0 || (p.ox_off % 4) != 0 || !al16(p.y) || !al16(p.residual) || !al16(p.res_mask) || !al16(p.res_sub) || !al16(p.out_mask) ||
        !al16(p.noise) || !al16(p.w))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: output rows must be 16-byte aligned multiples of 4 pixels");
    if ((size_t)p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull || (size_t)p.Cin * 16 * p.CoutP * sizeof(float) >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: one sample / the weight pack must stay below 4 GiB (32-bit buffer offsets)");
    if (p.tile_hint != 0) return l2i_set_error(L2I_E_ARG, "conv2d_wino: tile_hint must be 0 (one configuration)");
    if ((p.sq_ref != nullptr) != (p.sq_out != nullptr) || (((uintptr_t)p.sq_ref) % 16) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_wino: sq_ref (16-byte aligned) and sq_out go together");
    return launch_wino(p, (hipStream_t)stream);
}
