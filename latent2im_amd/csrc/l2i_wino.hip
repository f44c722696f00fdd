// l2i_wino.hip — 3x3 stride-1 correlation as Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950).
//
// The 3x3 stride-1 layers (StyleGAN2 styled convs, VGG-19, the ResNet-50 bottleneck 3x3s, discriminator convs and all of
// their input-gradients) are half of the step's matrix time, and the direct kernel of l2i_conv.hip already runs them at
// ~84 % of the clock-limited v_mfma_f32_32x32x2_f32 rate.  The remaining lever that keeps fp32 arithmetic is fewer
// multiplies: Y = A^T [ (G g G^T) . (B^T d B) ] A produces a 2x2 output tile from a 4x4 input patch with 16 multiplies
// per (cin, cout) instead of 36 — 2.25x less matrix work (the transform the vendor libraries pick for fp32 3x3 too).
//
// Mapping (wave64): the 16 transform positions are 16 independent GEMMs  M[pos] = U[pos] (Cout x Cin) * V[pos] (Cin x tiles).
//   MFMA-M = 32 output channels (A operand = pre-transformed weights U, packed on the host),
//   MFMA-N = 32 tiles (a 16 x 2 group of 2x2 output tiles = 32 x 4 pixels),   MFMA-K = 2 input channels.
// A wave keeps all 16 position accumulators of its 32 x 32 block (256 accumulator registers -> one wave per SIMD, one
// 256-thread block per CU), so the inverse transform is purely lane-local.
//
// Measured on MI355X (tools/probes/wino_loop_probe.hip): with one wave per SIMD, LDS reads issue for free beside the MFMAs
// (152 TFLOP/s of MFMA work with 8 ds_read_b128 per 16 MFMAs) but the wave's own VALU instructions do not — the 48-VALU
// input transform inside the MFMA stream cost 30 % (105 TFLOP/s) wherever it was placed.  So the K loop is MFMA + LDS
// reads only: the input transform V = B^T d B runs once per block (not once per consuming wave) in the staging phase,
// raw halo tile -> LDS -> V in LDS, and the MFMA loop reads V and U as ds_read_b128 fragments.
// With one block per CU there is no co-resident block to hide prologue/epilogue latency, so the kernel is persistent:
// each block walks work items (pixel tile x channel block) and K chunks as ONE flat software pipeline — the global loads
// and the register -> LDS writes of the following chunks are issued between the MFMAs of the current one (possibly for the
// next tile's first chunk), and the V/U stages are double buffered.
//
// Prologue fusions (style scale on V, activation-gradient mask on the raw tile) and the whole epilogue (demod, noise,
// bias, residual, masks, activation, gains, accumulate) match l2i_conv.hip; outputs go through a per-wave LDS transpose so
// every global access is 16 bytes wide.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef WINO_TIMING
__device__ unsigned long long g_wino_t[16];
#define TSTAMP(k) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); g_wino_t[k] += now_ - tlast; tlast = now_; } } while (0)
extern "C" int l2i_debug_wino_timing(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_t), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_t), 16 * sizeof(unsigned long long));
}
#else
#define TSTAMP(k) do {} while (0)
#endif

struct WinoLaunch {
    int tiles_x, tiles_y, mblocks;
    int total;                         // work items = B * tiles_y * tiles_x * mblocks
    int nchunks;                       // Cin / CK
};

template <int WCH> struct WinoGeom {
    static constexpr int WT = 4 / WCH;                 // waves along the pixel tile (each 32 x 4 pixels = 16 x 2 tiles)
    static constexpr int BM = 32 * WCH;                // output channels per block
    static constexpr int NT = 32 * WT;                 // 2x2 tiles per block (16 wide)
    static constexpr int CK = (WCH == 1) ? 4 : 8;      // input channels per chunk
    static constexpr int CKh = CK / 2;
    static constexpr int IH = 4 * WT + 2, IW = 34;     // raw halo tile (pitch = IW)
    static constexpr int PLANE = IH * IW;
    static constexpr int NRAW = CK * PLANE;
    static constexpr int NIN = (NRAW + 255) / 256;
    static constexpr int NV4 = CK * 4 * NT;            // float4 of V per chunk: [c][i][tile] x (4 j)
    static constexpr int NU4 = CK * 4 * BM;            // float4 of U per chunk: [c][i][ch] x (4 j)
    static constexpr int NWV = NU4 / 256;
    static constexpr int NSLOT = NIN;                  // register staging slots per thread per chunk (raw tile); U goes global -> LDS directly
    static constexpr int NITEM = CK * NT / 256;        // (channel, tile) transform items per thread per chunk
    static constexpr int STAGE = NV4 * 4 + NU4 * 4 + 512;         // floats: V | U | demod[256] | bias[256]
    static constexpr int RAWBUF = NRAW + 64;           // + a dump row for the slots past the end of the tile
    static constexpr bool EPI_ALIAS = (NV4 + NU4) * 4 >= 4 * 4096;   // the 64 KiB transpose strips fit on top of a consumed stage
    static constexpr int LDS_FLOATS = 2 * STAGE + RAWBUF + (EPI_ALIAS ? 0 : 4 * 4096);
};

template <int WCH, bool MASK>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const l2i_conv_params p, const WinoLaunch L) {
    typedef WinoGeom<WCH> Gm;
    constexpr int WT = Gm::WT, BM = Gm::BM, NT = Gm::NT, CK = Gm::CK, CKh = Gm::CKh, IW = Gm::IW, PLANE = Gm::PLANE;
    constexpr int NRAW = Gm::NRAW, NIN = Gm::NIN, NV4 = Gm::NV4, NU4 = Gm::NU4, NWV = Gm::NWV, NSLOT = Gm::NSLOT, NITEM = Gm::NITEM, STAGE = Gm::STAGE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* raw = smem + 2 * STAGE;                     // [CK][IH][IW] raw halo tile of the chunk being transformed

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wch = wave % WCH, wt = wave / WCH;
    const int txl = j & 15, tyl = j >> 4;              // this lane's 2x2 tile inside the wave's 16 x 2 group

    auto decode = [&](int w, int& b, int& m0, int& oy0, int& ox0) {
        const int mblk = w % L.mblocks; w /= L.mblocks;
        const int tx = w % L.tiles_x; w /= L.tiles_x;
        const int ty = w % L.tiles_y; w /= L.tiles_y;
        b = w; m0 = mblk * BM; oy0 = ty * (4 * WT); ox0 = tx * 32;
    };

    // blocks are dealt round-robin to the 8 XCDs: renumber so that consecutive work items (channel blocks of one pixel
    // tile, then x-neighbours) run on the same XCD and share its L2
    const int G = gridDim.x;
    const int vb = ((G & 7) == 0) ? (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int nmine = (L.total - vb + G - 1) / G;
    const int iters = nmine * L.nchunks;

    // ---- staging state: one register set; slot s of chunk f+1 is written to LDS and then reloaded with chunk f+2
    //      *inside* the MFMA stream of chunk f (memory instructions issue for free beside MFMAs, see the header) ----
    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * sizeof(float));        // bytes per channel plane (sample < 4 GiB)
    const unsigned in_bytes = (unsigned)p.Cin * plane_b;
    const unsigned u_bytes = (unsigned)((size_t)p.Cin * 16 * p.CoutP * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, u_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.x), 0, p.bias ? (unsigned)(p.Cout * sizeof(float)) : 0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = rs_null, rs_m = rs_null, rs_s = rs_null, rs_o = rs_null;
    unsigned voff[NIN];
    unsigned wvoff = 0, ovoff = 0;                         // wvoff: U cursor (one chunk behind the raw cursor)
    const unsigned wstep = (unsigned)((256 / BM) * p.CoutP * 16);
    unsigned rin[NIN];
    unsigned rmk[MASK ? NIN : 1];
    unsigned rsc[NITEM], ros = 0, rbi = 0;
    float sc_cur[NITEM];                                   // style scales of the chunk whose raw tile is in LDS

    auto setup = [&](int w) {
        asm volatile(".p2align 8");        // fixed placement of each phase: this kernel loses up to 30 % at unlucky code offsets (measured)
        int b, m0, oy0, ox0;
        decode(w, b, m0, oy0, ox0);
        const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
        const size_t smp = (size_t)b * p.Cin * ((size_t)p.H * p.W);
        rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + smp), 0, in_bytes, 0x00020000);
        rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)((MASK ? p.in_mask : p.x) + smp), 0, in_bytes, 0x00020000);
        rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)((p.in_scale ? p.in_scale : p.x) + (size_t)b * p.Cin), 0,
                                                 p.in_scale ? (unsigned)(p.Cin * sizeof(float)) : 0u, 0x00020000);
        rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)((p.out_scale ? p.out_scale : p.x) + (size_t)b * p.Cout), 0,
                                                 p.out_scale ? (unsigned)(p.Cout * sizeof(float)) : 0u, 0x00020000);
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            int e = tid + u * 256;
            asm volatile("" : "+v"(e));                // recompute the slot geometry per tile (~20 VALU) instead of keeping 3 hoisted registers per slot alive
            const int c = e / PLANE, rem = e - c * PLANE;
            const int iy = rem / IW, ix = rem - iy * IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const bool ok = (e < NRAW) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
            voff[u] = ok ? (unsigned)c * plane_b + (unsigned)(gy * p.W + gx) * 4u : in_bytes;
        }
        ovoff = (tid < BM) ? (unsigned)((m0 + tid) * sizeof(float)) : 0xFFFFFFF0u;
    };
    auto drain = [&]() { rs_x = rs_null; rs_m = rs_null; rs_s = rs_null; rs_o = rs_null; rs_b = rs_null; };   // loads past the last chunk: no traffic

    // slot s: one element of the raw halo tile.  Everything here is branch-free so that the instruction scheduler may
    // place it between MFMAs.
    auto issue_slot = [&](int s, int c0) {
        const unsigned so = (unsigned)c0 * plane_b;
        rin[s] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff[s], so, 0);
        if constexpr (MASK) rmk[s] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, voff[s], so, 0);
    };
    auto commit_slot = [&](int s) {
        const int e = tid + s * 256;
        float v = __uint_as_float(rin[s]);
        if constexpr (MASK) v *= (__uint_as_float(rmk[s]) > 0.f) ? p.mask_pos : p.mask_neg;
        raw[(s * 256 + 255 < NRAW || e < NRAW) ? e : NRAW + lane] = v;           // [c][iy][ix], pitch IW: the slot index is the LDS offset
    };
    // U of one chunk: NWV pieces of 1 KiB per wave, global -> LDS without touching registers (buffer_load_dwordx4 ... lds:
    // lane l of the wave lands at piece base + 16 l, which is exactly the [c][i][ch] float4 order of the pack)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_u = [&](int k, float* stage, int c0) {
        const unsigned sw = (unsigned)((size_t)c0 * 4 * p.CoutP * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(stage + NV4 * 4 + (k * 256 + wave_u * 64) * 4), 16,
                                                 wvoff, sw + k * wstep, 0, 0);
    };
    auto issue_misc = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NITEM; ++u)                        // style scale of the channel of this thread's transform item u
            rsc[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_s, (unsigned)(((tid + u * 256) / NT) * sizeof(float)), (unsigned)(c0 * sizeof(float)), 0);
        ros = __builtin_amdgcn_raw_buffer_load_b32(rs_o, ovoff, 0, 0);
        rbi = __builtin_amdgcn_raw_buffer_load_b32(rs_b, ovoff, 0, 0);
    };
    auto commit_misc = [&](float* stage) {
        float* tab = stage + NV4 * 4 + NU4 * 4;                // demod[256] | bias[256]; entries >= BM are never read
        tab[tid] = p.out_scale ? __uint_as_float(ros) : 1.f;
        tab[256 + tid] = p.bias ? __uint_as_float(rbi) : 0.f;
#pragma unroll
        for (int u = 0; u < NITEM; ++u) sc_cur[u] = p.in_scale ? __uint_as_float(rsc[u]) : 1.f;
    };
    // input transform of the chunk, once per block: V[c][i][tile] = (B^T d B)[i][0..3] * style(c)
    auto transform = [&](float* stage) {
        asm volatile(".p2align 8");        // fixed placement of each phase: this kernel loses up to 30 % at unlucky code offsets (measured)
        float4* V4 = reinterpret_cast<float4*>(stage);
#pragma unroll
        for (int u = 0; u < NITEM; ++u) {
            const int id = tid + u * 256;
            const int c = id / NT, t = id % NT;
            const float* rp = raw + c * PLANE + (2 * (t >> 4)) * IW + 2 * (t & 15);
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2*>(rp + r * IW);
                const float2 hi = *reinterpret_cast<const float2*>(rp + r * IW + 2);
                d[r][0] = lo.x; d[r][1] = lo.y; d[r][2] = hi.x; d[r][3] = hi.y;
            }
            const float s = sc_cur[u];
            float tt[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                          // B^T d
                tt[0][q] = d[0][q] - d[2][q];
                tt[1][q] = d[1][q] + d[2][q];
                tt[2][q] = d[2][q] - d[1][q];
                tt[3][q] = d[1][q] - d[3][q];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                          // (B^T d) B, times the style scale of the channel
                float4 v;
                v.x = (tt[i][0] - tt[i][2]) * s;
                v.y = (tt[i][1] + tt[i][2]) * s;
                v.z = (tt[i][2] - tt[i][1]) * s;
                v.w = (tt[i][1] - tt[i][3]) * s;
                V4[(c * 4 + i) * NT + t] = v;
            }
        }
    };

    f32x16 acc[16];      // defined only by MFMAs (first chunk of a tile accumulates onto a zero constant): stays in the accumulator file

    // ---- MFMA over one chunk: lanes 0-31 take channel cc, lanes 32-63 channel cc + CK/2.  Between the MFMAs: the LDS
    //      fragment reads of the next k-pair, and this k-pair's share of the staging slots (LDS write of chunk f+1, then
    //      the global load of chunk f+2 into the same register) ----
    auto compute = [&](const float* stage, float* nstage, int c0_raw, int c0_u, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile(".p2align 8");        // fixed placement of each phase: this kernel loses up to 30 % at unlucky code offsets (measured)
        const float4* vb4 = reinterpret_cast<const float4*>(stage) + half * CKh * 4 * NT + wt * 32 + j;
        const float4* ub4 = reinterpret_cast<const float4*>(stage + NV4 * 4) + half * CKh * 4 * BM + wch * 32 + j;
        float4 bn[4], an[4];
        auto fetch = [&](int cc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { bn[i] = vb4[(cc * 4 + i) * NT]; an[i] = ub4[(cc * 4 + i) * BM]; }
        };
        fetch(0);
#pragma unroll
        for (int cc = 0; cc < CKh; ++cc) {
            float4 a4[4], b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a4[i] = an[i]; b4[i] = bn[i]; }
            __builtin_amdgcn_sched_barrier(0);
            if (cc + 1 < CKh) fetch(cc + 1);              // next k-pair's fragments are in flight during these 16 MFMAs
            if (cc == 0) {                                // chunk f+1: registers -> LDS (raw tile, epilogue tables); U: global -> LDS
                commit_misc(nstage);
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) commit_slot(s);
            }
            if (cc == (CKh > 2 ? 1 : 0)) {
#pragma unroll
                for (int k = 0; k < NWV; ++k) dma_u(k, nstage, c0_u);
            }
            if (cc == CKh - 1) {                          // chunk f+2: global -> registers
                issue_misc(c0_raw);
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) issue_slot(s, c0_raw);
            }
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool z = FIRST && cc == 0;
                acc[i * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[i].x, z ? zero : acc[i * 4 + 0], 0, 0, 0);
                acc[i * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[i].y, z ? zero : acc[i * 4 + 1], 0, 0, 0);
                acc[i * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[i].z, z ? zero : acc[i * 4 + 2], 0, 0, 0);
                acc[i * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[i].w, z ? zero : acc[i * 4 + 3], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);      // one LDS instruction
                __builtin_amdgcn_sched_group_barrier(0x010, MASK ? 2 : 1, 0);      // global loads
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- inverse transform (lane-local) -> per-wave LDS transpose -> fused epilogue with 16-byte global accesses ----
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    auto epilogue = [&](int w, float* stage) {
        asm volatile(".p2align 8");        // fixed placement of each phase: this kernel loses up to 30 % at unlucky code offsets (measured)
        int b, m0, oy0, ox0;
        decode(w, b, m0, oy0, ox0);
        const float* tab = stage + NV4 * 4 + NU4 * 4;
        float* strip;
        if constexpr (Gm::EPI_ALIAS) {
            __syncthreads();                                   // every wave is done with the fragments of this stage
            strip = stage + wave * 4096;
        } else {
            strip = smem + 2 * STAGE + Gm::RAWBUF + wave * 4096;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
            float t0[4], t1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                          // A^T M
                t0[c] = acc[c][r] + acc[4 + c][r] + acc[8 + c][r];
                t1[c] = acc[4 + c][r] - acc[8 + c][r] - acc[12 + c][r];
            }
            float2 y0, y1;
            y0.x = t0[0] + t0[1] + t0[2]; y0.y = t0[1] - t0[2] - t0[3];
            y1.x = t1[0] + t1[1] + t1[2]; y1.y = t1[1] - t1[2] - t1[3];
            *reinterpret_cast<float2*>(&strip[ch * 128 + (2 * tyl) * 32 + 2 * txl]) = y0;
            *reinterpret_cast<float2*>(&strip[ch * 128 + (2 * tyl + 1) * 32 + 2 * txl]) = y1;
            __builtin_amdgcn_sched_barrier(0);         // one accumulator row at a time: keeps the staging registers resident
        }
        const int q = lane & 31, chl = lane >> 5;
        const int row = q >> 3, col = (q & 7) * 4;
        const int oy = oy0 + 4 * wt + row, ox = ox0 + col;
        const bool pok = (oy < p.OH) && (ox < p.OW);
        const size_t poff = (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pok && p.noise) {
            nz = *reinterpret_cast<const float4*>(p.noise + (size_t)b * plane_o + poff);
            nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
        }
#pragma unroll 2
        for (int i = 0; i < 16; ++i) {
            const int ch = 2 * i + chl;
            const int co = m0 + wch * 32 + ch;
            float4 v = *reinterpret_cast<const float4*>(&strip[ch * 128 + row * 32 + col]);
            if (pok && co < p.Cout) {
                const float sc = tab[wch * 32 + ch];
                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + poff;
                if (p.out_mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(p.out_mask + oidx);
                    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                }
                const float bv = tab[256 + wch * 32 + ch];
                v.x += nz.x + bv; v.y += nz.y + bv; v.z += nz.z + bv; v.w += nz.w + bv;
                if (p.residual) {
                    float4 rv = *reinterpret_cast<const float4*>(p.residual + oidx);
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
                v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain;
                if (p.accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4*>(p.y + oidx) = v;
            }
        }
    };

    // ---- flat pipeline over f = (work item, chunk): while the MFMAs of f run, the raw tile of f+1 moves registers -> LDS,
    //      U of f+1 global -> LDS and the raw tile of f+2 global -> registers; between two MFMA phases only the input
    //      transform of f+1 is exposed ----
    int iw = vb, ic = 0;                                   // raw cursor (work item, chunk): next chunk to load into registers
    int iwu = vb, icu = 0;                                 // U cursor: next chunk to copy into LDS
    auto advance = [&]() { if (++ic == L.nchunks) { ic = 0; iw += G; } };
    const unsigned wv_base = (unsigned)(((tid / BM) * p.CoutP + (tid % BM)) * 16);
    auto setup_u = [&]() { wvoff = wv_base + (unsigned)((iwu % L.mblocks) * BM * 16); };
    auto advance_u = [&]() { if (++icu == L.nchunks) { icu = 0; iwu += G; } };
    setup(iw);
    setup_u();
#pragma unroll
    for (int k = 0; k < NWV; ++k) dma_u(k, smem, 0);
    advance_u();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) issue_slot(s, 0);
    issue_misc(0);
    advance();
    commit_misc(smem);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) commit_slot(s);
    if (1 < iters) {
        if (ic == 0) setup(iw);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) issue_slot(s, ic * CK);
        issue_misc(ic * CK);
        advance();
    }
    __syncthreads();
    transform(smem);
    __syncthreads();
    int cw = vb;                                           // compute cursor
    int it = 0;
#ifdef WINO_TIMING
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    auto pre = [&]() {                                     // staging state for the chunks issued during this MFMA phase
        if (it + 2 < iters) { if (ic == 0) setup(iw); } else drain();
        if (it + 1 < iters) { if (icu == 0) setup_u(); } else rs_w = rs_null;
        TSTAMP(0);
    };
    auto post = [&](bool last_chunk) {
        TSTAMP(2);
        if (it + 2 < iters) advance();
        if (it + 1 < iters) advance_u();
        float* nxt = smem + ((it + 1) & 1) * STAGE;
        if (last_chunk) {
            epilogue(cw, smem + (it & 1) * STAGE);
            cw += G;
            TSTAMP(7);
        }
        __syncthreads();
        TSTAMP(4);
        if (it + 1 < iters) transform(nxt);
        TSTAMP(5);
        __syncthreads();
        TSTAMP(6);
        ++it;
    };
    for (int t = 0; t < nmine; ++t) {                      // the first chunk of a tile is peeled: its MFMAs start from a zero constant
        pre();
        compute(smem + (it & 1) * STAGE, smem + ((it + 1) & 1) * STAGE, ic * CK, icu * CK, std::true_type());
        for (int c = 1; c < L.nchunks; ++c) {
            post(false);
            pre();
            compute(smem + (it & 1) * STAGE, smem + ((it + 1) & 1) * STAGE, ic * CK, icu * CK, std::false_type());
        }
        post(true);
    }
}

template <int WCH>
static int launch_wino(const l2i_conv_params& p, hipStream_t st) {
    typedef WinoGeom<WCH> Gm;
    WinoLaunch L;
    L.tiles_x = (p.OW + 31) / 32;
    L.tiles_y = (p.OH + 4 * Gm::WT - 1) / (4 * Gm::WT);
    L.mblocks = (p.CoutP + Gm::BM - 1) / Gm::BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7fffffffL) return l2i_set_error(L2I_E_ARG, "conv2d_wino: too many tiles");
    L.total = (int)total;
    L.nchunks = p.Cin / Gm::CK;
    const size_t lds = (size_t)Gm::LDS_FLOATS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<WCH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<WCH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
        else n_cu = 256;
    }
    const int grid = total < n_cu ? (int)total : n_cu;
    if (p.in_mask) hipLaunchKernelGGL((conv_wino_kernel<WCH, true>), dim3(grid), dim3(256), lds, st, p, L);
    else hipLaunchKernelGGL((conv_wino_kernel<WCH, false>), dim3(grid), dim3(256), lds, st, p, L);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_conv2d_wino_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d_wino: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d_wino: null tensor");
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d_wino: non-positive dimension");
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.oy_step != 1 || p.ox_step != 1 || (p.Cin % 8) != 0)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: needs a 3x3 stride-1 dense-output layer with Cin % 8 == 0");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_wino: CoutP must be Cout rounded up to 32");
    if (p.oy_off < 0 || p.ox_off < 0 || p.OH + p.oy_off > p.OHf || p.OW + p.ox_off > p.OWf)
        return l2i_set_error(L2I_E_ARG, "conv2d_wino: output window exceeds the output tensor");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if ((p.OWf % 4) != 0 || (p.OW % 4) != 0 || (p.ox_off % 4) != 0 || !al16(p.y) || !al16(p.residual) || !al16(p.res_mask) || !al16(p.out_mask) ||
        !al16(p.noise) || !al16(p.w))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: output rows must be 16-byte aligned multiples of 4 pixels");
    if ((size_t)p.Cin * p.H * p.W * sizeof(float) >= 0xFFFFFFF0ull || (size_t)p.Cin * 16 * p.CoutP * sizeof(float) >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: one sample / the weight pack must stay below 4 GiB (32-bit buffer offsets)");
    if ((size_t)p.Cout * p.OHf * p.OWf * sizeof(float) >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: one output sample must stay below 4 GiB (32-bit buffer offsets)");
    hipStream_t st = (hipStream_t)stream;
    int wch = p.tile_hint;
    if (wch == 0) wch = (p.CoutP % 64 == 0) ? 2 : 1;
    if (wch == 2 && (p.CoutP % 64) != 0) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino: the 64-channel tile needs CoutP % 64 == 0");
    if (wch == 1) return launch_wino<1>(p, st);
    if (wch == 2) return launch_wino<2>(p, st);
    return l2i_set_error(L2I_E_ARG, "conv2d_wino: tile_hint must be 0, 1 or 2");
}
