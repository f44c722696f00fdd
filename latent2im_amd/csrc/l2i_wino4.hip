// l2i_wino4.hip — 3x3 stride-1 correlation as Winograd F(4x4, 3x3) on the fp32 matrix cores (gfx950).
//
// The F(2x2,3x3) kernel of l2i_wino.hip spends its time as (MFMA time) + (everything else), additively: an fp32 MFMA runs at the vector rate and
// does not overlap the wave's VALU.  The lever left is fewer MFMAs: Y = A^T [ (G g G^T) . (B^T d B) ] A with a 6x6 input patch gives a 4x4 output
// tile from 36 multiplies per (cin, cout) — 2.25 per output against 4 (F(2x2)) and 9 (direct): 1.78x less matrix work than l2i_wino.hip, and the
// input transform (72 packed VALU per patch) is paid once per 16 output pixels instead of once per 4.  Error against the float64 correlation is
// ~1e-6 .. 1e-5 of max|y| (F(2x2): 3e-7), inside the path's rtol 1e-3 / atol 1e-4; which layers take this kernel is decided by the parity suite
// (conv.WINO4).
//
// A wave owns 16 output channels x 16 tiles (one tile row: 64 x 4 pixels) on v_mfma_f32_16x16x4_f32: 36 positions x 4 registers = 144 accumulators,
// two blocks per CU.  Block = 4 waves = 16 channels x (64 x 16) pixels.  K chunks of 4 channels = ONE MFMA K-step of 36 MFMAs per wave:
//   * raw halo tile [4][18][66] global -> LDS by dword DMA (one element per lane; out-of-image elements through an out-of-range offset = zeros),
//     THREE chunks ahead into a ring of three stages; U (pre-transformed weights, stored in exactly the LDS image order) by 16-byte DMA one
//     chunk ahead, two stages; counted vmcnt waits, one barrier per chunk;
//   * lane (k = lane / 16, n = lane % 16) reads the 6x6 patch of (channel 4 ch + k, tile n) with ds_read_b64 (plane pitch = 2 mod 4 floats: the
//     two channel groups of a 32-lane access interleave on the banks) ONE CHUNK AHEAD of its use, column pair by column pair, and forms
//     V = B^T d B in registers: vertical pass on packed column pairs (12 v_pk per pair), horizontal pass per row with half-selecting packed
//     forms (6 v_pk per row; selections on src0 / src2 / op_sel_hi only — never src1's high half for the low result);
//   * no LDS in the epilogue: a tile row is 4 pixels = one 16-byte store, 16 lanes = 256 contiguous bytes per (channel, row).
//
// Unmasked launches only (plain, style-scaled, ReLU-on-load); a gradient mask would be a second DMA stream: those stay on l2i_wino.hip.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "l2i.h"
#include "l2i_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// timing-ablation switches (tools/probes/w4_ablate.sh: results are wrong by construction)
#ifdef L2I_W4_ABLATE_DMA
#define W4_NODMA 1
#else
#define W4_NODMA 0
#endif
#ifdef L2I_W4_ABLATE_XF
#define W4_NOXF 1
#else
#define W4_NOXF 0
#endif
#ifdef L2I_W4_ABLATE_MFMA
#define W4_NOMFMA 1
#else
#define W4_NOMFMA 0
#endif
#ifdef L2I_W4_ABLATE_LDSD
#define W4_NOLDSD 1
#else
#define W4_NOLDSD 0
#endif
#ifdef L2I_W4_ABLATE_EPI
#define W4_NOEPI 1
#else
#define W4_NOEPI 0
#endif
#ifdef L2I_W4_ABLATE_UREAD
#define W4_NOUREAD 1
#else
#define W4_NOUREAD 0
#endif
#ifdef L2I_W4_ABLATE_BAR
#define W4_NOBAR 1
#else
#define W4_NOBAR 0
#endif

#ifdef L2I_W4_ABLATE_HOTU
#define W4_HOTU 1
#else
#define W4_HOTU 0
#endif
#ifdef L2I_W4_ABLATE_HOTR
#define W4_HOTR 1
#else
#define W4_HOTR 0
#endif
#ifdef L2I_W4_ABLATE_NOUDMA
#define W4_NOUDMA 1
#else
#define W4_NOUDMA 0
#endif
#ifdef L2I_W4_ABLATE_NORDMA
#define W4_NORDMA 1
#else
#define W4_NORDMA 0
#endif
#ifdef L2I_W4_UPF
#define W4_UPF 1
#else
#define W4_UPF 0
#endif

namespace w4 {
constexpr int BM = 16, CK = 4;
constexpr int TW = 64, TH = 16;                         // block tile, pixels
constexpr int IH = TH + 2, IW = TW + 2;                 // 18 x 66 halo tile
constexpr int PLANE_E = IH * IW;                        // 1188 elements per channel
constexpr int NSL = (PLANE_E + 255) / 256;              // 5 DMA slots per channel
constexpr int PLANE = 1218;                             // LDS plane pitch in floats: = 2 (mod 4) -> conflict-free ds_read_b64 of the patches; >= 1216 so that
                                                        // the zero tail of wave 2's last slot stays inside its own plane (wave 3's goes to the dump)
constexpr int RAWST = CK * PLANE;                       // floats per raw stage
constexpr int UST = CK * 36 * BM;                       // floats per U stage (9216 B): [6 rows][4 k][16 ch] float4 (j 0..3) + [6][4][16] float2 (j 4, 5)
constexpr int UA = 6 * 64 * 4;                          // floats of the float4 part
constexpr int NUS = 3;                                  // 16-byte DMA slots per U chunk (576 lanes of 768)
constexpr int DUMP = 256;                               // floats: where slots without a target land
static_assert(NUS + CK * NSL == 23, "23 DMA instructions per wave and chunk, in 5 statements");
// Stage counts (RS raw, US weight stages; RS + US = 5 fits two blocks per CU): a raw tile is read ONE chunk before its MFMAs (the transform is
// pipelined in registers), so RS stages give it RS - 1 chunks of flight; U is read by its own chunk's MFMAs: US stages = US - 1 chunks of flight.
// (3, 2): the raw tile is the operand streamed from HBM (high-resolution layers, small weight pack);  (2, 3): the weight pack is
// (>= 256 channels: 9 .. 38 MB of U against an L2-resident tile).
template <int RS, int US> constexpr int lds_floats() { return RS * RAWST + US * UST + DUMP; }
static_assert(PLANE % 4 == 2 && PLANE >= 1216, "plane pitch");
}

struct Wino4Launch {
    int tiles_x, tiles_y, mblocks;
    int total;
    int nchunks;                       // Cin / 4
};

// ---- packed fp32 VALU helpers (inline asm: hipcc scalarises f32x2 arithmetic).  Constants come in SGPR pairs. ----
#define W4_PK3(name, text)                                                                                           \
    __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; asm(text : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 w4_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 w4_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 w4_mul(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// ReLU-on-load without a packed fp32 max: clamp(x * 2^-60) with the instruction's clamp bit ([0, 1]) is relu(x) * 2^-60 EXACTLY for |x| <= 2^60
// (a power-of-two scale is exact; values below 2^-66 flush to zero: an absolute error of 1e-20) — one v_pk_mul per pair instead of two v_max_f32.
// The accumulators then carry y * 2^-60 (products of order 2^-60 .. 2^-75: far above the denormal range) and the epilogue multiplies back.
#define W4_RELU_SCALE 0x1p-60f
#define W4_RELU_UNSCALE 0x1p60f
__device__ __forceinline__ f32x2 w4_relu_scaled(f32x2 a, f32x2 k) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(k)); return r; }
// a * k + c with the constant pair k in SGPRs
__device__ __forceinline__ f32x2 w4_fmak(f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; }
// horizontal pass (one row; x01, x23, x45 = the row's column pairs):
//   (a.lo * k.lo + c.lo, a.lo * k.hi + c.lo)
__device__ __forceinline__ f32x2 w4_fmak_lo(f32x2 a, f32x2 k, f32x2 c) {
    f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r;
}
//   (a.hi * k.lo + c.hi, a.hi * k.hi + c.hi)
__device__ __forceinline__ f32x2 w4_fmak_hi(f32x2 a, f32x2 k, f32x2 c) {
    f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r;
}
// the three results of a row that feed MFMAs carry the "VALU write -> MFMA read" wait states (hipcc cannot see hazards inside asm)
//   (a.lo + b.lo, a.lo - b.lo)
__device__ __forceinline__ f32x2 w4_lo_pm_lo_op(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b)); return r;
}
//   (c.hi + a.hi * k.lo, c.hi + a.hi * k.hi)
__device__ __forceinline__ f32x2 w4_fmak_hi_op(f32x2 a, f32x2 k, f32x2 c) {
    f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r;
}
__device__ __forceinline__ f32x2 w4_fmak_op(f32x2 a, f32x2 k, f32x2 c) {
    f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 1" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r;
}

// ds_read_b64 by hand: hipcc would merge two of them into ds_read2_b64, which the LDS serves per 16 lanes on 32 banks — the patches of 16 tiles are
// 16 bytes apart, a 2-way conflict on every access.  The matching wait is w4_lds_wait6 (the compiler does not count asm LDS reads).
__device__ __forceinline__ f32x2 w4_lds_b64(unsigned addr, int off) {
    f32x2 r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(off)); return r;
}
__device__ __forceinline__ void w4_lds_wait6(f32x2& a, f32x2& b, f32x2& c, f32x2& d, f32x2& e, f32x2& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}

template <bool SCALE, bool RELU, int RS>
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(const l2i_conv_params p, const Wino4Launch L) {
    using namespace w4;
    constexpr int US = 5 - RS;
    static_assert(RS == 2 || RS == 3, "stage split");
    constexpr bool UFIRST = US == 2;                   // issue order inside a chunk: the operand with ONE chunk of flight goes first, so that the
                                                       // top-of-chunk wait leaves exactly the other operand's youngest chunk in flight
    constexpr int TOPWAIT = UFIRST ? CK * NSL : NUS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rawbuf = smem;                              // RS x [CK][PLANE]
    float* ubuf = smem + RS * RAWST;                   // US x UST
    float* dump = ubuf + US * UST;
    float* stab = dump + DUMP;                         // SCALE: [Cin] style scales of this sample

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, n = lane & 15;           // MFMA K index (channel within the chunk) / tile column
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // XCD-aware order (blocks are dealt round-robin to the 8 XCDs): the channel blocks of one pixel tile and its x-neighbours share an XCD's L2
    const int G = gridDim.x;
    int w = (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    if (w >= L.total) return;
    const int mblk = w % L.mblocks; w /= L.mblocks;
    const int tx = w % L.tiles_x; w /= L.tiles_x;
    const int ty = w % L.tiles_y; w /= L.tiles_y;
    const int b = w, m0 = mblk * BM, oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;

    // ---- staging ----
    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * sizeof(float));
    const unsigned in_bytes = (unsigned)p.Cin * plane_b;
    const size_t smp = (size_t)b * p.Cin * ((size_t)p.H * p.W);
    const unsigned uchunk_b = (unsigned)(p.CoutP / BM) * (unsigned)(UST * sizeof(float));            // bytes of one 4-channel chunk of the weight pack
    constexpr unsigned XSHIFT = (NSL - 2) * 1024u;     // [r6] the descriptor starts this far before the sample: every register offset below stays >= 0
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.x + smp) - XSHIFT), 0, in_bytes + XSHIFT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)(p.Cin / CK) * uchunk_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);
    // element e = u * 256 + tid of the [18][66] plane -> byte offset in the channel plane.  Slots 0..3 of a channel go out in ONE statement with
    // immediate offsets u * 1024 (the immediate moves the LDS target AND the global address: voff carries XSHIFT - u * 1024 against a descriptor that
    // starts XSHIFT bytes early); out-of-image / past-the-plane
    // elements use an offset that is out of range whatever is added to it (one sample stays below 2 GiB: checked at launch) = zeros.
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff[NSL];
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
        const int e = u * 256 + tid;
        const int iy = e / IW, ix = e - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool ok = (e < PLANE_E) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
        voff[u] = ok ? (unsigned)(gy * p.W + gx) * 4u + (XSHIFT - (u < NSL - 1 ? (unsigned)u * 1024u : 0u)) : OOB;
    }
    const unsigned wvoff = (unsigned)tid * 16u;        // U: the chunk image is copied linearly, 16 bytes per lane and slot
    const unsigned lds_raw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)rawbuf;
    const unsigned lds_u = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ubuf;
    const unsigned lds_dump = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dump;

    // DMA of one chunk = 5 statements per wave: U (3 x 16 bytes per lane; lanes 0 .. 575 of 768 carry the image, the rest — slot 2 of waves 1-3 —
    // land in the dump) and one per channel of the raw tile (5 x 4 bytes per lane; wave 3's part of the last slot is past the plane: dump).
    // M0 is written inside the statement that uses it and not restored (hipcc keeps nothing in M0: guide section 5.7; M0 is a RESERVED register of
    // the target — naming it in the clobber list only earns "inline asm clobber list contains reserved registers", the allocator never uses it).  `on` false (no such
    // chunk): null descriptor = zeros, no traffic.
    auto issue_u = [&](int cu, int ust, bool on) {
        const unsigned base = lds_u + (unsigned)(ust * UST * 4) + (unsigned)wave_u * 1024u;
        const unsigned so = (unsigned)cu * uchunk_b + (unsigned)mblk * (unsigned)(UST * 4);
        const bool w0 = wave_u == 0;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %6 offen lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %7 offen lds\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %8 offen lds"
                     :: "v"(wvoff), "s"((on && !W4_NODMA) ? rs_w : rs_null), "s"((on && w0 && !W4_NODMA) ? rs_w : rs_null),
                        "s"(base), "s"(base + 4096u), "s"(w0 ? base + 8192u : lds_dump), "s"(so), "s"(so + 4096u), "s"(so + 8192u) : "memory");
    };
    auto issue_raw = [&](int c, int cr, int rst, bool on) {
        const unsigned base = lds_raw + (unsigned)((rst * RAWST + c * PLANE) * 4) + (unsigned)wave_u * 256u;
        const unsigned so = (unsigned)(cr * CK + c) * plane_b;
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "buffer_load_dword %0, %7, %8 offen lds\n\t"
                     "buffer_load_dword %1, %7, %8 offen offset:1024 lds\n\t"
                     "buffer_load_dword %2, %7, %8 offen offset:2048 lds\n\t"
                     "buffer_load_dword %3, %7, %8 offen offset:3072 lds\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "buffer_load_dword %4, %7, %8 offen lds"
                     :: "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "s"(base), "s"(wave_u < 3 ? base + 4096u : lds_dump),
                        "s"((on && !W4_NODMA) ? rs_x : rs_null), "s"(so) : "memory");
    };

    if constexpr (SCALE) {
        for (int i = tid; i < p.Cin; i += 256) stab[i] = p.in_scale[(size_t)b * p.Cin + i];
    }

    // constants of the transforms (SGPR pairs)
    const f32x2 km4 = {-4.f, -4.f}, k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, k2 = {2.f, 2.f}, km2 = {-2.f, -2.f};
    const f32x2 km4m1 = {-4.f, -1.f}, k2m2 = {2.f, -2.f};

    // ---- input transform, vertical pass of one column pair: rows P[0..5] (packed columns (2 cp, 2 cp + 1)) -> B^T P ----
    auto colpass = [&](f32x2 (&P)[6], f32x2 sc) {
        if (W4_NOXF) return;
        if constexpr (RELU) {
            const f32x2 krelu = {W4_RELU_SCALE, W4_RELU_SCALE};
#pragma unroll
            for (int r = 0; r < 6; ++r) P[r] = w4_relu_scaled(P[r], krelu);
        }
        if constexpr (SCALE) {
#pragma unroll
            for (int r = 0; r < 6; ++r) P[r] = w4_mul(P[r], sc);
        }
        const f32x2 a = w4_fmak(P[2], km4, P[4]), bq = w4_fmak(P[1], km4, P[3]);
        const f32x2 c = w4_sub(P[4], P[2]), e = w4_sub(P[3], P[1]);
        const f32x2 t0 = w4_fmak(P[0], k4, w4_fmak(P[2], km5, P[4]));
        const f32x2 t5 = w4_fmak(P[1], k4, w4_fmak(P[3], km5, P[5]));
        P[0] = t0; P[1] = w4_add(a, bq); P[2] = w4_sub(a, bq); P[3] = w4_fmak(e, k2, c); P[4] = w4_fmak(e, km2, c); P[5] = t5;
    };

    f32x4 acc[36];
    f32x2 Ta[6][3], Tb[6][3];                          // B^T d of the chunk being multiplied / of the next chunk (built during this chunk's MFMAs); the
                                                       // two swap roles every chunk (the loop is unrolled by two: no register copies)

    // patch of (channel kq, tile (wave, n)): byte address of its top-left element in raw stage 0
    const unsigned pa0 = lds_raw + (unsigned)((kq * PLANE + (4 * wave) * IW + 4 * n) * 4);
    auto load_pair = [&](f32x2 (&P)[6], unsigned pa, int cp) {
#pragma unroll
        for (int r = 0; r < 6; ++r) P[r] = W4_NOLDSD ? f32x2{(float)cp, 1.f} : w4_lds_b64(pa, (r * IW + 2 * cp) * 4);
    };
    auto scale_of = [&](int cch) -> f32x2 {            // style scale of channel 4 cch + kq (chunks past the end read a valid, unused entry)
        if constexpr (SCALE) { const int ci = min(cch * CK + kq, p.Cin - 1); const float s = stab[ci]; return f32x2{s, s}; }
        return f32x2{1.f, 1.f};
    };

    // ---- prologue: U(0), raw(0), raw(1), raw(2) ----
    {
        if constexpr (UFIRST) {                        // U(0); raw(0), raw(1), raw(2)
            issue_u(0, 0, true);
#pragma unroll
            for (int c = 0; c < CK; ++c) issue_raw(c, 0, 0, true);
#pragma unroll
            for (int c = 0; c < CK; ++c) issue_raw(c, 1, 1, 1 < L.nchunks);
#pragma unroll
            for (int c = 0; c < CK; ++c) issue_raw(c, 2, 2, 2 < L.nchunks);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * CK * NSL) : "memory");   // U(0), raw(0) landed
        } else {                                       // raw(0), U(0); raw(1), U(1)
#pragma unroll
            for (int c = 0; c < CK; ++c) issue_raw(c, 0, 0, true);
            issue_u(0, 0, true);
#pragma unroll
            for (int c = 0; c < CK; ++c) issue_raw(c, 1, 1, 1 < L.nchunks);
            issue_u(1, 1, 1 < L.nchunks);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CK * NSL + NUS) : "memory");  // raw(0), U(0) landed
        }
        __syncthreads();                                                          // (also: the style-scale table)
        const f32x2 sc = scale_of(0);
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
            f32x2 P[6];
            load_pair(P, pa0, cp);
            w4_lds_wait6(P[0], P[1], P[2], P[3], P[4], P[5]);
            colpass(P, sc);
#pragma unroll
            for (int r = 0; r < 6; ++r) Ta[r][cp] = P[r];
        }
    }

    // ---- one chunk: 36 MFMAs on T and U stage `ucur`; builds Tn from raw stage `rnext` (= raw(ch + 1)); issues the DMA of U(ch + US - 1) into
    //      the U stage before `ucur` and of raw(ch + RS) into the raw stage before `rnext` (the stages chunk ch - 1 was the last to read) ----
    auto chunk = [&](auto first_tag, int ch, int ucur, int rnext, f32x2 (&T)[6][3], f32x2 (&Tn)[6][3]) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const bool on_u = ch + US - 1 < L.nchunks, on_r = ch + RS < L.nchunks;
        const int rstw = rnext == 0 ? RS - 1 : rnext - 1;             // stage of raw(ch + RS) = stage of raw(ch)
        const int ustw = ucur == 0 ? US - 1 : ucur - 1;               // stage of U(ch + US - 1) = stage of U(ch - 1)
        const float4* ua = reinterpret_cast<const float4*>(ubuf + ucur * UST) + lane;              // [i][64 lanes] float4: positions (i, 0..3)
        const float2* ub = reinterpret_cast<const float2*>(ubuf + ucur * UST + UA) + lane;         // [i][64 lanes] float2: positions (i, 4..5)
        const unsigned pa = pa0 + (unsigned)(rnext * RAWST * 4);
        const f32x2 scn = scale_of(ch + 1);
        float4 a4 = W4_NOUREAD ? make_float4(1.f, 2.f, 3.f, 4.f) : ua[0];
        float2 a2 = W4_NOUREAD ? make_float2(5.f, 6.f) : ub[0];
        f32x2 P[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float4 c4 = a4;
            const float2 c2 = a2;
            if (i < 5 && !W4_NOUREAD) { a4 = ua[(i + 1) * 64]; a2 = ub[(i + 1) * 64]; }
            // the next chunk's patch, one column pair per two rows: loads at i = 0, 2, 4; vertical pass at i = 1, 3, 5 (into registers of dead T rows)
            if ((i & 1) == 0) load_pair(P, pa, i >> 1);
            if constexpr (UFIRST) {                                                // one DMA statement per row
                if (i == 0) issue_u(ch + US - 1, ustw, on_u);
                else if (i <= CK) issue_raw(i - 1, ch + RS, rstw, on_r);
            } else {
                if (i < CK) issue_raw(i, ch + RS, rstw, on_r);
                else if (i == CK) issue_u(ch + US - 1, ustw, on_u);
            }
            // horizontal pass of row i: V = (B^T d) B
#if W4_NOXF
            const f32x2 v05 = T[i][0], v12 = T[i][1], v34 = T[i][2];
#else
            const f32x2 ac = w4_fmak_lo(T[i][1], km4m1, T[i][2]);               // (x4 - 4 x2, x4 - x2)
            const f32x2 be = w4_fmak_hi(T[i][0], km4m1, T[i][1]);               // (x3 - 4 x1, x3 - x1)
            const f32x2 v05 = w4_fmak_op(T[i][0], k4, w4_fmak(T[i][1], km5, T[i][2]));   // (4 x0 - 5 x2 + x4, 4 x1 - 5 x3 + x5)
            const f32x2 v12 = w4_lo_pm_lo_op(ac, be);                            // (a + b, a - b)
            const f32x2 v34 = w4_fmak_hi_op(be, k2m2, ac);                       // (c + 2 e, c - 2 e)
#endif
            __builtin_amdgcn_sched_barrier(0);
#if W4_NOMFMA
            if (FIRST) {
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[6 * i + j] = zero;
            }
            acc[6 * i][0] += c4.x * v05.x + c4.y * v12.x + c4.z * v12.y + c4.w * v34.x + c2.x * v34.y + c2.y * v05.y;
#else
            acc[6 * i + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.x, v05.x, FIRST ? zero : acc[6 * i + 0], 0, 0, 0);
            acc[6 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.y, v12.x, FIRST ? zero : acc[6 * i + 1], 0, 0, 0);
            acc[6 * i + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.z, v12.y, FIRST ? zero : acc[6 * i + 2], 0, 0, 0);
            acc[6 * i + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.w, v34.x, FIRST ? zero : acc[6 * i + 3], 0, 0, 0);
            acc[6 * i + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, v34.y, FIRST ? zero : acc[6 * i + 4], 0, 0, 0);
            acc[6 * i + 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, v05.y, FIRST ? zero : acc[6 * i + 5], 0, 0, 0);
#endif
            if (i & 1) {
                w4_lds_wait6(P[0], P[1], P[2], P[3], P[4], P[5]);
                colpass(P, scn);
#pragma unroll
                for (int r = 0; r < 6; ++r) Tn[r][i >> 1] = P[r];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto top = [&]() {
        if (W4_NOBAR) return;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TOPWAIT) : "memory");        // U(ch) and raw(ch + 1) landed; the other operand's youngest chunk stays in flight
        __syncthreads();                                                      // ... for every wave; every wave is past chunk ch - 1: its stages may be refilled
    };
    auto nxt = [](int v, int m) { return v + 1 == m ? 0 : v + 1; };

    top();
    chunk(std::true_type(), 0, 0, 1 % RS, Ta, Tb);
    int ch = 1, rnext = 2 % RS, ucur = 1;
    for (; ch + 1 < L.nchunks; ch += 2) {
        top();
        chunk(std::false_type(), ch, ucur, rnext, Tb, Ta);
        rnext = nxt(rnext, RS); ucur = nxt(ucur, US);
        top();
        chunk(std::false_type(), ch + 1, ucur, rnext, Ta, Tb);
        rnext = nxt(rnext, RS); ucur = nxt(ucur, US);
    }
    if (ch < L.nchunks) {
        top();
        chunk(std::false_type(), ch, ucur, rnext, Tb, Ta);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // no DMA may outlive the block's LDS

    // ---- epilogue: lane-local inverse transform Y = A^T M A, then 16-byte row stores (lane (g, n): channels m0 + 4 g + 0..3, tile (wave, n)) ----
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const int oyb = oy0 + 4 * wave, ox = ox0 + 4 * n;
    const bool xok = ox < p.OW;
    float sq = 0.f;
    const float rc = p.res_sub ? p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // accumulator rows (2 h, 2 h + 1) = two channels as one register pair
        const int co0 = m0 + 4 * kq + 2 * h;
        f32x2 yv[4][6];                                // A^T M: rows 0..3, columns 0..5
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x2 m[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = h ? f32x2{acc[6 * r + j][2], acc[6 * r + j][3]} : f32x2{acc[6 * r + j][0], acc[6 * r + j][1]};
            const f32x2 pp = w4_add(m[1], m[2]), qq = w4_sub(m[1], m[2]), rr = w4_add(m[3], m[4]), ss = w4_sub(m[3], m[4]);
            yv[0][j] = w4_add(w4_add(m[0], pp), rr);
            yv[1][j] = w4_fmak(ss, k2, qq);
            yv[2][j] = w4_fmak(rr, k4, pp);
            yv[3][j] = w4_add(w4_fmak(ss, f32x2{8.f, 8.f}, qq), m[5]);
        }
        float scv[2], bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int co = co0 + q;
            scv[q] = (p.out_scale && co < p.Cout) ? p.out_scale[(size_t)b * p.Cout + co] : 1.f;
            bv[q] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        }
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) {
            const f32x2* m = yv[ry];
            const f32x2 pp = w4_add(m[1], m[2]), qq = w4_sub(m[1], m[2]), rr = w4_add(m[3], m[4]), ss = w4_sub(m[3], m[4]);
            const f32x2 y0 = w4_add(w4_add(m[0], pp), rr), y1 = w4_fmak(ss, k2, qq), y2 = w4_fmak(rr, k4, pp);
            const f32x2 y3 = w4_add(w4_fmak(ss, f32x2{8.f, 8.f}, qq), m[5]);
            const int oy = oyb + ry;
            const bool pok = xok && (oy < p.OH);
            const size_t poff = (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pok && p.noise) {
                nz = *reinterpret_cast<const float4*>(p.noise + (size_t)b * plane_o + poff);
                nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = co0 + q;
                if (!(pok && co < p.Cout)) continue;
                float4 v = q ? make_float4(y0.y, y1.y, y2.y, y3.y) : make_float4(y0.x, y1.x, y2.x, y3.x);
                if constexpr (RELU) { v.x *= W4_RELU_UNSCALE; v.y *= W4_RELU_UNSCALE; v.z *= W4_RELU_UNSCALE; v.w *= W4_RELU_UNSCALE; }
                const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + poff;
                if (p.out_scale) { v.x *= scv[q]; v.y *= scv[q]; v.z *= scv[q]; v.w *= scv[q]; }
                if (p.out_mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(p.out_mask + oidx);
                    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                }
                v.x += nz.x + bv[q]; v.y += nz.y + bv[q]; v.z += nz.z + bv[q]; v.w += nz.w + bv[q];
                if (p.residual) {
                    float4 rv = *reinterpret_cast<const float4*>(p.residual + oidx);
                    if (p.res_sub) {                               // residual term = res_coef * (residual - res_sub)
                        const float4 sb = *reinterpret_cast<const float4*>(p.res_sub + oidx);
                        rv.x = rc * (rv.x - sb.x); rv.y = rc * (rv.y - sb.y); rv.z = rc * (rv.z - sb.z); rv.w = rc * (rv.w - sb.w);
                    }
                    if (p.res_mask) {
                        const float4 mk = *reinterpret_cast<const float4*>(p.res_mask + oidx);
                        rv.x = mk.x > 0.f ? rv.x : 0.f; rv.y = mk.y > 0.f ? rv.y : 0.f; rv.z = mk.z > 0.f ? rv.z : 0.f; rv.w = mk.w > 0.f ? rv.w : 0.f;
                    }
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (p.act == L2I_ACT_LRELU) {                      // max(v, slope v) == (v > 0 ? v : slope v) for 0 <= slope <= 1
                    v.x = __builtin_fmaxf(v.x, v.x * p.act_slope) * p.act_gain; v.y = __builtin_fmaxf(v.y, v.y * p.act_slope) * p.act_gain;
                    v.z = __builtin_fmaxf(v.z, v.z * p.act_slope) * p.act_gain; v.w = __builtin_fmaxf(v.w, v.w * p.act_slope) * p.act_gain;
                } else if (p.act == L2I_ACT_RELU) {
                    v.x = __builtin_fmaxf(v.x, 0.f); v.y = __builtin_fmaxf(v.y, 0.f); v.z = __builtin_fmaxf(v.z, 0.f); v.w = __builtin_fmaxf(v.w, 0.f);
                }
                if (p.out_gain != 1.f) { v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain; }
                if (p.accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                if (!W4_NOEPI || v.x == 123.456f) *reinterpret_cast<float4*>(p.y + oidx) = v;
                if (p.sq_ref) {                                    // ContentLoss value of a VGG tap: sum (y - reference)^2 while y is in registers
                    const float4 rf = *reinterpret_cast<const float4*>(p.sq_ref + oidx);
                    const float d0 = v.x - rf.x, d1 = v.y - rf.y, d2 = v.z - rf.z, d3 = v.w - rf.w;
                    sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
        }
    }
    if (p.sq_ref) {                                                // (kernel argument: uniform branch) one atomic per block, 1024 slots
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        __syncthreads();                                           // every wave is past its vmcnt(0): no target-less DMA slot can still land in the dump
        if (lane == 0) dump[wave] = sq;
        __syncthreads();
        if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (dump[0] + dump[1]) + (dump[2] + dump[3]));
    }
}

// ====================================================================================================================================
// [r5] conv_wino4s_kernel — the same F(4x4,3x3) arithmetic, POSITION-SPLIT: the 36 Winograd positions of a (channel block, tile row) are shared by
// two waves (rows 0..2 / 3..5 of the 6x6 position grid), each of which carries them for TWO 16-channel groups: 18 positions x 2 groups x 4
// registers = the same 144 accumulators and two blocks per CU, but
//   * a wave's input transform is only the half of V = B^T d B it multiplies (3 of the 6 rows of B^T d: 6 v_pk per column pair instead of 12, then
//     the horizontal pass of 3 rows): 36 packed VALU per 36 MFMAs where the first kernel spends 72 — an fp32 MFMA shares its issue with the VALU
//     (tools/probes/mfma_valu_overlap_probe.hip), every VALU instruction of the stream is matrix time;
//   * a block owns 32 output channels (was 16): the raw tile is fetched and staged Cout/32 times, not Cout/16 times;
//   * the raw halo tile goes global -> LDS by 16-BYTE DMA: the LDS window is the 72 columns ox0 - 4 .. ox0 + 67, whose 4-column groups are
//     aligned with the image (W % 4 == 0), so a group is inside or outside the image as a whole (outside: out-of-range offset = zeros) and a
//     (4 channel x 10 row) chunk is 3 DMA instructions per wave instead of 11 dword ones.  The raw planes start 4 bytes off 16-byte alignment
//     (LDS-DMA writes do not care), which puts the patch column pairs (window columns 3 + 4 n + 2 cp) on 8-byte boundaries for ds_read_b64.
// Block = 4 waves = (position half h = wave & 1) x (tile row t = wave >> 1) = 32 channels x (64 x 8) pixels.  After the K loop the two waves of a
// tile row swap halves through LDS (the rings are dead by then): wave (h, t) hands over its rows of group 1 - h and finishes group h with the
// first kernel's lane-local inverse transform and fused epilogue.  Same products, same accumulation order, same inverse: BIT-IDENTICAL to
// conv_wino4_kernel (tests/test_kernels_gpu.py holds the two against each other).
namespace w4s {
constexpr int BMG = 2, BM = 16 * BMG, CK = 4;
// Tile of a block: 64 x 8 pixels (a wave's 16 tiles = one tile row) or, NARROW, 32 x 16 pixels for maps 32 .. 63 wide (a wave's 16 tiles = two
// tile rows of 8).  Both halo windows have 180 16-byte groups: 10 rows x 18 groups, or 18 rows x 10 groups.
// [r5, late] VAR 2 = TALL: 64 x 16 pixels = FOUR tile rows, eight waves (512 threads, one block per CU): the 18 KiB U slice of a chunk feeds 288
// MFMAs instead of 144 (U is 60 % of the kernel's L2 -> LDS traffic, 4 GB per 512 -> 512 @64^2 launch) and every wave issues three U slots per chunk
// instead of five; the raw window (18 x 18 groups per channel) is fetched three slots per wave as before, two waves per channel plane.
template <int VAR> struct Geo {
    static constexpr bool NARROW = VAR == 1, TALL = VAR == 2;
    static constexpr int NW = TALL ? 8 : 4;             // waves per block
    static constexpr int TW = NARROW ? 32 : 64, TH = (NARROW || TALL) ? 16 : 8;
    static constexpr int IH = TH + 2;                   // 10 / 18 halo rows
    static constexpr int IWG = NARROW ? 10 : 18, IWP = 4 * IWG;       // 16-byte groups / window columns per row (72 / 40)
    static constexpr int NGRP = IH * IWG;               // groups per channel plane: 180 (10 x 18 or 18 x 10) / 324 (18 x 18)
    static constexpr int NRSP = (NGRP + 63) / 64;       // 16-byte DMA slots per plane: 3 / 6
    static constexpr int PLANE = NRSP * 256 + 2;        // floats: >= 64 NRSP 4 (the idle lanes of the last slot write zeros inside their own plane), = 2 (mod 4)
    static constexpr int RAWST = 4 * PLANE;             // (CK planes)
    static_assert(NGRP == (TALL ? 324 : 180), "groups per channel plane");
};
constexpr int NRS = 3;                                  // raw DMA slots per WAVE and chunk (4 waves: wave w fetches channel w; TALL: waves w, w + 4 share channel w & 3)
constexpr int PLANE = 770;
constexpr int RAWST = CK * PLANE;                       // (VAR 0 / 1; the body uses Geo<VAR>::RAWST)
constexpr int UG = CK * 36 * 16;                        // floats of one 16-channel group image of a chunk (9216 B, as w4::UST)
constexpr int UST = BMG * UG;                           // a block's slice of a chunk: two consecutive group images = 18432 contiguous bytes of the pack
constexpr int UA = 6 * 64 * 4;
constexpr int NUSLOT = UST * 4 / 1024;                  // 18 real 1 KiB slots per chunk
constexpr int NUS = 5;                                  // slots per wave (slot = wave + 4 j; waves 2, 3: the fifth goes to the dump)
constexpr int RS = 3, US = 2;
constexpr int RAW0 = US * UST + 1;                      // float index of the raw ring: 4 bytes off 16-byte alignment
constexpr int DUMP = 256;
constexpr int LDS_FLOATS = US * UST + 4 + RS * RAWST + DUMP;
constexpr int XCH = 18 * 64 * 4;                        // floats a wave hands over in the epilogue (18 positions x 64 lanes x float4)
static_assert(Geo<0>::PLANE == PLANE && Geo<1>::PLANE == PLANE && Geo<2>::PLANE % 4 == 2, "plane pitch");
static_assert(4 * XCH <= US * UST + 4 + RS * RAWST, "the exchange area lives in the dead rings");
static_assert(NUS * 4 >= NUSLOT && (NUS - 1) * 4 < NUSLOT, "U slots");
}

struct Wino4sLaunch {
    int tiles_x, tiles_y, mblocks;
    int total;
    int nchunks;
};

template <bool SCALE, bool RELU, int H, int VAR>
__device__ __forceinline__ void wino4s_body(const l2i_conv_params& p, const Wino4sLaunch& L, float* smem) {
    using namespace w4s;
    using GE = Geo<VAR>;
    constexpr bool NARROW = GE::NARROW, TALL = GE::TALL;
    constexpr int TW = GE::TW, TH = GE::TH, IWG = GE::IWG, IWP = GE::IWP, NGRP = GE::NGRP, PLANE = GE::PLANE, RAWST = GE::RAWST, NW = GE::NW;
    float* ubuf = smem;                                // US x UST
    float* rawbuf = smem + RAW0;                       // RS x [CK][PLANE]
    float* dump = smem + US * UST + 4 + RS * RAWST;
    float* stab = dump + DUMP;                         // SCALE: [Cin] style scales of this sample

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, n = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int trow = wave_u >> 1;                      // tile row (NARROW: pair of tile rows; TALL: 0 .. 3) of this wave (H = wave & 1: position rows 3 H .. 3 H + 2)
    const int rpl = TALL ? (wave_u & 3) : wave_u;      // raw DMA: channel plane of the chunk this wave fetches ...
    const int rsl = TALL ? 3 * (wave_u >> 2) : 0;      // ... and its first 16-byte slot of that plane
    const int trow_n = NARROW ? 2 * trow + (n >> 3) : trow;          // this lane's tile: row / column inside the block
    const int tcol_n = NARROW ? (n & 7) : n;

    const int G = gridDim.x;
#ifdef L2I_W4_STAMP                                     // timing instrumentation (tools/probes/w4_stamp.py): thread 0 of every block writes the cycle counter at six points into p.ws
    unsigned long long stamp[6];
#define W4_STAMP(i) stamp[i] = __builtin_readcyclecounter()
#else
#define W4_STAMP(i)
#endif
    W4_STAMP(0);
    int w = (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    if (w >= L.total) return;
    const int mblk = w % L.mblocks; w /= L.mblocks;
    const int tx = w % L.tiles_x; w /= L.tiles_x;
    const int ty = w % L.tiles_y; w /= L.tiles_y;
    const int b = w, m0 = mblk * BM, oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 - p.pad_y;

    // ---- staging ----
    const unsigned plane_b = (unsigned)((size_t)p.H * p.W * sizeof(float));
    const unsigned in_bytes = (unsigned)p.Cin * plane_b;
    const size_t smp = (size_t)b * p.Cin * ((size_t)p.H * p.W);
    const unsigned uchunk_b = (unsigned)(p.CoutP / 16) * (unsigned)(UG * sizeof(float));
    // [r6] raw slot u rides on slot 0's M0 with an immediate of 1024 u (LDS target and global address alike): its register offset carries -1024 u.  The
    // descriptor starts XSHIFT bytes before the sample and is XSHIFT bytes longer, so that every register offset (byte + XSHIFT - 1024 u) is >= 0 — round 5
    // let the first pixels of a sample wrap below zero and relied on voffset + immediate being added modulo 2^32 before the range check (round-5 advice).
    constexpr unsigned XSHIFT = (NRS - 1) * 1024u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.x + smp) - XSHIFT), 0, in_bytes + XSHIFT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)(p.Cin / CK) * uchunk_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0u, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff[NRS];                                // group 64 u + lane of the [10][18] plane -> byte offset in the channel plane (+ XSHIFT - the immediate)
#pragma unroll
    for (int u = 0; u < NRS; ++u) {
        const int e = (rsl + u) * 64 + lane;
        const int iy = e / IWG, ig = e - iy * IWG;
        const int gy = iy0 + iy, gx = ox0 - 4 + 4 * ig;
        const bool ok = (e < NGRP) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
        voff[u] = ok ? (unsigned)(gy * p.W + gx) * 4u + (XSHIFT - (unsigned)u * 1024u) : OOB;
    }
    const unsigned wvoff = (unsigned)lane * 16u;
    const unsigned lds_raw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)rawbuf;
    const unsigned lds_u = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ubuf;
    const unsigned lds_dump = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dump;

    // DMA of one chunk = 2 statements per wave: U (5 x 16 bytes per lane: slots wave + 4 j of the 18 KiB image; the fifth slot of waves 2, 3 is
    // past the image: null descriptor into the dump) and the wave's channel plane of the raw tile (3 x 16 bytes per lane).
    auto issue_u = [&](int cu, int ust, bool on) {
        const unsigned base = lds_u + (unsigned)(ust * UST * 4) + (unsigned)wave_u * 1024u;
        const unsigned so = (unsigned)(W4_HOTU ? 0 : cu) * uchunk_b + (unsigned)mblk * (unsigned)(UST * 4) + (unsigned)wave_u * 1024u;
        const bool w01 = wave_u < 2;
        if (W4_NOUDMA) on = false;
        if constexpr (TALL) {                          // 18 slots over 8 waves: slots wave, wave + 8 and (waves 0, 1) wave + 16; the third slot of the others goes to the dump
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %6 offen lds\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %7 offen lds\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %8 offen lds"
                         :: "v"(wvoff), "s"((on && !W4_NODMA) ? rs_w : rs_null), "s"((on && w01 && !W4_NODMA) ? rs_w : rs_null),
                            "s"(base), "s"(base + 8192u), "s"(w01 ? base + 16384u : lds_dump),
                            "s"(so), "s"(so + 8192u), "s"(so + 16384u) : "memory");
        } else {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %8 offen lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %9 offen lds\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %10 offen lds\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %11 offen lds\n\t"
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %12 offen lds"
                     :: "v"(wvoff), "s"((on && !W4_NODMA) ? rs_w : rs_null), "s"((on && w01 && !W4_NODMA) ? rs_w : rs_null),
                        "s"(base), "s"(base + 4096u), "s"(base + 8192u), "s"(base + 12288u), "s"(w01 ? base + 16384u : lds_dump),
                        "s"(so), "s"(so + 4096u), "s"(so + 8192u), "s"(so + 12288u), "s"(so + 16384u) : "memory");
        }
    };
    auto issue_raw = [&](int cr, int rst, bool on) {
        const unsigned base = lds_raw + (unsigned)((rst * RAWST + rpl * PLANE) * 4) + (unsigned)rsl * 1024u;
        const unsigned so = (unsigned)((W4_HOTR ? 0 : cr) * CK + rpl) * plane_b;
        if (W4_NORDMA) on = false;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %4, %5 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %4, %5 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %2, %4, %5 offen offset:2048 lds"
                     :: "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(base), "s"((on && !W4_NODMA) ? rs_x : rs_null), "s"(so) : "memory");
    };

    // L2 prefetch of a U chunk: one dword of each of its 144 128-byte lines, DMA'd into the dump (no register target).  The weight pack of a
    // >= 256-channel layer (9 .. 38 MB) does not live in an XCD's 4 MB L2, and a U stage has ONE chunk (~1 us) of flight: the first block of an
    // XCD to ask for a slice would wait for HBM / the memory-side cache at every chunk.  Issued right AFTER the U DMA of a chunk, the prefetch
    // of U(ch + 3) has until the top of chunk ch + 2 (the in-order vmcnt makes it complete before U(ch + 2)) — two chunks — and the DMA of
    // U(ch + 3), issued at chunk ch + 2, then finds its lines in L2.
    const unsigned pfoff = tid < NUSLOT * 8 ? (unsigned)tid * 128u : OOB;
    auto issue_pf = [&](int cu, bool on) {
        const unsigned so = (unsigned)cu * uchunk_b + (unsigned)mblk * (unsigned)(UST * 4);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %0, %2, %3 offen lds"
                     :: "v"(pfoff), "s"(lds_dump + (unsigned)wave_u * 256u), "s"((on && !W4_NODMA) ? rs_w : rs_null), "s"(so) : "memory");
    };

    if constexpr (SCALE) {
        for (int i = tid; i < p.Cin; i += 64 * NW) stab[i] = p.in_scale[(size_t)b * p.Cin + i];
    }

    const f32x2 km4 = {-4.f, -4.f}, k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, k2 = {2.f, 2.f}, km2 = {-2.f, -2.f};
    const f32x2 km4m1 = {-4.f, -1.f}, k2m2 = {2.f, -2.f};

    // ---- input transform, vertical pass of one column pair: raw rows P[0..5] -> rows 3 H .. 3 H + 2 of B^T P (in P[0..2]) ----
    auto colpass = [&](f32x2 (&P)[6], f32x2 sc) {
        if (W4_NOXF) return;
        constexpr int R0 = H ? 1 : 0;                  // rows the half reads: 0..4 (H = 0), 1..5 (H = 1)
        if constexpr (RELU) {
            const f32x2 krelu = {W4_RELU_SCALE, W4_RELU_SCALE};
#pragma unroll
            for (int r = R0; r < R0 + 5; ++r) P[r] = w4_relu_scaled(P[r], krelu);
        }
        if constexpr (SCALE) {
#pragma unroll
            for (int r = R0; r < R0 + 5; ++r) P[r] = w4_mul(P[r], sc);
        }
        if constexpr (H == 0) {
            const f32x2 a = w4_fmak(P[2], km4, P[4]), bq = w4_fmak(P[1], km4, P[3]);
            const f32x2 t0 = w4_fmak(P[0], k4, w4_fmak(P[2], km5, P[4]));
            P[0] = t0; P[1] = w4_add(a, bq); P[2] = w4_sub(a, bq);
        } else {
            const f32x2 c = w4_sub(P[4], P[2]), e = w4_sub(P[3], P[1]);
            const f32x2 t5 = w4_fmak(P[1], k4, w4_fmak(P[3], km5, P[5]));
            P[0] = w4_fmak(e, k2, c); P[1] = w4_fmak(e, km2, c); P[2] = t5;
        }
    };

    f32x4 acc[BMG][18];
    f32x2 Ta[3][3], Tb[3][3];                          // [position row][column pair] of this half: the chunk being multiplied / the next chunk

    // patch of (channel kq, this lane's tile): its top-left element is window column 3 + 4 tcol_n of halo row 4 trow_n
    const unsigned pa0 = lds_raw + (unsigned)((kq * PLANE + (4 * trow_n) * IWP + 3 + 4 * tcol_n) * 4);
    auto load_pair = [&](f32x2 (&P)[6], unsigned pa, int cp) {
#pragma unroll
        for (int r = 0; r < 6; ++r) P[r] = w4_lds_b64(pa, (r * IWP + 2 * cp) * 4);
    };
    auto scale_of = [&](int cch) -> f32x2 {
        if constexpr (SCALE) { const int ci = min(cch * CK + kq, p.Cin - 1); const float s = stab[ci]; return f32x2{s, s}; }
        return f32x2{1.f, 1.f};
    };

    // ---- prologue: U(0); raw(0), raw(1), raw(2) ----
    issue_u(0, 0, true);
    issue_raw(0, 0, true);
    issue_raw(1, 1, 1 < L.nchunks);
    issue_raw(2, 2, 2 < L.nchunks);
    if (W4_UPF) { issue_pf(1, 1 < L.nchunks); issue_pf(2, 2 < L.nchunks); }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NRS + 2 * W4_UPF) : "memory");   // U(0), raw(0) landed
    __syncthreads();                                                          // (also: the style-scale table)
    W4_STAMP(1);
    {
        const f32x2 sc = scale_of(0);
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
            f32x2 P[6];
            load_pair(P, pa0, cp);
            w4_lds_wait6(P[0], P[1], P[2], P[3], P[4], P[5]);
            colpass(P, sc);
#pragma unroll
            for (int r = 0; r < 3; ++r) Ta[r][cp] = P[r];
        }
    }

    // ---- one chunk: 36 MFMAs (3 position rows x 6 columns x 2 channel groups) on T and U stage `ucur`; builds Tn from raw stage `rnext`
    //      (= raw(ch + 1)); issues U(ch + 1) into the other U stage and raw(ch + 3) into the raw stage chunk ch - 1 was the last to read ----
    auto chunk = [&](auto first_tag, int ch, int ucur, int rnext, f32x2 (&T)[3][3], f32x2 (&Tn)[3][3]) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const bool on_u = ch + 1 < L.nchunks, on_r = ch + RS < L.nchunks;
        const int rstw = rnext == 0 ? RS - 1 : rnext - 1;
        const int ustw = ucur ^ 1;
        const float4* ua = reinterpret_cast<const float4*>(ubuf + ucur * UST) + (3 * H) * 64 + lane;          // group 0, rows 3 H ..: positions (i, 0..3)
        const float2* ub = reinterpret_cast<const float2*>(ubuf + ucur * UST + UA) + (3 * H) * 64 + lane;     //                      positions (i, 4..5)
        const unsigned pa = pa0 + (unsigned)(rnext * RAWST * 4);
        const f32x2 scn = scale_of(ch + 1);
        float4 a4[BMG];
        float2 a2[BMG];
#pragma unroll
        for (int g = 0; g < BMG; ++g) { a4[g] = ua[g * (UG / 4)]; a2[g] = ub[g * (UG / 2)]; }
        f32x2 P[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float4 c4[BMG];
            float2 c2[BMG];
#pragma unroll
            for (int g = 0; g < BMG; ++g) { c4[g] = a4[g]; c2[g] = a2[g]; }
            if (i < 2) {
#pragma unroll
                for (int g = 0; g < BMG; ++g) { a4[g] = ua[g * (UG / 4) + (i + 1) * 64]; a2[g] = ub[g * (UG / 2) + (i + 1) * 64]; }
            }
            load_pair(P, pa, i);                                               // the next chunk's patch, one column pair per position row
            if (i == 0) { issue_u(ch + 1, ustw, on_u); if (W4_UPF) issue_pf(ch + 3, ch + 3 < L.nchunks); }
            else if (i == 1) issue_raw(ch + RS, rstw, on_r);
#if W4_NOXF
            const f32x2 v05 = T[i][0], v12 = T[i][1], v34 = T[i][2];
#else
            const f32x2 ac = w4_fmak_lo(T[i][1], km4m1, T[i][2]);               // (x4 - 4 x2, x4 - x2)
            const f32x2 be = w4_fmak_hi(T[i][0], km4m1, T[i][1]);               // (x3 - 4 x1, x3 - x1)
            const f32x2 v05 = w4_fmak_op(T[i][0], k4, w4_fmak(T[i][1], km5, T[i][2]));
            const f32x2 v12 = w4_lo_pm_lo_op(ac, be);
            const f32x2 v34 = w4_fmak_hi_op(be, k2m2, ac);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < BMG; ++g) {
#if W4_NOMFMA
                if (FIRST) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[g][6 * i + j] = zero;
                }
                acc[g][6 * i][0] += c4[g].x * v05.x + c4[g].y * v12.x + c4[g].z * v12.y + c4[g].w * v34.x + c2[g].x * v34.y + c2[g].y * v05.y;
#else
                acc[g][6 * i + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4[g].x, v05.x, FIRST ? zero : acc[g][6 * i + 0], 0, 0, 0);
                acc[g][6 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4[g].y, v12.x, FIRST ? zero : acc[g][6 * i + 1], 0, 0, 0);
                acc[g][6 * i + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4[g].z, v12.y, FIRST ? zero : acc[g][6 * i + 2], 0, 0, 0);
                acc[g][6 * i + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c4[g].w, v34.x, FIRST ? zero : acc[g][6 * i + 3], 0, 0, 0);
                acc[g][6 * i + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2[g].x, v34.y, FIRST ? zero : acc[g][6 * i + 4], 0, 0, 0);
                acc[g][6 * i + 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2[g].y, v05.y, FIRST ? zero : acc[g][6 * i + 5], 0, 0, 0);
#endif
            }
            w4_lds_wait6(P[0], P[1], P[2], P[3], P[4], P[5]);
            colpass(P, scn);
#pragma unroll
            for (int r = 0; r < 3; ++r) Tn[r][i] = P[r];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto top = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NRS + W4_UPF) : "memory");   // U(ch) and raw(ch + 1) landed; raw(ch + 2) (and the prefetch) stay in flight
        __syncthreads();
    };
    auto nxt = [](int v, int m) { return v + 1 == m ? 0 : v + 1; };

    W4_STAMP(2);
    top();
    chunk(std::true_type(), 0, 0, 1 % RS, Ta, Tb);
    int ch = 1, rnext = 2 % RS, ucur = 1;
    for (; ch + 1 < L.nchunks; ch += 2) {
        top();
        chunk(std::false_type(), ch, ucur, rnext, Tb, Ta);
        rnext = nxt(rnext, RS); ucur ^= 1;
        top();
        chunk(std::false_type(), ch + 1, ucur, rnext, Ta, Tb);
        rnext = nxt(rnext, RS); ucur ^= 1;
    }
    if (ch < L.nchunks) {
        top();
        chunk(std::false_type(), ch, ucur, rnext, Tb, Ta);
    }
    W4_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // no DMA may outlive the block's LDS
    __syncthreads();                                                           // every wave is past its last LDS read and its last DMA: the rings are dead

    // [r6] epilogue operands that do not depend on the accumulators — the demodulation / bias of this lane's four channels and the four noise rows of its
    // tile — are requested HERE, so that they travel during the half exchange below (36 LDS instructions and a barrier) instead of after it: inside the
    // row loop the noise row of ry + 1 could not even be requested before row ry's stores (they may alias as far as the compiler knows).  Same values,
    // same arithmetic: bit-identical; -11 % / -8 % on the generator's 1024^2 / 512^2 launches (profiles/r06_wino4s_strip_ab.txt).
    // -DL2I_W4_EPI_HOIST=0: the round-5 placement (A/B).
#ifndef L2I_W4_EPI_HOIST
#define L2I_W4_EPI_HOIST 1
#endif
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const int oyb = oy0 + 4 * trow_n, ox = ox0 + 4 * tcol_n;
    const bool xok = ox < p.OW;
    float scv_h[2][2], bv_h[2][2];
    float4 nz_h[4];
    if (L2I_W4_EPI_HOIST) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = m0 + 16 * H + 4 * kq + 2 * h + q;
                scv_h[h][q] = (p.out_scale && co < p.Cout) ? p.out_scale[(size_t)b * p.Cout + co] : 1.f;
                bv_h[h][q] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
            }
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) {
            const int oy = oyb + ry;
            nz_h[ry] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xok && oy < p.OH && p.noise) nz_h[ry] = *reinterpret_cast<const float4*>(p.noise + (size_t)b * plane_o + (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off);
        }
    }
    // ---- the two position halves of a tile row meet: wave (H, t) hands over its rows of group 1 - H and finishes group H ----
    {
        float4* xw = reinterpret_cast<float4*>(smem + wave_u * XCH) + lane;
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const f32x4 v = acc[1 - H][q];
            xw[q * 64] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    f32x4 oth[18];                                     // the partner's rows (3 (1 - H) ..) of group H
    {
        const float4* xr = reinterpret_cast<const float4*>(smem + (wave_u ^ 1) * XCH) + lane;
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const float4 v = xr[q * 64];
            oth[q] = f32x4{v.x, v.y, v.z, v.w};
        }
    }
    auto M = [&](int r, int j) -> const f32x4& { return (r / 3 == H) ? acc[H][6 * (r - 3 * H) + j] : oth[6 * (r - 3 * (1 - H)) + j]; };
    W4_STAMP(4);

    // ---- epilogue: lane-local inverse transform Y = A^T M A, then 16-byte row stores (lane (g, n): channels m0 + 16 H + 4 g + 0..3, tile (trow, n)) ----
    float sq = 0.f;
    const float rc = p.res_sub ? p.res_coef * (p.res_coef_dev ? p.res_coef_dev[0] : 1.f) : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // accumulator rows (2 h, 2 h + 1) = two channels as one register pair
        const int co0 = m0 + 16 * H + 4 * kq + 2 * h;
        f32x2 yv[4][6];                                // A^T M: rows 0..3, columns 0..5
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x2 m[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = h ? f32x2{M(r, j)[2], M(r, j)[3]} : f32x2{M(r, j)[0], M(r, j)[1]};
            const f32x2 pp = w4_add(m[1], m[2]), qq = w4_sub(m[1], m[2]), rr = w4_add(m[3], m[4]), ss = w4_sub(m[3], m[4]);
            yv[0][j] = w4_add(w4_add(m[0], pp), rr);
            yv[1][j] = w4_fmak(ss, k2, qq);
            yv[2][j] = w4_fmak(rr, k4, pp);
            yv[3][j] = w4_add(w4_fmak(ss, f32x2{8.f, 8.f}, qq), m[5]);
        }
        float scv[2], bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int co = co0 + q;
            if (L2I_W4_EPI_HOIST) { scv[q] = scv_h[h][q]; bv[q] = bv_h[h][q]; continue; }
            scv[q] = (p.out_scale && co < p.Cout) ? p.out_scale[(size_t)b * p.Cout + co] : 1.f;
            bv[q] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        }
        float4 prow[2];                                // pool_out: the even row of the current window pair, per channel of the register pair
        bool pvalid[2] = {false, false};
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) {
            const f32x2* m = yv[ry];
            const f32x2 pp = w4_add(m[1], m[2]), qq = w4_sub(m[1], m[2]), rr = w4_add(m[3], m[4]), ss = w4_sub(m[3], m[4]);
            const f32x2 y0 = w4_add(w4_add(m[0], pp), rr), y1 = w4_fmak(ss, k2, qq), y2 = w4_fmak(rr, k4, pp);
            const f32x2 y3 = w4_add(w4_fmak(ss, f32x2{8.f, 8.f}, qq), m[5]);
            const int oy = oyb + ry;
            const bool pok = xok && (oy < p.OH);
            const size_t poff = (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pok && p.noise) {
                nz = L2I_W4_EPI_HOIST ? nz_h[ry] : *reinterpret_cast<const float4*>(p.noise + (size_t)b * plane_o + poff);
                nz.x *= p.noise_w; nz.y *= p.noise_w; nz.z *= p.noise_w; nz.w *= p.noise_w;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = co0 + q;
                if (!(pok && co < p.Cout)) { pvalid[q] = false; continue; }
                float4 v = q ? make_float4(y0.y, y1.y, y2.y, y3.y) : make_float4(y0.x, y1.x, y2.x, y3.x);
                if constexpr (RELU) { v.x *= W4_RELU_UNSCALE; v.y *= W4_RELU_UNSCALE; v.z *= W4_RELU_UNSCALE; v.w *= W4_RELU_UNSCALE; }
                const size_t oidx = ((size_t)b * p.Cout + co) * plane_o + poff;
                if (p.out_scale) { v.x *= scv[q]; v.y *= scv[q]; v.z *= scv[q]; v.w *= scv[q]; }
                if (p.out_mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(p.out_mask + oidx);
                    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                }
                v.x += nz.x + bv[q]; v.y += nz.y + bv[q]; v.z += nz.z + bv[q]; v.w += nz.w + bv[q];
                if (p.residual) {
                    float4 rv = *reinterpret_cast<const float4*>(p.residual + oidx);
                    if (p.res_sub) {                               // residual term = res_coef * (residual - res_sub)
                        const float4 sb = *reinterpret_cast<const float4*>(p.res_sub + oidx);
                        rv.x = rc * (rv.x - sb.x); rv.y = rc * (rv.y - sb.y); rv.z = rc * (rv.z - sb.z); rv.w = rc * (rv.w - sb.w);
                    }
                    if (p.res_mask) {
                        const float4 mk = *reinterpret_cast<const float4*>(p.res_mask + oidx);
                        rv.x = mk.x > 0.f ? rv.x : 0.f; rv.y = mk.y > 0.f ? rv.y : 0.f; rv.z = mk.z > 0.f ? rv.z : 0.f; rv.w = mk.w > 0.f ? rv.w : 0.f;
                    }
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (p.act == L2I_ACT_LRELU) {                      // max(v, slope v) == (v > 0 ? v : slope v) for 0 <= slope <= 1
                    v.x = __builtin_fmaxf(v.x, v.x * p.act_slope) * p.act_gain; v.y = __builtin_fmaxf(v.y, v.y * p.act_slope) * p.act_gain;
                    v.z = __builtin_fmaxf(v.z, v.z * p.act_slope) * p.act_gain; v.w = __builtin_fmaxf(v.w, v.w * p.act_slope) * p.act_gain;
                } else if (p.act == L2I_ACT_RELU) {
                    v.x = __builtin_fmaxf(v.x, 0.f); v.y = __builtin_fmaxf(v.y, 0.f); v.z = __builtin_fmaxf(v.z, 0.f); v.w = __builtin_fmaxf(v.w, 0.f);
                }
                if (p.out_gain != 1.f) { v.x *= p.out_gain; v.y *= p.out_gain; v.z *= p.out_gain; v.w *= p.out_gain; }
                if (p.accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(p.y + oidx);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                if (!W4_NOEPI || v.x == 123.456f) *reinterpret_cast<float4*>(p.y + oidx) = v;
                if (p.pool_out) {                                  // [r5] MaxPool2d(2, 2) of y from the tile in registers (l2i.h: pool_out / pool_idx)
                    if ((ry & 1) == 0) { prow[q] = v; pvalid[q] = true; }
                    else if (pvalid[q]) {
                        const float4 u = prow[q];
                        float b0 = u.x, b1 = u.z;
                        unsigned i0 = 0u, i1 = 0u;
                        if (u.y > b0 || u.y != u.y) { b0 = u.y; i0 = 1u; }
                        if (v.x > b0 || v.x != v.x) { b0 = v.x; i0 = 2u; }
                        if (v.y > b0 || v.y != v.y) { b0 = v.y; i0 = 3u; }
                        if (u.w > b1 || u.w != u.w) { b1 = u.w; i1 = 1u; }
                        if (v.z > b1 || v.z != v.z) { b1 = v.z; i1 = 2u; }
                        if (v.w > b1 || v.w != v.w) { b1 = v.w; i1 = 3u; }
                        const size_t pidx = (((size_t)b * p.Cout + co) * (size_t)(p.OHf >> 1) + (size_t)(oy >> 1)) * (size_t)(p.OWf >> 1) + (size_t)(ox >> 1);
                        *reinterpret_cast<float2*>(p.pool_out + pidx) = make_float2(b0, b1);
                        *reinterpret_cast<unsigned short*>(p.pool_idx + pidx) = (unsigned short)(i0 | (i1 << 8));
                    }
                }
                if (p.sq_ref) {                                    // ContentLoss value of a VGG tap: sum (y - reference)^2 while y is in registers
                    const float4 rf = *reinterpret_cast<const float4*>(p.sq_ref + oidx);
                    const float d0 = v.x - rf.x, d1 = v.y - rf.y, d2 = v.z - rf.z, d3 = v.w - rf.w;
                    sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
        }
    }
    if (p.sq_ref) {                                                // (kernel argument: uniform branch) one atomic per block, 1024 slots
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        __syncthreads();                                           // every wave has read its partner's exchange area: the partial sums may overwrite it
        if (lane == 0) smem[wave_u] = sq;
        __syncthreads();
        if (tid == 0) {
            float t = (smem[0] + smem[1]) + (smem[2] + smem[3]);
            if constexpr (TALL) t += (smem[4] + smem[5]) + (smem[6] + smem[7]);
            atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), t);
        }
    }
#ifdef L2I_W4_STAMP
    W4_STAMP(5);
    if (p.ws && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) o[i] = stamp[i];
    }
#endif
}

template <bool SCALE, bool RELU, int VAR = 0>
__global__ __launch_bounds__(VAR == 2 ? 512 : 256, VAR == 2 ? 1 : 2) void conv_wino4s_kernel(const l2i_conv_params p, const Wino4sLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (a wave-uniform branch: the two halves run the same number of barriers)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) wino4s_body<SCALE, RELU, 1, VAR>(p, L, smem);
    else wino4s_body<SCALE, RELU, 0, VAR>(p, L, smem);
}

static int launch_wino4s(const l2i_conv_params& p, hipStream_t st) {
    Wino4sLaunch L;
    const bool narrow = p.OW < 64;                     // maps 32 .. 63 wide: the 32 x 16-pixel tile
    // TALL (64 x 16 pixels, eight waves): L2I_W4_TALL=1 takes it on every launch with >= 16 rows, 0 = never (the A/B: DESIGN.md section 4.0)
    static const int tall_env = getenv("L2I_W4_TALL") ? atoi(getenv("L2I_W4_TALL")) : 0;
    const bool tall = !narrow && p.OH >= 16 && (tall_env != 0 || p.tile_hint == 2);
    L.tiles_x = narrow ? (p.OW + 31) / 32 : (p.OW + 63) / 64;
    L.tiles_y = (narrow || tall) ? (p.OH + 15) / 16 : (p.OH + 7) / 8;
    L.mblocks = p.CoutP / w4s::BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: too many tiles");
    L.total = (int)total;
    L.nchunks = p.Cin / w4s::CK;
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    const bool relu_in = p.in_mask != nullptr;
    const bool scale = p.in_scale != nullptr;
    const int stab_f = scale ? ((p.Cin + 3) & ~3) : 0;
    size_t lds = (size_t)(w4s::LDS_FLOATS + stab_f) * sizeof(float);
    if (tall) {                                        // rings: U as before, three raw stages of 4 x 1538 floats; the epilogue's exchange area: eight waves x 18 KiB
        const size_t ring = (size_t)(w4s::US * w4s::UST + 4 + w4s::RS * w4s::Geo<2>::RAWST + w4s::DUMP + stab_f) * sizeof(float);
        const size_t xch = (size_t)8 * w4s::XCH * sizeof(float);
        lds = ring > xch ? ring : xch;
    }
    if (lds > (tall ? 160u : 80u) * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: too many input channels for the style-scale table");
#define L2I_WINO4S_(S_, R_, V_)                                                                                                         \
    do {                                                                                                                                \
        L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4s_kernel<S_, R_, V_>),                    \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (V_ == 2 ? 160 : 80) * 1024));       \
        hipLaunchKernelGGL((conv_wino4s_kernel<S_, R_, V_>), dim3(grid), dim3(V_ == 2 ? 512 : 256), lds, st, p, L);                      \
    } while (0)
#define L2I_WINO4S(S_, R_) do { if (narrow) L2I_WINO4S_(S_, R_, 1); else if (tall) L2I_WINO4S_(S_, R_, 2); else L2I_WINO4S_(S_, R_, 0); } while (0)
    if (relu_in) { if (scale) L2I_WINO4S(true, true); else L2I_WINO4S(false, true); }
    else { if (scale) L2I_WINO4S(true, false); else L2I_WINO4S(false, false); }
#undef L2I_WINO4S
#undef L2I_WINO4S_
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

static int launch_wino4(const l2i_conv_params& p, hipStream_t st) {
    Wino4Launch L;
    L.tiles_x = (p.OW + w4::TW - 1) / w4::TW;
    L.tiles_y = (p.OH + w4::TH - 1) / w4::TH;
    L.mblocks = p.CoutP / w4::BM;
    const long total = (long)p.B * L.tiles_y * L.tiles_x * L.mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: too many tiles");
    L.total = (int)total;
    L.nchunks = p.Cin / w4::CK;
    const unsigned grid = (unsigned)((total + 7) & ~7L);
    const bool relu_in = p.in_mask != nullptr;         // (checked by the caller: the mask IS the input, ReLU slopes)
    const bool scale = p.in_scale != nullptr;
    // which operand gets the deeper DMA ring (see w4::lds_floats): the weight pack once it no longer fits an XCD's L2 next to the tiles
    static const int rs_env = getenv("L2I_W4_RS") ? atoi(getenv("L2I_W4_RS")) : 0;
    const int rs = (rs_env == 2 || rs_env == 3) ? rs_env : (((size_t)p.Cin * p.CoutP * 36 * sizeof(float) > (size_t)(3u << 20)) ? 2 : 3);
    const size_t lds = (size_t)((rs == 3 ? w4::lds_floats<3, 2>() : w4::lds_floats<2, 3>()) + (scale ? ((p.Cin + 3) & ~3) : 0)) * sizeof(float);
    if (lds > 80 * 1024) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: too many input channels for the style-scale table");
#define L2I_WINO4(S_, R_, RS_)                                                                                                          \
    do {                                                                                                                                \
        L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4_kernel<S_, R_, RS_>),                    \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));                         \
        hipLaunchKernelGGL((conv_wino4_kernel<S_, R_, RS_>), dim3(grid), dim3(256), lds, st, p, L);                                      \
    } while (0)
#define L2I_WINO4_RS(S_, R_) do { if (rs == 3) L2I_WINO4(S_, R_, 3); else L2I_WINO4(S_, R_, 2); } while (0)
    if (relu_in) { if (scale) L2I_WINO4_RS(true, true); else L2I_WINO4_RS(false, true); }
    else { if (scale) L2I_WINO4_RS(true, false); else L2I_WINO4_RS(false, false); }
#undef L2I_WINO4_RS
#undef L2I_WINO4
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_conv2d_wino4_f32(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, true, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0)
        return l2i_set_error(L2I_E_ARG, "conv2d_wino4: non-positive dimension");
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.oy_step != 1 || p.ox_step != 1 || (p.Cin % 4) != 0)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: needs a 3x3 stride-1 dense-output layer with Cin % 4 == 0");
    const bool relu_in = p.in_mask && (const void*)p.in_mask == (const void*)p.x && p.mask_pos == 1.f && p.mask_neg == 0.f;
    if (p.in_mask && !relu_in) return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: gradient-masked launches take l2i_conv2d_wino_f32");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: CoutP must be Cout rounded up to a multiple of 32");
    if (p.oy_off < 0 || p.ox_off < 0 || p.OH + p.oy_off > p.OHf || p.OW + p.ox_off > p.OWf)
        return l2i_set_error(L2I_E_ARG, "conv2d_wino4: output window exceeds the output tensor");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if ((p.OWf % 4) != 0 || (p.OW % 4) != 0 || (p.ox_off % 4) != 0 || !al16(p.y) || !al16(p.residual) || !al16(p.res_mask) || !al16(p.res_sub) || !al16(p.out_mask) ||
        !al16(p.noise) || !al16(p.w))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: output rows must be 16-byte aligned multiples of 4 pixels");
    if ((size_t)p.Cin * p.H * p.W * sizeof(float) >= 0x7FFF0000ull || (size_t)p.Cin * 36 * p.CoutP * sizeof(float) >= 0xFFFFFFF0ull)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: one sample must stay below 2 GiB, the weight pack below 4 GiB (32-bit buffer offsets)");
    if (p.tile_hint < 0 || p.tile_hint > 2) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: tile_hint must be 0 (position-split kernel), 1 (the round-4 kernel) or 2 (position-split, 64 x 16 tile on eight waves)");
    if ((p.sq_ref != nullptr) != (p.sq_out != nullptr) || (((uintptr_t)p.sq_ref) % 16) != 0) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: sq_ref (16-byte aligned) and sq_out go together");
    if ((p.pool_out != nullptr) != (p.pool_idx != nullptr)) return l2i_set_error(L2I_E_ARG, "conv2d_wino4: pool_out and pool_idx go together");
    if (p.pool_out && (p.tile_hint == 1 || (p.OHf % 2) != 0 || (p.OWf % 4) != 0 || p.oy_off != 0 || p.ox_off != 0 || p.OH != p.OHf || p.OW != p.OWf || (((uintptr_t)p.pool_out) % 8) != 0 ||
                       (((uintptr_t)p.pool_idx) % 2) != 0))
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: pool_out needs the position-split kernel, a dense output with even height and OWf % 4 == 0, an 8-byte aligned pool_out");
    if (p.tile_hint == 1) return launch_wino4(p, (hipStream_t)stream);
    if (p.pad_x != 1 || (p.W % 4) != 0)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv2d_wino4: the position-split kernel needs pad_x == 1 and W % 4 == 0 (16-byte groups of the halo window aligned with the image)");
    return launch_wino4s(p, (hipStream_t)stream);
}
