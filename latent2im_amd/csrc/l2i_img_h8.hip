// l2i_img_h8.hip — [r5] the image-side convolutions of the 16-bit path: fp32 NCHW image with <= 4 channels in, 16-bit h8 feature map out, one
// v_mfma_f32_32x32x16_{bf16,f16} product per MAC (gfx950).  Entry points l2i_conv_img_h8 / l2i_conv_img_h8_f16.
//   VGG-19 conv1_1      3x3 / stride 1 / pad 1,  3 -> 64   (transform_base.py:426-454)
//   discriminator conv0 1x1 / stride 1,          3 -> C    (networks.py:568-575: ConvLayer(3, channels[size], 1))
//   ResNet-50 conv1     7x7 / stride 2 / pad 3,  3 -> 64   (transform_base.py:396-403 -> torchvision resnet50)
// Before: the image was cast into a zero-padded 16- or 32-channel h8 tensor (268 / 537 MB written and read back at 1024^2 batch 8 for 100 MB of
// image) and fed to the generic 16-bit conv, the stem ran on the fp32 implicit-GEMM kernel into an fp32 map (537 MB) followed by a cast pass.
// Here the kernel reads the fp32 image itself and stores what the next 16-bit kernel reads.
//
// im2col on the fly.  The contraction index of the MFMA is laid out as k' = 16 s + 8 half + e  <->  (row p = 2 s + half, column e):
// row p = (input channel c, kernel row ky) = (p / K, p % K), column e = kernel column kx (e >= K and p >= Cin K multiply zeros).  A lane of
// the B operand (pixel j, half) therefore reads K CONSECUTIVE floats of one staged image row per step — one address register per step
// (c PLANE + ky IWP + S j), the K columns as immediate offsets — and packs them to 16 bits (this is where the image is rounded, as the cast pass
// did).  Steps: ceil(Cin K / 2) = 2 / 5 / 11 for K = 1 / 3 / 7: up to 2.5x the MFMA work of a dense K index, on layers whose matrix time is
// a few per cent of their output-write time.  A operand: the weights in the same k' order, 16-bit planes [step][half][CoutP][8]
// (latent2im_amd/conv.py:pack_weight_img_h8), staged once per block into LDS.
// Block = 4 waves, output tile 8 rows x 64 columns x 64 channels; a wave walks 4 strips of 32 pixels.  Epilogue: bias, activation, out_gain,
// optional (y - sq_ref)^2 sum on the rounded output (the ContentLoss value of VGG conv_1, l2i.h: sq_ref / sq_out), 16-byte stores of whole
// h8 slots after the v_permlane32_swap exchange of l2i_conv_h8.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8_ __attribute__((ext_vector_type(8)));

namespace img8 {
constexpr int TH = 8, TW = 64, BM = 64;
template <int K, int S> struct Geo {
    static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
    static constexpr int IWP = IW | 1;                                     // odd pitch: the stride-2 reads of the stem spread over the banks
    static constexpr int PLANE = IH * IWP;
    static constexpr int ZPAD = 8;                                         // zeros behind the tile: rows p >= Cin K read them
    // tiles a block walks along x (the next tile's image loads fly during this tile's MFMAs and stores).  Measured at 1024^2 batch 8, NT = 1 / 4:
    // 1x1 0.192 / 0.160 ms, 7x7 stride 2 0.233 / 0.200, 3x3 0.364 / 0.400 (0.503 / 0.634 with the ContentLoss sum): the 3x3 keeps one tile per block
    static constexpr int NT = K == 3 ? 1 : 4;
};

template <bool F16> struct Elem;
template <> struct Elem<false> {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ unsigned pk(float lo, float hi) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Elem<true> {
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ unsigned pk(float lo, float hi) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
}  // namespace img8

template <bool F16, int K, int S>
__global__ __launch_bounds__(256, K <= 3 ? 4 : 2) void conv_img_h8_kernel(const l2i_conv_params p, int tiles_x, int tiles_y, int mblocks, int nsteps, int tile_f) {
    using namespace img8;
    using G = Geo<K, S>;
    using E = Elem<F16>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const tile = smem;                                              // [c][row][col] fp32, zero outside the image, ZPAD zeros behind
    u32x4* const wl = reinterpret_cast<u32x4*>(smem + tile_f);             // [step][half][BM] 16-byte slots (tile_f = Cin PLANE + ZPAD, rounded to 16 bytes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    int bid = blockIdx.x;
    const int mblk = bid % mblocks; bid /= mblocks;
    constexpr int NT = G::NT;
    const int xgroups = (tiles_x + NT - 1) / NT;
    const int txg = bid % xgroups; bid /= xgroups;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid, m0 = mblk * BM;
    const int oy0 = ty * TH;
    const int iy0 = oy0 * S - p.pad_y;
    const int tx_first = txg * NT, nt = tiles_x - tx_first < NT ? tiles_x - tx_first : NT;

    // ---- stage: weights (16-byte slots) and the halo tile ----
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.w_hi);
    for (int e = tid; e < nsteps * 2 * BM; e += 256) {
        const int r = e / BM, i = e - r * BM;                              // r = 2 step + half
        wl[e] = (m0 + i < p.CoutP) ? wg[(size_t)r * p.CoutP + m0 + i] : u32x4{0u, 0u, 0u, 0u};
    }
    // every load of the thread is issued before the first LDS write (a row-by-row loop exposed one memory round trip per row: 8 - 16 per block)
    const size_t plane_x = (size_t)p.H * p.W;
    const float* xs = p.x + (size_t)b * p.Cin * plane_x;
    constexpr int NE = (4 * G::PLANE + 255) / 256;                          // tile elements per thread at 4 input channels
    const int nelem = p.Cin * G::PLANE;
    float stg[NE];
    auto load_tile = [&](int tx) {
        const int ix0 = tx * TW * S - p.pad_x;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            const int c = e / G::PLANE, r2 = e - c * G::PLANE;
            const int r = r2 / G::IWP, col = r2 - r * G::IWP;
            const int gy = iy0 + r, gx = ix0 + col;
            stg[i] = (e < nelem && col < G::IW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? xs[(size_t)c * plane_x + (size_t)gy * p.W + gx] : 0.f;
        }
    };
    auto commit_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            if (e < nelem) tile[e] = stg[i];
        }
    };
    load_tile(tx_first);
    commit_tile();
    float* const zpad = tile + p.Cin * G::PLANE;
    if (tid < G::ZPAD) zpad[tid] = 0.f;
    __syncthreads();

    const int rows = p.Cin * K;
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const int cg = p.Cout >> 3;
    const bool two = m0 + 32 < p.CoutP;                                    // second 32-channel tile of the block exists (block-uniform)
    // epilogue constants of this lane's four (tile, quad pair) groups: group (m0 >> 3) + 4 m + 2 pr + half
    float bs[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co0 = ((m0 >> 3) + 4 * (q >> 1) + 2 * (q & 1) + half) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[q][e] = (p.bias && co0 + e < p.Cout) ? p.bias[co0 + e] : 0.f;
    }
    const float gpos = (p.act == L2I_ACT_LRELU ? p.act_gain : 1.f) * p.out_gain;
    const float gneg = p.act == L2I_ACT_LRELU ? p.act_slope * p.act_gain * p.out_gain : (p.act == L2I_ACT_RELU ? 0.f : p.out_gain);
    float sq = 0.f;

#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
    const int ox0 = (tx_first + t) * TW;
    if (t + 1 < nt) load_tile(tx_first + t + 1);
#pragma unroll 1
    for (int st = 0; st < 4; ++st) {
        const int row = 2 * wave + (st >> 1), cx = (st & 1) * 32;
        const float* bp = tile + (row * S) * G::IWP + (cx + j) * S;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int pr_ = 2 * s + half;
            const int c = pr_ / K, ky = pr_ - c * K;
            const float* rp = pr_ < rows ? bp + c * G::PLANE + ky * G::IWP : zpad;      // rows past Cin K: the zero pad
            // (a compiler-generated conversion, not the inline-asm pack of the epilogue: the hazard recognizer does not look inside inline asm, and an
            // MFMA that reads a VGPR written by the VALU instruction right before it gets the old value — measured: garbage in the first tile only)
            f32x8_ v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = e < K ? rp[e] : 0.f;
            const typename E::v8 bf = __builtin_convertvector(v, typename E::v8);
            const u32x4* wa = wl + (2 * s + half) * BM + j;
            acc0 = E::mfma(__builtin_bit_cast(typename E::v8, wa[0]), bf, acc0);
            if (two) acc1 = E::mfma(__builtin_bit_cast(typename E::v8, wa[32]), bf, acc1);
        }
        // ---- epilogue: quads -> whole 8-channel slots, max(v gpos, v gneg) = identity / ReLU / leaky ReLU with gains, one 16-byte store ----
        const int oy = oy0 + row, ox = ox0 + cx + j;
        const bool pok = oy < p.OH && ox < p.OW;
        const size_t slot0 = (size_t)b * cg * plane_o + (size_t)(oy + p.oy_off) * p.OWf + ox + p.ox_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= 2 && !two) break;
            const f32x16& a = q < 2 ? acc0 : acc1;
            const int pr = q & 1;
            float lo[4], hi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = a[8 * pr + e]; hi[e] = a[8 * pr + 4 + e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[e]), __float_as_uint(hi[e]), false, false);
                lo[e] = __uint_as_float(r[0]); hi[e] = __uint_as_float(r[1]);
            }
            const int grp = (m0 >> 3) + 4 * (q >> 1) + 2 * pr + half;
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[e] + bs[q][e];
                v[e] = fmaxf(t * gpos, t * gneg);
            }
            const u32x4 out = {E::pk(v[0], v[1]), E::pk(v[2], v[3]), E::pk(v[4], v[5]), E::pk(v[6], v[7])};
            if (pok && grp * 8 < p.Cout) {
                const size_t slot = slot0 + (size_t)grp * plane_o;
                reinterpret_cast<u32x4*>(p.y)[slot] = out;
                if (p.sq_ref) {
                    const u32x4 rf = reinterpret_cast<const u32x4*>(p.sq_ref)[slot];
                    const unsigned o4[4] = {out.x, out.y, out.z, out.w}, r4[4] = {rf.x, rf.y, rf.z, rf.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d0 = l2i_h8_lo(o4[e], F16) - l2i_h8_lo(r4[e], F16), d1 = l2i_h8_hi(o4[e], F16) - l2i_h8_hi(r4[e], F16);
                        sq += d0 * d0 + d1 * d1;
                    }
                }
            }
        }
    }
    if (t + 1 < nt) {
        __syncthreads();                                                   // every wave is done with this tile
        commit_tile();
        __syncthreads();
    }
    }
    if (p.sq_ref) {                                                        // one atomic per block into L2I_SQ_SLOTS slots
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        __syncthreads();                                                   // the staged tile is no longer read
        if (lane == 0) tile[wave] = sq;
        __syncthreads();
        if (tid == 0) atomicAdd(p.sq_out + (blockIdx.x & (L2I_SQ_SLOTS - 1)), (tile[0] + tile[1]) + (tile[2] + tile[3]));
    }
}

template <bool F16, int K, int S>
static int launch_img(const l2i_conv_params& p, hipStream_t st) {
    using namespace img8;
    using G = Geo<K, S>;
    const int nsteps = (p.Cin * K + 1) / 2;
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH, mblocks = (p.CoutP + BM - 1) / BM;
    const long total = (long)p.B * tiles_y * ((tiles_x + G::NT - 1) / G::NT) * mblocks;
    if (total <= 0 || total > 0x7ffffff0L) return l2i_set_error(L2I_E_ARG, "conv_img_h8: grid too large");
    const int tile_f = (p.Cin * G::PLANE + G::ZPAD + 3) & ~3;
    const size_t lds = (size_t)tile_f * sizeof(float) + (size_t)nsteps * 2 * BM * 16;
    L2I_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_img_h8_kernel<F16, K, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((conv_img_h8_kernel<F16, K, S>), dim3((unsigned)total), dim3(256), lds, st, p, tiles_x, tiles_y, mblocks, nsteps, tile_f);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

template <bool F16>
static int conv_img_h8(const l2i_conv_params* pp, void* stream) {
    if (!pp) return l2i_set_error(L2I_E_ARG, "conv_img_h8: null params");
    const l2i_conv_params& p = *pp;
    if (!p.x || !p.w_hi || !p.y) return l2i_set_error(L2I_E_ARG, "conv_img_h8: null tensor");
    if (const char* m = l2i_unsupported_v5_fields(p, false, false, false)) return l2i_set_error(L2I_E_UNSUPPORTED, m);
    if (p.B <= 0 || p.Cin <= 0 || p.Cin > 4 || p.Cout <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0) return l2i_set_error(L2I_E_ARG, "conv_img_h8: bad dimension (1 .. 4 input channels)");
    if (p.CoutP < p.Cout || (p.CoutP % 32) != 0 || (p.Cout % 8) != 0) return l2i_set_error(L2I_E_ARG, "conv_img_h8: CoutP = Cout rounded up to 32, Cout % 8 == 0");
    if (p.KH != p.KW || p.pad_y != p.pad_x || p.oy_step != 1 || p.ox_step != 1 || p.oy_off < 0 || p.ox_off < 0 || p.OH + p.oy_off > p.OHf || p.OW + p.ox_off > p.OWf)
        return l2i_set_error(L2I_E_ARG, "conv_img_h8: square kernel, symmetric padding, dense output window inside the output tensor");
    if (p.in_scale || p.in_mask || p.out_scale || p.noise || p.residual || p.res_mask || p.out_mask || p.res_sub || p.accumulate || p.ksplit > 1 || p.out_f32 || p.rgb_w)
        return l2i_set_error(L2I_E_UNSUPPORTED, "conv_img_h8: fused terms are bias, act, out_gain, sq_ref / sq_out");
    if ((p.sq_ref != nullptr) != (p.sq_out != nullptr)) return l2i_set_error(L2I_E_ARG, "conv_img_h8: sq_ref and sq_out go together");
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    if (!al16(p.w_hi) || !al16(p.y) || !al16(p.sq_ref)) return l2i_set_error(L2I_E_ARG, "conv_img_h8: weight planes, y and sq_ref must be 16-byte aligned");
    const bool gains_ok = p.out_gain > 0.f && (p.act != L2I_ACT_LRELU || (p.act_gain > 0.f && p.act_slope >= 0.f && p.act_slope <= 1.f));
    if (!gains_ok) return l2i_set_error(L2I_E_UNSUPPORTED, "conv_img_h8: positive gains, leaky slope in [0, 1]");
    if ((p.OH - 1) * p.stride + p.KH - p.pad_y > p.H + p.pad_y || (p.OW - 1) * p.stride + p.KW - p.pad_x > p.W + p.pad_x)
        return l2i_set_error(L2I_E_ARG, "conv_img_h8: output window reads past the padded input");
    hipStream_t st = (hipStream_t)stream;
    if (p.KH == 1 && p.stride == 1) return launch_img<F16, 1, 1>(p, st);
    if (p.KH == 3 && p.stride == 1) return launch_img<F16, 3, 1>(p, st);
    if (p.KH == 7 && p.stride == 2) return launch_img<F16, 7, 2>(p, st);
    return l2i_set_error(L2I_E_UNSUPPORTED, "conv_img_h8: built for 1x1 / stride 1, 3x3 / stride 1 and 7x7 / stride 2");
}

extern "C" int l2i_conv_img_h8(const l2i_conv_params* p, void* stream) { return conv_img_h8<false>(p, stream); }
extern "C" int l2i_conv_img_h8_f16(const l2i_conv_params* p, void* stream) { return conv_img_h8<true>(p, stream); }
