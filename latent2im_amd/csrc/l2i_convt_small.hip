// l2i_convt_small.hip — stride-2 transposed 7x7 (pad 3) convolution onto <= 4 output channels: the input-gradient of ResNet-50's stem
// conv1 (7x7 / stride 2 / pad 3, 3 -> 64; reference call site graphs/stylegan_v2_real/transform_base.py:396-403 -> torchvision resnet50
// conv1), the last step of the regressor loss' way back to the image.  With 3 output channels an MFMA tile is 90 % padding, so the four
// output parities used to run as four launches of the direct VALU kernel (l2i_conv.hip), each reading the whole 64-channel gradient and
// its ReLU mask: 4.3 GB for 1.07 GB of input at 1024^2 batch 8 (2.0 ms).  Here ONE launch reads the input once (0.82 ms):
//
//   y[b, co, 2t + py, 2s + px] = sum_ci sum_{ky in K(py)} sum_{kx in K(px)} x'[b, ci, t + d(py,ky), s + d(px,kx)] * w[co, ci, ky, kx]
//   K(0) = {1,3,5}, K(1) = {0,2,4,6}, d(p,k) = (p + 3 - k) / 2;   x' = x * (in_mask > 0 ? mask_pos : mask_neg)
//
// A thread owns two neighbouring input positions = a 2 x 4 patch of outputs x 3 channels (24 accumulators); per input channel it reads
// its 4 x 5 input window (20 LDS words) and the 49 taps (float4: 3 channels + pad) for 294 FMAs.  Block = 16 x 32 input
// positions (32 x 64 outputs); the (16+3) x 40 input tile of 8 channels is staged with aligned 16-byte loads (mask applied on the way),
// a handful of blocks per CU hide the staging latency.  The taps are wave-uniform: they come through the scalar cache into SGPRs
// (one SGPR operand per FMA), not through LDS.  VALU-bound: 147 FMA per input element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"
#include "l2i_epilogue.h"      // [r5] l2i_h8_lo / l2i_h8_hi: the 16-bit unpack of the in_h8 variant

namespace cts {
constexpr int K = 7, PAD = 3, KK = K * K;
constexpr int TH = 16, TW = 32;                    // input positions per block
constexpr int IH = TH + 3, IWV = (TW + 8) / 4;     // staged rows; float4 per staged row: columns s0 - 4 .. s0 + TW + 3
constexpr int PITCH = IWV * 4, CK = 8;
constexpr int NV = (CK * IH * IWV + 255) / 256;    // staging vectors per thread
constexpr int NS = (IH * PITCH + 255) / 256;       // [r5] in_h8: 16-byte pixel slots (8 channels = one chunk) per thread
}

// [r5] H8: x and in_mask are 16-bit h8 tensors [B, Cin/8, H, W, 8] (l2i_conv_params::in_h8): a chunk of CK = 8 channels is ONE 8-channel group, a
// staged element one 16-byte pixel slot (unpacked to the same fp32 [channel][row][column] LDS tile), half the bytes of the fp32 form and no cast pass
// between the 16-bit pool gradient and this kernel.
template <bool MASK, bool H8 = false>
__global__ __launch_bounds__(256) void convt7_small_kernel(const l2i_conv_params p, int tiles_x, int tiles_y) {
    using namespace cts;
    __shared__ __attribute__((aligned(16))) float tile[CK * IH * PITCH];
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid;
    const int t0 = ty * TH, s0 = tx * TW;
    const int u = threadIdx.x & 15, tr = threadIdx.x >> 4;           // column pair / row of this thread
    const size_t plane_x = (size_t)p.H * p.W;
    float acc[2][4][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int o = 0; o < 3; ++o) acc[a][c][o] = 0.f;

    // (measured, not kept: fetching the next chunk's vectors into registers during the FMAs — 48 more registers, one block per CU less: 0.82 -> 1.07 ms)
    for (int c0 = 0; c0 < p.Cin; c0 += CK) {
        __syncthreads();
        if constexpr (H8) {
            const bool f16 = p.in_h8 == 2;
            const size_t gbase = ((size_t)b * (p.Cin >> 3) + (c0 >> 3)) * plane_x;
            l2i_u32x4 q[NS], qm[NS];
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const int e = threadIdx.x + n * 256;
                const int r = e / PITCH, col = e - r * PITCH;
                const int gy = t0 - 1 + r, gx = s0 - 4 + col;
                q[n] = l2i_u32x4{0u, 0u, 0u, 0u};
                qm[n] = q[n];
                if (e < IH * PITCH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    const size_t slot = gbase + (size_t)gy * p.W + gx;
                    q[n] = reinterpret_cast<const l2i_u32x4*>(p.x)[slot];
                    if constexpr (MASK) qm[n] = reinterpret_cast<const l2i_u32x4*>(p.in_mask)[slot];
                }
            }
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const int e = threadIdx.x + n * 256;
                if (e < IH * PITCH) {
                    const unsigned u[4] = {q[n].x, q[n].y, q[n].z, q[n].w}, m[4] = {qm[n].x, qm[n].y, qm[n].z, qm[n].w};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        float a0 = l2i_h8_lo(u[h], f16), a1 = l2i_h8_hi(u[h], f16);
                        if constexpr (MASK) {
                            a0 *= l2i_h8_lo(m[h], f16) > 0.f ? p.mask_pos : p.mask_neg;
                            a1 *= l2i_h8_hi(m[h], f16) > 0.f ? p.mask_pos : p.mask_neg;
                        }
                        tile[(2 * h) * IH * PITCH + e] = a0;
                        tile[(2 * h + 1) * IH * PITCH + e] = a1;
                    }
                }
            }
        } else {
        float4 v[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int e = threadIdx.x + n * 256;
            const int c = e / (IH * IWV), rem = e - c * (IH * IWV);
            const int r = rem / IWV, q = rem - r * IWV;
            const int gy = t0 - 1 + r, gx = s0 - 4 + 4 * q;
            v[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < CK * IH * IWV && c0 + c < p.Cin && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {       // W % 4 == 0: a vector is inside or outside
                const size_t off = ((size_t)b * p.Cin + c0 + c) * plane_x + (size_t)gy * p.W + gx;
                v[n] = *reinterpret_cast<const float4*>(p.x + off);
                if constexpr (MASK) {
                    const float4 m = *reinterpret_cast<const float4*>(p.in_mask + off);
                    v[n].x *= m.x > 0.f ? p.mask_pos : p.mask_neg; v[n].y *= m.y > 0.f ? p.mask_pos : p.mask_neg;
                    v[n].z *= m.z > 0.f ? p.mask_pos : p.mask_neg; v[n].w *= m.w > 0.f ? p.mask_pos : p.mask_neg;
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int e = threadIdx.x + n * 256;
            if (e < CK * IH * IWV) *reinterpret_cast<float4*>(&tile[e * 4]) = v[n];          // e * 4 == (c * IH + r) * PITCH + 4 q
        }
        }
        __syncthreads();
        const int cn = p.Cin - c0 < CK ? p.Cin - c0 : CK;
        for (int c = 0; c < cn; ++c) {
            // window rows t - 1 .. t + 2, columns s - 1 .. s + 3 (s = s0 + 2u): staged column index = 4 + 2u - 1 + j
            const float* tc = tile + (c * IH + tr) * PITCH + 3 + 2 * u;
            float win[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) win[r][j] = tc[r * PITCH + j];
            // the taps are the same for every lane: scalar loads (constant address space -> s_load into SGPRs), one kernel row at a time;
            // as LDS broadcasts they cost 8 LDS cycles per 6 FMAs and bound the kernel (measured: 1.9 ms, no faster than four launches)
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(4))) const f32x4v cfloat4;
            cfloat4* wc = (cfloat4*)(uintptr_t)(p.w + (size_t)(c0 + c) * KK * 4);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int py = (ky + 1) & 1;                           // output-row parity this tap lands on: (py + 3 - ky) even
                const int ry = (py + PAD - ky) / 2 + 1;                // window row
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int px = (kx + 1) & 1;
                    const int rx = (px + PAD - kx) / 2 + 1;            // window column of the first position
                    const f32x4v w4 = wc[ky * K + kx];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float x = win[ry][rx + q];
                        acc[py][2 * q + px][0] += x * w4.x; acc[py][2 * q + px][1] += x * w4.y; acc[py][2 * q + px][2] += x * w4.z;
                    }
                }
            }
        }
    }
    const size_t plane_o = (size_t)p.OHf * p.OWf;
    const int t = t0 + tr, ox = 2 * (s0 + 2 * u);
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const int oy = 2 * t + py;
        if (oy >= p.OHf || ox >= p.OWf) continue;                      // OWf % 4 == 0: the four columns are inside together
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (o < p.Cout) {
                float* yp = p.y + ((size_t)b * p.Cout + o) * plane_o + (size_t)oy * p.OWf + ox;
                float4 r = make_float4(acc[py][0][o] * p.out_gain, acc[py][1][o] * p.out_gain, acc[py][2][o] * p.out_gain, acc[py][3][o] * p.out_gain);
                if (p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(yp); r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
                *reinterpret_cast<float4*>(yp) = r;
            }
        }
    }
}

// eligibility of l2i_conv_transpose2d_f32 launches for this kernel: 7x7 / pad 3 onto <= 3 channels with the [Cin][49][4] weight pack
// (CoutP == 4), whole 16-byte rows on both sides, no style / output scale
bool l2i_convt_small_eligible(const l2i_conv_params& p) {
    auto al16 = [](const void* q) { return (((uintptr_t)q) % 16) == 0; };
    return p.KH == 7 && p.KW == 7 && p.pad_y == 3 && p.pad_x == 3 && p.Cout <= 3 && p.CoutP == 4 && !p.in_scale && !p.out_scale && ((p.W % 4) == 0 || p.in_h8) &&
           (p.OWf % 4) == 0 && al16(p.x) && al16(p.in_mask) && al16(p.y) && al16(p.w) && p.ksplit <= 1 && (!p.in_h8 || ((p.in_h8 == 1 || p.in_h8 == 2) && (p.Cin % 8) == 0));
}

int l2i_launch_convt_small(const l2i_conv_params& p, hipStream_t st) {
    using namespace cts;
    // every output position of the window is produced: positions (t, s) up to ceil(OHf / 2), ceil(OWf / 2)
    const int th = (p.OHf + 1) / 2, tw = (p.OWf + 1) / 2;
    const int tiles_x = (tw + TW - 1) / TW, tiles_y = (th + TH - 1) / TH;
    const long grid = (long)p.B * tiles_x * tiles_y;
    if (grid <= 0 || grid > 0x7fffffffL) return l2i_set_error(L2I_E_ARG, "conv_transpose2d(small): grid too large");
    if (p.in_h8 && p.in_mask) hipLaunchKernelGGL((convt7_small_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, st, p, tiles_x, tiles_y);
    else if (p.in_h8) hipLaunchKernelGGL((convt7_small_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, st, p, tiles_x, tiles_y);
    else if (p.in_mask) hipLaunchKernelGGL((convt7_small_kernel<true>), dim3((unsigned)grid), dim3(256), 0, st, p, tiles_x, tiles_y);
    else hipLaunchKernelGGL((convt7_small_kernel<false>), dim3((unsigned)grid), dim3(256), 0, st, p, tiles_x, tiles_y);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
