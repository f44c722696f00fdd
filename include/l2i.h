/*
 * l2i.h — C ABI of libl2i_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the Latent2im
 * walk-training hot path.
 *
 * Drop-in boundary.  The reference's only native code is two pybind/CUDA extensions that are JIT-built at
 * import (graphs/stylegan_v2_real/op/fused_act.py:10-16, op/upfirdn2d.py:9-15).  The entry points below are
 * what an FFI for this path binds instead:
 *
 *   l2i_fused_bias_act_f32   replaces  fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *                                      op/fused_bias_act.cpp:11-21, op/fused_bias_act_kernel.cu:18-98
 *   l2i_upfirdn2d_f32        replaces  upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y,
 *                                      pad_x0, pad_x1, pad_y0, pad_y1)
 *                                      op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:140-272
 *   l2i_conv2d_f32           replaces  the F.conv2d(groups=batch) / F.conv_transpose2d(groups=batch) /
 *                                      weight-materialisation sequence of ModulatedConv2d.forward
 *                                      (networks.py:231-272), EqualConv2d.forward (networks.py:111-120) and the
 *                                      torchvision ResNet-50 / VGG-19 conv(+BN eval)+ReLU blocks called at
 *                                      transform_base.py:396-403,416-454, and their input-gradients
 *   l2i_torgb_* / l2i_sg2_act_bwd_f32 / l2i_dot_reduce_f32 / l2i_maxpool2d_* / l2i_sqdiff_f32
 *                            the streaming (HBM-bound) companions of the above on the same path:
 *                            ToRGB (networks.py:339-358), lrelu backward + style-gradient reductions,
 *                            MaxPool2d of ResNet/VGG, ContentLoss (transform_base.py:57-63).
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous float32 (NCHW for maps); the caller owns all
 * buffers (nothing is allocated here); `stream` is a hipStream_t passed as void*; kernels are enqueued
 * asynchronously; return value 0 = enqueued, negative = L2I_E_* (nothing enqueued).  Thread-safe for distinct
 * streams.  No torch types anywhere in this ABI.
 */
#ifndef L2I_H
#define L2I_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2I_OK 0
#define L2I_E_ARG (-1)      /* invalid argument combination */
#define L2I_E_LAUNCH (-2)   /* hipLaunch failed (see l2i_last_error) */
#define L2I_E_UNSUPPORTED (-3)

#define L2I_ACT_NONE 0
#define L2I_ACT_LRELU 1     /* y = (v > 0 ? v : v*slope) * gain */
#define L2I_ACT_RELU 2

/* Generic 2-D correlation as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
 *   y[b,co,oy*oy_step+oy_off,ox*ox_step+ox_off] = epi( sum_{ci,ky,kx} pro(x)[b,ci,oy*stride-pad_y+ky,ox*stride-pad_x+kx]
 *                                                       * w[ci][ky*KW+kx][co] )     for oy<OH, ox<OW
 *   pro(x) = x * in_scale[b,ci] * (in_mask ? (in_mask[same idx] > 0 ? mask_pos : mask_neg) : 1)
 *   epi(a) = act( a*out_scale[b,co] * (out_mask ? (out_mask[idx] > 0 ? 1 : 0) : 1) + noise[b,oyf,oxf]*noise_w + bias[co]
 *                 + R * (res_mask ? (res_mask[idx] > 0 ? 1 : 0) : 1) ) * out_gain  (+ y[idx] if accumulate)
 *   R = res_sub ? res_coef * res_coef_dev[0] * (residual[idx] - res_sub[idx]) : residual[idx]
 * Out-of-range input coordinates read as zero (that is the padding).  Transposed (stride-2) convolutions are issued
 * as one call per output phase with (oy_step, ox_step) = 2 and a per-phase packed sub-kernel. */
typedef struct l2i_conv_params {
    const float* x;         /* [B, Cin, H, W] */
    const float* w;         /* packed [Cin][KH*KW][CoutP], CoutP = Cout rounded up to 32, zero padded */
    float* y;               /* [B, Cout, OHf, OWf] */
    int32_t B, Cin, H, W, Cout, CoutP;
    int32_t KH, KW, stride, pad_y, pad_x;
    int32_t OH, OW, OHf, OWf;
    int32_t oy_step, ox_step, oy_off, ox_off;
    const float* in_scale;  /* [B, Cin] or NULL */
    const float* in_mask;   /* like x, or NULL */
    float mask_pos, mask_neg;
    const float* out_scale; /* [B, Cout] or NULL */
    const float* noise;     /* [B, 1, OHf, OWf] or NULL */
    float noise_w;
    const float* bias;      /* [Cout] or NULL */
    const float* residual;  /* like y, or NULL */
    const float* res_mask;  /* like y, or NULL */
    const float* out_mask;  /* like y, or NULL */
    int32_t act;
    float act_slope, act_gain, out_gain;
    int32_t accumulate;
    int32_t tile_hint;      /* 0 = auto, else 1..N selects a tile configuration (tuning / tests) */
    const void* w_hi;       /* l2i_conv2d_bf16x3_f32 only: bf16 planes [Cin/16][KH*KW][CoutP][2][8], hi = bf16(w), */
    const void* w_lo;       /*                              lo = bf16(w - hi); NULL for the fp32 entry points */
    float* ws;              /* l2i_conv2d_f32 split-K workspace: ksplit * B*Cout*OHf*OWf floats, or NULL */
    int32_t ksplit;         /* > 1: the input channels are cut into ksplit ranges computed by separate blocks (raw partial sums in ws),
                               then one reduction pass applies the epilogue: for 4x4..16x16 maps, whose whole K = Cin*KH*KW would
                               otherwise be walked serially by a dozen blocks.  0 / 1: off. */
    const float* res_sub;   /* like y, or NULL.  Not NULL: the residual term becomes res_coef * res_coef_dev[0] * (residual - res_sub) — the */
    float res_coef;         /* ContentLoss gradient 2/N * g * (feat - feat_org) of a VGG tap (transform_base.py:57-63) formed inside the     */
    const float* res_coef_dev;  /* gradient conv that consumes it, instead of a separate pass.  res_coef_dev: 1 float on the device or NULL */
    const float* sq_ref;    /* like y, or NULL.  Not NULL (l2i_conv2d_wino_f32, l2i_conv2d_bf16x3_f32 with dense 16-byte output rows, and launches */
                            /* of l2i_conv2d_f32 that take the <= 3-input-channel kernel, L2I_FAMILY_CIN3; everything else refuses it): the launch */
                            /* also adds                                                                                                          */
    float* sq_out;          /* sum (y - sq_ref)^2 over its outputs into sq_out[0 .. L2I_SQ_SLOTS-1] (fp32 atomics, one slot per block id mod   */
                            /* L2I_SQ_SLOTS; the caller zeroes the slots and sums them): the ContentLoss value of a VGG tap                  */
                            /* (transform_base.py:57-63, mse = the sum / N) without re-reading the feature map in a separate pass             */
    int64_t w_bstride;      /* l2i_conv2d_h8 / l2i_conv_transpose2d_h8: bytes between the weight planes of consecutive samples (the generator's */
                            /* modulated convs: style and demodulation folded into one plane set per sample), 0 = one set for every sample    */
    int32_t out_f32;        /* l2i_conv2d_h8: 1 = the output (and residual / res_mask / out_mask / res_sub) is fp32 NCHW instead of bf16 h8      */
    /* ---- ABI version 5: fp32 <-> 16-bit boundaries of the 16-bit path without a cast pass (see also l2i_conv_img_h8) ------------------------- */
    int32_t in_h8;          /* l2i_conv_transpose2d_f32, 7x7 / pad 3 onto <= 3 channels (the ResNet stem's input gradient): 1 / 2 = x and in_mask are      */
                            /* bf16 / fp16 h8 tensors [B, Cin/8, H, W, 8] (Cin % 8 == 0)                                                                  */
    const float* rgb_w;     /* l2i_conv2d_h8 with the h8 output and every output channel in one block (Cout = 32 or 64): not NULL = the launch also writes the   */
    const float* rgb_bias;  /* ToRGB image of its output, rgb_out[b, o, oy, ox] = rgb_bias[o] + sum_c rgb_w[b, o, c] * epi(.)[b, c, oy, ox] (o < 3; fp32     */
    float* rgb_out;         /* NCHW [B, 3, OHf, OWf]; networks.py:349-358 on the 512^2 / 1024^2 StyledConv outputs), instead of a pass that re-reads y        */
    float* pool_out;        /* l2i_conv2d_wino4_f32 (position-split kernel, dense output with even OHf, OWf % 4 == 0, zero output offsets): not NULL = the launch    */
    uint8_t* pool_idx;      /* also writes MaxPool2d(2, 2) of its output y, pool_out [B, Cout, OHf/2, OWf/2] fp32 and the window-local arg-max pool_idx (same     */
                            /* shape, bytes; taps in (ky, kx) order, first maximum, NaN propagates: l2i_maxpool2d_fwd_f32's rule) — VGG-19's pool after conv1_2   */
                            /* (transform_base.py:426-454) from the 4x4 output tile a lane holds anyway, instead of a pass that reads the 2 GB map again          */
    /* ---- ABI version 6: one-bit activation masks on the 16-bit path (l2i_conv2d_h8 / l2i_conv_transpose2d_h8) ------------------------------------------- */
    uint8_t* mask_out;      /* h8 output: not NULL = the launch also writes the SIGN PLANE of its output, one byte per 16-byte pixel slot:                              */
                            /* mask_out[((b * Cout/8 + g) * OHf + oy) * OWf + ox] bit e = (the stored 16-bit y[b, 8 g + e, oy, ox] > 0).  The input-gradient launches of   */
                            /* a frozen ReLU network need exactly this bit of every saved activation (torchvision's bottleneck relu(bn(conv)) called at                   */
                            /* transform_base.py:396-403,416-424), and read it at 1/16 of the map's bytes                                                                 */
    int32_t mask_bits;      /* 1 = out_mask / res_mask point to such sign planes ([B, Cout/8, OHf, OWf] bytes) instead of h8 maps shaped like y                           */
} l2i_conv_params;
#define L2I_SQ_SLOTS 1024

int l2i_conv2d_f32(const l2i_conv_params* p, void* stream);

/* Which kernel family l2i_conv2d_f32 runs `p` on (no launch; pure function of the parameters): the measurement code uses it to
 * price every launch against the roofline of the kernel that actually ran. */
#define L2I_FAMILY_IMPLICIT_GEMM 0  /* conv_mfma_kernel: LDS-staged halo tile, v_mfma_f32_32x32x2_f32 */
#define L2I_FAMILY_GEMM1X1 1        /* gemm1x1_kernel: DMA-fed plain GEMM of unscaled 1x1 stride-1 layers */
#define L2I_FAMILY_CIN3 2           /* conv_cin3_kernel: 3x3 convs of <= 3-channel images */
#define L2I_FAMILY_DIRECT_SMALL 3   /* conv_direct_small_kernel: <= 4 output channels, VALU */
int l2i_conv2d_family(const l2i_conv_params* p);

/* Stride-2 TRANSPOSED convolution, all four output parities in one launch:
 *   y[b,co,2*iy+ky-pad,2*ix+kx-pad] += x[b,ci,iy,ix] * w[co,ci,ky,kx]          (F.conv_transpose2d(stride=2) of the up
 *   layers, networks.py:246-255, and the input-gradient of every stride-2 conv of the path).
 * Same struct as l2i_conv2d_f32 with KH = KW = K, pad_y = pad_x = pad, y = [B,Cout,OHf,OWf] of the natural size
 * (H-1)*2-2*pad+K (or up to 8 more: rows / columns no input reaches are written as zeros); `w` is the FUSED pack [Cin][K*K][CoutP] whose tap order is the kernel's walk order
 * (latent2im_amd/conv.py:fused_transposed_taps).  Fused: in_scale, in_mask, out_scale, out_gain; everything else must
 * be unset.  Built for (K,pad) in {(3,0),(3,1),(7,3)}; other shapes return L2I_E_UNSUPPORTED (use the per-parity
 * l2i_conv2d_f32 calls). */
int l2i_conv_transpose2d_f32(const l2i_conv_params* p, void* stream);

/* Split-precision variant of l2i_conv2d_f32 (BASELINE config 5's "16-bit MFMA" path; latent2im_amd.conv.PRECISION = 'bf16x3' selects
 * it): operands are split x = bf16(x) + bf16(x - bf16(x)) and a*b is evaluated as ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16
 * with fp32 accumulation (relative product error <= ~2^-17: fp32-class results).  fp32 tensors in and out, same prologue / epilogue
 * fusions; `w` is ignored, `w_hi` / `w_lo` are the host-split weight planes [Cin/16][KH*KW][2][CoutP][8] (hi = bf16(w), lo = bf16(w - hi)).
 * Layers: 1x1 (pad 0, Cin % 32 == 0) and 3x3 (pad 1; stride 2 also pad 0; Cin % 16 == 0), stride 1 or 2, dense output window, OW >= 32,
 * W % 4 == 0, 16-byte aligned x / in_mask.  Anything else returns L2I_E_UNSUPPORTED (callers use l2i_conv2d_f32). */
int l2i_conv2d_bf16x3_f32(const l2i_conv_params* p, void* stream);

/* l2i_conv_transpose2d_f32 on the split-precision bf16 matrix path (same arithmetic as l2i_conv2d_bf16x3_f32): 3x3 stride-2 transposed
 * conv, pad 0 (the generator's up layers, networks.py:246-255) or pad 1 (input-gradient of a 3x3 stride-2 pad-1 conv), all four output
 * parities per launch.  `w_hi` / `w_lo`: the bf16 planes of the [Cout, Cin, 3, 3] correlation-form weight in the layout of
 * l2i_conv2d_bf16x3_f32 ([Cin/16][9][2][CoutP][8]); `w` is ignored.  Fused: in_scale, in_mask, out_scale, out_gain.  Needs Cin % 16 == 0,
 * W % 4 == 0, W >= 32 and the natural output size; other shapes return L2I_E_UNSUPPORTED (use l2i_conv_transpose2d_f32). */
int l2i_conv_transpose2d_bf16x3_f32(const l2i_conv_params* p, void* stream);

/* 3x3 stride-1 layers (KH = KW = 3, stride 1, dense output window, Cin % 8 == 0, output rows 16-byte aligned multiples of
 * 4 pixels) as Winograd F(2x2,3x3) on the fp32 matrix cores: exact-fp32 products and accumulation like l2i_conv2d_f32,
 * 2.25x fewer of them (the algorithm vendor libraries select for the reference's F.conv2d 3x3 calls too).  Same struct,
 * same prologue / epilogue fusions; `w` is the TRANSFORMED pack U = G g G^T laid out [Cin][4][CoutP][4]
 * (latent2im_amd/conv.py:pack_weight_wino).  tile_hint must be 0.
 * Shapes outside the constraints return L2I_E_UNSUPPORTED (callers use l2i_conv2d_f32). */
int l2i_conv2d_wino_f32(const l2i_conv_params* p, void* stream);

/* The same layers as Winograd F(4x4,3x3): 36 multiplies per 4x4 output tile and (cin, cout) — 1.78x fewer than F(2x2,3x3), 4x fewer than the
 * direct form; fp32 products and accumulation, error ~1e-6..1e-5 of max|y| against the exact correlation (F(2x2): 3e-7).  Unmasked launches
 * only: `in_mask` must be NULL or the input itself with ReLU slopes (1, 0) (ReLU-on-load); Cin % 4 == 0.  `w` is the transformed pack
 * U = G g G^T (6x6 per (cin, cout)) in the kernel's LDS image order [Cin/4][CoutP/16][ [6 i][4 cin][16 cout][4 j=0..3] ++ [6][4][16][2 j=4,5] ]
 * (latent2im_amd/conv.py:pack_weight_wino4); CoutP = Cout rounded up to 32.  Same epilogue fusions as l2i_conv2d_wino_f32 (incl. sq_ref).
 * tile_hint 0: the position-split kernel (round 5: a block owns 32 output channels, two waves share the 36 Winograd positions of a tile row;
 * needs pad_x == 1 and W % 4 == 0); tile_hint 1: the round-4 kernel (16 channels per block), same arithmetic, bit-identical results;
 * tile_hint 2: the position-split kernel on 64 x 16-pixel tiles / eight waves where the map has >= 64 columns and >= 16 rows (an A/B form: slower). */
int l2i_conv2d_wino4_f32(const l2i_conv_params* p, void* stream);

/* ---- the 16-bit path (BASELINE config 5: "fp16 MFMA"; bf16 here: fp32's exponent range, so gradients of 1e-9 need no loss scaling) ----
 * Tensors in the channel-blocked "h8" layout [B][C/8][H][W][8] bf16 (the 8 channels of a pixel = 16 contiguous bytes = one MFMA fragment);
 * one v_mfma_f32_32x32x16_bf16 product per MAC, fp32 accumulation; fp32 only for bias, noise, per-sample scales and the loss sums.
 * l2i_conv2d_h8: 1x1 (pad 0) / 3x3 (pad 0 or 1) correlation, stride 1 or 2, Cin % 32 == 0.  Same struct as l2i_conv2d_f32 with
 *   x, y, residual, res_mask, out_mask, res_sub, sq_ref -> bf16 h8 tensors (fp32 NCHW for y and the epilogue operands when out_f32 = 1);
 *   w_hi -> bf16 weight planes [Cin/16][KH*KW][2][CoutP][8] (latent2im_amd/conv.py:pack_weight_bf16x3, hi plane), per sample when w_bstride != 0;
 *   noise [B,1,OHf,OWf], bias [Cout], out_scale [B,Cout] fp32; `w`, `w_lo` ignored; no prologue fusions (scales live in the weights, masks in
 *   the producing epilogue): in_scale must be NULL, in_mask NULL or == x with mask (1, 0) (ReLU-on-load, 3x3 stride-1 layers); the output
 *   mask is leaky here: * (out_mask > 0 ? mask_pos : mask_neg).  h8 output needs Cout % 8 == 0.
 * l2i_conv_transpose2d_h8: stride-2 transposed 3x3 conv (pad 0 / 1), all four output parities per launch, h8 in and out. */
int l2i_conv2d_h8(const l2i_conv_params* p, void* stream);
int l2i_conv_transpose2d_h8(const l2i_conv_params* p, void* stream);

/* [ABI 7] Two chained 1x1 stride-1 convolutions on h8 maps in one launch (csrc/l2i_pair_h8.hip): `first` (Cin1 -> Cout1, with its bias / residual /
 * ReLU / sign-plane masks / sign-plane output, y = the wide map, written) feeds `second` (Cout1 -> Cout2 <= 256, bias / ReLU / sign-plane mask / sign-plane
 * output) from registers: the wide map is not read back.  ResNet-50's trunk (torchvision Bottleneck as called at transform_base.py:396-403, 416-424):
 * forward = conv3 + identity + ReLU of a block, then conv1 + ReLU of the next; backward = conv1's input gradient + trunk gradient, masked, then conv3's
 * input gradient of the block below.  Each struct is filled exactly as for l2i_conv2d_h8 (second->x may be NULL or first->y); results are bit-identical
 * to the two l2i_conv2d_h8 launches.  Refused (L2I_E_UNSUPPORTED) unless: both 1x1 / stride 1 / pad 0 / h8 output, first->residual set, masks given as sign
 * planes (mask_bits), H * W a multiple of 128, (Cin1, Cout2) one of (64, 64), (128, 128), (256, 256), (64, 128), (128, 256) — the caller then launches the two convs separately.
 * variant: 0 = 256-pixel tiles, 1 = 128-pixel tiles (more blocks on small maps). */
int l2i_conv1x1_pair_h8(const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream);
/* The same launch with the 3x3 stride-1 pad-1 conv in FRONT of the pair: head3x3 (C -> C, C <= 128 here: bias / ReLU / sign-plane mask / sign-plane output; its
 * map is written only if head3x3->y is set) -> first -> second: one launch per ResNet-50 bottleneck (conv2, conv3 + identity, the next block's conv1; backwards:
 * conv2's input gradient, conv1's + trunk gradient, conv3's of the block below).  The 3x3 runs exactly as l2i_conv2d_h8 would (16-channel chunks), so the
 * three results equal the three launches' bit for bit.  Needs W % 32 == 0, H % 4 == 0 and (C, Cout2) one of (64, 64), (128, 128), (64, 128). */
int l2i_conv_chain3_h8(const l2i_conv_params* head3x3, const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream);

/* [ABI 7] The fp32 twin of l2i_conv1x1_pair_h8 (csrc/l2i_pair_f32.hip): `first` = a 1x1 stride-1 conv of l2i_conv2d_f32 with bias / residual / ReLU (its y, the wide
 * map, is written), `second` = the 1x1 conv that reads it (bias / ReLU), one launch, the wide map not read back.  Both structs filled as for l2i_conv2d_f32
 * (w = the [Cin][1][CoutP] pack).  The second conv adds its channel pairs in another order than l2i_conv2d_f32: equal to fp32 rounding, not bit for bit.
 * Refused (L2I_E_UNSUPPORTED; the caller launches the two convs) unless H * W % 256 == 0 and (Cin1, Cout2) is (64, 64), (64, 128) or (128, 128). */
int l2i_conv1x1_pair_f32(const l2i_conv_params* first, const l2i_conv_params* second, void* stream);

/* [r5] The image-side convolutions of the 16-bit path (csrc/l2i_img_h8.hip): x = fp32 NCHW image [B, Cin <= 4, H, W], y = 16-bit h8
 * [B, Cout/8, OHf, OWf, 8], 16-bit MFMA with fp32 accumulation on operands rounded to the element type inside the kernel — VGG-19 conv1_1
 * (3x3 / stride 1; transform_base.py:426-454), the discriminator's from-RGB 1x1 (networks.py:568-575), ResNet-50's 7x7 / stride 2 stem
 * (transform_base.py:396-403) without a zero-padded 16-bit copy of the image, an fp32 stem map or a cast pass.  Built for (KH, stride) in
 * {(1, 1), (3, 1), (7, 2)}, any symmetric padding.  `w_hi`: 16-bit planes [ceil(Cin KH / 2)][2][CoutP][8] whose element (s, half, co, e) is
 * w[co, c, ky, kx = e] for (c, ky) = divmod(2 s + half, KH), zero for e >= KW or 2 s + half >= Cin KH (latent2im_amd/conv.py:
 * pack_weight_img_h8).  Fused: bias, act, out_gain (> 0), sq_ref / sq_out (sq_ref h8 like y).  Everything else must be unset. */
int l2i_conv_img_h8(const l2i_conv_params* p, void* stream);

/* Streaming companions on h8 maps (csrc/l2i_stream_h8.hip; same functions as the fp32 entry points below, arithmetic in fp32 registers):
 * layout casts fp32 NCHW <-> bf16 h8 (Cpad = channel count of the h8 tensor, a multiple of 8, zero filled above C);
 * the reference's upfirdn2d on h8 planes (kernels up to 4x4, up / down in {1, 2}) with the generator's fused epilogue
 *   y = act(fir(x) + noise[b,oy,ox] * noise_w + bias[c]) * act_gain;
 * ToRGB (x h8 -> rgb fp32 [B,3,HW]); the fused activation backward of a styled conv (dz h8; gin h8; grgb fp32; reductions fp32, zeroed by the caller);
 * out[b,c] += sum_p a * b; MaxPool2d forward (idx: [planes][OH][OW][8] bytes; relu = 1: y = max(pool, 0)) and backward (optionally + coef *
 *   coef_dev[0] * (b - a): the ContentLoss term of the pooled tap); the ContentLoss difference; y[2oy, 2ox] += c[oy, ox]; and the per-sample
 *   weight planes of a modulated conv: planes[b] = bf16(w32 * s[b, input channel]) with w32 = the fp32 weights in plane order. */
int l2i_cast_f32_to_h8(void* y, const float* x, int B, int C, int Cpad, int64_t HW, void* stream);
int l2i_cast_h8_to_f32(float* y, const void* x, int B, int C, int Cpad, int64_t HW, void* stream);
int l2i_upfirdn2d_h8(void* y, const void* x, const float* k, int64_t planes, int channels, int in_h, int in_w, int kh, int kw, int up, int down,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise, float noise_w, const float* bias, int act, float act_slope,
                     float act_gain, const void* mask, float mask_pos, float mask_neg, const void* addend, const float* k1y, const float* k1x, int mask_bits,
                     void* stream);
                     /* ... then * (mask > 0 ? mask_pos : mask_neg) (h8, like y: a gradient through a (leaky) ReLU) and + addend (h8, like y).
                        k1y / k1x: HOST pointers to four floats each, or NULL: the caller states that k = outer(k1y, k1x) (4x4, up = down = 1: the
                        register-streaming separable kernel).  [ABI 6] mask_bits = 1: `mask` is the SIGN PLANE of that map (one byte per pixel slot,
                        l2i_conv_params::mask_out) — separable 4x4 blur without resampling only */
int l2i_torgb_fwd_h8(float* rgb, const void* x, const float* wmod, const float* bias, int B, int C, int64_t HW, void* stream);
int l2i_sg2_act_bwd_h8(void* dz, const void* gin, const float* gin_scale, const float* grgb, const float* wmod_rgb, const void* y, const float* bias,
                       const float* noise, float noise_w, float slope, float gain, float* red_dz_z, float* red_x_grgb, float* red_gin_y, int B, int C, int64_t HW, void* stream);
int l2i_dot_reduce_h8(float* out, const void* a, const void* b, int B, int C, int64_t HW, void* stream);
int l2i_maxpool2d_fwd_h8(void* y, void* idx, const void* x, int64_t planes, int H, int W, int k, int s, int pad, int OH, int OW, int relu, void* stream);
int l2i_maxpool2d_bwd_h8(void* gx, const void* gy, const void* idx, const void* a, const void* b, float coef, const float* coef_dev, int64_t planes, int H, int W,
                         int k, int s, int pad, int OH, int OW, void* stream);
int l2i_sqdiff_h8(float* sum_out, void* grad, const void* a, const void* b, int64_t slots, float coef, const float* coef_dev, void* stream);
int l2i_add_zero_insert_h8(void* y, const void* c, const void* mask, int64_t planes, int H, int W, int OH, int OW, void* stream);   /* mask (h8 like y, or NULL): c * (mask[2oy,2ox] > 0) */
int l2i_mask_mul_h8(void* y, const void* g, const void* ref, float pos, float neg, int64_t slots, void* stream);                     /* y = g * (ref > 0 ? pos : neg) */
int l2i_mask_mul_bits_h8(void* y, const void* g, const void* bits, float pos, float neg, int64_t slots, void* stream);               /* [ABI 6] the same with ref's sign plane (one byte per slot) */
int l2i_modulate_planes_h8(void* planes, const float* w32, const float* s, int B, int Cs, int CinP, int KK, int CoutP, void* stream);
/* [r5] l2i_modulate_planes_h8 for every modulated conv of a generator pass in one launch.  `table` (device): nseg rows of eight int64 — w32 offset
 * (floats from `w32`), scale offset (floats from `s`: the layer's [B, Cs] block), output offset (16-byte slots from `planes`: the layer's
 * [B][slots per sample] block), slots per sample (= CinP/16 * KK * 2 * CoutP), KK, CoutP, Cs (= CinP), first block of the segment (ascending,
 * row 0 = 0); `nblocks` = grid size (blocks of a segment = next row's first block - its own).  networks.py:234-235 per layer. */
int l2i_modulate_planes_multi_h8(void* planes, const float* w32, const float* s, const void* table, int nseg, int B, int nblocks, void* stream);

/* [r5] The same sixteen entry points with IEEE fp16 (binary16) elements instead of bf16 — BASELINE configs[4] names "fp16 MFMA"; what the reference
 * would run under autocast on networks.py:231-272.  Identical signatures, layouts and fusions; the contraction is v_mfma_f32_32x32x16_f16 (same rate
 * as the bf16 instruction), conversions round to nearest even.  fp16 has 3 more mantissa bits than bf16 and 3 fewer exponent bits: the caller keeps
 * gradients inside its range with power-of-two loss scales (latent2im_amd/nets16.py: one per loss branch, undone on the fp32 side: exact). */
int l2i_conv2d_h8_f16(const l2i_conv_params* p, void* stream);
int l2i_conv_transpose2d_h8_f16(const l2i_conv_params* p, void* stream);
int l2i_conv1x1_pair_h8_f16(const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream);
int l2i_conv_chain3_h8_f16(const l2i_conv_params* head3x3, const l2i_conv_params* first, const l2i_conv_params* second, int variant, void* stream);
int l2i_conv_img_h8_f16(const l2i_conv_params* p, void* stream);
int l2i_cast_f32_to_h8_f16(void* y, const float* x, int B, int C, int Cpad, int64_t HW, void* stream);
int l2i_cast_h8_to_f32_f16(float* y, const void* x, int B, int C, int Cpad, int64_t HW, void* stream);
int l2i_upfirdn2d_h8_f16(void* y, const void* x, const float* k, int64_t planes, int channels, int in_h, int in_w, int kh, int kw, int up, int down,
                         int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise, float noise_w, const float* bias, int act, float act_slope,
                         float act_gain, const void* mask, float mask_pos, float mask_neg, const void* addend, const float* k1y, const float* k1x, int mask_bits,
                         void* stream);
int l2i_torgb_fwd_h8_f16(float* rgb, const void* x, const float* wmod, const float* bias, int B, int C, int64_t HW, void* stream);
int l2i_sg2_act_bwd_h8_f16(void* dz, const void* gin, const float* gin_scale, const float* grgb, const float* wmod_rgb, const void* y, const float* bias,
                           const float* noise, float noise_w, float slope, float gain, float* red_dz_z, float* red_x_grgb, float* red_gin_y, int B, int C, int64_t HW, void* stream);
int l2i_dot_reduce_h8_f16(float* out, const void* a, const void* b, int B, int C, int64_t HW, void* stream);
int l2i_maxpool2d_fwd_h8_f16(void* y, void* idx, const void* x, int64_t planes, int H, int W, int k, int s, int pad, int OH, int OW, int relu, void* stream);
int l2i_maxpool2d_bwd_h8_f16(void* gx, const void* gy, const void* idx, const void* a, const void* b, float coef, const float* coef_dev, int64_t planes, int H, int W,
                             int k, int s, int pad, int OH, int OW, void* stream);
int l2i_sqdiff_h8_f16(float* sum_out, void* grad, const void* a, const void* b, int64_t slots, float coef, const float* coef_dev, void* stream);
int l2i_add_zero_insert_h8_f16(void* y, const void* c, const void* mask, int64_t planes, int H, int W, int OH, int OW, void* stream);
int l2i_mask_mul_h8_f16(void* y, const void* g, const void* ref, float pos, float neg, int64_t slots, void* stream);
int l2i_mask_mul_bits_h8_f16(void* y, const void* g, const void* bits, float pos, float neg, int64_t slots, void* stream);
int l2i_modulate_planes_h8_f16(void* planes, const float* w32, const float* s, int B, int Cs, int CinP, int KK, int CoutP, void* stream);
int l2i_modulate_planes_multi_h8_f16(void* planes, const float* w32, const float* s, const void* table, int nseg, int B, int nblocks, void* stream);

/* out[i] = act_grad_table(x[i] + b[(i / step_b) % size_b], ref[i]) * scale   — the reference op, all six
 * act*10+grad cases (fused_bias_act_kernel.cu:36-47).  b / ref may be NULL (= the reference's empty tensors). */
int l2i_fused_bias_act_f32(float* y, const float* x, const float* b, const float* ref, int64_t n,
                           int64_t step_b, int64_t size_b, int act, int grad, float alpha, float scale, void* stream);

/* The same op for half tensors: the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF (fused_bias_act_kernel.cu:79).  y / x / b / ref
 * point to IEEE binary16; arithmetic follows the reference's scalar_t = Half instantiation operation by operation (each binary op is
 * the float op rounded to half; alpha and scale are rounded to half first), so results are bit-identical to it. */
int l2i_fused_bias_act_f16(void* y, const void* x, const void* b, const void* ref, int64_t n, int64_t step_b, int64_t size_b,
                           int act, int grad, float alpha, float scale, void* stream);

/* The reference op on [major, in_h, in_w] maps (minor_dim == 1, the only layout the path uses:
 * op/upfirdn2d.py:98) with an optional fused epilogue (all NULL/0 = the plain reference op):
 *   y = act( fir(x) + noise[b,oy,ox]*noise_w + bias[c] + addend[idx] ) * act_gain,   major = b*channels + c */
int l2i_upfirdn2d_f32(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw,
                      int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                      int channels, const float* noise, float noise_w, const float* bias, const float* addend,
                      int act, float act_slope, float act_gain, void* stream);

/* [r5] The same op with one more fused term, y *= (mask[idx] > 0 ? mask_pos : mask_neg) after the activation: a gradient that passes a (leaky) ReLU on its
 * way out of the FIR (the discriminator's backward, networks.py:530-536 / 574-583: the 3x3 gradient conv that follows then needs no mask operand and
 * takes the Winograd F(4x4) kernel).  mask is shaped like y.  NULL mask = l2i_upfirdn2d_f32. */
int l2i_upfirdn2d_masked_f32(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw,
                             int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                             int channels, const float* noise, float noise_w, const float* bias, const float* addend,
                             int act, float act_slope, float act_gain, const float* mask, float mask_pos, float mask_neg, void* stream);

/* upfirdn2d for half tensors (upfirdn2d_kernel.cu:225 dispatches half too): plain reference op, no fused epilogue.  y / x / k point to IEEE
 * binary16; like the reference kernel the taps and products are float and the accumulator is rounded to half after every tap, taps in
 * ascending input row / column order (upfirdn2d_kernel.cu:118-123): bit-identical results. */
int l2i_upfirdn2d_f16(void* y, const void* x, const void* k, int64_t major, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                      int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* ToRGB 1x1 modulated conv without demodulation (networks.py:346-351):
 *   rgb[b,o,p] = sum_c x[b,c,p] * wmod[b,o,c] + bias[o],  wmod = scale*W[o,c]*s[b,c] prepared by the caller [B,3,C] */
int l2i_torgb_fwd_f32(float* rgb, const float* x, const float* wmod, const float* bias, int B, int C, int64_t HW,
                      void* stream);

/* Fused elementwise backward of one StyledConv output (FusedLeakyReLU + NoiseInjection + demod bookkeeping) with the
 * ToRGB branch folded in:
 *   g      = gin[idx]*gin_scale[b,c] + sum_o wmod_rgb[b,o,c]*grgb[b,o,p]          (either part optional)
 *   dz     = g * (y > 0 ? gain : gain*slope)                                        -> dz[idx]
 *   zpre   = (y > 0 ? y/gain : y/(gain*slope)) - bias[c] - noise[b,p]*noise_w
 *   red_dz_z[b,c] += sum_p dz*zpre            (-> d demod)        red_x_grgb[b,c,o] += sum_p y*grgb[b,o,p]  (-> d s_rgb)
 *   [r5] red_gin_y[b,c] += sum_p gin*y  (NULL: not formed) — y is the INPUT of the next layer and gin the gradient w.r.t. that layer's modulated input,
 *   so this is the next layer's style gradient d s (networks.py:234-235: x * s), formed while both maps pass through registers instead of by a
 *   l2i_dot_reduce pass that reads them again.
 * Reduction buffers must be zeroed by the caller. */
int l2i_sg2_act_bwd_f32(float* dz, const float* gin, const float* gin_scale, const float* grgb, const float* wmod_rgb,
                        const float* y, const float* bias, const float* noise, float noise_w, float slope, float gain,
                        float* red_dz_z, float* red_x_grgb, float* red_gin_y, int B, int C, int64_t HW, void* stream);

/* out[r] (+)= sum_p a[r,p] * (b ? b[r,p] : 1), r < rows (rows = B*C).  `out` must be zeroed by the caller. */
int l2i_dot_reduce_f32(float* out, const float* a, const float* b, int64_t rows, int64_t cols, void* stream);

/* MaxPool2d(k, s, pad) on [N, H, W] planes; idx holds the window-local argmax (first maximum in row-major order, like
 * ATen) so that the backward is exact under ties. */
int l2i_maxpool2d_fwd_f32(float* y, uint8_t* idx, const float* x, int64_t planes, int H, int W, int k, int s, int pad,
                          int OH, int OW, void* stream);
int l2i_maxpool2d_bwd_f32(float* gx, const float* gy, const uint8_t* idx, int64_t planes, int H, int W, int k, int s,
                          int pad, int OH, int OW, void* stream);
/* MaxPool2d(2, 2) backward fused with the ContentLoss gradient of the pooled layer's input (VGG conv1_2 tap,
 * transform_base.py:57-63,440-452): gx = maxpool_bwd(gy) + coef * (coef_dev ? coef_dev[0] : 1) * (b - a), a / b / gx [planes, 2*OH, 2*OW].
 * One pass over a and b instead of sqdiff-gradient + pool backward + add. */
int l2i_maxpool2x2_bwd_add_diff_f32(float* gx, const float* gy, const uint8_t* idx, const float* a, const float* b, float coef,
                                    const float* coef_dev, int64_t planes, int OH, int OW, void* stream);

/* ContentLoss (transform_base.py:57-63): sum_out[0] += sum (a-b)^2 ; grad[i] = coef*(coef_dev ? coef_dev[0] : 1)*(b[i]-a[i])
 * (= d/d b of the scaled loss; coef_dev is a device scalar so that the upstream gradient needs no host sync).
 * sum_out / grad / coef_dev may each be NULL. */
int l2i_sqdiff_f32(float* sum_out, float* grad, const float* a, const float* b, int64_t n, float coef,
                   const float* coef_dev, void* stream);

/* y[i] = a[i]*sa[(i/step) % ...] ... small utilities */
int l2i_axpby_f32(float* y, const float* a, const float* b, float alpha, float beta, int64_t n, void* stream);

/* relu-masked copy: y = g * (ref > 0 ? 1 : 0) */
int l2i_relu_mask_f32(float* y, const float* g, const float* ref, int64_t n, void* stream);

/* ---- training a conv net (scene_regressor_256.py:118-171 trains the ResNet-50 regressor; the walk path never needs these) ----
 * Weight gradient of a correlation: dw[co,ci,ky,kx] += sum_{b,oy,ox} gy[b,co,oy,ox] * x[b,ci,oy*stride+ky-pad_y,ox*stride+kx-pad_x]
 * (out-of-range x reads as zero).  dw [Cout,Cin,KH,KW] must be zeroed (or hold the value to accumulate into) by the caller; partial sums
 * meet by fp32 atomics, so the summation order is not fixed.  Built for KW in {1,3,7}. */
int l2i_conv2d_wgrad_f32(float* dw, const float* x, const float* gy, int B, int Cin, int H, int W, int Cout, int OH, int OW,
                         int KH, int KW, int stride, int pad_y, int pad_x, void* stream);
/* BatchNorm2d in training mode on [B,C,HW] maps.  stats: sum[c] += sum x, sumsq[c] += sum x^2 in float64 (caller zeroes them).
 * apply: y = relu?(x*scale[c] + shift[c] (+ residual)).  bwd_reduce: with dy = gy * (out_mask > 0 if given), xhat = (x-mean[c])*invstd[c]:
 * sum_dy[c] += sum dy, sum_dyxh[c] += sum dy*xhat (float64, caller zeroes).  bwd_apply: dx = gamma[c]*invstd[c] * (dy - mean_dy[c] -
 * xhat*mean_dyxh[c]) with mean_* = the sums / (B*HW); dy_masked (optional) receives dy (the gradient a residual branch takes). */
int l2i_bn_stats_f32(double* sum, double* sumsq, const float* x, int B, int C, int64_t HW, void* stream);
int l2i_bn_apply_f32(float* y, const float* x, const float* scale, const float* shift, const float* residual, int relu, int B, int C,
                     int64_t HW, void* stream);
int l2i_bn_bwd_reduce_f32(double* sum_dy, double* sum_dyxh, const float* gy, const float* out_mask, const float* x, const float* mean,
                          const float* invstd, int B, int C, int64_t HW, void* stream);
int l2i_bn_bwd_apply_f32(float* dx, float* dy_masked, const float* gy, const float* out_mask, const float* x, const float* mean,
                         const float* invstd, const float* gamma, const float* mean_dy, const float* mean_dyxh, int B, int C, int64_t HW,
                         void* stream);

/* ---- PGGAN-256 generator (BASELINE config 1; reference graphs/pggan/model_256.py) ----
 * PixelNorm fused with the LeakyReLU that follows it (model_256.py:78-84, 128-150) on [B,C,HW] maps:
 *   y = lrelu(x / sqrt(mean_c x^2 + eps), slope)      (slope = 1: the plain PixelNorm of the latent code, model_256.py:230)
 * and its backward given gy = dL/dy and the forward INPUT x. */
int l2i_pixelnorm_act_f32(float* y, const float* x, int B, int C, int64_t HW, float eps, float slope, void* stream);
int l2i_pixelnorm_act_bwd_f32(float* dx, const float* gy, const float* x, int B, int C, int64_t HW, float eps, float slope, void* stream);
/* F.upsample(scale_factor=2) (nearest, model_256.py:240) on [planes,H,W] -> [planes,2H,2W], times `scale` (scale 0.25 = the adjoint of the
 * 2x2 mean below); and y[i,j] = scale * sum of the 2x2 window of x [planes,2*OH,2*OW] (scale 0.25 = F.upsample(size=half, bilinear) of
 * graphs/pggan/transform_base.py:320; scale 1 = the adjoint of the nearest upsample).  Even widths only. */
int l2i_upsample2x_nearest_f32(float* y, const float* x, int64_t planes, int H, int W, float scale, void* stream);
int l2i_pool2x2_f32(float* y, const float* x, int64_t planes, int OH, int OW, float scale, void* stream);

/* ---- style-dependent vectors of a generator pass (networks.py:148-156 EqualLinear modulation, :231-239 demodulation, and their gradients) ----
 * Segmented mat-vec:  out[b, r] = epi( sum over the segment's parts of  sum_k pre(in)[b, k] * w[w_off + k * w_pitch + r] ),  r < rows, b < B.
 * One launch evaluates every layer of a kind (all modulations s = w A^T + bias and the ToRGB weights; all demodulation factors; all
 * d s; the whole latent gradient).  `segs` / `block_seg` are DEVICE arrays built once per network (latent2im_amd/generator.py): block i works
 * on rows block_seg[2i+1]*64 .. +63 of segment block_seg[2i].  Offsets that scale with the batch are (constant, per-sample) pairs:
 * offset = off_c + B * off_b.
 *   pre: 0  x = in[idx]                       idx = in_off + b * in_bstride + k
 *        1  x = in[idx]^2                                                              (demodulation: sum s^2 T)
 *        2  x = in[idx] * in2[idx]^2                                                   (d demod * demod^3 = red / demod * demod^3)
 *        3  x = sum_o in2[in_off + (b * K + k) * 3 + o] * wrgb[aux_off + o * K + k]    (ToRGB: d s_rgb from the [B, C, 3] reduction, read from in2)
 *   epi: 0  out = acc + bias[bias_off + r]; if rgb_off_b >= 0 also wmod[B * rgb_off_b + (b * 3 + o) * rows + r] = wrgb[rgb_w_off + o * rows + r] * out
 *        1  out = rsqrt(acc + 1e-8)
 *        2  out = e1[e] - e2[e] * acc          e = e_off + b * e_bstride + r            (d s = q - s * (d demod demod^3) T)
 *        3  out = acc
 * K <= 512 and K % 4 == 0 for every part. */
typedef struct l2i_segmv_part {
    int32_t K, in_off_c, in_off_b, in_bstride, w_pitch, pre, aux_off, pad_;
    int64_t w_off;
} l2i_segmv_part;
typedef struct l2i_segmv_seg {
    int32_t rows, nparts, out_off_c, out_off_b, out_bstride, epi, bias_off, e_off_c, e_off_b, e_bstride, rgb_off_b, rgb_w_off;
    l2i_segmv_part part[2];
} l2i_segmv_seg;
int l2i_segmented_matvec_f32(float* out, const float* in, const float* in2, const float* w, const float* bias, const float* e1, const float* e2,
                             float* wmod, const float* wrgb, const l2i_segmv_seg* segs, const int32_t* block_seg, int nblocks, int B, void* stream);

/* ---- ABI version 6: the optimiser tail of the walk step on the fp16 path (l2i_optim.hip) ------------------------------------------------------
 * The reference ends a step with torch.optim.Adam on the walk tensor (transform_base.py:329-331, 487-488); under autocast it would do so through
 * a GradScaler.  These two entry points are that pair without a host synchronisation: all state lives on the device.
 *   state  int32[4]:  [L2I_LS_FOUND] 1 = a gradient of this step held an inf / NaN, [L2I_LS_TRACKER] clean steps since the last scale change,
 *                     [L2I_LS_SKIPPED] updates skipped so far, [L2I_LS_STEPS] steps seen so far
 *   scale  float[2]:  [0] the dynamic loss-scale factor (a power of two; multiplies every loss branch's incoming gradient), [1] its inverse */
#define L2I_LS_FOUND 0
#define L2I_LS_TRACKER 1
#define L2I_LS_SKIPPED 2
#define L2I_LS_STEPS 3
/* state[L2I_LS_FOUND] |= any(!isfinite(g[0..n))).  For walks with several parameter tensors: one call per gradient before the first update. */
int l2i_nonfinite_flag_f32(const float* g, int64_t n, int32_t* state, void* stream);
/* One Adam update of p (moments m, v; step = float step counter on the device) with torch.optim.Adam's arithmetic (no weight decay, no amsgrad),
 * SKIPPED — p, m, v, step untouched — when state[L2I_LS_FOUND] is set or (check_self) g itself holds an inf / NaN.  last != 0: afterwards the
 * scale state advances as torch._amp_update_scale_ does (found: scale[0] *= backoff, tracker = 0, skipped += 1; clean: ++tracker == interval ->
 * scale[0] = min(scale[0] * growth, max_scale) and tracker = 0), scale[1] = 1 / scale[0], steps += 1, and the flag is cleared.  scale may be NULL. */
int l2i_adam_guarded_f32(float* p, const float* g, float* m, float* v, float* step, int64_t n, float lr, float beta1, float beta2, float eps,
                         int32_t check_self, int32_t* state, float* scale, float growth, float backoff, int32_t interval, float max_scale,
                         int32_t last, void* stream);

/* [ABI 7] The regressor head and its BCE term in one launch (csrc/l2i_loss.hip) — transform_base.py:416-424: pred = fc(feat)[:, cols]; loss =
 * -mean(y log(max(pred, eps)) + (1 - y) log(max(1 - pred, eps))) (fp32 logs, float64 sum, like the torch expression on an fp32 pred and a float64 y).
 * feat [B, F], fc_w [A, F], fc_b [A] fp32; cols [K] int64 (K <= 64); target [B, K] float64 (target_f64 = 1) or fp32.  Writes loss[0] (float64), preds [B, K] and
 * g_feat [B, F] = d loss / d feat for an upstream gradient of 1 (clamp passes the gradient where its argument >= eps, as torch.clamp does). */
int l2i_reg_bce_f32(double* loss, float* preds, float* g_feat, const float* feat, const float* fc_w, const float* fc_b, const int64_t* cols,
                    const void* target, int target_f64, int B, int F, int K, float eps, void* stream);

const char* l2i_last_error(void);
/* Bumped whenever a struct of this header grows or an entry point changes meaning (1: round 1-2; 2: round 3, l2i_conv_params gained w_bstride /
 * out_f32; 3: round 4: l2i_conv2d_wino4_f32, l2i_sizeof_conv_params; 4: round 5: the l2i_*_h8_f16 entry points, wino4 tile_hint / CoutP % 32; 5: round 5: in_h8 / rgb_* fields, l2i_conv_img_h8; 6: round 6: l2i_nonfinite_flag_f32 / l2i_adam_guarded_f32, mask_out / mask_bits fields, l2i_mask_mul_bits_h8, the mask_bits argument of l2i_upfirdn2d_h8; 7: round 6: l2i_conv1x1_pair_h8, l2i_conv_chain3_h8, l2i_conv1x1_pair_f32, l2i_reg_bce_f32).  The ctypes binding (latent2im_amd/_lib.py) refuses a library whose version or struct size differs from its own mirror. */
#define L2I_ABI_VERSION 7
int l2i_abi_version(void);
int l2i_sizeof_conv_params(void);       /* sizeof(struct l2i_conv_params) of THIS build */

#ifdef __cplusplus
}
#endif
#endif /* L2I_H */
