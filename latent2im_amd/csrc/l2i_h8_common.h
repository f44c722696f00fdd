// Shared by the h8 (16-bit, channel-blocked) translation units, each compiled once per element type (bf16 as is, IEEE fp16 with -DL2I_H8_F16):
// element macros, fragment types, the pack / unpack converts, the sign-plane byte and the accumulator -> pixel-slot exchange.
#ifndef L2I_H8_COMMON_H
#define L2I_H8_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));

// [r5] Every h8 file is compiled twice (csrc/Makefile): as is = bf16 elements (l2i_conv2d_h8, l2i_conv_transpose2d_h8), and with -DL2I_H8_F16 = IEEE
// fp16 elements (the same entry points with the suffix _f16: BASELINE configs[4] says "fp16 MFMA").  The h8 layout, the DMA pipeline and the
// epilogues are element-type agnostic; what differs is the MFMA instruction (v_mfma_f32_32x32x16_{bf16,f16}: same rate), the two unpack
// converts and the packing convert (v_cvt_pk_{bf16,f16}_f32: one instruction per pair, round to nearest even, both).  ReLU-on-load stays the
// packed integer max: a negative fp16 is a negative int16 as well.  Everything lives in a per-type namespace: two objects with the same
// template kernels would otherwise be merged by the linker.
#ifdef L2I_H8_F16
#define H8_NS l2i_h8_f16
#define H8_NAME(n) n##_f16
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));          // (the fragment type keeps its name: "bf16x8" = eight 16-bit elements)
#define H8_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
#define H8_NS l2i_h8_bf16
#define H8_NAME(n) n
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define H8_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif

namespace H8_NS {

#ifdef L2I_H8_F16
__device__ __forceinline__ unsigned cvt_pk_bf16_h8(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo(unsigned u) { float r; asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(u)); return r; }
__device__ __forceinline__ float bf16_hi(unsigned u) { float r; asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(u)); return r; }
#else
__device__ __forceinline__ unsigned cvt_pk_bf16_h8(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#endif

__device__ __forceinline__ void h8_unpack(const u32x4& u, float (&v)[8]) {
    v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
    v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
}

// [r6] sign plane: bit e of the byte = (16-bit element e of the slot > 0).  Per dword (two elements): max with 0 as signed 16-bit integers (a negative
// fp16 / bf16, and -0, is a negative int16) then min with 1 as unsigned -> 0 / 1 in bits 0 and 16; the four dwords shifted by 0, 2, 4, 6 and folded.
__device__ __forceinline__ unsigned h8_sign_byte(const u32x4& o) {
    unsigned t[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm("v_pk_max_i16 %0, %0, 0" : "+v"(t[i]));
        asm("v_pk_min_u16 %0, %0, %1" : "+v"(t[i]) : "v"(0x00010001u));
    }
    const unsigned u = t[0] | (t[1] << 2) | (t[2] << 4) | (t[3] << 6);
    return (u | (u >> 15)) & 0xffu;
}

// one (m, n) accumulator tile -> the two 8-channel groups this lane finishes for quad pair pr: g[0..7]
__device__ __forceinline__ void h8_gather(const f32x16& a, int pr, int half, float (&g)[8]) {
    float lo[4], hi[4];                                            // quad 2 pr (group 2 pr) and quad 2 pr + 1 (group 2 pr + 1), own channel quad
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = a[8 * pr + e]; hi[e] = a[8 * pr + 4 + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // lanes 32-63 of lo <-> lanes 0-31 of hi: lower half then holds (own lo | upper's lo) = group 2 pr, upper half (lower's hi | own hi) = group 2 pr + 1
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[e]), __float_as_uint(hi[e]), false, false);
        lo[e] = __uint_as_float(r[0]); hi[e] = __uint_as_float(r[1]);
    }
    // lower half: lo = channels 0-3 (own), hi = channels 4-7 (from the upper half's lo)
    // upper half: lo = channels 0-3 of group 2 pr + 1 (from the lower half's hi), hi = channels 4-7 (own)
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[e] = lo[e]; g[4 + e] = hi[e]; }
    (void)half;
}

}  // namespace H8_NS
#endif
