// Shared by the translation units of libl2i_hip.so (not part of the public ABI).
#ifndef L2I_INTERNAL_H
#define L2I_INTERNAL_H
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

int l2i_set_error(int code, const char* msg);   // records msg for l2i_last_error(), returns code

struct l2i_conv_params;
bool l2i_gemm1x1_eligible(const l2i_conv_params& p);              // l2i_gemm.hip: DMA-fed GEMM form of unmasked 1x1 stride-1 layers
int l2i_launch_gemm1x1(const l2i_conv_params& p, hipStream_t st);
bool l2i_cin3_eligible(const l2i_conv_params& p);                 // l2i_cin3.hip: 3x3 convs of <= 3-channel images as one 27-long contraction
int l2i_launch_cin3(const l2i_conv_params& p, hipStream_t st);
bool l2i_convt_small_eligible(const l2i_conv_params& p);          // l2i_convt_small.hip: 7x7 / pad 3 stride-2 transposed conv onto <= 3 channels
int l2i_launch_convt_small(const l2i_conv_params& p, hipStream_t st);
bool l2i_conv3x3s2_eligible(const l2i_conv_params& p);            // l2i_conv_s2.hip: unmasked 3x3 stride-2 layers on maps >= 32 wide, both operands by LDS-DMA
int l2i_launch_conv3x3s2(const l2i_conv_params& p, hipStream_t st);
int l2i_launch_splitk_epilogue(const l2i_conv_params& q, hipStream_t st);     // l2i_conv.hip: y = epilogue(sum of q.ksplit partials in q.ws)

// [r5] The ABI-5 fields of l2i_conv_params belong to single entry points (rgb_*: l2i_conv2d_h8; pool_*: l2i_conv2d_wino4_f32; in_h8: the 7x7 kernel inside
// l2i_conv_transpose2d_f32).  Every other conv entry refuses a struct that carries them instead of silently not doing what was asked.
static inline const char* l2i_unsupported_v5_fields(const l2i_conv_params& p, bool allow_rgb, bool allow_pool, bool allow_in_h8) {
    if (!allow_rgb && (p.rgb_w || p.rgb_bias || p.rgb_out)) return "rgb_w / rgb_bias / rgb_out are fused in l2i_conv2d_h8 only";
    if (!allow_pool && (p.pool_out || p.pool_idx)) return "pool_out / pool_idx are fused in l2i_conv2d_wino4_f32 only";
    if (!allow_in_h8 && p.in_h8) return "in_h8 is read by the 7x7 kernel of l2i_conv_transpose2d_f32 only";
    if (!allow_rgb && (p.mask_out || p.mask_bits)) return "mask_out / mask_bits (one-bit activation masks) belong to l2i_conv2d_h8 / l2i_conv_transpose2d_h8";      // (allow_rgb = the h8 entries)
    return nullptr;
}

#define L2I_CHECK_LAUNCH()                                                      \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) return l2i_set_error(L2I_E_LAUNCH, hipGetErrorString(e_)); \
    } while (0)

// A function attribute (dynamic LDS limit) is PER DEVICE, and launches may come from several host threads: one bit per device id in an
// atomic word; the attribute is set before the bit (a racing thread at worst sets it a second time, which is harmless).
#define L2I_ONCE_PER_DEVICE(stmt)                                                       \
    do {                                                                                \
        static std::atomic<uint64_t> done_mask_{0};                                     \
        int dev_ = 0;                                                                   \
        (void)hipGetDevice(&dev_);                                                      \
        const uint64_t bit_ = 1ull << (dev_ & 63);                                      \
        if (!(done_mask_.load(std::memory_order_acquire) & bit_)) {                     \
            stmt;                                                                       \
            done_mask_.fetch_or(bit_, std::memory_order_release);                       \
        }                                                                               \
    } while (0)

static inline int l2i_grid_for(long long work_items, int per_block, int cap = 256 * 8) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
#endif
