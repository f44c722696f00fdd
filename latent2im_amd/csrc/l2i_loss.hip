// l2i_loss.hip — [r6] the regressor head and its BCE term in ONE launch (gfx950).  Entry point l2i_reg_bce_f32.
//
// Reference: transform_base.py:416-424 — `reg_preds = self.regressor(logit)` (torchvision ResNet-50: avg-pooled features -> fc), the attribute columns selected,
// `get_bce_loss(pred, y)` = -mean(y log(max(pred, eps)) + (1 - y) log(max(1 - pred, eps))) with y in float64 — and what autograd runs backwards through those
// ops.  As torch ops that is ~40 launches of a few microseconds each between the regressor's last conv and its first gradient conv, with nothing else on the
// stream (profiles/r06_timeline.txt: 0.2 - 0.3 ms without a large kernel running).  Here: block b finishes sample b — the K selected logits (2048-long dots
// with the fc rows, fp32), the predictions, d loss / d pred for an upstream gradient of 1 (fp32, the dtype autograd hands to the fp32 log nodes), and
// d loss / d feat = sum_k gpred[k] * fc_w[col_k][:]; block 0 also forms the loss itself over all samples in a fixed order (float64 sum of fp32 logs, like
// the torch expression): deterministic, no atomics.  The caller multiplies the stored gradient by the real upstream gradient (one launch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2i.h"
#include "l2i_internal.h"

namespace {
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void reg_bce_kernel(double* loss, float* preds, float* g_feat, const float* feat, const float* fc_w, const float* fc_b,
                                                      const int64_t* cols, const void* target, int target_f64, int B, int F, int K, float eps) {
    __shared__ float red[4];
    __shared__ float gp_s[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double inv_n = 1.0 / ((double)B * (double)K);
    auto tgt = [&](int bb, int k) -> double {
        return target_f64 ? reinterpret_cast<const double*>(target)[(size_t)bb * K + k] : (double)reinterpret_cast<const float*>(target)[(size_t)bb * K + k];
    };
    auto logit = [&](int bb, int k) -> float {
        const float* x = feat + (size_t)bb * F;
        const float* w = fc_w + (size_t)cols[k] * F;
        float s = 0.f;
        for (int i = tid; i < F; i += 256) s += x[i] * w[i];
        return block_sum_256(s, red) + fc_b[cols[k]];
    };
    for (int k = 0; k < K; ++k) {
        const float p = logit(b, k);
        if (tid == 0) {
            const double y = tgt(b, k);
            const float a = (float)(-y * inv_n), c = (float)(-(1.0 - y) * inv_n);                  // d loss / d log(.) of the two terms, upstream gradient 1
            const float q = 1.f - p;
            float g = 0.f;
            if (p >= eps) g += a / fmaxf(p, eps);                                                   // clamp(min = eps) passes the gradient where p >= eps
            if (q >= eps) g -= c / fmaxf(q, eps);                                                   // d (1 - p) / d p = -1
            preds[(size_t)b * K + k] = p;
            gp_s[k] = g;
        }
    }
    __syncthreads();
    for (int i = tid; i < F; i += 256) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += gp_s[k] * fc_w[(size_t)cols[k] * F + i];
        g_feat[(size_t)b * F + i] = s;
    }
    if (b == 0) {                                           // the loss: every sample again, fixed order
        double acc = 0.0;
        for (int bb = 0; bb < B; ++bb)
            for (int k = 0; k < K; ++k) {
                const float p = logit(bb, k);
                if (tid == 0) {
                    const double y = tgt(bb, k);
                    acc += y * (double)logf(fmaxf(p, eps)) + (1.0 - y) * (double)logf(fmaxf(1.f - p, eps));
                }
            }
        if (tid == 0) loss[0] = -acc * inv_n;
    }
}
}  // namespace

extern "C" int l2i_reg_bce_f32(double* loss, float* preds, float* g_feat, const float* feat, const float* fc_w, const float* fc_b, const int64_t* cols,
                               const void* target, int target_f64, int B, int F, int K, float eps, void* stream) {
    if (!loss || !preds || !g_feat || !feat || !fc_w || !fc_b || !cols || !target) return l2i_set_error(L2I_E_ARG, "reg_bce: null tensor");
    if (B <= 0 || F <= 0 || K <= 0 || K > 64) return l2i_set_error(L2I_E_ARG, "reg_bce: 1 <= K <= 64 selected attributes");
    hipLaunchKernelGGL(reg_bce_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, loss, preds, g_feat, feat, fc_w, fc_b, cols, target, target_f64, B, F, K, eps);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
