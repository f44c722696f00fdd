// l2i_optim.hip — the optimiser tail of the walk-training step on the fp16 path: finite check, guarded Adam, dynamic loss scale.
//
// The reference's step ends in `self.optimizers.step()` (transform_base.py:487-488: torch.optim.Adam, betas (0.5, 0.99)) on ONE tiny tensor,
// the walk [n_attr, n_latent, 512] (9 - 46 K floats).  Under autocast the reference would wrap that step in a torch GradScaler: skip the update
// when the scaled gradients hold an inf / NaN, halve the scale, grow it again after `interval` clean steps.  The fp16 path here (nets16.py) scales
// each loss branch by a static power of two times ONE dynamic factor that lives on the device; this file is the part of the GradScaler that must
// not cost a host synchronisation (torch's own reads `found_inf` back with .item() before every optimiser step):
//
//   l2i_nonfinite_flag_f32   state[FOUND] |= any(!isfinite(g))                    (multi-tensor walks: one call per gradient tensor first)
//   l2i_adam_guarded_f32     if !found: Adam update of (p, m, v, step) exactly as torch's single-tensor Adam computes it; found: nothing moves.
//                            `last`: then the scale state advances the way torch._amp_update_scale_ does and the flag is cleared.
//
// One 1024-thread block per call: the tensors are tens of KB, and a single block can order "every lane has read the flag / the step counter"
// before "lane 0 rewrites them" with a barrier instead of a second launch.  The linear walk's whole tail is ONE launch (check_self = last = 1)
// where torch's foreach Adam issues ~12.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "l2i.h"
#include "l2i_internal.h"

namespace {
constexpr int NT = 1024;

__device__ __forceinline__ bool nonfinite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

__device__ __forceinline__ int block_any(int pred, int* sh) {
    // wave-wide OR by ballot, block-wide through one LDS word per wave
    const unsigned long long b = __ballot(pred);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = b != 0ull;
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) any |= sh[w];
    __syncthreads();
    return any;
}
}  // namespace

__global__ __launch_bounds__(NT) void nonfinite_flag_kernel(const float* __restrict__ g, long n, int32_t* __restrict__ state) {
    __shared__ int sh[NT / 64];
    int bad = 0;
    for (long i = threadIdx.x; i < n; i += NT) bad |= nonfinite(g[i]);
    bad = block_any(bad, sh);
    if (bad && threadIdx.x == 0) state[L2I_LS_FOUND] = 1;
}

__global__ __launch_bounds__(NT) void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, float* __restrict__ step, long n, float lr, float beta1, float beta2,
                                                          float eps, int check_self, int32_t* __restrict__ state, float* __restrict__ scale,
                                                          float growth, float backoff, int interval, float max_scale, int last) {
    __shared__ int sh[NT / 64];
    __shared__ float s_size, s_bc2;
    int found = state[L2I_LS_FOUND];
    if (check_self) {
        int bad = 0;
        for (long i = threadIdx.x; i < n; i += NT) bad |= nonfinite(g[i]);
        found |= block_any(bad, sh);
    }
    const float t = step[0] + 1.f;
    if (threadIdx.x == 0) {
        // bias corrections in double like the python floats of torch's _single_tensor_adam
        const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
        s_size = (float)((double)lr / bc1);
        s_bc2 = (float)sqrt(bc2);
    }
    __syncthreads();                                             // every lane holds `found` and `t`: the words may be rewritten below
    if (!found) {
        const float size = s_size, bc2s = s_bc2, omb1 = 1.f - beta1, omb2 = 1.f - beta2;
        for (long i = threadIdx.x; i < n; i += NT) {
            const float gi = g[i];
            const float mi = m[i] + (gi - m[i]) * omb1;          // exp_avg.lerp_(grad, 1 - beta1)
            const float vi = v[i] * beta2 + omb2 * gi * gi;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / bc2s + eps;
            p[i] = p[i] - size * (mi / denom);                   // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
        if (threadIdx.x == 0) step[0] = t;
    }
    if (threadIdx.x == 0) {
        if (check_self && found) state[L2I_LS_FOUND] = 1;         // visible to the next tensor of a multi-tensor step
        if (last) {
            // torch._amp_update_scale_: found -> scale *= backoff, tracker = 0; clean -> ++tracker == interval -> scale *= growth (kept finite), tracker = 0
            if (found) {
                state[L2I_LS_TRACKER] = 0;
                state[L2I_LS_SKIPPED] += 1;
                if (scale) scale[0] *= backoff;
            } else {
                const int tr = state[L2I_LS_TRACKER] + 1;
                if (tr >= interval && interval > 0) {
                    state[L2I_LS_TRACKER] = 0;
                    if (scale) { const float s2 = scale[0] * growth; if (s2 <= max_scale) scale[0] = s2; }
                } else {
                    state[L2I_LS_TRACKER] = tr;
                }
            }
            if (scale) scale[1] = 1.f / scale[0];
            state[L2I_LS_STEPS] += 1;
            state[L2I_LS_FOUND] = 0;
        }
    }
}

extern "C" int l2i_nonfinite_flag_f32(const float* g, int64_t n, int32_t* state, void* stream) {
    if (!g || !state || n < 0) return l2i_set_error(L2I_E_ARG, "nonfinite_flag: null tensor");
    if (n == 0) return L2I_OK;
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, g, (long)n, state);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}

extern "C" int l2i_adam_guarded_f32(float* p, const float* g, float* m, float* v, float* step, int64_t n, float lr, float beta1, float beta2, float eps,
                                    int32_t check_self, int32_t* state, float* scale, float growth, float backoff, int32_t interval, float max_scale,
                                    int32_t last, void* stream) {
    if (!p || !g || !m || !v || !step || !state || n <= 0) return l2i_set_error(L2I_E_ARG, "adam_guarded: null tensor");
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return l2i_set_error(L2I_E_ARG, "adam_guarded: betas in [0, 1), eps >= 0");
    if (scale && !(growth >= 1.f && backoff > 0.f && backoff <= 1.f)) return l2i_set_error(L2I_E_ARG, "adam_guarded: growth >= 1, 0 < backoff <= 1");
    hipLaunchKernelGGL(adam_guarded_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, step, (long)n, lr, beta1, beta2, eps, (int)check_self, state,
                       scale, growth, backoff, (int)interval, max_scale, (int)last);
    L2I_CHECK_LAUNCH();
    return L2I_OK;
}
