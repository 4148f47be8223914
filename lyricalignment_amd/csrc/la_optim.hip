// la_optim.hip -- optimizer step of the fine-tune path (train_multitask.py:337-340, 683-686): global-norm gradient
// clipping (torch.nn.utils.clip_grad_norm_(params, 1.0)) fused into torch.optim.AdamW's update, on FLAT f32 buffers
// (one bucket per parameter group: the head at lr 5e-3, the Whisper backbone at 5e-6; weight_decay 1e-5).
// HBM-bound elementwise work: 16 B/lane accesses, grid-stride.  Per element: reads p, g, m, v (16 B), writes p, m, v (12 B).
#include "la_common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float *g, int64_t n, double *out) {
    __shared__ double red[4];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4 *>(g + i);
            s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            for (int64_t j = i; j < n; ++j) s += (double)g[j] * g[j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// clip coefficient = min(1, max_norm / (sqrt(sum_sq) + 1e-6)) read on the device: no host round trip between the
// all-reduce, the norm and the update
__global__ __launch_bounds__(256) void adamw_kernel(float *p, const float *g, float *m, float *v, int64_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const double *sum_sq, float max_norm, float grad_prescale) {
    float clip = 1.0f;
    if (sum_sq) {
        const float norm = (float)sqrt(*sum_sq) * grad_prescale;
        clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    const float gs = clip * grad_prescale;
    const float step = lr / bc1;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.0f - lr * wd);                  // decoupled weight decay
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

}  // namespace

extern "C" int la_grad_sqnorm_f32(const float *grad, int64_t n, double *sum_sq, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(grad && sum_sq && n > 0 && (uintptr_t)grad % 16 == 0, "grad_sqnorm: bad arguments");
    const int grid = (int)std::min<int64_t>(2048, la::cdiv(n, 1024));
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, grad, n, sum_sq);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_adamw_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                 const double *clip_sum_sq, float max_norm, float grad_prescale, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw_step: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const int grid = (int)std::min<int64_t>(4096, la::cdiv(n, 256));
    la::TimerScope ts("adamw", (hipStream_t)stream_);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, clip_sum_sq, max_norm, grad_prescale);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
