// Optimizer-side elementwise kernels of the training step for gfx950:
// fused AdamW (seaweed_apt/distilled_trainer.py:69-75 constructs torch AdamW,
// lr 5e-6, wd 0.01) and the EMA update (distilled_trainer.py:319-334, done on
// the CPU in the reference with a full GPU->CPU copy per step).
#include "omh_common.h"

namespace {

__global__ __launch_bounds__(256)
void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                  int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                  float inv_scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * inv_scale;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2 + eps;          // bc2 = sqrt(1 - beta2^t)
        p[i] = p[i] * (1.0f - lr * wd) - (lr / bc1) * (mi / denom);
    }
}

// one launch for many tensors: table[t] = {p, g, m, v, numel} (device int64 x 5), blockIdx.y = tensor
__global__ __launch_bounds__(256)
void adamw_multi_kernel(const int64_t* __restrict__ table, float lr, float beta1, float beta2, float eps, float wd,
                        float bc1, float bc2, float inv_scale) {
    const int64_t* e = table + 5 * (int64_t)blockIdx.y;
    float* p = (float*)e[0];
    const float* g = (const float*)e[1];
    float* m = (float*)e[2];
    float* v = (float*)e[3];
    const int64_t n = e[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * inv_scale;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] * (1.0f - lr * wd) - (lr / bc1) * (mi / (sqrtf(vi) / bc2 + eps));
    }
}

__global__ __launch_bounds__(256)
void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float decay) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        ema[i] = ema[i] * decay + p[i] * (1.0f - decay);
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

extern "C" int omh_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                              omh_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0 || grad_scale == 0.f) return OMH_E_BADARG;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.0f - powf(beta2, (float)step));
    omh_clear_status();
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, 1.0f / grad_scale);
    return omh_launch_status();
}

extern "C" int omh_adamw_multi(const int64_t* table, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int32_t step, float grad_scale, omh_stream_t stream) {
    if (!table || n_tensors <= 0 || step <= 0 || grad_scale == 0.f) return OMH_E_BADARG;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.0f - powf(beta2, (float)step));
    omh_clear_status();
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(64, n_tensors), dim3(256), 0, (hipStream_t)stream, table, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, 1.0f / grad_scale);
    return omh_launch_status();
}

extern "C" int omh_ema_update(float* ema, const float* p, int64_t n, float decay, omh_stream_t stream) {
    if (!ema || !p || n <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ema, p, n, decay);
    return omh_launch_status();
}
