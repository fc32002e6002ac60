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

// one expression for both EMA kernels (the compiler's fma contraction would otherwise pick per loop shape)
__device__ __forceinline__ float ema_one(float e, float p, float decay) {
    return __builtin_fmaf(e, decay, p * (1.0f - decay));
}

__global__ __launch_bounds__(256)
void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float decay) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        ema[i] = ema_one(ema[i], p[i], decay);
}

// all parameters of a model in ONE launch (distilled_trainer.py:319-334 is a Python loop over ~825 tensors): table[t] =
// {ema, p, numel, first chunk} (device int64 x 4), one workgroup per 4096-element chunk, binary search for its tensor;
// ema_kernel's arithmetic on every element (same bits)
__global__ __launch_bounds__(256)
void ema_multi_kernel(const int64_t* __restrict__ table, int n_entries, float decay) {
    const int64_t t = blockIdx.x;
    int lo = 0, hi = n_entries - 1;                               // last entry whose first chunk is <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * 4 + 3] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + (int64_t)lo * 4;
    float* ema = (float*)e[0];
    const float* p = (const float*)e[1];
    const int64_t n = e[2], i0 = (t - e[3]) * 4096;
    const int64_t i1 = i0 + 4096 < n ? i0 + 4096 : n;
    if (((((uintptr_t)ema) | ((uintptr_t)p)) & 15) == 0) {
        for (int64_t i = i0 + 4 * (int64_t)threadIdx.x; i + 3 < i1; i += 1024) {
            float4 a = *(const float4*)(ema + i);
            const float4 b = *(const float4*)(p + i);
            a.x = ema_one(a.x, b.x, decay); a.y = ema_one(a.y, b.y, decay);
            a.z = ema_one(a.z, b.z, decay); a.w = ema_one(a.w, b.w, decay);
            *(float4*)(ema + i) = a;
        }
        for (int64_t i = i0 + ((i1 - i0) & ~(int64_t)3) + threadIdx.x; i < i1; i += 256)
            ema[i] = ema_one(ema[i], p[i], decay);
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) ema[i] = ema_one(ema[i], p[i], decay);
    }
}

// ---------------------------------------------------------------------------------------------- weight packing
// After an optimizer step every bf16 operand copy of the fp32 master weights is stale.  The training step used to
// rebuild them per block with ~20 small cast / transpose launches (11 600 transposes + 8 200 casts per bench run);
// this is ONE launch for all of them: entry e of the device table = {src fp32 [rows, cols] contiguous, dst bf16
// [rows, ld_dst] or 0, dstT bf16 [cols, ld_dstT] or 0, rows, cols, ld_dst, ld_dstT, first tile, kind}.
//   kind 0: dst = bf16(src) and / or dstT = bf16(src)^T, one 64 x 64 tile per workgroup (through LDS);
//   kind 1: dst fp32 = src (rows = 1: the concatenated biases of fused projections), 4096 elements per workgroup.
constexpr int PK_COLS = 9;

__global__ __launch_bounds__(256)
void pack_weights_kernel(const int64_t* __restrict__ table, int n_entries) {
    __shared__ uint16_t tile[64][66];
    const int64_t t = blockIdx.x;
    int lo = 0, hi = n_entries - 1;                               // last entry whose first tile is <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * PK_COLS + 7] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + (int64_t)lo * PK_COLS;
    const float* src = (const float*)e[0];
    const int64_t rows = e[3], cols = e[4], ld_dst = e[5], ld_t = e[6];
    const int64_t local = t - e[7];
    const int tid = threadIdx.x;
    if (e[8] == 1) {                                              // fp32 copy
        float* dst = (float*)e[1];
        const int64_t n = rows * cols, i0 = local * 4096;
        for (int64_t i = i0 + tid; i < min(n, i0 + 4096); i += 256) dst[i] = src[i];
        return;
    }
    uint16_t* dst = (uint16_t*)e[1];
    uint16_t* dstT = (uint16_t*)e[2];
    const int64_t tiles_c = (cols + 63) >> 6;
    const int64_t r0 = (local / tiles_c) << 6, c0 = (local % tiles_c) << 6;
    const bool vec = (cols & 3) == 0;
    const int tr = tid >> 4, tc = (tid & 15) << 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = tr + 16 * j;
        const int64_t r = r0 + rl, c = c0 + tc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            if (vec && c + 3 < cols) v = *(const float4*)(src + r * cols + c);
            else {
                if (c < cols) v.x = src[r * cols + c];
                if (c + 1 < cols) v.y = src[r * cols + c + 1];
                if (c + 2 < cols) v.z = src[r * cols + c + 2];
                if (c + 3 < cols) v.w = src[r * cols + c + 3];
            }
        }
        const uint32_t lo2 = pack_bf2(v.x, v.y), hi2 = pack_bf2(v.z, v.w);
        if (dst && r < rows) {
            if ((ld_dst & 3) == 0 && c + 3 < cols) *(uint2*)(dst + r * ld_dst + c) = make_uint2(lo2, hi2);
            else {
                if (c < cols) dst[r * ld_dst + c] = (uint16_t)(lo2 & 0xffff);
                if (c + 1 < cols) dst[r * ld_dst + c + 1] = (uint16_t)(lo2 >> 16);
                if (c + 2 < cols) dst[r * ld_dst + c + 2] = (uint16_t)(hi2 & 0xffff);
                if (c + 3 < cols) dst[r * ld_dst + c + 3] = (uint16_t)(hi2 >> 16);
            }
        }
        *(uint32_t*)&tile[rl][tc] = lo2;
        *(uint32_t*)&tile[rl][tc + 2] = hi2;
    }
    if (!dstT) return;                                            // workgroup-uniform
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = tr + 16 * j;                               // source column = row of the transposed copy
        const int64_t c = c0 + cl, r = r0 + tc;
        if (c >= cols) continue;
        const uint16_t a0 = tile[tc][cl], a1 = tile[tc + 1][cl], a2 = tile[tc + 2][cl], a3 = tile[tc + 3][cl];
        if ((ld_t & 3) == 0 && r + 3 < rows) {
            *(uint2*)(dstT + c * ld_t + r) = make_uint2((uint32_t)a0 | ((uint32_t)a1 << 16), (uint32_t)a2 | ((uint32_t)a3 << 16));
        } else {
            if (r < rows) dstT[c * ld_t + r] = a0;
            if (r + 1 < rows) dstT[c * ld_t + r + 1] = a1;
            if (r + 2 < rows) dstT[c * ld_t + r + 2] = a2;
            if (r + 3 < rows) dstT[c * ld_t + r + 3] = a3;
        }
    }
}

// ---------------------------------------------------------------------------------------------- AdamW + weight packs
// Round 4: the optimizer step WRITES the bf16 operand copies the next training forward reads, instead of a second pass
// (pack_weights_kernel) that re-reads every updated fp32 parameter: entry e of the device table = {p, g, m, v fp32, dst,
// dstT, rows, cols, ld_dst, ld_dstT, first tile, kind}; kind 0: a [rows, cols] weight, one workgroup per 64 x 64 tile:
// AdamW on the tile, then dst = bf16(p) and / or dstT = bf16(p)^T through LDS (either may be 0); kind 1: a vector with an
// fp32 copy (dst fp32 = p: the concatenated biases of fused projections); kind 2: AdamW only; kinds 1 / 2 take 4096
// elements per workgroup.  The update is adamw_multi_kernel's arithmetic (equal up to the compiler's fma contraction), the copies are
// pack_weights_kernel's images of the parameters just written.
constexpr int AP_COLS = 12;

__device__ __forceinline__ float adamw_one(float p, float g, float& m, float& v, float lr, float beta1, float beta2, float eps,
                                           float wd, float bc1, float bc2, float inv_scale) {
    const float gi = g * inv_scale;
    const float mi = beta1 * m + (1.0f - beta1) * gi;
    const float vi = beta2 * v + (1.0f - beta2) * gi * gi;
    m = mi;
    v = vi;
    return p * (1.0f - lr * wd) - (lr / bc1) * (mi / (sqrtf(vi) / bc2 + eps));
}

__global__ __launch_bounds__(256)
void adamw_pack_kernel(const int64_t* __restrict__ table, int n_entries, float lr, float beta1, float beta2, float eps,
                       float wd, float bc1, float bc2, float inv_scale) {
    __shared__ uint16_t tile[64][66];
    const int64_t t = blockIdx.x;
    int lo = 0, hi = n_entries - 1;                               // last entry whose first tile is <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(int64_t)mid * AP_COLS + 10] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + (int64_t)lo * AP_COLS;
    float* P = (float*)e[0];
    const float* G = (const float*)e[1];
    float* M = (float*)e[2];
    float* V = (float*)e[3];
    const int64_t rows = e[6], cols = e[7], ld_dst = e[8], ld_t = e[9];
    const int64_t local = t - e[10];
    const int kind = (int)e[11];
    const int tid = threadIdx.x;
    if (kind != 0) {
        float* dst = kind == 1 ? (float*)e[4] : nullptr;
        const int64_t n = rows * cols, i0 = local * 4096;
        for (int64_t i = i0 + tid; i < min(n, i0 + 4096); i += 256) {
            float m = M[i], v = V[i];
            const float pn = adamw_one(P[i], G[i], m, v, lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale);
            M[i] = m; V[i] = v; P[i] = pn;
            if (dst) dst[i] = pn;
        }
        return;
    }
    uint16_t* dst = (uint16_t*)e[4];
    uint16_t* dstT = (uint16_t*)e[5];
    const int64_t tiles_c = (cols + 63) >> 6;
    const int64_t r0 = (local / tiles_c) << 6, c0 = (local % tiles_c) << 6;
    const bool vec = (cols & 3) == 0;
    const int tr = tid >> 4, tc = (tid & 15) << 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = tr + 16 * j;
        const int64_t r = r0 + rl, c = c0 + tc;
        float pv[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            const int64_t o = r * cols + c;
            if (vec && c + 3 < cols) {
                const float4 p4 = *(const float4*)(P + o), g4 = *(const float4*)(G + o);
                float4 m4 = *(const float4*)(M + o), v4 = *(const float4*)(V + o);
                pv[0] = adamw_one(p4.x, g4.x, m4.x, v4.x, lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale);
                pv[1] = adamw_one(p4.y, g4.y, m4.y, v4.y, lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale);
                pv[2] = adamw_one(p4.z, g4.z, m4.z, v4.z, lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale);
                pv[3] = adamw_one(p4.w, g4.w, m4.w, v4.w, lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale);
                *(float4*)(M + o) = m4;
                *(float4*)(V + o) = v4;
                *(float4*)(P + o) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            } else {
                for (int k = 0; k < 4; ++k)
                    if (c + k < cols) {
                        float m = M[o + k], v = V[o + k];
                        pv[k] = adamw_one(P[o + k], G[o + k], m, v, lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale);
                        M[o + k] = m; V[o + k] = v; P[o + k] = pv[k];
                    }
            }
        }
        const uint32_t lo2 = pack_bf2(pv[0], pv[1]), hi2 = pack_bf2(pv[2], pv[3]);
        if (dst && r < rows) {
            if ((ld_dst & 3) == 0 && c + 3 < cols) *(uint2*)(dst + r * ld_dst + c) = make_uint2(lo2, hi2);
            else {
                if (c < cols) dst[r * ld_dst + c] = (uint16_t)(lo2 & 0xffff);
                if (c + 1 < cols) dst[r * ld_dst + c + 1] = (uint16_t)(lo2 >> 16);
                if (c + 2 < cols) dst[r * ld_dst + c + 2] = (uint16_t)(hi2 & 0xffff);
                if (c + 3 < cols) dst[r * ld_dst + c + 3] = (uint16_t)(hi2 >> 16);
            }
        }
        *(uint32_t*)&tile[rl][tc] = lo2;
        *(uint32_t*)&tile[rl][tc + 2] = hi2;
    }
    if (!dstT) return;                                            // workgroup-uniform
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = tr + 16 * j;                               // source column = row of the transposed copy
        const int64_t c = c0 + cl, r = r0 + tc;
        if (c >= cols) continue;
        const uint16_t a0 = tile[tc][cl], a1 = tile[tc + 1][cl], a2 = tile[tc + 2][cl], a3 = tile[tc + 3][cl];
        if ((ld_t & 3) == 0 && r + 3 < rows) {
            *(uint2*)(dstT + c * ld_t + r) = make_uint2((uint32_t)a0 | ((uint32_t)a1 << 16), (uint32_t)a2 | ((uint32_t)a3 << 16));
        } else {
            if (r < rows) dstT[c * ld_t + r] = a0;
            if (r + 1 < rows) dstT[c * ld_t + r + 1] = a1;
            if (r + 2 < rows) dstT[c * ld_t + r + 2] = a2;
            if (r + 3 < rows) dstT[c * ld_t + r + 3] = a3;
        }
    }
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

extern "C" int omh_adamw_pack_multi(const int64_t* table, int32_t n_entries, int64_t total_tiles, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                                    omh_stream_t stream) {
    if (!table || n_entries <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL || step <= 0 || grad_scale == 0.f)
        return OMH_E_BADARG;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.0f - powf(beta2, (float)step));
    omh_clear_status();
    hipLaunchKernelGGL(adamw_pack_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table, n_entries,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2, 1.0f / grad_scale);
    return omh_launch_status();
}

extern "C" int omh_ema_update(float* ema, const float* p, int64_t n, float decay, omh_stream_t stream) {
    if (!ema || !p || n <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ema, p, n, decay);
    return omh_launch_status();
}

extern "C" int omh_ema_update_multi(const int64_t* table, int32_t n_entries, int64_t total_chunks, float decay,
                                    omh_stream_t stream) {
    if (!table || n_entries <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffLL) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(ema_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table, n_entries,
                       decay);
    return omh_launch_status();
}

extern "C" int omh_pack_weights_multi(const int64_t* table, int32_t n_entries, int64_t total_tiles, omh_stream_t stream) {
    if (!table || n_entries <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table,
                       n_entries);
    return omh_launch_status();
}
