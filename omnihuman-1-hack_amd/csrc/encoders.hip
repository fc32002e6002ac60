// Elementwise / row kernels of the prompt-side encoders: the umT5 text encoder
// (seaweed_apt/wan/modules/t5.py) and the vision tower of the CLIP (seaweed_apt/wan/modules/clip.py).  Their matrix
// products run on omh_gemm_bf16 (head_dim 64 / 80 attention as two batched GEMMs around the softmax below: both
// encoders run once per prompt on <= 512 tokens, nothing here is on the per-step path).
#include "omh_common.h"

namespace {

// out[r][:] = table[ids[r]][:]          nn.Embedding (t5.py:306)
__global__ __launch_bounds__(256)
void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ out,
                        int64_t rows, int dim, int64_t vocab) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float4* src = (const float4*)(table + id * dim);
    float4* dst = (float4*)(out + r * dim);
    for (int c = threadIdx.x & 63; c < (dim >> 2); c += 64) dst[c] = src[c];
}

// the same from a bf16 table (umt5-xxl ships as bf16: models_t5_umt5-xxl-enc-bf16.pth, text2video.py:64-70)
__global__ __launch_bounds__(256)
void gather_rows_bf16_kernel(const uint16_t* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ out,
                             int64_t rows, int dim, int64_t vocab) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const uint2* src = (const uint2*)(table + id * dim);
    float4* dst = (float4*)(out + r * dim);
    for (int c = threadIdx.x & 63; c < (dim >> 2); c += 64) {
        const uint2 u = src[c];
        dst[c] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                             __uint_as_float(u.y & 0xffff0000u));
    }
}

// T5LayerNorm (t5.py:55-69): y = w * x * rsqrt(mean(x^2) + eps); fp32 and / or bf16 result
__global__ __launch_bounds__(256)
void rmsnorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps, float* __restrict__ yf,
                        uint16_t* __restrict__ yb, int64_t rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float4* xr = (const float4*)(x + r * dim);
    const int nv = dim >> 2;
    float q = 0.f;
    for (int c = lane; c < nv; c += 64) {
        const float4 v = xr[c];
        q += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    const float rinv = rsqrtf(wave_sum(q) / dim + eps);
    for (int c = lane; c < nv; c += 64) {
        float4 v = xr[c];
        const float4 g = ((const float4*)w)[c];
        v.x *= rinv * g.x; v.y *= rinv * g.y; v.z *= rinv * g.z; v.w *= rinv * g.w;
        if (yf) ((float4*)(yf + r * dim))[c] = v;
        if (yb) ((uint2*)(yb + r * dim))[c] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
}

// nn.LayerNorm with affine, fp32 in / out (clip.py:47-50 pre_norm on the embedded tokens)
__global__ __launch_bounds__(256)
void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                          float eps, float* __restrict__ y, int64_t rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float4* xr = (const float4*)(x + r * dim);
    const int nv = dim >> 2;
    float s = 0.f;
    for (int c = lane; c < nv; c += 64) { const float4 v = xr[c]; s += v.x + v.y + v.z + v.w; }
    const float mean = wave_sum(s) / dim;
    float q = 0.f;
    for (int c = lane; c < nv; c += 64) {
        const float4 v = xr[c];
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
    const float rinv = rsqrtf(wave_sum(q) / dim + eps);
    for (int c = lane; c < nv; c += 64) {
        float4 v = xr[c];
        const float4 g = ((const float4*)w)[c], bb = ((const float4*)b)[c];
        v.x = (v.x - mean) * rinv * g.x + bb.x; v.y = (v.y - mean) * rinv * g.y + bb.y;
        v.z = (v.z - mean) * rinv * g.z + bb.z; v.w = (v.w - mean) * rinv * g.w + bb.w;
        ((float4*)(y + r * dim))[c] = v;
    }
}

// Row softmax with the additive terms of T5Attention (t5.py:101-113): y[h*L + i][j] = softmax_j( x * scale +
// table[bucket[i*L + j]][h] ) over keys j < klen (masked keys: exactly zero weight); columns L..ldy-1 are zeroed so
// that P can be the A operand of the P.V GEMM with K = ldy.
__global__ __launch_bounds__(256)
void softmax_bias_rows_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                              int H, int L, float scale, const int32_t* __restrict__ bucket,
                              const float* __restrict__ table, int klen) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (int64_t)H * L) return;
    const int h = (int)(r / L), i = (int)(r % L);
    const float* xr = x + r * ldx;
    uint16_t* yr = y + r * ldy;
    const int32_t* br = bucket ? bucket + (int64_t)i * L : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < klen; j += 64) {
        float v = xr[j] * scale;
        if (br) v += table[br[j] * H + h];
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
    for (int j = lane; j < klen; j += 64) {
        float v = xr[j] * scale;
        if (br) v += table[br[j] * H + h];
        s += __expf(v - mx);
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.0f / s : 0.f;
    for (int j = lane; j < (int)ldy; j += 64) {
        float p = 0.f;
        if (j < klen) {
            float v = xr[j] * scale;
            if (br) v += table[br[j] * H + h];
            p = __expf(v - mx) * inv;
        }
        yr[j] = f2bf(p);
    }
}

// out = a * b on bf16 (the gated product fc1(x) * gelu(gate(x)) of T5FeedForward, t5.py:137)
__global__ __launch_bounds__(256)
void mul_bf16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ o, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * blockDim.x * 2) {
        const uint32_t ua = *(const uint32_t*)(a + i), ub = *(const uint32_t*)(b + i);
        *(uint32_t*)(o + i) = pack_bf2(bf2f((uint16_t)(ua & 0xffff)) * bf2f((uint16_t)(ub & 0xffff)),
                                       bf2f((uint16_t)(ua >> 16)) * bf2f((uint16_t)(ub >> 16)));
    }
}

// out[b][0] = cls + pos[0]; out[b][1 + i] = tok[b][i] + pos[1 + i]        (clip.py:280-287)
__global__ __launch_bounds__(256)
void vit_embed_kernel(const float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                      float* __restrict__ out, int B, int n, int dim) {
    const int64_t total = (int64_t)B * (n + 1) * (dim >> 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (dim >> 2));
        const int64_t row = i / (dim >> 2);
        const int t = (int)(row % (n + 1)), b = (int)(row / (n + 1));
        const float4 p = ((const float4*)(pos + (int64_t)t * dim))[c];
        const float4 v = t == 0 ? ((const float4*)cls)[c] : ((const float4*)(tok + ((int64_t)b * n + t - 1) * dim))[c];
        ((float4*)(out + row * dim))[c] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
    }
}

inline unsigned rows4(int64_t rows) { return (unsigned)((rows + 3) / 4); }

}  // namespace

extern "C" int omh_gather_rows_f32(const float* table, const int64_t* ids, float* out, int64_t rows, int32_t dim,
                                   int64_t vocab, omh_stream_t stream) {
    if (!table || !ids || !out || rows <= 0 || dim <= 0 || vocab <= 0) return OMH_E_BADARG;
    if ((dim & 3) || ((uintptr_t)table & 15) || ((uintptr_t)out & 15)) return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows4(rows)), dim3(256), 0, (hipStream_t)stream, table, ids, out, rows,
                       dim, vocab);
    return omh_launch_status();
}

extern "C" int omh_gather_rows_bf16(const void* table_bf16, const int64_t* ids, float* out, int64_t rows, int32_t dim,
                                    int64_t vocab, omh_stream_t stream) {
    if (!table_bf16 || !ids || !out || rows <= 0 || dim <= 0 || vocab <= 0) return OMH_E_BADARG;
    if ((dim & 3) || ((uintptr_t)table_bf16 & 7) || ((uintptr_t)out & 15)) return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(rows4(rows)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)table_bf16, ids, out, rows, dim, vocab);
    return omh_launch_status();
}

extern "C" int omh_rmsnorm_f32(const float* x, const float* weight, float eps, float* y_f32, void* y_bf16, int64_t rows,
                               int32_t dim, omh_stream_t stream) {
    if (!x || !weight || (!y_f32 && !y_bf16) || rows <= 0 || dim <= 0) return OMH_E_BADARG;
    if ((dim & 3) || ((uintptr_t)x & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)y_f32 & 15) || ((uintptr_t)y_bf16 & 7))
        return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3(rows4(rows)), dim3(256), 0, (hipStream_t)stream, x, weight, eps, y_f32,
                       (uint16_t*)y_bf16, rows, dim);
    return omh_launch_status();
}

extern "C" int omh_layernorm_f32(const float* x, const float* weight, const float* bias, float eps, float* y,
                                 int64_t rows, int32_t dim, omh_stream_t stream) {
    if (!x || !weight || !bias || !y || rows <= 0 || dim <= 0) return OMH_E_BADARG;
    if ((dim & 3) || ((uintptr_t)x & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)y & 15))
        return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3(rows4(rows)), dim3(256), 0, (hipStream_t)stream, x, weight, bias, eps,
                       y, rows, dim);
    return omh_launch_status();
}

extern "C" int omh_softmax_bias_rows(const float* x, int64_t ldx, void* y_bf16, int64_t ldy, int32_t H, int32_t L,
                                     float scale, const int32_t* bucket, const float* table, int32_t klen,
                                     omh_stream_t stream) {
    if (!x || !y_bf16 || H <= 0 || L <= 0 || ldx < L || ldy < L || klen < 0 || klen > L) return OMH_E_BADARG;
    if ((bucket == nullptr) != (table == nullptr)) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(softmax_bias_rows_kernel, dim3(rows4((int64_t)H * L)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (uint16_t*)y_bf16, ldy, H, L, scale, bucket, table, klen);
    return omh_launch_status();
}

extern "C" int omh_mul_bf16(const void* a, const void* b, void* out, int64_t n, omh_stream_t stream) {
    if (!a || !b || !out || n <= 0) return OMH_E_BADARG;
    if ((n & 1) || ((uintptr_t)a & 3) || ((uintptr_t)b & 3) || ((uintptr_t)out & 3)) return OMH_E_ALIGN;
    int64_t g = (n / 2 + 255) / 256;
    g = g > 8192 ? 8192 : g;
    omh_clear_status();
    hipLaunchKernelGGL(mul_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)out, n);
    return omh_launch_status();
}

extern "C" int omh_vit_embed(const float* tok, const float* cls, const float* pos, float* out, int32_t B, int32_t n,
                             int32_t dim, omh_stream_t stream) {
    if (!tok || !cls || !pos || !out || B <= 0 || n <= 0 || dim <= 0) return OMH_E_BADARG;
    if ((dim & 3) || ((uintptr_t)tok & 15) || ((uintptr_t)cls & 15) || ((uintptr_t)pos & 15) || ((uintptr_t)out & 15))
        return OMH_E_ALIGN;
    int64_t g = ((int64_t)B * (n + 1) * (dim >> 2) + 255) / 256;
    g = g > 8192 ? 8192 : g;
    omh_clear_status();
    hipLaunchKernelGGL(vit_embed_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, out, B, n, dim);
    return omh_launch_status();
}
