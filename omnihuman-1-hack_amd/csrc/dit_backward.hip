// Backward-pass kernels of the DiT block for gfx950 (training step of
// seaweed_apt/distilled_trainer.py:241-316; the reference gets all of this from
// autograd over aten ops).  The matrix products of the backward pass (dgrad,
// wgrad, attention dQ/dK/dV) reuse omh_gemm_bf16 on transposed operands; this
// file holds what surrounds them: transposes, column sums (bias grads) and the
// backward of LayerNorm+modulate, RMSNorm+RoPE, GELU-tanh, the gated residual
// and the softmax, plus the tiny fp32 dense layers of the time embedding.
// Parameter / modulation gradients are accumulated with fp32 atomics.
#include "omh_common.h"

namespace {

constexpr int MAXV = 32;

// ------------------------------------------------------------------ transpose
// out[b][c][r] = in[b][r][c]   (bf16), 64x64 tiles through LDS
__global__ __launch_bounds__(256)
void transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C, int64_t ld_in,
                           int64_t ld_out, int64_t bs_in, int64_t bs_out) {
    __shared__ uint16_t tile[64][66];
    const uint16_t* src = in + (int64_t)blockIdx.z * bs_in;
    uint16_t* dst = out + (int64_t)blockIdx.z * bs_out;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(int64_t)r * ld_in + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(int64_t)c * ld_out + r] = tile[tx][i];
    }
}

// The same with 8-byte global accesses (4 elements per thread along the contiguous index on both sides): pitches and
// batch strides multiples of 4, 8-byte aligned pointers.  Round 4: the 2-byte version moved 38 MB in 17 us (the V^T -> V
// copies of the attention backward, 62 per training step).
__global__ __launch_bounds__(256)
void transpose_bf16_vec_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C, int64_t ld_in,
                               int64_t ld_out, int64_t bs_in, int64_t bs_out) {
    __shared__ uint16_t tile[64][68];
    const uint16_t* src = in + (int64_t)blockIdx.z * bs_in;
    uint16_t* dst = out + (int64_t)blockIdx.z * bs_out;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) << 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = tr + 16 * j, r = r0 + rl, c = c0 + tc;
        uint2 v = make_uint2(0u, 0u);
        if (r < R) {
            const uint16_t* p = src + (int64_t)r * ld_in + c;
            if (c + 3 < C) v = *(const uint2*)p;
            else {
                uint16_t e[4] = {0, 0, 0, 0};
                for (int k = 0; k < 4; ++k) if (c + k < C) e[k] = p[k];
                v = make_uint2((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16));
            }
        }
        *(uint2*)&tile[rl][tc] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = tr + 16 * j, c = c0 + cl, r = r0 + tc;      // source column cl = output row
        if (c >= C || r >= R) continue;
        const uint16_t a0 = tile[tc][cl], a1 = tile[tc + 1][cl], a2 = tile[tc + 2][cl], a3 = tile[tc + 3][cl];
        uint16_t* q = dst + (int64_t)c * ld_out + r;
        if (r + 3 < R) *(uint2*)q = make_uint2((uint32_t)a0 | ((uint32_t)a1 << 16), (uint32_t)a2 | ((uint32_t)a3 << 16));
        else {
            q[0] = a0;
            if (r + 1 < R) q[1] = a1;
            if (r + 2 < R) q[2] = a2;
        }
    }
}

// ------------------------------------------------------------------ column sums
template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<uint16_t>(const uint16_t* p) { return bf2f(*p); }

template <typename T>
__global__ __launch_bounds__(256)
void colsum_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t R, int C, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += ldf<T>(x + r * ld + c);
    atomicAdd(out + c, s);
}

// Up to OMH_COLSUM_MAX column sums in ONE launch (the ~10 bias gradients of a block backward: each alone is a
// 10-15 us latency- and atomics-bound kernel; together they overlap).  Descriptors travel in the kernel arguments.
__global__ __launch_bounds__(256)
void colsum_multi_kernel(const omh_colsum_batch b, const int rpb) {
    int e = 0, local = blockIdx.x;
#pragma unroll 1
    while (e + 1 < b.n && local >= b.blocks[e]) { local -= b.blocks[e]; ++e; }
    const int C = b.C[e];
    const int col_chunks = (C + 255) / 256;
    const int c = (local % col_chunks) * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t R = b.R[e], ld = b.ld[e];
    const int64_t r0 = (int64_t)(local / col_chunks) * rpb;
    const int64_t r1 = min(R, r0 + rpb);
    float s = 0.f;
    if (b.is_bf16[e]) {
        const uint16_t* x = (const uint16_t*)b.x[e];
        for (int64_t r = r0; r < r1; ++r) s += bf2f(x[r * ld + c]);
    } else {
        const float* x = (const float*)b.x[e];
        for (int64_t r = r0; r < r1; ++r) s += x[r * ld + c];
    }
    atomicAdd(b.out[e] + c, s);
}

// The same with 16-byte loads (round 3; the one-column-per-thread kernel above reads bf16 two bytes at a time and ran at
// 2.8 TB/s, 2.7 ms per training step at 4 clips): a block is 32 column groups of 8 columns x 8 row lanes, a thread adds
// rows r0 + lane, r0 + lane + 8, ... of its chunk, the 8 lanes of a column are combined in LDS in a fixed order, one
// atomic per column and chunk (none to combine with when rpb covers all rows: deterministic mode).
// Needs C % 8 == 0, ld % 8 == 0 and 16-byte aligned rows (the host checks).
__global__ __launch_bounds__(256)
void colsum_multi_vec_kernel(const omh_colsum_batch b, const int rpb) {
    __shared__ float red[8][256 + 8];
    int e = 0, local = blockIdx.x;
#pragma unroll 1
    while (e + 1 < b.n && local >= b.blocks[e]) { local -= b.blocks[e]; ++e; }
    const int C = b.C[e];
    const int col_chunks = (C + 255) / 256;
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int cbase = (local % col_chunks) * 256;
    const int c0 = cbase + cg * 8;
    const int64_t R = b.R[e], ld = b.ld[e];
    const int64_t r0 = (int64_t)(local / col_chunks) * rpb;
    const int64_t r1 = min(R, r0 + rpb);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (c0 < C) {
        if (b.is_bf16[e]) {
            const uint16_t* x = (const uint16_t*)b.x[e];
            for (int64_t r = r0 + rl; r < r1; r += 8) {
                const uint4 u = *(const uint4*)(x + r * ld + c0);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s[2 * k] += __uint_as_float(w[k] << 16);
                    s[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
                }
            }
        } else {
            const float* x = (const float*)b.x[e];
            for (int64_t r = r0 + rl; r < r1; r += 8) {
                const float4 a = *(const float4*)(x + r * ld + c0);
                const float4 c = *(const float4*)(x + r * ld + c0 + 4);
                s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
                s[4] += c.x; s[5] += c.y; s[6] += c.z; s[7] += c.w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl][cg * 8 + k] = s[k];
    __syncthreads();
    const int c = cbase + threadIdx.x;
    if (c < C) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][threadIdx.x];
        atomicAdd(b.out[e] + c, t);
    }
}

// ------------------------------------------------------------------ GELU-tanh fwd / bwd (bf16)

__global__ __launch_bounds__(256)
void gelu_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = f2bf(gelu_tanh(bf2f(x[i])));
}

__global__ __launch_bounds__(256)
void gelu_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ xpre, uint16_t* __restrict__ dx,
                     int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = f2bf(bf2f(dy[i]) * gelu_tanh_grad(bf2f(xpre[i])));
}

// ------------------------------------------------------------------ gated residual fwd / bwd
// fwd: xo = xi + y * gate ; bwd: dy = dx * gate (bf16), dgate[b][c] += sum_rows dx * y
__global__ __launch_bounds__(256)
void gated_resid_fwd_kernel(const float* __restrict__ xi, const uint16_t* __restrict__ y, float* __restrict__ xo,
                            int64_t rows, int dim, float gate_const, const float* __restrict__ gate0,
                            const float* __restrict__ gate1, int64_t gate1_stride, int64_t rows_per_batch) {
    const int64_t total = rows * dim;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % dim);
        const int64_t r = i / dim;
        float g = gate_const;
        if (gate0) g += gate0[c];
        if (gate1) g += gate1[(r / rows_per_batch) * gate1_stride + c];
        xo[i] = xi[i] + bf2f(y[i]) * g;
    }
}

// Round 4: four columns per thread (16-byte dx loads, 8-byte y loads / dy stores) and four row lanes per workgroup
// — the first version moved one column per thread with 2-byte stores and a serial row loop (28.7 us for 77 MB at
// 6 240 x 1536).  The gate gradient: per-thread sums over the row lane's rows (ascending), the four lanes added in
// lane order through LDS, then one atomicAdd per column and workgroup (one workgroup per (columns, batch element) in the
// deterministic mode, as before).
__global__ __launch_bounds__(256)
void gated_resid_bwd_kernel(const float* __restrict__ dx, const uint16_t* __restrict__ y, uint16_t* __restrict__ dy,
                            float* __restrict__ dgate, int64_t dgate_stride, int64_t rows, int dim, float gate_const,
                            const float* __restrict__ gate0, const float* __restrict__ gate1, int64_t gate1_stride,
                            int64_t rows_per_batch, int rows_per_block) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;                     // first of this thread's 4 columns (dim % 4 == 0)
    const bool live = c < dim;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t b = r0 / rows_per_batch;
    const int64_t r1 = min(min(rows, r0 + rows_per_block), (b + 1) * rows_per_batch);
    float4 g = make_float4(gate_const, gate_const, gate_const, gate_const);
    if (live && gate0) { const float4 t = *(const float4*)(gate0 + c); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
    if (live && gate1) { const float4 t = *(const float4*)(gate1 + b * gate1_stride + c); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const float4 d = *(const float4*)(dx + r * dim + c);
            *(uint2*)(dy + r * dim + c) = make_uint2(pack_bf2(d.x * g.x, d.y * g.y), pack_bf2(d.z * g.z, d.w * g.w));
            if (y) {
                const uint2 yy = *(const uint2*)(y + r * dim + c);
                acc.x += d.x * __uint_as_float(yy.x << 16); acc.y += d.y * __uint_as_float(yy.x & 0xffff0000u);
                acc.z += d.z * __uint_as_float(yy.y << 16); acc.w += d.w * __uint_as_float(yy.y & 0xffff0000u);
            }
        }
    }
    if (!(dgate && y)) return;                                      // workgroup-uniform
    red[rl][lane] = acc;
    __syncthreads();
    if (rl == 0 && live) {
        float4 t = red[0][lane];
#pragma unroll
        for (int i = 1; i < 4; ++i) { const float4 v = red[i][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float* o = dgate + b * dgate_stride + c;
        atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w);
    }
}

// ------------------------------------------------------------------ LayerNorm + modulate backward
// y = xhat * mul + add ;  dx += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * mul
// dmul[b][c] += dy * xhat ; dadd[b][c] += dy
// One wave walks RPW consecutive rows and keeps the per-column sums in registers, so the
// parameter-side atomics are issued once per RPW rows (they were 2/3 of this kernel's time).
constexpr int RPW = 4;      // rows per wave (measured at 6240 x 1536: 2 -> LN 119 / RMS 54 us, 4 -> 81 / 46, 8 -> 94 / 57)

template <int NV>
__global__ __launch_bounds__(256)
void layernorm_modulate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                   int64_t rows, int dim, float eps, float mul_const, const float* __restrict__ mul0,
                                   const float* __restrict__ mul1, int64_t mul1_stride, float* __restrict__ dmul,
                                   float* __restrict__ dadd, int64_t dstride, int64_t rows_per_batch) {
    // Parameter-side sums: a workgroup whose rows all belong to one batch element first combines its 4 waves
    // in LDS (ds_add_f32) and then issues ONE set of global atomics — at S = 1560 the 390 waves hammering the
    // same 2 x 1536 addresses were 2/3 of this kernel's time.  Workgroups that straddle a batch boundary keep
    // the per-wave flush.
    extern __shared__ float wred[];                                 // [2][dim]
    const int lane = threadIdx.x & 63;
    const int64_t wg_row0 = (int64_t)blockIdx.x * 4 * RPW;
    const int64_t wg_last = min(wg_row0 + 4 * RPW, rows) - 1;
    const bool one_batch = (wg_row0 / rows_per_batch) == (wg_last / rows_per_batch);   // workgroup-uniform
    if (one_batch) {
        for (int i = threadIdx.x; i < 2 * dim; i += 256) wred[i] = 0.f;
        __syncthreads();
    }
    const int64_t row0 = wg_row0 + (int64_t)(threadIdx.x >> 6) * RPW;
    const int nv = dim >> 2;
    const float4* m0 = (const float4*)mul0;
    float4 am[NV], aa[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { am[i] = make_float4(0, 0, 0, 0); aa[i] = make_float4(0, 0, 0, 0); }
    int64_t cur_b = min(row0, rows - 1) / rows_per_batch;

    auto flush = [&](int64_t b) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                if (dmul) {
                    float* d = dmul + b * dstride + 4 * c;
                    atomicAdd(d + 0, am[i].x); atomicAdd(d + 1, am[i].y); atomicAdd(d + 2, am[i].z); atomicAdd(d + 3, am[i].w);
                }
                if (dadd) {
                    float* d = dadd + b * dstride + 4 * c;
                    atomicAdd(d + 0, aa[i].x); atomicAdd(d + 1, aa[i].y); atomicAdd(d + 2, aa[i].z); atomicAdd(d + 3, aa[i].w);
                }
            }
            am[i] = make_float4(0, 0, 0, 0);
            aa[i] = make_float4(0, 0, 0, 0);
        }
    };

    for (int rr = 0; rr < RPW; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= rows) break;
        const int64_t b = row / rows_per_batch;
        if (!one_batch && b != cur_b) { flush(cur_b); cur_b = b; }
        const float4* xr = (const float4*)(x + row * dim);
        const float4* gr = (const float4*)(dy + row * dim);
        const float4* m1 = mul1 ? (const float4*)(mul1 + b * mul1_stride) : nullptr;
        float4 v[NV], g[NV];
        float s = 0.f;
        // x and dy of the row are requested together, before the reductions (latency-bound kernel, see RMSNorm backward)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) { v[i] = xr[c]; g[i] = gr[c]; }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        const float mean = wave_sum(s) / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / dim + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 d = g[i];
                float4 mu = make_float4(mul_const, mul_const, mul_const, mul_const);
                if (m0) { const float4 t = m0[c]; mu.x += t.x; mu.y += t.y; mu.z += t.z; mu.w += t.w; }
                if (m1) { const float4 t = m1[c]; mu.x += t.x; mu.y += t.y; mu.z += t.z; mu.w += t.w; }
                v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;          // xhat
                am[i].x += d.x * v[i].x; am[i].y += d.y * v[i].y; am[i].z += d.z * v[i].z; am[i].w += d.w * v[i].w;
                aa[i].x += d.x; aa[i].y += d.y; aa[i].z += d.z; aa[i].w += d.w;
                g[i] = make_float4(d.x * mu.x, d.y * mu.y, d.z * mu.z, d.w * mu.w);
                sg += g[i].x + g[i].y + g[i].z + g[i].w;
                sgx += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
            }
        }
        const float mg = wave_sum(sg) / dim, mgx = wave_sum(sgx) / dim;
        float4* dxr = (float4*)(dx + row * dim);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o = dxr[c];
                o.x += rstd * (g[i].x - mg - v[i].x * mgx);
                o.y += rstd * (g[i].y - mg - v[i].y * mgx);
                o.z += rstd * (g[i].z - mg - v[i].z * mgx);
                o.w += rstd * (g[i].w - mg - v[i].w * mgx);
                dxr[c] = o;
            }
        }
    }
    if (!one_batch) {
        if (row0 < rows) flush(cur_b);
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv && row0 < rows) {
            atomicAdd(&wred[4 * c + 0], am[i].x); atomicAdd(&wred[4 * c + 1], am[i].y);
            atomicAdd(&wred[4 * c + 2], am[i].z); atomicAdd(&wred[4 * c + 3], am[i].w);
            atomicAdd(&wred[dim + 4 * c + 0], aa[i].x); atomicAdd(&wred[dim + 4 * c + 1], aa[i].y);
            atomicAdd(&wred[dim + 4 * c + 2], aa[i].z); atomicAdd(&wred[dim + 4 * c + 3], aa[i].w);
        }
    }
    __syncthreads();
    const int64_t bo = (wg_row0 / rows_per_batch) * dstride;
    for (int i = threadIdx.x; i < dim; i += 256) {
        if (dmul) atomicAdd(dmul + bo + i, wred[i]);
        if (dadd) atomicAdd(dadd + bo + i, wred[dim + i]);
    }
}

// ------------------------------------------------------------------ RMSNorm (+RoPE) backward
// forward: y = rope( x * r * w ), r = rsqrt(mean(x^2)+eps).  g = unrope(dy);
// dw[c] += g*x*r ; dx = r*(g*w) - x * r^3 * mean(x * g*w)   -> bf16
// 4 consecutive elements of a row as fp32, from an fp32 or a bf16 tensor
template <typename T> __device__ __forceinline__ float4 ld4(const T* row, int c);
template <> __device__ __forceinline__ float4 ld4<float>(const float* row, int c) { return ((const float4*)row)[c]; }
template <> __device__ __forceinline__ float4 ld4<uint16_t>(const uint16_t* row, int c) {
    const uint2 u = ((const uint2*)row)[c];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <int NV, typename XT, typename GT>
__global__ __launch_bounds__(256)
void rmsnorm_rope_bwd_kernel(const XT* __restrict__ x, int64_t ldx, const GT* __restrict__ dy, int64_t lddy,
                             uint16_t* __restrict__ dx, int64_t lddx, float* __restrict__ dw, int64_t rows, int dim,
                             const float* __restrict__ weight, float eps, int do_norm,
                             const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, int rope_len,
                             int head_dim, const int* __restrict__ grid, int seq_len) {
    extern __shared__ float wred[];                                 // [dim]: the 4 waves' dw partials (see LN backward)
    const int lane = threadIdx.x & 63;
    if (dw) {
        for (int i = threadIdx.x; i < dim; i += 256) wred[i] = 0.f;
        __syncthreads();
    }
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    const int nv = dim >> 2;
    const float4* wv = (const float4*)weight;
    const int hc = head_dim >> 1, c3 = hc / 3, cf = hc - 2 * c3;
    float4 aw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) aw[i] = make_float4(0, 0, 0, 0);

    for (int rr = 0; rr < RPW; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= rows) break;
        const XT* xr = x + row * ldx;
        const GT* gr = dy + row * lddy;
        float4 v[NV], g[NV];
        float q = 0.f;
        // both operands of the row are requested before the first reduction: with ~6 waves per CU at S = 1560 the
        // kernel is latency bound, and dy used to be fetched only after wave_sum(x^2) had come back
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) { v[i] = ld4<XT>(xr, c); g[i] = ld4<GT>(gr, c); }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        const float r = do_norm ? rsqrtf(wave_sum(q) / dim + eps) : 1.0f;
        bool rot = false;
        int pf = 0, ph = 0, pw = 0;
        if (rope_cos) {
            const int b = (int)(row / seq_len), s = (int)(row % seq_len);
            const int gf = grid[3 * b], gh = grid[3 * b + 1], gw = grid[3 * b + 2];
            if (s < gf * gh * gw) { rot = true; pf = s / (gh * gw); ph = (s / gw) % gh; pw = s % gw; }
        }
        float sxg = 0.f;
        // as in rmsnorm_rope_kernel: with head_dim | 256 a lane's vectors all use the same two table entries
        const bool same_pairs = rot && (256 % head_dim) == 0;
        float cs2[2] = {1.f, 1.f}, sn2[2] = {0.f, 0.f};
        if (same_pairs) {
            const int p0 = ((4 * lane) % head_dim) >> 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pc = p0 + e;
                const int pos = pc < cf ? pf : (pc < cf + c3 ? ph : pw);
                const int idx = min(pos, rope_len - 1) * hc + pc;
                cs2[e] = rope_cos[idx]; sn2[e] = rope_sin[idx];
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 t = g[i];
                if (rot) {
                    const int p0 = ((4 * c) % head_dim) >> 1;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pc = p0 + e;
                        const int pos = pc < cf ? pf : (pc < cf + c3 ? ph : pw);
                        const int idx = min(pos, rope_len - 1) * hc + pc;
                        const float cs = same_pairs ? cs2[e] : rope_cos[idx], sn = same_pairs ? sn2[e] : rope_sin[idx];
                        float& re = e == 0 ? t.x : t.z;
                        float& im = e == 0 ? t.y : t.w;
                        const float nr = re * cs + im * sn;          // rotate by -theta
                        const float ni = -re * sn + im * cs;
                        re = nr; im = ni;
                    }
                }
                aw[i].x += t.x * v[i].x * r; aw[i].y += t.y * v[i].y * r;
                aw[i].z += t.z * v[i].z * r; aw[i].w += t.w * v[i].w * r;
                if (wv) { const float4 w4 = wv[c]; t.x *= w4.x; t.y *= w4.y; t.z *= w4.z; t.w *= w4.w; }
                g[i] = t;
                sxg += v[i].x * t.x + v[i].y * t.y + v[i].z * t.z + v[i].w * t.w;
            }
        }
        const float coef = do_norm ? wave_sum(sxg) / dim * r * r * r : 0.f;
        uint2* dxr = (uint2*)(dx + row * lddx);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                uint2 o;
                o.x = pack_bf2(r * g[i].x - v[i].x * coef, r * g[i].y - v[i].y * coef);
                o.y = pack_bf2(r * g[i].z - v[i].z * coef, r * g[i].w - v[i].w * coef);
                dxr[c] = o;
            }
        }
    }
    if (dw) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv && row0 < rows) {
                atomicAdd(&wred[4 * c + 0], aw[i].x); atomicAdd(&wred[4 * c + 1], aw[i].y);
                atomicAdd(&wred[4 * c + 2], aw[i].z); atomicAdd(&wred[4 * c + 3], aw[i].w);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < dim; i += 256) atomicAdd(dw + i, wred[i]);
    }
}

// ------------------------------------------------------------------ softmax backward
// dS = P * (dP - sum_j P*dP) * scale ; one 256-thread block per row
__global__ __launch_bounds__(256)
void softmax_bwd_rows_kernel(const uint16_t* __restrict__ p, int64_t ldp, const float* __restrict__ dp, int64_t lddp,
                             uint16_t* __restrict__ ds, int64_t ldds, int L, float scale) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const uint16_t* pr = p + r * ldp;
    const float* dr = dp + r * lddp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int j = tid; j < L; j += 256) s += bf2f(pr[j]) * dr[j];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float delta = red[0] + red[1] + red[2] + red[3];
    uint16_t* o = ds + r * ldds;
    for (int j = tid; j < L; j += 256) o[j] = f2bf(bf2f(pr[j]) * (dr[j] - delta) * scale);
}

// ------------------------------------------------------------------ unpatchify backward
// dtok[s][((a*ph+i)*pw+j)*C + c] = g[c][f*pt+a][h*ph+i][w*pw+j]   (bf16 out)
__global__ __launch_bounds__(256)
void unpatchify_bwd_kernel(const float* __restrict__ g, uint16_t* __restrict__ dtok, int Cout, int gf, int gh, int gw,
                           int pt, int ph, int pw) {
    const int F = gf * pt, H = gh * ph, W = gw * pw;
    const int ncol = pt * ph * pw * Cout;
    const int64_t total = (int64_t)gf * gh * gw * ncol;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int col = (int)(i % ncol);
        const int64_t s = i / ncol;
        const int c = col % Cout, j = (col / Cout) % pw, ii = (col / (Cout * pw)) % ph, a = col / (Cout * pw * ph);
        const int w = (int)(s % gw), h = (int)((s / gw) % gh), f = (int)(s / ((int64_t)gw * gh));
        dtok[i] = f2bf(g[(((int64_t)c * F + f * pt + a) * H + h * ph + ii) * W + w * pw + j]);
    }
}

// ------------------------------------------------------------------ tiny fp32 dense backward (time embedding)
// forward: y[b][n] = sum_k act(x[b][k]) W[n][k] + bias[n]
__global__ __launch_bounds__(256)
void dense_f32_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dW,
                            float* __restrict__ db, int B, int N, int K, int act_in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        float xv = x[(int64_t)b * K + k];
        if (act_in == 1) xv = xv / (1.0f + expf(-xv));
        const float d = dy[(int64_t)b * N + n];
        s += d * xv;
        sb += d;
    }
    dW[i] += s;
    if (db && k == 0) db[n] += sb;
}

// dx[b][k] (+)= act'(x[b][k]) * sum_n dy[b][n] W[n][k]; the n range is split over blockIdx.y and
// combined with atomics (dx must be zero-filled by the caller unless it accumulates).
__global__ __launch_bounds__(256)
void dense_f32_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ dy,
                            float* __restrict__ dx, int B, int N, int K, int act_in, int n_chunk) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * K) return;
    const int b = (int)(i / K), k = (int)(i % K);
    const int n0 = blockIdx.y * n_chunk, n1 = min(N, n0 + n_chunk);
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s += dy[(int64_t)b * N + n] * W[(int64_t)n * K + k];
    if (act_in == 1) {
        const float xv = x[i];
        const float sg = 1.0f / (1.0f + expf(-xv));
        s *= sg * (1.0f + xv * (1.0f - sg));
    }
    atomicAdd(dx + i, s);
}

inline int grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

extern "C" int omh_transpose_bf16(const void* in, void* out, int32_t R, int32_t C, int64_t ld_in, int64_t ld_out,
                                  int32_t batch, int64_t bs_in, int64_t bs_out, omh_stream_t stream) {
    if (!in || !out || R <= 0 || C <= 0 || batch <= 0 || ld_in < C || ld_out < R) return OMH_E_BADARG;
    dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
    omh_clear_status();
    const bool vec = !((ld_in | ld_out | bs_in | bs_out) & 3) && !(((uintptr_t)in | (uintptr_t)out) & 7);
    if (vec)
        hipLaunchKernelGGL(transpose_bf16_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in,
                           (uint16_t*)out, R, C, ld_in, ld_out, bs_in, bs_out);
    else
        hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in,
                           (uint16_t*)out, R, C, ld_in, ld_out, bs_in, bs_out);
    return omh_launch_status();
}

extern "C" int omh_colsum_accum(const void* x, int32_t is_bf16, int64_t ld, float* out, int64_t R, int32_t C,
                                omh_stream_t stream) {
    if (!x || !out || R <= 0 || C <= 0 || ld < C) return OMH_E_BADARG;
    if (omh_deterministic() && R > 0x7fffffff) return OMH_E_SHAPE;
    const int rpb = omh_deterministic() ? (int)R : 32;               // deterministic: one block adds a column's rows in order
    dim3 grid((C + 255) / 256, (unsigned)((R + rpb - 1) / rpb));
    omh_clear_status();
    if (is_bf16)
        hipLaunchKernelGGL(colsum_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ld,
                           out, R, C, rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, out, R,
                           C, rpb);
    return omh_launch_status();
}

extern "C" int omh_colsum_accum_multi(const omh_colsum_batch* batch, omh_stream_t stream) {
    if (!batch || batch->n <= 0 || batch->n > OMH_COLSUM_MAX) return OMH_E_BADARG;
    omh_colsum_batch b = *batch;
    int64_t total = 0;
    bool vec = true;                                                 // 16-byte loads: every matrix must allow them
    for (int i = 0; i < b.n; ++i) {
        if (!b.x[i] || !b.out[i] || b.R[i] <= 0 || b.C[i] <= 0 || b.ld[i] < b.C[i]) return OMH_E_BADARG;
        vec = vec && (b.C[i] & 7) == 0 && (b.ld[i] & 7) == 0 && ((uintptr_t)b.x[i] & 15) == 0;
    }
    const int rpb = omh_deterministic() ? 0x40000000 : (vec ? 128 : 32);    // deterministic: one row block per column chunk
    for (int i = 0; i < b.n; ++i) {
        b.blocks[i] = (int32_t)(((b.C[i] + 255) / 256) * ((b.R[i] + rpb - 1) / rpb));
        total += b.blocks[i];
    }
    if (total > 0x7fffffff) return OMH_E_SHAPE;
    omh_clear_status();
    if (vec) hipLaunchKernelGGL(colsum_multi_vec_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, b, rpb);
    else hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, b, rpb);
    return omh_launch_status();
}

// exact (erf) GELU of the i2v image-embedding MLP (model.py:366): forward and derivative
//   d/dz [ z Phi(z) ] = Phi(z) + z phi(z),  Phi(z) = 0.5 (1 + erf(z / sqrt 2)),  phi(z) = exp(-z^2/2) / sqrt(2 pi)
__global__ __launch_bounds__(256)
void gelu_erf_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float z = bf2f(x[i]);
        y[i] = f2bf(0.5f * z * (1.0f + erff(z * 0.7071067811865476f)));
    }
}
__global__ __launch_bounds__(256)
void gelu_erf_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ xpre, uint16_t* __restrict__ dx,
                         int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float z = bf2f(xpre[i]);
        const float g = 0.5f * (1.0f + erff(z * 0.7071067811865476f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
        dx[i] = f2bf(bf2f(dy[i]) * g);
    }
}

extern "C" int omh_gelu_erf_bf16(const void* x, void* y, int64_t n, omh_stream_t stream) {
    if (!x || !y || n <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(gelu_erf_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, n);
    return omh_launch_status();
}

extern "C" int omh_gelu_erf_bwd_bf16(const void* dy, const void* x_pre, void* dx, int64_t n, omh_stream_t stream) {
    if (!dy || !x_pre || !dx || n <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(gelu_erf_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dy, (const uint16_t*)x_pre, (uint16_t*)dx, n);
    return omh_launch_status();
}

extern "C" int omh_gelu_tanh_bf16(const void* x, void* y, int64_t n, omh_stream_t stream) {
    if (!x || !y || n <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, n);
    return omh_launch_status();
}

extern "C" int omh_gelu_tanh_bwd_bf16(const void* dy, const void* x_pre, void* dx, int64_t n, omh_stream_t stream) {
    if (!dy || !x_pre || !dx || n <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dy, (const uint16_t*)x_pre, (uint16_t*)dx, n);
    return omh_launch_status();
}

extern "C" int omh_gated_residual_fwd(const float* xi, const void* y_bf16, float* xo, int64_t rows, int32_t dim,
                                      float gate_const, const float* gate0, const float* gate1, int64_t gate1_stride,
                                      int64_t rows_per_batch, omh_stream_t stream) {
    if (!xi || !y_bf16 || !xo || rows <= 0 || dim <= 0 || rows_per_batch <= 0) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(gated_resid_fwd_kernel, dim3(grid_for(rows * dim, 256)), dim3(256), 0, (hipStream_t)stream, xi,
                       (const uint16_t*)y_bf16, xo, rows, dim, gate_const, gate0, gate1, gate1_stride, rows_per_batch);
    return omh_launch_status();
}

extern "C" int omh_gated_residual_bwd(const float* dx, const void* y_bf16, void* dy_bf16, float* dgate,
                                      int64_t dgate_stride, int64_t rows, int32_t dim, float gate_const,
                                      const float* gate0, const float* gate1, int64_t gate1_stride,
                                      int64_t rows_per_batch, omh_stream_t stream) {
    if (!dx || !dy_bf16 || rows <= 0 || dim <= 0 || rows_per_batch <= 0) return OMH_E_BADARG;
    int rpb = 32;
    while (rows_per_batch % rpb) --rpb;              // blocks never straddle two batch elements (1 560 rows: 30 per block)
    if (omh_deterministic() && dgate && rows_per_batch <= 0x7fffffff) rpb = (int)rows_per_batch;   // one adder per dgate element
    if ((dim & 3) || ((uintptr_t)dx & 15) || ((uintptr_t)dy_bf16 & 7) || ((uintptr_t)y_bf16 & 7) || (gate1_stride & 3) ||
        ((uintptr_t)gate0 & 15) || ((uintptr_t)gate1 & 15))
        return OMH_E_ALIGN;
    dim3 grid((dim + 255) / 256, (unsigned)((rows + rpb - 1) / rpb));
    omh_clear_status();
    hipLaunchKernelGGL(gated_resid_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dx, (const uint16_t*)y_bf16,
                       (uint16_t*)dy_bf16, dgate, dgate_stride, rows, dim, gate_const, gate0, gate1, gate1_stride,
                       rows_per_batch, rpb);
    return omh_launch_status();
}

extern "C" int omh_layernorm_modulate_bwd(const float* x, const float* dy, float* dx_accum, int64_t rows, int32_t dim,
                                          float eps, float mul_const, const float* mul0, const float* mul1,
                                          int64_t mul1_stride, float* dmul, float* dadd, int64_t dstride,
                                          int64_t rows_per_batch, omh_stream_t stream) {
    if (!x || !dy || !dx_accum || rows <= 0 || dim <= 0 || rows_per_batch <= 0) return OMH_E_BADARG;
    if ((dim & 3) || dim > MAXV * 256) return OMH_E_SHAPE;
    omh_clear_status();
    auto kern = dim <= 6 * 256 ? layernorm_modulate_bwd_kernel<6>
                               : (dim <= 20 * 256 ? layernorm_modulate_bwd_kernel<20> : layernorm_modulate_bwd_kernel<MAXV>);
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 4 * RPW - 1) / (4 * RPW))), dim3(256), 2 * dim * sizeof(float),
                       (hipStream_t)stream, x, dy, dx_accum, rows, dim, eps, mul_const, mul0, mul1, mul1_stride, dmul,
                       dadd, dstride, rows_per_batch);
    return omh_launch_status();
}

template <typename XT, typename GT>
static int rmsnorm_rope_bwd_launch(const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx_bf16, int64_t lddx,
                                   float* dweight, int64_t rows, int32_t dim, const float* weight, float eps,
                                   int32_t do_norm, const float* rope_cos, const float* rope_sin, int32_t rope_len,
                                   int32_t head_dim, const int32_t* grid, int32_t seq_len, omh_stream_t stream) {
    if (!x || !dy || !dx_bf16 || rows <= 0 || dim <= 0) return OMH_E_BADARG;
    if ((dim & 3) || dim > MAXV * 256 || (ldx & 3) || (lddy & 3) || (lddx & 3)) return OMH_E_SHAPE;
    if (rope_cos && (!rope_sin || !grid || seq_len <= 0 || head_dim <= 0)) return OMH_E_BADARG;
    omh_clear_status();
    auto kern = dim <= 6 * 256 ? rmsnorm_rope_bwd_kernel<6, XT, GT>
                               : (dim <= 20 * 256 ? rmsnorm_rope_bwd_kernel<20, XT, GT> : rmsnorm_rope_bwd_kernel<MAXV, XT, GT>);
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 4 * RPW - 1) / (4 * RPW))), dim3(256), dim * sizeof(float),
                       (hipStream_t)stream, (const XT*)x, ldx, (const GT*)dy, lddy, (uint16_t*)dx_bf16, lddx, dweight, rows,
                       dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid, seq_len);
    return omh_launch_status();
}

extern "C" int omh_rmsnorm_rope_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, void* dx_bf16,
                                    int64_t lddx, float* dweight, int64_t rows, int32_t dim, const float* weight,
                                    float eps, int32_t do_norm, const float* rope_cos, const float* rope_sin,
                                    int32_t rope_len, int32_t head_dim, const int32_t* grid, int32_t seq_len,
                                    omh_stream_t stream) {
    return rmsnorm_rope_bwd_launch<float, float>(x, ldx, dy, lddy, dx_bf16, lddx, dweight, rows, dim, weight, eps, do_norm,
                                                 rope_cos, rope_sin, rope_len, head_dim, grid, seq_len, stream);
}

extern "C" int omh_rmsnorm_rope_bwd_t(const void* x, int32_t x_bf16, int64_t ldx, const void* dy, int32_t dy_bf16,
                                      int64_t lddy, void* dx_bf16, int64_t lddx, float* dweight, int64_t rows, int32_t dim,
                                      const float* weight, float eps, int32_t do_norm, const float* rope_cos,
                                      const float* rope_sin, int32_t rope_len, int32_t head_dim, const int32_t* grid,
                                      int32_t seq_len, omh_stream_t stream) {
#define OMH_RRB(XT, GT)                                                                                               \
    return rmsnorm_rope_bwd_launch<XT, GT>(x, ldx, dy, lddy, dx_bf16, lddx, dweight, rows, dim, weight, eps, do_norm,  \
                                           rope_cos, rope_sin, rope_len, head_dim, grid, seq_len, stream)
    if (x_bf16 && dy_bf16) OMH_RRB(uint16_t, uint16_t);
    if (x_bf16) OMH_RRB(uint16_t, float);
    if (dy_bf16) OMH_RRB(float, uint16_t);
    OMH_RRB(float, float);
#undef OMH_RRB
}

extern "C" int omh_softmax_bwd_rows(const void* p_bf16, int64_t ldp, const float* dp, int64_t lddp, void* ds_bf16,
                                    int64_t ldds, int64_t R, int32_t L, float scale, omh_stream_t stream) {
    if (!p_bf16 || !dp || !ds_bf16 || R <= 0 || L <= 0 || R > 0x7fffffff) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)p_bf16, ldp, dp, lddp, (uint16_t*)ds_bf16, ldds, L, scale);
    return omh_launch_status();
}

extern "C" int omh_unpatchify_bwd(const float* g, void* dtok_bf16, int32_t Cout, int32_t f, int32_t h, int32_t w,
                                  int32_t pt, int32_t ph, int32_t pw, omh_stream_t stream) {
    if (!g || !dtok_bf16 || Cout <= 0 || f <= 0 || h <= 0 || w <= 0) return OMH_E_BADARG;
    const int64_t total = (int64_t)f * h * w * pt * ph * pw * Cout;
    omh_clear_status();
    hipLaunchKernelGGL(unpatchify_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       (uint16_t*)dtok_bf16, Cout, f, h, w, pt, ph, pw);
    return omh_launch_status();
}

extern "C" int omh_dense_f32_bwd(const float* x, const float* W, const float* dy, float* dW_accum, float* db_accum,
                                 float* dx, int32_t dx_accumulate, int32_t B, int32_t N, int32_t K, int32_t act_in,
                                 omh_stream_t stream) {
    if (!x || !W || !dy || B <= 0 || N <= 0 || K <= 0) return OMH_E_BADARG;
    omh_clear_status();
    if (dW_accum)
        hipLaunchKernelGGL(dense_f32_bwd_w_kernel, dim3((unsigned)(((int64_t)N * K + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x, dy, dW_accum, db_accum, B, N, K, act_in);
    if (dx) {
        if (!dx_accumulate) omh_zero_f32(dx, 1, (int64_t)B * K, (int64_t)B * K, (hipStream_t)stream);
        const int n_chunk = omh_deterministic() ? N : 64;
        dim3 grid((unsigned)(((int64_t)B * K + 255) / 256), (unsigned)((N + n_chunk - 1) / n_chunk));
        hipLaunchKernelGGL(dense_f32_bwd_x_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, W, dy, dx, B, N, K,
                           act_in, n_chunk);
    }
    return omh_launch_status();
}
