// Round-3 backward kernels of the two norms of the DiT block (autograd of model.py:292-293,313-315 and 144-145 under
// distilled_trainer.py:289-301), rebuilt for bandwidth and for repeatable results:
//
//   * round 2's kernels gave a wave 4 consecutive rows (390 workgroups at 6 240 rows: ~6 waves per CU, the loads
//     of one row at a time in flight) and added the per-column parameter sums with fp32 atomics, first in LDS then
//     in global memory: 117 us for 153 MB (1.3 TB/s) and a result whose last bits changed from run to run.
//   * here a workgroup takes ROWS_PER_WG rows of ONE batch element, interleaved over its 4 waves, every row's
//     operands requested up front; the per-column sums stay in registers, are combined across the waves through
//     LDS in a FIXED order, and each workgroup writes its partial to a workspace; a second small launch adds the
//     partials of a column in workgroup order.  No atomics anywhere: bit-repeatable.
//   * the LayerNorm kernel can also apply the gated-residual backward of the NEXT branch to the row it has just
//     finished (dy_next = bf16(dx * gate), dgate += dx * y): one pass over dx instead of two.
//   * the RMSNorm kernel takes one or two column segments per launch (q and k of the self-attention).
#include "omh_common.h"

namespace {

constexpr int MAXV2 = 32;
constexpr int RPW2 = 2;                    // rows per wave (RMSNorm kernel; LayerNorm kernel on short inputs)
constexpr int ROWS_PER_WG = 4 * RPW2;
// The LayerNorm kernel takes 4 rows per wave from 4 096 rows on: 390 workgroups at 6 240 rows are ONE round of the 512
// resident ones (2 per CU at 204 VGPRs) where 780 were a round and a half, and half as many partials go through the
// column-sum launch.  Measured at 6 240 rows (us, 2 -> 4 rows per wave): 50.3 -> 45.3 plain, 54.2 -> 48.8 with the next
// branch's residual backward, 70.9 -> 63.6 with its gate gradient too; the RMSNorm kernel (104 VGPRs, 4 waves per SIMD)
// does not gain (two segments 45.4 -> 43.5, one segment 23.1 -> 28.5).
constexpr int RPW_LN_LONG = 4, LN_LONG_ROWS = 4096;
static inline int ln_rows_per_wg(int64_t rows) { return 4 * (rows >= LN_LONG_ROWS ? RPW_LN_LONG : RPW2); }

template <typename T> __device__ __forceinline__ float4 ld4t(const T* row, int c);
template <> __device__ __forceinline__ float4 ld4t<float>(const float* row, int c) { return ((const float4*)row)[c]; }
template <> __device__ __forceinline__ float4 ld4t<uint16_t>(const uint16_t* row, int c) {
    const uint2 u = ((const uint2*)row)[c];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void acc4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// Combine NP register partials (float4 per column vector) of the 4 waves in the order 3, 2, 1, 0 through one LDS
// buffer [NP][dim] and let wave 0 store the result: part[k * dim + c].
template <int NV, int NP>
__device__ __forceinline__ void combine_and_store(float4 (&acc)[NP][NV], float* lds, float* __restrict__ part, int dim,
                                                  int nv, int lane, int wave) {
#pragma unroll 1
    for (int w = 3; w >= 0; --w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = lane + 64 * i;
                    if (c < nv) {
                        float4 v = acc[k][i];
                        if (w != 3) acc4(v, ((const float4*)(lds + k * dim))[c]);
                        if (w != 0) ((float4*)(lds + k * dim))[c] = v;
                        else ((float4*)(part + (int64_t)k * dim))[c] = v;
                    }
                }
        }
        if (w != 0) __syncthreads();
    }
}

// ------------------------------------------------------------------ LayerNorm + modulate backward (+ next residual)
template <int NV, typename GT, bool NEXT, bool GATE, int RPW>
__global__ __launch_bounds__(256)
void ln_bwd2_kernel(const omh_ln_bwd_args a, const int nj) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NP = 2 + (GATE ? 1 : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x, b = blockIdx.y;
    const int dim = a.dim, nv = dim >> 2;
    const int64_t rpb = a.rows_per_batch;
    const int64_t base = (int64_t)b * rpb + (int64_t)j * (4 * RPW);
    const int64_t end = min((int64_t)(b + 1) * rpb, a.rows);
    const float4* m0 = (const float4*)a.mul0;
    const float4* m1 = a.mul1 ? (const float4*)(a.mul1 + (int64_t)b * a.mul1_stride) : nullptr;
    const float4* g0 = (const float4*)a.gate0;
    const float4* g1 = a.gate1 ? (const float4*)(a.gate1 + (int64_t)b * a.gate1_stride) : nullptr;
    float4 acc[NP][NV];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[k][i] = make_float4(0.f, 0.f, 0.f, 0.f);

#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int64_t row = base + wave + 4 * rr;
        if (row >= end) continue;                                       // wave-uniform
        const float* xr = a.x + row * dim;
        const GT* gr = (const GT*)a.dy + row * dim;
        float* dxr = a.dx + row * dim;
        float4 v[NV], g[NV], o[NV], yn[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[i] = ((const float4*)xr)[c];
                g[i] = ld4t<GT>(gr, c);
                o[i] = ((const float4*)dxr)[c];
                if (GATE) yn[i] = ld4t<uint16_t>((const uint16_t*)a.y_next + row * dim, c);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) if (lane + 64 * i < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
        const float mean = wave_sum(s) / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            }
        const float rstd = rsqrtf(wave_sum(q) / dim + a.eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 d = g[i];
                float4 mu = make_float4(a.mul_const, a.mul_const, a.mul_const, a.mul_const);
                if (m0) acc4(mu, m0[c]);
                if (m1) acc4(mu, m1[c]);
                v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;          // xhat
                acc[0][i].x += d.x * v[i].x; acc[0][i].y += d.y * v[i].y; acc[0][i].z += d.z * v[i].z; acc[0][i].w += d.w * v[i].w;
                acc4(acc[1][i], d);
                g[i] = make_float4(d.x * mu.x, d.y * mu.y, d.z * mu.z, d.w * mu.w);
                sg += g[i].x + g[i].y + g[i].z + g[i].w;
                sgx += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
            }
        }
        const float mg = wave_sum(sg) / dim, mgx = wave_sum(sgx) / dim;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 r = o[i];
                r.x += rstd * (g[i].x - mg - v[i].x * mgx);
                r.y += rstd * (g[i].y - mg - v[i].y * mgx);
                r.z += rstd * (g[i].z - mg - v[i].z * mgx);
                r.w += rstd * (g[i].w - mg - v[i].w * mgx);
                ((float4*)dxr)[c] = r;
                if (NEXT) {                                             // the next branch's gated-residual backward
                    float4 gt = make_float4(a.gate_const, a.gate_const, a.gate_const, a.gate_const);
                    if (g0) acc4(gt, g0[c]);
                    if (g1) acc4(gt, g1[c]);
                    ((uint2*)((uint16_t*)a.dy_next + row * dim))[c] =
                        make_uint2(pack_bf2(r.x * gt.x, r.y * gt.y), pack_bf2(r.z * gt.z, r.w * gt.w));
                    if (GATE) {
                        acc[2][i].x += r.x * yn[i].x; acc[2][i].y += r.y * yn[i].y;
                        acc[2][i].z += r.z * yn[i].z; acc[2][i].w += r.w * yn[i].w;
                    }
                }
            }
        }
    }
    combine_and_store<NV, NP>(acc, lds, a.workspace + ((int64_t)b * nj + j) * NP * dim, dim, nv, lane, wave);
}

// out_k[b * stride_k + c] += sum_j part[((b * nj + j) * NP + k) * dim + c]   in a FIXED order: a workgroup takes 64
// columns (16 float4 lanes) x 16 slices of j; a thread adds its slice's partials (j = slice, slice + 16, ...), the 16
// slices are then added in slice order through LDS.
// stride_k == 0: the parameter is shared by all nb row groups — the b == 0 workgroups add all nb * nj partials.
__device__ __forceinline__
void partial_colsum_body(const float* __restrict__ part, int nj, int np, int nb, int dim, float* o0, int64_t s0,
                         float* o1, int64_t s1, float* o2, int64_t s2, const int by) {
    __shared__ float4 red[16][16];
    const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 16 + cl;                            // float4 column
    const int k = by % np, b = by / np;
    float* out = k == 0 ? o0 : (k == 1 ? o1 : o2);
    const int64_t st = k == 0 ? s0 : (k == 1 ? s1 : s2);
    if (!out || (st == 0 && b != 0)) return;                        // workgroup-uniform
    const int n = st == 0 ? nb * nj : nj;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * c4 < dim) {
        const float4* p = (const float4*)(part + (((int64_t)b * nj) * np + k) * dim) + c4;
        const int64_t step = (int64_t)np * dim / 4;
        // eight loads in flight per thread, added in the same ascending order as the plain loop (same bits): the loop
        // was latency-bound — 49 dependent trips of ~0.3 us at 6 240 rows (14.7 us per launch, 162 launches per step)
        int j = slice;
        for (; j + 16 * 7 < n; j += 16 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(j + 16 * u) * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; j < n; j += 16) {
            const float4 v = p[(int64_t)j * step];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[slice][cl] = s;
    __syncthreads();
    if (slice == 0 && 4 * c4 < dim) {
        float4 t = red[0][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) { const float4 v = red[i][cl]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float4* o = (float4*)(out + (int64_t)b * st) + c4;
        float4 w = *o;
        w.x += t.x; w.y += t.y; w.z += t.z; w.w += t.w;
        *o = w;
    }
}

__global__ __launch_bounds__(256)
void partial_colsum_kernel(const float* __restrict__ part, int nj, int np, int nb, int dim, float* o0, int64_t s0,
                           float* o1, int64_t s1, float* o2, int64_t s2) {
    partial_colsum_body(part, nj, np, nb, dim, o0, s0, o1, s1, o2, s2, (int)blockIdx.y);
}

// several deferred second stages in one launch (blockIdx.z = entry): the same body, workgroups outside an entry's own
// grid leave at once (workgroup-uniform)
__global__ __launch_bounds__(256)
void partial_colsum_multi_kernel(const omh_partial_reduce_batch bt) {
    const omh_partial_reduce& e = bt.e[blockIdx.z];
    if ((int)blockIdx.y >= e.grid_y || (int)blockIdx.x * 64 >= e.dim) return;
    partial_colsum_body(e.part, e.nj, e.np, e.nb, e.dim, e.out[0], e.stride[0], e.out[1], e.stride[1], e.out[2], e.stride[2],
                        (int)blockIdx.y);
}

// ------------------------------------------------------------------ RMSNorm (+RoPE) backward, 1 or 2 column segments
// forward: y = rope( x * r * w ), r = rsqrt(mean(x^2)+eps).  g = unrope(dy);
// dw[c] += g*x*r ; dx = r*(g*w) - x * r^3 * mean(x * g*w)   -> bf16 (may alias dy)
template <int NV, typename XT, typename GT>
__global__ __launch_bounds__(256)
void rms_bwd2_kernel(const omh_rms_bwd_args a, const int nj) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x, seg = blockIdx.y;
    const int dim = a.dim, nv = dim >> 2;
    const XT* X = (const XT*)a.x + (int64_t)seg * a.seg_x;
    const GT* G = (const GT*)a.dy + (int64_t)seg * a.seg_dy;
    uint16_t* DX = (uint16_t*)a.dx + (int64_t)seg * a.seg_dx;
    const float4* wv = a.weight[seg] ? (const float4*)a.weight[seg] : nullptr;
    const int hd = a.head_dim, hc = hd >> 1, c3 = hc / 3, cf = hc - 2 * c3;
    float4 acc[1][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[0][i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int rr = 0; rr < RPW2; ++rr) {
        const int64_t row = (int64_t)j * ROWS_PER_WG + wave + 4 * rr;
        if (row >= a.rows) continue;
        float4 v[NV], g[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) { v[i] = ld4t<XT>(X + row * a.ldx, c); g[i] = ld4t<GT>(G + row * a.lddy, c); }
        }
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        const float r = a.do_norm ? rsqrtf(wave_sum(q) / dim + a.eps) : 1.0f;
        bool rot = false;
        int pf = 0, ph = 0, pw = 0;
        if (a.rope_cos) {
            const int bb = (int)(row / a.seq_len), s = (int)(row % a.seq_len);
            const int gf = a.grid[3 * bb], gh = a.grid[3 * bb + 1], gw = a.grid[3 * bb + 2];
            if (s < gf * gh * gw) { rot = true; pf = s / (gh * gw); ph = (s / gw) % gh; pw = s % gw; }
        }
        const bool same_pairs = rot && (256 % hd) == 0;           // a lane's vectors all use the same two table entries
        float cs2[2] = {1.f, 1.f}, sn2[2] = {0.f, 0.f};
        if (same_pairs) {
            const int p0 = ((4 * lane) % hd) >> 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pc = p0 + e;
                const int pos = pc < cf ? pf : (pc < cf + c3 ? ph : pw);
                const int idx = min(pos, a.rope_len - 1) * hc + pc;
                cs2[e] = a.rope_cos[idx]; sn2[e] = a.rope_sin[idx];
            }
        }
        float sxg = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 t = g[i];
                if (rot) {
                    const int p0 = ((4 * c) % hd) >> 1;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float cs = cs2[e], sn = sn2[e];
                        if (!same_pairs) {
                            const int pc = p0 + e;
                            const int pos = pc < cf ? pf : (pc < cf + c3 ? ph : pw);
                            const int idx = min(pos, a.rope_len - 1) * hc + pc;
                            cs = a.rope_cos[idx]; sn = a.rope_sin[idx];
                        }
                        float& re = e == 0 ? t.x : t.z;
                        float& im = e == 0 ? t.y : t.w;
                        const float nr = re * cs + im * sn;          // rotate by -theta
                        const float ni = -re * sn + im * cs;
                        re = nr; im = ni;
                    }
                }
                acc[0][i].x += t.x * v[i].x * r; acc[0][i].y += t.y * v[i].y * r;
                acc[0][i].z += t.z * v[i].z * r; acc[0][i].w += t.w * v[i].w * r;
                if (wv) { const float4 w4 = wv[c]; t.x *= w4.x; t.y *= w4.y; t.z *= w4.z; t.w *= w4.w; }
                g[i] = t;
                sxg += v[i].x * t.x + v[i].y * t.y + v[i].z * t.z + v[i].w * t.w;
            }
        }
        const float coef = a.do_norm ? wave_sum(sxg) / dim * r * r * r : 0.f;
        uint2* dxr = (uint2*)(DX + row * a.lddx);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv)
                dxr[c] = make_uint2(pack_bf2(r * g[i].x - v[i].x * coef, r * g[i].y - v[i].y * coef),
                                    pack_bf2(r * g[i].z - v[i].z * coef, r * g[i].w - v[i].w * coef));
        }
    }
    if (a.dweight[seg])                                               // workgroup-uniform
        combine_and_store<NV, 1>(acc, lds, a.workspace + ((int64_t)seg * nj + j) * dim, dim, nv, lane, wave);
}

template <int NV>
int launch_ln(const omh_ln_bwd_args& a, int nj, int nb, hipStream_t s) {
    const bool next = a.dy_next != nullptr, gate = next && a.y_next && a.dgate;
    const dim3 grid((unsigned)nj, (unsigned)nb), blk(256);
    const size_t lds = (size_t)(gate ? 3 : 2) * a.dim * sizeof(float);
#define OMH_LN2(GT, NEXT, GATE)                                                                                       \
    do {                                                                                                              \
        if (ln_rows_per_wg(a.rows) == 4 * RPW_LN_LONG)                                                                \
            hipLaunchKernelGGL((ln_bwd2_kernel<NV, GT, NEXT, GATE, RPW_LN_LONG>), grid, blk, lds, s, a, nj);          \
        else                                                                                                          \
            hipLaunchKernelGGL((ln_bwd2_kernel<NV, GT, NEXT, GATE, RPW2>), grid, blk, lds, s, a, nj);                 \
    } while (0)
    if (a.dy_bf16) {
        if (gate) OMH_LN2(uint16_t, true, true); else if (next) OMH_LN2(uint16_t, true, false); else OMH_LN2(uint16_t, false, false);
    } else {
        if (gate) OMH_LN2(float, true, true); else if (next) OMH_LN2(float, true, false); else OMH_LN2(float, false, false);
    }
#undef OMH_LN2
    return gate ? 3 : 2;
}

template <int NV>
void launch_rms(const omh_rms_bwd_args& a, int nj, hipStream_t s) {
    const dim3 grid((unsigned)nj, (unsigned)a.n_seg), blk(256);
    const size_t lds = (size_t)a.dim * sizeof(float);
#define OMH_RMS2(XT, GT) hipLaunchKernelGGL((rms_bwd2_kernel<NV, XT, GT>), grid, blk, lds, s, a, nj)
    if (a.x_bf16 && a.dy_bf16) OMH_RMS2(uint16_t, uint16_t);
    else if (a.x_bf16) OMH_RMS2(uint16_t, float);
    else if (a.dy_bf16) OMH_RMS2(float, uint16_t);
    else OMH_RMS2(float, float);
#undef OMH_RMS2
}

}  // namespace

extern "C" int64_t omh_layernorm_modulate_bwd2_workspace(int64_t rows, int32_t dim, int64_t rows_per_batch) {
    if (rows <= 0 || dim <= 0 || rows_per_batch <= 0) return 0;
    const int64_t nb = (rows + rows_per_batch - 1) / rows_per_batch;
    const int64_t nj = (rows_per_batch + ln_rows_per_wg(rows) - 1) / ln_rows_per_wg(rows);
    return nb * nj * 3 * dim;
}

extern "C" int omh_layernorm_modulate_bwd2(const omh_ln_bwd_args* args, omh_stream_t stream) {
    if (!args) return OMH_E_BADARG;
    const omh_ln_bwd_args& a = *args;
    if (!a.x || !a.dy || !a.dx || !a.workspace || a.rows <= 0 || a.dim <= 0 || a.rows_per_batch <= 0) return OMH_E_BADARG;
    if ((a.dim & 3) || a.dim > MAXV2 * 256 || (a.mul1_stride & 3) || (a.gate1_stride & 3)) return OMH_E_SHAPE;
    if (((uintptr_t)a.x | (uintptr_t)a.dx | (uintptr_t)a.workspace) & 15) return OMH_E_ALIGN;
    if (((uintptr_t)a.dy | (uintptr_t)a.dy_next | (uintptr_t)a.y_next) & 7) return OMH_E_ALIGN;
    if ((((uintptr_t)a.dmul | (uintptr_t)a.dadd | (uintptr_t)a.dgate) & 15) || (a.dstride & 3) || (a.dgate_stride & 3)) return OMH_E_ALIGN;
    if (a.workspace_floats < omh_layernorm_modulate_bwd2_workspace(a.rows, a.dim, a.rows_per_batch)) return OMH_E_SHAPE;
    const int64_t nb = (a.rows + a.rows_per_batch - 1) / a.rows_per_batch;
    const int64_t nj = (a.rows_per_batch + ln_rows_per_wg(a.rows) - 1) / ln_rows_per_wg(a.rows);
    if (nj > 0x7fffffff || nb > 65535) return OMH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    omh_clear_status();
    const int np = a.dim <= 6 * 256 ? launch_ln<6>(a, (int)nj, (int)nb, s)
                                    : (a.dim <= 20 * 256 ? launch_ln<20>(a, (int)nj, (int)nb, s) : launch_ln<MAXV2>(a, (int)nj, (int)nb, s));
    if (a.deferred) {
        omh_partial_reduce& d = *a.deferred;
        d.part = a.workspace; d.nj = (int)nj; d.np = np; d.nb = (int)nb; d.dim = a.dim; d.grid_y = (int)(nb * np);
        d.out[0] = a.dmul; d.stride[0] = a.dstride; d.out[1] = a.dadd; d.stride[1] = a.dstride;
        d.out[2] = np == 3 ? a.dgate : nullptr; d.stride[2] = a.dgate_stride;
        return omh_launch_status();
    }
    hipLaunchKernelGGL(partial_colsum_kernel, dim3((a.dim + 63) / 64, (unsigned)(nb * np)), dim3(256), 0, s, a.workspace,
                       (int)nj, np, (int)nb, a.dim, a.dmul, a.dstride, a.dadd, a.dstride, np == 3 ? a.dgate : nullptr,
                       a.dgate_stride);
    return omh_launch_status();
}

extern "C" int omh_partial_colsum_multi(const omh_partial_reduce_batch* batch, omh_stream_t stream) {
    if (!batch || batch->n < 0 || batch->n > OMH_PARTIAL_REDUCE_MAX) return OMH_E_BADARG;
    if (batch->n == 0) return 0;
    int gx = 0, gy = 0;
    for (int i = 0; i < batch->n; ++i) {
        const omh_partial_reduce& e = batch->e[i];
        if (!e.part || e.nj <= 0 || e.np < 1 || e.np > 3 || e.nb < 1 || e.dim <= 0 || (e.dim & 3) || e.grid_y < 1) return OMH_E_BADARG;
        if ((((uintptr_t)e.part | (uintptr_t)e.out[0] | (uintptr_t)e.out[1] | (uintptr_t)e.out[2]) & 15)) return OMH_E_ALIGN;
        gx = gx > (e.dim + 63) / 64 ? gx : (e.dim + 63) / 64;
        gy = gy > e.grid_y ? gy : e.grid_y;
    }
    omh_clear_status();
    hipLaunchKernelGGL(partial_colsum_multi_kernel, dim3(gx, gy, batch->n), dim3(256), 0, (hipStream_t)stream, *batch);
    return omh_launch_status();
}

extern "C" int64_t omh_rmsnorm_rope_bwd2_workspace(int64_t rows, int32_t dim, int32_t n_seg) {
    if (rows <= 0 || dim <= 0 || n_seg <= 0) return 0;
    return (int64_t)n_seg * ((rows + ROWS_PER_WG - 1) / ROWS_PER_WG) * dim;
}

extern "C" int omh_rmsnorm_rope_bwd2(const omh_rms_bwd_args* args, omh_stream_t stream) {
    if (!args) return OMH_E_BADARG;
    const omh_rms_bwd_args& a = *args;
    if (!a.x || !a.dy || !a.dx || a.rows <= 0 || a.dim <= 0 || a.n_seg < 1 || a.n_seg > 2) return OMH_E_BADARG;
    if ((a.dim & 3) || a.dim > MAXV2 * 256 || (a.ldx & 3) || (a.lddy & 3) || (a.lddx & 3) || (a.seg_x & 3) || (a.seg_dy & 3) ||
        (a.seg_dx & 3))
        return OMH_E_SHAPE;
    if (a.rope_cos && (!a.rope_sin || !a.grid || a.seq_len <= 0 || a.head_dim <= 0)) return OMH_E_BADARG;
    if (a.n_seg == 2 && ((a.dweight[0] == nullptr) != (a.dweight[1] == nullptr))) return OMH_E_BADARG;
    const bool any_dw = a.dweight[0] != nullptr;
    if ((((uintptr_t)a.dweight[0] | (uintptr_t)a.dweight[1] | (uintptr_t)a.workspace) & 15)) return OMH_E_ALIGN;
    if (any_dw && (!a.workspace || a.workspace_floats < omh_rmsnorm_rope_bwd2_workspace(a.rows, a.dim, a.n_seg))) return OMH_E_SHAPE;
    const int64_t nj = (a.rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    if (nj > 0x7fffffff) return OMH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    omh_clear_status();
    if (a.dim <= 6 * 256) launch_rms<6>(a, (int)nj, s);
    else if (a.dim <= 20 * 256) launch_rms<20>(a, (int)nj, s);
    else launch_rms<MAXV2>(a, (int)nj, s);
    if (any_dw && a.deferred) {
        omh_partial_reduce& d = *a.deferred;
        d.part = a.workspace; d.nj = (int)nj; d.np = 1; d.nb = 1; d.dim = a.dim; d.grid_y = a.n_seg;
        d.out[0] = a.dweight[0]; d.stride[0] = (int64_t)(a.n_seg > 1 ? a.dweight[1] - a.dweight[0] : 1);
        d.out[1] = d.out[2] = nullptr; d.stride[1] = d.stride[2] = 0;
    } else if (any_dw)   // segment = "batch" of the column-sum launch, one partial array per workgroup
        hipLaunchKernelGGL(partial_colsum_kernel, dim3((a.dim + 63) / 64, (unsigned)a.n_seg), dim3(256), 0, s, a.workspace,
                           (int)nj, 1, 1, a.dim, a.dweight[0], (int64_t)(a.n_seg > 1 ? a.dweight[1] - a.dweight[0] : 1),
                           nullptr, 0, nullptr, 0);
    return omh_launch_status();
}
