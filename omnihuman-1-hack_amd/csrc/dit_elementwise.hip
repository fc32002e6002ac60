// HBM-bound row kernels of the DiT block for gfx950: LayerNorm+adaLN
// modulation, WanRMSNorm(+3-axis RoPE), casts, patchify / unpatchify, and the
// tiny fp32 dense layers of the time embedding.  One 64-lane wave owns one
// row; rows are read once with 16-byte loads and kept in registers.
//
// Reference arithmetic replaced (seaweed_apt/wan/modules/model.py):
//   layernorm_modulate  :91-104, 292-293, 313-315, 358
//   rmsnorm_rope        :72-88 (WanRMSNorm), :42-69 (rope_apply)
//   patchify/unpatchify :463,515-518 / :565-588
//   dense_f32, sinusoid :17-27, 469-471, 526-528
#include "omh_common.h"
#include <mutex>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int MAXV_GENERIC = 32;   // float4 chunks per lane -> dim <= 8192 (specialised: 6 = 1536, 20 = 5120)

// ------------------------------------------------------------------ LayerNorm + modulate
// A wave takes LN_RPW consecutive rows: the modulation vectors (up to four [dim] fp32 vectors, 24 floats per lane each at
// dim = 1536) are fetched once per wave and batch element instead of once per row — per row they were twice the
// loads of x itself — and the next row's x is requested before the current row's reductions.
// LN_RPW rows per wave: 4 on the long sequences (the modulation vectors are fetched once per 4 rows), 2 or 1 when the
// launch would otherwise not give every CU a few waves (omh_layernorm_modulate picks by the row count, from
// measurements).  The per-row arithmetic is the same: same bits.
template <int MAXV, int LN_RPW>
__global__ __launch_bounds__(256)
void layernorm_modulate_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t rows, int dim,
                               float eps, float mul_const, const float* __restrict__ mul0,
                               const float* __restrict__ mul1, int64_t mul1_stride,
                               const float* __restrict__ add0, const float* __restrict__ add1,
                               int64_t add1_stride, int64_t rows_per_batch) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
    if (row0 >= rows) return;
    const int nv = dim >> 2;
    float4 mu[MAXV], ad[MAXV], v[MAXV], nx[MAXV];
    int64_t have = -1;                                            // batch element whose vectors are in mu / ad
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) nx[i] = ((const float4*)(x + row0 * dim))[c];
    }
    for (int r = 0; r < LN_RPW; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) v[i] = nx[i];
        if (r + 1 < LN_RPW && row + 1 < rows) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) nx[i] = ((const float4*)(x + (row + 1) * dim))[c];
            }
        }
        const int64_t bidx = row / rows_per_batch;
        if (bidx != have) {                                       // wave-uniform
            have = bidx;
            const float4* m0 = (const float4*)mul0;
            const float4* m1 = mul1 ? (const float4*)(mul1 + bidx * mul1_stride) : nullptr;
            const float4* a0 = (const float4*)add0;
            const float4* a1 = add1 ? (const float4*)(add1 + bidx * add1_stride) : nullptr;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) {
                    float4 m = make_float4(mul_const, mul_const, mul_const, mul_const);
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m0) { const float4 t = m0[c]; m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w; }
                    if (m1) { const float4 t = m1[c]; m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w; }
                    if (a0) { const float4 t = a0[c]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
                    if (a1) { const float4 t = a1[c]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
                    mu[i] = m; ad[i] = a;
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        const float mean = wave_sum(s) / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += a * a + b * b + cc * cc + d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / dim + eps);
        uint2* yr = (uint2*)(y + row * dim);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                uint2 o;
                o.x = pack_bf2((v[i].x - mean) * rstd * mu[i].x + ad[i].x, (v[i].y - mean) * rstd * mu[i].y + ad[i].y);
                o.y = pack_bf2((v[i].z - mean) * rstd * mu[i].z + ad[i].z, (v[i].w - mean) * rstd * mu[i].w + ad[i].w);
                yr[c] = o;
            }
        }
    }
}

// ------------------------------------------------------------------ RMSNorm (+RoPE)
template <int MAXV, bool IN_BF16>
__device__ __forceinline__
void rmsnorm_rope_row(const void* __restrict__ xv, int64_t ldx, uint16_t* __restrict__ y, int64_t rows,
                      int dim, const float* __restrict__ weight, float eps, int do_norm,
                      const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                      int rope_len, int head_dim, const int* __restrict__ grid, int seq_len, float out_scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = dim >> 2;
    const float4* xr = (const float4*)((const float*)xv + row * ldx);
    const uint2* xh = (const uint2*)((const uint16_t*)xv + row * ldx);
    float4 v[MAXV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            if (IN_BF16) {
                const uint2 h = xh[c];
                v[i] = make_float4(bf2f((uint16_t)(h.x & 0xffff)), bf2f((uint16_t)(h.x >> 16)),
                                   bf2f((uint16_t)(h.y & 0xffff)), bf2f((uint16_t)(h.y >> 16)));
            } else {
                v[i] = xr[c];
            }
            q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    // out_scale rides on the normalisation factor: one fp32 multiply per element either way (1.0f is exact)
    const float rinv = (do_norm ? rsqrtf(wave_sum(q) / dim + eps) : 1.0f) * out_scale;

    // token position for RoPE
    bool rot = false;
    int pf = 0, ph = 0, pw = 0;
    const int hc = head_dim >> 1;                 // complex pairs per head
    const int c3 = hc / 3, cf = hc - 2 * c3;
    if (rope_cos) {
        const int b = (int)(row / seq_len), s = (int)(row % seq_len);
        const int gf = grid[3 * b], gh = grid[3 * b + 1], gw = grid[3 * b + 2];
        if (s < gf * gh * gw) {
            rot = true;
            pf = s / (gh * gw);
            ph = (s / gw) % gh;
            pw = s % gw;
        }
    }
    const float4* wv = (const float4*)weight;
    uint2* yr = (uint2*)(y + row * dim);
    // A lane's vectors sit 256 columns apart: when head_dim divides 256 (128 in every Wan model) they all fall on the
    // same two complex pairs of their heads, so the lane fetches its four table entries once per row instead of once
    // per vector (24 scattered 4-byte loads per lane and row before: the kernel ran at 2.7 TB/s on them).
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
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 t = v[i];
            t.x *= rinv; t.y *= rinv; t.z *= rinv; t.w *= rinv;
            if (wv) { const float4 g = wv[c]; t.x *= g.x; t.y *= g.y; t.z *= g.z; t.w *= g.w; }
            if (rot) {
                // columns 4c..4c+3 = complex pairs p0 = (4c mod head_dim)/2 and p0+1
                const int p0 = ((4 * c) % head_dim) >> 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int pc = p0 + e;
                    const int pos = pc < cf ? pf : (pc < cf + c3 ? ph : pw);
                    const int idx = min(pos, rope_len - 1) * hc + pc;
                    const float cs = same_pairs ? cs2[e] : rope_cos[idx], sn = same_pairs ? sn2[e] : rope_sin[idx];
                    float& re = e == 0 ? t.x : t.z;
                    float& im = e == 0 ? t.y : t.w;
                    const float nr = re * cs - im * sn;
                    const float ni = re * sn + im * cs;
                    re = nr; im = ni;
                }
            }
            uint2 o;
            o.x = pack_bf2(t.x, t.y);
            o.y = pack_bf2(t.z, t.w);
            yr[c] = o;
        }
    }
}

template <int MAXV, bool IN_BF16>
__global__ __launch_bounds__(256)
void rmsnorm_rope_kernel(const void* __restrict__ xv, int64_t ldx, uint16_t* __restrict__ y, int64_t rows,
                         int dim, const float* __restrict__ weight, float eps, int do_norm,
                         const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                         int rope_len, int head_dim, const int* __restrict__ grid, int seq_len, float out_scale) {
    rmsnorm_rope_row<MAXV, IN_BF16>(xv, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid,
                                    seq_len, out_scale);
}

// Two column segments of the same rows in ONE launch (blockIdx.y = segment): q and k of the self-attention out of the fused
// q|k projection, each with its own gain, output and output scale — the same row arithmetic as two launches of the kernel
// above (same bits), one launch fewer per block (latency-bound at one or two [16,1,60,104] clips: 2 x 8 us -> 10 us).
template <int MAXV>
__global__ __launch_bounds__(256)
void rmsnorm_rope_pair_kernel(const uint16_t* __restrict__ x, int64_t ldx, int64_t seg_x, uint16_t* __restrict__ y0,
                              uint16_t* __restrict__ y1, int64_t rows, int dim, const float* __restrict__ w0,
                              const float* __restrict__ w1, float eps, int do_norm, const float* __restrict__ rope_cos,
                              const float* __restrict__ rope_sin, int rope_len, int head_dim,
                              const int* __restrict__ grid, int seq_len, float scale0, float scale1) {
    const bool s = blockIdx.y != 0;                                  // workgroup-uniform
    rmsnorm_rope_row<MAXV, true>(x + (s ? seg_x : 0), ldx, s ? y1 : y0, rows, dim, s ? w1 : w0, eps, do_norm, rope_cos, rope_sin,
                                 rope_len, head_dim, grid, seq_len, s ? scale1 : scale0);
}

// Long inputs (round 6): ONE wave takes both segments of its row — the 2 x MAXV loads of q and k are in flight together
// (6 KB per wave instead of 3), the token position (four integer divisions) and the lane's four rotary-table entries are
// computed once for the two (q and k share the row and the head layout) — and the arithmetic of each segment is
// rmsnorm_rope_row's, operation for operation: the same bits.  rope on, head_dim dividing 256 (every Wan model).
// Measured at 32 760 rows x 2 x 1 536: 120 -> see DESIGN.md 4.1 (the two-workgroup form streams the q halves of all rows
// first and the k halves afterwards: 3 KB pieces of 6 KB rows).
template <int MAXV>
__global__ __launch_bounds__(256)
void rmsnorm_rope_pair_row_kernel(const uint16_t* __restrict__ x, int64_t ldx, int64_t seg_x, uint16_t* __restrict__ y0,
                                  uint16_t* __restrict__ y1, int64_t rows, int dim, const float* __restrict__ w0,
                                  const float* __restrict__ w1, float eps, int do_norm, const float* __restrict__ rope_cos,
                                  const float* __restrict__ rope_sin, int rope_len, int head_dim,
                                  const int* __restrict__ grid, int seq_len, float scale0, float scale1) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = dim >> 2;
    const uint2* xa = (const uint2*)(x + row * ldx);
    const uint2* xb = (const uint2*)(x + seg_x + row * ldx);
    uint2 ha[MAXV], hb[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        ha[i] = c < nv ? xa[c] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        hb[i] = c < nv ? xb[c] : make_uint2(0u, 0u);
    }
    // token position and the lane's two complex pairs (the same for every vector of the lane and for both segments)
    const int hc = head_dim >> 1, c3 = hc / 3, cf = hc - 2 * c3;
    bool rot = false;
    float cs2[2] = {1.f, 1.f}, sn2[2] = {0.f, 0.f};
    {
        const int b = (int)(row / seq_len), sidx = (int)(row % seq_len);
        const int gf = grid[3 * b], gh = grid[3 * b + 1], gw = grid[3 * b + 2];
        if (sidx < gf * gh * gw) {
            rot = true;
            const int pf = sidx / (gh * gw), ph = (sidx / gw) % gh, pw = sidx % gw;
            const int p0 = ((4 * lane) % head_dim) >> 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pc = p0 + e;
                const int pos = pc < cf ? pf : (pc < cf + c3 ? ph : pw);
                const int idx = min(pos, rope_len - 1) * hc + pc;
                cs2[e] = rope_cos[idx]; sn2[e] = rope_sin[idx];
            }
        }
    }
    auto finish = [&](uint2 (&h)[MAXV], uint16_t* __restrict__ y, const float* __restrict__ weight, float out_scale) {
        float4 v[MAXV];
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[i] = make_float4(bf2f((uint16_t)(h[i].x & 0xffff)), bf2f((uint16_t)(h[i].x >> 16)),
                                   bf2f((uint16_t)(h[i].y & 0xffff)), bf2f((uint16_t)(h[i].y >> 16)));
                q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            }
        }
        const float rinv = (do_norm ? rsqrtf(wave_sum(q) / dim + eps) : 1.0f) * out_scale;
        const float4* wv = (const float4*)weight;
        uint2* yr = (uint2*)(y + row * dim);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 t = v[i];
                t.x *= rinv; t.y *= rinv; t.z *= rinv; t.w *= rinv;
                if (wv) { const float4 g = wv[c]; t.x *= g.x; t.y *= g.y; t.z *= g.z; t.w *= g.w; }
                if (rot) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float cs = cs2[e], sn = sn2[e];
                        float& re = e == 0 ? t.x : t.z;
                        float& im = e == 0 ? t.y : t.w;
                        const float nr = re * cs - im * sn;
                        const float ni = re * sn + im * cs;
                        re = nr; im = ni;
                    }
                }
                uint2 o;
                o.x = pack_bf2(t.x, t.y);
                o.y = pack_bf2(t.z, t.w);
                yr[c] = o;
            }
        }
    };
    finish(ha, y0, w0, scale0);
    finish(hb, y1, w1, scale1);
}

// ------------------------------------------------------------------ cast
__global__ __launch_bounds__(256)
void cast_f32_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const float4 t = ((const float4*)x)[i];
        uint2 o;
        o.x = pack_bf2(t.x, t.y);
        o.y = pack_bf2(t.z, t.w);
        ((uint2*)y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        y[i] = f2bf(x[i]);
    }
}

// ------------------------------------------------------------------ patchify / unpatchify
__global__ __launch_bounds__(256)
void patchify_kernel(const float* __restrict__ x, uint16_t* __restrict__ tok, int C, int F, int H, int W,
                     int pt, int ph, int pw, int Kp) {
    const int gf = F / pt, gh = H / ph, gw = W / pw;
    const int64_t total = (int64_t)gf * gh * gw * Kp;
    const int kvalid = C * pt * ph * pw;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int k = (int)(i % Kp);
        const int64_t s = i / Kp;
        float val = 0.f;
        if (k < kvalid) {
            const int j = k % pw, ii = (k / pw) % ph, a = (k / (pw * ph)) % pt, c = k / (pw * ph * pt);
            const int w = (int)(s % gw), h = (int)((s / gw) % gh), f = (int)(s / ((int64_t)gw * gh));
            val = x[(((int64_t)c * F + f * pt + a) * H + h * ph + ii) * W + w * pw + j];
        }
        tok[i] = f2bf(val);
    }
}

__global__ __launch_bounds__(256)
void unpatchify_kernel(const float* __restrict__ tok, float* __restrict__ out, int Cout, int gf, int gh, int gw,
                       int pt, int ph, int pw) {
    // out[c][f*pt+a][h*ph+i][w*pw+j] = tok[(f,h,w)][((a*ph+i)*pw+j)*Cout + c]
    const int F = gf * pt, H = gh * ph, W = gw * pw;
    const int64_t total = (int64_t)Cout * F * H * W;
    const int ncol = pt * ph * pw * Cout;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ww = (int)(i % W), hh = (int)((i / W) % H), ff = (int)((i / ((int64_t)W * H)) % F);
        const int c = (int)(i / ((int64_t)W * H * F));
        const int w = ww / pw, j = ww % pw, h = hh / ph, ii = hh % ph, f = ff / pt, a = ff % pt;
        const int64_t s = ((int64_t)f * gh + h) * gw + w;
        out[i] = tok[s * ncol + ((a * ph + ii) * pw + j) * Cout + c];
    }
}

// ------------------------------------------------------------------ tiny fp32 dense / sinusoid
__global__ __launch_bounds__(256)
void dense_f32_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                      float* __restrict__ y, int B, int N, int K, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // output index b*N + n
    if (o >= (int64_t)B * N) return;
    const int b = (int)(o / N), n = (int)(o % N);
    const float* xr = x + (int64_t)b * K;
    const float* wr = W + (int64_t)n * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = xr[k];
        if (act_in == 1) xv = xv / (1.0f + expf(-xv));
        s = fmaf(xv, wr[k], s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (bias) s += bias[n];
        if (act_out == 1) s = s / (1.0f + expf(-s));
        y[o] = s;
    }
}

__global__ void sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i % half;
    const double ang = (double)t[b] * pow(10000.0, -(double)k / (double)half);
    out[(int64_t)b * dim + k] = (float)cos(ang);
    out[(int64_t)b * dim + half + k] = (float)sin(ang);
}

// ------------------------------------------------------------------ CFG + UniPC step
__global__ __launch_bounds__(256)
void cfg_unipc_kernel(const float* __restrict__ cond, const float* __restrict__ uncond, const float* x,
                      const float* last, const float* __restrict__ m1, const float* __restrict__ m2,
                      float* __restrict__ mt_out, float* xc_out, float* x_next, int64_t n, float guide, float sigma,
                      int use_corr, float ca_last, float ca_m1, float ca_m2, float ca_mt, float pb_x, float pb_mt,
                      float pb_m1) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float u = uncond[i];
        const float v = u + guide * (cond[i] - u);
        const float xv = x[i];
        const float mt = xv - sigma * v;
        const float a1 = m1 ? m1[i] : 0.f;
        float xc = xv;
        if (use_corr) {
            xc = ca_last * last[i] + ca_mt * mt;
            if (m1) xc += ca_m1 * a1;
            if (m2) xc += ca_m2 * m2[i];
        }
        float xn = pb_x * xc + pb_mt * mt;
        if (m1) xn += pb_m1 * a1;
        if (mt_out) mt_out[i] = mt;
        if (xc_out) xc_out[i] = xc;
        x_next[i] = xn;
    }
}

inline int grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int omh_layernorm_modulate(const float* x, void* y, int64_t rows, int32_t dim, float eps,
                                      float mul_const, const float* mul0, const float* mul1, int64_t mul1_stride,
                                      const float* add0, const float* add1, int64_t add1_stride,
                                      int64_t rows_per_batch, omh_stream_t stream) {
    if (!x || !y || rows <= 0 || dim <= 0 || rows_per_batch <= 0) return OMH_E_BADARG;
    if ((dim & 3) || dim > MAXV_GENERIC * 256) return OMH_E_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7) || (mul1_stride & 3) || (add1_stride & 3)) return OMH_E_ALIGN;
    omh_clear_status();
    // rows per wave: 4 on long inputs (the modulation vectors fetched once per 4 rows), fewer when that would leave the CUs
    // with a handful of waves.  Measured alone (us at 1536 columns, 1 / 2 / 4 rows per wave; tools/ln_rpw_probe.py):
    // 1 560 rows 6.6 / 7.7 / 10.9, 3 120 rows 8.0 / 8.8 / 11.4, 6 240 rows 19.4 / 16.3 / 17.5, 24 960 rows 69 / 58 / 52.
    // OMH_LN_RPW = 1 / 2 / 4 forces it (timing).  Same per-row arithmetic: same bits.
    const char* force = omh_opt(OMH_OPT_LN_RPW);
    // Round 5: 16 rows per wave from 24 576 rows on (one clip of 81 frames: 32 760) — us at 1 / 2 / 4 / 8 / 16 rows per wave:
    // 24 960 rows 69 / 58 / 52 / 53 / 48, 32 760 rows 87 / 73 / 62 / 60 / 60, 65 520 rows 186 / 152 / 136 / 123 / 124.
    int rpw = rows <= 4096 ? 1 : (rows < 16384 ? 2 : (rows < 24576 ? 4 : 16));
    if (force) { const int f = atoi(force); rpw = (f == 1 || f == 2 || f == 8 || f == 16) ? f : 4; }
#define OMH_LN_PICK(MV) (rpw == 1 ? layernorm_modulate_kernel<MV, 1> : (rpw == 2 ? layernorm_modulate_kernel<MV, 2> : \
                         (rpw == 8 ? layernorm_modulate_kernel<MV, 8> : (rpw == 16 ? layernorm_modulate_kernel<MV, 16> : layernorm_modulate_kernel<MV, 4>))))
    auto kern = dim <= 6 * 256 ? OMH_LN_PICK(6) : (dim <= 20 * 256 ? OMH_LN_PICK(20) : OMH_LN_PICK(MAXV_GENERIC));
#undef OMH_LN_PICK
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), dim3(256), 0,
                       (hipStream_t)stream, x, (uint16_t*)y, rows, dim, eps, mul_const, mul0, mul1, mul1_stride,
                       add0, add1, add1_stride, rows_per_batch);
    return omh_launch_status();
}

template <bool IN_BF16>
static int rmsnorm_rope_launch(const void* x, int64_t ldx, void* y, int64_t rows, int32_t dim, const float* weight,
                               float eps, int32_t do_norm, const float* rope_cos, const float* rope_sin,
                               int32_t rope_len, int32_t head_dim, const int32_t* grid, int32_t seq_len,
                               float out_scale, omh_stream_t stream) {
    if (!x || !y || rows <= 0 || dim <= 0) return OMH_E_BADARG;
    if ((dim & 3) || dim > MAXV_GENERIC * 256 || (ldx & 3)) return OMH_E_SHAPE;
    if (rope_cos && (!rope_sin || !grid || seq_len <= 0 || head_dim <= 0 || (head_dim & 3) || dim % head_dim))
        return OMH_E_BADARG;
    if (((uintptr_t)x & (IN_BF16 ? 7 : 15)) || ((uintptr_t)y & 7)) return OMH_E_ALIGN;
    omh_clear_status();
    auto kern = dim <= 6 * 256 ? rmsnorm_rope_kernel<6, IN_BF16>
                               : (dim <= 20 * 256 ? rmsnorm_rope_kernel<20, IN_BF16> : rmsnorm_rope_kernel<MAXV_GENERIC, IN_BF16>);
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, (uint16_t*)y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len,
                       head_dim, grid, seq_len, out_scale);
    return omh_launch_status();
}

extern "C" int omh_rmsnorm_rope(const float* x, int64_t ldx, void* y, int64_t rows, int32_t dim,
                                const float* weight, float eps, int32_t do_norm, const float* rope_cos,
                                const float* rope_sin, int32_t rope_len, int32_t head_dim, const int32_t* grid,
                                int32_t seq_len, omh_stream_t stream) {
    return rmsnorm_rope_launch<false>(x, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len,
                                      head_dim, grid, seq_len, 1.0f, stream);
}

extern "C" int omh_rmsnorm_rope_bf16(const void* x_bf16, int64_t ldx, void* y, int64_t rows, int32_t dim,
                                     const float* weight, float eps, int32_t do_norm, const float* rope_cos,
                                     const float* rope_sin, int32_t rope_len, int32_t head_dim,
                                     const int32_t* grid, int32_t seq_len, float out_scale, omh_stream_t stream) {
    return rmsnorm_rope_launch<true>(x_bf16, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len,
                                     head_dim, grid, seq_len, out_scale, stream);
}

extern "C" int omh_rmsnorm_rope_bf16_pair(const void* x_bf16, int64_t ldx, int64_t seg_x, void* y0, void* y1, int64_t rows,
                                          int32_t dim, const float* weight0, const float* weight1, float eps, int32_t do_norm,
                                          const float* rope_cos, const float* rope_sin, int32_t rope_len, int32_t head_dim,
                                          const int32_t* grid, int32_t seq_len, float out_scale0, float out_scale1,
                                          omh_stream_t stream) {
    if (!x_bf16 || !y0 || !y1 || rows <= 0 || dim <= 0) return OMH_E_BADARG;
    if ((dim & 3) || dim > MAXV_GENERIC * 256 || (ldx & 3) || (seg_x & 3)) return OMH_E_SHAPE;
    if (rope_cos && (!rope_sin || !grid || seq_len <= 0 || head_dim <= 0 || (head_dim & 3) || dim % head_dim))
        return OMH_E_BADARG;
    if (((uintptr_t)x_bf16 & 7) || ((uintptr_t)y0 & 7) || ((uintptr_t)y1 & 7)) return OMH_E_ALIGN;
    omh_clear_status();
    // long inputs with the rotary table on a 256-periodic head layout: one wave per row takes both segments (RMS_PAIR_ROW
    // = "0": never, "1": any row count)
    {
        const char* pr = omh_opt(OMH_OPT_RMS_PAIR_ROW);
        const bool off = pr && pr[0] == '0', on = pr && pr[0] == '1';
        if (!off && rope_cos && head_dim > 0 && (256 % head_dim) == 0 && dim <= 20 * 256 && (on || rows >= 8192)) {
            auto k2 = dim <= 6 * 256 ? rmsnorm_rope_pair_row_kernel<6> : rmsnorm_rope_pair_row_kernel<20>;
            hipLaunchKernelGGL(k2, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                               (const uint16_t*)x_bf16, ldx, seg_x, (uint16_t*)y0, (uint16_t*)y1, rows, dim, weight0, weight1,
                               eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid, seq_len, out_scale0, out_scale1);
            return omh_launch_status();
        }
    }
    auto kern = dim <= 6 * 256 ? rmsnorm_rope_pair_kernel<6>
                               : (dim <= 20 * 256 ? rmsnorm_rope_pair_kernel<20> : rmsnorm_rope_pair_kernel<MAXV_GENERIC>);
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 3) / 4), 2), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x_bf16, ldx, seg_x, (uint16_t*)y0, (uint16_t*)y1, rows, dim, weight0, weight1, eps,
                       do_norm, rope_cos, rope_sin, rope_len, head_dim, grid, seq_len, out_scale0, out_scale1);
    return omh_launch_status();
}

extern "C" int omh_cast_f32_bf16(const float* x, void* y, int64_t n, omh_stream_t stream) {
    if (!x || !y || n <= 0) return OMH_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, (uint16_t*)y, n);
    return omh_launch_status();
}

extern "C" int omh_patchify(const float* x, void* tok, int32_t C, int32_t F, int32_t H, int32_t W, int32_t pt,
                            int32_t ph, int32_t pw, int32_t Kp, omh_stream_t stream) {
    if (!x || !tok || C <= 0 || F <= 0 || H <= 0 || W <= 0 || pt <= 0 || ph <= 0 || pw <= 0) return OMH_E_BADARG;
    if (F % pt || H % ph || W % pw || Kp < C * pt * ph * pw) return OMH_E_SHAPE;
    const int64_t total = (int64_t)(F / pt) * (H / ph) * (W / pw) * Kp;
    omh_clear_status();
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (uint16_t*)tok, C, F, H, W, pt, ph, pw, Kp);
    return omh_launch_status();
}

extern "C" int omh_unpatchify(const float* tok, float* out, int32_t Cout, int32_t f, int32_t h, int32_t w,
                              int32_t pt, int32_t ph, int32_t pw, omh_stream_t stream) {
    if (!tok || !out || Cout <= 0 || f <= 0 || h <= 0 || w <= 0 || pt <= 0 || ph <= 0 || pw <= 0)
        return OMH_E_BADARG;
    const int64_t total = (int64_t)Cout * f * pt * h * ph * w * pw;
    omh_clear_status();
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, tok, out,
                       Cout, f, h, w, pt, ph, pw);
    return omh_launch_status();
}

extern "C" int omh_dense_f32(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t N,
                             int32_t K, int32_t act_in, int32_t act_out, omh_stream_t stream) {
    if (!x || !W || !y || B <= 0 || N <= 0 || K <= 0) return OMH_E_BADARG;
    const int64_t outs = (int64_t)B * N;
    omh_clear_status();
    hipLaunchKernelGGL(dense_f32_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, W,
                       bias, y, B, N, K, act_in, act_out);
    return omh_launch_status();
}

extern "C" int omh_sinusoidal_embedding(const float* t, float* out, int32_t B, int32_t dim, omh_stream_t stream) {
    if (!t || !out || B <= 0 || dim <= 0 || (dim & 1)) return OMH_E_BADARG;
    const int n = B * (dim / 2);
    omh_clear_status();
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, dim);
    return omh_launch_status();
}

extern "C" int omh_cfg_unipc_step(const float* cond, const float* uncond, const float* x, const float* last,
                                  const float* m1, const float* m2, float* mt_out, float* xc_out, float* x_next,
                                  int64_t n, float guide, float sigma, int32_t use_corr, float ca_last, float ca_m1,
                                  float ca_m2, float ca_mt, float pb_x, float pb_mt, float pb_m1,
                                  omh_stream_t stream) {
    if (!cond || !uncond || !x || !x_next || n <= 0) return OMH_E_BADARG;
    if (use_corr && !last) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(cfg_unipc_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, cond, uncond, x,
                       last, m1, m2, mt_out, xc_out, x_next, n, guide, sigma, use_corr, ca_last, ca_m1, ca_m2, ca_mt,
                       pb_x, pb_mt, pb_m1);
    return omh_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Process options (omh_common.h).  g_opt_val[i][0] == 0: unset.  The table is filled from the environment once.
namespace {
const char* const g_opt_name[OMH_OPT_COUNT] = {
#define X(n) #n,
    OMH_OPTIONS(X)
#undef X
};
char g_opt_val[OMH_OPT_COUNT][48];
char g_opt_env[OMH_OPT_COUNT][48];                                   // what the environment said (omh_set_option(NULL, NULL) restores it)
int g_deterministic = 0, g_deterministic_env = 0;
std::once_flag g_opt_once;

void opt_store(char* dst, const char* v) {
    size_t n = v ? strlen(v) : 0;
    if (n > 47) n = 47;
    if (n) memcpy(dst, v, n);
    dst[n] = 0;
}
void opt_init() {
    std::call_once(g_opt_once, [] {
        char name[64];
        for (int i = 0; i < OMH_OPT_COUNT; ++i) {
            snprintf(name, sizeof name, "OMH_%s", g_opt_name[i]);
            opt_store(g_opt_env[i], getenv(name));
            opt_store(g_opt_val[i], g_opt_env[i]);
        }
        const char* e = getenv("OMH_DETERMINISTIC");
        g_deterministic = g_deterministic_env = (e && e[0] == '1') ? 1 : 0;
    });
}
int opt_index(const char* key) {
    if (!key) return -1;
    if (!strncmp(key, "OMH_", 4)) key += 4;
    for (int i = 0; i < OMH_OPT_COUNT; ++i)
        if (!strcmp(key, g_opt_name[i])) return i;
    return -1;
}
}  // namespace

const char* omh_opt(int id) {
    opt_init();
    return g_opt_val[id][0] ? g_opt_val[id] : nullptr;
}
extern "C" int omh_set_option(const char* key, const char* value) {
    opt_init();
    if (!key) {                                                      // (NULL, NULL): back to what the environment said at start-up
        if (value) return OMH_E_BADARG;
        for (int i = 0; i < OMH_OPT_COUNT; ++i) opt_store(g_opt_val[i], g_opt_env[i]);
        g_deterministic = g_deterministic_env;
        return 0;
    }
    if (!strcmp(key, "DETERMINISTIC") || !strcmp(key, "OMH_DETERMINISTIC")) {
        g_deterministic = (value && value[0] == '1') ? 1 : 0;
        return 0;
    }
    const int i = opt_index(key);
    if (i < 0) return OMH_E_BADARG;
    if (value && strlen(value) > 47) return OMH_E_SHAPE;
    opt_store(g_opt_val[i], value);
    return 0;
}
extern "C" const char* omh_get_option(const char* key) {
    opt_init();
    if (key && (!strcmp(key, "DETERMINISTIC") || !strcmp(key, "OMH_DETERMINISTIC"))) return g_deterministic ? "1" : nullptr;
    const int i = opt_index(key);
    return i < 0 ? nullptr : omh_opt(i);
}
extern "C" int omh_option_count(void) { return OMH_OPT_COUNT; }
extern "C" const char* omh_option_name(int i) { return (i >= 0 && i < OMH_OPT_COUNT) ? g_opt_name[i] : nullptr; }

bool omh_deterministic() {
    opt_init();
    return g_deterministic == 1;
}
extern "C" int omh_set_deterministic(int on) {
    opt_init();
    if (on >= 0) g_deterministic = on ? 1 : 0;
    return g_deterministic;
}

extern "C" int omh_abi_version(void) { return OMH_ABI_VERSION; }
extern "C" const char* omh_build_arch(void) { return "gfx950"; }
