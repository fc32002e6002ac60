// HBM-bound kernels of the Wan 3D VAE on gfx950: per-voxel RMS norm + SiLU,
// layout converts at the NCTHW fp32 boundary, row softmax of the mid-block
// attention.  (seaweed_apt/wan/modules/vae.py:39-54, 223-262, 535-553, 661)
#include "omh_common.h"

namespace {

// One voxel (C channels, bf16) is handled by G consecutive lanes, 8 channels
// (16 bytes) per lane per pass.  A lane keeps its channels for every voxel it visits (grid-stride over voxels), so
// gamma sits in registers: round 2 fetched it with 8 scalar-width loads per chunk and voxel — 24 of the 30 vector
// memory instructions of a C = 96 voxel, which is what bounded the kernel (2.6 - 3.1 TB/s at 480x832).  Same
// arithmetic in the same order as before: the values do not move.
// OUT3 (the fp32-faithful VAE mode, omh_rms_silu_cl_split3): the result is not rounded once but written as a bf16 pair
// hi = bf16(v), lo = bf16(v - hi) in three blocks of C channels [hi | lo | hi] per voxel — the split-bf16 operand layout
// of omh_split3_f32.
template <int G, bool XF32, int OUT3 = 0>
__global__ __launch_bounds__(256)
void rms_silu_kernel(const void* __restrict__ xv, const float* __restrict__ gamma, uint16_t* __restrict__ y,
                     int64_t P, int C, int do_silu) {
    const int lane_g = threadIdx.x % G;
    const int nch = C >> 3;        // 8-channel chunks per voxel
    constexpr int MAXC = 4;        // chunks per lane  -> C <= 8*G*MAXC
    float gm[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane_g + G * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[i][e] = gamma[c * 8 + e];
        }
    }
    const float sqrtf_c = sqrtf((float)C);
    const int64_t per = blockDim.x / G;
    // a group's G lanes share p: they leave the loop together (G divides the wave)
    for (int64_t p = (int64_t)blockIdx.x * per + threadIdx.x / G; p < P; p += (int64_t)gridDim.x * per) {
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane_g + G * i;
        if (c < nch) {
            if (XF32) {
                const float4 a = *(const float4*)((const float*)xv + p * C + c * 8);
                const float4 b = *(const float4*)((const float*)xv + p * C + c * 8 + 4);
                v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
                v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
            } else {
                const uint4 u = *(const uint4*)((const uint16_t*)xv + p * C + c * 8);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[i][2 * e] = bf2f((uint16_t)(w[e] & 0xffff));
                    v[i][2 * e + 1] = bf2f((uint16_t)(w[e] >> 16));
                }
            }
            // one fma per value, in channel order: the fused epilogue of the convolution stream (gen_conv_w64.py,
            // norm kinds) adds in exactly this order, so a layer's values do not depend on which of the two ran
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(v[i][e], v[i][e], ss);
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    // sqrt(C) / max(||x||, 1e-12) on the raw v_sqrt_f32 / v_rcp_f32 (1 ulp each; the result is rounded to bf16)
    // (round 5: the split outputs too — the fused epilogue of the pair stream computes exactly these instructions, and a
    // 1-ulp reciprocal / exponential is far inside the fp32-faithful mode's 2e-4 bound: 1.6e-5 measured either way)
    const float inv = sqrtf_c * __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ss), 1e-12f));
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane_g + G * i;
        if (c < nch) {
            uint32_t o[4], ol[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = (v[i][2 * e] * inv) * gm[i][2 * e];
                float b = (v[i][2 * e + 1] * inv) * gm[i][2 * e + 1];
                if (do_silu) { a = silu(a); b = silu(b); }
                o[e] = pack_bf2(a, b);
                if (OUT3) ol[e] = pack_bf2(a - __uint_as_float(o[e] << 16), b - __uint_as_float(o[e] & 0xffff0000u));
            }
            if (OUT3 == 2) {                                      // pairs per 16 channels: [hi(16) | lo(16)] (omh_conv_args.pair)
                uint16_t* yo = y + p * 2 * C + (c >> 1) * 32 + (c & 1) * 8;
                *(uint4*)yo = make_uint4(o[0], o[1], o[2], o[3]);
                *(uint4*)(yo + 16) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
            } else if (OUT3) {
                uint16_t* yo = y + p * 3 * C + c * 8;
                *(uint4*)yo = make_uint4(o[0], o[1], o[2], o[3]);
                *(uint4*)(yo + C) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
                *(uint4*)(yo + 2 * C) = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
                *(uint4*)(y + p * C + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    }
}

__global__ __launch_bounds__(256)
void nchw_to_cl_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int C, int T, int H, int W, int Cp,
                       const float* __restrict__ mul, const float* __restrict__ add, int t_total, int t0) {
    const int64_t vox = (int64_t)T * H * W;
    const int64_t total = vox * Cp;
    const int64_t cstride = (int64_t)t_total * H * W, toff = (int64_t)t0 * H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % Cp);
        const int64_t v = i / Cp;
        float val = 0.f;
        if (c < C) {
            val = x[(int64_t)c * cstride + toff + v];
            if (mul) val *= mul[c];
            if (add) val += add[c];
        }
        y[i] = f2bf(val);
    }
}

__global__ __launch_bounds__(256)
void cl_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T, int H, int W, int Cp,
                       const float* __restrict__ mul, const float* __restrict__ add, float lo, float hi,
                       int t_total, int t0) {
    const int64_t vox = (int64_t)T * H * W;
    const int64_t total = vox * C;
    const int64_t cstride = (int64_t)t_total * H * W, toff = (int64_t)t0 * H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t v = i % vox;
        const int c = (int)(i / vox);
        float val = x[v * Cp + c];
        if (add) val += add[c];
        if (mul) val *= mul[c];
        y[(int64_t)c * cstride + toff + v] = fminf(fmaxf(val, lo), hi);
    }
}

// one 256-thread block per row
__global__ __launch_bounds__(256)
void softmax_rows_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy, int L,
                         float scale) {
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * ldx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int j = tid; j < L; j += 256) mx = fmaxf(mx, xr[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = scale * 1.4426950408889634f;
    float s = 0.f;
    for (int j = tid; j < L; j += 256) s += exp2f((xr[j] - mx) * sc);
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    uint16_t* yr = y + r * ldy;
    for (int j = tid; j < L; j += 256) yr[j] = f2bf(exp2f((xr[j] - mx) * sc) * inv);
}

// fp32 [rows, C] (row pitch ldx) -> split-bf16 operand [rows, 3 Cp] (row pitch ldy): with hi = bf16(x), lo = bf16(x - hi)
// (x = hi + lo to 2^-17 relative), pattern 0 writes the channel blocks [hi | lo | hi], pattern 1 [hi | hi | lo]; pad
// channels C..Cp-1 are zero.  A product over 3 Cp channels of a pattern-0 row with a pattern-1 row is
// x_hi w_hi + x_lo w_hi + x_hi w_lo = x w - x_lo w_lo: an fp32-class product on the bf16 MFMA path (three products per
// tile, fp32 accumulate).  One thread per 4 channels.
__global__ __launch_bounds__(256)
void split3_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy, int64_t rows,
                   int C, int Cp, int pattern) {
    const int q = Cp >> 2;                                           // 4-channel groups per row
    const int64_t total = rows * q;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (C & 3) == 0 && (ldx & 3) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* xr = x + r * ldx + c;
        if (vec && c + 3 < C) { const float4 t = *(const float4*)xr; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < C) v[e] = xr[e];
        }
        const uint32_t h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]);
        const uint32_t l0 = pack_bf2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
        const uint32_t l1 = pack_bf2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
        const uint2 hi = make_uint2(h0, h1), lo = make_uint2(l0, l1);
        if (pattern == 2) {                                          // pairs per 16 channels: [hi(16) | lo(16)], 2 Cp per row
            uint16_t* yp = y + r * ldy + (c >> 4) * 32 + (c & 15);
            *(uint2*)yp = hi;
            *(uint2*)(yp + 16) = lo;
            continue;
        }
        uint16_t* yr = y + r * ldy + c;
        *(uint2*)yr = hi;
        *(uint2*)(yr + Cp) = pattern == 0 ? lo : hi;
        *(uint2*)(yr + 2 * Cp) = pattern == 0 ? hi : lo;
    }
}

__global__ __launch_bounds__(256)
void nchw_to_cl_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T, int H, int W, int Cp,
                           const float* __restrict__ mul, const float* __restrict__ add, int t_total, int t0) {
    const int64_t vox = (int64_t)T * H * W;
    const int64_t total = vox * Cp;
    const int64_t cstride = (int64_t)t_total * H * W, toff = (int64_t)t0 * H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % Cp);
        const int64_t v = i / Cp;
        float val = 0.f;
        if (c < C) {
            val = x[(int64_t)c * cstride + toff + v];
            if (mul) val *= mul[c];
            if (add) val += add[c];
        }
        y[i] = val;
    }
}

// row softmax with an fp32 result (the fp32-faithful VAE mode): IEEE exp and division
__global__ __launch_bounds__(256)
void softmax_rows_f32_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int L,
                             float scale) {
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * ldx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int j = tid; j < L; j += 256) mx = fmaxf(mx, xr[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int j = tid; j < L; j += 256) s += expf((xr[j] - mx) * scale);
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    const float tot = red[4] + red[5] + red[6] + red[7];
    float* yr = y + r * ldy;
    for (int j = tid; j < L; j += 256) yr[j] = expf((xr[j] - mx) * scale) / tot;
}

inline int grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// in-place ReLU over bf16, 8 elements (16 bytes) per lane and trip: sign bit set -> +0
__global__ __launch_bounds__(256) void relu_bf16_kernel(uint4* __restrict__ x, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        uint4 v = x[i];
        uint32_t* w = (uint32_t*)&v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t u = w[e];
            if (u & 0x00008000u) u &= 0xffff0000u;
            if (u & 0x80000000u) u &= 0x0000ffffu;
            w[e] = u;
        }
        x[i] = v;
    }
}

// backward of ReLU on bf16: g = y > 0 ? dy : 0 (y = the ReLU's OUTPUT), 8 elements per lane and trip
__global__ __launch_bounds__(256) void relu_bwd_bf16_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y,
                                                            uint4* __restrict__ g, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        uint4 d = dy[i];
        const uint4 yy = y[i];
        uint32_t* w = (uint32_t*)&d;
        const uint32_t* q = (const uint32_t*)&yy;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t u = w[e];
            const uint32_t lo = q[e] & 0xffffu, hi = q[e] >> 16;
            if (lo == 0 || (lo & 0x8000u)) u &= 0xffff0000u;           // y <= 0 (or -0): no gradient
            if (hi == 0 || (hi & 0x8000u)) u &= 0x0000ffffu;
            w[e] = u;
        }
        g[i] = d;
    }
}

}  // namespace

extern "C" int omh_relu_bwd_bf16(const void* dy, const void* y, void* g, int64_t n, omh_stream_t stream) {
    if (!dy || !y || !g || n <= 0) return OMH_E_BADARG;
    if ((n & 7) || ((uintptr_t)dy & 15) || ((uintptr_t)y & 15) || ((uintptr_t)g & 15)) return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(relu_bwd_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)dy, (const uint4*)y, (uint4*)g, n / 8);
    return omh_launch_status();
}

extern "C" int omh_relu_bf16(void* x, int64_t n, omh_stream_t stream) {
    if (!x || n <= 0) return OMH_E_BADARG;
    if ((n & 7) || ((uintptr_t)x & 15)) return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(relu_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (uint4*)x,
                       n / 8);
    return omh_launch_status();
}

// workgroups: one pass over the voxels when there are few, else 8 per CU walking the rest
static unsigned rms_grid(int64_t P, int G) {
    const int64_t need = (P * G + 255) / 256;
    return (unsigned)(need < 2048 ? need : 2048);
}

template <bool XF32, int OUT3 = 0>
static int rms_silu_launch(const void* x, const float* gamma, void* y, int64_t P, int32_t C, int32_t do_silu,
                           omh_stream_t stream) {
    if (!x || !gamma || !y || P <= 0 || C <= 0) return OMH_E_BADARG;
    if ((C & 7) || C > 8 * 64 * 4 || (OUT3 == 2 && (C & 15))) return OMH_E_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return OMH_E_ALIGN;
    const int nch = C >> 3;
    hipStream_t s = (hipStream_t)stream;
    omh_clear_status();
    // lanes per voxel: enough that each lane holds <= 4 chunks, rounded to a power of two
    if (nch <= 16) {
        hipLaunchKernelGGL((rms_silu_kernel<4, XF32, OUT3>), dim3(rms_grid(P, 4)), dim3(256), 0, s,
                           x, gamma, (uint16_t*)y, P, C, do_silu);
    } else if (nch <= 64) {
        hipLaunchKernelGGL((rms_silu_kernel<16, XF32, OUT3>), dim3(rms_grid(P, 16)), dim3(256), 0, s,
                           x, gamma, (uint16_t*)y, P, C, do_silu);
    } else {
        hipLaunchKernelGGL((rms_silu_kernel<64, XF32, OUT3>), dim3(rms_grid(P, 64)), dim3(256), 0, s,
                           x, gamma, (uint16_t*)y, P, C, do_silu);
    }
    return omh_launch_status();
}

extern "C" int omh_rms_silu_cl(const void* x, const float* gamma, void* y, int64_t P, int32_t C, int32_t do_silu,
                               omh_stream_t stream) {
    return rms_silu_launch<false>(x, gamma, y, P, C, do_silu, stream);
}

extern "C" int omh_rms_silu_cl_f32in(const float* x, const float* gamma, void* y, int64_t P, int32_t C,
                                     int32_t do_silu, omh_stream_t stream) {
    return rms_silu_launch<true>(x, gamma, y, P, C, do_silu, stream);
}

extern "C" int omh_rms_silu_cl_split3(const float* x, const float* gamma, void* y, int64_t P, int32_t C,
                                      int32_t do_silu, omh_stream_t stream) {
    return rms_silu_launch<true, 1>(x, gamma, y, P, C, do_silu, stream);
}

extern "C" int omh_rms_silu_cl_pair(const float* x, const float* gamma, void* y, int64_t P, int32_t C,
                                    int32_t do_silu, omh_stream_t stream) {
    return rms_silu_launch<true, 2>(x, gamma, y, P, C, do_silu, stream);
}

extern "C" int omh_split3_f32(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t C, int32_t Cp,
                              int32_t pattern, omh_stream_t stream) {
    if (!x || !y || rows <= 0 || C <= 0 || Cp < C || pattern < 0 || pattern > 2) return OMH_E_BADARG;
    if ((Cp & 3) || (ldy & 3) || ldy < (pattern == 2 ? 2 : 3) * (int64_t)Cp || ldx < C) return OMH_E_SHAPE;
    if (pattern == 2 && (Cp & 15)) return OMH_E_SHAPE;               // whole 16-channel blocks
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return OMH_E_ALIGN;
    omh_clear_status();
    hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * (Cp / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (uint16_t*)y, ldy, rows, C, Cp, pattern);
    return omh_launch_status();
}

extern "C" int omh_nchw_to_cl_f32(const float* x, float* y, int32_t C, int32_t T, int32_t H, int32_t W, int32_t Cp,
                                  const float* mul, const float* add, int32_t t_total, int32_t t0, omh_stream_t stream) {
    if (!x || !y || C <= 0 || T <= 0 || H <= 0 || W <= 0 || Cp < C || t0 < 0 || t0 + T > t_total)
        return OMH_E_BADARG;
    const int64_t total = (int64_t)T * H * W * Cp;
    omh_clear_status();
    hipLaunchKernelGGL(nchw_to_cl_f32_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, C, T,
                       H, W, Cp, mul, add, t_total, t0);
    return omh_launch_status();
}

extern "C" int omh_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t R, int32_t L,
                                    float scale, omh_stream_t stream) {
    if (!x || !y || R <= 0 || L <= 0 || R > 0x7fffffff) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, L,
                       scale);
    return omh_launch_status();
}

extern "C" int omh_nchw_to_cl(const float* x, void* y, int32_t C, int32_t T, int32_t H, int32_t W, int32_t Cp,
                              const float* mul, const float* add, int32_t t_total, int32_t t0,
                              omh_stream_t stream) {
    if (!x || !y || C <= 0 || T <= 0 || H <= 0 || W <= 0 || Cp < C || t0 < 0 || t0 + T > t_total)
        return OMH_E_BADARG;
    const int64_t total = (int64_t)T * H * W * Cp;
    omh_clear_status();
    hipLaunchKernelGGL(nchw_to_cl_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (uint16_t*)y, C, T, H, W, Cp, mul, add, t_total, t0);
    return omh_launch_status();
}

extern "C" int omh_cl_to_nchw(const float* x, float* y, int32_t C, int32_t T, int32_t H, int32_t W, int32_t Cp,
                              const float* mul, const float* add, float lo, float hi, int32_t t_total,
                              int32_t t0, omh_stream_t stream) {
    if (!x || !y || C <= 0 || T <= 0 || H <= 0 || W <= 0 || Cp < C || t0 < 0 || t0 + T > t_total)
        return OMH_E_BADARG;
    const int64_t total = (int64_t)T * H * W * C;
    omh_clear_status();
    hipLaunchKernelGGL(cl_to_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, C, T,
                       H, W, Cp, mul, add, lo, hi, t_total, t0);
    return omh_launch_status();
}

extern "C" int omh_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t R, int32_t L, float scale,
                                omh_stream_t stream) {
    if (!x || !y || R <= 0 || L <= 0 || R > 0x7fffffff) return OMH_E_BADARG;
    omh_clear_status();
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (uint16_t*)y, ldy, L, scale);
    return omh_launch_status();
}
