// Implicit-GEMM causal convolution for the Wan 3D VAE on gfx950.
//
// Replaces aten/cuDNN conv under the reference's CausalConv3d (3x3x3, 3x1x1,
// 1x1x1), Conv2d 3x3 (stride 1, stride 2 with right/bottom zero pad) and the
// nearest-exact 2x upsample that precedes a Conv2d
// (seaweed_apt/wan/modules/vae.py:17-36, 57-63, 76-96, 127-141, 196-199).
//
// Activations are channels-last bf16 [T, H, W, C]; the temporal history a
// causal conv needs (the reference's feat_cache slots, vae.py:205-217) is the
// first frames of the input buffer, so this kernel never pads in time.
//   y[(to,y,x)][co] = bias[co] + resid + sum_{kt,dy,dx,ci}
//        w[co][((kt*KH+dy)*KW+dx)*Cin + ci] * x[to*st + kt][src(y*s+dy-ph)][src(x*s+dx-pw)][ci]
// src() is the identity, or >>1 when the 2x nearest upsample is folded in;
// out-of-range rows/columns contribute zero.
//
// GEMM view: M = Tout*Hout*Wout output voxels, N = Cout, K = taps*Cin with the
// k index running (tap, ci); same 128x128x64 MFMA tile, LDS swizzle and LDS-DMA
// double buffering as the small configuration of gemm_bf16.hip — only the
// A-operand source differs: each 16-byte chunk (8 channels of one tap) is
// fetched from the shifted voxel (out-of-image taps get an out-of-range buffer
// offset and arrive as zeros), so the 27-fold re-read of the input stays in L2.
#include "omh_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;

__device__ __forceinline__ uint32_t lds_slot_addr(int row, int slot) {
    return (uint32_t)(row * (BK * 2) + ((slot ^ ((row >> 1) & 7)) << 4));
}

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2)
void conv_cl_kernel(const omh_conv_args p, const int tiles_m, const int tiles_n) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_of(wid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = p.Tout * p.Hout * p.Wout;
    const int K = p.KT * p.KH * p.KW * p.Cin;
    const int Heff = p.up2 ? 2 * p.Hin : p.Hin, Weff = p.up2 ? 2 * p.Win : p.Win;

    const __bf16* __restrict__ X = (const __bf16*)p.x;
    const __bf16* __restrict__ Wt = (const __bf16*)p.w;

    // staging: chunk c = tid + 256 j lands at LDS byte 16 c of the tile (row c>>3, physical slot c&7) and
    // is fetched from logical slot (c&7) ^ ((row>>1)&7).  The four rows of a thread are 32 apart, so they
    // share the logical slot: one (tap, channel) decode per thread and k-step.
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)X, 0, (int)((int64_t)p.Tin * p.Hin * p.Win * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)((int64_t)p.Cout * K * 2), 0x00020000);
    const int lslot = (tid & 7) ^ ((tid >> 4) & 7);
    int a_t[4], a_y[4], a_x[4];
    bool a_ok[4];
    uint32_t voff_w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (tid + 256 * j) >> 3;
        const int m = m0 + row;
        a_ok[j] = m < M;
        const int mc = min(m, M - 1);
        const int xo = mc % p.Wout, yo = (mc / p.Wout) % p.Hout, to = mc / (p.Wout * p.Hout);
        a_t[j] = to * p.stride_t;
        a_y[j] = yo * p.stride_hw - p.pad_h;
        a_x[j] = xo * p.stride_hw - p.pad_w;
        voff_w[j] = (n0 + row < p.Cout) ? (uint32_t)(((int64_t)(n0 + row) * K + lslot * 8) * 2) : 0x80000000u;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;

#define CONV_DMA1(J, XA, XB)                                                                       \
    {                                                                                              \
        const int iy = a_y[J] + tap_y, ix = a_x[J] + tap_x;                                        \
        const bool ok = kok && a_ok[J] && iy >= 0 && iy < Heff && ix >= 0 && ix < Weff;            \
        const int sy = p.up2 ? (iy >> 1) : iy, sx = p.up2 ? (ix >> 1) : ix;                        \
        const uint32_t xo_ = ok ? (uint32_t)(((((a_t[J] + tap_t) * p.Hin + sy) * p.Win + sx) * p.Cin + ci) * 2) \
                                : 0x80000000u;                                                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)((XA) + wave_lds + (J) * 4096), 16, xo_, 0, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)((XB) + wave_lds + (J) * 4096), 16,               \
                                                 voff_w[J] + wko, 0, 0, 0);                        \
    }
#define CONV_DMA(KT_, BUF)                                                                         \
    {                                                                                              \
        const int kc = (KT_) * BK + lslot * 8;                                                     \
        const bool kok = kc < K;                                                                   \
        const uint32_t wko = kok ? (uint32_t)((KT_) * BK * 2) : 0x80000000u;                       \
        const int kcc = min(kc, K - 8);                                                            \
        const int tap = kcc / p.Cin, ci = kcc - tap * p.Cin;                                       \
        const int tap_x = tap % p.KW, tap_y = (tap / p.KW) % p.KH, tap_t = tap / (p.KW * p.KH);    \
        unsigned char* xa_ = smem + (BUF) * 2 * TILE_BYTES;                                        \
        unsigned char* xb_ = xa_ + TILE_BYTES;                                                     \
        CONV_DMA1(0, xa_, xb_) CONV_DMA1(1, xa_, xb_) CONV_DMA1(2, xa_, xb_) CONV_DMA1(3, xa_, xb_) \
    }

    CONV_DMA(0, 0)
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) CONV_DMA(kt + 1, buf ^ 1)
        const unsigned char* xa = smem + buf * 2 * TILE_BYTES;   // voxels (m)
        const unsigned char* xb = xa + TILE_BYTES;               // weights (n)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = *(const bf16x8*)(xb + lds_slot_addr(wn * 64 + i * 32 + li, 2 * kk + lh));
                xf[i] = *(const bf16x8*)(xa + lds_slot_addr(wm * 64 + i * 32 + li, 2 * kk + lh));
            }
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int in = 0; in < 2; ++in)
                    acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in], xf[im], acc[im][in], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---------------- epilogue: + bias (+ residual), bf16 or fp32, optional frame interleave
    const int HW = p.Hout * p.Wout;
    const int nsplit = p.split_n > 0 ? p.split_n : p.Cout;          // channels per output frame
    const int fmul = p.Cout / nsplit;                                 // frames produced per input frame
    float* Yf = (float*)p.y;
    uint16_t* Yh = (uint16_t*)p.y;
    const uint16_t* R = (const uint16_t*)p.resid;
    const bool vec_ok = (nsplit & 3) == 0;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int m = m0 + wm * 64 + im * 32 + li;
        if (m >= M) continue;
        const int to = m / HW, pix = m - to * HW;
#pragma unroll
        for (int in = 0; in < 2; ++in) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * 64 + in * 32 + 8 * gq + 4 * lh;
                if (n >= p.Cout) continue;
                const int jf = n / nsplit, c = n - jf * nsplit;
                const int64_t off = ((int64_t)(to * fmul + jf) * HW + pix) * nsplit + c;
                const bool full = vec_ok && (n + 3 < p.Cout);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[im][in][4 * gq + e];
                    if (p.bias && n + e < p.Cout) v[e] += p.bias[n + e];
                }
                if (R) {
                    if (full) {
                        const uint2 rr = *(const uint2*)(R + off);
                        v[0] += bf2f((uint16_t)(rr.x & 0xffff)); v[1] += bf2f((uint16_t)(rr.x >> 16));
                        v[2] += bf2f((uint16_t)(rr.y & 0xffff)); v[3] += bf2f((uint16_t)(rr.y >> 16));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) v[e] += bf2f(R[off + e]);
                    }
                }
                if (OUT_F32) {
                    if (full) *(float4*)(Yf + off) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) Yf[off + e] = v[e];
                    }
                } else {
                    if (full) {
                        uint2 pk;
                        pk.x = pack_bf2(v[0], v[1]);
                        pk.y = pack_bf2(v[2], v[3]);
                        *(uint2*)(Yh + off) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) Yh[off + e] = f2bf(v[e]);
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int omh_conv_cl_bf16(const omh_conv_args* args, omh_stream_t stream) {
    if (!args || !args->x || !args->w || !args->y) return OMH_E_BADARG;
    const omh_conv_args& a = *args;
    if (a.Tin <= 0 || a.Hin <= 0 || a.Win <= 0 || a.Cin <= 0 || a.Tout <= 0 || a.Hout <= 0 || a.Wout <= 0 ||
        a.Cout <= 0 || a.KT <= 0 || a.KH <= 0 || a.KW <= 0 || a.stride_t <= 0 || a.stride_hw <= 0)
        return OMH_E_BADARG;
    if (a.Cin & 7) return OMH_E_SHAPE;                      // 16-byte channel chunks
    if ((a.Tout - 1) * a.stride_t + a.KT > a.Tin) return OMH_E_SHAPE;   // history frames must be in the buffer
    if (a.split_n > 0 && (a.Cout % a.split_n)) return OMH_E_SHAPE;
    if (((uintptr_t)a.x & 15) || ((uintptr_t)a.w & 15) || ((uintptr_t)a.y & 15) || ((uintptr_t)a.resid & 7))
        return OMH_E_ALIGN;
    const int64_t M = (int64_t)a.Tout * a.Hout * a.Wout;
    if (M > 0x7fffffff) return OMH_E_SHAPE;
    // 32-bit buffer offsets
    if ((int64_t)a.Tin * a.Hin * a.Win * a.Cin * 2 >= 0x7fffffffLL) return OMH_E_SHAPE;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (a.Cout + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n);
    omh_clear_status();
    if (a.out_f32)
        hipLaunchKernelGGL(conv_cl_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL(conv_cl_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, tiles_m, tiles_n);
    return omh_launch_status();
}
