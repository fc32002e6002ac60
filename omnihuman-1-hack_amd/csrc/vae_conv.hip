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
#include <stdlib.h>

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
                if (R && p.resid_f32) {                             // fp32 residual trunk
                    const float* Rf = (const float*)p.resid;
                    if (full) {
                        const float4 rr = *(const float4*)(Rf + off);
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) v[e] += Rf[off + e];
                    }
                } else if (R) {
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


// Epilogue shared by the wide kernels: + bias (+ residual), bf16 or fp32, optional frame interleave, through a
// per-wave LDS patch so that global accesses are row-major 16-byte vectors.  Tile-local row t stands for output
// voxel m_base + t and is stored iff row_lo <= t <= row_hi and 0 <= m < M.
template <bool OUT_F32, int WM, int WN, int MT, int NT, int STAGE_BYTES_>
__device__ __forceinline__ void wide_epilogue(const omh_conv_args& p, f32x16 (&acc)[MT][NT], unsigned char* smem,
                                              const int wave, const int lane, const int wm, const int wn,
                                              const int m_base, const int n0, const int M, const int row_lo,
                                              const int row_hi) {
    constexpr int STAGE_BYTES = STAGE_BYTES_;
    const int li = lane & 31, lh = lane >> 5;
    // ---------------- epilogue: + bias (+ residual), bf16 or fp32, optional frame interleave
    constexpr int PITCH = NT * 32 + 4;                                // floats; 100 mod 32 = 4: conflict-free b128 writes
    constexpr int VEC = OUT_F32 ? 4 : 8;
    constexpr int CPR = NT * 32 / VEC;                                // 16-byte chunks per strip row
    constexpr int PASSES = 32 * CPR / 64;
    static_assert(8 * 32 * PITCH * 4 <= 2 * STAGE_BYTES, "epilogue patch does not fit the staging LDS");
    float* ep = (float*)smem + wave * (32 * PITCH);
    const int HW = p.Hout * p.Wout;
    const int nsplit = p.split_n > 0 ? p.split_n : p.Cout;            // channels per output frame
    const int fmul = p.Cout / nsplit;                                 // frames produced per input frame
    float* Yf = (float*)p.y;
    uint16_t* Yh = (uint16_t*)p.y;
    const uint16_t* R = (const uint16_t*)p.resid;
    const bool vec_all = (nsplit % VEC) == 0;
#pragma unroll
    for (int im = 0; im < MT; ++im) {
        const int trow = (wm * MT + im) * 32;                         // tile-local first row of the strip
        const int mrow = m_base + trow;
        if (mrow >= M) break;                                         // wave-uniform
        // output offsets of this lane's chunks, and the residual fetched up front so that its latency
        // hides under the LDS transposition
        int64_t offs[PASSES];
        uint4 rres[PASSES];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int q = ps * 64 + lane;
            const int r = q / CPR, cc = (q - r * CPR) * VEC;
            const int m = mrow + r, n = n0 + wn * (NT * 32) + cc;
            if (fmul == 1) {
                offs[ps] = (int64_t)m * p.Cout + n;
            } else {
                const int to = m / HW, pix = m - to * HW;
                const int jf = n / nsplit, c = n - jf * nsplit;
                offs[ps] = ((int64_t)(to * fmul + jf) * HW + pix) * nsplit + c;
            }
            rres[ps] = make_uint4(0, 0, 0, 0);
            const bool row_ok = trow + r >= row_lo && trow + r <= row_hi && m >= 0 && m < M;
            if (R && row_ok && vec_all && n + VEC <= p.Cout) {
                if (!OUT_F32 && !p.resid_f32) rres[ps] = *(const uint4*)(R + offs[ps]);                     // 8 bf16
                else if (OUT_F32 && p.resid_f32) rres[ps] = *(const uint4*)((const float*)p.resid + offs[ps]);   // 4 fp32
            }
        }
#pragma unroll
        for (int in = 0; in < NT; ++in)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)(ep + li * PITCH + in * 32 + 8 * gq + 4 * lh) =
                    make_float4(acc[im][in][4 * gq], acc[im][in][4 * gq + 1], acc[im][in][4 * gq + 2], acc[im][in][4 * gq + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int q = ps * 64 + lane;
            const int r = q / CPR, cc = (q - r * CPR) * VEC;
            const int m = mrow + r, n = n0 + wn * (NT * 32) + cc;
            float v[VEC];
#pragma unroll
            for (int h = 0; h < VEC / 4; ++h) {
                const float4 t = *(const float4*)(ep + r * PITCH + cc + 4 * h);
                v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
            }
            if (trow + r < row_lo || trow + r > row_hi || m < 0 || m >= M || n >= p.Cout) continue;
            const bool full = vec_all && n + VEC <= p.Cout;
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) if (n + e < p.Cout) v[e] += p.bias[n + e];
            }
            const int64_t off = offs[ps];
            if (R) {
                const uint32_t rw[4] = {rres[ps].x, rres[ps].y, rres[ps].z, rres[ps].w};
                if (full && !OUT_F32 && !p.resid_f32) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += bf2f((uint16_t)((rw[(e >> 1) & 3] >> (16 * (e & 1))) & 0xffff));
                } else if (full && OUT_F32 && p.resid_f32) {           // fp32 residual trunk -> fp32 trunk
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += __uint_as_float(rw[e & 3]);
                } else if (p.resid_f32) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.Cout) v[e] += ((const float*)p.resid)[off + e];
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.Cout) v[e] += bf2f(R[off + e]);
                }
            }
            if (OUT_F32) {
                if (full) *(float4*)(Yf + off) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.Cout) Yf[off + e] = v[e];
                }
            } else {
                if (full) {
                    uint4 pk;
                    pk.x = pack_bf2(v[0], v[1]);
                    pk.y = pack_bf2(v[2], v[3]);
                    pk.z = pack_bf2(v[4 % VEC], v[5 % VEC]);
                    pk.w = pack_bf2(v[6 % VEC], v[7 % VEC]);
                    *(uint4*)(Yh + off) = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.Cout) Yh[off + e] = f2bf(v[e]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------------------
// Wide configuration: 8 waves, each owning a 64(m) x 96(n) patch (2 x 3 MFMA tiles), so that the tile's N
// extent is a multiple of 96 — the VAE's channel counts are 96/192/384/768 and a 128-wide tile idles a
// quarter of the matrix pipe on them.  WM x WN = 8x1 (512 x 96, Cout <= 96) or 4x2 (256 x 192).
// Same LDS image, swizzle and DMA staging as above; the main loop is the one of gemm_bf16.hip's big tile
// (fragments read one 16-wide k group ahead, one barrier per k-step, stage kt+2 issued right after the
// barrier), and the epilogue goes through a per-wave LDS patch so that global accesses are row-major
// 16-byte vectors (a [*, 96] bf16 output is one contiguous 6 KiB run per 32-row strip).
template <bool OUT_F32, int WM, int WN>
__global__ __launch_bounds__(512)
void conv_cl_wide_kernel(const omh_conv_args p, const int tiles_m, const int tiles_n) {
    constexpr int MT = 2, NT = 3, THREADS = 512;
    constexpr int WBM = WM * MT * 32, WBN = WN * NT * 32;
    constexpr int A_BYTES = WBM * BK * 2, B_BYTES = WBN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int CA = WBM * 8 / THREADS;                            // 16-byte chunks per thread: voxels
    constexpr int CB = (WBN * 8 + THREADS - 1) / THREADS;            //                            weights
    static_assert(WM * WN == 8 && WBM * 8 % THREADS == 0, "8 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_of(wid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * WBM, n0 = tn * WBN;
    const int M = p.Tout * p.Hout * p.Wout;
    const int K = p.KT * p.KH * p.KW * p.Cin;
    const int Heff = p.up2 ? 2 * p.Hin : p.Hin, Weff = p.up2 ? 2 * p.Win : p.Win;
    const int HWin = p.Hin * p.Win;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (int)((int64_t)p.Tin * p.Hin * p.Win * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, 0, (int)((int64_t)p.Cout * K * 2), 0x00020000);
    // chunk c = tid + 512 j -> tile row (tid>>3) + 64 j, physical slot tid&7: all rows of a thread share the
    // logical slot, i.e. one (tap, channel) position per thread and k-step.
    const int lslot = (tid & 7) ^ ((tid >> 4) & 7);
    int a_vox[CA], a_yx[CA];                                         // frame base voxel; (y<<16 | x) of tap (0,0)
#pragma unroll
    for (int j = 0; j < CA; ++j) {
        const int m = m0 + (tid >> 3) + 64 * j;
        const int mc = min(m, M - 1);
        const int xo = mc % p.Wout, yo = (mc / p.Wout) % p.Hout, to = mc / (p.Wout * p.Hout);
        a_vox[j] = to * p.stride_t * HWin;
        const int ay = (m < M) ? yo * p.stride_hw - p.pad_h : -16384;  // rows past M never pass the bounds test
        const int ax = xo * p.stride_hw - p.pad_w;
        a_yx[j] = (ay << 16) | (ax & 0xffff);
    }
    static_assert(CB <= 3, "weight chunks per thread");
    uint32_t voff_w[3];                                              // (fixed extent: a dependent one loses the host-side kernel stub)
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int row = (tid >> 3) + 64 * j;
        voff_w[j] = (row < WBN && n0 + row < p.Cout) ? (uint32_t)(((int64_t)(n0 + row) * K + lslot * 8) * 2) : 0x80000000u;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    // running (tap, channel) position of this thread's slot, advanced by BK per staged k-step
    int s_ci = lslot * 8, s_tx = 0, s_ty = 0, s_tt = 0, s_k = lslot * 8;
    while (s_ci >= p.Cin) {
        s_ci -= p.Cin;
        if (++s_tx == p.KW) { s_tx = 0; if (++s_ty == p.KH) { s_ty = 0; ++s_tt; } }
    }
#define WCONV_DMA(KT_, BUF)                                                                        \
    {                                                                                              \
        unsigned char* xa_ = smem + (BUF) * STAGE_BYTES;                                           \
        unsigned char* xb_ = xa_ + A_BYTES;                                                        \
        const bool kok = s_k < K;                                                                  \
        const int tapvox = s_tt * HWin;                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < CA; ++j_) {                                        \
            const int iy = (a_yx[j_] >> 16) + s_ty, ix = (int)(short)(a_yx[j_] & 0xffff) + s_tx;   \
            const bool ok = kok && iy >= 0 && iy < Heff && ix >= 0 && ix < Weff;                   \
            const int sy = p.up2 ? (iy >> 1) : iy, sx = p.up2 ? (ix >> 1) : ix;                    \
            const uint32_t xo_ = ok ? (uint32_t)(((a_vox[j_] + tapvox + sy * p.Win + sx) * p.Cin + s_ci) * 2) \
                                    : 0x80000000u;                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(xa_ + wave_lds + j_ * 8192), 16, xo_, 0, 0, 0); \
        }                                                                                          \
        const uint32_t wko = kok ? (uint32_t)((KT_) * BK * 2) : 0x80000000u;                       \
        _Pragma("unroll") for (int j_ = 0; j_ < CB; ++j_)                                          \
            if (wave * 64 + 512 * j_ < WBN * 8)                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(xb_ + wave_lds + j_ * 8192), 16, \
                                                         voff_w[j_] + wko, 0, 0, 0);               \
        s_k += BK; s_ci += BK;                                                                     \
        while (s_ci >= p.Cin) {                                                                    \
            s_ci -= p.Cin;                                                                         \
            if (++s_tx == p.KW) { s_tx = 0; if (++s_ty == p.KH) { s_ty = 0; ++s_tt; } }            \
        }                                                                                          \
    }
#define WCONV_FRAGS(WF, XF, STAGE, KK)                                                             \
    {                                                                                              \
        const unsigned char* xa_ = smem + (STAGE) * STAGE_BYTES;                                   \
        const unsigned char* xb_ = xa_ + A_BYTES;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_)                                          \
            WF[i_] = *(const bf16x8*)(xb_ + lds_slot_addr((wn * NT + i_) * 32 + li, 2 * (KK) + lh)); \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                          \
            XF[i_] = *(const bf16x8*)(xa_ + lds_slot_addr((wm * MT + i_) * 32 + li, 2 * (KK) + lh)); \
    }
#define WCONV_MFMAS(WF, XF)                                                                        \
    _Pragma("unroll") for (int im_ = 0; im_ < MT; ++im_)                                           \
        _Pragma("unroll") for (int in_ = 0; in_ < NT; ++in_)                                       \
            acc[im_][in_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[in_], XF[im_], acc[im_][in_], 0, 0, 0);
#define WCONV_INTERLEAVE()                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < MT + NT; ++s_) {                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT) > 0 ? MT * NT - (MT + NT) : 0, 0);

    WCONV_DMA(0, 0)
    if (nk > 1) {
        WCONV_DMA(1, 1)
        // wait for stage 0 only: stage 1's loads (issued later) may still be in flight
        if (wave * 64 + 512 * (CB - 1) < WBN * 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB - 1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bf16x8 wf0[NT], xf0[MT], wf1[NT], xf1[MT];
    WCONV_FRAGS(wf0, xf0, 0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        WCONV_FRAGS(wf1, xf1, buf, 1)
        WCONV_MFMAS(wf0, xf0)
        WCONV_INTERLEAVE()
        WCONV_FRAGS(wf0, xf0, buf, 2)
        WCONV_MFMAS(wf1, xf1)
        WCONV_INTERLEAVE()
        WCONV_FRAGS(wf1, xf1, buf, 3)
        WCONV_MFMAS(wf0, xf0)
        WCONV_INTERLEAVE()
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk) WCONV_DMA(kt + 2, buf)
        if (kt + 1 < nk) WCONV_FRAGS(wf0, xf0, buf ^ 1, 0)
        WCONV_MFMAS(wf1, xf1)
    }

    wide_epilogue<OUT_F32, WM, WN, MT, NT, STAGE_BYTES>(p, acc, smem, wave, lane, wm, wn, m0, n0, M, 0, WBM - 1);
}

// ---------------------------------------------------------------------------------------------------------
// kw-shared configuration for the layers that carry the VAE's time: 3x3x3 / 1x3x3 "same" convolutions with
// stride 1, no folded upsample, Cin a multiple of 32 (the residual-block convs at 96 / 192 / 384 channels).
// Every MFMA kernel of this build sits on the same wall — about 22 B/clk per CU from L2 into LDS (DESIGN.md
// section 8) — so what decides the rate is bytes per flop, and an implicit GEMM re-fetches every input voxel once per
// tap: 27 times.  Here a stage is one (kt, kh) tap pair and one block of 32 input channels for ALL THREE kw taps:
//   A slab  [WBM voxels][32 ch]   consecutive output voxels v0-1 .. v0+WBM-2 in (t, y, x) order, fetched ONCE;
//   B tile  [WBN couts][3 taps x 32 ch].
// Tap kw of output row j is slab row j + kw - 1 — the same LDS bytes read with a shifted row index — so the voxel
// traffic per flop drops 3x (512x96 tile: 83 -> 184 flop per staged byte; 256x192: 112 -> 177).  Rows 0 and WBM-1
// of a tile lack a neighbour: they are computed and dropped, tiles advance by WBM-2 voxels.  The x neighbours of
// the first / last voxel of an image row are the previous / next row's end voxels in linear order: their A
// fragments are zeroed by an unconditional AND with a per-lane word (a wave-uniform branch around it, skipping the
// 12 of 13 strips without an edge voxel, cut the k-step into small basic blocks, the MFMA / ds_read interleave was
// lost and the stage time became the SUM of its DMA, LDS and MFMA times: 825 TF instead of the figure below).
// Out-of-image rows (kh) and frames past the buffer arrive as zeros from the buffer descriptor, as before.
__device__ __forceinline__ uint32_t a3_addr(int row, int slot) {      // [rows][32 ch]: 64-byte rows, 4 slots
    return (uint32_t)(row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
}
__device__ __forceinline__ uint32_t b3_addr(int row, int slot) {      // [rows][3 x 32]: 192-byte rows, 12 slots
    return (uint32_t)(row * 192 + ((slot ^ ((row >> 2) & 3)) << 4));
}

template <bool OUT_F32, int WM, int WN, int NT_>
__global__ __launch_bounds__(512)
void conv_cl_kw3_kernel(const omh_conv_args p, const int tiles_m, const int tiles_n) {
    constexpr int MT = 2, NT = NT_, THREADS = 512;                    // NT = 1: Cout <= 32 (the 3-channel head conv)
    constexpr int WBM = WM * MT * 32, WBN = WN * NT * 32, VM = WBM - 2;
    constexpr int A_BYTES = WBM * 64, B_BYTES = WBN * 192, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int CA = WBM * 4 / THREADS;                             // A chunks per thread and stage (4 or 2)
    constexpr int BCH = WBN * 12;                                     // B chunks per stage (1152 or 2304)
    constexpr int CB = (BCH + THREADS - 1) / THREADS;                 // rounds of B chunks (3 or 5)
    static_assert(WM * WN == 8 && BCH % 64 == 0, "8 waves; whole wave instructions");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_of(wid, tiles_m, tiles_n, tm, tn);
    const int M = p.Tout * p.Hout * p.Wout;
    const int vbase = tm * VM - 1;                                    // voxel of slab row 0 / output row 0
    const int n0 = tn * WBN;
    const int K = p.KT * 9 * p.Cin;
    const int HW = p.Hin * p.Win;
    const int cblocks = p.Cin >> 5;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (int)((int64_t)p.Tin * HW * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, 0, (int)((int64_t)p.Cout * K * 2), 0x00020000);

    // A staging: chunk c = tid + 512 j -> slab row c>>2, physical slot c&3, fetched from logical slot
    // (c&3) ^ ((row>>2)&3), i.e. channels 8*slot .. +7 of the stage's 32-channel block
    int a_off[CA], a_y[CA];                                           // element offset of (frame t, row 0, x) + slot; y - pad
#pragma unroll
    for (int j = 0; j < CA; ++j) {
        const int c = tid + THREADS * j;
        const int row = c >> 2;
        const int ls = (c & 3) ^ ((row >> 2) & 3);
        const int v = vbase + row;
        const bool ok = v >= 0 && v < M;
        const int vc = min(max(v, 0), M - 1);
        const int xo = vc % p.Wout, yo = (vc / p.Wout) % p.Hout, to = vc / (p.Wout * p.Hout);
        a_off[j] = (to * HW + xo) * p.Cin + ls * 8;
        a_y[j] = ok ? yo - p.pad_h : -16384;                          // rows outside the volume never pass the bounds test
    }
    // B staging: chunk c -> cout row c/12, physical slot c%12; logical slot = tap kw (slot>>2) and 8-channel group
    static_assert(CB <= 5, "weight chunk rounds");
    uint32_t w_off[5];                                                // (fixed extent: a dependent one loses the host-side kernel stub)
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int c = tid + THREADS * j;
        const int row = c / 12, ps = c - row * 12;
        const int ls = ps ^ ((row >> 2) & 3);
        const int kw = ls >> 2, c8 = ls & 3;
        w_off[j] = (c < BCH && n0 + row < p.Cout) ? (uint32_t)((((int64_t)(n0 + row)) * K + kw * p.Cin + c8 * 8) * 2)
                                                   : 0x80000000u;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // fragment addresses: output row r of strip i reads slab rows r-1, r, r+1 (clamped: rows 0 / WBM-1 are dropped)
    uint32_t xaddr[MT][3][2];
    uint32_t edge0[MT], edge2[MT];                                    // AND words: 0 where the x neighbour is outside the image
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = (wm * MT + i) * 32 + li;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int row = min(max(r + kw - 1, 0), WBM - 1);
            xaddr[i][kw][0] = a3_addr(row, lh);
            xaddr[i][kw][1] = a3_addr(row, 2 + lh);
        }
        const int v = vbase + r;
        const int xo = (v >= 0 && v < M) ? v % p.Wout : 1;
        edge0[i] = (xo == 0) ? 0u : 0xffffffffu;
        edge2[i] = (xo == p.Wout - 1) ? 0u : 0xffffffffu;
    }
    uint32_t waddr[NT], wsw[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int row = (wn * NT + i) * 32 + li;
        waddr[i] = (uint32_t)(row * 192);
        wsw[i] = (uint32_t)((row >> 2) & 3);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ns = p.KT * 3 * cblocks;                                // stages: (kt, kh, channel block)
    int s_cb = 0, s_kh = 0, s_kt = 0;                                 // position of the NEXT stage to be staged
#define K3_DMA(BUF)                                                                                \
    {                                                                                              \
        unsigned char* xa_ = smem + (BUF) * STAGE_BYTES;                                           \
        unsigned char* xb_ = xa_ + A_BYTES;                                                        \
        const int aoff_ = s_kt * HW * p.Cin + s_cb * 32;                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < CA; ++j_) {                                        \
            const int iy = a_y[j_] + s_kh;                                                         \
            const uint32_t xo_ = (iy >= 0 && iy < p.Hin) ? (uint32_t)((a_off[j_] + aoff_ + iy * p.Win * p.Cin) * 2) \
                                                         : 0x80000000u;                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(xa_ + wave_lds + j_ * 8192), 16, xo_, 0, 0, 0); \
        }                                                                                          \
        const uint32_t wk_ = (uint32_t)((((s_kt * 3 + s_kh) * 3) * p.Cin + s_cb * 32) * 2);        \
        _Pragma("unroll") for (int j_ = 0; j_ < CB; ++j_)                                          \
            if (wave * 64 + THREADS * j_ < BCH)                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(xb_ + wave_lds + j_ * 8192), 16, \
                                                         w_off[j_] + wk_, 0, 0, 0);                \
        if (++s_cb == cblocks) { s_cb = 0; if (++s_kh == 3) { s_kh = 0; ++s_kt; } }                \
    }
    // group G of a stage: tap kw = G>>1, 16-channel half G&1
#define K3_FRAGS(WF, XF, STAGE, G)                                                                 \
    {                                                                                              \
        const unsigned char* xa_ = smem + (STAGE) * STAGE_BYTES;                                   \
        const unsigned char* xb_ = xa_ + A_BYTES;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_)                                          \
            WF[i_] = *(const bf16x8*)(xb_ + waddr[i_] + ((((uint32_t)(2 * (G)) + lh) ^ wsw[i_]) << 4)); \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) {                                        \
            u32x4 t_ = *(const u32x4*)(xa_ + xaddr[i_][(G) >> 1][(G) & 1]);                        \
            if (((G) >> 1) != 1) {                         /* compile-time: the dx = -1 / +1 taps */ \
                const uint32_t m_ = ((G) >> 1) == 0 ? edge0[i_] : edge2[i_];                       \
                t_[0] &= m_; t_[1] &= m_; t_[2] &= m_; t_[3] &= m_;                                \
            }                                                                                      \
            XF[i_] = __builtin_bit_cast(bf16x8, t_);                                               \
        }                                                                                          \
    }

    K3_DMA(0)
    if (ns > 1) K3_DMA(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8 wf0[NT], xf0[MT], wf1[NT], xf1[MT];
    K3_FRAGS(wf0, xf0, 0, 0)
    for (int st = 0; st < ns; ++st) {
        const int buf = st & 1;
        K3_FRAGS(wf1, xf1, buf, 1)
        WCONV_MFMAS(wf0, xf0)
        WCONV_INTERLEAVE()
        K3_FRAGS(wf0, xf0, buf, 2)
        WCONV_MFMAS(wf1, xf1)
        WCONV_INTERLEAVE()
        K3_FRAGS(wf1, xf1, buf, 3)
        WCONV_MFMAS(wf0, xf0)
        WCONV_INTERLEAVE()
        K3_FRAGS(wf0, xf0, buf, 4)
        WCONV_MFMAS(wf1, xf1)
        WCONV_INTERLEAVE()
        K3_FRAGS(wf1, xf1, buf, 5)
        WCONV_MFMAS(wf0, xf0)
        WCONV_INTERLEAVE()
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (st + 2 < ns) K3_DMA(buf)
        if (st + 1 < ns) K3_FRAGS(wf0, xf0, buf ^ 1, 0)
        WCONV_MFMAS(wf1, xf1)
    }
    wide_epilogue<OUT_F32, WM, WN, MT, NT, STAGE_BYTES>(p, acc, smem, wave, lane, wm, wn, vbase, n0, M, 1, WBM - 2);
}

template <bool OUT_F32, int WM, int WN>
int launch_wide(const omh_conv_args& a, int64_t M, hipStream_t s) {
    constexpr int WBM = WM * 64, WBN = WN * 96;
    constexpr int LDS = 2 * (WBM + WBN) * BK * 2;
    auto kern = conv_cl_wide_kernel<OUT_F32, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int tiles_m = (int)((M + WBM - 1) / WBM), tiles_n = (a.Cout + WBN - 1) / WBN;
    omh_clear_status();
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), LDS, s, a, tiles_m, tiles_n);
    return omh_launch_status();
}

template <bool OUT_F32, int WM, int WN, int NT = 3>
int launch_kw3(const omh_conv_args& a, int64_t M, hipStream_t s) {
    constexpr int WBM = WM * 64, WBN = WN * NT * 32;
    constexpr int LDS = 2 * (WBM * 64 + WBN * 192);
    auto kern = conv_cl_kw3_kernel<OUT_F32, WM, WN, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int tiles_m = (int)((M + WBM - 3) / (WBM - 2)), tiles_n = (a.Cout + WBN - 1) / WBN;
    omh_clear_status();
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), LDS, s, a, tiles_m, tiles_n);
    return omh_launch_status();
}

}  // namespace

// conv_w64.hip: the kw-shared convolution on one wave per SIMD, generated stage loop
bool omh_conv_w64_takes(const omh_conv_args& a);
bool omh_conv_w64_pair_takes(const omh_conv_args& a);
int omh_launch_conv_w64(const omh_conv_args& a, hipStream_t s);

// The tile family of a call (by the layer's geometry only) and whether the stream kernel takes it
struct ConvRoute { bool wide, w64; };
static ConvRoute conv_route(const omh_conv_args& a) {
    // wide tiles (N extent 96 / 192) once they give every CU most of a workgroup; the 128x128 tile otherwise.  The
    // count is taken for TWO output frames of this geometry whatever Tout is (the executors convolve 1 - 8 frames per
    // call): the kernel choice — and with it the accumulation order of every output value — then depends on the layer
    // only, not on how many frames a call carries, so a prefix of a clip decodes / encodes bit for bit like the
    // whole clip's first frames (tests/test_gpu_config5.py).
    const char* force = omh_opt(OMH_OPT_CONV_TILE);                     // "wide" / "small": test / benchmarking override
    const bool narrow = a.Cout <= 96;
    const int64_t Mn = (int64_t)2 * a.Hout * a.Wout;
    const int64_t wide_tiles = narrow ? (Mn + 511) / 512 : ((Mn + 255) / 256) * ((a.Cout + 191) / 192);
    const char* wmin = omh_opt(OMH_OPT_CONV_WIDE_MIN);                  // A/B timing of the threshold
    bool wide = wide_tiles >= (wmin ? atoi(wmin) : 96);              // e.g. 384 channels at 60 x 104: 98
    if (force && force[0] == 'w') wide = true;
    if (force && force[0] == 's') wide = false;
    if (((uintptr_t)a.resid & 15) || ((uintptr_t)a.bias & 3)) wide = false;
    // the residual-block convolutions: the one-wave-per-SIMD stream kernel (same values as the kw-shared 8-wave
    // kernel).  OMH_CONV_TILE=w64 forces it wherever it applies, wide / small exclude it, OMH_CONV_W64=0 turns it off.
    const char* w64e = omh_opt(OMH_OPT_CONV_W64);
    const bool w64_forced = force && force[0] == 'w' && force[1] == '6';
    const bool w64_ok = !(w64e && w64e[0] == '0') && (w64_forced || (!force && wide));
    const bool w64 = w64_ok && omh_conv_w64_takes(a);
    if (w64_forced) wide = true;                                      // not taken: the wide kernels
    return {wide, w64};
}

extern "C" int omh_conv_pair_supported(const omh_conv_args* args) {
    if (!args || !args->pair) return 0;
    const omh_conv_args& a = *args;
    if (a.Tin <= 0 || a.Hin <= 0 || a.Win <= 0 || a.Cin <= 0 || a.Tout <= 0 || a.Hout <= 0 || a.Wout <= 0 || a.Cout <= 0) return 0;
    const ConvRoute route = conv_route(a);
    return (route.w64 && omh_conv_w64_pair_takes(a)) ? 1 : 0;
}

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
    const ConvRoute route = conv_route(a);
    if (a.pair) {                                           // split-bf16 pairs: the stream kernel or nothing (ABI v10)
        if (a.norm_gamma) {
            // the next layer's RMS norm + SiLU, written as pairs too ([M, 2 Cout]): in the stream's epilogue at Cout = 96,
            // as omh_rms_silu_cl_pair on y behind the convolution otherwise — the same values either way
            if (!a.norm_out || a.split_n > 0) return OMH_E_BADARG;
            if (((uintptr_t)a.norm_out & 15) || ((uintptr_t)a.norm_gamma & 3)) return OMH_E_ALIGN;
            const char* fe = omh_opt(OMH_OPT_CONV_FUSE_NORM);
            if (!(a.Cout == 96 && !(fe && fe[0] == '0'))) {
                omh_conv_args b = a;
                b.norm_gamma = nullptr; b.norm_out = nullptr; b.norm_only = 0;
                const int rc = omh_conv_cl_bf16(&b, stream);
                if (rc) return rc;
                return omh_rms_silu_cl_pair((const float*)a.y, a.norm_gamma, a.norm_out, M, a.Cout, 1, stream);
            }
        }
        if (!(route.w64 && omh_conv_w64_pair_takes(a))) return OMH_E_SHAPE;
        return omh_launch_conv_w64(a, (hipStream_t)stream);
    }
    if (a.norm_gamma) {
        // the next layer's RMS norm + SiLU (ABI v7): in the stream kernel's epilogue when one wave holds all the
        // channels of a voxel (Cout = 96), as a second launch over y otherwise — the two write the same values
        if (!a.norm_out || a.split_n > 0) return OMH_E_BADARG;
        if (((uintptr_t)a.norm_out & 15) || ((uintptr_t)a.norm_gamma & 3)) return OMH_E_ALIGN;
        const char* fe = omh_opt(OMH_OPT_CONV_FUSE_NORM);                // "0": never fused (tests / A/B timing)
        if (!(route.w64 && a.Cout == 96 && !(fe && fe[0] == '0'))) {
            omh_conv_args b = a;
            b.norm_gamma = nullptr; b.norm_out = nullptr; b.norm_only = 0;
            const int rc = omh_conv_cl_bf16(&b, stream);
            if (rc) return rc;
            return a.out_f32 ? omh_rms_silu_cl_f32in((const float*)a.y, a.norm_gamma, a.norm_out, M, a.Cout, 1, stream)
                             : omh_rms_silu_cl(a.y, a.norm_gamma, a.norm_out, M, a.Cout, 1, stream);
        }
    }
    if (route.w64) return omh_launch_conv_w64(a, (hipStream_t)stream);
    const bool wide = route.wide, narrow = a.Cout <= 96;
    if (wide) {
        hipStream_t s = (hipStream_t)stream;
        // 3x3 "same" convolutions with stride 1 at >= 32 channels: the kw-shared kernel (3x less voxel traffic)
        const char* kw3e = omh_opt(OMH_OPT_CONV_KW3);                     // "0": off (tests / A/B timing)
        const bool kw3 = !(kw3e && kw3e[0] == '0') && a.KW == 3 && a.KH == 3 && (a.KT == 3 || a.KT == 1) &&
                         a.stride_hw == 1 && a.stride_t == 1 && !a.up2 && a.pad_h == 1 && a.pad_w == 1 &&
                         a.Hout == a.Hin && a.Wout == a.Win && (a.Cin & 31) == 0 && a.split_n == 0 && a.Wout >= 3;
        if (kw3) {
            if (a.Cout <= 32) return a.out_f32 ? launch_kw3<true, 8, 1, 1>(a, M, s) : launch_kw3<false, 8, 1, 1>(a, M, s);
            if (narrow) return a.out_f32 ? launch_kw3<true, 8, 1>(a, M, s) : launch_kw3<false, 8, 1>(a, M, s);
            return a.out_f32 ? launch_kw3<true, 4, 2>(a, M, s) : launch_kw3<false, 4, 2>(a, M, s);
        }
        if (narrow) return a.out_f32 ? launch_wide<true, 8, 1>(a, M, s) : launch_wide<false, 8, 1>(a, M, s);
        return a.out_f32 ? launch_wide<true, 4, 2>(a, M, s) : launch_wide<false, 4, 2>(a, M, s);
    }
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (a.Cout + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n);
    omh_clear_status();
    if (a.out_f32)
        hipLaunchKernelGGL(conv_cl_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL(conv_cl_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, tiles_m, tiles_n);
    return omh_launch_status();
}
