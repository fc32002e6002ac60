// The VAE's 3x3(x3) "same" convolutions (vae.py:17-36 under the ResidualBlocks of :186-220) on one wave per SIMD: 4 waves,
// each a 128(voxels) x 96(couts) patch, three LDS stages, the stage loop and the epilogue a generated instruction
// stream (gen_conv_w64.py -> conv_w64_asm.inc; read its header).  Same contract and the same values, bit for bit, as
// vae_conv.hip's conv_cl_kw3_kernel for the layers it takes (omh_conv_w64_takes); this file computes the per-lane
// address table, the descriptors and the scalar arguments.
#include "omh_common.h"
#include "conv_w64_asm.inc"
#include <stdlib.h>

namespace {

constexpr int STAGE = 53312, A_OFF = 64;                  // [64 pad | A slab | B tile | 2 KiB sink], 3 stages
enum { CFG_P = 0, CFG_Q = 1 };                            // 512 x 96 (Cout = 96) / 256 x 192 (Cout % 192 == 0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffLL ? 0xffffffffu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)n, 0x00020000);
}
__device__ __forceinline__ uint64_t pack2(uint32_t lo, uint32_t hi) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32);
}
// 16-byte-slot swizzle of both LDS tiles, as in conv_cl_kw3_kernel.  ((row >> 1) & 3 would put 8 consecutive 64-byte
// rows on 8 distinct 16-byte bank groups instead of 4: measured, no difference — 762 vs 768 us on the 480x832 layer.)
#define SWZ(row) (((row) >> 2) & 3)
__device__ __forceinline__ uint32_t a3_addr(int row, int slot) {      // [rows][32 ch]: 64-byte rows, 4 slots
    return (uint32_t)(row * 64 + ((slot ^ SWZ(row)) << 4));
}

// v / d and v % d for 0 <= v < 2^24 through the float reciprocal (one correction step): the lane table needs ~40 of
// them per tile, and the compiler's exact 32-bit division is ~35 instructions each
__device__ __forceinline__ void divmod24(int v, int d, float rcp, int& q, int& r) {
    q = (int)((float)v * rcp);
    r = v - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

// NORM (P only): the stream's epilogue also writes the next layer's RMS norm + SiLU of the output (omh_conv_args.norm_*)
// PAIR (fp32 out, no norm): the split-bf16 pair stream — the loader sees an ordinary tensor of Cin = 2 C channels, the
// stage loop pairs the 16-channel halves of a stage as hi.hi, lo.hi, hi.lo (gen_conv_w64.py: main_loop_pair)
template <int CFG, bool OUT_F32, bool NORM, bool PAIR = false>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_cl_w64_kernel(const omh_conv_args p, const int tiles_m, const int tiles_n) {
    // + per wave [gamma(96) | bias(96)] fp32 (768 bytes): the epilogue's per-cout vectors come from LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE + 4 * 768];
    constexpr int WBM = CFG == CFG_P ? 512 : 256, WBN = CFG == CFG_P ? 96 : 192, VM = WBM - 2;
    constexpr int A_BYTES = WBM * 64;
    constexpr int NA = CFG == CFG_P ? 8 : 4, NB = CFG == CFG_P ? 5 : 9;
    constexpr int ES = OUT_F32 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = CFG == CFG_P ? w : (w >> 1), wn = CFG == CFG_P ? 0 : (w & 1);
    const int li = lane & 31, lh = lane >> 5;

    // Persistent when the grid is smaller than the tile count: workgroup b walks tiles b, b + grid, ... (the grid is a
    // multiple of 8, so a workgroup's tiles stay on its XCD's slice of xcd_remap's order).  The barrier keeps a fast
    // wave's table write off the stage buffers a slower wave of the previous tile is still reading.
    for (int vb = blockIdx.x; vb < tiles_m * tiles_n; vb += gridDim.x) {
    if (vb != (int)blockIdx.x) __syncthreads();
    const int wid = xcd_remap(vb, tiles_m * tiles_n);
    int tm, tn;
    tile_of(wid, tiles_m, tiles_n, tm, tn);
    const int M = p.Tout * p.Hout * p.Wout;
    const int vbase = tm * VM - 1;                                    // voxel of slab row 0 / tile row 0
    const int n0 = tn * WBN;
    const int K = p.KT * 9 * p.Cin;
    const int HW = p.Hin * p.Win;

    const float rcp_w = 1.0f / (float)p.Wout, rcp_h = 1.0f / (float)p.Hout;
    const int up = p.up2 ? 1 : 0;                                     // folded nearest-2x upsample: Hout = 2 Hin, Wout = 2 Win

    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    uint32_t* tab = (uint32_t*)smem + tid * 44;                       // this lane's table (gen_conv_w64.py's v[12:55])

    // A staging: piece q of this wave = slab rows 16 (first + q) .. +15; lane -> row lane >> 2, physical 16-byte slot
    // lane & 3, fetched from the logical slot (lane & 3) ^ ((row >> 2) & 3) = channels 8 slot .. +7 of the block
    const int a_first = (CFG == CFG_P ? 8 : 4) * w;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t off2 = 0, ay = (uint32_t)-16384;
        if (q < NA) {
            const int row = (a_first + q) * 16 + (lane >> 2);
            const int ls = (lane & 3) ^ SWZ(row);
            const int v = vbase + row;
            const bool ok = v >= 0 && v < M;
            const int vc = min(max(v, 0), M - 1);
            int xo, yo, to, vy;
            divmod24(vc, p.Wout, rcp_w, vy, xo);
            divmod24(vy, p.Hout, rcp_h, to, yo);
            off2 = (uint32_t)(((to * HW + (xo >> up)) * p.Cin + ls * 8) * 2);   // up2: input column x >> 1 (the slab holds the UPSAMPLED row)
            ay = ok ? (uint32_t)(yo - p.pad_h) : (uint32_t)-16384;   // rows outside the volume never pass the bounds test
        }
        tab[q] = off2;
        tab[8 + q] = ay;
    }
    // B staging: 16-byte chunk c of the tile -> cout row c / 12, physical slot c % 12; logical slot = tap kw (slot >> 2)
    // and 8-channel group (slot & 3).  P: 18 pieces, wave w takes 4 w .. 4 w + 3 and 16 + w (w < 2; the others' fifth
    // piece goes to the sink with an out-of-range source); Q: 36 pieces, 9 per wave.
    uint32_t blast_rel;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        uint32_t woff = 0x80000000u;
        if (q < NB) {
            int pb = CFG == CFG_P ? (q < 4 ? 4 * w + q : (w < 2 ? 16 + w : -1)) : 9 * w + q;
            if (pb >= 0) {
                const int c = pb * 64 + lane;
                const int row = c / 12, ps = c - row * 12;
                const int ls = ps ^ SWZ(row);
                const int kw = ls >> 2, c8 = ls & 3;
                if (row < WBN && n0 + row < p.Cout) woff = (uint32_t)((((int64_t)(n0 + row)) * K + kw * p.Cin + c8 * 8) * 2);
            }
        }
        tab[16 + q] = woff;
    }
    if (CFG == CFG_P) blast_rel = w < 2 ? (uint32_t)(A_OFF + A_BYTES + (16 + w) * 1024) : (uint32_t)(A_OFF + A_BYTES + 18432 + (w - 2) * 1024);
    else blast_rel = (uint32_t)(A_OFF + A_BYTES + (9 * w + 8) * 1024);
    // fragment addresses (buffer 0): voxel tile j adds 2048, cout tile i adds 6144, tap kw of the weights adds 64
    {
        const int r = wm * 128 + li;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int half = 0; half < 2; ++half)
                tab[25 + 2 * kw + half] = lds0 + A_OFF + a3_addr(r + kw - 1, 2 * half + lh);
        const int row = wn * 96 + li;
#pragma unroll
        for (int half = 0; half < 2; ++half)
            tab[31 + half] = lds0 + A_OFF + A_BYTES + (uint32_t)(row * 192) + (uint32_t)((((2 * half + lh) ^ SWZ(row))) << 4);
    }
    uint32_t rowmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = wm * 128 + 32 * j + li;
        const int v = vbase + t;
        int xo = 1, vy;
        if (v >= 0 && v < M) divmod24(v, p.Wout, rcp_w, vy, xo);
        tab[33 + j] = xo == 0 ? 0u : 0xffffffffu;                     // no x - 1 neighbour
        tab[37 + j] = xo == p.Wout - 1 ? 0u : 0xffffffffu;            // no x + 1 neighbour
        if (t >= 1 && t <= WBM - 2) rowmask |= 1u << j;               // rows 0 and WBM - 1 of a tile are dropped
    }
    tab[41] = (uint32_t)(((int64_t)(vbase + wm * 128 + li) * p.Cout + n0 + wn * 96 + 8 * lh) * ES);   // < 0: outside the descriptor
    tab[42] = rowmask;
    tab[43] = 0;
    const uint32_t vtab = lds0 + (uint32_t)tid * 176u;
    {   // this wave's gamma (norm kinds) and bias runs; ordered by the stream's first barrier
        float* vec = (float*)(smem + 3 * STAGE + w * 768);
        const int c0 = n0 + wn * 96;
        for (int c = lane; c < 96; c += 64) {
            if (NORM) vec[c] = p.norm_gamma[c];
            vec[96 + c] = (p.bias && c0 + c < p.Cout) ? p.bias[c0 + c] : 0.0f;
        }
    }

    const __amdgpu_buffer_rsrc_t rx = rsrc_of(p.x, (int64_t)p.Tin * HW * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(p.w, (int64_t)p.Cout * K * 2);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(p.y, (NORM && p.norm_only) ? 0 : (int64_t)M * p.Cout * ES);   // norm_only: y's stores fall outside
    const __amdgpu_buffer_rsrc_t rnorm = rsrc_of(p.norm_out, NORM ? (int64_t)M * p.Cout * (PAIR ? 4 : 2) : 0);   // pair: [M, 2 Cout] bf16
    const __amdgpu_buffer_rsrc_t rres = rsrc_of(p.resid, p.resid ? (int64_t)M * p.Cout * ES : 0);
    const int col0 = n0 + wn * 96;
    const __amdgpu_buffer_rsrc_t rbias = rsrc_of(p.bias ? p.bias + col0 : nullptr, p.bias ? (int64_t)(p.Cout - col0) * 4 : 0);

    const int cblocks = p.Cin >> 5;
    const uint64_t p0 = pack2(lds0 + A_OFF + (uint32_t)(a_first * 1024),
                              lds0 + A_OFF + A_BYTES + (uint32_t)((CFG == CFG_P ? 4 : 9) * w * 1024));
    const uint64_t p1 = pack2((uint32_t)(p.Win * p.Cin * 2), (uint32_t)(p.Hin << up));   // row bytes of the INPUT, rows of the upsampled image
    const uint64_t p2 = pack2((uint32_t)cblocks, (uint32_t)(p.KT * 3 * cblocks));
    const uint64_t p3 = pack2((uint32_t)(HW * p.Cin * 2), (uint32_t)(64 - p.Cin * 2));
    const uint64_t p4 = pack2((uint32_t)(4 * p.Cin + 64), lds0 + blast_rel);
    const uint64_t p5 = pack2((uint32_t)(32 * p.Cout * ES), (uint32_t)up);
    const uint64_t p6 = pack2(lds0 + 3u * STAGE + (uint32_t)w * 768u, __float_as_uint(sqrtf((float)p.Cout)));   // the wave's [gamma | bias] in LDS, sqrt(C)

#define OMH_CW64_RUN(ASM)                                                                                              \
    asm volatile(ASM                                                                                                   \
                 :                                                                                                     \
                 : [vtab] "v"(vtab), [rx] "s"(rx), [rw] "s"(rw), [ry] "s"(ry), [rres] "s"(rres), [rbias] "s"(rbias),   \
                   [rnorm] "s"(rnorm),                                                                                 \
                   [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1), [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3),          \
                   [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5), [p6] "{s[72:73]}"(p6)                                  \
                 : OMH_CONV_W64_CLOBBERS)
    if (PAIR) {
        if (CFG == CFG_P && NORM) OMH_CW64_RUN(OMH_CONV_W64_ASM_P_F32_PAIR_NORM);
        else if (CFG == CFG_P) OMH_CW64_RUN(OMH_CONV_W64_ASM_P_F32_PAIR);
        else OMH_CW64_RUN(OMH_CONV_W64_ASM_Q_F32_PAIR);
    }
    else if (CFG == CFG_P && NORM) { if (OUT_F32) OMH_CW64_RUN(OMH_CONV_W64_ASM_P_F32_NORM); else OMH_CW64_RUN(OMH_CONV_W64_ASM_P_BF16_NORM); }
    else if (CFG == CFG_P) { if (OUT_F32) OMH_CW64_RUN(OMH_CONV_W64_ASM_P_F32); else OMH_CW64_RUN(OMH_CONV_W64_ASM_P_BF16); }
    else { if (OUT_F32) OMH_CW64_RUN(OMH_CONV_W64_ASM_Q_F32); else OMH_CW64_RUN(OMH_CONV_W64_ASM_Q_BF16); }
#undef OMH_CW64_RUN
    }
}

template <int CFG, bool OUT_F32, bool NORM = false, bool PAIR = false>
int launch_cw64(const omh_conv_args& a, int64_t M, hipStream_t s) {
    constexpr int WBM = CFG == CFG_P ? 512 : 256, WBN = CFG == CFG_P ? 96 : 192;
    const int tiles_m = (int)((M + WBM - 3) / (WBM - 2)), tiles_n = a.Cout / WBN;
    // One workgroup per CU walks the tiles (no re-launch between a CU's tiles: +1-2 % on every layer, one box, interleaved:
    // 794 -> 784 us at 96 channels, 735 -> 722 at 192, whole decode 255.6 -> 258.3 frames/s); OMH_CONV_PERSIST=0: one
    // workgroup per tile (A/B timing).
    const char* pe = omh_opt(OMH_OPT_CONV_PERSIST);
    int grid = tiles_m * tiles_n;
    if (!(pe && pe[0] == '0')) {
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
            cus = cus / 8 * 8;
        }
        if (grid > cus) grid = cus;
    }
    omh_clear_status();
    hipLaunchKernelGGL((conv_cl_w64_kernel<CFG, OUT_F32, NORM, PAIR>), dim3(grid), dim3(256), 0, s, a, tiles_m, tiles_n);
    return omh_launch_status();
}

}  // namespace

// 3x3 taps in (h, w), 1 or 3 in t, stride 1, "same" padding (optionally through the folded 2x upsample), no frame interleave, Cin % 32 == 0,
// Cout = 96 or a multiple of 192, at least 15 stages (13 are peeled at the head of the stream), the residual (if any) in the output's type, 32-bit byte offsets.
bool omh_conv_w64_takes(const omh_conv_args& a) {
    const int64_t M = (int64_t)a.Tout * a.Hout * a.Wout;
    const int es = a.out_f32 ? 4 : 2;
    const int up = a.up2 ? 1 : 0;                            // (round 3: also through the folded nearest-2x upsample)
    if (up) { const char* ue = omh_opt(OMH_OPT_CONV_W64_UP2); if (ue && ue[0] == '0') return false; }   // A/B timing
    return a.KW == 3 && a.KH == 3 && (a.KT == 3 || a.KT == 1) && a.stride_hw == 1 && a.stride_t == 1 &&
           a.pad_h == 1 && a.pad_w == 1 && a.Hout == (a.Hin << up) && a.Wout == (a.Win << up) && (a.Cin & 31) == 0 && a.split_n == 0 &&
           a.Wout >= 3 && (a.Cout == 96 || a.Cout % 192 == 0) && a.KT * 3 * (a.Cin >> 5) >= 15 &&
           (!a.resid || (a.resid_f32 != 0) == (a.out_f32 != 0)) && (((uintptr_t)a.resid) & 15) == 0 &&
           (((uintptr_t)a.bias) & 15) == 0 && (int64_t)a.Win * a.Cin * 2 < (1 << 24) && (a.Hin << up) < 16384 && M < (1 << 24) &&
           (M + 1024) * a.Cout * es < 0x7fffffffLL && (int64_t)a.Tin * a.Hin * a.Win * a.Cin * 2 < 0x7fffffffLL &&
           (int64_t)a.Cout * a.KT * 9 * a.Cin * 2 < 0x7fffffffLL;
}

// The split-bf16 pair stream: what the stream takes, fp32 output (and residual), no fused norm, an even number of at
// least 16 stages (14 are peeled, the rolled loop and the tail run two at a time: the X fragment sets swap roles per tap)
bool omh_conv_w64_pair_takes(const omh_conv_args& a) {
    const int ns = a.KT * 3 * (a.Cin >> 5);
    // (a fused next-layer norm — written in the pair layout too — at Cout = 96 only, as for the bf16 streams)
    return a.pair && a.out_f32 && (!a.norm_gamma || a.Cout == 96) && (ns & 1) == 0 && ns >= 16 && omh_conv_w64_takes(a);
}

int omh_launch_conv_w64(const omh_conv_args& a, hipStream_t s) {
    const int64_t M = (int64_t)a.Tout * a.Hout * a.Wout;
    if (a.pair && a.Cout == 96 && a.norm_gamma) return launch_cw64<CFG_P, true, true, true>(a, M, s);
    if (a.pair) return a.Cout == 96 ? launch_cw64<CFG_P, true, false, true>(a, M, s) : launch_cw64<CFG_Q, true, false, true>(a, M, s);
    if (a.Cout == 96 && a.norm_gamma) return a.out_f32 ? launch_cw64<CFG_P, true, true>(a, M, s) : launch_cw64<CFG_P, false, true>(a, M, s);
    if (a.Cout == 96) return a.out_f32 ? launch_cw64<CFG_P, true>(a, M, s) : launch_cw64<CFG_P, false>(a, M, s);
    return a.out_f32 ? launch_cw64<CFG_Q, true>(a, M, s) : launch_cw64<CFG_Q, false>(a, M, s);
}
