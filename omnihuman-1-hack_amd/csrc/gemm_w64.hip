// bf16 GEMM C = A B^T on a 256(m) x 384(n) x 64 workgroup tile: 4 waves, one per SIMD, 384 accumulators per wave, the
// k loop a generated instruction stream (gen_gemm_w64.py -> gemm_w64_asm.inc; read its header).  Same contract as
// gemm_bf16.hip's omh_gemm_bf16 for the shapes it takes (see omh_gemm_w64_takes); this file only computes
// descriptors and per-lane offsets.
#include "omh_common.h"
#include "gemm_w64_asm.inc"
#include <stdlib.h>

namespace {

constexpr int TM = 256, TN = 384, BK = 64;
constexpr int TN192 = 192;       // the gated-residual stream with the old C tile requested during the k loop (round 4)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffLL ? 0xffffffffu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)n, 0x00020000);
}

enum { K_F32 = 0, K_BF16 = 1, K_GELU = 2, K_RESID = 3, K_GELUBWD = 4, K_BF16M = 5, K_GELUAUX = 6,
       K_RESID192 = 7, K_F32_192 = 8, K_BF16_192 = 9,                        // K_RESID192 .. K_BF16_192: the 256 x 192 tile
       K_QKV = 10,          // 256 x 384: columns < n_split -> C bf16 (the BF16 stream), columns >= n_split -> aux TRANSPOSED (BF16VT)
       // round 6: 256 x 256, the k loop with two k tiles of operands in flight in registers (gen_gemm_w64.py: main_loop_pgr)
       K_F32_P = 11, K_BF16_P = 12, K_GELU_P = 13, K_RESID_P = 14,
       K_ABL_A = 15, K_ABL_B = 16, K_ABL_C = 17, K_ABL_D = 18, K_ABL_E = 19,        // timing-only ablation builds of K_F32_P
       K_F32_M16 = 20 };    // timing-only: the 256 x 384 fp32 stream with every MFMA as two v_mfma_f32_16x16x32_bf16
constexpr bool is_n192(int kind) { return kind >= K_RESID192 && kind <= K_BF16_192; }
constexpr bool is_p256(int kind) { return kind >= K_F32_P && kind <= K_ABL_E; }
constexpr int TN256 = 256;
constexpr int tnw_of(int kind) { return is_n192(kind) ? TN192 : is_p256(kind) ? TN256 : TN; }

// two wave-uniform 32-bit scalars in one SGPR pair (inline asm takes at most 30 operands)
__device__ __forceinline__ uint64_t pack2(uint32_t lo, uint32_t hi) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32);
}

struct W64Tile { uint32_t sxb, swb, coff; int m0, n0; };
// Split K (ABI v9, K_F32_192 only): `n` slices of the contraction, slice s starts kbytes * s into every A / B row and
// writes its fp32 partial tile cbytes * s into C (= the workspace: n slices of [M, ldc]).  n = 1: the whole product.
struct W64Split { int n; uint32_t kbytes, cbytes; };

// Tile `idx` of this launch: XCD-contiguous work order (xcd_remap) walking 8 m-tiles x all n-tiles (tile_of), and the
// byte offsets of this WAVE's first X / W rows (wave w stages X rows 64 w .. and W rows 96 w ..).
template <int TNW>
__device__ __forceinline__ W64Tile w64_tile(const omh_gemm_args& p, int idx, int tiles_m, int tiles_n, int w, const W64Split sp) {
    int wid = xcd_remap(idx, tiles_m * tiles_n * sp.n);
    int sl = 0;
    if (sp.n > 1) { sl = wid / (tiles_m * tiles_n); wid -= sl * (tiles_m * tiles_n); }     // slice-major: a slice's tiles share rows
    int tm, tn;
    tile_of(wid, tiles_m, tiles_n, tm, tn, 8);
    W64Tile t;
    t.m0 = tm * TM; t.n0 = tn * TNW;
    t.sxb = (uint32_t)(((int64_t)(t.m0 + w * 64) * p.lda) * 2) + (uint32_t)sl * sp.kbytes;
    t.swb = (uint32_t)(((int64_t)(t.n0 + w * (TNW / 4)) * p.ldb) * 2) + (uint32_t)sl * sp.kbytes;
    t.coff = (uint32_t)sl * sp.cbytes;
    return t;
}

// Persistent: workgroup b computes tiles b, b + gridDim.x, ...  The prologue DMA of tile t + 1 is issued from inside
// tile t's stream, just before its epilogue (the LDS stages are free then), so it lands under the epilogue and the
// epilogue's stores drain under tile t + 1's k loop.
template <int KIND>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_bf16_nt_w64_kernel(const omh_gemm_args p, const int tiles_m, const int tiles_n, const W64Split sp) {
    constexpr int TNW = tnw_of(KIND);                                       // tile width; a wave owns TNW / 2 columns
    constexpr int WBYTES = TNW * 128, STAGE_B = 32768 + WBYTES;             // W tile, one stage (X 32 KiB | W)
    // (p256: both stages are filled whole by a tile's prologue, so the epilogue's column vectors get 16 KiB of their own)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B + (is_p256(KIND) ? 16384 : 0)];
    constexpr int ES = (KIND == K_BF16 || KIND == K_GELU || KIND == K_BF16_192 || KIND == K_GELUBWD || KIND == K_BF16M ||
                        KIND == K_GELUAUX || KIND == K_QKV || KIND == K_BF16_P || KIND == K_GELU_P) ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int r = lane & 31, h = lane >> 5;
    const int total = tiles_m * tiles_n * sp.n;

    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int x3 = (r >> 1) & 7;
    const uint32_t xh = (uint32_t)(x3 >> 1);
    const uint32_t hb = (uint32_t)((h ^ (x3 & 1)) << 4);
    const uint32_t xab = lds0 + (uint32_t)(wm * 16384 + r * 128) + hb;
    const uint32_t wab = lds0 + 32768u + (uint32_t)(wn * (WBYTES / 2) + r * 128) + hb;
    // LDS-DMA: a 1 KiB piece = 8 rows x 8 slots; lane -> (row lane >> 3, physical slot lane & 7), fetched from the
    // logical slot (lane & 7) ^ ((row >> 1) & 7) with (row >> 1) & 7 = 4 (piece & 1) + (lane >> 4)
    const int pr = lane >> 3, ps = lane & 7;
    const uint32_t vox0 = (uint32_t)((pr * p.lda + ((ps ^ (lane >> 4)) * 8)) * 2);
    const uint32_t vox1 = (uint32_t)((pr * p.lda + ((ps ^ ((lane >> 4) + 4)) * 8)) * 2);
    const uint32_t vow0 = (uint32_t)((pr * p.ldb + ((ps ^ (lane >> 4)) * 8)) * 2);
    const uint32_t vow1 = (uint32_t)((pr * p.ldb + ((ps ^ ((lane >> 4) + 4)) * 8)) * 2);
    // epilogue: after the permlane widening lane (r, h) holds, per tile (i, j) and run p, the 8 columns
    // 32 i + 16 p + 8 h ... of row 32 j + r of the wave's 128 x 192 patch; the tile's origin goes in the soffset
    const uint32_t voc = (uint32_t)(((int64_t)(wm * 128 + r) * p.ldc + 8 * h) * ES);
    const uint32_t vlane = (uint32_t)lane;

    // (split K: p.K is ONE slice's length, the rows hold sp.n of them; C = the workspace's sp.n slices)
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.A, (((int64_t)p.M - 1) * p.lda + (int64_t)p.K * sp.n) * 2);
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(p.B, (((int64_t)p.N - 1) * p.ldb + (int64_t)p.K * sp.n) * 2);
    // (K_QKV: C holds the columns below n_split only — the descriptor must end there, or the rows past M of a ragged
    // last m-tile would land inside it)
    const int c_cols = KIND == K_QKV ? p.n_split : p.N;
    const __amdgpu_buffer_rsrc_t rc = rsrc_of(p.C, (((int64_t)p.M - 1) * p.ldc + c_cols) * ES + (int64_t)(sp.n - 1) * sp.cbytes);
    // K_QKV, columns >= n_split: out^T [N - n_split, ldaux] bf16 in aux; lane (r, h) of wave (wm, wn) starts at row
    // wn * 192 + r, column wm * 128 + 8 h of the tile's patch; a lane's bias is that of its row
    const __amdgpu_buffer_rsrc_t rvt = rsrc_of(p.aux, KIND == K_QKV ? (((int64_t)(p.N - p.n_split) - 1) * p.ldaux + p.M) * 2 : 0);
    // (the stream derives the lane's part from vlane; the wave's part goes into the tile origin below)
    const __amdgpu_buffer_rsrc_t rbias = rsrc_of(p.bias, (p.bias && p.bias_mode == OMH_BIAS_N) ? (int64_t)p.N * 4 : 0);
    // K_BF16M: per-ROW bias (OMH_BIAS_M: the V^T projection, weights in the row slot)
    const __amdgpu_buffer_rsrc_t rbm = rsrc_of(p.bias, (KIND == K_BF16M && p.bias) ? (int64_t)p.M * 4 : 0);
    constexpr bool RES = KIND == K_RESID || KIND == K_RESID192 || KIND == K_RESID_P;
    const __amdgpu_buffer_rsrc_t rg0 = rsrc_of(p.gate0, (RES && p.gate0) ? (int64_t)p.N * 4 : 0);
    const bool has_g1 = RES && p.gate1 != nullptr;
    const int grows = has_g1 ? p.gate_rows : 1;
    const int nb = has_g1 ? (p.M + grows - 1) / grows : 1;
    const __amdgpu_buffer_rsrc_t rg1 = rsrc_of(p.gate1, has_g1 ? ((int64_t)(nb - 1) * p.gate1_stride + p.N) * 4 : 0);
    // resid192 (training epilogues, ABI v5): the old values may come from another tensor (c_in), y = bf16(acc + bias)
    // goes to aux (same row pitch as C; absent: an empty descriptor drops the stores)
    const __amdgpu_buffer_rsrc_t rcin = rsrc_of(p.c_in ? (const void*)p.c_in : (const void*)p.C, (((int64_t)p.M - 1) * p.ldc + p.N) * 4);
    const __amdgpu_buffer_rsrc_t raux = rsrc_of(p.aux, p.aux ? (((int64_t)p.M - 1) * p.ldc + p.N) * 2 : 0);   // (GELU_BWD: the pre-activations)

    const uint64_t p0 = pack2(lds0 + (uint32_t)w * 8192u, lds0 + (uint32_t)w * (uint32_t)(WBYTES / 4));
    const uint64_t p2 = pack2((uint32_t)(8 * p.lda * 2), (uint32_t)(8 * p.ldb * 2));
    const uint64_t p4 = pack2((uint32_t)(32 * p.ldc * ES), (uint32_t)p.N);
    const uint64_t p6 = pack2((uint32_t)(p.gate1_stride * 4), __float_as_uint(p.gate_const));
    const uint32_t nk = (uint32_t)(p.K / BK);
    const uint32_t region = lds0 + (uint32_t)(is_p256(KIND) ? 2 * STAGE_B : STAGE_B + 32768) + (uint32_t)w * 4096u;   // column vectors: stage 1's W region

    int idx = blockIdx.x;
    W64Tile t = w64_tile<TNW>(p, idx, tiles_m, tiles_n, w, sp);
#ifdef OMH_GEMM_W64_ASM_PRO256
    if (is_p256(KIND)) {
        const uint64_t p8 = pack2(t.sxb, t.swb);
        asm volatile(OMH_GEMM_W64_ASM_PRO256
                     :
                     : [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0), [vow1] "v"(vow1), [ra] "s"(ra), [rb] "s"(rb),
                       [p0] "{s[60:61]}"(p0), [p2] "{s[64:65]}"(p2), [p8] "{s[76:77]}"(p8)
                     : "memory", "scc", "s80", "s81", "s82");
    } else
#endif
    if (is_n192(KIND)) {
        const uint64_t p8 = pack2(t.sxb, t.swb);
        asm volatile(OMH_GEMM_W64_ASM_PRO192
                     :
                     : [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0), [vow1] "v"(vow1), [ra] "s"(ra), [rb] "s"(rb),
                       [p0] "{s[60:61]}"(p0), [p2] "{s[64:65]}"(p2), [p8] "{s[76:77]}"(p8)
                     : "memory", "scc", "s80", "s81", "s82");
    } else {
        const uint64_t p8 = pack2(t.sxb, t.swb);
        asm volatile(OMH_GEMM_W64_ASM_PRO
                     :
                     : [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0), [vow1] "v"(vow1), [ra] "s"(ra), [rb] "s"(rb),
                       [p0] "{s[60:61]}"(p0), [p2] "{s[64:65]}"(p2), [p8] "{s[76:77]}"(p8)
                     : "memory", "scc", "s80", "s81", "s82");
    }
#pragma nounroll
    while (true) {
        const int nidx = idx + (int)gridDim.x;
        const bool has_next = nidx < total;
        const W64Tile tn = w64_tile<TNW>(p, has_next ? nidx : idx, tiles_m, tiles_n, w, sp);
        const int mw = t.m0 + wm * 128, nw = t.n0 + wn * (TNW / 2);
        const int blo = has_g1 ? mw / grows : 0;                            // batch index of the patch's first row
        // rows of the wave's patch from here on take the gate of batch blo + 1
        const uint32_t mb = has_g1 ? (uint32_t)((blo + 1) * grows - mw) : 0xffffffffu;
        const uint64_t p1 = pack2(t.sxb, t.swb);
        const uint64_t p3 = pack2(nk, (uint32_t)(((int64_t)t.m0 * p.ldc + nw) * ES) + t.coff);
        const uint64_t p5 = pack2(mb, (uint32_t)(((int64_t)blo * p.gate1_stride + nw) * 4));
        const uint64_t p7 = pack2((uint32_t)(nw * 4), region);
        const uint64_t p8 = pack2(tn.sxb, tn.swb);
        const uint64_t p9 = pack2(has_next ? 1u : 0u, 0u);
#define OMH_GW64_RUN(ASM)                                                                                              \
    asm volatile(ASM                                                                                                   \
                 :                                                                                                     \
                 : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),  \
                   [vow1] "v"(vow1), [voc] "v"(voc), [vlane] "v"(vlane), [ra] "s"(ra), [rb] "s"(rb), [rc] "s"(rc),      \
                   [rbias] "s"(rbias), [rg0] "s"(rg0), [rg1] "s"(rg1), [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),   \
                   [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5),          \
                   [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)           \
                 : OMH_GEMM_W64_CLOBBERS)
        if (KIND == K_F32) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32);
        else if (KIND == K_BF16) OMH_GW64_RUN(OMH_GEMM_W64_ASM_BF16);
        else if (KIND == K_GELU) OMH_GW64_RUN(OMH_GEMM_W64_ASM_GELU);
        else if (KIND == K_RESID) OMH_GW64_RUN(OMH_GEMM_W64_ASM_RESID);
        else if (KIND == K_BF16M) {
            const uint32_t vbm = (uint32_t)((t.m0 + wm * 128 + r) * 4);
            asm volatile(OMH_GEMM_W64_ASM_BF16M
                 :
                 : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),
                   [vow1] "v"(vow1), [voc] "v"(voc), [vlane] "v"(vlane), [vbm] "v"(vbm), [ra] "s"(ra), [rb] "s"(rb), [rc] "s"(rc),
                   [rbias] "s"(rbias), [rg0] "s"(rg0), [rg1] "s"(rg1), [rbm] "s"(rbm),
                   [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),
                   [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5),
                   [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)
                 : OMH_GEMM_W64_CLOBBERS);
        } else if (KIND == K_GELUAUX)                  // gelu + the pre-activation to aux (C's shape and row pitch)
            asm volatile(OMH_GEMM_W64_ASM_GELUAUX
                 :
                 : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),
                   [vow1] "v"(vow1), [voc] "v"(voc), [vlane] "v"(vlane), [ra] "s"(ra), [rb] "s"(rb), [rc] "s"(rc),
                   [rbias] "s"(rbias), [rg0] "s"(rg0), [rg1] "s"(rg1), [raux] "s"(raux),
                   [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),
                   [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5),
                   [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)
                 : OMH_GEMM_W64_CLOBBERS);
        else if (KIND == K_GELUBWD)
            asm volatile(OMH_GEMM_W64_ASM_GELUBWD
                 :
                 : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),
                   [vow1] "v"(vow1), [voc] "v"(voc), [vlane] "v"(vlane), [ra] "s"(ra), [rb] "s"(rb), [rc] "s"(rc),
                   [rbias] "s"(rbias), [rg0] "s"(rg0), [rg1] "s"(rg1), [raux] "s"(raux),
                   [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),
                   [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5),
                   [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)
                 : OMH_GEMM_W64_CLOBBERS);
        else if (KIND == K_QKV) {
            // (neither stream reads the gate descriptors: rbias stands in for them, so that they cost no SGPRs here)
            if (t.n0 < p.n_split) {
                asm volatile(OMH_GEMM_W64_ASM_BF16
                     :
                     : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),
                       [vow1] "v"(vow1), [voc] "v"(voc), [vlane] "v"(vlane), [ra] "s"(ra), [rb] "s"(rb), [rc] "s"(rc),
                       [rbias] "s"(rbias), [rg0] "s"(rbias), [rg1] "s"(rbias), [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),
                       [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5),
                       [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)
                     : OMH_GEMM_W64_CLOBBERS);
            } else {
                const int left = p.M - mw;                                  // rows of the wave's patch inside M
                const uint64_t p3v = pack2(nk, (uint32_t)(((int64_t)(nw - p.n_split) * p.ldaux + mw) * 2));
                const uint64_t p4v = pack2((uint32_t)(32 * p.ldaux * 2), (uint32_t)p.N);
                const uint64_t p5v = pack2((uint32_t)(left > 0 ? left : 0), 0u);
                asm volatile(OMH_GEMM_W64_ASM_BF16VT
                     :
                     : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),
                       [vow1] "v"(vow1), [vlane] "v"(vlane), [ra] "s"(ra), [rb] "s"(rb),
                       [rc] "s"(rvt), [rbias] "s"(rbias), [rg0] "s"(rbias), [rg1] "s"(rbias),
                       [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),
                       [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3v), [p4] "{s[68:69]}"(p4v), [p5] "{s[70:71]}"(p5v),
                       [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)
                     : OMH_GEMM_W64_CLOBBERS);
            }
        }
#ifdef OMH_GEMM_W64_ASM_PRO256
        else if (KIND == K_F32_P) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_P256);
        else if (KIND == K_BF16_P) OMH_GW64_RUN(OMH_GEMM_W64_ASM_BF16_P256);
        else if (KIND == K_GELU_P) OMH_GW64_RUN(OMH_GEMM_W64_ASM_GELU_P256);
        else if (KIND == K_RESID_P) OMH_GW64_RUN(OMH_GEMM_W64_ASM_RESID_P256);
#endif
#ifdef OMH_GEMM_W64_ASM_F32_M16
        else if (KIND == K_F32_M16) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_M16);
#endif
#ifdef OMH_GEMM_W64_ASM_F32_P256_A
        else if (KIND == K_ABL_A) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_P256_A);
        else if (KIND == K_ABL_B) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_P256_B);
        else if (KIND == K_ABL_C) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_P256_C);
        else if (KIND == K_ABL_D) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_P256_D);
        else if (KIND == K_ABL_E) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_P256_E);
#endif
        else if (KIND == K_F32_192) OMH_GW64_RUN(OMH_GEMM_W64_ASM_F32_192);
        else if (KIND == K_BF16_192) OMH_GW64_RUN(OMH_GEMM_W64_ASM_BF16_192);
        else
            asm volatile(OMH_GEMM_W64_ASM_RESID192
                 :
                 : [xab] "v"(xab), [wab] "v"(wab), [xh] "v"(xh), [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0),
                   [vow1] "v"(vow1), [voc] "v"(voc), [vlane] "v"(vlane), [ra] "s"(ra), [rb] "s"(rb), [rc] "s"(rc),
                   [rbias] "s"(rbias), [rg0] "s"(rg0), [rg1] "s"(rg1), [rcin] "s"(rcin), [raux] "s"(raux),
                   [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1),
                   [p2] "{s[64:65]}"(p2), [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5),
                   [p6] "{s[72:73]}"(p6), [p7] "{s[74:75]}"(p7), [p8] "{s[76:77]}"(p8), [p9] "{s[78:79]}"(p9)
                 : OMH_GEMM_W64_CLOBBERS);
#undef OMH_GW64_RUN
        if (!has_next) break;
        idx = nidx;
        t = tn;
    }
}

template <int KIND>
int launch_w64(const omh_gemm_args& a, hipStream_t stream, const W64Split sp = W64Split{1, 0u, 0u}) {
    constexpr int TNW = tnw_of(KIND);
    const int tiles_m = (a.M + TM - 1) / TM, tiles_n = (a.N + TNW - 1) / TNW;
    const int total = tiles_m * tiles_n * sp.n;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        ncu -= ncu % 8;                                                   // whole XCD rounds keep xcd_remap's chunks aligned
        if (ncu < 8) ncu = 256;
    }
    hipLaunchKernelGGL(gemm_bf16_nt_w64_kernel<KIND>, dim3(total < ncu ? total : ncu), dim3(256), 0, stream, a, tiles_m, tiles_n, sp);
    return 0;
}

}  // namespace

// The shapes the stream kernel takes (everything else stays on gemm_bf16.hip's kernels): one batch, row-major B,
// K % 64 == 0, K >= 256, N % 8 == 0, M >= 256, epilogues F32 / BF16 / GELU_BF16 / RESID with an [N] bias or none, RESID's per-row
// gate changing at most once inside a wave's 128 rows (gate_rows >= 128), 32-bit byte offsets everywhere.
bool omh_gemm_w64_takes(const omh_gemm_args& a) {
    const bool gbwd = a.epilogue == OMH_EPI_GELU_BWD_BF16;       // aux = pre-activations with C's shape and row pitch
    if (gbwd && (!a.aux || a.ldaux != a.ldc || ((uintptr_t)a.aux & 15))) return false;
    // GELU with the pre-activation written to aux (the training forward's FFN-up): same shape and row pitch as C
    if (a.epilogue == OMH_EPI_GELU_BF16 && a.aux && (a.ldaux != a.ldc || ((uintptr_t)a.aux & 15))) return false;
    const bool bf16_out = a.epilogue == OMH_EPI_BF16 || a.epilogue == OMH_EPI_GELU_BF16 || gbwd;
    if (!(bf16_out || a.epilogue == OMH_EPI_F32 || a.epilogue == OMH_EPI_RESID)) return false;
    if (a.bias && a.bias_mode == OMH_BIAS_M) return false;
    if (a.epilogue == OMH_EPI_RESID && a.gate1 && (a.gate_rows < 128 || a.gate1_stride * 4 >= 0x7fffffffLL ||
                                                   ((int64_t)(a.M / a.gate_rows) * a.gate1_stride + a.N) * 4 >= 0x7fffffffLL))
        return false;
    const int es = bf16_out ? 2 : 4;
    return a.batch == 1 && !a.b_kmajor && (a.K % BK) == 0 && a.K >= 4 * BK && (a.N % 8) == 0 && a.M >= TM &&
           (a.ldc % (16 / es)) == 0 && (((uintptr_t)a.C) & 15) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0 &&
           (((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.B) & 15) == 0 &&
           ((int64_t)a.M + TM) * a.lda * 2 < 0x7fffffffLL && ((int64_t)a.N + TN) * a.ldb * 2 < 0x7fffffffLL &&
           ((int64_t)a.M + TM) * a.ldc * es < 0xffffffffLL;
}

// The 256 x 192 gated-residual stream (old C requested during the first 12 k steps): in-place RESID shapes the big
// stream takes, with 16 <= K / 64 (12 peeled steps + the tail logic) — chosen by omh_gemm_bf16 for short contractions,
// where the 256 x 384 stream's read-modify-write epilogue is as long as its k loop.
bool omh_gemm_w64_r192_takes(const omh_gemm_args& a) {
    if (a.aux && (a.ldaux != a.ldc || ((uintptr_t)a.aux & 15))) return false;       // aux offsets = C offsets / 2
    if (a.c_in && ((uintptr_t)a.c_in & 15)) return false;
    return a.epilogue == OMH_EPI_RESID && omh_gemm_w64_takes(a) && a.K >= 16 * BK;
}
int omh_launch_gemm_w64_r192(const omh_gemm_args& a, hipStream_t stream) { return launch_w64<K_RESID192>(a, stream); }
// bf16 output with a per-ROW bias (the V^T projection) on the 256 x 384 stream
bool omh_gemm_w64_bf16m_takes(const omh_gemm_args& a) {
    if (a.epilogue != OMH_EPI_BF16 || !a.bias || a.bias_mode != OMH_BIAS_M) return false;
    omh_gemm_args b = a;
    b.bias_mode = OMH_BIAS_N;
    return omh_gemm_w64_takes(b);
}
int omh_launch_gemm_w64_bf16m(const omh_gemm_args& a, hipStream_t stream) { return launch_w64<K_BF16M>(a, stream); }
// The same 256 x 192 tile for the plain fp32 / bf16 epilogues: twice the tiles of the 256 x 384 stream, for products whose
// 256 x 384 tiles would fill less than half of the chip (the training step's M = 6 240: 100 tiles -> 200).
bool omh_gemm_w64_n192_takes(const omh_gemm_args& a) {
    return (a.epilogue == OMH_EPI_F32 || a.epilogue == OMH_EPI_BF16) && omh_gemm_w64_takes(a);
}
int omh_launch_gemm_w64_n192(const omh_gemm_args& a, hipStream_t stream) {
    return a.epilogue == OMH_EPI_F32 ? launch_w64<K_F32_192>(a, stream) : launch_w64<K_BF16_192>(a, stream);
}

// The fused q | k | v projection (OMH_EPI_BF16_SPLIT_T, ABI v10): one launch over N = 3 dim columns; tiles below n_split
// run the BF16 stream into C, tiles from n_split on run the operand-swapped stream and store their result transposed
// into aux (V^T [dim, ld]).  h is read once and 128 x 12 = 1 536 tiles are six whole rounds of the chip where the
// separate V^T launch had 516 tiles of 256 x 384 = two rounds + four tiles (or three rounds of 256 x 256).
bool omh_gemm_w64_qkv_takes(const omh_gemm_args& a) {
    if (a.epilogue != OMH_EPI_BF16_SPLIT_T || !a.aux || a.n_split <= 0 || a.n_split >= a.N) return false;
    if ((a.n_split % TN) || (a.M & 7) || a.ldaux < a.M || (a.ldaux & 7) || ((uintptr_t)a.aux & 15)) return false;
    if ((int64_t)(a.N - a.n_split) * a.ldaux * 2 >= 0x7fffffffLL) return false;
    omh_gemm_args b = a;
    b.epilogue = OMH_EPI_BF16; b.aux = nullptr;
    return omh_gemm_w64_takes(b);
}
int omh_launch_gemm_w64_qkv(const omh_gemm_args& a, hipStream_t stream) { return launch_w64<K_QKV>(a, stream); }

// The 256 x 256 streams whose k loop keeps two k tiles of operands in flight in registers (round 6 EXPERIMENT: compiled
// only from a gemm_w64_asm.inc generated with OMH_GW64_P256=1; measured slower than the shipped streams and not part of
// the library — gen_gemm_w64.py, profiles/r06_gemm_p256.txt): the plain epilogues of the big stream, in place.
bool omh_gemm_w64_p256_takes(const omh_gemm_args& a) {
#ifndef OMH_GEMM_W64_ASM_PRO256
    (void)a;
    return false;
#else
    if (a.aux || a.c_in) return false;
    if (!(a.epilogue == OMH_EPI_F32 || a.epilogue == OMH_EPI_BF16 || a.epilogue == OMH_EPI_GELU_BF16 || a.epilogue == OMH_EPI_RESID))
        return false;
    return omh_gemm_w64_takes(a);
#endif
}
int omh_launch_gemm_w64_p256(const omh_gemm_args& a, hipStream_t stream) {
#ifndef OMH_GEMM_W64_ASM_PRO256
    (void)a; (void)stream;
    return OMH_E_SHAPE;
#else
#ifdef OMH_GEMM_W64_ASM_F32_M16
    { const char* m = omh_opt(OMH_OPT_GEMM_W64_P256); if (m && m[0] == 'm' && a.epilogue == OMH_EPI_F32) return launch_w64<K_F32_M16>(a, stream); }
#endif
#ifdef OMH_GEMM_W64_ASM_F32_P256_A
    const char* ab = omh_opt(OMH_OPT_GEMM_W64_P256);             // "a".."e": the timing-only ablations (results are garbage)
    if (ab && a.epilogue == OMH_EPI_F32) switch (ab[0]) {
        case 'a': return launch_w64<K_ABL_A>(a, stream);
        case 'b': return launch_w64<K_ABL_B>(a, stream);
        case 'c': return launch_w64<K_ABL_C>(a, stream);
        case 'd': return launch_w64<K_ABL_D>(a, stream);
        case 'e': return launch_w64<K_ABL_E>(a, stream);
        default: break;
    }
#endif
    switch (a.epilogue) {
        case OMH_EPI_F32:       return launch_w64<K_F32_P>(a, stream);
        case OMH_EPI_BF16:      return launch_w64<K_BF16_P>(a, stream);
        case OMH_EPI_GELU_BF16: return launch_w64<K_GELU_P>(a, stream);
        default:                return launch_w64<K_RESID_P>(a, stream);
    }
#endif
}

int omh_launch_gemm_w64(const omh_gemm_args& a, hipStream_t stream) {
    switch (a.epilogue) {
        case OMH_EPI_F32:       return launch_w64<K_F32>(a, stream);
        case OMH_EPI_BF16:      return launch_w64<K_BF16>(a, stream);
        case OMH_EPI_GELU_BF16: return a.aux ? launch_w64<K_GELUAUX>(a, stream) : launch_w64<K_GELU>(a, stream);
        case OMH_EPI_GELU_BWD_BF16: return launch_w64<K_GELUBWD>(a, stream);
        default:                return launch_w64<K_RESID>(a, stream);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split K (ABI v9).  A product with few rows and a long contraction — the FFN-down projection and the FFN-up input
// gradient at one or two [16,1,60,104] clips: M = 1 560 / 3 120, N = 1 536, K = 8 960 — has 56 / 104 tiles of 256 x 192
// for 256 CUs, each with 140 k steps.  Its contraction is cut into S = 4 / 2 equal slices: slice s of a tile is one
// workgroup of the fp32 256 x 192 stream (same instruction stream, k window moved by the tile's byte offsets) writing
// its partial to slice s of the workspace ([S][tiles_m * 256][N] fp32: the rows a ragged last m-tile writes past M land
// in the slice's own padding), and splitk_combine_kernel adds the slices in the order 0..S-1, the bias, and applies the
// epilogue.  No atomics, one fixed order: bit-repeatable; S is a function of (M, N, K) only, so the inference forward and
// the training forward (c_in / aux) of one shape add the same partials.
namespace {

template <bool RESID>
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ ws, const int S, const int64_t slice,
                                                             const omh_gemm_args p) {
    const int nq = p.N >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.M * nq) return;
    const int m = (int)(i / nq), n = (int)(i - (int64_t)m * nq) * 4;
    const float* w = ws + (int64_t)m * p.N + n;
    float4 acc = *(const float4*)w;
    for (int s = 1; s < S; ++s) {
        const float4 v = *(const float4*)(w + (int64_t)s * slice);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (p.bias) {
        const float4 b = *(const float4*)(p.bias + n);
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    float* c = (float*)p.C + (int64_t)m * p.ldc + n;
    if (!RESID) {
        *(float4*)c = acc;
        return;
    }
    float4 g = make_float4(p.gate_const, p.gate_const, p.gate_const, p.gate_const);
    if (p.gate0) { const float4 v = *(const float4*)(p.gate0 + n); g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w; }
    if (p.gate1) {
        const float4 v = *(const float4*)(p.gate1 + (int64_t)(m / p.gate_rows) * p.gate1_stride + n);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
    }
    const float4 o = *(const float4*)((p.c_in ? p.c_in : (const float*)p.C) + (int64_t)m * p.ldc + n);
    *(float4*)c = make_float4(o.x + acc.x * g.x, o.y + acc.y * g.y, o.z + acc.z * g.z, o.w + acc.w * g.w);
    if (p.aux) {                                                          // y = bf16(acc + bias): the branch output
        uint2 y;
        y.x = pack_bf2(acc.x, acc.y);
        y.y = pack_bf2(acc.z, acc.w);
        *(uint2*)((unsigned short*)p.aux + (int64_t)m * p.ldaux + n) = y;
    }
}

omh_gemm_args splitk_partial(const omh_gemm_args& a, int S, void* ws) {
    omh_gemm_args q = a;
    q.C = ws; q.ldc = a.N; q.K = a.K / S;
    q.epilogue = OMH_EPI_F32; q.bias = nullptr; q.bias_mode = OMH_BIAS_NONE;
    q.gate0 = q.gate1 = nullptr; q.c_in = nullptr; q.aux = nullptr; q.ldaux = 0;
    q.workspace = nullptr; q.workspace_bytes = 0;
    return q;
}

}  // namespace

// Number of slices omh_gemm_bf16 cuts this product's contraction into when it is handed a workspace (1: none).
// OMH_GEMM_SPLITK = 0 switches the path off (A/B timing, tests).
int omh_gemm_splitk_slices(const omh_gemm_args& a) {
    const char* e = omh_opt(OMH_OPT_GEMM_SPLITK);
    if (e && e[0] == '0') return 1;
    if (a.batch != 1 || a.b_kmajor || !(a.epilogue == OMH_EPI_RESID || a.epilogue == OMH_EPI_F32)) return 1;
    if (a.bias && a.bias_mode != OMH_BIAS_N) return 1;
    if (a.K < 4096 || a.M < TM || (a.N & 7) || (a.ldc & 3) || ((uintptr_t)a.C & 15)) return 1;
    if (a.bias && ((uintptr_t)a.bias & 15)) return 1;
    if (a.epilogue == OMH_EPI_RESID) {
        if ((a.gate0 && ((uintptr_t)a.gate0 & 15)) || (a.gate1 && (((uintptr_t)a.gate1 & 15) || (a.gate1_stride & 3) || a.gate_rows <= 0)))
            return 1;
        if ((a.c_in && ((uintptr_t)a.c_in & 15)) || (a.aux && ((a.ldaux & 3) || ((uintptr_t)a.aux & 7)))) return 1;
    } else if (a.aux || a.c_in) {
        return 1;
    }
    const int64_t t192 = (int64_t)((a.M + TM - 1) / TM) * ((a.N + TN192 - 1) / TN192);
    if (t192 > 128) return 1;                                             // half of the chip or more: no slices
    int S = 0;
    for (int s = 4; s >= 2 && !S; --s)
        if (t192 * s <= 256 && a.K % (BK * s) == 0 && a.K / s >= 1024) S = s;
    if (!S) return 1;
    const omh_gemm_args q = splitk_partial(a, S, (void*)a.C);             // (any 16-byte aligned address: shape check only)
    return omh_gemm_w64_n192_takes(q) ? S : 1;
}

int64_t omh_gemm_splitk_workspace(const omh_gemm_args& a) {
    const int S = omh_gemm_splitk_slices(a);
    return S > 1 ? (int64_t)S * ((a.M + TM - 1) / TM * TM) * a.N * 4 : 0;
}

int omh_launch_gemm_splitk(const omh_gemm_args& a, int S, hipStream_t stream) {
    const int64_t slice = (int64_t)((a.M + TM - 1) / TM * TM) * a.N;      // floats per slice
    const omh_gemm_args q = splitk_partial(a, S, a.workspace);
    launch_w64<K_F32_192>(q, stream, W64Split{S, (uint32_t)(q.K * 2), (uint32_t)(slice * 4)});
    const int64_t n4 = (int64_t)a.M * (a.N >> 2);
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (a.epilogue == OMH_EPI_RESID)
        hipLaunchKernelGGL(splitk_combine_kernel<true>, grid, dim3(256), 0, stream, (const float*)a.workspace, S, slice, a);
    else
        hipLaunchKernelGGL(splitk_combine_kernel<false>, grid, dim3(256), 0, stream, (const float*)a.workspace, S, slice, a);
    return 0;
}
