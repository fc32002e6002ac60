// Weight-gradient products C[M, N] (+)= A[K, M]^T B[K, N] (bf16 operands as stored, contraction on the rows, fp32 out) on a
// 256(m) x 384(n) x 64(k) workgroup tile: 4 waves, one per SIMD, 384 accumulators per wave, the k loop and the epilogue a
// generated instruction stream (gen_gemm_tn_w64.py -> gemm_tn_w64_asm.inc; read its header).  Persistent workgroups walk
// the tiles of up to OMH_TN_GROUP_MAX products (a block's q|k|v, o, cross q, cross k|v, cross o are 192 such tiles: one
// round of the chip).  Same contract and the same accumulation order as gemm_tn.hip's kernels without split K, which stay
// for everything this one does not take (omh_gemm_tn_w64_takes); this file only computes descriptors and per-lane offsets.
#include "omh_common.h"
#include "gemm_tn_w64_asm.inc"
#include <stdlib.h>

namespace {

constexpr int TM = 256, TN = 384, BK = 64;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffLL ? 0xffffffffu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)n, 0x00020000);
}

// two wave-uniform 32-bit scalars in one SGPR pair (inline asm takes at most 30 operands)
__device__ __forceinline__ uint64_t pack2(uint32_t lo, uint32_t hi) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32);
}

__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_bf16_tn_w64_kernel(const omh_gemm_tn_group g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 81920];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    // transposing fragment reads (gemm_tn.hip): 16-lane group gq, lane fr = 4 fe + fq of it reads k row 8 (gq >> 1) + fe (+ 4
    // for the second half, + 16 per k group) at the 8 bytes of columns 16 (gq & 1) + 4 fq ... of the 32-column tile
    const int gq = lane >> 4, fr = lane & 15, fe = fr >> 2, fq = fr & 3;
    const uint32_t low = (uint32_t)((2 * (gq & 1) + (fq >> 1)) * 16 + (fq & 1) * 8);
    const uint32_t mab = lds0 + (uint32_t)((8 * (gq >> 1) + fe) * 512) + low;
    const uint32_t nab = lds0 + 32768u + (uint32_t)((8 * (gq >> 1) + fe) * 768) + low;
    const int h = lane >> 5, sl = lane & 31;

    const uint64_t p0 = pack2(lds0 + (uint32_t)w * 8192u, lds0 + 32768u + (uint32_t)w * 12288u);
    const uint64_t p6 = pack2((uint32_t)(wm * 4), (uint32_t)(wn * 6));

#pragma nounroll
    for (int idx = blockIdx.x; idx < g.total_tiles; idx += (int)gridDim.x) {
        const int wid = xcd_remap(idx, g.total_tiles);
        int e = 0;
#pragma unroll 1
        while (e + 1 < g.n && wid >= g.first_tile[e + 1]) ++e;
        const omh_gemm_tn_args& p = g.problem[e];
        const int tiles_m = p.M / TM, tiles_n = (p.N + TN - 1) / TN;
        int tm, tn;
        tile_of(wid - g.first_tile[e], tiles_m, tiles_n, tm, tn, 8);
        const int m0 = tm * TM, n0 = tn * TN;
        const int lda = p.lda, ldb = p.ldb, ldc = p.ldc;

        // LDS-DMA source offsets.  A piece = 2 rows x 32 slots: lane -> (row h, physical slot sl), logical slot sl ^ ((row &
        // 3) << 2) with row & 3 = h (even pieces) / 2 + h (odd ones).  B piece = 64 of the 16 x 48 slots of the wave's rows:
        // three patterns (piece % 3), 4 rows per 3 pieces.
        const uint32_t voa0 = (uint32_t)((h * lda + ((sl ^ (h << 2)) * 8)) * 2);
        const uint32_t voa1 = (uint32_t)((h * lda + ((sl ^ ((2 + h) << 2)) * 8)) * 2);
        uint32_t vob[3];
#pragma unroll
        for (int pat = 0; pat < 3; ++pat) {
            const int c = 64 * pat + lane, rl = c / 48, lg = (c % 48) ^ ((rl & 3) << 2);
            vob[pat] = (n0 + 8 * lg < p.N) ? (uint32_t)((rl * ldb + 8 * lg) * 2) : 0x80000000u;
        }
        const uint32_t voc = (uint32_t)((((int64_t)(wm * 128 + 4 * h)) * ldc + wn * 192 + (lane & 31)) * 4);
        const int nrem = p.N - n0 - wn * 192;

        const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.A, (((int64_t)p.K - 1) * lda + p.M) * 2);
        const __amdgpu_buffer_rsrc_t rb = rsrc_of(p.B, (((int64_t)p.K - 1) * ldb + p.N) * 2);
        const __amdgpu_buffer_rsrc_t rc = rsrc_of(p.C, (((int64_t)p.M - 1) * ldc + p.N) * 4);
        const uint64_t p1 = pack2((uint32_t)(((int64_t)(16 * w) * lda + m0) * 2), (uint32_t)(((int64_t)(16 * w) * ldb + n0) * 2));
        const uint64_t p2 = pack2((uint32_t)(2 * lda * 2), (uint32_t)(4 * ldb * 2));
        const uint64_t p3 = pack2((uint32_t)((p.K + BK - 1) / BK), (uint32_t)(((int64_t)m0 * ldc + n0) * 4));
        const uint64_t p4 = pack2((uint32_t)(ldc * 4), (uint32_t)(nrem > 0 ? nrem : 0));
        const uint64_t p5 = pack2((uint32_t)(BK * lda * 2), (uint32_t)(BK * ldb * 2));
#define OMH_GTW64_RUN(ASM)                                                                                             \
    asm volatile(ASM                                                                                                   \
                 :                                                                                                     \
                 : [mab] "v"(mab), [nab] "v"(nab), [voa0] "v"(voa0), [voa1] "v"(voa1), [vob0] "v"(vob[0]), \
                   [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [voc] "v"(voc), [ra] "s"(ra), [rb] "s"(rb), \
                   [rc] "s"(rc), [p0] "{s[60:61]}"(p0), [p1] "{s[62:63]}"(p1), [p2] "{s[64:65]}"(p2),                   \
                   [p3] "{s[66:67]}"(p3), [p4] "{s[68:69]}"(p4), [p5] "{s[70:71]}"(p5), [p6] "{s[72:73]}"(p6)           \
                 : OMH_GEMM_TN_W64_CLOBBERS)
        if (p.accumulate) OMH_GTW64_RUN(OMH_GEMM_TN_W64_ASM_ACC);
        else OMH_GTW64_RUN(OMH_GEMM_TN_W64_ASM_ST);
#undef OMH_GTW64_RUN
    }
}

}  // namespace

// The products the stream kernel takes: M a multiple of the 256-row tile (the epilogue masks columns, not rows), at least
// three k tiles, 16-byte-aligned operand rows, 32-bit byte offsets everywhere.
bool omh_gemm_tn_w64_takes(const omh_gemm_tn_args& a) {
    return a.A && a.B && a.C && a.M > 0 && (a.M % TM) == 0 && a.N > 0 && (a.N & 7) == 0 && a.K >= 3 * BK &&
           (a.lda & 7) == 0 && (a.ldb & 7) == 0 && a.lda >= a.M && a.ldb >= a.N && a.ldc >= a.N &&
           (((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.B) & 15) == 0 && (((uintptr_t)a.C) & 3) == 0 &&
           ((int64_t)a.K + BK) * a.lda * 2 < 0x7fffffffLL && ((int64_t)a.K + BK) * a.ldb * 2 < 0x7fffffffLL &&
           ((int64_t)a.M + TM) * a.ldc * 4 < 0x7fffffffLL;
}

int64_t omh_gemm_tn_w64_tiles(const omh_gemm_tn_args& a) { return (int64_t)(a.M / TM) * ((a.N + TN - 1) / TN); }

// `g`: a validated group whose products all pass omh_gemm_tn_w64_takes; first_tile / total_tiles are filled here.
int omh_launch_gemm_tn_w64(omh_gemm_tn_group g, hipStream_t stream) {
    int64_t total = 0;
    for (int i = 0; i < g.n; ++i) {
        g.first_tile[i] = (int32_t)total;
        total += omh_gemm_tn_w64_tiles(g.problem[i]);
    }
    if (total <= 0 || total > 0x7fffffffLL) return OMH_E_SHAPE;
    g.total_tiles = (int32_t)total;
    int ncu = omh_cu_count();
    ncu -= ncu % 8;                                                       // whole XCD rounds keep xcd_remap's chunks aligned
    if (ncu < 8) ncu = 256;
    omh_clear_status();
    hipLaunchKernelGGL(gemm_bf16_tn_w64_kernel, dim3((unsigned)(total < ncu ? total : ncu)), dim3(256), 0, stream, g);
    return omh_launch_status();
}
