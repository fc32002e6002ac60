// bf16 MFMA GEMM with both operands K-MAJOR:  C[m][n] (+)= sum_k A[k][m] * B[k][n]   (fp32 out), for gfx950.
//
// This is the weight-gradient product of the training step — dW[n_out][k_in] = sum_r dy[r][n_out] x[r][k_in]
// (autograd of every nn.Linear of model.py under distilled_trainer.py:289-301) — on dy and x exactly as the
// backward produces them, row-major [R, *].  The NT kernel (gemm_bf16.hip) needs the contraction index contiguous
// in both operands, so each such product used to cost two transposes through omh_transpose_bf16 first (~20 per
// block, 10 % of the training step's GPU time).
//
// Operand tiles [64 k][BM m] bf16 go HBM -> LDS by LDS-DMA as in the NT kernel (1 KiB per wave instruction,
// lane-linear image, source-side swizzle).  The MFMA fragments — lane = row m, 8 consecutive k — are gathered by
// ds_read_b64_tr_b16: within a group of 16 lanes, result lane i, element e is the (i & 3)-th 16-bit element at the
// address lane 4e + (i >> 2) supplies; with source lane 4e + q pointing at tile[k0 + e][m0 + 4q .. +3], lane i gets
// tile[k0 .. k0+3][m0 + i].  Two such reads (k0, k0+4) are one v_mfma_f32_32x32x16_bf16 operand; per 16-lane
// group g: m0 = 16 (g & 1), k0 = 8 (g >> 1).  The four k rows of a read sit a whole row pitch (256 / 512 B) apart,
// i.e. on the same banks: the 16-byte slot index is XORed with (row & 3) << 2, on the DMA source address and on the
// read.  The result layout puts 32 consecutive n of one row m in the 32 lanes of a half-wave, so the fp32 output is
// written straight from the accumulators in 128-byte runs.
#include "omh_common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_bf4_ptr;
typedef __attribute__((address_space(3))) void* lds_vptr;

struct TnGeom { int tiles_m, tiles_n, splits, ksteps_per_split; };

// the body of one output tile (tm, tn) over k-steps [kt0, kt_end); out_mode: 0 store, 1 read-modify-write, 2 atomicAdd
template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void tn_tile(const omh_gemm_tn_args& p, const int tm, const int tn, const int kt0,
                                        const int kt_end, const int out_mode) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int SLA = BM / 8, SLB = BN / 8;                        // 16-byte slots per tile row
    constexpr int ROWA = BM * 2, ROWB = BN * 2;                      // row pitch, bytes
    constexpr int A_BYTES = BK * ROWA, B_BYTES = BK * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int CHA = BK * SLA / THREADS, CHB = BK * SLB / THREADS; // chunks per thread and stage
    static_assert(CHA == 4 && CHB == 4, "4 chunks per operand per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = tm * BM, n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.A, 0, (int)((((int64_t)p.K - 1) * p.lda + p.M) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, (int)((((int64_t)p.K - 1) * p.ldb + p.N) * 2), 0x00020000);

    // staging: chunk c = tid + THREADS j lands at LDS byte 16 c of the tile = (row c / SL, physical slot c % SL); it
    // is fetched from logical slot  physical ^ ((row & 3) << 2)  of that k row
    uint32_t voff_a[4], voff_b[4];
    int row_a[4], row_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + THREADS * j;
        row_a[j] = c / SLA;
        const int la = (c % SLA) ^ ((row_a[j] & 3) << 2);
        voff_a[j] = (m0 + la * 8 < p.M) ? (uint32_t)(((int64_t)row_a[j] * p.lda + m0 + la * 8) * 2) : 0x80000000u;
        row_b[j] = c / SLB;
        const int lb = (c % SLB) ^ ((row_b[j] & 3) << 2);
        voff_b[j] = (n0 + lb * 8 < p.N) ? (uint32_t)(((int64_t)row_b[j] * p.ldb + n0 + lb * 8) * 2) : 0x80000000u;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
    const uint32_t kstep_a = (uint32_t)(BK * p.lda * 2), kstep_b = (uint32_t)(BK * p.ldb * 2);

#define TN_DMA(KT_, BUF)                                                                              \
    {                                                                                                 \
        unsigned char* xa_ = smem + (BUF) * STAGE_BYTES;                                              \
        unsigned char* xb_ = xa_ + A_BYTES;                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                            \
            const uint32_t oa_ = ((KT_) * BK + row_a[j_] < p.K) ? voff_a[j_] + (uint32_t)(KT_) * kstep_a : 0x80000000u; \
            const uint32_t ob_ = ((KT_) * BK + row_b[j_] < p.K) ? voff_b[j_] + (uint32_t)(KT_) * kstep_b : 0x80000000u; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_vptr)(xa_ + wave_lds + j_ * THREADS * 16), 16, oa_, 0, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_vptr)(xb_ + wave_lds + j_ * THREADS * 16), 16, ob_, 0, 0, 0); \
        }                                                                                             \
    }

    // fragment gather addresses (see the header): group g' = lane>>4, i = lane&15, e = i>>2, q = i&3
    const int gq = lane >> 4, li = lane & 15, fe = li >> 2, fq = li & 3;
    uint32_t fa[MT][2], fb[NT][2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int row = 8 * (gq >> 1) + 4 * part + fe;               // + 16 kk per k group
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int l = ((wm * MT + i) * 32 + 16 * (gq & 1) + 4 * fq) >> 3;
            fa[i][part] = (uint32_t)(row * ROWA + ((l ^ (fe << 2)) << 4) + (fq & 1) * 8);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int l = ((wn * NT + i) * 32 + 16 * (gq & 1) + 4 * fq) >> 3;
            fb[i][part] = (uint32_t)(row * ROWB + ((l ^ (fe << 2)) << 4) + (fq & 1) * 8);
        }
    }
#define TN_FRAG(DST, BASE, ADDR, KK, ROWBYTES)                                                        \
    {                                                                                                 \
        const bf16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                \
            (lds_bf4_ptr)((__attribute__((address_space(3))) unsigned char*)(BASE) + ADDR[0] + (KK) * 16 * (ROWBYTES))); \
        const bf16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                \
            (lds_bf4_ptr)((__attribute__((address_space(3))) unsigned char*)(BASE) + ADDR[1] + (KK) * 16 * (ROWBYTES))); \
        DST = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                              \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split K (small outputs with a long contraction: 1536 x 1536 over 6240 rows is 144 tiles of 98 k-steps, one
    // latency-bound workgroup on half the CUs): blockIdx.y takes k-steps [kt0, nk) and adds its partial sums with
    // fp32 atomics (C zeroed by the launcher).  The summation order then varies from run to run: last-bit
    // differences in dW, like the bias / gain gradients of dit_backward.hip.
    const int nk = kt_end;
    if (kt0 >= nk) return;
    TN_DMA(kt0, 0)
    if (kt0 + 1 < nk) TN_DMA(kt0 + 1, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // main loop as in gemm_bf16.hip: the fragments of k group kk+1 are gathered while group kk is on the matrix pipe
    // (two register sets), one barrier per k-step, the DMA of stage kt+2 issued right behind it
#define TN_FRAGS(AF, BFR, STAGE, KK)                                                                  \
    {                                                                                                 \
        const unsigned char* xa_ = smem + (STAGE) * STAGE_BYTES;                                      \
        const unsigned char* xb_ = xa_ + A_BYTES;                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) TN_FRAG(AF[i_], xa_, fa[i_], KK, ROWA)      \
        _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_) TN_FRAG(BFR[i_], xb_, fb[i_], KK, ROWB)     \
    }
#define TN_MFMAS(AF, BFR)                                                                             \
    _Pragma("unroll") for (int im_ = 0; im_ < MT; ++im_)                                              \
        _Pragma("unroll") for (int in_ = 0; in_ < NT; ++in_)                                          \
            acc[im_][in_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[im_], BFR[in_], acc[im_][in_], 0, 0, 0);
    // one MFMA, then two of the (MT + NT) x 2 transposing reads, ...
#define TN_INTERLEAVE()                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < MT + NT; ++s_) {                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                            \
    }                                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT) > 0 ? MT * NT - (MT + NT) : 0, 0);

    bf16x8 af0[MT], bf0[NT], af1[MT], bf1[NT];
    TN_FRAGS(af0, bf0, 0, 0)
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        TN_FRAGS(af1, bf1, buf, 1)
        TN_MFMAS(af0, bf0)
        TN_INTERLEAVE()
        TN_FRAGS(af0, bf0, buf, 2)
        TN_MFMAS(af1, bf1)
        TN_INTERLEAVE()
        TN_FRAGS(af1, bf1, buf, 3)
        TN_MFMAS(af0, bf0)
        TN_INTERLEAVE()
        // every wave has read stage kt, and stage kt+1 (issued one k-step ago) has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk) TN_DMA(kt + 2, buf)
        if (kt + 1 < nk) TN_FRAGS(af0, bf0, buf ^ 1, 0)
        TN_MFMAS(af1, bf1)
    }

    // epilogue: lane (n = lane & 31, half h) holds C[m = 8 (r >> 2) + 4 h + (r & 3)][n] of each 32x32 tile
    float* Cb = p.C;
    const int h = lane >> 5, ln = lane & 31;
#pragma unroll
    for (int im = 0; im < MT; ++im)
#pragma unroll
        for (int in = 0; in < NT; ++in) {
            const int n = n0 + (wn * NT + in) * 32 + ln;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MT + im) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (m >= p.M) continue;
                float* dst = Cb + (int64_t)m * p.ldc + n;
                if (out_mode == 2) atomicAdd(dst, acc[im][in][r]);
                else if (out_mode == 1) *dst += acc[im][in][r];
                else *dst = acc[im][in][r];
            }
        }
#undef TN_DMA
#undef TN_FRAG
#undef TN_FRAGS
#undef TN_MFMAS
#undef TN_INTERLEAVE
}

template <bool ACCUM, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 1)
void gemm_bf16_tn_kernel(const omh_gemm_tn_args p, const TnGeom g) {
    const int wid = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    int tm, tn;
    tile_of(wid, g.tiles_m, g.tiles_n, tm, tn, 4);                   // 4 m-tiles x all n-tiles per XCD-resident group
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = blockIdx.y * g.ksteps_per_split;
    tn_tile<WM, WN, MT, NT>(p, tm, tn, kt0, min(nk_all, kt0 + g.ksteps_per_split), g.splits > 1 ? 2 : (ACCUM ? 1 : 0));
}

// Several independent products in ONE launch (the weight gradients of a block's backward): 128 x 128 tiles, every
// tile its whole K range — no split K, hence no atomics (bit-repeatable) — the tiles of all problems together fill the
// chip where each problem alone (144 tiles of 1536 x 1536 on 512 slots) did not.
template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void tn_grouped_body(const omh_gemm_tn_group& g) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int wid = xcd_remap(blockIdx.x, g.total_tiles);
    int e = 0;
#pragma unroll 1
    while (e + 1 < g.n && wid >= g.first_tile[e + 1]) ++e;
    const omh_gemm_tn_args& p = g.problem[e];
    const int local = wid - g.first_tile[e];
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    tile_of(local, tiles_m, tiles_n, tm, tn, 4);
    tn_tile<WM, WN, MT, NT>(p, tm, tn, 0, (p.K + BK - 1) / BK, p.accumulate ? 1 : 0);
}
__global__ __launch_bounds__(256, 2)
void gemm_bf16_tn_grouped_kernel(const omh_gemm_tn_group g) { tn_grouped_body<2, 2, 2, 2>(g); }
// the same on 256 x 256 tiles (half the staged bytes per flop) when the group's tiles still cover most of the chip in
// ONE round of one workgroup per CU: a block's {cross o, cross q, cross k|v, self o} is 180 such tiles, q|k|v 108
__global__ __launch_bounds__(512)
void gemm_bf16_tn_grouped_big_kernel(const omh_gemm_tn_group g) { tn_grouped_body<2, 4, 4, 2>(g); }

template <bool ACCUM, int WM, int WN, int MT, int NT>
int launch_tn(const omh_gemm_tn_args& a, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int LDS = 2 * BK * (BM + BN) * 2;
    auto kern = gemm_bf16_tn_kernel<ACCUM, WM, WN, MT, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    TnGeom g;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = (a.N + BN - 1) / BN;
    const int nk = (a.K + BK - 1) / BK;
    const int slots = (BM == 256) ? 256 : 512;                       // resident workgroups on the chip
    const int tiles = g.tiles_m * g.tiles_n;
    const char* spe = omh_opt(OMH_OPT_GEMM_TN_SPLIT);                    // forced split count (tests / timing)
    // split only when the tiles leave at least half the chip idle (more splits measured slower: 1536^2 x 6240 60 us
    // with 3, 77 with 4; 3072 x 1536 97 us unsplit, 125 with 2 — the atomics and the zero fill cost more than they win)
    int splits = spe ? atoi(spe) : (tiles * 2 <= slots ? slots / tiles : 1);
    if (omh_deterministic()) splits = 1;                              // no atomics: every tile its whole K range
    splits = splits < 1 ? 1 : (splits > 8 ? 8 : splits);
    if (nk < 16 * splits) splits = nk / 16 < 1 ? 1 : nk / 16;        // at least 16 k-steps per split (measured: 9 lose)
    g.ksteps_per_split = (nk + splits - 1) / splits;
    g.splits = (nk + g.ksteps_per_split - 1) / g.ksteps_per_split;
    omh_clear_status();
    if (g.splits > 1 && !ACCUM) {                                    // partial sums are added: start from zero
        omh_zero_f32(a.C, a.M, a.N, a.ldc, s);          // a kernel, not a memset node: omh_common.h
    }
    hipLaunchKernelGGL(kern, dim3(tiles, g.splits), dim3(64 * WM * WN), LDS, s, a, g);
    return omh_launch_status();
}

}  // namespace

// gemm_tn_w64.hip: the 256 x 384 stream kernel (one workgroup per CU, persistent)
bool omh_gemm_tn_w64_takes(const omh_gemm_tn_args& a);
int64_t omh_gemm_tn_w64_tiles(const omh_gemm_tn_args& a);
int omh_launch_gemm_tn_w64(omh_gemm_tn_group g, hipStream_t stream);
// OMH_GEMM_TN_W64=0: never the stream kernel (A/B timing and the bit-equality tests); =1: whenever it takes the shapes
static int tn_w64_mode() {
    const char* e = omh_opt(OMH_OPT_GEMM_TN_W64);
    return e ? atoi(e) : -1;
}

extern "C" int omh_gemm_bf16_tn(const omh_gemm_tn_args* args, omh_stream_t stream) {
    if (!args || !args->A || !args->B || !args->C) return OMH_E_BADARG;
    const omh_gemm_tn_args& a = *args;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return OMH_E_BADARG;
    if ((a.M & 7) || (a.N & 7) || (a.lda & 7) || (a.ldb & 7) || a.lda < a.M || a.ldb < a.N || a.ldc < a.N) return OMH_E_ALIGN;
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 3)) return OMH_E_ALIGN;
    if (((int64_t)a.K + 64) * a.lda * 2 >= 0x7fffffffLL || ((int64_t)a.K + 64) * a.ldb * 2 >= 0x7fffffffLL)
        return OMH_E_SHAPE;                                          // 32-bit buffer offsets
    hipStream_t s = (hipStream_t)stream;
    {   // one product that fills at least 3/8 of the chip with 256 x 384 tiles (the FFN weights: 140 / 144 tiles)
        const int mode = tn_w64_mode();
        if (mode != 0 && omh_gemm_tn_w64_takes(a) && (mode == 1 || omh_gemm_tn_w64_tiles(a) >= 96)) {
            omh_gemm_tn_group g1;
            g1.n = 1;
            g1.problem[0] = a;
            return omh_launch_gemm_tn_w64(g1, s);
        }
    }
    const int64_t big_tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const char* force = omh_opt(OMH_OPT_GEMM_TN_TILE);                  // "big" / "small": test override
    const bool big = force ? force[0] == 'b' : big_tiles >= 128;
    if (big) return a.accumulate ? launch_tn<true, 2, 4, 4, 2>(a, s) : launch_tn<false, 2, 4, 4, 2>(a, s);
    return a.accumulate ? launch_tn<true, 2, 2, 2, 2>(a, s) : launch_tn<false, 2, 2, 2, 2>(a, s);
}

extern "C" int omh_gemm_bf16_tn_grouped(const omh_gemm_tn_group* group, omh_stream_t stream) {
    if (!group || group->n <= 0 || group->n > OMH_TN_GROUP_MAX) return OMH_E_BADARG;
    omh_gemm_tn_group g = *group;
    int64_t total = 0;
    for (int i = 0; i < g.n; ++i) {
        const omh_gemm_tn_args& a = g.problem[i];
        if (!a.A || !a.B || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return OMH_E_BADARG;
        if ((a.M & 7) || (a.N & 7) || (a.lda & 7) || (a.ldb & 7) || a.lda < a.M || a.ldb < a.N || a.ldc < a.N) return OMH_E_ALIGN;
        if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 3)) return OMH_E_ALIGN;
        if (((int64_t)a.K + 64) * a.lda * 2 >= 0x7fffffffLL || ((int64_t)a.K + 64) * a.ldb * 2 >= 0x7fffffffLL) return OMH_E_SHAPE;
        g.first_tile[i] = (int32_t)total;
        total += (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    }
    if (total > 0x7fffffffLL) return OMH_E_SHAPE;
    {   // all of them on the 256 x 384 stream when every product qualifies and together they are worth a launch
        const int mode = tn_w64_mode();
        bool all = mode != 0;
        int64_t t384 = 0;
        for (int i = 0; all && i < g.n; ++i) {
            all = omh_gemm_tn_w64_takes(g.problem[i]);
            t384 += all ? omh_gemm_tn_w64_tiles(g.problem[i]) : 0;
        }
        if (all && (mode == 1 || t384 >= 48)) return omh_launch_gemm_tn_w64(g, (hipStream_t)stream);
    }
    int64_t total_big = 0;
    for (int i = 0; i < g.n; ++i) total_big += (int64_t)((g.problem[i].M + 255) / 256) * ((g.problem[i].N + 255) / 256);
    const char* te = omh_opt(OMH_OPT_GEMM_TN_GROUP_TILE);               // "big" / "small": test / timing override
    // measured (one box, interleaved, whole training step): 85.2 ms with 128 x 128 tiles, 86.5 with 256 x 256 at 4 clips
    // (43.5 / 45.2 at 1 clip) — one workgroup per CU shares the chip worse with the main stream's kernels — so: opt-in only
    const bool big = te ? te[0] == 'b' : false;
    (void)total_big;
    constexpr int LDS = 2 * BK * (128 + 128) * 2, LDS_BIG = 2 * BK * (256 + 256) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BIG);
        attr_set = true;
    }
    omh_clear_status();
    if (big) {
        total = 0;
        for (int i = 0; i < g.n; ++i) {
            g.first_tile[i] = (int32_t)total;
            total += (int64_t)((g.problem[i].M + 255) / 256) * ((g.problem[i].N + 255) / 256);
        }
        g.total_tiles = (int32_t)total;
        hipLaunchKernelGGL(gemm_bf16_tn_grouped_big_kernel, dim3((unsigned)total), dim3(512), LDS_BIG, (hipStream_t)stream, g);
    } else {
        g.total_tiles = (int32_t)total;
        hipLaunchKernelGGL(gemm_bf16_tn_grouped_kernel, dim3((unsigned)total), dim3(256), LDS, (hipStream_t)stream, g);
    }
    return omh_launch_status();
}
