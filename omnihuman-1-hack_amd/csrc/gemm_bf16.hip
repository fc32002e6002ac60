// bf16 MFMA GEMM  C[m][n] = epi( sum_k A[m][k] * B[n][k] )  for gfx950.
//
// Replaces aten addmm under every nn.Linear of the reference DiT
// (seaweed_apt/wan/modules/model.py:125-128,160,176-178,185,272-274,344,465-467)
// and the patch-embedding Conv3d-as-GEMM (model.py:463,515).
//
// Tiling (two instantiations, picked per problem size):
//   big   256(m) x 256(n) x 64(k), 512 threads = 8 waves as 2(m) x 4(n), wave tile 128x64
//         (4x2 MFMA accumulators): 128 flop per operand byte, for the S = 32 760 GEMMs;
//   small 128(m) x 128(n) x 64(k), 256 threads = 4 waves as 2x2, wave tile 64x64, two
//         workgroups per CU;
//   tiny  64 x 64 x 64, 4 waves, wave tile 32x32: keeps 256 CUs busy when M*N is small
//         (context projections, the S = 1560 training clips).
// Each MFMA is v_mfma_f32_32x32x16_bf16.
// The weight rows (n) go in the MFMA A slot and the activation rows (m) in
// the B slot, so each lane ends up with runs of 4 consecutive n for one m; the
// epilogue transposes through LDS to row-major 16-byte accesses.  Both operand tiles are staged
// HBM -> LDS directly with the LDS-DMA form of the buffer load
// (buffer_load_dwordx4 ... lds: no VGPR round trip and, more importantly, no
// ds_write_b128 pass — at 128x128 the register-staged version spent more LDS
// cycles on writes+reads than the MFMA pipe spent computing).  The LDS image is
// lane-linear per wave instruction, so the XOR swizzle ((row>>1)&7 on the
// 16-byte slot, conflict-free ds_read_b128) is applied to the per-lane SOURCE
// address and again on the read.  Rows past M/N and the K tail are out of the
// buffer descriptor's range and arrive as zeros.  Double buffered, one barrier
// per k-step; a stage's DMA is issued as soon as the barrier frees its buffer
// (one full k-step ahead), and the MFMA fragments are read one 16-wide k group ahead.
#include "omh_common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_bf4_ptr;
typedef __attribute__((address_space(3))) unsigned char* lds_u8_ptr;

struct GemmGeom { int tiles_m, tiles_n, group_m; };

__device__ __forceinline__ uint32_t lds_slot_addr(int row, int slot) {
    return (uint32_t)(row * (BK * 2) + ((slot ^ ((row >> 1) & 7)) << 4));
}

// WM x WN waves, each owning MT x NT 32x32 MFMA tiles; STAGES-deep LDS ring
// BKM: B is K-MAJOR ([K, N] row-major: dx = dy W with W as stored, the dgrad of every nn.Linear).  Its tile is then
// staged as [64 k][BN n] and the fragments (lane = row n, 8 consecutive k) are gathered with ds_read_b64_tr_b16,
// exactly as in gemm_tn.hip (which documents the addressing); everything else is shared.  Measured (tools/
// gemm_nn_probe.py): 15-30 % slower than the row-major-B kernel on the same product (6240 x 1536 x 8960: 233 vs 182 us)
// and all of it is the transposing read — with plain ds_read_b64 at the same addresses the two kernels tie.  The
// 256x256 tile already spends 384 of every 512 MFMA clocks on LDS reads; ds_read_b64_tr_b16 moves half the bytes
// of a ds_read_b128 in the same LDS time, which pushes the LDS pipe past the matrix pipe.  A weight transpose
// costs 10-18 us once per step, so the training step keeps transposed weight copies (OMH_DGRAD=nn selects this).
template <int EPI, int WM, int WN, int MT, int NT, int STAGES, bool BKM = false>
__global__ __launch_bounds__(64 * WM * WN, 2)
void gemm_bf16_nt_kernel(const omh_gemm_args p, const GemmGeom g) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int SLB = BN / 8, ROWB = BN * 2;                     // k-major B tile: 16-byte slots per k row, pitch
    static_assert(!BKM || (BK * SLB / THREADS == 4 && BM * 8 / THREADS == 4), "k-major B: 4 chunks per thread");
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int CA = BM * 8 / THREADS, CB = BN * 8 / THREADS;     // 16-byte chunks per thread and stage
    static_assert(CA * THREADS == BM * 8 && CB * THREADS == BN * 8 && CA >= 2 && CA <= 4 && CB >= 2 && CB <= 4,
                  "staging code assumes 2..4 whole chunks per operand per thread");
    constexpr int NCH = CA > CB ? CA : CB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    int tm, tn;
    tile_of(wid, g.tiles_m, g.tiles_n, tm, tn, g.group_m);
    const int m0 = tm * BM, n0 = tn * BN;

    const int64_t zb = blockIdx.z;
    const __bf16* __restrict__ A = (const __bf16*)p.A + zb * p.strideA;
    const __bf16* __restrict__ B = (const __bf16*)p.B + zb * p.strideB;

    // staging: 4 x 16-byte chunks per operand per thread.  Chunk c = tid + THREADS*j lands at LDS
    // byte c*16 of the tile (row c>>3, physical slot c&7); it is fetched from logical slot
    // (c&7) ^ ((row>>1)&7) of that row.  Offsets past the extent (rows >= M/N) read as zero.
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)A, 0, (int)((((int64_t)p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)B, 0, BKM ? (int)((((int64_t)p.K - 1) * p.ldb + p.N) * 2) : (int)((((int64_t)p.N - 1) * p.ldb + p.K) * 2),
        0x00020000);
    uint32_t voff_a[4], voff_b[4];          // fixed extents: a template-dependent extent loses hipcc's host stub
    int lslot[4], krow_b[4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = tid + THREADS * j;
        const int row = c >> 3;
        lslot[j] = (c & 7) ^ ((row >> 1) & 7);
        voff_a[j] = (m0 + row < p.M) ? (uint32_t)((((int64_t)(m0 + row)) * p.lda + lslot[j] * 8) * 2) : 0x80000000u;
        if (BKM) {      // chunk c = (k row c / SLB, physical slot c % SLB), fetched from logical slot ^ ((row & 3) << 2)
            krow_b[j] = c / SLB;
            const int lb = (c % SLB) ^ ((krow_b[j] & 3) << 2);
            voff_b[j] = (n0 + lb * 8 < p.N) ? (uint32_t)(((int64_t)krow_b[j] * p.ldb + n0 + lb * 8) * 2) : 0x80000000u;
        } else {
            krow_b[j] = 0;
            voff_b[j] = (n0 + row < p.N) ? (uint32_t)((((int64_t)(n0 + row)) * p.ldb + lslot[j] * 8) * 2) : 0x80000000u;
        }
    }
    const uint32_t kstep_b = (uint32_t)(BK * p.ldb * 2);
    // k-major B fragment gather addresses (gemm_tn.hip): 16-lane group gq, i = lane & 15, e = i >> 2, q = i & 3
    uint32_t fb[NT][2];
    {
        const int gq = lane >> 4, l15 = lane & 15, fe = l15 >> 2, fq = l15 & 3;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int row = 8 * (gq >> 1) + 4 * part + fe;            // + 16 KK per k group
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int l = ((wn * NT + i) * 32 + 16 * (gq & 1) + 4 * fq) >> 3;
                fb[i][part] = (uint32_t)(row * ROWB + ((l ^ (fe << 2)) << 4) + (fq & 1) * 8);
            }
        }
    }
    // wave-uniform LDS byte offset of this wave's 1 KiB piece of chunk group j
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define GEMM_DMA1(RSRC, VOFF, J, DST)                                                          \
    {                                                                                          \
        const uint32_t ko_ = (k0_ + lslot[J] * 8 < p.K) ? (uint32_t)(k0_ * 2) : 0x80000000u;   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, (lds_ptr_t)((DST) + wave_lds + (J) * THREADS * 16), 16, \
                                                 VOFF[J] + ko_, 0, 0, 0);                      \
    }
#define GEMM_DMA(KT_, BUF)                                                                     \
    {                                                                                          \
        const int k0_ = (KT_) * BK;                                                            \
        unsigned char* xa_ = smem + (BUF) * STAGE_BYTES;                                       \
        unsigned char* xb_ = xa_ + A_BYTES;                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < NCH; ++j_) {                                   \
            if (j_ < CA) GEMM_DMA1(rsrc_a, voff_a, j_, xa_)                                    \
            if (j_ >= CB) continue;                                                            \
            if (BKM) {                                                                         \
                const uint32_t ob_ = (k0_ + krow_b[j_] < p.K) ? voff_b[j_] + (uint32_t)(KT_) * kstep_b : 0x80000000u; \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(xb_ + wave_lds + j_ * THREADS * 16), 16, \
                                                         ob_, 0, 0, 0);                        \
            } else {                                                                           \
                GEMM_DMA1(rsrc_b, voff_b, j_, xb_)                                             \
            }                                                                                  \
        }                                                                                      \
    }

    // Fragment loads are software-pipelined one MFMA group ahead (register double buffer), across the
    // stage boundary too.  Per k-step: three groups read the next group's fragments while they compute;
    // then one wait+barrier proves (a) stage kt+1 has landed for every wave and (b) every wave has read
    // the last fragments of stage kt, so the DMA for stage kt+2 may overwrite it — it gets a whole
    // k-step to land.
#define GEMM_FRAGS(WF, XF, STAGE, KK)                                                          \
    {                                                                                          \
        const unsigned char* xa_ = smem + (STAGE) * STAGE_BYTES;                               \
        const unsigned char* xb_ = xa_ + A_BYTES;                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_) {                                    \
            if (BKM) {                                                                  \
                const bf16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                 \
                    (lds_bf4_ptr)((lds_u8_ptr)xb_ + fb[i_][0] + (KK) * 16 * ROWB));            \
                const bf16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                 \
                    (lds_bf4_ptr)((lds_u8_ptr)xb_ + fb[i_][1] + (KK) * 16 * ROWB));            \
                WF[i_] = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);            \
            } else {                                                                           \
                WF[i_] = *(const bf16x8*)(xb_ + lds_slot_addr((wn * NT + i_) * 32 + li, 2 * (KK) + lh)); \
            }                                                                                  \
        }                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                      \
            XF[i_] = *(const bf16x8*)(xa_ + lds_slot_addr((wm * MT + i_) * 32 + li, 2 * (KK) + lh)); \
    }
#define GEMM_MFMAS(WF, XF)                                                                     \
    _Pragma("unroll") for (int im_ = 0; im_ < MT; ++im_)                                       \
        _Pragma("unroll") for (int in_ = 0; in_ < NT; ++in_)                                   \
            acc[im_][in_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[in_], XF[im_], acc[im_][in_], 0, 0, 0);
    // one MFMA, then one LDS read, ... so the reads hide under the matrix pipe
#define GEMM_INTERLEAVE()                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < NT; ++s_) {       /* B fragments: two transposing reads each if BKM */ \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, BKM ? 2 : 1, 0);                           \
    }                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < MT; ++s_) {                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                     \
    }                                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT) > 0 ? MT * NT - (MT + NT) : 0, 0);

    // STAGES-deep ring: stages kt+1 .. kt+STAGES-1 are in flight while stage kt is consumed; rings deeper than 2
    // wait with a counted vmcnt (loads retire in order: allowing the STAGES-2 youngest stages to stay
    // outstanding proves stage kt+1 has landed).  Every shipped configuration uses 2: a deeper ring measured no
    // faster on any of them (the 64x64 tile is L2->LDS-bandwidth bound at ~13 TB/s aggregate, not latency bound,
    // and 64 KiB of LDS per workgroup halves the resident workgroups — DESIGN.md section 4.4).
    constexpr int PER_STAGE = CA + CB;                           // DMA instructions per wave and stage
#pragma unroll
    for (int st = 0; st < STAGES; ++st)
        if (st < nk) GEMM_DMA(st, st)
    if (nk >= STAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (STAGES - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8 wf0[NT], xf0[MT], wf1[NT], xf1[MT];
    GEMM_FRAGS(wf0, xf0, 0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % STAGES;
        GEMM_FRAGS(wf1, xf1, buf, 1)
        GEMM_MFMAS(wf0, xf0)
        GEMM_INTERLEAVE()
        GEMM_FRAGS(wf0, xf0, buf, 2)
        GEMM_MFMAS(wf1, xf1)
        GEMM_INTERLEAVE()
        GEMM_FRAGS(wf1, xf1, buf, 3)
        GEMM_MFMAS(wf0, xf0)
        GEMM_INTERLEAVE()
        // (in the tail fewer stages are in flight than the count assumes: wait for everything there)
        if (STAGES > 2 && kt + STAGES - 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PER_STAGE * (STAGES - 2)) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + STAGES < nk) GEMM_DMA(kt + STAGES, buf)
        if (kt + 1 < nk) GEMM_FRAGS(wf0, xf0, (kt + 1) % STAGES, 0)
        GEMM_MFMAS(wf1, xf1)
    }

    // ---------------- epilogue ----------------
    // The MFMA result layout gives each lane 4 consecutive n of ONE output row, 32 rows per
    // instruction: stored directly that is 64 separate 8/16-byte segments per store.  Instead each
    // wave transposes one 32(m) x NT*32(n) strip at a time through a private LDS patch (row pitch
    // NT*32+4 floats keeps the ds_write_b128 conflict-free) and then walks it row-major, so that
    // every global access is a 16-byte vector with LPR consecutive lanes covering one contiguous
    // row segment — full 128/256-byte lines for stores and for the residual read-modify-write.
    // All LDS reads of the main loop completed before its last barrier, so no extra barrier here.
    constexpr bool OUT_BF16 = (EPI == OMH_EPI_BF16 || EPI == OMH_EPI_GELU_BF16 || EPI == OMH_EPI_GELU_ERF_BF16 ||
                               EPI == OMH_EPI_GELU_BWD_BF16);
    constexpr int PITCH = NT * 32 + 4;                 // floats
    constexpr int VEC = OUT_BF16 ? 8 : 4;              // elements per 16-byte global access
    constexpr int LPR = NT * 32 / VEC;                 // lanes per row
    constexpr int RPP = 64 / LPR;                      // rows per pass
    constexpr int PASSES = 32 / RPP;
    static_assert(WM * WN * 32 * PITCH * 4 <= STAGES * STAGE_BYTES, "epilogue patch does not fit the staging LDS");
    float* ep = (float*)smem + wave * (32 * PITCH);
    float* Cf = (float*)p.C + zb * p.strideC;
    uint16_t* Ch = (uint16_t*)p.C + zb * p.strideC;
    // ABI v5 (batch == 1): out-of-place residual input, bf16 aux tensor (OUT: acc + bias of RESID / GELU; IN: the
    // pre-activation of GELU_BWD)
    const float* Cin = (EPI == OMH_EPI_RESID && p.c_in) ? p.c_in + zb * p.strideC : Cf;
    uint16_t* Ax = (uint16_t*)p.aux;
    constexpr bool AUX_OUT = (EPI == OMH_EPI_RESID || EPI == OMH_EPI_GELU_BF16);
    constexpr bool AUX_IN = (EPI == OMH_EPI_GELU_BWD_BF16);
    const bool aux_vec = Ax && (p.ldaux % 8) == 0 && (((uintptr_t)Ax) & 15) == 0;
    const int rr = lane / LPR, cc = (lane % LPR) * VEC;
    const int n = n0 + wn * NT * 32 + cc;
    const bool n_any = n < p.N;
    const bool vec_ok = (p.ldc % VEC) == 0 && (((uintptr_t)(OUT_BF16 ? (void*)Ch : (void*)Cf)) & 15) == 0 &&
                        n + VEC <= p.N;
    float bn[VEC], g0[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const bool ok = n + e < p.N;
        bn[e] = (p.bias_mode == OMH_BIAS_N && p.bias && ok) ? p.bias[n + e] : 0.f;
        g0[e] = (EPI == OMH_EPI_RESID) ? p.gate_const + ((p.gate0 && ok) ? p.gate0[n + e] : 0.f) : 0.f;
    }
#pragma unroll
    for (int im = 0; im < MT; ++im) {
        const int mrow = m0 + (wm * MT + im) * 32;
        if (mrow >= p.M) break;                                    // wave-uniform
        const float bias_m = (p.bias_mode == OMH_BIAS_M && p.bias && mrow + li < p.M) ? p.bias[mrow + li] : 0.f;
        // read-modify-write epilogues: fetch the old values first, their latency hides under the transposition
        constexpr bool RMW = (EPI == OMH_EPI_RESID || EPI == OMH_EPI_F32_ACCUM);
        float4 cold[RMW ? PASSES : 1];
        if (RMW && vec_ok) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int m = mrow + ps * RPP + rr;
                cold[ps] = (m < p.M) ? *(const float4*)(Cin + (int64_t)m * p.ldc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        uint4 apre[AUX_IN ? PASSES : 1];                            // GELU_BWD: the pre-activations of this strip
        if (AUX_IN && vec_ok && aux_vec) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int m = mrow + ps * RPP + rr;
                apre[ps] = (m < p.M) ? *(const uint4*)(Ax + (int64_t)m * p.ldaux + n) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int in = 0; in < NT; ++in)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)(ep + li * PITCH + in * 32 + 8 * gq + 4 * lh) =
                    make_float4(acc[im][in][4 * gq] + bias_m, acc[im][in][4 * gq + 1] + bias_m,
                                acc[im][in][4 * gq + 2] + bias_m, acc[im][in][4 * gq + 3] + bias_m);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-private patch: no barrier needed
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int r = ps * RPP + rr;
            const int m = mrow + r;
            float v[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 t = *(const float4*)(ep + r * PITCH + cc + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            if (m >= p.M || !n_any) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += bn[e];
            if (AUX_OUT && Ax) {                                       // bf16(acc + bias) on the side
                uint16_t* ap = Ax + (int64_t)m * p.ldaux + n;
                if (vec_ok && aux_vec) {
                    if (VEC == 8) {
                        uint4 pk;
                        pk.x = pack_bf2(v[0], v[1]);
                        pk.y = pack_bf2(v[2], v[3]);
                        pk.z = pack_bf2(v[4 % VEC], v[5 % VEC]);
                        pk.w = pack_bf2(v[6 % VEC], v[7 % VEC]);
                        *(uint4*)ap = pk;
                    } else {
                        *(uint2*)ap = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.N) ap[e] = f2bf(v[e]);
                }
            }
            if (AUX_IN) {                                              // dx = dy * gelu_tanh'(pre-activation)
                if (vec_ok && aux_vec) {
                    const uint4 a4 = apre[AUX_IN ? ps : 0];
                    const uint32_t w_[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        v[e] *= gelu_tanh_grad(__uint_as_float((e & 1) ? (w_[(e >> 1) & 3] & 0xffff0000u) : (w_[(e >> 1) & 3] << 16)));
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (n + e < p.N) v[e] *= gelu_tanh_grad(bf2f(Ax[(int64_t)m * p.ldaux + n + e]));
                }
            }
            if (EPI == OMH_EPI_GELU_BF16) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = gelu_tanh(v[e]);
            }
            if (EPI == OMH_EPI_GELU_ERF_BF16) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.7071067811865476f));
            }
            if (EPI == OMH_EPI_RESID) {
                if (p.gate1) {
                    const float* g1 = p.gate1 + (int64_t)(m / p.gate_rows) * p.gate1_stride + n;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] *= g0[e] + ((n + e < p.N) ? g1[e] : 0.f);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] *= g0[e];
                }
            }
            const int64_t off = (int64_t)m * p.ldc + n;
            if (OUT_BF16) {
                if (vec_ok) {
                    uint4 pk;
                    pk.x = pack_bf2(v[0], v[1]);
                    pk.y = pack_bf2(v[2], v[3]);
                    pk.z = pack_bf2(v[4 % VEC], v[5 % VEC]);
                    pk.w = pack_bf2(v[6 % VEC], v[7 % VEC]);
                    *(uint4*)(Ch + off) = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.N) Ch[off + e] = f2bf(v[e]);
                }
            } else if (EPI == OMH_EPI_F32) {
                if (vec_ok) {
                    *(float4*)(Cf + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.N) Cf[off + e] = v[e];
                }
            } else {  // RESID / F32_ACCUM: read-modify-write
                if (vec_ok) {
                    float4 o = cold[RMW ? ps : 0];
                    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                    *(float4*)(Cf + off) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) if (n + e < p.N) Cf[off + e] = Cin[off + e] + v[e];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // patch is rewritten by the next strip
    }
}

template <int EPI, int WM, int WN, int MT, int NT, int STAGES, bool BKM = false>
int launch_cfg(const omh_gemm_args& a, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int LDS = STAGES * (BM + BN) * BK * 2;
    auto kern = gemm_bf16_nt_kernel<EPI, WM, WN, MT, NT, STAGES, BKM>;
    static bool attr_set = false;            // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    GemmGeom g;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = (a.N + BN - 1) / BN;
    // m-tiles walked together by one XCD: as many A panels (BM x K bf16) as stay resident in ~3 MiB of its 4 MiB
    // L2 while the n-tiles stream past (measured: 4 at K=1536 and 2 at K>=6144 beat the former fixed 8 by 2-4 %)
    const char* gme = omh_opt(OMH_OPT_GEMM_GROUP_M);
    const int fit = (int)(3200000LL / ((int64_t)BM * a.K * 2));
    g.group_m = gme ? atoi(gme) : (fit < 2 ? 2 : (fit > 8 ? 8 : fit));
    dim3 grid(g.tiles_m * g.tiles_n, 1, a.batch);
    omh_clear_status();
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), LDS, s, a, g);
    return omh_launch_status();
}

// Estimated time (us) of the 256x256 and the 128x128 tile configurations on `a` (see launch() below for the model).
static float round_cost(int64_t tiles, int64_t slots, int64_t light_max, float light, float heavy) {
    const int64_t full = tiles / slots, last = tiles % slots;
    return (float)full * heavy + (last == 0 ? 0.0f : (last <= light_max ? light : heavy));
}
static void tile_costs(const omh_gemm_args& a, float& cost_big, float& cost_small) {
    const int64_t big_tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256) * a.batch;
    const int64_t mid_tiles = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
    const float k = (float)a.K * (1.0f / 1536.0f);
    cost_big = round_cost(big_tiles, 256, 192, 9.0f + 27.5f * k, 9.0f + 32.0f * k);
    cost_small = round_cost(mid_tiles, 512, 160, 4.0f + 15.5f * k, 5.0f + 21.0f * k);
}

template <int EPI, bool BKM = false>
int launch(const omh_gemm_args& a, hipStream_t s) {
    // Tile configuration by estimated time = sum over the rounds the tiles take on the chip, with per-round times
    // measured on MI355X (tools/gemm_tile_probe.py; k = K / 1536; a round is "light" when few workgroups share the
    // L2 / fabric, "heavy" when the chip is mostly full):
    //   256x256 tile, one workgroup per CU  (256 slots): light (<= 192 tiles) 9 + 27.5 k us, heavy 9 + 32 k us
    //   128x128 tile, two workgroups per CU (512 slots): light (<= 160 tiles) 4 + 15.5 k us, heavy 5 + 21 k us
    //   64x64 tile: only when even the 128x128 tiles cannot give every CU a workgroup (context projections, one
    //   training clip) — there it wins by 15-40 %.
    // The former rule (big iff >= 256 big tiles) put e.g. 6240 x 1536 x 1536 (150 big tiles = ONE round of 37 us) on
    // 588 small tiles = TWO rounds, and 300 big tiles (two rounds) ahead of 1176 small ones (2 + a light one):
    // 10-38 % slower on 11 of the 28 shapes probed, all of them in the training step and the S = 1560 forward.
    const int64_t big_tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256) * a.batch;
    const int64_t mid_tiles = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
    const char* force = omh_opt(OMH_OPT_GEMM_TILE);               // "big" / "small" / "tiny": test / benchmarking override
    if (force) {
        if (force[0] == 'm' && !BKM) return launch_cfg<EPI, 2, 4, 3, 2, 2, false>(a, s);      // "mid192"
        if (force[0] == 'b') return launch_cfg<EPI, 2, 4, 4, 2, 2, BKM>(a, s);
        if (force[0] == 't' && !BKM) return launch_cfg<EPI, 2, 2, 1, 1, 2>(a, s);
        return launch_cfg<EPI, 2, 2, 2, 2, 2, BKM>(a, s);
    }
    if (BKM && mid_tiles < 256) return launch_cfg<EPI, 2, 2, 2, 2, 2, BKM>(a, s);   // no 64x64 k-major variant
    if (mid_tiles < 256) return launch_cfg<EPI, 2, 2, 1, 1, 2>(a, s);
    const char* rule = omh_opt(OMH_OPT_GEMM_RULE);         // "old": the former rule, for A/B timing on one box
    if (rule && rule[0] == 'o')
        return big_tiles >= 256 ? launch_cfg<EPI, 2, 4, 4, 2, 2, BKM>(a, s) : launch_cfg<EPI, 2, 2, 2, 2, 2, BKM>(a, s);
    float cost_big, cost_small;
    tile_costs(a, cost_big, cost_small);
    // 192 x 256 tiles (8 waves, wave tile 96 x 64; round 3): a tile costs 3/4 of a 256 x 256 one, so when the big tiles
    // leave a round mostly empty (M = 6 240, N = 1 536: 150 tiles on 256 CUs) 198 of these take one round of 0.75
    if (!BKM) {
        const int64_t mid192 = (int64_t)((a.M + 191) / 192) * ((a.N + 255) / 256) * a.batch;
        const float k = (float)a.K * (1.0f / 1536.0f);
        const float cost_192 = round_cost(mid192, 256, 192, 8.0f + 21.0f * k, 8.0f + 24.5f * k);
        if (cost_192 < fminf(cost_big, cost_small)) return launch_cfg<EPI, 2, 4, 3, 2, 2, false>(a, s);
    }
    if (cost_big <= cost_small) return launch_cfg<EPI, 2, 4, 4, 2, 2, BKM>(a, s);
    return launch_cfg<EPI, 2, 2, 2, 2, 2, BKM>(a, s);
}

}  // namespace

// gemm_w64.hip: 256 x 384 tile, one wave per SIMD, generated k loop
bool omh_gemm_w64_takes(const omh_gemm_args& a);
int omh_launch_gemm_w64(const omh_gemm_args& a, hipStream_t stream);
// ... and its 256 x 192 gated-residual variant with the old C tile prefetched during the k loop
bool omh_gemm_w64_r192_takes(const omh_gemm_args& a);
int omh_launch_gemm_w64_r192(const omh_gemm_args& a, hipStream_t stream);
bool omh_gemm_w64_bf16m_takes(const omh_gemm_args& a);
int omh_launch_gemm_w64_bf16m(const omh_gemm_args& a, hipStream_t stream);
bool omh_gemm_w64_n192_takes(const omh_gemm_args& a);
int omh_launch_gemm_w64_n192(const omh_gemm_args& a, hipStream_t stream);
// ... split K over 2..4 slices of that stream + a combine launch, for few-row long-contraction products (ABI v9)
int omh_gemm_splitk_slices(const omh_gemm_args& a);
int64_t omh_gemm_splitk_workspace(const omh_gemm_args& a);
int omh_launch_gemm_splitk(const omh_gemm_args& a, int S, hipStream_t stream);

// ... the fused q | k | v projection: columns below n_split to C, the rest transposed to aux (ABI v10)
bool omh_gemm_w64_qkv_takes(const omh_gemm_args& a);
int omh_launch_gemm_w64_qkv(const omh_gemm_args& a, hipStream_t stream);
bool omh_gemm_w64_p256_takes(const omh_gemm_args& a);
int omh_launch_gemm_w64_p256(const omh_gemm_args& a, hipStream_t stream);

static int launch_8w(const omh_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case OMH_EPI_BF16:      return launch<OMH_EPI_BF16>(a, s);
        case OMH_EPI_F32:       return launch<OMH_EPI_F32>(a, s);
        case OMH_EPI_GELU_BF16: return launch<OMH_EPI_GELU_BF16>(a, s);
        case OMH_EPI_RESID:     return launch<OMH_EPI_RESID>(a, s);
        case OMH_EPI_F32_ACCUM: return launch<OMH_EPI_F32_ACCUM>(a, s);
        case OMH_EPI_GELU_ERF_BF16: return launch<OMH_EPI_GELU_ERF_BF16>(a, s);
        case OMH_EPI_GELU_BWD_BF16: return launch<OMH_EPI_GELU_BWD_BF16>(a, s);
        default: return OMH_E_BADARG;
    }
}

extern "C" int omh_gemm_bf16(const omh_gemm_args* args, omh_stream_t stream) {
    if (!args || !args->A || !args->B || !args->C) return OMH_E_BADARG;
    const omh_gemm_args& a = *args;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch <= 0) return OMH_E_BADARG;
    if ((a.K & 7) || (a.lda & 7) || (a.ldb & 7)) return OMH_E_ALIGN;
    if (a.epilogue == OMH_EPI_BF16_SPLIT_T) {
        // q | k | v in one product: one launch of the 256 x 384 stream where it fills the chip (>= one round of tiles),
        // else the two products it stands for, through this same entry point.  OMH GEMM_QKV = "0": always the two.
        if (!a.aux || a.batch != 1 || a.b_kmajor || a.c_in || a.n_split <= 0 || a.n_split >= a.N) return OMH_E_BADARG;
        if (a.bias && a.bias_mode != OMH_BIAS_N) return OMH_E_BADARG;
        if ((a.n_split & 7) || (a.M & 7) || (a.ldaux & 7) || a.ldaux < a.M || ((uintptr_t)a.aux & 15)) return OMH_E_ALIGN;
        if ((a.ldc & 7) || ((uintptr_t)a.C & 15)) return OMH_E_ALIGN;
        if (a.ldc < a.n_split) return OMH_E_SHAPE;                        // C rows are n_split wide: a narrower pitch would overlap them
        const char* q = omh_opt(OMH_OPT_GEMM_QKV);
        const char* gk = omh_opt(OMH_OPT_GEMM_KERNEL);
        const bool off = (q && q[0] == '0') || (gk && gk[0] == '8') || omh_opt(OMH_OPT_GEMM_TILE);
        const int64_t tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 383) / 384);
        if (!off && omh_gemm_w64_qkv_takes(a) && (tiles >= 256 || (q && q[0] == '1'))) {
            omh_clear_status();
            omh_launch_gemm_w64_qkv(a, (hipStream_t)stream);
            return omh_launch_status();
        }
        omh_gemm_args qk = a;                                             // C[M, n_split] = bf16(A B[:n_split]^T + bias)
        qk.N = a.n_split; qk.epilogue = OMH_EPI_BF16; qk.aux = nullptr; qk.ldaux = 0; qk.n_split = 0;
        const int rc = omh_gemm_bf16(&qk, stream);
        if (rc) return rc;
        omh_gemm_args v = a;                                              // aux[N - n_split, M] = bf16(B[n_split:] A^T + bias[m])
        v.A = (const char*)a.B + (int64_t)a.n_split * a.ldb * 2; v.lda = a.ldb;
        v.B = a.A; v.ldb = a.lda;
        v.C = a.aux; v.ldc = a.ldaux;
        v.M = a.N - a.n_split; v.N = a.M;
        v.epilogue = OMH_EPI_BF16; v.aux = nullptr; v.ldaux = 0; v.n_split = 0;
        v.bias = a.bias ? a.bias + a.n_split : nullptr; v.bias_mode = a.bias ? OMH_BIAS_M : OMH_BIAS_NONE;
        return omh_gemm_bf16(&v, stream);
    }
    if (a.b_kmajor) {                       // B = [K, N] row-major: dx = dy W on the weight as stored
        if ((a.N & 7) || a.ldb < a.N) return OMH_E_ALIGN;
        if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return OMH_E_ALIGN;
        if ((a.strideA & 7) || (a.strideB & 7) || (a.strideC & 3)) return OMH_E_ALIGN;
        if (((int64_t)a.M + 128) * a.lda * 2 >= 0x7fffffffLL || ((int64_t)a.K + 64) * a.ldb * 2 >= 0x7fffffffLL)
            return OMH_E_SHAPE;
        hipStream_t s = (hipStream_t)stream;
        switch (a.epilogue) {
            case OMH_EPI_BF16:      return launch<OMH_EPI_BF16, true>(a, s);
            case OMH_EPI_F32:       return launch<OMH_EPI_F32, true>(a, s);
            case OMH_EPI_F32_ACCUM: return launch<OMH_EPI_F32_ACCUM, true>(a, s);
            default: return OMH_E_SHAPE;     // the backward needs no other epilogue on a k-major B
        }
    }
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return OMH_E_ALIGN;
    if ((a.strideA & 7) || (a.strideB & 7) || (a.strideC & 3)) return OMH_E_ALIGN;
    if (a.epilogue == OMH_EPI_RESID && a.gate1 && a.gate_rows <= 0) return OMH_E_BADARG;
    const bool v5 = a.aux || a.c_in || a.epilogue == OMH_EPI_GELU_BWD_BF16;        // fused training epilogues
    if (v5) {
        if (a.batch != 1) return OMH_E_SHAPE;
        if (a.epilogue == OMH_EPI_GELU_BWD_BF16 && !a.aux) return OMH_E_BADARG;
        if (a.aux && a.epilogue != OMH_EPI_RESID && a.epilogue != OMH_EPI_GELU_BF16 && a.epilogue != OMH_EPI_GELU_BWD_BF16)
            return OMH_E_BADARG;
        if (a.c_in && a.epilogue != OMH_EPI_RESID) return OMH_E_BADARG;
        if (a.aux && (a.ldaux < a.N || (a.ldaux & 7) || ((uintptr_t)a.aux & 15))) return OMH_E_ALIGN;
        if (a.c_in && ((uintptr_t)a.c_in & 15)) return OMH_E_ALIGN;
    }
    // 32-bit buffer offsets: each operand (one batch element) must stay below 2 GiB
    if (((int64_t)a.M + 128) * a.lda * 2 >= 0x7fffffffLL || ((int64_t)a.N + 128) * a.ldb * 2 >= 0x7fffffffLL)
        return OMH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (a.workspace && !omh_opt(OMH_OPT_GEMM_TILE) && !omh_opt(OMH_OPT_GEMM_KERNEL)) {
        // few rows, long contraction (M = 1 560 / 3 120 against K = 8 960): the contraction in slices on the 256 x 192
        // stream, then one combine launch (gemm_w64.hip).  Only with a workspace of the size the query below returns.
        const int S = omh_gemm_splitk_slices(a);
        if (S > 1) {
            if (((uintptr_t)a.workspace & 15) || a.workspace_bytes < omh_gemm_splitk_workspace(a)) return OMH_E_BADARG;
            omh_clear_status();
            omh_launch_gemm_splitk(a, S, s);
            return omh_launch_status();
        }
    }
    {
        // OMH_GEMM_KERNEL = "w64": the 256 x 384 stream kernel wherever it applies; "8w": never; unset: where it
        // applies AND fills the chip (>= 256 tiles, last round of tiles at least 3/4 full or >= 4 rounds)
        const char* gk = omh_opt(OMH_OPT_GEMM_KERNEL);
        // (the fused training epilogues stay on the 8-wave kernels — except GELU_BWD, which the big stream has, and the
        // out-of-place / aux-writing gated residual, which the 256 x 192 stream has)
        // GELU_BWD on the 256 x 384 stream: built and bit-identical (test_gemm_w64_gelu_backward_stream), but measured EQUAL
        // to the 8-wave kernel in isolation (6240 x 8960 x 1536: 204-225 vs 202-208 us; 1560 rows 58 vs 56-62) — the 357 us
        // the 8-wave kernel shows inside a training step is the weight-gradient stream sharing the chip, not the kernel.
        // Opt-in: OMH_GEMM_W64_GBWD=1.
        const char* gbwd = omh_opt(OMH_OPT_GEMM_W64_GBWD);
        // GELU + pre-activation to aux (the FFN-up projection of a training forward, ABI v5) on the 256 x 384 stream
        // ("geluaux": bit-identical, test_gemm_w64_gelu_stream_with_the_pre_activation); OMH_GEMM_W64_GAUX=0: 8-wave kernels
        const char* gaux = omh_opt(OMH_OPT_GEMM_W64_GAUX);
        const bool gelu_aux = a.epilogue == OMH_EPI_GELU_BF16 && a.aux && !a.c_in && !(gaux && gaux[0] == '0');
        const bool v5_8w = v5 && !gelu_aux && !(a.epilogue == OMH_EPI_GELU_BWD_BF16 && !a.c_in && gbwd && gbwd[0] == '1');
        const bool force = gk && gk[0] == 'w', never = v5_8w || (gk && gk[0] == '8') || (!force && omh_opt(OMH_OPT_GEMM_TILE));
        // round 6 experiment (builds with OMH_GW64_P256=1 only; measured slower, never the default): the 256 x 256 streams
        // with two k tiles of operands in flight in registers (gemm_w64.hip: K_*_P).  GEMM_W64_P256 = "1": wherever they apply.
        {
            const char* p2 = omh_opt(OMH_OPT_GEMM_W64_P256);
            const bool off = (gk && gk[0] == '8') || (!force && omh_opt(OMH_OPT_GEMM_TILE));
            const bool on = p2 && p2[0] != '0';
            if (on && !off && !v5 && omh_gemm_w64_p256_takes(a)) {
                omh_clear_status();
                omh_launch_gemm_w64_p256(a, s);
                return omh_launch_status();
            }
        }
        // gated residual with a short contraction (o-projections: K = dim): the 256 x 192 stream that requests the old C
        // tile during its k loop.  OMH_GEMM_W64_R192 = 0 / 1 forces it off / on (A/B timing, tests).
        {
            const char* r192 = omh_opt(OMH_OPT_GEMM_W64_R192);
            const bool off = r192 && r192[0] == '0', on = r192 && r192[0] == '1';
            const bool never192 = (gk && gk[0] == '8') || (!force && omh_opt(OMH_OPT_GEMM_TILE));
            if (!never192 && !off && omh_gemm_w64_r192_takes(a)) {          // (also the training epilogues: c_in, aux)
                // measured (round 4, one box, R192 = 0 / 1): 32760 x 1536 x 1536 206 -> 186 us, 21840 rows 148 -> 119,
                // 6240 rows 51 -> 39 (K = 8960: 177 -> 151), 3120 rows 33.5 -> 35.4 (too few tiles), 32760 x 1536 x 8960
                // 697 -> 840 (a long k loop amortises the big stream's epilogue: stays there)
                const int64_t t192 = (int64_t)((a.M + 255) / 256) * ((a.N + 191) / 192);
                const int64_t t384 = (int64_t)((a.M + 255) / 256) * ((a.N + 383) / 384);
                if (on || (t192 >= 192 && (a.K <= 3072 || t384 < 256))) {
                    omh_clear_status();
                    omh_launch_gemm_w64_r192(a, s);
                    return omh_launch_status();
                }
            }
        }
        // bf16 output with a per-row bias (V^T = Wv h^T + bv: 1536 x 32 760 x 1536) on the 256 x 384 stream.  Its tile count
        // decides: 6 x 86 = 516 tiles is two rounds of 256 persistent workgroups plus FOUR tiles; the last tile column
        // (120 of the 32 760 columns) is therefore handed to the 8-wave kernels as a second small launch and the stream
        // takes 6 x 85 = 510 tiles = two rounds.  Built, bit-identical (test_gemm_w64_per_row_bias_stream) and measured
        // EQUAL to the 8-wave kernel (144-159 vs 141-149 us: 24 k steps per tile do not amortise the stream's per-tile
        // cost, and the tail is a second launch): opt-in, OMH_GEMM_W64_BF16M = 1.
        {
            const char* bm = omh_opt(OMH_OPT_GEMM_W64_BF16M);
            const bool on = bm && bm[0] == '1', off = !on;
            if (!never && !off && omh_gemm_w64_bf16m_takes(a)) {
                const int tm = (a.M + 255) / 256, tn = (a.N + 383) / 384;
                auto util = [&](int t) { const int64_t n = (int64_t)tm * t; return n <= 0 ? 0.0 : (double)n / (double)(((n + 255) / 256) * 256); };
                const bool cut = tn > 1 && util(tn - 1) > util(tn) + 0.05;
                const int tmain = cut ? tn - 1 : tn;
                if (on || ((int64_t)tm * tmain >= 256 && util(tmain) >= 0.9)) {
                    omh_gemm_args m = a;
                    if (cut) m.N = tmain * 384;
                    omh_clear_status();
                    omh_launch_gemm_w64_bf16m(m, s);
                    int rc = omh_launch_status();
                    if (rc || !cut) return rc;
                    omh_gemm_args r = a;                                  // columns [tmain * 384, N): rows of B, columns of C
                    r.N = a.N - m.N;
                    r.B = (const char*)a.B + (int64_t)m.N * a.ldb * 2;
                    r.C = (char*)a.C + (int64_t)m.N * 2;
                    return launch_8w(r, s);
                }
            }
        }
        // plain fp32 / bf16 products too small for the 256 x 384 stream (< 128 of its tiles) but with 160 .. 256 tiles
        // of 256 x 192: one round of the narrow stream (OMH_GEMM_W64_N192 = 0 / 1: off / wherever it applies)
        {
            const char* n192 = omh_opt(OMH_OPT_GEMM_W64_N192);
            const bool off = n192 && n192[0] == '0', on = n192 && n192[0] == '1';
            if (!never && !off && omh_gemm_w64_n192_takes(a)) {
                const int64_t t192 = (int64_t)((a.M + 255) / 256) * ((a.N + 191) / 192);
                const int64_t t384 = (int64_t)((a.M + 255) / 256) * ((a.N + 383) / 384);
                if (on || (!force && t384 < 128 && t192 >= 160 && t192 <= 256)) {
                    omh_clear_status();
                    omh_launch_gemm_w64_n192(a, s);
                    return omh_launch_status();
                }
            }
        }
        if (!never && omh_gemm_w64_takes(a)) {
            // (a ragged last tile column stays in this kernel, masked: handing N % 384 = 128 columns of the FFN's 8960
            // to the 8-wave kernel as a second launch measured 778 us against 757 us)
            // Rounds of 256 persistent workgroups at 10 + 47 K/1536 us per tile (57 us at K = 1536; a last round of
            // <= 128 tiles 0.8 of that) against the 8-wave kernels' model — whose full rounds at large K cost
            // 8 + 36 K/1536 us when compared here (tools/gemm_dispatch_probe.py: 15 shapes of the training step, config 4,
            // config 5 and the teacher pair); at least half a round of tiles.
            const int64_t tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 383) / 384);
            const float kk = (float)a.K * (1.0f / 1536.0f);
            float cost_big, cost_small;
            tile_costs(a, cost_big, cost_small);
            const int64_t big_tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
            cost_big = round_cost(big_tiles, 256, 192, 9.0f + 27.5f * kk, 8.0f + 36.0f * kk);
            const float tw = 10.0f + 47.0f * kk;
            const float cost_w64 = round_cost(tiles, 256, 128, 0.8f * tw, tw);
            if (force || (tiles >= 128 && cost_w64 < fminf(cost_big, cost_small))) {
                omh_clear_status();
                omh_launch_gemm_w64(a, s);
                return omh_launch_status();
            }
        }
    }
    return launch_8w(a, s);
}

extern "C" int64_t omh_gemm_workspace_bytes(const omh_gemm_args* args) {
    if (!args || !args->A || !args->B || !args->C || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
    if (omh_opt(OMH_OPT_GEMM_TILE) || omh_opt(OMH_OPT_GEMM_KERNEL)) return 0;     // a forced kernel family: no slices
    if ((args->K & 7) || (args->lda & 7) || (args->ldb & 7) || ((uintptr_t)args->A & 15) || ((uintptr_t)args->B & 15)) return 0;
    return omh_gemm_splitk_workspace(*args);
}
