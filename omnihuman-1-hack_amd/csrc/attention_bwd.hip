// Flash-attention backward, head_dim 128, bf16 operands, fp32 gradients, for gfx950.
//
// The reference gets this from autograd through flash_attn_varlen_func (attention.py:96-127) inside the
// per-block checkpoint recompute of the training step (model.py:544-548, distilled_trainer.py:289-301).
// Given q, k, v, the forward output o, its gradient dO and the forward's log-sum-exp:
//     P   = exp(scale * q k^T - lse)            (keys >= k_lens[b] excluded)
//     dV  = P^T dO
//     dP  = dO V^T ,  delta_i = sum_j P_ij dP_ij
//     dS  = P * (dP - delta)
//     dQ  = scale * dS K ,  dK = scale * dS^T Q
// Two launches, no atomics (bitwise repeatable), nothing of size Lq x Lk ever leaves the chip:
//   dQ kernel      one workgroup per 128 queries of one head (4 waves x 32 queries), loops over 32-key tiles
//                  twice: a first pass for delta (written out for the other kernel), a second for dQ;
//   dK/dV kernel   one workgroup per 128 keys of one head (4 waves x 32 keys), loops over 32-query tiles.
// delta is taken as sum_j P_ij dP_ij from the recomputed fp32 P and dP rather than flash-attn's rowsum(dO * o):
// with the forward's bf16 o the row sums of dS vanish only to bf16 accuracy, and that residue is all a
// near-null gradient (the K bias, to which the softmax is invariant) consists of — measured 15-18 % error on the
// cross-attention K-bias gradients of the 13-layer test model with rowsum(dO * o), inside that test's 10 % bound
// with this; the extra pass is 16 of 72 MFMAs per 32x32 block at sizes where the launch count is what matters.
// Both big kernels recompute S and dP with v_mfma_f32_32x32x16_bf16 in the orientation that leaves the
// contraction index of the NEXT product in consecutive registers of one lane (the forward kernel's trick,
// attention.hip): in the dK/dV kernel S is [query][key] with lane = key, so P and dS are directly the B operands
// of  dV^T = dO^T P  and  dK^T = Q^T dS ; in the dQ kernel S^T is [key][query] with lane = query, so dS^T is the B
// operand of  dQ^T = K^T dS^T  and lse / delta are per-lane scalars.  The A operands whose contraction index is
// the sequence position (dO^T, Q^T, K^T) are read from transposed bf16 copies [B, H*128, ld] made once per call by
// omh_transpose_bf16 (zero padded), the same layout the forward uses for V^T.
// Sized for the training clips of BASELINE config 3 (S = 1560 tokens, 512 context tokens): simple single-buffered
// LDS staging; the work per call is ~60 GFLOP, what matters there is 2 launches (+3 transposes) per call for the
// whole batch instead of ~14 per sample.
#include "omh_common.h"

namespace {

constexpr int D = 128;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int swap_bits23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
// [32 rows][128 d] bf16 tile: 256-byte rows = 16 slots of 16 B, slot ^= row & 15
__device__ __forceinline__ uint32_t r_addr(int row, int slot) {
    return (uint32_t)(row * 256 + ((slot ^ (row & 15)) << 4));
}
// [128 d][32 positions] bf16 tile: 64-byte rows = 4 slots, slot ^= (row >> 2) & 3
__device__ __forceinline__ uint32_t t_addr(int row, int slot) {
    return (uint32_t)(row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
}

// component-wise (a ?: on the whole struct becomes a select between two stack slots: scratch traffic)
__device__ __forceinline__ uint4 keep_if(bool ok, const uint4 v) {
    return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
}
__device__ __forceinline__ uint4 ld16(const uint16_t* p, bool ok) {
    return ok ? *(const uint4*)p : make_uint4(0u, 0u, 0u, 0u);
}

// P and dS of one 32x32 block from the two score accumulators; lv / dl are the log2-domain lse and delta of the
// position each REGISTER stands for (dK/dV kernel) or of the lane (dQ kernel, all 16 equal).
__device__ __forceinline__ void p_and_ds(const f32x16& s, const f32x16& dp, const float* lv, const float* dl,
                                         const bool* ok, float sc, float scale, bf16x8* pf, bf16x8* dsf) {
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = ok[r] ? __builtin_amdgcn_exp2f(fmaf(s[r], sc, -lv[r])) : 0.f;
        ds[r] = p[r] * (dp[r] - dl[r]) * scale;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        u32x4 cp, cd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cp[e] = pack_bf2(p[8 * a + 2 * e], p[8 * a + 2 * e + 1]);
            cd[e] = pack_bf2(ds[8 * a + 2 * e], ds[8 * a + 2 * e + 1]);
        }
        pf[a] = __builtin_bit_cast(bf16x8, cp);
        dsf[a] = __builtin_bit_cast(bf16x8, cd);
    }
}

// ---------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256)
void attn_bwd_dkdv_kernel(const omh_attn_bwd_args p, const int k_blocks) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 8192 + 256];
    unsigned char* Qs = smem;
    unsigned char* dOs = smem + 8192;
    unsigned char* QTs = smem + 16384;
    unsigned char* dOTs = smem + 24576;
    float* lse_s = (float*)(smem + 32768);
    float* dl_s = lse_s + 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wid = blockIdx.x;
    const int kb = wid % k_blocks, bh = wid / k_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);

    const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
    const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
    const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
    const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
    const uint16_t* QT = (const uint16_t*)p.qt + (int64_t)b * p.qt_bs + (int64_t)head * D * p.ldq;
    const uint16_t* DOT = (const uint16_t*)p.dot + (int64_t)b * p.qt_bs + (int64_t)head * D * p.ldq;
    const float* LSE = p.lse + ((int64_t)b * p.H + head) * p.Lq;
    const float* DEL = p.delta + ((int64_t)b * p.H + head) * p.Lq;

    // this lane's key (B-operand column) and its K / V rows, d = 16 kk + 8 lh + 0..7
    const int key = kb * 128 + wave * 32 + li;
    const bool key_ok = key < klen;
    bf16x8 kf[8], vf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int64_t off = (int64_t)key * p.k_rs + kk * 16 + lh * 8;
        kf[kk] = __builtin_bit_cast(bf16x8, ld16(K + off, key < p.Lk));
        vf[kk] = __builtin_bit_cast(bf16x8, ld16(V + off, key < p.Lk));
    }
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ok[r] = key_ok;

    f32x16 dv[4], dk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[i][r] = 0.f; dk[i][r] = 0.f; }
    // q_prescaled: q / qt already carry scale * log2(e) (the forward's operands): scores are in the log2 domain as
    // they come, and dK = dS^T q' / log2(e)
    const float sc = p.q_prescaled ? 1.0f : p.scale * LOG2E;
    const float ds_scale = p.q_prescaled ? (1.0f / LOG2E) : p.scale;
    const int qrow_l = swap_bits23(li);

    // global -> register prefetch one tile ahead: the loads of tile t+1 fly while tile t is computed.  Loads are
    // unconditional from a clamped row (a select on the loaded value would make the wave wait for it at once);
    // rows past the end are zeroed when the registers go to LDS, between the two barriers of the next trip.
    // (explicit scalars, not arrays: indexed by a loop variable inside a conditional block hipcc kept the arrays in
    // scratch memory — store after every load, reload before every LDS write — which made the prefetch a loss)
    uint4 gq0, gq1, gdo0, gdo1, gqt0, gqt1, gdot0, gdot1;
    float glse = 0.f, gdel = 0.f;
#define BWD_KV_PREFETCH1(Q0, J)                                                                               \
    {                                                                                                         \
        const int c = tid + 256 * J;                                                                          \
        const int row = min((Q0) + (c >> 4), p.Lq - 1), slot = c & 15;                                        \
        gq##J = *(const uint4*)(Q + (int64_t)row * p.q_rs + slot * 8);                                        \
        gdo##J = *(const uint4*)(DO + (int64_t)row * p.o_rs + slot * 8);                                      \
        const int64_t off = (int64_t)(c >> 2) * p.ldq + (Q0) + (c & 3) * 8;                                   \
        gqt##J = *(const uint4*)(QT + off);                                                                   \
        gdot##J = *(const uint4*)(DOT + off);                                                                 \
    }
#define BWD_KV_PREFETCH(Q0)                                                                                   \
    {                                                                                                         \
        BWD_KV_PREFETCH1(Q0, 0) BWD_KV_PREFETCH1(Q0, 1)                                                       \
        if (tid < 32) {                                                                                       \
            const int q = min((Q0) + tid, p.Lq - 1);                                                          \
            glse = LSE[q];                                                                                    \
            gdel = DEL[q];                                                                                    \
        }                                                                                                     \
    }
#define BWD_KV_TO_LDS1(Q0, J)                                                                                 \
    {                                                                                                         \
        const int c = tid + 256 * J;                                                                          \
        const bool in = (Q0) + (c >> 4) < p.Lq;                                                               \
        *(uint4*)(Qs + r_addr(c >> 4, c & 15)) = keep_if(in, gq##J);      /* Q, dO rows: 32 x 16 chunks */    \
        *(uint4*)(dOs + r_addr(c >> 4, c & 15)) = keep_if(in, gdo##J);                                        \
        *(uint4*)(QTs + t_addr(c >> 2, c & 3)) = gqt##J;                  /* Q^T, dO^T: 128 x 4 chunks */     \
        *(uint4*)(dOTs + t_addr(c >> 2, c & 3)) = gdot##J;                                                    \
    }
    BWD_KV_PREFETCH(0)
    for (int q0 = 0; q0 < p.Lq; q0 += 32) {
        __syncthreads();                                   // every wave is done with the previous tile
        BWD_KV_TO_LDS1(q0, 0) BWD_KV_TO_LDS1(q0, 1)
        if (tid < 32) {
            const bool in = q0 + tid < p.Lq;
            const float l = (in && glse > -INFINITY) ? glse : INFINITY;       // no keys / past the end: P = 0
            lse_s[tid] = l * LOG2E;
            dl_s[tid] = in ? gdel : 0.f;
        }
        __syncthreads();
        if (q0 + 32 < p.Lq) BWD_KV_PREFETCH(q0 + 32)

        // S = Q K^T and dP = dO V^T as [query][key], lane = key; register r <-> query q0 + 16(r>>3) + 8 lh + (r&7)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 qa = *(const bf16x8*)(Qs + r_addr(qrow_l, 2 * kk + lh));
            const bf16x8 da = *(const bf16x8*)(dOs + r_addr(qrow_l, 2 * kk + lh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[kk], dp, 0, 0, 0);
        }
        float lv[16], dl[16];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lv[8 * a + e] = lse_s[16 * a + 8 * lh + e];
                dl[8 * a + e] = dl_s[16 * a + 8 * lh + e];
            }
        bf16x8 pf[2], dsf[2];
        p_and_ds(s, dp, lv, dl, ok, sc, ds_scale, pf, dsf);
        // dV^T += dO^T P ,  dK^T += Q^T dS   ([d][key], lane = key)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 ta = *(const bf16x8*)(dOTs + t_addr(db * 32 + li, 2 * a + lh));
                const bf16x8 tq = *(const bf16x8*)(QTs + t_addr(db * 32 + li, 2 * a + lh));
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta, pf[a], dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq, dsf[a], dk[db], 0, 0, 0);
            }
    }
    if (key < p.Lk) {
        const int64_t eo = (int64_t)b * p.dk_bs + (int64_t)key * p.dk_rs + head * D;
        if (p.out_bf16) {
            uint16_t* DK = (uint16_t*)p.dk + eo;
            uint16_t* DV = (uint16_t*)p.dv + eo;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = db * 32 + g * 8 + lh * 4;
                    *(uint2*)(DK + d0) = make_uint2(pack_bf2(dk[db][4 * g], dk[db][4 * g + 1]), pack_bf2(dk[db][4 * g + 2], dk[db][4 * g + 3]));
                    *(uint2*)(DV + d0) = make_uint2(pack_bf2(dv[db][4 * g], dv[db][4 * g + 1]), pack_bf2(dv[db][4 * g + 2], dv[db][4 * g + 3]));
                }
        } else {
            float* DK = (float*)p.dk + eo;
            float* DV = (float*)p.dv + eo;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = db * 32 + g * 8 + lh * 4;
                    *(float4*)(DK + d0) = make_float4(dk[db][4 * g], dk[db][4 * g + 1], dk[db][4 * g + 2], dk[db][4 * g + 3]);
                    *(float4*)(DV + d0) = make_float4(dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------- dQ
// two workgroups per CU (<= 256 registers per lane): one wave per SIMD leaves every LDS-read -> MFMA -> exp2 chain
// exposed; a second resident workgroup fills those gaps
__global__ __launch_bounds__(256, 2)
void attn_bwd_dq_kernel(const omh_attn_bwd_args p, const int q_blocks) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 8192];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + 8192;
    unsigned char* KTs = smem + 16384;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wid = blockIdx.x;
    const int qb = wid % q_blocks, bh = wid / q_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);

    const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
    const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
    const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
    const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
    const uint16_t* KT = (const uint16_t*)p.kt + (int64_t)b * p.kt_bs + (int64_t)head * D * p.ldk;

    const int q_row = qb * 128 + wave * 32 + li;
    const bool q_ok = q_row < p.Lq;
    bf16x8 qf[8], dof[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        qf[kk] = __builtin_bit_cast(bf16x8, ld16(Q + (int64_t)q_row * p.q_rs + kk * 16 + lh * 8, q_ok));
        dof[kk] = __builtin_bit_cast(bf16x8, ld16(DO + (int64_t)q_row * p.o_rs + kk * 16 + lh * 8, q_ok));
    }
    float l2 = INFINITY;
    const int64_t row_i = ((int64_t)b * p.H + head) * p.Lq + q_row;
    if (q_ok) {
        const float l = p.lse[row_i];
        l2 = (l > -INFINITY) ? l * LOG2E : INFINITY;
    }
    const float sc = p.q_prescaled ? 1.0f : p.scale * LOG2E;      // (dq stays the gradient of the UNSCALED q: x scale)
    const int krow_l = swap_bits23(li);

    // ---- pass 1: delta_i = sum_j P_ij dP_ij (this lane's query; its two half-lanes hold different keys)
    float del = 0.f;
    // K / V / K^T tiles are prefetched into registers one tile ahead (unconditional loads from a clamped row; rows
    // past Lk are zeroed on the way to LDS), see the dK/dV kernel
    uint4 gk0, gk1, gv0, gv1, gkt0 = make_uint4(0u, 0u, 0u, 0u), gkt1 = gkt0;
#define BWD_Q_PREFETCH1(K0, WITH_KT, J)                                                                       \
    {                                                                                                         \
        const int c = tid + 256 * J;                                                                          \
        const int row = min((K0) + (c >> 4), p.Lk - 1), slot = c & 15;                                        \
        gk##J = *(const uint4*)(K + (int64_t)row * p.k_rs + slot * 8);                                        \
        gv##J = *(const uint4*)(V + (int64_t)row * p.k_rs + slot * 8);                                        \
        if (WITH_KT) gkt##J = *(const uint4*)(KT + (int64_t)(c >> 2) * p.ldk + (K0) + (c & 3) * 8);           \
    }
#define BWD_Q_PREFETCH(K0, WITH_KT) { BWD_Q_PREFETCH1(K0, WITH_KT, 0) BWD_Q_PREFETCH1(K0, WITH_KT, 1) }
#define BWD_Q_TO_LDS1(K0, WITH_KT, J)                                                                         \
    {                                                                                                         \
        const int c = tid + 256 * J;                                                                          \
        const bool in = (K0) + (c >> 4) < p.Lk;                                                               \
        *(uint4*)(Ks + r_addr(c >> 4, c & 15)) = keep_if(in, gk##J);                                          \
        *(uint4*)(Vs + r_addr(c >> 4, c & 15)) = keep_if(in, gv##J);                                          \
        if (WITH_KT) *(uint4*)(KTs + t_addr(c >> 2, c & 3)) = gkt##J;                                         \
    }
#define BWD_Q_TO_LDS(K0, WITH_KT) { BWD_Q_TO_LDS1(K0, WITH_KT, 0) BWD_Q_TO_LDS1(K0, WITH_KT, 1) }
    if (klen > 0) BWD_Q_PREFETCH(0, false)
    for (int k0 = 0; k0 < klen; k0 += 32) {
        __syncthreads();
        BWD_Q_TO_LDS(k0, false)
        __syncthreads();
        if (k0 + 32 < klen) BWD_Q_PREFETCH(k0 + 32, false)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 ka = *(const bf16x8*)(Ks + r_addr(krow_l, 2 * kk + lh));
            const bf16x8 va = *(const bf16x8*)(Vs + r_addr(krow_l, 2 * kk + lh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[kk], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool okr = (k0 + ((r >> 3) << 4) + lh * 8 + (r & 7)) < klen;
            const float pr = okr ? __builtin_amdgcn_exp2f(fmaf(s[r], sc, -l2)) : 0.f;
            del = fmaf(pr, dp[r], del);
        }
    }
    del += __shfl_xor(del, 32, 64);
    if (q_ok && lh == 0) p.delta[row_i] = del;
    float lv[16], dl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { lv[r] = l2; dl[r] = del; }

    f32x16 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    // ---- pass 2: dQ
    if (klen > 0) BWD_Q_PREFETCH(0, true)
    for (int k0 = 0; k0 < klen; k0 += 32) {
        __syncthreads();
        BWD_Q_TO_LDS(k0, true)
        __syncthreads();
        if (k0 + 32 < klen) BWD_Q_PREFETCH(k0 + 32, true)
        // S^T = K Q^T and dP^T = V dO^T as [key][query], lane = query; register r <-> key k0 + 16(r>>3) + 8 lh + (r&7)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 ka = *(const bf16x8*)(Ks + r_addr(krow_l, 2 * kk + lh));
            const bf16x8 va = *(const bf16x8*)(Vs + r_addr(krow_l, 2 * kk + lh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[kk], dp, 0, 0, 0);
        }
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ok[r] = (k0 + ((r >> 3) << 4) + lh * 8 + (r & 7)) < klen;
        bf16x8 pf[2], dsf[2];
        p_and_ds(s, dp, lv, dl, ok, sc, p.scale, pf, dsf);
        // dQ^T += K^T dS^T   ([d][query], lane = query)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 tk = *(const bf16x8*)(KTs + t_addr(db * 32 + li, 2 * a + lh));
                dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk, dsf[a], dq[db], 0, 0, 0);
            }
    }
    if (q_ok) {
        const int64_t eo = (int64_t)b * p.dq_bs + (int64_t)q_row * p.dq_rs + head * D;
        if (p.out_bf16) {
            uint16_t* DQ = (uint16_t*)p.dq + eo;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(uint2*)(DQ + db * 32 + g * 8 + lh * 4) =
                        make_uint2(pack_bf2(dq[db][4 * g], dq[db][4 * g + 1]), pack_bf2(dq[db][4 * g + 2], dq[db][4 * g + 3]));
        } else {
            float* DQ = (float*)p.dq + eo;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(DQ + db * 32 + g * 8 + lh * 4) =
                        make_float4(dq[db][4 * g], dq[db][4 * g + 1], dq[db][4 * g + 2], dq[db][4 * g + 3]);
        }
    }
}

}  // namespace

// attention_bwd2.hip: the round-3 kernel pair (needs the forward's fp32 output for delta)
int omh_launch_attn_bwd2(const omh_attn_bwd_args& a, hipStream_t s);

extern "C" int omh_flash_attn_bwd_d128(const omh_attn_bwd_args* args, omh_stream_t stream) {
    if (!args) return OMH_E_BADARG;
    const omh_attn_bwd_args& a = *args;
    if (!a.q || !a.k || !a.v || !a.dout || !a.lse || !a.delta || !a.dq || !a.dk || !a.dv) return OMH_E_BADARG;
    if (a.o32) {
        if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return OMH_E_BADARG;
        if ((a.q_rs & 7) || (a.k_rs & 7) || (a.o_rs & 7) || (a.q_bs & 7) || (a.k_bs & 7) || (a.o_bs & 7) || (a.dq_rs & 3) ||
            (a.dk_rs & 3) || (a.dq_bs & 3) || (a.dk_bs & 3))
            return OMH_E_ALIGN;
        if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.dout | (uintptr_t)a.dq | (uintptr_t)a.dk |
             (uintptr_t)a.dv) & 15)
            return OMH_E_ALIGN;
        omh_clear_status();
        const int rc = omh_launch_attn_bwd2(a, (hipStream_t)stream);
        return rc ? rc : omh_launch_status();
    }
    if (a.phase != 0) return OMH_E_BADARG;                           // phases exist for the round-3 kernels only
    if (!a.qt || !a.dot || !a.kt) return OMH_E_BADARG;
    if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return OMH_E_BADARG;
    if ((a.q_rs & 7) || (a.k_rs & 7) || (a.o_rs & 7) || (a.q_bs & 7) || (a.k_bs & 7) || (a.o_bs & 7) || (a.ldq & 7) ||
        (a.ldk & 7) || (a.qt_bs & 7) || (a.kt_bs & 7) || (a.dq_rs & 3) || (a.dk_rs & 3) || (a.dq_bs & 3) || (a.dk_bs & 3))
        return OMH_E_ALIGN;
    const uintptr_t ptrs = (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.dout |
                           (uintptr_t)a.qt | (uintptr_t)a.dot | (uintptr_t)a.kt | (uintptr_t)a.dq | (uintptr_t)a.dk |
                           (uintptr_t)a.dv;
    if (ptrs & 15) return OMH_E_ALIGN;
    // the transposed copies are read in whole 32-position tiles: they must be padded (with zeros) that far
    if (a.ldq < ((a.Lq + 31) / 32) * 32 || a.ldk < ((a.Lk + 31) / 32) * 32) return OMH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    omh_clear_status();
    const int k_blocks = (a.Lk + 127) / 128, q_blocks = (a.Lq + 127) / 128;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(q_blocks * a.H * a.B), dim3(256), 0, s, a, q_blocks);      // writes delta
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3(k_blocks * a.H * a.B), dim3(256), 0, s, a, k_blocks);   // reads it
    return omh_launch_status();
}
