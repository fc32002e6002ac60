// Flash attention forward for gfx950: head_dim 128, bf16 in/out, fp32
// accumulate, non-causal, per-batch key-length mask.
//
// Replaces flash_attn.flash_attn_varlen_func as called by the reference at
// seaweed_apt/wan/modules/attention.py:96-127 (self-attention model.py:151-156,
// cross-attention model.py:181,221-223).
//
// Formulation (everything "transposed" so softmax statistics are lane-local):
//   S^T = K Q^T     MFMA A = K rows (from LDS), B = Q rows (registers)
//   O^T = V^T P^T   MFMA A = V^T rows (from LDS), B = P (registers, bf16)
// With v_mfma_f32_32x32x16_bf16 the C/D layout gives lane l = (q = l&31,
// half h = l>>5) the scores of ONE query against 16 keys per 32-key block, so
// row max / row sum are 16-value in-register reductions plus one exchange
// with lane l^32, and O^T leaves each lane with its own query's outputs: the
// online-softmax rescale is a per-lane scalar.  K rows are fed to the MFMA
// with key bits 2 and 3 swapped, which makes a lane's 8 consecutive score
// registers correspond to 8 CONSECUTIVE keys — exactly the B-operand layout
// the P.V MFMA wants — so P never moves between lanes.  V is consumed
// transposed ([d][key], produced that way by the V-projection GEMM).
//
// Workgroup = 4 waves x 32 query rows = 128 rows of one (batch, head);
// KV tile = 64 keys; K tile [64][128] and V^T tile [128][64] bf16 staged
// HBM -> registers -> LDS (XOR-swizzled 16-byte slots, conflict-free
// ds_read_b128), double buffered, one barrier per tile, next tile's global
// loads issued before the current tile's MFMAs.
#include "omh_common.h"
#include <stdlib.h>

namespace {

constexpr int D = 128;
constexpr int QB = 128;   // query rows per workgroup
constexpr int KB = 64;    // keys per tile
constexpr int KT_BYTES = KB * D * 2;   // 16 KiB
constexpr int VT_BYTES = D * KB * 2;   // 16 KiB

// combine a value with the one held by lane l^32 (the other half of the same query row):
// one v_permlane32_swap instead of a ds_bpermute round trip through the LDS pipe
__device__ __forceinline__ void xhalf(float x, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

// max of three without the IEEE canonicalisation fmaxf() drags in: hipcc emits `v_max_f32 x, x, x` (sNaN
// quieting) in front of every fmaxf operand it cannot prove canonical — MFMA results are such — which made the
// row max 58 VALU instructions per 64-key tile instead of 16.  Scores are finite or -inf here.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ int swap_bits23(int i) {
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}
// K tile: [64 keys][128 d] bf16, 256-byte rows = 16 slots, slot ^= row & 15
__device__ __forceinline__ uint32_t k_addr(int row, int slot) {
    return (uint32_t)(row * 256 + ((slot ^ (row & 15)) << 4));
}
// V^T tile: [128 d][64 keys] bf16, 128-byte rows = 8 slots, slot ^= (row>>1)&7
__device__ __forceinline__ uint32_t v_addr(int row, int slot) {
    return (uint32_t)(row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
}

// Split-KV workers of the last round (omh_tail_split_plan, OMH_ATTN_ALLOW_SPLIT): ids >= n_regular.  Worker w takes tile
// n_regular + w / splits and key tiles [s * per, (s + 1) * per) of it, s = w % splits, and writes its NORMALISED fp32
// result + natural-log lse into slab (tail tile, s); attn_split_combine_kernel weights the slabs.
struct AttnSplit {
    int n_regular, n_tail, splits;
    float* ws_o;          // [n_tail][splits][128][128] fp32
    float* ws_lse;        // [n_tail][splits][128]      fp32 (-inf: no keys in the worker's range)
};

// WIN: the reference's flash_attention(causal=..., window_size=(left, right)) (attention.py:24-60,96-127 -> flash-attn's
// bottom-right aligned band): query i of a sample with qlen queries and klen keys sees key j iff
//   i + (klen - qlen) - left <= j <= i + (klen - qlen) + right     (a side < 0: unbounded; causal: right = 0);
// a row with no key in its band is written as zero.  The band limits the workgroup's key-tile range and masks the
// tiles on its edges; a separate instantiation, so the unlimited kernel keeps its instruction stream.
template <bool WIN>
__global__ __launch_bounds__(256, 2)
void flash_attn_fwd_d128_kernel(const omh_attn_args p, const int q_tiles, const AttnSplit wk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (KT_BYTES + VT_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // work id -> (batch*head, q tile); consecutive ids (same head) share an XCD
    const bool worker = (int)blockIdx.x >= wk.n_regular;
    int wid, split = 0;
    if (worker) {
        const int w = blockIdx.x - wk.n_regular;
        wid = wk.n_regular + w / wk.splits;
        split = w % wk.splits;
    } else {
        wid = xcd_remap(blockIdx.x, wk.n_regular);
    }
    const int bh = wid / q_tiles, qt = wid % q_tiles;
    const int b = bh / p.H, head = bh % p.H;

    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles_all = (klen + KB - 1) / KB;
    int t_first = 0, n_tiles = n_tiles_all;
    int band_shift = 0;                                   // WIN: key index of the band's centre for query row 0
    if constexpr (WIN) {
        int qlen = p.q_lens ? p.q_lens[b] : p.Lq;
        qlen = min(max(qlen, 0), p.Lq);
        band_shift = klen - qlen;
        const int q0 = qt * QB, q1 = min(q0 + QB, qlen) - 1;                  // live query rows of this workgroup
        const int lo = p.window_left < 0 ? 0 : max(0, q0 + band_shift - p.window_left);
        const int hi = p.window_right < 0 ? klen - 1 : min(klen - 1, q1 + band_shift + p.window_right);
        if (q1 < q0 || hi < lo) {
            n_tiles = 0;
        } else {
            t_first = lo / KB;
            n_tiles = hi / KB - t_first + 1;
        }
    }
    if (worker) {
        const int per = (n_tiles_all + wk.splits - 1) / wk.splits;
        t_first = min(split * per, n_tiles_all);
        n_tiles = min(t_first + per, n_tiles_all) - t_first;
    }

    const __bf16* __restrict__ Q = (const __bf16*)p.q + (int64_t)b * p.q_bs + head * D;
    const __bf16* __restrict__ K = (const __bf16*)p.k + (int64_t)b * p.k_bs + head * D;
    const __bf16* __restrict__ VT = (const __bf16*)p.vt + (int64_t)b * p.vt_bs + (int64_t)head * D * p.ldv;

    // ---- Q fragments (MFMA B operand): lane (q = li, h) holds Q[q][16kk + 8h .. +7]
    const int q_row = qt * QB + wave * 32 + li;
    const int q_ld = min(q_row, p.Lq - 1);
    int key_lo = 0, key_hi = klen - 1;                    // WIN: this lane's (query row's) band of keys
    if constexpr (WIN) {
        if (p.window_left >= 0) key_lo = max(0, q_row + band_shift - p.window_left);
        if (p.window_right >= 0) key_hi = min(klen - 1, q_row + band_shift + p.window_right);
    }
    bf16x8 qf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
        qf[kk] = *(const bf16x8*)(Q + (int64_t)q_ld * p.q_rs + kk * 16 + lh * 8);

    // ---- staging: 4 K chunks + 4 V^T chunks of 16 bytes per thread and tile, fetched with buffer
    // loads: the descriptor (SGPRs) carries base + extent, so the per-tile address is ONE v_add per
    // load and rows past the end of K read as zero in hardware (no clamping / select VALU).
    // K tile: chunk c -> row c>>4 (key), slot c&15 ; V^T tile: row c>>3 (d), slot c&7
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(
        (void*)K, 0, (int)((((int64_t)p.Lk - 1) * p.k_rs + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
        (void*)VT, 0, (int)((int64_t)D * p.ldv * 2), 0x00020000);
    uint32_t voff_k[4], voff_v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        voff_k[j] = (uint32_t)(((c >> 4) * (int)p.k_rs + (c & 15) * 8) * 2);
        voff_v[j] = (uint32_t)(((c >> 3) * p.ldv + (c & 7) * 8) * 2);
    }
    const uint32_t k_tile_bytes = (uint32_t)(KB * (int)p.k_rs * 2), v_tile_bytes = KB * 2;
    u32x4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
#define OMH_GLOAD(T)                                                                            \
    {                                                                                           \
        const uint32_t ko_ = (uint32_t)(T) * k_tile_bytes, vo_ = (uint32_t)(T) * v_tile_bytes;  \
        rk0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, voff_k[0] + ko_, 0, 0);             \
        rk1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, voff_k[1] + ko_, 0, 0);             \
        rk2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, voff_k[2] + ko_, 0, 0);             \
        rk3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, voff_k[3] + ko_, 0, 0);             \
        rv0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff_v[0] + vo_, 0, 0);             \
        rv1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff_v[1] + vo_, 0, 0);             \
        rv2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff_v[2] + vo_, 0, 0);             \
        rv3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff_v[3] + vo_, 0, 0);             \
    }
#define OMH_LSTORE1(RK, RV, J, KT, VTL)                                                         \
    {                                                                                           \
        const int c_ = tid + 256 * (J);                                                         \
        *(u32x4*)((KT) + k_addr(c_ >> 4, c_ & 15)) = RK;                                        \
        *(u32x4*)((VTL) + v_addr(c_ >> 3, c_ & 7)) = RV;                                        \
    }
#define OMH_LSTORE(BUF)                                                                         \
    {                                                                                           \
        unsigned char* kt_ = smem + (BUF) * (KT_BYTES + VT_BYTES);                              \
        unsigned char* vt_ = kt_ + KT_BYTES;                                                    \
        OMH_LSTORE1(rk0, rv0, 0, kt_, vt_) OMH_LSTORE1(rk1, rv1, 1, kt_, vt_)                   \
        OMH_LSTORE1(rk2, rv2, 2, kt_, vt_) OMH_LSTORE1(rk3, rv3, 3, kt_, vt_)                   \
    }

    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;   // scores -> log2 domain

    if (n_tiles > 0) {
        OMH_GLOAD(t_first)
        OMH_LSTORE(0)
    }
    __syncthreads();

    const int krow_l = swap_bits23(li);
    for (int tt = 0; tt < n_tiles; ++tt) {
        const int buf = tt & 1;
        const int t = t_first + tt;
        OMH_GLOAD(t_first + min(tt + 1, n_tiles - 1))   // unconditional: keeps the staging registers out of scratch
        const unsigned char* kt = smem + buf * (KT_BYTES + VT_BYTES);
        const unsigned char* vt = kt + KT_BYTES;

        // ---- S^T = K Q^T : two 32-key blocks; the two accumulators alternate so that
        //      consecutive MFMAs never wait on each other's result
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 kf0 = *(const bf16x8*)(kt + k_addr(krow_l, 2 * kk + lh));
            const bf16x8 kf1 = *(const bf16x8*)(kt + k_addr(32 + krow_l, 2 * kk + lh));
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[kk], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[kk], s[1], 0, 0, 0);
        }
        // register r of block kb  <->  key kv0 + 32kb + 16(r>>3) + 8h + (r&7)
        const int kv0 = t * KB;
        if constexpr (WIN) {                          // (key_hi <= klen - 1: the band mask covers the sequence end too)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + ((r >> 3) << 4) + lh * 8 + (r & 7);
                    if (key < key_lo || key > key_hi) s[kb][r] = -INFINITY;
                }
        } else if (__builtin_expect(kv0 + KB > klen, 0)) {   // only the last tile of a sequence
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + ((r >> 3) << 4) + lh * 8 + (r & 7);
                    if (key >= klen) s[kb][r] = -INFINITY;
                }
        }
        // ---- online softmax (log2 domain)
        // Both chains start from a COMPILER-VISIBLE max over one register of each score accumulator: hipcc's hazard
        // recognizer inserts the MFMA-write -> VALU-read wait states (19 for this MFMA) only for instructions it
        // knows, not for the inline-asm v_max3 below, and the last K·Q^T MFMA is issued one instruction earlier.
        // Without it the max was sometimes taken from scores missing their last k-slice: still a valid softmax
        // offset, but a different one from run to run, i.e. outputs that differed in the last bf16 bit.
        const float mx_seed = fmaxf(s[0][0], s[1][0]);
        float mx = mx_seed, mx2 = mx_seed;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                           // two independent v_max3 chains
            mx = vmax3(mx, s[0][r], s[1][r]);
            mx2 = vmax3(mx2, s[0][r + 1], s[1][r + 1]);
        }
        { float a_, b_; xhalf(vmax3(mx, mx2, mx2), a_, b_); mx = a_; mx2 = b_; }
        const float m_new = vmax3(m_run, mx * sc, mx2 * sc);
        // WIN: a row may have seen no key of its band yet (m_new = -inf): exponentials against 0 then, all of them 0
        const float m_use = (WIN && m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m_run - m_use);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(fmaf(s[kb][r], sc, -m_use));
                s[kb][r] = pv;
                rs += pv;
            }
        { float a_, b_; xhalf(rs, a_, b_); rs = a_ + b_; }
        l_run = l_run * alpha + rs;
        if (!__all(alpha == 1.0f)) {          // wave-uniform: after the first tiles the running max rarely moves
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }

        // ---- P -> bf16 MFMA B operands: pf[kb][a] covers keys 32kb + 16a + 8h + 0..7
        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                u32x4 cv;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cv[e] = pack_bf2(s[kb][8 * a + 2 * e], s[kb][8 * a + 2 * e + 1]);
                pf[kb][a] = __builtin_bit_cast(bf16x8, cv);
            }
        // ---- O^T += V^T P^T : four independent accumulators in rotation
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x8 vf = *(const bf16x8*)(vt + v_addr(db * 32 + li, 4 * kb + 2 * a + lh));
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][a], oacc[db], 0, 0, 0);
                }

        OMH_LSTORE(buf ^ 1)
        __syncthreads();
    }

    // ---- normalise and store: lane holds O[q][32db + 8g + 4h + 0..3]
    if (worker) {                                      // split-KV worker: fp32 slab, combined by attn_split_combine_kernel
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        const int64_t slab = (int64_t)(wid - wk.n_regular) * wk.splits + split;
        const int r = wave * 32 + li;
        float* so = wk.ws_o + (slab * QB + r) * D;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)(so + db * 32 + gq * 8 + lh * 4) =
                    make_float4(oacc[db][4 * gq] * inv, oacc[db][4 * gq + 1] * inv, oacc[db][4 * gq + 2] * inv,
                                oacc[db][4 * gq + 3] * inv);
        if (lh == 0)
            wk.ws_lse[slab * QB + r] = l_run > 0.f ? (m_run + log2f(l_run)) * 0.6931471805599453f : -INFINITY;
        return;
    }
    if (q_row < p.Lq) {
        // q_lens (ABI v10): a query row past its sample's length is a pad row of the reference's varlen batch -> zeros
        const bool live = !p.q_lens || q_row < p.q_lens[b];
        if (!live) l_run = 0.f;
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        uint16_t* O = (uint16_t*)p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + head * D;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 pk;
                pk.x = pack_bf2(oacc[db][4 * gq] * inv, oacc[db][4 * gq + 1] * inv);
                pk.y = pack_bf2(oacc[db][4 * gq + 2] * inv, oacc[db][4 * gq + 3] * inv);
                *(uint2*)(O + db * 32 + gq * 8 + lh * 4) = pk;
            }
        if (p.o32) {                                   // the same values before the rounding (training: delta of the backward)
            float* O32 = p.o32 + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_rs + head * D;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *(float4*)(O32 + db * 32 + gq * 8 + lh * 4) =
                        make_float4(oacc[db][4 * gq] * inv, oacc[db][4 * gq + 1] * inv, oacc[db][4 * gq + 2] * inv,
                                    oacc[db][4 * gq + 3] * inv);
        }
        if (p.lse && lh == 0) {
            // natural-log LSE of the scaled scores
            const float lse = l_run > 0.f ? (m_run + log2f(l_run)) * 0.6931471805599453f : -INFINITY;
            p.lse[((int64_t)b * p.H + head) * p.Lq + q_row] = lse;
        }
    }
}

// (Round 5: the 8-wave 256-row long-sequence kernel "pp" that preceded attention_w64.hip was retired — the generated
// 4 x 64 stream supersedes it on every shape it took; its history is in DESIGN_HISTORY.md 4.1 / 9.1.)
constexpr int QB2 = 256;            // query rows of a long-sequence workgroup (attention_w64.hip)

// out[row] = sum_s w_s O_s[row] / sum_s w_s,  w_s = exp(lse_s - max lse)  (flash-decoding reduction over the split-KV
// workers of a tail tile): one wave per query row, a lane per pair of channels; also the fp32 output and the lse.
__global__ __launch_bounds__(256)
void attn_split_combine_kernel(const omh_attn_args p, const int q_tiles, const AttnSplit wk) {
    const int lane = threadIdx.x & 63;
    const int rowid = blockIdx.x * 4 + (threadIdx.x >> 6);                // (tail tile, row in tile)
    const int tt = rowid / QB, r = rowid % QB;
    if (tt >= wk.n_tail) return;
    const int tile = wk.n_regular + tt;
    const int bh = tile / q_tiles, qt = tile % q_tiles;
    const int b = bh / p.H, head = bh % p.H;
    const int row = qt * QB + r;
    if (row >= p.Lq) return;
    float mx = -INFINITY;
    for (int s = 0; s < wk.splits; ++s) mx = fmaxf(mx, wk.ws_lse[((int64_t)tt * wk.splits + s) * QB + r]);
    float acc0 = 0.f, acc1 = 0.f, wsum = 0.f;
    if (mx > -INFINITY) {
        for (int s = 0; s < wk.splits; ++s) {                             // fixed order: repeatable bit for bit
            const int64_t part = (int64_t)tt * wk.splits + s;
            const float l = wk.ws_lse[part * QB + r];
            if (l == -INFINITY) continue;
            const float w = __expf(l - mx);
            const float2 v = *(const float2*)(wk.ws_o + (part * QB + r) * D + 2 * lane);
            acc0 += w * v.x;
            acc1 += w * v.y;
            wsum += w;
        }
    }
    const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
    const int64_t eo = (int64_t)b * p.o_bs + (int64_t)row * p.o_rs + head * D + 2 * lane;
    *(uint32_t*)((uint16_t*)p.o + eo) = pack_bf2(acc0 * inv, acc1 * inv);
    if (p.o32) *(float2*)(p.o32 + eo) = make_float2(acc0 * inv, acc1 * inv);
    if (p.lse && lane == 0)
        p.lse[((int64_t)b * p.H + head) * p.Lq + row] = wsum > 0.f ? mx + __logf(wsum) : -INFINITY;
}

}  // namespace

// attention_w64.hip: 4 waves x 64 query rows, asm-owned register file (long sequences)
int omh_launch_attn_w64(const omh_attn_args& a, hipStream_t stream);

static inline bool omh_attn_windowed(const omh_attn_args& a) { return a.window_left >= 0 || a.window_right >= 0; }

// Which forward kernel a call takes.  Option ATTN_KERNEL = "w64" / "base": test / benchmarking override (looked up
// per call — the tests flip it inside one process; a getenv is ~50 ns against ~3.5 us of launch).
struct AttnChoice { bool w64; };
static AttnChoice attn_choice(const omh_attn_args& a) {
    const char* force = omh_opt(OMH_OPT_ATTN_KERNEL);
    const int q_tiles2 = (a.Lq + QB2 - 1) / QB2;
    // long sequences that fill the chip with 256-row workgroups take the 4 x 64 kernel (attention_w64.hip)
    const bool big = (int64_t)q_tiles2 * a.H * a.B >= 512 && a.Lk >= 1024;
    // 32-bit buffer offsets inside one (batch, head) slice
    const bool fits32 = ((int64_t)a.Lq * a.q_rs * 2 < 0x7fffffffLL) && ((int64_t)a.Lk * a.k_rs * 2 < 0x7fffffffLL) &&
                        ((int64_t)a.Lq * a.o_rs * 2 < 0x7fffffffLL) && ((int64_t)D * a.ldv * 2 < 0x7fffffffLL);
    AttnChoice c;
    const bool short_only = a.o32 || a.q_lens || (a.flags & OMH_ATTN_SHORT_KERNEL) || omh_attn_windowed(a);   // fp32 output / q_lens / band: base kernel only
    c.w64 = short_only ? false : (force ? (force[0] == 'w' && fits32) : (big && fits32));
    return c;
}
// Split plan of the short-sequence kernel (OMH_ATTN_ALLOW_SPLIT; two workgroups per CU; >= 4 key tiles per worker)
static OmhSplitPlan base_split_plan(const omh_attn_args& a) {
    const int q_tiles = (a.Lq + QB - 1) / QB;
    const int nwg = q_tiles * a.H * a.B;
    OmhSplitPlan none = {nwg, 0, 1};
    const char* e = omh_opt(OMH_OPT_ATTN_SPLIT);                      // "0": never split (A/B timing; tests flip it in-process)
    if (!(a.flags & OMH_ATTN_ALLOW_SPLIT) || a.q_lens || omh_attn_windowed(a) || (e && e[0] == '0')) return none;
    // measured at one clip x 1560 keys: 32.1 -> 24.8 us + 12.8 us of combine (the workers' fp32 results + the bf16 / fp32 /
    // lse outputs are all HBM traffic): the forward's split only pays on long key loops — 16 key tiles per worker, and
    // only launches that do not fill the chip once.  OMH_ATTN_SPLIT=tail: 4 tiles per worker, any launch (tests, A/B).
    const bool tail = e && e[0] == 't';
    return omh_tail_split_plan(nwg, 2 * omh_cu_count(), (a.Lk + KB - 1) / KB, tail ? 4 : 16, !tail, tail ? 0.8 : 0.67);
}
int64_t omh_attn_base_workspace_bytes(const omh_attn_args& a) {
    const AttnChoice ch = attn_choice(a);
    if (ch.w64) return 0;
    const OmhSplitPlan pl = base_split_plan(a);
    return (int64_t)pl.n_tail * pl.splits * QB * (D + 1) * 4;
}
bool omh_attn_takes_w64(const omh_attn_args& a) { return attn_choice(a).w64; }

extern "C" int omh_flash_attn_fwd_d128(const omh_attn_args* args, omh_stream_t stream) {
    if (!args || !args->q || !args->k || !args->vt || !args->o) return OMH_E_BADARG;
    const omh_attn_args& a = *args;
    if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return OMH_E_BADARG;
    if ((a.q_rs & 7) || (a.k_rs & 7) || (a.o_rs & 3) || (a.ldv & 7) || (a.q_bs & 7) || (a.k_bs & 7) ||
        (a.vt_bs & 7) || (a.o_bs & 3))
        return OMH_E_ALIGN;
    if (((uintptr_t)a.q & 15) || ((uintptr_t)a.k & 15) || ((uintptr_t)a.vt & 15) || ((uintptr_t)a.o & 7))
        return OMH_E_ALIGN;
    if (a.ldv < ((a.Lk + KB - 1) / KB) * KB) return OMH_E_SHAPE;
    // 32-bit buffer offsets inside one (batch, head) slice of K / V^T (both kernels)
    if ((int64_t)a.Lk * a.k_rs * 2 >= 0x7fffffffLL || (int64_t)D * a.ldv * 2 >= 0x7fffffffLL) return OMH_E_SHAPE;
    if (a.o32 && (((uintptr_t)a.o32 & 15) || (a.o_rs & 3) || (a.o_bs & 3))) return OMH_E_ALIGN;
    const AttnChoice ch = attn_choice(a);
    const bool w64 = ch.w64;
    omh_clear_status();
    if (w64) {
        omh_launch_attn_w64(a, (hipStream_t)stream);
    } else {
        const int q_tiles = (a.Lq + QB - 1) / QB;
        OmhSplitPlan pl = base_split_plan(a);
        const int64_t need = (int64_t)pl.n_tail * pl.splits * QB * (D + 1) * 4;
        if (pl.n_tail && (!a.workspace || a.workspace_bytes < need || ((uintptr_t)a.workspace & 15))) {
            pl.n_regular += pl.n_tail; pl.n_tail = 0; pl.splits = 1;     // no workspace from the caller: one more round instead
        }
        AttnSplit wk;
        wk.n_regular = pl.n_regular; wk.n_tail = pl.n_tail; wk.splits = pl.splits;
        wk.ws_o = pl.n_tail ? (float*)a.workspace : nullptr;
        wk.ws_lse = pl.n_tail ? wk.ws_o + (int64_t)pl.n_tail * pl.splits * QB * D : nullptr;
        if (omh_attn_windowed(a))
            hipLaunchKernelGGL(flash_attn_fwd_d128_kernel<true>, dim3(pl.n_regular), dim3(256), 0, (hipStream_t)stream, a,
                               q_tiles, wk);
        else
            hipLaunchKernelGGL(flash_attn_fwd_d128_kernel<false>, dim3(pl.n_regular + pl.n_tail * pl.splits), dim3(256), 0,
                               (hipStream_t)stream, a, q_tiles, wk);
        if (pl.n_tail)
            hipLaunchKernelGGL(attn_split_combine_kernel, dim3((pl.n_tail * QB + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                               a, q_tiles, wk);
    }
    return omh_launch_status();
}
