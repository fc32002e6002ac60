// Flash-attention backward, round 3 (head_dim 128, bf16 operands) for gfx950 — the kernels omh_flash_attn_bwd_d128
// runs when the caller hands over the forward's fp32 output (omh_attn_bwd_args.o32).
//
// Same mathematics and the same two launches as attention_bwd.hip (which documents the orientation trick: each
// product is computed so that the NEXT product's contraction index sits in consecutive registers of one lane; no
// atomics, bit-repeatable):
//     P  = exp2(q'.k - lse)      dP = dO V^T      dS = P (dP - delta)
//     dV = P^T dO                dQ = scale dS K  dK = scale dS^T Q
// What changed, measured on the training step of BASELINE config 3 (4 clips, S = 1560: 574 us per self-attention
// backward, 230 us per cross-attention backward, 22 % of the step — 260 TFLOP/s on a path whose forward runs at 1 000):
//   * delta_i = sum_j P_ij dP_ij = sum_d dO_id O_id.  Round 2 took it from a first pass over all keys (2 of the 9
//     matrix products of a backward) because with the forward's bf16 O the row sums of dS vanish only to 2^-9.  The
//     forward now also writes O in fp32 (omh_attn_args.o32): the dQ kernel's prologue takes the dot product of the dO
//     row it already holds with that — one pass over the keys.
//   * 64-position tiles instead of 32 and a double-buffered LDS stage filled by LDS-DMA (buffer_load ... lds,
//     source-side swizzle, rows past the end arrive as zeros): ONE barrier per 64 positions instead of two per 32,
//     no staging registers, no ds_write pass.
//   * no transposed copies.  The operands whose contraction index is the sequence position (K^T for dQ; Q^T, dO^T
//     for dK, dV) were read from [d][position] tiles of transposed global copies made by three omh_transpose_bf16
//     launches per call.  They are now gathered from the SAME row-major [position][d] LDS tiles the score products
//     read, with ds_read_b64_tr_b16 (gemm_tn.hip documents the instruction): half the staged bytes, 3 launches less.
//     The 16-byte slot index of a tile row is XORed with ((row & 3) << 2) | ((row >> 2) & 3): rows 0..15 get 16
//     different values (conflict-free ds_read_b128 of 16 rows at one column) and 4 consecutive rows differ in bits
//     2..3 (conflict-free transposing reads of 4 rows x 64 bytes).
#include "omh_common.h"

namespace {

constexpr int D = 128;
constexpr int TB = 64;                               // positions per tile
constexpr int TILE_BYTES = TB * D * 2;               // 16 KiB
constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_bf4_ptr;
typedef __attribute__((address_space(3))) unsigned char* lds_u8_ptr;
typedef __attribute__((address_space(3))) void* lds_vptr;

__device__ __forceinline__ int swap_bits23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ uint32_t swz(int row) { return (uint32_t)(((row & 3) << 2) | ((row >> 2) & 3)); }

__device__ __forceinline__ uint4 ld16(const uint16_t* p, bool ok) {
    return ok ? *(const uint4*)p : make_uint4(0u, 0u, 0u, 0u);
}

// P and dS of one 32x32 block (see attention_bwd.hip)
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

// Per-lane constants of the two fragment read patterns on a [64][128] tile (byte offsets; + (32 hb) * 256 for the
// second 32 rows, + (16 a) * 256 for the second 16-row contraction group of a transposing read)
struct FragAddr {
    uint32_t row[8];      // ds_read_b128: tile row swap23(li), d = 16 kk + 8 lh .. +7
    uint32_t tr[4];       // transposing reads: d rows 32 db + (lane & 31), positions 8 lh + 0..7 (second read: + 4 rows)
};
__device__ __forceinline__ FragAddr frag_addr(int lane) {
    FragAddr f;
    const int li = lane & 31, lh = lane >> 5;
    const int r0 = swap_bits23(li);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) f.row[kk] = (uint32_t)(r0 * 256 + (((2 * kk + lh) ^ swz(r0)) << 4));
    const int gq = lane >> 4, i15 = lane & 15, fe = i15 >> 2, fq = i15 & 3;
    const int rlo = 8 * (gq >> 1) + fe;                              // (+4 for the second read: swz changes by 1 in bits 0..1)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const int slot = 4 * db + 2 * (gq & 1) + (fq >> 1);
        f.tr[db] = (uint32_t)(rlo * 256 + ((slot ^ swz(rlo)) << 4) + (fq & 1) * 8);
    }
    return f;
}
// the second transposing read sits 4 rows further: its swizzle differs from the first one's in bit 0 of (row >> 2),
// i.e. slot ^= 1 — tr_frag()'s "+ 4 * 256" alone would miss that, so the second address is derived here
__device__ __forceinline__ uint32_t tr_second(uint32_t first) { return (first + 4 * 256) ^ 16u; }

// one 32x32x16 MFMA operand whose 16-wide contraction index is the tile ROW (position): two transposing reads
__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* tile, uint32_t addr) {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)((lds_u8_ptr)tile + addr));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)((lds_u8_ptr)tile + tr_second(addr)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// LDS-DMA of one [64][128] bf16 tile: chunk c = tid + 256 j lands at byte 16 c (row c >> 4, physical slot c & 15)
// and is fetched from logical slot (c & 15) ^ swz(row) of source row `first_row + row`
struct TileSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[4];
    uint32_t tile_bytes;            // 64 rows of the source
};
__device__ __forceinline__ TileSrc tile_src(const uint16_t* base, int64_t rows, int64_t rs, int tid) {
    TileSrc t;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((rows - 1) * rs + D) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        const int row = c >> 4;
        t.voff[j] = (uint32_t)((row * rs + (((c & 15) ^ swz(row)) << 3)) * 2);
    }
    t.tile_bytes = (uint32_t)(TB * rs * 2);
    return t;
}
__device__ __forceinline__ void tile_dma(const TileSrc& t, int tile, unsigned char* dst, int wave_lds) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rsrc, (lds_vptr)(dst + wave_lds + j * 4096), 16,
                                                 t.voff[j] + (uint32_t)tile * t.tile_bytes, 0, 0, 0);
}

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---------------------------------------------------------------------------------------------- delta alone (phase 1)
// delta[b][h][q] = sum_d dO[q][d] o32[q][d] with the dQ kernel's arithmetic (two chains per (row, head): the sequential
// fma chain over the 8 runs of 8 channels d = 16 kk + 8 lh .., the two halves added) — the same bits as phase 0.
// A block takes 32 consecutive (row, head) pairs: their 8 KiB of dO and 16 KiB of o32 are fetched with coalesced
// 16-byte loads into LDS (the chains' own access pattern is 16 bytes every 32: 24 us per call when read directly),
// then 64 threads run the chains.
__global__ __launch_bounds__(256)
void attn_bwd2_delta_kernel(const omh_attn_bwd_args p) {
    constexpr int PD = 256 + 16, PO = 512 + 16;                     // row pitches: 16-byte reads of 32 rows spread over the banks
    __shared__ __attribute__((aligned(16))) unsigned char sdo[32 * PD];
    __shared__ __attribute__((aligned(16))) unsigned char so[32 * PO];
    const int tid = threadIdx.x;
    const int64_t pair0 = (int64_t)blockIdx.x * 32;
    const int64_t npairs = (int64_t)p.B * p.Lq * p.H;
    auto locate = [&](int64_t pair, int& b, int& q, int& head) {
        head = (int)(pair % p.H);
        const int64_t row = pair / p.H;
        b = (int)(row / p.Lq);
        q = (int)(row - (int64_t)b * p.Lq);
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                   // dO: 32 pairs x 16 chunks of 16 bytes
        const int c = tid + 256 * j, pr = c >> 4, ch = c & 15;
        if (pair0 + pr < npairs) {
            int b, q, head;
            locate(pair0 + pr, b, q, head);
            *(uint4*)(sdo + pr * PD + ch * 16) =
                *(const uint4*)((const uint16_t*)p.dout + (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + head * D + ch * 8);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                   // o32: 32 pairs x 32 chunks of 16 bytes
        const int c = tid + 256 * j, pr = c >> 5, ch = c & 31;
        if (pair0 + pr < npairs) {
            int b, q, head;
            locate(pair0 + pr, b, q, head);
            *(float4*)(so + pr * PO + ch * 16) = *(const float4*)(p.o32 + (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + head * D + ch * 4);
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    const int pr = tid >> 1, lh = tid & 1;
    const bool ok = pair0 + pr < npairs;
    float del = 0.f;
    if (ok) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const uint4 dw = *(const uint4*)(sdo + pr * PD + (2 * kk + lh) * 16);
            const float4 o0 = *(const float4*)(so + pr * PO + (2 * kk + lh) * 32);
            const float4 o1 = *(const float4*)(so + pr * PO + (2 * kk + lh) * 32 + 16);
            del = fmaf(bf_lo(dw.x), o0.x, del); del = fmaf(bf_hi(dw.x), o0.y, del);
            del = fmaf(bf_lo(dw.y), o0.z, del); del = fmaf(bf_hi(dw.y), o0.w, del);
            del = fmaf(bf_lo(dw.z), o1.x, del); del = fmaf(bf_hi(dw.z), o1.y, del);
            del = fmaf(bf_lo(dw.w), o1.z, del); del = fmaf(bf_hi(dw.w), o1.w, del);
        }
    }
    del += __shfl_xor(del, 1, 64);
    if (ok && lh == 0) {
        int b, q, head;
        locate(pair0 + pr, b, q, head);
        p.delta[((int64_t)b * p.H + head) * p.Lq + q] = del;
    }
}

// ---------------------------------------------------------------------------------------------- dQ (+ delta)
__global__ __launch_bounds__(256, 2)
void attn_bwd2_dq_kernel(const omh_attn_bwd_args p, const int q_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [2 stages][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wid = blockIdx.x;
    const int qb = wid % q_blocks, bh = wid / q_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles = (klen + TB - 1) / TB;

    const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
    const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
    const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
    const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
    const float* O32 = p.o32 + (int64_t)b * p.o_bs + head * D;

    const int q_row = qb * 128 + wave * 32 + li;
    const bool q_ok = q_row < p.Lq;
    bf16x8 qf[8], dof[8];
    float del = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        qf[kk] = __builtin_bit_cast(bf16x8, ld16(Q + (int64_t)q_row * p.q_rs + kk * 16 + lh * 8, q_ok));
        const uint4 dw = ld16(DO + (int64_t)q_row * p.o_rs + kk * 16 + lh * 8, q_ok);
        dof[kk] = __builtin_bit_cast(bf16x8, dw);
        if (q_ok && p.phase == 0) {                                  // delta = sum_d dO * O (fp32 O of the forward)
            const float4 o0 = *(const float4*)(O32 + (int64_t)q_row * p.o_rs + kk * 16 + lh * 8);
            const float4 o1 = *(const float4*)(O32 + (int64_t)q_row * p.o_rs + kk * 16 + lh * 8 + 4);
            del = fmaf(bf_lo(dw.x), o0.x, del); del = fmaf(bf_hi(dw.x), o0.y, del);
            del = fmaf(bf_lo(dw.y), o0.z, del); del = fmaf(bf_hi(dw.y), o0.w, del);
            del = fmaf(bf_lo(dw.z), o1.x, del); del = fmaf(bf_hi(dw.z), o1.y, del);
            del = fmaf(bf_lo(dw.w), o1.z, del); del = fmaf(bf_hi(dw.w), o1.w, del);
        }
    }
    del += __shfl_xor(del, 32, 64);
    float l2 = INFINITY;
    const int64_t row_i = ((int64_t)b * p.H + head) * p.Lq + q_row;
    if (q_ok) {
        const float l = p.lse[row_i];
        l2 = (l > -INFINITY) ? l * LOG2E : INFINITY;
        if (p.phase != 0) del = p.delta[row_i];                      // phase 2: a phase-1 launch computed it
        else if (lh == 0) p.delta[row_i] = del;                      // the dK/dV kernel reads it
    }
    const float sc = p.q_prescaled ? 1.0f : p.scale * LOG2E;
    float lv[16], dl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { lv[r] = l2; dl[r] = del; }

    const FragAddr fa = frag_addr(lane);
    const TileSrc ks = tile_src(K, p.Lk, p.k_rs, tid), vs = tile_src(V, p.Lk, p.k_rs, tid);
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;

    f32x16 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    if (n_tiles > 0) {
        tile_dma(ks, 0, smem, wave_lds);
        tile_dma(vs, 0, smem + TILE_BYTES, wave_lds);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < n_tiles; ++t) {
        const unsigned char* kt = smem + (t & 1) * 2 * TILE_BYTES;
        const unsigned char* vt = kt + TILE_BYTES;
        if (t + 1 < n_tiles) {                                       // the other stage was last read before the barrier
            unsigned char* nk = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
            tile_dma(ks, t + 1, nk, wave_lds);
            tile_dma(vs, t + 1, nk + TILE_BYTES, wave_lds);
        }
        const int k0 = t * TB;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            // S^T = K Q^T and dP^T = V dO^T as [key][query], lane = query; register r <-> key k0 + 32hb + 16(r>>3) + 8lh + (r&7)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8 ka = *(const bf16x8*)(kt + fa.row[kk] + hb * 32 * 256);
                const bf16x8 va = *(const bf16x8*)(vt + fa.row[kk] + hb * 32 * 256);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[kk], dp, 0, 0, 0);
            }
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ok[r] = (k0 + 32 * hb + ((r >> 3) << 4) + lh * 8 + (r & 7)) < klen;
            bf16x8 pf[2], dsf[2];
            p_and_ds(s, dp, lv, dl, ok, sc, p.scale, pf, dsf);
            // dQ^T += K^T dS^T   ([d][query], lane = query): K^T gathered from the row-major K tile
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x8 tk = tr_frag2(kt, fa.tr[db] + (32 * hb + 16 * a) * 256);
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk, dsf[a], dq[db], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// ---------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256)
void attn_bwd2_dkdv_kernel(const omh_attn_bwd_args p, const int k_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [2 stages][Q tile | dO tile] + lse/delta
    float* stat = (float*)(smem + 4 * TILE_BYTES);                              // [2 stages][lse 64 | delta 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wid = blockIdx.x;
    const int kb = wid % k_blocks, bh = wid / k_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles = (p.Lq + TB - 1) / TB;

    const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
    const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
    const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
    const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
    const float* LSE = p.lse + ((int64_t)b * p.H + head) * p.Lq;
    const float* DEL = p.delta + ((int64_t)b * p.H + head) * p.Lq;

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
    const float sc = p.q_prescaled ? 1.0f : p.scale * LOG2E;
    const float ds_scale = p.q_prescaled ? (1.0f / LOG2E) : p.scale;     // dK = dS^T q' / log2(e) on a pre-scaled q

    const FragAddr fa = frag_addr(lane);
    const TileSrc qs = tile_src(Q, p.Lq, p.q_rs, tid), dos = tile_src(DO, p.Lq, p.o_rs, tid);
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;

    f32x16 dv[4], dk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[i][r] = 0.f; dk[i][r] = 0.f; }

    // lse / delta of a tile's 64 queries: threads 0..63 fetch them one tile ahead and park them in LDS
    auto stat_load = [&](int tile, float& l, float& dd) {
        const int q = tile * TB + tid;
        const bool in = q < p.Lq;
        const float lraw = in ? LSE[q] : 0.f;
        l = (in && lraw > -INFINITY) ? lraw * LOG2E : INFINITY;           // no keys / past the end: P = 0
        dd = in ? DEL[q] : 0.f;
    };
    float gl = 0.f, gd = 0.f;
    tile_dma(qs, 0, smem, wave_lds);
    tile_dma(dos, 0, smem + TILE_BYTES, wave_lds);
    if (tid < 64) { stat_load(0, gl, gd); stat[tid] = gl; stat[64 + tid] = gd; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < n_tiles; ++t) {
        const unsigned char* qt = smem + (t & 1) * 2 * TILE_BYTES;
        const unsigned char* dot = qt + TILE_BYTES;
        const float* st = stat + (t & 1) * 128;
        const bool more = t + 1 < n_tiles;
        if (more) {
            unsigned char* nq = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
            tile_dma(qs, t + 1, nq, wave_lds);
            tile_dma(dos, t + 1, nq + TILE_BYTES, wave_lds);
            if (tid < 64) stat_load(t + 1, gl, gd);
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            // S = Q K^T and dP = dO V^T as [query][key], lane = key; register r <-> query 64t + 32hb + 16(r>>3) + 8lh + (r&7)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8 qa = *(const bf16x8*)(qt + fa.row[kk] + hb * 32 * 256);
                const bf16x8 da = *(const bf16x8*)(dot + fa.row[kk] + hb * 32 * 256);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[kk], dp, 0, 0, 0);
            }
            float lv[16], dl[16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    lv[8 * a + e] = st[32 * hb + 16 * a + 8 * lh + e];
                    dl[8 * a + e] = st[64 + 32 * hb + 16 * a + 8 * lh + e];
                }
            bf16x8 pf[2], dsf[2];
            p_and_ds(s, dp, lv, dl, ok, sc, ds_scale, pf, dsf);
            // dV^T += dO^T P ,  dK^T += Q^T dS   ([d][key], lane = key): dO^T, Q^T gathered from the row-major tiles
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const uint32_t ad = fa.tr[db] + (32 * hb + 16 * a) * 256;
                    const bf16x8 ta = tr_frag2(dot, ad);
                    const bf16x8 tq = tr_frag2(qt, ad);
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta, pf[a], dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq, dsf[a], dk[db], 0, 0, 0);
                }
        }
        if (more && tid < 64) {                                      // the other stage's statistics: last read a tile ago
            float* nst = stat + ((t + 1) & 1) * 128;
            nst[tid] = gl;
            nst[64 + tid] = gd;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

}  // namespace

// called by omh_flash_attn_bwd_d128 (attention_bwd.hip) when args->o32 is set; arguments already validated there
int omh_launch_attn_bwd2(const omh_attn_bwd_args& a, hipStream_t s) {
    // 32-bit buffer offsets inside one (batch, head) slice
    if ((int64_t)a.Lq * a.q_rs * 2 >= 0x7fffffffLL || (int64_t)a.Lk * a.k_rs * 2 >= 0x7fffffffLL ||
        (int64_t)a.Lq * a.o_rs * 2 >= 0x7fffffffLL)
        return OMH_E_SHAPE;
    if (((uintptr_t)a.o32 & 15) || (a.o_rs & 3) || (a.o_bs & 3)) return OMH_E_ALIGN;
    constexpr int LDS_DQ = 4 * TILE_BYTES, LDS_KV = 4 * TILE_BYTES + 2 * 128 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
        attr_set = true;
    }
    const int k_blocks = (a.Lk + 127) / 128, q_blocks = (a.Lq + 127) / 128;
    if (a.phase < 0 || a.phase > 3) return OMH_E_BADARG;
    if (a.phase == 1) {
        const int64_t pairs = (int64_t)a.B * a.Lq * a.H;
        hipLaunchKernelGGL(attn_bwd2_delta_kernel, dim3((unsigned)((pairs + 31) / 32)), dim3(256), 0, s, a);
        return 0;
    }
    if (a.phase == 0 || a.phase == 2)
        hipLaunchKernelGGL(attn_bwd2_dq_kernel, dim3(q_blocks * a.H * a.B), dim3(256), LDS_DQ, s, a, q_blocks);    // phase 0: writes delta
    if (a.phase == 0 || a.phase == 3)
        hipLaunchKernelGGL(attn_bwd2_dkdv_kernel, dim3(k_blocks * a.H * a.B), dim3(256), LDS_KV, s, a, k_blocks);  // reads it
    return 0;
}
