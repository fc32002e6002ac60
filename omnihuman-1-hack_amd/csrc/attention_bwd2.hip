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
#include "attention_bwd2_asm.inc"
#include <stdlib.h>
#include <type_traits>

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

// P and dS of one 32x32 block (see attention_bwd.hip), WITHOUT the factor `scale` of dS: the callers multiply their dQ /
// dK accumulators by it once, after the last tile (round 5; it was one multiply per score).
//   PRE   q carries scale * log2(e): the scores need no factor
//   FOLD  the score accumulators started from -lse / sc and the dP accumulators from -delta (MFMA C operand of the first
//         product): P = exp2(sc s), dS = P dp — the fma and the subtraction per score are gone
//   MASK  keys key_base + 16 (r >> 3) + (r & 7) at or past `klen` get P = 0 (the dQ kernel's last key tile only)
template <bool PRE, bool FOLD, bool MASK>
__device__ __forceinline__ void p_and_ds(const f32x16& s, const f32x16& dp, float lv, float dl, int key_base, int klen,
                                         float sc, bf16x8* pf, bf16x8* dsf) {
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float x = FOLD ? (PRE ? s[r] : s[r] * sc) : (PRE ? s[r] - lv : fmaf(s[r], sc, -lv));
        p[r] = __builtin_amdgcn_exp2f(x);
        if (MASK) p[r] = (key_base + ((r >> 3) << 4) + (r & 7) < klen) ? p[r] : 0.f;
        ds[r] = p[r] * (FOLD ? dp[r] : dp[r] - dl);
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
    u32x4 rsrc;                     // raw buffer descriptor: base, stride 0, bytes, 0x00020000
    uint32_t voff[8];               // 1024 / THREADS chunks per thread (4 for 256 threads, 8 for 128)
    uint32_t tile_bytes;            // 64 rows of the source
};
template <int THREADS = 256>
__device__ __forceinline__ TileSrc tile_src(const uint16_t* base, int64_t rows, int64_t rs, int tid) {
    TileSrc t;
    const uint64_t a = (uint64_t)base;
    t.rsrc = u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)(((rows - 1) * rs + D) * 2), 0x00020000u};
#pragma unroll
    for (int j = 0; j < 1024 / THREADS; ++j) {
        const int c = tid + THREADS * j;
        const int row = c >> 4;
        t.voff[j] = (uint32_t)((row * rs + (((c & 15) ^ swz(row)) << 3)) * 2);
    }
    t.tile_bytes = (uint32_t)(TB * rs * 2);
    return t;
}
// The loads are written as asm ON PURPOSE (round 5).  Through the builtin the compiler knows that a load into LDS is
// outstanding, cannot tell which bytes it will write, and puts `s_waitcnt vmcnt(0)` in front of the next LDS read —
// the first fragment read of the very iteration that issued the prefetch.  The "double buffer" then never overlapped
// anything: each 64-position tile paid one full memory round trip (measured: 5 700 cycles per tile of the dK / dV
// kernel with the LDS reads AND the softmax removed it was still 4 500, for 2 048 cycles of MFMA work).  As asm the
// loads are invisible to the compiler's counter bookkeeping; the kernels wait for them explicitly, at the end of the
// iteration (s_waitcnt vmcnt(0) + barrier).  Hidden loads can only make a compiler-placed vmcnt(N) wait longer, never
// shorter: the counter retires in order.  `lds_dst` is the wave-uniform LDS byte address of chunk row 0 of this wave.
template <int THREADS = 256>
__device__ __forceinline__ void tile_dma(const TileSrc& t, int tile, uint32_t lds_dst) {
#pragma unroll
    for (int j = 0; j < 1024 / THREADS; ++j)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(lds_dst + (uint32_t)(j * (THREADS * 16))), "v"(t.voff[j] + (uint32_t)tile * t.tile_bytes), "s"(t.rsrc)
                     : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_u8_ptr)p; }

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---------------------------------------------------------------------------------------------- delta
// delta[b][h][q] = sum_d dO[q][d] o32[q][d]: 16 lanes per (row, head) pair, 8 channels each (one 16-byte dO load and two
// 16-byte o32 loads per lane: a wave covers 4 pairs = 1 KiB + 2 KiB of contiguous memory when the tensors are dense),
// an fma chain over the lane's 8 channels, then a fixed butterfly over the 16 lanes — one arithmetic, whatever the
// phase (round 3 had a second copy of it in the dQ kernel's prologue and a 38 us LDS-staged kernel here; 57 MB at
// 4 clips: ~12 us of traffic).
__global__ __launch_bounds__(256)
void attn_bwd2_delta_kernel(const omh_attn_bwd_args p) {
    const int tid = threadIdx.x;
    const uint32_t npairs = (uint32_t)p.B * (uint32_t)p.Lq * (uint32_t)p.H;      // < 2^31: checked by the launcher
    const uint32_t pair = blockIdx.x * 16u + (uint32_t)(tid >> 4);
    const int sub = tid & 15;
    float del = 0.f;
    int b = 0, q = 0, head = 0;
    const bool ok = pair < npairs;
    if (ok) {
        const uint32_t row = pair / (uint32_t)p.H;
        head = (int)(pair - row * (uint32_t)p.H);
        b = (int)(row / (uint32_t)p.Lq);
        q = (int)(row - (uint32_t)b * (uint32_t)p.Lq);
        const int64_t off = (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + head * D + sub * 8;
        const uint4 dw = *(const uint4*)((const uint16_t*)p.dout + off);
        const float4 o0 = *(const float4*)(p.o32 + off);
        const float4 o1 = *(const float4*)(p.o32 + off + 4);
        del = bf_lo(dw.x) * o0.x;
        del = fmaf(bf_hi(dw.x), o0.y, del);
        del = fmaf(bf_lo(dw.y), o0.z, del); del = fmaf(bf_hi(dw.y), o0.w, del);
        del = fmaf(bf_lo(dw.z), o1.x, del); del = fmaf(bf_hi(dw.z), o1.y, del);
        del = fmaf(bf_lo(dw.w), o1.z, del); del = fmaf(bf_hi(dw.w), o1.w, del);
    }
    del += __shfl_xor(del, 1, 64);
    del += __shfl_xor(del, 2, 64);
    del += __shfl_xor(del, 4, 64);
    del += __shfl_xor(del, 8, 64);
    if (ok && sub == 0) p.delta[((int64_t)b * p.H + head) * p.Lq + q] = del;
}

// ---------------------------------------------------------------------------------------------- dQ (+ delta)
// Workers of the last, partly filled round (omh_tail_split_plan; ABI v8 omh_attn_bwd_args.workspace): ids >= n_regular.
// Worker w takes workgroup-tile n_regular + w / splits and the s-th share of its inner loop (s = w % splits; key tiles
// for dQ, query tiles for dK / dV) and writes fp32 partial sums into slab (tail tile, s) of `ws`; attn_bwd2_sum_kernel
// adds the slabs in the order of s.
struct BwdSplit {
    int n_regular, n_tail, splits;
    float* ws;            // dQ: [n_tail][splits][128][128];  dK, dV: [n_tail][splits][2][128][128]  fp32
};

template <bool PRE>
__global__ __launch_bounds__(256, 2)
void attn_bwd2_dq_kernel(const omh_attn_bwd_args p, const int q_blocks, const BwdSplit wk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [2 stages][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const bool worker = (int)blockIdx.x >= wk.n_regular;
    // regular ids go through xcd_remap: the 13 query blocks of one (batch, head) then run on ONE XCD and its K / V
    // (400 KB each at S = 1560) is fetched into one 4 MiB L2 instead of eight
    const int wid = worker ? wk.n_regular + ((int)blockIdx.x - wk.n_regular) / wk.splits : xcd_remap((int)blockIdx.x, wk.n_regular);
    const int split = worker ? ((int)blockIdx.x - wk.n_regular) % wk.splits : 0;
    const int qb = wid % q_blocks, bh = wid / q_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles_all = (klen + TB - 1) / TB;
    int t_first = 0, n_tiles = n_tiles_all;
    if (worker) {
        const int per = (n_tiles_all + wk.splits - 1) / wk.splits;
        t_first = min(split * per, n_tiles_all);
        n_tiles = min(t_first + per, n_tiles_all) - t_first;
    }

    const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
    const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
    const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
    const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
    const int q_row = qb * 128 + wave * 32 + li;
    const bool q_ok = q_row < p.Lq;
    bf16x8 qf[8], dof[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        qf[kk] = __builtin_bit_cast(bf16x8, ld16(Q + (int64_t)q_row * p.q_rs + kk * 16 + lh * 8, q_ok));
        dof[kk] = __builtin_bit_cast(bf16x8, ld16(DO + (int64_t)q_row * p.o_rs + kk * 16 + lh * 8, q_ok));
    }
    float l2 = INFINITY, del = 0.f;
    const int64_t row_i = ((int64_t)b * p.H + head) * p.Lq + q_row;
    if (q_ok) {
        const float l = p.lse[row_i];
        l2 = (l > -INFINITY) ? l * LOG2E : INFINITY;
        del = p.delta[row_i];                                        // attn_bwd2_delta_kernel ran before (every phase)
    }
    const float sc = PRE ? 1.0f : p.scale * LOG2E;

    const FragAddr fa = frag_addr(lane);
    const TileSrc ks = tile_src(K, p.Lk, p.k_rs, tid), vs = tile_src(V, p.Lk, p.k_rs, tid);
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);   // stage 0, K tile, this wave's rows

    f32x16 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    if (n_tiles > 0) {
        tile_dma(ks, t_first, wave_lds);
        tile_dma(vs, t_first, wave_lds + TILE_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // one tile of 64 keys; `masked`: the tile reaches past klen (the last one at most — every other tile skips the
    // compare + select per score)
    auto tile_body = [&](const unsigned char* kt, const unsigned char* vt, int k0, auto masked) {
        constexpr bool MASK = decltype(masked)::value;
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
            bf16x8 pf[2], dsf[2];
            p_and_ds<PRE, false, MASK>(s, dp, l2, del, k0 + 32 * hb + lh * 8, klen, sc, pf, dsf);
            // dQ^T += K^T dS^T   ([d][query], lane = query): K^T gathered from the row-major K tile
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x8 tk = tr_frag2(kt, fa.tr[db] + (32 * hb + 16 * a) * 256);
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk, dsf[a], dq[db], 0, 0, 0);
                }
        }
    };
    // the tile that reaches past klen (the last one of the sequence, if any) is peeled off the loop: two bodies inside
    // one loop keep both sets of hoisted addresses live and spill
    const bool edge = n_tiles > 0 && (t_first + n_tiles) * TB > klen;
    const int n_plain = n_tiles - (edge ? 1 : 0);
    for (int t = 0; t < n_plain; ++t) {
        const unsigned char* kt = smem + (t & 1) * 2 * TILE_BYTES;
        if (t + 1 < n_tiles) {                                       // the other stage was last read before the barrier
            const uint32_t nk = wave_lds + ((t + 1) & 1) * 2 * TILE_BYTES;
            tile_dma(ks, t_first + t + 1, nk);
            tile_dma(vs, t_first + t + 1, nk + TILE_BYTES);
        }
        tile_body(kt, kt + TILE_BYTES, (t_first + t) * TB, std::false_type());
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (edge) {
        const unsigned char* kt = smem + (n_plain & 1) * 2 * TILE_BYTES;
        tile_body(kt, kt + TILE_BYTES, (t_first + n_plain) * TB, std::true_type());
    }
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] *= p.scale;             // dQ = scale dS K
    if (worker) {                                                    // partial sums over this worker's keys
        float* W = wk.ws + (((int64_t)(wid - wk.n_regular) * wk.splits + split) * 128 + wave * 32 + li) * D;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(W + db * 32 + g * 8 + lh * 4) =
                    make_float4(dq[db][4 * g], dq[db][4 * g + 1], dq[db][4 * g + 2], dq[db][4 * g + 3]);
        return;
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

// dK (x ds_scale) and dV of one key (= one lane; `row` = the key's index in its 128-key block) -> bf16 / fp32 rows, or
// the fp32 slab of a split worker; keys past klen get zeros whatever the loop left in their column
__device__ __forceinline__ void dkdv_store(const omh_attn_bwd_args& p, const BwdSplit& wk, bool worker, int wid, int split, int b,
                                           int head, int row, int lh, int key, int klen, float ds_scale, f32x16* dk, f32x16* dv) {
    const bool key_in = key < klen;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[db][r] = key_in ? dk[db][r] * ds_scale : 0.f;          // (a select: the column may hold NaN)
            dv[db][r] = key_in ? dv[db][r] : 0.f;
        }
    if (worker) {                                                    // partial sums over this worker's queries
        float* W = wk.ws + ((((int64_t)(wid - wk.n_regular) * wk.splits + split) * 2) * 128 + row) * D;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + g * 8 + lh * 4;
                *(float4*)(W + d0) = make_float4(dk[db][4 * g], dk[db][4 * g + 1], dk[db][4 * g + 2], dk[db][4 * g + 3]);
                *(float4*)(W + 128 * D + d0) = make_float4(dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]);
            }
        return;
    }
    if (key >= p.Lk) return;
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

// ---------------------------------------------------------------------------------------------- dK, dV
// WAVES x KPW = 4 x 1: 4 waves of 32 keys.  The kernel is bound by the LDS pipe (1.5 LDS instructions per MFMA: each
// of the 4 waves re-reads the whole shared Q / dO tile for its 32 keys).  The template also expresses 2 waves of 64
// keys (<2, 2>: every fragment read feeds TWO key blocks, 0.75 per MFMA; two workgroups per CU, one wave per SIMD) —
// tried in round 4 and NOT instantiated: 256 accumulator registers fill the AGPRs, the remaining live set (128 K / V
// operand registers + 64 score + 32 packed P / dS + fragments + addresses) is ~290 > 256 arch VGPRs, and hipcc spills
// 334 registers (812 bytes of scratch per lane).  It needs an asm-owned register map with P / dS overlaid on the score
// registers and lse / delta folded into the MFMA C operand.
template <int WAVES, int KPW, bool PRE>
__global__ __launch_bounds__(64 * WAVES, 1)
void attn_bwd2_dkdv_kernel(const omh_attn_bwd_args p, const int k_blocks, const BwdSplit wk) {
    constexpr int THREADS = 64 * WAVES;
    static_assert(WAVES * KPW == 4, "a workgroup covers 128 keys");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [2 stages][Q tile | dO tile] + lse/delta
    float* stat = (float*)(smem + 4 * TILE_BYTES);                              // [2 stages][lse 64 | delta 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const bool worker = (int)blockIdx.x >= wk.n_regular;
    const int wid = worker ? wk.n_regular + ((int)blockIdx.x - wk.n_regular) / wk.splits : xcd_remap((int)blockIdx.x, wk.n_regular);
    const int split = worker ? ((int)blockIdx.x - wk.n_regular) % wk.splits : 0;
    const int kb = wid % k_blocks, bh = wid / k_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles_all = (p.Lq + TB - 1) / TB;
    int t_first = 0, n_tiles = n_tiles_all;
    if (worker) {                                                    // this worker's share of the query tiles
        const int per = (n_tiles_all + wk.splits - 1) / wk.splits;
        t_first = min(split * per, n_tiles_all);
        n_tiles = min(t_first + per, n_tiles_all) - t_first;
    }

    const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
    const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
    const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
    const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
    const float* LSE = p.lse + ((int64_t)b * p.H + head) * p.Lq;
    const float* DEL = p.delta + ((int64_t)b * p.H + head) * p.Lq;

    const int key0 = kb * 128 + wave * (32 * KPW) + li;              // key of key block 0; block c: + 32 c
    bf16x8 kf[KPW][8], vf[KPW][8];
#pragma unroll
    for (int c = 0; c < KPW; ++c)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int key = key0 + 32 * c;
            const int64_t off = (int64_t)key * p.k_rs + kk * 16 + lh * 8;
            kf[c][kk] = __builtin_bit_cast(bf16x8, ld16(K + off, key < p.Lk));
            vf[c][kk] = __builtin_bit_cast(bf16x8, ld16(V + off, key < p.Lk));
        }
    const float sc = PRE ? 1.0f : p.scale * LOG2E;
    const float neg_inv_sc = PRE ? -LOG2E : -1.0f / p.scale;             // -lse log2(e) / sc: the scores' starting value
    const float ds_scale = PRE ? (1.0f / LOG2E) : p.scale;               // dK = dS^T q' / log2(e) on a pre-scaled q

    const FragAddr fa = frag_addr(lane);
    const TileSrc qs = tile_src<THREADS>(Q, p.Lq, p.q_rs, tid), dos = tile_src<THREADS>(DO, p.Lq, p.o_rs, tid);
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);   // stage 0, Q tile, this wave's rows

    f32x16 dv[KPW][4], dk[KPW][4];
#pragma unroll
    for (int c = 0; c < KPW; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv[c][i][r] = 0.f; dk[c][i][r] = 0.f; }

    // lse / delta of a tile's 64 queries, fetched one tile ahead and parked in LDS — negated, as the values the score /
    // dP accumulators START from (p_and_ds FOLD).  stat_load only ISSUES the loads (every wave, clamped index: no
    // branch, no arithmetic on the result — either would put the wait for them right here); stat_value turns the raw
    // numbers into the parked ones when they are stored, a tile later.
    auto stat_load = [&](int tile, float& lraw, float& draw) {
        const int q = min(tile * TB + lane, p.Lq - 1);
        lraw = LSE[q];
        draw = DEL[q];
    };
    auto stat_store = [&](int tile, float lraw, float draw, float* dst) {
        const bool in = tile * TB + lane < p.Lq;
        dst[lane] = (in && lraw > -INFINITY) ? lraw * neg_inv_sc : -INFINITY;   // no keys / past the end: P = exp2(-inf) = 0
        dst[64 + lane] = in ? -draw : 0.f;
    };
    float gl = 0.f, gd = 0.f;
    tile_dma<THREADS>(qs, t_first, wave_lds);                        // (a tile index past the end arrives as zeros)
    tile_dma<THREADS>(dos, t_first, wave_lds + TILE_BYTES);
    stat_load(t_first, gl, gd);
    if (tid < 64) stat_store(t_first, gl, gd, stat);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < n_tiles; ++t) {
        const unsigned char* qt = smem + (t & 1) * 2 * TILE_BYTES;
        const unsigned char* dot = qt + TILE_BYTES;
        const float* st = stat + (t & 1) * 128;
        const bool more = t + 1 < n_tiles;
        if (more) {
            const uint32_t nq = wave_lds + ((t + 1) & 1) * 2 * TILE_BYTES;
            tile_dma<THREADS>(qs, t_first + t + 1, nq);
            tile_dma<THREADS>(dos, t_first + t + 1, nq + TILE_BYTES);
            stat_load(t_first + t + 1, gl, gd);
        }
        // One tile = two halves of 32 queries, each with three stages: A  S = Q K^T and dP = dO V^T (16 products,
        // [query][key], lane = key; register r <-> query 64t + 32hb + 16(r>>3) + 8lh + (r&7)); B  the softmax
        // arithmetic (VALU); C  dV^T += dO^T P, dK^T += Q^T dS (16 products, [d][key]).  With ONE wave per SIMD nothing
        // else covers a stage's latency, so the halves are software-pipelined:  A0 | A1 + B0 | C0 + B1 | C1  — the
        // sched_group_barrier sequences put ~5 VALU instructions into the shadow of each 8-pass MFMA, and every stage's
        // LDS fragments are requested one stage ahead (hipcc's own order was read, wait, MFMA, ... then all the VALU:
        // 6 500 cycles per tile for 2 048 cycles of MFMA work — profiles/r05_pmc_attn_bwd.json).
        static_assert(KPW == 1, "the pipelined body is written for one key block per wave");
        f32x16 s[2], dp[2];                                          // start from -lse / sc and -delta (p_and_ds FOLD)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s[hb][8 * a + e] = st[32 * hb + 16 * a + 8 * lh + e];
                    dp[hb][8 * a + e] = st[64 + 32 * hb + 16 * a + 8 * lh + e];
                }
        bf16x8 qa[8], da[8], ta[2][4], tq[2][4], pf[2][2], dsf[2][2];
        auto load_rows = [&](int hb) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                qa[kk] = *(const bf16x8*)(qt + fa.row[kk] + hb * 32 * 256);
                da[kk] = *(const bf16x8*)(dot + fa.row[kk] + hb * 32 * 256);
            }
        };
        auto load_tr = [&](int hb, int a) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const uint32_t ad = fa.tr[db] + (32 * hb + 16 * a) * 256;
                ta[a][db] = tr_frag2(dot, ad);
                tq[a][db] = tr_frag2(qt, ad);
            }
        };
        auto stage_a = [&](int hb) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                s[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[kk], kf[0][kk], s[hb], 0, 0, 0);
                dp[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[kk], vf[0][kk], dp[hb], 0, 0, 0);
            }
        };
        auto stage_c = [&](int hb, int a) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[a][db], pf[hb][a], dv[0][db], 0, 0, 0);
                dk[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[a][db], dsf[hb][a], dk[0][db], 0, 0, 0);
            }
        };
        // no key mask: a lane is ONE key, and a key at or past klen only spoils its own dK / dV column (zeroed after the loop)
        u32x4 cp[2][2], cd[2][2];                                    // packed P / dS: [half][16-query group]
        auto slice_b = [&](int hb, int i) {                          // scores 2i, 2i+1 of a half (p_and_ds FOLD, no mask)
            const float p0 = __builtin_amdgcn_exp2f(PRE ? s[hb][2 * i] : s[hb][2 * i] * sc);
            const float p1 = __builtin_amdgcn_exp2f(PRE ? s[hb][2 * i + 1] : s[hb][2 * i + 1] * sc);
            cp[hb][i >> 2][i & 3] = pack_bf2(p0, p1);
            cd[hb][i >> 2][i & 3] = pack_bf2(p0 * dp[hb][2 * i], p1 * dp[hb][2 * i + 1]);
        };
        load_rows(0);
        __builtin_amdgcn_sched_barrier(0);
        stage_a(0);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(1);
        load_tr(0, 0);
        load_tr(0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {                             // A1 + B0: two products, one slice (2 exp2 + 2 mul + 2 cvt)
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[kk], kf[0][kk], s[1], 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[kk], vf[0][kk], dp[1], 0, 0, 0);
            slice_b(0, kk);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            pf[0][a] = __builtin_bit_cast(bf16x8, cp[0][a]);
            dsf[0][a] = __builtin_bit_cast(bf16x8, cd[0][a]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)                                  // C0 + B1
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[a][db], pf[0][a], dv[0][db], 0, 0, 0);
                dk[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[a][db], dsf[0][a], dk[0][db], 0, 0, 0);
                slice_b(1, 4 * a + db);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            pf[1][a] = __builtin_bit_cast(bf16x8, cp[1][a]);
            dsf[1][a] = __builtin_bit_cast(bf16x8, cd[1][a]);
        }
        load_tr(1, 0);
        load_tr(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        stage_c(1, 0);
        stage_c(1, 1);
        if (more && tid < 64)                                        // the other stage's statistics: last read a tile ago
            stat_store(t_first + t + 1, gl, gd, stat + ((t + 1) & 1) * 128);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int c = 0; c < KPW; ++c)
        dkdv_store(p, wk, worker, wid, split, b, head, wave * (32 * KPW) + 32 * c + li, lh, key0 + 32 * c, klen, ds_scale, dk[c], dv[c]);
}

// ---------------------------------------------------------------------------------------------- dK, dV: the stream
// The same work decomposition, tiles, LDS layout and arithmetic as attn_bwd2_dkdv_kernel<4, 1>, with the loop written
// out instruction by instruction (gen_attn_bwd_w64.py -> attention_bwd2_asm.inc; that file's header has the schedule
// and the register map).  This function computes addresses and descriptors, hands them to the stream, takes the
// accumulators back through LDS and stores them like the HIP kernel does.
template <bool PRE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd2_dkdv_w64_kernel(const omh_attn_bwd_args p, const int k_blocks, const BwdSplit wk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // 4 x [Q tile | dO tile] + statistics
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool worker = (int)blockIdx.x >= wk.n_regular;
    const int wid = worker ? wk.n_regular + ((int)blockIdx.x - wk.n_regular) / wk.splits : xcd_remap((int)blockIdx.x, wk.n_regular);
    const int split = worker ? ((int)blockIdx.x - wk.n_regular) % wk.splits : 0;
    const int kb = wid % k_blocks, bh = wid / k_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles_all = (p.Lq + TB - 1) / TB;
    int t_first = 0, n_tiles = n_tiles_all;
    if (worker) {
        const int per = (n_tiles_all + wk.splits - 1) / wk.splits;
        t_first = min(split * per, n_tiles_all);
        n_tiles = min(t_first + per, n_tiles_all) - t_first;
    }
    if (n_tiles > 0) {
        const int key = kb * 128 + wave * 32 + li;
        const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
        const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
        const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
        const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
        const float* LSE = p.lse + ((int64_t)b * p.H + head) * p.Lq;
        const float* DEL = p.delta + ((int64_t)b * p.H + head) * p.Lq;
        auto rsrc = [](const void* base, int64_t bytes) {
            const uint64_t a = (uint64_t)base;
            return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a),
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)),
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bytes), 0x00020000u};
        };
        auto uni = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };   // (all of these are wave-uniform)
        // rows past the end read as zeros (the range check covers voffset + soffset + immediate)
        const u32x4 rq = rsrc(Q, (((int64_t)p.Lq - 1) * p.q_rs + D) * 2), rdo = rsrc(DO, (((int64_t)p.Lq - 1) * p.o_rs + D) * 2);
        const u32x4 rk = rsrc(K, (((int64_t)p.Lk - 1) * p.k_rs + D) * 2), rv = rsrc(V, (((int64_t)p.Lk - 1) * p.k_rs + D) * 2);
        const u32x4 rlse = rsrc(LSE, (int64_t)p.Lq * 4), rdel = rsrc(DEL, (int64_t)p.Lq * 4);
        const uint32_t lds0 = lds_addr(smem);
        // LDS-DMA sources: chunk c = tid (+ 256 j: 16 rows further, same swizzle) -> row tid >> 4, logical slot (tid & 15) ^ swz(row)
        const int drow = tid >> 4;
        uint32_t vodq = (uint32_t)((drow * (int)p.q_rs + (((tid & 15) ^ (int)swz(drow)) << 3)) * 2);
        uint32_t vodo = (uint32_t)((drow * (int)p.o_rs + (((tid & 15) ^ (int)swz(drow)) << 3)) * 2);
        uint32_t vokv = key < p.Lk ? (uint32_t)(((int64_t)key * p.k_rs + lh * 8) * 2) : 0x80000000u;   // out of range: zeros
        // fragment addresses (frag_addr): row fragments r0 = swap_bits23(li), slot (2 kk + lh) ^ swz(r0); transposed ones
        const int r0 = swap_bits23(li), z0 = (int)swz(r0);
        uint32_t kab = lds0 + (uint32_t)(r0 * 256 + ((lh ^ (z0 & 1)) << 4)), xh = (uint32_t)(z0 >> 1);
        const int gq = lane >> 4, i15 = lane & 15, fe = i15 >> 2, fq = i15 & 3;
        const int rlo = 8 * (gq >> 1) + fe, z1 = (int)swz(rlo), low2 = 2 * (gq & 1) + (fq >> 1);
        uint32_t tab = lds0 + (uint32_t)(rlo * 256 + ((low2 ^ (z1 & 3)) << 4) + (fq & 1) * 8), th = (uint32_t)(z1 >> 2);
        uint32_t vstr = lds0 + OMH_ATTN_BWD_W64_STAT_FIN + (uint32_t)(8 * lh * 4);
        uint32_t vraw = lds0 + OMH_ATTN_BWD_W64_STAT_RAW + (uint32_t)(lane * 4);
        uint32_t vost = (uint32_t)(lane * 4);
        uint32_t vdump = lds0 + (uint32_t)(wave * OMH_ATTN_BWD_W64_PARK_WAVE + lane * 16);
        // wave-uniform scalars
        const uint32_t ldsw = uni(lds0 + (uint32_t)wave * 1024u), sraw = uni(lds0 + OMH_ATTN_BWD_W64_STAT_RAW);
        const uint32_t sqp = uni((uint32_t)(16 * (int)p.q_rs * 2)), sop = uni((uint32_t)(16 * (int)p.o_rs * 2));
        uint32_t sqn = uni((uint32_t)t_first * 4u * sqp), son = uni((uint32_t)t_first * 4u * sop), sstn = uni((uint32_t)t_first * 256u);
        const uint32_t ntiles = uni((uint32_t)n_tiles);
        const uint32_t k1 = __builtin_amdgcn_readfirstlane(__float_as_uint(PRE ? -LOG2E : -1.0f / p.scale));
        const uint32_t sc = __builtin_amdgcn_readfirstlane(__float_as_uint(PRE ? 1.0f : p.scale * LOG2E));
#define OMH_BWD_W64_OPERANDS                                                                                          \
                 : [sqn] "+s"(sqn), [son] "+s"(son), [sstn] "+s"(sstn), [vodq] "+v"(vodq), [vodo] "+v"(vodo),         \
                   [vokv] "+v"(vokv), [kab] "+v"(kab), [xh] "+v"(xh), [tab] "+v"(tab), [th] "+v"(th), [vstr] "+v"(vstr),\
                   [vraw] "+v"(vraw), [vost] "+v"(vost), [vdump] "+v"(vdump)                                          \
                 : [rq] "s"(rq), [rdo] "s"(rdo), [rk] "s"(rk), [rv] "s"(rv), [rlse] "s"(rlse), [rdel] "s"(rdel),      \
                   [sqp] "s"(sqp), [sop] "s"(sop), [ldsw] "s"(ldsw), [sraw] "s"(sraw), [ntiles] "s"(ntiles),          \
                   [k1] "s"(k1), [sc] "s"(sc)                                                                         \
                 : OMH_ATTN_BWD_W64_CLOBBERS
        if constexpr (PRE) asm volatile(OMH_ATTN_BWD_W64_ASM_PRE OMH_BWD_W64_OPERANDS);
        else asm volatile(OMH_ATTN_BWD_W64_ASM_GEN OMH_BWD_W64_OPERANDS);
    }
    // The accumulators come back through LDS: the stream parked them in this wave's part of the ring, block (x, g) =
    // registers 4g .. 4g+3 of accumulator x (0..3 dV, 4..7 dK) as 16 bytes per lane (lane = key li + 32 lh; the registers are
    // d = 32 db + 8 g + 4 lh + 0..3).  They are read back ROW-major — 16 lanes per key for bf16 rows (256 bytes), 32 for
    // fp32 — so that a store instruction writes whole rows instead of 64 pieces of 8 bytes in 64 rows (that epilogue cost
    // 18 of the kernel's 147 us at 4 clips).
    const unsigned char* park = smem + wave * OMH_ATTN_BWD_W64_PARK_WAVE;
    const float ds_scale = PRE ? (1.0f / LOG2E) : p.scale;
    const bool have = n_tiles > 0;
    auto take = [&](int x, int db, int g, int src_lane) {              // x: 0 dV, 1 dK
        return have ? *(const float4*)(park + ((x * 4 + db) * 4 + g) * OMH_ATTN_BWD_W64_PARK_BLOCK + src_lane * 16)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (!worker && p.out_bf16) {
        const int c = lane & 15, db = c >> 2, g = c & 3;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 4 * it + (lane >> 4), key = kb * 128 + wave * 32 + row;
                float4 lo = take(x, db, g, row), hi = take(x, db, g, 32 + row);
                const float f = x == 1 ? ds_scale : 1.0f;
                const bool in = key < klen;                           // (a select: the column may hold NaN)
                lo = in ? make_float4(lo.x * f, lo.y * f, lo.z * f, lo.w * f) : make_float4(0.f, 0.f, 0.f, 0.f);
                hi = in ? make_float4(hi.x * f, hi.y * f, hi.z * f, hi.w * f) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (key < p.Lk) {
                    uint16_t* dst = (uint16_t*)(x == 1 ? p.dk : p.dv) + (int64_t)b * p.dk_bs + (int64_t)key * p.dk_rs + head * D + c * 8;
                    *(uint4*)dst = make_uint4(pack_bf2(lo.x, lo.y), pack_bf2(lo.z, lo.w), pack_bf2(hi.x, hi.y), pack_bf2(hi.z, hi.w));
                }
            }
    } else {
        const int c4 = lane & 31, db = c4 >> 3, g = (c4 >> 1) & 3, half = c4 & 1;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = 2 * it + (lane >> 5), key = kb * 128 + wave * 32 + row;
                float4 v = take(x, db, g, half * 32 + row);
                const float f = x == 1 ? ds_scale : 1.0f;
                v = key < klen ? make_float4(v.x * f, v.y * f, v.z * f, v.w * f) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (worker) {                                        // partial sums over this worker's queries: [dK | dV] slabs
                    float* W = wk.ws + ((((int64_t)(wid - wk.n_regular) * wk.splits + split) * 2 + (x == 1 ? 0 : 1)) * 128 + wave * 32 + row) * D;
                    *(float4*)(W + c4 * 4) = v;
                } else if (key < p.Lk) {
                    float* dst = (float*)(x == 1 ? p.dk : p.dv) + (int64_t)b * p.dk_bs + (int64_t)key * p.dk_rs + head * D + c4 * 4;
                    *(float4*)dst = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------- dQ: the stream
// attn_bwd2_dq_kernel's work decomposition and arithmetic (4 waves x 32 queries, tiles of 64 keys), the loop as an
// instruction stream (gen_attn_bwd_w64.py, "the dQ stream"): one wave per SIMD, so ONE workgroup per CU where the HIP
// kernel runs two.  S starts from -lse' (16 copies: MFMA C operand) instead of subtracting it per score, which changes
// the rounding of the exponent by an ulp: equal to the HIP kernel within 1e-5, not bit for bit.
template <bool PRE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd2_dq_w64_kernel(const omh_attn_bwd_args p, const int q_blocks, const BwdSplit wk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // 4 x [K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool worker = (int)blockIdx.x >= wk.n_regular;
    const int wid = worker ? wk.n_regular + ((int)blockIdx.x - wk.n_regular) / wk.splits : xcd_remap((int)blockIdx.x, wk.n_regular);
    const int split = worker ? ((int)blockIdx.x - wk.n_regular) % wk.splits : 0;
    const int qb = wid % q_blocks, bh = wid / q_blocks;
    const int b = bh / p.H, head = bh % p.H;
    int klen = p.k_lens ? p.k_lens[b] : p.Lk;
    klen = min(max(klen, 0), p.Lk);
    const int n_tiles_all = (klen + TB - 1) / TB;
    int t_first = 0, n_tiles = n_tiles_all;
    if (worker) {
        const int per = (n_tiles_all + wk.splits - 1) / wk.splits;
        t_first = min(split * per, n_tiles_all);
        n_tiles = min(t_first + per, n_tiles_all) - t_first;
    }
    constexpr int PARK_WAVE = 16 * OMH_ATTN_BWD_W64_PARK_BLOCK;
    if (n_tiles > 0) {
        const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + head * D;
        const uint16_t* DO = (const uint16_t*)p.dout + (int64_t)b * p.o_bs + head * D;
        const uint16_t* K = (const uint16_t*)p.k + (int64_t)b * p.k_bs + head * D;
        const uint16_t* V = (const uint16_t*)p.v + (int64_t)b * p.k_bs + head * D;
        auto uni = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };   // (all of these are wave-uniform)
        auto rsrc = [&](const void* base, int64_t bytes) {
            const uint64_t a = (uint64_t)base;
            return u32x4{uni((uint32_t)a), uni((uint32_t)(a >> 32) & 0xffffu), uni((uint32_t)bytes), 0x00020000u};
        };
        const u32x4 rq = rsrc(Q, (((int64_t)p.Lq - 1) * p.q_rs + D) * 2), rdo = rsrc(DO, (((int64_t)p.Lq - 1) * p.o_rs + D) * 2);
        const u32x4 rk = rsrc(K, (((int64_t)p.Lk - 1) * p.k_rs + D) * 2), rv = rsrc(V, (((int64_t)p.Lk - 1) * p.k_rs + D) * 2);
        const uint32_t lds0 = lds_addr(smem);
        const int q_row = qb * 128 + wave * 32 + li;
        const bool q_ok = q_row < p.Lq;
        const int64_t row_i = ((int64_t)b * p.H + head) * p.Lq + q_row;
        const float l = q_ok ? p.lse[row_i] : 0.f;
        const float k1f = PRE ? -LOG2E : -1.0f / p.scale;                 // -lse log2(e) / sc
        float negl = (q_ok && l > -INFINITY) ? l * k1f : -INFINITY;      // no keys / past the end: P = exp2(-inf) = 0
        float negd = q_ok ? -p.delta[row_i] : 0.f;
        uint32_t voq = q_ok ? (uint32_t)(((int64_t)q_row * p.q_rs + lh * 8) * 2) : 0x80000000u;    // out of range: zeros
        uint32_t vodof = q_ok ? (uint32_t)(((int64_t)q_row * p.o_rs + lh * 8) * 2) : 0x80000000u;
        const int drow = tid >> 4;
        uint32_t vodk = (uint32_t)((drow * (int)p.k_rs + (((tid & 15) ^ (int)swz(drow)) << 3)) * 2);
        const int r0 = swap_bits23(li), z0 = (int)swz(r0);
        uint32_t kab = lds0 + (uint32_t)(r0 * 256 + ((lh ^ (z0 & 1)) << 4)), xh = (uint32_t)(z0 >> 1);
        const int gq = lane >> 4, i15 = lane & 15, fe = i15 >> 2, fq = i15 & 3;
        const int rlo = 8 * (gq >> 1) + fe, z1 = (int)swz(rlo), low2 = 2 * (gq & 1) + (fq >> 1);
        uint32_t tab = lds0 + (uint32_t)(rlo * 256 + ((low2 ^ (z1 & 3)) << 4) + (fq & 1) * 8), th = (uint32_t)(z1 >> 2);
        uint32_t lh8 = (uint32_t)(8 * lh);
        uint32_t vdump = lds0 + (uint32_t)(wave * PARK_WAVE + lane * 16);
        const uint32_t ldsw = uni(lds0 + (uint32_t)wave * 1024u);
        const uint32_t skp = uni((uint32_t)(16 * (int)p.k_rs * 2));
        uint32_t skn = uni((uint32_t)t_first * 4u * skp), srem = uni((uint32_t)(klen - t_first * TB));
        const uint32_t ntiles = uni((uint32_t)n_tiles);
        const uint32_t sc = uni(__float_as_uint(PRE ? 1.0f : p.scale * LOG2E));
#define OMH_BWD_DQ_W64_OPERANDS                                                                                       \
                 : [skn] "+s"(skn), [srem] "+s"(srem), [vodk] "+v"(vodk), [voq] "+v"(voq), [vodof] "+v"(vodof),       \
                   [kab] "+v"(kab), [xh] "+v"(xh), [tab] "+v"(tab), [th] "+v"(th), [negl] "+v"(negl), [negd] "+v"(negd),\
                   [lh8] "+v"(lh8), [vdump] "+v"(vdump)                                                               \
                 : [rq] "s"(rq), [rdo] "s"(rdo), [rk] "s"(rk), [rv] "s"(rv), [skp] "s"(skp), [ldsw] "s"(ldsw),        \
                   [ntiles] "s"(ntiles), [sc] "s"(sc)                                                                 \
                 : OMH_ATTN_BWD_W64_CLOBBERS
        if constexpr (PRE) asm volatile(OMH_ATTN_BWD_DQ_W64_ASM_PRE OMH_BWD_DQ_W64_OPERANDS);
        else asm volatile(OMH_ATTN_BWD_DQ_W64_ASM_GEN OMH_BWD_DQ_W64_OPERANDS);
    }
    // accumulators back through LDS, row-major (see the dK / dV kernel): block (db, g) = dQ^T registers 4g .. 4g+3 of
    // accumulator db, 16 bytes per lane (lane = query li + 32 lh; d = 32 db + 8 g + 4 lh + 0..3); dQ = scale dS K
    const unsigned char* park = smem + wave * PARK_WAVE;
    const bool have = n_tiles > 0;
    auto take = [&](int db, int g, int src_lane) {
        float4 v = have ? *(const float4*)(park + (db * 4 + g) * OMH_ATTN_BWD_W64_PARK_BLOCK + src_lane * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        return make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
    };
    if (!worker && p.out_bf16) {
        const int c = lane & 15, db = c >> 2, g = c & 3;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 4 * it + (lane >> 4), q_row = qb * 128 + wave * 32 + row;
            const float4 lo = take(db, g, row), hi = take(db, g, 32 + row);
            if (q_row < p.Lq)
                *(uint4*)((uint16_t*)p.dq + (int64_t)b * p.dq_bs + (int64_t)q_row * p.dq_rs + head * D + c * 8) =
                    make_uint4(pack_bf2(lo.x, lo.y), pack_bf2(lo.z, lo.w), pack_bf2(hi.x, hi.y), pack_bf2(hi.z, hi.w));
        }
    } else {
        const int c4 = lane & 31, db = c4 >> 3, g = (c4 >> 1) & 3, half = c4 & 1;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = 2 * it + (lane >> 5), q_row = qb * 128 + wave * 32 + row;
            const float4 v = take(db, g, half * 32 + row);
            if (worker)                                              // partial sums over this worker's keys
                *(float4*)(wk.ws + (((int64_t)(wid - wk.n_regular) * wk.splits + split) * 128 + wave * 32 + row) * D + c4 * 4) = v;
            else if (q_row < p.Lq)
                *(float4*)((float*)p.dq + (int64_t)b * p.dq_bs + (int64_t)q_row * p.dq_rs + head * D + c4 * 4) = v;
        }
    }
}

// out[row] = sum_s slab_s[row] in the order of s (fixed: repeatable bit for bit).  One wave per (tail tile, row);
// NOUT = 1: dQ rows; NOUT = 2: dK and dV rows of the same key.
template <int NOUT>
__global__ __launch_bounds__(256)
void attn_bwd2_sum_kernel(const omh_attn_bwd_args p, const int blocks, const BwdSplit wk) {
    const int lane = threadIdx.x & 63;
    const int rowid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tt = rowid / 128, r = rowid % 128;
    if (tt >= wk.n_tail) return;
    const int tile = wk.n_regular + tt;
    const int blk = tile % blocks, bh = tile / blocks;
    const int b = bh / p.H, head = bh % p.H;
    const int row = blk * 128 + r;
    if (row >= (NOUT == 1 ? p.Lq : p.Lk)) return;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float a0 = 0.f, a1 = 0.f;
        for (int s = 0; s < wk.splits; ++s) {
            const float2 v = *(const float2*)(wk.ws + ((((int64_t)tt * wk.splits + s) * NOUT + o) * 128 + r) * D + 2 * lane);
            a0 += v.x;
            a1 += v.y;
        }
        const int64_t eo = (NOUT == 1 ? (int64_t)b * p.dq_bs + (int64_t)row * p.dq_rs
                                      : (int64_t)b * p.dk_bs + (int64_t)row * p.dk_rs) + head * D + 2 * lane;
        void* base = NOUT == 1 ? p.dq : (o == 0 ? p.dk : p.dv);
        if (p.out_bf16) *(uint32_t*)((uint16_t*)base + eo) = pack_bf2(a0, a1);
        else *(float2*)((float*)base + eo) = make_float2(a0, a1);
    }
}

}  // namespace

// Split plans: the dQ kernel runs two workgroups per CU (250 VGPRs), the dK / dV kernel one (214 VGPRs + 158 AGPRs:
// forcing two spills 179 registers); at least 4 tiles of 64 positions per worker
// option OMH_ATTN_BWD_W64: "0" both HIP kernels, "k" the dK / dV stream only, otherwise (default) both streams
static bool bwd2_dq_stream() {
    const char* e = omh_opt(OMH_OPT_ATTN_BWD_W64);
    return !(e && (e[0] == '0' || e[0] == 'k'));
}
static OmhSplitPlan bwd2_plan(const omh_attn_bwd_args& a, bool dq) {
    const int blocks = dq ? (a.Lq + 127) / 128 : (a.Lk + 127) / 128;
    const int nwg = blocks * a.H * a.B;
    OmhSplitPlan none = {nwg, 0, 1};
    const char* e = omh_opt(OMH_OPT_ATTN_SPLIT);                      // "0": never split (A/B timing; tests flip it in-process)
    if (e && e[0] == '0') return none;
    // OMH_ATTN_SPLIT=tail: also the last round of a launch that fills the chip (measured: no gain, omh_common.h)
    const bool tail = e && e[0] == 't';
    return omh_tail_split_plan(nwg, (dq && !bwd2_dq_stream() ? 2 : 1) * omh_cu_count(), ((dq ? a.Lk : a.Lq) + TB - 1) / TB, 4, !tail, tail ? 0.8 : 0.67);
}
static int64_t bwd2_ws_bytes(const OmhSplitPlan& pl, int nout) {
    return (int64_t)pl.n_tail * pl.splits * nout * 128 * D * 4;
}
extern "C" int64_t omh_flash_attn_bwd_workspace_bytes(const omh_attn_bwd_args* args) {
    if (!args || !args->o32 || args->B <= 0 || args->H <= 0 || args->Lq <= 0 || args->Lk <= 0) return 0;
    int64_t n = 0;
    if (args->phase == 0 || args->phase == 2) n += bwd2_ws_bytes(bwd2_plan(*args, true), 1);
    if (args->phase == 0 || args->phase == 3) n += bwd2_ws_bytes(bwd2_plan(*args, false), 2);
    return n;
}

// called by omh_flash_attn_bwd_d128 (attention_bwd.hip) when args->o32 is set; arguments already validated there
int omh_launch_attn_bwd2(const omh_attn_bwd_args& a, hipStream_t s) {
    // 32-bit buffer offsets inside one (batch, head) slice
    // (+ 4 tiles: the dK / dV stream requests up to three tiles past the end, which must stay out of range, not wrap)
    if (((int64_t)a.Lq + 4 * TB) * a.q_rs * 2 >= 0x7fffffffLL || ((int64_t)a.Lk + 4 * TB) * a.k_rs * 2 >= 0x7fffffffLL ||
        ((int64_t)a.Lq + 4 * TB) * a.o_rs * 2 >= 0x7fffffffLL)
        return OMH_E_SHAPE;
    if (((uintptr_t)a.o32 & 15) || (a.o_rs & 3) || (a.o_bs & 3)) return OMH_E_ALIGN;
    constexpr int LDS_DQ = 4 * TILE_BYTES, LDS_KV = 4 * TILE_BYTES + 2 * 128 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dkdv_kernel<4, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dkdv_kernel<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dkdv_w64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, OMH_ATTN_BWD_W64_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dkdv_w64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OMH_ATTN_BWD_W64_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dq_w64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, OMH_ATTN_BWD_DQ_W64_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd2_dq_w64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OMH_ATTN_BWD_DQ_W64_LDS);
        attr_set = true;
    }
    const int k_blocks = (a.Lk + 127) / 128, q_blocks = (a.Lq + 127) / 128;
    if (a.phase < 0 || a.phase > 3) return OMH_E_BADARG;
    const int64_t pairs = (int64_t)a.B * a.Lq * a.H;
    if (pairs >= 0x7fffffffLL) return OMH_E_SHAPE;
    if (a.phase == 0 || a.phase == 1) {                              // phase 0 = delta, then dQ, then dK / dV
        hipLaunchKernelGGL(attn_bwd2_delta_kernel, dim3((unsigned)((pairs + 15) / 16)), dim3(256), 0, s, a);
        if (a.phase == 1) return 0;
    }
    // workspace: [dQ slabs | dK, dV slabs]; without (enough of) it the kernels run unsplit
    OmhSplitPlan pq = bwd2_plan(a, true), pk = bwd2_plan(a, false);
    const bool run_q = a.phase == 0 || a.phase == 2, run_k = a.phase == 0 || a.phase == 3;
    const int64_t need_q = run_q ? bwd2_ws_bytes(pq, 1) : 0, need_k = run_k ? bwd2_ws_bytes(pk, 2) : 0;
    if (!a.workspace || ((uintptr_t)a.workspace & 15) || a.workspace_bytes < need_q + need_k) {
        pq.n_regular += pq.n_tail; pq.n_tail = 0; pq.splits = 1;
        pk.n_regular += pk.n_tail; pk.n_tail = 0; pk.splits = 1;
    }
    BwdSplit wq = {pq.n_regular, pq.n_tail, pq.splits, (float*)a.workspace};
    BwdSplit wkv = {pk.n_regular, pk.n_tail, pk.splits, (float*)((char*)a.workspace + (pq.n_tail ? need_q : 0))};
    if (run_q) {                                                                                               // phase 0: writes delta
        const dim3 grid(wq.n_regular + wq.n_tail * wq.splits);
        if (bwd2_dq_stream()) {
            if (a.q_prescaled) hipLaunchKernelGGL(attn_bwd2_dq_w64_kernel<true>, grid, dim3(256), OMH_ATTN_BWD_DQ_W64_LDS, s, a, q_blocks, wq);
            else hipLaunchKernelGGL(attn_bwd2_dq_w64_kernel<false>, grid, dim3(256), OMH_ATTN_BWD_DQ_W64_LDS, s, a, q_blocks, wq);
        } else if (a.q_prescaled) hipLaunchKernelGGL(attn_bwd2_dq_kernel<true>, grid, dim3(256), LDS_DQ, s, a, q_blocks, wq);
        else hipLaunchKernelGGL(attn_bwd2_dq_kernel<false>, grid, dim3(256), LDS_DQ, s, a, q_blocks, wq);
        if (wq.n_tail)
            hipLaunchKernelGGL(attn_bwd2_sum_kernel<1>, dim3((wq.n_tail * 128 + 3) / 4), dim3(256), 0, s, a, q_blocks, wq);
    }
    if (run_k) {                                                                                               // reads delta
        const dim3 grid(wkv.n_regular + wkv.n_tail * wkv.splits);
        const char* e = omh_opt(OMH_OPT_ATTN_BWD_W64);                // "0": the HIP kernel (A/B timing, tests)
        if (!(e && e[0] == '0')) {
            if (a.q_prescaled) hipLaunchKernelGGL(attn_bwd2_dkdv_w64_kernel<true>, grid, dim3(256), OMH_ATTN_BWD_W64_LDS, s, a, k_blocks, wkv);
            else hipLaunchKernelGGL(attn_bwd2_dkdv_w64_kernel<false>, grid, dim3(256), OMH_ATTN_BWD_W64_LDS, s, a, k_blocks, wkv);
        } else if (a.q_prescaled) hipLaunchKernelGGL((attn_bwd2_dkdv_kernel<4, 1, true>), grid, dim3(256), LDS_KV, s, a, k_blocks, wkv);
        else hipLaunchKernelGGL((attn_bwd2_dkdv_kernel<4, 1, false>), grid, dim3(256), LDS_KV, s, a, k_blocks, wkv);
        if (wkv.n_tail)
            hipLaunchKernelGGL(attn_bwd2_sum_kernel<2>, dim3((wkv.n_tail * 128 + 3) / 4), dim3(256), 0, s, a, k_blocks, wkv);
    }
    return 0;
}
