// bf16 MFMA GEMM  C[m][n] = epi( sum_k A[m][k] * B[n][k] )  for gfx950.
//
// Replaces aten addmm under every nn.Linear of the reference DiT
// (seaweed_apt/wan/modules/model.py:125-128,160,176-178,185,272-274,344,465-467)
// and the patch-embedding Conv3d-as-GEMM (model.py:463,515).
//
// Tiling: 128(m) x 128(n) x 64(k) per 256-thread workgroup, 4 waves as 2x2,
// each wave a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.
// The weight rows (n) go in the MFMA A slot and the activation rows (m) in
// the B slot, so each lane ends up with runs of 4 consecutive n for one m:
// 8-byte bf16 / 16-byte fp32 epilogue stores.  Both operand tiles are staged
// HBM -> registers -> LDS ([128][64] bf16, 16-byte slots XOR-swizzled by
// (row>>1)&7 so the ds_read_b128 fragment reads are bank-conflict free),
// double buffered with one barrier per k-step; the next tile's global loads
// are issued before the MFMA block of the current one.
#include "omh_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile

struct GemmGeom { int tiles_m, tiles_n; };

__device__ __forceinline__ uint32_t lds_slot_addr(int row, int slot) {
    return (uint32_t)(row * (BK * 2) + ((slot ^ ((row >> 1) & 7)) << 4));
}

template <int EPI>
__global__ __launch_bounds__(256, 2)
void gemm_bf16_nt_kernel(const omh_gemm_args p, const GemmGeom g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const int tm = wid / g.tiles_n, tn = wid % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int64_t zb = blockIdx.z;
    const __bf16* __restrict__ A = (const __bf16*)p.A + zb * p.strideA;
    const __bf16* __restrict__ B = (const __bf16*)p.B + zb * p.strideB;

    // staging assignment: 4 x 16-byte chunks per operand per thread
    int st_row[4], st_slot[4];
    const __bf16* a_src[4];
    const __bf16* b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 256 * j;
        st_row[j] = c >> 3;
        st_slot[j] = c & 7;
        const int am = min(m0 + st_row[j], p.M - 1);
        const int bn = min(n0 + st_row[j], p.N - 1);
        a_src[j] = A + (int64_t)am * p.lda + st_slot[j] * 8;
        b_src[j] = B + (int64_t)bn * p.ldb + st_slot[j] * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // K % 8 == 0: a chunk is entirely inside or outside [0, K); read a
            // clamped in-bounds address and zero the value (no divergent pointers)
            const int kc = k0 + st_slot[j] * 8;
            const int kd = min(kc, p.K - 8) - st_slot[j] * 8;
            ra[j] = *(const uint4*)(a_src[j] + kd);
            rb[j] = *(const uint4*)(b_src[j] + kd);
            if (kc >= p.K) { ra[j] = zero4; rb[j] = zero4; }
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* xa = smem + buf * 2 * TILE_BYTES;
        unsigned char* xb = xa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t off = lds_slot_addr(st_row[j], st_slot[j]);
            *(uint4*)(xa + off) = ra[j];
            *(uint4*)(xb + off) = rb[j];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* xa = smem + buf * 2 * TILE_BYTES;   // activations (m)
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
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    const bool vec_ok = (p.ldc & 3) == 0;
    float* Cf = (float*)p.C + zb * p.strideC;
    uint16_t* Ch = (uint16_t*)p.C + zb * p.strideC;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int m = m0 + wm * 64 + im * 32 + li;
        if (m >= p.M) continue;
        const float bias_m = (p.bias_mode == OMH_BIAS_M && p.bias) ? p.bias[m] : 0.f;
        const int64_t gb = (EPI == OMH_EPI_RESID && p.gate1) ? (int64_t)(m / p.gate_rows) * p.gate1_stride : 0;
#pragma unroll
        for (int in = 0; in < 2; ++in) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * 64 + in * 32 + 8 * gq + 4 * lh;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[im][in][4 * gq + e] + bias_m;
                const bool full = vec_ok && (n + 3 < p.N);
                if (p.bias_mode == OMH_BIAS_N && p.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += p.bias[n + e];
                }
                if (EPI == OMH_EPI_GELU_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                }
                if (EPI == OMH_EPI_GELU_ERF_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.7071067811865476f));
                }
                if (EPI == OMH_EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.N) {
                            float gt = p.gate_const;
                            if (p.gate0) gt += p.gate0[n + e];
                            if (p.gate1) gt += p.gate1[gb + n + e];
                            v[e] *= gt;
                        }
                    }
                }
                const int64_t off = (int64_t)m * p.ldc + n;
                if (EPI == OMH_EPI_BF16 || EPI == OMH_EPI_GELU_BF16 || EPI == OMH_EPI_GELU_ERF_BF16) {
                    if (full) {
                        uint2 pk;
                        pk.x = pack_bf2(v[0], v[1]);
                        pk.y = pack_bf2(v[2], v[3]);
                        *(uint2*)(Ch + off) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) Ch[off + e] = f2bf(v[e]);
                    }
                } else if (EPI == OMH_EPI_F32) {
                    if (full) {
                        *(float4*)(Cf + off) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) Cf[off + e] = v[e];
                    }
                } else {  // RESID / F32_ACCUM: read-modify-write
                    if (full) {
                        float4 o = *(const float4*)(Cf + off);
                        o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                        *(float4*)(Cf + off) = o;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) Cf[off + e] += v[e];
                    }
                }
            }
        }
    }
}

template <int EPI>
int launch(const omh_gemm_args& a, hipStream_t s) {
    GemmGeom g;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(g.tiles_m * g.tiles_n, 1, a.batch);
    omh_clear_status();
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI>, grid, dim3(256), 0, s, a, g);
    return omh_launch_status();
}

}  // namespace

extern "C" int omh_gemm_bf16(const omh_gemm_args* args, omh_stream_t stream) {
    if (!args || !args->A || !args->B || !args->C) return OMH_E_BADARG;
    const omh_gemm_args& a = *args;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch <= 0) return OMH_E_BADARG;
    if ((a.K & 7) || (a.lda & 7) || (a.ldb & 7)) return OMH_E_ALIGN;
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return OMH_E_ALIGN;
    if ((a.strideA & 7) || (a.strideB & 7) || (a.strideC & 3)) return OMH_E_ALIGN;
    if (a.epilogue == OMH_EPI_RESID && a.gate1 && a.gate_rows <= 0) return OMH_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    switch (a.epilogue) {
        case OMH_EPI_BF16:      return launch<OMH_EPI_BF16>(a, s);
        case OMH_EPI_F32:       return launch<OMH_EPI_F32>(a, s);
        case OMH_EPI_GELU_BF16: return launch<OMH_EPI_GELU_BF16>(a, s);
        case OMH_EPI_RESID:     return launch<OMH_EPI_RESID>(a, s);
        case OMH_EPI_F32_ACCUM: return launch<OMH_EPI_F32_ACCUM>(a, s);
        case OMH_EPI_GELU_ERF_BF16: return launch<OMH_EPI_GELU_ERF_BF16>(a, s);
        default: return OMH_E_BADARG;
    }
}
