// Shared device helpers for the gfx950 kernels of libomh.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omh.h"

// ---------------------------------------------------------------------------------------------------------------------
// Process options (ABI v10, include/omh.h: omh_set_option / omh_get_option).  Every dispatch switch of the library —
// kernel-family overrides for tests and A/B timing — lives in ONE table.  The environment is consulted exactly once per
// process (the first omh_opt() / omh_set_option() call: variable OMH_<NAME> seeds option <NAME>); after that the launch
// path never calls getenv and a switch changes only through omh_set_option().  omh_opt() returns NULL for an unset option.
#define OMH_OPTIONS(X)                                                                                                   \
    X(ATTN_KERNEL) X(ATTN_SPLIT) X(W64_SPLIT) X(W64_VARIANT) X(CONV_PERSIST) X(CONV_W64_UP2) X(LN_RPW) X(GEMM_GROUP_M)    \
    X(GEMM_TILE) X(GEMM_RULE) X(GEMM_KERNEL) X(GEMM_W64_GBWD) X(GEMM_W64_GAUX) X(GEMM_W64_R192) X(GEMM_W64_BF16M)        \
    X(GEMM_W64_N192) X(GEMM_TN_SPLIT) X(GEMM_TN_W64) X(GEMM_TN_TILE) X(GEMM_TN_GROUP_TILE) X(GEMM_SPLITK) X(CONV_TILE)   \
    X(CONV_WIDE_MIN) X(CONV_W64) X(CONV_FUSE_NORM) X(CONV_KW3) X(GEMM_QKV) X(ATTN_BWD_W64) X(GEMM_W64_P256) X(RMS_PAIR_ROW)
enum OmhOpt {
#define X(n) OMH_OPT_##n,
    OMH_OPTIONS(X)
#undef X
    OMH_OPT_COUNT
};
const char* omh_opt(int id);

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define OMH_WAVE 64

// round-to-nearest-even fp32 -> bf16 bit pattern (NaN kept quiet)
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// two fp32 -> packed bf16x2 with the gfx950 hardware convert v_cvt_pk_bf16_f32 (RNE, NaN-safe), through the vector
// conversion the compiler selects it for.  Rounds 1-4 had a one-line asm here; inline asm is opaque to the compiler's
// hazard recognizer, and on gfx950 a VALU instruction that reads the result of a transcendental one (v_exp_f32,
// v_rcp_f32, ...) in the very next issue slot gets the OLD register contents ("trans forwarding" hazard, 1 wait
// state).  Round 5's attention backward put exp2 -> pack back to back and produced garbage in dV; the conversion below
// lets the compiler insert the s_nop where needed (and schedule the convert like any other VALU instruction).
typedef float omh_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 omh_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const omh_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, omh_bf16x2));
}
// raw v_exp_f32 (2^x): no denormal fix-up sequence; inputs here are <= 0 and
// results below 2^-126 may flush to zero, which is what a softmax wants.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ))  ==  x * sigmoid(2u) = x / (1 + 2^(-2u log2 e)).
// Raw v_exp_f32 / v_rcp_f32 (1 ulp-class) instead of the IEEE division sequence: ~9 VALU per element,
// this runs 128 times per lane in a GEMM epilogue.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float x2 = x * x;
    const float k = -2.0f * 0.7978845608028654f * 1.4426950408889634f;          // -2 c log2(e)
    const float t = k * x * fmaf(0.044715f, x2, 1.0f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// d/dx gelu_tanh(x) = 0.5 (1 + t) + 0.5 x (1 - t^2) c (1 + 3 a x^2),  t = tanh(c (x + a x^3))
// With s = sigmoid(2u) = 1 / (1 + 2^(-2u log2 e)):  0.5 (1 + t) = s  and  0.5 (1 - t^2) = 2 s (1 - s), so
//   gelu' = s + x * 2 s (1 - s) * c (1 + 3 a x^2)      (raw v_exp_f32 / v_rcp_f32: ~12 VALU, no IEEE division —
// this runs 128 times per lane in the epilogue of the FFN dgrad GEMM)
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float c = 0.7978845608028654f, a = 0.044715f;
    const float x2 = x * x;
    const float k = -2.0f * c * 1.4426950408889634f;                  // -2 c log2(e)
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k * x * fmaf(a, x2, 1.0f)));
    return fmaf(x * (2.0f * c) * fmaf(3.0f * a, x2, 1.0f), s * (1.0f - s), s);
}
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware bijective remap of a 1-D block id: consecutive *work* ids land on
// the same XCD (block b runs on XCD b % 8 — observed, speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    const int q = nwg / nx, r = nwg % nx;
    const int xcd = bid % nx, idx = bid / nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Work id -> output tile.  Consecutive ids (which xcd_remap keeps on one XCD, ~64 resident at a
// time) walk 8 m-tiles x all n-tiles in m-fastest order, so the resident set is an 8 x 8 patch of
// tiles: 8 A panels + 8 B panels live in that XCD's 4 MiB L2 instead of 1 + 64.
__device__ __forceinline__ void tile_of(int wid, int tiles_m, int tiles_n, int& tm, int& tn, const int GM = 8) {
    const int per_group = GM * tiles_n;
    const int gid = wid / per_group, in_g = wid - gid * per_group;
    const int first_m = gid * GM;
    const int gsz = min(tiles_m - first_m, GM);
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

// Zero fill of rows x cols fp32 (row pitch ld) as a KERNEL.  Not hipMemsetAsync: under hipGraph stream capture a
// memset node recorded through this library was replayed out of stream order on ROCm 7 (round-2 finding: the
// input-gradient scratch of omh_dense_f32_bwd lives in recycled graph-pool memory, the memset ran early, a later
// kernel of the graph reused the block, and the atomics then added onto that kernel's data — every replay after
// the first returned wrong time-embedding gradients).  A kernel node keeps its place.
__global__ __launch_bounds__(256) static void omh_zero_f32_kernel(float* __restrict__ p, int64_t rows, int64_t cols,
                                                                  int64_t ld) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}
static inline void omh_zero_f32(float* p, int64_t rows, int64_t cols, int64_t ld, hipStream_t s) {
    const int64_t n = rows * cols;
    int64_t g = (n + 255) / 256;
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(omh_zero_f32_kernel, dim3((unsigned)g), dim3(256), 0, s, p, rows, cols, ld);
}

// omh_set_deterministic()'s state (dit_elementwise.hip): launchers that combine partial sums with atomics ask it
bool omh_deterministic();

// hipGetLastError() reports the last error of ANY earlier runtime call on this
// thread (e.g. a probe made by the host framework): clear it before a launch
// so the status returned after the launch belongs to that launch only.
static inline void omh_clear_status() { (void)hipGetLastError(); }
static inline int omh_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}


// ---- split of a launch's last, partly filled round of workgroups (host side; attention.hip, attention_bwd2.hip) ----
// A launch of `nwg` equal workgroups on `slots` resident slots takes ceil(nwg / slots) rounds; its last round holds
// r = nwg mod slots workgroups.  Those r are split into `splits` workers each over their inner loop (`loop_tiles` tiles,
// at least `min_tiles` per worker), dispatched as the LAST blocks of the same launch: they start when the last full
// round drains, and that round then costs ceil(r splits / slots) / splits of a full one — the smallest `splits` <= 8
// that minimises this is taken, if it saves at least a third of the round (the workers write partial results into a
// workspace that a small kernel combines in a fixed order: that is not free).  Examples (MI355X, 256 CUs): 624
// workgroups on 512 slots: r = 112, 4 workers each, 1.25 rounds instead of 2; 156 on 512 (one clip): 3 workers each,
// 1/3 of a round; 156 on 256: 3 workers, 2/3.
// `single_round_only`: split only launches that do not fill the chip once.  Measured on the training step (round 4,
// profiles/r04_attention_split_ab.txt): a workgroup alone on its CU runs almost twice as fast as one that shares it, so
// the partly filled LAST round of a long launch costs about half a round, not a whole one — splitting it saves 3-13 %
// of the kernel and the combine pass (10 us: the partial results are HBM traffic) takes that back; a launch of 156
// workgroups (one clip) on 256 CUs gains 25-37 % before the combine.
struct OmhSplitPlan { int n_regular, n_tail, splits; };
static inline OmhSplitPlan omh_tail_split_plan(int nwg, int slots, int loop_tiles, int min_tiles, bool single_round_only = false,
                                               double max_cost = 0.67) {
    OmhSplitPlan pl = {nwg, 0, 1};
    if (nwg <= 0 || slots <= 0) return pl;
    if (single_round_only && nwg >= slots) return pl;
    const int r = nwg % slots;
    if (r == 0) return pl;
    int smax = loop_tiles / min_tiles;
    if (smax > 8) smax = 8;
    int best_s = 1;
    double best = 1.0;
    for (int sp = 2; sp <= smax; ++sp) {
        const double c = (double)((r * sp + slots - 1) / slots) / sp;
        if (c < best - 1e-9) { best = c; best_s = sp; }
    }
    // at least a third of the round must go: 192 workgroups on 256 slots split 4 ways (0.75 of a round on paper)
    // measured 0.9 % SLOWER per training step at 4 clips — a lone workgroup already has its CU to itself
    if (best_s < 2 || best > max_cost) return pl;
    pl.n_tail = r;
    pl.n_regular = nwg - r;
    pl.splits = best_s;
    return pl;
}
static inline int omh_cu_count() {                       // whole XCDs; cached
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        ncu -= ncu % 8;
        if (ncu < 8) ncu = 256;
    }
    return ncu;
}
