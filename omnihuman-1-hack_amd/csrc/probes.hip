// Measurement entry point, not part of the product path: the rate at which this GPU sustains back-to-back
// v_mfma_f32_32x32x16_bf16 (the instruction every GEMM / attention / convolution stream of this library is made of),
// one wave per SIMD, no memory traffic at all — once with constant operands and once with operands that toggle like data.
// bench.py reports both next to the 2.5 PFLOP/s of the data sheet: on the boxes of round 5 the chip sustained 2.2 PFLOP/s
// on constant operands and 1.3 PFLOP/s on random ones (the clock follows the power the multipliers draw) — the attention
// stream's 1.4 PFLOP/s is to be read against THAT (profiles/r05_rate_probes.txt; tools/probes/mfma_rate_probe.hip is the
// stand-alone form with the issue-rate experiments).
#include "omh_common.h"

namespace {

template <bool RND>
__global__ __launch_bounds__(256, 1) void mfma_rate_kernel(float* out, int iters, int salt) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (RND) {                                               // hashed values in [-2, 2): every operand bit toggles
                uint32_t h = (threadIdx.x * 2654435761u) ^ ((i * 8 + e + salt) * 40503u + blockIdx.x * 977u);
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                uint32_t g = h * 3266489917u; g ^= g >> 16;
                ra[i][e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 16384.0f));
                rb[i][e] = (__bf16)(((int)(g & 0xffff) - 32768) * (1.0f / 16384.0f));
            } else {
                ra[i][e] = (__bf16)(float)((threadIdx.x & 63) + e + salt);
                rb[i][e] = (__bf16)(float)(e + salt);
            }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j & 7]) : "v"(ra[RND ? (j & 7) : 0]), "v"(rb[RND ? ((j >> 3) & 7) : 0]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The same measurement on v_mfma_f32_16x16x32_bf16 (the shape the vendor library's GEMM kernels use: MI16x16x1): same flops
// per cycle on paper, a quarter of the accumulator registers per instruction, twice the operand registers per flop.
template <bool RND>
__global__ __launch_bounds__(256, 1) void mfma16_rate_kernel(float* out, int iters, int salt) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8 ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (RND) {
                uint32_t h = (threadIdx.x * 2654435761u) ^ ((i * 8 + e + salt) * 40503u + blockIdx.x * 977u);
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                uint32_t g = h * 3266489917u; g ^= g >> 16;
                ra[i][e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 16384.0f));
                rb[i][e] = (__bf16)(((int)(g & 0xffff) - 32768) * (1.0f / 16384.0f));
            } else {
                ra[i][e] = (__bf16)(float)((threadIdx.x & 63) + e + salt);
                rb[i][e] = (__bf16)(float)(e + salt);
            }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 128; ++j)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j & 15]) : "v"(ra[RND ? (j & 7) : 0]), "v"(rb[RND ? ((j >> 3) & 7) : 0]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

extern "C" int omh_probe_mfma_tflops(int32_t random_operands, int32_t iters, float* scratch, int64_t scratch_floats,
                                     float* tflops_out, omh_stream_t stream) {
    if (!scratch || !tflops_out || iters <= 0) return OMH_E_BADARG;
    const int grid = omh_cu_count();
    if (scratch_floats < (int64_t)grid * 256) return OMH_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    hipError_t err = hipEventCreate(&e0);
    if (err != hipSuccess) return (int)err;
    err = hipEventCreate(&e1);
    if (err != hipSuccess) { (void)hipEventDestroy(e0); return (int)err; }
    // random_operands: 0 / 1 = 32x32x16 on constant / random operands (64 per iteration), 2 / 3 = 16x16x32 likewise (128 per
    // iteration: the same flops per iteration)
    auto launch = [&](int n) {
        if (random_operands == 1) hipLaunchKernelGGL(mfma_rate_kernel<true>, dim3(grid), dim3(256), 0, s, scratch, n, 1);
        else if (random_operands == 0) hipLaunchKernelGGL(mfma_rate_kernel<false>, dim3(grid), dim3(256), 0, s, scratch, n, 1);
        else if (random_operands == 3) hipLaunchKernelGGL(mfma16_rate_kernel<true>, dim3(grid), dim3(256), 0, s, scratch, n, 1);
        else hipLaunchKernelGGL(mfma16_rate_kernel<false>, dim3(grid), dim3(256), 0, s, scratch, n, 1);
    };
    launch(iters / 8 + 1);                                           // warm the clocks
    (void)hipEventRecord(e0, s);
    launch(iters);
    (void)hipEventRecord(e1, s);
    err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess) return (int)err;
    *tflops_out = (float)((double)grid * 4 * 64.0 * iters * 32768.0 / (ms * 1e-3) * 1e-12);
    return 0;
}


// A HIP stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask): the training step's weight-gradient stream
// can be kept off the CUs the main stream's latency-critical kernels run on (model_train.OMH_WGRAD_CU_MASK, an A/B
// switch; measured in DESIGN.md 4.2).  ``cus_per_32``: how many of every 32 consecutive mask bits are set (1..32), taken
// from the low end (``high`` != 0: from the high end — the complement of a low mask of 32 - cus_per_32) — whatever the
// driver's bit -> (XCD, CU) interleave is, every XCD then contributes the same number of CUs.
extern "C" int omh_stream_create_cu_mask(int32_t cus_per_32, int32_t high, omh_stream_t* stream_out) {
    if (!stream_out || cus_per_32 < 1 || cus_per_32 > 32) return OMH_E_BADARG;
    const int words = (omh_cu_count() + 31) / 32;
    uint32_t mask[32];
    if (words > 32) return OMH_E_SHAPE;
    uint32_t w = cus_per_32 == 32 ? 0xffffffffu : ((1u << cus_per_32) - 1u);
    if (high) w <<= (32 - cus_per_32);
    for (int i = 0; i < words; ++i) mask[i] = w;
    hipStream_t s = nullptr;
    const hipError_t err = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (err != hipSuccess) return -100 - (int)err;
    *stream_out = (omh_stream_t)s;
    return 0;
}
extern "C" int omh_stream_destroy(omh_stream_t stream) {
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : OMH_E_BADARG;
}
