// MFMA issue-rate probe for gfx950.  grid x 256 threads (4 waves per workgroup, __launch_bounds__(256, WPS): WPS = waves
// per SIMD the register budget allows), NACC independent accumulators used round-robin, ITER x 64
// v_mfma_f32_32x32x16_bf16 written as asm (the compiler neither reorders nor pads them), optionally NV v_add_f32 after
// each MFMA.  Prints ns per MFMA per wave and the aggregate rate.  Build: hipcc --offload-arch=gfx950 -O3 <this> -o probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC, int NV, bool AGPR, int WPS, bool RND = false, int NE = 0, int NL = 0>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, int salt) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b, ra[8], rb[8];
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e + salt); b[e] = (__bf16)(float)(e + salt); }
    if (RND)                                                       // operands that toggle like real data: 8 sets of hashed N(0,1)-ish values
        for (int i = 0; i < 8; ++i)
            for (int e = 0; e < 8; ++e) {
                uint32_t h = (threadIdx.x * 2654435761u) ^ ((i * 8 + e + salt) * 40503u + blockIdx.x * 977u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                uint32_t g = h * 3266489917u; g ^= g >> 16;
                ra[i][e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 16384.0f));
                rb[i][e] = (__bf16)(((int)(g & 0xffff) - 32768) * (1.0f / 16384.0f));
            }
    float x0 = salt, x1 = salt + 1, x2 = salt + 2, x3 = salt + 3;
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (threadIdx.x & 255) * 16;
    typedef __attribute__((ext_vector_type(4))) uint32_t u4; u4 lr = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (RND) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j % NACC]) : "v"(ra[j & 7]), "v"(rb[(j >> 3) & 7]));
            else if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j % NACC]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j % NACC]) : "v"(a), "v"(b));
            if (NV >= 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x0));
            if (NV >= 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x1));
            if (NV >= 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x2));
            if (NV >= 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x3));
            if (NE >= 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x2));
            if (NE >= 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x3));
            if (NL >= 1) asm volatile("ds_read_b128 %0, %1" : "=v"(lr) : "v"(la));
            if (NL >= 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(lr) : "v"(la));
            if (NL >= 1 && (j & 3) == 3) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            if (NV >= 6) { asm volatile("v_add_f32 %0, %0, %0" : "+v"(x0)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(x1)); }
        }
    }
    float s = x0 + x1 + x2 + x3 + (float)lr[0];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int NV, bool AGPR, int WPS, bool RND = false, int NE = 0, int NL = 0> void run(float* out, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    k<NACC, NV, AGPR, WPS, RND, NE, NL><<<grid, 256>>>(out, 10, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC, NV, AGPR, WPS, RND, NE, NL><<<grid, 256>>>(out, iters, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 / (iters * 64.0);
    printf("%s grid %4d  acc %d in %s, %d VALU + %d v_exp + %d ds_read_b128 per MFMA, %d wave(s)/SIMD allowed: %.1f ns per MFMA per wave (%.1f cycles at 2.4 GHz); %.0f TFLOP/s\n", RND ? "random operands  " : "constant operands", grid,
           NACC, AGPR ? "AGPR" : "VGPR", NV, NE, NL, WPS, per * 1e9, per * 2.4e9, grid * 4 * 32768.0 / per * 1e-12);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    for (int grid : {64, 256}) {
        run<1, 0, true, 1>(out, grid); run<2, 0, true, 1>(out, grid); run<4, 0, true, 1>(out, grid); run<8, 0, true, 1>(out, grid);
        run<4, 0, false, 1>(out, grid); run<4, 2, true, 1>(out, grid); run<4, 4, true, 1>(out, grid); run<4, 6, true, 1>(out, grid);
    }
    // (operands that toggle like data: csrc/probes.hip = omh_probe_mfma_tflops, whose loop is 64 bare MFMAs; the RND variant of THIS
    //  file compiles to extra register moves — 20 ns per MFMA even on 64 CUs — and is not run)
    // issue cost of what rides between the MFMAs of the streams (one wave per SIMD issues everything serially)
    for (int grid : {64}) {
        run<8, 0, true, 1, false, 1, 0>(out, grid); run<8, 0, true, 1, false, 2, 0>(out, grid); run<8, 2, true, 1, false, 1, 0>(out, grid);
        run<8, 0, true, 1, false, 0, 1>(out, grid); run<8, 0, true, 1, false, 0, 2>(out, grid); run<8, 2, true, 1, false, 1, 1>(out, grid);
        run<8, 4, true, 1, false, 1, 1>(out, grid);
    }
    for (int grid : {128, 512}) { run<4, 0, true, 2>(out, grid); run<4, 4, true, 2>(out, grid); run<2, 0, true, 2>(out, grid); }
    return 0;
}
