// LDS read-rate probe for gfx950: grid x 256 threads (4 waves per CU when grid <= 256), each wave issues ITER x 64
// reads (ds_read_b128, ds_read_b64, or ds_read_b64_tr_b16) of its own 4 KiB region, linear 16 / 8 bytes per lane, as asm.
// Prints ns per read instruction per wave and the bytes per clock per CU at an assumed 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 <this> -o lds_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
template <int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 64 * WAVES) ((uint32_t*)smem)[i] = i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192;
    const uint32_t addr = base + lane * (KIND == 0 ? 16 : 8);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (KIND == 0) { u32x4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"((j & 3) * 1024)); if (j == 63) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += r; } }
            else if (KIND == 1) { u32x2 r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"((j & 7) * 512)); if (j == 63) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc[0] += r[0]; } }
            else { u32x2 r; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"((j & 7) * 512)); if (j == 63) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc[0] += r[0]; } }
        }
    }
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = (float)(acc[0] + acc[1]);
}
template <int KIND, int WAVES> void run(float* out, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    k<KIND, WAVES><<<grid, 64 * WAVES>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND, WAVES><<<grid, 64 * WAVES>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 / (iters * 64.0);
    const int bytes = KIND == 0 ? 1024 : 512;
    printf("grid %3d, %d wave(s) per workgroup, %s: %.2f ns per read per wave; %.1f bytes per clock per CU at 2.4 GHz\n", grid, WAVES,
           KIND == 0 ? "ds_read_b128      " : KIND == 1 ? "ds_read_b64       " : "ds_read_b64_tr_b16", per * 1e9, WAVES * bytes / (per * 2.4e9));
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 512 * 4);
    for (int grid : {64, 256}) {
        run<0, 1>(out, grid); run<0, 4>(out, grid); run<0, 8>(out, grid);
        run<1, 1>(out, grid); run<1, 4>(out, grid);
        run<2, 1>(out, grid); run<2, 4>(out, grid); run<2, 8>(out, grid);
    }
    return 0;
}
