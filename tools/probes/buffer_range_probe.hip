// How does gfx950 range-check a raw buffer access: does the SGPR offset take part?  (decides how the k-major stream masks
// the rows of a partial last k tile)   hipcc --offload-arch=gfx950 -O2 buffer_range_probe.hip -o buffer_range_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const float* src, float* out, float* st, const uint32_t* vo, const uint32_t* so, int n, int nr) {
    __shared__ float lds[64];
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nr, 0x00020000);
    auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, nr, 0x00020000);
    for (int c = 0; c < n; ++c) {
        uint32_t v = vo[c], s = __builtin_amdgcn_readfirstlane(so[c]);
        float r;
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(v), "s"(rs), "s"(s) : "memory");
        if (threadIdx.x == 0) out[c] = r;
        float one = 1000.f + c;
        asm volatile("buffer_store_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" :: "v"(one), "v"(v), "s"(rd), "s"(s) : "memory");
        lds[threadIdx.x] = -5.f;
        __syncthreads();
        uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)"
                     :: "v"(v), "s"(rs), "s"(s), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
        __syncthreads();
        if (threadIdx.x == 0) out[64 + c] = lds[0];
        __syncthreads();
    }
}
int main() {
    const int N = 4096, NR = 256;
    float* h = new float[N];
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *src, *out, *st; uint32_t *vo, *so;
    hipMalloc(&src, N * 4); hipMalloc(&out, 128 * 4); hipMalloc(&st, N * 4); hipMalloc(&vo, 64 * 4); hipMalloc(&so, 64 * 4);
    hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipMemset(st, 0, N * 4);
    uint32_t hv[] = {0, 252, 256, 0, 0, 128, 0, 0x80000000u, 0x80000000u, 4, 0xfffffffcu};
    uint32_t hs[] = {0, 0, 0, 256, 512, 128, 252, 0, 128, 1024, 8};
    const int n = 11;
    hipMemcpy(vo, hv, n * 4, hipMemcpyHostToDevice); hipMemcpy(so, hs, n * 4, hipMemcpyHostToDevice);
    probe<<<1, 1>>>(src, out, st, vo, so, n, NR);
    float ho[128], hst[N];
    hipMemcpy(ho, out, 128 * 4, hipMemcpyDeviceToHost); hipMemcpy(hst, st, N * 4, hipMemcpyDeviceToHost);
    printf("num_records = %d bytes; source element i holds i\n", NR);
    for (int c = 0; c < n; ++c) {
        uint32_t e = (uint32_t)(((uint64_t)hv[c] + hs[c]) & 0xffffffffu) / 4;
        printf("voffset %10u soffset %5u -> load %8.1f  lds-load %8.1f  store landed: %s\n", hv[c], hs[c], ho[c], ho[64 + c],
               (e < N && hst[e] == 1000.f + c) ? "yes" : "no");
    }
    return 0;
}
