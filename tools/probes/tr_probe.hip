// What does ds_read_b64_tr_b16 return?  LDS holds uint16 values equal to their element index; lane l reads at
// byte address base + l*stride.  Prints the 4 values each lane gets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
__global__ void probe(uint16_t* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)((__attribute__((address_space(3))) char*)lds + lane * stride_bytes));
    union { bf16x4 b; uint16_t u[4]; } c; c.b = v;
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = c.u[e];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 16, 32, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes (= %d elements per lane row)\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    }
    return 0;
}
