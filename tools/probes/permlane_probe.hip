// what v_permlane16_swap / v_permlane32_swap deliver (gfx950): prints, per lane, the two results of swap(x, x) with x = lane id
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned u = threadIdx.x;
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[threadIdx.x * 4 + 0] = a[0]; out[threadIdx.x * 4 + 1] = a[1];
    out[threadIdx.x * 4 + 2] = b[0]; out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: p16 (%2u, %2u)  p32 (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
