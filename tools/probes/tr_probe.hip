#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off;   // element offset of this lane's 4-element chunk
    if (mode == 0) off = l * 4;                       // consecutive chunks
    else if (mode == 1) off = (l & 15) * 64 + (l >> 4) * 4;   // 16 rows of stride 64 elements, 4 column groups
    else off = (l & 15) * 4 + (l >> 4) * 256;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
