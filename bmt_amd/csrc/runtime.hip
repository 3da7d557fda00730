// Library-wide plumbing of libbmt_hip.so: the thread-local error string every entry point reports through, the ABI version,
// device properties, and bmt_colsum (bias gradients of layers whose upstream gradient is only available as an fp32 tensor).
#include <stdarg.h>

#include "common.h"

// ---------------------------------------------------------------- error state (shared by all .hip files)
static thread_local char g_err[512] = "";
void bmt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* bmt_last_error(void) { return g_err; }
extern "C" int bmt_version(void) { return BMT_ABI_VERSION; }
extern "C" int bmt_device_cus(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

namespace {

// ---------------------------------------------------------------- column sums (bias gradients)
// grid: (ceil(N/64), row chunks); block 256 = 4 row-lanes x 64 columns; atomic accumulate of chunk partials.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N,
                                                      float* __restrict__ out, int rows_per_blk) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_blk, rend = min(M, rbeg + rows_per_blk);
    float s = 0.f;
    if (c < N)
        for (int r = rbeg + rl; r < rend; r += 4) s += X[(int64_t)r * ldx + c];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int bmt_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, void* stream) {
    BMT_CHECK_ARG(X && out && M >= 0 && N > 0, "bmt_colsum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) {
            bmt_set_error("bmt_colsum: memset failed");
            return BMT_EHIP;
        }
    }
    if (M == 0) return BMT_OK;
    const int rows_per_blk = 256;
    dim3 grid(bmt_cdiv(N, 64), bmt_cdiv(M, rows_per_blk));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, X, ldx, M, N, out, rows_per_blk);
    BMT_CHECK_LAUNCH("bmt_colsum");
    return BMT_OK;
}
