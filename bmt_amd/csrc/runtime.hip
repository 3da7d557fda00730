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

// MANY small column-sum reductions in one launch: out_i[c] += sum_r part_i[r * ld_i + c] (c < D_i).  The step has ~90 of them -- the second
// stage of every LayerNorm's dgamma / dbeta, the per-tile bias partials of every attention backward -- each a 4-us kernel behind a 1.5-us
// kernel boundary when launched on its own.  The items travel BY VALUE in the kernel arguments (96 x 32 bytes): nothing is read from host
// memory when the launch executes, so it can be captured in a hipGraph.  grid (sum of ceil(D_i / 256), row slices).
struct ColsumItem {
    const float* part;
    float* out;
    int rows, ld, D, blk0;      // blk0: first blockIdx.x of this item
};
struct ColsumPack {
    ColsumItem it[BMT_COLSUM_MAX_ITEMS];
    int n;
};
static_assert(sizeof(ColsumPack) <= 4000, "the item pack must fit the kernel argument buffer");
__global__ __launch_bounds__(256) void colsum_multi_kernel(const ColsumPack pk) {
    int i = 0;
    while (i + 1 < pk.n && pk.it[i + 1].blk0 <= (int)blockIdx.x) ++i;      // (uniform: scalar loads from the argument segment)
    const float* part = pk.it[i].part;
    const int rows = pk.it[i].rows, ld = pk.it[i].ld, D = pk.it[i].D;
    const int c = ((int)blockIdx.x - pk.it[i].blk0) * 256 + (int)threadIdx.x;
    if (c >= D) return;
    float s0 = 0.f, s1 = 0.f;
    int r = blockIdx.y;
    for (; r + (int)gridDim.y < rows; r += 2 * gridDim.y) {
        s0 += part[(int64_t)r * ld + c];
        s1 += part[(int64_t)(r + gridDim.y) * ld + c];
    }
    if (r < rows) s0 += part[(int64_t)r * ld + c];
    atomicAdd(pk.it[i].out + c, s0 + s1);
}

// many small fp32 copies in one launch (the concatenated biases of the fused projection groups after an optimizer step): same by-value pack
struct CopyPack {
    struct { const float* src; float* dst; int n, blk0; } it[BMT_COLSUM_MAX_ITEMS];
    int n;
};
static_assert(sizeof(CopyPack) <= 4000, "the item pack must fit the kernel argument buffer");
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyPack pk) {
    int i = 0;
    while (i + 1 < pk.n && pk.it[i + 1].blk0 <= (int)blockIdx.x) ++i;
    const int c = ((int)blockIdx.x - pk.it[i].blk0) * 256 + (int)threadIdx.x;
    if (c < pk.it[i].n) pk.it[i].dst[c] = pk.it[i].src[c];
}

}  // namespace

extern "C" int bmt_copy_multi(const bmt_copy_item* items, int n, void* stream) {
    BMT_CHECK_ARG(items && n > 0, "bmt_copy_multi: bad args");
    for (int base = 0; base < n; base += BMT_COLSUM_MAX_ITEMS) {
        CopyPack pk;
        pk.n = n - base < BMT_COLSUM_MAX_ITEMS ? n - base : BMT_COLSUM_MAX_ITEMS;
        int blk = 0;
        for (int i = 0; i < pk.n; ++i) {
            const bmt_copy_item& a = items[base + i];
            BMT_CHECK_ARG(a.src && a.dst && a.n > 0, "bmt_copy_multi: bad item %d", base + i);
            pk.it[i].src = a.src; pk.it[i].dst = a.dst; pk.it[i].n = (int)a.n; pk.it[i].blk0 = blk;
            blk += bmt_cdiv(a.n, 256);
        }
        hipLaunchKernelGGL(copy_multi_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, pk);
        BMT_CHECK_LAUNCH("bmt_copy_multi");
    }
    return BMT_OK;
}

extern "C" int bmt_colsum_multi(const bmt_colsum_item* items, int n, void* stream) {
    BMT_CHECK_ARG(items && n > 0, "bmt_colsum_multi: bad args");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += BMT_COLSUM_MAX_ITEMS) {
        ColsumPack pk;
        pk.n = n - base < BMT_COLSUM_MAX_ITEMS ? n - base : BMT_COLSUM_MAX_ITEMS;
        int blk = 0, maxrows = 1;
        for (int i = 0; i < pk.n; ++i) {
            const bmt_colsum_item& a = items[base + i];
            BMT_CHECK_ARG(a.part && a.out && a.rows >= 0 && a.D > 0 && a.ld >= a.D, "bmt_colsum_multi: bad item %d", base + i);
            pk.it[i] = ColsumItem{a.part, a.out, a.rows, (int)a.ld, a.D, blk};
            blk += bmt_cdiv(a.D, 256);
            if (a.rows > maxrows) maxrows = a.rows;
        }
        const int slices = maxrows < 16 ? maxrows : 16;
        hipLaunchKernelGGL(colsum_multi_kernel, dim3(blk, slices), dim3(256), 0, st, pk);
        BMT_CHECK_LAUNCH("bmt_colsum_multi");
    }
    return BMT_OK;
}

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
