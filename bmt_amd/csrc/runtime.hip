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
                                                      float* __restrict__ out, int rows_per_blk, const int* __restrict__ rows_dev) {
    __shared__ float red[4][64];
    if (rows_dev != nullptr) M = min(M, *rows_dev);      // packed rows (bmt_gemm_bf16_args.rows_dev)
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

extern "C" int bmt_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, const int* rows_dev, void* stream) {
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
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, X, ldx, M, N, out, rows_per_blk, rows_dev);
    BMT_CHECK_LAUNCH("bmt_colsum");
    return BMT_OK;
}

// ---------------------------------------------------------------- small step-protocol kernels (replace ATen fills / copies / reductions)
namespace {

// zero fill, 16 bytes per thread and trip: the flat gradient arena of a step (202 MB at config[1]) in ONE launch instead of a fill per bucket
__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, int64_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
}
__global__ __launch_bounds__(256) void zero_tail_kernel(unsigned char* __restrict__ p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}

// training_loop's caption bookkeeping (epoch_loops/captioning_epoch_loops.py:130-134) in one launch of one workgroup:
//   x = caption_idx[:, :-1], y = caption_idx[:, 1:] (contiguous int64 copies), n_tokens = (y != pad_idx).sum()
__global__ __launch_bounds__(256) void caption_shift_kernel(const int64_t* __restrict__ caps, int64_t ld, int B, int T1, int64_t pad,
                                                             int64_t* __restrict__ x, int64_t* __restrict__ y, int64_t* __restrict__ n_tokens) {
    __shared__ int cnt[4];
    const int T = T1 - 1;
    int mine = 0;
    for (int i = threadIdx.x; i < B * T; i += 256) {
        const int b = i / T, t = i % T;
        const int64_t a = caps[(int64_t)b * ld + t], c = caps[(int64_t)b * ld + t + 1];
        x[i] = a;
        y[i] = c;
        mine += (c != pad) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) n_tokens[0] = (int64_t)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}

// loss = sum-KL / n_tokens and the gradient scale 1 / n_tokens the optimizer multiplies in (captioning_epoch_loops.py:135)
__global__ void loss_finish_kernel(const float* __restrict__ kl, const int64_t* __restrict__ n_tokens, float* __restrict__ loss,
                                   float* __restrict__ grad_scale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float n = (float)n_tokens[0];
        if (loss) loss[0] = kl[0] / n;
        if (grad_scale) grad_scale[0] = 1.0f / n;
    }
}

}  // namespace

extern "C" int bmt_zero(void* p, int64_t nbytes, void* stream) {
    BMT_CHECK_ARG(p && nbytes >= 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "bmt_zero: null or unaligned (16 bytes) pointer");
    if (nbytes == 0) return BMT_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n16 = nbytes / 16;
    if (n16 > 0) {
        int64_t blocks = (n16 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(p), n16);
    }
    if (nbytes % 16) hipLaunchKernelGGL(zero_tail_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<unsigned char*>(p) + n16 * 16, (int)(nbytes % 16));
    BMT_CHECK_LAUNCH("bmt_zero");
    return BMT_OK;
}

// ABI 10: host (pinned) -> device bytes on `stream` (hipMemcpyAsync): what carries a grouped launch's table image (bmt_gemm_bf16_grouped_image)
// to the device -- on a stream of the caller's choice, e.g. one that is NOT part of an ongoing capture
extern "C" int bmt_copy_h2d_async(void* dst, const void* src_host, int64_t nbytes, void* stream) {
    BMT_CHECK_ARG(dst && src_host && nbytes > 0, "bmt_copy_h2d_async: bad arguments");
    const hipError_t e = hipMemcpyAsync(dst, src_host, (size_t)nbytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
        bmt_set_error("bmt_copy_h2d_async: %s", hipGetErrorString(e));
        return BMT_EHIP;
    }
    return BMT_OK;
}

extern "C" int bmt_caption_shift(const int64_t* caption_idx, int64_t ld, int B, int T1, int64_t pad_idx, int64_t* x, int64_t* y,
                                 int64_t* n_tokens, void* stream) {
    BMT_CHECK_ARG(caption_idx && x && y && n_tokens && B > 0 && T1 > 1 && ld >= T1, "bmt_caption_shift: bad args");
    hipLaunchKernelGGL(caption_shift_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, caption_idx, ld, B, T1, pad_idx, x, y, n_tokens);
    BMT_CHECK_LAUNCH("bmt_caption_shift");
    return BMT_OK;
}

extern "C" int bmt_loss_finish(const float* kl, const int64_t* n_tokens, float* loss, float* grad_scale, void* stream) {
    BMT_CHECK_ARG(kl && n_tokens && (loss || grad_scale), "bmt_loss_finish: bad args");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kl, n_tokens, loss, grad_scale);
    BMT_CHECK_LAUNCH("bmt_loss_finish");
    return BMT_OK;
}
