// Generator log-softmax and the label-smoothed KL loss (K7 of SURVEY.md 2.3).
//   log_softmax over V ~ 10k per row: model/generators.py:19
//   LabelSmoothing.forward: loss/label_smoothing.py:12-32 -- the reference materialises a dense (B*Tc, V)
//   target distribution; here each row's KL is evaluated in closed form from three numbers
//   (sum_v pred, pred[target], pred[pad]), so the loss reads pred once and writes nothing but a scalar.
// HBM-bound: one 256-thread workgroup per row, float4 sweeps when V % 4 == 0.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(float* __restrict__ x, int64_t ldx, int rows, int V) {
    __shared__ float red[4];
    float* xr = x + (int64_t)blockIdx.x * ldx;
    float m = -__builtin_huge_valf();
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, xr[c]);
    m = block_max_256(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(xr[c] - m);
    s = block_sum_256(s, red);
    const float lse = m + logf(s);
    for (int c = threadIdx.x; c < V; c += 256) xr[c] = xr[c] - lse;
}

__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ logp, int64_t ldp, const float* __restrict__ dlogp,
                                                               int64_t ldd, float* __restrict__ dlogits, int64_t ldo, int rows, int V) {
    __shared__ float red[4];
    const float* pr = logp + (int64_t)blockIdx.x * ldp;
    const float* dr = dlogp + (int64_t)blockIdx.x * ldd;
    float* orow = dlogits + (int64_t)blockIdx.x * ldo;
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += dr[c];
    s = block_sum_256(s, red);
    for (int c = threadIdx.x; c < V; c += 256) orow[c] = dr[c] - expf(pr[c]) * s;
}

// flag[0] = 1 iff the pad rows are zeroed: the reference zeroes them only when the SUM of their flat row
// indices is > 0 (label_smoothing.py:26-30), i.e. not when the only pad target sits at flat index 0.
__global__ __launch_bounds__(256) void ls_padflag_kernel(const int64_t* __restrict__ target, int rows, int64_t pad_idx, int* flag) {
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int mine = 0;
    for (int r = threadIdx.x; r < rows; r += 256)
        if (target[r] == pad_idx && r > 0) mine = 1;
    if (mine) atomicOr(&any, 1);
    __syncthreads();
    if (threadIdx.x == 0) flag[0] = any;
}

__device__ __forceinline__ float xlogx(float u) { return u > 0.f ? u * logf(u) : 0.f; }

__global__ __launch_bounds__(256) void ls_kl_rows_kernel(const float* __restrict__ pred, int64_t ldp, const int64_t* __restrict__ target,
                                                          float* __restrict__ row_loss, const int* __restrict__ flag, int rows, int V,
                                                          float smoothing, int64_t pad_idx) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int64_t t = target[r];
    if (t == pad_idx && flag[0]) {
        if (threadIdx.x == 0) row_loss[r] = 0.f;
        return;
    }
    const float* pr = pred + (int64_t)r * ldp;
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += pr[c];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float u = smoothing / (float)(V - 2);
        const float conf = 1.f - smoothing;
        const float ppad = pr[pad_idx];
        float loss;
        if (t == pad_idx) {
            // un-zeroed pad row (the index-0 quirk): scatter put conf on the pad column, which is then zeroed
            loss = (float)(V - 1) * xlogx(u) - u * (s - ppad);
        } else {
            const float pt = pr[t];
            loss = (float)(V - 2) * xlogx(u) - u * (s - pt - ppad) + xlogx(conf) - conf * pt;
        }
        row_loss[r] = loss;
    }
}

__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void ls_kl_bwd_kernel(const int64_t* __restrict__ target, float* __restrict__ dpred, int64_t ldp,
                                                         const float* __restrict__ gscale, const int* __restrict__ flag, int rows, int V,
                                                         float smoothing, int64_t pad_idx) {
    const int r = blockIdx.x;
    const int64_t t = target[r];
    const float g = gscale[0];
    const bool zero = (t == pad_idx) && flag[0];
    const float u = zero ? 0.f : -g * (smoothing / (float)(V - 2));
    const float conf = zero ? 0.f : -g * (1.f - smoothing);
    float* dr = dpred + (int64_t)r * ldp;
    for (int c = threadIdx.x; c < V; c += 256) {
        float v = u;
        if (c == t) v = conf;
        if (c == pad_idx) v = 0.f;
        dr[c] = v;
    }
}

}  // namespace

extern "C" int bmt_log_softmax_fwd(float* x, int64_t ldx, int rows, int V, void* stream) {
    BMT_CHECK_ARG(x && rows >= 0 && V > 0, "bmt_log_softmax_fwd: bad args");
    if (rows == 0) return BMT_OK;
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, V);
    BMT_CHECK_LAUNCH("bmt_log_softmax_fwd");
    return BMT_OK;
}

extern "C" int bmt_log_softmax_bwd(const float* logp, int64_t ldp, const float* dlogp, int64_t ldd, float* dlogits, int64_t ldo,
                                   int rows, int V, void* stream) {
    BMT_CHECK_ARG(logp && dlogp && dlogits && rows >= 0 && V > 0, "bmt_log_softmax_bwd: bad args");
    if (rows == 0) return BMT_OK;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logp, ldp, dlogp, ldd, dlogits, ldo, rows, V);
    BMT_CHECK_LAUNCH("bmt_log_softmax_bwd");
    return BMT_OK;
}

// row_ws layout: [rows] floats of per-row loss followed by one int flag (so >= rows + 1 elements)
extern "C" int bmt_ls_kl_fwd(const float* pred, int64_t ldp, const int64_t* target, float* loss, float* row_ws, int rows, int V,
                             float smoothing, int64_t pad_idx, void* stream) {
    BMT_CHECK_ARG(pred && target && loss && row_ws && rows > 0 && V > 2, "bmt_ls_kl_fwd: bad args");
    BMT_CHECK_ARG(pad_idx >= 0 && pad_idx < V, "bmt_ls_kl_fwd: pad_idx out of range");
    hipStream_t st = (hipStream_t)stream;
    int* flag = reinterpret_cast<int*>(row_ws + rows);
    hipLaunchKernelGGL(ls_padflag_kernel, dim3(1), dim3(256), 0, st, target, rows, pad_idx, flag);
    hipLaunchKernelGGL(ls_kl_rows_kernel, dim3(rows), dim3(256), 0, st, pred, ldp, target, row_ws, flag, rows, V, smoothing, pad_idx);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, st, row_ws, rows, loss);
    BMT_CHECK_LAUNCH("bmt_ls_kl_fwd");
    return BMT_OK;
}

// row_ws: the same workspace the forward filled (its trailing int is the pad-row flag)
extern "C" int bmt_ls_kl_bwd(const int64_t* target, float* dpred, int64_t ldp, const float* gscale_dev, const float* row_ws, int rows,
                             int V, float smoothing, int64_t pad_idx, void* stream) {
    BMT_CHECK_ARG(target && dpred && gscale_dev && row_ws && rows > 0 && V > 2, "bmt_ls_kl_bwd: bad args");
    const int* flag = reinterpret_cast<const int*>(row_ws + rows);
    hipLaunchKernelGGL(ls_kl_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, target, dpred, ldp, gscale_dev, flag, rows, V, smoothing, pad_idx);
    BMT_CHECK_LAUNCH("bmt_ls_kl_bwd");
    return BMT_OK;
}
