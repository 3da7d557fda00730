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

__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ x, int64_t ldx, int rows, int V, float* __restrict__ out) {
    __shared__ float red[4];
    const float* xr = x + (int64_t)blockIdx.x * ldx;
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += xr[c];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
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


// ---------------------------------------------------------------- K7 in one pass each way
// Forward: log-softmax of a row with the row kept in REGISTERS (V <= 256 * 4 * NV: one HBM read, one write) and, from the same pass,
// rowsum[r] = sum_c logp[r][c] -- all the label-smoothed KL of that row needs beside two gathers (ls_kl_stats_kernel).  The unfused
// pair read the (B*Tc, V) tensor three times for the softmax (scalar loads) and once more for the KL.
template <int NV>
__global__ __launch_bounds__(256) void log_softmax_stats_kernel(float* __restrict__ x, int64_t ldx, int rows, int V, float* __restrict__ rowsum) {
    __shared__ float red[4];
    float* xr = x + (int64_t)blockIdx.x * ldx;
    float4 v[NV];
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (threadIdx.x + 256 * i) * 4;
        if (c < V) {       // (V % 4 == 0: whole float4 groups)
            v[i] = *reinterpret_cast<const float4*>(xr + c);
            m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
    m = block_max_256(m, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (threadIdx.x + 256 * i) * 4;
        if (c < V) s += (expf(v[i].x - m) + expf(v[i].y - m)) + (expf(v[i].z - m) + expf(v[i].w - m));
    }
    s = block_sum_256(s, red);
    const float lse = m + logf(s);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (threadIdx.x + 256 * i) * 4;
        if (c < V) {
            const float4 o = make_float4(v[i].x - lse, v[i].y - lse, v[i].z - lse, v[i].w - lse);
            *reinterpret_cast<float4*>(xr + c) = o;
            t += (o.x + o.y) + (o.z + o.w);
        }
    }
    t = block_sum_256(t, red);
    if (threadIdx.x == 0 && rowsum) rowsum[blockIdx.x] = t;
}

// the whole LabelSmoothing.forward from the row sums: ONE workgroup -- pad-row flag (the flat-index-0 quirk), the rows' closed-form KL, their
// sum.  row_ws: [rows] per-row losses + the int flag behind them (the layout bmt_ls_kl_fwd leaves, read by the backward kernels)
__global__ __launch_bounds__(256) void ls_kl_stats_kernel(const float* __restrict__ pred, int64_t ldp, const int64_t* __restrict__ target,
                                                           const float* __restrict__ rowsum, float* __restrict__ loss, float* __restrict__ row_ws,
                                                           int rows, int V, float smoothing, int64_t pad_idx) {
    __shared__ float red[4];
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int mine = 0;
    for (int r = threadIdx.x; r < rows; r += 256)
        if (target[r] == pad_idx && r > 0) mine = 1;
    if (mine) atomicOr(&any, 1);
    __syncthreads();
    const int flag = any;
    const float u = smoothing / (float)(V - 2), conf = 1.f - smoothing;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const int64_t t = target[r];
        float l = 0.f;
        if (!(t == pad_idx && flag)) {
            const float* pr = pred + (int64_t)r * ldp;
            const float s = rowsum[r], ppad = pr[pad_idx];
            if (t == pad_idx) l = (float)(V - 1) * xlogx(u) - u * (s - ppad);
            else {
                const float pt = pr[t];
                l = (float)(V - 2) * xlogx(u) - u * (s - pt - ppad) + xlogx(conf) - conf * pt;
            }
        }
        row_ws[r] = l;
        acc += l;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        loss[0] = acc;
        reinterpret_cast<int*>(row_ws + rows)[0] = flag;
    }
}

// Backward of {Linear -> log_softmax -> LabelSmoothing sum-KL} w.r.t. the LOGITS in one pass over the saved log-probabilities:
//     dlogits[r][c] = g * (exp(logp[r][c]) * R_r - dist[r][c])
// dist = the smoothed target distribution (never materialised; R_r its row sum: 1, (V - 1) u for the un-zeroed pad row of the quirk, 0 for a
// zeroed pad row).  Written straight as the bf16 operand plane the generator's dX / dW products read, with the column sums -- the
// generator's bias gradient -- from the same registers.  Replaces ls_kl_bwd -> log_softmax_bwd -> the plane conversion (+ its column
// sums): 38 + 38 + 38 MB written and 38 + 76 + 38 MB read become 38 MB read and 19 MB written.
// grid (ceil(pcols / 512), ceil(rows / RB)); a thread owns two adjacent columns over RB rows
constexpr int GENB_RB = 32;
template <bool VEC2>
__global__ __launch_bounds__(256) void gen_lskl_bwd_kernel(const float* __restrict__ logp, int64_t ldp, const int64_t* __restrict__ target,
                                                            const float* __restrict__ row_ws, const float* __restrict__ gscale, int rows, int V,
                                                            int pcols, float smoothing, int64_t pad_idx, uint16_t* __restrict__ hi, int64_t ldh,
                                                            float* __restrict__ colsum) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c0 >= pcols) return;
    const int flag = reinterpret_cast<const int*>(row_ws + rows)[0];
    const float g = gscale[0];
    const float u = smoothing / (float)(V - 2), conf = 1.f - smoothing;
    const int r0 = blockIdx.y * GENB_RB, r1 = min(rows, r0 + GENB_RB);
    float s0 = 0.f, s1 = 0.f;
    for (int r = r0; r < r1; ++r) {
        const int64_t t = target[r];
        float o0 = 0.f, o1 = 0.f;
        if (c0 < V && !(t == pad_idx && flag)) {
            const float R = (t == pad_idx) ? (float)(V - 1) * u : 1.f;
            float2 lp;
            if (VEC2 && c0 + 1 < V) lp = *reinterpret_cast<const float2*>(logp + (int64_t)r * ldp + c0);
            else lp = make_float2(logp[(int64_t)r * ldp + c0], (c0 + 1 < V) ? logp[(int64_t)r * ldp + c0 + 1] : 0.f);
            float d0 = (c0 == pad_idx) ? 0.f : (c0 == t ? conf : u);
            float d1 = (c0 + 1 == pad_idx) ? 0.f : (c0 + 1 == t ? conf : u);
            o0 = g * (expf(lp.x) * R - d0);
            o1 = (c0 + 1 < V) ? g * (expf(lp.y) * R - d1) : 0.f;
        }
        // the plane holds bf16(o); the bias gradient is the sum of the values the products will see
        const uint32_t w = pack_bf2(o0, o1);
        *reinterpret_cast<uint32_t*>(hi + (int64_t)r * ldh + c0) = w;
        s0 += __uint_as_float(w << 16);
        s1 += __uint_as_float(w & 0xffff0000u);
    }
    if (colsum) {
        if (c0 < V) atomicAdd(colsum + c0, s0);
        if (c0 + 1 < V) atomicAdd(colsum + c0 + 1, s1);
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


// log_softmax in place + rowsum[r] = sum_c logp[r][c] (optional), one HBM read and one write per element where a row fits the
// registers of a workgroup (V % 4 == 0, V <= 16384, 16-byte aligned rows); otherwise the three-sweep kernel + a row-sum pass
extern "C" int bmt_log_softmax_fwd_stats(float* x, int64_t ldx, int rows, int V, float* rowsum, void* stream) {
    BMT_CHECK_ARG(x && rows >= 0 && V > 0, "bmt_log_softmax_fwd_stats: bad args");
    if (rows == 0) return BMT_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (V % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && V <= 16384;
    if (!vec) {
        hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3(rows), dim3(256), 0, st, x, ldx, rows, V);
        if (rowsum) hipLaunchKernelGGL(row_sum_kernel, dim3(rows), dim3(256), 0, st, x, ldx, rows, V, rowsum);
        BMT_CHECK_LAUNCH("bmt_log_softmax_fwd_stats");
        return BMT_OK;
    }
    const int nv = bmt_cdiv(V, 1024);
#define BMT_LS(NV) hipLaunchKernelGGL(log_softmax_stats_kernel<NV>, dim3(rows), dim3(256), 0, st, x, ldx, rows, V, rowsum)
    if (nv <= 1) BMT_LS(1);
    else if (nv <= 2) BMT_LS(2);
    else if (nv <= 4) BMT_LS(4);
    else if (nv <= 10) BMT_LS(10);
    else BMT_LS(16);
#undef BMT_LS
    BMT_CHECK_LAUNCH("bmt_log_softmax_fwd_stats");
    return BMT_OK;
}

// LabelSmoothing.forward from the row sums of bmt_log_softmax_fwd_stats: one launch; row_ws as bmt_ls_kl_fwd ([rows] + the flag)
extern "C" int bmt_ls_kl_fwd_stats(const float* pred, int64_t ldp, const int64_t* target, const float* rowsum, float* loss, float* row_ws,
                                   int rows, int V, float smoothing, int64_t pad_idx, void* stream) {
    BMT_CHECK_ARG(pred && target && rowsum && loss && row_ws && rows > 0 && V > 2, "bmt_ls_kl_fwd_stats: bad args");
    BMT_CHECK_ARG(pad_idx >= 0 && pad_idx < V, "bmt_ls_kl_fwd_stats: pad_idx out of range");
    hipLaunchKernelGGL(ls_kl_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, ldp, target, rowsum, loss, row_ws, rows, V, smoothing,
                       pad_idx);
    BMT_CHECK_LAUNCH("bmt_ls_kl_fwd_stats");
    return BMT_OK;
}

// d(sum-KL * g) / d(logits) as the bf16 operand plane [rows][ldh] (columns [V, pcols) zero) + its column sums added into colsum[V]
// (optional); logp: the saved log-probabilities, row_ws: the forward's workspace (pad-row flag), gscale_dev: the loss's upstream gradient
extern "C" int bmt_gen_lskl_bwd(const float* logp, int64_t ldp, const int64_t* target, const float* row_ws, const float* gscale_dev, int rows,
                                int V, float smoothing, int64_t pad_idx, uint16_t* hi, int64_t ldh, float* colsum, void* stream) {
    BMT_CHECK_ARG(logp && target && row_ws && gscale_dev && hi && rows > 0 && V > 2, "bmt_gen_lskl_bwd: bad args");
    BMT_CHECK_ARG(pad_idx >= 0 && pad_idx < V, "bmt_gen_lskl_bwd: pad_idx out of range");
    const int pcols = (V + 63) / 64 * 64;
    BMT_CHECK_ARG(ldh >= pcols && ldh % 2 == 0 && ((reinterpret_cast<uintptr_t>(hi) & 3) == 0),
                  "bmt_gen_lskl_bwd: plane row stride %lld < %d (or odd), or a plane that is not 4-byte aligned", (long long)ldh, pcols);
    const bool vec2 = ldp % 2 == 0 && ((reinterpret_cast<uintptr_t>(logp) & 7) == 0);      // 8-byte loads of two adjacent log-probabilities
    const dim3 grid(bmt_cdiv(pcols, 512), bmt_cdiv(rows, GENB_RB));
    if (vec2) hipLaunchKernelGGL(gen_lskl_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, logp, ldp, target, row_ws, gscale_dev, rows, V, pcols,
                                 smoothing, pad_idx, hi, ldh, colsum);
    else hipLaunchKernelGGL(gen_lskl_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, logp, ldp, target, row_ws, gscale_dev, rows, V, pcols,
                            smoothing, pad_idx, hi, ldh, colsum);
    BMT_CHECK_LAUNCH("bmt_gen_lskl_bwd");
    return BMT_OK;
}
