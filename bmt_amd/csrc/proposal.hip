// Proposal-generator head post-processing (K10 of SURVEY.md 2.3): YOLO-style target assignment, decode and loss.
//   make_targets          model/proposal_generator.py:389-448  (masks / targets bit-exact; tiou_vectorized
//                         utilities/proposal_utils.py:11-57 restated with explicit, non-contracted fp32 ops)
//   decode + loss         model/proposal_generator.py:281-335  (sigmoid / exp / grid add, masked MSE + BCE means)
// The Conv1d stacks themselves run on the MFMA GEMM (bmt_conv1d in gemm.hip).  Everything here is elementwise /
// gather-scatter and HBM-bound: coalesced over the (B,S,3A) head output, block reduction + one atomic per block.
#include <stdlib.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void targets_init_kernel(uint8_t* obj, uint8_t* noobj, float* tx, float* tw, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        obj[i] = 0; noobj[i] = 1; tx[i] = 0.f; tw[i] = 0.f;
    }
}

// IoU of two centre-less segments (0, a) and (0, g) exactly as tiou_vectorized computes it in fp32
__device__ __forceinline__ float tiou_len(float a, float g) {
    const float s1 = __fsub_rn(0.f, __fdiv_rn(a, 2.f)), e1 = __fadd_rn(0.f, __fdiv_rn(a, 2.f));
    const float s2 = __fsub_rn(0.f, __fdiv_rn(g, 2.f)), e2 = __fadd_rn(0.f, __fdiv_rn(g, 2.f));
    const float inter = fmaxf(__fsub_rn(fminf(e1, e2), fmaxf(s1, s2)), 0.f);
    float uni = __fsub_rn(__fadd_rn(__fsub_rn(e1, s1), __fsub_rn(e2, s2)), inter);
    uni = fminf(__fsub_rn(fmaxf(e1, e2), fminf(s1, s2)), uni);
    return __fdiv_rn(inter, __fadd_rn(uni, 1e-8f));
}

// One workgroup.  Phase 1 (parallel over targets): best anchor, cell, target values.  Phase 2 (thread 0, in target
// order): the scatter, so duplicate (b, anchor, cell) assignments resolve "last write wins" like CPU index_put.
__global__ __launch_bounds__(256) void make_targets_kernel(const float* __restrict__ targets, int n, const float* __restrict__ anchors,
                                                            int A, int B, int G, float stride, uint8_t* obj, uint8_t* noobj, float* tx,
                                                            float* tw) {
    for (int base = 0; base < n; base += 256) {
        __shared__ int s_idx[256];
        __shared__ float s_x[256], s_w[256];
        const int i = base + threadIdx.x;
        if (i < n) {
            const int b = (int)targets[i * 4 + 0];
            const float gx = __fdiv_rn(targets[i * 4 + 1], stride);
            const float gw = __fdiv_rn(targets[i * 4 + 2], stride);
            int best = 0;
            float best_iou = -1.f;
            for (int a = 0; a < A; ++a) {
                const float iou = tiou_len(anchors[a], gw);
                if (iou > best_iou) { best_iou = iou; best = a; }   // first maximum wins, as torch.max(dim=0) on CPU
            }
            int cell = (int)gx;   // truncation toward zero == .long()
            cell = cell < 0 ? 0 : (cell > G - 1 ? G - 1 : cell);
            s_idx[threadIdx.x] = (b >= 0 && b < B) ? ((b * A + best) * G + cell) : -1;
            s_x[threadIdx.x] = __fsub_rn(gx, floorf(gx));
            s_w[threadIdx.x] = logf(__fadd_rn(__fdiv_rn(gw, anchors[best]), 1e-16f));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = min(256, n - base);
            for (int j = 0; j < m; ++j) {
                const int idx = s_idx[j];
                if (idx < 0) continue;
                obj[idx] = 1; noobj[idx] = 0; tx[idx] = s_x[j]; tw[idx] = s_w[j];
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// x: [B,S,A*3]; preds: [B, A*S, 3]; sums[0..5] += {sq_x, sq_w, bce_obj, bce_noobj, n_obj, n_noobj}
__global__ __launch_bounds__(256) void decode_loss_kernel(const float* __restrict__ x, const float* __restrict__ anchors, int B, int S, int A,
                                                           float stride, const uint8_t* __restrict__ obj, const uint8_t* __restrict__ noobj,
                                                           const float* __restrict__ tx, const float* __restrict__ tw,
                                                           float* __restrict__ preds, float* __restrict__ sums, int64_t pred_bs) {
    __shared__ float red[4];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t total = (int64_t)B * S * A;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates (b, s, a) in the memory order of x
        const int a = (int)(i % A);
        const int64_t bs = i / A;
        const int s = (int)(bs % S), b = (int)(bs / S);
        const float c = x[i * 3 + 0], l = x[i * 3 + 1], o = x[i * 3 + 2];
        const float sc = sigmoidf_(c), so = sigmoidf_(o);
        const int64_t pidx = ((int64_t)b * A + a) * S + s;   // (B, A, S) order of predictions / masks
        float* pp = preds + (int64_t)b * pred_bs + ((int64_t)a * S + s) * 3;      // (pred_bs: a head's slice of the generator's prediction buffer)
        pp[0] = (sc + (float)s) * stride;
        pp[1] = (anchors[a] * expf(l)) * stride;
        pp[2] = so;
        if (obj) {
            if (obj[pidx]) {
                const float dx = sc - tx[pidx], dw = l - tw[pidx];
                acc[0] += dx * dx; acc[1] += dw * dw;
                acc[2] += -fmaxf(logf(so), -100.f);
                acc[4] += 1.f;
            }
            if (noobj[pidx]) {
                acc[3] += -fmaxf(logf(1.f - so), -100.f);
                acc[5] += 1.f;
            }
        }
    }
    if (obj) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float v = block_sum_256(acc[k], red);
            if (threadIdx.x == 0 && v != 0.f) atomicAdd(sums + k, v);
        }
    }
}

// The same two kernels with both sides coalesced (ABI 5 default): the head output x is (B, S, A, 3) -- anchors fastest -- while predictions,
// masks and targets are (B, A, S) -- positions fastest: the element-per-thread kernels above scatter 12-byte prediction stores and
// single-byte mask loads S elements apart (81 us for 29 MB at configs[3]: 0.7 TB/s).  Here a workgroup owns 32 positions of one video:
// the (32 x A x 3) block of x goes through LDS (row stride padded to an odd number of words), the loop runs positions-fastest.
constexpr int PROP_TS = 32;
__global__ __launch_bounds__(256) void decode_loss_tiled_kernel(const float* __restrict__ x, const float* __restrict__ anchors, int B, int S, int A,
                                                                 float stride, const uint8_t* __restrict__ obj, const uint8_t* __restrict__ noobj,
                                                                 const float* __restrict__ tx, const float* __restrict__ tw,
                                                                 float* __restrict__ preds, float* __restrict__ sums, int64_t pred_bs) {
    extern __shared__ float xs[];                 // [PROP_TS][A * 3 + 1]
    __shared__ float red[4];
    const int b = blockIdx.y, s0 = blockIdx.x * PROP_TS, W = A * 3, WP = W + 1;
    const int ns = min(PROP_TS, S - s0);
    const float* src = x + ((int64_t)b * S + s0) * W;
    for (int i = threadIdx.x; i < ns * W; i += 256) xs[(i / W) * WP + i % W] = src[i];
    __syncthreads();
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = threadIdx.x; j < A * PROP_TS; j += 256) {
        const int a = j / PROP_TS, sl = j % PROP_TS;
        if (sl >= ns) continue;
        const int s = s0 + sl;
        const float c = xs[sl * WP + a * 3 + 0], l = xs[sl * WP + a * 3 + 1], o = xs[sl * WP + a * 3 + 2];
        const float sc = sigmoidf_(c), so = sigmoidf_(o);
        const int64_t pidx = ((int64_t)b * A + a) * S + s;
        float* pp = preds + (int64_t)b * pred_bs + ((int64_t)a * S + s) * 3;
        pp[0] = (sc + (float)s) * stride;
        pp[1] = (anchors[a] * expf(l)) * stride;
        pp[2] = so;
        if (obj) {
            if (obj[pidx]) {
                const float dx = sc - tx[pidx], dw = l - tw[pidx];
                acc[0] += dx * dx; acc[1] += dw * dw;
                acc[2] += -fmaxf(logf(so), -100.f);
                acc[4] += 1.f;
            }
            if (noobj[pidx]) {
                acc[3] += -fmaxf(logf(1.f - so), -100.f);
                acc[5] += 1.f;
            }
        }
    }
    if (obj) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float v = block_sum_256(acc[k], red);
            if (threadIdx.x == 0 && v != 0.f) atomicAdd(sums + k, v);
        }
    }
}

__global__ __launch_bounds__(256) void loss_bwd_tiled_kernel(const float* __restrict__ x, int B, int S, int A, const uint8_t* __restrict__ obj,
                                                              const uint8_t* __restrict__ noobj, const float* __restrict__ tx,
                                                              const float* __restrict__ tw, const float* __restrict__ sums, float obj_coeff,
                                                              float noobj_coeff, const float* __restrict__ gscale, float* __restrict__ dx) {
    extern __shared__ float xs[];                 // [PROP_TS][A * 3 + 1]: x in, dx out
    const int b = blockIdx.y, s0 = blockIdx.x * PROP_TS, W = A * 3, WP = W + 1;
    const int ns = min(PROP_TS, S - s0);
    const float g = gscale[0];
    const float inv_obj = 1.f / sums[4], inv_noobj = 1.f / sums[5];
    const float* src = x + ((int64_t)b * S + s0) * W;
    for (int i = threadIdx.x; i < ns * W; i += 256) xs[(i / W) * WP + i % W] = src[i];
    __syncthreads();
    for (int j = threadIdx.x; j < A * PROP_TS; j += 256) {
        const int a = j / PROP_TS, sl = j % PROP_TS;
        if (sl >= ns) continue;
        const int64_t pidx = ((int64_t)b * A + a) * S + s0 + sl;
        float* e = xs + sl * WP + a * 3;
        float dc = 0.f, dl = 0.f, dob = 0.f;
        const float so = sigmoidf_(e[2]);
        if (obj[pidx]) {
            const float sc = sigmoidf_(e[0]);
            dc = 2.f * (sc - tx[pidx]) * inv_obj * sc * (1.f - sc);
            dl = 2.f * (e[1] - tw[pidx]) * inv_obj;
            if (logf(so) > -100.f) dob += obj_coeff * inv_obj * (-(1.f - so));
        }
        if (noobj[pidx]) {
            if (logf(1.f - so) > -100.f) dob += noobj_coeff * inv_noobj * so;
        }
        e[0] = dc * g; e[1] = dl * g; e[2] = dob * g;          // (each element of the block belongs to exactly one thread)
    }
    __syncthreads();
    float* dst = dx + ((int64_t)b * S + s0) * W;
    for (int i = threadIdx.x; i < ns * W; i += 256) dst[i] = xs[(i / W) * WP + i % W];
}

__global__ void loss_finalize_kernel(const float* __restrict__ sums, float obj_coeff, float noobj_coeff, float* __restrict__ losses) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float lx = sums[0] / sums[4], lw = sums[1] / sums[4], lo = sums[2] / sums[4], ln = sums[3] / sums[5];
        losses[0] = lx; losses[1] = lw; losses[2] = lo; losses[3] = ln;
        losses[4] = (lx + lw) + (obj_coeff * lo + noobj_coeff * ln);
    }
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ x, int B, int S, int A, const uint8_t* __restrict__ obj,
                                                        const uint8_t* __restrict__ noobj, const float* __restrict__ tx,
                                                        const float* __restrict__ tw, const float* __restrict__ sums, float obj_coeff,
                                                        float noobj_coeff, const float* __restrict__ gscale, float* __restrict__ dx) {
    const float g = gscale[0];
    const float inv_obj = 1.f / sums[4], inv_noobj = 1.f / sums[5];
    const int64_t total = (int64_t)B * S * A;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int a = (int)(i % A);
        const int64_t bs = i / A;
        const int s = (int)(bs % S), b = (int)(bs / S);
        const int64_t pidx = ((int64_t)b * A + a) * S + s;
        float dc = 0.f, dl = 0.f, dob = 0.f;
        const float o = x[i * 3 + 2];
        const float so = sigmoidf_(o);
        if (obj[pidx]) {
            const float c = x[i * 3 + 0], l = x[i * 3 + 1];
            const float sc = sigmoidf_(c);
            dc = 2.f * (sc - tx[pidx]) * inv_obj * sc * (1.f - sc);
            dl = 2.f * (l - tw[pidx]) * inv_obj;
            // d(-log so)/do = -(1 - so); clamped branch (log so <= -100) has zero gradient
            if (logf(so) > -100.f) dob += obj_coeff * inv_obj * (-(1.f - so));
        }
        if (noobj[pidx]) {
            if (logf(1.f - so) > -100.f) dob += noobj_coeff * inv_noobj * so;
        }
        dx[i * 3 + 0] = dc * g; dx[i * 3 + 1] = dl * g; dx[i * 3 + 2] = dob * g;
    }
}

inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int bmt_targets_init(uint8_t* obj, uint8_t* noobj, float* tx, float* tw, int64_t n, void* stream) {
    BMT_CHECK_ARG(obj && noobj && tx && tw && n > 0, "bmt_targets_init: bad args");
    hipLaunchKernelGGL(targets_init_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, obj, noobj, tx, tw, n);
    BMT_CHECK_LAUNCH("bmt_targets_init");
    return BMT_OK;
}

extern "C" int bmt_make_targets(const float* targets, int n, const float* anchors, int A, int B, int G, float stride, uint8_t* obj,
                                uint8_t* noobj, float* tx, float* tw, void* stream) {
    BMT_CHECK_ARG(targets && anchors && obj && noobj && tx && tw && n >= 0 && A > 0 && B > 0 && G > 0, "bmt_make_targets: bad args");
    if (n == 0) return BMT_OK;
    hipLaunchKernelGGL(make_targets_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, targets, n, anchors, A, B, G, stride, obj, noobj, tx, tw);
    BMT_CHECK_LAUNCH("bmt_make_targets");
    return BMT_OK;
}

extern "C" int bmt_prop_decode_loss(const float* x, const float* anchors, int B, int S, int A, float stride, const uint8_t* obj,
                                    const uint8_t* noobj, const float* tx, const float* tw, float* preds, float* loss_ws, void* stream) {
    return bmt_prop_decode_loss2(x, anchors, B, S, A, stride, obj, noobj, tx, tw, preds, (int64_t)A * S * 3, loss_ws, 0, stream);
}

extern "C" int bmt_prop_decode_loss2(const float* x, const float* anchors, int B, int S, int A, float stride, const uint8_t* obj,
                                     const uint8_t* noobj, const float* tx, const float* tw, float* preds, int64_t pred_bs, float* loss_ws,
                                     int ws_zeroed, void* stream) {
    BMT_CHECK_ARG(x && anchors && preds && B > 0 && S > 0 && A > 0 && pred_bs >= (int64_t)A * S * 3, "bmt_prop_decode_loss: bad args");
    BMT_CHECK_ARG(!obj || (noobj && tx && tw && loss_ws), "bmt_prop_decode_loss: targets given without all of noobj/tx/tw/loss_ws");
    hipStream_t st = (hipStream_t)stream;
    if (obj && !ws_zeroed && hipMemsetAsync(loss_ws, 0, 8 * sizeof(float), st) != hipSuccess) {
        bmt_set_error("bmt_prop_decode_loss: memset failed");
        return BMT_EHIP;
    }
    const bool tiled = true;      // (the element-per-thread kernels serve head shapes whose tile does not fit the LDS)
    const size_t lds = (size_t)PROP_TS * (A * 3 + 1) * sizeof(float);
    if (tiled && lds <= 60 * 1024 && B <= 65535)
        hipLaunchKernelGGL(decode_loss_tiled_kernel, dim3(bmt_cdiv(S, PROP_TS), B), dim3(256), lds, st, x, anchors, B, S, A, stride, obj, noobj, tx, tw, preds,
                           loss_ws, pred_bs);
    else
        hipLaunchKernelGGL(decode_loss_kernel, dim3(grid_for((int64_t)B * S * A)), dim3(256), 0, st, x, anchors, B, S, A, stride, obj, noobj, tx, tw, preds, loss_ws,
                           pred_bs);
    BMT_CHECK_LAUNCH("bmt_prop_decode_loss");
    return BMT_OK;
}

// every head of a generator at once (round 6): head i's sums ws[i][0..5] -> losses[i][0..4] as loss_finalize_kernel does, and the column
// sums over all heads / the first n_first heads (modality A) / the others into sums[3][5] -- what the generator returns (total loss, the
// per-modality dictionaries of loss terms) without one finalize launch per head and ~100 scalar adds of the framework's
__global__ void loss_finalize_multi_kernel(float* __restrict__ ws, int n_heads, int n_first, const float* __restrict__ counts_first,
                                           const float* __restrict__ counts_second, float obj_coeff, float noobj_coeff,
                                           float* __restrict__ losses, float* __restrict__ sums) {
    __shared__ float sl[64][5];
    const int i = threadIdx.x;
    float l[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n_heads) {
        float* w = ws + 8 * i;
        const float* cn = i < n_first ? counts_first : counts_second;
        if (cn != nullptr) { w[4] = cn[0]; w[5] = cn[1]; }          // data parallel: LOCAL sums over GLOBAL cell counts (the backward reads them here)
        l[0] = w[0] / w[4]; l[1] = w[1] / w[4]; l[2] = w[2] / w[4]; l[3] = w[3] / w[5];
        l[4] = (l[0] + l[1]) + (obj_coeff * l[2] + noobj_coeff * l[3]);
        for (int c = 0; c < 5; ++c) losses[5 * i + c] = l[c];
    }
    if (i < 64) for (int c = 0; c < 5; ++c) sl[i][c] = l[c];
    __syncthreads();
    if (i < 5) {            // sums[g][c]: g = 1 the first n_first heads, 2 the rest -- running sums in head order, as the reference's -- and
        float a = 0.f, b = 0.f;      // g = 0 their sum (total_loss = total_loss_A + total_loss_V, reference :377)
        for (int h = 0; h < n_first; ++h) a += sl[h][i];
        for (int h = n_first; h < n_heads; ++h) b += sl[h][i];
        sums[5 + i] = a;
        sums[10 + i] = b;
        sums[i] = a + b;
    }
}

extern "C" int bmt_prop_loss_finalize_multi(float* loss_ws, int n_heads, int n_first, const float* counts_first, const float* counts_second,
                                            float obj_coeff, float noobj_coeff, float* losses, float* sums, void* stream) {
    BMT_CHECK_ARG(loss_ws && losses && sums && n_heads > 0 && n_heads <= 64 && n_first >= 0 && n_first <= n_heads, "bmt_prop_loss_finalize_multi: bad args");
    hipLaunchKernelGGL(loss_finalize_multi_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_ws, n_heads, n_first, counts_first, counts_second,
                       obj_coeff, noobj_coeff, losses, sums);
    BMT_CHECK_LAUNCH("bmt_prop_loss_finalize_multi");
    return BMT_OK;
}

extern "C" int bmt_prop_loss_finalize(const float* loss_ws, float obj_coeff, float noobj_coeff, float* losses, void* stream) {
    BMT_CHECK_ARG(loss_ws && losses, "bmt_prop_loss_finalize: bad args");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_ws, obj_coeff, noobj_coeff, losses);
    BMT_CHECK_LAUNCH("bmt_prop_loss_finalize");
    return BMT_OK;
}

extern "C" int bmt_prop_loss_bwd(const float* x, int B, int S, int A, const uint8_t* obj, const uint8_t* noobj, const float* tx,
                                 const float* tw, const float* loss_ws, float obj_coeff, float noobj_coeff, const float* gscale_dev,
                                 float* dx, void* stream) {
    BMT_CHECK_ARG(x && obj && noobj && tx && tw && loss_ws && gscale_dev && dx && B > 0 && S > 0 && A > 0, "bmt_prop_loss_bwd: bad args");
    const bool tiled = true;      // (the element-per-thread kernels serve head shapes whose tile does not fit the LDS)
    const size_t lds = (size_t)PROP_TS * (A * 3 + 1) * sizeof(float);
    if (tiled && lds <= 60 * 1024 && B <= 65535)
        hipLaunchKernelGGL(loss_bwd_tiled_kernel, dim3(bmt_cdiv(S, PROP_TS), B), dim3(256), lds, (hipStream_t)stream, x, B, S, A, obj, noobj, tx, tw, loss_ws,
                           obj_coeff, noobj_coeff, gscale_dev, dx);
    else
        hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for((int64_t)B * S * A)), dim3(256), 0, (hipStream_t)stream, x, B, S, A, obj, noobj, tx, tw, loss_ws, obj_coeff, noobj_coeff, gscale_dev, dx);
    BMT_CHECK_LAUNCH("bmt_prop_loss_bwd");
    return BMT_OK;
}
