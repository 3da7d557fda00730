// bmt_adam_step / bmt_grad_sqnorm -- fused multi-tensor optimizer (K11 of SURVEY.md 2.3).
// torch.optim.Adam semantics (scripts/train_captioning_module.py:46-48: lr 5e-5, betas (0.9,0.999), eps 1e-8, wd 0)
// and torch.nn.utils.clip_grad_norm_ (epoch_loops/captioning_epoch_loops.py:138-139).
// HBM-bound: 16 B read + 12 B written per parameter; ONE launch walks every tensor through a device-side
// pointer table (grid.y = tensor), the step counter lives in device memory so a captured hipGraph replays it.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes, int n_tensors,
                                                    const int64_t* __restrict__ step_dev, float lr, float beta1, float beta2, float eps,
                                                    float weight_decay, const float* __restrict__ grad_scale, int vec) {
    __shared__ float s_bc[2];
    const int t = blockIdx.y;
    if (threadIdx.x == 0) {
        const double step = (double)step_dev[0];
        s_bc[0] = (float)(1.0 - pow((double)beta1, step));
        s_bc[1] = (float)sqrt(1.0 - pow((double)beta2, step));
    }
    __syncthreads();
    const float bc1 = s_bc[0], bc2_sqrt = s_bc[1];
    const float step_size = lr / bc1;
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    float* p = reinterpret_cast<float*>(ptrs[t]);
    const float* g = reinterpret_cast<const float*>(ptrs[n_tensors + t]);
    float* m = reinterpret_cast<float*>(ptrs[2 * n_tensors + t]);
    float* v = reinterpret_cast<float*>(ptrs[3 * n_tensors + t]);
    const int64_t n = sizes[t];
    // four elements per thread and instruction where the tensor allows it (16-byte aligned, a multiple of 4 elements: every weight matrix and
    // all but a few odd-sized vectors) -- the same arithmetic per element
    if (vec && (n & 3) == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
        const int64_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const float4 gq = g4[i], pq = p4[i], mq = m4[i], vq = v4[i];
            float ge[4] = {gq.x * gs, gq.y * gs, gq.z * gs, gq.w * gs};
            const float pe[4] = {pq.x, pq.y, pq.z, pq.w}, me[4] = {mq.x, mq.y, mq.z, mq.w}, ve[4] = {vq.x, vq.y, vq.z, vq.w};
            float po[4], mo[4], vo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (weight_decay != 0.f) ge[q] = __fmaf_rn(weight_decay, pe[q], ge[q]);
                mo[q] = me[q] + (ge[q] - me[q]) * (1.f - beta1);
                vo[q] = ve[q] * beta2 + (1.f - beta2) * ge[q] * ge[q];
                const float denom = sqrtf(vo[q]) / bc2_sqrt + eps;
                po[q] = pe[q] - step_size * (mo[q] / denom);
            }
            p4[i] = make_float4(po[0], po[1], po[2], po[3]);
            m4[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
            v4[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gs;
        const float pi = p[i];
        if (weight_decay != 0.f) gi = __fmaf_rn(weight_decay, pi, gi);
        // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void step_inc_kernel(int64_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

__global__ __launch_bounds__(256) void sqnorm_kernel(void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes, float* __restrict__ out) {
    __shared__ float red[4];
    const float* g = reinterpret_cast<const float*>(ptrs[blockIdx.y]);
    const int64_t n = sizes[blockIdx.y];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(out, s);
}

__global__ void clip_coef_kernel(const float* sq, float max_norm, float* coef) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float c = max_norm / (sqrtf(sq[0]) + 1e-6f);
        coef[0] = c < 1.f ? c : 1.f;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes,
                                                     const float* __restrict__ coef) {
    float* g = reinterpret_cast<float*>(ptrs[blockIdx.y]);
    const int64_t n = sizes[blockIdx.y];
    const float c = coef[0];
    if (c == 1.f) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] *= c;
}

inline int blocks_for(int64_t max_size) {
    int64_t b = (max_size + 1023) / 1024;
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int bmt_adam_step(void* const* ptrs, const int64_t* sizes, int n_tensors, int64_t max_size, int64_t* step_dev, float lr,
                             float beta1, float beta2, float eps, float weight_decay, const float* grad_scale_dev, void* stream) {
    BMT_CHECK_ARG(ptrs && sizes && step_dev && n_tensors > 0 && n_tensors <= 65535 && max_size > 0, "bmt_adam_step: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, st, step_dev);
    const int vec = 1;      // four elements per thread and instruction wherever the tensor allows (0: one)
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(max_size), n_tensors), dim3(256), 0, st, ptrs, sizes, n_tensors, step_dev, lr, beta1,
                       beta2, eps, weight_decay, grad_scale_dev, vec);
    BMT_CHECK_LAUNCH("bmt_adam_step");
    return BMT_OK;
}

extern "C" int bmt_grad_sqnorm(void* const* ptrs, const int64_t* sizes, int n_tensors, int64_t max_size, float* out, float max_norm,
                               float* coef, void* stream) {
    BMT_CHECK_ARG(ptrs && sizes && out && n_tensors > 0 && n_tensors <= 65535 && max_size > 0, "bmt_grad_sqnorm: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) {
        bmt_set_error("bmt_grad_sqnorm: memset failed");
        return BMT_EHIP;
    }
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks_for(max_size), n_tensors), dim3(256), 0, st, ptrs, sizes, out);
    if (coef) hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, out, max_norm, coef);
    BMT_CHECK_LAUNCH("bmt_grad_sqnorm");
    return BMT_OK;
}

extern "C" int bmt_scale_tensors(void* const* ptrs, const int64_t* sizes, int n_tensors, int64_t max_size, const float* coef_dev,
                                 void* stream) {
    BMT_CHECK_ARG(ptrs && sizes && coef_dev && n_tensors > 0 && n_tensors <= 65535 && max_size > 0, "bmt_scale_tensors: bad args");
    hipLaunchKernelGGL(scale_kernel, dim3(blocks_for(max_size), n_tensors), dim3(256), 0, (hipStream_t)stream, ptrs, sizes, coef_dev);
    BMT_CHECK_LAUNCH("bmt_scale_tensors");
    return BMT_OK;
}
