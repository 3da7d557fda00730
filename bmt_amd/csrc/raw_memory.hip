// The decoder's cross-attention against the RAW encoder memory (ABI 8): the kernels between the small products.
//
// model/multihead_attention.py:62-84 projects the memory X (the encoder's output, 6 242 + 19 508 valid rows at configs[1]) to keys and
// values for every decoder layer -- the only large products of the decoder -- to attend from 29 queries per sample.  Reassociated,
//     S_h = (q_h W_k,h) X^T        (+ q_h . b_k: constant along the keys, softmax does not see it)
//     O_h = (P_h X) W_v,h^T + b_v
// the attention runs against X itself: K, V, dK and dV never exist.  The products are small GEMMs
// (bmt_gemm_small_batched); here: the transposed copy of the memory they need, and the softmax between them, forward and backward.
#include "common.h"

namespace {

// ---- bmt_memory_transposed: packed memory plane (fp16) -> per sample, transposed and padded: XT[b][d][k], k < Skp
//   xt_f16  fp16(X)                the B operand of O' = P . X (reduction over the keys)
//   xtc_bf  bf16(X - mean_b)       the B operand of dQ' = dS . X: sum_k dS[q][k] = 0 exactly, so the sample's mean key drops out of the
//                                  product -- and with it the systematic error of a ROUNDED dS that does not sum to zero (the mean-key
//                                  correction of the attention backward, bmt_attn_kmean, built into the operand)
// two launches: the column sums of every sample (64 x 64 tiles, atomics into a workspace the caller zeroed), then 64 x 64
// tiles through LDS (workgroup = (64 columns, 64 keys, sample): 16-byte loads along the rows, 16-byte stores along the keys)
__global__ __launch_bounds__(256) void memory_mean_kernel(const uint16_t* __restrict__ X, int64_t ld, const int* __restrict__ off, int D, int Skp,
                                                           float* __restrict__ sum) {
    // workgroup = (64 columns, 64 keys, sample): column sums of its keys, added into sum[b][d] (zeroed by the caller)
    __shared__ float colsum[4][64];
    const int b = blockIdx.z, d0 = blockIdx.x * 64, k0 = blockIdx.y * 64, tid = threadIdx.x;
    const int r0 = off[b], len = min(off[b + 1] - r0, Skp);
    if (k0 >= len) return;
    const int c = tid & 63, rg = tid >> 6;
    float s = 0.f;
    if (d0 + c < D) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int k = k0 + rg + 4 * i;
            if (k < len) s += h_bits2f(X[(int64_t)(r0 + k) * ld + d0 + c]);
        }
    }
    colsum[rg][c] = s;
    __syncthreads();
    if (tid < 64 && d0 + tid < D) atomicAdd(sum + (int64_t)b * D + d0 + tid, (colsum[0][tid] + colsum[1][tid]) + (colsum[2][tid] + colsum[3][tid]));
}

__global__ __launch_bounds__(256) void memory_transposed_kernel(const uint16_t* __restrict__ X, int64_t ld, const int* __restrict__ off, int D, int Skp,
                                                                 const float* __restrict__ mean, uint16_t* __restrict__ xt_f16,
                                                                 uint16_t* __restrict__ xtc_bf) {
    __shared__ uint16_t tile[64][72];                         // [key][column], rows of 144 bytes: 16-byte aligned
    const int b = blockIdx.z, d0 = blockIdx.x * 64, k0 = blockIdx.y * 64, tid = threadIdx.x;
    const int r0 = off[b], len = min(off[b + 1] - r0, Skp);
    const int piece = tid & 7;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {                          // 64 keys x 8 pieces of 8 columns
        const int kr = ps * 32 + (tid >> 3), k = k0 + kr, c = d0 + piece * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (k < len && c + 8 <= D) v = *reinterpret_cast<const u32x4*>(X + (int64_t)(r0 + k) * ld + c);
        else if (k < len) {
            uint16_t t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = c + q < D ? X[(int64_t)(r0 + k) * ld + c + q] : (uint16_t)0;
            v = *reinterpret_cast<const u32x4*>(t);
        }
        *reinterpret_cast<u32x4*>(&tile[kr][piece * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {                          // 64 columns x 8 pieces of 8 keys
        const int dc = ps * 32 + (tid >> 3), d = d0 + dc, kk = piece * 8;
        if (d >= D || k0 + kk >= Skp) continue;
        const float mu = (xtc_bf && len > 0) ? mean[(int64_t)b * D + d] / (float)len : 0.f;      // (mean = the column SUM over the sample's keys)
        uint16_t f[8], c_[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint16_t h = tile[kk + q][dc];
            f[q] = h;
            c_[q] = (uint16_t)f2bf_bits(k0 + kk + q < len ? h_bits2f(h) - mu : 0.f);
        }
        const int64_t o = ((int64_t)b * D + d) * Skp + k0 + kk;
        if (xt_f16) *reinterpret_cast<u32x4*>(xt_f16 + o) = *reinterpret_cast<const u32x4*>(f);
        if (xtc_bf) *reinterpret_cast<u32x4*>(xtc_bf + o) = *reinterpret_cast<const u32x4*>(c_);
    }
}

// ---- softmax over a sample's keys, one wave per (sample, head, query row): S fp32 [B][H][32][Skp] (scores before the scale) ->
//   p_f16   fp16 [B][H][32][Skp]                       the A operand of O' = P . X
//   p_bf    bf16 at p_bf + b * sb + h * sh + t * Skp   the k-major A operand of the memory's gradient (a row block of the per-sample stack)
// rows t >= Tq and keys >= the sample's length are written as zeros (they are reduction padding of the products that follow)
__global__ __launch_bounds__(256) void raw_softmax_fwd_kernel(const float* __restrict__ S, const int* __restrict__ off, int H, int Tq, int Skp, float scale,
                                                               uint16_t* __restrict__ p_f16, uint16_t* __restrict__ p_bf, int64_t sb, int64_t sh, int nrows) {
    constexpr int NG = 2;                                     // groups of 8 consecutive keys per lane: Skp <= 1024; 16-byte accesses
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const int t = row & 31, bh = row >> 5, b = bh / H, h = bh - b * H;
    const int len = min(off[b + 1] - off[b], Skp);
    const float* s = S + (int64_t)row * Skp;
    uint16_t* pf = p_f16 + (int64_t)row * Skp;
    uint16_t* pb = p_bf ? p_bf + b * sb + h * sh + (int64_t)t * Skp : nullptr;
    const bool live = t < Tq && len > 0;
    float v[NG][8];
    float m = -INFINITY;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int k0 = (lane + 64 * g) * 8;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (live && k0 < len) {                               // (a group that straddles the length: its tail may hold anything)
            a0 = *reinterpret_cast<const float4*>(s + k0);
            a1 = *reinterpret_cast<const float4*>(s + k0 + 4);
        }
        const float t8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[g][q] = (live && k0 + q < len) ? t8[q] : -INFINITY;
            m = fmaxf(m, v[g][q]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float sc = scale * 1.4426950408889634f;            // exp(x * scale) = exp2(x * scale * log2 e)
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[g][q] = (live && (lane + 64 * g) * 8 + q < len) ? exp2f((v[g][q] - m) * sc) : 0.f;
            sum += v[g][q];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = live ? 1.f / sum : 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int k0 = (lane + 64 * g) * 8;
        if (k0 >= Skp) continue;
        u32x4 f, bfv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float p0 = v[g][2 * q] * inv, p1 = v[g][2 * q + 1] * inv;
            f[q] = pack_h2(p0, p1);
            bfv[q] = pack_bf2(p0, p1);
        }
        *reinterpret_cast<u32x4*>(pf + k0) = f;
        if (pb) *reinterpret_cast<u32x4*>(pb + k0) = bfv;
    }
}

// ---- its backward: dS = P o (dP - rowsum(P o dP)) * scale as bf16 at ds_bf + b * sb + h * sh + t * Skp (A of dQ' = dS . X row-major, and k-major
// A of the memory's gradient); dP fp32 [B][H][32][Skp] = dO' . X^T; zeros for rows t >= Tq and keys >= the length
__global__ __launch_bounds__(256) void raw_softmax_bwd_kernel(const uint16_t* __restrict__ p_f16, const float* __restrict__ dP, const int* __restrict__ off,
                                                               int H, int Tq, int Skp, float scale, uint16_t* __restrict__ ds_bf, int64_t sb, int64_t sh,
                                                               int nrows) {
    constexpr int NG = 2;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const int t = row & 31, bh = row >> 5, b = bh / H, h = bh - b * H;
    const int len = min(off[b + 1] - off[b], Skp);
    const uint16_t* pf = p_f16 + (int64_t)row * Skp;
    const float* dp = dP + (int64_t)row * Skp;
    uint16_t* ds = ds_bf + b * sb + h * sh + (int64_t)t * Skp;
    const bool live = t < Tq && len > 0;
    float pv[NG][8], dv[NG][8];
    float delta = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int k0 = (lane + 64 * g) * 8;
        u32x4 ph = {0u, 0u, 0u, 0u};
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (live && k0 < len) {
            ph = *reinterpret_cast<const u32x4*>(pf + k0);
            a0 = *reinterpret_cast<const float4*>(dp + k0);
            a1 = *reinterpret_cast<const float4*>(dp + k0 + 4);
        }
        const float t8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = live && k0 + q < len;
            pv[g][q] = in ? h_bits2f((ph[q >> 1] >> (16 * (q & 1))) & 0xffffu) : 0.f;
            dv[g][q] = in ? t8[q] : 0.f;
            delta += pv[g][q] * dv[g][q];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int k0 = (lane + 64 * g) * 8;
        if (k0 >= Skp) continue;
        u32x4 o_;
#pragma unroll
        for (int q = 0; q < 4; ++q) o_[q] = pack_bf2(pv[g][2 * q] * (dv[g][2 * q] - delta) * scale, pv[g][2 * q + 1] * (dv[g][2 * q + 1] - delta) * scale);
        *reinterpret_cast<u32x4*>(ds + k0) = o_;
    }
}

}  // namespace

extern "C" int bmt_memory_transposed(const uint16_t* x_f16, int64_t ld, const int* off, int B, int D, int Skp, uint16_t* xt_f16, uint16_t* xtc_bf,
                                     float* mean_ws, void* stream) {
    BMT_CHECK_ARG(x_f16 && off && (xt_f16 || xtc_bf) && B > 0 && D > 0 && Skp > 0 && Skp % 64 == 0 && ld >= D && ld % 8 == 0 &&
                      (reinterpret_cast<uintptr_t>(x_f16) & 15) == 0,
                  "bmt_memory_transposed: bad arguments (Skp a multiple of 64; the plane 16-byte aligned with a row stride that is a multiple of 8)");
    BMT_CHECK_ARG(!xtc_bf || mean_ws, "bmt_memory_transposed: the centred plane needs the workspace for the column sums (B * D floats, ZEROED by the caller)");
    if (xtc_bf) hipLaunchKernelGGL(memory_mean_kernel, dim3(bmt_cdiv(D, 64), Skp / 64, B), dim3(256), 0, (hipStream_t)stream, x_f16, ld, off, D, Skp, mean_ws);
    hipLaunchKernelGGL(memory_transposed_kernel, dim3(bmt_cdiv(D, 64), Skp / 64, B), dim3(256), 0, (hipStream_t)stream, x_f16, ld, off, D, Skp, mean_ws, xt_f16,
                       xtc_bf);
    BMT_CHECK_LAUNCH("bmt_memory_transposed");
    return BMT_OK;
}

extern "C" int bmt_raw_softmax_fwd(const float* S, const int* off, int B, int H, int Tq, int Skp, float scale, uint16_t* p_f16, uint16_t* p_bf,
                                   int64_t p_bf_sb, int64_t p_bf_sh, void* stream) {
    BMT_CHECK_ARG(S && off && p_f16 && B > 0 && H > 0 && Tq > 0 && Tq <= 32 && Skp > 0 && Skp <= 1024 && Skp % 8 == 0 &&
                      !((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(p_f16) | reinterpret_cast<uintptr_t>(p_bf)) & 15) && !((p_bf_sb | p_bf_sh) & 7),
                  "bmt_raw_softmax_fwd: bad arguments (at most 32 queries per sample and head, Skp a multiple of 8 and at most 1024, 16-byte aligned buffers)");
    const int nrows = B * H * 32;
    hipLaunchKernelGGL(raw_softmax_fwd_kernel, dim3(bmt_cdiv(nrows, 4)), dim3(256), 0, (hipStream_t)stream, S, off, H, Tq, Skp, scale, p_f16, p_bf, p_bf_sb, p_bf_sh,
                       nrows);
    BMT_CHECK_LAUNCH("bmt_raw_softmax_fwd");
    return BMT_OK;
}

extern "C" int bmt_raw_softmax_bwd(const uint16_t* p_f16, const float* dP, const int* off, int B, int H, int Tq, int Skp, float scale, uint16_t* ds_bf,
                                   int64_t ds_sb, int64_t ds_sh, void* stream) {
    BMT_CHECK_ARG(p_f16 && dP && off && ds_bf && B > 0 && H > 0 && Tq > 0 && Tq <= 32 && Skp > 0 && Skp <= 1024 && Skp % 8 == 0 &&
                      !((reinterpret_cast<uintptr_t>(dP) | reinterpret_cast<uintptr_t>(p_f16) | reinterpret_cast<uintptr_t>(ds_bf)) & 15) && !((ds_sb | ds_sh) & 7),
                  "bmt_raw_softmax_bwd: bad arguments (Skp a multiple of 8, at most 1024; 16-byte aligned buffers)");
    const int nrows = B * H * 32;
    hipLaunchKernelGGL(raw_softmax_bwd_kernel, dim3(bmt_cdiv(nrows, 4)), dim3(256), 0, (hipStream_t)stream, p_f16, dP, off, H, Tq, Skp, scale, ds_bf, ds_sb, ds_sh,
                       nrows);
    BMT_CHECK_LAUNCH("bmt_raw_softmax_bwd");
    return BMT_OK;
}
