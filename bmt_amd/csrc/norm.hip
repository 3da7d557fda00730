// bmt_layernorm_fwd / bmt_layernorm_bwd -- nn.LayerNorm(size) of ResidualConnection / BridgeConnection
// (model/blocks.py:127,131,143,150): biased variance, eps inside the sqrt, affine.
//
// HBM-bound.  One 64-lane wave per row, float4 accesses when D % 4 == 0 (every width on the hot path:
// 128, 300, 600, 1024); the row is read from HBM once (second and third sweeps hit L1/L2).
// Backward keeps per-column dgamma/dbeta partials in registers over a chunk of rows, reduces the four
// waves of a workgroup through LDS and issues ONE atomic per column per workgroup.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <bool VEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y, int64_t ldy,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                      float eps, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                      int64_t ldp, int pcols, int lo_f16, const int* __restrict__ rows_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rows_dev != nullptr) rows = min(rows, *rows_dev);      // packed rows: the count is data (bmt_gemm_bf16_args.rows_dev)
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    float* yr = y ? y + (int64_t)row * ldy : nullptr;
    uint16_t* hr = hi ? hi + (int64_t)row * ldp : nullptr;      // bf16 operand planes of the output (GEMM / attention input)
    uint16_t* lr = lo ? lo + (int64_t)row * ldp : nullptr;
    float s = 0.f;
    if constexpr (VEC) {
        for (int c = lane * 4; c < D; c += 256) { const float4 v = ld4(xr + c); s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int c = lane; c < D; c += 64) s += xr[c];
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
    if constexpr (VEC) {
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = ld4(xr + c);
            const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    } else {
        for (int c = lane; c < D; c += 64) { const float a = xr[c] - mu; q += a * a; }
    }
    const float var = wave_sum(q) / (float)D;
    const float rs = 1.f / sqrtf(var + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    if constexpr (VEC) {
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = ld4(xr + c), g = ld4(gamma + c), b = ld4(beta + c);
            float4 o;
            o.x = (v.x - mu) * rs * g.x + b.x; o.y = (v.y - mu) * rs * g.y + b.y;
            o.z = (v.z - mu) * rs * g.z + b.z; o.w = (v.w - mu) * rs * g.w + b.w;
            if (yr) *reinterpret_cast<float4*>(yr + c) = o;
            if (hr) {
                uint32_t h0, l0, h1, l1;
                split_bf2(o.x, o.y, h0, l0);
                split_bf2(o.z, o.w, h1, l1);
                if (lo_f16) { l0 = pack_h2(o.x, o.y); l1 = pack_h2(o.z, o.w); }     // second plane = fp16(o) (fp16 forward operand)
                *reinterpret_cast<uint2*>(hr + c) = make_uint2(h0, h1);
                if (lr) *reinterpret_cast<uint2*>(lr + c) = make_uint2(l0, l1);
            }
        }
    } else {
        for (int c = lane; c < D; c += 64) {
            const float o = (xr[c] - mu) * rs * gamma[c] + beta[c];
            if (yr) yr[c] = o;
            if (hr) {
                const __bf16 h = (__bf16)o;
                hr[c] = __builtin_bit_cast(uint16_t, h);
                if (lr) lr[c] = lo_f16 ? __builtin_bit_cast(uint16_t, (_Float16)o) : __builtin_bit_cast(uint16_t, (__bf16)(o - (float)h));
            }
        }
    }
    if (hr)      // zero padding of the reduction extent (planes are read in 64-column steps)
        for (int c = D + lane; c < pcols; c += 64) {
            hr[c] = 0;
            if (lr) lr[c] = 0;
        }
}

// the same computation with the row held in registers (D <= 2048, 16-byte aligned rows): ONE read of x instead of three (sum, variance,
// output) -- same additions in the same order, so the same bits.  NV = ceil(D / 256) float4 groups per lane.
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, int64_t ldy,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                          float eps, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                          int64_t ldp, int pcols, int lo_f16, const int* __restrict__ rows_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rows_dev != nullptr) rows = min(rows, *rows_dev);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    float* yr = y ? y + (int64_t)row * ldy : nullptr;
    uint16_t* hr = hi ? hi + (int64_t)row * ldp : nullptr;
    uint16_t* lr = lo ? lo + (int64_t)row * ldp : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = (c < D) ? ld4(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < D) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rs = 1.f / sqrtf(var + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float4 g = ld4(gamma + c), b = ld4(beta + c);
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + b.x; o.y = (v[i].y - mu) * rs * g.y + b.y;
            o.z = (v[i].z - mu) * rs * g.z + b.z; o.w = (v[i].w - mu) * rs * g.w + b.w;
            if (yr) *reinterpret_cast<float4*>(yr + c) = o;
            if (hr) {
                uint32_t h0, l0, h1, l1;
                split_bf2(o.x, o.y, h0, l0);
                split_bf2(o.z, o.w, h1, l1);
                if (lo_f16) { l0 = pack_h2(o.x, o.y); l1 = pack_h2(o.z, o.w); }
                *reinterpret_cast<uint2*>(hr + c) = make_uint2(h0, h1);
                if (lr) *reinterpret_cast<uint2*>(lr + c) = make_uint2(l0, l1);
            }
        }
    }
    if (hr)
        for (int c = D + lane; c < pcols; c += 64) {
            hr[c] = 0;
            if (lr) lr[c] = 0;
        }
}

// rows each wave sweeps: sized so that a launch has ~512 workgroups (2 per CU); runtime parameter
static int ln_bwd_rows_per_wave(int rows) {
    const int waves = 4096;      // waves a launch aims for: 4096 = 1024 workgroups (measured: 2048 / 4096 / 8192 -> 8.49-8.50 / 8.44-8.45 / 8.52-8.54 ms per step, profiles/r04_q_ab_ln.txt)
    const int w = waves < 256 ? 256 : waves;
    const int r = (rows + w - 1) / w;
    return r < 1 ? 1 : r;
}

// NV = ceil(D / 256): float4 column groups per lane
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                      int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* dx, int64_t lddx,
                                                      const float* dx_add, int64_t ldadd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      float* __restrict__ partial, int rows_per_wave, int rows, int D,
                                                      const float* __restrict__ dx_add2, int64_t ldadd2,
                                                      uint16_t* __restrict__ gp_hi, int64_t gp_ld, float gp_drop_p, const uint64_t* __restrict__ gp_rng,
                                                      uint32_t gp_site, const int* __restrict__ rows_dev) {
    if (rows_dev != nullptr) rows = min(rows, *rows_dev);      // packed rows: workgroups past the count leave zero partials
    // gp_hi (optional): the bf16 operand plane of dropout_site(dx) -- what the PREVIOUS sublayer's last GEMM backward reads as its upstream
    // gradient (x_out = x + dropout(sublayer(LN x)): d x_out reaches that GEMM through the residual dropout's mask) -- and, in the third
    // block of the workgroup's partials, its column sums (that GEMM's bias gradient): the separate conversion pass over dx is gone
    extern __shared__ __attribute__((aligned(16))) float sred[];   // [2 or 3][3 waves][NV*256]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const DropCtx dcx = make_drop(gp_drop_p, gp_rng, gp_site);
    float4 ag[NV], ab[NV], gm[NV], am[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = ag[i];
        am[i] = ag[i];
        const int c = lane * 4 + 256 * i;
        gm[i] = (c < D) ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row0 = (blockIdx.x * 4 + wid) * rows_per_wave;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + (int64_t)row * ldx;
        const float* dr = dy + (int64_t)row * lddy;
        float4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                const float4 xv = ld4(xr + c), dv = ld4(dr + c);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * gm[i].x, dv.y * gm[i].y, dv.z * gm[i].z, dv.w * gm[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
            } else {
                xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                g[i] = xh[i];
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        float* dxr = dx + (int64_t)row * lddx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                float4 o;
                o.x = rs * (g[i].x - c1 - xh[i].x * c2); o.y = rs * (g[i].y - c1 - xh[i].y * c2);
                o.z = rs * (g[i].z - c1 - xh[i].z * c2); o.w = rs * (g[i].w - c1 - xh[i].w * c2);
                if (dx_add) {
                    const float4 p = ld4(dx_add + (int64_t)row * ldadd + c);
                    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                }
                if (dx_add2) {      // a THIRD consumer of the normalised tensor's input (the other modality's key / value projection)
                    const float4 p = ld4(dx_add2 + (int64_t)row * ldadd2 + c);
                    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                }
                *reinterpret_cast<float4*>(dxr + c) = o;
                if (gp_hi) {
                    float4 m = o;
                    if (dcx.on) {
                        const uint64_t e0 = (uint64_t)((int64_t)row * D + c);
                        m.x = drop_apply(dcx, m.x, e0); m.y = drop_apply(dcx, m.y, e0 + 1);
                        m.z = drop_apply(dcx, m.z, e0 + 2); m.w = drop_apply(dcx, m.w, e0 + 3);
                    }
                    *reinterpret_cast<uint2*>(gp_hi + (int64_t)row * gp_ld + c) = make_uint2(pack_bf2(m.x, m.y), pack_bf2(m.z, m.w));
                    am[i].x += m.x; am[i].y += m.y; am[i].z += m.z; am[i].w += m.w;
                }
            } else if (gp_hi && c < ((D + 63) & ~63)) {      // the plane's reduction padding (D = 300 -> columns 300 .. 319): zeros
                *reinterpret_cast<uint2*>(gp_hi + (int64_t)row * gp_ld + c) = make_uint2(0u, 0u);
            }
        }
    }
    // reduce the 4 waves' column partials through LDS; wave 0 issues the atomics
    float* sg = sred;                     // [3][NV*256]
    float* sb = sred + 3 * NV * 256;
    float* sm = sred + 6 * NV * 256;      // (only with gp_hi: the launch sizes the LDS for it)
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            *reinterpret_cast<float4*>(sg + (wid - 1) * NV * 256 + lane * 4 + 256 * i) = ag[i];
            *reinterpret_cast<float4*>(sb + (wid - 1) * NV * 256 + lane * 4 + 256 * i) = ab[i];
            if (gp_hi) *reinterpret_cast<float4*>(sm + (wid - 1) * NV * 256 + lane * 4 + 256 * i) = am[i];
        }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                float4 tg = ag[i], tb = ab[i];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float4 a = *reinterpret_cast<float4*>(sg + w * NV * 256 + lane * 4 + 256 * i);
                    const float4 b = *reinterpret_cast<float4*>(sb + w * NV * 256 + lane * 4 + 256 * i);
                    tg.x += a.x; tg.y += a.y; tg.z += a.z; tg.w += a.w;
                    tb.x += b.x; tb.y += b.y; tb.z += b.z; tb.w += b.w;
                }
                if (partial) {   // two-stage: plain stores of this workgroup's column partials, summed by ln_bwd_reduce_kernel
                    float* pr = partial + (int64_t)blockIdx.x * (gp_hi ? 3 : 2) * D;
                    *reinterpret_cast<float4*>(pr + c) = tg;
                    *reinterpret_cast<float4*>(pr + D + c) = tb;
                    if (gp_hi) {
                        float4 tm = am[i];
#pragma unroll
                        for (int w = 0; w < 3; ++w) {
                            const float4 a = *reinterpret_cast<float4*>(sm + w * NV * 256 + lane * 4 + 256 * i);
                            tm.x += a.x; tm.y += a.y; tm.z += a.z; tm.w += a.w;
                        }
                        *reinterpret_cast<float4*>(pr + 2 * D + c) = tm;
                    }
                    continue;
                }
                atomicAdd(dgamma + c + 0, tg.x); atomicAdd(dgamma + c + 1, tg.y);
                atomicAdd(dgamma + c + 2, tg.z); atomicAdd(dgamma + c + 3, tg.w);
                atomicAdd(dbeta + c + 0, tb.x); atomicAdd(dbeta + c + 1, tb.y);
                atomicAdd(dbeta + c + 2, tb.z); atomicAdd(dbeta + c + 3, tb.w);
            }
        }
    }
}

// second stage: dgamma[c] += sum_blk partial[blk][c], dbeta[c] += sum_blk partial[blk][D + c].  64 columns x 4 row groups per
// workgroup, blockIdx.y = slice of 64 partial rows (16 loads per thread); one atomic per column per slice.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int D) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;   // column in the concatenated [2*D] row
    const int b0 = blockIdx.y * 64, b1 = min(nblk, b0 + 64);
    float s0 = 0.f, s1 = 0.f;
    if (c < 2 * D) {
        int b = b0 + grp;
        for (; b + 4 < b1; b += 8) { s0 += partial[(int64_t)b * 2 * D + c]; s1 += partial[(int64_t)(b + 4) * 2 * D + c]; }
        if (b < b1) s0 += partial[(int64_t)b * 2 * D + c];
    }
    red[grp][cl] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c < 2 * D) {
        const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        atomicAdd(c < D ? dgamma + c : dbeta + (c - D), t);
    }
}

// generic fallback (any D, any alignment): one wave per row, per-element atomics for dgamma/dbeta
__global__ __launch_bounds__(256) void ln_bwd_scalar_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                             int64_t ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* dx, int64_t lddx, const float* dx_add, int64_t ldadd,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int D,
                                                             const int* __restrict__ rows_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rows_dev != nullptr) rows = min(rows, *rows_dev);
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (int64_t)row * ldx;
    const float* dr = dy + (int64_t)row * lddy;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xh = (xr[c] - mu) * rs, g = dr[c] * gamma[c];
        s1 += g; s2 += g * xh;
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
    float* dxr = dx + (int64_t)row * lddx;
    for (int c = lane; c < D; c += 64) {
        const float xh = (xr[c] - mu) * rs, g = dr[c] * gamma[c];
        const float o = rs * (g - c1 - xh * c2);
        dxr[c] = dx_add ? dx_add[(int64_t)row * ldadd + c] + o : o;
        atomicAdd(dgamma + c, dr[c] * xh);
        atomicAdd(dbeta + c, dr[c]);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int bmt_layernorm_fwd_planes(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                                        float* mean, float* rstd, uint16_t* hi, uint16_t* lo, int lo_f16, int64_t ldp, int rows, int D,
                                        float eps, const int* rows_dev, void* stream) {
    BMT_CHECK_ARG(x && gamma && beta && (y || hi) && rows >= 0 && D > 0, "bmt_layernorm_fwd: bad args");
    BMT_CHECK_ARG(!lo || hi, "bmt_layernorm_fwd_planes: lo plane without hi plane");
    BMT_CHECK_ARG(!hi || ldp >= D, "bmt_layernorm_fwd_planes: plane row stride %lld < D=%d", (long long)ldp, D);
    if (rows == 0) return BMT_OK;
    const int pad = (D + 63) / 64 * 64;
    const int pcols = hi ? (int)(pad < ldp ? pad : ldp) : 0;
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (!y || (ldy % 4 == 0 && al16(y))) && al16(x) && al16(gamma) && al16(beta) &&
                     (!hi || ((ldp % 4 == 0) && ((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 7) == 0));
    dim3 grid(bmt_cdiv(rows, 4)), block(256);
#define BMT_LNF(NV) hipLaunchKernelGGL(ln_fwd_reg_kernel<NV>, grid, block, 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, D, eps, hi, lo, ldp, pcols, lo_f16, rows_dev)
    if (vec && D <= 2048) {
        const int nv = bmt_cdiv(D, 256);
        if (nv <= 1) BMT_LNF(1);
        else if (nv <= 2) BMT_LNF(2);
        else if (nv <= 4) BMT_LNF(4);
        else BMT_LNF(8);
    } else if (vec) hipLaunchKernelGGL(ln_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, D, eps, hi, lo, ldp, pcols, lo_f16, rows_dev);
#undef BMT_LNF
    else hipLaunchKernelGGL(ln_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, D, eps, hi, lo, ldp, pcols, lo_f16, rows_dev);
    BMT_CHECK_LAUNCH("bmt_layernorm_fwd");
    return BMT_OK;
}

extern "C" int bmt_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                                 float* mean, float* rstd, int rows, int D, float eps, void* stream) {
    BMT_CHECK_ARG(y, "bmt_layernorm_fwd: bad args");
    return bmt_layernorm_fwd_planes(x, ldx, gamma, beta, y, ldy, mean, rstd, nullptr, nullptr, 0, 0, rows, D, eps, nullptr, stream);
}

extern "C" int bmt_layernorm_bwd_blocks(int rows) { return rows <= 0 ? 0 : bmt_cdiv(rows, 4 * ln_bwd_rows_per_wave(rows)); }

extern "C" int bmt_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, float* dx, int64_t lddx, int accumulate_dx,
                                 float* dgamma, float* dbeta, float* partial_ws, int rows, int D, void* stream) {
    return bmt_layernorm_bwd_add(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, accumulate_dx ? dx : nullptr, lddx, dgamma, dbeta,
                                 partial_ws, rows, D, nullptr, stream);
}

struct LnGradPlane { uint16_t* hi; int64_t ld; float drop_p; const uint64_t* rng; uint32_t site; };
static int ln_bwd_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                       float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, float* dgamma, float* dbeta, float* partial_ws, int rows,
                       int D, void* stream, bool leave_partials, const float* dx_add2 = nullptr, int64_t ldadd2 = 0,
                       const LnGradPlane* gp = nullptr, const int* rows_dev = nullptr);

extern "C" int bmt_layernorm_bwd_add(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                     const float* mean, const float* rstd, float* dx, int64_t lddx, const float* dx_add,
                                     int64_t ldadd, float* dgamma, float* dbeta, float* partial_ws, int rows, int D, const int* rows_dev,
                                     void* stream) {
    BMT_CHECK_ARG(dgamma && dbeta, "bmt_layernorm_bwd: bad args");
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ldadd, dgamma, dbeta, partial_ws, rows, D, stream, false, nullptr, 0, nullptr,
                       rows_dev);
}

extern "C" int bmt_layernorm_bwd_partial(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                                         const float* rstd, float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, float* partial_ws,
                                         int rows, int D, const int* rows_dev, void* stream) {
    BMT_CHECK_ARG(partial_ws, "bmt_layernorm_bwd_partial: needs the partial workspace");
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ldadd, nullptr, nullptr, partial_ws, rows, D, stream, true, nullptr, 0, nullptr,
                       rows_dev);
}

extern "C" int bmt_layernorm_bwd_partial2(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                                          const float* rstd, float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, const float* dx_add2,
                                          int64_t ldadd2, float* partial_ws, int rows, int D, const int* rows_dev, void* stream) {
    BMT_CHECK_ARG(partial_ws, "bmt_layernorm_bwd_partial2: needs the partial workspace");
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ldadd, nullptr, nullptr, partial_ws, rows, D, stream, true, dx_add2, ldadd2,
                       nullptr, rows_dev);
}

// bmt_layernorm_bwd_partial2 that ALSO emits the bf16 operand plane of dropout_site(dx) and its column partials (ABI 5): partial_ws is
// [blocks][3 D] then -- dgamma | dbeta | column sums of the masked dx.  gp_hi [rows][gp_ld] (gp_ld >= round_up(D, 64); columns D .. round_up(D, 64) - 1 are written as zeros: the consuming GEMM's reduction padding).
extern "C" int bmt_layernorm_bwd_emit(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, const float* dx_add2,
                                      int64_t ldadd2, float* partial_ws, uint16_t* gp_hi, int64_t gp_ld, float drop_p, const uint64_t* rng,
                                      uint32_t site, int rows, int D, const int* rows_dev, void* stream) {
    BMT_CHECK_ARG(partial_ws && gp_hi && gp_ld >= ((D + 63) & ~63) && D % 4 == 0 && gp_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(gp_hi) & 7) == 0,
                  "bmt_layernorm_bwd_emit: needs the partial workspace and an 8-byte aligned plane of at least round_up(D, 64) columns, D a multiple of 4");
    BMT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "bmt_layernorm_bwd_emit: bad dropout arguments");
    const LnGradPlane gp{gp_hi, gp_ld, drop_p, rng, site};
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ldadd, nullptr, nullptr, partial_ws, rows, D, stream, true, dx_add2, ldadd2,
                       &gp, rows_dev);
}

static int ln_bwd_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                       float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, float* dgamma, float* dbeta, float* partial_ws, int rows,
                       int D, void* stream, bool leave_partials, const float* dx_add2, int64_t ldadd2, const LnGradPlane* gp, const int* rows_dev) {
    BMT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows >= 0 && D > 0, "bmt_layernorm_bwd: bad args");
    if (rows == 0) return leave_partials ? 1 : BMT_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && al16(x) && al16(dy) && al16(dx) &&
                     al16(gamma) && D <= 2048 && (!dx_add || (al16(dx_add) && ldadd % 4 == 0)) && (!dx_add2 || (al16(dx_add2) && ldadd2 % 4 == 0));
    if (!vec && (leave_partials || dx_add2 || gp)) return 1;       // (the scalar kernel adds into dgamma / dbeta directly and knows one addend: the caller falls back)
    if (!vec) {
        hipLaunchKernelGGL(ln_bwd_scalar_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, dy, lddy, x, ldx, gamma, mean, rstd, dx,
                           lddx, dx_add, ldadd, dgamma, dbeta, rows, D, rows_dev);
        BMT_CHECK_LAUNCH("bmt_layernorm_bwd(scalar)");
        return BMT_OK;
    }
    dim3 grid(bmt_layernorm_bwd_blocks(rows)), block(256);
    const int nv = bmt_cdiv(D, 256);
#define BMT_LN(NV)                                                                                                         \
    hipLaunchKernelGGL(ln_bwd_kernel<NV>, grid, block, (gp ? 3 : 2) * 3 * NV * 256 * sizeof(float), st, dy, lddy, x, ldx, gamma, mean, rstd, \
                       dx, lddx, dx_add, ldadd, dgamma, dbeta, partial_ws, ln_bwd_rows_per_wave(rows), rows, D, dx_add2, ldadd2,           \
                       gp ? gp->hi : nullptr, gp ? gp->ld : 0, gp ? gp->drop_p : 0.f, gp ? gp->rng : nullptr, gp ? gp->site : 0u, rows_dev)
    if (nv <= 1) BMT_LN(1);
    else if (nv <= 2) BMT_LN(2);
    else if (nv <= 4) BMT_LN(4);
    else BMT_LN(8);
#undef BMT_LN
    BMT_CHECK_LAUNCH("bmt_layernorm_bwd");
    if (partial_ws && !leave_partials) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(bmt_cdiv(2 * D, 64), bmt_cdiv((int)grid.x, 64)), dim3(256), 0, st, partial_ws, (int)grid.x, dgamma, dbeta, D);
        BMT_CHECK_LAUNCH("bmt_layernorm_bwd(reduce)");
    }
    return BMT_OK;
}
