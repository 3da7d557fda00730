// Attention whose keys and values are NARROWER than a head (ABI 11): the two weight-side kernels of the reassociated form.
//
// model/multihead_attention.py:62-84 projects the audio stream x (d_a = 128 columns) to d_model = 1024 for H = 4 heads of d_k = 256 -- as the keys
// and values of the audio self-attention and of the video stream's attention over the audio stream.  k_h and v_h are rank-d_a images of x, and
// with queries projected from y (d_b columns: y = x for the self-attention, the 1024-wide video stream for the cross-attention)
//     S_h = q_h k_h^T = (y W'_h^T + c_h) x^T  (+ terms constant along the keys)     W'_h = W_k,h^T W_q,h  [d_a x d_b],  c_h = b_q,h W_k,h
//     O_h = P_h v_h   = (P_h x) W_v,h^T + b_v,h
// so the attention runs at width d_a against x itself (one key / value plane for all heads; bmt_attn_*_args.kv_shared).  What is left on
// the weight side is small (H d_a d_b d_k multiply-adds per module: 17 M / 134 M) and runs in fp32 on the vector units, straight from the fp32
// parameters -- no operand planes of W_q / W_k, no rounding of the weights or of their gradients:
//   bmt_rank_prep    W' and c from the weights, W' written as the operand planes the products read (fp16 hi + lo for q' = y W'^T + c, bf16
//                    for dy = dq' W'), once per optimizer step;
//   bmt_rank_chain   dW_q,h += W_k,h dW'_h,   dW_k,h += W_q,h dW'_h^T + b_q,h^T dc_h,   db_q,h += W_k,h dc_h
//                    from dW' = dq'^T y (one item of the step's grouped weight-gradient launch) and dc = column sums of dq' (bmt_rank_prep
//                    zeroes dW', the accumulator of that item, for the pass to come).
#include "common.h"

namespace {

// Both kernels are fp32 tile products: every workgroup one 64 x 64 output tile, 128 reduction indices of both operands whole in LDS ([k][64 + 4]:
// one load phase, every request in flight at once -- the products are small and latency is all there is), a 4 x 4 register tile per thread
// (two broadcast / contiguous 16-byte LDS reads per 16 multiply-adds).
constexpr int DA = 128, TS = 68;

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tile_mac(const float* As, const float* Bs, int ty, int tx, float (&acc)[4][4]) {
    v2f c[4][2];                     // packed multiply-adds (v_pk_fma_f32): two columns per instruction
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[i][0] = v2f{acc[i][0], acc[i][1]};
        c[i][1] = v2f{acc[i][2], acc[i][3]};
    }
#pragma unroll 8
    for (int k = 0; k < 128; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(As + k * TS + 4 * ty), b = *reinterpret_cast<const float4*>(Bs + k * TS + 4 * tx);
        const float av[4] = {a.x, a.y, a.z, a.w};
        const v2f b0 = v2f{b.x, b.y}, b1 = v2f{b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v2f ai = v2f{av[i], av[i]};
            c[i][0] = __builtin_elementwise_fma(ai, b0, c[i][0]);
            c[i][1] = __builtin_elementwise_fma(ai, b1, c[i][1]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i][0] = c[i][0].x; acc[i][1] = c[i][0].y; acc[i][2] = c[i][1].x; acc[i][3] = c[i][1].y;
    }
}
// rows of a [128 k][ld] source (the reduction index is the ROW): 64 columns from c0, stored as they are
__device__ __forceinline__ void tile_load_kmajor(const float* src, int64_t ld, int c0, float* dst, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, k = idx >> 4, n4 = (idx & 15) * 4;
        *reinterpret_cast<float4*>(dst + k * TS + n4) = *reinterpret_cast<const float4*>(src + (int64_t)k * ld + c0 + n4);
    }
}
// 64 rows of a [rows][ld] source whose COLUMNS are the reduction index (128 of them from k0): stored transposed
__device__ __forceinline__ void tile_load_rowmajor(const float* src, int64_t ld, int k0, float* dst, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, m = idx >> 5, k4 = (idx & 31) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)m * ld + k0 + k4);
        dst[(k4 + 0) * TS + m] = v.x; dst[(k4 + 1) * TS + m] = v.y; dst[(k4 + 2) * TS + m] = v.z; dst[(k4 + 3) * TS + m] = v.w;
    }
}

// W'_h[a][b] = sum_r W_k[h dk + r][a] W_q[h dk + r][b]: workgroup (h, at, nt) = rows 64 at ..., columns 64 nt ... of head h; both weights are read
// as they lie (their rows are the reduction index), dk in phases of 128.  The tile goes out as the operand planes (and / or fp32), the same tile of
// the dW' accumulator is zeroed, and the workgroups of the first column tile add c_h[a] = sum_r b_q[h dk + r] W_k[h dk + r][a].
__global__ __launch_bounds__(256) void rank_prep_kernel(const float* __restrict__ Wq, int64_t ldq, int d_b, const float* __restrict__ Wk, int64_t ldk,
                                                         const float* __restrict__ bq, int dk, int d_a, uint16_t* __restrict__ hi, uint16_t* __restrict__ fh,
                                                         uint16_t* __restrict__ fl, int64_t ldp, float* __restrict__ wp, float* __restrict__ c,
                                                         float* __restrict__ zero) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;
    float* Bs = sm + 128 * TS;
    float* Sb = sm + 2 * 128 * TS;       // b_q's 128 values of the phase (the workgroups of the first column tile: c_h)
    const int tid = threadIdx.x, nq = d_b / 64, na = d_a / 64;
    const int nt = blockIdx.x % nq, at = (blockIdx.x / nq) % na, h = blockIdx.x / (nq * na);
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float cs = 0.f;
    for (int r0 = 0; r0 < dk; r0 += 128) {
        if (r0) __syncthreads();
        tile_load_kmajor(Wk + (int64_t)(h * dk + r0) * ldk, ldk, at * 64, As, tid);
        tile_load_kmajor(Wq + (int64_t)(h * dk + r0) * ldq, ldq, nt * 64, Bs, tid);
        if (nt == 0 && tid < 128 && bq != nullptr) Sb[tid] = bq[h * dk + r0 + tid];      // (one load per value: the 128-step dot product below read b_q from
        __syncthreads();                                                                    // global memory step by step, on the workgroups that finish last)
        tile_mac(As, Bs, ty, tx, acc);
        if (nt == 0 && tid < 64 && bq != nullptr) {
#pragma unroll 8
            for (int k = 0; k < 128; ++k) cs = fmaf(Sb[k], As[k * TS + tid], cs);
        }
    }
    if (nt == 0 && tid < 64 && c != nullptr) c[h * d_a + at * 64 + tid] = cs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = (int64_t)h * d_a + at * 64 + 4 * ty + i;
        const int col = nt * 64 + 4 * tx;
        if (wp) *reinterpret_cast<float4*>(wp + row * d_b + col) = float4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        if (zero) *reinterpret_cast<float4*>(zero + row * d_b + col) = float4{0.f, 0.f, 0.f, 0.f};
        uint16_t h4[4], f4[4], l4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = acc[i][j];
            h4[j] = __builtin_bit_cast(uint16_t, (__bf16)v);
            const _Float16 f = (_Float16)v;
            f4[j] = __builtin_bit_cast(uint16_t, f);
            l4[j] = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)f));
        }
        if (hi) *reinterpret_cast<uint2*>(hi + row * ldp + col) = *reinterpret_cast<const uint2*>(h4);
        if (fh) *reinterpret_cast<uint2*>(fh + row * ldp + col) = *reinterpret_cast<const uint2*>(f4);
        if (fl) *reinterpret_cast<uint2*>(fl + row * ldp + col) = *reinterpret_cast<const uint2*>(l4);
    }
}

// The chain rule, same tiles over a reduction of 128 (d_a = 128):
//   Q tiles (h, mt, nt):      dW_q[h dk + 64 mt ...][64 nt ...] += W_k,h[64 mt ...][.] . dW'_h[.][64 nt ...]                    (reduction over a)
//                             ... those with nt = 0 also db_q[r] += sum_a W_k[r][a] dc[a]
//   K tiles (h, mt, at, kc):  dW_k[h dk + 64 mt ...][64 at ...] += W_q,h[64 mt ...][128 kc ...] . dW'_h[64 at ...][128 kc ...]^T   (reduction over the
//                             128 columns b of chunk kc; atomics when d_b has more than one chunk); kc = 0 adds b_q[r] dc[a]
__global__ __launch_bounds__(256) void rank_chain_kernel(const float* __restrict__ Wq, int64_t ldq, int d_b, const float* __restrict__ Wk, int64_t ldk,
                                                          const float* __restrict__ bq, int H, int dk, const float* __restrict__ dWp,
                                                          const float* __restrict__ dc, float* __restrict__ dWq, int64_t ldgq, float* __restrict__ dWk,
                                                          int64_t ldgk, float* __restrict__ dbq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;                  // [128 k][TS]: A[m][k] transposed
    float* Bs = sm + DA * TS;        // [128 k][TS]: B[k][n]
    const int tid = threadIdx.x, mts = dk / 64, nq = d_b / 64, nc = d_b / 128;
    const int nQ = H * mts * nq;
    int bid = blockIdx.x;
    const bool qrole = bid < nQ;     // workgroup-uniform
    int h, mt, nt = 0, at = 0, kc = 0;
    if (qrole) {
        nt = bid % nq; mt = (bid / nq) % mts; h = bid / (nq * mts);
    } else {
        bid -= nQ;
        kc = bid % nc; at = (bid / nc) % 2; mt = (bid / (2 * nc)) % mts; h = bid / (2 * nc * mts);
    }
    const int r0 = h * dk + mt * 64;
    const float* P = dWp + (int64_t)h * DA * d_b;
    if (qrole) {
        tile_load_rowmajor(Wk + (int64_t)r0 * ldk, ldk, 0, As, tid);
        tile_load_kmajor(P, d_b, nt * 64, Bs, tid);                                   // B[k = a][n = b]: rows of dW' as they are
    } else {
        tile_load_rowmajor(Wq + (int64_t)r0 * ldq, ldq, kc * 128, As, tid);
        tile_load_rowmajor(P + (int64_t)at * 64 * d_b, d_b, kc * 128, Bs, tid);      // B[k = b][n = a] = dW'[a][b]
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    tile_mac(As, Bs, ty, tx, acc);
    const float* dch = dc ? dc + h * DA : nullptr;
    if (qrole) {
        if (dWq != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4* o = reinterpret_cast<float4*>(dWq + (int64_t)(r0 + 4 * ty + i) * ldgq + nt * 64 + 4 * tx);
                float4 v = *o;
                v.x += acc[i][0]; v.y += acc[i][1]; v.z += acc[i][2]; v.w += acc[i][3];
                *o = v;
            }
        }
        if (nt == 0 && tid < 64 && dbq != nullptr && dch != nullptr) {      // As[a][m] = W_k[r0 + m][a]
            float s_ = 0.f;
            for (int a = 0; a < DA; ++a) s_ = fmaf(As[a * TS + tid], dch[a], s_);
            dbq[r0 + tid] += s_;
        }
    } else if (dWk != nullptr) {
        const bool bias = kc == 0 && bq != nullptr && dch != nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * ty + i;
            const float bqr = bias ? bq[r] : 0.f;
            float* o = dWk + (int64_t)r * ldgk + at * 64 + 4 * tx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[i][j] + (bias ? bqr * dch[at * 64 + 4 * tx + j] : 0.f);
                if (nc > 1) atomicAdd(o + j, v);
                else o[j] += v;
            }
        }
    }
}

}  // namespace

extern "C" int bmt_rank_prep(const float* Wq, int64_t ldq, int d_b, const float* Wk, int64_t ldk, const float* bq, int H, int dk, int d_a, uint16_t* wp_bf16,
                             uint16_t* wp_f16, uint16_t* wp_f16_lo, int64_t ldp, float* wp_f32, float* c, float* dWp_zero, void* stream) {
    BMT_CHECK_ARG(Wq && Wk && H > 0 && dk > 0 && dk % 128 == 0 && d_a > 0 && d_a % 64 == 0 && d_b > 0 && d_b % 64 == 0 && ldq >= d_b && ldk >= d_a &&
                      ((ldq | ldk | ldp) & 3) == 0 && ((((uintptr_t)Wq) | ((uintptr_t)Wk) | ((uintptr_t)wp_f32) | ((uintptr_t)dWp_zero)) & 15) == 0 &&
                      ((((uintptr_t)wp_bf16) | ((uintptr_t)wp_f16) | ((uintptr_t)wp_f16_lo)) & 7) == 0 && (wp_bf16 || wp_f16 || wp_f32) &&
                      (!wp_f16_lo || wp_f16) && ldp >= d_b,
                  "bmt_rank_prep: bad arguments (H=%d dk=%d d_a=%d d_b=%d: dk a multiple of 128, d_a and d_b of 64, 16-byte aligned rows)", H, dk, d_a, d_b);
    constexpr int lds = (2 * 128 * TS + 128) * (int)sizeof(float);
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)rank_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    hipLaunchKernelGGL(rank_prep_kernel, dim3(H * (d_a / 64) * (d_b / 64)), dim3(256), lds, (hipStream_t)stream, Wq, ldq, d_b, Wk, ldk, bq, dk, d_a, wp_bf16,
                       wp_f16, wp_f16_lo, ldp, wp_f32, c, dWp_zero);
    BMT_CHECK_LAUNCH("bmt_rank_prep");
    return BMT_OK;
}

extern "C" int bmt_rank_chain(const float* Wq, int64_t ldq, int d_b, const float* Wk, int64_t ldk, const float* bq, int H, int dk, int d_a, const float* dWp,
                              const float* dc, float* dWq, int64_t ldgq, float* dWk, int64_t ldgk, float* dbq, void* stream) {
    BMT_CHECK_ARG(Wq && Wk && dWp && H > 0 && dk > 0 && dk % 64 == 0 && d_a == DA && d_b > 0 && d_b % 128 == 0 && ldq >= d_b && ldk >= d_a && ldgq >= d_b &&
                      ldgk >= d_a && ((((uintptr_t)dWp) | ((uintptr_t)Wq) | ((uintptr_t)Wk) | ((uintptr_t)dWq)) & 15) == 0 && ((ldq | ldk | ldgq) & 3) == 0,
                  "bmt_rank_chain: bad arguments (H=%d dk=%d d_a=%d d_b=%d: d_a must be %d, d_b a multiple of 128, dk of 64; 16-byte aligned rows)", H, dk, d_a,
                  d_b, DA);
    constexpr int lds = 2 * DA * TS * (int)sizeof(float);
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)rank_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    const int blocks = H * (dk / 64) * (d_b / 64) + H * (dk / 64) * 2 * (d_b / 128);
    hipLaunchKernelGGL(rank_chain_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, Wq, ldq, d_b, Wk, ldk, bq, H, dk, dWp, dc, dWq, ldgq, dWk, ldgk, dbq);
    BMT_CHECK_LAUNCH("bmt_rank_chain");
    return BMT_OK;
}
