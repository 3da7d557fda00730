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


// LDS slot (16 bytes) of (row, piece) in a wave's [32 rows][8 pieces] staging image (gemm_small_kernel's swizzle)
__device__ __forceinline__ int raw_slot(int row, int s) { return row * 8 + (s ^ ((row >> 1) & 7)); }

// One wave's share of a product  C[32][32 n] = A[32][64 nch] . B[32 n + j][64 nch]^T  for the tiles n = w, w + 8, ... < ntiles:
//   A  the workgroup's 32-row operand in LDS (16-bit, a_stride 32-bit words per row);
//   B  rows of a plane behind the buffer descriptor rs (row r at byte r * b_row_bytes, the reduction index contiguous; rows >= b_rows read as zeros),
//      staged 64 reduction indices at a time through the wave's own 4-KB LDS area: coalesced 16-byte loads (eight lanes per 128-byte piece
//      of a row), NS chunks in flight in NS register sets (one CU streams bytes-in-flight / latency: with two sets ~45 GB/s, and a video
//      workgroup streams 1 MB), no barrier (gemm_small_kernel's scheme: a first version fetched the MFMA
//      fragments straight from the planes, 64 scattered 16-byte requests per instruction, and was slower than the three launches it replaces);
//   done(n, acc)  consumes a finished tile (the wave's LDS area is free by then).
// The (tile, chunk) pairs of the wave are one flat sequence, so the loads of the next tile are in flight under the last chunk of this one.
template <bool F16, int NS, int X3, typename Done>
__device__ __forceinline__ void raw_wave_product(const uint32_t* As, int a_stride, const __amdgpu_buffer_rsrc_t rs, int b_row_bytes, int b_rows, int ntiles,
                                                 int nch, char* wbase, int w, int lane, Done done, const uint32_t* As2 = nullptr,
                                                 const __amdgpu_buffer_rsrc_t rs2 = __amdgpu_buffer_rsrc_t()) {
    // X3 (a split-bf16 product: A2 = the low plane of A, rs2 = the low plane of B):
    //   1  the reduction runs twice into the same accumulator: over B with BOTH planes of A (one staged chunk of B feeds two MFMA chains), then over
    //      B2 with A;
    //   2  gemm_small_kernel's order, bit for bit: chunk by chunk both planes of B are staged (two register sets), and per 16 reduction indices the
    //      accumulator takes A2 . B, A . B2, A . B -- a caller that replaces a launch of that kernel hands on the same bits
    constexpr int OOB = 0x7ffffff0, NSEG = X3 ? 2 : 1;
    static_assert(X3 != 2 || NS % 2 == 0, "the interleaved form pairs the register sets");
    const int half = lane >> 5, l31 = lane & 31, lrow = lane >> 3, piece = lane & 7;
    const int ntw = w < ntiles ? (ntiles - w + 7) >> 3 : 0, total = ntw * nch * NSEG;
    if (total == 0) return;
    int lds_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_w[i] = raw_slot(i * 8 + lrow, piece) * 16;
    u32x4 rg[NS][4];
    int lt = w, lc = 0, lseg = 0, li = 0;                      // the next chunk to request: tile, chunk, segment, flat index
#define BMT_RA_LOAD(set_)                                                                              \
    do {                                                                                               \
        const bool in_ = li < total;                        /* wave-uniform */                         \
        const __amdgpu_buffer_rsrc_t rs_ = (X3 && lseg == 1) ? rs2 : rs;                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int r_ = lt * 32 + i * 8 + lrow;                                                     \
            const int vo_ = (in_ && r_ < b_rows) ? r_ * b_row_bytes + piece * 16 : OOB;                \
            rg[set_][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_, vo_, lc * 128, 0);                \
        }                                                                                              \
        ++li;                                                                                          \
        if constexpr (X3 == 2) {                            /* the plane is the inner index */         \
            if (++lseg == 2) {                                                                         \
                lseg = 0;                                                                              \
                if (++lc == nch) { lc = 0; lt += 8; }                                                  \
            }                                                                                          \
        } else {                                                                                       \
            if (++lc == nch) {                                                                         \
                lc = 0;                                                                                \
                if (++lseg == NSEG) { lseg = 0; lt += 8; }                                             \
            }                                                                                          \
        }                                                                                              \
    } while (0)
#pragma unroll
    for (int j = 0; j < NS; ++j) BMT_RA_LOAD(j);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int ct = w, cc = 0, cseg = 0;
    const uint32_t* ar = As + l31 * a_stride + half * 4;
    const uint32_t* ar2 = X3 ? As2 + l31 * a_stride + half * 4 : ar;
    bf16x8 fbh[4];                                            // (X3 == 2: the high plane's fragments of the chunk, kept while the low plane is staged)
    for (int idx = 0; idx < total; idx += NS) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(wbase + lds_w[i]) = rg[j][i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bf16x8 fa[4], fb[4], fa2[4];
            if constexpr (X3 == 2) {
                if ((j & 1) == 0) {                           // the high plane of B: fragments out, nothing else
#pragma unroll
                    for (int u = 0; u < 4; ++u) fbh[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + raw_slot(l31, 2 * u + half) * 16));
                    BMT_RA_LOAD(j);
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fb[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + raw_slot(l31, 2 * u + half) * 16));      // the low plane
                    fa[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(ar + cc * 32 + u * 8));
                    fa2[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(ar2 + cc * 32 + u * 8));
                }
                BMT_RA_LOAD(j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = mfma32t<F16>(fa2[u], fbh[u], acc);
                    acc = mfma32t<F16>(fa[u], fb[u], acc);
                    acc = mfma32t<F16>(fa[u], fbh[u], acc);
                }
                if (++cc == nch) {
                    if (idx + j < total) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        done(ct, acc);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    cc = 0;
                    ct += 8;
                }
            } else {
                const bool both = X3 && cseg == 0;            /* wave-uniform */
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fb[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + raw_slot(l31, 2 * u + half) * 16));
                    fa[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(ar + cc * 32 + u * 8));
                }
                if (both) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) fa2[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(ar2 + cc * 32 + u * 8));
                }
                BMT_RA_LOAD(j);
                __builtin_amdgcn_sched_barrier(0);
                if (both) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = mfma32t<F16>(fa2[u], fb[u], acc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma32t<F16>(fa[u], fb[u], acc);
                if (++cc == nch) {
                    cc = 0;
                    if (++cseg == NSEG) {
                        if (idx + j < total) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            done(ct, acc);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        cseg = 0;
                        ct += 8;
                    }
                }
            }
        }
    }
#undef BMT_RA_LOAD
}

// The row operation of raw_attn_kernel on a wave's four rows of the score tile in LDS (row t at Ss + t * lds_s, fp32 bits): raw_softmax_fwd_kernel's
// / raw_softmax_bwd_kernel's arithmetic; P (fp16 -> p_f16, bf16 -> the stack) or dS (bf16 -> the stack) to memory, and the 16-bit row -- the A operand
// of the second product -- over the head of its own fp32 row.  NG groups of 8 keys per lane; HALF: a row per half wave (Skp <= 256).
template <bool BWD, int NG, bool HALF>
__device__ __forceinline__ void raw_row_op(uint32_t* Ss, int lds_s, int w, int l, int b, int h, int H, int Tq, int Skp, int len, float scale, uint16_t* p_f16,
                                           uint16_t* stk, int64_t s_sb, int64_t s_sh) {
    constexpr int NP = HALF ? 2 : 4;                          // passes over the wave's four rows
    constexpr int LW = HALF ? 32 : 64;                        // lanes of a row
    const int ll = HALF ? (l & 31) : l;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int t = w * 4 + (HALF ? i * 2 + (l >> 5) : i);
        uint32_t* s = Ss + t * lds_s;
        const int64_t row = (int64_t)(b * H + h) * 32 + t;
        uint16_t* pf = p_f16 + row * Skp;
        uint16_t* sk = stk ? stk + b * s_sb + h * s_sh + (int64_t)t * Skp : nullptr;
        const bool live = t < Tq && len > 0;
        float v[NG][8];
        if constexpr (!BWD) {
            float m = -INFINITY;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int k0 = (ll + LW * g) * 8;
                u32x4 a0 = {0u, 0u, 0u, 0u}, a1_ = a0;
                if (live && k0 < len) {                       // (a group that straddles the length: its tail may hold anything)
                    a0 = *reinterpret_cast<const u32x4*>(s + k0);
                    a1_ = *reinterpret_cast<const u32x4*>(s + k0 + 4);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    v[g][q] = (live && k0 + q < len) ? __uint_as_float(q < 4 ? a0[q & 3] : a1_[q & 3]) : -INFINITY;
                    m = fmaxf(m, v[g][q]);
                }
            }
#pragma unroll
            for (int o = LW / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float sc = scale * 1.4426950408889634f;    // exp(x * scale) = exp2(x * scale * log2 e)
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    v[g][q] = (live && (ll + LW * g) * 8 + q < len) ? exp2f((v[g][q] - m) * sc) : 0.f;
                    sum += v[g][q];
                }
#pragma unroll
            for (int o = LW / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float inv = live ? 1.f / sum : 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int k0 = (ll + LW * g) * 8;
                if (k0 >= Skp) continue;
                u32x4 f, bfv;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float p0 = v[g][2 * q] * inv, p1 = v[g][2 * q + 1] * inv;
                    f[q] = pack_h2(p0, p1);
                    bfv[q] = pack_bf2(p0, p1);
                }
                *reinterpret_cast<u32x4*>(pf + k0) = f;
                if (sk) *reinterpret_cast<u32x4*>(sk + k0) = bfv;
                *reinterpret_cast<u32x4*>(s + (k0 >> 1)) = f;                 // the A operand of the second product, in place
            }
        } else {
            float pv[NG][8];
            float delta = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int k0 = (ll + LW * g) * 8;
                u32x4 ph = {0u, 0u, 0u, 0u}, a0 = ph, a1_ = ph;
                if (live && k0 < len) {
                    ph = *reinterpret_cast<const u32x4*>(pf + k0);
                    a0 = *reinterpret_cast<const u32x4*>(s + k0);
                    a1_ = *reinterpret_cast<const u32x4*>(s + k0 + 4);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool in = live && k0 + q < len;
                    pv[g][q] = in ? h_bits2f((ph[q >> 1] >> (16 * (q & 1))) & 0xffffu) : 0.f;
                    v[g][q] = in ? __uint_as_float(q < 4 ? a0[q & 3] : a1_[q & 3]) : 0.f;
                    delta += pv[g][q] * v[g][q];
                }
            }
#pragma unroll
            for (int o = LW / 2; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int k0 = (ll + LW * g) * 8;
                if (k0 >= Skp) continue;
                u32x4 o_;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o_[q] = pack_bf2(pv[g][2 * q] * (v[g][2 * q] - delta) * scale, pv[g][2 * q + 1] * (v[g][2 * q + 1] - delta) * scale);
                if (sk) *reinterpret_cast<u32x4*>(sk + k0) = o_;
                *reinterpret_cast<u32x4*>(s + (k0 >> 1)) = o_;
            }
        }
    }
}

// ---- the three steps between the block products as ONE launch per attention (round 6): workgroup = (sample, head), 8 waves.
//   forward   S = Q'_h X^T (fp16, one pass)  ->  P = softmax(S * scale) over the sample's keys  ->  O'_h = P X            (X^T from xt_f16)
//   backward  dP = dO'_h X^T (bf16)          ->  dS = P o (dP - rowsum(P o dP)) * scale         ->  dQ'_h = dS (X - mean)  (from xtc_bf)
// The 32 x Skp score tile never leaves LDS (the unfused form wrote it as fp32 and read it back: 2 x B H 32 Skp 4 bytes per attention and
// two dependent launches on the decoder's one-kernel-in-flight chain).  Same arithmetic as the three launches it replaces: the products
// accumulate over the reduction index in ascending 16-element steps into one fp32 accumulator (what gemm_small_kernel does), the row
// operations are raw_softmax_fwd_kernel's / raw_softmax_bwd_kernel's own, P (fp16 for the backward, bf16 for the memory's gradient) and dS
// (bf16) are written where those kernels wrote them.
// LDS: A [32][dm + 8] 16-bit | S [32][Skp + 4] fp32 -- the 16-bit P / dS row overwrites the head of its own fp32 row (one wave owns a row) --
// | 8 x 4 KB: the waves' staging areas for the memory's rows (raw_wave_product).
// All LDS traffic goes through 32-bit unsigned types (the in-place conversion must not be reordered under type-based aliasing).
// EDGES: the block products either side of the three steps in the same launch --
//   backward, in front   dO'_h = do_h W_v,h  (bf16, reduction over d_k): the A operand of the first product, computed into LDS instead of fetched, and
//                        written to the memory gradient's B stack as the unfused product wrote it (32 rows per (sample, head); rows t >= Tq zeros);
//   backward, behind     dq_h = dQ'_h W_k,h^T (bf16, reduction over dm) + its column sums (db_q): dQ'_h stays in the A operand's LDS area (free after
//                        the first product) beside its copy in memory (the operand of dW_k);
//   forward, in front    Q'_h = q_h W_k,h (split-bf16: three passes over d_k into one accumulator) -> fp16 into the A area (its copy in memory is gone:
//                        nobody else read it) and bf16 into the B stack.
// Two launches fewer per attention backward and one per forward on the decoder's chain.
struct RawEdges {
    const uint16_t* in_hi; const uint16_t* in_lo; int64_t ld_in;      // [M][ld_in] bf16, this head's columns at h * dk: do (backward) / q hi, lo (forward)
    const uint16_t* w_hi; const uint16_t* w_lo; int64_t ld_w;         // row d of dm: W[h dk + k][d] at d * ld_w + h * dk + k -- W_v^T (backward) / W_k^T hi, lo (forward)
    uint16_t* bst; int64_t bst_sb, bst_sh;         // the in-front product (bf16) -> bst + b * bst_sb + h * bst_sh + t * dm
    const uint16_t* wk; int64_t ld_wk;             // backward: row h dk + n of W_k's plane, dm contiguous
    uint16_t* dq; int64_t ld_dq;                   // backward: dq [M][ld_dq] bf16, column h dk + n
    float* dbq;                                    // backward: [H dk] += column sums of dq over the rows that exist (or null)
    int dk;
    // forward, optional (y_hi != null): the query projection in front of the in-front product, q_h = y W_q,h^T + b_q,h (split-bf16) -- then in_hi is
    // an OUTPUT (q's high plane, what the backward reads), in_lo is unused
    const uint16_t* y_hi; const uint16_t* y_lo; int64_t ld_y; int Kq;      // y [M][ld_y], Kq = its padded width (a multiple of 64, zeros past the true width)
    const uint16_t* wq_hi; const uint16_t* wq_lo; int64_t ld_wq;           // W_q's planes: row h dk + n, Kq contiguous
    const float* bq;                                                        // [H dk] or null
    // backward, optional (y_hi != null): the out-projection's dX in front of everything, do_h = dropout_mask(dy W_o[:, h dk ...]) (bf16, one pass): y_hi = dy's
    // plane [M][ld_y] (Kq its padded width), wq_hi = W_o^T's plane (row h dk + n, Kq contiguous) -- then in_hi is an OUTPUT (do's plane, the operand of
    // dW_v), its column sums are ADDED to dbv [H dk] (or null), and the attention-output dropout of the forward (p, site, the {seed, step} pair at rng; element
    // index row * in_ld + column) is re-applied
    float* dbv; float drop_p; const uint64_t* rng; uint32_t site;
};

// a finished 32 x 32 fp32 tile through the wave's 4-KB area ([32][32] fp32, 16-byte granules XOR-swizzled by the row): lane -> rows
// t = (l >> 2) + 16 it, 8 consecutive columns (l & 3) * 8; f(it, t, v0, v1)
template <typename F>
__device__ __forceinline__ void raw_tile_rows(float* mine, const f32x16& acc, int l, F f) {
    const int lr = l & 31, half = l >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, half);
        mine[t * 32 + (((lr >> 2) ^ (t & 7)) << 2) + (lr & 3)] = acc[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int t = it * 16 + (l >> 2), g = (l & 3) * 2;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(mine + t * 32 + ((g ^ (t & 7)) << 2));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(mine + t * 32 + (((g + 1) ^ (t & 7)) << 2));
        f(it, t, v0, v1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <bool BWD, bool EDGES = false>
__global__ __launch_bounds__(512) void raw_attn_kernel(const uint16_t* __restrict__ A1, int64_t a_sb, int64_t a_sh, int64_t a_ld,
                                                        const uint16_t* __restrict__ X, int64_t ldx, const int* __restrict__ off,
                                                        const uint16_t* __restrict__ XT, uint16_t* p_f16, uint16_t* __restrict__ stk, int64_t s_sb,
                                                        int64_t s_sh, uint16_t* __restrict__ o_hi, uint16_t* __restrict__ o_lo, int64_t ldo, int H, int Tq,
                                                        int dm, int Skp, float scale, const RawEdges eg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 31, half = l >> 5;
    // (the H workgroups of a sample on ONE XCD, in consecutive dispatch slots: the sample's rows come over the fabric once, not once per head)
    const int wg = xcd_remap(blockIdx.x, gridDim.x), b = wg / H, h = wg - b * H;
    const int r0 = off[b], len = min(off[b + 1] - r0, Skp);
    const int lda_s = dm + 8;                                  // 16-bit elements per A row
    const int lds_s = Skp + 4;                                 // 32-bit elements per S row (rows 4 banks apart: the second product's 16-byte fragment reads of 16 rows are conflict-free)
    uint32_t* As = reinterpret_cast<uint32_t*>(smem);
    uint32_t* Ss = reinterpret_cast<uint32_t*>(smem + (size_t)32 * lda_s * 2);
    // (the forward's edge product parks TWO planes of q_h in the score tile's area: hi | lo)
    // (... and the query projection in front of it two planes of y: in the A area where they fit, else behind q_h's)
    const size_t d2_bytes = !EDGES ? 0 : (size_t)(BWD ? 1 : 2) * 32 * (eg.dk + 8) * 2, y_bytes = (EDGES && eg.y_hi) ? (size_t)(BWD ? 1 : 2) * 32 * (eg.Kq + 8) * 2 : 0;
    const bool y_in_a = y_bytes <= (size_t)32 * lda_s * 2;
    const size_t ss_bytes = max((size_t)32 * lds_s * 4, d2_bytes + (y_in_a ? 0 : y_bytes));
    char* const wbase = smem + (size_t)32 * lda_s * 2 + ss_bytes + w * 4096;
    if constexpr (EDGES) {
        // ---- do_h / q_h hi, lo (32 x d_k, rows t >= Tq zeros) -> LDS (the score tile's area, free until the first product), then the in-front
        // product dO'_h = do_h W_v,h / Q'_h = q_h W_k,h -> the A area + the B stack
        const int dk = eg.dk, ldd_s = dk + 8, pc = dk >> 3;
        uint32_t* const D2 = Ss + 16 * ldd_s;                  // (the low plane: 32 rows of ldd_s 16-bit elements further)
        const int64_t ro = (int64_t)b * Tq * eg.ld_in + (int64_t)h * dk;
        const bool projected = eg.y_hi != nullptr;
        if (projected) {
            if constexpr (BWD) {
                // ---- do_h = mask(dy W_o[:, h dk ...]) from the sample's 32 rows of dy: its plane into LDS, the product (bf16) where the loaded do_h would
                // have gone and to memory (the operand of dW_v); the forward's attention-output dropout re-applied, column sums -> db_v
                const int Kq = eg.Kq, ldy_s = Kq + 8, pcq = Kq >> 3;
                uint32_t* const Y1 = y_in_a ? As : Ss + (d2_bytes >> 2);
                const int64_t yo = (int64_t)b * Tq * eg.ld_y;
                for (int i = tid; i < 32 * pcq; i += 512) {
                    const int t = i / pcq, c = (i - t * pcq) * 8;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (t < Tq) v = *reinterpret_cast<const u32x4*>(eg.y_hi + yo + (int64_t)t * eg.ld_y + c);
                    *reinterpret_cast<u32x4*>(Y1 + ((t * ldy_s + c) >> 1)) = v;
                }
                __syncthreads();
                const int qbytes = (int)((((int64_t)dk - 1) * eg.ld_wq + Kq) * 2);
                const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(eg.wq_hi + (int64_t)h * dk * eg.ld_wq), 0, qbytes, 0x00020000);
                uint16_t* const dout = const_cast<uint16_t*>(eg.in_hi);
                const DropCtx dc = make_drop(eg.drop_p, eg.rng, eg.site);
                raw_wave_product<false, 4, 0>(Y1, ldy_s >> 1, rsQ, (int)(eg.ld_wq * 2), dk, dk >> 5, Kq >> 6, wbase, w, l, [&](int n, const f32x16& acc) {
                    const int c = n * 32 + (l & 3) * 8;
                    float cs[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) cs[q] = 0.f;
                    raw_tile_rows(reinterpret_cast<float*>(wbase), acc, l, [&](int, int t, const f32x4& v0, const f32x4& v1) {
                        const bool on = t < Tq;
                        const uint64_t e0 = (uint64_t)((int64_t)(b * Tq + t) * eg.ld_in + (int64_t)h * dk + c);
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            v[q] = on ? drop_apply(dc, q < 4 ? v0[q & 3] : v1[q & 3], e0 + q) : 0.f;
                            cs[q] += v[q];
                        }
                        const u32x4 hv = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                        *reinterpret_cast<u32x4*>(Ss + ((t * ldd_s + c) >> 1)) = hv;
                        if (on) *reinterpret_cast<u32x4*>(dout + ro + (int64_t)t * eg.ld_in + c) = hv;
                    });
                    if (eg.dbv) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
#pragma unroll
                            for (int o = 4; o < 64; o <<= 1) cs[q] += __shfl_xor(cs[q], o, 64);
                        }
                        if (l < 4) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) atomicAdd(eg.dbv + h * dk + c + q, cs[q]);
                        }
                    }
                });
            }
            if constexpr (!BWD) {
                // ---- q_h = y W_q,h^T + b_q,h (split-bf16) from the sample's 32 rows of y: its two planes into LDS, the product's hi / lo planes where
                // the loaded q_h would have gone, the high plane to memory too (the backward's operand of dW_k)
                const int Kq = eg.Kq, ldy_s = Kq + 8, pcq = Kq >> 3;
                uint32_t* const Y1 = y_in_a ? As : Ss + (d2_bytes >> 2);
                uint32_t* const Y2 = Y1 + 16 * ldy_s;
                const int64_t yo = (int64_t)b * Tq * eg.ld_y;
                for (int i = tid; i < 32 * pcq; i += 512) {
                    const int t = i / pcq, c = (i - t * pcq) * 8;
                    u32x4 v = {0u, 0u, 0u, 0u}, v2 = v;
                    if (t < Tq) {
                        v = *reinterpret_cast<const u32x4*>(eg.y_hi + yo + (int64_t)t * eg.ld_y + c);
                        v2 = *reinterpret_cast<const u32x4*>(eg.y_lo + yo + (int64_t)t * eg.ld_y + c);
                    }
                    *reinterpret_cast<u32x4*>(Y1 + ((t * ldy_s + c) >> 1)) = v;
                    *reinterpret_cast<u32x4*>(Y2 + ((t * ldy_s + c) >> 1)) = v2;
                }
                __syncthreads();
                const int qbytes = (int)((((int64_t)dk - 1) * eg.ld_wq + Kq) * 2);
                const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(eg.wq_hi + (int64_t)h * dk * eg.ld_wq), 0, qbytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsQ2 = __builtin_amdgcn_make_buffer_rsrc((void*)(eg.wq_lo + (int64_t)h * dk * eg.ld_wq), 0, qbytes, 0x00020000);
                uint16_t* const qout = const_cast<uint16_t*>(eg.in_hi);
                raw_wave_product<false, 4, 2>(Y1, ldy_s >> 1, rsQ, (int)(eg.ld_wq * 2), dk, dk >> 5, Kq >> 6, wbase, w, l, [&](int n, const f32x16& acc) {
                    const int c = n * 32 + (l & 3) * 8;
                    float bb[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) bb[q] = eg.bq ? eg.bq[h * dk + c + q] : 0.f;
                    raw_tile_rows(reinterpret_cast<float*>(wbase), acc, l, [&](int, int t, const f32x4& v0, const f32x4& v1) {
                        const bool on = t < Tq;                 // (rows t >= Tq stay zeros: the bias must not reach them)
                        uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
                        split_bf2(on ? v0[0] + bb[0] : 0.f, on ? v0[1] + bb[1] : 0.f, h0, l0);
                        split_bf2(on ? v0[2] + bb[2] : 0.f, on ? v0[3] + bb[3] : 0.f, h1, l1);
                        split_bf2(on ? v1[0] + bb[4] : 0.f, on ? v1[1] + bb[5] : 0.f, h2, l2);
                        split_bf2(on ? v1[2] + bb[6] : 0.f, on ? v1[3] + bb[7] : 0.f, h3, l3);
                        const u32x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
                        *reinterpret_cast<u32x4*>(Ss + ((t * ldd_s + c) >> 1)) = hi;
                        *reinterpret_cast<u32x4*>(D2 + ((t * ldd_s + c) >> 1)) = lo;
                        if (on) *reinterpret_cast<u32x4*>(qout + ro + (int64_t)t * eg.ld_in + c) = hi;
                    });
                }, Y2, rsQ2);
            }
        } else {
            for (int i = tid; i < 32 * pc; i += 512) {
                const int t = i / pc, c = (i - t * pc) * 8;
                u32x4 v = {0u, 0u, 0u, 0u}, v2 = v;
                if (t < Tq) {
                    v = *reinterpret_cast<const u32x4*>(eg.in_hi + ro + (int64_t)t * eg.ld_in + c);
                    if constexpr (!BWD) v2 = *reinterpret_cast<const u32x4*>(eg.in_lo + ro + (int64_t)t * eg.ld_in + c);
                }
                *reinterpret_cast<u32x4*>(Ss + ((t * ldd_s + c) >> 1)) = v;
                if constexpr (!BWD) *reinterpret_cast<u32x4*>(D2 + ((t * ldd_s + c) >> 1)) = v2;
            }
        }
        __syncthreads();
        const int wbytes = (int)((((int64_t)dm - 1) * eg.ld_w + dk) * 2);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(eg.w_hi + (int64_t)h * dk), 0, wbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)((BWD ? eg.w_hi : eg.w_lo) + (int64_t)h * dk), 0, wbytes, 0x00020000);
        uint16_t* bs = eg.bst + b * eg.bst_sb + h * eg.bst_sh;
        raw_wave_product<false, 4, BWD ? 0 : 1>(Ss, ldd_s >> 1, rsW, (int)(eg.ld_w * 2), dm, dm >> 5, dk >> 6, wbase, w, l, [&](int n, const f32x16& acc) {
            raw_tile_rows(reinterpret_cast<float*>(wbase), acc, l, [&](int, int t, const f32x4& v0, const f32x4& v1) {
                const u32x4 bv = {pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]), pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3])};
                const int c = n * 32 + (l & 3) * 8;
                if constexpr (BWD) {
                    *reinterpret_cast<u32x4*>(As + ((t * lda_s + c) >> 1)) = bv;
                } else {
                    const u32x4 hv = {pack_h2(v0[0], v0[1]), pack_h2(v0[2], v0[3]), pack_h2(v1[0], v1[1]), pack_h2(v1[2], v1[3])};
                    *reinterpret_cast<u32x4*>(As + ((t * lda_s + c) >> 1)) = hv;
                }
                *reinterpret_cast<u32x4*>(bs + (int64_t)t * dm + c) = bv;
            });
        }, D2, rsW2);
    } else {
    // ---- A (the 32 query rows of this head; rows t >= Tq are zeros) -> LDS
        const uint16_t* a1 = A1 + b * a_sb + h * a_sh;
        const int pc = dm >> 3;
        for (int i = tid; i < 32 * pc; i += 512) {
            const int t = i / pc, c = (i - t * pc) * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (t < Tq) v = *reinterpret_cast<const u32x4*>(a1 + (int64_t)t * a_ld + c);
            *reinterpret_cast<u32x4*>(As + ((t * lda_s + c) >> 1)) = v;
        }
    }
    __syncthreads();
    // ---- first product: a wave per 32 keys (keys past the sample's length read as zeros: the row operation masks them)
    {
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (int64_t)r0 * ldx), 0, (int)((int64_t)len * ldx * 2), 0x00020000);
        raw_wave_product<!BWD, 4, 0>(As, lda_s >> 1, rsX, (int)(ldx * 2), len, (len + 31) >> 5, dm >> 6, wbase, w, l, [&](int n, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ss[acc_row(r, half) * lds_s + n * 32 + lr] = __float_as_uint(acc[r]);
        });
    }
    __syncthreads();
    // ---- the row operation: four rows per wave.  The lanes of a row hold groups of 8 consecutive keys; with Skp <= 256 a row is HALF a wave (two
    // rows per pass), with Skp <= 512 one group per lane: the VALU work follows the keys that exist (a lane's 16 masked keys cost what live ones do)
    if (Skp <= 256) raw_row_op<BWD, 1, true>(Ss, lds_s, w, l, b, h, H, Tq, Skp, len, scale, p_f16, stk, s_sb, s_sh);
    else if (Skp <= 512) raw_row_op<BWD, 1, false>(Ss, lds_s, w, l, b, h, H, Tq, Skp, len, scale, p_f16, stk, s_sb, s_sh);
    else raw_row_op<BWD, 2, false>(Ss, lds_s, w, l, b, h, H, Tq, Skp, len, scale, p_f16, stk, s_sb, s_sh);
    __syncthreads();
    // ---- second product: a wave per 32 columns of the memory; the reduction runs over the sample's keys in chunks of 64 (P / dS and the
    // transposed memory are zero from the length to Skp, a multiple of 64)
    {
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(XT + (int64_t)b * dm * Skp), 0, (int)((int64_t)dm * Skp * 2), 0x00020000);
        raw_wave_product<!BWD, 4, 0>(Ss, lds_s, rsT, Skp * 2, dm, dm >> 5, max(1, (len + 63) >> 6), wbase, w, l, [&](int n, const f32x16& acc) {
            raw_tile_rows(reinterpret_cast<float*>(wbase), acc, l, [&](int, int t, const f32x4& v0, const f32x4& v1) {
                uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
                split_bf2(v0[0], v0[1], h0, l0);
                split_bf2(v0[2], v0[3], h1, l1);
                split_bf2(v1[0], v1[1], h2, l2);
                split_bf2(v1[2], v1[3], h3, l3);
                const u32x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
                const int c = n * 32 + (l & 3) * 8;
                if constexpr (EDGES && BWD) *reinterpret_cast<u32x4*>(As + ((t * lda_s + c) >> 1)) = hi;      // (rows t >= Tq: zeros, dS's are)
                if (t >= Tq) return;
                const int64_t o = (int64_t)(b * Tq + t) * ldo + (int64_t)h * dm + c;
                *reinterpret_cast<u32x4*>(o_hi + o) = hi;
                if (o_lo) *reinterpret_cast<u32x4*>(o_lo + o) = lo;
            });
        });
    }
    if constexpr (EDGES && BWD) {
        // ---- dq_h = dQ'_h W_k,h^T: a wave per 32 columns of the head, the reduction over dm; column sums over the rows that exist -> db_q
        __syncthreads();
        const int dk = eg.dk;
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(eg.wk + (int64_t)h * dk * eg.ld_wk), 0, (int)((((int64_t)dk - 1) * eg.ld_wk + dm) * 2), 0x00020000);
        raw_wave_product<false, 4, 0>(As, lda_s >> 1, rsK, (int)(eg.ld_wk * 2), dk, dk >> 5, dm >> 6, wbase, w, l, [&](int n, const f32x16& acc) {
            float cs[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cs[q] = 0.f;
            const int c = h * dk + n * 32 + (l & 3) * 8;
            raw_tile_rows(reinterpret_cast<float*>(wbase), acc, l, [&](int, int t, const f32x4& v0, const f32x4& v1) {
                if (t >= Tq) return;
                const u32x4 hv = {pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]), pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3])};
                *reinterpret_cast<u32x4*>(eg.dq + (int64_t)(b * Tq + t) * eg.ld_dq + c) = hv;
#pragma unroll
                for (int q = 0; q < 4; ++q) { cs[q] += v0[q]; cs[4 + q] += v1[q]; }
            });
            if (eg.dbq) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int o = 4; o < 64; o <<= 1) cs[q] += __shfl_xor(cs[q], o, 64);
                }
                if (l < 4) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) atomicAdd(eg.dbq + c + q, cs[q]);
                }
            }
        });
    }
}

inline size_t raw_attn_lds(int dm, int Skp) { return (size_t)32 * (dm + 8) * 2 + (size_t)32 * (Skp + 4) * 4 + 8 * 4096; }
inline size_t raw_attn_lds_bwd_proj(int dm, int Skp, int dk, int Kq) {
    const size_t a = (size_t)32 * (dm + 8) * 2, ss = (size_t)32 * (Skp + 4) * 4, d1 = (size_t)32 * (dk + 8) * 2, y = (size_t)32 * (Kq + 8) * 2;
    const size_t need = d1 + (y <= a ? 0 : y);
    return a + (ss > need ? ss : need) + 8 * 4096;
}
inline size_t raw_attn_lds_fwd_edges(int dm, int Skp, int dk, int Kq = 0) {
    const size_t a = (size_t)32 * (dm + 8) * 2, ss = (size_t)32 * (Skp + 4) * 4, d2 = (size_t)2 * 32 * (dk + 8) * 2;
    const size_t y = Kq > 0 ? (size_t)2 * 32 * (Kq + 8) * 2 : 0, need = d2 + (y <= a ? 0 : y);
    return a + (ss > need ? ss : need) + 8 * 4096;
}
// the kernel's dynamic-LDS ceiling only ever grows (one record per instantiation, whichever entry point launches it)
template <bool BWD, bool EDGES>
void raw_attn_set_lds(size_t lds) {
    static size_t lds_set = 0;
    if (lds > lds_set) {
        (void)hipFuncSetAttribute((const void*)raw_attn_kernel<BWD, EDGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set = lds;
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

extern "C" int bmt_raw_attn_ok(int dm, int Skp) {
    return dm > 0 && dm % 64 == 0 && Skp > 0 && Skp % 64 == 0 && Skp <= 1024 && raw_attn_lds(dm, Skp) <= (size_t)160 * 1024;
}

extern "C" int bmt_raw_attn_fwd(const uint16_t* q_f16, int64_t q_sb, int64_t q_sh, int64_t ldq, const uint16_t* x_f16, int64_t ldx, const int* off,
                                const uint16_t* xt_f16, int B, int H, int Tq, int dm, int Skp, float scale, uint16_t* p_f16, uint16_t* p_bf,
                                int64_t p_bf_sb, int64_t p_bf_sh, uint16_t* o_hi, uint16_t* o_lo, int64_t ldo, void* stream) {
    BMT_CHECK_ARG(q_f16 && x_f16 && off && xt_f16 && p_f16 && o_hi && B > 0 && H > 0 && Tq > 0 && Tq <= 32, "bmt_raw_attn_fwd: null pointer or bad extents (at most 32 queries per sample and head)");
    BMT_CHECK_ARG(bmt_raw_attn_ok(dm, Skp), "bmt_raw_attn_fwd: dm and Skp multiples of 64, Skp <= 1024, and 64 (dm + 8) + 128 (Skp + 4) + 32 768 bytes of LDS <= 160 KB (bmt_raw_attn_ok)");
    BMT_CHECK_ARG(!((reinterpret_cast<uintptr_t>(q_f16) | reinterpret_cast<uintptr_t>(x_f16) | reinterpret_cast<uintptr_t>(xt_f16) | reinterpret_cast<uintptr_t>(p_f16) |
                     reinterpret_cast<uintptr_t>(p_bf)) & 15) && !((q_sb | q_sh | ldq | ldx | p_bf_sb | p_bf_sh) & 7) && ldx >= dm && ldo >= (int64_t)H * dm,
                  "bmt_raw_attn_fwd: 16-byte aligned operands, strides that are multiples of 8 elements");
    const size_t lds = raw_attn_lds(dm, Skp);
    raw_attn_set_lds<false, false>(lds);
    hipLaunchKernelGGL(raw_attn_kernel<false>, dim3(B * H), dim3(512), lds, (hipStream_t)stream, q_f16, q_sb, q_sh, ldq, x_f16, ldx, off, xt_f16, p_f16, p_bf, p_bf_sb,
                       p_bf_sh, o_hi, o_lo, ldo, H, Tq, dm, Skp, scale, RawEdges{});
    BMT_CHECK_LAUNCH("bmt_raw_attn_fwd");
    return BMT_OK;
}

extern "C" int bmt_raw_attn_bwd(const uint16_t* do_bf, int64_t do_sb, int64_t do_sh, int64_t lddo, const uint16_t* x_bf, int64_t ldx, const int* off,
                                const uint16_t* xtc_bf, const uint16_t* p_f16, int B, int H, int Tq, int dm, int Skp, float scale, uint16_t* ds_bf,
                                int64_t ds_sb, int64_t ds_sh, uint16_t* dq_bf, int64_t lddq, void* stream) {
    BMT_CHECK_ARG(do_bf && x_bf && off && xtc_bf && p_f16 && dq_bf && B > 0 && H > 0 && Tq > 0 && Tq <= 32, "bmt_raw_attn_bwd: null pointer or bad extents (at most 32 queries per sample and head)");
    BMT_CHECK_ARG(bmt_raw_attn_ok(dm, Skp), "bmt_raw_attn_bwd: dm and Skp multiples of 64, Skp <= 1024, and 64 (dm + 8) + 128 (Skp + 4) + 32 768 bytes of LDS <= 160 KB (bmt_raw_attn_ok)");
    BMT_CHECK_ARG(!((reinterpret_cast<uintptr_t>(do_bf) | reinterpret_cast<uintptr_t>(x_bf) | reinterpret_cast<uintptr_t>(xtc_bf) | reinterpret_cast<uintptr_t>(p_f16) |
                     reinterpret_cast<uintptr_t>(ds_bf)) & 15) && !((do_sb | do_sh | lddo | ldx | ds_sb | ds_sh) & 7) && ldx >= dm && lddq >= (int64_t)H * dm,
                  "bmt_raw_attn_bwd: 16-byte aligned operands, strides that are multiples of 8 elements");
    const size_t lds = raw_attn_lds(dm, Skp);
    raw_attn_set_lds<true, false>(lds);
    hipLaunchKernelGGL(raw_attn_kernel<true>, dim3(B * H), dim3(512), lds, (hipStream_t)stream, do_bf, do_sb, do_sh, lddo, x_bf, ldx, off, xtc_bf,
                       const_cast<uint16_t*>(p_f16), ds_bf, ds_sb, ds_sh, dq_bf, (uint16_t*)nullptr, lddq, H, Tq, dm, Skp, scale, RawEdges{});
    BMT_CHECK_LAUNCH("bmt_raw_attn_bwd");
    return BMT_OK;
}

extern "C" int bmt_raw_attn_edges_ok(int dm, int Skp, int dk) {
    return bmt_raw_attn_ok(dm, Skp) && dk > 0 && dk % 64 == 0 && (size_t)32 * (dk + 8) * 2 <= (size_t)32 * (Skp + 4) * 4;
}

extern "C" int bmt_raw_attn_bwd_edges(const uint16_t* do_bf, int64_t ld_do, const uint16_t* wvT_bf, int64_t ld_wvT, uint16_t* bstack, int64_t b_sb, int64_t b_sh,
                                      const uint16_t* x_bf, int64_t ldx, const int* off, const uint16_t* xtc_bf, const uint16_t* p_f16, int B, int H, int Tq,
                                      int dm, int Skp, int dk, float scale, uint16_t* ds_bf, int64_t ds_sb, int64_t ds_sh, uint16_t* dqp_bf, int64_t lddqp,
                                      const uint16_t* wk_bf, int64_t ld_wk, uint16_t* dq_bf, int64_t ld_dq, float* dbq, void* stream) {
    BMT_CHECK_ARG(do_bf && wvT_bf && bstack && x_bf && off && xtc_bf && p_f16 && dqp_bf && wk_bf && dq_bf && B > 0 && H > 0 && Tq > 0 && Tq <= 32,
                  "bmt_raw_attn_bwd_edges: null pointer or bad extents (at most 32 queries per sample and head)");
    BMT_CHECK_ARG(bmt_raw_attn_edges_ok(dm, Skp, dk), "bmt_raw_attn_bwd_edges: bmt_raw_attn_ok(dm, Skp), d_k a multiple of 64 with 64 (d_k + 8) <= 128 (Skp + 4)");
    BMT_CHECK_ARG(!((reinterpret_cast<uintptr_t>(do_bf) | reinterpret_cast<uintptr_t>(wvT_bf) | reinterpret_cast<uintptr_t>(bstack) | reinterpret_cast<uintptr_t>(x_bf) |
                     reinterpret_cast<uintptr_t>(xtc_bf) | reinterpret_cast<uintptr_t>(p_f16) | reinterpret_cast<uintptr_t>(ds_bf) | reinterpret_cast<uintptr_t>(dqp_bf) |
                     reinterpret_cast<uintptr_t>(wk_bf) | reinterpret_cast<uintptr_t>(dq_bf)) & 15) &&
                      !((ld_do | ld_wvT | b_sb | b_sh | ldx | ds_sb | ds_sh | lddqp | ld_wk | ld_dq) & 7) && ldx >= dm && lddqp >= (int64_t)H * dm && ld_wk >= dm &&
                      ld_do >= (int64_t)H * dk && ld_dq >= (int64_t)H * dk && ld_wvT >= (int64_t)H * dk,
                  "bmt_raw_attn_bwd_edges: 16-byte aligned operands, strides that are multiples of 8 elements and cover their rows");
    const size_t lds = raw_attn_lds(dm, Skp);
    raw_attn_set_lds<true, true>(lds);
    const RawEdges eg{do_bf, nullptr, ld_do, wvT_bf, nullptr, ld_wvT, bstack, b_sb, b_sh, wk_bf, ld_wk, dq_bf, ld_dq, dbq, dk, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, nullptr, 0.f, nullptr, 0u};
    hipLaunchKernelGGL((raw_attn_kernel<true, true>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, x_bf,
                       ldx, off, xtc_bf, const_cast<uint16_t*>(p_f16), ds_bf, ds_sb, ds_sh, dqp_bf, (uint16_t*)nullptr, lddqp, H, Tq, dm, Skp, scale, eg);
    BMT_CHECK_LAUNCH("bmt_raw_attn_bwd_edges");
    return BMT_OK;
}

extern "C" int bmt_raw_attn_fwd_edges_ok(int dm, int Skp, int dk) {
    return bmt_raw_attn_ok(dm, Skp) && dk > 0 && dk % 64 == 0 && raw_attn_lds_fwd_edges(dm, Skp, dk) <= (size_t)160 * 1024;
}

extern "C" int bmt_raw_attn_fwd_edges(const uint16_t* q_hi, const uint16_t* q_lo, int64_t ld_q, const uint16_t* wkT_hi, const uint16_t* wkT_lo, int64_t ld_wkT,
                                      uint16_t* bstack, int64_t b_sb, int64_t b_sh, const uint16_t* x_f16, int64_t ldx, const int* off, const uint16_t* xt_f16,
                                      int B, int H, int Tq, int dm, int Skp, int dk, float scale, uint16_t* p_f16, uint16_t* p_bf, int64_t p_bf_sb,
                                      int64_t p_bf_sh, uint16_t* o_hi, uint16_t* o_lo, int64_t ldo, void* stream) {
    BMT_CHECK_ARG(q_hi && q_lo && wkT_hi && wkT_lo && bstack && x_f16 && off && xt_f16 && p_f16 && o_hi && B > 0 && H > 0 && Tq > 0 && Tq <= 32,
                  "bmt_raw_attn_fwd_edges: null pointer or bad extents (at most 32 queries per sample and head)");
    BMT_CHECK_ARG(bmt_raw_attn_fwd_edges_ok(dm, Skp, dk), "bmt_raw_attn_fwd_edges: bmt_raw_attn_ok(dm, Skp), d_k a multiple of 64, LDS with two planes of q_h <= 160 KB");
    BMT_CHECK_ARG(!((reinterpret_cast<uintptr_t>(q_hi) | reinterpret_cast<uintptr_t>(q_lo) | reinterpret_cast<uintptr_t>(wkT_hi) | reinterpret_cast<uintptr_t>(wkT_lo) |
                     reinterpret_cast<uintptr_t>(bstack) | reinterpret_cast<uintptr_t>(x_f16) | reinterpret_cast<uintptr_t>(xt_f16) | reinterpret_cast<uintptr_t>(p_f16) |
                     reinterpret_cast<uintptr_t>(p_bf) | reinterpret_cast<uintptr_t>(o_hi) | reinterpret_cast<uintptr_t>(o_lo)) & 15) &&
                      !((ld_q | ld_wkT | b_sb | b_sh | ldx | p_bf_sb | p_bf_sh | ldo) & 7) && ldx >= dm && ldo >= (int64_t)H * dm && ld_q >= (int64_t)H * dk &&
                      ld_wkT >= (int64_t)H * dk,
                  "bmt_raw_attn_fwd_edges: 16-byte aligned operands, strides that are multiples of 8 elements and cover their rows");
    const size_t lds = raw_attn_lds_fwd_edges(dm, Skp, dk);
    raw_attn_set_lds<false, true>(lds);
    const RawEdges eg{q_hi, q_lo, ld_q, wkT_hi, wkT_lo, ld_wkT, bstack, b_sb, b_sh, nullptr, 0, nullptr, 0, nullptr, dk, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, nullptr, 0.f, nullptr, 0u};
    hipLaunchKernelGGL((raw_attn_kernel<false, true>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, x_f16,
                       ldx, off, xt_f16, p_f16, p_bf, p_bf_sb, p_bf_sh, o_hi, o_lo, ldo, H, Tq, dm, Skp, scale, eg);
    BMT_CHECK_LAUNCH("bmt_raw_attn_fwd_edges");
    return BMT_OK;
}

extern "C" int bmt_raw_attn_fwd_proj_ok(int dm, int Skp, int dk, int Kq) {
    return bmt_raw_attn_ok(dm, Skp) && dk > 0 && dk % 64 == 0 && Kq > 0 && Kq % 64 == 0 && raw_attn_lds_fwd_edges(dm, Skp, dk, Kq) <= (size_t)160 * 1024;
}

extern "C" int bmt_raw_attn_fwd_proj(const uint16_t* y_hi, const uint16_t* y_lo, int64_t ld_y, int Kq, const uint16_t* wq_hi, const uint16_t* wq_lo, int64_t ld_wq,
                                     const float* bq, uint16_t* q_hi_out, int64_t ld_q, const uint16_t* wkT_hi, const uint16_t* wkT_lo, int64_t ld_wkT,
                                     uint16_t* bstack, int64_t b_sb, int64_t b_sh, const uint16_t* x_f16, int64_t ldx, const int* off, const uint16_t* xt_f16,
                                     int B, int H, int Tq, int dm, int Skp, int dk, float scale, uint16_t* p_f16, uint16_t* p_bf, int64_t p_bf_sb,
                                     int64_t p_bf_sh, uint16_t* o_hi, uint16_t* o_lo, int64_t ldo, void* stream) {
    BMT_CHECK_ARG(y_hi && y_lo && wq_hi && wq_lo && q_hi_out && wkT_hi && wkT_lo && bstack && x_f16 && off && xt_f16 && p_f16 && o_hi && B > 0 && H > 0 && Tq > 0 &&
                      Tq <= 32,
                  "bmt_raw_attn_fwd_proj: null pointer or bad extents (at most 32 queries per sample and head)");
    BMT_CHECK_ARG(bmt_raw_attn_fwd_proj_ok(dm, Skp, dk, Kq), "bmt_raw_attn_fwd_proj: bmt_raw_attn_ok(dm, Skp), d_k and Kq multiples of 64, LDS with the planes of y and q_h <= 160 KB");
    BMT_CHECK_ARG(!((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo) | reinterpret_cast<uintptr_t>(wq_hi) | reinterpret_cast<uintptr_t>(wq_lo) |
                     reinterpret_cast<uintptr_t>(q_hi_out) | reinterpret_cast<uintptr_t>(wkT_hi) | reinterpret_cast<uintptr_t>(wkT_lo) | reinterpret_cast<uintptr_t>(bstack) |
                     reinterpret_cast<uintptr_t>(x_f16) | reinterpret_cast<uintptr_t>(xt_f16) | reinterpret_cast<uintptr_t>(p_f16) | reinterpret_cast<uintptr_t>(p_bf) |
                     reinterpret_cast<uintptr_t>(o_hi) | reinterpret_cast<uintptr_t>(o_lo)) & 15) &&
                      !((ld_y | ld_wq | ld_q | ld_wkT | b_sb | b_sh | ldx | p_bf_sb | p_bf_sh | ldo) & 7) && ldx >= dm && ldo >= (int64_t)H * dm && ld_q >= (int64_t)H * dk &&
                      ld_wkT >= (int64_t)H * dk && ld_y >= Kq && ld_wq >= Kq,
                  "bmt_raw_attn_fwd_proj: 16-byte aligned operands, strides that are multiples of 8 elements and cover their rows");
    const size_t lds = raw_attn_lds_fwd_edges(dm, Skp, dk, Kq);
    raw_attn_set_lds<false, true>(lds);
    const RawEdges eg{q_hi_out, nullptr, ld_q, wkT_hi, wkT_lo, ld_wkT, bstack, b_sb, b_sh, nullptr, 0, nullptr, 0, nullptr, dk, y_hi, y_lo, ld_y, Kq, wq_hi, wq_lo, ld_wq, bq, nullptr, 0.f, nullptr, 0u};
    hipLaunchKernelGGL((raw_attn_kernel<false, true>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, x_f16,
                       ldx, off, xt_f16, p_f16, p_bf, p_bf_sb, p_bf_sh, o_hi, o_lo, ldo, H, Tq, dm, Skp, scale, eg);
    BMT_CHECK_LAUNCH("bmt_raw_attn_fwd_proj");
    return BMT_OK;
}

extern "C" int bmt_raw_attn_bwd_proj_ok(int dm, int Skp, int dk, int Kq) {
    return bmt_raw_attn_edges_ok(dm, Skp, dk) && Kq > 0 && Kq % 64 == 0 && raw_attn_lds_bwd_proj(dm, Skp, dk, Kq) <= (size_t)160 * 1024;
}

extern "C" int bmt_raw_attn_bwd_proj(const uint16_t* dy_bf, int64_t ld_dy, int Kq, const uint16_t* woT_bf, int64_t ld_woT, float drop_p, const uint64_t* rng,
                                     uint32_t site, float* dbv, uint16_t* do_out, int64_t ld_do, const uint16_t* wvT_bf, int64_t ld_wvT, uint16_t* bstack,
                                     int64_t b_sb, int64_t b_sh, const uint16_t* x_bf, int64_t ldx, const int* off, const uint16_t* xtc_bf, const uint16_t* p_f16,
                                     int B, int H, int Tq, int dm, int Skp, int dk, float scale, uint16_t* ds_bf, int64_t ds_sb, int64_t ds_sh, uint16_t* dqp_bf,
                                     int64_t lddqp, const uint16_t* wk_bf, int64_t ld_wk, uint16_t* dq_bf, int64_t ld_dq, float* dbq, void* stream) {
    BMT_CHECK_ARG(dy_bf && woT_bf && do_out && wvT_bf && bstack && x_bf && off && xtc_bf && p_f16 && dqp_bf && wk_bf && dq_bf && B > 0 && H > 0 && Tq > 0 && Tq <= 32 &&
                      drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || (rng && ld_do == (int64_t)H * dk)),
                  "bmt_raw_attn_bwd_proj: null pointer or bad extents (at most 32 queries per sample and head; a dropout rate needs its {seed, step} pair and the "
                  "forward's element index: ld_do = H d_k)");
    BMT_CHECK_ARG(bmt_raw_attn_bwd_proj_ok(dm, Skp, dk, Kq), "bmt_raw_attn_bwd_proj: bmt_raw_attn_edges_ok(dm, Skp, dk), Kq a multiple of 64, LDS with the plane of dy <= 160 KB");
    BMT_CHECK_ARG(!((reinterpret_cast<uintptr_t>(dy_bf) | reinterpret_cast<uintptr_t>(woT_bf) | reinterpret_cast<uintptr_t>(do_out) | reinterpret_cast<uintptr_t>(wvT_bf) |
                     reinterpret_cast<uintptr_t>(bstack) | reinterpret_cast<uintptr_t>(x_bf) | reinterpret_cast<uintptr_t>(xtc_bf) | reinterpret_cast<uintptr_t>(p_f16) |
                     reinterpret_cast<uintptr_t>(ds_bf) | reinterpret_cast<uintptr_t>(dqp_bf) | reinterpret_cast<uintptr_t>(wk_bf) | reinterpret_cast<uintptr_t>(dq_bf)) & 15) &&
                      !((ld_dy | ld_woT | ld_do | ld_wvT | b_sb | b_sh | ldx | ds_sb | ds_sh | lddqp | ld_wk | ld_dq) & 7) && ldx >= dm && lddqp >= (int64_t)H * dm &&
                      ld_wk >= dm && ld_do >= (int64_t)H * dk && ld_dq >= (int64_t)H * dk && ld_wvT >= (int64_t)H * dk && ld_dy >= Kq && ld_woT >= Kq,
                  "bmt_raw_attn_bwd_proj: 16-byte aligned operands, strides that are multiples of 8 elements and cover their rows");
    const size_t lds = raw_attn_lds_bwd_proj(dm, Skp, dk, Kq);
    raw_attn_set_lds<true, true>(lds);
    const RawEdges eg{do_out, nullptr, ld_do, wvT_bf, nullptr, ld_wvT, bstack, b_sb, b_sh, wk_bf, ld_wk, dq_bf, ld_dq, dbq, dk,
                      dy_bf, nullptr, ld_dy, Kq, woT_bf, nullptr, ld_woT, nullptr, dbv, drop_p, rng, site};
    hipLaunchKernelGGL((raw_attn_kernel<true, true>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int64_t)0, x_bf,
                       ldx, off, xtc_bf, const_cast<uint16_t*>(p_f16), ds_bf, ds_sb, ds_sh, dqp_bf, (uint16_t*)nullptr, lddqp, H, Tq, dm, Skp, scale, eg);
    BMT_CHECK_LAUNCH("bmt_raw_attn_bwd_proj");
    return BMT_OK;
}
