// bmt_attn_fwd / bmt_attn_bwd -- flash-style masked attention for gfx950 (CDNA4).
//
// Replaces attention() of the reference (model/multihead_attention.py:8-26): the (B,H,Sq,Sk)
// score tensor -- 327 MB for the 800x800 audio self-attention at config[1], materialised four
// times by the reference -- never exists; scores live in MFMA accumulators.
//
// Everything is computed TRANSPOSED so that the softmax axis is lane-local:
//     S^T[key][q] = K . Q^T        (A = K tile from LDS, B = Q fragments held in VGPRs)
//     O^T[d][q]  += V^T . P^T      (A = V^T tile from LDS, B = P^T straight from the S^T accumulators)
// With v_mfma_f32_32x32x16_bf16 the accumulator of lane l holds column q = l&31 and 16 rows
// (keys, then d): the row max / row sum over keys is an in-lane reduction plus ONE cross-lane
// exchange (lane ^ 32), the running max/sum and the rescale factor are per-lane scalars, and
// the probabilities feed the second MFMA as its B operand without leaving registers.
// The two MFMAs only need the reduction index to be enumerated consistently by both operands,
// so V^T is staged with the key order the S^T accumulator layout produces (4-key groups).
//
//   workgroup = 4 waves x 32 query rows = 128 query rows of one (batch, head); K/V tiles of BC keys
//   are shared through LDS (K row-major [key][d], 16-B slots XOR-swizzled; V transposed [d][key]
//   in 8-B key-quads, quad index XOR-swizzled -- both conflict-free for the MFMA operand reads).
//   fp32 in HBM -> bf16 (hi [+ lo residual]) while staging; BMT_PREC_BF16X3 issues
//   hi*hi + hi*lo + lo*hi for both products (the reference is fp32; plain bf16 operands miss the
//   1e-3 log-prob tolerance, see DESIGN.md "precision").
//   Tiles for step t+1 are fetched into registers while tile t is multiplied (T14 split staging).
#include "common.h"

namespace {

constexpr float NEG_INF = -__builtin_huge_valf();

template <int DK>
struct Geo {
    static constexpr int BC = (DK == 256) ? 32 : 64;   // keys per staged tile
    static constexpr int NSUB = BC / 32;               // 32-key MFMA sub-tiles per staged tile
    static constexpr int SPR = DK / 8;                 // 16-B slots per K row
    static constexpr int KG = BC / 4;                  // 8-B key-quads per V^T row
    static constexpr int DT = DK / 32;                 // 32-row tiles of O^T
    static constexpr int NKS = BC * DK / 8 / 256;      // K slots staged per thread
    static constexpr int NVB = BC * DK / 16 / 256;     // 4x4 V blocks staged per thread
    static constexpr int K_BYTES = BC * DK * 2;
    static constexpr int V_BYTES = DK * BC * 2;
};

template <int DK>
__device__ __forceinline__ int kslot(int row, int slot) {
    if constexpr (DK == 32) return row * 4 + (slot ^ ((row >> 2) & 3));
    else if constexpr (DK == 64) return row * 8 + (slot ^ ((row >> 1) & 7));
    else return row * (DK / 8) + (slot ^ (row & 15));
}
template <int BC>
__device__ __forceinline__ int vunit(int d, int kg) {
    if constexpr (BC == 64) return d * 16 + (kg ^ ((d >> 1) & 15));
    else return d * 8 + (kg ^ ((d >> 2) & 7));
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- staging helpers shared by forward and backward -------------------------------------------
// row-major tile [ROWS][DK] (fp32, row stride ld) -> registers; rows >= nrows read as zero
// per-thread register counts of the two staging shapes (ceil-divided; short tiles leave some threads idle)
template <int DK, int ROWS> constexpr int rows_n() { return (ROWS * DK / 8 + 255) / 256; }
template <int DK, int ROWS> constexpr int rowsT_n() { return (ROWS * DK / 16 + 255) / 256; }

template <int DK, int ROWS>
__device__ __forceinline__ void tile_gload(const float* base, int64_t ld, int row0, int nrows, int tid,
                                           float (&v)[rows_n<DK, ROWS>() * 8]) {
    constexpr int SPR = DK / 8, N = rows_n<DK, ROWS>();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = tid + 256 * i;
        const int row = c / SPR, slot = c % SPR;
        if (c < ROWS * SPR && row0 + row < nrows) {
            const float* src = base + (int64_t)(row0 + row) * ld + slot * 8;
            const float4 x = ldg4(src), y = ldg4(src + 4);
            v[i * 8 + 0] = x.x; v[i * 8 + 1] = x.y; v[i * 8 + 2] = x.z; v[i * 8 + 3] = x.w;
            v[i * 8 + 4] = y.x; v[i * 8 + 5] = y.y; v[i * 8 + 6] = y.z; v[i * 8 + 7] = y.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i * 8 + j] = 0.f;
        }
    }
}
// registers -> LDS row-major swizzled image (K-style)
template <int DK, int ROWS, int NPASS>
__device__ __forceinline__ void tile_lstore_rows(uint4* hi, uint4* lo, int tid, const float (&v)[rows_n<DK, ROWS>() * 8]) {
    constexpr int SPR = DK / 8, N = rows_n<DK, ROWS>();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = tid + 256 * i;
        if (c >= ROWS * SPR) break;
        const int row = c / SPR, slot = c % SPR;
        uint4 h, l;
        if constexpr (NPASS == 3) {
            split_bf2(v[i * 8 + 0], v[i * 8 + 1], h.x, l.x);
            split_bf2(v[i * 8 + 2], v[i * 8 + 3], h.y, l.y);
            split_bf2(v[i * 8 + 4], v[i * 8 + 5], h.z, l.z);
            split_bf2(v[i * 8 + 6], v[i * 8 + 7], h.w, l.w);
            lo[kslot<DK>(row, slot)] = l;
        } else {
            h.x = pack_bf2(v[i * 8 + 0], v[i * 8 + 1]);
            h.y = pack_bf2(v[i * 8 + 2], v[i * 8 + 3]);
            h.z = pack_bf2(v[i * 8 + 4], v[i * 8 + 5]);
            h.w = pack_bf2(v[i * 8 + 6], v[i * 8 + 7]);
        }
        hi[kslot<DK>(row, slot)] = h;
    }
}
// transposed staging: tile [ROWS][DK] -> registers as 4(row) x 4(d) blocks
template <int DK, int ROWS>
__device__ __forceinline__ void tileT_gload(const float* base, int64_t ld, int row0, int nrows, int tid,
                                            float (&v)[rowsT_n<DK, ROWS>() * 16]) {
    constexpr int DQ = DK / 4, N = rowsT_n<DK, ROWS>();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = tid + 256 * i;
        const int dq = c % DQ, kg = c / DQ;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + kg * 4 + r;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < DQ * (ROWS / 4) && row < nrows) x = ldg4(base + (int64_t)row * ld + dq * 4);
            v[i * 16 + r * 4 + 0] = x.x; v[i * 16 + r * 4 + 1] = x.y; v[i * 16 + r * 4 + 2] = x.z; v[i * 16 + r * 4 + 3] = x.w;
        }
    }
}
// registers -> LDS transposed image [DK][ROWS] in 8-byte row-quads (V^T-style)
template <int DK, int ROWS, int NPASS>
__device__ __forceinline__ void tileT_lstore(uint2* hi, uint2* lo, int tid, const float (&v)[rowsT_n<DK, ROWS>() * 16]) {
    constexpr int DQ = DK / 4, N = rowsT_n<DK, ROWS>();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = tid + 256 * i;
        if (c >= DQ * (ROWS / 4)) break;
        const int dq = c % DQ, kg = c / DQ;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int d = dq * 4 + cc;
            uint2 h, l;
            if constexpr (NPASS == 3) {
                split_bf2(v[i * 16 + 0 + cc], v[i * 16 + 4 + cc], h.x, l.x);
                split_bf2(v[i * 16 + 8 + cc], v[i * 16 + 12 + cc], h.y, l.y);
                lo[vunit<ROWS>(d, kg)] = l;
            } else {
                h.x = pack_bf2(v[i * 16 + 0 + cc], v[i * 16 + 4 + cc]);
                h.y = pack_bf2(v[i * 16 + 8 + cc], v[i * 16 + 12 + cc]);
            }
            hi[vunit<ROWS>(d, kg)] = h;
        }
    }
}
// MFMA operand from the transposed image: rows (d) across lanes, reduction = two 4-row quads kg, kg+2
template <int ROWS>
__device__ __forceinline__ bf16x8 tfrag(const uint2* img, int d, int kg) {
    const uint2 a = img[vunit<ROWS>(d, kg)], b = img[vunit<ROWS>(d, kg + 2)];
    return as_bf16x8(make_uint4(a.x, a.y, b.x, b.y));
}
// 8 consecutive fp32 -> bf16x8 hi (+lo)
template <int NPASS>
__device__ __forceinline__ void cvt8(const float* src, bool ok, bf16x8& hi, bf16x8& lo) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
    if (ok) { x = ldg4(src); y = ldg4(src + 4); }
    uint4 h, l = make_uint4(0, 0, 0, 0);
    if constexpr (NPASS == 3) {
        split_bf2(x.x, x.y, h.x, l.x); split_bf2(x.z, x.w, h.y, l.y);
        split_bf2(y.x, y.y, h.z, l.z); split_bf2(y.z, y.w, h.w, l.w);
    } else {
        h.x = pack_bf2(x.x, x.y); h.y = pack_bf2(x.z, x.w); h.z = pack_bf2(y.x, y.y); h.w = pack_bf2(y.z, y.w);
    }
    hi = as_bf16x8(h);
    lo = as_bf16x8(l);
}
// 8 accumulator values (one reduction step of the transposed product) -> bf16x8 hi (+lo)
template <int NPASS>
__device__ __forceinline__ void pack_p(const float (&p)[16], int s2, bf16x8& hi, bf16x8& lo) {
    uint4 h, l = make_uint4(0, 0, 0, 0);
    const int o = 8 * s2;
    if constexpr (NPASS == 3) {
        split_bf2(p[o + 0], p[o + 1], h.x, l.x); split_bf2(p[o + 2], p[o + 3], h.y, l.y);
        split_bf2(p[o + 4], p[o + 5], h.z, l.z); split_bf2(p[o + 6], p[o + 7], h.w, l.w);
    } else {
        h.x = pack_bf2(p[o + 0], p[o + 1]); h.y = pack_bf2(p[o + 2], p[o + 3]);
        h.z = pack_bf2(p[o + 4], p[o + 5]); h.w = pack_bf2(p[o + 6], p[o + 7]);
    }
    hi = as_bf16x8(h);
    lo = as_bf16x8(l);
}

struct AttnP {
    const float *Q, *K, *V, *O, *dO, *lse;
    float *Ow, *lsew, *dQ, *dK, *dV, *delta;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    const uint8_t* mask;
    int64_t mask_bs, mask_qs;
    int B, H, Sq, Sk;
    float scale, drop_p;
    const uint64_t* rng;
    uint32_t site;
};

// mask byte for (b, q, key): key-padding masks (mask_qs == 0) are read through the LDS copy
__device__ __forceinline__ bool mask_ok(const AttnP& p, const uint8_t* smask, int b, int q, int key_local, int key) {
    if (key >= p.Sk) return false;
    if (p.mask == nullptr) return true;
    if (p.mask_qs == 0) return smask[key_local] != 0;
    if (q >= p.Sq) return false;
    return p.mask[(int64_t)b * p.mask_bs + (int64_t)q * p.mask_qs + key] != 0;
}

// =================================================================================== forward
template <int DK, int NPASS>
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel(const AttnP p) {
    using G = Geo<DK>;
    constexpr int BC = G::BC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sKh = reinterpret_cast<uint4*>(smem);
    uint2* sVh = reinterpret_cast<uint2*>(smem + G::K_BYTES);
    uint4* sKl = reinterpret_cast<uint4*>(smem + G::K_BYTES + G::V_BYTES);
    uint2* sVl = reinterpret_cast<uint2*>(smem + 2 * G::K_BYTES + G::V_BYTES);
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + (NPASS == 3 ? 2 : 1) * (G::K_BYTES + G::V_BYTES));

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;   // this lane's query row
    const bool qok = q < p.Sq;

    const float* Kb = p.K + (int64_t)b * p.bsk + h * DK;
    const float* Vb = p.V + (int64_t)b * p.bsv + h * DK;

    // Q^T fragments (B operand of S^T = K.Q^T): lane holds q = l31, d = 16 s + 8 half + j
    bf16x8 qh[DK / 16], ql[DK / 16];
    {
        const float* Qr = p.Q + (int64_t)b * p.bsq + (int64_t)q * p.ldq + h * DK + 8 * half;
#pragma unroll
        for (int s = 0; s < DK / 16; ++s) cvt8<NPASS>(Qr + 16 * s, qok, qh[s], ql[s]);
    }

    f32x16 o[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;

    float kv[rows_n<DK, BC>() * 8], vv[rowsT_n<DK, BC>() * 16];
    const int ntile = (p.Sk + BC - 1) / BC;
    tile_gload<DK, BC>(Kb, p.ldk, 0, p.Sk, tid, kv);
    tileT_gload<DK, BC>(Vb, p.ldv, 0, p.Sk, tid, vv);

    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * BC;
        tile_lstore_rows<DK, BC, NPASS>(sKh, sKl, tid, kv);
        tileT_lstore<DK, BC, NPASS>(sVh, sVl, tid, vv);
        if (p.mask != nullptr && p.mask_qs == 0 && tid < BC)
            sMask[tid] = (key0 + tid < p.Sk) ? p.mask[(int64_t)b * p.mask_bs + key0 + tid] : (uint8_t)0;
        __syncthreads();
        if (t + 1 < ntile) {
            tile_gload<DK, BC>(Kb, p.ldk, key0 + BC, p.Sk, tid, kv);
            tileT_gload<DK, BC>(Vb, p.ldv, key0 + BC, p.Sk, tid, vv);
        }
#pragma unroll
        for (int sub = 0; sub < G::NSUB; ++sub) {
            // ---- S^T = K . Q^T for 32 keys x 32 queries
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < DK / 16; ++s) {
                const int idx = kslot<DK>(sub * 32 + l31, 2 * s + half);
                const bf16x8 kh = as_bf16x8(sKh[idx]);
                if constexpr (NPASS == 3) {
                    const bf16x8 kl = as_bf16x8(sKl[idx]);
                    st = mfma32(kl, qh[s], st);
                    st = mfma32(kh, ql[s], st);
                }
                st = mfma32(kh, qh[s], st);
            }
            // ---- online softmax over the key axis (rows of S^T: in-lane + one lane^32 exchange)
            float pv[16];
            float tmax = NEG_INF;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl_ = sub * 32 + acc_row(r, half);
                const bool ok = mask_ok(p, sMask, b, q, kl_, key0 + kl_);
                pv[r] = ok ? st[r] * p.scale : NEG_INF;
                tmax = fmaxf(tmax, pv[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
            const float alpha = __expf(m_run - m_use);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __expf(pv[r] - m_use);
                psum += pv[r];
            }
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            // ---- O^T += V^T . P^T
            bf16x8 ph[2], pl[2];
            pack_p<NPASS>(pv, 0, ph[0], pl[0]);
            pack_p<NPASS>(pv, 1, ph[1], pl[1]);
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int d = dt * 32 + l31, kg = sub * 8 + 4 * s2 + half;
                    const bf16x8 vh = tfrag<BC>(sVh, d, kg);
                    if constexpr (NPASS == 3) {
                        const bf16x8 vl = tfrag<BC>(sVl, d, kg);
                        o[dt] = mfma32(vl, ph[s2], o[dt]);
                        o[dt] = mfma32(vh, pl[s2], o[dt]);
                    }
                    o[dt] = mfma32(vh, ph[s2], o[dt]);
                }
        }
        __syncthreads();
    }

    // ---- epilogue: normalise, dropout, store O[b][q][h*DK + d] (4 consecutive d per register quad)
    if (qok) {
        const float inv = 1.f / l_run;   // l == 0 (fully masked row) -> 0 * inf = NaN, as the reference's softmax
        const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
        const int64_t rowoff = (int64_t)b * p.bso + (int64_t)q * p.ldo + h * DK;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = dt * 32 + 8 * r4 + 4 * half;
                float4 v;
                v.x = drop_apply(dc, o[dt][4 * r4 + 0] * inv, (uint64_t)(rowoff + d + 0));
                v.y = drop_apply(dc, o[dt][4 * r4 + 1] * inv, (uint64_t)(rowoff + d + 1));
                v.z = drop_apply(dc, o[dt][4 * r4 + 2] * inv, (uint64_t)(rowoff + d + 2));
                v.w = drop_apply(dc, o[dt][4 * r4 + 3] * inv, (uint64_t)(rowoff + d + 3));
                *reinterpret_cast<float4*>(p.Ow + rowoff + d) = v;
            }
        if (half == 0) p.lsew[((int64_t)b * p.H + h) * p.Sq + q] = m_run + __logf(l_run);
    }
}

// =================================================================================== backward
// delta[b,h,q] = (1-p) * sum_d dO[b,q,h*DK+d] * O[b,q,h*DK+d]   (O is the saved post-dropout output)
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnP p, int DK) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;   // over B*H*Sq
    const int64_t total = (int64_t)p.B * p.H * p.Sq;
    if (row >= total) return;
    const int q = (int)(row % p.Sq);
    const int bh = (int)(row / p.Sq);
    const int b = bh / p.H, h = bh % p.H;
    const int64_t off = (int64_t)b * p.bso + (int64_t)q * p.ldo + h * DK;
    float s = 0.f;
    for (int d = lane * 4; d < DK; d += 256) {
        const float4 a = ldg4(p.dO + off + d), c = ldg4(p.O + off + d);
        s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
    s = wave_sum(s);
    if (lane == 0) p.delta[row] = s * (1.f - p.drop_p);
}

// dQ kernel: workgroup = 128 query rows (4 waves x 32), loops over key tiles of 32.
//   S^T = K.Q^T, dP^T = V.dO^T (A from LDS row-major tiles, B = Q / dO fragments in VGPRs),
//   dS^T = P^T * (dP^T - delta) in registers, dQ^T[d][q] += K^T[d][key] . dS^T[key][q]  (A = K^T image).
template <int DK>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(const AttnP p) {
    constexpr int BC = 32, DT = DK / 32;
    constexpr int TB = BC * DK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sK = reinterpret_cast<uint4*>(smem);
    uint4* sV = reinterpret_cast<uint4*>(smem + TB);
    uint2* sKt = reinterpret_cast<uint2*>(smem + 2 * TB);
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + 3 * TB);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;

    const float* Kb = p.K + (int64_t)b * p.bsk + h * DK;
    const float* Vb = p.V + (int64_t)b * p.bsv + h * DK;

    bf16x8 qf[DK / 16], dof[DK / 16], unused;
    {
        const float* Qr = p.Q + (int64_t)b * p.bsq + (int64_t)q * p.ldq + h * DK + 8 * half;
        const float* Dr = p.dO + (int64_t)b * p.bso + (int64_t)q * p.ldo + h * DK + 8 * half;
#pragma unroll
        for (int s = 0; s < DK / 16; ++s) {
            cvt8<1>(Qr + 16 * s, qok, qf[s], unused);
            cvt8<1>(Dr + 16 * s, qok, dof[s], unused);
        }
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.Sq + q;
    const float lse = qok ? p.lse[stat] : 0.f;
    const float delta = qok ? p.delta[stat] : 0.f;

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    // K is staged twice (row-major for S^T, transposed for dQ^T): the second read of the tile is an L2 hit
    float kv[rows_n<DK, BC>() * 8], vv[rows_n<DK, BC>() * 8], ktv[rowsT_n<DK, BC>() * 16];
    const int ntile = (p.Sk + BC - 1) / BC;
    tile_gload<DK, BC>(Kb, p.ldk, 0, p.Sk, tid, kv);
    tile_gload<DK, BC>(Vb, p.ldv, 0, p.Sk, tid, vv);
    tileT_gload<DK, BC>(Kb, p.ldk, 0, p.Sk, tid, ktv);
    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * BC;
        tile_lstore_rows<DK, BC, 1>(sK, nullptr, tid, kv);
        tile_lstore_rows<DK, BC, 1>(sV, nullptr, tid, vv);
        tileT_lstore<DK, BC, 1>(sKt, nullptr, tid, ktv);
        if (p.mask != nullptr && p.mask_qs == 0 && tid < BC)
            sMask[tid] = (key0 + tid < p.Sk) ? p.mask[(int64_t)b * p.mask_bs + key0 + tid] : (uint8_t)0;
        __syncthreads();
        if (t + 1 < ntile) {
            tile_gload<DK, BC>(Kb, p.ldk, key0 + BC, p.Sk, tid, kv);
            tile_gload<DK, BC>(Vb, p.ldv, key0 + BC, p.Sk, tid, vv);
            tileT_gload<DK, BC>(Kb, p.ldk, key0 + BC, p.Sk, tid, ktv);
        }
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < DK / 16; ++s) {
            const int idx = kslot<DK>(l31, 2 * s + half);
            st = mfma32(as_bf16x8(sK[idx]), qf[s], st);
            dp = mfma32(as_bf16x8(sV[idx]), dof[s], dp);
        }
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl_ = acc_row(r, half);
            const bool ok = qok && mask_ok(p, sMask, b, q, kl_, key0 + kl_);
            const float pr = ok ? __expf(st[r] * p.scale - lse) : 0.f;
            ds[r] = pr * (dp[r] - delta) * p.scale;
        }
        bf16x8 dsf[2];
        pack_p<1>(ds, 0, dsf[0], unused);
        pack_p<1>(ds, 1, dsf[1], unused);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                dq[dt] = mfma32(tfrag<BC>(sKt, dt * 32 + l31, 4 * s2 + half), dsf[s2], dq[dt]);
        __syncthreads();
    }
    if (qok) {
        const int64_t rowoff = (int64_t)b * p.bsq + (int64_t)q * p.ldq + h * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = dt * 32 + 8 * r4 + 4 * half;
                *reinterpret_cast<float4*>(p.dQ + rowoff + d) =
                    make_float4(dq[dt][4 * r4 + 0], dq[dt][4 * r4 + 1], dq[dt][4 * r4 + 2], dq[dt][4 * r4 + 3]);
            }
    }
}

// dK/dV kernel: workgroup = 64 keys, loops over query tiles of 32.  The 4 waves split by ROLE so that each
// holds one 32x DK accumulator (128 VGPRs at DK = 256): waves 0,1 produce dV for keys [0,32) / [32,64),
// waves 2,3 produce dK for the same keys.
//   S[q][key] = Q.K^T  (A = Q tile from LDS, B = this wave's K rows from the LDS K image); accumulator lane = key,
//   rows = q, so P and dS feed the second products as B operands:
//     dV^T[d][key] += dO^T[d][q] . P[q][key]            (A = transposed dO image)
//     dK^T[d][key] += Q^T[d][q]  . dS[q][key]           (A = transposed Q image; dS needs dP = dO.V^T as well)
template <int DK>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const AttnP p) {
    constexpr int BQ = 32, DT = DK / 32, KB = 64;
    constexpr int TB = BQ * DK * 2;            // one 32-row tile image
    constexpr int KVB = KB * DK * 2;           // the workgroup's 64 keys of K (or V), row-major image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sK = reinterpret_cast<uint4*>(smem);
    uint4* sV = reinterpret_cast<uint4*>(smem + KVB);
    uint4* sQ = reinterpret_cast<uint4*>(smem + 2 * KVB);
    uint4* sdO = reinterpret_cast<uint4*>(smem + 2 * KVB + TB);
    uint2* sQt = reinterpret_cast<uint2*>(smem + 2 * KVB + 2 * TB);
    uint2* sdOt = reinterpret_cast<uint2*>(smem + 2 * KVB + 3 * TB);
    float* sLse = reinterpret_cast<float*>(smem + 2 * KVB + 4 * TB);
    float* sDelta = sLse + BQ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int role = __builtin_amdgcn_readfirstlane(wid >> 1);   // 0: dV, 1: dK
    const int kgrp = __builtin_amdgcn_readfirstlane(wid & 1);
    const int nkt = (p.Sk + KB - 1) / KB;
    const int w = xcd_remap(blockIdx.x, nkt * p.B * p.H);
    const int kt = w % nkt, bh = w / nkt;
    const int b = bh / p.H, h = bh % p.H;
    const int key = kt * KB + kgrp * 32 + l31;   // this lane's key
    const bool kok = key < p.Sk;

    const float* Qb = p.Q + (int64_t)b * p.bsq + h * DK;
    const float* dOb = p.dO + (int64_t)b * p.bso + h * DK;

    // the workgroup's K and V rows -> LDS once (row-major swizzled image, 64 rows)
    {
        const float* Kb = p.K + (int64_t)b * p.bsk + h * DK;
        const float* Vb = p.V + (int64_t)b * p.bsv + h * DK;
        float tmp[rows_n<DK, KB>() * 8];
        tile_gload<DK, KB>(Kb, p.ldk, kt * KB, p.Sk, tid, tmp);
        tile_lstore_rows<DK, KB, 1>(sK, nullptr, tid, tmp);
        tile_gload<DK, KB>(Vb, p.ldv, kt * KB, p.Sk, tid, tmp);
        tile_lstore_rows<DK, KB, 1>(sV, nullptr, tid, tmp);
    }
    const int myrow = kgrp * 32 + l31;
    // key-padding mask for this lane's key is loop invariant
    bool kmask = kok;
    if (kok && p.mask != nullptr && p.mask_qs == 0) kmask = p.mask[(int64_t)b * p.mask_bs + key] != 0;

    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    const int ntile = (p.Sq + BQ - 1) / BQ;
    for (int t = 0; t < ntile; ++t) {
        const int q0 = t * BQ;
        __syncthreads();   // previous tile fully consumed
        {
            float tmp[rows_n<DK, BQ>() * 8];
            tile_gload<DK, BQ>(Qb, p.ldq, q0, p.Sq, tid, tmp);
            tile_lstore_rows<DK, BQ, 1>(sQ, nullptr, tid, tmp);
            tile_gload<DK, BQ>(dOb, p.ldo, q0, p.Sq, tid, tmp);
            tile_lstore_rows<DK, BQ, 1>(sdO, nullptr, tid, tmp);
            float tmpT[rowsT_n<DK, BQ>() * 16];
            tileT_gload<DK, BQ>(Qb, p.ldq, q0, p.Sq, tid, tmpT);
            tileT_lstore<DK, BQ, 1>(sQt, nullptr, tid, tmpT);
            tileT_gload<DK, BQ>(dOb, p.ldo, q0, p.Sq, tid, tmpT);
            tileT_lstore<DK, BQ, 1>(sdOt, nullptr, tid, tmpT);
        }
        if (tid < BQ) {
            const int qq = q0 + tid;
            const int64_t stat = ((int64_t)b * p.H + h) * p.Sq + qq;
            sLse[tid] = (qq < p.Sq) ? p.lse[stat] : 0.f;
            sDelta[tid] = (qq < p.Sq) ? p.delta[stat] : 0.f;
        }
        __syncthreads();
        // S[q][key] (both roles) and dP[q][key] (dK role): A rows = q (from LDS), B cols = key (this wave's K / V rows)
        f32x16 sacc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < DK / 16; ++s) {
            const int ia = kslot<DK>(l31, 2 * s + half), ib = kslot<DK>(myrow, 2 * s + half);
            sacc = mfma32(as_bf16x8(sQ[ia]), as_bf16x8(sK[ib]), sacc);
            if (role == 1) dp = mfma32(as_bf16x8(sdO[ia]), as_bf16x8(sV[ib]), dp);
        }
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql_ = acc_row(r, half);
            const int qq = q0 + ql_;
            bool ok = kmask && qq < p.Sq;
            if (ok && p.mask != nullptr && p.mask_qs != 0)
                ok = p.mask[(int64_t)b * p.mask_bs + (int64_t)qq * p.mask_qs + key] != 0;
            pr[r] = ok ? __expf(sacc[r] * p.scale - sLse[ql_]) : 0.f;
            if (role == 1) pr[r] = pr[r] * (dp[r] - sDelta[ql_]) * p.scale;   // dS
        }
        bf16x8 bf[2], unused;
        pack_p<1>(pr, 0, bf[0], unused);
        pack_p<1>(pr, 1, bf[1], unused);
        const uint2* timg = (role == 1) ? sQt : sdOt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                acc[dt] = mfma32(tfrag<BQ>(timg, dt * 32 + l31, 4 * s2 + half), bf[s2], acc[dt]);
    }
    if (kok) {
        float* dst = (role == 1) ? p.dK + (int64_t)b * p.bsk + (int64_t)key * p.ldk + h * DK
                                 : p.dV + (int64_t)b * p.bsv + (int64_t)key * p.ldv + h * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = dt * 32 + 8 * r4 + 4 * half;
                *reinterpret_cast<float4*>(dst + d) =
                    make_float4(acc[dt][4 * r4 + 0], acc[dt][4 * r4 + 1], acc[dt][4 * r4 + 2], acc[dt][4 * r4 + 3]);
            }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int DK, int NPASS>
int launch_fwd(const AttnP& p, hipStream_t st) {
    using G = Geo<DK>;
    const int lds = (NPASS == 3 ? 2 : 1) * (G::K_BYTES + G::V_BYTES) + 64;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<DK, NPASS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    hipLaunchKernelGGL((attn_fwd_kernel<DK, NPASS>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_attn_fwd");
    return BMT_OK;
}

template <int DK>
int launch_bwd(const AttnP& p, hipStream_t st) {
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, DK);
    BMT_CHECK_LAUNCH("bmt_attn_bwd(delta)");
    {
        const int lds = 3 * 32 * DK * 2 + 64;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done = true;
        }
        const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DK>), dim3(nblk), dim3(256), lds, st, p);
        BMT_CHECK_LAUNCH("bmt_attn_bwd(dq)");
    }
    {
        const int lds = 2 * 64 * DK * 2 + 4 * 32 * DK * 2 + 2 * 32 * 4;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done = true;
        }
        const int nblk = ((p.Sk + 63) / 64) * p.B * p.H;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<DK>), dim3(nblk), dim3(256), lds, st, p);
        BMT_CHECK_LAUNCH("bmt_attn_bwd(dkv)");
    }
    return BMT_OK;
}

int check_common(const char* who, const void* Q, const void* K, const void* V, const void* O, int64_t ldq, int64_t ldk,
                 int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, int B, int H, int Sq, int Sk,
                 int dk) {
    BMT_CHECK_ARG(Q && K && V && O, "%s: null pointer", who);
    BMT_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0, "%s: bad sizes B=%d H=%d Sq=%d Sk=%d", who, B, H, Sq, Sk);
    BMT_CHECK_ARG(dk == 32 || dk == 64 || dk == 128 || dk == 256, "%s: d_k=%d not in {32,64,128,256}", who, dk);
    if (!(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(O)) ||
        ((ldq | ldk | ldv | ldo | bsq | bsk | bsv | bso) & 3)) {
        bmt_set_error("%s: pointers must be 16-byte aligned and strides multiples of 4 floats", who);
        return BMT_EALIGN;
    }
    return BMT_OK;
}

}  // namespace

extern "C" int bmt_attn_fwd(const bmt_attn_fwd_args* a, void* stream) {
    BMT_CHECK_ARG(a, "bmt_attn_fwd: null args");
    int rc = check_common("bmt_attn_fwd", a->Q, a->K, a->V, a->O, a->ldq, a->ldk, a->ldv, a->ldo, a->bsq, a->bsk, a->bsv,
                          a->bso, a->B, a->H, a->Sq, a->Sk, a->dk);
    if (rc) return rc;
    BMT_CHECK_ARG(a->lse, "bmt_attn_fwd: lse is required");
    BMT_CHECK_ARG(a->precision == BMT_PREC_BF16 || a->precision == BMT_PREC_BF16X3, "bmt_attn_fwd: bad precision");
    AttnP p;
    memset(&p, 0, sizeof(p));
    p.Q = a->Q; p.K = a->K; p.V = a->V; p.Ow = a->O; p.lsew = a->lse;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    hipStream_t st = (hipStream_t)stream;
#define BMT_FWD(D)                                                                      \
    if (a->dk == D) return a->precision == BMT_PREC_BF16X3 ? launch_fwd<D, 3>(p, st) : launch_fwd<D, 1>(p, st);
    BMT_FWD(32) BMT_FWD(64) BMT_FWD(128) BMT_FWD(256)
#undef BMT_FWD
    return BMT_EINVAL;
}

extern "C" int bmt_attn_bwd(const bmt_attn_bwd_args* a, void* stream) {
    BMT_CHECK_ARG(a, "bmt_attn_bwd: null args");
    int rc = check_common("bmt_attn_bwd", a->Q, a->K, a->V, a->O, a->ldq, a->ldk, a->ldv, a->ldo, a->bsq, a->bsk, a->bsv,
                          a->bso, a->B, a->H, a->Sq, a->Sk, a->dk);
    if (rc) return rc;
    BMT_CHECK_ARG(a->dO && a->lse && a->dQ && a->dK && a->dV && a->delta_ws, "bmt_attn_bwd: null pointer");
    if (!(aligned16(a->dO) && aligned16(a->dQ) && aligned16(a->dK) && aligned16(a->dV))) {
        bmt_set_error("bmt_attn_bwd: gradient pointers must be 16-byte aligned");
        return BMT_EALIGN;
    }
    AttnP p;
    memset(&p, 0, sizeof(p));
    p.Q = a->Q; p.K = a->K; p.V = a->V; p.O = a->O; p.dO = a->dO; p.lse = a->lse;
    p.dQ = a->dQ; p.dK = a->dK; p.dV = a->dV; p.delta = a->delta_ws;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p;
    hipStream_t st = (hipStream_t)stream;
    if (a->dk == 32) return launch_bwd<32>(p, st);
    if (a->dk == 64) return launch_bwd<64>(p, st);
    if (a->dk == 128) return launch_bwd<128>(p, st);
    if (a->dk == 256) return launch_bwd<256>(p, st);
    return BMT_EINVAL;
}
