// bmt_attn_fwd_bf16 / bmt_attn_bwd_bf16 -- flash-style masked attention over PRE-SPLIT bf16 operand planes.
//
// Same mathematics and the same transposed MFMA formulation as attention.hip (S^T = K.Q^T, O^T += V^T.P^T, softmax
// axis lane-local), but Q/K/V arrive as bf16 planes (hi [+ lo residual]) written by the projection GEMM's epilogue:
//   * the K/V staging loop moves 16-byte bf16 slots global -> registers -> LDS with NO conversion (the fp32 version
//     spent 13 VALU instructions per MFMA, PMC: profiles/r01_b_attn_fwd_pmc.csv) and half the bytes;
//   * key-padding masks are classified per staged tile (all valid / partly valid / all masked): fully valid tiles take
//     an unmasked fast path, fully masked tiles are skipped (exact: they contribute exp(-inf) = 0);
//   * the O^T accumulator is rescaled only when some lane's running maximum actually moved (exact, alpha == 1 otherwise);
//   * the backward dK/dV kernel prefetches the next query tile into registers under the current tile's MFMAs.
#include "common.h"

namespace {

constexpr float NEG_INF = -__builtin_huge_valf();

template <int DK>
struct Geo {
    static constexpr int BC = (DK == 256) ? 32 : 64;
    static constexpr int NSUB = BC / 32;
    static constexpr int DT = DK / 32;
    static constexpr int K_BYTES = BC * DK * 2;
    static constexpr int V_BYTES = DK * BC * 2;
};

template <int DK>
__device__ __forceinline__ int kslot(int row, int slot) {
    if constexpr (DK == 32) return row * 4 + (slot ^ ((row >> 2) & 3));
    else if constexpr (DK == 64) return row * 8 + (slot ^ ((row >> 1) & 7));
    else return row * (DK / 8) + (slot ^ (row & 15));
}
template <int ROWS>
__device__ __forceinline__ int vunit(int d, int kg) {
    if constexpr (ROWS == 64) return d * 16 + (kg ^ ((d >> 1) & 15));
    else return d * 8 + (kg ^ ((d >> 2) & 7));
}
template <int DK, int ROWS, int NT = 256> constexpr int rows_n() { return (ROWS * DK / 8 + NT - 1) / NT; }
template <int DK, int ROWS, int NT = 256> constexpr int rowsT_n() { return (ROWS * DK / 16 + NT - 1) / NT; }

// ---- row-major bf16 tile [ROWS][DK]: global -> registers (16-B slots) -> swizzled LDS image
template <int DK, int ROWS, int NT = 256>
__device__ __forceinline__ void tile_gload(const uint16_t* base, int64_t ld, int row0, int nrows, int tid,
                                           u32x4 (&v)[rows_n<DK, ROWS, NT>()]) {
    // UNCONDITIONAL loads (row clamped to the last valid row): a guarded load into a register array makes hipcc either wait
    // vmcnt(0) at the join or demote the array to scratch.  Clamped rows hold finite data and are neutralised downstream
    // (masked scores / zero probabilities / rows that are never stored).
    constexpr int SPR = DK / 8;
#pragma unroll
    for (int i = 0; i < rows_n<DK, ROWS, NT>(); ++i) {
        const int c = (ROWS * SPR % NT == 0) ? tid + NT * i : min(tid + NT * i, ROWS * SPR - 1);
        const int row = min(row0 + c / SPR, nrows - 1), slot = c % SPR;
        v[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + slot * 8);
    }
}
template <int DK, int ROWS, int NT = 256>
__device__ __forceinline__ void tile_lstore(u32x4* img, int tid, const u32x4 (&v)[rows_n<DK, ROWS, NT>()]) {
    constexpr int SPR = DK / 8;
#pragma unroll
    for (int i = 0; i < rows_n<DK, ROWS, NT>(); ++i) {
        const int c = tid + NT * i;
        if constexpr (ROWS * SPR % NT != 0) {
            if (c < ROWS * SPR) img[kslot<DK>(c / SPR, c % SPR)] = v[i];
        } else {
            img[kslot<DK>(c / SPR, c % SPR)] = v[i];
        }
    }
}
// ---- transposed image [DK][ROWS] in 8-byte row-quads: each thread owns 4(row) x 4(d) blocks
template <int DK, int ROWS, int NT = 256>
__device__ __forceinline__ void tileT_gload(const uint16_t* base, int64_t ld, int row0, int nrows, int tid,
                                            u32x2 (&v)[rowsT_n<DK, ROWS, NT>() * 4]) {
    constexpr int DQ = DK / 4, NB = DQ * (ROWS / 4);
#pragma unroll
    for (int i = 0; i < rowsT_n<DK, ROWS, NT>(); ++i) {
        const int c = (NB % NT == 0) ? tid + NT * i : min(tid + NT * i, NB - 1);
        const int dq = c % DQ, kg = c / DQ;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + kg * 4 + r, nrows - 1);
            v[i * 4 + r] = *reinterpret_cast<const u32x2*>(base + (int64_t)row * ld + dq * 4);
        }
    }
}
template <int DK, int ROWS, int NT = 256>
__device__ __forceinline__ void tileT_lstore(u32x2* img, int tid, const u32x2 (&v)[rowsT_n<DK, ROWS, NT>() * 4]) {
    constexpr int DQ = DK / 4;
#pragma unroll
    for (int i = 0; i < rowsT_n<DK, ROWS, NT>(); ++i) {
        const int c = tid + NT * i;
        if constexpr (DQ * (ROWS / 4) % NT != 0) {
            if (c >= DQ * (ROWS / 4)) continue;
        }
        const int dq = c % DQ, kg = c / DQ;
        const u32x2 a = v[i * 4 + 0], b = v[i * 4 + 1], cc = v[i * 4 + 2], d = v[i * 4 + 3];
        // 4x4 bf16 transpose: output row = d index, 4 consecutive source rows packed low -> high
        const u32x2 o0 = {(a.x & 0xffffu) | (b.x << 16), (cc.x & 0xffffu) | (d.x << 16)};
        const u32x2 o1 = {(a.x >> 16) | (b.x & 0xffff0000u), (cc.x >> 16) | (d.x & 0xffff0000u)};
        const u32x2 o2 = {(a.y & 0xffffu) | (b.y << 16), (cc.y & 0xffffu) | (d.y << 16)};
        const u32x2 o3 = {(a.y >> 16) | (b.y & 0xffff0000u), (cc.y >> 16) | (d.y & 0xffff0000u)};
        img[vunit<ROWS>(dq * 4 + 0, kg)] = o0;
        img[vunit<ROWS>(dq * 4 + 1, kg)] = o1;
        img[vunit<ROWS>(dq * 4 + 2, kg)] = o2;
        img[vunit<ROWS>(dq * 4 + 3, kg)] = o3;
    }
}
template <int ROWS>
__device__ __forceinline__ bf16x8 tfrag(const u32x2* img, int d, int kg) {
    const u32x2 a = img[vunit<ROWS>(d, kg)], b = img[vunit<ROWS>(d, kg + 2)];
    const u32x4 r = {a.x, a.y, b.x, b.y};
    return as_bf16x8(r);
}
template <int NPASS, bool F16 = false>
__device__ __forceinline__ void pack_p(const float (&p)[16], int s2, bf16x8& hi, bf16x8& lo) {
    uint4 h, l = make_uint4(0, 0, 0, 0);
    const int o = 8 * s2;
    if constexpr (F16) {
        h.x = pack_h2(p[o + 0], p[o + 1]); h.y = pack_h2(p[o + 2], p[o + 3]);
        h.z = pack_h2(p[o + 4], p[o + 5]); h.w = pack_h2(p[o + 6], p[o + 7]);
    } else if constexpr (NPASS == 3) {
        split_bf2(p[o + 0], p[o + 1], h.x, l.x); split_bf2(p[o + 2], p[o + 3], h.y, l.y);
        split_bf2(p[o + 4], p[o + 5], h.z, l.z); split_bf2(p[o + 6], p[o + 7], h.w, l.w);
    } else {
        h.x = pack_bf2(p[o + 0], p[o + 1]); h.y = pack_bf2(p[o + 2], p[o + 3]);
        h.z = pack_bf2(p[o + 4], p[o + 5]); h.w = pack_bf2(p[o + 6], p[o + 7]);
    }
    hi = as_bf16x8(h);
    lo = as_bf16x8(l);
}
// UNCONDITIONAL load, then a select: `p` must be readable even when !ok (callers clamp the row: min(q, Sq - 1)).  The guarded form
// (`if (ok) v = *p`) made hipcc 7.2 branch around every load of a fragment array and wait vmcnt(0) behind it: the 16 - 48 row loads of a
// kernel prologue became as many serialized memory round trips (tools/exp_isa.sh listing, round 3).
__device__ __forceinline__ bf16x8 ldfrag(const uint16_t* p, bool ok) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    return as_bf16x8(v);
}

// one gradient output (dQ, dK or dV) in up to four forms, all optional
struct GradOut {
    float* f32; int64_t f_ld, f_bs;        // fp32 [B,S,D]
    uint16_t* hi; int64_t h_ld, h_bs;      // bf16 plane, row b*S+s at hi + b*h_bs + s*h_ld   (dX operand of the projection)
    uint16_t* hiT; int64_t t_ld;           // transposed bf16 plane [D][t_ld], column b*S+s      (dW operand)
    float* bsum;                           // fp32 [D] += sum over (b,s)                          (bias gradient)
    float* bpart;                          // split backward: the tile's column sums go to bpart[(b * tiles + tile) * bp_ld + h * d_k + d] instead
    int64_t bp_ld;                         // (plain stores; attn_bias_finish_kernel adds the rows up into bsum: no contended atomics)
};

struct AttnPB {
    const uint16_t *Qh, *Ql, *Kh, *Kl, *Vh, *Vl, *dOh;
    const float *O, *dO, *lse;
    const uint16_t *Oph, *Opl;             // saved forward output as planes (backward, when O == nullptr)
    float *Ow, *lsew, *delta;
    int fuse_delta;                        // backward, 16-wide kernels: the dQ kernel computes delta = rowsum(dO * O) itself (and stores it for dK/dV)
    uint16_t *Owh, *Owl;                   // forward output planes (optional), strides ldop / bsop: bf16(o) and bf16(o - hi), or fp16(o) when ow_f16
    int ow_f16;
    const uint16_t* Opf;                   // backward: the saved output's fp16 plane (delta = rowsum(dO * O) reads it instead of Oph + Opl)
    int64_t ldop, bsop;
    GradOut gq, gk, gv;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;     // plane strides for Q/K/V (and dOh: ldo/bso); fp32 O/dO share ldo/bso
    const uint8_t* mask;
    int64_t mask_bs, mask_qs;
    int B, H, Sq, Sk;
    float scale, drop_p;
    const uint64_t* rng;
    uint32_t site;
    const float* kmean;                    // backward, optional: fp32 [B][H * d_k] mean key over the valid keys (bmt_attn_kmean)
    int qkv_f16;                           // backward, 16-wide kernels: Qh / Kh / Vh hold fp16 (the forward's planes); converted to bf16 on load
    // backward, split form (attn_bwd_dq32e_kernel -> attn_bwd_dkvg_kernel): the dQ kernel leaves P and dS (bf16, [B*H][Sq][ws_pitch]) and a
    // bf16 copy of q ([B][Sq][H * d_k]: ldqb / bsqb) in workspaces, dK / dV are two plain products over them
    uint16_t *Pws, *dSws, *Qbws;
    int64_t ws_pitch, ws_tile, ws_slab, ldqb, bsqb;     // (batch, head) slab bh at bh * ws_slab; in it element (q, key) of a (batch, head) slab at (key / 128) * ws_tile + q * ws_pitch + key % 128
    int xp;                                    // experiments only: parts of a loop switched off (timing probes)
    int defer_bias;                            // the per-tile bias sums stay in bpart: the caller adds them up (bmt_colsum_multi)
    // split backward: which 32-query groups have a non-zero dO at all.  qlive[(b * H + h) * ceil(Sq / 128) + tile] = bit w set iff rows
    // tile * 128 + 32 w .. + 31 of head h carry a non-zero gradient; written by the dQ kernel (which has those rows in registers anyway),
    // read by the dK / dV kernel.  Padded positions of a sequence get EXACTLY zero gradient in the encoder (their keys are masked everywhere
    // downstream), a quarter of the query rows under configs[1]'s ragged lengths: a dQ tile without a live row skips its key loop and
    // emits nothing, the dK / dV loop ends at the last live 32-query stage.  Data-driven, so a caller whose padded rows DO carry gradient
    // loses nothing but the shortcut.  nullptr: off.
    int* qlive;
    float* qamax;                              // recompute form: qamax[(b * H + h) * ceil(Sq / 128) + tile] = max |dO| over the tile's rows of head h (written by the dQ kernel)
    // PACKED ROWS (ABI 7; bmt_attn_fwd_bf16_args.q_off / k_off): the valid positions of a ragged batch are stored compacted -- sample b's query
    // rows are rows q_off[b] .. q_off[b + 1] - 1 of every query-side plane (its key rows k_off[b] ..), nothing is masked.  SqP / SkP keep the
    // PADDED extents the launch was sized for: they map workgroups to (batch, head, tile) and index what stays per padded position
    // (lse, delta, the split backward's workspaces, the per-tile bias partials); attn_rebase turns Sq / Sk into this sample's lengths.
    const int *q_off, *k_off;
    const int* b_order;                        // (ABI 10) work items are numbered with sample b_order[i] in place of sample i (attn_sample)
    int kvh;                                   // (ABI 10) elements between two heads' columns in the K / V planes: d_k, or 0 = ONE key / value plane of width
                                               // d_k shared by the heads (attention against an un-projected input: ops.MHAFn's rank path)
    int SqP, SkP;
    int64_t drop_off;                          // element index of this sample's first output row in the dropout mask's index space (attn_rebase)
};

// the sample a work item's sample-major index stands for: the launch's own numbering, or the caller's balanced order (bmt_pack_rows_ordered)
__device__ __forceinline__ int attn_sample(const AttnPB& p, int i) { return p.b_order != nullptr ? p.b_order[i] : i; }

// packed rows: make `p` describe sample b alone -- base pointers at its first row, batch strides 0, Sq / Sk = its lengths.  A no-op for
// padded callers (q_off == k_off == nullptr: SqP == Sq, SkP == Sk as the host set them).
__device__ __forceinline__ void attn_rebase(AttnPB& p, int b) {
    if (p.q_off != nullptr) {
        const int r0 = p.q_off[b];
        p.Sq = p.q_off[b + 1] - r0;
        const int64_t oq = (int64_t)r0 * p.ldq, oo = (int64_t)r0 * p.ldo, op = (int64_t)r0 * p.ldop;
        p.Qh += oq;
        if (p.Ql) p.Ql += oq;
        if (p.dOh) p.dOh += oo;
        if (p.O) p.O += oo;
        if (p.dO) p.dO += oo;
        if (p.Ow) p.Ow += oo;
        if (p.Oph) p.Oph += op;
        if (p.Opl) p.Opl += op;
        if (p.Opf) p.Opf += op;
        if (p.Owh) p.Owh += op;
        if (p.Owl) p.Owl += op;
        if (p.gq.f32) p.gq.f32 += (int64_t)r0 * p.gq.f_ld;
        if (p.gq.hi) p.gq.hi += (int64_t)r0 * p.gq.h_ld;
        p.bsq = 0; p.bso = 0; p.bsop = 0; p.gq.f_bs = 0; p.gq.h_bs = 0;
        p.drop_off = oo;                     // the output dropout's mask is indexed by the element of the whole (packed) tensor
    }
    if (p.k_off != nullptr) {
        const int r0 = p.k_off[b];
        p.Sk = p.k_off[b + 1] - r0;
        const int64_t ok = (int64_t)r0 * p.ldk, ov = (int64_t)r0 * p.ldv;
        p.Kh += ok;
        if (p.Kl) p.Kl += ok;
        p.Vh += ov;
        if (p.Vl) p.Vl += ov;
        if (p.gk.f32) p.gk.f32 += (int64_t)r0 * p.gk.f_ld;
        if (p.gk.hi) p.gk.hi += (int64_t)r0 * p.gk.h_ld;
        if (p.gv.f32) p.gv.f32 += (int64_t)r0 * p.gv.f_ld;
        if (p.gv.hi) p.gv.hi += (int64_t)r0 * p.gv.h_ld;
        p.bsk = 0; p.bsv = 0; p.gk.f_bs = 0; p.gk.h_bs = 0; p.gv.f_bs = 0; p.gv.h_bs = 0;
    }
}
// byte extent of `rows` rows of a (batch, head)'s plane slice (row stride ld elements, DK of them used): what a buffer descriptor over it
// may read; no rows -> nothing
__device__ __forceinline__ int plane_extent(int rows, int64_t ld, int DK) { return rows > 0 ? (int)(((int64_t)(rows - 1) * ld + DK) * 2) : 0; }


// 8 fp16 -> 8 bf16 (round to nearest even) in one 16-byte register slot: q / k / v exist as fp16 planes only under the fp16 attention
// policy (4 instead of 6 bytes per element written by the projections); the backward's bf16 products convert them while staging
__device__ __forceinline__ uint32_t h2_to_b2(uint32_t w) {
    const f32x2_t f = __builtin_convertvector(__builtin_bit_cast(f16x2_t, w), f32x2_t);
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
// (element by element from scalars: a loop that assigns r[q] of an uninitialised ext_vector came out of hipcc 7.2 with r[1..3] all
// equal to r[0] -- the check in tools/probes/qkv_f16_bwd_check.py compares the mean-key kernel on the two plane kinds)
__device__ __forceinline__ u32x4 h8_to_b8(u32x4 v) {
    const uint32_t a = h2_to_b2(v[0]), b = h2_to_b2(v[1]), c = h2_to_b2(v[2]), d = h2_to_b2(v[3]);
    return u32x4{a, b, c, d};
}
__device__ __forceinline__ bf16x8 h8_to_b8(bf16x8 v) { return as_bf16x8(h8_to_b8(__builtin_bit_cast(u32x4, v))); }

// dQ_i = sum_j dS_ij K_j with sum_j dS_ij = 0 exactly: the bf16 rounding of dS leaves a residue (sum_j round(dS_ij)) that the
// product multiplies by the keys' common component -- 10-25 % of |dQ| where attention is near uniform over many similar keys
// (the decoder's cross-attention, the encoder's second layer; tests/study_attn_bwd_centering.py).  The dQ kernels add up the
// ROUNDED dS they feed to the MFMA (rs) and take rs * mean key out again in fp32: error 25 % -> 0.5 %.

// stage the key-padding mask bytes of one tile and classify it: 0 = fully masked, 1 = partial, 2 = fully valid.
// Called by every thread; result valid after the next __syncthreads().
template <int BC>
__device__ __forceinline__ void stage_mask(const AttnPB& p, int b, int key0, int tid, uint8_t* sMask, int* sFlag) {
    if (tid < 64) {
        uint8_t m = 0;
        if (tid < BC && key0 + tid < p.Sk)
            m = (p.mask != nullptr && p.mask_qs == 0) ? p.mask[(int64_t)b * p.mask_bs + key0 + tid] : (uint8_t)1;
        if (tid < BC) sMask[tid] = m;
        const unsigned long long valid = __ballot(m != 0);
        const unsigned long long full = (BC == 64) ? ~0ull : ((1ull << BC) - 1ull);
        if (tid == 0) {
            int f = (valid == 0) ? 0 : ((valid == full) ? 2 : 1);
            if (p.mask != nullptr && p.mask_qs != 0 && f == 2) f = 1;   // general (B,Sq,Sk) mask: checked per element
            sFlag[0] = f;
        }
    }
}

// ---- gradient tile epilogue.  acc[dt][r] = G^T[d = dt*32 + acc_row(r, half)][token = this lane's l31 column].
template <int DK>
__device__ __forceinline__ void grad_store_rows(const GradOut& g, const f32x16 (&acc)[DK / 32], int b, int h, int tok, bool ok, int half) {
    if (!ok) return;
    if (g.f32) {
        float* dst = g.f32 + (int64_t)b * g.f_bs + (int64_t)tok * g.f_ld + h * DK;
#pragma unroll
        for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4*>(dst + dt * 32 + 8 * r4 + 4 * half) =
                    make_float4(acc[dt][4 * r4 + 0], acc[dt][4 * r4 + 1], acc[dt][4 * r4 + 2], acc[dt][4 * r4 + 3]);
    }
    if (g.hi) {
        uint16_t* dst = g.hi + (int64_t)b * g.h_bs + (int64_t)tok * g.h_ld + h * DK;
#pragma unroll
        for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                u32x2 v;
                v[0] = pack_bf2(acc[dt][4 * r4 + 0], acc[dt][4 * r4 + 1]);
                v[1] = pack_bf2(acc[dt][4 * r4 + 2], acc[dt][4 * r4 + 3]);
                *reinterpret_cast<u32x2*>(dst + dt * 32 + 8 * r4 + 4 * half) = v;
            }
    }
}
// the transposed plane and the bias sums go through an LDS image [DK][NTOK + 8] of bf16 (tokens contiguous)
template <int DK, int NTOK>
__device__ __forceinline__ void grad_tile_write(uint16_t* tile, const f32x16 (&acc)[DK / 32], int tc, bool ok, int l31, int half) {
    constexpr int TS = NTOK + 8;
#pragma unroll
    for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const __bf16 hv = (__bf16)acc[dt][r];
            tile[(dt * 32 + acc_row(r, half)) * TS + tc + l31] = ok ? __builtin_bit_cast(uint16_t, hv) : (uint16_t)0;
        }
}
template <int DK, int NTOK, int NT = 256>
__device__ __forceinline__ void grad_tile_flush(const uint16_t* tile, const GradOut& g, int b, int h, int tok0, int S, int tid) {
    constexpr int TS = NTOK + 8, GR = NTOK / 8;
    if (g.hiT) {
        uint16_t* base = g.hiT + (int64_t)(h * DK) * g.t_ld + (int64_t)b * S + tok0;
        const bool vec = (S % 8 == 0) && (g.t_ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(g.hiT) & 15) == 0);
        if (vec) {
            for (int idx = tid; idx < DK * GR; idx += NT) {
                const int row = idx / GR, gq = idx % GR;
                if (tok0 + gq * 8 < S)
                    *reinterpret_cast<u32x4*>(base + (int64_t)row * g.t_ld + gq * 8) = *reinterpret_cast<const u32x4*>(tile + row * TS + gq * 8);
            }
        } else {
            for (int idx = tid; idx < DK * NTOK; idx += NT) {
                const int row = idx / NTOK, c = idx % NTOK;
                if (tok0 + c < S) base[(int64_t)row * g.t_ld + c] = tile[row * TS + c];
            }
        }
    }
    if (g.bsum) {
        for (int d = tid; d < DK; d += NT) {
            float sum = 0.f;
#pragma unroll
            for (int gq = 0; gq < GR; ++gq) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(tile + d * TS + gq * 8);
#pragma unroll
                for (int c = 0; c < 4; ++c) sum += __uint_as_float(v[c] << 16) + __uint_as_float(v[c] & 0xffff0000u);
            }
            atomicAdd(g.bsum + h * DK + d, sum);
        }
    }
}

// ---- gradient tile epilogue, row-major through LDS.  The tile's accumulators are G^T[d][token] with the token on the lane: written straight
// to the plane a lane stores 8 bytes per (d-tile, register quad) at a 2-KB row stride -- 32 partial-line requests per instruction, 64
// instructions per wave; measured on attn_bwd_dkvg_kernel (profiles/r03_h_split_probes.txt): 73 of the kernel's 154 us.  Here every wave
// writes its registers into an image [128 tokens][d_k + 8] (ds_write_b64), the workgroup stores the image as whole 512-byte rows (16 bytes per
// lane, consecutive lanes along d) and takes the bias column sums from the same image.
// a workgroup barrier for LDS hand-overs inside an epilogue: __syncthreads() is a fence too -- it waits vmcnt(0), i.e. until every global
// store the wave has issued is acknowledged -- and the epilogues below issue a gradient tile's 64 KB of row stores right before their
// barriers: each tile's stores were drained twice on the critical path of a workgroup that owns its CU alone (round 6).  LDS writes are
// complete at lgkmcnt(0); the "memory" clobbers keep the compiler's LDS accesses on their side of the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int DK, int NTL>
__device__ __forceinline__ void grad_rm_write(uint16_t* img, const f32x16 (&acc)[NTL], int trow, bool ok, int hh, int dt0) {
    constexpr int PITCH = DK + 8;
    uint16_t* row = img + trow * PITCH + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < NTL; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 v;
            v[0] = ok ? pack_bf2(acc[dt][4 * i + 0], acc[dt][4 * i + 1]) : 0u;
            v[1] = ok ? pack_bf2(acc[dt][4 * i + 2], acc[dt][4 * i + 3]) : 0u;
            *reinterpret_cast<u32x2*>(row + (dt0 + dt) * 32 + 8 * i) = v;
        }
}
// (called by all NT threads after a barrier; `red` = NT * 2 floats of LDS scratch behind the image)
template <int DK, int NT>
__device__ __forceinline__ void grad_rm_flush(const uint16_t* img, float* red, const GradOut& g, int b, int h, int tok0, int S, int SP, int tid) {
    // S: rows of this (batch) sequence that exist; SP: the PADDED sequence length the per-tile partial rows are laid out for (packed rows: SP >= S)
    constexpr int PITCH = DK + 8, CPRW = DK / 8;
    if (g.hi) {
        uint16_t* base = g.hi + (int64_t)b * g.h_bs + (int64_t)tok0 * g.h_ld + h * DK;
#pragma unroll 4
        for (int c = tid; c < 128 * CPRW; c += NT) {
            const int row = c / CPRW, col = c % CPRW;
            if (tok0 + row < S)
                *reinterpret_cast<u32x4*>(base + (int64_t)row * g.h_ld + col * 8) = *reinterpret_cast<const u32x4*>(img + row * PITCH + col * 8);
        }
    }
    if (g.f32) {
        float* base = g.f32 + (int64_t)b * g.f_bs + (int64_t)tok0 * g.f_ld + h * DK;
        for (int c = tid; c < 128 * (DK / 4); c += NT) {
            const int row = c / (DK / 4), col = c % (DK / 4);
            if (tok0 + row < S) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(img + row * PITCH + col * 4);
                *reinterpret_cast<float4*>(base + (int64_t)row * g.f_ld + col * 4) =
                    make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u));
            }
        }
    }
    if (g.bsum) {
        constexpr int NCW = DK / 2, NRB = NT / NCW;          // dword columns (two d values), row blocks
        const int cw = tid % NCW, rb = tid / NCW;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int row = rb; row < 128; row += NRB) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(img + row * PITCH + 2 * cw);
            s0 += __uint_as_float(v << 16);
            s1 += __uint_as_float(v & 0xffff0000u);
        }
        red[2 * tid] = s0;
        red[2 * tid + 1] = s1;
        lds_barrier();
        if (tid < NCW) {
#pragma unroll
            for (int r = 1; r < NRB; ++r) { s0 += red[2 * (tid + r * NCW)]; s1 += red[2 * (tid + r * NCW) + 1]; }
            if (g.bpart) {      // one row of partial sums per tile: ~900 workgroups adding into the same 1024 floats took 30 us of atomics
                *reinterpret_cast<float2*>(g.bpart + ((int64_t)b * ((SP + 127) / 128) + tok0 / 128) * g.bp_ld + h * DK + 2 * cw) = make_float2(s0, s1);
            } else {
                atomicAdd(g.bsum + h * DK + 2 * cw, s0);
                atomicAdd(g.bsum + h * DK + 2 * cw + 1, s1);
            }
        }
    }
}
// the whole epilogue of one gradient tile (NT threads; the transposed plane, if anybody asks for it, still goes through the old image)
template <int DK, int NTL, int NT>
__device__ __forceinline__ void grad_rm_epilogue(char* smem, const GradOut& g, const f32x16 (&acc)[NTL], int b, int h, int tok0, int trow, bool ok,
                                                 int hh, int dt0, int S, int SP, int tid) {
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + 128 * (DK + 8) * 2);
    grad_rm_write<DK, NTL>(img, acc, trow, ok, hh, dt0);
    lds_barrier();
    grad_rm_flush<DK, NT>(img, red, g, b, h, tok0, S, SP, tid);
    lds_barrier();
}

// the same image from the 16x16 accumulator layout of the 16-wide kernels: acc[dt][r] = G^T[d = 16 dt + 4 g + r][token = this lane's column c]
template <int DK>
__device__ __forceinline__ void grad_rm_write16(uint16_t* img, const f32x4 (&acc)[DK / 16], int trow, bool ok, int g) {
    constexpr int PITCH = DK + 8;
    uint16_t* row = img + trow * PITCH + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt) {
        u32x2 v;
        v[0] = ok ? pack_bf2(acc[dt][0], acc[dt][1]) : 0u;
        v[1] = ok ? pack_bf2(acc[dt][2], acc[dt][3]) : 0u;
        *reinterpret_cast<u32x2*>(row + dt * 16) = v;
    }
}
template <int DK, int NT>
__device__ __forceinline__ void grad_rm_epilogue16(char* smem, const GradOut& g_, const f32x4 (&acc)[DK / 16], int b, int h, int tok0, int trow, bool ok,
                                                   int g, int S, int SP, int tid) {
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + 128 * (DK + 8) * 2);
    grad_rm_write16<DK>(img, acc, trow, ok, g);
    lds_barrier();
    grad_rm_flush<DK, NT>(img, red, g_, b, h, tok0, S, SP, tid);
    lds_barrier();
}

// =================================================================================== forward
constexpr float RESCALE_TAU = 8.f;   // in units of the scaled scores (natural log): P <= e^8

// F16: Q / K / V planes hold fp16 values and P is rounded to fp16 (one pass, v_mfma_*_f16) -- the forward's operand policy
template <int DK, int NPASS, bool F16 = false>
__global__ __launch_bounds__(256, 1) void attn_fwd_bf16_kernel(const AttnPB p) {
    static_assert(!F16 || NPASS == 1, "fp16 operands: single pass");
    using G = Geo<DK>;
    constexpr int BC = G::BC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* sKh = reinterpret_cast<u32x4*>(smem);
    u32x2* sVh = reinterpret_cast<u32x2*>(smem + G::K_BYTES);
    u32x4* sKl = reinterpret_cast<u32x4*>(smem + G::K_BYTES + G::V_BYTES);
    u32x2* sVl = reinterpret_cast<u32x2*>(smem + 2 * G::K_BYTES + G::V_BYTES);
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + (NPASS == 3 ? 2 : 1) * (G::K_BYTES + G::V_BYTES));
    int* sFlag = reinterpret_cast<int*>(sMask + 64);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;

    const int64_t koff = (int64_t)b * p.bsk + h * p.kvh, voff = (int64_t)b * p.bsv + h * p.kvh;

    bf16x8 qh[DK / 16], ql[DK / 16];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * half;
#pragma unroll
        for (int s = 0; s < DK / 16; ++s) {
            qh[s] = ldfrag(p.Qh + qo + 16 * s, qok);
            if constexpr (NPASS == 3) ql[s] = ldfrag(p.Ql + qo + 16 * s, qok);
        }
    }

    f32x16 o[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;

    u32x4 kh[rows_n<DK, BC>()], kl[rows_n<DK, BC>()];
    u32x2 vh[rowsT_n<DK, BC>() * 4], vl[rowsT_n<DK, BC>() * 4];
    const int ntile = (p.Sk + BC - 1) / BC;
    // Register arrays are filled and drained inside ONE loop iteration (fetch tile t+1 -> MFMAs of tile t -> barrier ->
    // write tile t+1 to LDS -> barrier): a loop-carried or conditionally written array is demoted to scratch by hipcc,
    // with a vmcnt(0) after every load (measured: profiles/r01_*).  The last iteration re-fetches its own tile.
#define BMT_FWD_FETCH(key0_)                                                        \
    do {                                                                            \
        tile_gload<DK, BC>(p.Kh + koff, p.ldk, (key0_), p.Sk, tid, kh);             \
        tileT_gload<DK, BC>(p.Vh + voff, p.ldv, (key0_), p.Sk, tid, vh);            \
        if constexpr (NPASS == 3) {                                                 \
            tile_gload<DK, BC>(p.Kl + koff, p.ldk, (key0_), p.Sk, tid, kl);         \
            tileT_gload<DK, BC>(p.Vl + voff, p.ldv, (key0_), p.Sk, tid, vl);        \
        }                                                                           \
    } while (0)
#define BMT_FWD_STORE(key0_)                                                        \
    do {                                                                            \
        tile_lstore<DK, BC>(sKh, tid, kh);                                          \
        tileT_lstore<DK, BC>(sVh, tid, vh);                                         \
        if constexpr (NPASS == 3) {                                                 \
            tile_lstore<DK, BC>(sKl, tid, kl);                                      \
            tileT_lstore<DK, BC>(sVl, tid, vl);                                     \
        }                                                                           \
        stage_mask<BC>(p, b, (key0_), tid, sMask, sFlag);                           \
    } while (0)
    BMT_FWD_FETCH(0);
    BMT_FWD_STORE(0);
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * BC;
        const int kn = min(key0 + BC, (ntile - 1) * BC);
        BMT_FWD_FETCH(kn);
        const int flag = sFlag[0];
        if (flag != 0) {
#pragma unroll
            for (int sub = 0; sub < G::NSUB; ++sub) {
                f32x16 st;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
                for (int s = 0; s < DK / 16; ++s) {
                    const int idx = kslot<DK>(sub * 32 + l31, 2 * s + half);
                    const bf16x8 a = as_bf16x8(sKh[idx]);
                    if constexpr (NPASS == 3) {
                        st = mfma32(as_bf16x8(sKl[idx]), qh[s], st);
                        st = mfma32(a, ql[s], st);
                    }
                    st = mfma32t<F16>(a, qh[s], st);
                }
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = st[r] * p.scale;
                if (flag != 2) {
                    if (p.mask != nullptr && p.mask_qs != 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = key0 + sub * 32 + acc_row(r, half);
                            const bool ok = qok && key < p.Sk &&
                                            p.mask[(int64_t)b * p.mask_bs + (int64_t)q * p.mask_qs + key] != 0;
                            pv[r] = ok ? pv[r] : NEG_INF;
                        }
                    } else {
                        // this lane's 16 keys are four 4-byte groups of the staged mask bytes
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint32_t mw = *reinterpret_cast<const uint32_t*>(sMask + sub * 32 + 8 * g + 4 * half);
#pragma unroll
                            for (int c = 0; c < 4; ++c) pv[4 * g + c] = ((mw >> (8 * c)) & 0xffu) ? pv[4 * g + c] : NEG_INF;
                        }
                    }
                }
                float tmax = pv[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, pv[r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                // Online softmax with a STALE reference: exponentials are taken relative to m_run, which is only moved (and the
                // 32 x d_k accumulator only rescaled: 128 AGPR reads, multiplies and writes per lane) when some query of this
                // wave sees a score more than RESCALE_TAU above it.  softmax is shift invariant, so this is exact; P stays
                // below e^TAU (fp32 accumulation, bf16 operands keep their relative precision).  With the exact running max
                // the rescale ran on ~90 % of the key tiles (32 queries x fresh keys), at VALU cost comparable to the MFMAs.
                if (__any(tmax > m_run + RESCALE_TAU)) {
                    const float m_new = fmaxf(m_run, tmax);
                    const float alpha = __expf(m_run - ((m_new == NEG_INF) ? 0.f : m_new));
                    l_run *= alpha;
#pragma unroll
                    for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                    m_run = m_new;
                }
                const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = __expf(pv[r] - m_use);
                    psum += pv[r];
                }
                psum += __shfl_xor(psum, 32, 64);
                l_run += psum;
                bf16x8 ph[2], pl[2];
                pack_p<NPASS, F16>(pv, 0, ph[0], pl[0]);
                pack_p<NPASS, F16>(pv, 1, ph[1], pl[1]);
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int d = dt * 32 + l31, kg = sub * 8 + 4 * s2 + half;
                        const bf16x8 a = tfrag<BC>(sVh, d, kg);
                        if constexpr (NPASS == 3) {
                            o[dt] = mfma32(tfrag<BC>(sVl, d, kg), ph[s2], o[dt]);
                            o[dt] = mfma32(a, pl[s2], o[dt]);
                        }
                        o[dt] = mfma32t<F16>(a, ph[s2], o[dt]);
                    }
            }
        }
        __syncthreads();
        BMT_FWD_STORE(kn);
        __syncthreads();
    }
#undef BMT_FWD_FETCH
#undef BMT_FWD_STORE

    if (qok) {
        const float inv = 1.f / l_run;   // fully masked row: 0 * inf = NaN, as the reference's softmax
        const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
        const int64_t rowoff = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK;
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
                if (p.Ow) *reinterpret_cast<float4*>(p.Ow + rowoff + d) = v;
                if (p.Owh) {      // operand planes of the out-projection, written here instead of by a conversion pass
                    const int64_t po = (int64_t)b * p.bsop + (int64_t)min(q, p.Sq - 1) * p.ldop + h * DK + d;
                    uint32_t h0, l0, h1, l1;
                    split_bf2(v.x, v.y, h0, l0);
                    split_bf2(v.z, v.w, h1, l1);
                    if (p.ow_f16) { l0 = pack_h2(v.x, v.y); l1 = pack_h2(v.z, v.w); }
                    u32x2 hh, ll;
                    hh[0] = h0; hh[1] = h1; ll[0] = l0; ll[1] = l1;
                    *reinterpret_cast<u32x2*>(p.Owh + po) = hh;
                    if (p.Owl) *reinterpret_cast<u32x2*>(p.Owl + po) = ll;
                }
            }
        if (half == 0) p.lsew[((int64_t)b * p.H + h) * p.Sq + q] = m_run + __logf(l_run);
    }
}

// ---- padded row-major image + hardware transpose read (gfx950 ds_read_b64_tr_b16), used by the 16-wide kernels.
// Image: [ROWS][DK] bf16, row stride DK*2 + 32 bytes.  It serves two access patterns without bank conflicts:
//   * row fragments (A/B operand with the reduction along d): lane (row c, 16-B slot s) -> ds_read_b128;
//   * column fragments (operand [16 d x 32 rows], reduction along the ROWS): ds_read_b64_tr_b16.  Measured semantics
//     (tools/probes/tr_probe.hip): within a 16-lane group lane m supplies the address of a 4-element chunk and lane i receives
//     element j = chunk[4 j + (i >> 2)][i & 3].  With lane m pointing at image[r0 + (m >> 2)][d0 + 4 (m & 3) ..] lane i gets
//     image[r0 + j][d0 + i], j < 4: a 4-row x 16-column block transposed in flight.  Two reads (rows r0 = 4 g and 16 + 4 g)
//     give lane (c, g) the MFMA fragment for reduction indices 8 g + j in exactly the order the preceding product leaves its
//     probabilities in (see attn_fwd16_kernel).  The d-tile enters as a compile-time byte offset: one address VGPR per kernel.
// This replaces the separately loaded and transposed images (a second global read of every tile, 4x4 register transposes and
// 8-byte LDS stores).
template <int DK> constexpr int pad_rs() { return DK * 2 + 32; }
template <int DK, int ROWS, int NT>
__device__ __forceinline__ void tile_lstore_pad(char* img, int tid, const u32x4 (&v)[rows_n<DK, ROWS, NT>()]) {
    constexpr int SPR = DK / 8;
#pragma unroll
    for (int i = 0; i < rows_n<DK, ROWS, NT>(); ++i) {
        const int c = tid + NT * i;
        if constexpr (ROWS * SPR % NT != 0) {
            if (c >= ROWS * SPR) continue;
        }
        *reinterpret_cast<u32x4*>(img + (c / SPR) * pad_rs<DK>() + (c % SPR) * 16) = v[i];
    }
}
template <int DK>
__device__ __forceinline__ bf16x8 rowfrag_pad(const char* img, int row, int slot) {
    return as_bf16x8(*reinterpret_cast<const u32x4*>(img + row * pad_rs<DK>() + slot * 16));
}
typedef short v4s16 __attribute__((ext_vector_type(4)));
// lane_base = img + (4 g + (c >> 2)) * RS + 8 * (c & 3)   (bytes); fragment of d-tile dt = columns [16 dt, 16 dt + 16)
template <int DK>
__device__ __forceinline__ bf16x8 trfrag(const char* lane_base, int dt) {
    typedef v4s16 __attribute__((address_space(3))) * lds_v4s;
    const v4s16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lane_base + dt * 32));
    const v4s16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lane_base + dt * 32 + 16 * pad_rs<DK>()));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ int tr_lane_off(int rs, int c, int g) { return (4 * g + (c >> 2)) * rs + 8 * (c & 3); }

// =================================================================================== forward, 8 waves x 16 queries
// Same algorithm, different decomposition, for d_k >= 128.  The 4-wave kernel above gives each wave 32 queries: its
// 32 x d_k accumulator (128 registers at d_k = 256) plus the Q fragments push it past 256 registers, so it runs ONE wave per
// SIMD, hipcc selects AGPR-form MFMAs and copies the accumulator between the register files every key tile, and MFMA, VALU
// and LDS phases of the single wave serialise (PMC, profiles/r01_e_attn_pmc.csv: MFMA busy 20 %, VALU 25 %, waiting 41 %).
// Here a wave owns 16 queries (v_mfma_f32_16x16x32_bf16): 16 x d_k accumulator = 64 registers, everything fits in 256
// VGPRs, MFMAs are VGPR-form (the VALU rescales the accumulator in place) and TWO waves share each SIMD, so one wave's
// softmax runs under the other's MFMAs.  Cost: K / V^T fragments are re-read from LDS by twice as many waves.
//   S^T tile [16 keys x 16 q] = K[16 x 32] . Q^T[32 x 16]:   A = K rows (LDS), B = Q (registers);  lane (c = l&15, g = l>>4)
//   holds S^T[key = 4g + r][q = c], r < 4.  Two key tiles per 32-key stage -> this lane's 8 probabilities are keys
//   {4g..4g+3} and {16+4g..16+4g+3}: exactly the B operand of O^T[16 d x 16 q] += V^T[16 x 32] . P^T[32 x 16] if the MFMA's
//   reduction index kk = 8g + j is read as key(kk) = (j < 4 ? 4g + j : 16 + 4g + j - 4); the A operand (V^T image, 4-key
//   units) then takes units g and 4 + g.  No cross-lane movement between the two products.
typedef __attribute__((ext_vector_type(4))) float f32x4v;
__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) {
    // D[16x16] += A[16x32] * B[32x16]; lane l: A[row = l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4) + j][col = l&15],
    // D[row = 4*(l>>4) + r][col = l&15]
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4v mfma16t(bf16x8 a, bf16x8 b, f32x4v c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int DK, int NPASS, bool F16 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd16_kernel(const AttnPB p) {
    static_assert(!F16 || NPASS == 1, "fp16 operands: single pass");
    constexpr int BC = 32, NT = 512, KS = DK / 32, DT = DK / 16;
    constexpr int KB = BC * pad_rs<DK>(), VB = BC * pad_rs<DK>();   // padded rows: K read by row, V through the transpose unit
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sKh = smem;
    char* sVh = smem + KB;
    char* sKl = smem + KB + VB;
    char* sVl = smem + 2 * KB + VB;
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + (NPASS == 3 ? 2 : 1) * (KB + VB));
    int* sFlag = reinterpret_cast<int*>(sMask + 64);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 16 + c;
    const bool qok = q < p.Sq;
    const int64_t koff = (int64_t)b * p.bsk + h * p.kvh, voff = (int64_t)b * p.bsv + h * p.kvh;

    bf16x8 qh[KS], ql[KS];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[ks] = ldfrag(p.Qh + qo + 32 * ks, qok);
            if constexpr (NPASS == 3) ql[ks] = ldfrag(p.Ql + qo + 32 * ks, qok);
        }
    }
    f32x4v o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float m_run = NEG_INF, l_run = 0.f;
    const int troff = tr_lane_off(pad_rs<DK>(), c, g);

    u32x4 kh[rows_n<DK, BC, NT>()], kl[rows_n<DK, BC, NT>()];
    u32x4 vh[rows_n<DK, BC, NT>()], vl[rows_n<DK, BC, NT>()];
    const int ntile = (p.Sk + BC - 1) / BC;
#define BMT_F16_FETCH(key0_)                                                            \
    do {                                                                                \
        tile_gload<DK, BC, NT>(p.Kh + koff, p.ldk, (key0_), p.Sk, tid, kh);             \
        tile_gload<DK, BC, NT>(p.Vh + voff, p.ldv, (key0_), p.Sk, tid, vh);            \
        if constexpr (NPASS == 3) {                                                     \
            tile_gload<DK, BC, NT>(p.Kl + koff, p.ldk, (key0_), p.Sk, tid, kl);         \
            tile_gload<DK, BC, NT>(p.Vl + voff, p.ldv, (key0_), p.Sk, tid, vl);        \
        }                                                                               \
    } while (0)
#define BMT_F16_STORE(key0_)                                                            \
    do {                                                                                \
        tile_lstore_pad<DK, BC, NT>(sKh, tid, kh);                                          \
        tile_lstore_pad<DK, BC, NT>(sVh, tid, vh);                                         \
        if constexpr (NPASS == 3) {                                                     \
            tile_lstore_pad<DK, BC, NT>(sKl, tid, kl);                                      \
            tile_lstore_pad<DK, BC, NT>(sVl, tid, vl);                                     \
        }                                                                               \
        stage_mask<BC>(p, b, (key0_), tid, sMask, sFlag);                               \
    } while (0)
    BMT_F16_FETCH(0);
    BMT_F16_STORE(0);
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * BC;
        const int kn = min(key0 + BC, (ntile - 1) * BC);
        BMT_F16_FETCH(kn);
        const int flag = sFlag[0];
        if (flag != 0) {
            f32x4v st[2];
            st[0] = f32x4v{0.f, 0.f, 0.f, 0.f};
            st[1] = st[0];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const bf16x8 a = rowfrag_pad<DK>(sKh, kt * 16 + c, 4 * ks + g);
                    if constexpr (NPASS == 3) {
                        st[kt] = mfma16(rowfrag_pad<DK>(sKl, kt * 16 + c, 4 * ks + g), qh[ks], st[kt]);
                        st[kt] = mfma16(a, ql[ks], st[kt]);
                    }
                    st[kt] = mfma16t<F16>(a, qh[ks], st[kt]);
                }
            float pv[8];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[4 * kt + r] = st[kt][r] * p.scale;
            if (flag != 2) {
                if (p.mask != nullptr && p.mask_qs != 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int key = key0 + 16 * (i >> 2) + 4 * g + (i & 3);
                        const bool ok = qok && key < p.Sk && p.mask[(int64_t)b * p.mask_bs + (int64_t)q * p.mask_qs + key] != 0;
                        pv[i] = ok ? pv[i] : NEG_INF;
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const uint32_t mw = *reinterpret_cast<const uint32_t*>(sMask + 16 * kt + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[4 * kt + r] = ((mw >> (8 * r)) & 0xffu) ? pv[4 * kt + r] : NEG_INF;
                    }
                }
            }
            float tmax = pv[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) tmax = fmaxf(tmax, pv[i]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m_run + RESCALE_TAU)) {     // stale-reference online softmax, see the 4-wave kernel
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __expf(m_run - ((m_new == NEG_INF) ? 0.f : m_new));
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
                m_run = m_new;
            }
            const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pv[i] = __expf(pv[i] - m_use);
                psum += pv[i];
            }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l_run += psum;
            u32x4 phw, plw = {0u, 0u, 0u, 0u};
            if constexpr (NPASS == 3) {
                uint32_t hh, ll;
                split_bf2(pv[0], pv[1], hh, ll); phw[0] = hh; plw[0] = ll;
                split_bf2(pv[2], pv[3], hh, ll); phw[1] = hh; plw[1] = ll;
                split_bf2(pv[4], pv[5], hh, ll); phw[2] = hh; plw[2] = ll;
                split_bf2(pv[6], pv[7], hh, ll); phw[3] = hh; plw[3] = ll;
            } else {
                phw[0] = pack_2<F16>(pv[0], pv[1]); phw[1] = pack_2<F16>(pv[2], pv[3]);
                phw[2] = pack_2<F16>(pv[4], pv[5]); phw[3] = pack_2<F16>(pv[6], pv[7]);
            }
            const bf16x8 ph = as_bf16x8(phw), pl = as_bf16x8(plw);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 a = trfrag<DK>(sVh + troff, dt);
                if constexpr (NPASS == 3) {
                    o[dt] = mfma16(trfrag<DK>(sVl + troff, dt), ph, o[dt]);
                    o[dt] = mfma16(a, pl, o[dt]);
                }
                o[dt] = mfma16t<F16>(a, ph, o[dt]);
            }
        }
        __syncthreads();
        BMT_F16_STORE(kn);
        __syncthreads();
    }
#undef BMT_F16_FETCH
#undef BMT_F16_STORE

    if (qok) {
        const float inv = 1.f / l_run;   // fully masked row: 0 * inf = NaN, as the reference's softmax
        const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
        const int64_t rowoff = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + 4 * g;
            float4 v;
            v.x = drop_apply(dc, o[dt][0] * inv, (uint64_t)(rowoff + d + 0));
            v.y = drop_apply(dc, o[dt][1] * inv, (uint64_t)(rowoff + d + 1));
            v.z = drop_apply(dc, o[dt][2] * inv, (uint64_t)(rowoff + d + 2));
            v.w = drop_apply(dc, o[dt][3] * inv, (uint64_t)(rowoff + d + 3));
            if (p.Ow) *reinterpret_cast<float4*>(p.Ow + rowoff + d) = v;
            if (p.Owh) {
                const int64_t po = (int64_t)b * p.bsop + (int64_t)min(q, p.Sq - 1) * p.ldop + h * DK + d;
                uint32_t h0, l0, h1, l1;
                split_bf2(v.x, v.y, h0, l0);
                split_bf2(v.z, v.w, h1, l1);
                if (p.ow_f16) { l0 = pack_h2(v.x, v.y); l1 = pack_h2(v.z, v.w); }
                u32x2 hh, ll;
                hh[0] = h0; hh[1] = h1; ll[0] = l0; ll[1] = l1;
                *reinterpret_cast<u32x2*>(p.Owh + po) = hh;
                if (p.Owl) *reinterpret_cast<u32x2*>(p.Owl + po) = ll;
            }
        }
        if (g == 0) p.lsew[((int64_t)b * p.H + h) * p.Sq + q] = m_run + __logf(l_run);
    }
}

// =================================================================================== forward, single pass, 64-key stages
// The single-pass forward (fp16 or bf16 operands) rebuilt around its instruction budget.  PMC / ISA of attn_fwd16_kernel
// (profiles/r02_*): per 32-key stage a wave issues 32 MFMAs (~540 matrix-pipe cycles) next to ~270 vector, ~175 scalar and ~60 LDS
// instructions -- the loop is issue-bound, not MFMA- or LDS-bound.  Here:
//   * 64-key stages (the per-stage bookkeeping -- mask classification, rescale test, loop control -- is paid half as often);
//   * K / V tiles are fetched with buffer loads: per-lane byte offsets computed once, the key offset of a stage in an SGPR,
//     rows past Sk read as zero by the descriptor's bounds check (no clamp / 64-bit address arithmetic per stage);
//   * the LDS images are double buffered: the next stage is written while the current one is read, ONE barrier per stage;
//   * softmax in the log2 domain: p = exp2(fma(s, scale * log2 e, -m)) is one fma + one v_exp_f32 per score;
//   * the two cross-lane reductions (max, sum over the lanes l ^ 16, l ^ 32 that share a query) use v_permlane16/32_swap
//     instead of ds_bpermute round trips.
// Same decomposition as attn_fwd16_kernel: 8 waves x 16 queries, S^T = K.Q^T with v_mfma_f32_16x16x32, P^T feeds O^T += V^T.P^T
// from registers, V^T fragments through ds_read_b64_tr_b16 from the padded row image.
// v_permlane16_swap / v_permlane32_swap exchange lanes between TWO registers: with both holding x, afterwards a = {x.row0, x.row0,
// x.row2, x.row2} and b = {x.row1, x.row1, x.row3, x.row3} (rows of 16 lanes) resp. a = {x.lo32, x.lo32}, b = {x.hi32, x.hi32}
// (tools/probes/permlane_probe.hip): op(a, b) is the reduction over lanes l ^ 16 resp. l ^ 32 without an LDS round trip.  Inline
// asm: hipcc (ROCm 7.2) folds the builtin's two results into one when they feed a commutative op (max(r0, r1) vanished, r0 + r1
// became r0 + r0); the `s_nop 1` covers the VALU-write -> permlane-read hazard the compiler would otherwise pad.
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float xlane_max(float x) {          // max over lanes {l, l^16, l^32, l^48}
    float a = x, b = x;
    swap16(a, b);
    a = b = fmaxf(a, b);
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float xlane_sum(float x) {
    float a = x, b = x;
    swap16(a, b);
    a = b = a + b;
    swap32(a, b);
    return a + b;
}
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

template <int DK, bool F16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd64_kernel(const AttnPB pin) {
    AttnPB p = pin;
    constexpr int BC = 64, NT = 512, KS = DK / 32, DT = DK / 16, RS = pad_rs<DK>();
    constexpr int TILE = BC * RS, STAGE = 2 * TILE;            // K image | V image
    constexpr int NR = rows_n<DK, BC, NT>();                   // 16-byte slots per thread and operand (4 at d_k 256)
    constexpr int SPR = DK / 8;
    static_assert(BC * SPR % NT == 0, "whole slots per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + 2 * STAGE);       // [2][64]
    int* sFlag = reinterpret_cast<int*>(sMask + 128);                    // [2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bhw = w / nqt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (qt * 128 >= p.Sq) return;            // (a query tile past the sample's length)
    const int q = qt * 128 + wid * 16 + c;
    const bool qok = q < p.Sq;
    // a wave whose 16 queries all lie past Sq (the decoder: 29 queries per tile of 128) only helps to stage the tiles
    const bool wave_on = qt * 128 + wid * 16 < p.Sq;

    bf16x8 qf[KS];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = ldfrag(p.Qh + qo + 32 * ks, qok);
    }
    f32x4v o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float m_run = NEG_INF, l_run = 0.f;       // running maximum in log2 units (scores are scaled by scale * log2 e)
    const float sc2 = p.scale * LOG2E;
    constexpr float TAU2 = RESCALE_TAU * LOG2E;
    const int troff = tr_lane_off(RS, c, g);

    // K / V rows of this (b, h): descriptors end after the last key row, so a stage that runs past Sk reads zeros there
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * p.kvh), 0, plane_extent(p.Sk, p.ldk, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * p.kvh), 0, plane_extent(p.Sk, p.ldv, DK), 0x00020000);
    int kvo[NR], vvo[NR], lso[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int s_ = tid + NT * i;
        kvo[i] = (s_ / SPR) * (int)p.ldk * 2 + (s_ % SPR) * 16;
        vvo[i] = (s_ / SPR) * (int)p.ldv * 2 + (s_ % SPR) * 16;
        lso[i] = (s_ / SPR) * RS + (s_ % SPR) * 16;
    }
    const int ntile = (p.Sk + BC - 1) / BC;
    u32x4 kr[NR], vr[NR];
#define BMT_F64_FETCH(t_)                                                                              \
    do {                                                                                               \
        const int so_k = (t_) * BC * (int)p.ldk * 2, so_v = (t_) * BC * (int)p.ldv * 2;                \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo[i], so_k, 0); \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo[i], so_v, 0); \
    } while (0)
#define BMT_F64_STORE(t_, buf_)                                                                        \
    do {                                                                                               \
        char* sk_ = smem + (buf_) * STAGE;                                                             \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) *reinterpret_cast<u32x4*>(sk_ + lso[i]) = kr[i];        \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) *reinterpret_cast<u32x4*>(sk_ + TILE + lso[i]) = vr[i]; \
        stage_mask<BC>(p, b, (t_) * BC, tid, sMask + (buf_) * 64, sFlag + (buf_));                      \
    } while (0)
    BMT_F64_FETCH(0);
    BMT_F64_STORE(0, 0);
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int tn = min(t + 1, ntile - 1);              // the last stage re-fetches itself (branch-free)
        BMT_F64_FETCH(tn);
        const int flag = sFlag[cur];
        if (flag != 0 && wave_on) {
            const char* sK = smem + cur * STAGE;
            const char* sV = sK + TILE;
            const int key0 = t * BC;
            f32x4v st[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) st[kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    st[kt] = mfma16t<F16>(rowfrag_pad<DK>(sK, kt * 16 + c, 4 * ks + g), qf[ks], st[kt]);
            __builtin_amdgcn_s_setprio(0);
            float x[16];                                   // scores in log2 units; lane (c, g): key = key0 + 16 kt + 4 g + r, query c
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[4 * kt + r] = st[kt][r] * sc2;
            if (flag != 2) {
                if (p.mask != nullptr && p.mask_qs != 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = key0 + 16 * (i >> 2) + 4 * g + (i & 3);
                        const bool ok = qok && key < p.Sk && p.mask[(int64_t)b * p.mask_bs + (int64_t)q * p.mask_qs + key] != 0;
                        x[i] = ok ? x[i] : NEG_INF;
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) {
                        const uint32_t mw = *reinterpret_cast<const uint32_t*>(sMask + cur * 64 + 16 * kt + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[4 * kt + r] = ((mw >> (8 * r)) & 0xffu) ? x[4 * kt + r] : NEG_INF;
                    }
                }
            }
            float tmax = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
#pragma unroll
            for (int i = 4; i < 16; i += 4) tmax = fmaxf(tmax, fmaxf(fmaxf(x[i], x[i + 1]), fmaxf(x[i + 2], x[i + 3])));
            tmax = xlane_max(tmax);
            if (__any(tmax > m_run + TAU2)) {              // stale-reference online softmax (see attn_fwd_bf16_kernel): exact
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == NEG_INF) ? 0.f : m_new));
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
                m_run = m_new;
            }
            const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                x[i] = __builtin_amdgcn_exp2f(x[i] - m_use);
                psum += x[i];
            }
            l_run += xlane_sum(psum);
            // B operands of O^T += V^T . P^T, one per 32-key half: reduction index kk = 8 g + j <-> key 4 g + j (j < 4) / 16 + 4 g + j - 4
            bf16x8 pf[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                u32x4 pw;
                pw[0] = pack_2<F16>(x[8 * hf + 0], x[8 * hf + 1]); pw[1] = pack_2<F16>(x[8 * hf + 2], x[8 * hf + 3]);
                pw[2] = pack_2<F16>(x[8 * hf + 4], x[8 * hf + 5]); pw[3] = pack_2<F16>(x[8 * hf + 6], x[8 * hf + 7]);
                pf[hf] = as_bf16x8(pw);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[dt] = mfma16t<F16>(trfrag<DK>(sV + troff, dt), pf[0], o[dt]);
                o[dt] = mfma16t<F16>(trfrag<DK>(sV + troff + 32 * RS, dt), pf[1], o[dt]);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        BMT_F64_STORE(tn, cur ^ 1);                        // nobody reads that image now: its readers passed the previous barrier
        __syncthreads();
    }
#undef BMT_F64_FETCH
#undef BMT_F64_STORE

    if (qok) {
        const float inv = 1.f / l_run;   // fully masked row: 0 * inf = NaN, as the reference's softmax
        const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
        const int64_t rowoff = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + 4 * g;
            float4 v;
            v.x = drop_apply(dc, o[dt][0] * inv, (uint64_t)(p.drop_off + rowoff + d + 0));
            v.y = drop_apply(dc, o[dt][1] * inv, (uint64_t)(p.drop_off + rowoff + d + 1));
            v.z = drop_apply(dc, o[dt][2] * inv, (uint64_t)(p.drop_off + rowoff + d + 2));
            v.w = drop_apply(dc, o[dt][3] * inv, (uint64_t)(p.drop_off + rowoff + d + 3));
            if (p.Ow) *reinterpret_cast<float4*>(p.Ow + rowoff + d) = v;
            if (p.Owh) {
                const int64_t po = (int64_t)b * p.bsop + (int64_t)min(q, p.Sq - 1) * p.ldop + h * DK + d;
                uint32_t h0, l0, h1, l1;
                split_bf2(v.x, v.y, h0, l0);
                split_bf2(v.z, v.w, h1, l1);
                if (p.ow_f16) { l0 = pack_h2(v.x, v.y); l1 = pack_h2(v.z, v.w); }
                u32x2 hh, ll;
                hh[0] = h0; hh[1] = h1; ll[0] = l0; ll[1] = l1;
                *reinterpret_cast<u32x2*>(p.Owh + po) = hh;
                if (p.Owl) *reinterpret_cast<u32x2*>(p.Owl + po) = ll;
            }
        }
        if (g == 0) p.lsew[((int64_t)b * p.H + h) * p.SqP + q] = m_run * LN2 + __logf(l_run);
    }
}

// =================================================================================== forward, single pass, 32 queries per wave + LDS-DMA
// The encoder's large attention shapes (>= 2 workgroups per CU) on v_mfma_f32_32x32x16.
// Why: attn_fwd64_kernel (8 waves x 16 queries, v_mfma_f32_16x16x32) reads one 1-KB operand fragment from LDS per 16-cycle MFMA; four
// SIMDs ask for 256 B / clk, the LDS peak (MI355X_MICROARCH.md, LDS table), and its K / V staging adds 64 KB of ds_write_b128 per stage
// on a path that moves 79 B / clk: the loop measures 21-23 % MFMA busy (profiles/r02_h_attn_fwd64_pmc.csv).  Here:
//   * a wave owns 32 queries: S^T[32 keys x 32 q] = K[32 x 16] . Q^T[16 x 32] per MFMA (A = K rows from LDS, B = Q in registers),
//     O^T[32 d x 32 q] += V^T[32 x 16] . P^T[16 x 32] (A = V^T through ds_read_b64_tr_b16, B = P from the softmax registers):
//     1 KB of LDS per 32-cycle MFMA, half the bytes per matrix cycle;
//   * K / V tiles go global -> LDS by buffer_load ... lds (no staging registers -- the 32 x 256 accumulator takes 128 and Q 64 of the
//     256 registers two waves per SIMD leave -- and no ds_write); the swizzles that keep both read patterns conflict free sit on
//     the SOURCE side: K image chunk (16 B) position = chunk ^ (row & 15) [ds_read_b128 row fragments], V image position =
//     chunk ^ 4 (row & 3) [4-row x 16-column transposing reads: the four rows of a read land in the four 64-byte bank quarters];
//     SQ_LDS_BANK_CONFLICT = 0 (profiles/r02_q_attn_fwd32_pmc.csv);
//   * 4 waves (128 queries) per workgroup, 32-key stages, two stage buffers = 64 KB: TWO workgroups per CU, whose phases drift apart,
//     put one wave's softmax under the other's MFMAs (the role the second wave of a SIMD plays in the 8-wave kernels);
//   * the key-padding mask row of the batch element is staged into LDS once; no global load besides the DMA is in flight inside the
//     loop (cdna_hip_programming.md: an ordinary load beside an LDS-DMA makes hipcc drain vmcnt(0)).
// Lane algebra (l31 = lane & 31, hh = lane >> 5), checked by tools/probes/attn_fwd32_layout.py against numpy:
//   S^T register r = 4 i + j of lane (l31, hh): key 8 i + 4 hh + j, query l31.  The 16-key MFMA kk of O^T += V^T . P^T takes
//   registers 8 kk .. 8 kk + 7 in order as its B operand: reduction index 8 hh + jj <-> key 16 kk + 4 hh + jj (jj < 4),
//   16 kk + 8 + 4 hh + jj - 4 (jj >= 4); the A operand reads V rows 16 kk + 4 hh + (0..3) and 16 kk + 8 + 4 hh + (0..3).
// Measured (tools/probes/attn_fwd32_check.py, profiles/r02_q_*): configs[1] audio self-attention 194 -> 135 us (620 TF/s), audio <- video
// 96 -> 75 us; the shapes with one workgroup per CU (256 queries) gain nothing and stay on attn_fwd64_kernel.
__device__ __forceinline__ float half_max(float x) { float a = x, b = x; swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float x) { float a = x, b = x; swap32(a, b); return a + b; }

__device__ __forceinline__ void swap32u(uint32_t& a, uint32_t& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// LDS fragment reads as inline asm, waits counted by hand.  Two reasons: (1) with an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in
// front of the ds_read_tr builtin (it cannot tell the read from the DMA's target buffer) and drains the prefetch in the middle of the
// stage; (2) it issues read -> lgkmcnt(0) -> MFMA one fragment at a time, LDS latency exposed per MFMA.  The waits carry the fragment
// registers as "+v" operands so that the MFMA that consumes them cannot be scheduled above the wait.  LDS operations retire in order:
// when lgkmcnt <= N only the N youngest can be pending, whatever else the compiler has in flight -- its own waits only get stronger.
template <int OFF>
__device__ __forceinline__ u32x4 lds_b128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr_b64(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// compile-time repetition by the preprocessor: the step bodies need their index as a constant expression (immediate offsets and wait counts
// of the asm above)
#define BMT_X_REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
template <int N>
__device__ __forceinline__ void lgkm_wait(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void lgkm_wait(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

#ifndef BMT_FWD32_XP          // build-variant timing probes (BMT_VARIANT_FLAGS, csrc/build.sh)
#define BMT_FWD32_XP 0
#endif
// DMAV: where the next tile's 2 PPW DMA requests are issued (an experiment axis): 0 = one per MFMA on the first MFMAs of S, 1 = one per
// two MFMAs of S, 2 = all before S, 3 = K pieces one per four MFMAs of S + V pieces one per two MFMAs at the start of PV.
// PRIO: raise the wave's priority around its MFMA phases.
template <int DK, bool F16, int DMAV = 0, bool PRIO = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd32_kernel(const AttnPB pin) {
    AttnPB p = pin;
    constexpr int BC = 32, NT = 256, KS = DK / 16, DT = DK / 32, ROWB = DK * 2, TILE = BC * ROWB, STAGE = 2 * TILE;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NP = BC / RPP, PPW = NP / 4;       // 16-B chunks per row, rows per 1-KB piece, pieces per tile / wave
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];    // the fragment addresses XOR bits 5 .. 8: the base must not carry into them
    char* sMask = smem + 2 * STAGE;                                      // [ntile * 32] bytes: 1 = valid key

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bhw = w / nqt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (qt * 128 >= p.Sq) return;            // (a query tile past the sample's length)
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;
    const bool wave_on = qt * 128 + wid * 32 < p.Sq;                     // waves past Sq only move tiles
    const int ntile = (p.Sk + BC - 1) / BC;

    // ---- LDS-DMA: piece = 1 KB = RPP rows; wave w moves pieces w * PPW .. of the K and of the V tile
    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * p.kvh), 0, plane_extent(p.Sk, p.ldk, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * p.kvh), 0, plane_extent(p.Sk, p.ldv, DK), 0x00020000);
    // (fixed extent: with the template-dependent extent PPW the DMA builtin's call becomes type-dependent and hipcc 7.2's host pass drops
    // the whole kernel instantiation WITHOUT a diagnostic -- the library then fails to load with the kernel's stub undefined)
    int kvo[4], vvo[4];
    static_assert(PPW <= 4, "pieces per wave");
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wid * PPW + j) * RPP + lane / CPR, cpos = lane % CPR;
        kvo[j] = row * (int)p.ldk * 2 + ((cpos ^ (row & 15)) * 16);
        vvo[j] = row * (int)p.ldv * 2 + ((cpos ^ (4 * (row & 3))) * 16);
    }
    const int sstep_k = BC * (int)p.ldk * 2, sstep_v = BC * (int)p.ldv * 2;
#define BMT_X_DMA_K_(j_, t_, buf_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + (buf_) * STAGE + (wid * PPW + (j_)) * 1024), 16, kvo[j_], (t_) * sstep_k, 0, 0)
#define BMT_X_DMA_V_(j_, t_, buf_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + (buf_) * STAGE + TILE + (wid * PPW + (j_)) * 1024), 16, vvo[j_], (t_) * sstep_v, 0, 0)
    // (timing probes, build variants only: BMT_FWD32_XP bit 0 = the loop moves no tile -- every stage computes on what the prologue staged;
    // bit 1 = the loop computes nothing -- tiles move, barriers stay)
#define BMT_X_DMA_K(j_, t_, buf_) do { if constexpr (!(BMT_FWD32_XP & 1)) BMT_X_DMA_K_(j_, t_, buf_); } while (0)
#define BMT_X_DMA_V(j_, t_, buf_) do { if constexpr (!(BMT_FWD32_XP & 1)) BMT_X_DMA_V_(j_, t_, buf_); } while (0)

    // stage 0 in flight first, then the Q fragments and the mask row
#pragma unroll
    for (int j = 0; j < PPW; ++j) BMT_X_DMA_K_(j, 0, 0);
#pragma unroll
    for (int j = 0; j < PPW; ++j) BMT_X_DMA_V_(j, 0, 0);
    if constexpr (BMT_FWD32_XP & 1) {      // (both buffers hold data: the stages alternate between them)
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_X_DMA_K_(j, 0, 1);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_X_DMA_V_(j, 0, 1);
    }
    bf16x8 qf[KS];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = ldfrag(p.Qh + qo + 16 * ks, qok);
    }
    for (int i = tid; i < ntile * BC; i += NT) {
        uint8_t m = 0;
        if (i < p.Sk) m = (p.mask != nullptr) ? (uint8_t)(p.mask[(int64_t)b * p.mask_bs + i] != 0) : (uint8_t)1;
        sMask[i] = m;
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;       // m_run in log2 units, identical in the two lanes of a query; l_run: THIS lane's 16 keys per stage
    const float sc2 = p.scale * LOG2E;
    constexpr float TAU2 = RESCALE_TAU * LOG2E;

    // fragment addresses (LDS bytes); the k-step / d-tile enters by XOR on bits the lane part leaves free
    const int s15 = l31 & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t kA0 = lds0 + l31 * ROWB + 32 * (s15 >> 1) + 16 * (hh ^ (s15 & 1));
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t vL0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 8 * mr;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int tn = min(t + 1, ntile - 1);             // the last stage re-fetches itself into the idle buffer (branch-free)
        const int key0 = t * BC;
        uint32_t mw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mw[i] = *reinterpret_cast<const uint32_t*>(sMask + key0 + 8 * i + 4 * hh);
        const bool none_valid = __all((mw[0] | mw[1] | mw[2] | mw[3]) == 0u);
        const bool all_valid = __all((mw[0] & mw[1] & mw[2] & mw[3]) == 0x01010101u);
        if (none_valid || !wave_on || (BMT_FWD32_XP & 2)) {                     // nothing to compute: only move the next tile
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_X_DMA_K(j, tn, cur ^ 1);
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_X_DMA_V(j, tn, cur ^ 1);
        } else {
            const uint32_t kA = kA0 + cur * STAGE, vL = vL0 + cur * STAGE + TILE;
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            // ---- S^T = K . Q^T: fragments two ahead of the MFMA that uses them; the next tile's DMA requests ride on the first MFMAs
            u32x4 kf[3];
            kf[0] = lds_b128<0>(kA);
            kf[1] = lds_b128<0>(kA ^ (1 << 5));
            if constexpr (DMAV == 2) {
#pragma unroll
                for (int j = 0; j < PPW; ++j) BMT_X_DMA_K(j, tn, cur ^ 1);
#pragma unroll
                for (int j = 0; j < PPW; ++j) BMT_X_DMA_V(j, tn, cur ^ 1);
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#define BMT_X_SSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) kf[((ks_) + 2) % 3] = lds_b128<0>(kA ^ (((ks_) + 2) << 5));      \
        if constexpr (DMAV == 0) {                                                                     \
            if constexpr ((ks_) < PPW) BMT_X_DMA_K((ks_) % PPW, tn, cur ^ 1);                          \
            else if constexpr ((ks_) < 2 * PPW) BMT_X_DMA_V((ks_) % PPW, tn, cur ^ 1);                 \
        } else if constexpr (DMAV == 1) {                                                              \
            if constexpr ((ks_) % 2 == 0 && (ks_) / 2 < PPW) BMT_X_DMA_K(((ks_) / 2) % PPW, tn, cur ^ 1);             \
            else if constexpr ((ks_) % 2 == 0 && (ks_) / 2 < 2 * PPW) BMT_X_DMA_V(((ks_) / 2) % PPW, tn, cur ^ 1);   \
        } else if constexpr (DMAV == 3) {                                                              \
            if constexpr ((ks_) % 4 == 0 && (ks_) / 4 < PPW) BMT_X_DMA_K(((ks_) / 4) % PPW, tn, cur ^ 1);             \
        }                                                                                              \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(kf[(ks_) % 3]);                             \
        st = mfma32t<F16>(as_bf16x8(kf[(ks_) % 3]), qf[(ks_)], st);                                    \
    }
            BMT_X_REP16(BMT_X_SSTEP)
#undef BMT_X_SSTEP
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            // ---- softmax in the log2 domain on the raw scores: register 4 i + j = key key0 + 8 i + 4 hh + j
            if (!all_valid) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) st[4 * i + j] = ((mw[i] >> (8 * j)) & 0xffu) ? st[4 * i + j] : NEG_INF;
            }
            float tmax = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
#pragma unroll
            for (int i = 4; i < 16; i += 4) tmax = fmaxf(tmax, fmaxf(fmaxf(st[i], st[i + 1]), fmaxf(st[i + 2], st[i + 3])));
            tmax = half_max(tmax) * sc2;                   // sc2 > 0: the maximum of the scaled scores
            if (__any(tmax > m_run + TAU2)) {              // stale-reference online softmax (attn_fwd_bf16_kernel): exact
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == NEG_INF) ? 0.f : m_new));
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
                m_run = m_new;
            }
            const float m_use = (m_run == NEG_INF) ? 0.f : m_run;
            float x[16];
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], sc2, -m_use));
                psum += x[i];
            }
            l_run += psum;
            bf16x8 pf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 pw;
                pw[0] = pack_2<F16>(x[8 * kk + 0], x[8 * kk + 1]); pw[1] = pack_2<F16>(x[8 * kk + 2], x[8 * kk + 3]);
                pw[2] = pack_2<F16>(x[8 * kk + 4], x[8 * kk + 5]); pw[3] = pack_2<F16>(x[8 * kk + 6], x[8 * kk + 7]);
                pf[kk] = as_bf16x8(pw);
            }
            // ---- O^T += V^T . P^T: MFMA n = 2 dt + kk; fragment n = rows 16 kk + 4 hh .. (first read) and 16 kk + 8 + 4 hh .. (second)
            u32x2 va[3], vb[3];
#define BMT_X_VFRAG(n_)                                                                   \
    do {                                                                                  \
        const uint32_t a_ = vL ^ (((n_) >> 1) << 6);                                      \
        va[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1)) * ROWB>(a_);                          \
        vb[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1) + 8) * ROWB>(a_);                      \
    } while (0)
            BMT_X_VFRAG(0);
            BMT_X_VFRAG(1);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#define BMT_X_VSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + 2 < 2 * DT) BMT_X_VFRAG((n_) + 2);                                        \
        if constexpr (DMAV == 3 && (n_) % 2 == 0 && (n_) / 2 < PPW) BMT_X_DMA_V(((n_) / 2) % PPW, tn, cur ^ 1); \
        lgkm_wait<((n_) + 2 < 2 * DT) ? 4 : 2 * (2 * DT - 1 - (n_))>(va[(n_) % 3], vb[(n_) % 3]);      \
        const u32x4 av = {va[(n_) % 3][0], va[(n_) % 3][1], vb[(n_) % 3][0], vb[(n_) % 3][1]};         \
        o[(n_) >> 1] = mfma32t<F16>(as_bf16x8(av), pf[(n_) & 1], o[(n_) >> 1]);                        \
    }
            BMT_X_REP16(BMT_X_VSTEP)
#undef BMT_X_VSTEP
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
#undef BMT_X_VFRAG
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile has landed (this wave's share); the barrier publishes all shares
        __syncthreads();                                       // and says every wave is done reading tile t (tile t + 2 overwrites it)
    }
#undef BMT_X_DMA_K
#undef BMT_X_DMA_V
#undef BMT_X_DMA_K_
#undef BMT_X_DMA_V_

    // ---- epilogue.  Lane (l31, hh) holds O^T[d = 32 dt + 8 i + 4 hh + j][q] in register 4 i + j: the two lanes of a query own alternate
    // 4-column groups.  For the 16-bit planes one v_permlane32_swap per dword regroups a PAIR of groups (i = 2 ip, 2 ip + 1) so that the
    // lower lane holds columns 8 i .. 8 i + 7 of the first and the upper lane those of the second: 16-byte stores instead of 8-byte ones
    // (cdna_hip_programming.md T21).  Both lanes of a query are active or inactive together (same q).
    const float l_tot = half_sum(l_run);
    const float inv = 1.f / l_tot;   // fully masked row: 0 * inf = NaN, as the reference's softmax
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    const int64_t rowoff = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK;
    const int64_t po = (int64_t)b * p.bsop + (int64_t)min(q, p.Sq - 1) * p.ldop + h * DK;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            float v[2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = dt * 32 + 8 * (2 * ip + e) + 4 * hh + j;
                    v[e][j] = drop_apply(dc, o[dt][4 * (2 * ip + e) + j] * inv, (uint64_t)(p.drop_off + rowoff + d));
                }
            if (p.Ow && qok) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    *reinterpret_cast<float4*>(p.Ow + rowoff + dt * 32 + 8 * (2 * ip + e) + 4 * hh) = make_float4(v[e][0], v[e][1], v[e][2], v[e][3]);
            }
            if (p.Owh) {
                const int col = dt * 32 + 8 * (2 * ip + hh);
                uint32_t ha[2], hb[2], la[2], lb[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    split_bf2(v[e][0], v[e][1], ha[e], la[e]);
                    split_bf2(v[e][2], v[e][3], hb[e], lb[e]);
                    if (p.ow_f16) { la[e] = pack_h2(v[e][0], v[e][1]); lb[e] = pack_h2(v[e][2], v[e][3]); }
                }
                swap32u(ha[0], ha[1]);
                swap32u(hb[0], hb[1]);
                if (qok) *reinterpret_cast<u32x4*>(p.Owh + po + col) = u32x4{ha[0], hb[0], ha[1], hb[1]};
                if (p.Owl) {
                    swap32u(la[0], la[1]);
                    swap32u(lb[0], lb[1]);
                    if (qok) *reinterpret_cast<u32x4*>(p.Owl + po + col) = u32x4{la[0], lb[0], la[1], lb[1]};
                }
            }
        }
    if (qok && hh == 0) p.lsew[((int64_t)b * p.H + h) * p.SqP + q] = m_run * LN2 + __logf(l_tot);
}


// =================================================================================== backward
// delta[b,h,q] = (1-p) * sum_d dO[b,q,h*DK+d] * O[b,q,h*DK+d]  (fp32 inputs), and the bf16 plane of dO for the MFMAs
__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(const AttnPB pin, int DK, uint16_t* dOh) {
    AttnPB p = pin;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    const int64_t total = (int64_t)p.B * p.H * p.Sq;
    if (row >= total) return;
    const int q = (int)(row % p.Sq);
    const int bh = (int)(row / p.Sq);
    const int b = bh / p.H, h = bh % p.H;
    attn_rebase(p, b);                       // packed rows (delta stays indexed by the padded position: row)
    if (q >= p.Sq) return;
    dOh += p.drop_off;                       // (= this sample's first row x ldo: the same offset as the fp32 output's)
    const int64_t off = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK;
    const int64_t poff = (int64_t)b * p.bsop + (int64_t)min(q, p.Sq - 1) * p.ldop + h * DK;
    float s = 0.f;
    for (int d = lane * 4; d < DK; d += 256) {
        float4 a;
        if (p.dO) {
            a = *reinterpret_cast<const float4*>(p.dO + off + d);
        } else {          // dO arrives as the bf16 plane the MFMAs read (written by the out-projection's dX epilogue)
            const u32x2 gh = *reinterpret_cast<const u32x2*>(dOh + off + d);
            a.x = __uint_as_float(gh[0] << 16); a.y = __uint_as_float(gh[0] & 0xffff0000u);
            a.z = __uint_as_float(gh[1] << 16); a.w = __uint_as_float(gh[1] & 0xffff0000u);
        }
        float4 c;
        if (p.O) {
            c = *reinterpret_cast<const float4*>(p.O + off + d);
        } else if (p.Opf) {
            const u32x2 hh = *reinterpret_cast<const u32x2*>(p.Opf + poff + d);
            c.x = h_bits2f(hh[0] & 0xffffu); c.y = h_bits2f(hh[0] >> 16);
            c.z = h_bits2f(hh[1] & 0xffffu); c.w = h_bits2f(hh[1] >> 16);
        } else {
            const u32x2 hh = *reinterpret_cast<const u32x2*>(p.Oph + poff + d);
            u32x2 ll; ll[0] = 0u; ll[1] = 0u;
            if (p.Opl) ll = *reinterpret_cast<const u32x2*>(p.Opl + poff + d);
            c.x = __uint_as_float(hh[0] << 16) + __uint_as_float(ll[0] << 16);
            c.y = __uint_as_float(hh[0] & 0xffff0000u) + __uint_as_float(ll[0] & 0xffff0000u);
            c.z = __uint_as_float(hh[1] << 16) + __uint_as_float(ll[1] << 16);
            c.w = __uint_as_float(hh[1] & 0xffff0000u) + __uint_as_float(ll[1] & 0xffff0000u);
        }
        s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        if (p.dO) *reinterpret_cast<uint2*>(dOh + off + d) = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
    }
    s = wave_sum(s);
    if (lane == 0) p.delta[row] = s * (1.f - p.drop_p);
}

template <int DK>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_bf16_kernel(const AttnPB p) {
    constexpr int BC = 32, DT = DK / 32;
    constexpr int TB = BC * DK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* sK = reinterpret_cast<u32x4*>(smem);
    u32x4* sV = reinterpret_cast<u32x4*>(smem + TB);
    u32x2* sKt = reinterpret_cast<u32x2*>(smem + 2 * TB);
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + 3 * TB);
    int* sFlag = reinterpret_cast<int*>(sMask + 64);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;
    const int64_t koff = (int64_t)b * p.bsk + h * p.kvh, voff = (int64_t)b * p.bsv + h * p.kvh;

    // Q fragments stay in registers; the dO rows of this workgroup's 128 queries live in LDS (swizzled like K) and are read
    // as B operands.  Holding both in registers needs dq (128) + Q (64) + dO (64) + S, dP (32) = 288 accumulator-file
    // registers: more than the 256 AGPRs, and hipcc then moved all of dq out and back every key tile (288 v_accvgpr moves per
    // 48 MFMAs).
    u32x4* sdOq = reinterpret_cast<u32x4*>(smem + 3 * TB + 256);
    bf16x8 qf[DK / 16];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * half;
#pragma unroll
        for (int s = 0; s < DK / 16; ++s) qf[s] = ldfrag(p.Qh + qo + 16 * s, qok);
        u32x4 tmp[rows_n<DK, 128>()];
        tile_gload<DK, 128>(p.dOh + (int64_t)b * p.bso + h * DK, p.ldo, qt * 128, p.Sq, tid, tmp);
        tile_lstore<DK, 128>(sdOq, tid, tmp);
    }
    const int dorow = wid * 32 + l31;
    const int64_t stat = ((int64_t)b * p.H + h) * p.Sq + q;
    const float lse = qok ? p.lse[stat] : 0.f;
    const float delta = qok ? p.delta[stat] : 0.f;
    float rs = 0.f;                    // sum over keys of the bf16-rounded dS of this lane's query

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    u32x4 kv[rows_n<DK, BC>()], vv[rows_n<DK, BC>()];
    u32x2 ktv[rowsT_n<DK, BC>() * 4];
    const int ntile = (p.Sk + BC - 1) / BC;
#define BMT_DQ_FETCH(key0_)                                                         \
    do {                                                                            \
        tile_gload<DK, BC>(p.Kh + koff, p.ldk, (key0_), p.Sk, tid, kv);             \
        tile_gload<DK, BC>(p.Vh + voff, p.ldv, (key0_), p.Sk, tid, vv);             \
        tileT_gload<DK, BC>(p.Kh + koff, p.ldk, (key0_), p.Sk, tid, ktv);           \
    } while (0)
#define BMT_DQ_STORE(key0_)                                                         \
    do {                                                                            \
        tile_lstore<DK, BC>(sK, tid, kv);                                           \
        tile_lstore<DK, BC>(sV, tid, vv);                                           \
        tileT_lstore<DK, BC>(sKt, tid, ktv);                                        \
        stage_mask<BC>(p, b, (key0_), tid, sMask, sFlag);                           \
    } while (0)
    BMT_DQ_FETCH(0);
    BMT_DQ_STORE(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * BC;
        const int kn = min(key0 + BC, (ntile - 1) * BC);
        BMT_DQ_FETCH(kn);
        const int flag = sFlag[0];
        if (flag != 0) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < DK / 16; ++s) {
                const int idx = kslot<DK>(l31, 2 * s + half);
                st = mfma32(as_bf16x8(sK[idx]), qf[s], st);
                dp = mfma32(as_bf16x8(sV[idx]), as_bf16x8(sdOq[kslot<DK>(dorow, 2 * s + half)]), dp);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = qok ? __expf(st[r] * p.scale - lse) : 0.f;
                ds[r] = pr * (dp[r] - delta) * p.scale;
            }
            if (flag != 2) {
                if (p.mask != nullptr && p.mask_qs != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + acc_row(r, half);
                        const bool ok = qok && key < p.Sk && p.mask[(int64_t)b * p.mask_bs + (int64_t)q * p.mask_qs + key] != 0;
                        ds[r] = ok ? ds[r] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t mw = *reinterpret_cast<const uint32_t*>(sMask + 8 * g + 4 * half);
#pragma unroll
                        for (int c = 0; c < 4; ++c) ds[4 * g + c] = ((mw >> (8 * c)) & 0xffu) ? ds[4 * g + c] : 0.f;
                    }
                }
            }
            bf16x8 dsf[2], unused;
            pack_p<1>(ds, 0, dsf[0], unused);
            pack_p<1>(ds, 1, dsf[1], unused);
#pragma unroll
            for (int j = 0; j < 8; ++j) rs += (float)dsf[0][j] + (float)dsf[1][j];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    dq[dt] = mfma32(tfrag<BC>(sKt, dt * 32 + l31, 4 * s2 + half), dsf[s2], dq[dt]);
        }
        __syncthreads();
        BMT_DQ_STORE(kn);
        __syncthreads();
    }
#undef BMT_DQ_FETCH
#undef BMT_DQ_STORE
    if (p.kmean != nullptr) {          // the query's keys are split over the lanes l31 and l31 + 32
        rs += __shfl_xor(rs, 32, 64);
        const float* km = p.kmean + (p.kvh != 0 ? ((int64_t)b * p.H + h) : (int64_t)b) * DK      /* (shared keys: ONE mean key per sample) */;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] -= rs * km[dt * 32 + acc_row(r, half)];
    }
    grad_store_rows<DK>(p.gq, dq, b, h, q, qok, half);
    if (p.gq.hiT || p.gq.bsum) {       // uniform; the loop ended on a barrier, so the stage images are free
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
        grad_tile_write<DK, 128>(tile, dq, wid * 32, qok, l31, half);
        __syncthreads();
        grad_tile_flush<DK, 128>(tile, p.gq, b, h, qt * 128, p.Sq, tid);
    }
}

// dK/dV: workgroup = 64 keys, 4 waves split by role (0,1: dV of keys [0,32)/[32,64); 2,3: dK), loop over 32-query tiles
// with the next tile's four images prefetched into registers.   dkv_ld / dkv_bs: strides of the fp32 dK / dV outputs.
template <int DK>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_bf16_kernel(const AttnPB p) {
    constexpr int BQ = 32, DT = DK / 32, KB = 64;
    constexpr int TB = BQ * DK * 2;
    constexpr int KVB = KB * DK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* sK = reinterpret_cast<u32x4*>(smem);
    u32x4* sV = reinterpret_cast<u32x4*>(smem + KVB);
    u32x4* sQ = reinterpret_cast<u32x4*>(smem + 2 * KVB);
    u32x4* sdO = reinterpret_cast<u32x4*>(smem + 2 * KVB + TB);
    u32x2* sQt = reinterpret_cast<u32x2*>(smem + 2 * KVB + 2 * TB);
    u32x2* sdOt = reinterpret_cast<u32x2*>(smem + 2 * KVB + 3 * TB);
    float* sLse = reinterpret_cast<float*>(smem + 2 * KVB + 4 * TB);
    float* sDelta = sLse + BQ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int role = __builtin_amdgcn_readfirstlane(wid >> 1);
    const int kgrp = __builtin_amdgcn_readfirstlane(wid & 1);
    const int nkt = (p.Sk + KB - 1) / KB;
    const int w = xcd_remap(blockIdx.x, nkt * p.B * p.H);
    const int kt = w % nkt, bh = w / nkt;
    const int b = bh / p.H, h = bh % p.H;
    const int key = kt * KB + kgrp * 32 + l31;
    const bool kok = key < p.Sk;

    const uint16_t* Qb = p.Qh + (int64_t)b * p.bsq + h * DK;
    const uint16_t* dOb = p.dOh + (int64_t)b * p.bso + h * DK;

    bool kmask = kok;
    if (kok && p.mask != nullptr && p.mask_qs == 0) kmask = p.mask[(int64_t)b * p.mask_bs + key] != 0;
    // every key of this workgroup masked: the gradients are exactly zero (the loop is skipped, the epilogue writes zeros)
    const bool dead = __syncthreads_or(kmask ? 1 : 0) == 0;
    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    if (!dead) {
    {
        u32x4 tmp[rows_n<DK, KB>()];
        tile_gload<DK, KB>(p.Kh + (int64_t)b * p.bsk + h * p.kvh, p.ldk, kt * KB, p.Sk, tid, tmp);
        tile_lstore<DK, KB>(sK, tid, tmp);
        tile_gload<DK, KB>(p.Vh + (int64_t)b * p.bsv + h * p.kvh, p.ldv, kt * KB, p.Sk, tid, tmp);
        tile_lstore<DK, KB>(sV, tid, tmp);
    }
    const int myrow = kgrp * 32 + l31;

    u32x4 rq[rows_n<DK, BQ>()], rdo[rows_n<DK, BQ>()];
    u32x2 rqt[rowsT_n<DK, BQ>() * 4], rdot[rowsT_n<DK, BQ>() * 4];
    float rl = 0.f, rd = 0.f;
    const int ntile = (p.Sq + BQ - 1) / BQ;
#define BMT_FETCH(q0_)                                                             \
    do {                                                                           \
        tile_gload<DK, BQ>(Qb, p.ldq, (q0_), p.Sq, tid, rq);                       \
        tile_gload<DK, BQ>(dOb, p.ldo, (q0_), p.Sq, tid, rdo);                     \
        tileT_gload<DK, BQ>(Qb, p.ldq, (q0_), p.Sq, tid, rqt);                     \
        tileT_gload<DK, BQ>(dOb, p.ldo, (q0_), p.Sq, tid, rdot);                   \
        {                                                                          \
            const int qq_ = min((q0_) + (tid & (BQ - 1)), p.Sq - 1);               \
            const int64_t stat_ = ((int64_t)b * p.H + h) * p.Sq + qq_;             \
            rl = p.lse[stat_];                                                     \
            rd = p.delta[stat_];                                                   \
        }                                                                          \
    } while (0)
#define BMT_DKV_STORE()                                                              \
    do {                                                                            \
        tile_lstore<DK, BQ>(sQ, tid, rq);                                           \
        tile_lstore<DK, BQ>(sdO, tid, rdo);                                         \
        tileT_lstore<DK, BQ>(sQt, tid, rqt);                                        \
        tileT_lstore<DK, BQ>(sdOt, tid, rdot);                                      \
        if (tid < BQ) { sLse[tid] = rl; sDelta[tid] = rd; }                         \
    } while (0)
    BMT_FETCH(0);
    BMT_DKV_STORE();
    __syncthreads();   // also makes the K/V images visible
    for (int t = 0; t < ntile; ++t) {
        const int q0 = t * BQ;
        BMT_FETCH(min(q0 + BQ, (ntile - 1) * BQ));
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
            if (role == 1) pr[r] = pr[r] * (dp[r] - sDelta[ql_]) * p.scale;
        }
        bf16x8 bf[2], unused;
        pack_p<1>(pr, 0, bf[0], unused);
        pack_p<1>(pr, 1, bf[1], unused);
        const u32x2* timg = (role == 1) ? sQt : sdOt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                acc[dt] = mfma32(tfrag<BQ>(timg, dt * 32 + l31, 4 * s2 + half), bf[s2], acc[dt]);
        __syncthreads();   // tile t fully consumed
        BMT_DKV_STORE();
        __syncthreads();
    }
#undef BMT_FETCH
#undef BMT_DKV_STORE
    }   // !dead
    const GradOut& g = (role == 1) ? p.gk : p.gv;
    grad_store_rows<DK>(g, acc, b, h, key, kok, half);
    if (p.gk.hiT || p.gk.bsum || p.gv.hiT || p.gv.bsum) {
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);       // [2 roles][DK][64 + 8]
        grad_tile_write<DK, KB>(tile + role * DK * (KB + 8), acc, kgrp * 32, kok, l31, half);
        __syncthreads();
        grad_tile_flush<DK, KB>(tile, p.gv, b, h, kt * KB, p.Sk, tid);
        grad_tile_flush<DK, KB>(tile + DK * (KB + 8), p.gk, b, h, kt * KB, p.Sk, tid);
    }
}

// ---- gradient epilogue for the 16x16 accumulator layout: acc[dt][r] = G^T[d = 16 dt + 4 g + r][token = this lane's c column]
template <int DK>
__device__ __forceinline__ void grad_store_rows16(const GradOut& gr, const f32x4v (&acc)[DK / 16], int b, int h, int tok, bool ok, int g) {
    if (!ok) return;
    if (gr.f32) {
        float* dst = gr.f32 + (int64_t)b * gr.f_bs + (int64_t)tok * gr.f_ld + h * DK + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt)
            *reinterpret_cast<float4*>(dst + dt * 16) = make_float4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
    }
    if (gr.hi) {
        uint16_t* dst = gr.hi + (int64_t)b * gr.h_bs + (int64_t)tok * gr.h_ld + h * DK + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            u32x2 v;
            v[0] = pack_bf2(acc[dt][0], acc[dt][1]);
            v[1] = pack_bf2(acc[dt][2], acc[dt][3]);
            *reinterpret_cast<u32x2*>(dst + dt * 16) = v;
        }
    }
}
template <int DK, int NTOK>
__device__ __forceinline__ void grad_tile_write16(uint16_t* tile, const f32x4v (&acc)[DK / 16], int tc, bool ok, int c, int g) {
    constexpr int TS = NTOK + 8;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const __bf16 hv = (__bf16)acc[dt][r];
            tile[(dt * 16 + 4 * g + r) * TS + tc + c] = ok ? __builtin_bit_cast(uint16_t, hv) : (uint16_t)0;
        }
}

// dQ, 8 waves x 16 queries (see attn_fwd16_kernel for the decomposition and the key permutation of the second product)
// 16-wide dQ kernels: the keys of a query are split over the four lanes (c, g); accumulator dq[dt][r] is d = 16 dt + 4 g + r
template <int DK>
__device__ __forceinline__ void dq_rowsum_fix(const AttnPB& p, f32x4v (&dq)[DK / 16], float rs, int b, int h, int g) {
    if (p.kmean == nullptr) return;
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    const float* km = p.kmean + (p.kvh != 0 ? ((int64_t)b * p.H + h) : (int64_t)b) * DK      /* (shared keys: ONE mean key per sample) */ + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt) {
        const float4 k4 = *reinterpret_cast<const float4*>(km + 16 * dt);
        dq[dt][0] -= rs * k4.x; dq[dt][1] -= rs * k4.y; dq[dt][2] -= rs * k4.z; dq[dt][3] -= rs * k4.w;
    }
}

// dQ with 64-key stages: the forward's lean loop (attn_fwd64_kernel) applied to attn_bwd_dq16_kernel -- K / V tiles by buffer loads
// with the stage offset in an SGPR and rows past Sk read as zero, double-buffered LDS images and ONE barrier per stage, the
// probabilities recomputed in the log2 domain (p = exp2(fma(s, scale log2 e, -lse log2 e))), per-stage bookkeeping paid half as often.
template <int DK, int BC>
__device__ __forceinline__ void attn_bwd_dq16b_body(const AttnPB& pin, const int bid) {
    AttnPB p = pin;
    constexpr int KT = BC / 16, NT = 512, KS = DK / 32, DT = DK / 16, RS = pad_rs<DK>();
    constexpr int TILE = BC * RS, STAGE = 2 * TILE;
    constexpr int NR = rows_n<DK, BC, NT>();
    constexpr int SPR = DK / 8;
    static_assert(BC * SPR % NT == 0, "whole slots per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* sMask = reinterpret_cast<uint8_t*>(smem + 2 * STAGE);       // [2][BC]
    int* sFlag = reinterpret_cast<int*>(sMask + 2 * BC);                    // [2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(bid, nqt * p.B * p.H);
    const int qt = w % nqt, bhw = w / nqt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (qt * 128 >= p.Sq) {                  // a query tile past the sample's length: a zero row of bias partials, nothing else
        if (p.gq.bpart != nullptr)
            for (int d = tid; d < DK; d += NT) p.gq.bpart[((int64_t)b * nqt + qt) * p.gq.bp_ld + h * DK + d] = 0.f;
        return;
    }
    const int q = qt * 128 + wid * 16 + c;
    const bool qok = q < p.Sq;
    const bool wave_on = qt * 128 + wid * 16 < p.Sq;      // see attn_fwd64_kernel

    bf16x8 qf[KS], dof[KS];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * g;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = ldfrag(p.Qh + qo + 32 * ks, qok);
            if (p.qkv_f16) qf[ks] = h8_to_b8(qf[ks]);
            dof[ks] = ldfrag(p.dOh + oo + 32 * ks, qok);
        }
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.SqP + q;
    const float lse2 = qok ? p.lse[stat] * LOG2E : 0.f;
    float delta;
    if (p.fuse_delta) {      // see attn_bwd_dq16_kernel
        const int64_t po = (int64_t)b * p.bsop + (int64_t)min(q, p.Sq - 1) * p.ldop + h * DK + 8 * g;
        float acc = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (p.Opf) {
                const f16x8 of = __builtin_bit_cast(f16x8, ldfrag(p.Opf + po + 32 * ks, qok));
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)dof[ks][j] * (float)of[j];
            } else {
                const bf16x8 oh = ldfrag(p.Oph + po + 32 * ks, qok);
                bf16x8 ol = oh;
                if (p.Opl) ol = ldfrag(p.Opl + po + 32 * ks, qok);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float ov = (float)oh[j];
                    if (p.Opl) ov += (float)ol[j];
                    acc += (float)dof[ks][j] * ov;
                }
            }
        }
        acc = xlane_sum(acc);
        delta = acc * (1.f - p.drop_p);
        if (qok && g == 0) p.delta[stat] = delta;
    } else {
        delta = qok ? p.delta[stat] : 0.f;
    }
    const float sc2 = p.scale * LOG2E;
    const int troff = tr_lane_off(RS, c, g);
    float rs = 0.f;

    f32x4v dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * p.kvh), 0, plane_extent(p.Sk, p.ldk, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * p.kvh), 0, plane_extent(p.Sk, p.ldv, DK), 0x00020000);
    int kvo[NR], vvo[NR], lso[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int s_ = tid + NT * i;
        kvo[i] = (s_ / SPR) * (int)p.ldk * 2 + (s_ % SPR) * 16;
        vvo[i] = (s_ / SPR) * (int)p.ldv * 2 + (s_ % SPR) * 16;
        lso[i] = (s_ / SPR) * RS + (s_ % SPR) * 16;
    }
    const int ntile = (p.Sk + BC - 1) / BC;
    u32x4 kr[NR], vr[NR];
#define BMT_DQ64_FETCH(t_)                                                                             \
    do {                                                                                               \
        const int so_k = (t_) * BC * (int)p.ldk * 2, so_v = (t_) * BC * (int)p.ldv * 2;                \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo[i], so_k, 0); \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo[i], so_v, 0); \
    } while (0)
#define BMT_DQ64_STORE(t_, buf_)                                                                       \
    do {                                                                                               \
        char* sk_ = smem + (buf_) * STAGE;                                                             \
        if (p.qkv_f16) {                                                                               \
            _Pragma("unroll") for (int i = 0; i < NR; ++i) { kr[i] = h8_to_b8(kr[i]); vr[i] = h8_to_b8(vr[i]); }  \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) *reinterpret_cast<u32x4*>(sk_ + lso[i]) = kr[i];        \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) *reinterpret_cast<u32x4*>(sk_ + TILE + lso[i]) = vr[i]; \
        stage_mask<BC>(p, b, (t_) * BC, tid, sMask + (buf_) * BC, sFlag + (buf_));                      \
    } while (0)
    BMT_DQ64_FETCH(0);
    BMT_DQ64_STORE(0, 0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int tn = min(t + 1, ntile - 1);
        BMT_DQ64_FETCH(tn);
        const int flag = sFlag[cur];
        if (flag != 0 && wave_on) {
            const char* sK = smem + cur * STAGE;
            const char* sV = sK + TILE;
            const int key0 = t * BC;
            f32x4v st[KT], dp[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) { st[kt] = f32x4v{0.f, 0.f, 0.f, 0.f}; dp[kt] = st[kt]; }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    st[kt] = mfma16(rowfrag_pad<DK>(sK, kt * 16 + c, 4 * ks + g), qf[ks], st[kt]);
                    dp[kt] = mfma16(rowfrag_pad<DK>(sV, kt * 16 + c, 4 * ks + g), dof[ks], dp[kt]);
                }
            __builtin_amdgcn_s_setprio(0);
            float ds[4 * KT];
#pragma unroll
            for (int i = 0; i < 4 * KT; ++i) {
                const float pr = qok ? __builtin_amdgcn_exp2f(st[i >> 2][i & 3] * sc2 - lse2) : 0.f;
                ds[i] = pr * (dp[i >> 2][i & 3] - delta) * p.scale;
            }
            if (flag != 2) {
                if (p.mask != nullptr && p.mask_qs != 0) {
#pragma unroll
                    for (int i = 0; i < 4 * KT; ++i) {
                        const int key = key0 + 16 * (i >> 2) + 4 * g + (i & 3);
                        const bool ok = qok && key < p.Sk && p.mask[(int64_t)b * p.mask_bs + (int64_t)q * p.mask_qs + key] != 0;
                        ds[i] = ok ? ds[i] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const uint32_t mw = *reinterpret_cast<const uint32_t*>(sMask + cur * BC + 16 * kt + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[4 * kt + r] = ((mw >> (8 * r)) & 0xffu) ? ds[4 * kt + r] : 0.f;
                    }
                }
            }
            bf16x8 dsf[BC / 32];
#pragma unroll
            for (int hf = 0; hf < BC / 32; ++hf) {
                u32x4 dw;
                dw[0] = pack_bf2(ds[8 * hf + 0], ds[8 * hf + 1]); dw[1] = pack_bf2(ds[8 * hf + 2], ds[8 * hf + 3]);
                dw[2] = pack_bf2(ds[8 * hf + 4], ds[8 * hf + 5]); dw[3] = pack_bf2(ds[8 * hf + 6], ds[8 * hf + 7]);
#pragma unroll
                for (int j = 0; j < 4; ++j) rs += __uint_as_float(dw[j] << 16) + __uint_as_float(dw[j] & 0xFFFF0000u);
                dsf[hf] = as_bf16x8(dw);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dq[dt] = mfma16(trfrag<DK>(sK + troff, dt), dsf[0], dq[dt]);
                if constexpr (BC == 64) dq[dt] = mfma16(trfrag<DK>(sK + troff + 32 * RS, dt), dsf[BC / 32 - 1], dq[dt]);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        BMT_DQ64_STORE(tn, cur ^ 1);
        __syncthreads();
    }
#undef BMT_DQ64_FETCH
#undef BMT_DQ64_STORE
    dq_rowsum_fix<DK>(p, dq, rs, b, h, g);
    // (plane / fp32 / bias sums through the row-major LDS image: whole rows out, see grad_rm_write; the transposed plane keeps the old image)
    __syncthreads();
    grad_rm_epilogue16<DK, NT>(smem, p.gq, dq, b, h, qt * 128, wid * 16 + c, qok, g, p.Sq, p.SqP, tid);
    if (p.gq.hiT) {
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
        grad_tile_write16<DK, 128>(tile, dq, wid * 16, qok, c, g);
        __syncthreads();
        GradOut gt = p.gq;
        gt.bsum = nullptr;
        grad_tile_flush<DK, 128, NT>(tile, gt, b, h, qt * 128, p.Sq, tid);
    }
}
// dK / dV, 8 waves x 16 keys (128 keys per workgroup), loop over 32-query stages.  Every wave computes S[q][key] = Q . K^T and
// dP[q][key] = dO . V^T once (A = staged Q / dO rows, B = this wave's K / V rows held in registers) and accumulates BOTH
// dV^T += dO^T . P and dK^T += Q^T . dS for its keys (an earlier version split the waves into a dV and a dK role: S was
// computed twice and the Q / dO fragments were read by twice as many waves).  The lane's 8 probabilities are queries
// {4g..4g+3} and {16+4g..} of the stage -> B operand of the second products with the same permutation of the reduction index
// as in the forward kernel; the A operands come through the transpose unit from the same padded Q / dO images.
// dK / dV with the lean stage loop: Q / dO tiles by buffer loads (stage offset in an SGPR, rows past Sq read as zero), double-buffered
// LDS images with ONE barrier per 32-query stage, probabilities recomputed in the log2 domain.  Decomposition and fragment layouts as
// in attn_bwd_dkv16_kernel (which documents them).
// QMASK: the mask has a row per query (the decoder's causal mask); the key-padding masks of the encoder and of every cross-attention
// are per key (kmask), and without the per-element mask addressing the d_k = 256 kernel fits its 256 registers (no spills).
template <int DK, bool QMASK>
__device__ __forceinline__ void attn_bwd_dkv32_body(const AttnPB& pin, const int bid) {
    AttnPB p = pin;
    constexpr int BQ = 32, NT = 512, KS = DK / 32, DT = DK / 16, KBLK = 128, RS = pad_rs<DK>();
    constexpr int TP = BQ * RS, STAGE = 2 * TP;
    constexpr int NR = rows_n<DK, BQ, NT>();
    constexpr int SPR = DK / 8;
    static_assert(BQ * SPR % NT == 0, "whole slots per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sStat = reinterpret_cast<float*>(smem + 2 * STAGE);       // [2 buffers][lse log2 e | delta][32]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int nkt = (p.Sk + KBLK - 1) / KBLK;
    const int w = xcd_remap(bid, nkt * p.B * p.H);
    const int kt = w % nkt, bhw = w / nkt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (kt * KBLK >= p.Sk) {                 // a key block past the sample's length: zero rows of bias partials, nothing else
        for (int d = tid; d < DK; d += NT) {
            if (p.gk.bpart != nullptr) p.gk.bpart[((int64_t)b * nkt + kt) * p.gk.bp_ld + h * DK + d] = 0.f;
            if (p.gv.bpart != nullptr) p.gv.bpart[((int64_t)b * nkt + kt) * p.gv.bp_ld + h * DK + d] = 0.f;
        }
        return;
    }
    const int key = kt * KBLK + wid * 16 + c;
    const bool kok = key < p.Sk;
    const int troff = tr_lane_off(RS, c, g);

    bool kmask = kok;
    if (!QMASK && kok && p.mask != nullptr) kmask = p.mask[(int64_t)b * p.mask_bs + key] != 0;
    const bool dead = __syncthreads_or(kmask ? 1 : 0) == 0;   // every key of the workgroup masked: gradients exactly zero
    f32x4v accv[DT], acck[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { accv[dt] = f32x4v{0.f, 0.f, 0.f, 0.f}; acck[dt] = accv[dt]; }
    if (!dead) {
        bf16x8 kf[KS], vf[KS];
        {
            const int krow = min(key, p.Sk - 1);
            const int64_t ko = (int64_t)b * p.bsk + (int64_t)krow * p.ldk + h * p.kvh + 8 * g;
            const int64_t vo = (int64_t)b * p.bsv + (int64_t)krow * p.ldv + h * p.kvh + 8 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[ks] = ldfrag(p.Kh + ko + 32 * ks, true);
                vf[ks] = ldfrag(p.Vh + vo + 32 * ks, true);
                if (p.qkv_f16) { kf[ks] = h8_to_b8(kf[ks]); vf[ks] = h8_to_b8(vf[ks]); }
            }
        }
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Qh + (int64_t)b * p.bsq + h * DK), 0, plane_extent(p.Sq, p.ldq, DK), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dOh + (int64_t)b * p.bso + h * DK), 0, plane_extent(p.Sq, p.ldo, DK), 0x00020000);
        int qvo[NR], ovo[NR], lso[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int s_ = tid + NT * i;
            qvo[i] = (s_ / SPR) * (int)p.ldq * 2 + (s_ % SPR) * 16;
            ovo[i] = (s_ / SPR) * (int)p.ldo * 2 + (s_ % SPR) * 16;
            lso[i] = (s_ / SPR) * RS + (s_ % SPR) * 16;
        }
        const float sc2 = p.scale * LOG2E;
        u32x4 rq[NR], rdo[NR];
        float rl = 0.f, rd = 0.f;
        const int ntile = (p.Sq + BQ - 1) / BQ;
        const int64_t stat0 = ((int64_t)b * p.H + h) * p.SqP;
#define BMT_DKV32_FETCH(t_)                                                                            \
    do {                                                                                               \
        const int so_q = (t_) * BQ * (int)p.ldq * 2, so_o = (t_) * BQ * (int)p.ldo * 2;                \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) rq[i] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qvo[i], so_q, 0);  \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) rdo[i] = __builtin_amdgcn_raw_buffer_load_b128(rsO, ovo[i], so_o, 0); \
        if (tid < BQ) {                                                                                \
            const int qq_ = max(0, min((t_) * BQ + tid, p.Sq - 1));                                    \
            rl = p.lse[stat0 + qq_] * LOG2E;                                                           \
            rd = p.delta[stat0 + qq_];                                                                 \
        }                                                                                              \
    } while (0)
#define BMT_DKV32_STORE(buf_)                                                                          \
    do {                                                                                               \
        char* sq_ = smem + (buf_) * STAGE;                                                             \
        if (p.qkv_f16) {                                                                               \
            _Pragma("unroll") for (int i = 0; i < NR; ++i) rq[i] = h8_to_b8(rq[i]);                           \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) *reinterpret_cast<u32x4*>(sq_ + lso[i]) = rq[i];       \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) *reinterpret_cast<u32x4*>(sq_ + TP + lso[i]) = rdo[i]; \
        if (tid < BQ) { sStat[(buf_) * 64 + tid] = rl; sStat[(buf_) * 64 + 32 + tid] = rd; }           \
    } while (0)
        BMT_DKV32_FETCH(0);
        BMT_DKV32_STORE(0);
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            const int cur = t & 1;
            const int q0 = t * BQ;
            BMT_DKV32_FETCH(min(t + 1, ntile - 1));
            const char* sQ = smem + cur * STAGE;
            const char* sdO = sQ + TP;
            const float* sLse = sStat + cur * 64;
            const float* sDelta = sLse + 32;
            f32x4v sacc[2], dp[2];
            sacc[0] = f32x4v{0.f, 0.f, 0.f, 0.f};
            sacc[1] = sacc[0]; dp[0] = sacc[0]; dp[1] = sacc[0];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    sacc[qi] = mfma16(rowfrag_pad<DK>(sQ, qi * 16 + c, 4 * ks + g), kf[ks], sacc[qi]);
                    dp[qi] = mfma16(rowfrag_pad<DK>(sdO, qi * 16 + c, 4 * ks + g), vf[ks], dp[qi]);
                }
            __builtin_amdgcn_s_setprio(0);
            float pr[8], ds[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ql_ = 16 * (i >> 2) + 4 * g + (i & 3);
                const int qq = q0 + ql_;
                bool ok = kmask && qq < p.Sq;
                if constexpr (QMASK) {
                    if (ok) ok = p.mask[(int64_t)b * p.mask_bs + (int64_t)qq * p.mask_qs + key] != 0;
                }
                pr[i] = ok ? __builtin_amdgcn_exp2f(sacc[i >> 2][i & 3] * sc2 - sLse[ql_]) : 0.f;
                ds[i] = pr[i] * (dp[i >> 2][i & 3] - sDelta[ql_]) * p.scale;
            }
            u32x4 pw, dw;
            pw[0] = pack_bf2(pr[0], pr[1]); pw[1] = pack_bf2(pr[2], pr[3]);
            pw[2] = pack_bf2(pr[4], pr[5]); pw[3] = pack_bf2(pr[6], pr[7]);
            dw[0] = pack_bf2(ds[0], ds[1]); dw[1] = pack_bf2(ds[2], ds[3]);
            dw[2] = pack_bf2(ds[4], ds[5]); dw[3] = pack_bf2(ds[6], ds[7]);
            const bf16x8 pf = as_bf16x8(pw), dsf = as_bf16x8(dw);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                accv[dt] = mfma16(trfrag<DK>(sdO + troff, dt), pf, accv[dt]);
                acck[dt] = mfma16(trfrag<DK>(sQ + troff, dt), dsf, acck[dt]);
            }
            __builtin_amdgcn_s_setprio(0);
            BMT_DKV32_STORE(cur ^ 1);          // nobody reads that image now: its readers passed the previous barrier
            __syncthreads();
        }
#undef BMT_DKV32_FETCH
#undef BMT_DKV32_STORE
    }
    // one [128][DK + 8] row-major image at a time (dV, then dK) in the stage buffers: whole rows out, bias sums from the image
    static_assert(KBLK == 128, "the row-major epilogue image holds 128 tokens");
    __syncthreads();
    grad_rm_epilogue16<DK, NT>(smem, p.gv, accv, b, h, kt * KBLK, wid * 16 + c, kok, g, p.Sk, p.SkP, tid);
    grad_rm_epilogue16<DK, NT>(smem, p.gk, acck, b, h, kt * KBLK, wid * 16 + c, kok, g, p.Sk, p.SkP, tid);
    if (p.gk.hiT || p.gv.hiT) {
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
        GradOut gt = p.gv;
        gt.bsum = nullptr;
        grad_tile_write16<DK, KBLK>(tile, accv, wid * 16, kok, c, g);
        __syncthreads();
        grad_tile_flush<DK, KBLK, NT>(tile, gt, b, h, kt * KBLK, p.Sk, tid);
        __syncthreads();
        gt = p.gk;
        gt.bsum = nullptr;
        grad_tile_write16<DK, KBLK>(tile, acck, wid * 16, kok, c, g);
        __syncthreads();
        grad_tile_flush<DK, KBLK, NT>(tile, gt, b, h, kt * KBLK, p.Sk, tid);
    }
}
// the two kernels of the two-kernel backward in ONE launch: workgroups [0, nq) are the dQ kernel's, the rest the dK / dV kernel's.  Neither
// reads what the other writes once delta comes from its own small kernel (attn_delta_bf16_kernel), and on the shapes this form serves -- the
// decoder's 30-query attentions -- the dQ kernel is B x H workgroups walking the whole key range one after the other (latency: ~50 us over
// 800 keys on half the CUs) while the dK / dV kernel streams 2 x Sk x d_k of keys and gradients (HBM): side by side they take the longer of the
// two instead of the sum.
template <int DK, bool QMASK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_pair_kernel(const AttnPB p, const int nq) {
    if ((int)blockIdx.x < nq) attn_bwd_dq16b_body<DK, 32>(p, (int)blockIdx.x);
    else attn_bwd_dkv32_body<DK, QMASK>(p, (int)blockIdx.x - nq);
}

// mean key per (batch, column) over the valid keys: K plane [B][Sk][ldk] (bf16), key-padding mask [B][Sk] (or none / a per-query
// mask: every key counts).  Block = 128 columns (16 threads x 16 bytes) x 32 key groups; every load is unconditional and independent
// (the mask byte is a multiplier) so the key loop unrolls into batches of loads in flight -- a `continue` on the mask byte made
// every iteration a dependent L2 round trip (100 us per call instead of 7).  grid (B, ceil(D / 128)).
// ks: every ks-th key is read.  The mean key is a SHIFT, not a quantity of the mathematics: rows of dS sum to zero, so dS . (K - m) = dS . K
// for any m, and what the correction needs from m is the keys' common component -- which the mean of every 8th key carries as well as the
// mean of all of them, for an eighth of the 52 MB an 800-key audio memory costs to read (31 us per launch, 12 launches per step).
__global__ __launch_bounds__(512) void attn_kmean_kernel(const uint16_t* __restrict__ Kh, int64_t ldk, int64_t bsk, const uint8_t* __restrict__ mask,
                                                         int64_t mask_bs, int Sk, int D, float* __restrict__ out, int f16, int ks,
                                                         const int* __restrict__ k_off) {
    constexpr int KG = 32;
    __shared__ float red[KG][129];
    __shared__ float cnt[KG];
    const int b = blockIdx.x, ct = threadIdx.x & 15, kg = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 128 + ct * 8;
    const bool cok = c0 < D;                       // D is a multiple of 8
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float n = 0.f;
    int64_t boff = (int64_t)b * bsk;
    if (k_off != nullptr) {                        // packed rows: this sample's keys are rows k_off[b] .. k_off[b + 1] - 1, all valid
        boff = (int64_t)k_off[b] * ldk;
        Sk = k_off[b + 1] - k_off[b];
    }
    const uint16_t* base = Kh + boff + (cok ? c0 : 0);
    const uint8_t* mb = mask ? mask + (int64_t)b * mask_bs : nullptr;
#pragma unroll 8
    for (int k = kg * ks; k < Sk; k += KG * ks) {
        const float m = mb ? (mb[k] != 0 ? 1.f : 0.f) : 1.f;
        u32x4 v = *reinterpret_cast<const u32x4*>(base + (int64_t)k * ldk);
        if (f16) v = h8_to_b8(v);          // (uniform) the fp16 plane: same values the backward kernels will multiply
        n += m;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[2 * q] += m * bf_bits2f(v[q] & 0xFFFFu);
            a[2 * q + 1] += m * bf_bits2f(v[q] >> 16);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[kg][ct * 8 + q] = a[q];
    if (ct == 0) cnt[kg] = n;
    __syncthreads();
    const int c = blockIdx.y * 128 + threadIdx.x;
    if (threadIdx.x < 128 && c < D) {
        float total = 0.f, sum = 0.f;
#pragma unroll
        for (int i = 0; i < KG; ++i) { total += cnt[i]; sum += red[i][threadIdx.x]; }
        out[(int64_t)b * D + c] = total > 0.f ? sum / total : 0.f;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int DK, bool F16, int DMAV = 0, bool PRIO = true>
int launch_fwd32(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds = 2 * 2 * 32 * DK * 2 + ((ntile * 32 + 15) & ~15);
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel<DK, F16, DMAV, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 32 * DK * 2 + 8192);
        done = true;
    }
    hipLaunchKernelGGL((attn_fwd32_kernel<DK, F16, DMAV, PRIO>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_attn_fwd_bf16");
    return BMT_OK;
}

template <int DK, int NPASS, bool F16 = false>
int launch_fwd(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    if constexpr (DK >= 128 && NPASS == 1) {
        // K / V rows are fetched with 32-bit byte offsets from the (batch, head) base
        const bool fits = ((int64_t)p.Sk * p.ldk * 2 < (1ll << 31)) && ((int64_t)p.Sk * p.ldv * 2 < (1ll << 31));
        // the 32-query kernel: key-padding masks (its mask row lives in LDS: Sk <= 8192), and two workgroups per CU to overlap --
        // with fewer it ties or loses against the 16-query kernel (V-self / V<-A of configs[1]: 0.95x / 0.98x, the decoder 0.86x)
        const bool can32 = fits && (p.mask == nullptr || p.mask_qs == 0) && p.Sk <= 8192;
        if (can32 && nblk >= 2 * bmt_device_cus()) return launch_fwd32<DK, F16>(p, st);
        if (fits) {
            const int lds = 2 * 2 * 64 * (DK * 2 + 32) + 256;
            static bool done64 = false;
            if (!done64) {
                (void)hipFuncSetAttribute((const void*)attn_fwd64_kernel<DK, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                done64 = true;
            }
            hipLaunchKernelGGL((attn_fwd64_kernel<DK, F16>), dim3(nblk), dim3(512), lds, st, p);
            BMT_CHECK_LAUNCH("bmt_attn_fwd_bf16");
            return BMT_OK;
        }
    }
    if (p.q_off || p.k_off) {
        bmt_set_error("bmt_attn_fwd_bf16: packed rows are not taken by this shape's kernel");
        return BMT_EINVAL;
    }
    if constexpr (DK >= 128) {        // 8 waves x 16 queries, two waves per SIMD
        const int lds = (NPASS == 3 ? 2 : 1) * (2 * 32 * (DK * 2 + 32)) + 128;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)attn_fwd16_kernel<DK, NPASS, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done = true;
        }
        hipLaunchKernelGGL((attn_fwd16_kernel<DK, NPASS, F16>), dim3(nblk), dim3(512), lds, st, p);
    } else {
        using G = Geo<DK>;
        const int lds = (NPASS == 3 ? 2 : 1) * (G::K_BYTES + G::V_BYTES) + 128;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<DK, NPASS, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done = true;
        }
        hipLaunchKernelGGL((attn_fwd_bf16_kernel<DK, NPASS, F16>), dim3(nblk), dim3(256), lds, st, p);
    }
    BMT_CHECK_LAUNCH("bmt_attn_fwd_bf16");
    return BMT_OK;
}


// =================================================================================== backward, split form (d_k >= 128, fp16 q / k / v planes)
// The two-kernel backward above computes S = Q K^T and dP = dO V^T in BOTH kernels: 7 products of Sq x Sk x d_k for the 5 the mathematics
// has.  Here the dQ kernel, which has P and dS in registers anyway, LEAVES them in HBM workspaces (bf16, one 128-key block of Sq rows
// after the other per (batch, head): [B*H][ceil(Sk / 128)][Sq][128]) together with a bf16 copy of its q rows (scaled by the per-query
// power of two the dS operand carries), and dK / dV are two plain products over them with no softmax arithmetic at all:
//     dV^T[d][key] += dO^T[d x q] . P[q x key]          dK^T[d][key] += Qb^T[d x q] . dS'[q x key]
// Measured on configs[1]'s audio self-attention (B 32, H 4, 800 x 800, d_k 256, ragged lengths; tools/probes/attn_bwd_split_check.py,
// profiles/r03_*_split_*): 504 -> 347 us for the whole backward (video <- audio 217 -> 147, audio <- video 220 -> 200, video self 80 -> 65),
// dQ 2.4e-3 instead of 4.7e-3 relative error (fp16 operands with per-query scales), dK / dV 3.3e-3 / 2.9e-3 instead of 4.7e-3 / 4.0e-3.
// What bounds it now (probes + PMC in DESIGN.md): a workgroup's fixed costs -- 196 KB of q / dO / O rows in, 64 KB of Qb out before the
// first MFMA, 64 KB per gradient tile out after the last -- move at one CU's ~10 B / clk, and the 512-register kernels cannot share a CU.
__device__ __forceinline__ int kswz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ float bfbits_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfbits_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

#define BMT_B_BAR()                              \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;

template <int OFF>
__device__ __forceinline__ u32x2 lds_b64(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ void st128(__amdgpu_buffer_rsrc_t rs, uint32_t a, uint32_t b, uint32_t c, uint32_t d, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{a, b, c, d}, rs, voff + OFF, soff, 0);
}


// ------------------------------------------------------------------------------------------------------------ the dQ kernel, software-pipelined
// One wave per SIMD issues in order: a stage that runs {S MFMAs} {softmax VALU} {dP MFMAs} {dS VALU} {dQ MFMAs} leaves the matrix pipe idle
// during every VALU block (attn_bwd_dq32e_kernel: 590 instructions per 48 MFMAs, 38 % of the pipe).  Here the VALU work of a tile runs under
// the MFMAs of its neighbours -- iteration t:
//     phase A: S(t+1) = K . Q^T            under  dS'(t) = P(t) (dP'(t) - delta') scale  [+ the stores of P(t)]
//     phase B: dQ'^T += K(t)^T . dS'(t)^T   under  P(t+1) = exp2(S(t+1) scale log2 e - lse)
//     phase C: dP'(t+1) = V . dO'^T         under  the stores of dS(t), address stepping
// and the per-element cost is cut: fragment addresses of the two swizzled images are lane constants per (k-step & 7) / (d-tile & 3, row
// block) -- bit 8 of the address is free, so k-step >> 3 and d-tile >> 2 are immediates -- stepped by one add per stage instead of one
// v_xor per read; V uses the K image's dual-purpose swizzle (vA = kA + TILE); the key mask is applied to the packed fp16 dS' by ANDing with a
// 16-bit-per-key mask image in LDS (8 v_and per stage instead of 16 selects + their compares: P of a masked key may be anything, its dS'
// is zeroed bit-wise, its P / dS columns in the workspace belong to keys whose gradients attn_bwd_dkvg_kernel zeroes); fully masked tiles at
// the END of the key range (prefix masks) shorten the loop for the whole workgroup (ntile_run), a fully masked tile in the middle is simply
// computed -- no wave-level skip paths (they cost hipcc 7.2 its register allocation, see attn_bwd_dkvg_kernel); the emitted dS keeps the
// per-query scale (bf16 has the range) and the bf16 copy of q carries 2^-k(q) instead: dK = sum_q (q 2^-k) . (dS 2^k) needs no extra multiply;
// delta = (1 - p) rowsum(dO * O) is computed in the prologue from the saved output plane (fuse_delta), nobody else needs it.
// XP (timing probes): bit 3 = no epilogue stores, bit 4 = no loop
// EMIT = false (round 6, the recompute form: attn_bwd_dkvr_kernel rebuilds P and dS from q / k / v / dO itself): nothing is emitted -- no P / dS
// workspace stores, no scaled copy of q -- and the kernel leaves what the key-side kernel needs instead: delta per query (AttnPB.delta) and
// the largest |dO| of the tile's rows (AttnPB.qamax, next to the live-query bits: the key side scales its fp16 dS by one power of two per
// (batch, head)).
template <int DK, int XP = 0, bool EMIT = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq32p_kernel(const AttnPB pin) {
    AttnPB p = pin;
    constexpr int BC = 32, NT = 256, KS = DK / 16, DT = DK / 32, ROWB = DK * 2, TILE = BC * ROWB, STAGE = 2 * TILE, NS = 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NP = BC / RPP, PPW = NP / 4;
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    uint16_t* sMask = reinterpret_cast<uint16_t*>(smem + NS * STAGE);    // [ntile * 32] 0xffff = valid key, 0 = masked / past Sk
    int* sLast = reinterpret_cast<int*>(smem + NS * STAGE + (((p.Sk + BC - 1) / BC) * BC * 2 + 15) / 16 * 16);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    // work order: (batch, head)-major, so that the query tiles of a (batch, head) run together and share its K / V in their XCD's L2
    const int qt = w % nqt, bhw = w / nqt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (qt * 128 >= p.Sq) {                  // a query tile past the sample's length: no live query, a zero row of bias partials, nothing else
        if (p.qlive != nullptr && tid == 0) p.qlive[(int64_t)bh * nqt + qt] = 0;
        if (p.qamax != nullptr && tid == 0) p.qamax[(int64_t)bh * nqt + qt] = 0.f;
        if (p.gq.bpart != nullptr)
            for (int d = tid; d < DK; d += NT) p.gq.bpart[((int64_t)b * nqt + qt) * p.gq.bp_ld + h * DK + d] = 0.f;
        return;
    }
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;
    const int ntile = (p.Sk + BC - 1) / BC;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * p.kvh), 0, plane_extent(p.Sk, p.ldk, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * p.kvh), 0, plane_extent(p.Sk, p.ldv, DK), 0x00020000);
    const int64_t slab = (int64_t)bh * p.ws_slab;
    // (the descriptor covers the whole (batch, head) slab: the range check takes the soffset into account -- raw buffers are out of range at
    // voffset >= num_records - soffset -- so a one-block range dropped every store to key blocks past the first; rows past Sq are kept
    // out by their voffset instead)
    const int slab_bytes = (int)(p.ws_slab * 2);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Pws + slab), 0, slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dSws + slab), 0, slab_bytes, 0x00020000);
    const int ws_tile2 = (int)p.ws_tile * 2;
    const int wvo = qok ? (int)(((int64_t)q * p.ws_pitch + 8 * hh) * 2) : 0x7fffff00;
    int kvo[4], vvo[4];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wid * PPW + j) * RPP + lane / CPR, cpos = lane % CPR;
        kvo[j] = row * (int)p.ldk * 2 + ((cpos ^ kswz(row)) * 16);
        vvo[j] = row * (int)p.ldv * 2 + ((cpos ^ kswz(row)) * 16);
    }
    const int sstep_k = BC * (int)p.ldk * 2, sstep_v = BC * (int)p.ldv * 2;
#define BMT_P_DMA_K(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + (slot_) * STAGE + (wid * PPW + (j_)) * 1024), 16, kvo[j_], (t_) * sstep_k, 0, 0)
#define BMT_P_DMA_V(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + (slot_) * STAGE + TILE + (wid * PPW + (j_)) * 1024), 16, vvo[j_], (t_) * sstep_v, 0, 0)

    // ---- prologue: tiles 0, 1, 2 in flight; mask image; q, dO (-> scale, delta), lse
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        const int tl = min(s, ntile - 1);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_P_DMA_K(j, tl, s);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_P_DMA_V(j, tl, s);
    }
    if (tid == 0) { sLast[0] = -1; sLast[1] = 0; sLast[2] = 0; }
    __syncthreads();
    {
        int last = -1;
        for (int i0 = tid; i0 < ntile * BC; i0 += 4 * NT) {      // four independent (clamped, unconditional) loads per round trip
            uint8_t mb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mb[j] = (p.mask != nullptr) ? p.mask[(int64_t)b * p.mask_bs + min(i0 + j * NT, p.Sk - 1)] : (uint8_t)1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * NT;
                const bool m = i < p.Sk && mb[j] != 0;
                if (i < ntile * BC) sMask[i] = m ? (uint16_t)0xffffu : (uint16_t)0;
                if (m) last = i;
            }
        }
        last = (int)wave_max((float)last);
        if (lane == 0 && last >= 0) atomicMax(sLast, last);
    }
    bf16x8 qf[KS];
    u32x4 dob[KS];
    float dsum = 0.f;
    {
        // UNCONDITIONAL loads from a clamped row: a guarded load into a register array makes hipcc 7.2 branch around every load and wait
        // vmcnt(0) behind it -- 48 serialized memory round trips, 66 of this kernel's 217 us (profiles/r03_j_split_probes.txt).  Rows past
        // Sq carry the last row's values: nothing of theirs is stored.
        const int qc = min(q, p.Sq - 1);
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)qc * p.ldq + h * DK + 8 * hh;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)qc * p.ldo + h * DK + 8 * hh;
        const int64_t po = (int64_t)b * p.bsop + (int64_t)qc * p.ldop + h * DK + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = as_bf16x8(*reinterpret_cast<const u32x4*>(p.Qh + qo + 16 * ks));
            dob[ks] = *reinterpret_cast<const u32x4*>(p.dOh + oo + 16 * ks);
        }
        if (p.fuse_delta) {      // (uniform branches OUTSIDE the load loops: every path is one batch of 16 loads)
            u32x4 of[KS];
            if (p.Opf) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) of[ks] = *reinterpret_cast<const u32x4*>(p.Opf + po + 16 * ks);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        dsum += bfbits_lo(dob[ks][j]) * h_bits2f(of[ks][j] & 0xffffu) + bfbits_hi(dob[ks][j]) * h_bits2f(of[ks][j] >> 16);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) of[ks] = *reinterpret_cast<const u32x4*>(p.Oph + po + 16 * ks);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dsum += bfbits_lo(dob[ks][j]) * bfbits_lo(of[ks][j]) + bfbits_hi(dob[ks][j]) * bfbits_hi(of[ks][j]);
                if (p.Opl) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) of[ks] = *reinterpret_cast<const u32x4*>(p.Opl + po + 16 * ks);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int j = 0; j < 4; ++j) dsum += bfbits_lo(dob[ks][j]) * bfbits_lo(of[ks][j]) + bfbits_hi(dob[ks][j]) * bfbits_hi(of[ks][j]);
                }
            }
        }
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.SqP + min(q, p.Sq - 1);
    const float lse2 = p.lse[stat] * LOG2E;
    const float delta = p.fuse_delta ? half_sum(dsum) * (1.f - p.drop_p) : p.delta[stat];
    float amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bfbits_lo(dob[ks][j])), fabsf(bfbits_hi(dob[ks][j]))));
    amax = half_max(amax);
    if (__ballot(qok && !(amax == 0.f)) != 0ull && lane == 0) atomicOr(&sLast[1], 1 << wid);      // this wave's 32 queries carry gradient (a NaN row counts: it must propagate)
    if constexpr (!EMIT) {
        // (non-negative floats order as their bit patterns; a NaN row makes the largest pattern: the key side's scale clamps)
        const float wmax = wave_max(qok ? amax : 0.f);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(&sLast[2]), __float_as_uint(wmax));
        if (qok && hh == 0 && p.delta != nullptr) p.delta[stat] = delta;
    }
    int kexp = 0;
    if (amax > 0.f) kexp = 6 - ((int)((__float_as_uint(amax) >> 23) & 0xffu) - 127);
    kexp = max(-60, min(60, kexp));
    const float up = __uint_as_float((uint32_t)(127 + kexp) << 23), down = __uint_as_float((uint32_t)(127 - kexp) << 23);
    bf16x8 dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const uint32_t w0 = pack_h2(bfbits_lo(dob[ks][0]) * up, bfbits_hi(dob[ks][0]) * up);
        const uint32_t w1 = pack_h2(bfbits_lo(dob[ks][1]) * up, bfbits_hi(dob[ks][1]) * up);
        const uint32_t w2 = pack_h2(bfbits_lo(dob[ks][2]) * up, bfbits_hi(dob[ks][2]) * up);
        const uint32_t w3 = pack_h2(bfbits_lo(dob[ks][3]) * up, bfbits_hi(dob[ks][3]) * up);
        dof[ks] = as_bf16x8(u32x4{w0, w1, w2, w3});
    }
    if (EMIT && qok) {        // Qb = bf16(q 2^-k(q)): the A operand of dK^T = Qb^T . dS' in attn_bwd_dkvg_kernel (dS' keeps the 2^k)
        uint16_t* qb = p.Qbws + (int64_t)b * p.bsqb + (int64_t)q * p.ldqb + h * DK + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 qv = __builtin_bit_cast(u32x4, qf[ks]);
            const uint32_t w0 = pack_bf2(h_bits2f(qv[0] & 0xffffu) * down, h_bits2f(qv[0] >> 16) * down);
            const uint32_t w1 = pack_bf2(h_bits2f(qv[1] & 0xffffu) * down, h_bits2f(qv[1] >> 16) * down);
            const uint32_t w2 = pack_bf2(h_bits2f(qv[2] & 0xffffu) * down, h_bits2f(qv[2] >> 16) * down);
            const uint32_t w3 = pack_bf2(h_bits2f(qv[3] & 0xffffu) * down, h_bits2f(qv[3] >> 16) * down);
            *reinterpret_cast<u32x4*>(qb + 16 * ks) = u32x4{w0, w1, w2, w3};
        }
    }
    const float dsc = -delta * up * p.scale;      // dS' = P ((dP' - delta') scale)
    const float sc2 = p.scale * LOG2E;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    float rs = 0.f;
    const h2_t ones = {(_Float16)1.f, (_Float16)1.f};

    // fragment addresses (LDS bytes) as lane constants
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int fk = kswz(l31);
    const uint32_t kA0 = lds0 + l31 * ROWB + 32 * (fk >> 1) + 16 * (hh ^ (fk & 1));
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t kT0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1);
    uint32_t kax[8], ktx[4][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) kax[j] = kA0 ^ (j << 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ktx[j][0] = kT0 ^ (j << 6);
        ktx[j][1] = kT0 ^ ((j << 6) | 32);
    }
    const uint32_t mA = lds0 + NS * STAGE + 8 * hh;          // mask image: keys 8 i + 4 hh .. + 3 of a tile = 8 bytes at key0 * 2 + 16 i + 8 hh

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (0 when every key is masked -- or, with p.qlive, when no query of the tile has a non-zero dO: dQ = 0 exactly, and the P / dS blocks of
    // these rows are never read: attn_bwd_dkvg8_kernel skips or wipes the stages whose live bit is clear)
    const int livemask = sLast[1];
    if (p.qlive != nullptr && tid == 0) p.qlive[(int64_t)bh * nqt + qt] = livemask;
    if (!EMIT && p.qamax != nullptr && tid == 0) p.qamax[(int64_t)bh * nqt + qt] = __uint_as_float((uint32_t)sLast[2]);
    const int ntile_run = ((XP & 16) || (p.qlive != nullptr && livemask == 0)) ? 0 : sLast[0] / BC + 1;

    // A dependent MFMA issues back to back with its predecessor or waits out its latency (MI355X_MICROARCH.md: any instruction between two
    // MFMAs on one accumulator costs ~43 cycles): consecutive MFMAs here always belong to DIFFERENT accumulators -- S and dP' steps alternate
    // (phase AC), dQ walks the d-tiles inside a 16-key step (phase B).  Fragment reads run PF steps ahead of their MFMA (one wave per SIMD:
    // nothing else hides the LDS latency).
    constexpr int PF = 4, RR = PF + 1;
#define BMT_P_ROWFRAG(i_) lds_b128<(DK == 256 ? (((i_) >> 1) >> 3) * 256 : 0) + (((i_) & 1) ? TILE : 0)>(kan[((i_) >> 1) & 7])
    float pr[16], sth[8];
    f32x16 dp;
#define BMT_P_EXPR(r_, src_) pr[(r_) & 15] = __builtin_amdgcn_exp2f(__builtin_fmaf((src_), sc2, -lse2))
    if (ntile_run > 0) {
        // ---- head: S(0) and dP'(0), the lower half of P(0); the upper half of S(0) waits in registers (as in every iteration)
        uint32_t kan[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kan[j] = kax[j];
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
        u32x4 fr[RR];
        fr[0] = BMT_P_ROWFRAG(0); fr[1] = BMT_P_ROWFRAG(1); fr[2] = BMT_P_ROWFRAG(2); fr[3] = BMT_P_ROWFRAG(3);
#define BMT_P_HSTEP(i_)                                                                                \
    if constexpr ((i_) < 2 * KS) {                                                                     \
        if constexpr ((i_) + PF < 2 * KS) fr[((i_) + PF) % RR] = BMT_P_ROWFRAG((i_) + PF);             \
        lgkm_wait<((i_) + PF < 2 * KS) ? PF : (2 * KS - 1 - (i_))>(fr[(i_) % RR]);                     \
        if constexpr (((i_) & 1) == 0) st = mfma32t<true>(as_bf16x8(fr[(i_) % RR]), qf[(i_) >> 1], st); \
        else dp = mfma32t<true>(as_bf16x8(fr[(i_) % RR]), dof[(i_) >> 1], dp);                         \
    }
#define BMT_P_HSTEP2(j_) BMT_P_HSTEP(2 * (j_)) BMT_P_HSTEP(2 * (j_) + 1)
        BMT_X_REP16(BMT_P_HSTEP2)
#undef BMT_P_HSTEP2
#undef BMT_P_HSTEP
#pragma unroll
        for (int r = 0; r < 8; ++r) { BMT_P_EXPR(r, st[r]); sth[r] = st[8 + r]; }
    }

    for (int t = 0; t < ntile_run; ++t) {
        const int slot = t % NS, slotn = (t + NS - 1) % NS;
        const int tn = min(t + NS - 1, ntile - 1);
        const int tx = min(t + 1, ntile_run - 1);               // the "next" tile of the pipeline (the last iteration recomputes its own)
        const uint32_t so = slot * STAGE, sx = (tx % NS) * STAGE;
        uint32_t kan[8], ktc[4][2];
#pragma unroll
        for (int j = 0; j < 8; ++j) kan[j] = kax[j] + sx;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ktc[j][0] = ktx[j][0] + so; ktc[j][1] = ktx[j][1] + so; }
        const uint32_t mAt = mA + t * (BC * 2);
        u32x2 mm[4];
        mm[0] = lds_b64<0>(mAt); mm[1] = lds_b64<16>(mAt); mm[2] = lds_b64<32>(mAt); mm[3] = lds_b64<48>(mAt);
        const int wso = (t >> 2) * ws_tile2 + (t & 3) * 64;

        // ---- phase AC: S(t+1) and dP'(t+1), alternating, under: upper half of P(t), dS'(t), the stores of P(t) and of the first half of dS'(t)
        float dpv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {          // dP'(t) leaves the accumulator registers before the new chain starts in them
            dpv[r] = dp[r];
            asm volatile("" : "+v"(dpv[r]));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
        float av[16];
        uint32_t dwv[8], pw[8], sw[8];
        u32x4 fr[RR];
        fr[0] = BMT_P_ROWFRAG(0); fr[1] = BMT_P_ROWFRAG(1); fr[2] = BMT_P_ROWFRAG(2); fr[3] = BMT_P_ROWFRAG(3);
        lgkm_wait<PF>(mm[0], mm[1]); lgkm_wait<PF>(mm[2], mm[3]);
#define BMT_P_DEL(r_)                                                                                   \
    do {                                                                                                \
        av[(r_) & 15] = pr[(r_) & 15] * __builtin_fmaf(dpv[(r_) & 15], p.scale, dsc);                                   \
        if constexpr (((r_) & 1) == 1) {                                                                \
            const float c0_ = __builtin_amdgcn_fmed3f(av[((r_) - 1) & 15], -60000.f, 60000.f);                 \
            const float c1_ = __builtin_amdgcn_fmed3f(av[(r_) & 15], -60000.f, 60000.f);                     \
            dwv[((r_) >> 1) & 7] = pack_h2(c0_, c1_) & mm[((r_) >> 2) & 3][((r_) >> 1) & 1];                        \
            if constexpr (EMIT) sw[((r_) >> 1) & 7] = pack_bf2(av[((r_) - 1) & 15], av[(r_) & 15]);                       \
            rs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, dwv[((r_) >> 1) & 7]), ones, rs, false);     \
        }                                                                                               \
    } while (0)
    // VALU work of combined step i_ (x = step index scaled to 32 steps: d_k 128 has 16 steps, each does two slots)
#define BMT_P_ACWORK(x_)                                                                                \
    do {                                                                                                \
        if constexpr ((x_) < 8) BMT_P_EXPR(8 + (x_), sth[(x_) & 7]);                                        \
        if constexpr (EMIT && (x_) >= 8 && (x_) < 16) pw[((x_) - 8) & 7] = pack_bf2(pr[(2 * ((x_) - 8)) & 15], pr[(2 * ((x_) - 8) + 1) & 15]); \
        if constexpr (((x_) & 1) == 0) BMT_P_DEL((x_) >> 1);                                            \
        if constexpr (EMIT && (x_) == 17) { swap32u(pw[0], pw[2]); swap32u(pw[1], pw[3]); st128<0>(rsP, pw[0], pw[1], pw[2], pw[3], wvo, wso); }  \
        if constexpr (EMIT && (x_) == 19) { swap32u(pw[4], pw[6]); swap32u(pw[5], pw[7]); st128<32>(rsP, pw[4], pw[5], pw[6], pw[7], wvo, wso); } \
        if constexpr (EMIT && (x_) == 21) { swap32u(sw[0], sw[2]); swap32u(sw[1], sw[3]); st128<0>(rsS, sw[0], sw[1], sw[2], sw[3], wvo, wso); }  \
    } while (0)
#define BMT_P_ACSTEP(i_)                                                                               \
    if constexpr ((i_) < 2 * KS) {                                                                     \
        if constexpr ((i_) + PF < 2 * KS) fr[((i_) + PF) % RR] = BMT_P_ROWFRAG((i_) + PF);             \
        if constexpr ((i_) < PPW) BMT_P_DMA_K((i_) % PPW, tn, slotn);                                  \
        else if constexpr ((i_) < 2 * PPW) BMT_P_DMA_V((i_) % PPW, tn, slotn);                         \
        lgkm_wait<((i_) + PF < 2 * KS) ? PF : (2 * KS - 1 - (i_))>(fr[(i_) % RR]);                     \
        if constexpr (((i_) & 1) == 0) st = mfma32t<true>(as_bf16x8(fr[(i_) % RR]), qf[(i_) >> 1], st); \
        else dp = mfma32t<true>(as_bf16x8(fr[(i_) % RR]), dof[(i_) >> 1], dp);                         \
        if constexpr (KS == 16) { BMT_P_ACWORK((i_)); }                                                \
        else { BMT_P_ACWORK(2 * (i_)); BMT_P_ACWORK(2 * (i_) + 1); }                                   \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
#define BMT_P_ACSTEP2(j_) BMT_P_ACSTEP(2 * (j_)) BMT_P_ACSTEP(2 * (j_) + 1)
        BMT_X_REP16(BMT_P_ACSTEP2)
#undef BMT_P_ACSTEP2
#undef BMT_P_ACSTEP
#undef BMT_P_ACWORK
#undef BMT_P_DEL
        bf16x8 dsf[2];
        dsf[0] = as_bf16x8(u32x4{dwv[0], dwv[1], dwv[2], dwv[3]});
        dsf[1] = as_bf16x8(u32x4{dwv[4], dwv[5], dwv[6], dwv[7]});

        // ---- phase B: dQ'^T += K(t)^T . dS'(t)^T (d-tiles inside a 16-key step: no MFMA follows one on its own accumulator) under the lower
        // half of P(t+1) and the last store of dS'(t)
        u32x2 ta[RR], tb[RR];
#define BMT_P_TFRAG(n_)                                                                                       \
    do {                                                                                                      \
        constexpr int dt__ = (n_) % DT, kk__ = (n_) / DT;                                                     \
        constexpr int off__ = (DK == 256 ? (dt__ >> 2) * 256 : 0) + 16 * kk__ * ROWB;                         \
        ta[(n_) % RR] = lds_tr_b64<off__>(ktc[dt__ & 3][0]);                                                  \
        tb[(n_) % RR] = lds_tr_b64<off__ + 8 * ROWB>(ktc[dt__ & 3][1]);                                       \
    } while (0)
        BMT_P_TFRAG(0); BMT_P_TFRAG(1); BMT_P_TFRAG(2); BMT_P_TFRAG(3);
#define BMT_P_BWORK(x_)                                                                                 \
    do {                                                                                                \
        if constexpr (EMIT && (x_) == 0) { swap32u(sw[4], sw[6]); swap32u(sw[5], sw[7]); st128<32>(rsS, sw[4], sw[5], sw[6], sw[7], wvo, wso); } \
        if constexpr ((x_) >= 2 && (x_) < 10) BMT_P_EXPR((x_) - 2, st[((x_) - 2) & 15]);                        \
        if constexpr ((x_) >= 8 && (x_) < 16) { sth[((x_) - 8) & 7] = st[(x_) & 15]; asm volatile("" : "+v"(sth[((x_) - 8) & 7])); } \
    } while (0)
#define BMT_P_BSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + PF < 2 * DT) BMT_P_TFRAG((n_) + PF);                                      \
        lgkm_wait<((n_) + PF < 2 * DT) ? 2 * PF : 2 * (2 * DT - 1 - (n_))>(ta[(n_) % RR], tb[(n_) % RR]); \
        const u32x4 fv = {ta[(n_) % RR][0], ta[(n_) % RR][1], tb[(n_) % RR][0], tb[(n_) % RR][1]};     \
        dq[(n_) % DT] = mfma32t<true>(as_bf16x8(fv), dsf[(n_) / DT], dq[(n_) % DT]);                   \
        if constexpr (2 * DT == 16) { BMT_P_BWORK((n_)); }                                             \
        else { BMT_P_BWORK(2 * (n_)); BMT_P_BWORK(2 * (n_) + 1); }                                     \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
        BMT_X_REP16(BMT_P_BSTEP)
#undef BMT_P_BSTEP
#undef BMT_P_BWORK
#undef BMT_P_TFRAG
        // tile t + 2 has landed once only tile t + 3's requests (this iteration's) may be pending; loads only are counted (a store
        // younger than them can only make the wait longer)
        if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        BMT_B_BAR();
    }
#undef BMT_P_EXPR
#undef BMT_P_ROWFRAG
#undef BMT_P_DMA_K
#undef BMT_P_DMA_V

    if (p.kmean != nullptr) {
        const float rst = half_sum(rs);
        const float* km = p.kmean + (p.kvh != 0 ? ((int64_t)b * p.H + h) : (int64_t)b) * DK      /* (shared keys: ONE mean key per sample) */;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] -= rst * km[dt * 32 + acc_row(r, hh)];
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] *= down;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (XP & 8) { if (dq[0][0] == 1234.5f && dq[1][1] == 3.25f) p.gq.bsum[0] = 1.f; return; }
    grad_rm_epilogue<DK, DT, 256>(smem, p.gq, dq, b, h, qt * 128, wid * 32 + l31, qok, hh, 0, p.Sq, p.SqP, tid);
    if (p.gq.hiT) {
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
        grad_tile_write<DK, 128>(tile, dq, wid * 32, qok, l31, hh);
        __syncthreads();
        GradOut gt = p.gq;
        gt.bsum = nullptr;
        grad_tile_flush<DK, 128>(tile, gt, b, h, qt * 128, p.Sq, tid);
    }
}

template <int DK, int XP = 0, bool EMIT = true>
int launch_dq32p(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds_loop = 4 * 2 * 32 * DK * 2 + ((ntile * 64 + 15) & ~15) + 16, lds_epi = 128 * (DK + 8) * 2 + 256 * 8;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq32p_kernel<DK, XP, EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dq32p_kernel<DK, XP, EMIT>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(dq, pipelined)");
    return BMT_OK;
}

// ------------------------------------------------------------------------------------------------------------ the key side, recomputing (round 6)
// attn_bwd_dq32p_kernel<.., EMIT = false> -> attn_bwd_dkvr_kernel: the split backward WITHOUT its P / dS round trip.  The emitting form wrote
// P and dS' (bf16, 2 x Sq x Sk x 2 bytes per (batch, head): 145 MB per launch of configs[1]'s audio self-attention) and a scaled copy of q, and
// attn_bwd_dkvg8_kernel read them back (profiles/r05_z_pmc_traffic.json: 488.6 MB moved per attention for ~237 MB of operands and gradients):
// both kernels ran HBM-shaped at 2.7 / 3.4 TB/s.  Bytes were the bound, so recomputing S = Q K^T and dP = dO V^T on the key side is nearly free:
// this kernel is the MIRROR IMAGE of the dQ kernel -- a wave owns 32 KEYS, the query side streams through LDS in 32-query stages (the fp16 q
// tile and the bf16 dO tile by LDS-DMA, the dual-purpose swizzle of the dQ kernel's K image: row fragments for S / dP, transposing reads for
// the two gradient products), and BOTH gradients accumulate in registers:
//     S[q x key] = Q . K^T   (fp16)        dP[q x key] = dO . V^T   (bf16: dO is a bf16 plane)
//     dV^T[d x key] += dO^T[d x q] . P[q x key]       (bf16, P <= 1)
//     dK^T[d x key] += Q^T[d x q]  . dS'[q x key]     (fp16: q is an fp16 plane; dS' = dS 2^g with ONE power of two per (batch, head), from the
//                                                      largest |dO| the dQ kernel saw: a sum over queries needs no per-query scale -- a row whose
//                                                      dO is five decades below the largest contributes five decades less to the sum)
// S leaves the MFMA with the KEY on the lane and the queries along the registers, which is exactly the B operand of the two gradient
// products (reduction over the queries, registers 8 kk .. 8 kk + 7 in order): P and dS' never leave the registers.  The softmax statistics
// vary along the REGISTERS here (register r of lane half hh = query 8 (r >> 2) + 4 hh + (r & 3) of the stage): lse log2 e and -delta scale 2^g
// of all the sample's queries are staged into LDS once (rows past Sq: lse = +huge, delta = 0, so P = dS' = 0 whatever the ring holds), a lane
// reads its 2 x 16 values per stage as 8 ds_read_b128.  Nothing is masked in the loop: a key's column of dK / dV depends on no other key, a
// key that is masked or past Sk has its column zeroed / not stored at the end.
// Registers decide the rest.  dK and dV of 32 keys x d_k 256 are 2 x 128 accumulator registers -- the whole AGPR half of a 512-register wave --
// and with K AND V fragments in registers as well (2 x 64) hipcc 7.2 shuttled accumulator tiles between the register files (352 v_accvgpr
// moves per stage) and spilled V fragments, whose scratch reloads count in vmcnt next to the DMA ring.  So: V stays in registers (64), the
// workgroup's 128 K rows sit in LDS (64 KB, the same swizzled image, read as the B operand of S: one more ds_read_b128 per S step), the
// stage ring is two deep (64 KB) and a stage is finished before the next one starts:
//     phase AC: S(t), dP(t) alternating (no MFMA follows one on its own accumulator); the DMA of stage t + 1 rides on the first steps
//     VALU    : P, dS' of registers 0 .. 7 (the first 16-query step's operands)
//     phase B0: dV, dK over queries 0 .. 15 of the stage, alternating, under the VALU of registers 8 .. 15
//     phase B1: dV, dK over queries 16 .. 31
// 4 products of 2 Sq Sk d_k on this side + 3 on the dQ side = 7 for the mathematics' 5: MFMA work bought with HBM bytes.
#define BMT_X_REP32(M) BMT_X_REP16(M) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
template <int N>
__device__ __forceinline__ void lgkm_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
// LDS reads issued behind step i's fragments when steps i + 1 .. min(i + pf, n - 1) have been prefetched: one fragment per step, two on even steps
// S and dP accumulate in ARCHITECTURAL registers, by inline asm: dK and dV fill the accumulator file (256), and hipcc 7.2 -- which prefers the
// accumulator file for every MFMA result -- made room for S / dP there by moving gradient tiles out and back (320 v_accvgpr moves per stage).
// A chain's links take the accumulator whole as C (no wait states between them: cdna_hip_programming.md section 5.7, item 2); the first
// link starts from the inline constant 0 (early-clobber: the result must not share registers with its operands); the readers after the
// last link wait 16 states (mfma_v_done).
template <bool F16, bool FIRST>
__device__ __forceinline__ void mfma_v(f32x16& c, const u32x4& a, const u32x4& b) {
    if constexpr (FIRST) {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    } else {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
}
__device__ __forceinline__ void mfma_v_done(f32x16& a, f32x16& b) { asm volatile("s_nop 15" : "+v"(a), "+v"(b)); }
constexpr int dkvr_younger(int i, int pf, int n) {
    int c = 0;
    for (int s = i + 1; s <= i + pf && s < n; ++s) c += 1 + ((s & 1) == 0 ? 1 : 0);
    return c;
}

#ifndef BMT_DKVR_XP          // build-variant probes (BMT_VARIANT_FLAGS, csrc/build.sh): XP bit 3 = no epilogue, bit 4 = no loop; PF = fragment prefetch depth
#define BMT_DKVR_XP 0
#endif
#ifndef BMT_DKVR_PF
#define BMT_DKVR_PF 4
#endif
template <int DK, int XP = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkvr_kernel(const AttnPB pin) {
    AttnPB p = pin;
    constexpr int BQ = 32, NT = 256, KS = DK / 16, DT = DK / 32, ROWB = DK * 2, TILE = BQ * ROWB, STAGE = 2 * TILE, NS = 2;
    constexpr int KBLK = 128 * ROWB;                                       // the workgroup's K rows
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NP = BQ / RPP, PPW = NP / 4, KPW = (128 / RPP) / 4;
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [ring: NS stages][K block][statistics]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l31 = lane & 31;
    const int nkt = (p.Sk + 127) / 128, nqt = (p.SqP + 127) / 128, nstP = (p.SqP + BQ - 1) / BQ;
    const int w = xcd_remap(blockIdx.x, nkt * p.B * p.H);
    // work order: (batch, head)-major -- the key blocks of a (batch, head) run together and share its q / dO rows in their XCD's L2
    const int kt = w % nkt, bhw = w / nkt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (kt * 128 >= p.Sk) {                  // a key block past the sample's length: zero rows of bias partials, nothing else
        for (int d = tid; d < DK; d += NT) {
            if (p.gk.bpart != nullptr) p.gk.bpart[((int64_t)b * nkt + kt) * p.gk.bp_ld + h * DK + d] = 0.f;
            if (p.gv.bpart != nullptr) p.gv.bpart[((int64_t)b * nkt + kt) * p.gv.bp_ld + h * DK + d] = 0.f;
        }
        return;
    }
    const int nst = (p.Sq + BQ - 1) / BQ;
    const int key = kt * 128 + wid * 32 + l31;
    const bool kin = key < p.Sk;
    const bool kok = kin && (p.mask == nullptr || p.mask[(int64_t)b * p.mask_bs + key] != 0);
    // live 32-query stages (AttnPB.qlive, written by the dQ kernel): the loop ends behind the last one -- a dead stage has dO = 0, hence dP = delta = 0
    // and contributes exactly nothing, so dead stages in front of a live one are simply computed
    uint64_t smask = ~0ull;
    float am = 0.f;
    if (p.qlive != nullptr) {
        smask = 0ull;
        for (int i = 0; i < nqt; ++i) smask |= (uint64_t)(uint32_t)(p.qlive[(int64_t)bh * nqt + i] & 15) << (4 * i);
    }
    for (int i = 0; i < nqt; ++i) am = fmaxf(am, p.qamax[(int64_t)bh * nqt + i]);
    const int live_end = smask == 0ull ? 0 : 64 - __builtin_clzll(smask);
    const int nst_run = ((XP & 16) || !__syncthreads_or((int)kok)) ? 0 : min(nst, live_end);
    // one power of two per (batch, head) puts the largest |dO| at 2^6 .. 2^7: dS' = P (dP - delta) scale 2^g stays inside fp16 (clamped at +-60000)
    int kexp = 0;
    if (am > 0.f) kexp = 6 - ((int)((__float_as_uint(am) >> 23) & 0xffu) - 127);
    kexp = max(-60, min(60, kexp));
    const float up = __uint_as_float((uint32_t)(127 + kexp) << 23), down = __uint_as_float((uint32_t)(127 - kexp) << 23);
    const float sg = p.scale * up, sc2 = p.scale * LOG2E;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Qh + (int64_t)b * p.bsq + h * DK), 0, plane_extent(p.Sq, p.ldq, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dOh + (int64_t)b * p.bso + h * DK), 0, plane_extent(p.Sq, p.ldo, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * p.kvh), 0, plane_extent(p.Sk, p.ldk, DK), 0x00020000);
    // LDS-DMA pieces (1 KB = RPP rows): wave w moves pieces w, w + 4, w + 8, .. of each tile, so that the rows of its pieces differ by multiples of
    // 4 RPP -- the swizzle term of piece j is then the first piece's (d_k 128: rows 16 j apart) or the first piece's ^ 32 bytes for odd j
    // (d_k 256: rows 8 j apart flip bit 1 of (row >> 2) & 3) and the row term goes into the scalar offset: two offset registers per tile
    // instead of one per piece
    static_assert(PPW <= 4, "pieces per wave");
    constexpr int ODD = DK == 256 ? 32 : 0;
    int qv0, ov0, kv0, qv1, ov1, kv1;
    {
        const int row = wid * RPP + lane / CPR, cpos = lane % CPR;
        const int cs = (cpos ^ kswz(row)) * 16;
        qv0 = row * (int)p.ldq * 2 + cs; qv1 = row * (int)p.ldq * 2 + (cs ^ ODD);
        ov0 = row * (int)p.ldo * 2 + cs; ov1 = row * (int)p.ldo * 2 + (cs ^ ODD);
        kv0 = row * (int)p.ldk * 2 + cs; kv1 = row * (int)p.ldk * 2 + (cs ^ ODD);
    }
    const int sstep_q = BQ * (int)p.ldq * 2, sstep_o = BQ * (int)p.ldo * 2;
    const int pstep_q = 4 * RPP * (int)p.ldq * 2, pstep_o = 4 * RPP * (int)p.ldo * 2, pstep_k = 4 * RPP * (int)p.ldk * 2;
#define BMT_R_DMA_Q(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + (slot_) * STAGE + (wid + 4 * (j_)) * 1024), 16, ((j_) & 1) ? qv1 : qv0, (t_) * sstep_q + (j_) * pstep_q, 0, 0)
#define BMT_R_DMA_O(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + (slot_) * STAGE + TILE + (wid + 4 * (j_)) * 1024), 16, ((j_) & 1) ? ov1 : ov0, (t_) * sstep_o + (j_) * pstep_o, 0, 0)

    // ---- prologue: the K block and stage 0 in flight; the softmax statistics of every query; this wave's V rows
    {
        const int kbase = kt * 128 * (int)p.ldk * 2;
#pragma unroll
        for (int j = 0; j < KPW; ++j)      // (rows past Sk: out of the descriptor's range, zeros)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + NS * STAGE + (wid + 4 * j) * 1024), 16, (j & 1) ? kv1 : kv0, kbase + j * pstep_k, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PPW; ++j) BMT_R_DMA_Q(j, 0, 0);
#pragma unroll
    for (int j = 0; j < PPW; ++j) BMT_R_DMA_O(j, 0, 0);
    float* l2s = reinterpret_cast<float*>(smem + NS * STAGE + KBLK);     // [nstP * 32] lse log2 e  (+huge past Sq)
    float* dcs = l2s + nstP * BQ;                                         // [nstP * 32] -delta scale 2^g  (0 past Sq)
    {
        const int64_t sbase = ((int64_t)b * p.H + h) * p.SqP;
        for (int i = tid; i < nst * BQ; i += NT) {
            const bool ok = i < p.Sq;
            const int64_t si = sbase + min(i, p.Sq - 1);
            const float l = p.lse[si], d = p.delta[si];
            l2s[i] = ok ? l * LOG2E : 1e30f;
            dcs[i] = ok ? -d * sg : 0.f;
        }
    }
    u32x4 vf[KS];
    {
        // (UNCONDITIONAL loads from a clamped row, as everywhere: a key past Sk carries the last key's values, its columns are never stored)
        const int kc = min(key, p.Sk - 1);
        const int64_t vo = (int64_t)b * p.bsv + (int64_t)kc * p.ldv + h * p.kvh + 8 * hh;
        u32x4 vr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vr[ks] = *reinterpret_cast<const u32x4*>(p.Vh + vo + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks] = h8_to_b8(vr[ks]);      // dP = dO . V^T runs in bf16 (dO is a bf16 plane)
    }
    f32x16 dka[DT], dva[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }

    // fragment addresses (LDS bytes): the dQ kernel's algebra with the roles exchanged (rows of a stage image = queries).  Every address is
    // (lane constant + stage offset) ^ (step bits 5 .. 7), formed per read -- one v_xor -- instead of being kept in 16 + 16 registers
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int fk = kswz(l31);
    const uint32_t kA0 = lds0 + l31 * ROWB + 32 * (fk >> 1) + 16 * (hh ^ (fk & 1));
    const uint32_t kB0 = kA0 + NS * STAGE + wid * 32 * ROWB;          // this lane's key row in the K block (same row & 15, same swizzle)
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t kT0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1);
    const uint32_t lA = lds0 + NS * STAGE + KBLK + 16 * hh;      // this lane's statistics: queries 8 i + 4 hh .. + 3 of stage t at lA + t * 128 + 32 i
    const uint32_t dA = lA + nstP * BQ * 4;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int PF = BMT_DKVR_PF, RR = PF + 1;
#define BMT_R_ROWFRAG(i_) lds_b128<(DK == 256 ? (((i_) >> 1) >> 3) * 256 : 0) + (((i_) & 1) ? TILE : 0)>(kAs ^ ((((i_) >> 1) & 7) << 5))
#define BMT_R_KFRAG(ks_) lds_b128<(DK == 256 ? ((ks_) >> 3) * 256 : 0)>(kB0 ^ (((ks_) & 7) << 5))
    // register r of the S / dP accumulators -> P, dS'.  x_ even: P; x_ odd: dS' (and, on the odd register of a pair, the two packed words)
#define BMT_R_PS(x_)                                                                                                            \
    do {                                                                                                                        \
        constexpr int r__ = ((x_) >> 1) & 15;                                                                                  \
        if constexpr (((x_) & 1) == 0) {                                                                                       \
            pr[r__ & 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r__], sc2, -__uint_as_float(ls[r__ >> 2][r__ & 3])));       \
        } else {                                                                                                               \
            av[r__ & 1] = __builtin_amdgcn_fmed3f(pr[r__ & 1] * __builtin_fmaf(dp[r__], sg, __uint_as_float(dc[r__ >> 2][r__ & 3])), -60000.f, 60000.f); \
            if constexpr ((r__ & 1) == 1) {                                                                                    \
                pfw[r__ >> 1] = pack_bf2(pr[0], pr[1]);                                                                        \
                sfw[r__ >> 1] = pack_h2(av[0], av[1]);                                                                         \
            }                                                                                                                  \
        }                                                                                                                      \
    } while (0)

    for (int t = 0; t < nst_run; ++t) {
        const int slot = t & 1, slotn = slot ^ 1;
        const int tn = min(t + 1, nst - 1);                  // the last stage re-fetches itself into the idle slot (branch-free)
        const uint32_t so = slot * STAGE;
        f32x16 st, dp;
        {
            const uint32_t kAs = kA0 + so;
            u32x4 fr[RR], fkk[RR];         // step i's stage fragment (even: q rows, odd: dO rows) in fr[i % RR]; an even step's K fragment in fkk[(i / 2) % RR]
#define BMT_R_ACPRE(i_)                                                                               \
    if constexpr ((i_) < PF) {                                                                         \
        fr[(i_) % RR] = BMT_R_ROWFRAG(i_);                                                             \
        if constexpr (((i_) & 1) == 0) fkk[((i_) >> 1) % RR] = BMT_R_KFRAG((i_) >> 1);                 \
    }
            BMT_X_REP16(BMT_R_ACPRE)
#undef BMT_R_ACPRE
#define BMT_R_ACSTEP(i_)                                                                               \
    if constexpr ((i_) < 2 * KS) {                                                                     \
        if constexpr ((i_) + PF < 2 * KS) {                                                            \
            fr[((i_) + PF) % RR] = BMT_R_ROWFRAG((i_) + PF);                                           \
            if constexpr ((((i_) + PF) & 1) == 0) fkk[(((i_) + PF) >> 1) % RR] = BMT_R_KFRAG(((i_) + PF) >> 1); \
        }                                                                                              \
        if constexpr ((i_) < PPW) BMT_R_DMA_Q((i_) % PPW, tn, slotn);                                  \
        else if constexpr ((i_) < 2 * PPW) BMT_R_DMA_O((i_) % PPW, tn, slotn);                         \
        if constexpr (((i_) & 1) == 0) {                                                               \
            lgkm_wait2<dkvr_younger((i_), PF, 2 * KS)>(fr[(i_) % RR], fkk[((i_) >> 1) % RR]);          \
            mfma_v<true, (i_) == 0>(st, fr[(i_) % RR], fkk[((i_) >> 1) % RR]);                         \
        } else {                                                                                       \
            lgkm_wait<dkvr_younger((i_), PF, 2 * KS)>(fr[(i_) % RR]);                                  \
            mfma_v<false, (i_) == 1>(dp, fr[(i_) % RR], vf[(i_) >> 1]);                                \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
            BMT_X_REP32(BMT_R_ACSTEP)
#undef BMT_R_ACSTEP
            mfma_v_done(st, dp);
        }
        uint32_t pfw[8], sfw[8];           // P as bf16 pairs, dS' as fp16 pairs: word w = registers 2 w, 2 w + 1
        float pr[2], av[2];
        u32x4 ls[4], dc[4];                // this lane's statistics of the stage: queries 8 i + 4 hh .. + 3 in ls[i] / dc[i]
        {
            const uint32_t lAt = lA + t * (BQ * 4), dAt = dA + t * (BQ * 4);
            ls[0] = lds_b128<0>(lAt); ls[1] = lds_b128<32>(lAt); ls[2] = lds_b128<64>(lAt); ls[3] = lds_b128<96>(lAt);
            dc[0] = lds_b128<0>(dAt); dc[1] = lds_b128<32>(dAt); dc[2] = lds_b128<64>(dAt); dc[3] = lds_b128<96>(dAt);
        }
        const uint32_t kTs = kT0 + so;
        constexpr int NSTEP = 4 * DT;
        u32x2 ta[RR], tb[RR];
        // step m: product m & 1 (0: dV from the dO tile, 1: dK from the q tile), d-tile (m >> 1) % DT, 16-query step (m >> 1) / DT
#define BMT_R_TFRAG(m_)                                                                                       \
    do {                                                                                                      \
        constexpr int dt__ = ((m_) >> 1) % DT, kk__ = ((m_) >> 1) / DT;                                       \
        constexpr int off__ = (((m_) & 1) ? 0 : TILE) + (DK == 256 ? (dt__ >> 2) * 256 : 0) + 16 * kk__ * ROWB; \
        ta[(m_) % RR] = lds_tr_b64<off__>(kTs ^ ((dt__ & 3) << 6));                                           \
        tb[(m_) % RR] = lds_tr_b64<off__ + 8 * ROWB>(kTs ^ (((dt__ & 3) << 6) | 32));                         \
    } while (0)
#define BMT_R_BPRE(m_) if constexpr ((m_) < PF) { BMT_R_TFRAG(m_); }
        BMT_X_REP16(BMT_R_BPRE)
#undef BMT_R_BPRE
        lgkm_wait4<2 * PF>(ls[0], ls[1], ls[2], ls[3]);          // (LDS operations retire in order: the statistics are older than the fragments)
        lgkm_wait4<2 * PF>(dc[0], dc[1], dc[2], dc[3]);
#define BMT_R_V1(x_) if constexpr ((x_) < 16) { BMT_R_PS(x_); }
        BMT_X_REP16(BMT_R_V1)
#undef BMT_R_V1
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 pf[2], sf[2];
        pf[0] = as_bf16x8(u32x4{pfw[0], pfw[1], pfw[2], pfw[3]});
        sf[0] = as_bf16x8(u32x4{sfw[0], sfw[1], sfw[2], sfw[3]});
        // VALU work of step m_ < NSTEP / 2: the 16 half-steps of registers 8 .. 15 spread over the first half's steps
#define BMT_R_BSTEP(m_)                                                                                \
    if constexpr ((m_) < NSTEP) {                                                                      \
        if constexpr ((m_) == NSTEP / 2) {                                                             \
            pf[1] = as_bf16x8(u32x4{pfw[4], pfw[5], pfw[6], pfw[7]});                                  \
            sf[1] = as_bf16x8(u32x4{sfw[4], sfw[5], sfw[6], sfw[7]});                                  \
        }                                                                                              \
        if constexpr ((m_) + PF < NSTEP) BMT_R_TFRAG((m_) + PF);                                       \
        lgkm_wait<((m_) + PF < NSTEP) ? 2 * PF : 2 * (NSTEP - 1 - (m_))>(ta[(m_) % RR], tb[(m_) % RR]); \
        const u32x4 fv = {ta[(m_) % RR][0], ta[(m_) % RR][1], tb[(m_) % RR][0], tb[(m_) % RR][1]};     \
        if constexpr (((m_) & 1) == 0) dva[((m_) >> 1) % DT] = mfma32t<false>(as_bf16x8(fv), pf[((m_) >> 1) / DT], dva[((m_) >> 1) % DT]); \
        else dka[((m_) >> 1) % DT] = mfma32t<true>(as_bf16x8(fv), sf[((m_) >> 1) / DT], dka[((m_) >> 1) % DT]); \
        if constexpr ((m_) < NSTEP / 2) {                                                              \
            if constexpr (NSTEP == 32) { BMT_R_PS(16 + (m_)); }                                        \
            else { BMT_R_PS(16 + 2 * (m_)); BMT_R_PS(16 + 2 * (m_) + 1); }                             \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
        BMT_X_REP32(BMT_R_BSTEP)
#undef BMT_R_BSTEP
#undef BMT_R_TFRAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage t + 1 has landed (this wave's share); the barrier publishes all shares
        BMT_B_BAR();                                          // and says every wave is done reading stage t (stage t + 2 overwrites it)
    }
#undef BMT_R_PS
#undef BMT_R_ROWFRAG
#undef BMT_R_KFRAG
#undef BMT_R_DMA_Q
#undef BMT_R_DMA_O

    if (!kok) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dka[dt] *= down;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (XP & 8) { if (dka[0][0] == 1234.5f && dva[1][1] == 3.25f) p.gk.bsum[0] = 1.f; return; }
    grad_rm_epilogue<DK, DT, 256>(smem, p.gv, dva, b, h, kt * 128, wid * 32 + l31, kin, hh, 0, p.Sk, p.SkP, tid);
    grad_rm_epilogue<DK, DT, 256>(smem, p.gk, dka, b, h, kt * 128, wid * 32 + l31, kin, hh, 0, p.Sk, p.SkP, tid);
}

template <int DK, int XP = 0>
int launch_dkvr(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sk + 127) / 128) * p.B * p.H;
    const int nstP = (p.Sq + 31) / 32;
    const int lds_loop = 2 * 2 * 32 * DK * 2 + 128 * DK * 2 + nstP * 32 * 8, lds_epi = 128 * (DK + 8) * 2 + 256 * 8;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkvr_kernel<DK, XP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dkvr_kernel<DK, XP>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_attn_bwd_bf16 (recompute form, dK / dV)");
    return BMT_OK;
}

// ---- the same products on 8 waves = two per SIMD: wave (kg = wid & 3, dh = wid >> 2) owns 32 keys x HALF of d_k for both gradients (2 x 64
// accumulator registers: two waves fit a SIMD).  Why: every operand is a ds_read_b64_tr_b16, and one wave per SIMD cannot issue them fast
// enough (PMC of attn_bwd_dkvg_kernel, profiles/r03_f_split_pmc.csv: MFMA 23 % busy, waves issue-stalled 41 % of the time, no bank conflict;
// with the MFMAs AND the DMA switched off the loop still takes 60 % of its time: MI355X_MICROARCH.md, LDS: 4- and 8-byte reads reach their
// rate only from several waves per SIMD).  The P / dS fragments are read by both d-halves (+8 reads per stage and wave pair).
template <int DK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkvg8_kernel(const AttnPB pin) {
    AttnPB p = pin;
    constexpr int BQ = 32, DT = DK / 32, DTL = DT / 2, ROWB = DK * 2, XT = BQ * ROWB, YT = BQ * 256, STAGE = 2 * XT + 2 * YT;
    constexpr int NS = (DK == 256) ? 3 : 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NPX = BQ / RPP, PPW = NPX / 8;      // X tiles: 16 (8) pieces of 1 KB over 8 waves; Y tiles: 8 pieces
    constexpr int NDMA = 2 * PPW + 2;                                               // requests per wave and stage
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wid & 3, dh = wid >> 2;
    const int hh = lane >> 5, l31 = lane & 31;
    const int nkt = (p.Sk + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nkt * p.B * p.H);
    const int kt = w % nkt, bhw = w / nkt;
    const int h = bhw % p.H, b = attn_sample(p, bhw / p.H), bh = b * p.H + h;
    attn_rebase(p, b);                       // packed rows: this sample's rows and lengths
    if (kt * 128 >= p.Sk) {                  // a key block past the sample's length: zero rows of bias partials, nothing else
        for (int d = tid; d < DK; d += 512) {
            if (p.gk.bpart != nullptr) p.gk.bpart[((int64_t)b * nkt + kt) * p.gk.bp_ld + h * DK + d] = 0.f;
            if (p.gv.bpart != nullptr) p.gv.bpart[((int64_t)b * nkt + kt) * p.gv.bp_ld + h * DK + d] = 0.f;
        }
        return;
    }
    const int nst = (p.Sq + BQ - 1) / BQ;
    const int key = kt * 128 + kg * 32 + l31;
    const bool kin = key < p.Sk;
    const bool kok = kin && (p.mask == nullptr || p.mask[(int64_t)b * p.mask_bs + key] != 0);
    // live 32-query stages of this (batch, head) (AttnPB.qlive: bit t = stage t has a non-zero dO): the loop ends behind the last live
    // one; a dead stage BEFORE it (never in this model: padding is a suffix) runs with its P / dS fragments wiped -- nobody wrote them
    uint64_t smask = ~0ull;
    if (p.qlive != nullptr) {
        smask = 0ull;
        const int nqt = (p.SqP + 127) / 128;
        for (int i = 0; i < nqt; ++i) smask |= (uint64_t)(uint32_t)(p.qlive[(int64_t)bh * nqt + i] & 15) << (4 * i);
    }
    const int live_end = smask == 0ull ? 0 : 64 - __builtin_clzll(smask);
    const int nst_run = __syncthreads_or((int)kok) ? min(nst, live_end) : 0;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Qbws + (int64_t)b * p.bsqb + h * DK), 0, plane_extent(p.Sq, p.ldqb, DK), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dOh + (int64_t)b * p.bso + h * DK), 0, plane_extent(p.Sq, p.ldo, DK), 0x00020000);
    const int64_t slab = (int64_t)bh * p.ws_slab + kt * p.ws_tile;
    const int slab_bytes = (int)(((int64_t)p.Sq * p.ws_pitch - (p.ws_tile == 128 ? kt * 128 : 0)) * 2);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Pws + slab), 0, slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dSws + slab), 0, slab_bytes, 0x00020000);
    int xq[2], xo[2], yo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wid * PPW + (j % PPW)) * RPP + lane / CPR, cpos = lane % CPR;
        xq[j] = row * (int)p.ldqb * 2 + ((cpos ^ kswz(row)) * 16);
        xo[j] = row * (int)p.ldo * 2 + ((cpos ^ kswz(row)) * 16);
    }
    {
        const int row = wid * 4 + lane / 16, cpos = lane % 16;
        yo = row * (int)p.ws_pitch * 2 + ((cpos ^ ((row & 3) << 2)) * 16);
    }
    const int sstep_q = BQ * (int)p.ldqb * 2, sstep_o = BQ * (int)p.ldo * 2, sstep_y = BQ * (int)p.ws_pitch * 2;
#define BMT_H_DMA(i_, slot_)                                                                                                              \
    do {                                                                                                                                  \
        if constexpr ((i_) < PPW)                                                                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + (slot_) * STAGE + (wid * PPW + (i_)) * 1024), 16, xq[(i_) % 2], 0, 0, 0); \
        else if constexpr ((i_) < 2 * PPW)                                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + (slot_) * STAGE + XT + (wid * PPW + (i_) - PPW) * 1024), 16, xo[((i_) - PPW) % 2], 0, 0, 0); \
        else if constexpr ((i_) == 2 * PPW)                                                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lptr_t)(smem + (slot_) * STAGE + 2 * XT + wid * 1024), 16, yo, 0, 0, 0);       \
        else if constexpr ((i_) == 2 * PPW + 1)                                                                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lptr_t)(smem + (slot_) * STAGE + 2 * XT + YT + wid * 1024), 16, yo, 0, 0, 0);  \
    } while (0)
#define BMT_H_ADVANCE()                                                       \
    do {                                                                      \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) { xq[j] += sstep_q; xo[j] += sstep_o; } \
        yo += sstep_y;                                                        \
    } while (0)
#define BMT_H_DMA_ALL(slot_) \
    do { BMT_H_DMA(0, slot_); BMT_H_DMA(1, slot_); BMT_H_DMA(2, slot_); BMT_H_DMA(3, slot_); BMT_H_DMA(4, slot_); BMT_H_DMA(5, slot_); } while (0)

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        BMT_H_DMA_ALL(s);
        BMT_H_ADVANCE();
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t xT0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1);
    // this wave's d-tiles are dh * DTL + (0 .. DTL - 1): residues dt & 3 = all four at d_k 256 (DTL 4, dt >> 2 = dh), two at d_k 128 (DTL 2)
    uint32_t xa[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dt = dh * DTL + (j % DTL);
        const uint32_t base = xT0 + (DK == 256 ? (dt >> 2) * 256 : 0);
        xa[j][0] = base ^ ((dt & 3) << 6);
        xa[j][1] = base ^ (((dt & 3) << 6) | 32);
    }
    const uint32_t yB0 = lds0 + 2 * XT + (4 * hh + mq) * 256 + 64 * (kg ^ mq) + 32 * gi + 8 * mr;

    f32x16 dka[DTL], dva[DTL];
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nst_run; ++t) {
        const int slot = t % NS, slotn = (t + NS - 1) % NS;
        {
            const uint32_t so = slot * STAGE;
            uint32_t xs[4][2];
#pragma unroll
            for (int j = 0; j < DTL; ++j) { xs[j][0] = xa[j][0] + so; xs[j][1] = xa[j][1] + so; }
            const uint32_t yB = yB0 + so;
            u32x2 yr[8];
            yr[0] = lds_tr_b64<0 * 256>(yB);           yr[1] = lds_tr_b64<8 * 256>(yB);
            yr[2] = lds_tr_b64<16 * 256>(yB);          yr[3] = lds_tr_b64<24 * 256>(yB);
            yr[4] = lds_tr_b64<YT + 0 * 256>(yB);      yr[5] = lds_tr_b64<YT + 8 * 256>(yB);
            yr[6] = lds_tr_b64<YT + 16 * 256>(yB);     yr[7] = lds_tr_b64<YT + 24 * 256>(yB);
            // step m: 16-query step kk = m / (2 DTL), local d-tile (m % (2 DTL)) >> 1, m & 1 = 0: dV from the dO tile, 1: dK from the Qb tile
            constexpr int PF = 3, RR = PF + 1, NSTEP = 4 * DTL;
            u32x2 ta[RR], tb[RR];
#define BMT_H_TFRAG(m_)                                                                                                   \
    do {                                                                                                                  \
        constexpr int kk__ = (m_) / (2 * DTL), dl__ = ((m_) % (2 * DTL)) >> 1;                                            \
        constexpr int off__ = (((m_) & 1) ? 0 : XT) + 16 * kk__ * ROWB;                                                   \
        ta[(m_) % RR] = lds_tr_b64<off__>(xs[dl__][0]);                                                                   \
        tb[(m_) % RR] = lds_tr_b64<off__ + 8 * ROWB>(xs[dl__][1]);                                                        \
    } while (0)
            BMT_H_TFRAG(0); BMT_H_TFRAG(1); BMT_H_TFRAG(2);
            lgkm_wait<2 * PF>(yr[0], yr[1]); lgkm_wait<2 * PF>(yr[2], yr[3]); lgkm_wait<2 * PF>(yr[4], yr[5]); lgkm_wait<2 * PF>(yr[6], yr[7]);
            const bool live_t = (smask >> t) & 1ull;
            if ((t == nst - 1 && (p.Sq & 31) != 0) || !live_t) {       // rows past Sq (and the rows of a dead stage) must not contribute whatever the ring holds there
                const int rem = live_t ? p.Sq - t * BQ : 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = (i >> 1) & 1, u = i & 1;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int q0 = 16 * kk + 8 * u + 4 * hh + 2 * c;
                        const uint32_t m = (q0 < rem ? 0xffffu : 0u) | (q0 + 1 < rem ? 0xffff0000u : 0u);
                        yr[i][c] &= m;
                    }
                }
            }
            bf16x8 pf[2], sf[2];
            pf[0] = as_bf16x8(u32x4{yr[0][0], yr[0][1], yr[1][0], yr[1][1]});
            pf[1] = as_bf16x8(u32x4{yr[2][0], yr[2][1], yr[3][0], yr[3][1]});
            sf[0] = as_bf16x8(u32x4{yr[4][0], yr[4][1], yr[5][0], yr[5][1]});
            sf[1] = as_bf16x8(u32x4{yr[6][0], yr[6][1], yr[7][0], yr[7][1]});
#define BMT_H_STEP(m_)                                                                                     \
    if constexpr ((m_) < NSTEP) {                                                                          \
        if constexpr ((m_) + PF < NSTEP) BMT_H_TFRAG((m_) + PF);                                           \
        if constexpr ((m_) < NDMA) BMT_H_DMA((m_), slotn);                                                 \
        lgkm_wait<((m_) + PF < NSTEP) ? 2 * PF : 2 * (NSTEP - 1 - (m_))>(ta[(m_) % RR], tb[(m_) % RR]);    \
        const u32x4 av = {ta[(m_) % RR][0], ta[(m_) % RR][1], tb[(m_) % RR][0], tb[(m_) % RR][1]};         \
        if constexpr (((m_) & 1) == 0) dva[((m_) % (2 * DTL)) >> 1] = mfma32t<false>(as_bf16x8(av), pf[(m_) / (2 * DTL)], dva[((m_) % (2 * DTL)) >> 1]); \
        else dka[((m_) % (2 * DTL)) >> 1] = mfma32t<false>(as_bf16x8(av), sf[(m_) / (2 * DTL)], dka[((m_) % (2 * DTL)) >> 1]); \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
            BMT_X_REP16(BMT_H_STEP)
#undef BMT_H_STEP
#undef BMT_H_TFRAG
        }
        BMT_H_ADVANCE();
        if constexpr (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        BMT_B_BAR();
    }
#undef BMT_H_DMA_ALL
#undef BMT_H_ADVANCE
#undef BMT_H_DMA

    if (!kok) {
#pragma unroll
        for (int dt = 0; dt < DTL; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    grad_rm_epilogue<DK, DTL, 512>(smem, p.gv, dva, b, h, kt * 128, kg * 32 + l31, kin, hh, dh * DTL, p.Sk, p.SkP, tid);
    grad_rm_epilogue<DK, DTL, 512>(smem, p.gk, dka, b, h, kt * 128, kg * 32 + l31, kin, hh, dh * DTL, p.Sk, p.SkP, tid);
}

template <int DK>
int launch_dkvg8(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sk + 127) / 128) * p.B * p.H;
    const int NS = DK == 256 ? 3 : 4;
    const int lds_loop = NS * (2 * 32 * DK * 2 + 2 * 32 * 256), lds_epi = 128 * (DK + 8) * 2 + 512 * 8;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkvg8_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dkvg8_kernel<DK>), dim3(nblk), dim3(512), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(dkv, 8 waves)");
    return BMT_OK;
}

// rows of per-tile column sums -> the bias gradients: out[c] += sum_r part[r][c]; grid (D / 256, chunks of rows, 3 gradients)
__global__ __launch_bounds__(256) void attn_bias_finish_kernel(const float* pq, int rq, float* oq, const float* pk, const float* pv, int rk, float* ok,
                                                               float* ov, int D) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const float* part = blockIdx.z == 0 ? pq : (blockIdx.z == 1 ? pk : pv);
    float* out = blockIdx.z == 0 ? oq : (blockIdx.z == 1 ? ok : ov);
    const int rows = blockIdx.z == 0 ? rq : rk;
    if (c >= D || out == nullptr || part == nullptr) return;
    float s = 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) s += part[(int64_t)r * D + c];
    atomicAdd(out + c, s);
}

template <int DK>
void launch_bias_finish(const AttnPB& p, hipStream_t st) {
    if (!(p.gq.bpart || p.gk.bpart || p.gv.bpart) || p.defer_bias) return;
    const int D = p.H * DK, rq = p.B * ((p.Sq + 127) / 128), rk = p.B * ((p.Sk + 127) / 128);
    const int chunks = rk > rq ? (rk < 32 ? rk : 32) : (rq < 32 ? rq : 32);
    hipLaunchKernelGGL(attn_bias_finish_kernel, dim3(bmt_cdiv(D, 256), chunks, 3), dim3(256), 0, st, p.gq.bpart, rq, p.gq.bsum, p.gk.bpart,
                       p.gv.bpart, rk, p.gk.bsum, p.gv.bsum, D);
}

template <int DK>
int launch_bwd(const AttnPB& p, uint16_t* dOh, hipStream_t st) {
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    const int nblk_q = ((p.Sq + 127) / 128) * p.B * p.H;
    if constexpr (DK >= 128) {
        if (p.Pws == nullptr && p.qamax != nullptr) {      // recompute form (round 6): the dQ kernel emits nothing but delta, the live-query bits and
                                                           // max |dO|; the key side rebuilds P and dS from q / k / v / dO
            AttnPB pf = p;
            pf.fuse_delta = (p.dO == nullptr && p.O == nullptr && (p.Oph != nullptr || p.Opf != nullptr)) ? 1 : 0;
            if (!pf.fuse_delta) hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, DK, dOh);
            int rc = launch_dq32p<DK, 0, false>(pf, st);
            if (rc != BMT_OK) return rc;
            rc = launch_dkvr<DK, BMT_DKVR_XP>(pf, st);
            if (rc != BMT_OK) return rc;
            launch_bias_finish<DK>(p, st);
            BMT_CHECK_LAUNCH("bmt_attn_bwd_bf16 (recompute)");
            return BMT_OK;
        }
        if (p.Pws != nullptr) {       // split form (bmt_attn_bwd_bf16 checked shapes and sizes): dQ + emission (delta = rowsum(dO * O) in its prologue), dK / dV
                                      // as plain products, bias sums
            AttnPB pf = p;
            pf.fuse_delta = (p.dO == nullptr && p.O == nullptr && (p.Oph != nullptr || p.Opf != nullptr)) ? 1 : 0;
            if (!pf.fuse_delta) hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, DK, dOh);
            int rc = launch_dq32p<DK>(pf, st);
            if (rc != BMT_OK) return rc;
            rc = launch_dkvg8<DK>(pf, st);
            if (rc != BMT_OK) return rc;
            launch_bias_finish<DK>(p, st);
            BMT_CHECK_LAUNCH("bmt_attn_bwd_bf16 (split)");
            return BMT_OK;
        }
        // the two-kernel form as ONE launch (attn_bwd_pair_kernel: the decoder's 30-query attentions, per-query masks): delta from its own
        // small kernel, then dQ and dK / dV workgroups side by side.  Operand rows are fetched with 32-bit byte offsets from the (batch, head) base
        if (!((int64_t)p.Sk * p.ldk * 2 < (1ll << 31) && (int64_t)p.Sk * p.ldv * 2 < (1ll << 31) && (int64_t)p.Sq * p.ldq * 2 < (1ll << 31) &&
              (int64_t)p.Sq * p.ldo * 2 < (1ll << 31))) {
            bmt_set_error("bmt_attn_bwd_bf16: a sample's plane slice of 2 GiB or more");
            return BMT_EINVAL;
        }
        hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, DK, dOh);
        const int nblk_k16 = ((p.Sk + 127) / 128) * p.B * p.H;
        const int lds_epi = 128 * (DK + 8) * 2 + 512 * 8;
        const int lds_dq = 2 * 2 * 32 * (DK * 2 + 32) + 256, lds_dkv = 2 * 2 * 32 * (DK * 2 + 32) + 512;
        int lds = lds_dq > lds_dkv ? lds_dq : lds_dkv;
        lds = lds > lds_epi ? lds : lds_epi;
        const bool qmask = p.mask != nullptr && p.mask_qs != 0;
        static bool done_p[2] = {false, false};
        if (!done_p[qmask]) {
            if (qmask) (void)hipFuncSetAttribute((const void*)attn_bwd_pair_kernel<DK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            else (void)hipFuncSetAttribute((const void*)attn_bwd_pair_kernel<DK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done_p[qmask] = true;
        }
        if (qmask) hipLaunchKernelGGL((attn_bwd_pair_kernel<DK, true>), dim3(nblk_q + nblk_k16), dim3(512), lds, st, p, nblk_q);
        else hipLaunchKernelGGL((attn_bwd_pair_kernel<DK, false>), dim3(nblk_q + nblk_k16), dim3(512), lds, st, p, nblk_q);
        launch_bias_finish<DK>(p, st);
        BMT_CHECK_LAUNCH("bmt_attn_bwd_bf16 (dQ and dK / dV in one launch)");
        return BMT_OK;
    } else {
        if (p.q_off || p.k_off) {
            bmt_set_error("bmt_attn_bwd_bf16: packed rows need d_k >= 128");
            return BMT_EINVAL;
        }
        hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, DK, dOh);
        const int nblk_k = ((p.Sk + 63) / 64) * p.B * p.H;
        {
            const int lds_loop = 3 * 32 * DK * 2 + 256 + 128 * DK * 2, lds_epi = DK * (128 + 8) * 2;   // stage images / transposed gradient tile
            const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
            static bool done = false;
            if (!done) {
                (void)hipFuncSetAttribute((const void*)attn_bwd_dq_bf16_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                done = true;
            }
            hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DK>), dim3(nblk_q), dim3(256), lds, st, p);
        }
        {
            const int lds = 2 * 64 * DK * 2 + 4 * 32 * DK * 2 + 2 * 32 * 4;
            static bool done = false;
            if (!done) {
                (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_bf16_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                done = true;
            }
            hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DK>), dim3(nblk_k), dim3(256), lds, st, p);
        }
    }
    BMT_CHECK_LAUNCH("bmt_attn_bwd_bf16");
    return BMT_OK;
}

}  // namespace

extern "C" int bmt_attn_fwd_bf16(const bmt_attn_fwd_bf16_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && (a->O || a->Oh) && a->lse, "bmt_attn_fwd_bf16: null pointer");
    BMT_CHECK_ARG(a->B > 0 && a->H > 0 && a->Sq > 0 && a->Sk > 0, "bmt_attn_fwd_bf16: bad sizes");
    BMT_CHECK_ARG(a->dk == 32 || a->dk == 64 || a->dk == 128 || a->dk == 256, "bmt_attn_fwd_bf16: d_k=%d not in {32,64,128,256}", a->dk);
    BMT_CHECK_ARG(a->precision == BMT_PREC_BF16 || a->precision == BMT_PREC_F16 || (a->precision == BMT_PREC_BF16X3 && a->Ql && a->Kl && a->Vl),
                  "bmt_attn_fwd_bf16: precision must be BF16, F16 or BF16X3 (which needs the lo planes)");
    BMT_CHECK_ARG(!(a->Ol && a->Of), "bmt_attn_fwd_bf16: Ol and Of are alternatives (one second output plane)");
    if (!(al16(a->Qh) && al16(a->Kh) && al16(a->Vh) && al16(a->O)) || ((a->ldq | a->ldk | a->ldv | a->bsq | a->bsk | a->bsv) & 7) ||
        ((a->ldo | a->bso) & 3) || (a->Ql && !(al16(a->Ql) && al16(a->Kl) && al16(a->Vl))) ||
        (a->Oh && (!al16(a->Oh) || !al16(a->Ol) || !al16(a->Of) || ((a->ldop | a->bsop) & 7)))) {
        bmt_set_error("bmt_attn_fwd_bf16: planes must be 16-byte aligned with strides multiples of 8 elements");
        return BMT_EALIGN;
    }
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Ql = a->Ql; p.Kh = a->Kh; p.Kl = a->Kl; p.Vh = a->Vh; p.Vl = a->Vl;
    p.Ow = a->O; p.lsew = a->lse;
    p.Owh = a->Oh; p.Owl = a->Of ? a->Of : a->Ol; p.ow_f16 = a->Of != nullptr; p.ldop = a->ldop; p.bsop = a->bsop;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.SqP = a->Sq; p.SkP = a->Sk; p.q_off = a->q_off; p.k_off = a->k_off; p.b_order = a->b_order; p.kvh = a->kv_shared ? 0 : a->dk;
    BMT_CHECK_ARG(!(a->q_off || a->k_off) || (a->dk >= 128 && a->precision != BMT_PREC_BF16X3 && (!a->k_off || !a->mask) && (a->mask == nullptr || a->mask_qs == 0) &&
                                              (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) && (int64_t)a->Sk * a->ldv * 2 < (1ll << 31)),
                  "bmt_attn_fwd_bf16: packed rows (q_off / k_off) are taken by the one-pass d_k >= 128 kernels, without a mask over packed keys");
    p.scale = a->scale; p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    hipStream_t st = (hipStream_t)stream;
#define BMT_FWD(D) \
    if (a->dk == D) return a->precision == BMT_PREC_BF16X3 ? launch_fwd<D, 3>(p, st) : (a->precision == BMT_PREC_F16 ? launch_fwd<D, 1, true>(p, st) : launch_fwd<D, 1>(p, st));
    BMT_FWD(32) BMT_FWD(64) BMT_FWD(128) BMT_FWD(256)
#undef BMT_FWD
    return BMT_EINVAL;
}

extern "C" int bmt_attn_bwd_bf16(const bmt_attn_bwd_bf16_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && (a->O || a->Oh || a->Of) && a->lse && a->delta_ws && a->dOh_ws,
                  "bmt_attn_bwd_bf16: null pointer");
    BMT_CHECK_ARG((a->dQ || a->dQh) && (a->dK || a->dKh) && (a->dV || a->dVh), "bmt_attn_bwd_bf16: every gradient needs an fp32 or a plane output");
    BMT_CHECK_ARG(a->B > 0 && a->H > 0 && a->Sq > 0 && a->Sk > 0, "bmt_attn_bwd_bf16: bad sizes");
    BMT_CHECK_ARG(a->dk == 32 || a->dk == 64 || a->dk == 128 || a->dk == 256, "bmt_attn_bwd_bf16: d_k=%d not in {32,64,128,256}", a->dk);
    if (!(al16(a->Qh) && al16(a->Kh) && al16(a->Vh) && al16(a->O) && al16(a->dO) && al16(a->dQ) && al16(a->dK) && al16(a->dV) &&
          al16(a->dOh_ws) && al16(a->Oh) && al16(a->Ol) && al16(a->Of) && al16(a->dQh) && al16(a->dKh) && al16(a->dVh)) ||
        ((a->ldq | a->ldk | a->ldv | a->bsq | a->bsk | a->bsv | a->ldo | a->bso | a->dkv_ld | a->dkv_bs | a->ldop | a->bsop |
          a->gq_ld | a->gq_bs | a->gkv_ld | a->gkv_bs) & 7)) {
        bmt_set_error("bmt_attn_bwd_bf16: pointers must be 16-byte aligned with strides multiples of 8 elements");
        return BMT_EALIGN;
    }
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Kh = a->Kh; p.Vh = a->Vh; p.dOh = a->dOh_ws;
    p.O = a->O; p.Oph = a->Oh; p.Opl = a->Ol; p.Opf = a->Of; p.ldop = a->ldop; p.bsop = a->bsop;
    p.dO = a->dO; p.lse = a->lse; p.delta = a->delta_ws;
    p.gq = GradOut{a->dQ, a->ldo, a->bso, a->dQh, a->gq_ld, a->gq_bs, a->dQT, a->gqT_ld, a->dbq};
    p.gk = GradOut{a->dK, a->dkv_ld, a->dkv_bs, a->dKh, a->gkv_ld, a->gkv_bs, a->dKT, a->gkvT_ld, a->dbk};
    p.gv = GradOut{a->dV, a->dkv_ld, a->dkv_bs, a->dVh, a->gkv_ld, a->gkv_bs, a->dVT, a->gkvT_ld, a->dbv};
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.SqP = a->Sq; p.SkP = a->Sk; p.q_off = a->q_off; p.k_off = a->k_off; p.b_order = a->b_order; p.kvh = a->kv_shared ? 0 : a->dk;
    BMT_CHECK_ARG(!(a->q_off || a->k_off) || (a->dk >= 128 && (!a->k_off || !a->mask) && (a->mask == nullptr || a->mask_qs == 0) && !a->dQT && !a->dKT && !a->dVT &&
                                              !a->O && !a->dO && (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) && (int64_t)a->Sk * a->ldv * 2 < (1ll << 31) &&
                                              (int64_t)a->Sq * a->ldq * 2 < (1ll << 31) && (int64_t)a->Sq * a->ldo * 2 < (1ll << 31)),
                  "bmt_attn_bwd_bf16: packed rows (q_off / k_off) are taken by the d_k >= 128 kernels over plane operands, without a mask over packed "
                  "keys or transposed outputs");
    p.scale = a->scale; p.drop_p = a->drop_p;
    p.kmean = a->kmean;
    p.qkv_f16 = a->qkv_f16;
    BMT_CHECK_ARG(!a->qkv_f16 || a->dk >= 128, "bmt_attn_bwd_bf16: fp16 q / k / v planes are taken by the d_k >= 128 kernels only");
    if (a->bias_ws && a->dk >= 128) {      // per-tile partial column sums instead of contended atomics (every d_k >= 128 kernel), added up by a last launch
        if (!al16(a->bias_ws)) {
            bmt_set_error("bmt_attn_bwd_bf16: workspaces must be 16-byte aligned");
            return BMT_EALIGN;
        }
        const int64_t D = (int64_t)a->H * a->dk, rq = (int64_t)a->B * ((a->Sq + 127) / 128), rk = (int64_t)a->B * ((a->Sk + 127) / 128);
        if (p.gq.bsum) { p.gq.bpart = a->bias_ws; p.gq.bp_ld = D; }
        if (p.gk.bsum) { p.gk.bpart = a->bias_ws + rq * D; p.gk.bp_ld = D; }
        if (p.gv.bsum) { p.gv.bpart = a->bias_ws + (rq + rk) * D; p.gv.bp_ld = D; }
        p.defer_bias = a->defer_bias;
    }
    if (a->P_ws || a->dS_ws || a->Qb_ws) {
        BMT_CHECK_ARG(a->P_ws && a->dS_ws && a->Qb_ws && a->bias_ws, "bmt_attn_bwd_bf16: the split backward needs all four workspaces");
        int64_t n_pds, n_qb, n_bias;
        const bool takes = bmt_attn_bwd_split_ws(a->B, a->H, a->Sq, a->Sk, a->dk, &n_pds, &n_qb, &n_bias) == BMT_OK && a->qkv_f16 &&
                           (a->mask == nullptr || a->mask_qs == 0) && (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) &&
                           (int64_t)a->Sk * a->ldv * 2 < (1ll << 31) && (int64_t)a->Sq * a->ldo * 2 < (1ll << 31) && !a->dQT && !a->dKT && !a->dVT;
        if (takes) {
            if (!(al16(a->P_ws) && al16(a->dS_ws) && al16(a->Qb_ws))) {
                bmt_set_error("bmt_attn_bwd_bf16: workspaces must be 16-byte aligned");
                return BMT_EALIGN;
            }
            p.Pws = a->P_ws; p.dSws = a->dS_ws; p.Qbws = a->Qb_ws;
            p.ws_pitch = 128; p.ws_tile = (int64_t)a->Sq * 128; p.ws_slab = (int64_t)((a->Sk + 127) / 128) * p.ws_tile;
            p.ldqb = (int64_t)a->H * a->dk; p.bsqb = (int64_t)a->Sq * a->H * a->dk;
            // the live-query bits sit behind the scaled copy of q in Qb_ws (bmt_attn_bwd_split_ws sized it for them); one 64-bit mask
            // per (batch, head) in the dK / dV kernel: 64 stages of 32 queries
            p.qlive = a->Sq <= 2048 ? reinterpret_cast<int*>(a->Qb_ws + (int64_t)a->B * p.bsqb) : nullptr;
        }
    }
    if (a->rc_ws) {
        BMT_CHECK_ARG(!a->P_ws && !a->dS_ws && !a->Qb_ws && a->bias_ws, "bmt_attn_bwd_bf16: rc_ws goes with bias_ws and without the emitting form's workspaces");
        int64_t n_rc, n_bias;
        const bool takes = bmt_attn_bwd_rc_ws(a->B, a->H, a->Sq, a->Sk, a->dk, &n_rc, &n_bias) == BMT_OK && a->qkv_f16 &&
                           (a->mask == nullptr || a->mask_qs == 0) && (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) &&
                           (int64_t)a->Sk * a->ldv * 2 < (1ll << 31) && (int64_t)a->Sq * a->ldo * 2 < (1ll << 31) &&
                           (int64_t)a->Sq * a->ldq * 2 < (1ll << 31) && !a->dQT && !a->dKT && !a->dVT;
        BMT_CHECK_ARG(takes, "bmt_attn_bwd_bf16: rc_ws given for a problem the recompute form does not take (fp16 q / k / v planes, d_k 128 / 256, "
                             "64 <= Sq <= 2048, Sk <= 8192, a key-padding mask or none, no transposed outputs)");
        if ((reinterpret_cast<uintptr_t>(a->rc_ws) & 15) != 0) {
            bmt_set_error("bmt_attn_bwd_bf16: workspaces must be 16-byte aligned");
            return BMT_EALIGN;
        }
        const int64_t nqt = (a->Sq + 127) / 128;
        p.qlive = a->rc_ws;
        p.qamax = reinterpret_cast<float*>(a->rc_ws + (int64_t)a->B * a->H * nqt);
    }
    hipStream_t st = (hipStream_t)stream;
    if (a->dk == 32) return launch_bwd<32>(p, a->dOh_ws, st);
    if (a->dk == 64) return launch_bwd<64>(p, a->dOh_ws, st);
    if (a->dk == 128) return launch_bwd<128>(p, a->dOh_ws, st);
    if (a->dk == 256) return launch_bwd<256>(p, a->dOh_ws, st);
    return BMT_EINVAL;
}

extern "C" int bmt_attn_bwd_split_ws(int B, int H, int Sq, int Sk, int dk, int64_t* n_pds, int64_t* n_qb, int64_t* n_bias) {
    if (n_pds) *n_pds = 0;
    if (n_qb) *n_qb = 0;
    if (n_bias) *n_bias = 0;
    BMT_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0 && n_pds && n_qb && n_bias, "bmt_attn_bwd_split_ws: bad arguments");
    // (one 128-key block of a (batch, head) is addressed through a 32-bit byte offset; a query tile below 64 rows leaves the dQ kernel's
    // workgroups mostly idle: the decoder's 30-query attentions stay on the two-kernel form)
    // (and the dQ kernel keeps a 2-byte-per-key mask image next to its 128 KB K / V ring in LDS: Sk <= 8192)
    if (!(dk == 128 || dk == 256) || Sq < 64 || (int64_t)((Sk + 127) / 128) * Sq * 128 * 2 >= (1ll << 31) || Sk > 8192) {
        bmt_set_error("bmt_attn_bwd_split_ws: the split backward takes d_k 128 / 256, Sq >= 64, Sk <= 8192");
        return BMT_EINVAL;
    }
    const int64_t nkt = (Sk + 127) / 128, nqt = (Sq + 127) / 128;
    *n_pds = (int64_t)B * H * nkt * Sq * 128;
    *n_qb = (int64_t)B * Sq * H * dk + (((int64_t)2 * B * H * nqt + 7) & ~(int64_t)7);      // + one int of live-query bits per (batch, head, query tile)
    *n_bias = ((int64_t)B * nqt + 2 * (int64_t)B * nkt) * H * dk;
    return BMT_OK;
}

extern "C" int bmt_attn_bwd_rc_ws(int B, int H, int Sq, int Sk, int dk, int64_t* n_rc, int64_t* n_bias) {
    if (n_rc) *n_rc = 0;
    if (n_bias) *n_bias = 0;
    BMT_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0 && n_rc && n_bias, "bmt_attn_bwd_rc_ws: bad arguments");
    // (the key side keeps lse and delta of all the sample's queries in LDS behind its 128-KB stage ring and one live bit per 32-query stage in
    // a 64-bit word: Sq <= 2048; the dQ kernel keeps a 2-byte-per-key mask image next to its K / V ring: Sk <= 8192; below 64 queries the dQ
    // kernel's 128-query workgroups are mostly idle -- the decoder's 30-query attentions stay on the two-kernel form)
    if (!(dk == 128 || dk == 256) || Sq < 64 || Sq > 2048 || Sk > 8192) {
        bmt_set_error("bmt_attn_bwd_rc_ws: the recompute form takes d_k 128 / 256, 64 <= Sq <= 2048, Sk <= 8192");
        return BMT_EINVAL;
    }
    const int64_t nkt = (Sk + 127) / 128, nqt = (Sq + 127) / 128;
    *n_rc = ((int64_t)2 * B * H * nqt + 3) & ~(int64_t)3;       // live-query bits (int32) + max |dO| (fp32) per (batch, head, 128-query tile)
    *n_bias = ((int64_t)B * nqt + 2 * (int64_t)B * nkt) * H * dk;
    return BMT_OK;
}

extern "C" int64_t bmt_attn_bwd_bias_ws(int B, int H, int Sq, int Sk, int dk) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || dk < 128) return 0;
    return ((int64_t)B * ((Sq + 127) / 128) + 2 * (int64_t)B * ((Sk + 127) / 128)) * H * dk;
}

extern "C" int bmt_attn_kmean(const uint16_t* Kh, int64_t ldk, int64_t bsk, const uint8_t* mask, int64_t mask_bs, int64_t mask_qs, int B, int Sk,
                              int D, float* out, int k_f16, const int* k_off, void* stream) {
    BMT_CHECK_ARG(Kh && out && B > 0 && Sk > 0 && D > 0 && D % 8 == 0 && ldk % 8 == 0 && bsk % 8 == 0 &&
                      (reinterpret_cast<uintptr_t>(Kh) & 15) == 0,
                  "bmt_attn_kmean: bad args (D, ldk, bsk multiples of 8, 16-byte aligned plane)");
    const int ks = Sk >= 256 ? 8 : 1;
    hipLaunchKernelGGL(attn_kmean_kernel, dim3(B, bmt_cdiv(D, 128)), dim3(512), 0, (hipStream_t)stream, Kh, ldk, bsk,
                       (mask_qs == 0 && !k_off) ? mask : nullptr, mask_bs, Sk, D, out, k_f16, ks, k_off);
    BMT_CHECK_LAUNCH("bmt_attn_kmean");
    return BMT_OK;
}
