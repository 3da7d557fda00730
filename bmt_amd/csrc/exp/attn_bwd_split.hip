// EXPERIMENT (libbmt_exp.so; harness tools/probes/attn_bwd_split_check.py): the attention backward in SPLIT form.
//
// Why: the two-kernel backward (dQ per query tile, dK / dV per key tile) computes S = Q K^T and dP = dO V^T twice -- 7 products of
// Sq x Sk x d_k (8 in the two-pass dK / dV form d_k = 256 needs) for the 5 the mathematics has.  Here the dQ kernel, which has P and dS
// in registers anyway, LEAVES them in HBM workspaces (bf16, [B*H][Sq][pitch]) together with a bf16 copy of its q rows, and dK / dV are
// two plain products over them with no softmax arithmetic at all:
//     dV^T[d][key] += dO^T[d x q] . P[q x key]          dK^T[d][key] += Qb^T[d x q] . dS[q x key]
//   * attn_bwd_dq32e_kernel: attn_bwd_dq32_kernel (32 queries per wave on v_mfma_f32_32x32x16_f16, K / V by LDS-DMA into a 4-deep ring,
//     per-query power-of-two scale on the gradient operands) with ONE accumulator chain per product, the row sum of the rounded dS by
//     v_dot2c_f32_f16, and the emission: P and dS (true scale) rounded to bf16, column groups of the two half-waves exchanged by
//     v_permlane32_swap so that every lane stores 16 contiguous bytes (8 keys) per 16-key group;
//   * attn_bwd_dkvg_kernel: 4 waves x 32 keys, both gradient tiles (2 x 32 x d_k fp32 = the accumulator half of the register file),
//     loop over 32-query stages: the Qb and dO tiles (32 x d_k) and the P and dS column blocks (32 x 128 keys) come by LDS-DMA into a
//     3-deep ring (4 at d_k 128); EVERY MFMA operand is a transposing read (the reduction index q is the row of all four images); no
//     VALU work in the loop beyond address stepping.
// Lane algebra of the new operand (tools/probes/attn_bwd_split_layout.py): B[k][n = key] of the 16-query step kk: fragment element
// 4 u + j of lane (l31, hh) <-> q = 16 kk + 8 u + 4 hh + j (the same reduction-index permutation the A fragments use), read u of lane
// (hh, gi, mq, mr) addresses row 16 kk + 8 u + 4 hh + mq, bytes 64 (wave ^ mq) + 32 gi + 8 mr of the 256-byte row: 16-byte chunk
// position = chunk ^ ((row & 3) << 2), applied on the SOURCE side of the DMA -- the four rows of a read fall into the four 64-byte bank
// quarters.
#ifndef BMT_EXP_LIB
#include "../attention_bf16.hip"
#include "attn_bwd32.hip"
#endif

namespace {

template <int DK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq32e_kernel(const AttnPB p, float* kq_out) {
    constexpr int BC = 32, NT = 256, KS = DK / 16, DT = DK / 32, ROWB = DK * 2, TILE = BC * ROWB, STAGE = 2 * TILE, NS = 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NP = BC / RPP, PPW = NP / 4;
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    static_assert(PPW <= 4, "pieces per wave");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sMask = smem + NS * STAGE;                                     // [ntile * 32] bytes: 1 = valid key

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l31 = lane & 31;
    const int nqt = (p.Sq + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nqt * p.B * p.H);
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;
    const bool wave_on = qt * 128 + wid * 32 < p.Sq;
    const int ntile = (p.Sk + BC - 1) / BC;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sk - 1) * p.ldk + DK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sk - 1) * p.ldv + DK) * 2), 0x00020000);
    // the (batch, head) slab of the P / dS workspaces: rows past Sq fall outside the descriptor and are dropped
    // (the descriptor covers the Sq rows of ONE 128-key tile block; the block's offset travels in the soffset, which the range check ignores)
    const int64_t slab = (int64_t)bh * p.ws_slab;
    // (the descriptor covers the whole (batch, head) slab: the range check takes the soffset into account -- raw buffers are out of range at
    // voffset >= num_records - soffset -- so a one-block range dropped every store to key blocks past the first; rows past Sq are kept
    // out by their voffset instead)
    const int slab_bytes = (int)(p.ws_slab * 2);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Pws + slab), 0, slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dSws + slab), 0, slab_bytes, 0x00020000);
    const int ws_tile2 = (int)p.ws_tile * 2;
    const int wvo = qok ? (int)(((int64_t)q * p.ws_pitch + 8 * hh) * 2) : 0x7fffff00;
    int kvo[4], vvo[4];      // (fixed extent: see attn_fwd32_kernel)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wid * PPW + j) * RPP + lane / CPR, cpos = lane % CPR;
        kvo[j] = row * (int)p.ldk * 2 + ((cpos ^ kswz(row)) * 16);
        vvo[j] = row * (int)p.ldv * 2 + ((cpos ^ (row & 15)) * 16);
    }
    const int sstep_k = BC * (int)p.ldk * 2, sstep_v = BC * (int)p.ldv * 2;
#define BMT_E_DMA_K(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + (slot_) * STAGE + (wid * PPW + (j_)) * 1024), 16, kvo[j_], (t_) * sstep_k, 0, 0)
#define BMT_E_DMA_V(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + (slot_) * STAGE + TILE + (wid * PPW + (j_)) * 1024), 16, vvo[j_], (t_) * sstep_v, 0, 0)

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        const int tl = min(s, ntile - 1);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_E_DMA_K(j, tl, s);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_E_DMA_V(j, tl, s);
    }
    bf16x8 qf[KS];
    u32x4 dob[KS];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * hh;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = ldfrag(p.Qh + qo + 16 * ks, qok);
            dob[ks] = __builtin_bit_cast(u32x4, ldfrag(p.dOh + oo + 16 * ks, qok));
        }
        if (qok) {        // the bf16 copy of this lane's q values: the A operand of dK^T = Qb^T . dS in attn_bwd_dkvg_kernel
            uint16_t* qb = p.Qbws + (int64_t)b * p.bsqb + (int64_t)q * p.ldqb + h * DK + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<u32x4*>(qb + 16 * ks) = __builtin_bit_cast(u32x4, h8_to_b8(qf[ks]));
        }
    }
    for (int i = tid; i < ntile * BC; i += NT) {
        uint8_t m = 0;
        if (i < p.Sk) m = (p.mask != nullptr) ? (uint8_t)(p.mask[(int64_t)b * p.mask_bs + i] != 0) : (uint8_t)1;
        sMask[i] = m;
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.Sq + q;
    const float lse2 = qok ? p.lse[stat] * LOG2E : 0.f;
    const float delta = qok ? p.delta[stat] : 0.f;
    float amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bfbits_lo(dob[ks][j])), fabsf(bfbits_hi(dob[ks][j]))));
    amax = half_max(amax);
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
    // dS' = P ((dP' - delta') scale): dsc = -delta' scale in the fma
    const float dsc = -delta * up * p.scale;
    if (kq_out != nullptr && qok && hh == 0) kq_out[stat] = amax > 0.f ? up : 3.0e38f;
    const float sc2 = p.scale * LOG2E;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    float rs = 0.f;          // sum of the ROUNDED dS' this lane fed to the MFMAs (mean-key correction)
    const h2_t ones = {(_Float16)1.f, (_Float16)1.f};

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int fk = kswz(l31), s15 = l31 & 15;
    const uint32_t kA0 = lds0 + l31 * ROWB + 32 * (fk >> 1) + 16 * (hh ^ (fk & 1));
    const uint32_t vA0 = lds0 + TILE + l31 * ROWB + 32 * (s15 >> 1) + 16 * (hh ^ (s15 & 1));
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t kT0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int slot = t % NS, slotn = (t + NS - 1) % NS;
        const int tn = min(t + NS - 1, ntile - 1);
        const int key0 = t * BC;
        uint32_t mw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mw[i] = *reinterpret_cast<const uint32_t*>(sMask + key0 + 8 * i + 4 * hh);
        const bool none_valid = __all((mw[0] | mw[1] | mw[2] | mw[3]) == 0u);
        const bool all_valid = __all((mw[0] & mw[1] & mw[2] & mw[3]) == 0x01010101u);
        if (none_valid || !wave_on) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_E_DMA_K(j, tn, slotn);
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_E_DMA_V(j, tn, slotn);
        } else {
            const uint32_t kA = kA0 + slot * STAGE, vA = vA0 + slot * STAGE, kT = kT0 + slot * STAGE;
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
            u32x4 kf[3];
            kf[0] = lds_b128<0>(kA);
            kf[1] = lds_b128<0>(kA ^ (1 << 5));
#define BMT_E_SSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) kf[((ks_) + 2) % 3] = lds_b128<0>(kA ^ (((ks_) + 2) << 5));      \
        if constexpr ((ks_) < PPW) BMT_E_DMA_K((ks_) % PPW, tn, slotn);                                \
        else if constexpr ((ks_) < 2 * PPW) BMT_E_DMA_V((ks_) % PPW, tn, slotn);                       \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(kf[(ks_) % 3]);                             \
        st = mfma32t<true>(as_bf16x8(kf[(ks_) % 3]), qf[(ks_)], st);                                   \
    }
            BMT_X_REP16(BMT_E_SSTEP)
#undef BMT_E_SSTEP
            float pr[16];
            u32x4 vf[3];
            vf[0] = lds_b128<0>(vA);
            vf[1] = lds_b128<0>(vA ^ (1 << 5));
#define BMT_E_PSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) vf[((ks_) + 2) % 3] = lds_b128<0>(vA ^ (((ks_) + 2) << 5));      \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(vf[(ks_) % 3]);                             \
        dp = mfma32t<true>(as_bf16x8(vf[(ks_) % 3]), dof[(ks_)], dp);                                  \
        if constexpr ((ks_) < 16 && (ks_) >= 2) {                                                      \
            const float pe_ = __builtin_amdgcn_exp2f(__builtin_fmaf(st[(ks_) - 2], sc2, -lse2));       \
            pr[(ks_) - 2] = (all_valid || ((mw[((ks_) - 2) >> 2] >> (8 * (((ks_) - 2) & 3))) & 0xffu)) ? pe_ : 0.f; \
        }                                                                                              \
    }
            BMT_X_REP16(BMT_E_PSTEP)
#undef BMT_E_PSTEP
#pragma unroll
            for (int r = (KS < 16 ? KS - 2 : 14); r < 16; ++r) {
                const float pe_ = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], sc2, -lse2));
                pr[r] = (all_valid || ((mw[r >> 2] >> (8 * (r & 3))) & 0xffu)) ? pe_ : 0.f;
            }
            // ---- dS' = P (dP' - delta') scale, rounded to fp16 (clamped: an overflow must not become inf); the true-scale dS for the
            // workspace: dS' 2^-k
            bf16x8 dsf[2];
            uint32_t sw[8];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint32_t dwv[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int r0 = 8 * kk + 2 * j2;
                    float a0 = pr[r0] * __builtin_fmaf(dp[r0], p.scale, dsc);
                    float a1 = pr[r0 + 1] * __builtin_fmaf(dp[r0 + 1], p.scale, dsc);
                    sw[4 * kk + j2] = pack_bf2(a0 * down, a1 * down);
                    a0 = __builtin_amdgcn_fmed3f(a0, -60000.f, 60000.f);
                    a1 = __builtin_amdgcn_fmed3f(a1, -60000.f, 60000.f);
                    dwv[j2] = pack_h2(a0, a1);
                    rs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, dwv[j2]), ones, rs, false);
                }
                dsf[kk] = as_bf16x8(u32x4{dwv[0], dwv[1], dwv[2], dwv[3]});
            }
            // ---- emission: registers 4 i + j = key 8 i + 4 hh + j; after the half exchange the lower half-wave holds keys 8 i .. 8 i + 7
            // of group i, the upper one those of group i + 1 (i even): 16 contiguous bytes per lane
            {
                uint32_t pw[8];
#pragma unroll
                for (int j2 = 0; j2 < 8; ++j2) pw[j2] = pack_bf2(pr[2 * j2], pr[2 * j2 + 1]);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    swap32u(pw[4 * g + 0], pw[4 * g + 2]);
                    swap32u(pw[4 * g + 1], pw[4 * g + 3]);
                    swap32u(sw[4 * g + 0], sw[4 * g + 2]);
                    swap32u(sw[4 * g + 1], sw[4 * g + 3]);
                }
                const int wso = (t >> 2) * ws_tile2 + (t & 3) * 64;
                st128<0>(rsP, pw[0], pw[1], pw[2], pw[3], wvo, wso);
                st128<32>(rsP, pw[4], pw[5], pw[6], pw[7], wvo, wso);
                st128<0>(rsS, sw[0], sw[1], sw[2], sw[3], wvo, wso);
                st128<32>(rsS, sw[4], sw[5], sw[6], sw[7], wvo, wso);
            }
            // ---- dQ'^T += K^T . dS'^T
            u32x2 ta[3], tb[3];
#define BMT_E_TFRAG(n_)                                                                   \
    do {                                                                                  \
        ta[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1)) * ROWB>(kT ^ (((n_) >> 1) << 6));     \
        tb[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1) + 8) * ROWB>(kT ^ ((((n_) >> 1) << 6) | 32)); \
    } while (0)
            BMT_E_TFRAG(0);
            BMT_E_TFRAG(1);
#define BMT_E_QSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + 2 < 2 * DT) BMT_E_TFRAG((n_) + 2);                                        \
        lgkm_wait<((n_) + 2 < 2 * DT) ? 4 : 2 * (2 * DT - 1 - (n_))>(ta[(n_) % 3], tb[(n_) % 3]);      \
        const u32x4 av = {ta[(n_) % 3][0], ta[(n_) % 3][1], tb[(n_) % 3][0], tb[(n_) % 3][1]};         \
        dq[(n_) >> 1] = mfma32t<true>(as_bf16x8(av), dsf[(n_) & 1], dq[(n_) >> 1]);                    \
    }
            BMT_X_REP16(BMT_E_QSTEP)
#undef BMT_E_QSTEP
#undef BMT_E_TFRAG
        }
        // (loads only are counted: a store younger than the DMA requests can only make the wait longer, never shorter)
        if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BMT_B_BAR();
    }
#undef BMT_E_DMA_K
#undef BMT_E_DMA_V

    if (p.kmean != nullptr) {
        const float rst = half_sum(rs);
        const float* km = p.kmean + ((int64_t)b * p.H + h) * DK;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] -= rst * km[dt * 32 + acc_row(r, hh)];
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] *= down;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    grad_store_rows<DK>(p.gq, dq, b, h, q, qok, hh);
    if (p.gq.hiT || p.gq.bsum) {
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
        grad_tile_write<DK, 128>(tile, dq, wid * 32, qok, l31, hh);
        __syncthreads();
        grad_tile_flush<DK, 128>(tile, p.gq, b, h, qt * 128, p.Sq, tid);
    }
}

template <int DK>
int launch_dq32e(const AttnPB& p, float* kq_out, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds_loop = 4 * 2 * 32 * DK * 2 + ((ntile * 32 + 15) & ~15), lds_epi = DK * (128 + 8) * 2;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq32e_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dq32e_kernel<DK>), dim3(nblk), dim3(256), lds, st, p, kq_out);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(dq)");
    return BMT_OK;
}

// ------------------------------------------------------------------------------------------------------------ dK / dV as two plain products
// XP (timing probes only): bit 0 = no DMA inside the loop, bit 1 = no MFMA, bit 2 = no P / dS DMA inside the loop
template <int DK, int XP = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkvg_kernel(const AttnPB p) {
    constexpr int BQ = 32, DT = DK / 32, ROWB = DK * 2, XT = BQ * ROWB, YT = BQ * 256, STAGE = 2 * XT + 2 * YT;
    constexpr int NS = (DK == 256) ? 3 : 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NPX = BQ / RPP, PPW = NPX / 4;      // X tiles: 16 (8) pieces of 1 KB; Y tiles: 8 pieces
    constexpr int NDMA = 2 * PPW + 4;                                               // requests per wave and stage
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int nst = (p.Sq + BQ - 1) / BQ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l31 = lane & 31;
    const int nkt = (p.Sk + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nkt * p.B * p.H);
    const int kt = w % nkt, bh = w / nkt;
    const int b = bh / p.H, h = bh % p.H;
    const int key = kt * 128 + wid * 32 + l31;
    const bool kin = key < p.Sk;
    const bool kok = kin && (p.mask == nullptr || p.mask[(int64_t)b * p.mask_bs + key] != 0);
    // a key tile without a valid key (ragged lengths: a quarter of them at configs[1]) writes zeros and leaves; inside a tile every wave
    // runs the whole loop (a per-wave skip around the two 128-register accumulator sets made hipcc 7.2 shuttle them between the register
    // files: 500 v_accvgpr moves per stage)
    const int nst_run = (__syncthreads_or((int)kok) && !(XP & 16)) ? nst : 0;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Qbws + (int64_t)b * p.bsqb + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sq - 1) * p.ldqb + DK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dOh + (int64_t)b * p.bso + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sq - 1) * p.ldo + DK) * 2), 0x00020000);
    const int64_t slab = (int64_t)bh * p.ws_slab + kt * p.ws_tile;
    const int slab_bytes = (int)(((int64_t)p.Sq * p.ws_pitch - (p.ws_tile == 128 ? kt * 128 : 0)) * 2);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Pws + slab), 0, slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dSws + slab), 0, slab_bytes, 0x00020000);
    // per-lane byte offsets of the NEXT tile to request (stepped through the voffset: the descriptor's range check covers it, rows
    // past Sq read as zero; an soffset would bypass the check)
    int xq[4], xo[4], yo[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid * PPW + (j % PPW)) * RPP + lane / CPR, cpos = lane % CPR;
        xq[j] = row * (int)p.ldqb * 2 + ((cpos ^ kswz(row)) * 16);
        xo[j] = row * (int)p.ldo * 2 + ((cpos ^ kswz(row)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wid * 2 + j) * 4 + lane / 16, cpos = lane % 16;
        yo[j] = row * (int)p.ws_pitch * 2 + ((cpos ^ ((row & 3) << 2)) * 16);
    }
    const int sstep_q = BQ * (int)p.ldqb * 2, sstep_o = BQ * (int)p.ldo * 2, sstep_y = BQ * (int)p.ws_pitch * 2;
#define BMT_G_DMA(i_, slot_)                                                                                                              \
    do {                                                                                                                                  \
        if constexpr ((i_) < PPW)                                                                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + (slot_) * STAGE + (wid * PPW + (i_)) * 1024), 16, xq[(i_) % 4], 0, 0, 0); \
        else if constexpr ((i_) < 2 * PPW)                                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + (slot_) * STAGE + XT + (wid * PPW + (i_) - PPW) * 1024), 16, xo[((i_) - PPW) % 4], 0, 0, 0); \
        else if constexpr ((i_) < 2 * PPW + 2)                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lptr_t)(smem + (slot_) * STAGE + 2 * XT + (wid * 2 + (i_) - 2 * PPW) * 1024), 16, yo[((i_) - 2 * PPW) % 2], 0, 0, 0); \
        else if constexpr ((i_) < 2 * PPW + 4)                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lptr_t)(smem + (slot_) * STAGE + 2 * XT + YT + (wid * 2 + (i_) - 2 * PPW - 2) * 1024), 16, yo[((i_) - 2 * PPW - 2) % 2], 0, 0, 0); \
    } while (0)
#define BMT_G_ADVANCE()                                                       \
    do {                                                                      \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) { xq[j] += sstep_q; xo[j] += sstep_o; } \
        yo[0] += sstep_y; yo[1] += sstep_y;                                   \
    } while (0)
#define BMT_G_DMA_ALL(slot_)                                                                                                      \
    do {                                                                                                                          \
        BMT_G_DMA(0, slot_); BMT_G_DMA(1, slot_); BMT_G_DMA(2, slot_); BMT_G_DMA(3, slot_); BMT_G_DMA(4, slot_); BMT_G_DMA(5, slot_); \
        BMT_G_DMA(6, slot_); BMT_G_DMA(7, slot_); BMT_G_DMA(8, slot_); BMT_G_DMA(9, slot_); BMT_G_DMA(10, slot_); BMT_G_DMA(11, slot_); \
    } while (0)

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        BMT_G_DMA_ALL(s);
        BMT_G_ADVANCE();
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    // transposing reads of the X tiles: base of d-tile residue j (dt & 3) and row block u; d-tile dt >> 2, the 16-query step and u's
    // 8 rows enter as immediates
    const uint32_t xT0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1);
    uint32_t xa[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xa[j][0] = xT0 ^ (j << 6);
        xa[j][1] = xT0 ^ ((j << 6) | 32);
    }
    const uint32_t yB0 = lds0 + 2 * XT + (4 * hh + mq) * 256 + 64 * (wid ^ mq) + 32 * gi + 8 * mr;

    f32x16 dka[DT], dva[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
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
            for (int j = 0; j < 4; ++j) { xs[j][0] = xa[j][0] + so; xs[j][1] = xa[j][1] + so; }
            const uint32_t yB = yB0 + so;
            // ---- B fragments of the stage: P and dS, 16-query steps kk = 0, 1 (two transposing reads each)
            u32x2 yr[8];
            yr[0] = lds_tr_b64<0 * 256>(yB);           yr[1] = lds_tr_b64<8 * 256>(yB);
            yr[2] = lds_tr_b64<16 * 256>(yB);          yr[3] = lds_tr_b64<24 * 256>(yB);
            yr[4] = lds_tr_b64<YT + 0 * 256>(yB);      yr[5] = lds_tr_b64<YT + 8 * 256>(yB);
            yr[6] = lds_tr_b64<YT + 16 * 256>(yB);     yr[7] = lds_tr_b64<YT + 24 * 256>(yB);
            // step m: 16-query step kk = m / (2 DT), d-tile (m % (2 DT)) >> 1, m & 1 = 0: dV from the dO tile, 1: dK from the Qb tile -- an
            // accumulator comes round again after 2 DT MFMAs; fragment reads PF steps ahead (one wave per SIMD: nothing else hides LDS latency)
            constexpr int PF = 4, RR = PF + 1;
            u32x2 ta[RR], tb[RR];
#define BMT_G_TFRAG(m_)                                                                                                   \
    do {                                                                                                                  \
        constexpr int kk__ = (m_) / (2 * DT), dt__ = ((m_) % (2 * DT)) >> 1;                                              \
        constexpr int off__ = (((m_) & 1) ? 0 : XT) + (DK == 256 ? (dt__ >> 2) * 256 : 0) + 16 * kk__ * ROWB;             \
        ta[(m_) % RR] = lds_tr_b64<off__>(xs[dt__ & 3][0]);                                                               \
        tb[(m_) % RR] = lds_tr_b64<off__ + 8 * ROWB>(xs[dt__ & 3][1]);                                                    \
    } while (0)
            BMT_G_TFRAG(0); BMT_G_TFRAG(1); BMT_G_TFRAG(2); BMT_G_TFRAG(3);
            lgkm_wait<2 * PF>(yr[0], yr[1]); lgkm_wait<2 * PF>(yr[2], yr[3]); lgkm_wait<2 * PF>(yr[4], yr[5]); lgkm_wait<2 * PF>(yr[6], yr[7]);
            if (t == nst - 1 && (p.Sq & 31) != 0) {       // rows past Sq must not contribute whatever the ring holds there
                const int rem = p.Sq - t * BQ;
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
#define BMT_G_STEP(m_)                                                                                     \
    if constexpr ((m_) < 4 * DT) {                                                                         \
        if constexpr ((m_) + PF < 4 * DT) BMT_G_TFRAG((m_) + PF);                                          \
        if constexpr ((m_) < NDMA && !(XP & 1) && !((XP & 4) && (m_) >= 2 * PPW)) BMT_G_DMA((m_), slotn);  \
        lgkm_wait<((m_) + PF < 4 * DT) ? 2 * PF : 2 * (4 * DT - 1 - (m_))>(ta[(m_) % RR], tb[(m_) % RR]);  \
        const u32x4 av = {ta[(m_) % RR][0], ta[(m_) % RR][1], tb[(m_) % RR][0], tb[(m_) % RR][1]};         \
        if constexpr (XP & 2) dva[0][0] += __uint_as_float(av[0] ^ av[3]);                                 \
        else if constexpr (((m_) & 1) == 0) dva[((m_) % (2 * DT)) >> 1] = mfma32t<false>(as_bf16x8(av), pf[(m_) / (2 * DT)], dva[((m_) % (2 * DT)) >> 1]); \
        else dka[((m_) % (2 * DT)) >> 1] = mfma32t<false>(as_bf16x8(av), sf[(m_) / (2 * DT)], dka[((m_) % (2 * DT)) >> 1]); \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define BMT_G_STEP2(i_) BMT_G_STEP(2 * (i_)) BMT_G_STEP(2 * (i_) + 1)
            BMT_X_REP16(BMT_G_STEP2)
#undef BMT_G_STEP2
#undef BMT_G_STEP
#undef BMT_G_TFRAG
        }
        BMT_G_ADVANCE();
        if constexpr (NDMA == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BMT_B_BAR();
    }
#undef BMT_G_DMA_ALL
#undef BMT_G_ADVANCE
#undef BMT_G_DMA

    if (!kok) {      // a masked key's columns of P / dS were never written where the dQ kernel skipped its (fully masked) 32-key tile
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dka[dt][r] = 0.f; dva[dt][r] = 0.f; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
    if constexpr (XP & 8) { if (dva[0][0] == 1234.5f && dka[1][1] == 3.25f) p.gv.bsum[0] = 1.f; return; }
    grad_rm_epilogue<DK, DT, 256>(smem, p.gv, dva, b, h, kt * 128, wid * 32 + l31, kin, hh, 0, p.Sk, tid);
    grad_rm_epilogue<DK, DT, 256>(smem, p.gk, dka, b, h, kt * 128, wid * 32 + l31, kin, hh, 0, p.Sk, tid);
    (void)tile;
}

template <int DK, int XP = 0>
int launch_dkvg(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sk + 127) / 128) * p.B * p.H;
    const int NS = DK == 256 ? 3 : 4;
    const int lds_loop = NS * (2 * 32 * DK * 2 + 2 * 32 * 256), lds_epi = 128 * (DK + 8) * 2 + 256 * 8;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkvg_kernel<DK, XP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dkvg_kernel<DK, XP>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(dkv)");
    return BMT_OK;
}

template <int NTL>
__device__ __forceinline__ void grad_store_part(const GradOut& g, const f32x16 (&acc)[NTL], int DK, int b, int h, int tok, bool ok, int half, int dt0) {
    if (!ok) return;
    if (g.f32) {
        float* dst = g.f32 + (int64_t)b * g.f_bs + (int64_t)tok * g.f_ld + h * DK;
#pragma unroll
        for (int dt = 0; dt < NTL; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4*>(dst + (dt0 + dt) * 32 + 8 * r4 + 4 * half) =
                    make_float4(acc[dt][4 * r4 + 0], acc[dt][4 * r4 + 1], acc[dt][4 * r4 + 2], acc[dt][4 * r4 + 3]);
    }
    if (g.hi) {
        uint16_t* dst = g.hi + (int64_t)b * g.h_bs + (int64_t)tok * g.h_ld + h * DK;
#pragma unroll
        for (int dt = 0; dt < NTL; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                u32x2 v;
                v[0] = pack_bf2(acc[dt][4 * r4 + 0], acc[dt][4 * r4 + 1]);
                v[1] = pack_bf2(acc[dt][4 * r4 + 2], acc[dt][4 * r4 + 3]);
                *reinterpret_cast<u32x2*>(dst + (dt0 + dt) * 32 + 8 * r4 + 4 * half) = v;
            }
    }
}
template <int NTL, int NTOK>
__device__ __forceinline__ void grad_tile_write_part(uint16_t* tile, const f32x16 (&acc)[NTL], int tc, bool ok, int l31, int half, int dt0) {
    constexpr int TS = NTOK + 8;
#pragma unroll
    for (int dt = 0; dt < NTL; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const __bf16 hv = (__bf16)acc[dt][r];
            tile[((dt0 + dt) * 32 + acc_row(r, half)) * TS + tc + l31] = ok ? __builtin_bit_cast(uint16_t, hv) : (uint16_t)0;
        }
}

int launch_dkvg_probe(const AttnPB& p, int dk, hipStream_t st) {
    if (dk == 128) return launch_dkvg<128>(p, st);
    switch (p.xp) {
        case 1: return launch_dkvg<256, 1>(p, st);
        case 2: return launch_dkvg<256, 2>(p, st);
        case 3: return launch_dkvg<256, 3>(p, st);
        case 4: return launch_dkvg<256, 4>(p, st);
        case 6: return launch_dkvg<256, 6>(p, st);
        case 8: return launch_dkvg<256, 8>(p, st);
        case 16: return launch_dkvg<256, 16>(p, st);
        case 24: return launch_dkvg<256, 24>(p, st);
        default: return launch_dkvg<256>(p, st);
    }
}

}  // namespace

// the attention backward in split form, drop-in for bmt_attn_bwd_bf16 (same argument block) plus the workspaces: Pws / dSws bf16
// [B*H][Sq][pitch] (pitch a multiple of 32, >= Sk rounded up to 32), Qbws bf16 [B][Sq][H * d_k].  which: bit 0 the product's delta kernel,
// bit 1 the dQ kernel, bit 2 the dK / dV kernel, bit 3 the pipelined dQ kernel (the harness times them one by one).
extern "C" int bmt_exp_attn_bwd_split(const bmt_attn_bwd_bf16_args* a, uint16_t* Pws, uint16_t* dSws, uint16_t* Qbws, int64_t pitch, int which,
                                      float* bias_ws, void* stream) {
    // bias_ws (optional): fp32 [(B * ceil(Sq / 128) + 2 * B * ceil(Sk / 128)) * H * d_k]: per-tile column sums, added up by a finishing launch
    // which bits 8.. : bit 8 = tile-major workspaces ([bh][key tile of 128][Sq][128]: `pitch` ignored), bits 12-14 = loop probes of the dK / dV kernel
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && (a->O || a->Oh || a->Of) && a->lse && a->delta_ws && a->dOh_ws && Pws && dSws && Qbws,
                  "bmt_exp_attn_bwd_split: null pointer");
    BMT_CHECK_ARG(a->qkv_f16 && (a->dk == 128 || a->dk == 256), "bmt_exp_attn_bwd_split: fp16 q / k / v planes, d_k 128 / 256");
    BMT_CHECK_ARG(a->mask == nullptr || a->mask_qs == 0, "bmt_exp_attn_bwd_split: key-padding masks only");
    BMT_CHECK_ARG(pitch % 32 == 0 && pitch >= ((a->Sk + 31) / 32) * 32 && (int64_t)a->Sq * pitch * 2 < (1ll << 31), "bmt_exp_attn_bwd_split: bad pitch");
    BMT_CHECK_ARG((int64_t)a->Sk * a->ldk * 2 < (1ll << 31) && (int64_t)a->Sk * a->ldv * 2 < (1ll << 31) && (int64_t)a->Sq * a->ldo * 2 < (1ll << 31),
                  "bmt_exp_attn_bwd_split: sequence too long");
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Kh = a->Kh; p.Vh = a->Vh; p.dOh = a->dOh_ws;
    p.O = a->O; p.Oph = a->Oh; p.Opl = a->Ol; p.Opf = a->Of; p.ldop = a->ldop; p.bsop = a->bsop;
    p.dO = a->dO; p.lse = a->lse; p.delta = a->delta_ws;
    p.gq = GradOut{a->dQ, a->ldo, a->bso, a->dQh, a->gq_ld, a->gq_bs, a->dQT, a->gqT_ld, a->dbq, nullptr, 0};
    p.gk = GradOut{a->dK, a->dkv_ld, a->dkv_bs, a->dKh, a->gkv_ld, a->gkv_bs, a->dKT, a->gkvT_ld, a->dbk, nullptr, 0};
    p.gv = GradOut{a->dV, a->dkv_ld, a->dkv_bs, a->dVh, a->gkv_ld, a->gkv_bs, a->dVT, a->gkvT_ld, a->dbv, nullptr, 0};
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p;
    p.kmean = a->kmean;
    p.qkv_f16 = 1;
    p.Pws = Pws; p.dSws = dSws; p.Qbws = Qbws;
    const int64_t nkt128 = (a->Sk + 127) / 128;
    if (which & 256) { p.ws_pitch = 128; p.ws_tile = (int64_t)a->Sq * 128; p.ws_slab = nkt128 * p.ws_tile; }
    else { p.ws_pitch = pitch; p.ws_tile = 128; p.ws_slab = (int64_t)a->Sq * pitch; }
    p.xp = (which >> 12) & 31;
    p.ldqb = (int64_t)a->H * a->dk; p.bsqb = (int64_t)a->Sq * a->H * a->dk;
    hipStream_t st = (hipStream_t)stream;
    const int D = a->H * a->dk, rq = a->B * ((a->Sq + 127) / 128), rk = a->B * ((a->Sk + 127) / 128);
    if (bias_ws) {
        if (p.gq.bsum) { p.gq.bpart = bias_ws; p.gq.bp_ld = D; }
        if (p.gk.bsum) { p.gk.bpart = bias_ws + (int64_t)rq * D; p.gk.bp_ld = D; }
        if (p.gv.bsum) { p.gv.bpart = bias_ws + (int64_t)(rq + rk) * D; p.gv.bp_ld = D; }
    }
    if (which & 1) {
        const int64_t rows = (int64_t)a->B * a->H * a->Sq;
        hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, a->dk, a->dOh_ws);
        BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(delta)");
    }
    if (which & 2) {
        const int rc = a->dk == 256 ? launch_dq32e<256>(p, nullptr, st) : launch_dq32e<128>(p, nullptr, st);
        if (rc != BMT_OK) return rc;
    }
    if (which & 8) {      // the pipelined dQ kernel; without bit 0 it computes delta itself (needs the saved output as a plane and dO as the bf16 plane)
        p.fuse_delta = (which & 1) ? 0 : 1;
        BMT_CHECK_ARG(!p.fuse_delta || (a->dO == nullptr && (a->Of || a->Oh)), "bmt_exp_attn_bwd_split: the fused delta reads planes");
        const int rc = a->dk == 128 ? launch_dq32p<128>(p, st) : (p.xp == 8 ? launch_dq32p<256, 8>(p, st) : (p.xp == 16 ? launch_dq32p<256, 16>(p, st) : (p.xp == 24 ? launch_dq32p<256, 24>(p, st) : launch_dq32p<256>(p, st))));
        if (rc != BMT_OK) return rc;
    }
    int rc = BMT_OK;
    if (which & 16) rc = a->dk == 256 ? launch_dkvg8<256>(p, st) : launch_dkvg8<128>(p, st);      // the 8-wave dK / dV kernel
    else if (which & 4) rc = launch_dkvg_probe(p, a->dk, st);
    if (rc != BMT_OK) return rc;
    if (bias_ws && (which & (4 | 8 | 16))) {
        const bool qd = (which & 8) != 0, kd = (which & (4 | 16)) != 0;
        hipLaunchKernelGGL(attn_bias_finish_kernel, dim3(bmt_cdiv(D, 256), 32, 3), dim3(256), 0, st, qd ? p.gq.bpart : nullptr, rq, p.gq.bsum,
                           kd ? p.gk.bpart : nullptr, kd ? p.gv.bpart : nullptr, rk, p.gk.bsum, p.gv.bsum, D);
        BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(bias)");
    }
    return BMT_OK;
}
