// EXPERIMENT (built by exp/build.sh into bmt_amd/lib/libbmt_exp.so, never loaded by the product; NOT yet run on a GPU -- written at the
// end of round 2 after the GPU budget was spent, lane algebra checked on the CPU by tools/probes/attn_bwd32_layout.py, harness
// tools/probes/attn_bwd32_check.py): the attention backward's dQ kernel on the recipe of attn_fwd32_kernel, DESIGN.md section 7 item 1.
//
//   * 32 queries per wave on v_mfma_f32_32x32x16_f16, ONE wave per SIMD (the whole 512-register budget): Q and dO fragments (64 + 64
//     registers) and the 32 x d_k accumulator (128, no VALU touches it inside the loop) do not leave room for a second wave;
//   * K / V tiles of 32 keys by LDS-DMA into a 4-deep ring (128 KB, one workgroup per CU), three tiles in flight, counted vmcnt and a
//     raw s_barrier per stage;
//   * the K image serves both read patterns (row fragments for S^T = K . Q^T, transposing reads for dQ^T += K^T . dS^T): 16-byte chunk
//     position = chunk ^ ((row & 3) << 2 | (row >> 2) & 3); the V image (rows only, dP^T = V . dO^T): chunk ^ (row & 15);
//   * q / k / v exist as fp16 planes only (an LDS-DMA cannot convert): every product runs on fp16 MFMAs.  S is computed exactly as the
//     forward computes it.  The gradient operands carry a per-query power-of-two scale 2^k(q), k(q) = 6 - floor(log2 max|dO(q, :)|):
//     dO' = dO 2^k in fp16 (|dO'| < 128), dP' = V . dO', delta' = delta 2^k, dS' = P (dP' - delta') scale -- all linear in 2^k --
//     and dQ = (K^T . dS') 2^-k in the epilogue.  11 significand bits on every backward product instead of bf16's 8;
//   * the element-wise work is dealt out between the MFMAs by hand (the asm reads pin the instruction order): the probability of
//     element r right after dP's MFMA r.
// Same lane algebra as the forward: S^T / dP^T register 4 i + j of lane (l31, hh) = key 8 i + 4 hh + j, query l31; registers
// 8 kk .. 8 kk + 7 of dS' are the B operand of 16-key MFMA kk of the dQ product.
#ifndef BMT_EXP_LIB      // exp_lib.hip compiles the experiment files as ONE translation unit over one copy of the product file
#include "../attention_bf16.hip"
#endif

namespace {

__device__ __forceinline__ int kswz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ float bfbits_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfbits_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

#define BMT_B_BAR()                              \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

template <int DK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq32_kernel(const AttnPB p) {
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
    int kvo[4], vvo[4];      // (fixed extent: see attn_fwd32_kernel)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wid * PPW + j) * RPP + lane / CPR, cpos = lane % CPR;
        kvo[j] = row * (int)p.ldk * 2 + ((cpos ^ kswz(row)) * 16);
        vvo[j] = row * (int)p.ldv * 2 + ((cpos ^ (row & 15)) * 16);
    }
    const int sstep_k = BC * (int)p.ldk * 2, sstep_v = BC * (int)p.ldv * 2;
#define BMT_B_DMA_K(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + (slot_) * STAGE + (wid * PPW + (j_)) * 1024), 16, kvo[j_], (t_) * sstep_k, 0, 0)
#define BMT_B_DMA_V(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + (slot_) * STAGE + TILE + (wid * PPW + (j_)) * 1024), 16, vvo[j_], (t_) * sstep_v, 0, 0)

    // ---- prologue: tiles 0, 1, 2 in flight (a tile index past the end re-fetches the last tile: harmless, uniform counts), then the
    // mask row, Q, dO (row maximum -> scale -> fp16), lse, delta
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        const int tl = min(s, ntile - 1);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_B_DMA_K(j, tl, s);
#pragma unroll
        for (int j = 0; j < PPW; ++j) BMT_B_DMA_V(j, tl, s);
    }
    bf16x8 qf[KS];
    u32x4 dob[KS];
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)q * p.ldq + h * DK + 8 * hh;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)q * p.ldo + h * DK + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = ldfrag(p.Qh + qo + 16 * ks, qok);
            dob[ks] = __builtin_bit_cast(u32x4, ldfrag(p.dOh + oo + 16 * ks, qok));
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
    // per-query scale: |dO'| = |dO| 2^k in [64, 128) at the row maximum
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
        // (element by element from scalars: hipcc 7.2 miscompiles a loop that assigns hw[j] of an uninitialised ext_vector, see h8_to_b8)
        const uint32_t w0 = pack_h2(bfbits_lo(dob[ks][0]) * up, bfbits_hi(dob[ks][0]) * up);
        const uint32_t w1 = pack_h2(bfbits_lo(dob[ks][1]) * up, bfbits_hi(dob[ks][1]) * up);
        const uint32_t w2 = pack_h2(bfbits_lo(dob[ks][2]) * up, bfbits_hi(dob[ks][2]) * up);
        const uint32_t w3 = pack_h2(bfbits_lo(dob[ks][3]) * up, bfbits_hi(dob[ks][3]) * up);
        dof[ks] = as_bf16x8(u32x4{w0, w1, w2, w3});
    }
    const float deltas = delta * up;
    const float sc2 = p.scale * LOG2E;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    float rs = 0.f;          // sum of the ROUNDED dS' this lane fed to the MFMAs (mean-key correction, see dq_rowsum_fix)

    // fragment addresses (LDS bytes); k-step / d-tile / second read enter by XOR on bits the lane part leaves free
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
            for (int j = 0; j < PPW; ++j) BMT_B_DMA_K(j, tn, slotn);
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_B_DMA_V(j, tn, slotn);
        } else {
            const uint32_t kA = kA0 + slot * STAGE, vA = vA0 + slot * STAGE, kT = kT0 + slot * STAGE;
            // ---- S^T = K . Q^T on two accumulators (even / odd k-steps: no MFMA waits for its predecessor)
            f32x16 st0, st1, dp0, dp1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st0[r] = 0.f; st1[r] = 0.f; dp0[r] = 0.f; dp1[r] = 0.f; }
            u32x4 kf[3];
            kf[0] = lds_b128<0>(kA);
            kf[1] = lds_b128<0>(kA ^ (1 << 5));
#define BMT_B_SSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) kf[((ks_) + 2) % 3] = lds_b128<0>(kA ^ (((ks_) + 2) << 5));      \
        if constexpr ((ks_) < PPW) BMT_B_DMA_K((ks_) % PPW, tn, slotn);                                \
        else if constexpr ((ks_) < 2 * PPW) BMT_B_DMA_V((ks_) % PPW, tn, slotn);                       \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(kf[(ks_) % 3]);                             \
        if constexpr (((ks_) & 1) == 0) st0 = mfma32t<true>(as_bf16x8(kf[(ks_) % 3]), qf[(ks_)], st0); \
        else st1 = mfma32t<true>(as_bf16x8(kf[(ks_) % 3]), qf[(ks_)], st1);                            \
    }
            BMT_X_REP16(BMT_B_SSTEP)
#undef BMT_B_SSTEP
            // ---- dP'^T = V . dO'^T; the probability of element r = k-step r is computed right behind MFMA r
            float pr[16];
            u32x4 vf[3];
            vf[0] = lds_b128<0>(vA);
            vf[1] = lds_b128<0>(vA ^ (1 << 5));
#define BMT_B_PSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) vf[((ks_) + 2) % 3] = lds_b128<0>(vA ^ (((ks_) + 2) << 5));      \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(vf[(ks_) % 3]);                             \
        if constexpr (((ks_) & 1) == 0) dp0 = mfma32t<true>(as_bf16x8(vf[(ks_) % 3]), dof[(ks_)], dp0); \
        else dp1 = mfma32t<true>(as_bf16x8(vf[(ks_) % 3]), dof[(ks_)], dp1);                           \
        if constexpr ((ks_) < 16) {                                                                    \
            const float pe_ = __builtin_amdgcn_exp2f(__builtin_fmaf(st0[(ks_) & 15] + st1[(ks_) & 15], sc2, -lse2)); \
            pr[(ks_) & 15] = (all_valid || ((mw[((ks_) & 15) >> 2] >> (8 * ((ks_) & 3))) & 0xffu)) ? pe_ : 0.f; \
        }                                                                                              \
    }
            BMT_X_REP16(BMT_B_PSTEP)
#undef BMT_B_PSTEP
            if constexpr (KS < 16) {       // d_k 128: eight k-steps carried probabilities 0 .. 7
#pragma unroll
                for (int r = KS; r < 16; ++r) {
                    const float pe_ = __builtin_amdgcn_exp2f(__builtin_fmaf(st0[r] + st1[r], sc2, -lse2));
                    pr[r] = (all_valid || ((mw[r >> 2] >> (8 * (r & 3))) & 0xffu)) ? pe_ : 0.f;
                }
            }
            // ---- dS' = P (dP' - delta') scale, rounded to fp16 (clamped: an overflow must not become inf)
            bf16x8 dsf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint32_t dwv[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int r0 = 8 * kk + 2 * j2;
                    float a0 = pr[r0] * ((dp0[r0] + dp1[r0]) - deltas) * p.scale;
                    float a1 = pr[r0 + 1] * ((dp0[r0 + 1] + dp1[r0 + 1]) - deltas) * p.scale;
                    a0 = fminf(fmaxf(a0, -60000.f), 60000.f);
                    a1 = fminf(fmaxf(a1, -60000.f), 60000.f);
                    dwv[j2] = pack_h2(a0, a1);
                    rs += h_bits2f(dwv[j2] & 0xffffu) + h_bits2f(dwv[j2] >> 16);
                }
                dsf[kk] = as_bf16x8(u32x4{dwv[0], dwv[1], dwv[2], dwv[3]});
            }
            // ---- dQ'^T += K^T . dS'^T: MFMA n = 2 dt + kk, fragment = rows 16 kk + 4 hh .. and 16 kk + 8 + 4 hh .. through the transpose unit
            u32x2 ta[3], tb[3];
#define BMT_B_TFRAG(n_)                                                                   \
    do {                                                                                  \
        ta[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1)) * ROWB>(kT ^ (((n_) >> 1) << 6));     \
        tb[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1) + 8) * ROWB>(kT ^ ((((n_) >> 1) << 6) | 32)); \
    } while (0)
            BMT_B_TFRAG(0);
            BMT_B_TFRAG(1);
#define BMT_B_QSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + 2 < 2 * DT) BMT_B_TFRAG((n_) + 2);                                        \
        lgkm_wait<((n_) + 2 < 2 * DT) ? 4 : 2 * (2 * DT - 1 - (n_))>(ta[(n_) % 3], tb[(n_) % 3]);      \
        const u32x4 av = {ta[(n_) % 3][0], ta[(n_) % 3][1], tb[(n_) % 3][0], tb[(n_) % 3][1]};         \
        dq[(n_) >> 1] = mfma32t<true>(as_bf16x8(av), dsf[(n_) & 1], dq[(n_) >> 1]);                    \
    }
            BMT_X_REP16(BMT_B_QSTEP)
#undef BMT_B_QSTEP
#undef BMT_B_TFRAG
        }
        // tile t + 1 has landed (this wave's share) once at most the two younger tiles' requests are pending; the barrier publishes all
        // shares and says every wave is done with tile t (its slot is the target of the requests of iteration t + 1)
        if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BMT_B_BAR();
    }
#undef BMT_B_DMA_K
#undef BMT_B_DMA_V

    // ---- epilogue: mean-key correction in scaled units, scale out, outputs in the forms the projection backward takes
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-fetches of the last tile may still be landing in the ring
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
int launch_dq32(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds_loop = 4 * 2 * 32 * DK * 2 + ((ntile * 32 + 15) & ~15), lds_epi = DK * (128 + 8) * 2;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq32_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dq32_kernel<DK>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_dq32");
    return BMT_OK;
}

}  // namespace

// the dQ half of bmt_attn_bwd_bf16 (same argument block; include/bmt_hip.h) on the experimental kernel.  Needs what the product call
// leaves in its workspaces: delta_ws (delta = (1 - p) rowsum(dO * O)) and dOh_ws (the bf16 plane of dO) -- run the product first with
// the same workspaces, then this entry with its own dQ outputs.  fp16 q / k / v planes (qkv_f16), d_k 128 / 256, key-padding masks.
extern "C" int bmt_exp_attn_bwd_dq32(const bmt_attn_bwd_bf16_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && a->lse && a->delta_ws && a->dOh_ws && (a->dQ || a->dQh), "bmt_exp_attn_bwd_dq32: null pointer");
    BMT_CHECK_ARG(a->qkv_f16 && (a->dk == 128 || a->dk == 256), "bmt_exp_attn_bwd_dq32: fp16 q / k / v planes, d_k 128 / 256");
    BMT_CHECK_ARG(a->mask == nullptr || a->mask_qs == 0, "bmt_exp_attn_bwd_dq32: key-padding masks only");
    BMT_CHECK_ARG(a->Sk <= 8192 && (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) && (int64_t)a->Sk * a->ldv * 2 < (1ll << 31), "bmt_exp_attn_bwd_dq32: Sk too large");
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Kh = a->Kh; p.Vh = a->Vh; p.dOh = a->dOh_ws;
    p.lse = a->lse; p.delta = a->delta_ws;
    p.gq = GradOut{a->dQ, a->ldo, a->bso, a->dQh, a->gq_ld, a->gq_bs, a->dQT, a->gqT_ld, a->dbq};
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p;
    p.kmean = a->kmean;
    p.qkv_f16 = 1;
    hipStream_t st = (hipStream_t)stream;
    return a->dk == 256 ? launch_dq32<256>(p, st) : launch_dq32<128>(p, st);
}
