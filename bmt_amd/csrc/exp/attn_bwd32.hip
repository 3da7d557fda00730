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

// (kswz, bfbits_lo / bfbits_hi and BMT_B_BAR live in the product file now: the split backward uses them)

template <int DK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq32_kernel(const AttnPB p, float* kq_out) {
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
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)min(q, p.Sq - 1) * p.ldq + h * DK + 8 * hh;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)min(q, p.Sq - 1) * p.ldo + h * DK + 8 * hh;
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
    // 2^k(q): the dK / dV kernel derives its (one) scale from the smallest of these; a row without gradient does not take part
    if (kq_out != nullptr && qok && hh == 0) kq_out[stat] = amax > 0.f ? up : 3.0e38f;
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
int launch_dq32(const AttnPB& p, float* kq_out, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds_loop = 4 * 2 * 32 * DK * 2 + ((ntile * 32 + 15) & ~15), lds_epi = DK * (128 + 8) * 2;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq32_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dq32_kernel<DK>), dim3(nblk), dim3(256), lds, st, p, kq_out);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_dq32");
    return BMT_OK;
}


// ----------------------------------------------------------------------------------------------------------------- dK / dV
// 32 keys per wave (K fp16 and V bf16 fragments in registers, both 32 x d_k accumulators in AGPRs), 4 waves = 128 keys per workgroup,
// loop over 32-query stages: the Q tile (fp16) and the dO tile (bf16) come by LDS-DMA into a 4-deep ring, BOTH with the dual-purpose
// swizzle of the dQ kernel's K image (row fragments for S / dP, transposing reads for dK^T / dV^T):
//   S[q][key]   = Qtile . K^T     (fp16; A = Q rows from LDS, B = K registers)  -> lane (key = l31, hh) register 4 i + j: query 8 i + 4 hh + j
//   dP[q][key]  = dOtile . V^T    (bf16; V converted once in the prologue)
//   dV^T[d][key] += dO^T . P      (bf16; A = dO^T through the transpose unit, B = P registers 8 kk .. 8 kk + 7)
//   dK^T[d][key] += Q^T . dS'     (fp16; dS' = dS 2^g with ONE scale per (batch, head): the reduction runs over the queries, a per-query
//                                  scale could not be taken out again; g = min_q k(q) of the dQ kernel's row scales, so the largest row
//                                  is where the dQ kernel put it and smaller rows lose absolute, not relative, accuracy)
// lse (in log2 units), delta and nothing else per query are staged into LDS once (no ordinary load in the loop); rows past Sq get
// lse = +huge (P = 0).  A masked key is a lane whose P is zero throughout: its gradient rows are written as zeros.
// OPEN (compiler output, no GPU run yet): at d_k = 256 the kernel needs K, V (128 registers) + two gradient tiles (256) + S, dP (32) +
// ~90 working registers = the whole 512-register file, and hipcc 7.2 spills 636 bytes per lane -- it keeps the K / V fragments in
// scratch and reloads them every stage (32 scratch_load_dwordx4 in the loop).  Still correct (its waits are conservative, and extra
// VMEM operations only make the counted vmcnt of the stage end stronger), but a scratch reload waits, in order, behind the DMA requests
// issued before it: the three-tiles-ahead prefetch collapses to the latency of the newest request.  Ways out: the TWO-pass form below
// (dV with K only, then dK with K and V, in one launch: 256 + 222 registers, no scratch; S is computed twice = 80 instead of 64 MFMAs per
// stage); or pinning the register classes by hand (MFMAs as inline asm: K / V fragments and dK in AGPRs, dV + everything the VALU touches
// in VGPRs -- needs the gfx950 MFMA hazard table for the s_nops the compiler no longer inserts).  d_k = 128 (configs[4]) is spill-free
// in one pass.
// one pass over the query stages of a (batch, head): DO_DV / DO_DK select the gradient products (both: one pass, 64 MFMAs per stage; the
// two-pass kernel runs <true, false> then <false, true>: S twice = 80 MFMAs per stage pair, but K + one gradient tile (+ V in the second
// pass) fit the register file without scratch)
template <int DK, bool DO_DV, bool DO_DK>
__device__ __forceinline__ void dkv_pass(const AttnPB& p, char* smem, const int nst, const bool compute, const bool kok, const int wid, const int hh,
                                         const uint32_t lds0, const uint32_t qA0, const uint32_t qT0, const __amdgpu_buffer_rsrc_t rsQ,
                                         const __amdgpu_buffer_rsrc_t rsO, const int (&qvo)[4], const int (&ovo)[4], const bf16x8 (&kf)[DK / 16],
                                         const bf16x8 (&vf)[DK / 16], f32x16 (&dka)[DK / 32], f32x16 (&dva)[DK / 32], const float sc2, const float scu) {
    constexpr int BQ = 32, KS = DK / 16, DT = DK / 32, ROWB = DK * 2, TILE = BQ * ROWB, STAGE = 2 * TILE, NS = 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NP = BQ / RPP, PPW = NP / 4;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int sstep_q = BQ * (int)p.ldq * 2, sstep_o = BQ * (int)p.ldo * 2;
#define BMT_B_DMA_Q(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + (slot_) * STAGE + (wid * PPW + (j_)) * 1024), 16, qvo[j_], (t_) * sstep_q, 0, 0)
#define BMT_B_DMA_O(j_, t_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + (slot_) * STAGE + TILE + (wid * PPW + (j_)) * 1024), 16, ovo[j_], (t_) * sstep_o, 0, 0)
    for (int t = 0; t < nst; ++t) {
        const int slot = t % NS, slotn = (t + NS - 1) % NS;
        const int tn = min(t + NS - 1, nst - 1);
        if (!compute) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_B_DMA_Q(j, tn, slotn);
#pragma unroll
            for (int j = 0; j < PPW; ++j) BMT_B_DMA_O(j, tn, slotn);
        } else {
            const uint32_t qA = qA0 + slot * STAGE, oA = qA + TILE, qT = qT0 + slot * STAGE, oT = qT + TILE;
            // lse of this lane's 16 queries (8 i + 4 hh + j), pinned here by asm reads (a plain load would be hoisted and lengthen its live range)
            const uint32_t statA = lds0 + NS * STAGE + (t * BQ + 4 * hh) * 4;
            u32x4 lsr[4];
            lsr[0] = lds_b128<0>(statA); lsr[1] = lds_b128<32>(statA); lsr[2] = lds_b128<64>(statA); lsr[3] = lds_b128<96>(statA);
            f32x16 st0, dp0;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st0[r] = 0.f; dp0[r] = 0.f; }
            // ---- S = Qtile . K^T (fp16)
            u32x4 af[3];
            af[0] = lds_b128<0>(qA);
            af[1] = lds_b128<0>(qA ^ (1 << 5));
#define BMT_B_SSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) af[((ks_) + 2) % 3] = lds_b128<0>(qA ^ (((ks_) + 2) << 5));      \
        if constexpr ((ks_) < PPW) BMT_B_DMA_Q((ks_) % PPW, tn, slotn);                                \
        else if constexpr ((ks_) < 2 * PPW) BMT_B_DMA_O((ks_) % PPW, tn, slotn);                       \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(af[(ks_) % 3]);                             \
        st0 = mfma32t<true>(as_bf16x8(af[(ks_) % 3]), kf[(ks_)], st0);                                 \
    }
            BMT_X_REP16(BMT_B_SSTEP)
#undef BMT_B_SSTEP
            // ---- P = exp2(S scale log2 e - lse), rounded to bf16 at once: the B operand of dV AND (unpacked again) the factor of dS --
            // the fp32 probabilities would otherwise live through the dP block; dS carries bf16's 2^-9 from P, as every dS of the bf16 kernels does
            lgkm_wait<0>(lsr[0]); lgkm_wait<0>(lsr[1]); lgkm_wait<0>(lsr[2]); lgkm_wait<0>(lsr[3]);
            uint32_t pw[8];
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) {
                const int r0 = 2 * j2;
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st0[r0], sc2, -__uint_as_float(lsr[r0 >> 2][r0 & 3])));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st0[r0 + 1], sc2, -__uint_as_float(lsr[(r0 + 1) >> 2][(r0 + 1) & 3])));
                pw[j2] = kok ? pack_bf2(p0, p1) : 0u;
            }
            u32x2 ta[3], tb[3];
#define BMT_B_TFRAG(base_, n_)                                                            \
    do {                                                                                  \
        ta[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1)) * ROWB>((base_) ^ (((n_) >> 1) << 6));     \
        tb[(n_) % 3] = lds_tr_b64<(16 * ((n_) & 1) + 8) * ROWB>((base_) ^ ((((n_) >> 1) << 6) | 32)); \
    } while (0)
            if constexpr (DO_DV) {
                // ---- dV^T += dO^T . P (bf16), A through the transpose unit
                bf16x8 pf[2];
                pf[0] = as_bf16x8(u32x4{pw[0], pw[1], pw[2], pw[3]});
                pf[1] = as_bf16x8(u32x4{pw[4], pw[5], pw[6], pw[7]});
                BMT_B_TFRAG(oT, 0);
                BMT_B_TFRAG(oT, 1);
#define BMT_B_VSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + 2 < 2 * DT) BMT_B_TFRAG(oT, (n_) + 2);                                    \
        lgkm_wait<((n_) + 2 < 2 * DT) ? 4 : 2 * (2 * DT - 1 - (n_))>(ta[(n_) % 3], tb[(n_) % 3]);      \
        const u32x4 av = {ta[(n_) % 3][0], ta[(n_) % 3][1], tb[(n_) % 3][0], tb[(n_) % 3][1]};         \
        dva[(n_) >> 1] = mfma32t<false>(as_bf16x8(av), pf[(n_) & 1], dva[(n_) >> 1]);                  \
    }
                BMT_X_REP16(BMT_B_VSTEP)
#undef BMT_B_VSTEP
            }
            if constexpr (DO_DK) {
                // ---- dP = dOtile . V^T (bf16)
                u32x4 bfg[3];
                bfg[0] = lds_b128<0>(oA);
                bfg[1] = lds_b128<0>(oA ^ (1 << 5));
#define BMT_B_PSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + 2 < KS) bfg[((ks_) + 2) % 3] = lds_b128<0>(oA ^ (((ks_) + 2) << 5));     \
        lgkm_wait<((ks_) + 2 < KS) ? 2 : (KS - 1 - (ks_))>(bfg[(ks_) % 3]);                            \
        dp0 = mfma32t<false>(as_bf16x8(bfg[(ks_) % 3]), vf[(ks_)], dp0);                               \
    }
                BMT_X_REP16(BMT_B_PSTEP)
#undef BMT_B_PSTEP
                // ---- dS' = P (dP - delta) scale 2^g (fp16, clamped)
                u32x4 dlr[4];
                const uint32_t delA = statA + nst * BQ * 4;
                dlr[0] = lds_b128<0>(delA); dlr[1] = lds_b128<32>(delA); dlr[2] = lds_b128<64>(delA); dlr[3] = lds_b128<96>(delA);
                lgkm_wait<0>(dlr[0]); lgkm_wait<0>(dlr[1]); lgkm_wait<0>(dlr[2]); lgkm_wait<0>(dlr[3]);
                uint32_t dw[8];
#pragma unroll
                for (int j2 = 0; j2 < 8; ++j2) {
                    const int r0 = 2 * j2;
                    float a0 = bfbits_lo(pw[j2]) * (dp0[r0] - __uint_as_float(dlr[r0 >> 2][r0 & 3])) * scu;
                    float a1 = bfbits_hi(pw[j2]) * (dp0[r0 + 1] - __uint_as_float(dlr[(r0 + 1) >> 2][(r0 + 1) & 3])) * scu;
                    a0 = fminf(fmaxf(a0, -60000.f), 60000.f);
                    a1 = fminf(fmaxf(a1, -60000.f), 60000.f);
                    dw[j2] = pack_h2(a0, a1);
                }
                bf16x8 dsf[2];
                dsf[0] = as_bf16x8(u32x4{dw[0], dw[1], dw[2], dw[3]});
                dsf[1] = as_bf16x8(u32x4{dw[4], dw[5], dw[6], dw[7]});
                // ---- dK'^T += Q^T . dS' (fp16)
                BMT_B_TFRAG(qT, 0);
                BMT_B_TFRAG(qT, 1);
#define BMT_B_KSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + 2 < 2 * DT) BMT_B_TFRAG(qT, (n_) + 2);                                    \
        lgkm_wait<((n_) + 2 < 2 * DT) ? 4 : 2 * (2 * DT - 1 - (n_))>(ta[(n_) % 3], tb[(n_) % 3]);      \
        const u32x4 av = {ta[(n_) % 3][0], ta[(n_) % 3][1], tb[(n_) % 3][0], tb[(n_) % 3][1]};         \
        dka[(n_) >> 1] = mfma32t<true>(as_bf16x8(av), dsf[(n_) & 1], dka[(n_) >> 1]);                  \
    }
                BMT_X_REP16(BMT_B_KSTEP)
#undef BMT_B_KSTEP
            }
#undef BMT_B_TFRAG
        }
        if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BMT_B_BAR();
    }
#undef BMT_B_DMA_Q
#undef BMT_B_DMA_O
}

// TWO: two passes over the queries in one launch (dV with K only, then dK with K and V): no scratch at d_k 256, S computed twice
template <int DK, bool TWO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkv32x_kernel(const AttnPB p, const float* kq) {
    constexpr int BQ = 32, NT = 256, KS = DK / 16, DT = DK / 32, ROWB = DK * 2, TILE = BQ * ROWB, STAGE = 2 * TILE, NS = 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NP = BQ / RPP, PPW = NP / 4;
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    static_assert(PPW <= 4, "pieces per wave");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int nst = (p.Sq + BQ - 1) / BQ;
    float* sLse = reinterpret_cast<float*>(smem + NS * STAGE);          // [nst * 32] lse * log2 e (+huge past Sq)
    float* sDel = sLse + nst * BQ;                                       // [nst * 32] delta
    float* sRed = sDel + nst * BQ;                                       // [4] block reduction scratch

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
    const bool wave_on = kt * 128 + wid * 32 < p.Sk;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Qh + (int64_t)b * p.bsq + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sq - 1) * p.ldq + DK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dOh + (int64_t)b * p.bso + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sq - 1) * p.ldo + DK) * 2), 0x00020000);
    int qvo[4], ovo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid * PPW + (j % PPW)) * RPP + lane / CPR, cpos = lane % CPR;
        qvo[j] = row * (int)p.ldq * 2 + ((cpos ^ kswz(row)) * 16);
        ovo[j] = row * (int)p.ldo * 2 + ((cpos ^ kswz(row)) * 16);
    }
    const int sstep_q = BQ * (int)p.ldq * 2, sstep_o = BQ * (int)p.ldo * 2;
    // the first NS - 1 tiles of a pass (a tile index past the end re-fetches the last tile: harmless, uniform counts)
#define BMT_B_PRIME()                                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NS - 1; ++s_) {                                                                  \
        const int tl_ = min(s_, nst - 1);                                                                                    \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j)                                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + s_ * STAGE + (wid * PPW + j) * 1024), 16, qvo[j], tl_ * sstep_q, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j)                                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + s_ * STAGE + TILE + (wid * PPW + j) * 1024), 16, ovo[j], tl_ * sstep_o, 0, 0); \
    }
    BMT_B_PRIME()
    bf16x8 kf[KS], vf[KS];      // K as fp16 (B operand of S), V as bf16 (B operand of dP)
    const int64_t ko = (int64_t)b * p.bsk + (int64_t)min(key, p.Sk - 1) * p.ldk + h * DK + 8 * hh;
    const int64_t vo = (int64_t)b * p.bsv + (int64_t)min(key, p.Sk - 1) * p.ldv + h * DK + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = ldfrag(p.Kh + ko + 16 * ks, kin);
        if constexpr (!TWO) vf[ks] = h8_to_b8(ldfrag(p.Vh + vo + 16 * ks, kin));
    }
    const int64_t stat0 = ((int64_t)b * p.H + h) * p.Sq;
    float gmin = 3.0e38f;
    for (int i = tid; i < nst * BQ; i += NT) {
        const bool in = i < p.Sq;
        sLse[i] = in ? p.lse[stat0 + i] * LOG2E : 3.0e38f;
        sDel[i] = in ? p.delta[stat0 + i] : 0.f;
        if (in && kq != nullptr) gmin = fminf(gmin, kq[stat0 + i]);
    }
    gmin = -wave_max(-gmin);
    if (lane == 0) sRed[wid] = gmin;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gmin = fminf(fminf(sRed[0], sRed[1]), fminf(sRed[2], sRed[3]));
    const float up = (kq != nullptr && gmin < 1.0e38f) ? gmin : 1.f, down = 1.f / up;     // powers of two
    const float sc2 = p.scale * LOG2E, scu = p.scale * up;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int fk = kswz(l31);
    const uint32_t qA0 = lds0 + l31 * ROWB + 32 * (fk >> 1) + 16 * (hh ^ (fk & 1));          // row fragments of the Q tile (+ TILE: dO tile)
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t qT0 = lds0 + (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1);   // transposing reads
    const bool compute = wave_on && __any(kok);
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem);

    f32x16 acc[DT];             // two-pass: dV, then dK'; one pass: dK' (dV in acc2)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    if constexpr (TWO) {
        dkv_pass<DK, true, false>(p, smem, nst, compute, kok, wid, hh, lds0, qA0, qT0, rsQ, rsO, qvo, ovo, kf, kf, acc, acc, sc2, scu);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        grad_store_rows<DK>(p.gv, acc, b, h, key, kin, hh);
        if (p.gv.hiT || p.gv.bsum) {
            grad_tile_write<DK, 128>(tile, acc, wid * 32, kin, l31, hh);
            __syncthreads();
            grad_tile_flush<DK, 128>(tile, p.gv, b, h, kt * 128, p.Sk, tid);
        }
        __syncthreads();
        BMT_B_PRIME()
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks] = h8_to_b8(ldfrag(p.Vh + vo + 16 * ks, kin));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dkv_pass<DK, false, true>(p, smem, nst, compute, kok, wid, hh, lds0, qA0, qT0, rsQ, rsO, qvo, ovo, kf, vf, acc, acc, sc2, scu);
    } else {
        f32x16 acc2[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[dt][r] = 0.f;
        dkv_pass<DK, true, true>(p, smem, nst, compute, kok, wid, hh, lds0, qA0, qT0, rsQ, rsO, qvo, ovo, kf, vf, acc, acc2, sc2, scu);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        grad_store_rows<DK>(p.gv, acc2, b, h, key, kin, hh);
        if (p.gv.hiT || p.gv.bsum) {
            grad_tile_write<DK, 128>(tile, acc2, wid * 32, kin, l31, hh);
            __syncthreads();
            grad_tile_flush<DK, 128>(tile, p.gv, b, h, kt * 128, p.Sk, tid);
            __syncthreads();
        }
    }
#undef BMT_B_PRIME
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] *= down;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    grad_store_rows<DK>(p.gk, acc, b, h, key, kin, hh);
    if (p.gk.hiT || p.gk.bsum) {
        grad_tile_write<DK, 128>(tile, acc, wid * 32, kin, l31, hh);
        __syncthreads();
        grad_tile_flush<DK, 128>(tile, p.gk, b, h, kt * 128, p.Sk, tid);
    }
}

template <int DK, bool TWO>
int launch_dkv32(const AttnPB& p, const float* kq, hipStream_t st) {
    const int nblk = ((p.Sk + 127) / 128) * p.B * p.H;
    const int nst = (p.Sq + 31) / 32;
    const int lds_loop = 4 * 2 * 32 * DK * 2 + 2 * nst * 32 * 4 + 64, lds_epi = DK * (128 + 8) * 2;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv32x_kernel<DK, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dkv32x_kernel<DK, TWO>), dim3(nblk), dim3(256), lds, st, p, kq);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_dkv32");
    return BMT_OK;
}

}  // namespace

// the dQ half of bmt_attn_bwd_bf16 (same argument block; include/bmt_hip.h) on the experimental kernel.  Needs what the product call
// leaves in its workspaces: delta_ws (delta = (1 - p) rowsum(dO * O)) and dOh_ws (the bf16 plane of dO) -- run the product first with
// the same workspaces, then this entry with its own dQ outputs.  fp16 q / k / v planes (qkv_f16), d_k 128 / 256, key-padding masks.
// kq_out (optional, fp32 [B][H][Sq]): the per-query scale 2^k(q) for bmt_exp_attn_bwd_dkv32.
extern "C" int bmt_exp_attn_bwd_dq32(const bmt_attn_bwd_bf16_args* a, float* kq_out, void* stream) {
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
    return a->dk == 256 ? launch_dq32<256>(p, kq_out, st) : launch_dq32<128>(p, kq_out, st);
}

// the dK / dV half, same protocol (delta_ws and dOh_ws from a product call); kq: the row scales bmt_exp_attn_bwd_dq32 left (nullptr: dS
// unscaled); two_pass: dV and dK in two passes over the queries (no scratch at d_k 256, S twice).  Sq <= 3072 (lse / delta of the (batch, head)
// live in LDS), no per-query mask.
extern "C" int bmt_exp_attn_bwd_dkv32(const bmt_attn_bwd_bf16_args* a, const float* kq, int two_pass, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && a->lse && a->delta_ws && a->dOh_ws && (a->dK || a->dKh) && (a->dV || a->dVh),
                  "bmt_exp_attn_bwd_dkv32: null pointer");
    BMT_CHECK_ARG(a->qkv_f16 && (a->dk == 128 || a->dk == 256), "bmt_exp_attn_bwd_dkv32: fp16 q / k / v planes, d_k 128 / 256");
    BMT_CHECK_ARG(a->mask == nullptr || a->mask_qs == 0, "bmt_exp_attn_bwd_dkv32: key-padding masks only");
    BMT_CHECK_ARG(a->Sq <= 3072 && (int64_t)a->Sq * a->ldq * 2 < (1ll << 31) && (int64_t)a->Sq * a->ldo * 2 < (1ll << 31), "bmt_exp_attn_bwd_dkv32: Sq too large");
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Kh = a->Kh; p.Vh = a->Vh; p.dOh = a->dOh_ws;
    p.lse = a->lse; p.delta = a->delta_ws;
    p.gk = GradOut{a->dK, a->dkv_ld, a->dkv_bs, a->dKh, a->gkv_ld, a->gkv_bs, a->dKT, a->gkvT_ld, a->dbk};
    p.gv = GradOut{a->dV, a->dkv_ld, a->dkv_bs, a->dVh, a->gkv_ld, a->gkv_bs, a->dVT, a->gkvT_ld, a->dbv};
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p;
    p.qkv_f16 = 1;
    hipStream_t st = (hipStream_t)stream;
    if (a->dk == 256) return two_pass ? launch_dkv32<256, true>(p, kq, st) : launch_dkv32<256, false>(p, kq, st);
    return two_pass ? launch_dkv32<128, true>(p, kq, st) : launch_dkv32<128, false>(p, kq, st);
}

// the whole attention backward on the experimental kernels, drop-in for bmt_attn_bwd_bf16 (same argument block): the product's delta
// kernel (delta = (1 - p) rowsum(dO * O), and the bf16 plane of dO when dO arrives as fp32), then dQ (leaving the row scales in kq_ws,
// fp32 [B][H][Sq]), then dK / dV.  ops.attn_bwd_planes routes here under BMT_ATTN_BWD32=1 (off by default).
extern "C" int bmt_exp_attn_bwd_all(const bmt_attn_bwd_bf16_args* a, float* kq_ws, int two_pass, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && (a->O || a->Oh || a->Of) && a->lse && a->delta_ws && a->dOh_ws && kq_ws, "bmt_exp_attn_bwd_all: null pointer");
    BMT_CHECK_ARG(a->qkv_f16 && (a->dk == 128 || a->dk == 256), "bmt_exp_attn_bwd_all: fp16 q / k / v planes, d_k 128 / 256");
    BMT_CHECK_ARG(a->mask == nullptr || a->mask_qs == 0, "bmt_exp_attn_bwd_all: key-padding masks only");
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.dOh = a->dOh_ws; p.O = a->O; p.Oph = a->Oh; p.Opl = a->Ol; p.Opf = a->Of; p.ldop = a->ldop; p.bsop = a->bsop;
    p.dO = a->dO; p.delta = a->delta_ws;
    p.ldo = a->ldo; p.bso = a->bso;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk; p.drop_p = a->drop_p;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)a->B * a->H * a->Sq;
    hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3(bmt_cdiv(rows, 4)), dim3(256), 0, st, p, a->dk, a->dOh_ws);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_all(delta)");
    int rc = bmt_exp_attn_bwd_dq32(a, kq_ws, stream);
    if (rc != BMT_OK) return rc;
    return bmt_exp_attn_bwd_dkv32(a, kq_ws, two_pass, stream);
}
