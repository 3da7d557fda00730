// EXPERIMENT driver (not part of libbmt_hip.so; built by tools/experiments/build.sh into tools/experiments/libbmt_exp.so, driven by
// tools/probes/attn_fwd32_check.py): the instruction-placement variants of attn_fwd32_kernel (attention_bf16.hip) and a switch back to
// the 16-query kernel, behind the product's argument block -- old and new kernel, and every variant, timed in ONE process on one box.
//   variant v < 8: DMAV = v % 4 (where the next tile's DMA requests are issued), PRIO = v < 4 (s_setprio around the MFMA phases);
//   variant 100:   attn_fwd64_kernel (the 16-query kernel), whatever the shape;
//   variant 200 + XP (d_k 256 fp16): the probe copy below with loop parts switched off (XP bits: 1 no DMA in the loop, 2 no softmax
//                  arithmetic, 4 no end-of-stage wait + barrier, 8 no output stores) -- timing only, the output is not attention;
//   variant 300 + XP: the same with 40 KB of extra LDS per workgroup: ONE workgroup per CU instead of two;
//   variant 400 + 100 ORD + 10 DEPTH + XP: prefetch depth of the fragment reads and MFMA ordering of the probe copy (XP 0 or 7).
// Result (profiles/r02_q_attn_fwd32_variants.txt): the eight variants are within +-3 % of each other on every shape.
// (attention_bf16.hip is included by exp_lib.hip ahead of this file)

namespace {
int g_variant = 0;

// ---- probe copy of attn_fwd32_kernel (attention_bf16.hip) with parts of the loop switched off: what does each part cost?
// XP (probe bits, results are NOT attention any more): 1 = no DMA inside the loop (every stage reads tile 0), 2 = no softmax arithmetic
// (P = the raw scores converted), 4 = no end-of-stage wait + barrier (use with 1)
// DEPTH: fragments requested ahead of the MFMA that uses them (ring of DEPTH + 1 register sets).  ORD bit 0: the PV MFMAs in key-half-major
// order (consecutive MFMAs never share an accumulator), bit 1: S on two accumulators (even / odd k-steps).
template <int DK, bool F16, int XP, int DEPTH = 2, int ORD = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd32x_kernel(const AttnPB p) {
    constexpr int DMAV = 0;
    constexpr bool PRIO = true;
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
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;
    const bool wave_on = qt * 128 + wid * 32 < p.Sq;                     // waves past Sq only move tiles
    const int ntile = (p.Sk + BC - 1) / BC;

    // ---- LDS-DMA: piece = 1 KB = RPP rows; wave w moves pieces w * PPW .. of the K and of the V tile
    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sk - 1) * p.ldk + DK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sk - 1) * p.ldv + DK) * 2), 0x00020000);
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
#define BMT_X_DMA_K(j_, t_, buf_) \
    do { if (!((XP & 1) && inloop)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + (buf_) * STAGE + (wid * PPW + (j_)) * 1024), 16, kvo[j_], (t_) * sstep_k, 0, 0); } while (0)
#define BMT_X_DMA_V(j_, t_, buf_) \
    do { if (!((XP & 1) && inloop)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + (buf_) * STAGE + TILE + (wid * PPW + (j_)) * 1024), 16, vvo[j_], (t_) * sstep_v, 0, 0); } while (0)

    // stage 0 in flight first, then the Q fragments and the mask row
    constexpr bool inloop0 = false;
    { constexpr bool inloop = inloop0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) BMT_X_DMA_K(j, 0, 0);
#pragma unroll
    for (int j = 0; j < PPW; ++j) BMT_X_DMA_V(j, 0, 0);
    }
    constexpr bool inloop = true;
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
        const int cur = (XP & 1) ? 0 : (t & 1);
        const int tn = min(t + 1, ntile - 1);             // the last stage re-fetches itself into the idle buffer (branch-free)
        const int key0 = t * BC;
        uint32_t mw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mw[i] = *reinterpret_cast<const uint32_t*>(sMask + key0 + 8 * i + 4 * hh);
        const bool none_valid = __all((mw[0] | mw[1] | mw[2] | mw[3]) == 0u);
        const bool all_valid = __all((mw[0] & mw[1] & mw[2] & mw[3]) == 0x01010101u);
        if (none_valid || !wave_on) {                     // nothing to compute: only move the next tile
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
            constexpr int RB = DEPTH + 1;
            u32x4 kf[RB];
            f32x16 st2;
#pragma unroll
            for (int r = 0; r < 16; ++r) st2[r] = 0.f;
#define BMT_X_KPRE(i_) if constexpr ((i_) < DEPTH && (i_) < KS) kf[(i_)] = lds_b128<0>(kA ^ ((i_) << 5));
            BMT_X_KPRE(0) BMT_X_KPRE(1) BMT_X_KPRE(2) BMT_X_KPRE(3) BMT_X_KPRE(4) BMT_X_KPRE(5)
#undef BMT_X_KPRE
            if constexpr (DMAV == 2) {
#pragma unroll
                for (int j = 0; j < PPW; ++j) BMT_X_DMA_K(j, tn, cur ^ 1);
#pragma unroll
                for (int j = 0; j < PPW; ++j) BMT_X_DMA_V(j, tn, cur ^ 1);
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#define BMT_X_SSTEP(ks_)                                                                               \
    if constexpr ((ks_) < KS) {                                                                        \
        if constexpr ((ks_) + DEPTH < KS) kf[((ks_) + DEPTH) % RB] = lds_b128<0>(kA ^ (((ks_) + DEPTH) << 5)); \
        if constexpr (DMAV == 0) {                                                                     \
            if constexpr ((ks_) < PPW) BMT_X_DMA_K((ks_) % PPW, tn, cur ^ 1);                          \
            else if constexpr ((ks_) < 2 * PPW) BMT_X_DMA_V((ks_) % PPW, tn, cur ^ 1);                 \
        }                                                                                              \
        lgkm_wait<((ks_) + DEPTH < KS) ? DEPTH : (KS - 1 - (ks_))>(kf[(ks_) % RB]);                    \
        if constexpr ((ORD & 2) && ((ks_) & 1)) st2 = mfma32t<F16>(as_bf16x8(kf[(ks_) % RB]), qf[(ks_)], st2); \
        else st = mfma32t<F16>(as_bf16x8(kf[(ks_) % RB]), qf[(ks_)], st);                              \
    }
            BMT_X_REP16(BMT_X_SSTEP)
            if constexpr (ORD & 2) st += st2;
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
            if (!(XP & 2) && __any(tmax > m_run + TAU2)) {              // stale-reference online softmax (attn_fwd_bf16_kernel): exact
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
                if constexpr (XP & 2) x[i] = st[i];
                else x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], sc2, -m_use));
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
            u32x2 va[RB], vb[RB];
            // MFMA n of the PV phase: (d-tile, key half) = (n >> 1, n & 1), or key-half-major (n % DT, n / DT)
#define BMT_X_DTI(n_) ((ORD & 1) ? (n_) % DT : (n_) >> 1)
#define BMT_X_KKI(n_) ((ORD & 1) ? (n_) / DT : (n_) & 1)
#define BMT_X_VFRAG(n_)                                                                   \
    do {                                                                                  \
        const uint32_t a_ = vL ^ (BMT_X_DTI(n_) << 6);                                    \
        va[(n_) % RB] = lds_tr_b64<(16 * BMT_X_KKI(n_)) * ROWB>(a_);                      \
        vb[(n_) % RB] = lds_tr_b64<(16 * BMT_X_KKI(n_) + 8) * ROWB>(a_);                  \
    } while (0)
#define BMT_X_VPRE(i_) if constexpr ((i_) < DEPTH && (i_) < 2 * DT) BMT_X_VFRAG(i_);
            BMT_X_VPRE(0) BMT_X_VPRE(1) BMT_X_VPRE(2) BMT_X_VPRE(3) BMT_X_VPRE(4) BMT_X_VPRE(5)
#undef BMT_X_VPRE
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#define BMT_X_VSTEP(n_)                                                                                \
    if constexpr ((n_) < 2 * DT) {                                                                     \
        if constexpr ((n_) + DEPTH < 2 * DT) BMT_X_VFRAG((n_) + DEPTH);                                \
        lgkm_wait<((n_) + DEPTH < 2 * DT) ? 2 * DEPTH : 2 * (2 * DT - 1 - (n_))>(va[(n_) % RB], vb[(n_) % RB]); \
        const u32x4 av = {va[(n_) % RB][0], va[(n_) % RB][1], vb[(n_) % RB][0], vb[(n_) % RB][1]};     \
        o[BMT_X_DTI(n_)] = mfma32t<F16>(as_bf16x8(av), pf[BMT_X_KKI(n_)], o[BMT_X_DTI(n_)]);           \
    }
            BMT_X_REP16(BMT_X_VSTEP)
#undef BMT_X_VSTEP
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
#undef BMT_X_VFRAG
        }
        if constexpr (!(XP & 4)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
#undef BMT_X_DMA_K
#undef BMT_X_DMA_V

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
                    v[e][j] = drop_apply(dc, o[dt][4 * (2 * ip + e) + j] * inv, (uint64_t)(rowoff + d));
                }
            if (!(XP & 8) && p.Ow && qok) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    *reinterpret_cast<float4*>(p.Ow + rowoff + dt * 32 + 8 * (2 * ip + e) + 4 * hh) = make_float4(v[e][0], v[e][1], v[e][2], v[e][3]);
            }
            if (!(XP & 8) && p.Owh) {
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
    float keep = 0.f;
    if constexpr ((XP & 8) != 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += o[dt][r];
    }
    if (qok && hh == 0) p.lsew[((int64_t)b * p.H + h) * p.Sq + q] = m_run * LN2 + __logf(l_tot) + 1e-30f * keep;
}



template <int XP, int DEPTH = 2, int ORD = 0>
int launch_probe(const AttnPB& p, hipStream_t st, int lds_extra) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds = 2 * 2 * 32 * 256 * 2 + ((ntile * 32 + 15) & ~15) + lds_extra;
    (void)hipFuncSetAttribute((const void*)attn_fwd32x_kernel<256, true, XP, DEPTH, ORD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_fwd32x_kernel<256, true, XP, DEPTH, ORD>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_attn_fwd32 probe");
    return BMT_OK;
}
}

extern "C" void bmt_exp_set_variant(int v) { g_variant = v; }

// same argument block as bmt_attn_fwd_bf16 (include/bmt_hip.h); single-pass precisions, d_k 128 / 256, key-padding masks only
extern "C" int bmt_exp_attn_fwd32(const bmt_attn_fwd_bf16_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && (a->O || a->Oh) && a->lse, "bmt_exp_attn_fwd32: null pointer");
    BMT_CHECK_ARG(a->dk == 128 || a->dk == 256, "bmt_exp_attn_fwd32: d_k=%d not in {128,256}", a->dk);
    BMT_CHECK_ARG(a->precision == BMT_PREC_BF16 || a->precision == BMT_PREC_F16, "bmt_exp_attn_fwd32: single-pass precisions only");
    BMT_CHECK_ARG(a->mask == nullptr || a->mask_qs == 0, "bmt_exp_attn_fwd32: key-padding masks only");
    BMT_CHECK_ARG(a->Sk <= 8192 && (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) && (int64_t)a->Sk * a->ldv * 2 < (1ll << 31), "bmt_exp_attn_fwd32: Sk too large");
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Kh = a->Kh; p.Vh = a->Vh;
    p.Ow = a->O; p.lsew = a->lse;
    p.Owh = a->Oh; p.Owl = a->Of ? a->Of : a->Ol; p.ow_f16 = a->Of != nullptr; p.ldop = a->ldop; p.bsop = a->bsop;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    hipStream_t st = (hipStream_t)stream;
    const bool f16 = a->precision == BMT_PREC_F16;
    if (g_variant == 100) {
        if (a->dk == 256) return f16 ? launch_fwd<256, 1, true>(p, st, 0) : launch_fwd<256, 1, false>(p, st, 0);
        return f16 ? launch_fwd<128, 1, true>(p, st, 0) : launch_fwd<128, 1, false>(p, st, 0);
    }
    if (a->dk == 256 && f16 && g_variant >= 400) {      // 400 + 100 * ORD + 10 * DEPTH + XP (XP 0 or 7)
        const int ord = (g_variant - 400) / 100, depth = (g_variant % 100) / 10, xp = g_variant % 10;
#define BMT_X_CASE(o_, d_) if (ord == o_ && depth == d_) return xp ? launch_probe<7, d_, o_>(p, st, 0) : launch_probe<0, d_, o_>(p, st, 0);
        BMT_X_CASE(0, 1) BMT_X_CASE(0, 2) BMT_X_CASE(0, 3) BMT_X_CASE(0, 4) BMT_X_CASE(1, 2) BMT_X_CASE(1, 4) BMT_X_CASE(2, 2) BMT_X_CASE(3, 2) BMT_X_CASE(3, 3)
#undef BMT_X_CASE
        bmt_set_error("bmt_exp_attn_fwd32: no such probe variant %d", g_variant);
        return BMT_EINVAL;
    }
    if (a->dk == 256 && f16 && g_variant >= 200) {
        const int extra = g_variant >= 300 ? 40960 : 0;
        switch (g_variant % 100) {
            case 1: return launch_probe<1>(p, st, extra);
            case 2: return launch_probe<2>(p, st, extra);
            case 3: return launch_probe<3>(p, st, extra);
            case 5: return launch_probe<5>(p, st, extra);
            case 7: return launch_probe<7>(p, st, extra);
            case 8: return launch_probe<8>(p, st, extra);
            case 15: return launch_probe<15>(p, st, extra);
            default: return launch_probe<0>(p, st, extra);
        }
    }
    if (a->dk == 256 && f16) {
        switch (g_variant) {
            case 1: return launch_fwd32<256, true, 1, true>(p, st);
            case 2: return launch_fwd32<256, true, 2, true>(p, st);
            case 3: return launch_fwd32<256, true, 3, true>(p, st);
            case 4: return launch_fwd32<256, true, 0, false>(p, st);
            case 5: return launch_fwd32<256, true, 1, false>(p, st);
            case 6: return launch_fwd32<256, true, 2, false>(p, st);
            case 7: return launch_fwd32<256, true, 3, false>(p, st);
            default: return launch_fwd32<256, true, 0, true>(p, st);
        }
    }
    if (a->dk == 256) return launch_fwd32<256, false>(p, st);
    return f16 ? launch_fwd32<128, true>(p, st) : launch_fwd32<128, false>(p, st);
}
