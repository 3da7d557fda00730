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


// ---- gradient tile epilogue, row-major through LDS.  The tile's accumulators are G^T[d][token] with the token on the lane: written straight
// to the plane a lane stores 8 bytes per (d-tile, register quad) at a 2-KB row stride -- 32 partial-line requests per instruction, 64
// instructions per wave; measured on attn_bwd_dkvg_kernel (profiles/r03_h_split_probes.txt): 73 of the kernel's 154 us.  Here every wave
// writes its registers into an image [128 tokens][d_k + 8] (ds_write_b64), the workgroup stores the image as whole 512-byte rows (16 bytes per
// lane, consecutive lanes along d) and takes the bias column sums from the same image.
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
__device__ __forceinline__ void grad_rm_flush(const uint16_t* img, float* red, const GradOut& g, int b, int h, int tok0, int S, int tid) {
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
        __syncthreads();
        if (tid < NCW) {
#pragma unroll
            for (int r = 1; r < NRB; ++r) { s0 += red[2 * (tid + r * NCW)]; s1 += red[2 * (tid + r * NCW) + 1]; }
            atomicAdd(g.bsum + h * DK + 2 * cw, s0);
            atomicAdd(g.bsum + h * DK + 2 * cw + 1, s1);
        }
    }
}
// the whole epilogue of one gradient tile (NT threads; the transposed plane, if anybody asks for it, still goes through the old image)
template <int DK, int NTL, int NT>
__device__ __forceinline__ void grad_rm_epilogue(char* smem, const GradOut& g, const f32x16 (&acc)[NTL], int b, int h, int tok0, int trow, bool ok,
                                                 int hh, int dt0, int S, int tid) {
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + 128 * (DK + 8) * 2);
    grad_rm_write<DK, NTL>(img, acc, trow, ok, hh, dt0);
    __syncthreads();
    grad_rm_flush<DK, NT>(img, red, g, b, h, tok0, S, tid);
    __syncthreads();
}

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
    const int slab_bytes = (int)((int64_t)p.Sq * p.ws_pitch * 2);
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
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)q * p.ldq + h * DK + 8 * hh;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)q * p.ldo + h * DK + 8 * hh;
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
template <int DK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq32p_kernel(const AttnPB p) {
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
    const int qt = w % nqt, bh = w / nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q = qt * 128 + wid * 32 + l31;
    const bool qok = q < p.Sq;
    const int ntile = (p.Sk + BC - 1) / BC;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Kh + (int64_t)b * p.bsk + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sk - 1) * p.ldk + DK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vh + (int64_t)b * p.bsv + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sk - 1) * p.ldv + DK) * 2), 0x00020000);
    const int64_t slab = (int64_t)bh * p.ws_slab;
    const int slab_bytes = (int)((int64_t)p.Sq * p.ws_pitch * 2);
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
    if (tid == 0) sLast[0] = -1;
    __syncthreads();
    {
        int last = -1;
        for (int i = tid; i < ntile * BC; i += NT) {
            bool m = false;
            if (i < p.Sk) m = (p.mask != nullptr) ? (p.mask[(int64_t)b * p.mask_bs + i] != 0) : true;
            sMask[i] = m ? (uint16_t)0xffffu : (uint16_t)0;
            if (m) last = i;
        }
        last = (int)wave_max((float)last);
        if (lane == 0 && last >= 0) atomicMax(sLast, last);
    }
    bf16x8 qf[KS];
    u32x4 dob[KS];
    float dsum = 0.f;
    {
        const int64_t qo = (int64_t)b * p.bsq + (int64_t)q * p.ldq + h * DK + 8 * hh;
        const int64_t oo = (int64_t)b * p.bso + (int64_t)q * p.ldo + h * DK + 8 * hh;
        const int64_t po = (int64_t)b * p.bsop + (int64_t)q * p.ldop + h * DK + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = ldfrag(p.Qh + qo + 16 * ks, qok);
            dob[ks] = __builtin_bit_cast(u32x4, ldfrag(p.dOh + oo + 16 * ks, qok));
        }
        if (p.fuse_delta) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (p.Opf) {
                    const u32x4 of = __builtin_bit_cast(u32x4, ldfrag(p.Opf + po + 16 * ks, qok));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        dsum += bfbits_lo(dob[ks][j]) * h_bits2f(of[j] & 0xffffu) + bfbits_hi(dob[ks][j]) * h_bits2f(of[j] >> 16);
                } else {
                    const u32x4 oh = __builtin_bit_cast(u32x4, ldfrag(p.Oph + po + 16 * ks, qok));
                    u32x4 ol = {0u, 0u, 0u, 0u};
                    if (p.Opl) ol = __builtin_bit_cast(u32x4, ldfrag(p.Opl + po + 16 * ks, qok));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        dsum += bfbits_lo(dob[ks][j]) * (bfbits_lo(oh[j]) + bfbits_lo(ol[j])) + bfbits_hi(dob[ks][j]) * (bfbits_hi(oh[j]) + bfbits_hi(ol[j]));
                }
            }
        }
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.Sq + q;
    const float lse2 = qok ? p.lse[stat] * LOG2E : 0.f;
    const float delta = p.fuse_delta ? half_sum(dsum) * (1.f - p.drop_p) : (qok ? p.delta[stat] : 0.f);
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
    if (qok) {        // Qb = bf16(q 2^-k(q)): the A operand of dK^T = Qb^T . dS' in attn_bwd_dkvg_kernel (dS' keeps the 2^k)
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
    const int ntile_run = sLast[0] / BC + 1;                 // (0 when every key is masked)

    // A dependent MFMA issues back to back with its predecessor or waits out its latency (MI355X_MICROARCH.md: any instruction between two
    // MFMAs on one accumulator costs ~43 cycles): consecutive MFMAs here always belong to DIFFERENT accumulators -- S and dP' steps alternate
    // (phase AC), dQ walks the d-tiles inside a 16-key step (phase B).  Fragment reads run PF steps ahead of their MFMA (one wave per SIMD:
    // nothing else hides the LDS latency).
    constexpr int PF = 4, RR = PF + 1;
#define BMT_P_ROWFRAG(i_) lds_b128<(DK == 256 ? (((i_) >> 1) >> 3) * 256 : 0) + (((i_) & 1) ? TILE : 0)>(kan[((i_) >> 1) & 7])
    float pr[16], sth[8];
    f32x16 dp;
#define BMT_P_EXPR(r_, src_) pr[(r_)] = __builtin_amdgcn_exp2f(__builtin_fmaf((src_), sc2, -lse2))
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
        av[(r_)] = pr[(r_)] * __builtin_fmaf(dpv[(r_)], p.scale, dsc);                                   \
        if constexpr (((r_) & 1) == 1) {                                                                \
            const float c0_ = __builtin_amdgcn_fmed3f(av[(r_) - 1], -60000.f, 60000.f);                 \
            const float c1_ = __builtin_amdgcn_fmed3f(av[(r_)], -60000.f, 60000.f);                     \
            dwv[(r_) >> 1] = pack_h2(c0_, c1_) & mm[(r_) >> 2][((r_) >> 1) & 1];                        \
            sw[(r_) >> 1] = pack_bf2(av[(r_) - 1], av[(r_)]);                                           \
            rs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, dwv[(r_) >> 1]), ones, rs, false);     \
        }                                                                                               \
    } while (0)
    // VALU work of combined step i_ (x = step index scaled to 32 steps: d_k 128 has 16 steps, each does two slots)
#define BMT_P_ACWORK(x_)                                                                                \
    do {                                                                                                \
        if constexpr ((x_) < 8) BMT_P_EXPR(8 + (x_), sth[(x_)]);                                        \
        if constexpr ((x_) >= 8 && (x_) < 16) pw[(x_) - 8] = pack_bf2(pr[2 * ((x_) - 8)], pr[2 * ((x_) - 8) + 1]); \
        if constexpr (((x_) & 1) == 0) BMT_P_DEL((x_) >> 1);                                            \
        if constexpr ((x_) == 17) { swap32u(pw[0], pw[2]); swap32u(pw[1], pw[3]); st128<0>(rsP, pw[0], pw[1], pw[2], pw[3], wvo, wso); }  \
        if constexpr ((x_) == 19) { swap32u(pw[4], pw[6]); swap32u(pw[5], pw[7]); st128<32>(rsP, pw[4], pw[5], pw[6], pw[7], wvo, wso); } \
        if constexpr ((x_) == 21) { swap32u(sw[0], sw[2]); swap32u(sw[1], sw[3]); st128<0>(rsS, sw[0], sw[1], sw[2], sw[3], wvo, wso); }  \
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
        if constexpr ((x_) == 0) { swap32u(sw[4], sw[6]); swap32u(sw[5], sw[7]); st128<32>(rsS, sw[4], sw[5], sw[6], sw[7], wvo, wso); } \
        if constexpr ((x_) >= 2 && (x_) < 10) BMT_P_EXPR((x_) - 2, st[(x_) - 2]);                        \
        if constexpr ((x_) >= 8 && (x_) < 16) { sth[(x_) - 8] = st[(x_)]; asm volatile("" : "+v"(sth[(x_) - 8])); } \
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
    grad_rm_epilogue<DK, DT, 256>(smem, p.gq, dq, b, h, qt * 128, wid * 32 + l31, qok, hh, 0, p.Sq, tid);
    if (p.gq.hiT) {
        uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
        grad_tile_write<DK, 128>(tile, dq, wid * 32, qok, l31, hh);
        __syncthreads();
        GradOut gt = p.gq;
        gt.bsum = nullptr;
        grad_tile_flush<DK, 128>(tile, gt, b, h, qt * 128, p.Sq, tid);
    }
}

template <int DK>
int launch_dq32p(const AttnPB& p, hipStream_t st) {
    const int nblk = ((p.Sq + 127) / 128) * p.B * p.H;
    const int ntile = (p.Sk + 31) / 32;
    const int lds_loop = 4 * 2 * 32 * DK * 2 + ((ntile * 64 + 15) & ~15) + 16, lds_epi = 128 * (DK + 8) * 2 + 256 * 8;
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq32p_kernel<DK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_bwd_dq32p_kernel<DK>), dim3(nblk), dim3(256), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_attn_bwd_split(dq, pipelined)");
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

// ---- the same products on 8 waves = two per SIMD: wave (kg = wid & 3, dh = wid >> 2) owns 32 keys x HALF of d_k for both gradients (2 x 64
// accumulator registers: two waves fit a SIMD).  Why: every operand is a ds_read_b64_tr_b16, and one wave per SIMD cannot issue them fast
// enough (PMC of attn_bwd_dkvg_kernel, profiles/r03_f_split_pmc.csv: MFMA 23 % busy, waves issue-stalled 41 % of the time, no bank conflict;
// with the MFMAs AND the DMA switched off the loop still takes 60 % of its time: MI355X_MICROARCH.md, LDS: 4- and 8-byte reads reach their
// rate only from several waves per SIMD).  The P / dS fragments are read by both d-halves (+8 reads per stage and wave pair).
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

template <int DK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkvg8_kernel(const AttnPB p) {
    constexpr int BQ = 32, DT = DK / 32, DTL = DT / 2, ROWB = DK * 2, XT = BQ * ROWB, YT = BQ * 256, STAGE = 2 * XT + 2 * YT;
    constexpr int NS = (DK == 256) ? 3 : 4;
    constexpr int CPR = DK / 8, RPP = 64 / CPR, NPX = BQ / RPP, PPW = NPX / 8;      // X tiles: 16 (8) pieces of 1 KB over 8 waves; Y tiles: 8 pieces
    constexpr int NDMA = 2 * PPW + 2;                                               // requests per wave and stage
    static_assert(DK == 128 || DK == 256, "d_k 128 / 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int nst = (p.Sq + BQ - 1) / BQ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wid & 3, dh = wid >> 2;
    const int hh = lane >> 5, l31 = lane & 31;
    const int nkt = (p.Sk + 127) / 128;
    const int w = xcd_remap(blockIdx.x, nkt * p.B * p.H);
    const int kt = w % nkt, bh = w / nkt;
    const int b = bh / p.H, h = bh % p.H;
    const int key = kt * 128 + kg * 32 + l31;
    const bool kin = key < p.Sk;
    const bool kok = kin && (p.mask == nullptr || p.mask[(int64_t)b * p.mask_bs + key] != 0);
    const int nst_run = __syncthreads_or((int)kok) ? nst : 0;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Qbws + (int64_t)b * p.bsqb + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sq - 1) * p.ldqb + DK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dOh + (int64_t)b * p.bso + h * DK), 0,
                                                                         (int)(((int64_t)(p.Sq - 1) * p.ldo + DK) * 2), 0x00020000);
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
    grad_rm_epilogue<DK, DTL, 512>(smem, p.gv, dva, b, h, kt * 128, kg * 32 + l31, kin, hh, dh * DTL, p.Sk, tid);
    grad_rm_epilogue<DK, DTL, 512>(smem, p.gk, dka, b, h, kt * 128, kg * 32 + l31, kin, hh, dh * DTL, p.Sk, tid);
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

}  // namespace

// the attention backward in split form, drop-in for bmt_attn_bwd_bf16 (same argument block) plus the workspaces: Pws / dSws bf16
// [B*H][Sq][pitch] (pitch a multiple of 32, >= Sk rounded up to 32), Qbws bf16 [B][Sq][H * d_k].  which: bit 0 the product's delta kernel,
// bit 1 the dQ kernel, bit 2 the dK / dV kernel, bit 3 the pipelined dQ kernel (the harness times them one by one).
extern "C" int bmt_exp_attn_bwd_split(const bmt_attn_bwd_bf16_args* a, uint16_t* Pws, uint16_t* dSws, uint16_t* Qbws, int64_t pitch, int which,
                                      void* stream) {
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
    p.gq = GradOut{a->dQ, a->ldo, a->bso, a->dQh, a->gq_ld, a->gq_bs, a->dQT, a->gqT_ld, a->dbq};
    p.gk = GradOut{a->dK, a->dkv_ld, a->dkv_bs, a->dKh, a->gkv_ld, a->gkv_bs, a->dKT, a->gkvT_ld, a->dbk};
    p.gv = GradOut{a->dV, a->dkv_ld, a->dkv_bs, a->dVh, a->gkv_ld, a->gkv_bs, a->dVT, a->gkvT_ld, a->dbv};
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
        const int rc = a->dk == 256 ? launch_dq32p<256>(p, st) : launch_dq32p<128>(p, st);
        if (rc != BMT_OK) return rc;
    }
    if (which & 16) return a->dk == 256 ? launch_dkvg8<256>(p, st) : launch_dkvg8<128>(p, st);      // the 8-wave dK / dV kernel
    if (which & 4) {
        if (a->dk == 128) return launch_dkvg<128>(p, st);
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
    return BMT_OK;
}
