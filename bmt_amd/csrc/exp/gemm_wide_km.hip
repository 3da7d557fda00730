// EXPERIMENT (exp/build.sh -> libbmt_exp.so; NOT yet run on a GPU: written after round 2's GPU budget was spent, harness
// tools/probes/gemm_wide_km_check.py): the 256 x 256 ping-pong GEMM kernel (gemm_wide_kernel, ../gemm_bf16.hip) with a K-MAJOR weight
// operand -- the dX product of every nn.Linear, dX[M][K_in] = dY[M][N_out] . W[N_out][K_in]: the reduction index is the ROW of W.  Today
// that product runs on the register-staged 128-row loop (1.52 ms per step, 18 % of the nominal matrix peak).
// GENERATED from the product kernel's text by tools/probes/make_wide_km.py (the ping-pong structure, the activation operand and the
// epilogue are the product's, line for line); what differs:
//   * W half-tile = [64 reduction rows][128 output columns] (rows of 256 B, 16 LDS-DMA pieces of 4 rows), 16-byte chunk position =
//     chunk ^ 4 (row & 3) -- the image attn_fwd32_kernel uses for V (emulated in tools/probes/gemm_wide_km_layout.py);
//   * W fragments (MFMA A operand: 32 output columns x 16 reduction rows) through ds_read_b64_tr_b16, reduction index in natural order,
//     as inline asm under the phase's own lgkmcnt(0);
//   * one plane, bf16 (the backward's operand format).
// Not there yet: column sums in the epilogue (the dX launches that also produce a bias gradient, FFN-2's among them) and a way to fill the
// chip with 8192 x 1024 outputs (128 tiles): see DESIGN.md section 7 item 3.
#ifndef BMT_EXP_LIB
#include "../gemm_bf16.hip"
#endif

namespace {

template <bool F16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wide_km_kernel(const GemmB p) {
    constexpr int HT = 16384, SLOT = 4 * HT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int half = lane >> 5, l31 = lane & 31;

    // tile order: XCD remap, then groups of 8 activation panels walked panel-first (the ~32 tiles an XCD runs together share
    // 8 activation panels and 4 weight panels in its L2)
    const int tiles_m = p.tiles_m, tiles_n = p.tiles_n;       // activation / weight panels of 256 rows
    const int w = xcd_remap((int)blockIdx.x, tiles_m * tiles_n);
    const int per_group = 8 * tiles_n;
    const int g = w / per_group, first_m = g * 8;
    const int gsz = min(tiles_m - first_m, 8);
    const int wi = w - g * per_group;
    const int m0 = (first_m + wi % gsz) * 256, n0 = (wi / gsz) * 256;

    // ---- LDS-DMA: a half-tile is 16 pieces of 1 KB (8 rows x 128 B); wave w fills pieces 2w, 2w+1 (rows 16w .. 16w+15)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ah, 0, (int)((int64_t)p.M * p.lda * 2), 0x00020000);
    // k-major weight operand: plane [reduction rows = p.krows][output columns], rows past the reduction read as zero
    const __amdgpu_buffer_rsrc_t rsWh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)((int64_t)p.krows * p.ldb * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWl = rsWh;      // (one plane: single-pass bf16)
    int xvo[2], wvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * wid + 8 * j + (lane >> 3);
        const int ks = (lane & 7) ^ ((row >> 1) & 7);
        xvo[j] = row * (int)p.lda * 2 + ks * 16;
        // W half-tile = [64 reduction rows][128 output columns] = 16 pieces of 4 rows x 256 B; 16-byte chunk position = chunk ^ 4 (row & 3):
        // the four rows of a transposing read fall into the four 64-byte bank quarters (tools/probes/gemm_wide_km_layout.py)
        const int rowk = 4 * (2 * wid + j) + (lane >> 4);
        wvo[j] = rowk * (int)p.ldb * 2 + (((lane & 15) ^ (4 * (rowk & 3))) * 16);
    }
    const int T1 = p.Kpad / 64;
    const int T = p.Bl ? 2 * T1 : T1;
#define BMT_W_DMA_W(t_, slot_)                                                                                       \
    do {                                                                                                             \
        const bool lo_ = (t_) >= T1;                                                                                 \
        const __amdgpu_buffer_rsrc_t rs_ = lo_ ? rsWl : rsWh;                                                        \
        const int k_ = ((t_) - (lo_ ? T1 : 0)) * 128;                                                                \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lptr_t)(smem + (slot_) * SLOT + hf * HT + (2 * wid + j) * 1024), 16, wvo[j], \
                                                         (k_ >> 1) * (int)p.ldb * 2 + (n0 + 128 * hf) * 2, 0, 0);   \
    } while (0)
#define BMT_W_DMA_X(t_, slot_)                                                                                       \
    do {                                                                                                             \
        const int k_ = ((t_) >= T1 ? (t_) - T1 : (t_)) * 128;                                                        \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(smem + (slot_) * SLOT + (2 + hf) * HT + (2 * wid + j) * 1024), 16, xvo[j], \
                                                         (m0 + 128 * hf) * (int)p.lda * 2 + k_, 0, 0);              \
    } while (0)

    // ---- fragment addresses: lane (l31 = row of the 32-row fragment, half) reads slot (2 s + half) ^ swizzle(row) for k16 step s
    int offX[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int o = l31 * 128 + (((2 * s + half) ^ ((l31 >> 1) & 7)) * 16);
        offX[s] = o + (2 + (wc >> 1)) * HT + (wc & 1) * 8192;
    }
#define BMT_W_FRAG(slot_, off_, i_) as_bf16x8(*reinterpret_cast<const u32x4*>(smem + (slot_) * SLOT + (off_) + (i_) * 4096))
    // W fragments through the transpose unit: A operand row = output column 32 blk + l31 of this group's half-tile, k-index 8 half + jj
    // = reduction row 16 s + 8 half + jj (natural order: the activation fragments stay plain row reads).  Two ds_read_b64_tr_b16 (rows
    // .. + 0-3 and + 4-7); lane (m16 = lane & 15, gi = (lane >> 4) & 1) points at row 8 half + (m16 >> 2), columns 32 blk + 16 gi + 4 (m16 & 3).
    // Inline asm (a builtin ds_read_tr next to an LDS-DMA in flight gets vmcnt(0) from hipcc); the waits are the phase's own lgkmcnt(0).
    const int m16 = lane & 15, gi = (lane >> 4) & 1, mq = m16 >> 2, mr = m16 & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t wT0 = lds0 + wr * HT + (8 * half + mq) * 256 + 64 * mq + 32 * gi + 8 * mr;
    u32x2 ra[2][4], rb[2][4];
#define BMT_W_KREAD(slot_, blk_, i_, s_)                                                                             \
    do {                                                                                                             \
        const uint32_t a_ = (wT0 + (slot_) * SLOT) ^ ((blk_) << 6);                                                  \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ra[i_][s_]) : "v"(a_), "n"((s_) * 4096));         \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(rb[i_][s_]) : "v"(a_), "n"((s_) * 4096 + 1024));  \
    } while (0)
#define BMT_W_KREAD8(slot_, blk0_)                                                                                   \
    do {                                                                                                             \
        BMT_W_KREAD(slot_, (blk0_), 0, 0); BMT_W_KREAD(slot_, (blk0_), 0, 1); BMT_W_KREAD(slot_, (blk0_), 0, 2); BMT_W_KREAD(slot_, (blk0_), 0, 3); \
        BMT_W_KREAD(slot_, (blk0_) + 1, 1, 0); BMT_W_KREAD(slot_, (blk0_) + 1, 1, 1); BMT_W_KREAD(slot_, (blk0_) + 1, 1, 2); BMT_W_KREAD(slot_, (blk0_) + 1, 1, 3); \
    } while (0)
    // after the phase's lgkmcnt(0): tie the read registers to the wait (nothing that uses them may be scheduled above it), then pack
#define BMT_W_KPACK()                                                                                                \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                          \
                asm volatile("" : "+v"(ra[i][s]), "+v"(rb[i][s]));                                                   \
                wa[i][s] = as_bf16x8(u32x4{ra[i][s][0], ra[i][s][1], rb[i][s][0], rb[i][s][1]});                     \
            }                                                                                                        \
    } while (0)
#define BMT_W_BAR()                                  \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        __builtin_amdgcn_s_barrier();                \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;
    bf16x8 wa[2][4], xb0[4], xb1[4];

    // one K-tile in slot e (compile-time): 4 phases
#define BMT_W_MFMA(ib_, xb_, bcol_)                                                                                  \
    do {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
                acc[(ib_) + i][bcol_] = mfma32t<F16>(wa[i][s], xb_[s], acc[(ib_) + i][bcol_]);                       \
        __builtin_amdgcn_s_setprio(0);                                                                               \
    } while (0)
#define BMT_W_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BMT_W_KTILE(e_, t_)                                                                                          \
    do {                                                                                                             \
        const bool next2_ = (t_) + 2 < T;                                                                            \
        /* phase 0: quadrant (w0, x0) */                                                                             \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) xb0[s] = BMT_W_FRAG(e_, offX[s], 0);                           \
        BMT_W_KREAD8(e_, 0);                                                                                         \
        BMT_W_LGKM0();                                                                                               \
        BMT_W_KPACK();                                                                                               \
        BMT_W_BAR();                                                                                                 \
        BMT_W_MFMA(0, xb0, 0);                                                                                       \
        BMT_W_BAR();                                                                                                 \
        /* phase 1: (w0, x1) */                                                                                      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) xb1[s] = BMT_W_FRAG(e_, offX[s], 1);                           \
        BMT_W_LGKM0();                                                                                               \
        BMT_W_BAR();                                                                                                 \
        BMT_W_MFMA(0, xb1, 1);                                                                                       \
        BMT_W_BAR();                                                                                                 \
        /* phase 2: (w1, x1); the activation half-tiles of this slot were last read in phase 1 by both groups (their reads      \
           retired before the barrier that ended it): K-tile t + 2 may overwrite them */                             \
        BMT_W_KREAD8(e_, 2);                                                                                         \
        if (next2_) BMT_W_DMA_X((t_) + 2, e_);                                                                       \
        BMT_W_LGKM0();                                                                                               \
        BMT_W_KPACK();                                                                                               \
        BMT_W_BAR();                                                                                                 \
        BMT_W_MFMA(2, xb1, 1);                                                                                       \
        BMT_W_BAR();                                                                                                 \
        /* phase 3: (w1, x0), operands in registers.  K-tile t + 1 has landed (only this phase 2's requests are younger) before  \
           the barrier that precedes its first read; the weight half-tiles of this slot are free now */              \
        if (next2_) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        if (next2_) BMT_W_DMA_W((t_) + 2, e_);                                                                       \
        BMT_W_BAR();                                                                                                 \
        BMT_W_MFMA(2, xb0, 0);                                                                                       \
        BMT_W_BAR();                                                                                                 \
    } while (0)

#ifdef BMT_EXP
    const int tile_id = (int)blockIdx.x;
#endif
    BMT_STAMP(0);
    BMT_W_DMA_W(0, 0);
    BMT_W_DMA_X(0, 0);
    if (T > 1) {
        BMT_W_DMA_W(1, 1);
        BMT_W_DMA_X(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BMT_W_BAR();
    BMT_STAMP(1);
    if (wr == 1) BMT_W_BAR();                  // group 1 runs one barrier behind group 0
    for (int t = 0; t < T; t += 2) {
        BMT_W_KTILE(0, t);
        if (t + 1 < T) BMT_W_KTILE(1, t + 1);
    }
    if (wr == 0) BMT_W_BAR();
    BMT_STAMP(2);
#undef BMT_W_LGKM0
#undef BMT_W_KTILE
#undef BMT_W_MFMA
#undef BMT_W_BAR
#undef BMT_W_FRAG
#undef BMT_W_KREAD
#undef BMT_W_KREAD8
#undef BMT_W_KPACK
#undef BMT_W_DMA_W
#undef BMT_W_DMA_X

    // ---------------- epilogue (order: alpha, bias, dropout_pre, relu, dropout_post, gate, residual).
    // acc[i][b][r]: output row m = m0 + 64 wc + 32 b + l31, column n = n0 + 128 wr + 32 i + 8 (r >> 2) + 4 half + (r & 3): a lane
    // holds 4 consecutive columns of a row per register group.  Stored from there every instruction would touch 32 rows x 32 bytes
    // (measured: 2.4 TB/s over the chip, 28 us per tile).  Each wave therefore turns its tile through a PRIVATE 8 KB LDS chunk
    // ([32 rows][64 columns] fp32, 16-byte slots XOR-swizzled by the row, no barrier -- only the wave's own lgkmcnt) and writes
    // 8-column row segments: an instruction covers 8 rows x 256 contiguous bytes (C) / 128 bytes (each plane).  The residual /
    // gate segments of the next step are requested before this step's stores (the stores may alias them, so the compiler would
    // otherwise serialise load -> store round trips).
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    const unsigned f = p.flags;
    const int pcols = p.Chi ? p.plane_cols : 0;
    const bool c_al = p.C && ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
    const bool pre_r = (f & BMT_EPI_RESIDUAL) && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0);
    const bool pre_g = (f & BMT_EPI_GATE) && ((p.ldg & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.gate) & 15) == 0);
    char* chunk = smem + SLOT + wid * 8192;                  // the second K-tile slot is idle now
    const int cgl = (lane & 7) * 8, rsel = lane >> 3;        // this lane's 8 columns of the 64-column chunk, row within a step of 8
    float4 rr[2][2];
    u32x4 rg[2];
    // step st (0 .. 15): chunk ch = st >> 2 = (b, i pair), rows 8 (st & 3) + rsel of the chunk
    auto seg_row = [&](int st) { return m0 + 64 * wc + 32 * (st >> 3) + 8 * (st & 3) + rsel; };
    auto seg_col = [&](int st) { return n0 + 128 * wr + 64 * ((st >> 2) & 1) + cgl; };
    auto prefetch = [&](int st, int buf) {
        const int row = seg_row(st), col = seg_col(st);
        const bool ok = row < p.M && col + 8 <= p.N;
        if (pre_r && ok) {
            const float* rp = p.residual + (int64_t)row * p.ldr + col;
            rr[buf][0] = *reinterpret_cast<const float4*>(rp);
            rr[buf][1] = *reinterpret_cast<const float4*>(rp + 4);
        }
        if (pre_g && ok) rg[buf] = *reinterpret_cast<const u32x4*>(p.gate + (int64_t)row * p.ldg + col);
    };
    float bv[8];
    prefetch(0, 0);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const int b = ch >> 1, ip = ch & 1;
        // registers -> chunk: fragment i = 2 ip + ii, register group j: 4 columns 32 ii + 8 j + 4 half .. of row l31
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = (32 * ii + 8 * j + 4 * half) >> 2;
                float4 t;
                t.x = acc[2 * ip + ii][b][4 * j + 0] * p.alpha; t.y = acc[2 * ip + ii][b][4 * j + 1] * p.alpha;
                t.z = acc[2 * ip + ii][b][4 * j + 2] * p.alpha; t.w = acc[2 * ip + ii][b][4 * j + 3] * p.alpha;
                *reinterpret_cast<float4*>(chunk + l31 * 256 + ((slot ^ (l31 & 15)) * 16)) = t;
            }
        if (f & BMT_EPI_BIAS) {
            const int col = n0 + 128 * wr + 64 * ip + cgl;
#pragma unroll
            for (int q = 0; q < 8; ++q) bv[q] = (col + q < p.N) ? p.bias[col + q] : 0.f;
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int st = 4 * ch + s4, buf = st & 1;
            if (st + 1 < 16) prefetch(st + 1, buf ^ 1);
            const int rl = 8 * s4 + rsel;
            const int row = seg_row(st), col = seg_col(st);
            float v[8];
            {
                const float4 t0 = *reinterpret_cast<const float4*>(chunk + rl * 256 + (((cgl >> 2) ^ (rl & 15)) * 16));
                const float4 t1 = *reinterpret_cast<const float4*>(chunk + rl * 256 + ((((cgl >> 2) + 1) ^ (rl & 15)) * 16));
                v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
            }
            if (row >= p.M || (col >= p.N && col >= pcols)) continue;
            const bool full = col + 8 <= p.N;
            const int64_t idx = (int64_t)row * p.ldc + col;
            if (f & BMT_EPI_BIAS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += bv[q];
            }
            if (f & BMT_EPI_DROP_PRE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)(idx + q));
            }
            if (f & BMT_EPI_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if (f & BMT_EPI_DROP_POST) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)(idx + q));
            }
            if (f & BMT_EPI_GATE) {
                if (full && pre_g) {
                    const u32x4 gv = rg[buf];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] = (gv[q] & 0x00007FFFu) ? v[2 * q] * p.gate_scale : 0.f;
                        v[2 * q + 1] = (gv[q] & 0x7FFF0000u) ? v[2 * q + 1] * p.gate_scale : 0.f;
                    }
                } else {
                    const uint16_t* gp = p.gate + (int64_t)row * p.ldg + col;
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (col + q < p.N && (gp[q] & 0x7fffu)) ? v[q] * p.gate_scale : 0.f;
                }
            }
            if (f & BMT_EPI_RESIDUAL) {
                if (full && pre_r) {
                    const float4 t0 = rr[buf][0], t1 = rr[buf][1];
                    v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
                } else {
                    const float* rp = p.residual + (int64_t)row * p.ldr + col;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (col + q < p.N) v[q] += rp[q];
                }
            }
            if (!full) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (col + q >= p.N) v[q] = 0.f;
            }
            if (p.C) {
                if (full && c_al) {
                    *reinterpret_cast<float4*>(p.C + idx) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(p.C + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (col + q < p.N) p.C[idx + q] = v[q];
                }
            }
            if (p.Chi && col < pcols) {
                u32x4 h, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t h_, l_;
                    split_bf2(v[2 * q], v[2 * q + 1], h_, l_);
                    h[q] = h_;
                    l[q] = p.second_f16 ? pack_h2(v[2 * q], v[2 * q + 1]) : l_;
                    if (p.hi_f16) h[q] = pack_h2(v[2 * q], v[2 * q + 1]);
                }
                const int64_t pi = (int64_t)row * p.ldp + col;
                if (p.plane_vec) {
                    *reinterpret_cast<u32x4*>(p.Chi + pi) = h;
                    if (p.Clo) *reinterpret_cast<u32x4*>(p.Clo + pi) = l;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (col + q < pcols) {
                            p.Chi[pi + q] = (uint16_t)(h[q >> 1] >> (16 * (q & 1)));
                            if (p.Clo) p.Clo[pi + q] = (uint16_t)(l[q >> 1] >> (16 * (q & 1)));
                        }
                }
            }
        }
    }
    BMT_STAMP(3);
}


int launch_wide_km(const GemmB& p, hipStream_t st) {
    constexpr int lds = 2 * 4 * 16384;
    (void)hipFuncSetAttribute((const void*)gemm_wide_km_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((gemm_wide_km_kernel<false>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_exp_gemm_wide_km");
    return BMT_OK;
}

}  // namespace

// bmt_gemm_bf16's argument block (include/bmt_hip.h): row-major A, k-major B, BMT_PREC_BF16, no split-K / column sums / accumulation
extern "C" int bmt_exp_gemm_wide_km(const bmt_gemm_bf16_args* a, void* stream) {
    BMT_CHECK_ARG(a && !a->a_kmajor && a->b_kmajor && !a->conv_mode && a->precision == BMT_PREC_BF16, "bmt_exp_gemm_wide_km: row-major A, k-major B, bf16");
    BMT_CHECK_ARG(!a->colsum && !(a->flags & BMT_EPI_ACCUM) && a->splitk <= 1 && a->N >= 256, "bmt_exp_gemm_wide_km: plain epilogue, N >= 256");
    BMT_CHECK_ARG((int64_t)(a->M + 256) * a->lda * 2 < (1ll << 31) && (int64_t)(a->Kpad + 64) * a->ldb * 2 < (1ll << 31), "bmt_exp_gemm_wide_km: plane too large");
    GemmB p;
    int splitk = 1;
    const int rc = gemm_prepare(a, p, splitk, false);
    if (rc != BMT_OK) return rc;
    p.bm = 256;
    p.pipe = 3;
    p.Bl = nullptr;
    p.tiles_m = bmt_cdiv(a->M, 256);
    p.tiles_n = bmt_cdiv(p.Chi ? (p.plane_cols > a->N ? p.plane_cols : a->N) : a->N, 256);
    return launch_wide_km(p, (hipStream_t)stream);
}
