// EXPERIMENT (round 4, not built into the library): the weight-gradient products on 256 x 256 tiles -- gemm_wide_kernel's 8-phase loop with
// both operands k-major (LDS-DMA of [64 reduction rows][128 columns] half-tiles, transposing fragment reads as inline asm), one workgroup per CU,
// atomics straight from the accumulators.  Correct (tests/test_gpu_kernels.py::test_grouped_weight_gradient_launch over 14 shapes), and slower
// than the 128 x 128 register-staged tile it was meant to replace: the video stream's problems 797 us at 34 % MFMA-busy and 1.8 TB/s of fetch
// (profiles/r04_n_pmc_dw_wide.json) against ~746 us; grouped class 1.15 vs 1.07 ms, step 8.55-8.59 vs 8.48-8.49 (profiles/r04_n_ab_dw_wide.txt).
// Why: both operands of a weight gradient STREAM from HBM (every byte is used by a handful of tiles), two K-tile slots of 64 KB are all the LDS
// holds, so a K-tile's requests have one K-tile period (0.85 us at the MFMA-bound rate) to land -- and the slowest of them takes an HBM round
// trip: the loop runs at ~2.2 us per K-tile.  The 128-row tile hides that latency with two workgroups per CU and three stages in flight each.
// This file is the kernel as it was in bmt_amd/csrc/gemm_bf16.hip (it needs that translation unit's GemmB, XcdSeg, km_sw_off, mfma32t ...); the
// host side split the grouped call's problems by tile kind (M, N >= 256 and <= 10 % more padding than 128-wide tiles).

// ===================================================================== 256 x 256 tile for the weight gradients: both operands k-major
// dW[n][k] += sum_r dY[r][n] X[r][k] over a chunk of rows, in the 8-phase structure of gemm_wide_kernel: half the operand bytes per FLOP
// of the 128 x 128 tile (its grouped launch fetched 3.9 GB for ~1 GB of unique operands) and 1.5 transposing reads per MFMA instead of 3.
//   * a half-tile is [64 reduction rows][128 columns] as stored (256-byte rows), DMA'd in 1-KB pieces of 4 rows, 16-byte slots XOR-ed with
//     4 (row & 3) on the source side -- the image km_frag_sw / km_sw_off read (the four rows of a transposing read fall into the four
//     64-byte bank quarters);
//   * MFMA A = dY columns (output rows n), B = X columns (output columns k): a lane's accumulator registers are 4 consecutive n of one k,
//     the 32 lanes of a half-wave 32 consecutive k -- the atomics of the epilogue are 128-byte contiguous requests, straight from registers;
//   * fragment reads are inline asm (hipcc drains the DMA queue -- vmcnt(0) -- in front of a ds_read_tr builtin) behind the phase's own
//     lgkmcnt(0);
//   * one work item = (tile, chunk of p.kchunk rows) of a problem, found through the XCD segment lists exactly as gemm_bf16_grouped_kernel does.
template <int OFF>
__device__ __forceinline__ u32x2 gw_tr_b64(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int S>
__device__ __forceinline__ bf16x8 gw_km_frag(uint32_t addr) {      // reduction indices 16 S .. 16 S + 15 of this lane's column (km_frag_sw)
    const u32x2 lo = gw_tr_b64<S * 4096>(addr), hi = gw_tr_b64<S * 4096 + 1024>(addr);
    return __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wide_km_grouped_kernel(
    const GemmB* __restrict__ table, const XcdSeg* __restrict__ segs, const int* __restrict__ nseg) {
    constexpr int HT = 16384, SLOT = 4 * HT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const XcdSeg* sx = segs + x * XCD_MAXSEG;
    int lo = 0, hi = nseg[x] - 1;
    if (hi < 0) return;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sx[mid].first_slot <= slot) lo = mid;
        else hi = mid - 1;
    }
    const XcdSeg sg = sx[lo];
    const int local = slot - sg.first_slot;
    if (local >= sg.count * sg.nsplit) return;
    const GemmB p = table[sg.prob];
    const int tile = sg.tile_off + local % sg.count, split = local / sg.count;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int half = lane >> 5, l31 = lane & 31;
    // tile -> (row panel of 256 dY columns, column panel of 256 X columns): groups of 8 row panels walked column by column
    const int per_group = 8 * p.tiles_n;
    const int g = tile / per_group, first_m = g * 8;
    const int gsz = min(p.tiles_m - first_m, 8);
    const int wi = tile - g * per_group;
    const int m0 = (first_m + wi % gsz) * 256, n0 = (wi / gsz) * 256;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.Kpad, kbeg + p.kchunk);
    const int T = (kend - kbeg) / 64;

    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ah, 0, (int)((int64_t)p.krows * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)((int64_t)p.krows * p.ldb * 2), 0x00020000);
    int avo[2][2], bvo[2][2];
    {
        const int kr = lane >> 4, sl = (lane & 15) ^ (kr << 2);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (2 * wid + j) * 4 + kr;
                avo[hf][j] = row * (int)p.lda * 2 + min(m0 + 128 * hf + sl * 8, (int)p.lda - 8) * 2;
                bvo[hf][j] = row * (int)p.ldb * 2 + min(n0 + 128 * hf + sl * 8, (int)p.ldb - 8) * 2;
            }
    }
#define BMT_K_DMA_A(t_, slot_)                                                                                       \
    do {                                                                                                             \
        const int so_ = (kbeg + (t_) * 64) * (int)p.lda * 2;                                                          \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + (slot_) * SLOT + hf * HT + (2 * wid + j) * 1024), 16, avo[hf][j], so_, 0, 0); \
    } while (0)
#define BMT_K_DMA_B(t_, slot_)                                                                                       \
    do {                                                                                                             \
        const int so_ = (kbeg + (t_) * 64) * (int)p.ldb * 2;                                                          \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(smem + (slot_) * SLOT + (2 + hf) * HT + (2 * wid + j) * 1024), 16, bvo[hf][j], so_, 0, 0); \
    } while (0)

    // fragment addresses (LDS byte addresses of K-tile slot 0; slot 1 is + SLOT, past the 16-bit immediate): A fragment i = dY columns
    // 128 wr + 32 i .., B fragment b = X columns 64 wc + 32 b ..
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    uint32_t adA[2][4], adB[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int i = 0; i < 4; ++i) adA[e][i] = lds0 + e * SLOT + wr * HT + km_sw_off(32 * i, lane);
#pragma unroll
        for (int b = 0; b < 2; ++b) adB[e][b] = lds0 + e * SLOT + (2 + (wc >> 1)) * HT + km_sw_off((wc & 1) * 64 + 32 * b, lane);
    }
#define BMT_K_BAR()                                  \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        __builtin_amdgcn_s_barrier();                \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
#define BMT_K_FRAGS4(dst_, ad_) \
    do { dst_[0] = gw_km_frag<0>(ad_); dst_[1] = gw_km_frag<1>(ad_); dst_[2] = gw_km_frag<2>(ad_); dst_[3] = gw_km_frag<3>(ad_); } while (0)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;
    bf16x8 wa[2][4], xb0[4], xb1[4];
#define BMT_K_MFMA(ib_, xb_, bcol_)                                                                                  \
    do {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
                acc[(ib_) + i][bcol_] = mfma32t<false>(wa[i][s], xb_[s], acc[(ib_) + i][bcol_]);                     \
        __builtin_amdgcn_s_setprio(0);                                                                               \
    } while (0)
#define BMT_K_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BMT_K_KTILE(e_, t_)                                                                                          \
    do {                                                                                                             \
        const bool next2_ = (t_) + 2 < T;                                                                            \
        /* phase 0: (A fragments 0, 1) x B fragment 0 */                                                             \
        BMT_K_FRAGS4(xb0, adB[e_][0]);                                                                               \
        BMT_K_FRAGS4(wa[0], adA[e_][0]);                                                                             \
        BMT_K_FRAGS4(wa[1], adA[e_][1]);                                                                             \
        BMT_K_LGKM0();                                                                                               \
        BMT_K_BAR();                                                                                                 \
        BMT_K_MFMA(0, xb0, 0);                                                                                       \
        BMT_K_BAR();                                                                                                 \
        /* phase 1: (A 0, 1) x B 1 */                                                                                \
        BMT_K_FRAGS4(xb1, adB[e_][1]);                                                                               \
        BMT_K_LGKM0();                                                                                               \
        BMT_K_BAR();                                                                                                 \
        BMT_K_MFMA(0, xb1, 1);                                                                                       \
        BMT_K_BAR();                                                                                                 \
        /* phase 2: (A 2, 3) x B 1; the B half-tiles of this slot were last read in phase 1: K-tile t + 2 may overwrite them */ \
        BMT_K_FRAGS4(wa[0], adA[e_][2]);                                                                             \
        BMT_K_FRAGS4(wa[1], adA[e_][3]);                                                                             \
        if (next2_) BMT_K_DMA_B((t_) + 2, e_);                                                                       \
        BMT_K_LGKM0();                                                                                               \
        BMT_K_BAR();                                                                                                 \
        BMT_K_MFMA(2, xb1, 1);                                                                                       \
        BMT_K_BAR();                                                                                                 \
        /* phase 3: (A 2, 3) x B 0 from registers; K-tile t + 1 has landed (only this phase 2's requests are younger); the A          \
           half-tiles of this slot are free */                                                                       \
        if (next2_) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        if (next2_) BMT_K_DMA_A((t_) + 2, e_);                                                                       \
        BMT_K_BAR();                                                                                                 \
        BMT_K_MFMA(2, xb0, 0);                                                                                       \
        BMT_K_BAR();                                                                                                 \
    } while (0)

    BMT_K_DMA_A(0, 0);
    BMT_K_DMA_B(0, 0);
    if (T > 1) {
        BMT_K_DMA_A(1, 1);
        BMT_K_DMA_B(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BMT_K_BAR();
    if (wr == 1) BMT_K_BAR();                  // group 1 runs one barrier behind group 0
    for (int t = 0; t < T; t += 2) {
        BMT_K_KTILE(0, t);
        if (t + 1 < T) BMT_K_KTILE(1, t + 1);
    }
    if (wr == 0) BMT_K_BAR();
#undef BMT_K_LGKM0
#undef BMT_K_KTILE
#undef BMT_K_MFMA
#undef BMT_K_BAR
#undef BMT_K_FRAGS4
#undef BMT_K_DMA_A
#undef BMT_K_DMA_B
    // C += alpha * acc: acc[i][b][r] is output row n = m0 + 128 wr + 32 i + acc_row(r, half), column k = n0 + 64 wc + 32 b + l31
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + 64 * wc + 32 * b + l31;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 128 * wr + 32 * i + acc_row(r, half);
                if (row < p.M) atomicAdd(p.C + (int64_t)row * p.ldc + col, acc[i][b][r] * p.alpha);
            }
        }
}

