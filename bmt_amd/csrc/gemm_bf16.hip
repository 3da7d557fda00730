// bmt_gemm_bf16 / bmt_planes -- MFMA GEMM over PRE-SPLIT bf16 operand planes, and the kernels that make the planes.
//
// Why a second GEMM: in gemm.hip the x1 and x3 variants of one shape take the same time
// (profiles/r01_b_microbench.txt) -- it is bound by staging fp32 operands (64 KB per 128x128x64 stage through L2->LDS, plus
// the conversion VALU), not by the matrix pipe.  Here every operand is converted ONCE per tensor into bf16 planes
// (hi = bf16(x), lo = bf16(x - hi)), reduction index contiguous and zero-padded to a multiple of 64, so the main loop is
// nothing but {16-byte global loads -> ds_write_b128 -> ds_read_b128 -> MFMA}: half the bytes, no conversion, no edge
// branches (rows are clamped, the K tail is zero padding).
//
//   tile 128x128, 4 waves (2x2), 64x64 per wave = 2x2 v_mfma_f32_32x32x16_bf16; stage = 32 KB in both precisions
//   (x1: BK = 64 of one plane per operand, x3: BK = 32 of hi+lo planes), LDS double-buffered (64 KB -> 2 workgroups/CU),
//   ONE barrier per stage: global loads for stage t+1 are issued before the MFMAs of stage t and written to the other
//   buffer after them.  16-B LDS slots XOR-swizzled so ds_read_b128 operand reads are conflict-free.
//   Tile order: XCD-aware remap, then 8-row-panel groups, so the 64 workgroups resident on one XCD cover an ~8x8 block
//   of tiles and both operand panels are re-read from that XCD's 4 MB L2.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BN = 128;          // tile columns; tile rows are 64 * WM (WM = waves along M: 2 -> 128 rows, 4 -> 256 rows)

struct GemmB {
    const uint16_t *Ah, *Al, *Bh, *Bl;   // planes [M][lda], [N][ldb]; reduction index contiguous, zero padded to Kpad
    int64_t lda, ldb;
    float* C;
    int64_t ldc;
    uint16_t *Chi, *Clo;                 // output planes: Chi = bf16(c) (fp16(c) when hi_f16: the fp16 plane alone); Clo = bf16(c - hi), or fp16(c) when second_f16
    int second_f16, hi_f16;
    int64_t ldp;
    int plane_cols;                      // planes are written for col < plane_cols (zeros for col >= N)
    int plane_vec;                       // plane outputs 16-B aligned with ldp, plane_cols multiples of 8: staged through LDS
    int M, N, Kpad;
    int tiles_m, tiles_n, kchunk;
    int bm;                              // tile rows (128 or 256)
    int pipe;                            // host: launch the LDS-DMA pipelined kernel (256 x 128 tiles)
    int conv_cin;                        // > 0: implicit Conv1d, channels per tap (padded width of the activation plane); see bmt_gemm_bf16_args
    int conv_rows;                       // rows of the halo-padded activation plane reachable from its base pointer
    int conv_S, conv_halo;               // sequence length and halo rows on each side of a sequence
    int conv_tap_minor;                  // Conv1d dW: column tiles in (channel block, tap) order instead of (tap, channel block)
    int krows;                           // k-major operands: valid rows (the true reduction length); rows [krows, Kpad) read as 0
    float alpha;
    unsigned flags;
    const float* bias;
    const float* residual;
    int64_t ldr;
    const uint16_t* gate;                // bf16 plane of the saved forward output (relu / dropout backward)
    int64_t ldg;
    float* colsum;                       // optional: colsum[col] += sum over rows of the epilogue value (bias gradient of the next layer back)
    float gate_scale;
    float drop_p;
    const uint64_t* rng;
    uint32_t site;
    float* ws;                           // split-K partials [split][tiles_m*128][tiles_n*128] (two-pass mode), or nullptr
    int nsplit;
    int nk_rg, nk_upw;                   // gemm_k128_kernel: row groups, 32-row units per row group
    int a_blk_n, a_blk_k;                // gemm_k128_kernel, block products (bmt_gemm_bf16_args.a_blk_n): output column block j reads A's column block j
    int rows_is_k;                       // rows_dev bounds the REDUCTION (A k-major: a weight gradient), not the output rows
    const int *c_row_dev, *m_dev;        // grouped launch, packed OUTPUT rows: C starts *c_row_dev rows lower, only *m_dev of the M rows exist
    const int* rows_dev;                 // packed rows (bmt_gemm_bf16_args.rows_dev): the rows actually present, in device memory; the launch is sized for M
                                         // (k-major A: for krows) and every kernel takes min(M, *rows_dev) -- graph-static grids over data-dependent extents
};

// LDS slot of (row, 16-byte slot) for rows of SPR slots
template <int SPR>
__device__ __forceinline__ int slot_of(int row, int s) {
    if constexpr (SPR == 8) return row * 8 + (s ^ ((row >> 1) & 7));
    else return row * 4 + (s ^ ((row >> 2) & 3));
}

// ---- operand tiles: global -> registers.  Every load is a buffer_load_dwordx4 through a wave-uniform descriptor: the per-lane byte
// offset (row, 16-byte slot) is computed ONCE per tile, the reduction offset of a stage travels in an SGPR, and rows past the
// operand's extent (k-major planes: reduction rows >= the true K) read as zero by the descriptor's bounds check -- no address
// arithmetic, clamp or select in the stage loop (it was 25 vector instructions per stage and wave, PMC: profiles/r02_e_*).
// Operand planes are therefore limited to 2 GiB each (checked by the host).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const uint16_t* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
template <int N>
__device__ __forceinline__ void plane_bload(const __amdgpu_buffer_rsrc_t rs, const int (&voff)[N], int soff, u32x4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0);
}
// per-lane byte offsets of the 16-byte slots thread tid moves (slot c = tid + NT i).  Row-major tile [ROWS rows][SPR slots]:
template <int SPR, int ROWS, int NT>
__device__ __forceinline__ void plane_voff(int64_t ld, int r0, int nrows, int tid, int (&vo)[ROWS * SPR / NT]) {
#pragma unroll
    for (int i = 0; i < ROWS * SPR / NT; ++i) {
        const int c = tid + NT * i;
        vo[i] = (int)((int64_t)min(r0 + c / SPR, nrows - 1) * ld * 2) + (c % SPR) * 16;
    }
}
// k-major tile [BK reduction rows][COLS columns] (reduction row of the stage added through the scalar offset); `shift`: extra rows
// (implicit Conv1d dW: the activation rows of this tile's tap)
template <int COLS, int BK, int NT>
__device__ __forceinline__ void plane_voff_km(int64_t ld, int c0, int shift, int tid, int (&vo)[COLS * BK / 8 / NT]) {
    constexpr int SPC = COLS / 8;
#pragma unroll
    for (int i = 0; i < COLS * BK / 8 / NT; ++i) {
        const int c = tid + NT * i;
        const int col = min(c0 + (c % SPC) * 8, (int)ld - 8);          // columns past the operand's extent: duplicates, discarded
        vo[i] = (int)((int64_t)(c / SPC + shift) * ld * 2) + col * 2;
    }
}
template <int SPR, int ROWS, int NT>
__device__ __forceinline__ void plane_lstore(u32x4* img, int tid, const u32x4 (&v)[ROWS * SPR / NT]) {
#pragma unroll
    for (int i = 0; i < ROWS * SPR / NT; ++i) {
        const int c = tid + NT * i;
        img[slot_of<SPR>(c / SPR, c % SPR)] = v[i];
    }
}

// ---- k-major operands (single-pass kernel only).  The operand is stored with the REDUCTION index as its row: for the weight
// gradient dW = dY^T . X both dY [rows, N] and X [rows, K] are used as they are, for dX = dY . W the weight plane [N, K] is.  A
// stage tile is [BK reduction rows][COLS columns], staged row-major with a padded row stride (COLS*2 + 64 bytes), and the MFMA
// fragment (8 reduction indices of one column per lane) is read with ds_read_b64_tr_b16 (semantics: attention_bf16.hip /
// tools/probes/tr_probe.hip).  This removes every transposed plane from the model: weights, activations and gradients are
// converted once, in one orientation.
template <int COLS> constexpr int km_rs() { return COLS * 2 + 64; }
// reduction indices per stage: 64 (128-byte rows: a full cache line per row and DMA request) wherever a ring of >= 2 stages fits
// the LDS; the two-plane products on the 128-row tile and the three-pass product stage 32
constexpr int gemm_bk(int npass, bool pipe, int ti) { return (npass == 1 || (npass == 2 && pipe && ti == 2)) ? 64 : 32; }
constexpr int pipe_ring(int stage_bytes, int ti) {
    const int budget = (ti == 2 ? 163840 : 81920) - 1024;
    const int r = budget / stage_bytes;
    return r > 4 ? 4 : (r < 2 ? 2 : r);
}
template <int COLS, int BK, int NT>
__device__ __forceinline__ void plane_lstore_km(char* img, int tid, const u32x4 (&v)[COLS * BK / 8 / NT]) {
    constexpr int SPC = COLS / 8;
#pragma unroll
    for (int i = 0; i < COLS * BK / 8 / NT; ++i) {
        const int c = tid + NT * i;
        *reinterpret_cast<u32x4*>(img + (c / SPC) * km_rs<COLS>() + (c % SPC) * 16) = v[i];
    }
}
// fragment for MFMA rows/cols [colbase, colbase + 32) and reduction indices [k16, k16 + 16): lane l -> column colbase + (l & 31),
// k = k16 + 8 (l >> 5) + j
template <int COLS>
__device__ __forceinline__ bf16x8 km_frag(const char* img, int k16, int colbase, int lane) {
    typedef short v4s16 __attribute__((ext_vector_type(4)));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    typedef v4s16 __attribute__((address_space(3))) * lds_v4s;
    const int m = lane & 15;
    const char* a = img + (k16 + 8 * (lane >> 5) + (m >> 2)) * km_rs<COLS>() + (colbase + 16 * ((lane >> 4) & 1) + 4 * (m & 3)) * 2;
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)a);
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a + 4 * km_rs<COLS>()));
    return __builtin_bit_cast(bf16x8, (v8s16)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// The same fragment from the UNPADDED image the LDS-DMA ring fills ([BK rows][128 columns], 256-byte rows): the four rows a 16-lane
// group of a transpose read touches would share their banks, so the 16-byte slots of a row are XOR-ed with 4 (row & 3) -- applied to
// the source address by the DMA and to the read address here.  row & 3 = (lane & 15) >> 2 for every fragment, so the swizzle folds
// into one per-lane byte offset per fragment column base (km_sw_off), the k16 step and the second read are immediates.
__device__ __forceinline__ int km_sw_off(int colbase, int lane) {
    const int m = lane & 15;
    const int cb = (colbase + 16 * ((lane >> 4) & 1) + 4 * (m & 3)) * 2;         // byte column of this lane's 8 bytes
    return (8 * (lane >> 5) + (m >> 2)) * 256 + (((cb >> 4) ^ ((m >> 2) << 2)) << 4) + (cb & 15);
}
__device__ __forceinline__ bf16x8 km_frag_sw(const char* img, int off, int k16) {
    typedef short v4s16 __attribute__((ext_vector_type(4)));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    typedef v4s16 __attribute__((address_space(3))) * lds_v4s;
    const char* a = img + off + k16 * 256;
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)a);
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a + 4 * 256));
    return __builtin_bit_cast(bf16x8, (v8s16)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// WM = 2: 128x128 tile, 4 waves, two workgroups per CU.  WM = 4: 256x128 tile, 8 waves, one workgroup per CU -- the same two
// waves per SIMD, but 3/4 of the operand bytes per FLOP: the 128x128 kernel moves ~7.5 TB/s of operand tiles L2 -> LDS at
// 29 % MFMA utilisation (profiles/r01_d_gemm_l2_pmc.txt), i.e. it is bound by the L2 -> CU fabric, not by L1, LDS or MFMA.
// TI = 32-row MFMA tiles per wave along M (wave tile 32 TI x 64): TI = 2 is the layout above; TI = 1 doubles the waves of a
// tile (more waves per SIMD to hide the LDS / barrier latency of the stage loop, 1.5x the fragment reads).
// CONV = 1: operand A is a halo-padded activation plane read with a per-stage row shift (Conv1d forward and dX);
// CONV = 2: operand B (k-major) is that plane read with a per-TILE row shift (Conv1d dW: output column block = tap).
// NPASS: 1 = A.B (one plane each); 2 = A.(Bh + Bl) (the activation as ONE plane, the weight split: the fp16 forward policy, two
// MFMA passes); 3 = split-bf16 (Ah.Bh + Ah.Bl + Al.Bh).  F16: the planes hold fp16 values (v_mfma_f32_32x32x16_f16).
// PIPE: the deep-pipelined main loop (row-major operands, 256 x 128 tile = WM 4, TI 2, one workgroup per CU): operand tiles go
// global -> LDS by LDS-DMA (global_load_lds, no staging registers, no ds_write), a ring of R stage buffers with R - 1 tiles in
// flight ACROSS the (single, raw) barrier of a stage, counted vmcnt -- cdna_hip_programming.md section 5 "pipelining across
// barriers".  Everything around the loop (tile order, split-K, epilogue) is shared with the register-staged loop.
// TAG: a distinct specialization per calling kernel template.  hipcc 7.2 (host pass) rejects the call of one and the same k-major
// PIPE specialization from a second kernel template with an unexplained "substitution failure"; the device code is identical.
template <int NPASS, int WM, int TI, bool AKM, bool BKM, int CONV = 0, bool F16 = false, bool PIPE = false, int TAG = 0>
__device__ __forceinline__ void gemm_bf16_tile(const GemmB& p, const int tile_id, const int split_id, const bool raw_order = false) {
    static_assert(NPASS == 1 || (!AKM && !BKM), "k-major operands: single-pass kernel only");
    static_assert(!PIPE || (WM == 4 && (CONV == 0 || CONV == 1) && NPASS <= 2 && (!(AKM || BKM) || (NPASS == 1 && (TI == 1 || (!AKM && BKM))))),
                  "pipelined loop: 8 waves; k-major operands on the one-plane 128-row tile");
    static_assert(CONV == 0 || (CONV == 1 && !AKM && !BKM) || (CONV == 2 && AKM && BKM), "conv modes: row-major A, or k-major A and B");
    constexpr int BK = gemm_bk(NPASS, PIPE, TI);
    constexpr bool ALO = NPASS == 3, BLO = NPASS >= 2;
    constexpr int SPR = BK / 8;
    constexpr int BM = 32 * TI * WM, NT = 128 * WM;       // WM waves along M x 2 along N
    // bytes of one A / B plane tile (k-major: [BK rows][columns], padded rows in the register-staged loop, swizzled rows in the DMA ring)
    constexpr int PA = (AKM && !PIPE) ? BK * km_rs<BM>() : BM * BK * 2, PBB = (BKM && !PIPE) ? BK * km_rs<BN>() : BN * BK * 2;
    constexpr int STAGE_BYTES = (ALO ? 2 : 1) * PA + (BLO ? 2 : 1) * PBB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // packed rows: the rows (row-major A) / reduction rows (k-major A: the weight gradients) actually present come from device memory; the
    // tile order is built over the tiles that exist, so that the workgroups that leave at once are the LAST of every XCD's share
    int Mr = p.M, krows = p.krows, Kp = p.Kpad, tiles_m = p.tiles_m;
    if (p.rows_dev != nullptr) {
        const int nrows = *p.rows_dev;
        if constexpr (AKM) {
            krows = min(krows, nrows);
            if (raw_order) Kp = (krows + 63) & ~63;        // (the grouped launch accumulates with atomics: a chunk past the rows adds nothing)
        } else {
            Mr = min(Mr, nrows);
            tiles_m = (Mr + BM - 1) / BM;
        }
    }
    // tile order: XCD remap, then groups of 8 row panels walked column by column
    const int ntiles = tiles_m * p.tiles_n;
    if (!raw_order && tile_id >= ntiles) return;
    const int w = raw_order ? tile_id : xcd_remap(tile_id, ntiles);      // raw: the caller already placed this tile on its XCD
    const int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int g = w / per_group;
    const int first_m = g * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int wi = w - g * per_group;
    const int tm = first_m + wi % gsz, tn = wi / gsz;
    const int m0 = tm * BM;
    if constexpr (AKM) {
        if (p.m_dev != nullptr) {            // (grouped launch: the output's rows are data too)
            Mr = min(Mr, *p.m_dev);
            if (m0 >= Mr) return;
        }
    }
    int n0_ = tn * BN;
    if constexpr (CONV == 2) {
        // Conv1d dW: output column block = (tap, 128 channels).  Walk the TAPS of one channel block before the next channel block, so that the
        // tiles an XCD runs together read the same rows of the activation plane, shifted by a tap each: one fetch serves a whole run of taps
        // (tap-major order shared a block between two taps only: the launch fetched 3.5 GB at 4.6 TB/s, profiles/r04_w_prop_pmc_traffic.json)
        const int nb = p.conv_cin / BN;
        if (p.conv_tap_minor && nb > 0 && p.tiles_n % nb == 0) {
            const int taps = p.tiles_n / nb;
            n0_ = (tn % taps) * p.conv_cin + (tn / taps) * BN;
        }
    }
    const int n0 = n0_;
    int kbeg_ = split_id * p.kchunk;
    int kend_ = min(Kp, kbeg_ + p.kchunk);
    int n0b_ = n0;                       // the tile's first column in B
    if constexpr (!AKM && BKM && !PIPE && CONV == 0) {
        // block products (bmt_gemm_bf16_args.a_blk_n, B k-major): output column block j = A[:, j a_blk_k ...] . B[j a_blk_k ..., 0 : a_blk_n] -- the
        // tile reduces over ITS block's rows of B (= columns of A) and reads B's columns from the start
        if (p.a_blk_n > 0) {
            const int j = n0 / p.a_blk_n;
            kbeg_ = j * p.a_blk_k;
            kend_ = kbeg_ + p.a_blk_k;
            n0b_ = n0 - j * p.a_blk_n;
        }
    }
    const int kbeg = kbeg_, kend = kend_, n0b = n0b_;
    if (kbeg >= kend) return;            // (packed rows, grouped launch: this reduction chunk lies past the rows present)

    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int kma_off[TI], kmb_off[2];          // k-major fragments of the DMA ring: per-lane byte offsets (km_sw_off)
    if constexpr (PIPE && AKM) {
#pragma unroll
        for (int i = 0; i < TI; ++i) kma_off[i] = km_sw_off(wr * 32 * TI + i * 32, lane);
    }
    if constexpr (PIPE && BKM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) kmb_off[j] = km_sw_off(wc * 64 + j * 32, lane);
    }
    // Two register sets: the global loads of stage t+2 are issued before the MFMAs of stage t, so every load has two
    // iterations (two barriers) to land -- with 2 workgroups per CU and ~0.2 us of MFMA work per stage a single stage of
    // prefetch leaves the loop waiting on HBM/L2 latency (profiles/r01_c: 16 iterations of a 24-tile GEMM took 50 us).
    constexpr int NRA = BM * SPR / NT, NRB = BN * SPR / NT;
    u32x4 ra0[NRA], rb0[NRB], ral0[NRA], rbl0[NRB];
    u32x4 ra1[NRA], rb1[NRB], ral1[NRA], rbl1[NRB];
    // per-lane byte offsets of this thread's slots (loop invariant) and the descriptors of the operand planes
    int avo[NRA], bvo[NRB];
    if constexpr (!PIPE) {
        if constexpr (CONV == 1) {           // halo-padded activation plane: output row m = b * S + s reads plane rows m + 2 b * halo (+ tap, per stage)
#pragma unroll
            for (int i = 0; i < NRA; ++i) {
                const int c = tid + NT * i;
                const int row = min(m0 + c / SPR, Mr - 1);
                avo[i] = (int)((int64_t)(row + 2 * (row / p.conv_S) * p.conv_halo) * p.lda * 2) + (c % SPR) * 16;
            }
        } else if constexpr (AKM) plane_voff_km<BM, BK, NT>(p.lda, m0, 0, tid, avo);
        else plane_voff<SPR, BM, NT>(p.lda, m0, Mr, tid, avo);
        if constexpr (CONV == 2) plane_voff_km<BN, BK, NT>(p.ldb, n0 % p.conv_cin, n0 / p.conv_cin, tid, bvo);
        else if constexpr (BKM) plane_voff_km<BN, BK, NT>(p.ldb, n0b, 0, tid, bvo);
        else plane_voff<SPR, BN, NT>(p.ldb, n0, p.N, tid, bvo);
    }
    // extents: row-major planes end after their last row; k-major planes after reduction row K (later rows read as zero); the
    // Conv1d activation plane after the rows reachable from its (advanced) base pointer
    const int64_t a_bytes = (CONV == 1 ? (int64_t)p.conv_rows : (AKM ? (int64_t)krows : (int64_t)Mr)) * p.lda * 2;
    const int64_t b_bytes = (CONV == 2 ? (int64_t)p.conv_rows : (BKM ? (int64_t)krows : (int64_t)p.N)) * p.ldb * 2;
    const __amdgpu_buffer_rsrc_t rsAh = plane_rsrc(p.Ah, a_bytes), rsAl = plane_rsrc(ALO ? p.Al : p.Ah, a_bytes);
    const __amdgpu_buffer_rsrc_t rsBh = plane_rsrc(p.Bh, b_bytes), rsBl = plane_rsrc(BLO ? p.Bl : p.Bh, b_bytes);
    // stage image: A hi | B hi | [A lo] | [B lo]
#define stage_ptr(buf_, which_) reinterpret_cast<u32x4*>(smem + (buf_) * STAGE_BYTES + ((which_) == 0 ? 0 : (which_) == 1 ? PA : (which_) == 2 ? PA + PBB : (ALO ? 2 : 1) * PA + PBB))
#define BMT_GLOAD(s_, RA, RB, RAL, RBL)                                                   \
    do {                                                                                  \
        const int k_ = min(kbeg + (s_) * BK, kend - BK);   /* branch-free tail: re-fetch the last stage */ \
        /* scalar byte offset of the stage: k columns (row-major), k rows (k-major), (tap rows, channel) (Conv1d) */ \
        const int sa_ = (CONV == 1) ? ((k_ / p.conv_cin) * (int)p.lda + k_ % p.conv_cin) * 2 : (AKM ? k_ * (int)p.lda * 2 : k_ * 2); \
        const int sb_ = BKM ? k_ * (int)p.ldb * 2 : k_ * 2;                               \
        plane_bload<NRA>(rsAh, avo, sa_, RA);                                             \
        plane_bload<NRB>(rsBh, bvo, sb_, RB);                                             \
        if constexpr (ALO) plane_bload<NRA>(rsAl, avo, sa_, RAL);                         \
        if constexpr (BLO) plane_bload<NRB>(rsBl, bvo, sb_, RBL);                         \
    } while (0)
#define BMT_LSTORE(buf_, RA, RB, RAL, RBL)                                                \
    do {                                                                                  \
        if constexpr (AKM) plane_lstore_km<BM, BK, NT>(reinterpret_cast<char*>(stage_ptr(buf_, 0)), tid, RA);  \
        else plane_lstore<SPR, BM, NT>(stage_ptr(buf_, 0), tid, RA);                      \
        if constexpr (BKM) plane_lstore_km<BN, BK, NT>(reinterpret_cast<char*>(stage_ptr(buf_, 1)), tid, RB);  \
        else plane_lstore<SPR, BN, NT>(stage_ptr(buf_, 1), tid, RB);                      \
        if constexpr (ALO) plane_lstore<SPR, BM, NT>(stage_ptr(buf_, 2), tid, RAL);       \
        if constexpr (BLO) plane_lstore<SPR, BN, NT>(stage_ptr(buf_, 3), tid, RBL);       \
    } while (0)
#define BMT_COMPUTE(buf_)                                                                 \
    do {                                                                                  \
        const u32x4* sAh = stage_ptr(buf_, 0);                                            \
        const u32x4* sBh = stage_ptr(buf_, 1);                                            \
        const u32x4* sAl = stage_ptr(buf_, 2);                                            \
        const u32x4* sBl = stage_ptr(buf_, 3);                                            \
        __builtin_amdgcn_s_setprio(1);                                                    \
        _Pragma("unroll") for (int s = 0; s < BK / 16; ++s) {                             \
            const int sl = 2 * s + half;                                                  \
            bf16x8 ah[TI], bh[2], al[TI], bl[2];                                          \
            _Pragma("unroll") for (int i = 0; i < TI; ++i) {                              \
                const int ia = slot_of<SPR>(wr * 32 * TI + i * 32 + l31, sl);             \
                if constexpr (AKM && PIPE) ah[i] = km_frag_sw(reinterpret_cast<const char*>(sAh), kma_off[i], 16 * s);  \
                else if constexpr (AKM) ah[i] = km_frag<BM>(reinterpret_cast<const char*>(sAh), 16 * s, wr * 32 * TI + i * 32, lane); \
                else ah[i] = as_bf16x8(sAh[ia]);                                          \
                if constexpr (ALO) al[i] = as_bf16x8(sAl[ia]);                            \
            }                                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                               \
                const int ib = slot_of<SPR>(wc * 64 + i * 32 + l31, sl);                  \
                if constexpr (BKM && PIPE) bh[i] = km_frag_sw(reinterpret_cast<const char*>(sBh), kmb_off[i], 16 * s);  \
                else if constexpr (BKM) bh[i] = km_frag<BN>(reinterpret_cast<const char*>(sBh), 16 * s, wc * 64 + i * 32, lane); \
                else bh[i] = as_bf16x8(sBh[ib]);                                          \
                if constexpr (BLO) bl[i] = as_bf16x8(sBl[ib]);                            \
            }                                                                             \
            _Pragma("unroll") for (int i = 0; i < TI; ++i)                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                           \
                    if constexpr (ALO) acc[i][j] = mfma32t<F16>(al[i], bh[j], acc[i][j]); \
                    if constexpr (BLO) acc[i][j] = mfma32t<F16>(ah[i], bl[j], acc[i][j]); \
                    acc[i][j] = mfma32t<F16>(ah[i], bh[j], acc[i][j]);                    \
                }                                                                         \
        }                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                    \
    } while (0)

    if constexpr (PIPE) {
        // ---- ring geometry: one step = BK reduction indices of A [BM rows] | B hi [128 rows] | (B lo); 1 KB pieces, one per
        // wave-instruction (64 lanes x 16 B, LDS destination = piece base + 16 lane); the XOR swizzle of the 16-byte slots is applied
        // on the SOURCE side (lane -> which k-slot of its row it fetches), so the image is exactly what slot_of<SPR> reads.
        // Loads are buffer_load ... lds: the per-lane byte offset is loop invariant, the reduction offset advances in an SGPR and
        // the LDS address goes through M0 from scalar arithmetic -- no vector instruction per load.  The step loop is unrolled by
        // the ring depth so that every LDS address of a step is a loop-invariant register plus an immediate.
        constexpr int RPP = 1024 / (BK * 2);                   // rows per piece: 16 (BK 32) / 8 (BK 64)
        constexpr int PA_PIECES = BM / RPP, PB_PIECES = BN / RPP;
        constexpr int APW = PA_PIECES / 8, BPW = PB_PIECES / 8; // pieces per wave
        constexpr int LPT = APW + (BLO ? 2 : 1) * BPW;          // LDS-DMA instructions per thread per step
        constexpr int R = pipe_ring(STAGE_BYTES, TI);
        const int wid_s = __builtin_amdgcn_readfirstlane(wid);
        const int rl = lane / SPR, sp = lane % SPR;             // row within the piece, slot POSITION within the row
        // (k-major operand: a piece is 4 reduction rows x 256 B; lane -> row lane >> 4, slot position lane & 15, source slot
        // position ^ 4 (row & 3); the descriptor ends after reduction row K, later rows read as zero; columns past the operand's
        // extent are duplicates of its last 8, discarded by the epilogue)
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ah, 0, (int)a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)b_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsBl = __builtin_amdgcn_make_buffer_rsrc((void*)(BLO ? p.Bl : p.Bh), 0, (int)b_bytes, 0x00020000);
        int avo[APW], bvo[BPW];
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            if constexpr (AKM) {
                const int kr = lane >> 4, sl = (lane & 15) ^ (kr << 2);
                avo[i] = ((wid_s * APW + i) * 4 + kr) * (int)p.lda * 2 + min(m0 + sl * 8, (int)p.lda - 8) * 2;
            } else {
                const int row = (wid_s * APW + i) * RPP + rl;
                const int ks = (SPR == 8) ? (sp ^ ((row >> 1) & 7)) : (sp ^ ((row >> 2) & 3));
                int grow = min(m0 + row, Mr - 1);
                if constexpr (CONV == 1) grow += 2 * (grow / p.conv_S) * p.conv_halo;      // output row b S + s reads plane row (+ tap, per step)
                avo[i] = (int)((int64_t)grow * p.lda * 2) + ks * 16;
            }
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            if constexpr (BKM) {
                const int kr = lane >> 4, sl = (lane & 15) ^ (kr << 2);
                bvo[i] = ((wid_s * BPW + i) * 4 + kr) * (int)p.ldb * 2 + min(n0 + sl * 8, (int)p.ldb - 8) * 2;
            } else {
                const int row = (wid_s * BPW + i) * RPP + rl;
                const int ks = (SPR == 8) ? (sp ^ ((row >> 1) & 7)) : (sp ^ ((row >> 2) & 3));
                bvo[i] = (int)((int64_t)min(n0 + row, p.N - 1) * p.ldb * 2) + ks * 16;
            }
        }
        typedef __attribute__((address_space(3))) void* lptr_t;
#define BMT_DMA(step_, slot_)                                                                                    \
        do {                                                                                                     \
            char* base_ = smem + (slot_) * STAGE_BYTES;                                                          \
            const int k_ = kbeg + (step_) * BK;                                                                  \
            const int so_ = (CONV == 1) ? ((k_ / p.conv_cin) * (int)p.lda + k_ % p.conv_cin) * 2 : (AKM ? k_ * (int)p.lda * 2 : k_ * 2);  \
            const int sob_ = BKM ? k_ * (int)p.ldb * 2 : k_ * 2;                                                 \
            _Pragma("unroll") for (int i = 0; i < APW; ++i)                                                      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base_ + (wid_s * APW + i) * 1024), 16, avo[i], so_, 0, 0);          \
            _Pragma("unroll") for (int i = 0; i < BPW; ++i)                                                      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(base_ + PA + (wid_s * BPW + i) * 1024), 16, bvo[i], sob_, 0, 0);    \
            if constexpr (BLO) {                                                                                 \
                _Pragma("unroll") for (int i = 0; i < BPW; ++i)                                                  \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBl, (lptr_t)(base_ + PA + PBB + (wid_s * BPW + i) * 1024), 16, bvo[i], sob_, 0, 0); \
            }                                                                                                    \
        } while (0)
#define BMT_VMWAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
        // one step: tile `step_` (in ring slot `slot_`, a compile-time constant) has landed once at most min(R - 2, later tiles)
        // younger tiles are still in flight; after the barrier every wave's pieces of it are in LDS and every wave is done reading
        // tile step_ - 1, whose slot the next DMA overwrites
#define BMT_STEP(step_, slot_)                                                                                   \
        do {                                                                                                     \
            const int younger_ = min(R - 2, niter - 1 - (step_));                                                \
            if (younger_ >= 2) BMT_VMWAIT(2 * LPT);                                                              \
            else if (younger_ == 1) BMT_VMWAIT(LPT);                                                             \
            else BMT_VMWAIT(0);                                                                                  \
            __builtin_amdgcn_s_barrier();                                                                        \
            if ((step_) + R - 1 < niter) BMT_DMA((step_) + R - 1, ((slot_) + R - 1) % R);      \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            BMT_COMPUTE(slot_);                                                              \
        } while (0)
        const int niter = (kend - kbeg) / BK;
#pragma unroll
        for (int s0 = 0; s0 < R - 1; ++s0)
            if (s0 < niter) BMT_DMA(s0, s0);
        int t = 0;
        for (; t + R <= niter; t += R) {
#pragma unroll
            for (int r = 0; r < R; ++r) BMT_STEP(t + r, r);
        }
#pragma unroll
        for (int r = 0; r < R - 1; ++r)
            if (t + r < niter) BMT_STEP(t + r, r);
#undef BMT_STEP
#undef BMT_DMA
#undef BMT_VMWAIT
        __syncthreads();      // the epilogue reuses the stage buffers
    } else {
    const int niter = (kend - kbeg) / BK;     // >= 1: the host never launches an empty split
    BMT_GLOAD(0, ra0, rb0, ral0, rbl0);
    BMT_GLOAD(1, ra1, rb1, ral1, rbl1);
    BMT_LSTORE(0, ra0, rb0, ral0, rbl0);
    __syncthreads();
    int t = 0;
    // invariant at the top of a pair: LDS buffer 0 = stage t, set 1 = stage t+1 (in flight), set 0 free
    for (; t + 2 <= niter; t += 2) {
        BMT_GLOAD(t + 2, ra0, rb0, ral0, rbl0);
        __builtin_amdgcn_sched_barrier(0);    // keep the loads ahead of the MFMAs (the scheduler otherwise sinks them below)
        BMT_COMPUTE(0);
        BMT_LSTORE(1, ra1, rb1, ral1, rbl1);
        __syncthreads();
        BMT_GLOAD(t + 3, ra1, rb1, ral1, rbl1);
        __builtin_amdgcn_sched_barrier(0);
        BMT_COMPUTE(1);
        BMT_LSTORE(0, ra0, rb0, ral0, rbl0);
        __syncthreads();
    }
    if (t < niter) BMT_COMPUTE(0);
    __syncthreads();      // the epilogue reuses the stage buffers

    }
    // ---------------- split-K, two passes: each split stores its raw partial tile (plain coalesced fp32 stores) into
    // ws[split][Mpad][Npad]; splitk_epilogue_kernel sums the splits and runs the epilogue.  Used for GEMMs with too few tiles to
    // fill the chip and a long reduction (the k-loop is a chain of dependent ~2 us memory round trips: 24 tiles x 16 stages is
    // ~50 us of latency on 24 CUs, 24 x 8 splits of 2 stages is ~10 us on 192) and for the weight gradients (reduction over
    // B*S rows), where it replaces tiles*splits*16K fp32 atomics by streaming traffic.
    if (p.ws != nullptr) {
        const int64_t ldw = (int64_t)p.tiles_n * BN;
        float* part = p.ws + (int64_t)split_id * p.tiles_m * BM * ldw;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    part[(int64_t)(m0 + wr * 32 * TI + i * 32 + acc_row(r, half)) * ldw + n0 + wc * 64 + j * 32 + l31] = acc[i][j][r];
        return;
    }
    // ---------------- epilogue (order: alpha, bias, dropout_pre, relu, dropout_post, gate, residual), in row-segment form.
    // The accumulators go through the (now idle) stage buffers as an fp32 [BM][128] tile; then each thread owns 8 consecutive
    // columns (fixed for the thread: bias and column sums live in registers) and walks rows.  Every global access of a segment is
    // a 16-byte vector (C: 2 x float4, planes: 8 x 16 bit, gate plane, residual), the flags are tested once per segment instead of
    // once per element, and there is one bounds decision per segment.  (The element-wise form this replaces -- one divergent
    // row / column test, six flag tests and a 64-bit index per accumulator register -- took 35 % of a K = 1024 product:
    // the probes of round 2.)
    if ((p.flags == BMT_EPI_ACCUM || (p.flags == 0 && p.c_row_dev != nullptr)) && !p.Chi) {
        float* const Cacc = p.c_row_dev ? p.C + (int64_t)(*p.c_row_dev) * p.ldc : p.C;
        // C += alpha * acc and nothing else (weight gradients, possibly several products into one buffer): atomics straight from the
        // accumulators -- 32 lanes of an instruction hit 32 consecutive floats of a row, which the L2 handles as one 128-byte request.
        // flags 0 with a placed output (grouped launch, an unsplit reduction: ONE writer per element -- the gradient of a packed encoder
        // memory): C = alpha * acc by plain stores, so that nobody has to zero the output first (round 6)
        const bool add = p.flags == BMT_EPI_ACCUM;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wc * 64 + j * 32 + l31;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wr * 32 * TI + i * 32 + acc_row(r, half);
                    if (row < Mr) {
                        if (add) atomicAdd(Cacc + (int64_t)row * p.ldc + col, acc[i][j][r] * p.alpha);
                        else Cacc[(int64_t)row * p.ldc + col] = acc[i][j][r] * p.alpha;
                    }
                }
            }
        return;
    }
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(wr * 32 * TI + i * 32 + acc_row(r, half)) * BN + wc * 64 + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    const unsigned f = p.flags;
    const int cg = (tid & 15) * 8;
    const int col = n0 + cg;
    const int pcols = p.Chi ? p.plane_cols : 0;
    const bool active = col < p.N || col < pcols;
    const bool full = col + 8 <= p.N;                                         // all 8 columns inside the product
    const bool c_vec = p.C && full && ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
    const bool r_vec = full && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0);
    const bool g_vec = full && ((p.ldg & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.gate) & 15) == 0);
    float bv[8], cs8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        bv[q] = ((f & BMT_EPI_BIAS) && col + q < p.N) ? p.bias[col + q] : 0.f;
        cs8[q] = 0.f;
    }
#pragma unroll 2
    for (int ps = 0; ps < BM * 16 / NT; ++ps) {
        const int rl = ps * (NT / 16) + (tid >> 4);
        const int row = m0 + rl;
        if (row >= Mr || !active) continue;
        float v[8];
        {
            const float4 t0 = *reinterpret_cast<const float4*>(ct + rl * BN + cg);
            const float4 t1 = *reinterpret_cast<const float4*>(ct + rl * BN + cg + 4);
            v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        }
        const int64_t idx = (int64_t)row * p.ldc + col;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = v[q] * p.alpha + bv[q];
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
        if (f & BMT_EPI_GATE) {            // keep an element iff the saved forward output is non-zero (sign bit ignored)
            const uint16_t* gp = p.gate + (int64_t)row * p.ldg + col;
            if (g_vec) {
                const u32x4 gv = *reinterpret_cast<const u32x4*>(gp);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = (gv[q] & 0x00007FFFu) ? v[2 * q] * p.gate_scale : 0.f;
                    v[2 * q + 1] = (gv[q] & 0x7FFF0000u) ? v[2 * q + 1] * p.gate_scale : 0.f;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (col + q < p.N && (gp[q] & 0x7fffu)) ? v[q] * p.gate_scale : 0.f;
            }
        }
        if (f & BMT_EPI_RESIDUAL) {
            const float* rp = p.residual + (int64_t)row * p.ldr + col;
            if (r_vec) {
                const float4 t0 = *reinterpret_cast<const float4*>(rp), t1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (col + q < p.N) v[q] += rp[q];
            }
        }
        if (!full) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (col + q >= p.N) v[q] = 0.f;                               // plane columns past N hold zeros
        }
        if (f & BMT_EPI_ACCUM) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (col + q < p.N) atomicAdd(p.C + idx + q, v[q]);
        } else if (c_vec) {
            *reinterpret_cast<float4*>(p.C + idx) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(p.C + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (p.C) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (col + q < p.N) p.C[idx + q] = v[q];
        }
        if (p.colsum) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cs8[q] += v[q];
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
    if (p.colsum) {                  // uniform per launch.  Lanes 16 apart share a column group: fold them, then the waves through LDS
        __syncthreads();             // every staged row has been read
        float* cs = reinterpret_cast<float*>(smem);      // [NT / 64][128]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float t = cs8[q];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (lane < 16) cs[wid * BN + cg + q] = t;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float t = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < NT / 64; ++w_) t += cs[w_ * BN + tid];
            atomicAdd(p.colsum + n0 + tid, t);
        }
    }
}

template <int NPASS, int WM, int TI, bool AKM, bool BKM, int CONV = 0, bool F16 = false>
__global__ __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(TI == 2 ? 2 : 4, TI == 2 ? 2 : 4))) void gemm_bf16_kernel(const GemmB p) {
    gemm_bf16_tile<NPASS, WM, TI, AKM, BKM, CONV, F16>(p, blockIdx.x, blockIdx.y);
}
template <int NPASS, bool F16, int TI, bool AKM = false, bool BKM = false, int CONV = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(TI == 2 ? 2 : 4, TI == 2 ? 2 : 4))) void gemm_pipe_kernel(const GemmB p) {
    // persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8, so a workgroup stays on
    // the XCD its tiles were ordered for); its stores drain while the next tile's operands are already on their way
    int ntiles = p.tiles_m * p.tiles_n;
    if (p.rows_dev != nullptr && !AKM) ntiles = ((min(p.M, *p.rows_dev) + 128 * TI - 1) / (128 * TI)) * p.tiles_n;      // packed rows: the tiles that exist
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        gemm_bf16_tile<NPASS, 4, TI, AKM, BKM, CONV, F16, true, 1>(p, t, (int)blockIdx.y, false);
        __syncthreads();          // the stage buffers (epilogue tile) are free again
    }
}


// ===================================================================== 256 x 256 tile, 8 waves, two wave groups in ping-pong
// The 128-row tiles above are bound by the L2 -> LDS path (tools/probes/gemm_exp.py: DMA alone 46 us, MFMA + LDS alone 55 us, both
// 73 us for 8192 x 4096 x 1024; ~23 TB/s of operand tiles) and by LDS reads (1.5 KB per 32x32x16 MFMA).  This kernel follows the
// 256 x 256 "8-phase" structure of cdna_hip_programming.md: half the operand bytes per FLOP, 0.75 KB of fragment reads per MFMA.
//   * C^T = W . X^T: the weight rows take the MFMA's A role, the activation rows the B role, so a lane's accumulator registers
//     are consecutive OUTPUT COLUMNS of one output row (4 per register group, 8 after one v_permlane32_swap): the epilogue
//     stores 16-byte row segments straight from registers -- no LDS round trip, no barrier;
//   * 8 waves = 2 (W halves of 128 rows: the two groups) x 4 (64 activation rows each); wave tile 128 x 64 = 4 x 2 accumulators
//     of 32 x 32; a K-tile of 64 is worked in 4 phases of 8 MFMAs (quadrants (w0,x0) (w0,x1) (w1,x1) (w1,x0): 8 + 4, 4, 8, 0
//     fragment reads), every phase = {fragment reads + this phase's share of the next K-tile's LDS-DMA} barrier {MFMAs} barrier.
//     Group 1 runs one barrier behind group 0, so on every SIMD one wave issues MFMAs while the other reads -- the matrix pipe
//     never waits for a barrier or for LDS latency of its own wave;
//   * LDS: 2 K-tile slots x {W0, W1, X0, X1} half-tiles of [128 rows][64 k] (16 KB, XOR-swizzled 16-byte slots, filled by
//     buffer_load ... lds with the swizzle on the source side) = 128 KB; K-tile t+1 is requested during phases 0 and 1 of K-tile t
//     and waited for (vmcnt(0): nothing younger is in flight) in phase 3, one barrier before its first read;
//   * NPASS 2 (activation fp16 x weight fp16 hi + lo) runs the K loop twice over the activation with the second weight plane --
//     the same instruction stream, 2 K / 64 K-tiles.
template <bool F16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wide_kernel(const GemmB p) {
    constexpr int HT = 16384, SLOT = 4 * HT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int half = lane >> 5, l31 = lane & 31;

    // tile order: XCD remap, then groups of 8 activation panels walked panel-first (the ~32 tiles an XCD runs together share
    // 8 activation panels and 4 weight panels in its L2)
    int Mr = p.M, tiles_m = p.tiles_m;                        // activation / weight panels of 256 rows
    const int tiles_n = p.tiles_n;
    if (p.rows_dev != nullptr) {                              // packed rows: the rows present are in device memory (see gemm_bf16_tile)
        Mr = min(Mr, *p.rows_dev);
        tiles_m = (Mr + 255) / 256;
    }
    if ((int)blockIdx.x >= tiles_m * tiles_n) return;
    const int w = xcd_remap((int)blockIdx.x, tiles_m * tiles_n);
    const int per_group = 8 * tiles_n;
    const int g = w / per_group, first_m = g * 8;
    const int gsz = min(tiles_m - first_m, 8);
    const int wi = w - g * per_group;
    const int m0 = (first_m + wi % gsz) * 256, n0 = (wi / gsz) * 256;

    // ---- LDS-DMA: a half-tile is 16 pieces of 1 KB (8 rows x 128 B); wave w fills pieces 2w, 2w+1 (rows 16w .. 16w+15)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ah, 0, (int)((int64_t)Mr * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)((int64_t)p.N * p.ldb * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bl ? p.Bl : p.Bh), 0, (int)((int64_t)p.N * p.ldb * 2), 0x00020000);
    int xvo[2], wvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * wid + 8 * j + (lane >> 3);
        const int ks = (lane & 7) ^ ((row >> 1) & 7);
        xvo[j] = row * (int)p.lda * 2 + ks * 16;
        wvo[j] = row * (int)p.ldb * 2 + ks * 16;
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
                                                         (n0 + 128 * hf) * (int)p.ldb * 2 + k_, 0, 0);              \
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
    int offW[4], offX[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int o = l31 * 128 + (((2 * s + half) ^ ((l31 >> 1) & 7)) * 16);
        offW[s] = o + wr * HT;
        offX[s] = o + (2 + (wc >> 1)) * HT + (wc & 1) * 8192;
    }
#define BMT_W_FRAG(slot_, off_, i_) as_bf16x8(*reinterpret_cast<const u32x4*>(smem + (slot_) * SLOT + (off_) + (i_) * 4096))
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
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) wa[i][s] = BMT_W_FRAG(e_, offW[s], i);                     \
        BMT_W_LGKM0();                                                                                               \
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
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) wa[i][s] = BMT_W_FRAG(e_, offW[s], 2 + i);                 \
        if (next2_) BMT_W_DMA_X((t_) + 2, e_);                                                                       \
        BMT_W_LGKM0();                                                                                               \
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
    if (wr == 1) BMT_W_BAR();                  // group 1 runs one barrier behind group 0
    for (int t = 0; t < T; t += 2) {
        BMT_W_KTILE(0, t);
        if (t + 1 < T) BMT_W_KTILE(1, t + 1);
    }
    if (wr == 0) BMT_W_BAR();
#undef BMT_W_LGKM0
#undef BMT_W_KTILE
#undef BMT_W_MFMA
#undef BMT_W_BAR
#undef BMT_W_FRAG
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
        const bool ok = row < Mr && col + 8 <= p.N;
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
            if (row >= Mr || (col >= p.N && col >= pcols)) continue;
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
}

// ===================================================================== reduction of 128 (the audio stream's projections: d_model_audio = 128)
// M x N outputs from a reduction of 128: 2 FLOP per output byte-pair -- these launches are bound by WRITING the result (25600 x 3072 fp16
// = 157 MB for the fused audio q|k|v), and the tile kernels above spend 102 us on it (1.5 TB/s: every 128 x 128 tile re-stages 64 KB of
// weight planes for 32 KB of output, and its load / MFMA / store phases do not overlap; profiles/r03_v_last_step_trace.csv).  Here
//   * a workgroup keeps a CHUNK of the weight (128 rows x 128 k, both planes: 64 KB) in LDS for its whole life (one LDS-DMA burst,
//     XOR-swizzled 16-byte slots), and walks down its share of the activation rows;
//   * a wave owns a 32-row unit: its activation fragments (8 k-steps x 16 bytes per lane) come straight from global memory into
//     registers -- they are used for every column block of the chunk -- requested by hand TWO units ahead with a counted vmcnt;
//   * per 32-column block: 8 (or 16: second weight plane) MFMAs on one accumulator that starts as the bias; two blocks go through the
//     wave's private 8-KB LDS chunk together and leave as 8 rows x 64 columns per store instruction -- whole 128-byte lines of a 16-bit
//     plane, 256 bytes of a row of C (16 rows x 32 columns, 64-byte half lines, wrote at ~2.5 TB/s; round 4);
//   * after the initial barrier the 8 waves never synchronise: stores, LDS turns and MFMAs of different waves overlap freely.
// As in gemm_wide_kernel the weight rows take the MFMA's A role (C^T = W . X^T), so a lane's accumulator registers are consecutive
// output columns of one output row.
// Memory operations of a wave retire IN ORDER on this architecture (one vmcnt for loads and stores): a wait for any load issued after
// a store is a wait for that store's acknowledgement from L2 / HBM -- several microseconds when every CU is writing -- hence the counted
// waits that leave the stores of the last two units in flight.  (Round 3 also ran a variant with seven computing waves and a loading wave;
// the eight-wave form with hand-counted prefetch measured faster and is the one kept: profiles/r03_w_k128_pmc_*.csv.)
// Instruction diet: the column-block loop is unrolled (LDS offsets are immediates), the plane format is branched on, not selected, 32-bit
// store offsets, the next block's fragments are requested before this block's epilogue.
template <bool F16, bool TWO, int NCB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_k128_kernel(const GemmB p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int NCW = 32 * NCB, PLANE = NCW * 256, NPL = TWO ? 2 : 1;
    constexpr int CKB = 8192;                                                                    // a wave's turn chunk: [32 rows][64 columns] fp32 (two blocks)
    constexpr int CK_OFF = NPL * PLANE, BIAS_OFF = CK_OFF + 8 * CKB;                              // LDS: weight chunk | turn chunks | bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nch = p.tiles_n;
    // workgroups of one row group (they read the same activation rows) are neighbours in the remapped order: same XCD, same L2
    const int w = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int rg = w / nch, chunk = w - rg * nch;
    const int n0 = chunk * NCW;
    const int a_col = p.a_blk_n > 0 ? (n0 / p.a_blk_n) * p.a_blk_k : 0;      // block products: this chunk's columns of A
    const int lrow = lane >> 4, lslot = lane & 15;          // LDS-DMA: an instruction fills 4 rows of 256 B; lane = (row, 16-byte slot)

    // ---- the weight chunk: pieces of 4 rows (1 KB); rows >= N are outside the descriptor (zeros)
    {
        const __amdgpu_buffer_rsrc_t rsWh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)((int64_t)p.N * p.ldb * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsWl = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? p.Bl : p.Bh), 0, (int)((int64_t)p.N * p.ldb * 2), 0x00020000);
        for (int pc = wid; pc < NCW / 4; pc += 8) {
            const int row = 4 * pc + lrow;
            const int vo = (n0 + row) * (int)p.ldb * 2 + ((lslot ^ (row & 15)) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWh, (lptr_t)(smem + pc * 1024), 16, vo, 0, 0, 0);
            if constexpr (TWO) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWl, (lptr_t)(smem + PLANE + pc * 1024), 16, vo, 0, 0, 0);
        }
    }
    float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
    const unsigned f = p.flags;
    if (tid < NCW) sbias[tid] = ((f & BMT_EPI_BIAS) && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;

    int Mr = p.M, upw = p.nk_upw;
    if (p.rows_dev != nullptr) {            // packed rows: the rows present are in device memory; the row groups share what is there
        Mr = min(Mr, *p.rows_dev);
        upw = (((Mr + 31) >> 5) + p.nk_rg - 1) / p.nk_rg;
    }
    const int units = (Mr + 31) >> 5;
    const int u0 = rg * upw, u_end = min(units, (rg + 1) * upw);
    char* ck = smem + CK_OFF + wid * CKB;
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    const int pcols = p.Chi ? p.plane_cols : 0;
    const int ncols = pcols > p.N ? pcols : p.N;
    const int ncb = (min(NCW, ncols - n0) + 31) >> 5;
    const int seg = lane & 3, rsel = lane >> 2;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, p.C ? (int)((int64_t)p.M * p.ldc * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)p.Chi, 0, p.Chi ? (int)((int64_t)p.M * p.ldp * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)p.Clo, 0, p.Clo ? (int)((int64_t)p.M * p.ldp * 2) : 0, 0x00020000);
    const bool slow_epi = (f & (BMT_EPI_DROP_PRE | BMT_EPI_DROP_POST | BMT_EPI_GATE | BMT_EPI_RESIDUAL)) != 0;
    constexpr int CLIP = 0x7fffff00;
    // fragment address of k-step s: + fragb[s] (+ 8192 * block, + PLANE) in the weight chunk, + 8192 * wave in a ring slot
    int fragb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) fragb[s] = l31 * 256 + (((2 * s + half) ^ (l31 & 15)) * 16);
    const int wr_off = l31 * 128;
    const int ldc4 = (int)p.ldc * 4, ldp2 = (int)p.ldp * 2;

    bf16x8 wh[8], wl[TWO ? 8 : 1];
#define BMT_K128_FRAGS(cb_)                                                                                          \
    do {                                                                                                             \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                              \
            wh[s] = as_bf16x8(*reinterpret_cast<const u32x4*>(smem + fragb[s] + (cb_) * 8192));                      \
            if constexpr (TWO) wl[s] = as_bf16x8(*reinterpret_cast<const u32x4*>(smem + fragb[s] + (cb_) * 8192 + PLANE)); \
        }                                                                                                            \
    } while (0)

    auto process = [&](const bf16x8 (&a)[8], int u) {
        BMT_K128_FRAGS(0);
        const int row_a = 32 * u + rsel;                   // pass 0 row of this lane (pass 1: + 16); a unit past u_end stores nothing
        const int m_lim = u < u_end ? Mr : 0;
        const int seg8 = lane & 7, rsel8 = lane >> 3;      // this lane's 8 columns of the pair's 64, its row within a pass of 8
#pragma unroll
        for (int cb = 0; cb < NCB; cb += 2) {
            if (cb < ncb) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {           // the pair's two blocks, one accumulator after the other (no extra registers)
                    if (cb + hb < ncb) {
                        f32x16 acc0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 bq = *reinterpret_cast<const float4*>(sbias + 32 * (cb + hb) + 8 * j + 4 * half);
                            acc0[4 * j + 0] = bq.x; acc0[4 * j + 1] = bq.y; acc0[4 * j + 2] = bq.z; acc0[4 * j + 3] = bq.w;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        {
#pragma unroll
                            for (int s = 0; s < 8; ++s) {
                                if constexpr (TWO) acc0 = mfma32t<F16>(wl[s], a[s], acc0);
                                acc0 = mfma32t<F16>(wh[s], a[s], acc0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (cb + hb + 1 < NCB) {
                            if (cb + hb + 1 < ncb) BMT_K128_FRAGS(cb + hb + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // chunk row l31 (256 B = 16 slots of 16 B, slot ^ (row & 15)): columns 32 hb + 8 j + 4 half .. + 3
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float4 t;
                            t.x = acc0[4 * j + 0]; t.y = acc0[4 * j + 1]; t.z = acc0[4 * j + 2]; t.w = acc0[4 * j + 3];
                            *reinterpret_cast<float4*>(ck + l31 * 256 + (((8 * hb + 2 * j + half) ^ (l31 & 15)) * 16)) = t;
                        }
                    }
                }
                const int col = n0 + 32 * cb + 8 * seg8;
                const bool in_n = col < p.N;
                const bool in_p = col < pcols;
                const bool in_pair = 8 * seg8 < 32 * min(2, ncb - cb);       // (an odd last block: the pair's second half was not computed)
#pragma unroll 2
                for (int ps = 0; ps < 4; ++ps) {
                    const int rl = 8 * ps + rsel8;
                    const float4 t0 = *reinterpret_cast<const float4*>(ck + rl * 256 + (((2 * seg8) ^ (rl & 15)) * 16));
                    const float4 t1 = *reinterpret_cast<const float4*>(ck + rl * 256 + (((2 * seg8 + 1) ^ (rl & 15)) * 16));
                    float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    const int row = 32 * u + rl;
                    const bool rok = row < m_lim && in_pair;
                    if (f & BMT_EPI_RELU) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                    }
                    if (slow_epi) {
                        const int64_t idx = (int64_t)row * p.ldc + col;
                        if (f & BMT_EPI_DROP_PRE) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)(idx + q));
                        }
                        if (f & BMT_EPI_DROP_POST) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)(idx + q));
                        }
                        if ((f & BMT_EPI_GATE) && rok && in_n) {
                            const u32x4 gv = *reinterpret_cast<const u32x4*>(p.gate + (int64_t)row * p.ldg + col);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[2 * q] = (gv[q] & 0x00007FFFu) ? v[2 * q] * p.gate_scale : 0.f;
                                v[2 * q + 1] = (gv[q] & 0x7FFF0000u) ? v[2 * q + 1] * p.gate_scale : 0.f;
                            }
                        }
                        if ((f & BMT_EPI_RESIDUAL) && rok && in_n) {
                            const float* rp = p.residual + (int64_t)row * p.ldr + col;
                            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        }
                    }
                    if (32 * (cb + 2) > p.N - n0) {
                        if (!in_n) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = 0.f;
                        }
                    }
                    if (p.C) {
                        const int vo = (rok && in_n) ? row * ldc4 + col * 4 : CLIP;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rsC, vo, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rsC, vo, 16, 0);
                    }
                    if (p.Chi) {
                        const int vo = (rok && in_p) ? row * ldp2 + col * 2 : CLIP;
                        if (p.hi_f16) {
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])}, rsH, vo, 0, 0);
                        } else {
                            u32x4 h, l;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                uint32_t h_, l_;
                                split_bf2(v[2 * q], v[2 * q + 1], h_, l_);
                                h[q] = h_;
                                l[q] = l_;
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(h, rsH, vo, 0, 0);
                            if (p.Clo) {
                                if (p.second_f16) l = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
                                __builtin_amdgcn_raw_buffer_store_b128(l, rsL, vo, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    };
    // every wave computes; its activation rows come straight from global memory into registers, requested TWO units ahead by hand
    // (asm: the compiler does not track them) and waited for with a counted vmcnt that leaves the younger operations -- the other
    // register set's loads and the stores of the last two units -- in flight.  All stores are issued unconditionally (clipped by the
    // descriptor) so that the count is exact.
    u32x4 n0r[8], n1r[8];
#define BMT_K128_LD(dst_, i_) asm volatile("global_load_dwordx4 %0, %1, off offset:" #i_ "*32" : "=&v"(dst_[i_]) : "v"(ap_) : "memory")
#define BMT_K128_LDA(dst_, uu_)                                                                                     \
do {                                                                                                             \
    const int row_ = max(0, min(32 * min((uu_), units - 1) + l31, Mr - 1));                                             \
    const uint16_t* ap_ = p.Ah + (int64_t)row_ * p.lda + a_col + 8 * half;                                       \
    BMT_K128_LD(dst_, 0); BMT_K128_LD(dst_, 1); BMT_K128_LD(dst_, 2); BMT_K128_LD(dst_, 3);                      \
    BMT_K128_LD(dst_, 4); BMT_K128_LD(dst_, 5); BMT_K128_LD(dst_, 6); BMT_K128_LD(dst_, 7);                      \
} while (0)
#define BMT_K128_W(dst_, n_)                                                                                         \
asm volatile("s_waitcnt vmcnt(" #n_ ")"                                                                         \
             : "+v"(dst_[0]), "+v"(dst_[1]), "+v"(dst_[2]), "+v"(dst_[3]), "+v"(dst_[4]), "+v"(dst_[5]), "+v"(dst_[6]), "+v"(dst_[7])::"memory")
#define BMT_K128_WAIT(dst_, cnt_)                                                                                    \
do {                                                                                                             \
    const int c_ = (cnt_);                     /* operations younger than the loads waited for */               \
    if (c_ >= 56) BMT_K128_W(dst_, 56);                                                                          \
    else if (c_ >= 40) BMT_K128_W(dst_, 40);                                                                     \
    else if (c_ >= 24) BMT_K128_W(dst_, 24);                                                                     \
    else if (c_ >= 16) BMT_K128_W(dst_, 16);                                                                     \
    else if (c_ >= 8) BMT_K128_W(dst_, 8);                                                                       \
    else BMT_K128_W(dst_, 0);                                                                                    \
} while (0)
    const int nst = (p.C ? 2 : 0) + (p.Chi ? 1 : 0) + (p.Clo ? 1 : 0);      // store instructions per pass; 2 passes per block
    const int kw = 4 * nst * ((ncb + 1) >> 1);                                            // ... per unit (4 passes per pair of blocks)
    int u = u0 + wid;
    BMT_K128_LDA(n0r, u);
    BMT_K128_LDA(n1r, u + 8);
    BMT_K128_W(n0r, 8);                        // the weight chunk and the first unit's rows (older than the second unit's 8 loads)
    __syncthreads();
    bool first = true;
    for (; u < u_end; u += 16) {
        bf16x8 a[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = as_bf16x8(n0r[s]);
        BMT_K128_LDA(n0r, u + 16);
        process(a, u);
        if (u + 8 >= u_end) break;
        BMT_K128_WAIT(n1r, first ? 8 + kw : 8 + 2 * kw);
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = as_bf16x8(n1r[s]);
        BMT_K128_LDA(n1r, u + 24);
        process(a, u + 8);
        BMT_K128_WAIT(n0r, 8 + 2 * kw);
        first = false;
    }
#undef BMT_K128_LD
#undef BMT_K128_LDA
#undef BMT_K128_W
#undef BMT_K128_WAIT
#undef BMT_K128_FRAGS
}

// MANY independent GEMMs in one launch (the weight gradients of a whole step: each dW = dY^T . X is too small to fill the chip
// on its own -- 64 tiles for a 1024 x 1024 weight -- which is why the single launches split their reduction and pay an
// epilogue kernel plus the workspace traffic; together they are ~3000 tiles, enough to run every reduction unsplit).
// Every XCD has its own L2, so a problem spread over all 8 of them streams its operand panels from HBM 8 times; here the host
// packs the problems onto XCDs (whole problems, large ones in 2-8 contiguous parts): XCD x = workgroup index mod 8 works through
// its own list of segments {first slot, problem, first tile, tile count}, a tile's neighbours in the L2-friendly order run
// on the same XCD, and an operand panel is fetched by as few XCDs as the load balance allows.
// ---- small products (round 5): a decoder layer's own GEMMs -- 928 rows, 300 ... 1200 columns, reductions of 300 ... 1200 -- and the greedy
// decoder's.  On 128 x 128 tiles such a product is 24 ... 80 workgroups walking a chain of dependent stages (one CU moves ~25 GB/s with two
// stages in flight), or a split reduction plus a second kernel; both cost 20 ... 30 us for microseconds of MFMA work, and the decoder is a
// chain of ~40 of them with nothing else to run beside it (profiles/r05_d_replay_dispatches.csv).  Here every WAVE is on its own: a 32 x 32
// output tile and a range of the reduction, operands staged 64 reduction indices at a time through a wave-private LDS area (coalesced
// 16-byte loads, eight lanes per 128-byte line of a plane row, two chunks ahead in two register sets; ds_write_b128 into the swizzled
// [32 rows][8 slots] image the MFMA fragments are read from) -- no barrier in the loop.  Two groupings of a workgroup's four waves, chosen
// by the host (GemmB.nk_rg): 4 = one tile, the REDUCTION split over the waves, summed through LDS (long reductions over few tiles: the
// hand-over stays inside a workgroup, so no device-scope release / acquire between XCDs, which is what made the one-kernel split-K lose
// in round 4); 1 = a 64 x 64 block of four tiles, each wave the whole reduction of its own (short reductions).  (A first version fetched
// the MFMA fragments straight from the planes -- lane (row, half) its own 16 bytes -- and was slower than the kernels it replaces:
// 64 scattered 16-byte requests per instruction cost the texture path ~4x a coalesced one, profiles/r05_g_gemm_small_time.txt.)
// Full epilogue, same order of operations as gemm_bf16_tile's.
struct SmallBatch {                  // bmt_gemm_small_batched: product (o, i) = blockIdx.y / nb_inner, % nb_inner at these element offsets; nb_inner 0 = one product
    int nb_inner;
    int64_t a_off_o, a_off_i, b_off_o, b_off_i, c_off_o, c_off_i, p_off_o, p_off_i, p2_off_o, p2_off_i, ldp2, bias_off_i, drop_off_o, drop_off_i;
    const int* b_rows_dev;
    int a_div, c_div, p_div, p2_div;     // row r of A / C / the planes at (r / div) * qs + (r % div) * ld when div > 0
    int64_t a_qs, c_qs, p_qs, p2_qs;
};

// element offset of row r under the block-row addressing of SmallBatch
__device__ __forceinline__ int64_t small_row(int r, int div, int64_t qs, int64_t ld) {
    return div > 0 ? (int64_t)(r / div) * qs + (int64_t)(r % div) * ld : (int64_t)r * ld;
}

template <int NPASS, bool F16>
__global__ __launch_bounds__(256) void gemm_small_kernel(const GemmB p_, const SmallBatch bt) {
    GemmB p = p_;
    int64_t drop_base = 0, ldp2 = p.ldp;
    if (bt.nb_inner > 0) {
        const int by = blockIdx.y, bo = by / bt.nb_inner, bi = by - bo * bt.nb_inner;
        const int64_t ao = bo * bt.a_off_o + bi * bt.a_off_i;
        int64_t bof = bo * bt.b_off_o + bi * bt.b_off_i;
        if (bt.b_rows_dev != nullptr) {          // a packed memory: this sample's rows
            const int r0 = bt.b_rows_dev[bo];
            p.N = min(p.N, bt.b_rows_dev[bo + 1] - r0);
            bof += (int64_t)r0 * p.ldb;
        }
        p.Ah += ao; p.Bh += bof;
        if (p.Al) p.Al += ao;
        if (p.Bl) p.Bl += bof;
        if (p.C) p.C += bo * bt.c_off_o + bi * bt.c_off_i;
        if (p.Chi) p.Chi += bo * bt.p_off_o + bi * bt.p_off_i;
        if (p.Clo) p.Clo += bo * bt.p2_off_o + bi * bt.p2_off_i;
        if (bt.ldp2) ldp2 = bt.ldp2;
        if (p.bias) p.bias += bi * bt.bias_off_i;
        if (p.colsum) p.colsum += bi * bt.bias_off_i;
        drop_base = bo * bt.drop_off_o + bi * bt.drop_off_i;
    }
    constexpr bool ALO = NPASS == 3, BLO = NPASS >= 2;
    constexpr int NPL = 2 + (ALO ? 1 : 0) + (BLO ? 1 : 0);       // planes staged per chunk: A hi | B hi | [A lo] | [B lo]
    constexpr int WBYTES = NPL * 4096;                            // a wave's area: [plane][32 rows][8 slots of 16 B], swizzled (slot_of<8>)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
    char* const wbase = smem + wid * WBYTES;
    int Mr = p.M;
    if (p.rows_dev != nullptr) Mr = min(Mr, *p.rows_dev);
    const int ncols = p.Chi ? max(p.plane_cols, p.N) : p.N;      // columns that are written (planes: zeros past N)
    const bool ksplit = p.nk_rg == 4;
    const int bm = blockIdx.x % p.tiles_m, bn = blockIdx.x / p.tiles_m;
    const int m0 = ksplit ? bm * 32 : bm * 64 + 32 * (wid >> 1), n0 = ksplit ? bn * 32 : bn * 64 + 32 * (wid & 1);
    if ((ksplit ? m0 : bm * 64) >= Mr || (ksplit && n0 >= ncols)) return;      // (whole workgroup)
    if (!ksplit && (m0 >= Mr || n0 >= ncols)) return;             // a wave of the 64 x 64 block past the extents: it shares nothing
    const int nch = p.Kpad / 64, per = ksplit ? (nch + 3) / 4 : nch;
    const int c0 = ksplit ? wid * per : 0, c1 = min(nch, c0 + per);
    // (block rows: the rows are not one contiguous range; the descriptor ends just below the out-of-range offset the masked lanes use)
    const int64_t a_bytes = bt.a_div > 0 ? 0x7ffffff0 : (int64_t)Mr * p.lda * 2, b_bytes = (int64_t)p.N * p.ldb * 2;
    const __amdgpu_buffer_rsrc_t rsAh = plane_rsrc(p.Ah, a_bytes), rsAl = plane_rsrc(ALO ? p.Al : p.Ah, a_bytes);
    const __amdgpu_buffer_rsrc_t rsBh = plane_rsrc(p.Bh, b_bytes), rsBl = plane_rsrc(BLO ? p.Bl : p.Bh, b_bytes);
    constexpr int OOB = 0x7ffffff0;                               // past any plane (< 2 GiB, checked by the host): reads as zero
    int voa[4], vob[4], lds_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + (lane >> 3), piece = lane & 7;
        voa[i] = (m0 + row < Mr) ? (int)(small_row(m0 + row, bt.a_div, bt.a_qs, p.lda) * 2) + piece * 16 : OOB;
        vob[i] = (n0 + row < p.N) ? (int)((int64_t)(n0 + row) * p.ldb * 2) + piece * 16 : OOB;
        lds_w[i] = slot_of<8>(row, piece) * 16;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NS = 2;                                         // chunks in flight per wave (four of them for the one-pass products: slower, fewer waves per SIMD)
    u32x4 rg[NS][NPL][4];
#define BMT_SM_LOAD(set_, c_)                                                                       \
    do {                                                                                            \
        const bool in_ = (c_) < c1;                        /* wave-uniform */                       \
        const int so_ = (c_) * 128;                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
            const int va_ = in_ ? voa[i] : OOB, vb_ = in_ ? vob[i] : OOB;                           \
            rg[set_][0][i] = __builtin_amdgcn_raw_buffer_load_b128(rsAh, va_, so_, 0);              \
            rg[set_][1][i] = __builtin_amdgcn_raw_buffer_load_b128(rsBh, vb_, so_, 0);              \
            if constexpr (ALO) rg[set_][2][i] = __builtin_amdgcn_raw_buffer_load_b128(rsAl, va_, so_, 0); \
            if constexpr (BLO) rg[set_][NPL - 1][i] = __builtin_amdgcn_raw_buffer_load_b128(rsBl, vb_, so_, 0); \
        }                                                                                           \
    } while (0)
#define BMT_SM_STAGE(set_)                                                                          \
    do {   /* LDS operations of a wave execute in order: these writes follow the previous chunk's fragment reads */ \
        _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(wbase + pl * 4096 + lds_w[i]) = rg[set_][pl][i]; \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        /* every fragment of the chunk first (one LDS round trip), then the MFMA chain */           \
        bf16x8 fa_[4], fb_[4], fal_[4], fbl_[4];                                                    \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                             \
            const int fo_ = slot_of<8>(l31, 2 * u + half) * 16;                                     \
            fa_[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + fo_));                      \
            fb_[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + 4096 + fo_));               \
            if constexpr (ALO) fal_[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + 2 * 4096 + fo_)); \
            if constexpr (BLO) fbl_[u] = as_bf16x8(*reinterpret_cast<const u32x4*>(wbase + (NPL - 1) * 4096 + fo_)); \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                             \
            if constexpr (ALO) acc = mfma32t<F16>(fal_[u], fb_[u], acc);                            \
            if constexpr (BLO) acc = mfma32t<F16>(fa_[u], fbl_[u], acc);                            \
            acc = mfma32t<F16>(fa_[u], fb_[u], acc);                                                \
        }                                                                                           \
    } while (0)
    // branch-free inside the loop: a chunk past the wave's range is fetched out of range (zeros, no memory access) and multiplied anyway, so
    // that the compiler's vmcnt counts are exact (with a conditional load it waits for the YOUNGER set before touching the older one)
    if (c0 < c1) {
#pragma unroll
        for (int j = 0; j < NS; ++j) BMT_SM_LOAD(j, c0 + j);
        for (int c = c0; c < c1; c += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                BMT_SM_STAGE(j);
                BMT_SM_LOAD(j, c + NS + j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#undef BMT_SM_LOAD
#undef BMT_SM_STAGE
    // ---- the wave's 32 x 32 partial as fp32 [32][36] in its own area; reduction split: the four partials meet behind one barrier
    float* const mine = reinterpret_cast<float*>(wbase);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[acc_row(r, half) * 36 + l31] = acc[r];
    if (ksplit) {
        __syncthreads();
        if (wid >= 2) return;
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- epilogue (order: alpha, bias, dropout_pre, relu, dropout_post, gate, residual): a lane = one row x 8 consecutive columns; reduction
    // split: waves 0 / 1 take rows 0-15 / 16-31 of the summed tile; 64 x 64 block: every wave its own tile, 16 rows at a time
    const unsigned f = p.flags;
    const int nparts = ksplit ? 4 : 1, niter = ksplit ? 1 : 2, cg = (lane & 3) * 8, col = n0 + cg;
    const int pcols = p.Chi ? p.plane_cols : 0;
    const bool col_on = col < p.N || col < pcols, full = col + 8 <= p.N;
    const bool c_vec = p.C && full && ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    float bv[8], cs8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        bv[q] = ((f & BMT_EPI_BIAS) && col + q < p.N) ? p.bias[col + q] : 0.f;
        cs8[q] = 0.f;
    }
    for (int it = 0; it < niter; ++it) {
        const int rl = (ksplit ? wid * 16 : it * 16) + (lane >> 2), row = m0 + rl;
        if (row >= Mr || !col_on) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        for (int w_ = 0; w_ < nparts; ++w_) {
            const float* src = reinterpret_cast<const float*>(ksplit ? smem + w_ * WBYTES : wbase) + rl * 36 + cg;
            const float4 t0 = *reinterpret_cast<const float4*>(src), t1 = *reinterpret_cast<const float4*>(src + 4);
            v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
        }
        const int64_t idx = (int64_t)row * p.ldc + col, cidx = small_row(row, bt.c_div, bt.c_qs, p.ldc) + col;      // (idx: the dropout's element index)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = v[q] * p.alpha + bv[q];
        if (f & BMT_EPI_DROP_PRE) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)(drop_base + idx + q));
        }
        if (f & BMT_EPI_RELU) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (f & BMT_EPI_DROP_POST) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)(drop_base + idx + q));
        }
        if (f & BMT_EPI_GATE) {            // keep an element iff the saved forward output is non-zero (sign bit ignored)
            const uint16_t* gp = p.gate + (int64_t)row * p.ldg + col;
            if (full && ((p.ldg & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.gate) & 15) == 0)) {      // one 16-byte load instead of eight 2-byte ones
                const u32x4 gv = *reinterpret_cast<const u32x4*>(gp);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = (gv[q] & 0x00007FFFu) ? v[2 * q] * p.gate_scale : 0.f;
                    v[2 * q + 1] = (gv[q] & 0x7FFF0000u) ? v[2 * q + 1] * p.gate_scale : 0.f;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (col + q < p.N && (gp[q] & 0x7fffu)) ? v[q] * p.gate_scale : 0.f;
            }
        }
        if (f & BMT_EPI_RESIDUAL) {
            const float* rp = p.residual + (int64_t)row * p.ldr + col;
            if (full && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0)) {
                const float4 t0 = *reinterpret_cast<const float4*>(rp), t1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (col + q < p.N) v[q] += rp[q];
            }
        }
        if (!full) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (col + q >= p.N) v[q] = 0.f;                               // plane columns past N hold zeros
        }
        if (f & BMT_EPI_ACCUM) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (col + q < p.N) atomicAdd(p.C + cidx + q, v[q]);
        } else if (c_vec) {
            *reinterpret_cast<float4*>(p.C + cidx) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(p.C + cidx + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (p.C) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (col + q < p.N) p.C[cidx + q] = v[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) cs8[q] += v[q];
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
            const int64_t pi = small_row(row, bt.p_div, bt.p_qs, p.ldp) + col, pi2 = small_row(row, bt.p2_div, bt.p2_qs, ldp2) + col;
            if (p.plane_vec) {
                *reinterpret_cast<u32x4*>(p.Chi + pi) = h;
                if (p.Clo) *reinterpret_cast<u32x4*>(p.Clo + pi2) = l;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (col + q < pcols) {
                        p.Chi[pi + q] = (uint16_t)(h[q >> 1] >> (16 * (q & 1)));
                        if (p.Clo) p.Clo[pi2 + q] = (uint16_t)(l[q >> 1] >> (16 * (q & 1)));
                    }
            }
        }
    }
    if (p.colsum) {                  // uniform per launch.  A wave's rows live in lanes 4 apart: fold them, one atomic per column and wave
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float t = cs8[q];
            t += __shfl_xor(t, 4, 64);
            t += __shfl_xor(t, 8, 64);
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (lane < 4 && col + q < p.N) atomicAdd(p.colsum + col + q, t);
        }
    }
}

struct XcdSeg {
    int first_slot, prob, tile_off, count;
    int nsplit;       // reduction chunks of the problem (p.kchunk rows each); the segment's work items are chunk-major: item L = chunk
                      // L / count of tile tile_off + L % count, so the workgroups an XCD runs together stream the SAME rows of
                      // neighbouring tiles' operand panels (they would drift apart over a 25600-row reduction and out of the 4 MB L2)
};
constexpr int XCD_MAXSEG = 64;

template <int NPASS, int WM, int TI, bool AKM, bool BKM, bool PIPE = false>
__global__ __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(TI == 2 ? 2 : 4, TI == 2 ? 2 : 4))) void gemm_bf16_grouped_kernel(
    const GemmB* __restrict__ table, const XcdSeg* __restrict__ segs, const int* __restrict__ nseg) {
    const int x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const XcdSeg* sx = segs + x * XCD_MAXSEG;
    int lo = 0, hi = nseg[x] - 1;                     // last segment of this XCD whose first slot is <= slot (uniform: scalar loads)
    if (hi < 0) return;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sx[mid].first_slot <= slot) lo = mid;
        else hi = mid - 1;
    }
    const XcdSeg sg = sx[lo];
    const int local = slot - sg.first_slot;
    if (local >= sg.count * sg.nsplit) return;        // past the end of this XCD's list
    const GemmB p = table[sg.prob];
    const int tile = sg.tile_off + local % sg.count, split = local / sg.count;
    if constexpr (PIPE) gemm_bf16_tile<NPASS, WM, TI, AKM, BKM, 0, false, true, 2>(p, tile, split, true);
    else gemm_bf16_tile<NPASS, WM, TI, AKM, BKM, 0>(p, tile, split, true);
}

// the same launch under another name for the per-sample products of an encoder memory's gradient (output rows placed by device-side offsets:
// bmt_gemm_bf16_args.c_row_dev): kernel statistics and counters keep the step's weight-gradient launch apart from them
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_memory_grad_kernel(const GemmB* __restrict__ table,
                                                                                                            const XcdSeg* __restrict__ segs,
                                                                                                            const int* __restrict__ nseg) {
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
    gemm_bf16_tile<1, 4, 1, true, true, 0>(p, sg.tile_off + local % sg.count, local / sg.count, true);
}

// second pass of the two-pass split-K: sum the partials of one output element group (4 consecutive columns) in split order
// and run the same epilogue as the GEMM kernel.  One writer per element: ACCUM is a plain read-modify-write.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmB p) {
    const int pc = p.Chi ? max(p.plane_cols, p.N) : p.N;
    const int ncg = (pc + 3) / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * ncg) return;
    const int row = (int)(idx / ncg), c0 = (int)(idx % ncg) * 4;
    if (p.rows_dev != nullptr && !p.rows_is_k && row >= *p.rows_dev) return;      // packed rows: the tiles past the rows present wrote no partials
    const int64_t ldw = (int64_t)p.tiles_n * BN;
    const int64_t slab = (int64_t)p.tiles_m * p.bm * ldw;
    const float* src = p.ws + (int64_t)row * ldw + c0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sidx = 0; sidx < p.nsplit; ++sidx) {
        const float4 v = *reinterpret_cast<const float4*>(src + sidx * slab);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const float acc4[4] = {a.x, a.y, a.z, a.w};
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    const unsigned f = p.flags;
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = c0 + c;
        float v = 0.f;
        if (col < p.N) {
            v = acc4[c] * p.alpha + ((f & BMT_EPI_BIAS) ? p.bias[col] : 0.f);
            const int64_t ci = (int64_t)row * p.ldc + col;
            if (f & BMT_EPI_DROP_PRE) v = drop_apply(dc, v, (uint64_t)ci);
            if (f & BMT_EPI_RELU) v = fmaxf(v, 0.f);
            if (f & BMT_EPI_DROP_POST) v = drop_apply(dc, v, (uint64_t)ci);
            if (f & BMT_EPI_GATE) v = (p.gate[(int64_t)row * p.ldg + col] & 0x7fffu) ? v * p.gate_scale : 0.f;
            if (f & BMT_EPI_RESIDUAL) v += p.residual[(int64_t)row * p.ldr + col];
            if (f & BMT_EPI_ACCUM) p.C[ci] += v;
            else if (p.C) p.C[ci] = v;
        }
        out[c] = v;
    }
    if (p.Chi) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = c0 + c;
            if (col < p.plane_cols) {
                const __bf16 hv = (__bf16)out[c];
                const int64_t pi = (int64_t)row * p.ldp + col;
                p.Chi[pi] = p.hi_f16 ? __builtin_bit_cast(uint16_t, (_Float16)out[c]) : __builtin_bit_cast(uint16_t, hv);
                if (p.Clo) p.Clo[pi] = p.second_f16 ? __builtin_bit_cast(uint16_t, (_Float16)out[c])
                                                    : __builtin_bit_cast(uint16_t, (__bf16)(out[c] - (float)hv));
            }
        }
    }
}

// ---------------------------------------------------------------- plane construction
// 64x64 fp32 tile -> bf16 planes, straight (hi/lo [R][ldp]) and/or transposed (hiT/loT [C][ldpT]), zero padded;
// optional column sums of the source (bias gradients: the gradient tensor is being read here anyway).
struct PlaneDesc {
    const float* src; int64_t ld; int R, C;
    uint16_t *hi, *lo; int64_t ldp; int pcols;
    uint16_t *fh, *fl;                                              // optional fp16 planes, same [R][ldp] layout: fh = fp16(x), fl = fp16(x - fh)
    uint16_t *hiT, *loT; int64_t ldpT; int pcolsT;
    float* colsum;
    int tiles_x, tiles_y;
    float drop_p; uint32_t drop_site; const uint64_t* drop_rng;     // optional: the source is masked (inverted dropout) first
    int rows_ok;                                                    // bmt_planes_desc: eligible for the 16-byte row kernel
    const float* gate; int64_t ldgate; float gate_scale;            // optional: v = gate[r][c] != 0 ? v * gate_scale : 0 (relu / dropout derivative
                                                                    // from the saved forward output: bmt_planes_gate)
    const int* rows_dev;                                            // optional (packed rows): only rows < *rows_dev exist; the launch is sized for R
};

__device__ __forceinline__ void planes_tile(const PlaneDesc& d_, int bx, int by, float (*tile)[65]) {
    PlaneDesc d = d_;
    if (d.rows_dev != nullptr) d.R = min(d.R, *d.rows_dev);
    const int r0 = by * 64, c0 = bx * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 row groups
    float csum = 0.f;
    const DropCtx dc = make_drop(d.drop_p, d.drop_rng, d.drop_site);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        float v = (r < d.R && c < d.C) ? d.src[(int64_t)r * d.ld + c] : 0.f;
        if (d.gate && r < d.R && c < d.C) v = d.gate[(int64_t)r * d.ldgate + c] != 0.f ? v * d.gate_scale : 0.f;
        if (dc.on) v = drop_apply(dc, v, (uint64_t)((int64_t)r * d.C + c));      // element index of the [R][C] tensor
        csum += v;
        tile[ty * 16 + i][tx] = v;
        if (d.hi && r < d.R && c < d.pcols) {
            const __bf16 h = (__bf16)v;
            d.hi[(int64_t)r * d.ldp + c] = __builtin_bit_cast(uint16_t, h);
            if (d.lo) d.lo[(int64_t)r * d.ldp + c] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
        }
        if (d.fh && r < d.R && c < d.pcols) {
            const _Float16 h = (_Float16)v;
            d.fh[(int64_t)r * d.ldp + c] = __builtin_bit_cast(uint16_t, h);
            if (d.fl) d.fl[(int64_t)r * d.ldp + c] = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)h));
        }
    }
    if (d.hiT) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty * 16 + i, r = r0 + tx;     // output row = source column c, output column = source row r
            if (c < d.C && r < d.pcolsT) {
                const float v = tile[tx][ty * 16 + i];
                const __bf16 h = (__bf16)v;
                d.hiT[(int64_t)c * d.ldpT + r] = __builtin_bit_cast(uint16_t, h);
                if (d.loT) d.loT[(int64_t)c * d.ldpT + r] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
            }
        }
    }
    if (d.colsum) {
        __syncthreads();
        tile[ty][tx] = csum;          // rows 0..3 of the tile reused as the 4 partial sums per column
        __syncthreads();
        if (ty == 0 && c0 + tx < d.C) atomicAdd(d.colsum + c0 + tx, (tile[0][tx] + tile[1][tx]) + (tile[2][tx] + tile[3][tx]));
    }
}

__global__ __launch_bounds__(256) void planes_kernel(const PlaneDesc d) {
    __shared__ float tile[64][65];
    planes_tile(d, blockIdx.x, blockIdx.y, tile);
}

// Straight planes only (no transposed output -- the common case since the backward GEMMs read k-major operands): each thread
// converts 8 consecutive columns of a row, two 16-byte loads in, one 16-byte store per plane out (planes_tile moves 4 / 2 bytes
// per access because its thread mapping serves the LDS transpose).  Block = 64 rows x 128 columns; column sums: 8 partials per
// thread, folded over the 16 row-threads of a column group through LDS, one atomic per column per block.
__device__ __forceinline__ void planes_rows_tile(const PlaneDesc& d_, int bx, int by, float (*cs)[129]) {
    PlaneDesc d = d_;
    if (d.rows_dev != nullptr) d.R = min(d.R, *d.rows_dev);
    const int tid = threadIdx.x;
    const int cg = (tid & 15) * 8, rt = tid >> 4;
    const int c0 = bx * 128 + cg, r0 = by * 64;
    const DropCtx dc = make_drop(d.drop_p, d.drop_rng, d.drop_site);
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < d.pcols) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = r0 + ps * 16 + rt;
            if (r >= d.R) continue;
            float v[8];
            if (c0 + 8 <= d.C) {
                const float4 a = *reinterpret_cast<const float4*>(d.src + (int64_t)r * d.ld + c0);
                const float4 b = *reinterpret_cast<const float4*>(d.src + (int64_t)r * d.ld + c0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (c0 + q < d.C) ? d.src[(int64_t)r * d.ld + c0 + q] : 0.f;     // ragged edge / zero padding
            }
            if (d.gate) {
                if (c0 + 8 <= d.C && (d.ldgate & 3) == 0) {
                    const float4 a = *reinterpret_cast<const float4*>(d.gate + (int64_t)r * d.ldgate + c0);
                    const float4 b = *reinterpret_cast<const float4*>(d.gate + (int64_t)r * d.ldgate + c0 + 4);
                    const float gq[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = gq[q] != 0.f ? v[q] * d.gate_scale : 0.f;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (c0 + q < d.C && d.gate[(int64_t)r * d.ldgate + c0 + q] != 0.f) ? v[q] * d.gate_scale : 0.f;
                }
            }
            if (dc.on) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = drop_apply(dc, v[q], (uint64_t)((int64_t)r * d.C + c0 + q));
            }
            u32x4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t hh, ll;
                split_bf2(v[2 * q], v[2 * q + 1], hh, ll);
                h[q] = hh; l[q] = ll;
                part[2 * q] += v[2 * q]; part[2 * q + 1] += v[2 * q + 1];
            }
            if (d.hi) *reinterpret_cast<u32x4*>(d.hi + (int64_t)r * d.ldp + c0) = h;
            if (d.lo) *reinterpret_cast<u32x4*>(d.lo + (int64_t)r * d.ldp + c0) = l;
            if (d.fh) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t hh, ll;
                    split_h2(v[2 * q], v[2 * q + 1], hh, ll);
                    h[q] = hh; l[q] = ll;
                }
                *reinterpret_cast<u32x4*>(d.fh + (int64_t)r * d.ldp + c0) = h;
                if (d.fl) *reinterpret_cast<u32x4*>(d.fl + (int64_t)r * d.ldp + c0) = l;
            }
        }
    }
    if (d.colsum) {
#pragma unroll
        for (int q = 0; q < 8; ++q) cs[rt][cg + q] = part[q];
        __syncthreads();
        if (tid < 128 && bx * 128 + tid < d.C) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += cs[i][tid];
            atomicAdd(d.colsum + bx * 128 + tid, t);
        }
    }
}

__global__ __launch_bounds__(256) void planes_rows_kernel(const PlaneDesc d) {
    __shared__ float cs[16][129];
    planes_rows_tile(d, blockIdx.x, blockIdx.y, cs);
}

// the vector kernel applies when there is a straight hi plane and nothing transposed, everything 16-byte aligned
static bool planes_straight_ok(const PlaneDesc& d) {
    return (d.hi || d.fh) && (d.ld % 4 == 0) && (d.ldp % 8 == 0) && (d.pcols % 8 == 0) &&
           ((reinterpret_cast<uintptr_t>(d.src) | reinterpret_cast<uintptr_t>(d.hi) | reinterpret_cast<uintptr_t>(d.lo) |
             reinterpret_cast<uintptr_t>(d.fh) | reinterpret_cast<uintptr_t>(d.fl)) & 15) == 0;
}
static bool planes_rows_ok(const PlaneDesc& d) { return planes_straight_ok(d) && !d.hiT; }

// every weight of the model in ONE launch: blockIdx.y = tensor, blockIdx.x strides over its 64x64 tiles
__global__ __launch_bounds__(256) void planes_multi_kernel(const PlaneDesc* __restrict__ table) {
    __shared__ float tile[64][65];
    const PlaneDesc d = table[blockIdx.y];
    if (d.rows_ok) {                 // straight planes aligned: the 16-byte path (64 x 128 tiles)
        const int ntx = (d.pcols + 127) / 128, nt = ntx * ((d.R + 63) / 64);
        for (int t = blockIdx.x; t < nt; t += gridDim.x) planes_rows_tile(d, t % ntx, t / ntx, nullptr);
        if (d.rows_ok == 2) {        // + a transposed bf16 plane (the row-major B operand of the dX products): 64 x 64 tiles through LDS
            PlaneDesc dt = d;
            dt.hi = dt.lo = dt.fh = dt.fl = nullptr;
            dt.pcols = 0;
            const int tx_ = (d.C + 63) / 64, nt2 = tx_ * d.tiles_y;
            for (int t = blockIdx.x; t < nt2; t += gridDim.x) {
                planes_tile(dt, t % tx_, t / tx_, tile);
                __syncthreads();
            }
        }
        return;
    }
    const int nt = d.tiles_x * d.tiles_y;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        planes_tile(d, t % d.tiles_x, t / d.tiles_x, tile);
        __syncthreads();
    }
}

// the same work as a FLAT tile list: workgroup = one tile of one tensor, found by bisection of the prefix sums of the tensors' tile counts
// (planes_desc_tiles; the host builds them with the table).  planes_multi_kernel gives every tensor 64 workgroups that stride over its
// tiles: the four 4096 x 1024 FFN weights (512 tiles each) were converted by 64 workgroups apiece while the workgroups of ~60 small
// tensors idled -- 0.50 ms per optimizer step for ~100 us of HBM traffic (profiles/r02_r_kernel_stats.csv, r03_p_kernel_stats.csv).
__host__ __device__ inline int planes_desc_tiles(const PlaneDesc& d) {
    if (d.rows_ok) {
        int nt = ((d.pcols + 127) / 128) * ((d.R + 63) / 64);
        if (d.rows_ok == 2) nt += ((d.C + 63) / 64) * d.tiles_y;
        return nt;
    }
    return d.tiles_x * d.tiles_y;
}
__global__ __launch_bounds__(256) void planes_multi_flat_kernel(const PlaneDesc* __restrict__ table, const int* __restrict__ prefix, int n) {
    __shared__ float tile[64][65];
    const int t = blockIdx.x;
    int lo = 0, hi = n;                       // prefix[i] <= t < prefix[i + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= t) lo = mid; else hi = mid;
    }
    const PlaneDesc d = table[lo];
    int lt = t - prefix[lo];
    if (d.rows_ok) {
        const int ntx = (d.pcols + 127) / 128, nt = ntx * ((d.R + 63) / 64);
        if (lt < nt) {
            planes_rows_tile(d, lt % ntx, lt / ntx, nullptr);
        } else {
            lt -= nt;
            PlaneDesc dt = d;
            dt.hi = dt.lo = dt.fh = dt.fl = nullptr;
            dt.pcols = 0;
            const int tx_ = (d.C + 63) / 64;
            planes_tile(dt, lt % tx_, lt / tx_, tile);
        }
        return;
    }
    planes_tile(d, lt % d.tiles_x, lt / d.tiles_x, tile);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NPASS, int WM, int TI, bool AKM = false, bool BKM = false, int CONV = 0, bool F16 = false>
int launch(const GemmB& p, int splitk, hipStream_t st) {
    constexpr int BK = (NPASS == 1) ? 64 : 32;
    constexpr int BMr = 32 * TI * WM;
    constexpr int stage = (NPASS == 3 ? 2 : 1) * (AKM ? BK * km_rs<BMr>() : BMr * BK * 2) + (NPASS >= 2 ? 2 : 1) * (BKM ? BK * km_rs<BN>() : BN * BK * 2);
    constexpr int lds = (2 * stage > BMr * BN * 4) ? 2 * stage : BMr * BN * 4;   // two stage buffers / the packed plane tile of the epilogue
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<NPASS, WM, TI, AKM, BKM, CONV, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<NPASS, WM, TI, AKM, BKM, CONV, F16>), dim3(p.tiles_m * p.tiles_n, splitk), dim3(128 * WM), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16");
    return BMT_OK;
}

template <int NPASS, bool F16, int TI, bool AKM = false, bool BKM = false, int CONV = 0>
int launch_pipe(const GemmB& p, int splitk, hipStream_t st) {
    constexpr int BK = gemm_bk(NPASS, true, TI);
    constexpr int BMr = 128 * TI;
    constexpr int stage = BMr * BK * 2 + (NPASS >= 2 ? 2 : 1) * BN * BK * 2;
    constexpr int R = pipe_ring(stage, TI);
    constexpr int lds = (R * stage > BMr * BN * 4) ? R * stage : BMr * BN * 4;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)gemm_pipe_kernel<NPASS, F16, TI, AKM, BKM, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    const int tiles = p.tiles_m * p.tiles_n, slots = bmt_device_cus() * (TI == 2 ? 1 : 2);
    const int gx = (tiles > slots && slots % 8 == 0) ? slots : tiles;          // persistent: a workgroup per slot walks the tiles
    hipLaunchKernelGGL((gemm_pipe_kernel<NPASS, F16, TI, AKM, BKM, CONV>), dim3(gx, splitk), dim3(512), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16(pipelined)");
    return BMT_OK;
}

template <bool F16>
int launch_wide(const GemmB& p, hipStream_t st) {
    constexpr int lds = 2 * 4 * 16384;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    hipLaunchKernelGGL((gemm_wide_kernel<F16>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16(256 x 256)");
    return BMT_OK;
}

// the weight-chunk-resident kernel: 128-column chunks (64 KB of LDS with two weight planes), eight computing waves, two 32-column blocks
// through a wave's LDS chunk together
template <bool F16, bool TWO>
int launch_k128_(const GemmB& p, hipStream_t st) {
    constexpr int NCB = 4;
    constexpr int lds = (TWO ? 2 : 1) * NCB * 32 * 256 + 8 * 8192 + NCB * 32 * 4;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)gemm_k128_kernel<F16, TWO, NCB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    hipLaunchKernelGGL((gemm_k128_kernel<F16, TWO, NCB>), dim3(p.tiles_n * p.nk_rg), dim3(512), lds, st, p);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16(reduction of 128)");
    return BMT_OK;
}
template <bool F16>
int launch_k128(const GemmB& p, hipStream_t st) {
    return p.Bl ? launch_k128_<F16, true>(p, st) : launch_k128_<F16, false>(p, st);
}

template <int NPASS, bool F16>
int launch_small(const GemmB& p, hipStream_t st, const SmallBatch* bt = nullptr, int nbatch = 1) {
    constexpr int lds = 4 * (2 + (NPASS == 3 ? 1 : 0) + (NPASS >= 2 ? 1 : 0)) * 4096;
    SmallBatch none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((gemm_small_kernel<NPASS, F16>), dim3(p.tiles_m * p.tiles_n, nbatch), dim3(256), lds, st, p, bt ? *bt : none);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16(32 x 32 tiles)");
    return BMT_OK;
}

}  // namespace


// products of at most this many outputs run on gemm_small_kernel (a build with -DBMT_SMALL_TILE_OUTPUTS=0 is the A/B arm without it:
// tools/gpu_r5.sh ablib)
#ifndef BMT_SMALL_TILE_OUTPUTS
#define BMT_SMALL_TILE_OUTPUTS (3ll << 19)
#endif
extern "C" long long bmt_gemm_small_outputs(void) { return BMT_SMALL_TILE_OUTPUTS; }

// validate the arguments and fill the kernel descriptor; splitk: in = 0 (decide here) / forced value, out = splits to launch
static int gemm_prepare(const bmt_gemm_bf16_args* a, GemmB& p, int& splitk, bool allow_split) {
    BMT_CHECK_ARG(a && a->A_hi && a->B_hi && (a->C || a->C_hi || a->C_f16), "bmt_gemm_bf16: null pointer");
    BMT_CHECK_ARG(a->M > 0 && a->N > 0 && a->Kpad > 0 && a->Kpad % 64 == 0, "bmt_gemm_bf16: bad sizes M=%d N=%d Kpad=%d (Kpad %% 64 != 0?)",
                  a->M, a->N, a->Kpad);
    BMT_CHECK_ARG(a->precision == BMT_PREC_BF16 || a->precision == BMT_PREC_F16 || (a->precision == BMT_PREC_BF16X3 && a->A_lo && a->B_lo) ||
                      (a->precision == BMT_PREC_F16W2 && a->B_lo),
                  "bmt_gemm_bf16: BF16X3 needs both lo planes, F16W2 the lo plane of B");
    BMT_CHECK_ARG(!(a->C_lo && a->C_f16), "bmt_gemm_bf16: C_lo and C_f16 are alternatives (one second output plane)");
    BMT_CHECK_ARG(!(a->C_f16 && a->colsum), "bmt_gemm_bf16: colsum is taken from bf16 hi + lo of the staged value, not with C_f16");
    BMT_CHECK_ARG(!(a->a_kmajor || a->b_kmajor) || (a->precision == BMT_PREC_BF16 && a->K > 0 && a->K <= a->Kpad && a->Kpad - a->K < 64),
                  "bmt_gemm_bf16: k-major operands need BMT_PREC_BF16 and the true reduction length K (Kpad = K rounded up to 64)");
    BMT_CHECK_ARG(a->conv_mode == 0 || (a->conv_cin > 0 && a->conv_cin % 64 == 0 && a->conv_rows > 0 &&
                                        (a->conv_mode == 1 ? (!a->a_kmajor && !a->b_kmajor && a->Kpad % a->conv_cin == 0 && a->lda >= a->conv_cin)
                                                           : (a->conv_mode == 2 && a->a_kmajor && a->b_kmajor && a->conv_cin % 128 == 0 && a->N % a->conv_cin == 0))),
                  "bmt_gemm_bf16: bad implicit-convolution arguments");
    BMT_CHECK_ARG(((a->a_kmajor || a->conv_mode == 1) ? a->lda >= 8 : a->lda >= a->Kpad) && (a->b_kmajor ? a->ldb >= 8 : a->ldb >= a->Kpad),
                  "bmt_gemm_bf16: plane row stride smaller than Kpad");
    if (!(al16(a->A_hi) && al16(a->B_hi)) || ((a->lda | a->ldb) & 7) || (a->A_lo && !al16(a->A_lo)) || (a->B_lo && !al16(a->B_lo))) {
        bmt_set_error("bmt_gemm_bf16: planes must be 16-byte aligned with row strides multiples of 8 elements");
        return BMT_EALIGN;
    }
    {   // operand tiles are fetched with 32-bit byte offsets (buffer loads): a plane must stay below 2 GiB
        const int64_t a_rows = a->conv_mode == 1 ? a->conv_rows : (a->a_kmajor ? a->K : a->M);
        const int64_t b_rows = a->conv_mode == 2 ? a->conv_rows : (a->b_kmajor ? a->K : a->N);
        BMT_CHECK_ARG(a_rows * a->lda * 2 < (1ll << 31) && b_rows * a->ldb * 2 < (1ll << 31) &&
                          (!a->a_kmajor || (int64_t)a->Kpad * a->lda * 2 < (1ll << 31)) && (!a->b_kmajor || (int64_t)a->Kpad * a->ldb * 2 < (1ll << 31)),
                      "bmt_gemm_bf16: an operand plane of 2 GiB or more (rows %lld x ld %lld / rows %lld x ld %lld)", (long long)a_rows,
                      (long long)a->lda, (long long)b_rows, (long long)a->ldb);
    }
    splitk = a->splitk < 1 ? 1 : a->splitk;
    if (!allow_split) splitk = 1;
    const bool accum = (a->flags & BMT_EPI_ACCUM) != 0;
    const bool two_pass = allow_split && a->splitk_ws != nullptr;             // split-K through a workspace: any epilogue
    const unsigned nonlin = BMT_EPI_RELU | BMT_EPI_DROP_PRE | BMT_EPI_DROP_POST | BMT_EPI_GATE | BMT_EPI_BIAS | BMT_EPI_RESIDUAL;
    BMT_CHECK_ARG(splitk == 1 || two_pass || (accum && !(a->flags & nonlin) && !a->C_hi && !a->C_f16),
                  "bmt_gemm_bf16: splitk>1 needs either a split-K workspace or BMT_EPI_ACCUM with no other epilogue op / plane output");
    BMT_CHECK_ARG(!(a->flags & BMT_EPI_ACCUM) || a->C, "bmt_gemm_bf16: ACCUM needs the fp32 output");
    BMT_CHECK_ARG(!(a->flags & BMT_EPI_BIAS) || a->bias, "bmt_gemm_bf16: BIAS flag without pointer");
    BMT_CHECK_ARG(!(a->flags & BMT_EPI_RESIDUAL) || a->residual, "bmt_gemm_bf16: RESIDUAL flag without pointer");
    BMT_CHECK_ARG(!(a->flags & BMT_EPI_GATE) || a->gate, "bmt_gemm_bf16: GATE flag without pointer");
    memset(&p, 0, sizeof(p));
    p.Ah = a->A_hi; p.Al = a->A_lo; p.Bh = a->B_hi; p.Bl = a->B_lo; p.lda = a->lda; p.ldb = a->ldb;
    p.C = a->C; p.ldc = a->ldc; p.Chi = a->C_hi; p.Clo = a->C_f16 ? a->C_f16 : a->C_lo; p.second_f16 = a->C_f16 != nullptr; p.ldp = a->ldp;
    p.hi_f16 = 0;
    if (!a->C_hi && a->C_f16) {          // the fp16 plane alone (q / k / v under the fp16 attention policy: the backward converts on load)
        BMT_CHECK_ARG(!a->C_lo && !a->colsum, "bmt_gemm_bf16: C_f16 without C_hi excludes C_lo and colsum");
        p.Chi = a->C_f16; p.Clo = nullptr; p.second_f16 = 0; p.hi_f16 = 1;
    }
    p.plane_cols = p.Chi ? (int)((a->N + 63) / 64 * 64 < a->ldp ? (a->N + 63) / 64 * 64 : a->ldp) : 0;
    p.plane_vec = p.Chi && al16(p.Chi) && (!p.Clo || al16(p.Clo)) && (a->ldp % 8 == 0) && (p.plane_cols % 8 == 0);
    p.M = a->M; p.N = a->N; p.Kpad = a->Kpad; p.krows = a->K;
    p.rows_dev = a->rows_dev; p.rows_is_k = a->a_kmajor != 0;
    p.c_row_dev = a->c_row_dev; p.m_dev = a->m_dev;
    p.conv_cin = a->conv_cin; p.conv_rows = a->conv_rows; p.conv_S = a->conv_S > 0 ? a->conv_S : 1; p.conv_halo = a->conv_halo;
    // Conv1d dW: the taps of a channel block before the next channel block (halves the launch's HBM fetch, profiles/r04_x_ab_conv_dw_order.txt)
    p.conv_tap_minor = (a->conv_mode == 2 && a->conv_cin % BN == 0) ? 1 : 0;
    p.tiles_n = bmt_cdiv(p.Chi ? (p.plane_cols > a->N ? p.plane_cols : a->N) : a->N, BN);
    // Which kernel (every choice below was measured against its alternatives on the shapes of the step; DESIGN.md section 6 has the numbers):
    //   pipe 0  register-staged loop on 128-row tiles, 8 waves: k-major operands (every dX / dW), the three-pass product, Conv1d dW
    //   pipe 2  LDS-DMA pipelined loop on 128-row tiles, two workgroups per CU: every other row-major product
    //   pipe 1  ... on 256 x 128 tiles, one workgroup per CU: launches that are whole rounds of 256 such tiles with a reduction >= 512, and
    //           the Conv1d forward where that is at least one tile per CU
    //   pipe 3  the 256 x 256 ping-pong kernel: >= one round of its tiles, reduction >= 256, plain epilogue
    //   pipe 4  the weight-chunk-resident kernel: a reduction of exactly 128 over >= 2048 rows (the audio stream's projections)
    //   pipe 5  32 x 32 tiles, one per workgroup, the reduction over its waves: three-pass and one-pass bf16 products of <= 1.5 M outputs
    //           (a decoder layer's own GEMMs forward, and their dX with the weight plane transposed)
    p.bm = 128;
    p.pipe = 0;
    if (!a->a_kmajor && !a->b_kmajor && !a->conv_mode && a->precision != BMT_PREC_BF16X3) p.pipe = 2;
    if (p.pipe == 2 && a->Kpad >= 512) {
        const int t256 = bmt_cdiv(a->M, 256) * p.tiles_n, cus = bmt_device_cus();
        if (t256 >= cus && t256 % cus == 0) p.pipe = 1;
    }
    if (a->conv_mode == 1 && !a->a_kmajor && !a->b_kmajor && (a->precision == BMT_PREC_F16 || a->precision == BMT_PREC_F16W2) &&
        bmt_cdiv(a->M, 256) * p.tiles_n >= bmt_device_cus())
        p.pipe = 1;
    const bool wide_ok = p.pipe != 0 && !a->conv_mode && !a->a_kmajor && !a->b_kmajor && !a->colsum && !(a->flags & BMT_EPI_ACCUM) && a->splitk <= 1 && a->N >= 256 &&
                         (int64_t)(a->M + 256) * a->lda * 2 < (1ll << 31) && (int64_t)(a->N + 256) * a->ldb * 2 < (1ll << 31);
    const int wide_tiles = bmt_cdiv(a->M, 256) * bmt_cdiv(p.Chi ? (p.plane_cols > a->N ? p.plane_cols : a->N) : a->N, 256);
    if (wide_ok && wide_tiles >= bmt_device_cus() && a->Kpad >= 256) p.pipe = 3;
    if (p.pipe == 3) {
        p.bm = 256;
        p.tiles_n = bmt_cdiv(p.Chi ? (p.plane_cols > a->N ? p.plane_cols : a->N) : a->N, 256);
        if (a->precision != BMT_PREC_F16W2) p.Bl = nullptr;      // the kernel runs its second pass iff there is a second weight plane
    }
    if (p.pipe == 1) p.bm = 256;
    const bool blk = a->a_blk_n > 0, blk_km = blk && a->b_kmajor && !a->a_kmajor;
    BMT_CHECK_ARG(!blk || blk_km ||
                      (a->a_blk_k == 128 && a->Kpad == 128 && a->a_blk_n % 128 == 0 && a->N % a->a_blk_n == 0 && p.pipe == 2 &&
                       a->lda >= (int64_t)(a->N / a->a_blk_n) * a->a_blk_k),
                  "bmt_gemm_bf16: block products (a_blk_n) take row-major one- / two-plane operands with a_blk_k = Kpad = 128 and a_blk_n a multiple of 128 that divides N");
    BMT_CHECK_ARG(!blk_km || (!a->conv_mode && a->a_blk_n % BN == 0 && a->N % a->a_blk_n == 0 && a->a_blk_k % 64 == 0 &&
                              (int64_t)(a->N / a->a_blk_n) * a->a_blk_k == a->K && a->K == a->Kpad && a->splitk <= 1),
                  "bmt_gemm_bf16: block products with a k-major B: a_blk_n a multiple of 128 that divides N, a_blk_k a multiple of 64, K = (N / a_blk_n) a_blk_k, unsplit");
    if (blk_km) { p.a_blk_n = a->a_blk_n; p.a_blk_k = a->a_blk_k; }
    if (p.pipe == 2 && a->Kpad == 128 && (a->M >= 2048 || blk) && a->N >= 128 && !a->colsum &&
        !(a->flags & BMT_EPI_ACCUM) && a->splitk <= 1 && (int64_t)(a->N + 256) * a->ldb * 2 < (1ll << 31) && a->alpha == 1.f &&
        // its stores are 16-byte buffer stores clipped by 32-bit descriptors, its gate / residual loads 16-byte loads
        a->N % 8 == 0 && (!p.Chi || (p.plane_vec && (int64_t)a->M * a->ldp * 2 < (1ll << 31))) &&
        (!a->C || (al16(a->C) && a->ldc % 4 == 0 && (int64_t)a->M * a->ldc * 4 < (1ll << 31))) &&
        (!(a->flags & BMT_EPI_GATE) || (al16(a->gate) && a->ldg % 8 == 0)) &&
        (!(a->flags & BMT_EPI_RESIDUAL) || (al16(a->residual) && a->ldr % 4 == 0))) {
        if (a->precision != BMT_PREC_F16W2) p.Bl = nullptr;
        // 128-column weight chunks; the row groups (workgroups of one chunk) share the 32-row units of the activation
        const int cols = p.Chi ? (p.plane_cols > a->N ? p.plane_cols : a->N) : a->N;
        const int units = bmt_cdiv(a->M, 32), cus = bmt_device_cus();
        const int nch = bmt_cdiv(cols, 128);
        int rg = cus / nch < 1 ? 1 : cus / nch;
        const int upw = bmt_cdiv(units, rg);
        p.nk_rg = bmt_cdiv(units, upw); p.nk_upw = upw; p.tiles_n = nch;
        p.a_blk_n = a->a_blk_n; p.a_blk_k = a->a_blk_k;
        p.pipe = 4;
    }
    BMT_CHECK_ARG(!blk || blk_km || p.pipe == 4, "bmt_gemm_bf16: block products (a_blk_n) need the reduction-of-128 kernel's alignment (16-byte planes, N %% 8 == 0, alpha 1, no colsum / accumulate / split)");
    // pipe 5: the three-pass product over few rows (a decoder layer's own GEMMs, the greedy decoder's): 32 x 32 tiles, reduction over the waves
    if ((p.pipe == 0 || p.pipe == 2) && !a->a_kmajor && !a->b_kmajor && !a->conv_mode && a->splitk <= 1 && (int64_t)a->M * a->N <= BMT_SMALL_TILE_OUTPUTS &&
        (a->precision == BMT_PREC_BF16X3 || a->precision == BMT_PREC_BF16)) {      // (one bf16 pass: the small dX products, weight plane transposed)
        // long reductions over few tiles: one 32 x 32 tile per workgroup, the reduction over its four waves; else 64 x 64 blocks
        const int cols = p.Chi ? (p.plane_cols > a->N ? p.plane_cols : a->N) : a->N;
        p.pipe = 5;
        p.nk_rg = (a->Kpad >= 512 && bmt_cdiv(a->M, 32) * bmt_cdiv(cols, 32) <= 2 * bmt_device_cus()) ? 4 : 1;
        p.bm = p.nk_rg == 4 ? 32 : 64;
        p.tiles_n = bmt_cdiv(cols, p.bm);
    }
    p.tiles_m = bmt_cdiv(a->M, p.bm);
    const int bk = (a->precision == BMT_PREC_BF16X3 || (a->precision == BMT_PREC_F16W2 && p.pipe != 1)) ? 32 : 64;
    const int ktiles = a->Kpad / bk;
    const int tiles = p.tiles_m * p.tiles_n;
    // split-K (two passes through the workspace) for launches of fewer than 180 tiles with >= 12 stages: ~2 workgroups per CU, at least two
    // stages per split.  (Splitting the 200-tile products of the audio stream costs more in the workspace pass than their idle CUs.)
    if (a->splitk == 0 && two_pass && tiles < 180 && ktiles >= 12 && p.pipe != 3 && p.pipe != 4 && p.pipe != 5 && !blk_km) {
        int want = 512 / tiles, cap = ktiles / 2;
        if (want > 32) want = 32;
        splitk = want < cap ? want : cap;
        if (splitk < 1) splitk = 1;
    }
    if (splitk > ktiles) splitk = ktiles;
    if (splitk > 1 && two_pass) {              // bounded by the workspace
        const int64_t slab = (int64_t)tiles * p.bm * BN * (int64_t)sizeof(float);
        const int64_t fit = a->splitk_ws_bytes / slab;
        if (fit < splitk) splitk = (int)(fit < 1 ? 1 : fit);
    }
    p.kchunk = bmt_cdiv(ktiles, splitk) * bk;
    splitk = bmt_cdiv(a->Kpad, p.kchunk);
    BMT_CHECK_ARG(splitk == 1 || two_pass || accum, "bmt_gemm_bf16: split-K workspace too small");
    if (a->colsum && splitk > 1) splitk = 1, p.kchunk = a->Kpad;      // column sums come from the main kernel's epilogue only
    if (splitk > 1 && two_pass) { p.ws = a->splitk_ws; p.nsplit = splitk; }
    p.alpha = a->alpha; p.flags = a->flags; p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr;
    p.gate = a->gate; p.ldg = a->ldg; p.gate_scale = a->gate_scale;
    p.colsum = a->colsum;
    BMT_CHECK_ARG(!a->colsum || (a->C_hi && p.plane_vec && !a->C), "bmt_gemm_bf16: colsum needs 16-byte aligned plane-only output");
    p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    return BMT_OK;
}

extern "C" int bmt_gemm_bf16(const bmt_gemm_bf16_args* a, void* stream) {
    GemmB p;
    int splitk = 0;
    int rc = gemm_prepare(a, p, splitk, true);
    if (rc != BMT_OK) return rc;
    BMT_CHECK_ARG(!a->c_row_dev && !a->m_dev, "bmt_gemm_bf16: c_row_dev / m_dev belong to bmt_gemm_bf16_grouped");
    hipStream_t st_ = (hipStream_t)stream;
    const bool akm = a->a_kmajor != 0, bkm = a->b_kmajor != 0;
    const bool f16 = a->precision == BMT_PREC_F16 || a->precision == BMT_PREC_F16W2;
    if (f16 && (akm || bkm || a->conv_mode == 2)) {
        bmt_set_error("bmt_gemm_bf16: the fp16 precisions take row-major operands (forward products) only");
        return BMT_EINVAL;
    }
    // (the register-staged kernels run 8 waves of 32 x 64 on a 128-row tile: four waves per SIMD across two workgroups hide the stage
    // loop's LDS / barrier latency better than 4 waves of 64 x 64 did -- whole step -6 % in round 1)
    if (p.pipe == 3) {
        rc = f16 ? launch_wide<true>(p, st_) : launch_wide<false>(p, st_);
    } else if (p.pipe == 4) {
        rc = f16 ? launch_k128<true>(p, st_) : launch_k128<false>(p, st_);
    } else if (p.pipe == 5) {
        rc = a->precision == BMT_PREC_BF16X3 ? launch_small<3, false>(p, st_) : launch_small<1, false>(p, st_);
    } else if (a->conv_mode == 1 && f16) {      // the Conv1d forward through the LDS-DMA ring (the A rows shift by a tap per step: one scalar offset)
        if (p.bm == 256) rc = a->precision == BMT_PREC_F16W2 ? launch_pipe<2, true, 2, false, false, 1>(p, splitk, st_) : launch_pipe<1, true, 2, false, false, 1>(p, splitk, st_);
        else rc = a->precision == BMT_PREC_F16W2 ? launch_pipe<2, true, 1, false, false, 1>(p, splitk, st_) : launch_pipe<1, true, 1, false, false, 1>(p, splitk, st_);
    } else if (p.pipe == 1) {
        if (a->precision == BMT_PREC_F16W2) rc = launch_pipe<2, true, 2>(p, splitk, st_);
        else if (a->precision == BMT_PREC_F16) rc = launch_pipe<1, true, 2>(p, splitk, st_);
        else rc = launch_pipe<1, false, 2>(p, splitk, st_);
    } else if (p.pipe == 2) {
        if (a->precision == BMT_PREC_F16W2) rc = launch_pipe<2, true, 1>(p, splitk, st_);
        else if (a->precision == BMT_PREC_F16) rc = launch_pipe<1, true, 1>(p, splitk, st_);
        else rc = launch_pipe<1, false, 1>(p, splitk, st_);
    } else if (a->conv_mode == 1) {          // implicit Conv1d forward (split-bf16) / dX
        rc = a->precision == BMT_PREC_BF16X3 ? launch<3, 4, 1, false, false, 1>(p, splitk, st_) : launch<1, 4, 1, false, false, 1>(p, splitk, st_);
    } else if (a->conv_mode == 2) {          // implicit Conv1d dW
        rc = launch<1, 4, 1, true, true, 2>(p, splitk, st_);
    } else if (akm || bkm) {                 // k-major operands: the register-staged loop with its two tiles of prefetch (the LDS-DMA ring has room
                                             // for two stages only at two workgroups per CU and lost: dX class 1.74 -> 2.06 ms, round 3)
        if (akm && bkm) rc = launch<1, 4, 1, true, true>(p, splitk, st_);
        else if (bkm) rc = launch<1, 4, 1, false, true>(p, splitk, st_);
        else rc = launch<1, 4, 1, true, false>(p, splitk, st_);
    } else if (a->precision == BMT_PREC_BF16X3) {
        rc = launch<3, 4, 1>(p, splitk, st_);
    } else {
        bmt_set_error("bmt_gemm_bf16: no kernel for this operand combination");
        return BMT_EINVAL;
    }
    if (rc != BMT_OK || p.ws == nullptr) return rc;
    const int pc = p.Chi ? (p.plane_cols > p.N ? p.plane_cols : p.N) : p.N;
    const int64_t groups = (int64_t)p.M * ((pc + 3) / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)bmt_cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream, p);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16(split-K epilogue)");
    return BMT_OK;
}

extern "C" int bmt_gemm_small_batched(const bmt_gemm_bf16_args* a, const bmt_gemm_batch* b, void* stream) {
    BMT_CHECK_ARG(a && b && b->nb_outer > 0 && b->nb_inner > 0 && (int64_t)b->nb_outer * b->nb_inner <= 65535, "bmt_gemm_small_batched: bad batch");
    BMT_CHECK_ARG(BMT_SMALL_TILE_OUTPUTS > 0, "bmt_gemm_small_batched: the library was built without the 32 x 32 tile kernel");
    BMT_CHECK_ARG(!a->a_kmajor && !a->b_kmajor && !a->conv_mode && (a->splitk <= 1 || a->splitk == 4) && !a->rows_dev && !a->c_row_dev && !a->m_dev &&
                      !(a->flags & (BMT_EPI_RESIDUAL | BMT_EPI_GATE | BMT_EPI_ACCUM)),
                  "bmt_gemm_small_batched: row-major operands, no split, no residual / gate / accumulate");
    BMT_CHECK_ARG(a->precision == BMT_PREC_BF16 || a->precision == BMT_PREC_F16 || a->precision == BMT_PREC_BF16X3,
                  "bmt_gemm_small_batched: BMT_PREC_BF16, BMT_PREC_F16 or BMT_PREC_BF16X3");
    GemmB p;
    int splitk = 1;
    bmt_gemm_bf16_args a1 = *a;
    a1.splitk = 1;
    int rc = gemm_prepare(&a1, p, splitk, false);
    if (rc != BMT_OK) return rc;
    // every product writes exactly its N columns (a neighbour's may follow); 16-byte plane stores need aligned offsets
    if (p.Chi) p.plane_cols = a->N;
    const int64_t ldp2 = b->ldp2 ? b->ldp2 : a->ldp;
    p.plane_vec = p.plane_vec && (a->N % 8 == 0) && (ldp2 % 8 == 0) && !((b->p_off_o | b->p_off_i | b->p2_off_o | b->p2_off_i) & 7);
    const int cols = a->N, nb = b->nb_outer * b->nb_inner;
    p.pipe = 5;
    // splitk 0: the library chooses (the reduction over a workgroup's waves for long reductions over few tiles; measured per product at
    // configs[1]'s shapes: profiles/r05_s_raw_products_time.txt); 1 / 4: the caller does (64 x 64 blocks / one tile per workgroup)
    p.nk_rg = a->splitk == 4 ? 4 : (a->splitk == 1 ? 1 : ((a->Kpad >= 512 && (int64_t)bmt_cdiv(a->M, 32) * bmt_cdiv(cols, 32) * nb <= 2 * bmt_device_cus()) ? 4 : 1));
    p.bm = p.nk_rg == 4 ? 32 : 64;
    p.tiles_m = bmt_cdiv(a->M, p.bm);
    p.tiles_n = bmt_cdiv(cols, p.bm);
    SmallBatch bt;
    bt.nb_inner = b->nb_inner;
    bt.a_off_o = b->a_off_o; bt.a_off_i = b->a_off_i; bt.b_off_o = b->b_off_o; bt.b_off_i = b->b_off_i;
    bt.c_off_o = b->c_off_o; bt.c_off_i = b->c_off_i; bt.p_off_o = b->p_off_o; bt.p_off_i = b->p_off_i;
    bt.p2_off_o = b->p2_off_o; bt.p2_off_i = b->p2_off_i; bt.ldp2 = b->ldp2;
    bt.bias_off_i = b->bias_off_i; bt.drop_off_o = b->drop_off_o; bt.drop_off_i = b->drop_off_i;
    bt.b_rows_dev = b->b_rows_dev;
    bt.a_div = b->a_div; bt.c_div = b->c_div; bt.p_div = b->p_div; bt.p2_div = b->p2_div;
    bt.a_qs = b->a_qs; bt.c_qs = b->c_qs; bt.p_qs = b->p_qs; bt.p2_qs = b->p2_qs;
    BMT_CHECK_ARG(b->a_div >= 0 && b->c_div >= 0 && b->p_div >= 0 && b->p2_div >= 0 && !((b->a_qs | b->p_qs | b->p2_qs) & 7) && !(b->c_qs & 3),
                  "bmt_gemm_small_batched: block-row strides must keep 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    if (a->precision == BMT_PREC_BF16X3) return launch_small<3, false>(p, st, &bt, nb);
    if (a->precision == BMT_PREC_F16) return launch_small<1, true>(p, st, &bt, nb);
    return launch_small<1, false>(p, st, &bt, nb);
}

// ---- grouped launch (see gemm_bf16_grouped_kernel).  The descriptor table and the per-XCD segment lists live in device memory
// and are written by small kernels whose ARGUMENTS carry them: nothing is read from host memory when the launch executes, so the
// sequence can be captured in a hipGraph and replayed (a memcpy node would re-read a host buffer that may have changed).
constexpr int GEMM_PACK_N = 12;      // descriptors per table-writer launch (they travel in its kernel arguments)
struct GemmPack {
    GemmB d[GEMM_PACK_N];
    int n, base;
};
static_assert(sizeof(GemmPack) <= 4000, "descriptor pack must fit the kernel argument buffer");
struct SegPack {
    XcdSeg s[2][XCD_MAXSEG];
    int n[2], xcd0;
};
static_assert(sizeof(SegPack) <= 4000, "segment pack must fit the kernel argument buffer");

__global__ void gemm_table_write_kernel(const GemmPack pk, GemmB* __restrict__ table) {
    const int i = blockIdx.x;
    if (i >= pk.n) return;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&pk.d[i]);
    uint32_t* dst = reinterpret_cast<uint32_t*>(table + pk.base + i);
    for (int w = threadIdx.x; w < (int)(sizeof(GemmB) / 4); w += blockDim.x) dst[w] = src[w];
}
__global__ void gemm_segs_write_kernel(const SegPack pk, XcdSeg* __restrict__ segs, int* __restrict__ nseg) {
    const int x = blockIdx.x;                  // 0, 1: XCD pk.xcd0 + x
    for (int i = threadIdx.x; i < pk.n[x]; i += blockDim.x) segs[(pk.xcd0 + x) * XCD_MAXSEG + i] = pk.s[x][i];
    if (threadIdx.x == 0) nseg[pk.xcd0 + x] = pk.n[x];
}

extern "C" size_t bmt_gemm_bf16_grouped_ws_bytes(int nprob) {
    return nprob <= 0 ? 0 : (size_t)nprob * sizeof(GemmB) + 8 * XCD_MAXSEG * sizeof(XcdSeg) + 8 * sizeof(int) + 512;
}

// ABI 9: the grouped launch in two calls, so that the ~25 table-writer launches (4 us each, dependent on nothing but the buffers' addresses)
// can sit on ANOTHER stream than the product -- a stream the caller forks from the step's beginning: inside a captured step they then run
// beside the forward pass instead of in front of the grouped launch on the critical path (profiles/r05_zz_replay_dispatches.csv: ~100 us of
// tiny kernels before the weight-gradient launch, ~50 us before each memory-gradient launch).  launch[0] = grid slots, launch[1] = 1 when an
// output is a packed row range: what bmt_gemm_bf16_grouped_run needs besides the workspace.
static int grouped_tables(const bmt_gemm_bf16_args* args, int nprob, void* ws, size_t ws_bytes, int* launch, hipStream_t st, void* host_image);
static int grouped_run(void* ws, int nprob, const int* launch, hipStream_t st);

extern "C" int bmt_gemm_bf16_grouped_tables(const bmt_gemm_bf16_args* args, int nprob, void* ws, size_t ws_bytes, int* launch, void* stream) {
    BMT_CHECK_ARG(launch != nullptr, "bmt_gemm_bf16_grouped_tables: launch is NULL");
    return grouped_tables(args, nprob, ws, ws_bytes, launch, (hipStream_t)stream, nullptr);
}
// ABI 10 -- the SAME bytes _tables leaves in ws (descriptor table | per-XCD segment lists | segment counts), written into HOST memory
// instead: no launch, no device access.  For a caller that moves them itself -- a captured step copies them into a table buffer of its own
// ONCE, at capture time, on a stream that is not capturing (the addresses a captured launch works on never change: profiles/r06_o_*: the
// ~11 table-writer launches per grouped launch, wherever their graph branch forks from, execute where they were captured -- in front of the
// product), and replays _run alone.
extern "C" int bmt_gemm_bf16_grouped_image(const bmt_gemm_bf16_args* args, int nprob, void* host_image, size_t bytes, int* launch) {
    BMT_CHECK_ARG(launch != nullptr && host_image != nullptr, "bmt_gemm_bf16_grouped_image: NULL argument");
    return grouped_tables(args, nprob, host_image, bytes, launch, nullptr, host_image);
}
extern "C" int bmt_gemm_bf16_grouped_run(void* ws, int nprob, const int* launch, void* stream) {
    BMT_CHECK_ARG(ws && launch && nprob > 0 && launch[0] > 0, "bmt_gemm_bf16_grouped_run: bad arguments");
    return grouped_run(ws, nprob, launch, (hipStream_t)stream);
}
extern "C" int bmt_gemm_bf16_grouped(const bmt_gemm_bf16_args* args, int nprob, void* ws, size_t ws_bytes, void* stream) {
    int launch[2] = {0, 0};
    const int rc = grouped_tables(args, nprob, ws, ws_bytes, launch, (hipStream_t)stream, nullptr);
    return rc != BMT_OK ? rc : grouped_run(ws, nprob, launch, (hipStream_t)stream);
}

static int grouped_tables(const bmt_gemm_bf16_args* args, int nprob, void* ws, size_t ws_bytes, int* launch, hipStream_t st, void* host_image) {
    BMT_CHECK_ARG(args && ws && nprob > 0 && nprob <= 4096, "bmt_gemm_bf16_grouped: bad arguments");
    BMT_CHECK_ARG(ws_bytes >= bmt_gemm_bf16_grouped_ws_bytes(nprob) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                  "bmt_gemm_bf16_grouped: workspace too small or not 16-byte aligned");
    static_assert(sizeof(GemmB) % 4 == 0, "descriptor is copied word-wise");
    struct Prob { GemmB p; int tiles, stages, nsplit; };
    // reduction chunk (64-row stages) of a work item; 0 = whole reductions.  Partial products accumulate with the fp32 atomics the
    // unsplit launch uses too (C += alpha A B is the only epilogue here)
    // measured (tools/gpu_ab.sh, whole step; PMC: the unsplit launch fetched 6 GB for ~1 GB of unique operands at 5.1 TB/s): chunks
    // of 16 / 32 / 64 / 100 / 134 / 200 stages -> 1.48 / 1.14 / 1.02 / 1.02 / 1.06 / 1.05 ms against 1.235 ms unsplit
    constexpr int chunk = 64;
    Prob* pr = (Prob*)malloc(sizeof(Prob) * (size_t)nprob);
    int* order = (int*)malloc(sizeof(int) * (size_t)nprob);
    SegPack* sp = (SegPack*)malloc(sizeof(SegPack) * 4);
    if (!pr || !order || !sp) { free(pr); free(order); free(sp); bmt_set_error("bmt_gemm_bf16_grouped: out of host memory"); return BMT_EINVAL; }
    int rc = BMT_OK;
    double total_work = 0.0;
    bool placed = false;                 // some output is a packed row range (c_row_dev): the launch runs under its own kernel name
    for (int i = 0; i < nprob && rc == BMT_OK; ++i) {
        const bmt_gemm_bf16_args* a = args + i;
        placed = placed || a->c_row_dev != nullptr;
        if (!(a->precision == BMT_PREC_BF16 && a->a_kmajor && a->b_kmajor && a->conv_mode == 0 && a->C && !a->C_hi && !a->colsum)) {
            bmt_set_error("bmt_gemm_bf16_grouped: problem %d: the grouped launch takes single-pass GEMMs with both operands k-major and "
                          "fp32 output", i);
            rc = BMT_EINVAL;
            break;
        }
        if (!(a->flags == BMT_EPI_ACCUM || (a->flags == 0 && a->c_row_dev != nullptr && a->Kpad / 64 <= chunk + chunk / 2))) {
            bmt_set_error("bmt_gemm_bf16_grouped: problem %d: flags must be BMT_EPI_ACCUM, or 0 (plain stores) for a placed output whose reduction "
                          "is not split (Kpad <= %d)", i, 64 * (chunk + chunk / 2));
            rc = BMT_EINVAL;
            break;
        }
        int splitk = 1;
        rc = gemm_prepare(a, pr[i].p, splitk, false);
        pr[i].tiles = pr[i].p.tiles_m * pr[i].p.tiles_n;
        pr[i].stages = a->Kpad / 64;
        pr[i].nsplit = 1;
        if (chunk > 0 && pr[i].stages > chunk + chunk / 2) {
            pr[i].nsplit = bmt_cdiv(pr[i].stages, chunk);
            pr[i].p.kchunk = bmt_cdiv(pr[i].stages, pr[i].nsplit) * 64;
            pr[i].nsplit = bmt_cdiv(a->Kpad, pr[i].p.kchunk);
        }
        total_work += (double)pr[i].tiles * (pr[i].stages + 2);      // + prologue / epilogue of a tile
        order[i] = i;
    }
    if (rc != BMT_OK) { free(pr); free(order); free(sp); return rc; }
    // largest problems first (stable insertion sort: nprob is small)
    auto work = [&](int i) { return (double)pr[i].tiles * (pr[i].stages + 2); };
    for (int i = 1; i < nprob; ++i) {
        const int o = order[i];
        int j = i - 1;
        while (j >= 0 && work(order[j]) < work(o)) { order[j + 1] = order[j]; --j; }
        order[j + 1] = o;
    }
    // pack onto the 8 XCDs: a problem goes to the least loaded XCD; one that is more than ~60 % of an XCD's fair share is cut
    // into 2, 4 or 8 contiguous tile ranges first (placed independently)
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int slots[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double share = total_work / 8.0;
    bool overflow = false;
    for (int oi = 0; oi < nprob && !overflow; ++oi) {
        const int i = order[oi];
        int parts = 1;
        while (parts < 8 && work(i) / parts > 0.6 * share) parts *= 2;
        if (parts > pr[i].tiles) parts = 1;
        const int per = (pr[i].tiles + parts - 1) / parts;
        for (int part = 0; part < parts; ++part) {
            const int t0 = part * per, cnt = (t0 + per <= pr[i].tiles) ? per : pr[i].tiles - t0;
            if (cnt <= 0) break;
            int x = 0;
            for (int k = 1; k < 8; ++k) if (load[k] < load[x]) x = k;
            if (ns[x] >= XCD_MAXSEG) { overflow = true; break; }
            XcdSeg& sg = sp[x / 2].s[x & 1][ns[x]++];
            sg.first_slot = slots[x]; sg.prob = i; sg.tile_off = t0; sg.count = cnt; sg.nsplit = pr[i].nsplit;
            slots[x] += cnt * pr[i].nsplit;
            load[x] += (double)cnt * (pr[i].stages + 2);
        }
    }
    if (overflow) { free(pr); free(order); free(sp); bmt_set_error("bmt_gemm_bf16_grouped: too many segments for one XCD"); return BMT_EINVAL; }
    GemmB* table = reinterpret_cast<GemmB*>(ws);
    char* tail = reinterpret_cast<char*>(ws) + (((size_t)nprob * sizeof(GemmB) + 15) & ~(size_t)15);
    XcdSeg* segs = reinterpret_cast<XcdSeg*>(tail);
    int* nseg = reinterpret_cast<int*>(tail + 8 * XCD_MAXSEG * sizeof(XcdSeg));
    int max_slots = 0;
    for (int x = 0; x < 8; ++x) max_slots = slots[x] > max_slots ? slots[x] : max_slots;
    if (host_image != nullptr) {         // (ws IS the host image: table, segs and nseg point into it)
        memset(host_image, 0, bmt_gemm_bf16_grouped_ws_bytes(nprob));
        for (int i = 0; i < nprob; ++i) table[i] = pr[i].p;
        for (int x = 0; x < 8; ++x) {
            for (int i = 0; i < ns[x]; ++i) segs[x * XCD_MAXSEG + i] = sp[x / 2].s[x & 1][i];
            nseg[x] = ns[x];
        }
        free(pr);
        free(order);
        free(sp);
        launch[0] = max_slots;
        launch[1] = placed ? 1 : 0;
        return BMT_OK;
    }
    GemmPack pk;
    for (int base = 0; base < nprob; base += GEMM_PACK_N) {
        pk.n = nprob - base < GEMM_PACK_N ? nprob - base : GEMM_PACK_N;
        pk.base = base;
        for (int i = 0; i < pk.n; ++i) pk.d[i] = pr[base + i].p;
        hipLaunchKernelGGL(gemm_table_write_kernel, dim3(pk.n), dim3(64), 0, st, pk, table);
    }
    for (int q = 0; q < 4; ++q) {
        sp[q].n[0] = ns[2 * q]; sp[q].n[1] = ns[2 * q + 1]; sp[q].xcd0 = 2 * q;
        hipLaunchKernelGGL(gemm_segs_write_kernel, dim3(2), dim3(64), 0, st, sp[q], segs, nseg);
    }
    free(pr);
    free(order);
    free(sp);
    BMT_CHECK_LAUNCH("bmt_gemm_bf16_grouped(table)");
    launch[0] = max_slots;
    launch[1] = placed ? 1 : 0;
    return BMT_OK;
}

static int grouped_run(void* ws, int nprob, const int* launch, hipStream_t st) {
    const int max_slots = launch[0];
    const bool placed = launch[1] != 0;
    GemmB* table = reinterpret_cast<GemmB*>(ws);
    char* tail = reinterpret_cast<char*>(ws) + (((size_t)nprob * sizeof(GemmB) + 15) & ~(size_t)15);
    XcdSeg* segs = reinterpret_cast<XcdSeg*>(tail);
    int* nseg = reinterpret_cast<int*>(tail + 8 * XCD_MAXSEG * sizeof(XcdSeg));
    constexpr int BK = 64, BMr = 128;
    constexpr int stage = BK * km_rs<BMr>() + BK * km_rs<BN>();
    constexpr int lds = (2 * stage > BMr * BN * 4) ? 2 * stage : BMr * BN * 4;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<1, 4, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    if (placed) {
        static bool done2 = false;
        if (!done2) {
            (void)hipFuncSetAttribute((const void*)gemm_memory_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done2 = true;
        }
        hipLaunchKernelGGL(gemm_memory_grad_kernel, dim3(8 * max_slots), dim3(512), lds, st, table, segs, nseg);
    } else {
        hipLaunchKernelGGL((gemm_bf16_grouped_kernel<1, 4, 1, true, true>), dim3(8 * max_slots), dim3(512), lds, st, table, segs, nseg);
    }
    BMT_CHECK_LAUNCH("bmt_gemm_bf16_grouped");
    return BMT_OK;
}

static int fill_desc(PlaneDesc& d, const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp,
                     uint16_t* hiT, uint16_t* loT, int64_t ldpT, float* colsum) {
    BMT_CHECK_ARG(src && (hi || fh || hiT || colsum) && R > 0 && C > 0, "bmt_planes: bad args");
    BMT_CHECK_ARG(!(hi || fh) || ldp >= C, "bmt_planes: ldp < C");
    BMT_CHECK_ARG((!lo || hi) && (!fl || fh), "bmt_planes: a lo plane without its hi plane");
    BMT_CHECK_ARG(!hiT || ldpT >= R, "bmt_planes: ldpT < R");
    // padding written with zeros: up to the next multiple of 64 (bounded by the row stride)
    d.src = src; d.ld = ld; d.R = R; d.C = C; d.hi = hi; d.lo = lo; d.fh = fh; d.fl = fl; d.ldp = ldp; d.hiT = hiT; d.loT = loT; d.ldpT = ldpT;
    d.colsum = colsum;
    d.drop_p = 0.f; d.drop_site = 0; d.drop_rng = nullptr;
    d.gate = nullptr; d.ldgate = 0; d.gate_scale = 1.f;
    d.rows_dev = nullptr;
    d.pcols = (hi || fh) ? (int)(((C + 63) / 64 * 64) < ldp ? ((C + 63) / 64 * 64) : ldp) : 0;
    d.pcolsT = hiT ? (int)(((R + 63) / 64 * 64) < ldpT ? ((R + 63) / 64 * 64) : ldpT) : 0;
    d.tiles_x = bmt_cdiv(d.pcols > C ? d.pcols : C, 64);
    d.tiles_y = bmt_cdiv(d.pcolsT > R ? d.pcolsT : R, 64);
    return BMT_OK;
}

extern "C" int bmt_planes(const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp,
                          uint16_t* hiT, uint16_t* loT, int64_t ldpT, float* colsum, const int* rows_dev, void* stream) {
    PlaneDesc d;
    int rc = fill_desc(d, src, ld, R, C, hi, lo, fh, fl, ldp, hiT, loT, ldpT, colsum);
    if (rc) return rc;
    BMT_CHECK_ARG(!rows_dev || !hiT, "bmt_planes: packed rows (rows_dev) with a transposed plane");
    d.rows_dev = rows_dev;
    if (planes_rows_ok(d)) hipLaunchKernelGGL(planes_rows_kernel, dim3(bmt_cdiv(d.pcols, 128), bmt_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(planes_kernel, dim3(d.tiles_x, d.tiles_y), dim3(256), 0, (hipStream_t)stream, d);
    BMT_CHECK_LAUNCH("bmt_planes");
    return BMT_OK;
}

extern "C" int bmt_planes_dropout(const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp,
                                  uint16_t* hiT, uint16_t* loT, int64_t ldpT, float* colsum, float drop_p, const uint64_t* rng, uint32_t site,
                                  const int* rows_dev, void* stream) {
    PlaneDesc d;
    int rc = fill_desc(d, src, ld, R, C, hi, lo, fh, fl, ldp, hiT, loT, ldpT, colsum);
    if (rc) return rc;
    BMT_CHECK_ARG(!rows_dev || !hiT, "bmt_planes_dropout: packed rows (rows_dev) with a transposed plane");
    d.rows_dev = rows_dev;
    BMT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "bmt_planes_dropout: bad dropout arguments");
    d.drop_p = drop_p; d.drop_site = site; d.drop_rng = rng;
    if (planes_rows_ok(d)) hipLaunchKernelGGL(planes_rows_kernel, dim3(bmt_cdiv(d.pcols, 128), bmt_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(planes_kernel, dim3(d.tiles_x, d.tiles_y), dim3(256), 0, (hipStream_t)stream, d);
    BMT_CHECK_LAUNCH("bmt_planes_dropout");
    return BMT_OK;
}

// planes (+ column sums) of  dz = (gate != 0) ? src * gate_scale : 0  -- the gradient through relu (/ dropout before it) taken from the saved
// forward output, without a dz tensor: bmt_gate -> bmt_planes + bmt_colsum in one pass (the 1 x 1 layers of the proposal heads, ABI 5)
extern "C" int bmt_planes_gate(const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp,
                               float* colsum, const float* gate, int64_t ldgate, float gate_scale, void* stream) {
    PlaneDesc d;
    int rc = fill_desc(d, src, ld, R, C, hi, lo, fh, fl, ldp, nullptr, nullptr, 0, colsum);
    if (rc) return rc;
    BMT_CHECK_ARG(gate && ldgate >= C, "bmt_planes_gate: bad gate");
    d.gate = gate; d.ldgate = ldgate; d.gate_scale = gate_scale;
    if (planes_rows_ok(d) && al16(gate)) hipLaunchKernelGGL(planes_rows_kernel, dim3(bmt_cdiv(d.pcols, 128), bmt_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(planes_kernel, dim3(d.tiles_x, d.tiles_y), dim3(256), 0, (hipStream_t)stream, d);
    BMT_CHECK_LAUNCH("bmt_planes_gate");
    return BMT_OK;
}

// host helper: fill one descriptor of the multi-tensor table (the caller uploads the table to device memory)
extern "C" int bmt_planes_desc(void* desc_out, const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl,
                               int64_t ldp, uint16_t* hiT, uint16_t* loT, int64_t ldpT) {
    BMT_CHECK_ARG(desc_out, "bmt_planes_desc: null");
    PlaneDesc d;
    memset(&d, 0, sizeof(d));
    int rc = fill_desc(d, src, ld, R, C, hi, lo, fh, fl, ldp, hiT, loT, ldpT, nullptr);
    if (rc) return rc;
    d.rows_ok = planes_straight_ok(d) ? (d.hiT ? 2 : 1) : 0;
    memcpy(desc_out, &d, sizeof(d));
    return BMT_OK;
}
extern "C" int bmt_planes_desc_bytes(void) { return (int)sizeof(PlaneDesc); }

extern "C" int bmt_planes_multi(const void* table_dev, int n_tensors, void* stream) {
    BMT_CHECK_ARG(table_dev && n_tensors > 0 && n_tensors <= 65535, "bmt_planes_multi: bad args");
    hipLaunchKernelGGL(planes_multi_kernel, dim3(64, n_tensors), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PlaneDesc*>(table_dev));
    BMT_CHECK_LAUNCH("bmt_planes_multi");
    return BMT_OK;
}

extern "C" int bmt_planes_desc_tiles(const void* desc_host) {
    if (!desc_host) return 0;
    PlaneDesc d;
    memcpy(&d, desc_host, sizeof(d));
    return planes_desc_tiles(d);
}
extern "C" int bmt_planes_multi_flat(const void* table_dev, const int* prefix_dev, int n_tensors, int total_tiles, void* stream) {
    BMT_CHECK_ARG(table_dev && prefix_dev && n_tensors > 0 && total_tiles > 0, "bmt_planes_multi_flat: bad args");
    hipLaunchKernelGGL(planes_multi_flat_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PlaneDesc*>(table_dev), prefix_dev, n_tensors);
    BMT_CHECK_LAUNCH("bmt_planes_multi_flat");
    return BMT_OK;
}

// bf16 [R][ld] -> transposed bf16 [C][ldT] (zero padded up to min(round_up(R,64), ldT)): operand planes of saved activations
// (FFN hidden) for the weight-gradient GEMM, whose reduction runs over rows.
namespace {
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ src, int64_t ld, int R, int C,
                                                              uint16_t* __restrict__ dst, int64_t ldT, int pcolsT) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < R && c < C) ? src[(int64_t)r * ld + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < C && r < pcolsT) dst[(int64_t)c * ldT + r] = tile[tx][ty * 16 + i];
    }
}
}  // namespace

extern "C" int bmt_transpose_bf16(const uint16_t* src, int64_t ld, int R, int C, uint16_t* dst, int64_t ldT, void* stream) {
    BMT_CHECK_ARG(src && dst && R > 0 && C > 0 && ldT >= R, "bmt_transpose_bf16: bad args");
    const int pcolsT = (int)(((R + 63) / 64 * 64) < ldT ? ((R + 63) / 64 * 64) : ldT);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(bmt_cdiv(C, 64), bmt_cdiv(pcolsT, 64)), dim3(256), 0, (hipStream_t)stream, src, ld, R, C,
                       dst, ldT, pcolsT);
    BMT_CHECK_LAUNCH("bmt_transpose_bf16");
    return BMT_OK;
}

// ---------------------------------------------------------------- halo-padded planes for the implicit Conv1d
// x fp32 (B, S, C) -> bf16 planes [B * (S + 2 halo) + tail][ldp]: row b * (S + 2 halo) + halo + s holds x[b, s, :] (hi = bf16(x),
// lo = bf16(x - hi), optional), every other row and the columns [C, ldp) are zero.
namespace {
__global__ __launch_bounds__(256) void pad_planes_kernel(const float* __restrict__ x, int B, int S, int C, int halo, int tail,
                                                          uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int lo_f16, int64_t ldp) {
    const int groups = (int)(ldp / 8);
    const int64_t total = ((int64_t)B * (S + 2 * halo) + tail) * groups;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t r = idx / groups;
    const int c0 = (int)(idx % groups) * 8;
    const int SP = S + 2 * halo;
    const int b = (int)(r / SP), s = (int)(r % SP) - halo;
    u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
    if (b < B && s >= 0 && s < S) {
        const float* src = x + ((int64_t)b * S + s) * C;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? src[c0 + j] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t hh, ll;
            split_bf2(v[2 * j], v[2 * j + 1], hh, ll);
            if (lo_f16) ll = pack_h2(v[2 * j], v[2 * j + 1]);        // second plane = fp16(x) (fp16 forward operand)
            h[j] = hh; l[j] = ll;
        }
    }
    *reinterpret_cast<u32x4*>(hi + r * ldp + c0) = h;
    if (lo) *reinterpret_cast<u32x4*>(lo + r * ldp + c0) = l;
}
}  // namespace

extern "C" int bmt_pad_planes(const float* x, int B, int S, int C, int halo, int tail, uint16_t* hi, uint16_t* lo, int lo_f16, int64_t ldp,
                              void* stream) {
    BMT_CHECK_ARG(x && hi && B > 0 && S > 0 && C > 0 && halo >= 0 && tail >= 0 && ldp >= C && ldp % 8 == 0, "bmt_pad_planes: bad args");
    if (!al16(hi) || (lo && !al16(lo))) { bmt_set_error("bmt_pad_planes: planes must be 16-byte aligned"); return BMT_EALIGN; }
    const int64_t total = ((int64_t)B * (S + 2 * halo) + tail) * (ldp / 8);
    hipLaunchKernelGGL(pad_planes_kernel, dim3((unsigned)bmt_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, B, S, C, halo, tail, hi, lo, lo_f16, ldp);
    BMT_CHECK_LAUNCH("bmt_pad_planes");
    return BMT_OK;
}


// ---------------------------------------------------------------- the same with the relu / dropout derivative folded in, and the column sums
// halo-padded bf16 plane of  dz = (y != 0) ? dy * scale : 0  (ConvKFn.backward: the gradient operand of the implicit Conv1d's dX and dW)
// plus colsum[c] += sum over (b, s) of the bf16-rounded dz -- the convolution's bias gradient -- from the same pass: bmt_gate ->
// bmt_pad_planes -> bmt_colsum over (B, S, C) fp32 became one read of dy and y.  Block = 64 plane rows; a thread owns one 8-column group
// and walks the rows of its row lane.
namespace {
__global__ __launch_bounds__(256) void pad_planes_gate_kernel(const float* __restrict__ dy, const float* __restrict__ y, float scale, int B, int S, int C,
                                                               int halo, int tail, uint16_t* __restrict__ hi, int64_t ldp, float* __restrict__ colsum) {
    // block = 64 plane rows x 128 columns (the geometry of planes_rows_tile): a thread owns 8 columns of 4 rows, 16-byte loads / stores
    __shared__ float cs[16][129];
    const int tid = threadIdx.x;
    const int cg = (tid & 15) * 8, rt = tid >> 4;
    const int c0 = blockIdx.x * 128 + cg;
    const int64_t rows = (int64_t)B * (S + 2 * halo) + tail, r0 = (int64_t)blockIdx.y * 64;
    const int SP = S + 2 * halo;
    const bool vec = (C % 8 == 0);
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < ldp) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int64_t r = r0 + ps * 16 + rt;
            if (r >= rows) continue;
            const int b = (int)(r / SP), s = (int)(r % SP) - halo;
            u32x4 h = {0u, 0u, 0u, 0u};
            if (b < B && s >= 0 && s < S && c0 < C) {
                const int64_t o = ((int64_t)b * S + s) * C + c0;
                float v[8], g[8];
                if (vec) {
                    const float4 a0 = *reinterpret_cast<const float4*>(dy + o), a1 = *reinterpret_cast<const float4*>(dy + o + 4);
                    const float4 g0 = *reinterpret_cast<const float4*>(y + o), g1 = *reinterpret_cast<const float4*>(y + o + 4);
                    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
                    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] = (c0 + j < C) ? dy[o + j] : 0.f; g[j] = (c0 + j < C) ? y[o + j] : 0.f; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = g[j] != 0.f ? v[j] * scale : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = pack_bf2(v[2 * j], v[2 * j + 1]);
                    h[j] = w;
                    part[2 * j] += __uint_as_float(w << 16);
                    part[2 * j + 1] += __uint_as_float(w & 0xffff0000u);
                }
            }
            *reinterpret_cast<u32x4*>(hi + r * ldp + c0) = h;
        }
    }
    if (colsum) {
#pragma unroll
        for (int q = 0; q < 8; ++q) cs[rt][cg + q] = part[q];
        __syncthreads();
        if (tid < 128 && blockIdx.x * 128 + tid < C) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += cs[i][tid];
            if (t != 0.f) atomicAdd(colsum + blockIdx.x * 128 + tid, t);
        }
    }
}

// Conv1d weight [N][C][k] (state_dict layout, contiguous) <-> the tap-major operand [N][k][cin_pad] of the implicit GEMM:
//   TO_PLANES:  planes[n][tap * cin_pad + c] = W[n][c][tap]  (c >= C: zero), any of the four planes, row stride ldp
//   else:       G[n][c][tap] += dWp[n][tap * cin_pad + c]     (the weight gradient back into the parameter's layout)
// One workgroup per (n, 64-channel block): the (64 x k) sub-matrix goes through LDS, both sides are read / written in contiguous runs.
// Replaces zeros + a permuting copy + bmt_planes (forward) and a strided framework add (backward): 612 MB of fp32 weights per step at
// configs[3] were moved four times.
template <bool TO_PLANES>
__global__ __launch_bounds__(256) void conv_weight_kernel(float* __restrict__ W, int N, int C, int k, int cin_pad, uint16_t* __restrict__ hi,
                                                           uint16_t* __restrict__ lo, uint16_t* __restrict__ fh, uint16_t* __restrict__ fl, int64_t ldp,
                                                           float* __restrict__ dWp, int64_t ldw) {
    extern __shared__ float tile[];                   // [64][k + 1]
    const int n = blockIdx.x, c0 = blockIdx.y * 64, kp = k + 1;
    const int nc = min(64, cin_pad - c0);             // channels of this block (zero padded past C)
    if (TO_PLANES) {
        for (int i = threadIdx.x; i < 64 * k; i += 256) {
            const int c = i / k, t = i % k;
            tile[c * kp + t] = (c0 + c < C) ? W[((int64_t)n * C + c0 + c) * k + t] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < k * nc; i += 256) {
            const int t = i / nc, c = i % nc;
            const float v = tile[c * kp + t];
            const int64_t o = (int64_t)n * ldp + (int64_t)t * cin_pad + c0 + c;
            if (hi) {
                const __bf16 h = (__bf16)v;
                hi[o] = __builtin_bit_cast(uint16_t, h);
                if (lo) lo[o] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
            }
            if (fh) {
                const _Float16 h = (_Float16)v;
                fh[o] = __builtin_bit_cast(uint16_t, h);
                if (fl) fl[o] = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)h));
            }
        }
    } else {
        for (int i = threadIdx.x; i < k * nc; i += 256) {
            const int t = i / nc, c = i % nc;
            tile[c * kp + t] = dWp[(int64_t)n * ldw + (int64_t)t * cin_pad + c0 + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * k; i += 256) {
            const int c = i / k, t = i % k;
            if (c0 + c < C) W[((int64_t)n * C + c0 + c) * k + t] += tile[c * kp + t];
        }
    }
}
}  // namespace

extern "C" int bmt_pad_planes_gate(const float* dy, const float* y, float scale, int B, int S, int C, int halo, int tail, uint16_t* hi, int64_t ldp,
                                   float* colsum, void* stream) {
    BMT_CHECK_ARG(dy && y && hi && B > 0 && S > 0 && C > 0 && halo >= 0 && tail >= 0 && ldp >= C && ldp % 8 == 0,
                  "bmt_pad_planes_gate: bad args (row stride a multiple of 8)");
    BMT_CHECK_ARG(C % 8 != 0 || (((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y)) & 15) == 0), "bmt_pad_planes_gate: 16-byte aligned tensors");
    if (!al16(hi)) { bmt_set_error("bmt_pad_planes_gate: the plane must be 16-byte aligned"); return BMT_EALIGN; }
    const int64_t rows = (int64_t)B * (S + 2 * halo) + tail;
    hipLaunchKernelGGL(pad_planes_gate_kernel, dim3((unsigned)bmt_cdiv(ldp, 128), (unsigned)bmt_cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream, dy, y,
                       scale, B, S, C, halo, tail, hi, ldp, colsum);
    BMT_CHECK_LAUNCH("bmt_pad_planes_gate");
    return BMT_OK;
}

extern "C" int bmt_conv_weight_planes(const float* W, int N, int C, int k, int cin_pad, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl,
                                      int64_t ldp, void* stream) {
    BMT_CHECK_ARG(W && (hi || fh) && N > 0 && C > 0 && k > 0 && cin_pad >= C && cin_pad % 64 == 0 && ldp >= (int64_t)k * cin_pad && k <= 600,
                  "bmt_conv_weight_planes: bad args (cin_pad a multiple of 64 >= C, ldp >= k * cin_pad, k <= 600)");
    BMT_CHECK_ARG((!lo || hi) && (!fl || fh), "bmt_conv_weight_planes: a lo plane without its hi plane");
    const size_t lds = (size_t)64 * (k + 1) * sizeof(float);
    if (hipFuncSetAttribute((const void*)conv_weight_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        bmt_set_error("conv weight re-layout: %zu bytes of LDS per workgroup for k = %d taps are more than this device offers", lds, k);
        return BMT_EINVAL;
    }
    hipLaunchKernelGGL(conv_weight_kernel<true>, dim3(N, cin_pad / 64), dim3(256), lds, (hipStream_t)stream, const_cast<float*>(W), N, C, k, cin_pad, hi, lo,
                       fh, fl, ldp, nullptr, 0);
    BMT_CHECK_LAUNCH("bmt_conv_weight_planes");
    return BMT_OK;
}

extern "C" int bmt_conv_weight_grad(const float* dWp, int64_t ldw, int N, int C, int k, int cin_pad, float* grad, void* stream) {
    BMT_CHECK_ARG(dWp && grad && N > 0 && C > 0 && k > 0 && cin_pad >= C && cin_pad % 64 == 0 && ldw >= (int64_t)k * cin_pad && k <= 600,
                  "bmt_conv_weight_grad: bad args");
    const size_t lds = (size_t)64 * (k + 1) * sizeof(float);
    if (hipFuncSetAttribute((const void*)conv_weight_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        bmt_set_error("conv weight re-layout: %zu bytes of LDS per workgroup for k = %d taps are more than this device offers", lds, k);
        return BMT_EINVAL;
    }
    hipLaunchKernelGGL(conv_weight_kernel<false>, dim3(N, cin_pad / 64), dim3(256), lds, (hipStream_t)stream, grad, N, C, k, cin_pad, nullptr, nullptr,
                       nullptr, nullptr, 0, const_cast<float*>(dWp), ldw);
    BMT_CHECK_LAUNCH("bmt_conv_weight_grad");
    return BMT_OK;
}
