/*
 * bmt_hip.h -- C ABI of libbmt_hip.so: the MI355X (gfx950) kernels behind the bi-modal
 * transformer hot path of v-iashin/BMT.
 *
 * The reference has no FFI of its own (it is pure PyTorch); the boundary it offers is the
 * nn.Module surface of model/ and loss/ (SURVEY.md 8b).  Each entry point below replaces the ATen op
 * sequence of the reference lines it cites; bmt_amd/ops.py binds them with ctypes and
 * bmt_amd/model/ puts them back behind the reference's class names.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller; the library never allocates,
 *     frees or retains device memory (workspaces are passed in);
 *   - tensors are fp32 row-major unless stated; sizes/strides are in ELEMENTS;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), stateless
 *     and re-entrant; one device per call (the stream's device);
 *   - return value: 0 = ok, negative = BMT_E*; bmt_last_error() gives a thread-local
 *     message.  Nothing throws or exits across the ABI.
 *   - dropout: `rng` points at two device uint64 {seed, step}; a call site is identified by
 *     `site` (any 32-bit constant); element i of the dropped tensor keeps iff
 *     hash(seed, step, site, i) >= p.  p == 0 or rng == NULL disables it.  Reading the
 *     seed/step from device memory keeps captured hipGraphs replayable.
 */
#ifndef BMT_HIP_H
#define BMT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMT_OK 0
#define BMT_EINVAL (-1)   /* bad argument / unsupported shape */
#define BMT_EHIP (-2)     /* HIP runtime error (message has the hipError string) */
#define BMT_ENOENT (-3)   /* a feature file does not exist / cannot be opened (the reference catches FileNotFoundError) */
#define BMT_EALIGN (-4)   /* pointer or stride alignment requirement violated */

#define BMT_ABI_VERSION 12

int bmt_version(void);
const char* bmt_last_error(void);
/* device properties the host side sizes grids/workspaces with: CU count of the stream's device */
int bmt_device_cus(void);

/* ---------------------------------------------------------------- precision modes */
#define BMT_PREC_BF16 1    /* one bf16 MFMA pass, fp32 accumulate                      */
#define BMT_PREC_BF16X3 3  /* split-bf16: hi*hi + hi*lo + lo*hi, ~2^-16 relative error */
#define BMT_PREC_F16 4     /* one fp16 MFMA pass (planes hold fp16 values, 11 significand bits), fp32 accumulate: forward attention cores */
#define BMT_PREC_F16W2 5   /* two fp16 passes: A as ONE fp16 plane, B (the weight) as fp16 hi + lo: the weight is exact to ~2^-21, the
                            * activation rounds to 2^-11 -- what the per-site study (tests/study_precision_policy.py) showed to matter */

/* ---------------------------------------------------------------- GEMM epilogue flags */
#define BMT_EPI_BIAS 1u        /* + bias[n]                                                    */
#define BMT_EPI_RELU 2u        /* max(0, .)                                                    */
#define BMT_EPI_DROP_PRE 4u    /* dropout BEFORE relu  (bridge, proposal heads: blocks.py:152) */
#define BMT_EPI_DROP_POST 8u   /* dropout AFTER relu   (FFN hidden: blocks.py:170-171)         */
#define BMT_EPI_RESIDUAL 16u   /* + residual[m, n]     (ResidualConnection: blocks.py:136)     */
#define BMT_EPI_GATE 32u       /* *= (gate[m,n] != 0 ? gate_scale : 0)  (relu/dropout backward) */
#define BMT_EPI_ACCUM 64u      /* C += result (atomic; required when splitk > 1)               */

/*
 * bmt_gemm_bf16: C[M,N] = epilogue( alpha * sum_k A[m,k] * B[n,k] ) over PRE-CONVERTED 16-bit operand planes (both operands
 * reduction-contiguous).  Replaces nn.Linear forward (model/multihead_attention.py:66-68,84; model/blocks.py:151,168,172;
 * model/generators.py:18) and its two backward products (dX = dY.W, dW = dY^T.X).
 * Epilogue order: alpha -> +bias -> dropout_pre -> relu -> dropout_post -> gate -> +residual -> store/accumulate.
 *   BMT_PREC_BF16   A_hi.B_hi                           (bf16 planes, one pass: the backward products)
 *   BMT_PREC_BF16X3 A_hi.B_hi + A_hi.B_lo + A_lo.B_hi   (bf16 hi/lo planes)
 *   BMT_PREC_F16    A_hi.B_hi                           (fp16 planes)
 *   BMT_PREC_F16W2  A_hi.(B_hi + B_lo)                  (fp16 planes; A_lo unused)
 * Planes are [rows][ld] bf16 with the reduction extent zero-padded to Kpad (a multiple of 64); bmt_planes makes them from
 * fp32 tensors (straight and/or transposed), bmt_gemm_bf16 can emit them for its own result (C_hi/C_lo,
 * zero-padded up to min(round_up(N,64), ldp)).  gate is the bf16 hi plane of the saved forward output.
 */
typedef struct {
    const uint16_t *A_hi, *A_lo; int64_t lda;
    const uint16_t *B_hi, *B_lo; int64_t ldb;
    float* C; int64_t ldc;
    uint16_t *C_hi, *C_lo; int64_t ldp;
    int M, N, Kpad;
    float alpha;
    unsigned flags;
    const float* bias;
    const float* residual; int64_t ldr;
    const uint16_t* gate; int64_t ldg; float gate_scale;
    float drop_p; const uint64_t* rng; uint32_t site;
    int precision;
    int splitk;                 /* >1: split the reduction; 0: let the library choose (needs the workspace below); 1: never */
    /* two-pass split-K: every split stores its partial output into this fp32 scratch (splitk * ceil128(M) * ceil128(N) floats,
     * the split count is reduced to what fits) and a second kernel sums them in split order and runs the epilogue -- any
     * epilogue, deterministic, ACCUM without atomics.  Without a workspace splitk>1 needs BMT_EPI_ACCUM and a plain epilogue
     * (atomic accumulation).  One workspace serves every launch of one stream. */
    float* splitk_ws; int64_t splitk_ws_bytes;
    /* k-major operands (BMT_PREC_BF16 only): the operand's plane has the REDUCTION index as its row -- A_hi is [K][lda] with M
     * valid columns, B_hi is [K][ldb] with N valid columns (lda/ldb multiples of 8; rows [K, Kpad) are treated as zero, so the
     * plane needs no row padding).  dW = dY^T . X takes both gradient and activation planes as they are (a_kmajor = b_kmajor = 1),
     * dX = dY . W takes the weight plane [N][K] as it is (b_kmajor = 1): no transposed copies of anything.  K: true reduction
     * length, Kpad = K rounded up to a multiple of 64. */
    int a_kmajor, b_kmajor, K;
    /* implicit Conv1d over a HALO-PADDED activation plane X [rows][conv_cin] (bf16 planes; every sequence is followed by zero
     * rows, so that row r + tap never leaves its sequence's halo; conv_rows = rows reachable from the given base pointer):
     *   conv_mode 1 (forward, dX): A = X, reduction index = tap * conv_cin + c, Kpad = taps * conv_cin, B = weights [N][Kpad]
     *                C[r][n] = epilogue( sum_{tap,c} X[r + tap][c] * B[n][tap * conv_cin + c] )      (move A_hi by halo - pad rows)
     *   conv_mode 2 (dW): A = dY [K rows][M cols] k-major, B = X k-major; output column block j = tap * conv_cin + c
     *                C[m][tap * conv_cin + c] = sum_r dY[r][m] * X[r + tap][c]                       (N = taps * conv_cin) */
    int conv_mode, conv_cin, conv_rows;
    int conv_S, conv_halo;      /* conv_mode 1: output rows are compact (m = b * S + s); they read rows m + 2 b * halo + tap of the advanced plane */
    /* optional fp32 [N]: colsum[n] += sum over rows of the epilogue value (before bf16 rounding).  The column sums of a dX GEMM's
     * output are the bias gradient of the Linear below it; asking for them disables split-K for the launch. */
    float* colsum;
    /* alternative to C_lo: the second output plane holds fp16(c) (same layout as C_hi).  A forward activation is then stored as
     * {bf16(c) for the single-pass bf16 backward, fp16(c) for the fp16 forward consumer} -- the same bytes as hi + lo.
     * With C_hi == NULL the fp16 plane is the ONLY plane written (q / k / v under the fp16 attention policy, whose backward converts
     * them on load: bmt_attn_bwd_bf16_args.qkv_f16); excludes C_lo and colsum. */
    uint16_t* C_f16;
    /* ABI 7 -- PACKED ROWS.  The padded positions of a ragged batch carry no information (masked keys everywhere downstream, exactly zero
     * gradient): the encoder's row-wise tensors hold the valid rows only, compacted (bmt_pack_rows), and HOW MANY there are this step is
     * data.  rows_dev (optional, device int32): the launch is sized for M rows (k-major A -- a weight gradient: for K reduction rows)
     * and every kernel works on min(M | K, *rows_dev) of them, read when it runs -- the same launch (a hipGraph node) serves every batch.
     * Rows past the count are neither read (they may hold anything) nor written; column sums (colsum) leave them out. */
    const int* rows_dev;
    /* ABI 8, bmt_gemm_bf16_grouped only (a product whose OUTPUT lives in a packed row layout: the gradient of a packed encoder memory,
     * dX_b = dS_b^T . Q'_b of the decoder's cross-attention against the raw memory): the output rows start c_row_dev[0] rows below C and
     * only the first m_dev[0] of the M rows exist -- both read from device memory when the launch runs.  NULL = as before.
     * flags of a grouped problem: BMT_EPI_ACCUM (C += A B; several products may share a buffer), or -- with c_row_dev set and a reduction of at
     * most 96 x 64 rows, which the launch does not split -- 0: C = A B by plain stores, one writer per element, nothing to zero beforehand. */
    const int* c_row_dev;
    const int* m_dev;
    /* ABI 11 -- BLOCK PRODUCTS (the value product of the rank-form self-attention, O_h = O'_h W_v,h^T: the heads' d_in-wide attention outputs
     * sit side by side in A, every head multiplies its own row block of the weight): with a_blk_n > 0 the output columns
     * [j a_blk_n, (j + 1) a_blk_n) read A's columns [j a_blk_k, (j + 1) a_blk_k) -- a_blk_k = Kpad = 128, a_blk_n a multiple of 128, row-major
     * operands of a one- or two-plane fp16 / bf16 product (the reduction-of-128 kernel, whatever M).  0 = one product over all of A. */
    int a_blk_n, a_blk_k;
} bmt_gemm_bf16_args;
int bmt_gemm_bf16(const bmt_gemm_bf16_args* args, void* stream);
/* MANY independent single-pass GEMMs with both operands k-major and fp32 (accumulating) output in ONE launch -- the weight
 * gradients dW = dY^T . X of a whole training step (each `weight.grad` product of autograd's backward is too small to fill the
 * chip alone and would have to split its reduction).  args: host array of nprob argument structs (the same struct as
 * bmt_gemm_bf16; precision BMT_PREC_BF16, a_kmajor = b_kmajor = 1, fp32 C, no plane output, no split).  ws: device scratch of
 * bmt_gemm_bf16_grouped_ws_bytes(nprob) bytes, 16-byte aligned, that must stay untouched until the launch has executed; it is
 * filled by kernels that carry the descriptors in their arguments, so the call can be captured in a hipGraph. */
size_t bmt_gemm_bf16_grouped_ws_bytes(int nprob);

/* Products of at most this many outputs (M x N) with row-major operands in BMT_PREC_BF16X3 or BMT_PREC_BF16 and splitk 0 / 1 run on 32 x 32
 * tiles, one per workgroup, with the reduction split over the workgroup's waves (one launch, no split-K workspace pass): a decoder layer's own
 * nn.Linear products (model/decoders.py:60-95: 928 rows at configs[1]) and, with the weight's plane transposed, their dX.  A caller that
 * keeps transposed weight planes for dX asks here which products qualify; 0 = the library was built without the kernel. */
long long bmt_gemm_small_outputs(void);
int bmt_gemm_bf16_grouped(const bmt_gemm_bf16_args* args, int nprob, void* ws, size_t ws_bytes, void* stream);
/* ABI 9 -- the same launch as two calls: _tables writes the descriptor table and the per-XCD segment lists into ws (~25 launches of a few
 * microseconds, on `stream`: a stream of the caller's choice, e.g. one forked from the beginning of the step, so that they do not sit in
 * front of the product on its critical path) and returns what the product launch needs in launch[2]; _run launches the product on ITS
 * stream.  The caller orders the two (an event between the streams) and keeps ws untouched until the product has executed. */
int bmt_gemm_bf16_grouped_tables(const bmt_gemm_bf16_args* args, int nprob, void* ws, size_t ws_bytes, int* launch, void* stream);
int bmt_gemm_bf16_grouped_run(void* ws, int nprob, const int* launch, void* stream);
/* ABI 10 -- the bytes _tables leaves in ws (descriptor table | per-XCD segment lists | counts), written into HOST memory of
 * bmt_gemm_bf16_grouped_ws_bytes(nprob) bytes instead: no launch, no device access.  A captured step copies them into a device table of its
 * own once, at capture time, on a stream that is not capturing (bmt_copy_h2d_async) -- the addresses its launch works on never change --
 * and its graph holds _run alone: the ~11 table-writer launches of _tables execute where they were captured, in front of the product
 * (profiles/r06_o_replay_dispatches.csv), wherever their branch of the graph forks from. */
int bmt_gemm_bf16_grouped_image(const bmt_gemm_bf16_args* args, int nprob, void* host_image, size_t bytes, int* launch);

/* ABI 8 -- MANY small products of one shape in one launch on the 32 x 32 tile kernel (row-major operands; BMT_PREC_BF16, BMT_PREC_F16 or
 * BMT_PREC_BF16X3; epilogue: alpha, bias, dropout, relu, column sums; no residual / gate / accumulate).  `args` describes ONE product
 * (M, N, Kpad, planes, outputs, strides); product (o, i), 0 <= o < nb_outer, 0 <= i < nb_inner, reads and writes at the element offsets
 * o * x_off_o + i * x_off_i from those pointers.  What it replaces: the per-(sample, head) products of attention() when the heads' key /
 * value projections are reassociated onto the queries (model/multihead_attention.py:8-26, 62-84 with K = X W_k^T never formed):
 * S = (q W_k,h) X^T, O' = P X, and the per-head block products q_h W_k,h / O'_h W_v,h^T around them.
 *   b_rows_dev  optional, device int32 [nb_outer + 1]: the B operand of outer index o is rows b_rows_dev[o] .. b_rows_dev[o + 1] of B_hi
 *               (a packed encoder memory: bmt_pack_rows' `off`), N = their number (at most args->N); columns past it are not written;
 *   p2_*, ldp2  the second output plane (C_lo / C_f16) has its own row stride and offsets (ldp2 = 0: those of C_hi);
 *   bias_off_i  column offset of bias and colsum per inner index;  drop_off_*: dropout element index = offset + row * ldc + col. */
typedef struct {
    int nb_outer, nb_inner;
    int64_t a_off_o, a_off_i, b_off_o, b_off_i;
    const int* b_rows_dev;
    int64_t c_off_o, c_off_i, p_off_o, p_off_i, p2_off_o, p2_off_i, ldp2;
    int64_t bias_off_i, drop_off_o, drop_off_i;
    /* row r of the A operand / of an output lives at (r / x_div) * x_qs + (r % x_div) * ld instead of r * ld when x_div > 0: blocks of x_div
     * rows that are x_qs elements apart (the queries of one head inside a (sample, query) x (head, d) tensor; the 32-row blocks of a stack) */
    int a_div, c_div, p_div, p2_div;
    int64_t a_qs, c_qs, p_qs, p2_qs;
} bmt_gemm_batch;
int bmt_gemm_small_batched(const bmt_gemm_bf16_args* args, const bmt_gemm_batch* batch, void* stream);
/* ABI 11 -- attention whose keys and values are narrower than a head (csrc/rank_attn.hip; model/multihead_attention.py:62-84 with
 * d_model_K = d_model_V = d_a <= d_k / 2: the encoder's audio self-attention and the video stream's attention over the audio stream): k_h and
 * v_h are rank-d_a images of the key / value input x, so with queries projected from y (d_b columns; y = x for the self-attention)
 *     S_h = (y W'_h^T + c_h) x^T (+ terms constant along the keys),  W'_h = W_k,h^T W_q,h [d_a][d_b],  c_h = b_q,h W_k,h;   O_h = (P_h x) W_v,h^T + b_v,h
 * and the attention runs at width d_a against x itself (kv_shared).  The weight side, fp32 on the vector units from the fp32 parameters
 * (W_q [H dk][ldq >= d_b], W_k [H dk][ldk >= d_a]: head h = rows [h dk, (h + 1) dk)):
 *   bmt_rank_prep   W' [H d_a][d_b] (row h d_a + a = W'_h[a][.]) as bf16 / fp16 / fp16 lo planes of row stride ldp and / or fp32 (row stride
 *                   d_b), each optional; c [H d_a] (zeros without a query bias), optional; dWp_zero (optional): fp32 [H d_a][d_b] set to zero --
 *                   the accumulator of dW' for the pass to come;
 *   bmt_rank_chain  from dW' = dq'^T y (fp32 [H d_a][d_b]) and dc = column sums of dq' (or NULL):  dW_q,h += W_k,h dW'_h,
 *                   dW_k,h += W_q,h dW'_h^T + b_q,h^T dc_h,  db_q,h += W_k,h dc_h  (row strides ldgq / ldgk; each output optional; d_a = 128, d_b a
 *                   multiple of 128, dk of 64; 64 x 64 fp32 tiles, dW_k by atomics when d_b > 128). */
int bmt_rank_prep(const float* Wq, int64_t ldq, int d_b, const float* Wk, int64_t ldk, const float* bq, int H, int dk, int d_a, uint16_t* wp_bf16,
                  uint16_t* wp_f16, uint16_t* wp_f16_lo, int64_t ldp, float* wp_f32, float* c, float* dWp_zero, void* stream);
int bmt_rank_chain(const float* Wq, int64_t ldq, int d_b, const float* Wk, int64_t ldk, const float* bq, int H, int dk, int d_a, const float* dWp,
                   const float* dc, float* dWq, int64_t ldgq, float* dWk, int64_t ldgk, float* dbq, void* stream);
/* ... and the kernels between those products (csrc/raw_memory.hip).  `off` = bmt_pack_rows' offsets of the memory (int32, off[b] = first
 * packed row of sample b, off[B] = the row count), Skp = the padded key extent of the per-sample buffers (a multiple of 64, >= every length):
 *   bmt_memory_transposed  packed fp16 plane X [rows][ld] -> xt_f16[b][d][k] = fp16(X) and xtc_bf[b][d][k] = bf16(X - mean key of sample b)
 *                          (either may be NULL; the second takes a workspace for the samples' mean keys), zeros for k >= the sample's length;
 *   bmt_raw_softmax_fwd    S fp32 [B][H][32][Skp] (unscaled scores, rows t < Tq <= 32) -> P = softmax over the sample's keys of S * scale as
 *                          fp16 [B][H][32][Skp] and, optionally, bf16 at p_bf + b * p_bf_sb + h * p_bf_sh + t * Skp; rows t >= Tq and keys past
 *                          the length are zeros.  A sample without a valid key gives zeros (the reference gives NaN);
 *   bmt_raw_softmax_bwd    dS = P o (dP - rowsum(P o dP)) * scale as bf16 at ds_bf + b * ds_sb + h * ds_sh + t * Skp. */
int bmt_memory_transposed(const uint16_t* x_f16, int64_t ld, const int* off, int B, int D, int Skp, uint16_t* xt_f16, uint16_t* xtc_bf,
                          float* mean_ws /* B * D floats, ZEROED by the caller: needed with xtc_bf */, void* stream);
int bmt_raw_softmax_fwd(const float* S, const int* off, int B, int H, int Tq, int Skp, float scale, uint16_t* p_f16, uint16_t* p_bf, int64_t p_bf_sb,
                        int64_t p_bf_sh, void* stream);
int bmt_raw_softmax_bwd(const uint16_t* p_f16, const float* dP, const int* off, int B, int H, int Tq, int Skp, float scale, uint16_t* ds_bf, int64_t ds_sb,
                        int64_t ds_sh, void* stream);
/* ABI 12 -- the two products against the memory and the row operation between them as ONE launch per attention (workgroup = (sample, head);
 * the 32 x Skp score tile stays in LDS).  Replaces, with the same arithmetic and the same outputs, bmt_gemm_small_batched (S or dP) ->
 * bmt_raw_softmax_fwd / _bwd -> bmt_gemm_small_batched (O' or dQ') of model/multihead_attention.py:8-26 in the reassociated form:
 *   bmt_raw_attn_fwd  q_f16: row t of (b, h) at q_f16 + b * q_sb + h * q_sh + t * ldq (dm fp16 values = Q'_h, rows t < Tq); x_f16 the packed
 *                     memory plane, xt_f16 its per-sample transposed copy (bmt_memory_transposed) -> p_f16 [B][H][32][Skp], p_bf (optional) as
 *                     bmt_raw_softmax_fwd writes them, and O' = P X as split-bf16 planes o_hi / o_lo (o_lo optional) at row b * Tq + t,
 *                     column h * dm + d, row stride ldo;
 *   bmt_raw_attn_bwd  do_bf: row t of (b, h) at do_bf + b * do_sb + h * do_sh + t * lddo (dO'_h, bf16); x_bf the memory's bf16 plane, xtc_bf its
 *                     centred transposed copy, p_f16 the forward's P -> ds_bf (optional) as bmt_raw_softmax_bwd writes it and dQ' = dS (X - mean key)
 *                     as a bf16 plane dq_bf (row b * Tq + t, column h * dm + d, row stride lddq);
 *   bmt_raw_attn_ok   1 where the form applies: dm and Skp multiples of 64, Skp <= 1024, 64 (dm + 8) + 128 (Skp + 4) + 32 768 bytes of LDS <= 160 KB. */
int bmt_raw_attn_ok(int dm, int Skp);
/*   bmt_raw_attn_bwd_edges  bmt_raw_attn_bwd with the block products either side of it in the same launch: in front dO'_h = do_h W_v,h (do_bf [B Tq][ld_do]
 *                     bf16, this head's columns at h dk; wvT_bf: row d of dm holds W_v[h dk + k][d] at d * ld_wvT + h dk + k -- the transposed weight
 *                     group's plane) -> the A operand, and to bstack + b * b_sb + h * b_sh + t * dm (32 rows per (sample, head), rows t >= Tq zeros);
 *                     behind dq_h = dQ'_h W_k,h^T (wk_bf: W_k's plane, row h dk + n, dm contiguous) -> dq_bf [B Tq][ld_dq] bf16 at column h dk + n, its
 *                     column sums ADDED to dbq [H dk] (optional).  dqp_bf = dQ' as bmt_raw_attn_bwd writes it.  bmt_raw_attn_edges_ok: the form
 *                     applies (bmt_raw_attn_ok, dk a multiple of 64, 64 (dk + 8) <= 128 (Skp + 4)). */
int bmt_raw_attn_edges_ok(int dm, int Skp, int dk);
/*   bmt_raw_attn_fwd_edges  bmt_raw_attn_fwd with the block product in front of it in the same launch: Q'_h = q_h W_k,h (split-bf16: q_hi / q_lo [B Tq][ld_q],
 *                     this head's columns at h dk; wkT_hi / wkT_lo: row d of dm holds W_k[h dk + k][d] at d * ld_wkT + h dk + k) -> fp16 into the A operand
 *                     (no copy in memory) and bf16 to bstack + b * b_sb + h * b_sh + t * dm (32 rows, rows t >= Tq zeros).  bmt_raw_attn_fwd_edges_ok:
 *                     bmt_raw_attn_ok, dk a multiple of 64, and the LDS with two planes of q_h parked in the score tile's area <= 160 KB. */
int bmt_raw_attn_fwd_edges_ok(int dm, int Skp, int dk);
/*   bmt_raw_attn_fwd_proj   ... and the query projection in front of that: q_h = y W_q,h^T + b_q,h (split-bf16) from the sample's rows of y (y_hi / y_lo
 *                     [B Tq][ld_y], Kq = the padded width, a multiple of 64 with zeros past the true width; wq_hi / wq_lo: W_q's planes, row h dk + n,
 *                     Kq contiguous; bq [H dk] or NULL) -> q_h's two planes in LDS (the operand of Q'_h = q_h W_k,h) and its high plane to
 *                     q_hi_out [B Tq][ld_q] (what the backward reads).  linear_Q2d of model/multihead_attention.py:62 inside the launch. */
int bmt_raw_attn_fwd_proj_ok(int dm, int Skp, int dk, int Kq);
/*   bmt_raw_attn_bwd_proj   bmt_raw_attn_bwd_edges with the out-projection's dX in front of it: do_h = mask(dy W_o[:, h dk ...]) (bf16, one pass; dy_bf [B Tq][ld_dy],
 *                     Kq = its padded width; woT_bf = W_o^T's plane: row h dk + n, Kq contiguous) with the forward's attention-output dropout re-applied
 *                     (drop_p, the {seed, step} pair at rng, site; element index row * ld_do + column, ld_do = H dk) -> do_out [B Tq][ld_do] (the operand of
 *                     dW_v) and the A operand of dO'_h = do_h W_v,h; its column sums are ADDED to dbv [H dk] (optional).  The backward of
 *                     model/multihead_attention.py:84-86 (linear_d2Q's dX and the dropout of :22-23) inside the launch. */
int bmt_raw_attn_bwd_proj_ok(int dm, int Skp, int dk, int Kq);
int bmt_raw_attn_bwd_proj(const uint16_t* dy_bf, int64_t ld_dy, int Kq, const uint16_t* woT_bf, int64_t ld_woT, float drop_p, const uint64_t* rng, uint32_t site,
                          float* dbv, uint16_t* do_out, int64_t ld_do, const uint16_t* wvT_bf, int64_t ld_wvT, uint16_t* bstack, int64_t b_sb, int64_t b_sh,
                          const uint16_t* x_bf, int64_t ldx, const int* off, const uint16_t* xtc_bf, const uint16_t* p_f16, int B, int H, int Tq, int dm, int Skp,
                          int dk, float scale, uint16_t* ds_bf, int64_t ds_sb, int64_t ds_sh, uint16_t* dqp_bf, int64_t lddqp, const uint16_t* wk_bf, int64_t ld_wk,
                          uint16_t* dq_bf, int64_t ld_dq, float* dbq, void* stream);
int bmt_raw_attn_fwd_proj(const uint16_t* y_hi, const uint16_t* y_lo, int64_t ld_y, int Kq, const uint16_t* wq_hi, const uint16_t* wq_lo, int64_t ld_wq,
                          const float* bq, uint16_t* q_hi_out, int64_t ld_q, const uint16_t* wkT_hi, const uint16_t* wkT_lo, int64_t ld_wkT, uint16_t* bstack,
                          int64_t b_sb, int64_t b_sh, const uint16_t* x_f16, int64_t ldx, const int* off, const uint16_t* xt_f16, int B, int H, int Tq, int dm,
                          int Skp, int dk, float scale, uint16_t* p_f16, uint16_t* p_bf, int64_t p_bf_sb, int64_t p_bf_sh, uint16_t* o_hi, uint16_t* o_lo,
                          int64_t ldo, void* stream);
int bmt_raw_attn_fwd_edges(const uint16_t* q_hi, const uint16_t* q_lo, int64_t ld_q, const uint16_t* wkT_hi, const uint16_t* wkT_lo, int64_t ld_wkT,
                           uint16_t* bstack, int64_t b_sb, int64_t b_sh, const uint16_t* x_f16, int64_t ldx, const int* off, const uint16_t* xt_f16, int B, int H,
                           int Tq, int dm, int Skp, int dk, float scale, uint16_t* p_f16, uint16_t* p_bf, int64_t p_bf_sb, int64_t p_bf_sh, uint16_t* o_hi,
                           uint16_t* o_lo, int64_t ldo, void* stream);
int bmt_raw_attn_bwd_edges(const uint16_t* do_bf, int64_t ld_do, const uint16_t* wvT_bf, int64_t ld_wvT, uint16_t* bstack, int64_t b_sb, int64_t b_sh,
                           const uint16_t* x_bf, int64_t ldx, const int* off, const uint16_t* xtc_bf, const uint16_t* p_f16, int B, int H, int Tq, int dm,
                           int Skp, int dk, float scale, uint16_t* ds_bf, int64_t ds_sb, int64_t ds_sh, uint16_t* dqp_bf, int64_t lddqp,
                           const uint16_t* wk_bf, int64_t ld_wk, uint16_t* dq_bf, int64_t ld_dq, float* dbq, void* stream);
int bmt_raw_attn_fwd(const uint16_t* q_f16, int64_t q_sb, int64_t q_sh, int64_t ldq, const uint16_t* x_f16, int64_t ldx, const int* off,
                     const uint16_t* xt_f16, int B, int H, int Tq, int dm, int Skp, float scale, uint16_t* p_f16, uint16_t* p_bf, int64_t p_bf_sb,
                     int64_t p_bf_sh, uint16_t* o_hi, uint16_t* o_lo, int64_t ldo, void* stream);
int bmt_raw_attn_bwd(const uint16_t* do_bf, int64_t do_sb, int64_t do_sh, int64_t lddo, const uint16_t* x_bf, int64_t ldx, const int* off,
                     const uint16_t* xtc_bf, const uint16_t* p_f16, int B, int H, int Tq, int dm, int Skp, float scale, uint16_t* ds_bf, int64_t ds_sb,
                     int64_t ds_sh, uint16_t* dq_bf, int64_t lddq, void* stream);
/* x fp32 (B,S,C) -> halo-padded planes [B*(S+2*halo) + tail][ldp] (zero halo rows, zero columns >= C): hi = bf16(x) and, optionally,
 * lo = bf16(x - hi) or (lo_f16) fp16(x) */
int bmt_pad_planes(const float* x, int B, int S, int C, int halo, int tail, uint16_t* hi, uint16_t* lo, int lo_f16, int64_t ldp,
                   void* stream);
/* ABI 5 -- the proposal heads' backward without its fp32 intermediates:
 *   bmt_planes_gate       planes (+ column sums added into colsum[C], optional) of dz = (gate != 0) ? src * gate_scale : 0: the gradient
 *                         through relu (and a dropout in front of it) from the saved forward output, no dz tensor (bmt_gate -> bmt_planes);
 *   bmt_pad_planes_gate   the halo-padded bf16 plane of the same dz for a (B, S, C) gradient (bmt_pad_planes layout) + its column sums;
 *   bmt_conv_weight_planes  Conv1d weight [N][C][k] (contiguous) -> operand planes [N][k * cin_pad], tap-major, channels zero padded:
 *                         planes[n][tap * cin_pad + c] = W[n][c][tap] (any of hi / lo / fh / fl, row stride ldp);
 *   bmt_conv_weight_grad  the inverse for the weight gradient: grad[n][c][tap] += dWp[n * ldw + tap * cin_pad + c]. */
int bmt_planes_gate(const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp, float* colsum,
                    const float* gate, int64_t ldgate, float gate_scale, void* stream);
int bmt_pad_planes_gate(const float* dy, const float* y, float scale, int B, int S, int C, int halo, int tail, uint16_t* hi, int64_t ldp,
                        float* colsum, void* stream);
int bmt_conv_weight_planes(const float* W, int N, int C, int k, int cin_pad, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp,
                           void* stream);
int bmt_conv_weight_grad(const float* dWp, int64_t ldw, int N, int C, int k, int cin_pad, float* grad, void* stream);
/* fp32 [R][C] (row stride ld) -> bf16 planes: hi/lo [R][ldp] and/or transposed hiT/loT [C][ldpT]; any output may be NULL
 * (lo only with hi, loT only with hiT); padding up to the next multiple of 64 (bounded by the row stride) is zero-filled. */
int bmt_planes(const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh /* fp16(x), optional */,
               uint16_t* fl /* fp16(x - fh), optional */, int64_t ldp, uint16_t* hiT, uint16_t* loT, int64_t ldpT,
               float* colsum /* optional: colsum[c] += sum_r src[r][c] (atomic) */,
               const int* rows_dev /* optional, device: only rows < *rows_dev exist (packed rows, see bmt_gemm_bf16_args); not with hiT */, void* stream);
/* bmt_planes of dropout(src): the fp32 source is masked with the library's counter-based dropout (site, element index
 * r * C + c -- the mask bmt_dropout / the GEMM epilogues draw for a contiguous [R][C] tensor) before it is split; colsum sums
 * the masked values.  Backward of `x + dropout(sub)` fused into the gradient's operand conversion. */
int bmt_planes_dropout(const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl, int64_t ldp,
                       uint16_t* hiT, uint16_t* loT, int64_t ldpT, float* colsum, float drop_p, const uint64_t* rng, uint32_t site,
                       const int* rows_dev, void* stream);
/* the same for MANY tensors in one launch (all weights of a model after an optimizer step): the caller fills a host table of
 * bmt_planes_desc_bytes()-sized descriptors with bmt_planes_desc, uploads it, and passes the device pointer. */
int bmt_planes_desc_bytes(void);
int bmt_planes_desc(void* desc_out, const float* src, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, uint16_t* fh, uint16_t* fl,
                    int64_t ldp, uint16_t* hiT, uint16_t* loT, int64_t ldpT);
int bmt_planes_multi(const void* table_dev, int n_tensors, void* stream);
/* the same conversion as a flat tile list (one workgroup per 64-row tile of one tensor, no idle workgroups behind small tensors):
 * prefix_dev = int32 [n_tensors + 1] prefix sums of bmt_planes_desc_tiles over the table, total_tiles = its last entry */
int bmt_planes_desc_tiles(const void* desc_host);
int bmt_planes_multi_flat(const void* table_dev, const int* prefix_dev, int n_tensors, int total_tiles, void* stream);

/* bf16 [R][ld] -> transposed bf16 [C][ldT], zero padded up to min(round_up(R,64), ldT) */
int bmt_transpose_bf16(const uint16_t* src, int64_t ld, int R, int C, uint16_t* dst, int64_t ldT, void* stream);

/* column sums: out[n] (+)= sum_m X[m*ldx + n]   (bias gradients) */
int bmt_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate,
               const int* rows_dev /* optional, device (ABI 7): only rows < *rows_dev are summed */, void* stream);
/* many small reductions in ONE launch: out_i[c] += sum over r < rows_i of part_i[r * ld_i + c], c < D_i (fp32).  The second stage of the
 * LayerNorm dgamma / dbeta partials (bmt_layernorm_bwd_partial) and of the attention backward's per-tile bias sums (defer_bias): the
 * host side queues them over a backward pass and issues them together.  `items` is read on the HOST during the call (and passed to the
 * kernel by value, BMT_COLSUM_MAX_ITEMS per launch): capturable in a hipGraph. */
#define BMT_COLSUM_MAX_ITEMS 96
typedef struct {
    const float* part;
    float* out;
    int rows;
    int D;
    int64_t ld;
} bmt_colsum_item;
int bmt_colsum_multi(const bmt_colsum_item* items, int n, void* stream);
/* many small fp32 copies (dst_i[0 .. n_i) = src_i[0 .. n_i)) in one launch, items by value as above */
typedef struct {
    const float* src;
    float* dst;
    int64_t n;
} bmt_copy_item;
int bmt_copy_multi(const bmt_copy_item* items, int n, void* stream);

/*
 * bmt_attn_fwd: O = dropout( softmax(Q K^T * scale, masked) V )   per (batch, head)
 * Replaces attention() model/multihead_attention.py:8-26 together with the head split/merge
 * views of :71-73,82 (heads are addressed in place: head h is columns [h*dk, (h+1)*dk)).
 *   Q:[B,Sq,*] K,V:[B,Sk,*] O:[B,Sq,*]; row strides ldq/ldk/ldv/ldo, batch strides bsq/...
 *   mask: uint8/bool, element (b,q,k) at mask[b*mask_bs + q*mask_qs + k]; mask_qs == 0 for a
 *         key-padding mask (B,1,Sk); NULL = no mask.  mask==0 -> score = -inf (scale applied first).
 *   lse: [B,H,Sq] log-sum-exp of the scaled, masked scores (saved for backward).
 *   A fully masked row gives NaN, as the reference does.
 *   dk in {32, 64, 128, 256}.
 */
typedef struct {
    const float* Q; const float* K; const float* V; float* O; float* lse;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    const uint8_t* mask; int64_t mask_bs, mask_qs;
    int B, H, Sq, Sk, dk;
    float scale;
    float drop_p; const uint64_t* rng; uint32_t site; /* dropout on O, indexed b*bso + q*ldo + col */
    int precision;
} bmt_attn_fwd_args;
int bmt_attn_fwd(const bmt_attn_fwd_args* args, void* stream);

/*
 * bmt_attn_bwd: gradients of the softmax-attention core (bf16 MFMA, probabilities recomputed
 * from lse).  dO is the gradient w.r.t. the PRE-dropout attention output (the out-projection's
 * dX epilogue has already applied the dropout mask); O is the saved POST-dropout output and
 * delta is rebuilt as (1-p) * rowsum(dO*O).   delta_ws: [B,H,Sq] fp32 workspace.
 */
typedef struct {
    const float* Q; const float* K; const float* V; const float* O; const float* dO; const float* lse;
    float* dQ; float* dK; float* dV; float* delta_ws;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;   /* dO shares O's strides; dQ/dK/dV share Q/K/V's */
    const uint8_t* mask; int64_t mask_bs, mask_qs;
    int B, H, Sq, Sk, dk;
    float scale, drop_p;
} bmt_attn_bwd_args;
int bmt_attn_bwd(const bmt_attn_bwd_args* args, void* stream);

/*
 * bmt_attn_fwd_bf16 / bmt_attn_bwd_bf16: the same attention core over PRE-SPLIT bf16 operand planes (hi = bf16(x),
 * lo = bf16(x - hi)) as written by bmt_gemm_bf16's C_hi / C_lo epilogue outputs: no conversion in the K/V loop, half the
 * staged bytes, per-tile mask classification (fully masked key tiles are skipped, fully valid ones run unmasked).
 * Plane strides are in bf16 elements and must be multiples of 8; O / dO / dQ are fp32 [B,Sq,H*dk] (ldo, bso),
 * dK / dV fp32 with (dkv_ld, dkv_bs).  Backward reads hi planes only (single-pass bf16); dOh_ws: bf16 workspace the size of O.
 *
 * Outputs that feed GEMMs can be produced directly as operand planes, so that no conversion pass runs over them:
 *   forward : Oh / Ol (hi, lo planes of the post-dropout output; row b*Sq+q at b*bsop + q*ldop).  O (fp32) may then be NULL;
 *             ldo / bso must still describe the logical fp32 layout (they index the dropout mask).
 *   backward: the saved output comes back as Oh / Ol when O is NULL; with dO == NULL, dOh_ws is an INPUT: the gradient of the
 *             output already as a bf16 plane (ldo / bso strides), e.g. written by the out-projection's dX GEMM epilogue.  Each gradient has up to four forms, all optional
 *             but at least one of {fp32, hi plane}: fp32 (dQ|dK|dV), bf16 plane (dQh|dKh|dVh: row b*S+s at b*g?_bs + s*g?_ld),
 *             transposed bf16 plane (dQT|dKT|dVT: [H*dk][g?T_ld], column b*S+s; the caller zero-fills columns past B*S) and
 *             bias sums (dbq|dbk|dbv: fp32 [H*dk] += sum over (b,s), atomics; summed from the bf16-rounded values).
 */
typedef struct {
    const uint16_t *Qh, *Ql, *Kh, *Kl, *Vh, *Vl;      /* lo planes may be NULL for BMT_PREC_BF16 */
    float* O; float* lse;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    const uint8_t* mask; int64_t mask_bs, mask_qs;
    int B, H, Sq, Sk, dk;
    float scale;
    float drop_p; const uint64_t* rng; uint32_t site;
    int precision;
    uint16_t *Oh, *Ol; int64_t ldop, bsop;            /* optional plane outputs */
    uint16_t* Of;                                     /* alternative to Ol: fp16(o) plane (consumer: an fp16-policy out-projection) */
    /* ABI 7 -- PACKED ROWS (bmt_pack_rows): with q_off (int32 [B + 1], device) the query-side planes (Q, O / Oh / Ol / Of) hold the valid
     * rows of the batch compacted -- sample b's queries are rows q_off[b] .. q_off[b + 1] - 1, the batch strides are ignored; with k_off the
     * same for K / V (mask must be NULL: every packed key is valid).  Sq / Sk stay the PADDED lengths (they size the launch and index lse,
     * which keeps its [B, H, Sq] layout by position within the sample).  Taken by the one-pass d_k >= 128 kernels. */
    const int *q_off, *k_off;
    /* ABI 10 (optional, device int32 [B], a permutation of 0 .. B - 1; packed rows only): the launch's work items (batch, head, tile) are
     * numbered with sample b_order[i] in place of sample i -- which sample a workgroup (and, through the XCD-contiguous numbering, an XCD)
     * works on, not what it computes (bmt_pack_rows_ordered builds a length-balanced one). */
    const int* b_order;
    /* ABI 10: 1 = K and V are ONE plane of width dk each, shared by the H heads (ldk / ldv rows of dk columns; no per-head column offset) --
     * attention against an un-projected input whose width is the head dimension: S_h = q'_h X^T, O'_h = P_h X.  Q, O and every gradient
     * keep their per-head column blocks (the key-side gradients come out per head: the caller sums them).  d_k >= 128 plane kernels. */
    int kv_shared;
} bmt_attn_fwd_bf16_args;
int bmt_attn_fwd_bf16(const bmt_attn_fwd_bf16_args* args, void* stream);

typedef struct {
    const uint16_t *Qh, *Kh, *Vh;
    const float *O, *dO, *lse;
    float *dQ, *dK, *dV, *delta_ws;
    uint16_t* dOh_ws;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, dkv_ld, dkv_bs;
    const uint8_t* mask; int64_t mask_bs, mask_qs;
    int B, H, Sq, Sk, dk;
    float scale, drop_p;
    const uint16_t *Oh, *Ol; int64_t ldop, bsop;      /* saved forward output as planes (when O == NULL) */
    uint16_t *dQh, *dKh, *dVh; int64_t gq_ld, gq_bs, gkv_ld, gkv_bs;
    uint16_t *dQT, *dKT, *dVT; int64_t gqT_ld, gkvT_ld;
    float *dbq, *dbk, *dbv;
    const uint16_t* Of;                               /* saved forward output as an fp16 plane (delta = rowsum(dO * O) reads it when set) */
    const float* kmean;                               /* optional fp32 [B][H*dk]: mean key over the valid keys (bmt_attn_kmean).  With it dQ is
                                                         corrected by (row sum of the bf16-rounded dS) x mean key: the rounding residue of dS
                                                         times the keys' common component, 10-25 % of |dQ| under near-uniform attention */
    int qkv_f16;                                      /* Qh / Kh / Vh are the FORWARD's fp16 planes (d_k >= 128): converted to bf16 while staging; the
                                                         projections then write 2 instead of 4 bytes per element of q, k and v */
    /* ABI 4 -- workspaces of the SPLIT backward (all four, or bias_ws alone, or none; sizes from bmt_attn_bwd_split_ws / bmt_attn_bwd_bias_ws).  With them, fp16 q / k / v planes
     * (qkv_f16), d_k 128 / 256, a key-padding mask (or none) and Sq >= 64 the backward runs as {dQ kernel that leaves P and dS in
     * P_ws / dS_ws and a scaled bf16 copy of q in Qb_ws} -> {dK / dV as two plain products over them} -> {bias sums from per-tile
     * partials in bias_ws}; S = Q K^T and dP = dO V^T are computed once instead of twice.  Contents are scratch: the caller may hand the
     * same buffers to every call on a stream. */
    uint16_t *P_ws, *dS_ws, *Qb_ws;
    float* bias_ws;
    int defer_bias;                                   /* with bias_ws: leave the per-tile sums there (rows b * tiles + tile of [D] floats: dbq's B * ceil(Sq / 128)
                                                         rows, then dbk's and dbv's B * ceil(Sk / 128) each) and skip the finishing launch: the caller
                                                         adds them up itself (bmt_colsum_multi, together with other reductions) */
    const int *q_off, *k_off;                         /* ABI 7: packed rows as in bmt_attn_fwd_bf16_args -- q side: Q, O planes, dOh_ws, dQh; k side: K, V, dKh, dVh.  lse, delta_ws,
                                                         the split backward's workspaces and the per-tile bias partials keep their padded layouts */
    /* ABI 9 -- the RECOMPUTE form of the split backward: rc_ws (int32, *n_rc elements from bmt_attn_bwd_rc_ws) + bias_ws, P_ws = dS_ws = Qb_ws =
     * NULL.  {dQ kernel: S, dP', dQ -- emits nothing but delta (delta_ws), one word of live-query bits and the largest |dO| per (batch, head,
     * 128-query tile) into rc_ws} -> {key-side kernel: a wave keeps 32 keys' K and V rows in registers, streams the q / dO rows through LDS,
     * rebuilds S, P, dP and dS (one power-of-two scale per (batch, head) on the fp16 dS) and accumulates dK and dV} -> bias sums.  7 products of
     * Sq x Sk x d_k instead of 5, and none of the 2 x Sq x Sk x 2 bytes per (batch, head) written and read back (model/multihead_attention.py:8-26's
     * autograd).  Same eligibility as the emitting form plus Sq <= 2048; an ineligible problem with rc_ws set is an error. */
    int* rc_ws;
    const int* b_order;                               /* ABI 10: as in bmt_attn_fwd_bf16_args */
    int kv_shared;                                    /* ABI 10: as in bmt_attn_fwd_bf16_args (dK / dV per head in their column blocks) */
} bmt_attn_bwd_bf16_args;
int bmt_attn_bwd_bf16(const bmt_attn_bwd_bf16_args* args, void* stream);
/* element counts of the split backward's workspaces for a problem: P_ws and dS_ws (bf16) take *n_pds each, Qb_ws (bf16) *n_qb, bias_ws
 * (fp32) *n_bias.  (ABI 5: *n_qb includes, behind the scaled copy of q, one int per (batch, head, 128-query tile) of LIVE-QUERY bits -- which
 * 32-query groups have a non-zero dO at all.  The dQ kernel finds that out from the rows it holds anyway; a tile without a live row skips its
 * key loop, the dK / dV kernel's query loop ends at the last live stage.  Padded positions get exactly zero gradient in the encoder: a
 * quarter of the query rows of configs[1]'s ragged batches.  Data-driven -- nothing is assumed about the caller's padding.)  Returns BMT_EINVAL (and zeros) for a problem the split form does not take (d_k < 128, Sq < 64, sizes past 2^31 bytes per
 * (batch, head) block): the caller then passes NULL workspaces and the two-kernel form runs. */
int bmt_attn_bwd_split_ws(int B, int H, int Sq, int Sk, int dk, int64_t* n_pds, int64_t* n_qb, int64_t* n_bias);
/* ABI 9: element counts of the recompute form's workspaces: rc_ws (int32) *n_rc, bias_ws (fp32) *n_bias.  BMT_EINVAL (and zeros) for a problem
 * it does not take (d_k not 128 / 256, Sq < 64 or > 2048, Sk > 8192): the caller then uses the emitting form or the two-kernel form. */
int bmt_attn_bwd_rc_ws(int B, int H, int Sq, int Sk, int dk, int64_t* n_rc, int64_t* n_bias);
/* bias_ws alone (P_ws = dS_ws = Qb_ws = NULL) is taken by every d_k >= 128 backward: the tiles' column sums are stored per tile and added
 * up by a finishing launch instead of ~900 workgroups adding into the same H * d_k floats.  Element count (0 for d_k < 128: pass NULL): */
int64_t bmt_attn_bwd_bias_ws(int B, int H, int Sq, int Sk, int dk);
/* out[b][c] = mean over the valid keys k of K[b][k][c] (bf16 plane, row stride ldk, batch stride bsk; mask: key-padding bytes [B][Sk]
   when mask_qs == 0, otherwise -- no mask or one row per query -- every key counts).  From Sk = 256 on the mean is taken over every 8th key:
   the vector is a shift (any value leaves dS . K unchanged in exact arithmetic since rows of dS sum to zero), what matters is that it
   carries the keys' common component.  No counterpart in the reference: numerical aid of the 16-bit backward
   (model/multihead_attention.py:8-26 is exact in fp32). */
int bmt_attn_kmean(const uint16_t* Kh, int64_t ldk, int64_t bsk, const uint8_t* mask, int64_t mask_bs, int64_t mask_qs, int B, int Sk, int D,
                   float* out, int k_f16 /* the plane holds fp16 */,
                   const int* k_off /* optional (ABI 7): packed rows -- sample b's keys are rows k_off[b] .. k_off[b + 1] - 1, bsk and mask ignored */, void* stream);

/* ---------------------------------------------------------------- LayerNorm (model/blocks.py:127,131,143,150) */
/* y = (x-mean)/sqrt(var+eps)*gamma+beta over the last dim D (biased variance).  mean/rstd: [rows] saved for backward. */
int bmt_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                      float* mean, float* rstd, int rows, int D, float eps, void* stream);
/* bmt_layernorm_fwd that also (or only: y == NULL) writes the bf16 operand planes of y: hi = bf16(y), lo = bf16(y - hi)
 * (lo may be NULL), row stride ldp >= D, columns D .. min(pad64(D), ldp) zero filled -- the next GEMM / attention kernel reads
 * them directly, no separate conversion pass. */
int bmt_layernorm_fwd_planes(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                             float* mean, float* rstd, uint16_t* hi, uint16_t* lo, int lo_f16 /* lo receives fp16(y) instead */,
                             int64_t ldp, int rows, int D, float eps,
                             const int* rows_dev /* optional, device: rows = min(rows, *rows_dev) when the kernel runs (packed rows, ABI 7) */, void* stream);
/* dx (+)= LN backward; dgamma/dbeta += column reductions (accumulate into pre-zeroed or live grads).
 * dx[i] = (accumulate_dx ? dx[i] : 0) + ...
 * partial_ws: NULL -> one atomic per column per workgroup; else fp32 [bmt_layernorm_bwd_blocks(rows)][2][D] scratch for a
 * two-stage reduction (store per-workgroup partials, then sum them 64 at a time: one atomic per column per 64 workgroups). */
int bmt_layernorm_bwd_blocks(int rows);
/* bmt_layernorm_bwd_add WITHOUT its second stage: the dgamma / dbeta column partials of the bmt_layernorm_bwd_blocks(rows) workgroups stay
 * in partial_ws ([blocks][2 D]: dgamma | dbeta) for the caller to reduce (bmt_colsum_multi).  Returns 1 -- and does nothing -- where the
 * vector kernel does not apply (D > 2048, unaligned rows): the caller then uses bmt_layernorm_bwd_add. */
int bmt_layernorm_bwd_partial(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                              float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, float* partial_ws, int rows, int D,
                              const int* rows_dev /* optional (ABI 7): as in bmt_layernorm_fwd_planes; workgroups past the count leave zero partials */, void* stream);
/* bmt_layernorm_bwd_partial with a SECOND addend: dx = dx_add + dx_add2 + LN backward.  The input of a ResidualConnection's LayerNorm that
 * is also the key / value input of the other modality's cross-attention (model/encoders.py:63-79) has three consumers; their gradients meet
 * in this kernel instead of in an add kernel of autograd's (ABI 5).  Returns 1 where the vector kernel does not apply. */
int bmt_layernorm_bwd_partial2(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                               float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, const float* dx_add2, int64_t ldadd2,
                               float* partial_ws, int rows, int D, const int* rows_dev, void* stream);
/* ... and with the NEXT consumer's operand conversion folded in (ABI 5): besides dx the kernel writes gp_hi [rows][gp_ld] = bf16 of
 * dropout(dx) under the mask of (drop_p, rng, site) over the contiguous [rows][D] index space -- the upstream-gradient operand of the
 * previous sublayer's last GEMM backward (x_out = x + dropout(sublayer(LN x))) -- and leaves that plane's column partials (the GEMM's bias
 * gradient) as a THIRD block of partial_ws, which is [blocks][3 D] here: dgamma | dbeta | column sums.  D a multiple of 4 (ABI 6: any such width; gp_ld >= round_up(D, 64), the pad columns are written as zeros).  Returns 1
 * where the vector kernel does not apply. */
int bmt_layernorm_bwd_emit(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                           float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, const float* dx_add2, int64_t ldadd2, float* partial_ws,
                           uint16_t* gp_hi, int64_t gp_ld, float drop_p, const uint64_t* rng, uint32_t site, int rows, int D, const int* rows_dev,
                           void* stream);
int bmt_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, float* dx, int64_t lddx, int accumulate_dx,
                      float* dgamma, float* dbeta, float* partial_ws, int rows, int D, void* stream);
/* dx = dx_add + LN backward (dx_add may be NULL or alias dx): the residual stream's gradient joins here, so that the sum
 * autograd would form with a separate add kernel is produced by the LayerNorm backward itself. */
int bmt_layernorm_bwd_add(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                          const float* rstd, float* dx, int64_t lddx, const float* dx_add, int64_t ldadd, float* dgamma,
                          float* dbeta, float* partial_ws, int rows, int D, const int* rows_dev, void* stream);

/* ---------------------------------------------------------------- input prep / elementwise (K8) */
/* out[b,s,:] = dropout( (a[b,s,:] (+ b2[b,s,:])) * in_scale + PE[s,:] )      model/captioning_module.py:165,174-176, blocks.py:101-107
 * PE is the reference's table (sin on even j, cos on odd j, exponent j/D for both), passed in as fp32 [>=S, D]. */
int bmt_prep_features(const float* a, const float* b2, const float* pe, float* out, int B, int S, int D,
                      float drop_p, const uint64_t* rng, uint32_t site, void* stream);
/* ABI 7 -- PACKED ROWS (epoch_loops/captioning_epoch_loops.py:105-112 derives the masks; model/multihead_attention.py:17 is the only place the
 * reference reads a padded position: as a masked key).  bmt_pack_rows turns a key-padding mask (uint8 [B][S], batch stride mask_bs, 1 = valid)
 * into the layout of the valid rows: off[b] (int32 [2 B + 2]: B + 1 offsets, then scratch) = number of valid positions of the samples before b -- off[B] = their total, the
 * `rows_dev` of every row-wise kernel --, row_map[r] (int32 [B * S]) = b * S + t of packed row r (in (b, t) order; derived from the mask,
 * so a hole inside a sequence is as good as a padded tail).  bmt_prep_features_packed is bmt_prep_features writing packed rows:
 * out[r] = dropout(a[row_map[r]] (+ b2[row_map[r]]) + PE[row_map[r] % S]) for r < off[B]; the dropout mask is indexed by the packed element. */
int bmt_pack_rows(const uint8_t* mask, int64_t mask_bs, int B, int S, int* off, int* row_map, void* stream);
/* ABI 10 -- ... and `order` (optional, int32 [B], B <= 256): a permutation of the samples that balances the attention kernels' work over the
 * eight XCDs -- ranked by valid length, dealt to the eight contiguous ranges of the sample-major work order in serpentine order, each range
 * longest first (bmt_attn_fwd_bf16_args.b_order / bmt_attn_bwd_bf16_args.b_order).  Results do not depend on it. */
int bmt_pack_rows_ordered(const uint8_t* mask, int64_t mask_bs, int B, int S, int* off, int* row_map, int* order, void* stream);
int bmt_prep_features_packed(const float* a, const float* b2, const float* pe, float* out, int B, int S, int D, float drop_p,
                             const uint64_t* rng, uint32_t site, const int* row_map, const int* rows_dev, void* stream);
/* out[b,s,:] = dropout( W[ids[b,s],:] * emb_scale + PE[s,:] )                model/blocks.py:42-46 + pos enc */
int bmt_prep_embed(const int64_t* ids, const float* W, const float* pe, float* out, int B, int S, int D, int V,
                   float emb_scale, float drop_p, const uint64_t* rng, uint32_t site, void* stream);
/* dW[ids[b,s],:] += dout[b,s,:] * keep(b,s,:) * emb_scale   (trainable-embedding path) */
int bmt_prep_embed_bwd(const int64_t* ids, const float* dout, float* dW, int B, int S, int D, int V,
                       float emb_scale, float drop_p, const uint64_t* rng, uint32_t site, void* stream);
/* masks, bit-exact: model/masking.py:14-21 + epoch_loops/captioning_epoch_loops.py:105-112
 *   src_mask[b,0,s] = feat[b,s,0] != pad  (feat row stride ld, batch stride bs)
 *   trg_mask[b,i,j] = (trg[b,j] != pad_idx) & (j <= i)                                   */
int bmt_mask_from_features(const float* feat, int64_t bs, int64_t ld, float pad, uint8_t* out, int B, int S, void* stream);
int bmt_mask_from_tokens(const int64_t* trg, int64_t pad_idx, uint8_t* src_mask, uint8_t* trg_mask, int B, int S, void* stream);
/* y = dropout(x) and its mask re-application dy*keep/(1-p) (standalone nn.Dropout sites, residual backward) */
int bmt_dropout(const float* x, float* y, int64_t n, float drop_p, const uint64_t* rng, uint32_t site, void* stream);
/* out = dy * (y != 0 ? scale : 0): backward of relu/dropout fused epilogues, read off the saved output y */
int bmt_gate(const float* dy, const float* y, float scale, float* out, int64_t n, void* stream);
/* out = res + dropout(x)   (ResidualConnection.forward model/blocks.py:134-136) */
int bmt_dropout_add(const float* x, const float* res, float* out, int64_t n, float drop_p, const uint64_t* rng,
                    uint32_t site, void* stream);
/* out = a + b (n elements) ; out may alias a */
int bmt_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* ABI 5: out[r] = a[r] | b[r] (concatenation along the last dimension, rows with strides lda / ldb / ldo) and its inverse: the decoder
 * layer's torch.cat([Ca, Cv], -1) in front of the bridge (model/decoders.py:83) and the two halves of that tensor's gradient */
int bmt_cat2(const float* a, int64_t lda, int Da, const float* b, int64_t ldb, int Db, float* out, int64_t ldo, int rows, void* stream);
int bmt_split2(const float* in, int64_t ldi, float* a, int64_t lda, int Da, float* b, int64_t ldb, int Db, int rows, void* stream);
/* rng[1] += 1 (advance the dropout step counter on device; graph-capturable) */
int bmt_rng_advance(uint64_t* rng, void* stream);
/* out[i0][i1][i2] (contiguous) (+)= in[i0*s0 + i1*s1 + i2*s2]  -- Conv1d weight re-layout
 * ([Dout][Din][k] state_dict layout <-> the tap-major layouts the implicit-convolution GEMM consumes) */
int bmt_copy3d(const float* in, int64_t s0, int64_t s1, int64_t s2, float* out, int n0, int n1, int n2, int accumulate,
               void* stream);

/* ---------------------------------------------------------------- generator + loss (K7) */
/* in-place row log_softmax over V: model/generators.py:19 */
int bmt_log_softmax_fwd(float* x, int64_t ldx, int rows, int V, void* stream);
/* dlogits = dlogp - exp(logp) * rowsum(dlogp)   (dlogp may alias dlogits) */
int bmt_log_softmax_bwd(const float* logp, int64_t ldp, const float* dlogp, int64_t ldd, float* dlogits, int64_t ldo,
                        int rows, int V, void* stream);
/* LabelSmoothing.forward loss/label_smoothing.py:12-32: sum-KL between the smoothed target
 * distribution and pred (log-probs), incl. the quirk that pad rows are zeroed only when the
 * SUM of their flat row indices is > 0.   loss: 1 float (overwritten).  row_ws: >= rows + 1 floats. */
int bmt_ls_kl_fwd(const float* pred, int64_t ldp, const int64_t* target, float* loss, float* row_ws,
                  int rows, int V, float smoothing, int64_t pad_idx, void* stream);
/* dpred = -dist * (*gscale_dev)  (gscale_dev: device scalar = upstream grad; row_ws: the workspace the forward
 * filled -- its trailing int carries the pad-row decision) */
int bmt_ls_kl_bwd(const int64_t* target, float* dpred, int64_t ldp, const float* gscale_dev, const float* row_ws,
                  int rows, int V, float smoothing, int64_t pad_idx, void* stream);
/* K7 in one pass each way (ABI 5; model/generators.py:18-19 + loss/label_smoothing.py:12-32 as ONE forward and ONE backward kernel over
 * the (B*Tc, V) tensor):
 *   bmt_log_softmax_fwd_stats  log_softmax in place with the row in registers (one HBM read, one write) and rowsum[r] = sum_c logp[r][c]
 *                              (rowsum may be NULL);
 *   bmt_ls_kl_fwd_stats        LabelSmoothing.forward from those row sums and two gathers per row -- one launch of one workgroup; row_ws as
 *                              bmt_ls_kl_fwd leaves it ([rows] row losses + the int pad-row flag);
 *   bmt_gen_lskl_bwd           d(loss * *gscale_dev) / d(logits) from the saved log-probabilities: dlogits = g (softmax * rowsum(dist) - dist),
 *                              written as the bf16 operand plane hi [rows][ldh] (ldh >= round_up(V, 64), pad columns zeroed) that the
 *                              generator's dX / dW products read, its column sums (the generator's bias gradient) ADDED into colsum[V]
 *                              (optional).  Replaces bmt_ls_kl_bwd -> bmt_log_softmax_bwd -> bmt_planes. */
int bmt_log_softmax_fwd_stats(float* x, int64_t ldx, int rows, int V, float* rowsum, void* stream);
int bmt_ls_kl_fwd_stats(const float* pred, int64_t ldp, const int64_t* target, const float* rowsum, float* loss, float* row_ws, int rows, int V,
                        float smoothing, int64_t pad_idx, void* stream);
int bmt_gen_lskl_bwd(const float* logp, int64_t ldp, const int64_t* target, const float* row_ws, const float* gscale_dev, int rows, int V,
                     float smoothing, int64_t pad_idx, uint16_t* hi, int64_t ldh, float* colsum, void* stream);

/* ---------------------------------------------------------------- step protocol (ABI 5): what training_loop does between the kernels
 * (epoch_loops/captioning_epoch_loops.py:128-135), as library launches instead of framework fills / copies / reductions */
/* p[0 .. nbytes) = 0 (p 16-byte aligned): optimizer.zero_grad() over the flat gradient arena in one launch */
int bmt_zero(void* p, int64_t nbytes, void* stream);
/* ABI 10: nbytes from (pinned) host memory to the device on `stream` (hipMemcpyAsync) */
int bmt_copy_h2d_async(void* dst, const void* src_host, int64_t nbytes, void* stream);
/* x = caption_idx[:, :-1], y = caption_idx[:, 1:] as contiguous int64 [B][T1 - 1] and n_tokens[0] = (y != pad_idx).sum(); caption_idx [B][T1]
 * with row stride ld */
int bmt_caption_shift(const int64_t* caption_idx, int64_t ld, int B, int T1, int64_t pad_idx, int64_t* x, int64_t* y, int64_t* n_tokens,
                      void* stream);
/* loss[0] = kl[0] / n_tokens[0], grad_scale[0] = 1 / n_tokens[0] (either output may be NULL): loss = criterion(pred, y) / n_tokens */
int bmt_loss_finish(const float* kl, const int64_t* n_tokens, float* loss, float* grad_scale, void* stream);

/* ---------------------------------------------------------------- optimizer (K11) */
/* torch.optim.Adam semantics (scripts/train_captioning_module.py:46-48) over n_tensors tensors.
 * ptrs: device array of 4*n_tensors pointers laid out [p_0..][g_0..][m_0..][v_0..]; sizes: device int64[n].
 * step_dev: device int64 step counter, incremented by the kernel launch itself (graph-replayable).
 * grad_scale_dev: optional device float multiplied into every gradient (e.g. clip coefficient or 1/world); NULL = 1. */
int bmt_adam_step(void* const* ptrs, const int64_t* sizes, int n_tensors, int64_t max_size, int64_t* step_dev,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  const float* grad_scale_dev, void* stream);
/* sum of squares of n_tensors tensors -> out[0] (overwritten); clip coefficient helper:
 * coef[0] = min(1, max_norm / (sqrt(out[0]) + 1e-6))   (torch.nn.utils.clip_grad_norm_) */
int bmt_grad_sqnorm(void* const* ptrs, const int64_t* sizes, int n_tensors, int64_t max_size, float* out,
                    float max_norm, float* coef, void* stream);

/* every element of every tensor *= coef_dev[0] (the in-place scaling step of clip_grad_norm_) */
int bmt_scale_tensors(void* const* ptrs, const int64_t* sizes, int n_tensors, int64_t max_size, const float* coef_dev,
                      void* stream);

/* ---------------------------------------------------------------- proposal generator (K9, K10) */
/* The Conv1d heads (model/proposal_generator.py:29,41-45) run on bmt_gemm_bf16's implicit-convolution modes (conv_mode 1 / 2
 * over halo-padded activation planes, bmt_pad_planes); the (B,S,D) <-> (B,D,S) permutes at :41,45 vanish in the addressing. */
/* make_targets model/proposal_generator.py:389-448 (bit-exact masks / targets):
 * targets [n,4] f32 (batch idx, center s, length s, meta); anchors [A] already divided by stride.
 * obj/noobj: uint8 [B,A,G] (caller pre-fills obj=0, noobj=1, tx=tw=0 via bmt_targets_init);
 * duplicates of one (b,a,cell) resolve in target order, last write wins (CPU index_put order). */
int bmt_targets_init(uint8_t* obj, uint8_t* noobj, float* tx, float* tw, int64_t n, void* stream);
int bmt_make_targets(const float* targets, int n, const float* anchors, int A, int B, int G, float stride,
                     uint8_t* obj, uint8_t* noobj, float* tx, float* tw, void* stream);
/* decode + YOLO loss for one head (model/proposal_generator.py:281-335).
 * x: head output [B,S,A*3]; preds: [B, A*S, 3] written as (center_s, length_s, conf);
 * loss_ws: 8 floats {sum_x, sum_w, sum_bce_obj, sum_bce_noobj, n_obj, n_noobj, -, -} (overwritten);
 * with targets==NULL semantics (obj == NULL) only preds are produced. */
int bmt_prop_decode_loss(const float* x, const float* anchors, int B, int S, int A, float stride,
                         const uint8_t* obj, const uint8_t* noobj, const float* tx, const float* tw,
                         float* preds, float* loss_ws, void* stream);
/* ABI 9: the same with the predictions written into a SLICE of a larger buffer (batch element b at preds + b * pred_bs: the generator's
 * (B, sum over heads of A * S, 3) result -- the reference concatenates the heads' predictions, model/proposal_generator.py:380-383) and,
 * ws_zeroed != 0, into a loss_ws the caller has zeroed already (one fill for all heads). */
int bmt_prop_decode_loss2(const float* x, const float* anchors, int B, int S, int A, float stride,
                          const uint8_t* obj, const uint8_t* noobj, const float* tx, const float* tw,
                          float* preds, int64_t pred_bs, float* loss_ws, int ws_zeroed, void* stream);
/* ABI 9: bmt_prop_loss_finalize for n_heads heads at once (loss_ws [n_heads][8], losses [n_heads][5]) + the sums over all heads, the first
 * n_first heads and the others into sums [3][5] (overwritten): the generator's total loss and per-modality loss-term sums
 * (model/proposal_generator.py:363-378).  counts_first / counts_second (optional, device, 2 floats each): the GLOBAL {obj, noobj} cell counts
 * of the two modalities under data parallelism, written into loss_ws[i][4..5] before the means are taken. */
int bmt_prop_loss_finalize_multi(float* loss_ws, int n_heads, int n_first, const float* counts_first, const float* counts_second,
                                 float obj_coeff, float noobj_coeff, float* losses, float* sums, void* stream);
/* losses[0..3] = {mse_x, mse_w, bce_obj, bce_noobj} (means), losses[4] = total with coefficients */
int bmt_prop_loss_finalize(const float* loss_ws, float obj_coeff, float noobj_coeff, float* losses, void* stream);
/* dx[B,S,A*3] = d total / d x * (*gscale_dev) */
int bmt_prop_loss_bwd(const float* x, int B, int S, int A, const uint8_t* obj, const uint8_t* noobj,
                      const float* tx, const float* tw, const float* loss_ws, float obj_coeff, float noobj_coeff,
                      const float* gscale_dev, float* dx, void* stream);

/* ---------------------------------------------------------------- proposal post-processing
 * Replaces utilities/proposal_utils.py:115-121 (get_corner_coords), :136-149 (select_topk_predictions: argsort over all S
 * candidates), :152-161 (trim_proposals), :163-172 (remove_very_short_segments), :175-194 (non_max_suppresion) and their
 * compositions :196-212 (postprocess_preds) and sample/single_video_prediction.py:176-186 (generate_proposals).
 * preds: [B, S, 3] fp32 rows (center_s, length_s, confidence) -- or (start, end, confidence) without BMT_PP_CORNERS.
 * Selection: the k rows of largest confidence per video, descending; equal confidences in candidate-index order (a stable
 * descending sort).  With BMT_PP_FILTER only rows whose transformed segment has end - start > min_len are candidates.
 * out: [B, k, 3] transformed rows (start, end, confidence), rows >= count[b] are zero; out_idx (optional): [B, k] int64
 * candidate indices (-1 beyond count[b]); nms_thresh >= 0 applies greedy NMS to the selected rows (kept while tIoU <
 * nms_thresh against every kept row before it), count[b] is the number that survive. k <= 2048, S < 2^32. */
#define BMT_PP_CORNERS 1u   /* (center, length) -> (start, end) */
#define BMT_PP_TRIM 2u      /* start = min(max(start, 0), duration), end = min(end, duration) */
#define BMT_PP_FILTER 4u    /* candidates must satisfy end - start > min_len (after CORNERS / TRIM) */
typedef struct {
    const float* preds;
    int B;
    int64_t S;
    int k;
    unsigned flags;
    const float* durations; /* [B] seconds (device); required with BMT_PP_TRIM */
    float min_len;
    float nms_thresh;       /* < 0: no NMS */
    float* out;
    int64_t* out_idx;       /* may be NULL */
    int* count;             /* [B] */
    void* ws;               /* bmt_select_proposals_ws_bytes(B, S, k) bytes */
    size_t ws_bytes;
} bmt_select_proposals_args;
size_t bmt_select_proposals_ws_bytes(int B, int64_t S, int k);
int bmt_select_proposals(const bmt_select_proposals_args* a, void* stream);
/* in place over every candidate: the CORNERS / TRIM transforms above (get_corner_coords, trim_proposals) */
int bmt_transform_proposals(float* preds, int B, int64_t S, unsigned flags, const float* durations, void* stream);

/* ---------------------------------------------------------------- feature ingest
 * Host: read rows [row0, row1) (row1 < 0: to the end; the range is clipped to the array) of a 1-D / 2-D little-endian
 * float32 or float64 C-ordered .npy file into dst as float32 -- np.load + torch.from_numpy(x).float() + x[row0:row1] of
 * datasets/load_features.py:50-53,67-71,19-33 without intermediates.  dst == NULL: only *rows / *cols of the range.
 * Returns BMT_ENOENT when the file cannot be opened.  Thread-safe; no HIP call. */
int bmt_npy_shape(const char* path, int64_t* rows, int64_t* cols, int* elem_bytes);
int bmt_npy_read_rows(const char* path, int64_t row0, int64_t row1, float* dst, int64_t dst_floats, int64_t* rows,
                      int64_t* cols);
/* Device: packed [offsets[B], D] ragged rows (sample b = rows offsets[b] .. offsets[b+1]) -> out [B, T, D], rows beyond a
 * sample's length filled with pad: pad_sequence(batch_first=True, padding_value=pad) of
 * datasets/captioning_dataset.py:259-261 / pad_segment of datasets/load_features.py:38-44.  offsets: int64 [B+1], device. */
int bmt_pad_batch(const float* packed, const int64_t* offsets, int B, int T, int D, float pad, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BMT_HIP_H */
