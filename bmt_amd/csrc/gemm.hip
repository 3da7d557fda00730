// bmt_gemm / bmt_conv1d / bmt_colsum -- MFMA GEMM for gfx950 (CDNA4).
//
// One kernel template covers nn.Linear forward, both of its backward products and the Conv1d
// heads (as an implicit GEMM: no im2col buffer ever exists in HBM).
//
//   block tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in a 2x2 grid, 64x64 per wave
//   = 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 acc VGPRs).
//   Operands live in HBM as fp32; the staging pass converts to bf16 on the way into LDS and,
//   in BMT_PREC_BF16X3 mode, also stores the rounding residual (lo = bf16(x - hi)), so that
//   acc += hi*hi + hi*lo + lo*hi reproduces an fp32 product to ~2^-16 while staying on the
//   bf16 matrix pipe (3 passes ~ 830 TFLOP/s peak vs 157 TFLOP/s for the native f32 MFMA).
//   LDS image: [128 rows][8 slots of 16 B] per operand and split half, slot index XOR-swizzled with
//   (row>>1)&7 so a ds_read_b128 lane group (16 rows, same k-slot) touches 16 distinct 16-B
//   slots of the 256-B bank row (conflict-free), cdna_hip_programming.md T2.
//   Global->LDS goes through registers (fp32->bf16 conversion is needed anyway): the loads for
//   tile t+1 are issued before the MFMA block of tile t and written to LDS after it (T14),
//   two workgroups per CU cover each other's barriers.
//   1-D grid with XCD-aware remap (T1) so tiles sharing an A row-panel sit on one XCD's L2.
#include <stdarg.h>

#include "common.h"

// ---------------------------------------------------------------- error state (shared by all .hip files)
static thread_local char g_err[512] = "";
void bmt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* bmt_last_error(void) { return g_err; }
extern "C" int bmt_version(void) { return BMT_ABI_VERSION; }
extern "C" int bmt_device_cus(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // one operand, one split half: 16 KiB

enum { OP_KC = 0, OP_RC = 1, OP_CONV_KC = 2, OP_CONV_RC = 3 };

struct Operand {
    const float* p;
    int64_t ld;
    int rows;   // extent of the non-reduction index (M or N)
    int vec;    // float4 loads legal (ld % 4 == 0 and base 16-B aligned)
    // implicit-conv addressing
    int S;      // sequence length (rows per batch item)
    int Dc;     // channels per tap
    int sign;   // +1: src row = s + t + shift0 ; -1: src row = s - t + shift0
    int shift0;
};

struct GemmP {
    Operand a, b;
    float* C;
    int64_t ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    int kchunk;  // K range per split (multiple of BK)
    float alpha;
    unsigned flags;
    const float* bias;
    const float* residual;
    int64_t ldr;
    const float* gate;
    int64_t ldg;
    float gate_scale;
    float drop_p;
    const uint64_t* rng;
    uint32_t site;
    uint16_t* Chi;
    uint16_t* Clo;
    int64_t ldp;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- global -> registers: 32 floats per thread of a [128 rows][64 k] operand tile.
// KC-style modes: v[i*8 + j] = elem(row = rbase + 32 i, k = koct*8 + j)      (i<4, j<8)
// RC-style modes: v[j*4 + c] = elem(row = mq*4 + c,     k = koct*8 + j)      (j<8, c<4)
template <int MODE>
__device__ __forceinline__ void gload(const Operand& op, int r0, int k0, int kend, int tid, float (&v)[32]) {
    if constexpr (MODE == OP_KC || MODE == OP_CONV_KC) {
        const int koct = tid & 7, rbase = tid >> 3;
        const int k = k0 + koct * 8;
        int tshift = 0, c0 = k;
        if constexpr (MODE == OP_CONV_KC) {
            const int t = k / op.Dc;
            c0 = k - t * op.Dc;
            tshift = op.sign * t + op.shift0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + rbase + 32 * i;
            bool ok = row < op.rows;
            const float* src;
            if constexpr (MODE == OP_CONV_KC) {
                const int s = row % op.S;
                const int sp = s + tshift;
                ok = ok && sp >= 0 && sp < op.S;
                src = op.p + (int64_t)(row + tshift) * op.ld + c0;
            } else {
                src = op.p + (int64_t)row * op.ld + k;
            }
            if (ok && op.vec && k + 7 < kend) {
                const float4 x = ld4(src), y = ld4(src + 4);
                v[i * 8 + 0] = x.x; v[i * 8 + 1] = x.y; v[i * 8 + 2] = x.z; v[i * 8 + 3] = x.w;
                v[i * 8 + 4] = y.x; v[i * 8 + 5] = y.y; v[i * 8 + 6] = y.z; v[i * 8 + 7] = y.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i * 8 + j] = (ok && k + j < kend) ? src[j] : 0.f;
            }
        }
    } else {
        const int mq = tid & 31, koct = tid >> 5;
        const int row = r0 + mq * 4;
        int tshift = 0, c0 = row;
        if constexpr (MODE == OP_CONV_RC) {
            const int t = row / op.Dc;
            c0 = row - t * op.Dc;
            tshift = op.sign * t + op.shift0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + koct * 8 + j;   // reduction index
            bool ok = k < kend;
            const float* src;
            if constexpr (MODE == OP_CONV_RC) {
                const int s = k % op.S;
                const int sp = s + tshift;
                ok = ok && sp >= 0 && sp < op.S;
                src = op.p + (int64_t)(k + tshift) * op.ld + c0;
            } else {
                src = op.p + (int64_t)k * op.ld + row;
            }
            if (ok && op.vec && row + 3 < op.rows) {
                const float4 x = ld4(src);
                v[j * 4 + 0] = x.x; v[j * 4 + 1] = x.y; v[j * 4 + 2] = x.z; v[j * 4 + 3] = x.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[j * 4 + c] = (ok && row + c < op.rows) ? src[c] : 0.f;
            }
        }
    }
}

__device__ __forceinline__ int lds_slot(int row, int oct) { return row * 8 + (oct ^ ((row >> 1) & 7)); }

// ---- registers -> LDS (bf16 hi [+ lo])
template <int MODE, int NPASS>
__device__ __forceinline__ void lstore(uint4* hi, uint4* lo, int tid, const float (&v)[32]) {
    if constexpr (MODE == OP_KC || MODE == OP_CONV_KC) {
        const int koct = tid & 7, rbase = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rbase + 32 * i;
            uint4 h, l;
            if constexpr (NPASS == 3) {
                split_bf2(v[i * 8 + 0], v[i * 8 + 1], h.x, l.x);
                split_bf2(v[i * 8 + 2], v[i * 8 + 3], h.y, l.y);
                split_bf2(v[i * 8 + 4], v[i * 8 + 5], h.z, l.z);
                split_bf2(v[i * 8 + 6], v[i * 8 + 7], h.w, l.w);
                lo[lds_slot(row, koct)] = l;
            } else {
                h.x = pack_bf2(v[i * 8 + 0], v[i * 8 + 1]);
                h.y = pack_bf2(v[i * 8 + 2], v[i * 8 + 3]);
                h.z = pack_bf2(v[i * 8 + 4], v[i * 8 + 5]);
                h.w = pack_bf2(v[i * 8 + 6], v[i * 8 + 7]);
            }
            hi[lds_slot(row, koct)] = h;
        }
    } else {
        const int mq = tid & 31, koct = tid >> 5;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = mq * 4 + c;
            uint4 h, l;
            if constexpr (NPASS == 3) {
                split_bf2(v[0 * 4 + c], v[1 * 4 + c], h.x, l.x);
                split_bf2(v[2 * 4 + c], v[3 * 4 + c], h.y, l.y);
                split_bf2(v[4 * 4 + c], v[5 * 4 + c], h.z, l.z);
                split_bf2(v[6 * 4 + c], v[7 * 4 + c], h.w, l.w);
                lo[lds_slot(row, koct)] = l;
            } else {
                h.x = pack_bf2(v[0 * 4 + c], v[1 * 4 + c]);
                h.y = pack_bf2(v[2 * 4 + c], v[3 * 4 + c]);
                h.z = pack_bf2(v[4 * 4 + c], v[5 * 4 + c]);
                h.w = pack_bf2(v[6 * 4 + c], v[7 * 4 + c]);
            }
            hi[lds_slot(row, koct)] = h;
        }
    }
}

template <int AMODE, int BMODE, int NPASS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sAh = reinterpret_cast<uint4*>(smem);
    uint4* sBh = reinterpret_cast<uint4*>(smem + TILE_BYTES);
    uint4* sAl = reinterpret_cast<uint4*>(smem + 2 * TILE_BYTES);
    uint4* sBl = reinterpret_cast<uint4*>(smem + 3 * TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int w = xcd_remap(blockIdx.x, ntiles);
    const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float va[32], vb[32];
    if (kbeg < kend) {
        gload<AMODE>(p.a, m0, kbeg, kend, tid, va);
        gload<BMODE>(p.b, n0, kbeg, kend, tid, vb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        lstore<AMODE, NPASS>(sAh, sAl, tid, va);
        lstore<BMODE, NPASS>(sBh, sBl, tid, vb);
        __syncthreads();
        if (k0 + BK < kend) {  // prefetch the next tile while this one is multiplied
            gload<AMODE>(p.a, m0, k0 + BK, kend, tid, va);
            gload<BMODE>(p.b, n0, k0 + BK, kend, tid, vb);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int oct = 2 * s + half;
            bf16x8 ah[2], bh[2], al[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
                ah[i] = as_bf16x8(sAh[lds_slot(ra, oct)]);
                bh[i] = as_bf16x8(sBh[lds_slot(rb, oct)]);
                if constexpr (NPASS == 3) {
                    al[i] = as_bf16x8(sAl[lds_slot(ra, oct)]);
                    bl[i] = as_bf16x8(sBl[lds_slot(rb, oct)]);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (NPASS == 3) {
                        acc[i][j] = mfma32(al[i], bh[j], acc[i][j]);
                        acc[i][j] = mfma32(ah[i], bl[j], acc[i][j]);
                    }
                    acc[i][j] = mfma32(ah[i], bh[j], acc[i][j]);
                }
        }
        __syncthreads();
    }

    // ---------------- epilogue
    const DropCtx dc = make_drop(p.drop_p, p.rng, p.site);
    const unsigned f = p.flags;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (f & BMT_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + i * 32 + acc_row(r, half);
                if (row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha + bv;
                const int64_t idx = (int64_t)row * p.ldc + col;
                if (f & BMT_EPI_DROP_PRE) v = drop_apply(dc, v, (uint64_t)idx);
                if (f & BMT_EPI_RELU) v = fmaxf(v, 0.f);
                if (f & BMT_EPI_DROP_POST) v = drop_apply(dc, v, (uint64_t)idx);
                if (f & BMT_EPI_GATE) v = (p.gate[(int64_t)row * p.ldg + col] != 0.f) ? v * p.gate_scale : 0.f;
                if (f & BMT_EPI_RESIDUAL) v += p.residual[(int64_t)row * p.ldr + col];
                if (f & BMT_EPI_ACCUM) atomicAdd(p.C + idx, v);
                else if (p.C) p.C[idx] = v;
                if (p.Chi) {
                    const __bf16 h = (__bf16)v;
                    const int64_t pi = (int64_t)row * p.ldp + col;
                    p.Chi[pi] = __builtin_bit_cast(uint16_t, h);
                    if (p.Clo) p.Clo[pi] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
                }
            }
        }
}

template <int AMODE, int BMODE>
int launch_modes(const GemmP& p, int precision, int splitk, hipStream_t st) {
    dim3 grid(p.tiles_m * p.tiles_n, splitk), block(256);
    if (precision == BMT_PREC_BF16X3) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)gemm_kernel<AMODE, BMODE, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      4 * TILE_BYTES);
            attr_done = true;
        }
        hipLaunchKernelGGL((gemm_kernel<AMODE, BMODE, 3>), grid, block, 4 * TILE_BYTES, st, p);
    } else {
        hipLaunchKernelGGL((gemm_kernel<AMODE, BMODE, 1>), grid, block, 2 * TILE_BYTES, st, p);
    }
    BMT_CHECK_LAUNCH("bmt_gemm");
    return BMT_OK;
}

int launch_gemm(GemmP& p, int amode, int bmode, int precision, int splitk, hipStream_t st) {
    BMT_CHECK_ARG(precision == BMT_PREC_BF16 || precision == BMT_PREC_BF16X3, "bmt_gemm: bad precision %d", precision);
    BMT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K >= 0, "bmt_gemm: bad sizes M=%d N=%d K=%d", p.M, p.N, p.K);
    BMT_CHECK_ARG(splitk >= 1, "bmt_gemm: splitk=%d", splitk);
    const unsigned nonlin = BMT_EPI_RELU | BMT_EPI_DROP_PRE | BMT_EPI_DROP_POST | BMT_EPI_GATE | BMT_EPI_BIAS | BMT_EPI_RESIDUAL;
    BMT_CHECK_ARG(splitk == 1 || ((p.flags & BMT_EPI_ACCUM) && !(p.flags & nonlin)),
                  "bmt_gemm: splitk>1 needs BMT_EPI_ACCUM and no other epilogue op (flags=0x%x)", p.flags);
    BMT_CHECK_ARG(!(p.flags & BMT_EPI_BIAS) || p.bias, "bmt_gemm: BIAS flag without bias pointer");
    BMT_CHECK_ARG(!(p.flags & BMT_EPI_RESIDUAL) || p.residual, "bmt_gemm: RESIDUAL flag without pointer");
    BMT_CHECK_ARG(!(p.flags & BMT_EPI_GATE) || p.gate, "bmt_gemm: GATE flag without pointer");
    p.tiles_m = bmt_cdiv(p.M, BM);
    p.tiles_n = bmt_cdiv(p.N, BN);
    const int ktiles = bmt_cdiv(p.K > 0 ? p.K : 1, BK);
    if (splitk > ktiles) splitk = ktiles;
    p.kchunk = bmt_cdiv(ktiles, splitk) * BK;
    splitk = bmt_cdiv(p.K > 0 ? p.K : 1, p.kchunk);
#define BMT_CASE(am, bm) \
    if (amode == am && bmode == bm) return launch_modes<am, bm>(p, precision, splitk, st);
    BMT_CASE(OP_KC, OP_KC)
    BMT_CASE(OP_KC, OP_RC)
    BMT_CASE(OP_RC, OP_RC)
    BMT_CASE(OP_RC, OP_KC)
    BMT_CASE(OP_CONV_KC, OP_KC)
    BMT_CASE(OP_RC, OP_CONV_RC)
#undef BMT_CASE
    bmt_set_error("bmt_gemm: unsupported operand mode pair (%d,%d)", amode, bmode);
    return BMT_EINVAL;
}

inline int vec_ok(const float* p, int64_t ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0); }

// ---------------------------------------------------------------- column sums (bias gradients)
// grid: (ceil(N/64), row chunks); block 256 = 4 row-lanes x 64 columns; atomic accumulate of chunk partials.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N,
                                                      float* __restrict__ out, int rows_per_blk) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_blk, rend = min(M, rbeg + rows_per_blk);
    float s = 0.f;
    if (c < N)
        for (int r = rbeg + rl; r < rend; r += 4) s += X[(int64_t)r * ldx + c];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int bmt_gemm(const bmt_gemm_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->A && a->B && (a->C || a->C_hi), "bmt_gemm: null pointer");
    BMT_CHECK_ARG(!(a->flags & BMT_EPI_ACCUM) || a->C, "bmt_gemm: ACCUM needs the fp32 output");
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.a = Operand{a->A, a->lda, a->M, 0, 1, 1, 1, 0};
    p.b = Operand{a->B, a->ldb, a->N, 0, 1, 1, 1, 0};
    p.a.vec = vec_ok(a->A, a->lda);
    p.b.vec = vec_ok(a->B, a->ldb);
    p.C = a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K;
    p.alpha = a->alpha; p.flags = a->flags; p.bias = a->bias;
    p.residual = a->residual; p.ldr = a->ldr; p.gate = a->gate; p.ldg = a->ldg; p.gate_scale = a->gate_scale;
    p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    p.Chi = a->C_hi; p.Clo = a->C_lo; p.ldp = a->ldp;
    return launch_gemm(p, a->a_kcontig ? OP_KC : OP_RC, a->b_kcontig ? OP_KC : OP_RC, a->precision,
                       a->splitk < 1 ? 1 : a->splitk, (hipStream_t)stream);
}

extern "C" int bmt_conv1d(const bmt_conv1d_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->x && a->W && a->y, "bmt_conv1d: null pointer");
    BMT_CHECK_ARG(a->k >= 1 && (a->k & 1), "bmt_conv1d: kernel size %d must be odd", a->k);
    BMT_CHECK_ARG(a->Din % 8 == 0 && a->Dout % 8 == 0, "bmt_conv1d: Din=%d / Dout=%d must be multiples of 8", a->Din, a->Dout);
    const int pad = a->k / 2;
    const int M = a->B * a->S;
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.alpha = 1.f; p.flags = a->flags; p.bias = a->bias;
    p.gate = a->gate; p.gate_scale = a->gate_scale;
    p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    if (a->mode == 0) {
        // y[m, o] = sum_{t,c} x[m + t - pad, c] * Wp[o, t, c];  W given as [Dout][k][Din] (tap-major) by the host side
        p.a = Operand{a->x, a->Din, M, vec_ok(a->x, a->Din), a->S, a->Din, +1, -pad};
        p.b = Operand{a->W, (int64_t)a->k * a->Din, a->Dout, vec_ok(a->W, (int64_t)a->k * a->Din), 1, 1, 1, 0};
        p.C = a->y; p.ldc = a->Dout; p.ldg = a->Dout; p.M = M; p.N = a->Dout; p.K = a->k * a->Din;
        return launch_gemm(p, OP_CONV_KC, OP_KC, a->precision, 1, (hipStream_t)stream);
    } else if (a->mode == 1) {
        // dx[m, c] = sum_{t,o} dy[m - t + pad, o] * Wt[c, t, o];  W given as [Din][k][Dout]
        p.a = Operand{a->x, a->Dout, M, vec_ok(a->x, a->Dout), a->S, a->Dout, -1, +pad};
        p.b = Operand{a->W, (int64_t)a->k * a->Dout, a->Din, vec_ok(a->W, (int64_t)a->k * a->Dout), 1, 1, 1, 0};
        p.C = a->y; p.ldc = a->Din; p.ldg = a->Din; p.M = M; p.N = a->Din; p.K = a->k * a->Dout;
        return launch_gemm(p, OP_CONV_KC, OP_KC, a->precision, 1, (hipStream_t)stream);
    } else if (a->mode == 2) {
        // dWp[o, (t,c)] += sum_m dy[m, o] * x[m + t - pad, c];  here x = dy (a->x), W = layer input (a->W), y = dWp
        p.a = Operand{a->x, a->Dout, a->Dout, vec_ok(a->x, a->Dout), 1, 1, 1, 0};
        p.b = Operand{a->W, a->Din, a->k * a->Din, vec_ok(a->W, a->Din), a->S, a->Din, +1, -pad};
        p.C = a->y; p.ldc = (int64_t)a->k * a->Din; p.M = a->Dout; p.N = a->k * a->Din; p.K = M;
        p.flags = BMT_EPI_ACCUM;
        return launch_gemm(p, OP_RC, OP_CONV_RC, a->precision, a->splitk < 1 ? 1 : a->splitk, (hipStream_t)stream);
    }
    bmt_set_error("bmt_conv1d: bad mode %d", a->mode);
    return BMT_EINVAL;
}

extern "C" int bmt_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, void* stream) {
    BMT_CHECK_ARG(X && out && M >= 0 && N > 0, "bmt_colsum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) {
            bmt_set_error("bmt_colsum: memset failed");
            return BMT_EHIP;
        }
    }
    if (M == 0) return BMT_OK;
    const int rows_per_blk = 256;
    dim3 grid(bmt_cdiv(N, 64), bmt_cdiv(M, rows_per_blk));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, X, ldx, M, N, out, rows_per_blk);
    BMT_CHECK_LAUNCH("bmt_colsum");
    return BMT_OK;
}
