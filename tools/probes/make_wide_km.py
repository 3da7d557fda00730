#!/usr/bin/env python
"""regenerates bmt_amd/csrc/exp/gemm_wide_km.hip from the text of gemm_wide_kernel (bmt_amd/csrc/gemm_bf16.hip): the same 256 x 256
ping-pong kernel with a K-MAJOR weight operand (the dX product of every nn.Linear).  The ping-pong structure, the activation operand
and the epilogue stay the product's, line for line; the patches below change the weight's DMA geometry + swizzle and its fragment reads.
Every patch asserts that its anchor text exists, so a change of the product kernel that the patches no longer fit fails loudly."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "bmt_amd", "csrc", "gemm_bf16.hip")).read()
a = src.index('template <bool F16>\n__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wide_kernel(const GemmB p) {')
b = src.index('// ===================================================================== reduction of 128')
k = src[a:b]


def R(old, new):
    global k
    assert k.count(old) >= 1, "anchor not found: " + old[:80]
    k = k.replace(old, new, 1)


R('void gemm_wide_kernel(const GemmB p) {', 'void gemm_wide_km_kernel(const GemmB p) {')
R('const __amdgpu_buffer_rsrc_t rsWh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)((int64_t)p.N * p.ldb * 2), 0x00020000);',
  '// k-major weight operand: plane [reduction rows = p.krows][output columns], rows past the reduction read as zero\n'
  '    const __amdgpu_buffer_rsrc_t rsWh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bh, 0, (int)((int64_t)p.krows * p.ldb * 2), 0x00020000);')
R('const __amdgpu_buffer_rsrc_t rsWl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bl ? p.Bl : p.Bh), 0, (int)((int64_t)p.N * p.ldb * 2), 0x00020000);',
  'const __amdgpu_buffer_rsrc_t rsWl = rsWh;      // (one plane: single-pass bf16)')
R('''        xvo[j] = row * (int)p.lda * 2 + ks * 16;
        wvo[j] = row * (int)p.ldb * 2 + ks * 16;''', '''        xvo[j] = row * (int)p.lda * 2 + ks * 16;
        // W half-tile = [64 reduction rows][128 output columns] = 16 pieces of 4 rows x 256 B; 16-byte chunk position = chunk ^ 4 (row & 3):
        // the four rows of a transposing read fall into the four 64-byte bank quarters (tools/probes/gemm_wide_km_layout.py)
        const int rowk = 4 * (2 * wid + j) + (lane >> 4);
        wvo[j] = rowk * (int)p.ldb * 2 + (((lane & 15) ^ (4 * (rowk & 3))) * 16);''')
# k_ = t * 128 bytes of a row-major K-tile = 64 reduction elements: (k_ >> 1) = 64 t reduction rows
R('''                                                         (n0 + 128 * hf) * (int)p.ldb * 2 + k_, 0, 0);              \\''',
  '''                                                         (k_ >> 1) * (int)p.ldb * 2 + (n0 + 128 * hf) * 2, 0, 0);   \\''')
R('''        offW[s] = o + wr * HT;
''', '')
R('    int offW[4], offX[4];', '    int offX[4];')
R('#define BMT_W_FRAG(slot_, off_, i_) as_bf16x8(*reinterpret_cast<const u32x4*>(smem + (slot_) * SLOT + (off_) + (i_) * 4096))',
  r'''#define BMT_W_FRAG(slot_, off_, i_) as_bf16x8(*reinterpret_cast<const u32x4*>(smem + (slot_) * SLOT + (off_) + (i_) * 4096))
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
    } while (0)''')
R('''        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \\
            _Pragma("unroll") for (int s = 0; s < 4; ++s) wa[i][s] = BMT_W_FRAG(e_, offW[s], i);                     \\
        BMT_W_LGKM0();                                                                                               \\''',
  '''        BMT_W_KREAD8(e_, 0);                                                                                         \\
        BMT_W_LGKM0();                                                                                               \\
        BMT_W_KPACK();                                                                                               \\''')
R('''        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \\
            _Pragma("unroll") for (int s = 0; s < 4; ++s) wa[i][s] = BMT_W_FRAG(e_, offW[s], 2 + i);                 \\
        if (next2_) BMT_W_DMA_X((t_) + 2, e_);                                                                       \\
        BMT_W_LGKM0();                                                                                               \\''',
  '''        BMT_W_KREAD8(e_, 2);                                                                                         \\
        if (next2_) BMT_W_DMA_X((t_) + 2, e_);                                                                       \\
        BMT_W_LGKM0();                                                                                               \\
        BMT_W_KPACK();                                                                                               \\''')
R('#undef BMT_W_FRAG\n', '#undef BMT_W_FRAG\n#undef BMT_W_KREAD\n#undef BMT_W_KREAD8\n#undef BMT_W_KPACK\n')
assert 'offW' not in k

HDR = '''// EXPERIMENT (exp/build.sh -> libbmt_exp.so; NOT yet run on a GPU: written after round 2's GPU budget was spent, harness
// tools/probes/gemm_wide_km_check.py): the 256 x 256 ping-pong GEMM kernel (gemm_wide_kernel, ../gemm_bf16.hip) with a K-MAJOR weight
// operand -- the dX product of every nn.Linear, dX[M][K_in] = dY[M][N_out] . W[N_out][K_in]: the reduction index is the ROW of W.  Today
// that product runs on the register-staged 128-row loop (1.52 ms per step, 18 % of the nominal matrix peak, DESIGN.md section 6).
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

'''
TAIL = '''
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
'''

OUT = os.path.join(ROOT, "bmt_amd", "csrc", "exp", "gemm_wide_km.hip")


def generate() -> str:
    return HDR + k + TAIL


if __name__ == "__main__":
    open(OUT, "w").write(generate())
    print("wrote", OUT)
