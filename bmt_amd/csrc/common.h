// Shared device/host helpers for libbmt_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/bmt_hip.h"

// ---------------------------------------------------------------- host-side error plumbing
void bmt_set_error(const char* fmt, ...);

#define BMT_CHECK_ARG(cond, ...)                     \
    do {                                             \
        if (!(cond)) {                               \
            bmt_set_error(__VA_ARGS__);              \
            return BMT_EINVAL;                       \
        }                                            \
    } while (0)

#define BMT_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            bmt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return BMT_EHIP;                                                     \
        }                                                                        \
    } while (0)

static inline int bmt_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- vector types for MFMA
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;  // 8 bf16 = 4 VGPR (MFMA A/B operand)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

// native vector types for 16-B / 8-B register slots: first-class SSA values (HIP's uint4/uint2 are structs whose array
// copies go through memcpy and can keep a whole register array in scratch)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    // D[32x32] += A[32x16] * B[16x32]; lane l holds A[row=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][col=l&31],
    // D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]  (cdna_hip_programming.md section 3)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fp16 operands: the same MFMA shapes and rate, 11 significand bits instead of 8 (the forward's operand policy, DESIGN.md
// "precision").  Fragments travel as bf16x8 = four untyped 32-bit registers; F16 selects the instruction.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16>
__device__ __forceinline__ f32x16 mfma32t(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// row of accumulator register r for this lane's half (l>>5)
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---------------------------------------------------------------- bf16 helpers (round-to-nearest-even)
__device__ __forceinline__ uint32_t f2bf_bits(float x) {
    uint32_t u = __float_as_uint(x);
    // NaN stays NaN (quiet); everything else RNE on the top 16 bits
    uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    return ((u & 0x7FFFFFFFu) > 0x7F800000u) ? ((u >> 16) | 0x40u) : (r >> 16);
}
__device__ __forceinline__ float bf_bits2f(uint32_t b) { return __uint_as_float(b << 16); }

// pack two floats into one dword of 2 bf16 (lo half = a, hi half = b): one v_cvt_pk_bf16_f32 (RNE) on gfx950
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// split (a, b) into hi = bf16(x) and lo = bf16(x - hi), both packed: 2 v_cvt_pk + 2 unpack + 1 packed subtract
__device__ __forceinline__ void split_bf2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));
}

// fp16 (RNE): pack two floats; split (a, b) into hi = fp16(x) and lo = fp16(x - hi)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ void split_h2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
}
__device__ __forceinline__ float h_bits2f(uint32_t b) { return (float)__builtin_bit_cast(_Float16, (uint16_t)b); }
template <bool F16>
__device__ __forceinline__ uint32_t pack_2(float a, float b) {
    if constexpr (F16) return pack_h2(a, b);
    else return pack_bf2(a, b);
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, step, site, i) = hash(i; key(seed, step, site)) >= p * 2^32.  The same function is used by every fused epilogue and by the
// standalone kernels, so forward and backward agree by construction.  Two levels:
//   * the KEY, once per kernel and thread: a 64-bit avalanche mix (two splitmix64 rounds) of (seed, step, site) -> (k0, k1);
//   * per ELEMENT: a 32-bit avalanche hash (two multiplies, three xor-shifts) of i + k0 with k1 folded in between the multiplies, so
//     that streams of different keys are not shifted copies of one sequence.
// (Round 1 ran the 64-bit mix per element: 6 quarter-rate 32-bit multiplies and ~20 other VALU operations for every value an epilogue
// writes -- 15 % of the attention forward at d_k = 256, tools/probes/attn_fwd32_check.py --variants.  The per-element part is now
// 2 multiplies + 8 operations; keep rate, pairwise agreement of streams, lag correlation and row / column rates of the 2-D layouts
// were checked against their binomial expectations on 4 M elements per key.)
struct DropCtx {
    uint32_t k0, k1;
    uint32_t thresh;  // keep iff hash >= thresh, thresh = p * 2^32
    float inv_keep;
    bool on;
};
__device__ __forceinline__ uint32_t bmt_hash32(uint32_t k0, uint32_t k1, uint64_t i) {
    uint32_t x = (uint32_t)i + (uint32_t)(i >> 32) + k0;      // (tensors past 2^32 elements repeat the stream shifted by one)
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x ^= k1;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ DropCtx make_drop(float p, const uint64_t* rng, uint32_t site) {
    DropCtx d;
    d.on = (p > 0.f) && (rng != nullptr);
    const uint64_t seed = d.on ? rng[0] : 0, step = d.on ? rng[1] : 0;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1u);
    z ^= step * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    d.k0 = (uint32_t)z;
    d.k1 = (uint32_t)(z >> 32);
    double t = (double)p * 4294967296.0;
    d.thresh = (p >= 1.f) ? 0xFFFFFFFFu : (uint32_t)t;
    d.inv_keep = (p < 1.f) ? 1.f / (1.f - p) : 0.f;
    return d;
}
__device__ __forceinline__ float drop_apply(const DropCtx& d, float v, uint64_t idx) {
    if (!d.on) return v;
    return (bmt_hash32(d.k0, d.k1, idx) >= d.thresh) ? v * d.inv_keep : 0.f;
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); `red` is a 4-float LDS scratch. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// XCD-aware remap of a 1-D block id: consecutive work ids land on the same XCD (block b runs on XCD b%8),
// so neighbouring tiles share that XCD's L2 (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
