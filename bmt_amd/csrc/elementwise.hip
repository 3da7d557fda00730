// Input preparation and small HBM-bound kernels (K8 of SURVEY.md 2.3):
//   rgb+flow add + positional table + dropout   (model/captioning_module.py:165,174-176; model/blocks.py:101-107)
//   vocabulary gather * sqrt(d) + positional table + dropout (model/blocks.py:42-46)
//   padding / causal masks, bit-exact            (model/masking.py:3-21; epoch_loops/captioning_epoch_loops.py:105-112)
//   standalone dropout, add, RNG step advance, strided 3-D copy (Conv1d weight re-layout).
// All are grid-stride, float4 where the widths allow (D % 4 == 0 on every hot-path tensor), coalesced over
// the padded (B,T,d) feature tensors.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

template <bool VEC>
__global__ __launch_bounds__(256) void prep_features_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                             const float* __restrict__ pe, float* __restrict__ out, int B, int S,
                                                             int D, float drop_p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = make_drop(drop_p, rng, site);
    const int64_t SD = (int64_t)S * D;
    const int64_t total = (int64_t)B * SD;
    if constexpr (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
            float4 v = ld4(a + i);
            if (b2) { const float4 w = ld4(b2 + i); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
            const float4 p = ld4(pe + (i % SD));
            v.x = drop_apply(dc, v.x + p.x, (uint64_t)i + 0); v.y = drop_apply(dc, v.y + p.y, (uint64_t)i + 1);
            v.z = drop_apply(dc, v.z + p.z, (uint64_t)i + 2); v.w = drop_apply(dc, v.w + p.w, (uint64_t)i + 3);
            *reinterpret_cast<float4*>(out + i) = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            float v = a[i];
            if (b2) v += b2[i];
            out[i] = drop_apply(dc, v + pe[i % SD], (uint64_t)i);
        }
    }
}

// ---- packed rows (bmt_pack_rows, bmt_prep_features_packed): the valid positions of a ragged batch, compacted in (b, t) order
// count: off[B + 1 + b] = number of valid positions of sample b (scratch half of `off`)
__global__ __launch_bounds__(256) void pack_count_kernel(const uint8_t* __restrict__ mask, int64_t mask_bs, int B, int S, int* __restrict__ off) {
    __shared__ int red[4];
    const int b = blockIdx.x;
    int c = 0;
    for (int t = threadIdx.x; t < S; t += 256) c += mask[(int64_t)b * mask_bs + t] != 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) off[B + 1 + b] = red[0] + red[1] + red[2] + red[3];
}
// place: off[b] = valid positions of the samples before b (off[B] = their total), row_map[off[b] + i] = b * S + t of the i-th valid position
// order (optional, int32 [B]; block 0 writes it): the samples in the order the attention kernels walk them (bmt_attn_*_args.b_order) -- a
// kernel's (batch, head, tile) work items are numbered sample-major and XCD x runs the x-th eighth of them (xcd_remap), i.e. B / 8 whole
// samples whose lengths are whatever the batch put there: at configs[1] (32 samples, lengths ~ U[T / 2, T], cost ~ length^2) the busiest XCD
// carries 1.2 - 1.4x the mean (profiles/r06_s_attn_order.txt: audio self-attention backward 399 us as drawn, 350 dealt out, 346 with equal
// lengths).  Here the samples are ranked by length and dealt to the eight XCD ranges in serpentine order (rank r: round r / 8, range r % 8
// forward in even rounds, backward in odd ones), each range longest first.
__global__ __launch_bounds__(256) void pack_place_kernel(const uint8_t* __restrict__ mask, int64_t mask_bs, int B, int S, int* __restrict__ off,
                                                          int* __restrict__ row_map, int* __restrict__ order) {
    __shared__ int red[4];
    __shared__ int wcnt[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int* cnt = off + B + 1;
    if (order != nullptr && b == 0 && (int)threadIdx.x < B) {
        const int j = threadIdx.x, cj = cnt[j];
        int rank = 0;
        for (int i = 0; i < B; ++i) {
            const int ci = cnt[i];
            rank += (ci > cj || (ci == cj && i < j)) ? 1 : 0;
        }
        const int rounds = B >> 3, rem = B & 7;                 // full rounds; samples of the last, partial round
        const int round = rank >> 3, pos = rank & 7, g = (round & 1) ? 7 - pos : pos;
        int base = 0;
        for (int gg = 0; gg < g; ++gg) {                         // range gg holds one sample per full round + one of the partial round's, if it got one
            const int pp = (rounds & 1) ? 7 - gg : gg;           // (the position that maps to range gg in the partial round)
            base += rounds + (pp < rem ? 1 : 0);
        }
        order[base + round] = j;
    }
    int c = 0;
    for (int i = threadIdx.x; i < b; i += 256) c += cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) red[wid] = c;
    __syncthreads();
    const int start = red[0] + red[1] + red[2] + red[3];
    if (threadIdx.x == 0) {
        off[b] = start;
        if (b == B - 1) off[B] = start + cnt[b];
    }
    int run = start;
    for (int t0 = 0; t0 < S; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const bool v = t < S && mask[(int64_t)b * mask_bs + t] != 0;
        const unsigned long long bal = __ballot(v);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();                      // (the previous round's wcnt has been read)
        if (lane == 0) wcnt[wid] = __popcll(bal);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += wcnt[w];
        if (v) row_map[run + wbase + before] = b * S + t;
        run += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void prep_features_packed_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                                    const float* __restrict__ pe, float* __restrict__ out, int S, int D,
                                                                    float drop_p, const uint64_t* rng, uint32_t site,
                                                                    const int* __restrict__ row_map, const int* __restrict__ rows_dev, int cap) {
    const DropCtx dc = make_drop(drop_p, rng, site);
    const int n = min(cap, *rows_dev);
    const int64_t total = (int64_t)n * D;
    if constexpr (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
            const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
            const int src = row_map[r];
            const int64_t so = (int64_t)src * D + c;
            float4 v = ld4(a + so);
            if (b2) { const float4 w = ld4(b2 + so); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
            const float4 p = ld4(pe + (int64_t)(src % S) * D + c);
            v.x = drop_apply(dc, v.x + p.x, (uint64_t)i + 0); v.y = drop_apply(dc, v.y + p.y, (uint64_t)i + 1);
            v.z = drop_apply(dc, v.z + p.z, (uint64_t)i + 2); v.w = drop_apply(dc, v.w + p.w, (uint64_t)i + 3);
            *reinterpret_cast<float4*>(out + i) = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
            const int src = row_map[r];
            float v = a[(int64_t)src * D + c];
            if (b2) v += b2[(int64_t)src * D + c];
            out[i] = drop_apply(dc, v + pe[(int64_t)(src % S) * D + c], (uint64_t)i);
        }
    }
}

__global__ __launch_bounds__(256) void prep_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ W,
                                                          const float* __restrict__ pe, float* __restrict__ out, int B, int S, int D,
                                                          int V, float emb_scale, float drop_p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = make_drop(drop_p, rng, site);
    const int64_t total = (int64_t)B * S * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t tok = i / D;
        const int d = (int)(i - tok * D);
        const int s = (int)(tok % S);
        int64_t id = ids[tok];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        out[i] = drop_apply(dc, W[id * D + d] * emb_scale + pe[(int64_t)s * D + d], (uint64_t)i);
    }
}

__global__ __launch_bounds__(256) void prep_embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                              float* __restrict__ dW, int B, int S, int D, int V, float emb_scale,
                                                              float drop_p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = make_drop(drop_p, rng, site);
    const int64_t total = (int64_t)B * S * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t tok = i / D;
        const int d = (int)(i - tok * D);
        int64_t id = ids[tok];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        atomicAdd(dW + id * D + d, drop_apply(dc, dout[i], (uint64_t)i) * emb_scale);
    }
}

__global__ __launch_bounds__(256) void mask_feat_kernel(const float* __restrict__ feat, int64_t bs, int64_t ld, float pad,
                                                         uint8_t* __restrict__ out, int B, int S) {
    const int64_t total = (int64_t)B * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / S), s = (int)(i % S);
        out[i] = feat[(int64_t)b * bs + (int64_t)s * ld] != pad ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void mask_tok_kernel(const int64_t* __restrict__ trg, int64_t pad_idx, uint8_t* __restrict__ src_mask,
                                                        uint8_t* __restrict__ trg_mask, int B, int S) {
    const int64_t total = (int64_t)B * S * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % S);
        const int64_t bi = i / S;
        const int r = (int)(bi % S);
        const int b = (int)(bi / S);
        const bool nonpad = trg[(int64_t)b * S + j] != pad_idx;
        if (trg_mask) trg_mask[i] = (nonpad && j <= r) ? 1 : 0;
        if (src_mask && r == 0) src_mask[(int64_t)b * S + j] = nonpad ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float drop_p,
                                                       const uint64_t* rng, uint32_t site) {
    const DropCtx dc = make_drop(drop_p, rng, site);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = drop_apply(dc, x[i], (uint64_t)i);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                   int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] + b[i];
}

__global__ void rng_advance_kernel(uint64_t* rng) {
    if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}

// the dropout stream of one part of a batch that is stepped in parts (micro-batches in flight together): the device's step counter with a seed
// of its own, so that element i of two parts never shares a mask

// out[i0][i1][i2] (contiguous) (+)= in[i0*s0 + i1*s1 + i2*s2]
__global__ __launch_bounds__(256) void copy3d_kernel(const float* __restrict__ in, int64_t s0, int64_t s1, int64_t s2,
                                                      float* __restrict__ out, int n0, int n1, int n2, int accumulate) {
    const int64_t total = (int64_t)n0 * n1 * n2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int i2 = (int)(i % n2);
        const int64_t r = i / n2;
        const int i1 = (int)(r % n1), i0 = (int)(r / n1);
        const float v = in[i0 * s0 + i1 * s1 + i2 * s2];
        out[i] = accumulate ? out[i] + v : v;
    }
}

}  // namespace

extern "C" int bmt_prep_features(const float* a, const float* b2, const float* pe, float* out, int B, int S, int D, float drop_p,
                                 const uint64_t* rng, uint32_t site, void* stream) {
    BMT_CHECK_ARG(a && pe && out && B > 0 && S > 0 && D > 0, "bmt_prep_features: bad args");
    const int64_t total = (int64_t)B * S * D;
    const bool vec = (D % 4 == 0) && al16(a) && al16(pe) && al16(out) && (!b2 || al16(b2));
    if (vec) hipLaunchKernelGGL(prep_features_kernel<true>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, a, b2, pe, out, B, S, D, drop_p, rng, site);
    else hipLaunchKernelGGL(prep_features_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b2, pe, out, B, S, D, drop_p, rng, site);
    BMT_CHECK_LAUNCH("bmt_prep_features");
    return BMT_OK;
}

extern "C" int bmt_pack_rows_ordered(const uint8_t* mask, int64_t mask_bs, int B, int S, int* off, int* row_map, int* order, void* stream) {
    BMT_CHECK_ARG(mask && off && row_map && B > 0 && S > 0 && mask_bs >= S && (int64_t)B * S < (1ll << 31), "bmt_pack_rows: bad args");
    BMT_CHECK_ARG(order == nullptr || B <= 256, "bmt_pack_rows_ordered: the balanced sample order is built for at most 256 samples");
    hipLaunchKernelGGL(pack_count_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mask, mask_bs, B, S, off);
    hipLaunchKernelGGL(pack_place_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mask, mask_bs, B, S, off, row_map, order);
    BMT_CHECK_LAUNCH("bmt_pack_rows");
    return BMT_OK;
}
extern "C" int bmt_pack_rows(const uint8_t* mask, int64_t mask_bs, int B, int S, int* off, int* row_map, void* stream) {
    return bmt_pack_rows_ordered(mask, mask_bs, B, S, off, row_map, nullptr, stream);
}

extern "C" int bmt_prep_features_packed(const float* a, const float* b2, const float* pe, float* out, int B, int S, int D, float drop_p,
                                        const uint64_t* rng, uint32_t site, const int* row_map, const int* rows_dev, void* stream) {
    BMT_CHECK_ARG(a && pe && out && row_map && rows_dev && B > 0 && S > 0 && D > 0, "bmt_prep_features_packed: bad args");
    const int64_t total = (int64_t)B * S * D;
    const bool vec = (D % 4 == 0) && al16(a) && al16(pe) && al16(out) && (!b2 || al16(b2));
    if (vec) hipLaunchKernelGGL(prep_features_packed_kernel<true>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, a, b2, pe, out, S, D, drop_p, rng, site, row_map, rows_dev, B * S);
    else hipLaunchKernelGGL(prep_features_packed_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b2, pe, out, S, D, drop_p, rng, site, row_map, rows_dev, B * S);
    BMT_CHECK_LAUNCH("bmt_prep_features_packed");
    return BMT_OK;
}

extern "C" int bmt_prep_embed(const int64_t* ids, const float* W, const float* pe, float* out, int B, int S, int D, int V,
                              float emb_scale, float drop_p, const uint64_t* rng, uint32_t site, void* stream) {
    BMT_CHECK_ARG(ids && W && pe && out && B > 0 && S > 0 && D > 0 && V > 0, "bmt_prep_embed: bad args");
    hipLaunchKernelGGL(prep_embed_kernel, dim3(grid_for((int64_t)B * S * D)), dim3(256), 0, (hipStream_t)stream, ids, W, pe, out, B, S, D, V, emb_scale, drop_p, rng, site);
    BMT_CHECK_LAUNCH("bmt_prep_embed");
    return BMT_OK;
}

extern "C" int bmt_prep_embed_bwd(const int64_t* ids, const float* dout, float* dW, int B, int S, int D, int V, float emb_scale,
                                  float drop_p, const uint64_t* rng, uint32_t site, void* stream) {
    BMT_CHECK_ARG(ids && dout && dW && B > 0 && S > 0 && D > 0 && V > 0, "bmt_prep_embed_bwd: bad args");
    hipLaunchKernelGGL(prep_embed_bwd_kernel, dim3(grid_for((int64_t)B * S * D)), dim3(256), 0, (hipStream_t)stream, ids, dout, dW, B, S, D, V, emb_scale, drop_p, rng, site);
    BMT_CHECK_LAUNCH("bmt_prep_embed_bwd");
    return BMT_OK;
}

extern "C" int bmt_mask_from_features(const float* feat, int64_t bs, int64_t ld, float pad, uint8_t* out, int B, int S, void* stream) {
    BMT_CHECK_ARG(feat && out && B > 0 && S > 0, "bmt_mask_from_features: bad args");
    hipLaunchKernelGGL(mask_feat_kernel, dim3(grid_for((int64_t)B * S)), dim3(256), 0, (hipStream_t)stream, feat, bs, ld, pad, out, B, S);
    BMT_CHECK_LAUNCH("bmt_mask_from_features");
    return BMT_OK;
}

extern "C" int bmt_mask_from_tokens(const int64_t* trg, int64_t pad_idx, uint8_t* src_mask, uint8_t* trg_mask, int B, int S, void* stream) {
    BMT_CHECK_ARG(trg && (src_mask || trg_mask) && B > 0 && S > 0, "bmt_mask_from_tokens: bad args");
    hipLaunchKernelGGL(mask_tok_kernel, dim3(grid_for((int64_t)B * S * S)), dim3(256), 0, (hipStream_t)stream, trg, pad_idx, src_mask, trg_mask, B, S);
    BMT_CHECK_LAUNCH("bmt_mask_from_tokens");
    return BMT_OK;
}

extern "C" int bmt_dropout(const float* x, float* y, int64_t n, float drop_p, const uint64_t* rng, uint32_t site, void* stream) {
    BMT_CHECK_ARG(x && y && n >= 0, "bmt_dropout: bad args");
    if (n == 0) return BMT_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, drop_p, rng, site);
    BMT_CHECK_LAUNCH("bmt_dropout");
    return BMT_OK;
}

namespace {
// out[r][0 .. Da) = a[r][:], out[r][Da .. Da + Db) = b[r][:]  (cat along the last dimension) / the inverse split: the decoder layer's
// torch.cat([Ca, Cv], -1) in front of the bridge (model/decoders.py:83) and the two halves of its gradient
template <bool SPLIT>
__global__ __launch_bounds__(256) void cat2_kernel(float* __restrict__ a, int64_t lda, int Da, float* __restrict__ b, int64_t ldb, int Db,
                                                    float* __restrict__ o, int64_t ldo, int rows) {
    const int W = Da + Db;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * W; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / W), c = (int)(i % W);
        float* src = c < Da ? a + (int64_t)r * lda + c : b + (int64_t)r * ldb + (c - Da);
        if (SPLIT) *src = o[(int64_t)r * ldo + c];
        else o[(int64_t)r * ldo + c] = *src;
    }
}
}  // namespace

extern "C" int bmt_cat2(const float* a, int64_t lda, int Da, const float* b, int64_t ldb, int Db, float* out, int64_t ldo, int rows, void* stream) {
    BMT_CHECK_ARG(a && b && out && rows >= 0 && Da > 0 && Db > 0 && lda >= Da && ldb >= Db && ldo >= Da + Db, "bmt_cat2: bad args");
    if (rows == 0) return BMT_OK;
    hipLaunchKernelGGL(cat2_kernel<false>, dim3(grid_for((int64_t)rows * (Da + Db))), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(a), lda, Da,
                       const_cast<float*>(b), ldb, Db, out, ldo, rows);
    BMT_CHECK_LAUNCH("bmt_cat2");
    return BMT_OK;
}

extern "C" int bmt_split2(const float* in, int64_t ldi, float* a, int64_t lda, int Da, float* b, int64_t ldb, int Db, int rows, void* stream) {
    BMT_CHECK_ARG(a && b && in && rows >= 0 && Da > 0 && Db > 0 && lda >= Da && ldb >= Db && ldi >= Da + Db, "bmt_split2: bad args");
    if (rows == 0) return BMT_OK;
    hipLaunchKernelGGL(cat2_kernel<true>, dim3(grid_for((int64_t)rows * (Da + Db))), dim3(256), 0, (hipStream_t)stream, a, lda, Da, b, ldb, Db,
                       const_cast<float*>(in), ldi, rows);
    BMT_CHECK_LAUNCH("bmt_split2");
    return BMT_OK;
}

extern "C" int bmt_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    BMT_CHECK_ARG(a && b && out && n >= 0, "bmt_add: bad args");
    if (n == 0) return BMT_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    BMT_CHECK_LAUNCH("bmt_add");
    return BMT_OK;
}

extern "C" int bmt_rng_advance(uint64_t* rng, void* stream) {
    BMT_CHECK_ARG(rng, "bmt_rng_advance: null");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng);
    BMT_CHECK_LAUNCH("bmt_rng_advance");
    return BMT_OK;
}


extern "C" int bmt_copy3d(const float* in, int64_t s0, int64_t s1, int64_t s2, float* out, int n0, int n1, int n2, int accumulate,
                          void* stream) {
    BMT_CHECK_ARG(in && out && n0 > 0 && n1 > 0 && n2 > 0, "bmt_copy3d: bad args");
    hipLaunchKernelGGL(copy3d_kernel, dim3(grid_for((int64_t)n0 * n1 * n2)), dim3(256), 0, (hipStream_t)stream, in, s0, s1, s2, out, n0, n1, n2, accumulate);
    BMT_CHECK_LAUNCH("bmt_copy3d");
    return BMT_OK;
}

// ---- backward helpers for fused activation/dropout epilogues
namespace {
// out = dy * (y != 0 ? scale : 0): derivative of relu(dropout(.)) / dropout(relu(.)) read off the saved OUTPUT y
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ dy, const float* __restrict__ y, float scale,
                                                    float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (y[i] != 0.f) ? dy[i] * scale : 0.f;
}
// out = res + dropout(x)   (ResidualConnection: model/blocks.py:134-136)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ out,
                                                           int64_t n, float drop_p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = make_drop(drop_p, rng, site);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = res[i] + drop_apply(dc, x[i], (uint64_t)i);
}
}  // namespace

extern "C" int bmt_gate(const float* dy, const float* y, float scale, float* out, int64_t n, void* stream) {
    BMT_CHECK_ARG(dy && y && out && n >= 0, "bmt_gate: bad args");
    if (n == 0) return BMT_OK;
    hipLaunchKernelGGL(gate_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, scale, out, n);
    BMT_CHECK_LAUNCH("bmt_gate");
    return BMT_OK;
}

extern "C" int bmt_dropout_add(const float* x, const float* res, float* out, int64_t n, float drop_p, const uint64_t* rng,
                               uint32_t site, void* stream) {
    BMT_CHECK_ARG(x && res && out && n >= 0, "bmt_dropout_add: bad args");
    if (n == 0) return BMT_OK;
    hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, res, out, n, drop_p, rng, site);
    BMT_CHECK_LAUNCH("bmt_dropout_add");
    return BMT_OK;
}
