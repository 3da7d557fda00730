// Proposal post-processing on the device.
//
// Replaces, for a (B, S, 3) prediction tensor [center_s, length_s, confidence] with S up to ~2.9 M candidates per video:
//   select_topk_predictions   utilities/proposal_utils.py:136-149   (full argsort over S, gather, [:k])
//   get_corner_coords         utilities/proposal_utils.py:115-121
//   trim_proposals            utilities/proposal_utils.py:152-161
//   remove_very_short_segments utilities/proposal_utils.py:163-172  (as a validity filter BEFORE the selection)
//   non_max_suppresion        utilities/proposal_utils.py:175-194   (greedy, on the k selected, tiou_vectorized :11-57)
//   postprocess_preds         utilities/proposal_utils.py:196-212,  generate_proposals sample/single_video_prediction.py:176-186
//
// HBM-bound integer work: one pass reads the predictions (12 B / candidate) and writes a 4-byte order-preserving key; an
// exact 32-bit radix select (11 + 11 + 10 bits) over the keys finds the k-th largest confidence; equal keys are taken in
// index order (what a stable descending sort does); the <= k winners are sorted in LDS and emitted transformed.
// Algorithmic bytes per candidate: 12 (read) + 4 (key write) + 4 x 4 (key reads: 2 histogram passes, tie count, gather) = 32.
#include "common.h"

namespace {

constexpr int PP_NT = 256;
constexpr int PP_CHUNK = 4096;           // candidates per block
constexpr int PP_BINS = 2048;
constexpr int PP_MAXK = 2048;
// per-video selection state (uint32 words)
enum { ST_PREFIX = 0, ST_KREM = 1, ST_KEFF = 2, ST_NGT = 3, ST_GTCTR = 4, ST_WORDS = 8 };

struct Seg {
    float s, e;
};
// (center, length) -> (start, end), clamp to the duration: the reference's fp32 operation order
__device__ __forceinline__ Seg pp_transform(float c0, float c1, int flags, float dur) {
    Seg r;
    if (flags & BMT_PP_CORNERS) {
        const float h = c1 / 2.f;
        r.s = c0 - h;
        r.e = c0 + h;
    } else {
        r.s = c0;
        r.e = c1;
    }
    if (flags & BMT_PP_TRIM) {
        r.s = fminf(fmaxf(r.s, 0.f), dur);
        r.e = fminf(r.e, dur);
    }
    return r;
}
// order-preserving float -> uint32 (larger float -> larger key); 0 is reserved for "not a candidate"
__device__ __forceinline__ uint32_t pp_key(float conf) {
    uint32_t u = __float_as_uint(conf);
    if (u == 0x80000000u) u = 0u;          // -0.0 == +0.0: one tie group
    const uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return k == 0u ? 1u : k;
}

__device__ __forceinline__ void hist_flush(const uint32_t* lh, uint32_t* gh, int bins) {
    for (int i = threadIdx.x; i < bins; i += PP_NT) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

// pass 1: predictions -> keys, histogram of the top 11 key bits
__global__ __launch_bounds__(PP_NT) void pp_keys_kernel(const float* __restrict__ preds, int64_t S, int flags,
                                                        const float* __restrict__ durations, float min_len,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[PP_BINS];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < PP_BINS; i += PP_NT) lh[i] = 0;
    __syncthreads();
    const float dur = (flags & BMT_PP_TRIM) ? durations[b] : 0.f;
    const float* row = preds + (int64_t)b * S * 3;
    uint32_t* kb = keys + (int64_t)b * S;
    const int64_t i0 = (int64_t)blockIdx.x * PP_CHUNK;
#pragma unroll 4
    for (int j = 0; j < PP_CHUNK / PP_NT; ++j) {
        const int64_t i = i0 + j * PP_NT + threadIdx.x;
        if (i >= S) break;
        const float c0 = row[i * 3], c1 = row[i * 3 + 1], conf = row[i * 3 + 2];
        bool valid = true;
        if (flags & BMT_PP_FILTER) {
            const Seg sg = pp_transform(c0, c1, flags, dur);
            valid = (sg.e - sg.s) > min_len;
        }
        const uint32_t k = valid ? pp_key(conf) : 0u;
        kb[i] = k;
        if (k) atomicAdd(&lh[k >> 21], 1u);
    }
    __syncthreads();
    hist_flush(lh, hist + (int64_t)b * 3 * PP_BINS, PP_BINS);
}

// passes 2, 3: histogram of the next digit among keys that match the prefix found so far
template <int SHIFT, int BITS, int HSHIFT>
__global__ __launch_bounds__(PP_NT) void pp_hist_kernel(const uint32_t* __restrict__ keys, int64_t S,
                                                        const uint32_t* __restrict__ state, uint32_t* __restrict__ hist,
                                                        int pass) {
    __shared__ uint32_t lh[1 << BITS];
    const int b = blockIdx.y;
    const uint32_t* st = state + b * ST_WORDS;
    if (st[ST_KEFF] == 0) return;
    for (int i = threadIdx.x; i < (1 << BITS); i += PP_NT) lh[i] = 0;
    __syncthreads();
    const uint32_t prefix = st[ST_PREFIX] >> HSHIFT;
    const uint32_t* kb = keys + (int64_t)b * S;
    const int64_t i0 = (int64_t)blockIdx.x * PP_CHUNK;
#pragma unroll 4
    for (int j = 0; j < PP_CHUNK / PP_NT; ++j) {
        const int64_t i = i0 + j * PP_NT + threadIdx.x;
        if (i >= S) break;
        const uint32_t k = kb[i];
        if (k && (k >> HSHIFT) == prefix) atomicAdd(&lh[(k >> SHIFT) & ((1u << BITS) - 1u)], 1u);
    }
    __syncthreads();
    hist_flush(lh, hist + ((int64_t)b * 3 + pass) * PP_BINS, 1 << BITS);
}

// one block per video: walk the histogram from the top bin down to the bin that holds the k-th largest key
__global__ __launch_bounds__(PP_NT) void pp_pick_kernel(const uint32_t* __restrict__ hist, uint32_t* __restrict__ state, int pass,
                                                        int bins, int shift, int k) {
    __shared__ uint32_t part[PP_NT];
    __shared__ uint32_t total;
    const int b = blockIdx.x, t = threadIdx.x;
    uint32_t* st = state + b * ST_WORDS;
    const uint32_t* h = hist + ((int64_t)b * 3 + pass) * PP_BINS;
    const int per = bins / PP_NT;          // 8 or 4 bins per thread, thread 0 owns the TOP bins
    // state of the previous pass: read by every thread BEFORE the barriers below, written by one thread after them
    const uint32_t krem_in = st[ST_KREM], prev = (pass == 0) ? 0u : st[ST_PREFIX], keff_in = st[ST_KEFF];
    uint32_t loc[8];
    uint32_t sum = 0;
    for (int j = 0; j < per; ++j) {
        loc[j] = h[bins - 1 - (t * per + j)];
        sum += loc[j];
    }
    part[t] = sum;
    __syncthreads();
    // inclusive scan over threads (256 entries, Hillis-Steele)
    for (int o = 1; o < PP_NT; o <<= 1) {
        const uint32_t v = (t >= o) ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t == PP_NT - 1) total = part[t];
    __syncthreads();
    uint32_t krem;
    if (pass == 0) {
        krem = min((uint32_t)k, total);     // fewer candidates than k: take them all
        if (t == 0) {
            st[ST_KEFF] = krem;
            st[ST_GTCTR] = 0;
            if (krem == 0) { st[ST_PREFIX] = 0; st[ST_KREM] = 0; st[ST_NGT] = 0; }
        }
    } else {
        krem = krem_in;
    }
    if (krem == 0) return;
    const uint32_t before = part[t] - sum;  // keys in bins above this thread's bins
    if (before < krem && krem <= part[t]) {
        uint32_t acc = before;
        for (int j = 0; j < per; ++j) {
            if (acc + loc[j] >= krem) {
                const uint32_t digit = (uint32_t)(bins - 1 - (t * per + j));
                st[ST_PREFIX] = prev | (digit << shift);
                st[ST_KREM] = krem - acc;                       // still to take inside this bin
                if (pass == 2) st[ST_NGT] = keff_in - (krem - acc);   // keys strictly above the threshold key
                break;
            }
            acc += loc[j];
        }
    }
}

// keys equal to the threshold, per block (ties are taken in index order)
__global__ __launch_bounds__(PP_NT) void pp_count_eq_kernel(const uint32_t* __restrict__ keys, int64_t S,
                                                            const uint32_t* __restrict__ state, uint32_t* __restrict__ blockcnt) {
    __shared__ uint32_t red[PP_NT / 64];
    const int b = blockIdx.y;
    const uint32_t* st = state + b * ST_WORDS;
    uint32_t n = 0;
    if (st[ST_KEFF]) {
        const uint32_t T = st[ST_PREFIX];
        const uint32_t* kb = keys + (int64_t)b * S;
        const int64_t i0 = (int64_t)blockIdx.x * PP_CHUNK;
        for (int j = 0; j < PP_CHUNK / PP_NT; ++j) {
            const int64_t i = i0 + j * PP_NT + threadIdx.x;
            if (i < S) n += (kb[i] == T);
        }
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[(int64_t)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// winners -> sel[b][slot] = key << 32 | ~index   (slot order is irrelevant: they are sorted afterwards)
__global__ __launch_bounds__(PP_NT) void pp_gather_kernel(const uint32_t* __restrict__ keys, int64_t S, uint32_t* __restrict__ state,
                                                          const uint32_t* __restrict__ blockcnt, uint64_t* __restrict__ sel, int kpad) {
    __shared__ uint32_t red[PP_NT / 64];
    __shared__ uint32_t wbase[PP_NT / 64 + 1];
    const int b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    uint32_t* st = state + b * ST_WORDS;
    if (st[ST_KEFF] == 0) return;
    const uint32_t T = st[ST_PREFIX], krem = st[ST_KREM], ngt = st[ST_NGT];
    // ties in the blocks before this one
    uint32_t eqb = 0;
    for (int i = t; i < (int)blockIdx.x; i += PP_NT) eqb += blockcnt[(int64_t)b * gridDim.x + i];
    for (int o = 32; o > 0; o >>= 1) eqb += __shfl_xor(eqb, o, 64);
    if (lane == 0) red[wave] = eqb;
    __syncthreads();
    uint32_t eq_seen = red[0] + red[1] + red[2] + red[3];
    const uint32_t* kb = keys + (int64_t)b * S;
    uint64_t* out = sel + (int64_t)b * kpad;
    const int64_t i0 = (int64_t)blockIdx.x * PP_CHUNK;
    for (int j = 0; j < PP_CHUNK / PP_NT; ++j) {
        const int64_t i = i0 + j * PP_NT + t;
        const uint32_t k = (i < S) ? kb[i] : 0u;
        if (k > T) {
            const uint32_t slot = atomicAdd(&st[ST_GTCTR], 1u);
            out[slot] = ((uint64_t)k << 32) | (uint32_t)(~(uint32_t)i);
        }
        // block-wide rank, in index order, of the keys equal to T (uniform trip count: no early exit around the barriers)
        const bool eq = (k == T);
        const uint64_t m = __ballot(eq);
        if (lane == 0) wbase[wave + 1] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t base = eq_seen, tot = 0;
        for (int w = 0; w < PP_NT / 64; ++w) {
            const uint32_t c = wbase[w + 1];
            if (w < wave) base += c;
            tot += c;
        }
        if (eq) {
            const uint32_t rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (rank < krem) out[ngt + rank] = ((uint64_t)k << 32) | (uint32_t)(~(uint32_t)i);
        }
        eq_seen += tot;
        __syncthreads();
    }
}

__device__ __forceinline__ float pp_tiou(float s1, float e1, float s2, float e2) {
    // tiou_vectorized(center_length=False), utilities/proposal_utils.py:41-57, same operation order
    const float is = fmaxf(s1, s2), ie = fminf(e1, e2);
    const float inter = fmaxf(ie - is, 0.f);
    float uni = (e1 - s1) + (e2 - s2) - inter;
    uni = fminf(fmaxf(e1, e2) - fminf(s1, s2), uni);
    return inter / (uni + 1e-8f);
}

// one block per video: sort the winners (confidence descending, index ascending), transform, optional greedy NMS, emit
__global__ __launch_bounds__(1024) void pp_emit_kernel(const float* __restrict__ preds, int64_t S, int flags,
                                                       const float* __restrict__ durations, float nms_thresh,
                                                       const uint32_t* __restrict__ state, const uint64_t* __restrict__ sel, int kpad,
                                                       int k, float* __restrict__ out, int64_t* __restrict__ out_idx,
                                                       int* __restrict__ count) {
    __shared__ uint64_t v[PP_MAXK];
    __shared__ float ss[PP_MAXK], se[PP_MAXK], sc[PP_MAXK];
    __shared__ uint8_t alive[PP_MAXK];
    __shared__ uint32_t wsum[16];
    const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    const int n = (int)state[b * ST_WORDS + ST_KEFF];
    const uint64_t* in = sel + (int64_t)b * kpad;
    for (int i = t; i < kpad; i += nt) v[i] = (i < n) ? in[i] : 0ull;
    __syncthreads();
    // bitonic sort, descending
    for (int sz = 2; sz <= kpad; sz <<= 1) {
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int i = t; i < kpad; i += nt) {
                const int p = i ^ st;
                if (p > i) {
                    const uint64_t a = v[i], c = v[p];
                    const bool desc = ((i & sz) == 0);
                    if (desc ? (a < c) : (a > c)) { v[i] = c; v[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    const float dur = (flags & BMT_PP_TRIM) ? durations[b] : 0.f;
    const float* row = preds + (int64_t)b * S * 3;
    for (int i = t; i < n; i += nt) {
        const int64_t idx = (int64_t)(uint32_t)(~(uint32_t)(v[i] & 0xFFFFFFFFull));
        const float c0 = row[idx * 3], c1 = row[idx * 3 + 1], conf = row[idx * 3 + 2];
        const Seg sg = pp_transform(c0, c1, flags, dur);
        ss[i] = sg.s; se[i] = sg.e; sc[i] = conf;
        alive[i] = 1;
    }
    __syncthreads();
    if (nms_thresh >= 0.f) {
        // greedy NMS over the sorted list: a kept segment removes every later one whose tIoU with it is not < threshold
        for (int i = 0; i < n; ++i) {
            if (alive[i]) {     // uniform: read after the barrier below
                const float s1 = ss[i], e1 = se[i];
                for (int j = i + 1 + t; j < n; j += nt)
                    if (alive[j] && !(pp_tiou(s1, e1, ss[j], se[j]) < nms_thresh)) alive[j] = 0;
            }
            __syncthreads();
        }
    }
    // ordered compaction of the survivors
    int base = 0;
    float* ob = out + (int64_t)b * k * 3;
    int64_t* oi = out_idx ? out_idx + (int64_t)b * k : nullptr;
    for (int i0 = 0; i0 < n; i0 += nt) {
        const int i = i0 + t;
        const bool a = (i < n) && alive[i];
        const uint64_t m = __ballot(a);
        if ((t & 63) == 0) wsum[t >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        int pos = base, tot = 0;
        for (int w = 0; w < nt / 64; ++w) {
            if (w < (t >> 6)) pos += wsum[w];
            tot += wsum[w];
        }
        if (a) {
            pos += __popcll(m & ((1ull << (t & 63)) - 1ull));
            ob[pos * 3] = ss[i]; ob[pos * 3 + 1] = se[i]; ob[pos * 3 + 2] = sc[i];
            if (oi) oi[pos] = (int64_t)(uint32_t)(~(uint32_t)(v[i] & 0xFFFFFFFFull));
        }
        base += tot;
        __syncthreads();
    }
    for (int i = base + t; i < k; i += nt) {
        ob[i * 3] = 0.f; ob[i * 3 + 1] = 0.f; ob[i * 3 + 2] = 0.f;
        if (oi) oi[i] = -1;
    }
    if (t == 0) count[b] = base;
}

// in-place get_corner_coords / trim_proposals over every candidate
__global__ __launch_bounds__(PP_NT) void pp_transform_kernel(float* __restrict__ preds, int64_t S, int flags,
                                                             const float* __restrict__ durations) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * PP_NT + threadIdx.x;
    if (i >= S) return;
    float* r = preds + ((int64_t)b * S + i) * 3;
    const Seg sg = pp_transform(r[0], r[1], flags, (flags & BMT_PP_TRIM) ? durations[b] : 0.f);
    r[0] = sg.s;
    r[1] = sg.e;
}

int pp_kpad(int k) {
    int p = 2;
    while (p < k) p <<= 1;
    return p;
}
struct PpLayout {
    size_t keys, hist, state, blockcnt, sel, total;
    int nblk, kpad;
};
PpLayout pp_layout(int B, int64_t S, int k) {
    PpLayout L;
    L.nblk = bmt_cdiv(S, PP_CHUNK);
    L.kpad = pp_kpad(k);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    L.hist = o; o += al((size_t)B * 3 * PP_BINS * 4);
    L.state = o; o += al((size_t)B * ST_WORDS * 4);
    L.blockcnt = o; o += al((size_t)B * L.nblk * 4);
    L.sel = o; o += al((size_t)B * L.kpad * 8);
    L.keys = o; o += al((size_t)B * S * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" size_t bmt_select_proposals_ws_bytes(int B, int64_t S, int k) {
    if (B <= 0 || S <= 0 || k <= 0) return 0;
    return pp_layout(B, S, k).total;
}

extern "C" int bmt_select_proposals(const bmt_select_proposals_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->preds && a->out && a->count && a->ws, "bmt_select_proposals: null argument");
    BMT_CHECK_ARG(a->B > 0 && a->S > 0 && a->k > 0, "bmt_select_proposals: B=%d S=%lld k=%d must be positive", a->B, (long long)a->S, a->k);
    BMT_CHECK_ARG(a->k <= PP_MAXK, "bmt_select_proposals: k=%d exceeds %d", a->k, PP_MAXK);
    BMT_CHECK_ARG(a->S < (1ll << 32), "bmt_select_proposals: S=%lld does not fit a 32-bit candidate index", (long long)a->S);
    BMT_CHECK_ARG(!(a->flags & BMT_PP_TRIM) || a->durations, "bmt_select_proposals: BMT_PP_TRIM needs durations");
    const PpLayout L = pp_layout(a->B, a->S, a->k);
    BMT_CHECK_ARG(a->ws_bytes >= L.total, "bmt_select_proposals: workspace %zu < %zu bytes", a->ws_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)a->ws;
    uint32_t* keys = (uint32_t*)(w + L.keys);
    uint32_t* hist = (uint32_t*)(w + L.hist);
    uint32_t* state = (uint32_t*)(w + L.state);
    uint32_t* blockcnt = (uint32_t*)(w + L.blockcnt);
    uint64_t* sel = (uint64_t*)(w + L.sel);
    if (hipMemsetAsync(w, 0, L.blockcnt, s) != hipSuccess) {   // histograms + state
        bmt_set_error("bmt_select_proposals: hipMemsetAsync failed");
        return BMT_EHIP;
    }
    const dim3 grid(L.nblk, a->B);
    pp_keys_kernel<<<grid, PP_NT, 0, s>>>(a->preds, a->S, (int)a->flags, a->durations, a->min_len, keys, hist);
    pp_pick_kernel<<<a->B, PP_NT, 0, s>>>(hist, state, 0, 2048, 21, a->k);
    pp_hist_kernel<10, 11, 21><<<grid, PP_NT, 0, s>>>(keys, a->S, state, hist, 1);
    pp_pick_kernel<<<a->B, PP_NT, 0, s>>>(hist, state, 1, 2048, 10, a->k);
    pp_hist_kernel<0, 10, 10><<<grid, PP_NT, 0, s>>>(keys, a->S, state, hist, 2);
    pp_pick_kernel<<<a->B, PP_NT, 0, s>>>(hist, state, 2, 1024, 0, a->k);
    pp_count_eq_kernel<<<grid, PP_NT, 0, s>>>(keys, a->S, state, blockcnt);
    pp_gather_kernel<<<grid, PP_NT, 0, s>>>(keys, a->S, state, blockcnt, sel, L.kpad);
    const int nt = L.kpad >= 1024 ? 1024 : (L.kpad < 64 ? 64 : L.kpad);
    pp_emit_kernel<<<a->B, nt, 0, s>>>(a->preds, a->S, (int)a->flags, a->durations, a->nms_thresh, state, sel, L.kpad, a->k, a->out,
                                       a->out_idx, a->count);
    BMT_CHECK_LAUNCH("bmt_select_proposals");
    return BMT_OK;
}

extern "C" int bmt_transform_proposals(float* preds, int B, int64_t S, unsigned flags, const float* durations, void* stream) {
    BMT_CHECK_ARG(preds && B > 0 && S > 0, "bmt_transform_proposals: bad arguments");
    BMT_CHECK_ARG(!(flags & BMT_PP_TRIM) || durations, "bmt_transform_proposals: BMT_PP_TRIM needs durations");
    pp_transform_kernel<<<dim3(bmt_cdiv(S, PP_NT), B), PP_NT, 0, (hipStream_t)stream>>>(preds, S, (int)flags, durations);
    BMT_CHECK_LAUNCH("bmt_transform_proposals");
    return BMT_OK;
}
