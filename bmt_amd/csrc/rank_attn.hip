// The encoder's self-attention over an input NARROWER than a head (ABI 11): the two weight-side kernels of the reassociated form.
//
// model/multihead_attention.py:62-84 projects the audio stream x (d_in = 128 columns) to d_model = 1024 for H = 4 heads of d_k = 256:
// q_h, k_h, v_h are rank-d_in images of the same x, and
//     S_h = q_h k_h^T = (x W'_h^T + c_h) x^T  (+ terms constant along the keys)     W'_h = W_k,h^T W_q,h  [d_in x d_in],  c_h = b_q,h W_k,h
//     O_h = P_h v_h   = (P_h x) W_v,h^T + b_v,h
// so the attention runs at width d_in against x itself (one key / value plane for all heads; bmt_attn_*_args.kv_shared).  What is left on
// the weight side is tiny (H d_in^2 d_k = 17 M multiply-adds per module) and runs in fp32 on the vector units, straight from the fp32
// parameters -- no operand planes of W_q / W_k, no rounding of the weights or of their gradients:
//   bmt_rank_prep    W' and c from the weights, W' written as the operand planes the products read (fp16 hi + lo for q' = x W'^T + c, bf16
//                    for dx = dq' W'), once per optimizer step;
//   bmt_rank_chain   dW_q,h += W_k,h dW'_h,   dW_k,h += W_q,h dW'_h^T + b_q,h^T dc_h,   db_q,h += W_k,h dc_h
//                    from dW' = dq'^T x (one item of the step's grouped weight-gradient launch) and dc = column sums of dq'; the last
//                    workgroup to finish zeroes dW' for the next accumulation.
#include "common.h"

namespace {

// workgroup = one row (h, a) of W': W'_h[a][b] = sum_r W_k[h dk + r][a] W_q[h dk + r][b].  Four waves split the reduction (wave w: r = w, w + 4,
// ...), lane l owns columns 2 l, 2 l + 1 (+ 128, ...): a row of W_q is a coalesced 8-byte read per lane, W_k[r][a] one scalar per wave; eight
// reduction steps are in flight per lane.  The four partial sums meet in LDS; wave 0 writes the planes.  c_h[a] rides along (lane 0's extra sum).
__global__ __launch_bounds__(256) void rank_prep_kernel(const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ bq,
                                                         int64_t ldw, int dk, int d_in, uint16_t* __restrict__ hi, uint16_t* __restrict__ fh,
                                                         uint16_t* __restrict__ fl, int64_t ldp, float* __restrict__ wp, float* __restrict__ c) {
    __shared__ float part[4][130];
    const int a = blockIdx.x % d_in, h = blockIdx.x / d_in;
    const int64_t row = (int64_t)h * d_in + a;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* wq = Wq + (int64_t)h * dk * ldw;
    const float* wk = Wk + (int64_t)h * dk * ldw + a;
    for (int b0 = 0; b0 < d_in; b0 += 128) {
        const int b = b0 + 2 * lane;
        float s0 = 0.f, s1 = 0.f, sc = 0.f;
#pragma unroll 8
        for (int r = w; r < dk; r += 4) {
            const float k = wk[(int64_t)r * ldw];
            const float2 q = b < d_in ? *reinterpret_cast<const float2*>(wq + (int64_t)r * ldw + b) : float2{0.f, 0.f};
            s0 = fmaf(k, q.x, s0);
            s1 = fmaf(k, q.y, s1);
            if (b0 == 0 && bq != nullptr) sc = fmaf(bq[h * dk + r], k, sc);      // (the same value in every lane)
        }
        part[w][2 * lane] = s0;
        part[w][2 * lane + 1] = s1;
        if (lane == 0 && b0 == 0) part[w][128] = sc;
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = 2 * lane + i, bb = b0 + col;
                if (bb < d_in) {
                    const float acc = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
                    if (wp) wp[row * d_in + bb] = acc;
                    if (hi) hi[row * ldp + bb] = __builtin_bit_cast(uint16_t, (__bf16)acc);
                    if (fh) {
                        const _Float16 f = (_Float16)acc;
                        fh[row * ldp + bb] = __builtin_bit_cast(uint16_t, f);
                        if (fl) fl[row * ldp + bb] = __builtin_bit_cast(uint16_t, (_Float16)(acc - (float)f));
                    }
                }
            }
            if (lane == 0 && b0 == 0 && c != nullptr) c[row] = (part[0][128] + part[1][128]) + (part[2][128] + part[3][128]);
        }
        __syncthreads();
    }
}

// workgroup = (head h, RT weight rows r0 ...), d_in = 128.  dW'_h sits in LDS ([a][b], rows of 129 floats: a column walk and a row walk are both
// conflict-free), the tile's weight rows beside it as [j][RT] (one broadcast 16-byte read per four rows).  Threads 0 .. 127 own column b of
//   dW_q[r][b] += sum_a W_k[r][a] dW'[a][b]
// threads 128 .. 255 column a of
//   dW_k[r][a] += sum_b W_q[r][b] dW'[a][b] + b_q[r] dc[a]
// and the first RT threads   db_q[r] += sum_a W_k[r][a] dc[a].
// Every workgroup of a head reads all of dW'_h, so it can only be zeroed when all are done: the last one to draw its ticket does it (and
// resets the ticket).
constexpr int RT = 8, DIN = 128;
__global__ __launch_bounds__(256) void rank_chain_kernel(const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ bq,
                                                          int64_t ldw, int dk, int H, float* __restrict__ dWp, const float* __restrict__ dc,
                                                          float* __restrict__ dWq, float* __restrict__ dWk, float* __restrict__ dbq, int64_t ldg,
                                                          int* __restrict__ ticket) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sP = sm;                              // [DIN][DIN + 1]
    float* sq = sm + DIN * (DIN + 1);            // [DIN][RT]
    static_assert((DIN * (DIN + 1)) % 4 == 0, "weight tiles 16-byte aligned");
    float* sk = sq + DIN * RT;                   // [DIN][RT]
    const int tiles = dk / RT, tid = threadIdx.x;
    const int h = blockIdx.x / tiles, r0 = h * dk + (blockIdx.x % tiles) * RT;
    const float* P = dWp + (int64_t)h * DIN * DIN;
    for (int i = tid; i < DIN * DIN / 4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(P)[i];
        const int a = (4 * i) / DIN, b = (4 * i) % DIN;
        float* d = sP + a * (DIN + 1) + b;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = tid; i < RT * DIN; i += 256) {
        const int r = i / DIN, col = i % DIN;
        sq[col * RT + r] = Wq[(int64_t)(r0 + r) * ldw + col];
        sk[col * RT + r] = Wk[(int64_t)(r0 + r) * ldw + col];
    }
    __syncthreads();
    const float* dch = dc ? dc + h * DIN : nullptr;
    const bool kside = tid >= DIN;               // wave-uniform (waves 2, 3)
    const int t = tid & (DIN - 1);
    const float* w8 = kside ? sq : sk;
    float acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int j = 0; j < DIN; ++j) {
        const float pv = kside ? sP[t * (DIN + 1) + j] : sP[j * (DIN + 1) + t];
        const float4 w0 = *reinterpret_cast<const float4*>(w8 + j * RT), w1 = *reinterpret_cast<const float4*>(w8 + j * RT + 4);
        acc[0] = fmaf(w0.x, pv, acc[0]); acc[1] = fmaf(w0.y, pv, acc[1]); acc[2] = fmaf(w0.z, pv, acc[2]); acc[3] = fmaf(w0.w, pv, acc[3]);
        acc[4] = fmaf(w1.x, pv, acc[4]); acc[5] = fmaf(w1.y, pv, acc[5]); acc[6] = fmaf(w1.z, pv, acc[6]); acc[7] = fmaf(w1.w, pv, acc[7]);
    }
    float* out = kside ? dWk : dWq;
    if (out != nullptr) {
        const float dct = (kside && bq && dch) ? dch[t] : 0.f;
#pragma unroll
        for (int r = 0; r < RT; ++r) out[(int64_t)(r0 + r) * ldg + t] += acc[r] + (dct != 0.f ? bq[r0 + r] * dct : 0.f);
    }
    if (tid < RT && dbq != nullptr && dch != nullptr) {
        float s = 0.f;
        for (int a = 0; a < DIN; ++a) s = fmaf(sk[a * RT + tid], dch[a], s);
        dbq[r0 + tid] += s;
    }
    // the last workgroup of the launch zeroes dW' (every other one has copied what it needs: its ticket was drawn after the copy)
    __shared__ int last;
    if (tid == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        const int64_t n = (int64_t)H * DIN * DIN;
        for (int64_t i = tid; i < n / 4; i += 256) reinterpret_cast<float4*>(dWp)[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (tid == 0) *ticket = 0;
    }
}

}  // namespace

extern "C" int bmt_rank_prep(const float* Wq, const float* Wk, const float* bq, int64_t ldw, int H, int dk, int d_in, uint16_t* wp_bf16, uint16_t* wp_f16,
                             uint16_t* wp_f16_lo, int64_t ldp, float* wp_f32, float* c, void* stream) {
    BMT_CHECK_ARG(Wq && Wk && H > 0 && dk > 0 && d_in > 0 && d_in % 2 == 0 && ldw >= d_in && ldw % 2 == 0 && (((uintptr_t)Wq) & 7) == 0 &&
                      (wp_bf16 || wp_f16 || wp_f32) && (!wp_f16_lo || wp_f16) && ldp >= d_in,
                  "bmt_rank_prep: bad arguments (H=%d dk=%d d_in=%d)", H, dk, d_in);
    hipLaunchKernelGGL(rank_prep_kernel, dim3(H * d_in), dim3(256), 0, (hipStream_t)stream, Wq, Wk, bq, ldw, dk, d_in, wp_bf16, wp_f16, wp_f16_lo, ldp,
                       wp_f32, c);
    BMT_CHECK_LAUNCH("bmt_rank_prep");
    return BMT_OK;
}

extern "C" int bmt_rank_chain(const float* Wq, const float* Wk, const float* bq, int64_t ldw, int H, int dk, int d_in, float* dWp, const float* dc,
                              float* dWq, float* dWk, float* dbq, int64_t ldg, int* ticket, void* stream) {
    BMT_CHECK_ARG(Wq && Wk && dWp && ticket && H > 0 && dk > 0 && dk % RT == 0 && d_in == DIN && ldw >= d_in && ldg >= d_in && (((uintptr_t)dWp) & 15) == 0,
                  "bmt_rank_chain: bad arguments (H=%d dk=%d d_in=%d: d_in must be %d, dk a multiple of %d)", H, dk, d_in, DIN, RT);
    constexpr int lds = (DIN * (DIN + 1) + 2 * DIN * RT) * (int)sizeof(float);
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)rank_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    hipLaunchKernelGGL(rank_chain_kernel, dim3(H * (dk / RT)), dim3(256), lds, (hipStream_t)stream, Wq, Wk, bq, ldw, dk, H, dWp, dc, dWq, dWk, dbq, ldg, ticket);
    BMT_CHECK_LAUNCH("bmt_rank_chain");
    return BMT_OK;
}
