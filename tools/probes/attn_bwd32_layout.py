#!/usr/bin/env python
"""CPU check of the lane algebra of the 32-query attention backward dQ kernel (round 2's experiment, since removed; kept for kswz / mfma, which tools/probes/attn_bwd_split_layout.py uses for the product's split backward; no GPU, numpy only).

Same emulation as attn_fwd32_layout.py (LDS image built by the LDS-DMA with source-side swizzles, ds_read_b128 row fragments,
ds_read_b64_tr_b16 transposing reads, v_mfma_f32_32x32x16 operand / result layouts, bank model).  What is new here:
  * the K image serves BOTH read patterns: row fragments for S^T = K . Q^T and transposing reads for dQ^T += K^T . dS^T.  Chunk position
    = chunk ^ f(row), f(row) = (row & 3) << 2 | (row >> 2) & 3: a bijection of row & 15 (the 16 keys of a ds_read_b128 lane group fall on
    16 different 16-byte bank slots) whose bits 2-3 come from row & 3 (the four rows of a transposing read fall into the four 64-byte
    quarters);
  * the V image is read by rows only (dP^T = V . dO^T): position = chunk ^ (row & 15), as the forward's K image."""
import sys

import numpy as np

from attn_fwd32_layout import B128_GROUPS, conflicts


def kswz(row):
    return ((row & 3) << 2) | ((row >> 2) & 3)


def mfma(Alanes, Blanes, hh, l31):
    """D = A . B with lane l holding A[row = l31][k = 8 hh + j] and B[k = 8 hh + j][col = l31]; returns the 32 x 32 product"""
    Am = np.zeros((32, 16), dtype=np.int64)
    Bm = np.zeros((16, 32), dtype=np.int64)
    for l in range(64):
        Am[l31[l], 8 * hh[l]:8 * hh[l] + 8] = Alanes[l]
        Bm[8 * hh[l]:8 * hh[l] + 8, l31[l]] = Blanes[l]
    return Am @ Bm


def check(DK):
    rng = np.random.default_rng(100 + DK)
    BC, KS, DT, ROWB = 32, DK // 16, DK // 32, DK * 2
    TILE = BC * ROWB
    CPR, RPP = DK // 8, 64 // (DK // 8)
    PPW = (BC // RPP) // 4
    K = rng.integers(-3, 4, size=(BC, DK)).astype(np.int64)
    V = rng.integers(-3, 4, size=(BC, DK)).astype(np.int64)
    Q = rng.integers(-3, 4, size=(32, DK)).astype(np.int64)
    dO = rng.integers(-3, 4, size=(32, DK)).astype(np.int64)

    imgK = np.zeros(TILE // 2, dtype=np.int64)
    imgV = np.zeros(TILE // 2, dtype=np.int64)
    for wid in range(4):
        for j in range(PPW):
            for lane in range(64):
                row = (wid * PPW + j) * RPP + lane // CPR
                cpos = lane % CPR
                ck = cpos ^ kswz(row)
                cv = cpos ^ (row & 15)
                dst = ((wid * PPW + j) * 1024 + 16 * lane) // 2
                imgK[dst:dst + 8] = K[row, 8 * ck:8 * ck + 8]
                imgV[dst:dst + 8] = V[row, 8 * cv:8 * cv + 8]

    lanes = np.arange(64)
    hh, l31 = lanes >> 5, lanes & 31

    # ---- S^T = K . Q^T and dP^T = V . dO^T (row fragments)
    fk = np.array([kswz(r) for r in l31])
    kA0 = l31 * ROWB + 32 * (fk >> 1) + 16 * (hh ^ (fk & 1))
    s15 = l31 & 15
    vA0 = l31 * ROWB + 32 * (s15 >> 1) + 16 * (hh ^ (s15 & 1))
    st = np.zeros((64, 16), dtype=np.int64)
    dp = np.zeros((64, 16), dtype=np.int64)
    worst_k = worst_v = 1
    for ks in range(KS):
        ak, av = kA0 ^ (ks << 5), vA0 ^ (ks << 5)
        worst_k = max(worst_k, conflicts(ak, 16, B128_GROUPS))
        worst_v = max(worst_v, conflicts(av, 16, B128_GROUPS))
        Ak = [imgK[a // 2:a // 2 + 8] for a in ak]
        Av = [imgV[a // 2:a // 2 + 8] for a in av]
        Bq = [Q[l31[l], 16 * ks + 8 * hh[l]:16 * ks + 8 * hh[l] + 8] for l in lanes]
        Bo = [dO[l31[l], 16 * ks + 8 * hh[l]:16 * ks + 8 * hh[l] + 8] for l in lanes]
        Ds, Dp = mfma(Ak, Bq, hh, l31), mfma(Av, Bo, hh, l31)
        for l in lanes:
            for r in range(16):
                row = (r & 3) + 8 * (r >> 2) + 4 * hh[l]
                st[l, r] += Ds[row, l31[l]]
                dp[l, r] += Dp[row, l31[l]]
    refS, refP = K @ Q.T, V @ dO.T
    for l in lanes:
        for r in range(16):
            key = 8 * (r >> 2) + 4 * hh[l] + (r & 3)
            assert st[l, r] == refS[key, l31[l]], ("S", DK, l, r)
            assert dp[l, r] == refP[key, l31[l]], ("dP", DK, l, r)

    # ---- dQ^T = K^T . dS^T with dS = any function of the registers (here S + 2 dP), B operand = registers 8 kk .. 8 kk + 7
    ds = st + 2 * dp
    m16, gi = lanes & 15, (lanes >> 4) & 1
    mq, mr = m16 >> 2, m16 & 3
    L0 = (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1)
    dq = np.zeros((64, DT, 16), dtype=np.int64)
    TRG = [list(range(0, 32)), list(range(32, 64))]
    worst_t = 1
    for n in range(2 * DT):
        dt, kk = n >> 1, n & 1
        frag = np.zeros((64, 8), dtype=np.int64)
        for u in range(2):
            addr = (L0 ^ ((dt << 6) | (u << 5))) + (16 * kk + 8 * u) * ROWB
            worst_t = max(worst_t, conflicts(addr, 8, TRG))
            for l in lanes:
                grp, i = l & ~15, l & 15
                for j in range(4):
                    src = grp + 4 * j + (i >> 2)
                    frag[l, 4 * u + j] = imgK[addr[src] // 2 + (i & 3)]
        D = mfma(frag, [ds[l, 8 * kk:8 * kk + 8] for l in lanes], hh, l31)
        for l in lanes:
            for r in range(16):
                dq[l, dt, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh[l], l31[l]]
    refdS = refS + 2 * refP                         # [key][q]
    refdQ = K.T @ refdS                             # [d][q] = sum_key K[key][d] dS[key][q]
    for l in lanes:
        for dt in range(DT):
            for r in range(16):
                d = 32 * dt + 8 * (r >> 2) + 4 * hh[l] + (r & 3)
                assert dq[l, dt, r] == refdQ[d, l31[l]], ("dQ", DK, l, dt, r)
    print(f"d_k {DK}: S^T, dP^T and dQ^T match numpy; worst bank multiplicity: K rows {worst_k}, V rows {worst_v}, K transposing reads {worst_t}")
    return worst_k == 1 and worst_v == 1 and worst_t == 1


if __name__ == "__main__":
    ok = all([check(256), check(128)])
    sys.exit(0 if ok else 1)
