#!/usr/bin/env python
"""CPU check of the lane algebra of attn_fwd32_kernel (bmt_amd/csrc/attention_bf16.hip) (no GPU, numpy only).

Emulates, with the formulas of the kernel source, (1) the LDS image the LDS-DMA builds (1-KB pieces, swizzle on the source side),
(2) the ds_read_b128 row fragments and the ds_read_b64_tr_b16 transposing reads, (3) v_mfma_f32_32x32x16 with the operand / result
lane layouts of common.h, and compares S^T = K.Q^T and O^T = V^T.P^T with numpy.  Also counts LDS bank conflicts of both read patterns
under the model of MI355X_MICROARCH.md (ds_read_b128: lane groups of 16, ds_read_b64_tr_b16: lane groups of 32, 64 banks x 4 B).

ds_read_b64_tr_b16 semantics as measured by tools/probes/tr_probe.hip (attention_bf16.hip): within a 16-lane group lane m supplies the
address of a 4-element chunk, lane i receives element j = chunk[lane 4 j + (i >> 2)][i & 3]."""
import sys

import numpy as np

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in g] for g in B128_GROUPS]


def conflicts(addr_bytes, nbytes, groups):
    """worst number of distinct addresses on one bank within a lane group"""
    worst = 1
    for g in groups:
        banks = {}
        for lane in g:
            for w in range(nbytes // 4):
                a = addr_bytes[lane] + 4 * w
                banks.setdefault((a // 4) % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def check(DK):
    rng = np.random.default_rng(DK)
    BC, KS, DT, ROWB = 32, DK // 16, DK // 32, DK * 2
    TILE = BC * ROWB
    CPR, RPP = DK // 8, 64 // (DK // 8)
    NP = BC // RPP
    PPW = NP // 4
    K = rng.integers(-3, 4, size=(BC, DK)).astype(np.int64)
    V = rng.integers(-3, 4, size=(BC, DK)).astype(np.int64)
    Q = rng.integers(-3, 4, size=(32, DK)).astype(np.int64)          # one wave's 32 queries

    # ---- (1) DMA: wave wid, piece j, lane -> LDS position (wid * PPW + j) * 1024 + 16 * lane holds global chunk (row, c)
    imgK = np.zeros(TILE // 2, dtype=np.int64)                         # element index = byte / 2
    imgV = np.zeros(TILE // 2, dtype=np.int64)
    for wid in range(4):
        for j in range(PPW):
            for lane in range(64):
                row = (wid * PPW + j) * RPP + lane // CPR
                cpos = lane % CPR
                ck = cpos ^ (row & 15)
                cv = cpos ^ (4 * (row & 3))
                dst = ((wid * PPW + j) * 1024 + 16 * lane) // 2
                imgK[dst:dst + 8] = K[row, 8 * ck:8 * ck + 8]
                imgV[dst:dst + 8] = V[row, 8 * cv:8 * cv + 8]

    lanes = np.arange(64)
    hh, l31 = lanes >> 5, lanes & 31

    # ---- (2a) S^T = K . Q^T
    s15 = l31 & 15
    kA0 = l31 * ROWB + 32 * (s15 >> 1) + 16 * (hh ^ (s15 & 1))
    st = np.zeros((64, 16), dtype=np.int64)
    worst_k = 1
    for ks in range(KS):
        addr = kA0 ^ (ks << 5)
        worst_k = max(worst_k, conflicts(addr, 16, B128_GROUPS))
        A = np.stack([imgK[a // 2:a // 2 + 8] for a in addr])                                   # lane: A[row = l31][k = 8 hh + j]
        Bq = np.stack([Q[l31[l], 16 * ks + 8 * hh[l]:16 * ks + 8 * hh[l] + 8] for l in lanes])  # lane: B[k = 8 hh + j][col = l31]
        Am = np.zeros((32, 16), dtype=np.int64)
        Bm = np.zeros((16, 32), dtype=np.int64)
        for l in lanes:
            Am[l31[l], 8 * hh[l]:8 * hh[l] + 8] = A[l]
            Bm[8 * hh[l]:8 * hh[l] + 8, l31[l]] = Bq[l]
        D = Am @ Bm
        for l in lanes:
            for r in range(16):
                st[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh[l], l31[l]]
    ref = K @ Q.T                                                                                # [key][q]
    for l in lanes:
        for r in range(16):
            key = 8 * (r >> 2) + 4 * hh[l] + (r & 3)
            assert st[l, r] == ref[key, l31[l]], ("S", DK, l, r)

    # ---- (2b) O^T = V^T . P^T with P = the S registers (any values do)
    m16, gi = lanes & 15, (lanes >> 4) & 1
    mq, mr = m16 >> 2, m16 & 3
    vL0 = (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 8 * mr
    o = np.zeros((64, DT, 16), dtype=np.int64)
    TRG = [list(range(0, 32)), list(range(32, 64))]
    worst_v = 1
    for n in range(2 * DT):
        dt, kk = n >> 1, n & 1
        frag = np.zeros((64, 8), dtype=np.int64)
        for u in range(2):
            addr = (vL0 ^ (dt << 6)) + (16 * kk + 8 * u) * ROWB
            worst_v = max(worst_v, conflicts(addr, 8, TRG))
            for l in lanes:
                grp, i = l & ~15, l & 15
                for j in range(4):
                    src = grp + 4 * j + (i >> 2)
                    frag[l, 4 * u + j] = imgV[addr[src] // 2 + (i & 3)]
        Am = np.zeros((32, 16), dtype=np.int64)
        Bm = np.zeros((16, 32), dtype=np.int64)
        for l in lanes:
            Am[l31[l], 8 * hh[l]:8 * hh[l] + 8] = frag[l]
            Bm[8 * hh[l]:8 * hh[l] + 8, l31[l]] = st[l, 8 * kk:8 * kk + 8]
        D = Am @ Bm
        for l in lanes:
            for r in range(16):
                o[l, dt, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh[l], l31[l]]
    refo = V.T @ ref                                                                             # [d][q] = sum_key V[key][d] S[key][q]
    for l in lanes:
        for dt in range(DT):
            for r in range(16):
                d = 32 * dt + 8 * (r >> 2) + 4 * hh[l] + (r & 3)
                assert o[l, dt, r] == refo[d, l31[l]], ("O", DK, l, dt, r)
    print(f"d_k {DK}: S^T and O^T match numpy; worst bank multiplicity: K row fragments {worst_k}, V transposing reads {worst_v}")
    return worst_k == 1 and worst_v == 1


if __name__ == "__main__":
    ok = all([check(256), check(128)])
    sys.exit(0 if ok else 1)
