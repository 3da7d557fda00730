#!/usr/bin/env python
"""CPU check of the lane algebra of attn_bwd_dkvg_kernel (bmt_amd/csrc/attention_bf16.hip; numpy only): both operands of
dV^T[d][key] += dO^T[d x q] . P[q x key] (and dK^T += Qb^T . dS) are transposing reads of row-major images whose ROW is the reduction
index q.  X image (32 q x d_k, dual-purpose swizzle of the dQ kernel's K image) -> A fragments as in attn_bwd32_layout.py; Y image
(32 q x 128 keys, 256-byte rows, 16-byte chunk position = chunk ^ ((row & 3) << 2) applied by the DMA's source addresses) -> B fragments:
element 4 u + j of lane (l31, hh) <-> q = 16 kk + 8 u + 4 hh + j, key = 32 wave + l31."""
import sys

import numpy as np

from attn_fwd32_layout import conflicts
from attn_bwd32_layout import kswz, mfma


def tr_read(img, addr, lanes):
    """ds_read_b64_tr_b16: lane l of a 16-lane group gets element (l & 3) of the 8-byte rows addressed by lanes grp + 4 j + ((l & 15) >> 2)"""
    out = np.zeros((64, 4), dtype=np.int64)
    for l in lanes:
        grp, i = l & ~15, l & 15
        for j in range(4):
            src = grp + 4 * j + (i >> 2)
            out[l, j] = img[addr[src] // 2 + (i & 3)]
    return out


def check(DK):
    rng = np.random.default_rng(7 + DK)
    BQ, DT, ROWB = 32, DK // 32, DK * 2
    CPR, RPP = DK // 8, 64 // (DK // 8)
    PPW = (BQ // RPP) // 4
    X = rng.integers(-3, 4, size=(BQ, DK)).astype(np.int64)          # dO (or Qb) tile
    Y = rng.integers(-3, 4, size=(BQ, 128)).astype(np.int64)         # P (or dS) column block of the workgroup's 128 keys
    imgX = np.zeros(BQ * DK, dtype=np.int64)
    imgY = np.zeros(BQ * 128, dtype=np.int64)
    for wid in range(4):
        for j in range(PPW):
            for lane in range(64):
                row, cpos = (wid * PPW + j) * RPP + lane // CPR, lane % CPR
                c = cpos ^ kswz(row)
                dst = ((wid * PPW + j) * 1024 + 16 * lane) // 2
                imgX[dst:dst + 8] = X[row, 8 * c:8 * c + 8]
        for j in range(2):
            for lane in range(64):
                row, cpos = (wid * 2 + j) * 4 + lane // 16, lane % 16
                c = cpos ^ ((row & 3) << 2)
                dst = ((wid * 2 + j) * 1024 + 16 * lane) // 2
                imgY[dst:dst + 8] = Y[row, 8 * c:8 * c + 8]
    lanes = np.arange(64)
    hh, l31 = lanes >> 5, lanes & 31
    m16, gi = lanes & 15, (lanes >> 4) & 1
    mq, mr = m16 >> 2, m16 & 3
    xT0 = (4 * hh + mq) * ROWB + 64 * mq + 32 * gi + 16 * ((mr >> 1) ^ hh) + 8 * (mr & 1)
    TRG = [list(range(0, 32)), list(range(32, 64))]
    worst = 1
    ref = X.T @ Y                                                     # [d][key]
    for wid in range(4):
        yB0 = (4 * hh + mq) * 256 + 64 * (wid ^ mq) + 32 * gi + 8 * mr
        acc = np.zeros((64, DT, 16), dtype=np.int64)
        for n in range(2 * DT):
            dt, kk = n >> 1, n & 1
            A = np.zeros((64, 8), dtype=np.int64)
            Bf = np.zeros((64, 8), dtype=np.int64)
            for u in range(2):
                xa = (xT0 ^ (((dt & 3) << 6) | (u << 5))) + (dt >> 2) * 256 + (16 * kk + 8 * u) * ROWB
                ya = yB0 + (16 * kk + 8 * u) * 256
                worst = max(worst, conflicts(xa, 8, TRG), conflicts(ya, 8, TRG))
                A[:, 4 * u:4 * u + 4] = tr_read(imgX, xa, lanes)
                Bf[:, 4 * u:4 * u + 4] = tr_read(imgY, ya, lanes)
            D = mfma(A, Bf, hh, l31)
            for l in lanes:
                for r in range(16):
                    acc[l, dt, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh[l], l31[l]]
        for l in lanes:
            for dt in range(DT):
                for r in range(16):
                    d = 32 * dt + 8 * (r >> 2) + 4 * hh[l] + (r & 3)
                    assert acc[l, dt, r] == ref[d, 32 * wid + l31[l]], ("dkvg", DK, wid, l, dt, r)
    print(f"d_k {DK}: dV^T / dK^T fragments match numpy for all four waves; worst bank multiplicity of the transposing reads {worst}")
    return worst == 1


def check_emit():
    """the dQ kernel's emission: registers 4 i + j of lane (l31, hh) = key 8 i + 4 hh + j; after swapping (pw[4g], pw[4g+2]) and
    (pw[4g+1], pw[4g+3]) between the half-waves every lane stores 8 consecutive keys at key offset 16 g + 8 hh"""
    lanes = np.arange(64)
    hh, l31 = lanes >> 5, lanes & 31
    key_of = lambda r, h: 8 * (r >> 2) + 4 * h + (r & 3)
    pw = np.zeros((64, 8, 2), dtype=np.int64)              # packed pairs: dword j2 = registers 2 j2, 2 j2 + 1 -> (query, key) codes
    for l in lanes:
        for j2 in range(8):
            for e in range(2):
                pw[l, j2, e] = 1000 * l31[l] + key_of(2 * j2 + e, hh[l])

    def swap(a, b):                                        # v_permlane32_swap a, b: a[32..63] <-> b[0..31]
        ta = pw[32:, a].copy()
        pw[32:, a] = pw[:32, b]
        pw[:32, b] = ta
    for g in range(2):
        swap(4 * g + 0, 4 * g + 2)
        swap(4 * g + 1, 4 * g + 3)
    for l in lanes:
        for g in range(2):
            got = [pw[l, 4 * g + d, e] for d in range(4) for e in range(2)]
            want = [1000 * l31[l] + 16 * g + 8 * hh[l] + k for k in range(8)]
            assert got == want, (l, g, got, want)
    print("emission: every lane holds 8 consecutive keys of its query row after the half exchange")
    return True


if __name__ == "__main__":
    ok = all([check(256), check(128), check_emit()])
    sys.exit(0 if ok else 1)
