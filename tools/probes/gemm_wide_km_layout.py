#!/usr/bin/env python
"""CPU check of the weight-operand lane algebra of gemm_wide_km_kernel (bmt_amd/csrc/exp/gemm_wide_km.hip; numpy only).

One K-tile (64 reduction rows) of the k-major weight W[reduction][output column] and of the row-major activation X[row][reduction]:
  * the LDS image of a W half-tile ([64 rows][128 columns], 16 LDS-DMA pieces of 4 rows x 256 B, wave w fills pieces 2 w, 2 w + 1, 16-byte
    chunk position = chunk ^ 4 (row & 3)) as the kernel's source-side offsets build it;
  * the W fragment of MFMA (block blk of 32 columns, k16 step s): two ds_read_b64_tr_b16 (semantics as in attn_fwd32_layout.py) at the
    kernel's lane addresses -> A operand [32 output columns][16 reduction rows], reduction index in natural order;
  * C^T[32 columns][32 rows] = A . B with B = the activation fragment of the product kernel (row l31, reduction 16 s + 8 half + j),
    accumulated over s, against numpy; bank multiplicity of the transposing reads under the documented model."""
import sys

import numpy as np

from attn_fwd32_layout import conflicts


def check():
    rng = np.random.default_rng(7)
    NR, NC, ROWB, HT = 64, 128, 256, 16384                     # half-tile rows (reduction), columns, bytes per row, bytes per half-tile
    W = rng.integers(-3, 4, size=(NR, NC)).astype(np.int64)    # [reduction][output column]
    X = rng.integers(-3, 4, size=(32, NR)).astype(np.int64)    # 32 activation rows x 64 reduction
    img = np.zeros(HT // 2, dtype=np.int64)
    for wid in range(8):
        for j in range(2):
            for lane in range(64):
                rowk = 4 * (2 * wid + j) + (lane >> 4)
                c = (lane & 15) ^ (4 * (rowk & 3))
                dst = ((2 * wid + j) * 1024 + 16 * lane) // 2
                img[dst:dst + 8] = W[rowk, 8 * c:8 * c + 8]
    lanes = np.arange(64)
    half, l31 = lanes >> 5, lanes & 31
    m16, gi = lanes & 15, (lanes >> 4) & 1
    mq, mr = m16 >> 2, m16 & 3
    wT0 = (8 * half + mq) * ROWB + 64 * mq + 32 * gi + 8 * mr
    TRG = [list(range(0, 32)), list(range(32, 64))]
    worst = 1
    for blk in range(4):
        acc = np.zeros((32, 32), dtype=np.int64)
        for s in range(4):
            frag = np.zeros((64, 8), dtype=np.int64)
            for u in range(2):
                addr = (wT0 ^ (blk << 6)) + s * 4096 + u * 1024
                worst = max(worst, conflicts(addr, 8, TRG))
                for l in lanes:
                    grp, i = l & ~15, l & 15
                    for jj in range(4):
                        src = grp + 4 * jj + (i >> 2)
                        frag[l, 4 * u + jj] = img[addr[src] // 2 + (i & 3)]
            Am = np.zeros((32, 16), dtype=np.int64)
            Bm = np.zeros((16, 32), dtype=np.int64)
            for l in lanes:
                Am[l31[l], 8 * half[l]:8 * half[l] + 8] = frag[l]
                Bm[8 * half[l]:8 * half[l] + 8, l31[l]] = X[l31[l], 16 * s + 8 * half[l]:16 * s + 8 * half[l] + 8]
            acc += Am @ Bm
        ref = W[:, 32 * blk:32 * blk + 32].T @ X.T                # [column][row]
        assert (acc == ref).all(), ("block", blk)
    print(f"k-major W fragments: C^T blocks match numpy; worst bank multiplicity of the transposing reads {worst}")
    return worst == 1


if __name__ == "__main__":
    sys.exit(0 if check() else 1)
