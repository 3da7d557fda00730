#!/usr/bin/env python
"""a decoder layer's own products through bmt_gemm_small_batched as single products, under both groupings of a workgroup's waves (split 1: 64 x 64
blocks, 4: one tile per workgroup, the reduction over its waves): which reduction length should switch?  python tools/probes/small_split_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402
from tools.probes.gemm_small_time import timed  # noqa: E402

DEV = "cuda"


def main():
    flush = torch.zeros(64 << 20, device=DEV)
    A_ = ops._addr
    for prec, name in ((ops.PREC_BF16X3, "x3"), (ops.PREC_BF16, "bf16")):
        for M, N, K in [(928, 900, 300), (928, 300, 300), (928, 1024, 300), (928, 300, 1024), (928, 300, 600), (928, 1200, 300), (928, 300, 1200), (928, 1024, 1024)]:
            Kp = ops._pad64(K)
            bf = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16)
            a_hi, a_lo, b_hi, b_lo = bf(M, Kp), bf(M, Kp), bf(N, Kp), bf(N, Kp)
            out = torch.empty(M, N, device=DEV)
            row = []
            for sp in (1, 4):
                f = lambda: ops.gemm_batched(prec, M, N, Kp, 1, 1, A_(a_hi), A_(a_lo) if prec == ops.PREC_BF16X3 else None, Kp, A_(b_hi),
                                             A_(b_lo) if prec == ops.PREC_BF16X3 else None, Kp, C_=A_(out), ldc=N, split=sp)
                f()
                row.append(f"split {sp}: {timed(f, 30):5.1f} hot {timed(f, 12, flush):5.1f} cold")
            print(f"{name:5s} {M:4d} x {N:4d} x {K:4d}   " + "    ".join(row), flush=True)


if __name__ == "__main__":
    main()
