#!/usr/bin/env python
"""GEMM loop experiments (alt build: BMT_ALT_FLAGS=-DBMT_EXP bash bmt_amd/csrc/build.sh; BMT_LIB_PATH=bmt_amd/lib/libbmt_hip_alt.so):
env BMT_EXP bit 1 = no DMA inside the k-loop, 2 = no MFMA, 4 = no epilogue.  Prints us per call for a few shapes / precisions with
fp32 output and with plane-only output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402

dev = "cuda"


def timeit(f, iters=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            f()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / iters)
    return sorted(ts)[2]


line = [f"EXP={os.environ.get('BMT_EXP', '0')}"]
for name, M, N, K in (("Vffn1", 8192, 4096, 1024), ("Vffn2", 8192, 1024, 4096), ("Vqkv", 8192, 3072, 1024), ("Aoproj", 25600, 1024, 1024)):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.03
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for prec in (ops.PREC_F16, ops.PREC_F16W2):
        A = ops.make_planes(x, ops.act_fmt(prec))
        ops.weight_planes(W, ops.weight_fmt(prec))
        t32 = timeit(lambda: ops.linear_fwd(A, W, b, out=out, precision=prec))
        tpl = timeit(lambda: ops.linear_fwd_planes(A, W, b, precision=prec, out_fmt="f16"))
        fl = 2.0 * M * N * K
        line.append(f"{name} {ops.prec_name(prec)[:12]}: f32 {t32:6.1f} us ({fl / t32 / 1e6:4.0f} TF) planes {tpl:6.1f} us ({fl / tpl / 1e6:4.0f} TF)")
print("\n  ".join(line), flush=True)
