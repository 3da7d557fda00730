#!/usr/bin/env python
"""the audio stream's reduction-of-128 projections (configs[1]: 25600 rows, d_model_audio 128): us per launch and output GB/s, plane-only
outputs as in the step.  A/B: BMT_GEMM_K128=0 (tile kernels) vs default; BMT_GEMM_K128_NCW=128|256 pins the chunk width."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402

dev = "cuda"


def timeit(f, iters=20):
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


M = 25600
ONE = int(sys.argv[sys.argv.index("--one") + 1]) if "--one" in sys.argv else 0      # only the shape with this N, 10 launches (counter runs)
x = torch.randn(M, 128, device=dev)
print(f"K128={os.environ.get('BMT_GEMM_K128', 'default')} NCW={os.environ.get('BMT_GEMM_K128_NCW', 'auto')}")
for name, N, prec, fmt, epi in (("A q|k|v", 3072, ops.PREC_F16W2, "f16only", {}), ("A->V k|v", 2048, ops.PREC_F16W2, "f16only", {}),
                                ("A q", 1024, ops.PREC_F16W2, "f16only", {}), ("A ffn1", 512, ops.PREC_F16W2, "f16", dict(relu=True)),
                                ("dX-like bf16", 1024, ops.PREC_BF16, "bwd", {}), ("fp32 out", 1024, ops.PREC_F16W2, None, {})):
    if ONE and (N != ONE or fmt != "f16only"):
        continue
    W = torch.randn(N, 128, device=dev) * 0.03
    b = torch.randn(N, device=dev)
    A = ops.make_planes(x, ops.act_fmt(prec))
    ops.weight_planes(W, ops.weight_fmt(prec))
    if fmt is None:
        out = torch.empty(M, N, device=dev)
        f = lambda: ops.linear_fwd(A, W, b, out=out, precision=prec, **epi)
        nbytes = M * N * 4
    else:
        f = lambda: ops.linear_fwd_planes(A, W, b, precision=prec, out_fmt=fmt, **epi)
        nbytes = M * N * 2 * len(ops._FMT[fmt])
    if ONE:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        continue
    us = timeit(f)
    print(f"  {name:14s} N={N:5d}  {us:7.1f} us   {nbytes / us / 1e3:7.0f} GB/s written")
