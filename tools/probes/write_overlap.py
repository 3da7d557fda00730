#!/usr/bin/env python
"""how fast can 134 MB be written, and do HBM writes overlap a GEMM main loop that reads from L2?  (alt build, BMT_EXP=4 = GEMM without its epilogue)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402

dev = "cuda"
M, N, K = 8192, 4096, 1024
x = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) * 0.03
out = torch.empty(M, N, device=dev)
out2 = torch.empty(M, N, device=dev)
src = torch.randn(M, N, device=dev)
A = ops.make_planes(x, "f16")
ops.weight_planes(W, "f16")


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


mb = M * N * 4 / 1e6
t = timeit(lambda: out2.fill_(1.0))
print(f"fill {mb:.0f} MB: {t:.1f} us = {mb / t:.2f} TB/s")
t = timeit(lambda: out2.copy_(src))
print(f"copy {mb:.0f} MB (read + write): {t:.1f} us = {2 * mb / t:.2f} TB/s total")
g = lambda: ops.linear_fwd(A, W, None, out=out, precision=ops.PREC_F16)
print(f"gemm (BMT_EXP={os.environ.get('BMT_EXP', '0')}): {timeit(g):.1f} us")
s2 = torch.cuda.Stream()


def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        out2.fill_(1.0)
    g()
    torch.cuda.current_stream().wait_stream(s2)


print(f"gemm + concurrent fill on a second stream: {timeit(both):.1f} us")
