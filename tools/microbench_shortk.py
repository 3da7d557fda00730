#!/usr/bin/env python
"""Short-reduction GEMMs of the audio branch (K = d_aud = 128): where does the time go?  25600 x N x 128, x3 and x1, with
fp32 / hi+lo planes / hi-only outputs."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import ops  # noqa: E402

dev = "cuda"
M, K = 25600, 128


def t(fn, name, nbytes, flops, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{name:52s} {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s written  {flops / us / 1e6:6.0f} TF/s issued", flush=True)


x = ops.make_planes(torch.randn(M, K, device=dev), lo=True)[0]
for N in (1024, 3072):
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    hi = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lo = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Wp = ops.weight_planes(W)
    fl = 2.0 * M * N * K
    t(lambda: ops.gemm_bf16(x, Wp, out, ldc=N, bias=b, precision=3), f"N={N} x3 fp32 out", M * N * 4, 3 * fl)
    t(lambda: ops.gemm_bf16(x, Wp, None, bias=b, precision=3, out_planes=ops.Planes(hi, lo, M, N)), f"N={N} x3 hi+lo planes", M * N * 4, 3 * fl)
    t(lambda: ops.gemm_bf16(x, Wp, None, bias=b, precision=3, out_planes=ops.Planes(hi, None, M, N)), f"N={N} x3 hi plane only", M * N * 2, 3 * fl)
    t(lambda: ops.gemm_bf16(x, Wp, None, bias=b, precision=1, out_planes=ops.Planes(hi, None, M, N)), f"N={N} x1 hi plane only", M * N * 2, fl)
    t(lambda: ops.gemm_bf16(x, Wp, out, ldc=N, bias=b, precision=1), f"N={N} x1 fp32 out", M * N * 4, fl)
src = torch.randn(M, 3072, device=dev)
t(lambda: ops.make_planes(src, lo=True), "planes_kernel fp32 -> hi+lo (25600x3072)", M * 3072 * 4, 0.0)
t(lambda: src.clone(), "torch clone fp32 (25600x3072)", M * 3072 * 4, 0.0)
