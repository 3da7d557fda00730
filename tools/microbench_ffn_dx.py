#!/usr/bin/env python
"""The FFN fc2 dX GEMM (dh[8192,4096] = dY[8192,1024] . W2[1024,4096], gate = saved hidden) in its variants."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import ops  # noqa: E402

M, Dm, Dff = 8192, 1024, 4096
dy = torch.randn(M, Dm, device="cuda")
W2 = torch.randn(Dm, Dff, device="cuda") * 0.03
W1 = torch.randn(Dff, Dm, device="cuda") * 0.03
h = ops.make_planes(torch.relu(torch.randn(M, Dff, device="cuda")), lo=False)[0]
dyP = ops.make_planes(dy, lo=False)[0]
dhf = torch.randn(M, Dff, device="cuda")
dhP = ops.make_planes(dhf, lo=False)[0]


def t(fn, name, flops=2.0 * M * Dm * Dff, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{name:60s} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s", flush=True)


out = torch.empty(M, Dff, device="cuda")
op = ops.Planes(torch.empty(M, Dff, device="cuda", dtype=torch.bfloat16), None, M, Dff)
t(lambda: ops.linear_dx(dyP, W2, out=out), "fc2 dX  fp32 out, no gate")
t(lambda: ops.linear_dx(dyP, W2, out=out, gate=h, gate_scale=1.1), "fc2 dX  fp32 out, gate")
t(lambda: ops.linear_dx(dyP, W2, out_planes=op, gate=h, gate_scale=1.1), "fc2 dX  plane out, gate")
t(lambda: ops.linear_dx(dyP, W2, out_planes=op), "fc2 dX  plane out, no gate")
out1 = torch.empty(M, Dm, device="cuda")
t(lambda: ops.linear_dx(dhP, W1, out=out1), "fc1 dX  fp32 out (K=4096)")
t(lambda: ops.make_planes(dhf, lo=False), "planes of dh (fp32 -> hi)", flops=0.0)
