#!/usr/bin/env python
"""one kernel configuration, a few launches (for rocprofv3 --pmc passes):
    tools/pmc_case.py gemm M N K prec        prec: 1 bf16, 3 bf16x3, 4 fp16, 5 fp16w2
    tools/pmc_case.py attnfwd B H Sq Sk dk prec
    tools/pmc_case.py attnbwd B H Sq Sk dk"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmt_amd import ops  # noqa: E402

dev = "cuda"
kind, a = sys.argv[1], [int(x) for x in sys.argv[2:]]
if kind == "gemm":
    M, N, K, prec = a
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.03
    out = torch.empty(M, N, device=dev)
    A = ops.make_planes(x, ops.act_fmt(prec))
    for _ in range(4):
        ops.linear_fwd(A, W, None, out=out, precision=prec)
elif kind in ("attnfwd", "attnbwd"):
    B, H, Sq, Sk, dk = a[:5]
    D = H * dk
    mk = lambda S: ops.make_planes(torch.randn(B * S, D, device=dev), "all")
    q, k, v = mk(Sq), mk(Sk), mk(Sk)
    mask = torch.ones(B, 1, Sk, dtype=torch.bool, device=dev)
    if kind == "attnfwd":
        prec = a[5]
        for _ in range(4):
            ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, precision=prec, out_fmt={3: "x3", 4: "f16", 1: "bwd"}[prec])
    else:
        o, lse = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, precision=ops.PREC_F16, out_fmt="f16")
        do = ops.make_planes(torch.randn(B * Sq, D, device=dev), "bwd")
        do = ops.Planes(do.hi[:, :D].contiguous(), None, B * Sq, D)
        for _ in range(4):
            ops.attn_bwd_planes(q, k, v, o, do, lse, B, Sq, Sk, D, mask, H, 0.0, (None, None, None))
torch.cuda.synchronize()
