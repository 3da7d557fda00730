#!/usr/bin/env python
"""kernel timings at configs[1] shapes, per operand format (HIP events, median of interleaved rounds):
    python tools/microbench.py [gemm] [attn]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmt_amd import ops  # noqa: E402

dev = "cuda"
what = set(sys.argv[1:]) or {"gemm", "attn"}


def timeit(fns, iters=20, rounds=5):
    """fns: {name: callable}; interleaved rounds, returns {name: median us per call}"""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) * 1e3 / iters)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


if "gemm" in what:
    print("== forward GEMMs  y = x W^T + b  (fp32 out)")
    for name, M, N, K in (("V qkv", 8192, 3072, 1024), ("V ffn1", 8192, 4096, 1024), ("V ffn2", 8192, 1024, 4096), ("V oproj", 8192, 1024, 1024),
                          ("A qkv", 25600, 3072, 128), ("A oproj", 25600, 128, 1024), ("A<-V kv", 8192, 2048, 1024), ("V<-A kv", 25600, 2048, 128),
                          ("dec q", 928, 1024, 300), ("gen", 928, 10000, 300)):
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.03
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        fns = {}
        for prec in (ops.PREC_BF16, ops.PREC_F16, ops.PREC_F16W2, ops.PREC_BF16X3):
            A = ops.make_planes(x, ops.act_fmt(prec))
            ops.weight_planes(W, ops.weight_fmt(prec))
            fns[ops.prec_name(prec)] = (lambda A=A, prec=prec: ops.linear_fwd(A, W, b, out=out, precision=prec))
        r = timeit(fns)
        fl = 2.0 * M * N * K
        print(f"{name:9s} {M:6d}x{N:5d}x{K:5d} " + "  ".join(f"{k}: {v:7.1f} us {fl / v / 1e6:6.0f} TF" for k, v in r.items()), flush=True)

if "attn" in what:
    print("== attention forward (planes in, planes out)")
    for name, B, H, Sq, Sk, dk in (("A self", 32, 4, 800, 800, 256), ("V self", 32, 4, 256, 256, 256), ("A<-V", 32, 4, 800, 256, 256),
                                   ("V<-A", 32, 4, 256, 800, 256), ("C<-A", 32, 4, 29, 800, 256), ("deep A self", 16, 8, 800, 800, 128)):
        D = H * dk
        mk = lambda S: ops.make_planes(torch.randn(B * S, D, device=dev), "all")
        q, k, v = mk(Sq), mk(Sk), mk(Sk)
        mask = torch.ones(B, 1, Sk, dtype=torch.bool, device=dev)
        for bi in range(B):
            mask[bi, 0, Sk - (bi * Sk) // (2 * B):] = False
        fns = {}
        for prec, fmt in ((ops.PREC_BF16, "bwd"), (ops.PREC_F16, "f16"), (ops.PREC_BF16X3, "x3")):
            fns[ops.prec_name(prec)] = (lambda prec=prec, fmt=fmt: ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, precision=prec, out_fmt=fmt))
        r = timeit(fns)
        fl = 4.0 * B * Sq * Sk * D
        print(f"{name:11s} B{B} H{H} {Sq}x{Sk} dk{dk} " + "  ".join(f"{k}: {v:7.1f} us {fl / v / 1e6:6.0f} TF" for k, v in r.items()), flush=True)
    print("== attention backward (dq, dk, dv planes)")
    for name, B, H, Sq, Sk, dk in (("A self", 32, 4, 800, 800, 256), ("V self", 32, 4, 256, 256, 256), ("A<-V", 32, 4, 800, 256, 256),
                                   ("V<-A", 32, 4, 256, 800, 256), ("C<-A", 32, 4, 29, 800, 256)):
        D = H * dk
        mk = lambda S: ops.make_planes(torch.randn(B * S, D, device=dev), "all")
        q, k, v = mk(Sq), mk(Sk), mk(Sk)
        mask = torch.ones(B, 1, Sk, dtype=torch.bool, device=dev)
        o, lse = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, precision=ops.PREC_F16, out_fmt="f16")
        do = ops.make_planes(torch.randn(B * Sq, D, device=dev), "bwd")
        do = ops.Planes(do.hi[:, :D].contiguous(), None, B * Sq, D)
        r = timeit({"bwd": lambda: ops.attn_bwd_planes(q, k, v, o, do, lse, B, Sq, Sk, D, mask, H, 0.0, (None, None, None))})
        fl = 10.0 * B * Sq * Sk * D
        print(f"{name:11s} B{B} H{H} {Sq}x{Sk} dk{dk} " + "  ".join(f"{k}: {v:7.1f} us {fl / v / 1e6:6.0f} TF (algorithmic)" for k, v in r.items()), flush=True)
