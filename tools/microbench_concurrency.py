#!/usr/bin/env python
"""Do an HBM-write-bound GEMM (audio QKV projection, 25600x3072x128, x3, plane outputs) and an MFMA-bound GEMM (video FFN fc1,
8192x4096x1024, x3) overlap when issued on two streams?  Serial vs concurrent wall time per pair (HIP events on a fork/join)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import ops  # noqa: E402

dev = "cuda"
xa = ops.make_planes(torch.randn(25600, 128, device=dev), lo=True)[0]
Wa = torch.randn(3072, 128, device=dev) * 0.05
ba = torch.randn(3072, device=dev)
xv = ops.make_planes(torch.randn(8192, 1024, device=dev), lo=True)[0]
Wv = torch.randn(4096, 1024, device=dev) * 0.03
bv = torch.randn(4096, device=dev)
q = torch.randn(32, 800, 1024, device=dev)
pl = lambda t: (t.to(torch.bfloat16), (t - t.to(torch.bfloat16).float()).to(torch.bfloat16))
(qh, ql) = pl(q)
mask = torch.ones(32, 1, 800, dtype=torch.bool, device=dev)


def ga():
    return ops.linear_fwd_planes(xa, Wa, ba, want_lo=True)


def gv():
    return ops.linear_fwd_planes(xv, Wv, bv, want_lo=True, pad=True, relu=True)


def att():
    return ops.attn_fwd_bf16(qh, ql, qh, ql, qh, ql, mask, 4, precision=3)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ops.splitk_workspace(torch.device(dev))


def timed(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def both(f1, f2):
    def run():
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1):
            r1 = f1()
        with torch.cuda.stream(s2):
            r2 = f2()
        main.wait_stream(s1); main.wait_stream(s2)
        return r1, r2
    return run


for n1, f1, n2, f2 in (("audio QKV gemm (HBM-write bound)", ga, "video FFN gemm (MFMA bound)", gv),
                       ("audio QKV gemm", ga, "audio self-attention fwd x3", att),
                       ("video FFN gemm", gv, "audio self-attention fwd x3", att),
                       ("video FFN gemm", gv, "video FFN gemm", gv)):
    t1, t2 = timed(f1), timed(f2)
    ts = timed(lambda: (f1(), f2()))
    tc = timed(both(f1, f2))
    print(f"{n1}: {t1:.1f} us | {n2}: {t2:.1f} us | serial {ts:.1f} us | two streams {tc:.1f} us  ({ts / tc:.2f}x)", flush=True)
