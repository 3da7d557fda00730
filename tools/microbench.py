#!/usr/bin/env python
"""Kernel microbenchmarks on the config[1] shapes (HIP-event timed through the C ABI).
usage: microbench.py [gemm] [attn] [attnbwd] [--iters N]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import ops  # noqa: E402

DEV = "cuda"
ITERS = 5
for i, a in enumerate(sys.argv):
    if a == "--iters":
        ITERS = int(sys.argv[i + 1])


def timeit(fn, flops, name, iters=None):
    iters = iters or ITERS
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{name:58s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


def gemm_bench():
    shapes = [("ffn_v fc1   8192x4096x1024", 8192, 4096, 1024), ("v proj      8192x1024x1024", 8192, 1024, 1024),
              ("a proj     25600x1024x128 ", 25600, 1024, 128), ("a d2Q      25600x128x1024 ", 25600, 128, 1024),
              ("dec q       960x1024x300  ", 960, 1024, 300), ("generator   960x10000x300 ", 960, 10000, 300)]
    flt = os.environ.get("MB_FILTER")
    for name, M, N, K in shapes:
        if flt and flt not in name:
            continue
        x = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV)
        b = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        dy = torch.randn(M, N, device=DEV)
        for path in (True, False):
            ops.USE_PLANE_GEMM = path
            tag = "planes" if path else "fp32st"
            for prec in (3, 1):
                xin = ops.make_planes(x, lo=True)[0] if path else x
                timeit(lambda: ops.linear_fwd(xin, W, b, out=out, precision=prec), 2.0 * M * N * K, f"linear_fwd {tag} x{prec} {name}")
            dyP, dyT, _ = ops.grad_planes(dy)
            xT = ops.input_t(x)
            timeit(lambda: ops.linear_dx(dyP, W), 2.0 * M * N * K, f"linear_dx  {tag} x1 {name}")
            timeit(lambda: ops.linear_dw(dyT, xT), 2.0 * M * N * K, f"linear_dw  {tag} x1 {name}")
        ops.USE_PLANE_GEMM = True
        timeit(lambda: ops.make_planes(x, lo=True), 0.0, f"make_planes(hi,lo)      {name}")
        timeit(lambda: ops.grad_planes(dy), 0.0, f"grad_planes(hi,hiT)     {name}")


def attn_bench(bwd=False):
    B, H, dk = 32, 4, 256
    D = H * dk
    for name, Sq, Sk in (("A-self 800x800", 800, 800), ("V-self 256x256", 256, 256), ("A<-V 800x256", 800, 256),
                         ("V<-A 256x800", 256, 800), ("C<-A 30x800", 30, 800)):
        if os.environ.get("MB_FILTER") and os.environ["MB_FILTER"] not in name:
            continue
        q = torch.randn(B, Sq, D, device=DEV)
        k = torch.randn(B, Sk, D, device=DEV)
        v = torch.randn(B, Sk, D, device=DEV)
        mask = torch.ones(B, 1, Sk, dtype=torch.bool, device=DEV)
        mask[:, :, Sk - Sk // 5:] = False
        fl = 4.0 * B * Sq * Sk * D
        pl = lambda t: (t.to(torch.bfloat16), (t - t.to(torch.bfloat16).float()).to(torch.bfloat16))
        (qh, ql), (kh, kl), (vh, vl) = pl(q), pl(k), pl(v)
        if not bwd:
            for prec in (3, 1):
                if "v1" in sys.argv:
                    timeit(lambda: ops.attn_fwd(q, k, v, mask, H, precision=prec), fl, f"attn_fwd  fp32-in x{prec} {name}")
                timeit(lambda: ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask, H, precision=prec), fl, f"attn_fwd  planes  x{prec} {name}")
            timeit(lambda: ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, None, H, precision=3), fl, f"attn_fwd  planes  x3 nomask {name}")
        else:
            o, lse = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask, H, precision=3)
            do = torch.randn_like(o)
            if "v1" in sys.argv:
                timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, mask, H), 2.5 * fl, f"attn_bwd  fp32-in x1 {name}")
            timeit(lambda: ops.attn_bwd_bf16(qh, kh, vh, o, do, lse, mask, H), 2.5 * fl, f"attn_bwd  planes  x1 {name}")


if __name__ == "__main__":
    if "gemm" in sys.argv:
        gemm_bench()
    if "attn" in sys.argv:
        attn_bench()
    if "attnbwd" in sys.argv:
        attn_bench(bwd=True)
