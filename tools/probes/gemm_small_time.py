#!/usr/bin/env python
"""the decoder's own products (928 rows) one by one: time per launch of the forward three-pass product and of its dX, back to back (operands
hot in the L2) and with a 256 MB buffer swept between launches (operands cold, as inside a step).  Run once per library build:
BMT_LIB_PATH=.../libbmt_hip_nosmall.so python tools/probes/gemm_small_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402

DEV = "cuda"
SHAPES = [(928, 900, 300), (928, 300, 300), (928, 1024, 300), (928, 300, 1024), (928, 300, 600), (928, 1200, 300), (928, 300, 1200), (32, 1024, 300)]


def graph_ms(body, reps=5):
    """GPU time of one replay of a captured sequence (the host is out of the loop)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            body()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def timed(fn, n, flush=None):
    """us per launch inside a replayed graph of n launches; flush: a 256 MB sweep before every launch (its own time subtracted)"""
    def seq(with_fn):
        def body():
            for _ in range(n):
                if flush is not None:
                    flush.add_(1.0)
                if with_fn:
                    fn()
        return body
    t = graph_ms(seq(True))
    if flush is not None:
        t -= graph_ms(seq(False))
    return t * 1e3 / n


def main():
    print(f"library: {os.environ.get('BMT_LIB_PATH', 'default')}  small-tile outputs: {ops.SMALL_DX_OUTPUTS}")
    flush = torch.zeros(64 << 20, device=DEV)
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) * 0.1
        b = torch.randn(N, device=DEV)
        A = ops.as_planes(x, "x3")
        out = torch.empty(M, N, device=DEV)
        dy = ops.make_planes(torch.randn(M, N, device=DEV), "bwd")
        dx = torch.empty(M, K, device=DEV)
        f = lambda: ops.linear_fwd(A, W, b, out=out, precision=ops.PREC_BF16X3)
        g = lambda: ops.linear_dx(dy, W, out=dx)
        for _ in range(5):
            f(); g()
        print(f"{M:5d} x {N:5d} x {K:5d}   fwd x3 {timed(f, 50):6.1f} us hot {timed(f, 20, flush):6.1f} us cold    dX {timed(g, 50):6.1f} us hot {timed(g, 20, flush):6.1f} us cold")


if __name__ == "__main__":
    main()
