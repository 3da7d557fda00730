#!/usr/bin/env python
"""per-tile timeline of one dX = dY . W launch (k-major weight operand), experiment build (BMT_ALT_FLAGS=-DBMT_EXP, BMT_EXP=16):
where a 46 us backward product of the step spends its time.  usage: dx_timeline.py [M N_out K_in]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import _lib, ops  # noqa: E402

dev = "cuda"
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (8192, 1024, 1024)
dy = torch.randn(M, N, device=dev)
W = torch.randn(N, K, device=dev) * 0.03
P = ops.make_planes(dy, "bwd")
out = torch.empty(M, K, device=dev)


def timeit(f, iters=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


t = timeit(lambda: ops.linear_dx(P, W, out=out))
print(f"dX {M}x{K} (reduction {N}): {t:.1f} us = {2.0 * M * N * K / t / 1e6:.0f} TF   [BMT_EXP={os.environ.get('BMT_EXP', '0')} KM_PIPE={os.environ.get('BMT_GEMM_KM_PIPE', '0')}]")
if os.environ.get("BMT_EXP") == "16":
    lib = _lib.load()
    ntile = ((M + 127) // 128) * ((K + 127) // 128)
    buf = (C.c_ulonglong * (8 * ntile))()
    lib.bmt_dbg_read.argtypes = [C.c_void_p, C.c_int]
    assert lib.bmt_dbg_read(buf, 8 * ntile) == 0
    d = np.frombuffer(buf, dtype=np.uint64).reshape(ntile, 8).astype(np.int64)
    st, le, sg, dn = (d[:, i] for i in range(4))
    tot = np.median(dn - st)
    print(f"per tile (median, share of the tile's life): k-loop {np.median(le - st) / tot:.0%}  staging {np.median(sg - le) / tot:.0%}  stores {np.median(dn - sg) / tot:.0%}")
