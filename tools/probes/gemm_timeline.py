#!/usr/bin/env python
"""per-tile timeline of one GEMM launch (alt build with -DBMT_EXP, env BMT_EXP=16): s_memtime stamps at tile start, end of the k-loop,
accumulators staged, stores issued; prints the phase durations and how the workgroups line up in time."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import _lib, ops  # noqa: E402

dev = "cuda"
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (8192, 4096, 1024)
prec = int(sys.argv[4]) if len(sys.argv) > 4 else ops.PREC_F16
x = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) * 0.03
out = torch.empty(M, N, device=dev)
A = ops.make_planes(x, ops.act_fmt(prec))
ops.weight_planes(W, ops.weight_fmt(prec))
for _ in range(3):
    ops.linear_fwd(A, W, None, out=out, precision=prec)
torch.cuda.synchronize()
lib = _lib.load()
tsz = 256 if os.environ.get('BMT_GEMM_WIDE') == '1' else 128
ntile = ((M + tsz - 1) // tsz) * ((N + tsz - 1) // tsz)
buf = (C.c_ulonglong * (8 * ntile))()
lib.bmt_dbg_read.argtypes = [C.c_void_p, C.c_int]
assert lib.bmt_dbg_read(buf, 8 * ntile) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(ntile, 8).astype(np.int64)
t0 = d[:, 0].min()
st, le, sg, dn = (d[:, i] - t0 for i in range(4))
hw = d[:, 4]
xcc = hw >> 32
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
f = 100e6          # s_memtime ticks at the 100 MHz reference clock on gfx9
us = lambda v: v / f * 1e6
print(f"{ntile} tiles, {len(np.unique(cuid))} distinct CUs; span {us(dn.max()):.1f} us")
print("(256 x 256 kernel: stamps are start, first K-tile landed, K loop done, stores issued)" if tsz == 256 else "")
print(f"k-loop   : median {us(np.median(le - st)):.2f} us  p10 {us(np.percentile(le - st, 10)):.2f}  p90 {us(np.percentile(le - st, 90)):.2f}")
print(f"stage acc: median {us(np.median(sg - le)):.2f} us  p90 {us(np.percentile(sg - le, 90)):.2f}")
print(f"stores   : median {us(np.median(dn - sg)):.2f} us  p10 {us(np.percentile(dn - sg, 10)):.2f}  p90 {us(np.percentile(dn - sg, 90)):.2f}")
order = np.argsort(st)
print("start times of tiles (sorted), every 128th:", " ".join(f"{us(st[i]):.1f}" for i in order[::128]))
# per CU: the sequence of (start, loop end, done)
for c in (np.unique(cuid)[:3] if tsz == 128 else []):
    idx = np.where(cuid == c)[0]
    idx = idx[np.argsort(st[idx])]
    print(f"CU {c}: " + "  ".join(f"[{us(st[i]):.1f} {us(le[i]):.1f} {us(dn[i]):.1f}]" for i in idx))
