#!/usr/bin/env python
"""does a kernel slow down under sustained load?  one GEMM shape timed in bursts of 20 launches over ~3 s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmt_amd import ops
M, N, K = 8192, 1024, 1024
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
xin = ops.make_planes(x, lo=True)[0]
fn = lambda: ops.linear_fwd(xin, W, b, out=out, precision=3)
fn(); torch.cuda.synchronize()
t0 = time.time(); res = []
while time.time() - t0 < 3.0:
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    res.append(s.elapsed_time(e) * 1e3 / 20)
print("bursts", len(res), "first 5:", [round(r, 1) for r in res[:5]], "last 5:", [round(r, 1) for r in res[-5:]], "max", round(max(res), 1))
# now interleave with a big memory-bound kernel to evict caches
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
res2 = []
for _ in range(30):
    big.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    res2.append(s.elapsed_time(e) * 1e3)
print("cold (after a 512 MB memset):", [round(r, 1) for r in res2[:10]])
