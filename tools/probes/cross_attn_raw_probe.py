"""What the attention part of the decoder's cross-attention costs when it runs against the RAW audio memory (the key / value
projections reassociated away), with the kernels the library has today: the heads become extra queries of ONE head of width d_memory = 128
(Sq = H x 30 = 120, K = V = the memory plane), against today's 4 heads x 30 queries x d_k 256 over the projected K / V planes.
    python tools/probes/cross_attn_raw_probe.py        (GPU; prints microseconds per launch, forward and backward)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402

DEV = "cuda"
B, Sc, Sk, H = 32, 30, 800, 4
torch.manual_seed(0)
lens = torch.randint(Sk // 2, Sk + 1, (B,))
mask = (torch.arange(Sk)[None, :] < lens[:, None]).view(B, 1, Sk).to(DEV)


def planes(rows, cols, fmt):
    return ops.make_planes(torch.randn(rows, cols, device=DEV) * 0.5, fmt)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, Sq, Hh, D, Dkv in (("today: 4 heads x 30 queries, d_k 256, projected K / V [25600 x 1024]", Sc, H, 1024, 1024),
                             ("raw memory: 1 head x 120 queries, d 128, K = V = memory [25600 x 128]", Sc * H, 1, 128, 128)):
    q = planes(B * Sq, D, "f16")
    k = planes(B * Sk, Dkv, "f16")
    v = k if Hh == 1 else planes(B * Sk, Dkv, "f16")
    out = {}

    def fwd():
        out["o"], out["lse"] = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, Hh, precision=ops.PREC_F16, out_fmt="x3")
    t_f = timeit(fwd)
    do = ops.make_planes(torch.randn(B * Sq, D, device=DEV) * 0.1, "bwd")

    def bwd():
        ops.attn_bwd_planes(q, k, v, out["o"], do, out["lse"], B, Sq, Sk, D, mask, Hh, 0.0, (None, None, None))
    t_b = timeit(bwd)
    print(f"{name}\n    forward {t_f:7.1f} us   backward {t_b:7.1f} us (incl. the mean-key launch)", flush=True)
