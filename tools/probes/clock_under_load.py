#!/usr/bin/env python
"""engine clock and power while a kernel class runs back to back (rocm-smi polled from a side thread): are the MFMA loops priced against the
right clock?   python tools/probes/clock_under_load.py > gpurun_out/clock_under_load.txt"""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bmt_amd import ops  # noqa: E402

dev = "cuda"


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "fclk", "Power", "power"))]
    return " | ".join(keep)


def under_load(name, fn, seconds=1.8):
    fn()
    torch.cuda.synchronize()
    stop = False
    samples = []

    def poll():
        time.sleep(0.6)
        while not stop:
            samples.append(smi())
            time.sleep(0.3)

    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.time() - t0
    stop = True
    th.join()
    print(f"{name}: {n} launches, {dt / n * 1e6:.1f} us each", flush=True)
    for s_ in samples[:4]:
        print("    " + s_, flush=True)


print("idle: " + smi(), flush=True)
B, H, S, dk = 32, 4, 800, 256
D = H * dk
g = torch.Generator().manual_seed(1)
mk = lambda n: ops.make_planes(torch.randn(B * n, D, generator=g).to(dev), "all")
q, k, v = mk(S), mk(S), mk(S)
mask = torch.ones(B, 1, S, dtype=torch.bool, device=dev)
under_load("attention forward A-self (32-query kernel)",
           lambda: ops.attn_fwd_planes(q, k, v, B, S, S, D, mask, H, drop_p=0.0, site=3, precision=ops.PREC_F16, out_fmt="f16"))
x = torch.randn(8192, 1024, device=dev)
W = torch.randn(4096, 1024, device=dev) * 0.03
out = torch.empty(8192, 4096, device=dev)
A = ops.make_planes(x, ops.act_fmt(ops.PREC_F16W2))
under_load("two-plane fp16 GEMM 8192x4096x1024 (256x256 kernel)", lambda: ops.linear_fwd(A, W, None, out=out, precision=ops.PREC_F16W2))
a16, b16 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16), torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
under_load("torch.matmul bf16 8192^3 (vendor GEMM, yardstick)", lambda: torch.matmul(a16, b16), seconds=1.5)
