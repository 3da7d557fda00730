#!/usr/bin/env python
"""yardstick, not product: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the step's product shapes with 16-bit
operands and 16-bit output (no bias, no dropout, no planes) -- the ceiling the hand-written kernels in gemm_bf16.hip are compared with."""
import torch

dev = "cuda"
shapes = [("V qkv", 8192, 3072, 1024), ("V ffn1", 8192, 4096, 1024), ("V ffn2", 8192, 1024, 4096), ("V oproj", 8192, 1024, 1024),
          ("A qkv", 25600, 3072, 128), ("A oproj", 25600, 128, 1024), ("A ffn1", 25600, 512, 128), ("A<-V kv", 8192, 2048, 1024),
          ("gen", 928, 10000, 300), ("dW V ffn1", 4096, 1024, 8192), ("dW A qkv", 3072, 128, 25600)]
for name, M, N, K in shapes:
    for dt in (torch.bfloat16, torch.float16):
        a = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)
        for _ in range(5):
            y = a @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            y = a @ w.t()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{name:10s} {M:6d}x{N:5d}x{K:5d} {str(dt)[6:]:9s} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
