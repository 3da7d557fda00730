#!/usr/bin/env python
"""GPU check + timing of the experimental k-major 256 x 256 GEMM kernel (bmt_amd/csrc/exp/gemm_wide_km.hip, libbmt_exp.so) against the
product's dX path (ops.linear_dx -> bmt_gemm_bf16 with a k-major B operand -> the register-staged 128-row loop) and a float64 reference
on the same bf16-rounded operands.  NOT yet run (written after round 2's GPU budget was spent; the weight operand's lane algebra is checked
on the CPU by gemm_wide_km_layout.py).

    bash bmt_amd/csrc/exp/build.sh && python tools/probes/gemm_wide_km_check.py > gpurun_out/gemm_wide_km_check.txt

The experiment entry takes bmt_gemm_bf16's argument block: ops.lib is wrapped by a proxy that routes the eligible k-major calls to it while
`use_exp` is set, so the product's Python (plane conversion, epilogue arguments) drives both."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bmt_amd import _lib, ops  # noqa: E402

EXP = C.CDLL(os.path.join(ROOT, "bmt_amd", "lib", "libbmt_exp.so"))
EXP.bmt_exp_gemm_wide_km.restype = C.c_int
EXP.bmt_exp_gemm_wide_km.argtypes = [C.POINTER(_lib.GemmBf16Args), C.c_void_p]
EXP.bmt_last_error.restype = C.c_char_p
dev = "cuda"


class Proxy:
    use_exp = False
    routed = 0

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if name != "bmt_gemm_bf16":
            return real

        def call(a, st):
            args = a._obj if hasattr(a, "_obj") else a
            if Proxy.use_exp and args.b_kmajor and not args.a_kmajor and args.N >= 256 and args.splitk <= 1:
                Proxy.routed += 1
                rc = EXP.bmt_exp_gemm_wide_km(a, st)
                if rc != 0:
                    raise RuntimeError(f"bmt_exp_gemm_wide_km rc={rc}: {EXP.bmt_last_error().decode()}")
                return 0
            return real(a, st)
        return call


ops.lib = Proxy(ops.lib)


def dx(dy, W, exp, **epi):
    Proxy.use_exp = exp
    try:
        out = ops.linear_dx(dy, W, **epi)
    finally:
        Proxy.use_exp = False
    torch.cuda.synchronize()
    return out


def case(M, N_out, K_in, g, time_it=False):
    dy = (torch.randn(M, N_out, generator=g) * 1e-3).to(dev)
    W = (torch.randn(N_out, K_in, generator=g) * 0.03).to(dev)
    routed0 = Proxy.routed
    old, new = dx(dy, W, False), dx(dy, W, True)
    took = Proxy.routed > routed0
    ref = dy.bfloat16().double() @ W.bfloat16().double()
    rel = lambda x: float((x.double() - ref).norm() / ref.norm())
    ok = took and bool(torch.isfinite(new).all()) and rel(new) <= max(1.5 * rel(old), 1e-5)
    print(f"  dX[{M} x {K_in}] = dY[{M} x {N_out}] . W: |old-ref| {rel(old):.2e}  |new-ref| {rel(new):.2e}  max|new-old| {float((new - old).abs().max()):.2e}  "
          f"{'OK' if ok else ('FAIL' if took else 'NOT ROUTED')}", flush=True)
    if time_it:
        res = []
        for exp in (False, True):
            out = torch.empty(M, K_in, device=dev)
            P = ops.as_planes(dy, "bwd")
            f = lambda: dx(P, W, exp, out=out)
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            Proxy.use_exp = exp
            e0.record()
            for _ in range(20):
                ops.linear_dx(P, W, out=out)
            e1.record()
            Proxy.use_exp = False
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
        fl = 2.0 * M * N_out * K_in
        print(f"      product k-major loop {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF/s)   k-major 256 x 256 kernel {res[1]:7.1f} us ({fl / res[1] / 1e6:6.1f} TF/s)   "
              f"x{res[0] / res[1]:.2f}", flush=True)
    return ok


def main():
    g = torch.Generator().manual_seed(0)
    ok = True
    for c in [(512, 256, 256), (300, 1024, 1024), (1000, 320, 1024), (2048, 4096, 1024), (777, 1024, 4096)]:
        try:
            ok &= case(*c, g)
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"  {c}: EXCEPTION {e}", flush=True)
    print("PARITY", "OK" if ok else "FAILED", flush=True)
    if "--no-time" not in sys.argv:
        # the step's dX shapes (configs[1], B = 32): video stream 8192 rows, audio stream 25600 rows
        for c in [(8192, 1024, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (8192, 3072, 1024), (25600, 1024, 1024)]:
            case(*c, g, time_it=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
