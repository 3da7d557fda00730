#!/usr/bin/env python
"""GPU check + timing of the experimental attention forward (tools/experiments/attn_fwd32.hip, libbmt_exp.so) against the product
16-query kernel and an fp32 torch reference on the same rounded operands (both kernels live in attention_bf16.hip; the experiment
library pins which one runs, whatever the shape -- tools/experiments/attn_fwd32.hip).

    bash tools/experiments/build.sh && python tools/probes/attn_fwd32_check.py [--no-time] > gpurun_out/attn_fwd32_check.txt
    ... --variants | --probe | --probe2 | --probe3 [--ragged]: loop variants / the probe copy with parts switched off (profiles/r02_q_*, r02_s_*)

The experiment entry takes the product's argument block, so the product's Python (ops.attn_fwd_bf16 / ops.attn_fwd_planes) drives both:
ops.lib is wrapped by a proxy that routes bmt_attn_fwd_bf16 to the experiment entry."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bmt_amd import _lib, ops  # noqa: E402

import subprocess
subprocess.run(["bash", os.path.join(ROOT, "tools", "experiments", "build.sh")], check=True, stdout=subprocess.DEVNULL)      # on demand
EXP = C.CDLL(os.path.join(ROOT, "tools", "experiments", "libbmt_exp.so"))
EXP.bmt_exp_attn_fwd32.restype = C.c_int
EXP.bmt_exp_attn_fwd32.argtypes = [C.POINTER(_lib.AttnFwdBf16Args), C.c_void_p]
EXP.bmt_last_error.restype = C.c_char_p
EXP.bmt_exp_set_variant.argtypes = [C.c_int]


class Proxy:
    """use_exp False: the 16-query kernel (experiment entry, variant 100); True: the 32-query kernel, variant Proxy.variant"""
    use_exp = False
    variant = 0

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        if name == "bmt_attn_fwd_bf16":
            def call(a, st):
                EXP.bmt_exp_set_variant(Proxy.variant if Proxy.use_exp else 100)
                rc = EXP.bmt_exp_attn_fwd32(a, st)
                if rc != 0:
                    raise RuntimeError(f"bmt_exp_attn_fwd32 rc={rc}: {EXP.bmt_last_error().decode()}")
                return 0
            return call
        return getattr(self._real, name)


ops.lib = Proxy(ops.lib)
dev = "cuda"


def ragged_mask(B, Sk, g):
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    if B > 1:
        lens[1] = max(1, Sk - 37)          # a partly valid last tiles
    return (torch.arange(Sk)[None, :] < lens[:, None]).view(B, 1, Sk).to(dev)


def reference(q, k, v, mask, H):
    B, Sq, D = q.shape
    dk = D // H
    qh, kh, vh = (x.float().view(B, -1, H, dk).transpose(1, 2) for x in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) / dk ** 0.5
    s = s.masked_fill(~mask.view(B, 1, 1, -1), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    o = torch.softmax(s, dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, Sq, D), lse


def run_raw(q, k, v, mask, H, prec, exp, drop_p=0.0):
    Proxy.use_exp = exp
    try:
        o, lse = ops.attn_fwd_bf16(q, None, k, None, v, None, mask, H, drop_p=drop_p, site=7, precision=prec)
    finally:
        Proxy.use_exp = False
    torch.cuda.synchronize()
    return o, lse


def check_case(B, H, Sq, Sk, dk, prec, g):
    D = H * dk
    dt = torch.float16 if prec == ops.PREC_F16 else torch.bfloat16
    q = (torch.randn(B, Sq, D, generator=g) * 1.5).to(dev).to(dt)
    k = (torch.randn(B, Sk, D, generator=g) * 1.5).to(dev).to(dt)
    v = torch.randn(B, Sk, D, generator=g).to(dev).to(dt)
    k[0, 5, :dk] *= 6.0                    # a spike: the rescale branch of the stale-maximum softmax fires late in some rows
    mask = ragged_mask(B, Sk, g)
    qa, ka, va = q, k, v                   # the C ABI takes 16-bit planes by pointer and element strides
    o_ref, lse_ref = reference(q, k, v, mask, H)
    o_old, lse_old = run_raw(qa, ka, va, mask, H, prec, False)
    o_new, lse_new = run_raw(qa, ka, va, mask, H, prec, True)
    e_old = float((o_old - o_ref).abs().max())
    e_new = float((o_new - o_ref).abs().max())
    e_on = float((o_new - o_old).abs().max())
    l_new = float((lse_new - lse_ref).abs().max())
    l_old = float((lse_old - lse_ref).abs().max())
    ok = e_new <= max(2.0 * e_old, 2e-3) and l_new <= max(2.0 * l_old, 1e-3) and bool(torch.isfinite(o_new).all())
    print(f"  B{B} H{H} Sq{Sq} Sk{Sk} dk{dk} {ops.prec_name(prec):5s}: |o-ref| old {e_old:.2e} new {e_new:.2e}  |new-old| {e_on:.2e}  "
          f"|lse-ref| old {l_old:.2e} new {l_new:.2e}  {'OK' if ok else 'FAIL'}", flush=True)
    # dropout: the same counter-based mask in both kernels -> same zeros
    od_old, _ = run_raw(qa, ka, va, mask, H, prec, False, drop_p=0.1)
    od_new, _ = run_raw(qa, ka, va, mask, H, prec, True, drop_p=0.1)
    same_zeros = bool(((od_old == 0) == (od_new == 0)).all())
    e_d = float((od_new - od_old).abs().max())
    print(f"      dropout 0.1: zero pattern {'same' if same_zeros else 'DIFFERENT'}, |new-old| {e_d:.2e}", flush=True)
    return ok and same_zeros and e_d <= max(4.0 * e_on, 2e-3)


def planes_case(B, H, Sq, Sk, dk, g):
    """the training path: fp16 planes in, bf16 + fp16 planes out (attn_fwd_planes), dropout on"""
    D = H * dk
    mk = lambda S: ops.make_planes(torch.randn(B * S, D, generator=g).to(dev), "all")
    q, k, v = mk(Sq), mk(Sk), mk(Sk)
    mask = ragged_mask(B, Sk, g)
    outs = []
    for exp in (False, True):
        Proxy.use_exp = exp
        try:
            o, lse = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, drop_p=0.1, site=3, precision=ops.PREC_F16, out_fmt="f16")
        finally:
            Proxy.use_exp = False
        torch.cuda.synchronize()
        outs.append((o.hi.float(), o.fh.float(), lse))
    eh = float((outs[0][0] - outs[1][0]).abs().max())
    ef = float((outs[0][1] - outs[1][1]).abs().max())
    el = float((outs[0][2] - outs[1][2]).abs().max())
    ok = eh <= 2e-2 and ef <= 4e-3 and el <= 1e-3
    print(f"  planes B{B} H{H} Sq{Sq} Sk{Sk} dk{dk}: |hi new-old| {eh:.2e}  |fp16 new-old| {ef:.2e}  |lse| {el:.2e}  {'OK' if ok else 'FAIL'}", flush=True)
    return ok


RAGGED = "--ragged" in sys.argv      # valid key lengths ~ U[Sk / 2, Sk] per batch element (the bench's synthetic batches) instead of full length


def time_one(B, H, Sq, Sk, dk, exp, drop_p=0.1, iters=20):
    """us per launch of the training-path call (fp16 planes in, planes out)"""
    D = H * dk
    g = torch.Generator().manual_seed(1)
    mk = lambda S: ops.make_planes(torch.randn(B * S, D, generator=g).to(dev), "all")
    q, k, v = mk(Sq), mk(Sk), mk(Sk)
    mask = ragged_mask(B, Sk, g) if RAGGED else torch.ones(B, 1, Sk, dtype=torch.bool, device=dev)
    Proxy.use_exp = exp
    try:
        f = lambda: ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, drop_p=drop_p, site=3, precision=ops.PREC_F16, out_fmt="f16")
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    finally:
        Proxy.use_exp = False


def variants():
    """every (DMA placement, priority) variant of the d_k 256 fp16 kernel: parity on one ragged case, then the two shapes it is meant for"""
    g = torch.Generator().manual_seed(0)
    shapes = [("A-self", 32, 4, 800, 800, 256), ("A<-V", 32, 4, 800, 256, 256), ("V<-A", 32, 4, 256, 800, 256)]
    for name, *sh in shapes:
        print(f"  old kernel {name}: dropout 0.1 {time_one(*sh, False):7.1f} us   no dropout {time_one(*sh, False, drop_p=0.0):7.1f} us", flush=True)
    for v in range(8):
        Proxy.variant = v
        print(f"variant {v} (DMA placement {v % 4}, setprio {'on' if v < 4 else 'off'}):", flush=True)
        ok = check_case(2, 4, 800, 800, 256, ops.PREC_F16, g)
        line = "  " + ("parity OK  " if ok else "PARITY FAILED  ")
        for name, *sh in shapes:
            line += f"{name}: {time_one(*sh, True):7.1f} us (no dropout {time_one(*sh, True, drop_p=0.0):7.1f})   "
        print(line, flush=True)
    Proxy.variant = 0


def probe():
    """what each part of the 32-query loop costs: the probe copy of the kernel with parts switched off (timing only), two / one
    workgroup(s) per CU"""
    shapes = [("A-self", 32, 4, 800, 800, 256), ("A<-V", 32, 4, 800, 256, 256), ("V<-A", 32, 4, 256, 800, 256)]
    names = {0: "everything on", 1: "no DMA in the loop", 2: "no softmax arithmetic", 3: "no DMA, no softmax", 5: "no DMA, no wait / barrier",
             7: "MFMA + fragment reads only"}
    for base, label in ((200, "two workgroups per CU"), (300, "ONE workgroup per CU (40 KB of LDS padding)")):
        print(label + ":", flush=True)
        for xp in (0, 1, 2, 3, 5, 7):
            Proxy.variant = base + xp
            line = f"  XP {xp} ({names[xp]:28s}): "
            for name, *sh in shapes:
                line += f"{name} {time_one(*sh, True, drop_p=0.0):7.1f} us   "
            print(line, flush=True)
    Proxy.variant = 0
    for name, *sh in shapes:
        print(f"  product kernel {name}: no dropout {time_one(*sh, True, drop_p=0.0):7.1f} us, dropout 0.1 {time_one(*sh, True):7.1f} us", flush=True)


def probe2():
    """prefetch depth of the fragment reads (1 .. 4 MFMAs ahead) and MFMA ordering (PV key-half-major: no two consecutive MFMAs on
    one accumulator; S on two accumulators) of the probe copy: whole loop (XP 0) and MFMA + fragment reads only (XP 7)"""
    shapes = [("A-self", 32, 4, 800, 800, 256), ("A<-V", 32, 4, 800, 256, 256)]
    for ord_, depth in ((0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 4), (2, 2), (3, 2), (3, 3)):
        line = f"  order {ord_} depth {depth}: "
        for xp in (0, 7):
            Proxy.variant = 400 + 100 * ord_ + 10 * depth + xp
            line += ("whole loop " if xp == 0 else "| MFMA + reads only ")
            for name, *sh in shapes:
                line += f"{name} {time_one(*sh, True, drop_p=0.0):7.1f} us  "
        print(line, flush=True)
    Proxy.variant = 0


def time_case(name, B, H, Sq, Sk, dk, iters=20):
    D = H * dk
    g = torch.Generator().manual_seed(1)
    mk = lambda S: ops.make_planes(torch.randn(B * S, D, generator=g).to(dev), "all")
    q, k, v = mk(Sq), mk(Sk), mk(Sk)
    mask = torch.ones(B, 1, Sk, dtype=torch.bool, device=dev)
    flops = 4.0 * B * H * Sq * Sk * dk
    res = []
    for exp in (False, True):
        Proxy.use_exp = exp
        try:
            f = lambda: ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, drop_p=0.1, site=3, precision=ops.PREC_F16, out_fmt="f16")
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / iters)
        finally:
            Proxy.use_exp = False
    print(f"  {name:8s} B{B} H{H} Sq{Sq} Sk{Sk} dk{dk}: old {res[0] * 1e3:7.1f} us ({flops / res[0] / 1e9:6.1f} TF/s)   new {res[1] * 1e3:7.1f} us "
          f"({flops / res[1] / 1e9:6.1f} TF/s)   x{res[0] / res[1]:.2f}", flush=True)


def main():
    if "--pmc-case" in sys.argv:         # a few launches of each kernel on the A-self shape (for a rocprofv3 --pmc pass)
        time_case("A-self", 32, 4, 800, 800, 256, iters=2)
        return 0
    if "--probe3" in sys.argv:          # is the fixed part of the kernel its output? (XP bit 8: no plane stores)
        shapes = [("A-self", 32, 4, 800, 800, 256), ("A<-V", 32, 4, 800, 256, 256), ("V<-A", 32, 4, 256, 800, 256)]
        for xp, label in ((0, "everything on"), (8, "no output stores"), (7, "MFMA + fragment reads only"), (15, "MFMA + reads, no output stores")):
            Proxy.variant = 200 + xp
            line = f"  XP {xp:2d} ({label:32s}): "
            for name, *sh in shapes:
                line += f"{name} {time_one(*sh, True, drop_p=0.0):7.1f} us   "
            print(line, flush=True)
        Proxy.variant = 0
        return 0
    if "--probe2" in sys.argv:
        probe2()
        return 0
    if "--probe" in sys.argv:
        probe()
        return 0
    if "--variants" in sys.argv:
        variants()
        return 0
    t0 = time.time()
    g = torch.Generator().manual_seed(0)
    ok = True
    print("parity (fp32 output, against torch fp32 on the same rounded operands):", flush=True)
    for case in [(2, 4, 800, 800, 256), (2, 4, 256, 800, 256), (2, 4, 800, 256, 256), (3, 4, 29, 800, 256), (2, 2, 130, 45, 256), (2, 8, 300, 333, 128)]:
        for prec in (ops.PREC_F16, ops.PREC_BF16):
            try:
                ok &= check_case(*case, prec, g)
            except Exception as e:  # noqa: BLE001
                ok = False
                print(f"  {case} {ops.prec_name(prec)}: EXCEPTION {e}", flush=True)
    print("training path (planes out, dropout 0.1):", flush=True)
    for case in [(2, 4, 800, 800, 256), (2, 4, 256, 800, 256), (2, 8, 300, 333, 128)]:
        try:
            ok &= planes_case(*case, g)
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"  planes {case}: EXCEPTION {e}", flush=True)
    print("PARITY", "OK" if ok else "FAILED", f"({time.time() - t0:.0f} s)", flush=True)
    if "--no-time" not in sys.argv:
        print("timing (configs[1] encoder shapes, fp16 planes, dropout 0.1, planes out):", flush=True)
        time_case("A-self", 32, 4, 800, 800, 256)
        time_case("V-self", 32, 4, 256, 256, 256)
        time_case("A<-V", 32, 4, 800, 256, 256)
        time_case("V<-A", 32, 4, 256, 800, 256)
        time_case("dec C<-A", 32, 4, 29, 800, 256)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
