#!/usr/bin/env python
"""GPU check + timing of the experimental dQ and dK/dV kernels of the attention backward (bmt_amd/csrc/exp/attn_bwd32.hip, libbmt_exp.so) against
the product's backward and an fp64 torch reference on the same rounded operands.  NOT yet run (written after round 2's GPU budget was
spent; the kernel's lane algebra is checked on the CPU by attn_bwd32_layout.py).

    bash bmt_amd/csrc/exp/build.sh && python tools/probes/attn_bwd32_check.py > gpurun_out/attn_bwd32_check.txt

Protocol: forward on fp16 planes (product) -> o, lse; product backward with fp32 outputs (fills the delta and bf16-dO workspaces) -> dq_old;
experiment entry on the same argument block with its own dQ buffer -> dq_new; fp64 autograd of softmax(QK^T/sqrt(dk))V on the fp16-rounded
q / k / v -> dq_ref.  The upstream gradient spans five decades from row to row (the per-row power-of-two scale of the new kernel)."""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bmt_amd import _lib, ops  # noqa: E402
from bmt_amd._lib import AttnBwdBf16Args  # noqa: E402

EXP = C.CDLL(os.path.join(ROOT, "bmt_amd", "lib", "libbmt_exp.so"))
EXP.bmt_exp_attn_bwd_dq32.restype = C.c_int
EXP.bmt_exp_attn_bwd_dq32.argtypes = [C.POINTER(AttnBwdBf16Args), C.c_void_p, C.c_void_p]
EXP.bmt_exp_attn_bwd_dkv32.restype = C.c_int
EXP.bmt_exp_attn_bwd_dkv32.argtypes = [C.POINTER(AttnBwdBf16Args), C.c_void_p, C.c_int, C.c_void_p]
EXP.bmt_last_error.restype = C.c_char_p
dev = "cuda"
_p, _st = ops._p, ops._st


def make_args(q, k, v, o, do, lse, mask, H, dq, dk_, dv, delta, doh, km):
    B, Sq, D = q.shape
    Sk = k.shape[1]
    dk = D // H
    keep, mptr, mbs, mqs = ops._mask_args(mask, B, Sq, Sk)
    a = AttnBwdBf16Args(Qh=_p(q), Kh=_p(k), Vh=_p(v), O=_p(o), dO=_p(do), lse=_p(lse), dQ=_p(dq), dK=_p(dk_), dV=_p(dv),
                        delta_ws=_p(delta), dOh_ws=_p(doh), ldq=q.stride(1), ldk=k.stride(1), ldv=v.stride(1), ldo=o.stride(1),
                        bsq=q.stride(0), bsk=k.stride(0), bsv=v.stride(0), bso=o.stride(0), dkv_ld=dk_.stride(1), dkv_bs=dk_.stride(0),
                        mask=mptr, mask_bs=mbs, mask_qs=mqs, B=B, H=H, Sq=Sq, Sk=Sk, dk=dk, scale=1.0 / math.sqrt(dk), drop_p=0.0,
                        kmean=_p(km), qkv_f16=1)
    return a, keep


def reference(q, k, v, do, mask, H):
    B, Sq, D = q.shape
    dk = D // H
    qd, kd, vd = (x.double().detach().requires_grad_(True) for x in (q, k, v))
    qh = qd.view(B, Sq, H, dk).transpose(1, 2)
    kh, vh = (x.view(B, -1, H, dk).transpose(1, 2) for x in (kd, vd))
    s = (qh @ kh.transpose(-1, -2)) / dk ** 0.5
    s = s.masked_fill(~mask.view(B, 1, 1, -1), float("-inf"))
    o = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(B, Sq, D)
    (o * do.double()).sum().backward()
    return qd.grad, kd.grad, vd.grad


def case(B, H, Sq, Sk, dk, g, time_it=False):
    D = H * dk
    q = (torch.randn(B, Sq, D, generator=g)).to(dev).half()
    k = (torch.randn(B, Sk, D, generator=g) + 0.5).to(dev).half()       # a common component: the mean-key correction has work to do
    v = torch.randn(B, Sk, D, generator=g).to(dev).half()
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = (torch.arange(Sk)[None, :] < lens[:, None]).view(B, 1, Sk).to(dev)
    rowscale = 10.0 ** (-1.0 - 5.0 * torch.rand(B, Sq, 1, generator=g))
    do = (torch.randn(B, Sq, D, generator=g) * rowscale).to(dev)
    o, lse = ops.attn_fwd_bf16(q, None, k, None, v, None, mask, H, precision=ops.PREC_F16)
    f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    dq_old, dq_new, dk_, dv = f32(B, Sq, D), torch.zeros(B, Sq, D, device=dev), f32(B, Sk, D), f32(B, Sk, D)
    dk_new, dv_new, kq = torch.zeros(B, Sk, D, device=dev), torch.zeros(B, Sk, D, device=dev), f32(B, H, Sq)
    delta, doh = f32(B, H, Sq), torch.empty(B, Sq, D, device=dev, dtype=torch.bfloat16)
    keepm = ops._mask_args(mask, B, Sq, Sk)
    km = ops.attn_kmean(k, k.stride(1), k.stride(0), B, Sk, D, keepm, f16=True)
    a, keep = make_args(q, k, v, o, do, lse, mask, H, dq_old, dk_, dv, delta, doh, km)
    _lib.check(ops.lib.bmt_attn_bwd_bf16(C.byref(a), _st()), "bmt_attn_bwd_bf16")
    a2, keep2 = make_args(q, k, v, o, do, lse, mask, H, dq_new, dk_, dv, delta, doh, km)
    rc = EXP.bmt_exp_attn_bwd_dq32(C.byref(a2), _p(kq), _st())
    if rc != 0:
        raise RuntimeError(f"bmt_exp_attn_bwd_dq32 rc={rc}: {EXP.bmt_last_error().decode()}")
    a3, keep3 = make_args(q, k, v, o, do, lse, mask, H, dq_new, dk_new, dv_new, delta, doh, km)
    dk_two, dv_two = torch.zeros_like(dk_new), torch.zeros_like(dv_new)
    a4, keep4 = make_args(q, k, v, o, do, lse, mask, H, dq_new, dk_two, dv_two, delta, doh, km)
    for args_, two in ((a3, 0), (a4, 1)):          # one pass (spills at d_k 256) and two passes over the queries
        rc = EXP.bmt_exp_attn_bwd_dkv32(C.byref(args_), _p(kq), two, _st())
        if rc != 0:
            raise RuntimeError(f"bmt_exp_attn_bwd_dkv32 rc={rc}: {EXP.bmt_last_error().decode()}")
    torch.cuda.synchronize()
    ref, refk, refv = reference(q, k, v, do, mask, H)
    rel = lambda x, r=None: float((x.double() - (ref if r is None else r)).norm() / (ref if r is None else r).norm())
    rows = lambda x: float(((x.double() - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-300)).max())
    ok = bool(torch.isfinite(dq_new).all()) and rel(dq_new) <= max(1.5 * rel(dq_old), 2e-3)
    print(f"  B{B} H{H} Sq{Sq} Sk{Sk} dk{dk}: |dq-ref|/|ref| old {rel(dq_old):.2e} new {rel(dq_new):.2e}; worst row old {rows(dq_old):.2e} "
          f"new {rows(dq_new):.2e}  {'OK' if ok else 'FAIL'}", flush=True)
    okk = bool(torch.isfinite(dk_new).all() and torch.isfinite(dv_new).all()) and rel(dk_new, refk) <= max(1.5 * rel(dk_, refk), 2e-3) \
        and rel(dv_new, refv) <= max(1.5 * rel(dv, refv), 2e-3)
    print(f"      |dk-ref|/|ref| old {rel(dk_, refk):.2e} new {rel(dk_new, refk):.2e};  |dv-ref|/|ref| old {rel(dv, refv):.2e} new {rel(dv_new, refv):.2e}  "
          f"{'OK' if okk else 'FAIL'}", flush=True)
    same = float((dk_two - dk_new).abs().max()), float((dv_two - dv_new).abs().max())
    ok2 = same[0] <= 1e-6 * float(dk_new.abs().max()) + 1e-30 and same[1] <= 1e-6 * float(dv_new.abs().max()) + 1e-30
    print(f"      two-pass vs one-pass kernel: max |dk| diff {same[0]:.2e}, |dv| diff {same[1]:.2e}  {'OK' if ok2 else 'FAIL'}", flush=True)
    ok = ok and okk and ok2
    if time_it:
        def timed(f, iters=20):
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
        t_all = timed(lambda: ops.lib.bmt_attn_bwd_bf16(C.byref(a), _st()))
        t_new = timed(lambda: EXP.bmt_exp_attn_bwd_dq32(C.byref(a2), _p(kq), _st()))
        t_kv = timed(lambda: EXP.bmt_exp_attn_bwd_dkv32(C.byref(a3), _p(kq), 0, _st()))
        t_kv2 = timed(lambda: EXP.bmt_exp_attn_bwd_dkv32(C.byref(a4), _p(kq), 1, _st()))
        fl = 2.0 * B * H * Sq * Sk * dk
        print(f"      product backward (delta + dQ + dK/dV kernels) {t_all:7.1f} us; experimental dQ {t_new:7.1f} us ({3 * fl / t_new / 1e6:6.1f} TF/s), "
              f"dK/dV one pass {t_kv:7.1f} us ({4 * fl / t_kv / 1e6:6.1f} TF/s algorithmic), two passes {t_kv2:7.1f} us ({4 * fl / t_kv2 / 1e6:6.1f}); "
              f"the product's kernels one by one: rocprofv3 --kernel-trace of this script", flush=True)
    return ok


def main():
    g = torch.Generator().manual_seed(0)
    ok = True
    for c in [(2, 4, 800, 800, 256), (2, 4, 256, 800, 256), (2, 4, 800, 256, 256), (3, 4, 29, 800, 256), (2, 2, 130, 45, 256), (2, 8, 300, 333, 128)]:
        try:
            ok &= case(*c, g)
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"  {c}: EXCEPTION {e}", flush=True)
    print("PARITY", "OK" if ok else "FAILED", flush=True)
    if "--no-time" not in sys.argv:
        for c in [(32, 4, 800, 800, 256), (32, 4, 800, 256, 256), (32, 4, 256, 800, 256)]:
            case(*c, g, time_it=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
