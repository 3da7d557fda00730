#!/usr/bin/env python
"""GPU check + timing of the SPLIT attention backward (bmt_amd/csrc/exp/attn_bwd_split.hip, libbmt_exp.so): the dQ kernel that leaves P, dS
and a bf16 copy of q in workspaces, and the dK / dV kernel that is two plain products over them -- against the product's backward and an fp64
torch reference on the same rounded operands.

    bash bmt_amd/csrc/exp/build.sh && python tools/probes/attn_bwd_split_check.py > gpurun_out/attn_bwd_split_check.txt

Workspaces are filled with NaN first: whatever the kernels read without having written it shows up.  Masks: ragged prefixes (one batch
element at full length, one with whole 32- and 128-key tiles masked)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bmt_amd import _lib, ops  # noqa: E402
from bmt_amd._lib import AttnBwdBf16Args  # noqa: E402
from attn_bwd32_check import make_args, reference  # noqa: E402

EXP = C.CDLL(os.path.join(ROOT, "bmt_amd", "lib", "libbmt_exp.so"))
EXP.bmt_exp_attn_bwd_split.restype = C.c_int
EXP.bmt_exp_attn_bwd_split.argtypes = [C.POINTER(AttnBwdBf16Args), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
EXP.bmt_last_error.restype = C.c_char_p
dev = "cuda"
_p, _st = ops._p, ops._st


def case(B, H, Sq, Sk, dk, g, time_it=False, short=None, pmc=False):
    D = H * dk
    q = (torch.randn(B, Sq, D, generator=g)).to(dev).half()
    k = (torch.randn(B, Sk, D, generator=g) + 0.5).to(dev).half()
    v = torch.randn(B, Sk, D, generator=g).to(dev).half()
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    if B > 1 and short is not None:
        lens[1] = short                          # whole tiles masked
    mask = (torch.arange(Sk)[None, :] < lens[:, None]).view(B, 1, Sk).to(dev)
    rowscale = 10.0 ** (-1.0 - 5.0 * torch.rand(B, Sq, 1, generator=g))
    do = (torch.randn(B, Sq, D, generator=g) * rowscale).to(dev)
    o, lse = ops.attn_fwd_bf16(q, None, k, None, v, None, mask, H, precision=ops.PREC_F16)
    f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    dq_old, dk_old, dv_old = f32(B, Sq, D), f32(B, Sk, D), f32(B, Sk, D)
    nanf = lambda *s: torch.full(s, float("nan"), device=dev, dtype=torch.float32)
    dq_new, dk_new, dv_new = nanf(B, Sq, D), nanf(B, Sk, D), nanf(B, Sk, D)
    delta, doh = f32(B, H, Sq), torch.empty(B, Sq, D, device=dev, dtype=torch.bfloat16)
    keepm = ops._mask_args(mask, B, Sq, Sk)
    km = ops.attn_kmean(k, k.stride(1), k.stride(0), B, Sk, D, keepm, f16=True)
    a, keep = make_args(q, k, v, o, do, lse, mask, H, dq_old, dk_old, dv_old, delta, doh, km)
    _lib.check(ops.lib.bmt_attn_bwd_bf16(C.byref(a), _st()), "bmt_attn_bwd_bf16")
    pitch = (Sk + 31) // 32 * 32
    nanb = lambda *s: torch.full(s, float("nan"), device=dev, dtype=torch.bfloat16)
    wsn = max(Sq * pitch, (Sk + 127) // 128 * Sq * 128)
    Pws, dSws, Qb = nanb(B * H, wsn), nanb(B * H, wsn), nanb(B, Sq, D)
    a2, keep2 = make_args(q, k, v, o, do, lse, mask, H, dq_new, dk_new, dv_new, delta, doh, km)

    def run(which):
        rc = EXP.bmt_exp_attn_bwd_split(C.byref(a2), _p(Pws), _p(dSws), _p(Qb), pitch, which, None, _st())
        if rc != 0:
            raise RuntimeError(f"bmt_exp_attn_bwd_split rc={rc}: {EXP.bmt_last_error().decode()}")
    run(7 | LAYOUT)
    torch.cuda.synchronize()
    ref, refk, refv = reference(q, k, v, do, mask, H)
    rel = lambda x, r: float((x.double() - r).norm() / r.norm())
    rows = lambda x: float(((x.double() - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-300)).max())
    fin = all(bool(torch.isfinite(t).all()) for t in (dq_new, dk_new, dv_new))
    ok = fin and rel(dq_new, ref) <= max(1.5 * rel(dq_old, ref), 2e-3) and rel(dk_new, refk) <= max(1.5 * rel(dk_old, refk), 2e-3) \
        and rel(dv_new, refv) <= max(1.5 * rel(dv_old, refv), 2e-3)
    # the workspaces themselves, where they were written: P against the fp64 softmax on the valid keys
    print(f"  B{B} H{H} Sq{Sq} Sk{Sk} dk{dk}{'' if short is None else f' (one element with {short} keys)'}: finite {fin}  "
          f"dq old {rel(dq_old, ref):.2e} new {rel(dq_new, ref):.2e} (worst row {rows(dq_old):.2e} / {rows(dq_new):.2e});  "
          f"dk old {rel(dk_old, refk):.2e} new {rel(dk_new, refk):.2e};  dv old {rel(dv_old, refv):.2e} new {rel(dv_new, refv):.2e}  "
          f"{'OK' if ok else 'FAIL'}", flush=True)
    if not ok:
        for name, new, r in (("dq", dq_new, ref), ("dk", dk_new, refk), ("dv", dv_new, refv)):
            bad = (~torch.isfinite(new)).sum().item()
            e = (new.double().nan_to_num() - r).abs()
            bi = int(e.view(-1).argmax())
            print(f"      {name}: non-finite {bad}, max |err| {float(e.max()):.3e} at flat {bi} -> (b, s, d) = "
                  f"({bi // (new.shape[1] * D)}, {bi // D % new.shape[1]}, {bi % D}); |ref| max {float(r.abs().max()):.3e}", flush=True)
    # ---- the product's calling convention: dO as the bf16 plane, O as the saved fp16 plane (delta fused into the dQ kernels), gradients as
    # bf16 planes + bias column sums; old = bmt_attn_bwd_bf16, new = pipelined dQ kernel (fused delta) + dK / dV kernel
    of16 = o.half()
    bfp = lambda *s_: torch.zeros(*s_, device=dev, dtype=torch.bfloat16)

    def plane_args(gq, gk, gv, dbs):
        ap, keepp = make_args(q, k, v, o, do, lse, mask, H, dq_new, dk_new, dv_new, delta, doh, km)
        ap.O, ap.dO, ap.dQ, ap.dK, ap.dV = None, None, None, None, None
        ap.Of, ap.ldop, ap.bsop = _p(of16), of16.stride(1), of16.stride(0)
        ap.dQh, ap.dKh, ap.dVh = _p(gq), _p(gk), _p(gv)
        ap.gq_ld, ap.gq_bs, ap.gkv_ld, ap.gkv_bs = D, Sq * D, D, Sk * D
        ap.dbq, ap.dbk, ap.dbv = _p(dbs[0]), _p(dbs[1]), _p(dbs[2])
        return ap, keepp
    go = (bfp(B, Sq, D), bfp(B, Sk, D), bfp(B, Sk, D))
    gn = (bfp(B, Sq, D), bfp(B, Sk, D), bfp(B, Sk, D))
    dbo = [torch.zeros(D, device=dev) for _ in range(3)]
    dbn = [torch.zeros(D, device=dev) for _ in range(3)]
    apo, ko = plane_args(*go, dbo)
    apn, kn = plane_args(*gn, dbn)
    _lib.check(ops.lib.bmt_attn_bwd_bf16(C.byref(apo), _st()), "bmt_attn_bwd_bf16 (planes)")

    def runp(which):
        rc = EXP.bmt_exp_attn_bwd_split(C.byref(apn), _p(Pws), _p(dSws), _p(Qb), pitch, which, _p(bws), _st())
        if rc != 0:
            raise RuntimeError(f"bmt_exp_attn_bwd_split rc={rc}: {EXP.bmt_last_error().decode()}")
    Pws.fill_(float("nan")); dSws.fill_(float("nan")); Qb.fill_(float("nan"))
    bws = torch.full(((B * ((Sq + 127) // 128) + 2 * B * ((Sk + 127) // 128)) * D,), float("nan"), device=dev)
    refs = (ref, refk, refv)
    for kvbit, kvname in ((4, "4-wave dK/dV"), (16, "8-wave dK/dV")):
        for t_ in gn + tuple(dbn):
            t_.zero_()
        Pws.fill_(float("nan")); dSws.fill_(float("nan")); Qb.fill_(float("nan")); bws.fill_(float("nan"))
        runp(8 | kvbit | LAYOUT)
        torch.cuda.synchronize()
        okp = True
        line = []
        for name, po, pn, r, bo, bn in zip(("dq", "dk", "dv"), go, gn, refs, dbo, dbn):
            eo, en = rel(po.float(), r), rel(pn.float(), r)
            bref = r.sum(dim=(0, 1))
            ebo, ebn = float((bo.double() - bref).norm() / bref.norm()), float((bn.double() - bref).norm() / bref.norm())
            fin_ = bool(torch.isfinite(pn.float()).all() and torch.isfinite(bn).all())
            okp = okp and fin_ and en <= max(1.5 * eo, 4e-3) and (ebn <= max(2.0 * ebo, 4e-3) or name == "dk")
            line.append(f"{name} old {eo:.2e} new {en:.2e} (bias {ebo:.1e} / {ebn:.1e})")
        print(f"      planes + fused delta, pipelined dQ, {kvname}: " + ";  ".join(line) + f"  {'OK' if okp else 'FAIL'}", flush=True)
        ok = ok and okp
    # is the mean-key correction still worth its kernel with fp16 dS (11 significand bits)?  dQ without it:
    apn.kmean = None
    gn[0].zero_()
    runp(8 | LAYOUT)
    torch.cuda.synchronize()
    e_nokm = rel(gn[0].float(), ref)
    apo.kmean = None
    go[0].zero_()
    _lib.check(ops.lib.bmt_attn_bwd_bf16(C.byref(apo), _st()), "bmt_attn_bwd_bf16 (planes, no kmean)")
    torch.cuda.synchronize()
    print(f"      dq WITHOUT the mean-key correction: split {e_nokm:.2e}, two-kernel {rel(go[0].float(), ref):.2e}", flush=True)
    apn.kmean, apo.kmean = _p(km), _p(km)
    if pmc:          # a few launches of every kernel for rocprofv3 --pmc (tile-major workspaces)
        for _ in range(3):
            ops.lib.bmt_attn_bwd_bf16(C.byref(apo), _st())
            runp(8 | 256)
            runp(4 | 256)
            runp(16 | 256)
        torch.cuda.synchronize()
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
        t_all = timed(lambda: ops.lib.bmt_attn_bwd_bf16(C.byref(apo), _st()))
        fl = 2.0 * B * H * Sq * Sk * dk
        for lay in (0, 256):
            t_e, t_p, t_kv, t_new = timed(lambda: runp(1 | 2 | lay)), timed(lambda: runp(8 | lay)), timed(lambda: runp(4 | lay)), timed(lambda: runp(8 | 16 | lay))
            t_kv8 = timed(lambda: runp(16 | lay))
            print(f"      [{'tile-major' if lay else 'row-major '} workspaces; plane outputs] product backward {t_all:7.1f} us;  split: dQ-emit (delta kernel + first form) {t_e:7.1f}, "
                  f"pipelined with fused delta {t_p:7.1f} ({3 * fl / t_p / 1e6:6.1f} TF/s) + dK/dV 4-wave {t_kv:7.1f} / 8-wave {t_kv8:7.1f} ({2 * fl / t_kv8 / 1e6:6.1f} TF/s) = {t_new:7.1f} us together "
                  f"({5 * fl / t_new / 1e6:6.1f} TF/s algorithmic)", flush=True)
            if dk == 256 and "--probes" in sys.argv:
                probes = {3: "no DMA, no MFMA", 8: "no epilogue stores", 16: "no loop (prologue + epilogue only)", 24: "neither loop nor epilogue"}
                print("        dK/dV probes: " + "; ".join(f"{n} {timed(lambda: runp(4 | lay | (x << 12))):6.1f}" for x, n in probes.items()), flush=True)
                del probes[3]
                print("        dQ    probes: " + "; ".join(f"{n} {timed(lambda: runp(8 | lay | (x << 12))):6.1f}" for x, n in probes.items()), flush=True)
        runp(8 | 4 | LAYOUT)
    return ok


LAYOUT = 0 if "--row-major" in sys.argv else 256      # the product uses the tile-major workspaces


def main():
    g = torch.Generator().manual_seed(0)
    ok = True
    if "--pmc-case" in sys.argv:
        shape = (32, 4, 800, 800, 256) if "--shape" not in sys.argv else tuple(int(x) for x in sys.argv[sys.argv.index("--shape") + 1].split(","))
        return 0 if case(*shape, g, pmc=True) else 1
    cases = [((2, 4, 800, 800, 256), 300), ((2, 4, 256, 800, 256), 97), ((2, 4, 800, 256, 256), 128), ((3, 4, 29, 800, 256), 31),
             ((2, 2, 130, 45, 256), None), ((2, 8, 300, 333, 128), 100), ((2, 4, 256, 256, 256), 1)]
    for c, short in cases:
        try:
            ok &= case(*c, g, short=short)
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"  {c}: EXCEPTION {e}", flush=True)
    print("PARITY", "OK" if ok else "FAILED", flush=True)
    if "--no-time" not in sys.argv:
        shapes = [(32, 4, 800, 800, 256), (32, 4, 800, 256, 256), (32, 4, 256, 800, 256), (32, 4, 256, 256, 256), (64, 8, 800, 800, 128)]
        if "--batch-sweep" in sys.argv:
            shapes = [(8, 4, 800, 800, 256), (16, 4, 800, 800, 256), (32, 4, 800, 800, 256)]
        for c in shapes:
            case(*c, g, time_it=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
