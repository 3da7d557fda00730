#!/usr/bin/env python
"""GPU: the encoder's four attention backward launches of configs[1] (B 32, H 4, d_k 256, ragged lengths), each split form timed with device
events: "recompute" (round 6: the key side rebuilds P / dS), "emit" (rounds 3-5: P / dS through HBM workspaces), and the forward for scale.
Packed rows (the step's layout) when --packed, else padded with key masks.  Prints us per launch and the executed FLOPs' rate."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402

DEV = "cuda:0"


def planes16(t):
    pl = ops.make_planes(t, "f16")
    return ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--order", default="random", choices=("random", "desc", "asc", "equal", "snake"),
                    help="the samples' lengths as drawn, sorted (longest / shortest sample first), or all equal to their rms (same executed FLOPs)")
    ap.add_argument("--forms", default="recompute,emit,two-kernel")
    args = ap.parse_args()
    B, H, dk = args.B, 4, 256
    D = H * dk
    g = torch.Generator().manual_seed(7)
    Lv = torch.randint(128, 257, (B,), generator=g)
    Lv[0] = 256
    if args.order == "desc":
        Lv = torch.sort(Lv, descending=True).values
    elif args.order == "asc":
        Lv = torch.sort(Lv).values
    elif args.order == "snake":      # samples 4x .. 4x + 3 (the (batch, head) range one XCD works through) with balanced sums, longest first
        sv = torch.sort(Lv, descending=True).values
        n8 = B // 8
        Lv = torch.stack([torch.sort(torch.stack([sv[r * 8 + (g if r % 2 == 0 else 7 - g)] for r in range(n8)]), descending=True).values for g in range(8)]).reshape(-1)
    elif args.order == "equal":
        Lv = torch.full_like(Lv, int(round(float((Lv.double() ** 2).mean().sqrt()))))
    La = torch.round(Lv.float() * 800 / 256).long()
    shapes = [("A-self", 800, 800, La, La), ("V-self", 256, 256, Lv, Lv), ("A<-V", 800, 256, La, Lv), ("V<-A", 256, 800, Lv, La)]
    for name, Sq, Sk, Lq, Lk in shapes:
        q = (torch.randn(B * Sq, D, generator=g) * 0.7).to(DEV)
        k = (torch.randn(B * Sk, D, generator=g) * 0.7 + 0.3).to(DEV)
        v = torch.randn(B * Sk, D, generator=g).to(DEV)
        do = (torch.randn(B * Sq, D, generator=g) * 1e-3)
        qmask = (torch.arange(Sq)[None, :] < Lq[:, None])
        do = (do.view(B, Sq, D) * qmask[..., None]).view(B * Sq, D).to(DEV)          # padded query rows carry no gradient (the encoder's pattern)
        mask = (torch.arange(Sk)[None, :] < Lk[:, None]).view(B, 1, Sk).to(DEV)
        qp, kp, vp = planes16(q), planes16(k), planes16(v)
        o, lse = ops.attn_fwd_planes(qp, kp, vp, B, Sq, Sk, D, mask, H, precision=ops.PREC_F16, out_fmt="f16")
        dop = ops.make_planes(do, "bwd")
        dop = ops.Planes(dop.hi[:, :D].contiguous(), None, B * Sq, D)
        flops = float((Lq.double() * Lk.double()).sum()) * D * 2
        res = {}
        for form in args.forms.split(","):
            ops.ATTN_BWD_SPLIT = form != "two-kernel"
            ops.ATTN_BWD_RECOMPUTE = form == "recompute"
            for _ in range(3):
                r = ops.attn_bwd_planes(qp, kp, vp, o, dop, lse, B, Sq, Sk, D, mask, H, 0.0, (None, None, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                r = ops.attn_bwd_planes(qp, kp, vp, o, dop, lse, B, Sq, Sk, D, mask, H, 0.0, (None, None, None))
            e1.record()
            torch.cuda.synchronize()
            res[form] = (e0.elapsed_time(e1) / args.reps * 1e3, [x[0].hi[:, :D].float() for x in r[:3]])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.attn_fwd_planes(qp, kp, vp, B, Sq, Sk, D, mask, H, precision=ops.PREC_F16, out_fmt="f16")
        e1.record()
        torch.cuda.synchronize()
        tf = e0.elapsed_time(e1) / args.reps * 1e3
        if set(res) >= {"recompute", "emit", "two-kernel"}:
            ref = res["emit"][1]
            diffs = " ".join(f"{n} {float((a - b).norm() / (b.norm() + 1e-30)):.2e}" for n, a, b in zip(("dq", "dk", "dv"), res["recompute"][1], ref))
            print(f"{name:7s} {Sq}x{Sk}: fwd {tf:7.1f} us ({2 * flops / tf * 1e-6:6.1f} TF/s)  bwd recompute {res['recompute'][0]:7.1f} us "
                  f"({5 * flops / res['recompute'][0] * 1e-6:6.1f} TF/s on 5 products)  emit {res['emit'][0]:7.1f}  two-kernel {res['two-kernel'][0]:7.1f}"
                  f"   recompute vs emit: {diffs}", flush=True)
        else:
            print(f"{name:7s} {Sq}x{Sk} order={args.order}: fwd {tf:7.1f} us ({2 * flops / tf * 1e-6:6.1f} TF/s)  " +
                  "  ".join(f"bwd {f} {res[f][0]:7.1f} us ({5 * flops / res[f][0] * 1e-6:6.1f} TF/s)" for f in res), flush=True)


if __name__ == "__main__":
    main()
