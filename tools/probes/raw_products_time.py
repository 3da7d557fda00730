#!/usr/bin/env python
"""the batched small products of the decoder's cross-attention against the raw memories (ops.RawCrossAttnFn) one by one at configs[1]'s shapes:
us per launch inside a replayed graph, for each grouping of a workgroup's waves (split 1: 64 x 64 blocks, 4: one tile, the reduction over the
waves).  python tools/probes/raw_products_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402
from tools.probes.gemm_small_time import timed  # noqa: E402

DEV = "cuda"
B, H, Tq, D = 32, 4, 29, 1024
dk, M = D // H, B * Tq


def main():
    flush = torch.zeros(64 << 20, device=DEV)
    for name, dm, S in (("video", 1024, 256), ("audio", 128, 800)):
        Skp = ops._pad64(S)
        lens = torch.randint(S // 2, S + 1, (B,))
        m = (torch.arange(S)[None, :] < lens[:, None]).view(B, 1, S).to(DEV)
        pk = ops.pack_rows(m)
        bf = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16)
        hf = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
        q_hi, q_lo = bf(M, D), bf(M, D)
        wT_hi, wT_lo = bf(dm, 2 * D), bf(dm, 2 * D)
        w_hi, w_lo = bf(2 * D, dm), bf(2 * D, dm)
        x_fh, x_hi = hf(B * S, dm), bf(B * S, dm)
        xt = hf(B, dm, Skp)
        xtc = bf(B, dm, Skp)
        qf = hf(M, H * dm)
        nat_hi, nat_lo = bf(M, H * dm), bf(M, H * dm)
        o_hi, o_lo = bf(M, D), bf(M, D)
        stackB = bf(B, 2, 2, H, 32, dm)
        stackA = bf(B, 2, 2, H, 32, Skp)
        S_ = torch.empty(B, H, 32, Skp, device=DEV)
        Pf = hf(B, H, 32, Skp)
        bsb, bsh, asb, ash = 2 * 2 * H * 32 * dm, 32 * dm, 2 * 2 * H * 32 * Skp, 32 * Skp
        A = ops._addr
        cases = {
            "Q' = q_h W_k,h (x3)": lambda sp: ops.gemm_batched(ops.PREC_BF16X3, M, dm, dk, 1, H, A(q_hi), A(q_lo), D, A(wT_hi), A(wT_lo), 2 * D, a_off=(0, dk), b_off=(0, dk),
                                                              p1=A(stackB), ldp=dm, p_off=(0, bsh), p_div=(Tq, bsb), p2=A(qf), p2_f16=True, ldp2=H * dm, p2_off=(0, dm), p2_div=(0, 0), split=sp),
            "S = Q' X^T (f16)": lambda sp: ops.gemm_batched(ops.PREC_F16, H * Tq, S, dm, B, 1, A(qf), None, H * dm, A(x_fh), None, dm, a_off=(Tq * H * dm, 0), a_div=(Tq, dm),
                                                            b_rows=pk.off_ptr, C_=A(S_), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp), split=sp),
            "O' = P X (f16)": lambda sp: ops.gemm_batched(ops.PREC_F16, H * Tq, dm, Skp, B, 1, A(Pf), None, Skp, A(xt), None, Skp, a_off=(H * 32 * Skp, 0), a_div=(Tq, 32 * Skp),
                                                          b_off=(dm * Skp, 0), p1=A(nat_hi), p2=A(nat_lo), ldp=H * dm, p_off=(Tq * H * dm, 0), p_div=(Tq, dm), split=sp),
            "out_h = O'_h W_v,h^T (x3)": lambda sp: ops.gemm_batched(ops.PREC_BF16X3, M, dk, dm, 1, H, A(nat_hi), A(nat_lo), H * dm, A(w_hi), A(w_lo), dm, a_off=(0, dm), b_off=(0, dk * dm),
                                                                    p1=A(o_hi), p2=A(o_lo), ldp=D, p_off=(0, dk), ldc=D, split=sp),
            "dO' = do_h W_v,h (bf16)": lambda sp: ops.gemm_batched(ops.PREC_BF16, M, dm, dk, 1, H, A(o_hi), None, D, A(wT_hi, D), None, 2 * D, a_off=(0, dk), b_off=(0, dk),
                                                                  p1=A(stackB), ldp=dm, p_off=(0, bsh), p_div=(Tq, bsb), split=sp),
            "dP = dO' X^T (bf16)": lambda sp: ops.gemm_batched(ops.PREC_BF16, H * Tq, S, dm, B, 1, A(stackB), None, dm, A(x_hi), None, dm, a_off=(bsb, 0), a_div=(Tq, bsh),
                                                              b_rows=pk.off_ptr, C_=A(S_), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp), split=sp),
            "dQ' = dS X (bf16)": lambda sp: ops.gemm_batched(ops.PREC_BF16, H * Tq, dm, Skp, B, 1, A(stackA), None, Skp, A(xtc), None, Skp, a_off=(asb, 0), a_div=(Tq, ash),
                                                            b_off=(dm * Skp, 0), p1=A(nat_hi), ldp=H * dm, p_off=(Tq * H * dm, 0), p_div=(Tq, dm), split=sp),
            "dq_h = dQ'_h W_k,h^T (bf16)": lambda sp: ops.gemm_batched(ops.PREC_BF16, M, dk, dm, 1, H, A(nat_hi), None, H * dm, A(w_hi), None, dm, a_off=(0, dm), b_off=(0, dk * dm),
                                                                      p1=A(o_hi), ldp=D, p_off=(0, dk), split=sp),
        }
        lib, ck = ops.lib, ops._lib.check
        dbq = torch.zeros(D, device=DEV)
        y_hi, y_lo, wq_hi, wq_lo = bf(M, 320), bf(M, 320), bf(D, 320), bf(D, 320)
        fused = {
            "fused S -> softmax -> O' (f16)": lambda: ck(lib.bmt_raw_attn_fwd(A(qf), Tq * H * dm, dm, H * dm, A(x_fh), dm, pk.off_ptr, A(xt), B, H, Tq, dm, Skp, 0.0625, A(Pf),
                                                                                A(stackA), asb, ash, A(nat_hi), A(nat_lo), H * dm, ops._st()), "f"),
            "fused dP -> dS -> dQ' (bf16)": lambda: ck(lib.bmt_raw_attn_bwd(A(stackB), bsb, bsh, dm, A(x_hi), dm, pk.off_ptr, A(xtc), A(Pf), B, H, Tq, dm, Skp, 0.0625,
                                                                              A(stackA), asb, ash, A(nat_hi), H * dm, ops._st()), "b"),
            "fused dO' -> ... -> dq (bf16)": lambda: ck(lib.bmt_raw_attn_bwd_edges(A(o_hi), D, A(wT_hi, D), 2 * D, A(stackB), bsb, bsh, A(x_hi), dm, pk.off_ptr, A(xtc), A(Pf), B, H, Tq,
                                                                                     dm, Skp, dk, 0.0625, A(stackA), asb, ash, A(nat_hi), H * dm, A(w_hi), dm, A(q_hi), D, A(dbq),
                                                                                     ops._st()), "e"),
            "fused Q' -> ... -> O' (x3 + f16)": lambda: ck(lib.bmt_raw_attn_fwd_edges(A(q_hi), A(q_lo), D, A(wT_hi), A(wT_lo), 2 * D, A(stackB), bsb, bsh, A(x_fh), dm, pk.off_ptr, A(xt),
                                                                                       B, H, Tq, dm, Skp, dk, 0.0625, A(Pf), A(stackA), asb, ash, A(nat_hi), A(nat_lo), H * dm,
                                                                                       ops._st()), "fe"),
            "fused q -> Q' -> ... -> O' (proj)": lambda: ck(lib.bmt_raw_attn_fwd_proj(A(y_hi), A(y_lo), 320, 320, A(wq_hi), A(wq_lo), 320, A(dbq), A(q_hi), D, A(wT_hi), A(wT_lo), 2 * D,
                                                                                        A(stackB), bsb, bsh, A(x_fh), dm, pk.off_ptr, A(xt), B, H, Tq, dm, Skp, dk, 0.0625, A(Pf),
                                                                                        A(stackA), asb, ash, A(nat_hi), A(nat_lo), H * dm, ops._st()), "fp"),
            "fused do -> dO' -> ... -> dq (proj)": lambda: ck(lib.bmt_raw_attn_bwd_proj(A(y_hi), 320, 320, A(wq_hi), 320, 0.1, A(ops.rng_tensor()), 7, A(dbq), A(o_hi), D, A(wT_hi, D), 2 * D,
                                                                                          A(stackB), bsb, bsh, A(x_hi), dm, pk.off_ptr, A(xtc), A(Pf), B, H, Tq, dm, Skp, dk, 0.0625,
                                                                                          A(stackA), asb, ash, A(nat_hi), H * dm, A(w_hi), dm, A(q_hi), D, A(dbq), ops._st()), "bp"),
            "softmax forward alone": lambda: ck(lib.bmt_raw_softmax_fwd(A(S_), pk.off_ptr, B, H, Tq, Skp, 0.0625, A(Pf), A(stackA), asb, ash, ops._st()), "s"),
        }
        print(f"--- {name} memory: d = {dm}, {S} keys (capacity), {B} samples x {H} heads x {Tq} queries")
        for label, f in fused.items():
            f()
            print(f"{label:32s} {timed(f, 20):6.1f} us hot {timed(f, 10, flush):6.1f} us cold", flush=True)
        for label, fn in cases.items():
            row = []
            for sp in (1, 4):
                f = lambda: fn(sp)
                f()
                row.append(f"split {sp}: {timed(f, 20):6.1f} us hot {timed(f, 10, flush):6.1f} us cold")
            print(f"{label:32s} " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
