"""-m gpu: the decoder's cross-attention against the RAW encoder memory (bmt_amd.ops.RawCrossAttnFn, ABI 8).

model/multihead_attention.py:62-84 projects the memory to keys and values; with 29 queries per sample the products reassociate --
S_h = (q_h W_k,h) X^T, O_h = (P_h X) W_v,h^T + b_v -- and K, V, dK, dV never exist.  The form must be indistinguishable from the reference's:
every check is against fp64 autograd over the reference's own formulas (operands as they are: the bars are the operand formats' -- fp16 products
against the memory, split-bf16 block products, bf16 backward), and the whole model against the CPU oracle and against its own projected form."""
import copy
import math

import numpy as np
import pytest
import torch

from bmt_amd import synthetic as syn
from tests.gpu_util import assert_close, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from bmt_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def amax(t):
    return float(t.float().abs().max()) if t.numel() else 0.0


def _mask(B, S, seed, holes=False):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(S // 3, 1), S + 1, (B,), generator=g)
    lens[0] = S
    m = torch.arange(S)[None, :] < lens[:, None]
    if holes:
        for b in range(1, B, 2):
            m[b, int(torch.randint(0, max(int(lens[b]) - 1, 1), (1,), generator=g))] = False
    return m.view(B, 1, S)


def _packed(ops, x, m):
    """(packed copy of the padded (B, S, D) tensor x under mask m with its RowPack attached, row indices of the valid positions)"""
    B, S, D = x.shape
    rows = torch.nonzero(m.view(-1)).view(-1)
    xp = torch.zeros(B, S, D)
    xp.view(-1, D)[:rows.numel()] = x.view(-1, D)[rows]
    xp = xp.to(DEV)
    pk = ops.pack_rows(m.to(DEV))
    ops.carry_pack(pk, xp)
    return xp, rows


# ------------------------------------------------------------------------------------------ the kernels between the products
@pytest.mark.parametrize("B,S,D", [(3, 70, 128), (4, 256, 1024), (2, 800, 128), (5, 33, 192)])
def test_memory_transposed(ops, B, S, D):
    m = _mask(B, S, seed=B + S, holes=True)
    x = rnd(B, S, D, seed=1) + 0.7                      # a common component: the mean key is not small
    xp, rows = _packed(ops, x, m)
    pl = ops.make_planes(xp.view(-1, D), "f16", pack=ops.pack_of(xp))
    Skp = ops._pad64(S)
    xt = torch.full((B, D, Skp), 3.0, device=DEV, dtype=torch.float16)
    xc = torch.full((B, D, Skp), 3.0, device=DEV, dtype=torch.bfloat16)
    ws = torch.zeros(B, D, device=DEV)
    ops._lib.check(ops.lib.bmt_memory_transposed(ops._p(pl.fh), pl.fh.stride(0), ops.pack_of(xp).off_ptr, B, D, Skp, ops._p(xt), ops._p(xc), ops._p(ws), ops._st()), "t")
    mm = m.view(B, S)
    for b in range(B):
        xb = x[b][mm[b]].to(torch.float16)               # (len, D): the sample's valid rows in order
        n = xb.shape[0]
        assert torch.equal(xt[b, :, :n].cpu(), xb.t()), b
        assert amax(xt[b, :, n:]) == 0.0
        want = (xb.float() - xb.float().mean(0, keepdim=True)).t()
        assert_close(xc[b, :, :n].float(), want.double(), atol=2e-2, rtol=2 ** -8, name=f"centred sample {b}")
        assert amax(xc[b, :, n:]) == 0.0


@pytest.mark.parametrize("B,H,Tq,S", [(3, 4, 29, 256), (2, 2, 30, 800), (4, 4, 5, 70)])
def test_raw_softmax_forward_and_backward(ops, B, H, Tq, S):
    m = _mask(B, S, seed=7 * B + S, holes=True)
    pk = ops.pack_rows(m.to(DEV))
    lens = m.view(B, S).sum(1)
    Skp, scale = ops._pad64(S), 1.0 / 16.0
    Sc = (rnd(B, H, 32, Skp, seed=2) * 8.0).to(DEV)
    Sc[:, :, :, S:] = float("nan")                       # never read: past every length
    Pf = torch.full((B, H, 32, Skp), 3.0, device=DEV, dtype=torch.float16)
    stack = torch.full((B, 2, H, 32, Skp), 3.0, device=DEV, dtype=torch.bfloat16)      # a stack with two row blocks per sample: P goes to block 1
    sb, sh = 2 * H * 32 * Skp, 32 * Skp
    ops._lib.check(ops.lib.bmt_raw_softmax_fwd(ops._p(Sc), pk.off_ptr, B, H, Tq, Skp, scale, ops._p(Pf),
                                               ops.C.c_void_p(stack.data_ptr() + 2 * H * 32 * Skp), sb, sh, ops._st()), "f")
    dP = rnd(B, H, 32, Skp, seed=3).to(DEV)
    ops._lib.check(ops.lib.bmt_raw_softmax_bwd(ops._p(Pf), ops._p(dP), pk.off_ptr, B, H, Tq, Skp, scale, ops._p(stack), sb, sh, ops._st()), "b")
    for b in range(B):
        n = int(lens[b])
        s = Sc[b, :, :Tq, :n].double().cpu() * scale
        P = torch.softmax(s, -1)
        assert_close(Pf[b, :, :Tq, :n].float(), P, atol=1e-3, rtol=2e-3, name="P")
        assert amax(Pf[b, :, :, n:]) == 0.0 and amax(Pf[b, :, Tq:]) == 0.0
        assert torch.equal(stack[b, 1, :, :Tq, :n].float().cpu(), Pf[b, :, :Tq, :n].float().to(torch.bfloat16).float().cpu()) or \
            rel_err(stack[b, 1, :, :Tq, :n].float().cpu(), P.float()) < 5e-3
        Pq = Pf[b, :, :Tq, :n].double().cpu()
        d = dP[b, :, :Tq, :n].double().cpu()
        dS = Pq * (d - (Pq * d).sum(-1, keepdim=True)) * scale
        assert_close(stack[b, 0, :, :Tq, :n].float(), dS, atol=2e-4, rtol=2 ** -7, name="dS")
        assert amax(stack[b, 0, :, :, n:]) == 0.0 and amax(stack[b, 0, :, Tq:]) == 0.0


def test_batched_small_products(ops):
    """bmt_gemm_small_batched: (sample, head) products at offsets, a packed B operand with device-side rows, two output planes with their own
    geometry, bias and column sums per inner index"""
    B, H, Tq, dm, dk, S = 3, 4, 29, 128, 64, 70
    m = _mask(B, S, seed=11, holes=True)
    lens = m.view(B, S).sum(1)
    X = rnd(B, S, dm, seed=1) * 0.5
    xp, rows = _packed(ops, X, m)
    pk = ops.pack_of(xp)
    xpl = ops.make_planes(xp.view(-1, dm), "f16", pack=pk)
    q = (rnd(B * Tq, H * dm, seed=2) * 0.5).to(DEV)
    qpl = ops.make_planes(q, "f16")
    Skp = ops._pad64(S)
    # S[b][h][t][k] = q[(b, t)][h dm : (h + 1) dm] . X_b[k]: one product per sample over the H Tq rows (h, t), block rows on both sides
    S_ = torch.full((B, H, 32, Skp), 7.0, device=DEV)
    ops.gemm_batched(ops.PREC_F16, H * Tq, S, dm, B, 1, ops._addr(qpl.fh), None, H * dm, ops._addr(xpl.fh), None, xpl.fh.stride(0),
                     a_off=(Tq * H * dm, 0), a_div=(Tq, dm), b_rows=pk.off_ptr, C_=ops._addr(S_), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp))
    mm = m.view(B, S)
    for b in range(B):
        n = int(lens[b])
        xb = X[b][mm[b]].to(torch.float16).double()
        want = torch.einsum("thd,kd->htk", q[b * Tq:(b + 1) * Tq].cpu().view(Tq, H, dm).to(torch.float16).double(), xb)
        assert_close(S_[b, :, :Tq, :n], want, atol=2e-3, rtol=1e-4, name=f"S sample {b}")
        assert bool((S_[b, :, Tq:] == 7.0).all()) and bool((S_[b, :, :, n:] == 7.0).all())
    # block products over the heads with two planes, bias and column sums: out[:, h dk : (h + 1) dk] = A[:, h dm : (h + 1) dm] W[h dk : (h + 1) dk]^T + b
    M = B * Tq
    A = (rnd(M, H * dm, seed=3) * 0.5).to(DEV)
    W = (rnd(H * dk, dm, seed=4) * 0.2).to(DEV)
    bias = rnd(H * dk, seed=5).to(DEV)
    Apl, Wpl = ops.make_planes(A, "x3"), ops.make_planes(W, "x3")
    o = ops._alloc_planes(M, H * dk, "x3", DEV, ld=H * dk)
    cs = torch.zeros(H * dk, device=DEV)
    ops.gemm_batched(ops.PREC_BF16X3, M, dk, dm, 1, H, ops._addr(Apl.hi), ops._addr(Apl.lo), H * dm, ops._addr(Wpl.hi), ops._addr(Wpl.lo), Wpl.hi.stride(0),
                     a_off=(0, dm), b_off=(0, dk * Wpl.hi.stride(0)), p1=ops._addr(o.hi), p2=ops._addr(o.lo), ldp=H * dk, p_off=(0, dk), bias=bias,
                     bias_off_i=dk, colsum=cs)
    want = torch.cat([A[:, h * dm:(h + 1) * dm].double().cpu() @ W[h * dk:(h + 1) * dk].double().cpu().t() for h in range(H)], 1) + bias.double().cpu()
    got = o.hi.float().double().cpu() + o.lo.float().double().cpu()
    assert_close(got, want, atol=3e-4, rtol=3e-5, name="block products (hi + lo)")
    assert_close(cs, want.sum(0), atol=2e-2, rtol=1e-4, name="column sums")


@pytest.mark.parametrize("dm,S,Tq,H", [(128, 800, 29, 4), (1024, 256, 29, 4), (128, 128, 7, 2), (256, 768, 32, 8), (64, 64, 1, 1), (128, 300, 30, 4)])
def test_fused_launch_equals_the_three_launches(ops, dm, S, Tq, H):
    """bmt_raw_attn_fwd / _bwd (ABI 12: both products against the memory and the row operation between them in one launch, the score tile in
    LDS) against the three launches they replace, on the same operands (P fp16 and bf16, O' hi + lo, dS, dQ'): the same arithmetic up to the
    order of the fp32 sums (the small-product kernel splits a long reduction over its four waves), so the outputs agree to a rounding of
    their 16-bit formats; ragged samples, one sample WITHOUT a key, query rows t >= Tq"""
    import ctypes as C
    lib = ops.lib
    B = 5
    assert lib.bmt_raw_attn_ok(dm, ops._pad64(S)) == 1
    m = _mask(B, S, seed=S + dm)
    m[2] = False                                         # a sample without a valid key: zeros, not NaN (bmt_raw_softmax_fwd's convention)
    lens = m.view(B, S).sum(1)
    X = rnd(B, S, dm, seed=1) * 0.7 + 0.3
    xp, rows = _packed(ops, X, m)
    pk = ops.pack_of(xp)
    Skp = ops._pad64(S)
    xpl = ops.make_planes(xp.view(-1, dm), "f16", pack=pk)
    xt = torch.empty(B, dm, Skp, device=DEV, dtype=torch.float16)
    xtc = torch.empty(B, dm, Skp, device=DEV, dtype=torch.bfloat16)
    ksum = torch.zeros(B, dm, device=DEV)
    ops._lib.check(lib.bmt_memory_transposed(ops._p(xpl.fh), xpl.fh.stride(0), pk.off_ptr, B, dm, Skp, ops._p(xt), ops._p(xtc), ops._p(ksum), None), "t")
    M = B * Tq
    scale = 1.0 / math.sqrt(256)
    qf = (rnd(M, H * dm, seed=2) * 0.5).to(DEV).to(torch.float16)
    sb, sh = 2 * H * 32 * Skp, 32 * Skp                  # a stack [B][2][H][32][Skp]: kind 0 = dS, kind 1 = P
    bsb, bsh = H * 32 * dm, 32 * dm

    def forward(fused):
        Pf = torch.full((B, H, 32, Skp), 3.0, device=DEV, dtype=torch.float16)
        stack = torch.full((B, 2, H, 32, Skp), 5.0, device=DEV, dtype=torch.bfloat16)
        hi = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
        lo = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
        pb = C.c_void_p(ops._addr(stack, H * 32 * Skp))
        if fused:
            ops._lib.check(lib.bmt_raw_attn_fwd(ops._addr(qf), Tq * H * dm, dm, H * dm, ops._addr(xpl.fh), xpl.fh.stride(0), pk.off_ptr, ops._addr(xt), B, H, Tq,
                                                dm, Skp, scale, ops._p(Pf), pb, sb, sh, ops._addr(hi), ops._addr(lo), H * dm, None), "f")
        else:
            S_ = torch.empty(B, H, 32, Skp, device=DEV)
            ops.gemm_batched(ops.PREC_F16, H * Tq, S, dm, B, 1, ops._addr(qf), None, H * dm, ops._addr(xpl.fh), None, xpl.fh.stride(0),
                             a_off=(Tq * H * dm, 0), a_div=(Tq, dm), b_rows=pk.off_ptr, C_=ops._addr(S_), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp))
            ops._lib.check(lib.bmt_raw_softmax_fwd(ops._p(S_), pk.off_ptr, B, H, Tq, Skp, scale, ops._p(Pf), pb, sb, sh, None), "s")
            ops.gemm_batched(ops.PREC_F16, H * Tq, dm, Skp, B, 1, ops._addr(Pf), None, Skp, ops._addr(xt), None, Skp, a_off=(H * 32 * Skp, 0),
                             a_div=(Tq, 32 * Skp), b_off=(dm * Skp, 0), p1=ops._addr(hi), p2=ops._addr(lo), ldp=H * dm, p_off=(Tq * H * dm, 0), p_div=(Tq, dm))
        torch.cuda.synchronize()
        return Pf, stack, hi, lo

    a, b_ = forward(True), forward(False)
    # (one fp16 / bf16 rounding step of the largest value: 2^-11 / 2^-8 relative)
    assert_close(a[0].float(), b_[0].float(), atol=1.1 * 2.0 ** -11 * amax(b_[0]), rtol=0, name="P fp16")
    assert_close(a[1].float(), b_[1].float(), atol=1.1 * 2.0 ** -8 * amax(b_[0]), rtol=0, name="P bf16 (stack; kind 0 untouched)")
    oa, ob = a[2].float() + a[3].float(), b_[2].float() + b_[3].float()
    assert_close(oa, ob, atol=2e-3 * amax(ob), rtol=0, name="O' (hi + lo)")
    assert rel_err(oa, ob) < 5e-4
    Pf = a[0]
    assert amax(Pf[2]) == 0.0 and amax(Pf[:, :, Tq:]) == 0.0
    rs = Pf[:, :, :Tq].float().sum(-1)
    assert bool(((rs - 1).abs() < 2e-3)[lens > 0].all())
    # backward: dO' rows from a B stack [B][H][32][dm] whose rows t >= Tq are zero
    dO = torch.zeros(B, H, 32, dm, device=DEV, dtype=torch.bfloat16)
    dO[:, :, :Tq] = (rnd(B, H, Tq, dm, seed=3) * 0.3).to(DEV).to(torch.bfloat16)

    def backward(fused):
        stack = torch.full((B, 2, H, 32, Skp), 5.0, device=DEV, dtype=torch.bfloat16)
        dq = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
        ds = C.c_void_p(ops._addr(stack))
        if fused:
            ops._lib.check(lib.bmt_raw_attn_bwd(ops._addr(dO), bsb, bsh, dm, ops._addr(xpl.hi), xpl.hi.stride(0), pk.off_ptr, ops._addr(xtc), ops._p(Pf), B, H, Tq,
                                                dm, Skp, scale, ds, sb, sh, ops._addr(dq), H * dm, None), "b")
        else:
            dP = torch.empty(B, H, 32, Skp, device=DEV)
            ops.gemm_batched(ops.PREC_BF16, H * Tq, S, dm, B, 1, ops._addr(dO), None, dm, ops._addr(xpl.hi), None, xpl.hi.stride(0),
                             a_off=(bsb, 0), a_div=(Tq, bsh), b_rows=pk.off_ptr, C_=ops._addr(dP), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp))
            ops._lib.check(lib.bmt_raw_softmax_bwd(ops._p(Pf), ops._p(dP), pk.off_ptr, B, H, Tq, Skp, scale, ds, sb, sh, None), "sb")
            ops.gemm_batched(ops.PREC_BF16, H * Tq, dm, Skp, B, 1, ops._addr(stack), None, Skp, ops._addr(xtc), None, Skp,
                             a_off=(sb, 0), a_div=(Tq, sh), b_off=(dm * Skp, 0), p1=ops._addr(dq), ldp=H * dm, p_off=(Tq * H * dm, 0), p_div=(Tq, dm))
        torch.cuda.synchronize()
        return stack, dq

    a, b_ = backward(True), backward(False)
    assert_close(a[0][:, 0].float(), b_[0][:, 0].float(), atol=1.1 * 2.0 ** -8 * amax(b_[0][:, 0]), rtol=0, name="dS (stack)")
    assert bool((a[0][:, 1] == 5.0).all())
    assert_close(a[1].float(), b_[1].float(), atol=1e-2 * amax(b_[1]), rtol=0, name="dQ'")
    assert rel_err(a[1].float(), b_[1].float()) < 4e-3 and amax(a[1]) > 0.0
    # ... and with the block products either side in the same launch (bmt_raw_attn_bwd_edges): dO'_h = do_h W_v,h in front, dq_h = dQ'_h W_k,h^T and
    # its column sums behind, against the three launches + the two block products
    dk = 128 if dm == 64 else 256
    if not lib.bmt_raw_attn_edges_ok(dm, Skp, dk):
        assert (dm, S) == (64, 64)
        return
    D = H * dk
    do = (rnd(M, D, seed=4) * 0.3).to(DEV).to(torch.bfloat16)
    wvT = (rnd(dm, D + 64, seed=5) * 0.1).to(DEV).to(torch.bfloat16)          # row d: W_v[h dk + k][d] at column 64 + h dk + k (a plane with other columns in front)
    wk = (rnd(D, dm, seed=6) * 0.1).to(DEV).to(torch.bfloat16)

    def chain(fused):
        stack = torch.full((B, 2, H, 32, Skp), 5.0, device=DEV, dtype=torch.bfloat16)
        bst = torch.zeros(B, H, 32, dm, device=DEV, dtype=torch.bfloat16)
        dqp = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
        dq = torch.zeros(M, D, device=DEV, dtype=torch.bfloat16)
        dbq = torch.ones(D, device=DEV)
        ds = C.c_void_p(ops._addr(stack))
        if fused:
            ops._lib.check(lib.bmt_raw_attn_bwd_edges(ops._addr(do), D, ops._addr(wvT, 64), wvT.stride(0), ops._addr(bst), bsb, bsh, ops._addr(xpl.hi), xpl.hi.stride(0),
                                                      pk.off_ptr, ops._addr(xtc), ops._p(Pf), B, H, Tq, dm, Skp, dk, scale, ds, sb, sh, ops._addr(dqp), H * dm,
                                                      ops._addr(wk), wk.stride(0), ops._addr(dq), D, ops._p(dbq), None), "e")
        else:
            ops.gemm_batched(ops.PREC_BF16, M, dm, dk, 1, H, ops._addr(do), None, D, ops._addr(wvT, 64), None, wvT.stride(0), a_off=(0, dk), b_off=(0, dk),
                             p1=ops._addr(bst), ldp=dm, p_off=(0, bsh), p_div=(Tq, bsb))
            ops._lib.check(lib.bmt_raw_attn_bwd(ops._addr(bst), bsb, bsh, dm, ops._addr(xpl.hi), xpl.hi.stride(0), pk.off_ptr, ops._addr(xtc), ops._p(Pf), B, H, Tq,
                                                dm, Skp, scale, ds, sb, sh, ops._addr(dqp), H * dm, None), "b")
            ops.gemm_batched(ops.PREC_BF16, M, dk, dm, 1, H, ops._addr(dqp), None, H * dm, ops._addr(wk), None, wk.stride(0), a_off=(0, dm),
                             b_off=(0, dk * wk.stride(0)), p1=ops._addr(dq), ldp=D, p_off=(0, dk), colsum=dbq, bias_off_i=dk)
        torch.cuda.synchronize()
        return bst, stack, dqp, dq, dbq

    a, b_ = chain(True), chain(False)
    assert amax(a[0][:, :, Tq:]) == 0.0 and amax(a[0]) > 0.0
    # ... and with the out-projection's dX in front of that (bmt_raw_attn_bwd_proj): do_h = mask(dy W_o[:, h-th block]) with the forward's dropout mask
    # re-applied and its column sums, against ops.linear_dx (the same transposed plane of W_o, the same element index of the mask) + bmt_raw_attn_bwd_edges
    if lib.bmt_raw_attn_bwd_proj_ok(dm, Skp, dk, 320) and M * D <= ops.SMALL_DX_OUTPUTS:
        Wo = (rnd(300, D, seed=12) * 0.05).to(DEV)
        P_ = ops.make_planes((rnd(M, 300, seed=13) * 0.3).to(DEV), "bwd")
        woT = ops.weight_planes_t(Wo)
        assert P_.hi.stride(0) == 320 and woT.hi.shape[0] == D
        for pdrop in (0.0, 0.1):
            def ochain(fused):
                stack = torch.full((B, 2, H, 32, Skp), 5.0, device=DEV, dtype=torch.bfloat16)
                bst = torch.zeros(B, H, 32, dm, device=DEV, dtype=torch.bfloat16)
                dqp = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
                dq = torch.zeros(M, D, device=DEV, dtype=torch.bfloat16)
                dbq, dbv = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
                do_ = torch.zeros(M, D, device=DEV, dtype=torch.bfloat16)
                ds = C.c_void_p(ops._addr(stack))
                if fused:
                    ops._lib.check(lib.bmt_raw_attn_bwd_proj(ops._addr(P_.hi), 320, 320, ops._addr(woT.hi), woT.hi.stride(0), pdrop, ops._p(ops.rng_tensor()) if pdrop else None, 7,
                                                             ops._p(dbv), ops._addr(do_), D, ops._addr(wvT, 64), wvT.stride(0), ops._addr(bst), bsb, bsh, ops._addr(xpl.hi),
                                                             xpl.hi.stride(0), pk.off_ptr, ops._addr(xtc), ops._p(Pf), B, H, Tq, dm, Skp, dk, scale, ds, sb, sh, ops._addr(dqp), H * dm,
                                                             ops._addr(wk), wk.stride(0), ops._addr(dq), D, ops._p(dbq), None), "bp")
                else:
                    ops.linear_dx(P_, Wo, out_planes=ops.Planes(do_, None, M, D), drop_post=True, drop_p=pdrop, site=7, colsum=dbv)
                    ops._lib.check(lib.bmt_raw_attn_bwd_edges(ops._addr(do_), D, ops._addr(wvT, 64), wvT.stride(0), ops._addr(bst), bsb, bsh, ops._addr(xpl.hi), xpl.hi.stride(0),
                                                              pk.off_ptr, ops._addr(xtc), ops._p(Pf), B, H, Tq, dm, Skp, dk, scale, ds, sb, sh, ops._addr(dqp), H * dm,
                                                              ops._addr(wk), wk.stride(0), ops._addr(dq), D, ops._p(dbq), None), "e")
                torch.cuda.synchronize()
                return do_, dbv, bst, dqp, dq, dbq

            oa, ob = ochain(True), ochain(False)
            assert_close(oa[0].float(), ob[0].float(), atol=1.1 * 2.0 ** -8 * amax(ob[0]), rtol=0, name=f"do (p = {pdrop})")
            if pdrop:
                za, zb = (oa[0] == 0), (ob[0] == 0)
                assert 0.05 < float(zb.float().mean()) < 0.15 and bool((za == zb).all()), "the dropout mask of the launch is not linear_dx's"
            for x_, y_, n, bar in zip(oa[1:], ob[1:], ("db_v", "dO' (B stack)", "dQ'", "dq", "db_q"), (1e-3, 4e-3, 1e-2, 1e-2, 1e-3)):
                e = rel_err(x_.float(), y_.float())
                assert e < bar, f"out-projection dX in the launch (p = {pdrop}): {n}: {e:.3e}"
    # the forward with ITS in-front block product in the launch (bmt_raw_attn_fwd_edges): Q'_h = q_h W_k,h (split-bf16, three passes) -> the A operand
    # (fp16, no copy in memory) + the B stack (bf16), against bmt_gemm_small_batched (x3) + bmt_raw_attn_fwd
    if lib.bmt_raw_attn_fwd_edges_ok(dm, Skp, dk):
        qv = rnd(M, D, seed=7) * 0.5
        q_pl = ops.make_planes(qv.to(DEV), "x3")
        wkT = ops.make_planes((rnd(dm, D, seed=8) * 0.1).to(DEV), "x3")

        def fchain(fused):
            Pf2 = torch.full((B, H, 32, Skp), 3.0, device=DEV, dtype=torch.float16)
            stack = torch.full((B, 2, H, 32, Skp), 5.0, device=DEV, dtype=torch.bfloat16)
            bst = torch.zeros(B, H, 32, dm, device=DEV, dtype=torch.bfloat16)
            hi = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
            lo = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
            pb = C.c_void_p(ops._addr(stack, H * 32 * Skp))
            if fused:
                ops._lib.check(lib.bmt_raw_attn_fwd_edges(ops._addr(q_pl.hi), ops._addr(q_pl.lo), q_pl.hi.stride(0), ops._addr(wkT.hi), ops._addr(wkT.lo), wkT.hi.stride(0),
                                                          ops._addr(bst), bsb, bsh, ops._addr(xpl.fh), xpl.fh.stride(0), pk.off_ptr, ops._addr(xt), B, H, Tq, dm, Skp, dk, scale,
                                                          ops._p(Pf2), pb, sb, sh, ops._addr(hi), ops._addr(lo), H * dm, None), "fe")
            else:
                qf2 = torch.empty(M, H * dm, device=DEV, dtype=torch.float16)
                ops.gemm_batched(ops.PREC_BF16X3, M, dm, dk, 1, H, ops._addr(q_pl.hi), ops._addr(q_pl.lo), q_pl.hi.stride(0), ops._addr(wkT.hi), ops._addr(wkT.lo),
                                 wkT.hi.stride(0), a_off=(0, dk), b_off=(0, dk), p1=ops._addr(bst), ldp=dm, p_off=(0, bsh), p_div=(Tq, bsb), p2=ops._addr(qf2), p2_f16=True,
                                 ldp2=H * dm, p2_off=(0, dm), p2_div=(0, 0))
                ops._lib.check(lib.bmt_raw_attn_fwd(ops._addr(qf2), Tq * H * dm, dm, H * dm, ops._addr(xpl.fh), xpl.fh.stride(0), pk.off_ptr, ops._addr(xt), B, H, Tq,
                                                    dm, Skp, scale, ops._p(Pf2), pb, sb, sh, ops._addr(hi), ops._addr(lo), H * dm, None), "f")
            torch.cuda.synchronize()
            return bst, Pf2, stack, hi.float() + lo.float()

        fa, fb = fchain(True), fchain(False)
        assert amax(fa[0][:, :, Tq:]) == 0.0 and amax(fa[0]) > 0.0
        for x_, y_, n, bar in zip(fa, fb, ("Q' (B stack)", "P fp16", "P bf16 (stack)", "O'"), (4e-3, 3e-3, 6e-3, 3e-3)):
            e = rel_err(x_.float(), y_.float())
            assert e < bar, f"forward edges: {n}: {e:.3e}"
        assert bool((fa[2][:, 0] == 5.0).all())
        # ... and with the query projection in front of that (bmt_raw_attn_fwd_proj): q_h = y W_q,h^T + b_q,h from the sample's rows of y (300 columns padded to
        # 320), against the product of the operands' hi + lo planes in fp64 fed to bmt_raw_attn_fwd_edges
        Kq = 320
        if lib.bmt_raw_attn_fwd_proj_ok(dm, Skp, dk, Kq):
            y_pl = ops.make_planes((rnd(M, 300, seed=9) * 0.8).to(DEV), "x3")
            Wq32 = (rnd(D, 300, seed=10) * 0.06).to(DEV)
            wq_pl = ops.make_planes(Wq32, "x3")
            bq = (rnd(D, seed=11) * 0.2).to(DEV)
            assert y_pl.hi.stride(0) == Kq and wq_pl.hi.stride(0) == Kq
            q64 = (y_pl.hi.double() + y_pl.lo.double()) @ (wq_pl.hi.double() + wq_pl.lo.double()).t() + bq.double()
            q_ref = ops.make_planes(q64.float().contiguous(), "x3")

            def pchain(fused):
                Pf2 = torch.full((B, H, 32, Skp), 3.0, device=DEV, dtype=torch.float16)
                stack = torch.full((B, 2, H, 32, Skp), 5.0, device=DEV, dtype=torch.bfloat16)
                bst = torch.zeros(B, H, 32, dm, device=DEV, dtype=torch.bfloat16)
                hi = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
                lo = torch.zeros(M, H * dm, device=DEV, dtype=torch.bfloat16)
                qo = torch.zeros(M, D, device=DEV, dtype=torch.bfloat16)
                pb = C.c_void_p(ops._addr(stack, H * 32 * Skp))
                if fused:
                    ops._lib.check(lib.bmt_raw_attn_fwd_proj(ops._addr(y_pl.hi), ops._addr(y_pl.lo), Kq, Kq, ops._addr(wq_pl.hi), ops._addr(wq_pl.lo), Kq, ops._p(bq), ops._addr(qo), D,
                                                             ops._addr(wkT.hi), ops._addr(wkT.lo), wkT.hi.stride(0), ops._addr(bst), bsb, bsh, ops._addr(xpl.fh), xpl.fh.stride(0),
                                                             pk.off_ptr, ops._addr(xt), B, H, Tq, dm, Skp, dk, scale, ops._p(Pf2), pb, sb, sh, ops._addr(hi), ops._addr(lo), H * dm,
                                                             None), "fp")
                else:
                    qo = q_ref.hi.clone()
                    ops._lib.check(lib.bmt_raw_attn_fwd_edges(ops._addr(q_ref.hi), ops._addr(q_ref.lo), q_ref.hi.stride(0), ops._addr(wkT.hi), ops._addr(wkT.lo), wkT.hi.stride(0),
                                                              ops._addr(bst), bsb, bsh, ops._addr(xpl.fh), xpl.fh.stride(0), pk.off_ptr, ops._addr(xt), B, H, Tq, dm, Skp, dk, scale,
                                                              ops._p(Pf2), pb, sb, sh, ops._addr(hi), ops._addr(lo), H * dm, None), "fe")
                torch.cuda.synchronize()
                return qo, bst, Pf2, hi.float() + lo.float()

            pa, pb_ = pchain(True), pchain(False)
            assert_close(pa[0].float(), pb_[0].float(), atol=1.1 * 2.0 ** -8 * amax(pb_[0]), rtol=0, name="q (high plane)")
            # the projection inside the launch adds in gemm_small_kernel's order (per 16 reduction indices: lo . hi, hi . lo, hi . hi into one accumulator):
            # q's high plane is, bit for bit, what ops.linear_fwd_planes -- the launch it replaces -- writes from the same operands
            wreg = ops.weight_planes(Wq32, "x3")
            assert torch.equal(wreg.hi.view(torch.int16), wq_pl.hi.view(torch.int16)) and torch.equal(wreg.lo.view(torch.int16), wq_pl.lo.view(torch.int16))
            q_lin = ops.linear_fwd_planes(y_pl, Wq32, bq, precision=ops.PREC_BF16X3, out_fmt="x3")
            torch.cuda.synchronize()
            neq = pa[0].view(torch.int16) != q_lin.hi.view(torch.int16)
            assert not bool(neq.any()), (f"q differs from the replaced launch's in {int(neq.sum())} of {neq.numel()} elements (max {amax(pa[0].float() - q_lin.hi.float()):.3e}; "
                                         f"rows {sorted(set(torch.nonzero(neq)[:, 0].tolist()))[:8]}, columns {sorted(set(torch.nonzero(neq)[:, 1].tolist()))[:8]})")
            for x_, y_, n, bar in zip(pa[1:], pb_[1:], ("Q' (B stack)", "P fp16", "O'"), (4e-3, 3e-3, 3e-3)):
                e = rel_err(x_.float(), y_.float())
                assert e < bar, f"query projection in the launch: {n}: {e:.3e}"
            assert amax(pa[1][:, :, Tq:]) == 0.0
    for x_, y_, n, bar in zip(a, b_, ("dO' (B stack)", "dS (stack)", "dQ'", "dq", "db_q (accumulated onto ones)"), (4e-3, 1e-2, 1e-2, 1e-2, 1e-3)):
        e = rel_err(x_.float(), y_.float())
        assert e < bar, f"{n}: {e:.3e}"
    assert bool((a[1][:, 1] == 5.0).all())


# ------------------------------------------------------------------------------------------ one attention module, both forms, against fp64
def _reference_mha(Q, X, m, P, H):
    """model/multihead_attention.py:55-86 in fp64 (dropout off); X padded (B, S, dm), m (B, 1, S)"""
    q = Q @ P["Wq"].t() + P["bq"]
    k = X @ P["Wk"].t() + P["bk"]
    v = X @ P["Wv"].t() + P["bv"]
    B, Tq, D = q.shape
    dk = D // H
    sp = lambda t: t.view(B, -1, H, dk).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dk)
    s = s.masked_fill(m.unsqueeze(1) == 0, -float("inf"))
    o = (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(B, Tq, D)
    return o @ P["Wo"].t() + P["bo"]


@pytest.mark.parametrize("dm,S,holes", [(128, 800, False), (1024, 256, True), (128, 100, True)])
def test_cross_attention_against_the_raw_memory(ops, dm, S, holes):
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    B, Tq, Dq, D, H, L = 4, 29, 300, 1024, 4, 2
    torch.manual_seed(0)
    att = ops.tag_policy(MultiheadedAttention(Dq, dm, dm, H, 0.0, D), "dec").to(DEV)
    with torch.no_grad():
        att.linear_K2d.bias.normal_(0, 0.5)              # a key bias that matters if it is (wrongly) kept or dropped in the wrong place
    m = _mask(B, S, seed=S + dm, holes=holes)
    X = rnd(B, S, dm, seed=1) * 0.7 + 0.3
    Q = rnd(B, Tq, Dq, seed=2)
    G = rnd(B, Tq, Dq, seed=3) * 0.1
    # fp64 reference
    P64 = {n: getattr(att, ln).__getattr__(pn).detach().double().cpu().requires_grad_(True)
           for n, ln, pn in (("Wq", "linear_Q2d", "weight"), ("bq", "linear_Q2d", "bias"), ("Wk", "linear_K2d", "weight"), ("bk", "linear_K2d", "bias"),
                             ("Wv", "linear_V2d", "weight"), ("bv", "linear_V2d", "bias"), ("Wo", "linear_d2Q", "weight"), ("bo", "linear_d2Q", "bias"))}
    Q64, X64 = Q.double().requires_grad_(True), X.double().requires_grad_(True)
    y64 = _reference_mha(Q64, X64, m, P64, H)
    y64.backward(G.double())
    # the reassociated form on the packed memory (layer 1 of a two-layer state: layer 0's slots stay empty)
    xp, rows = _packed(ops, X, m)
    xp.requires_grad_(True)
    Qd = Q.to(DEV).requires_grad_(True)
    mem = ops.raw_memory(xp, L, H, Tq, ops.policy_of(att))
    assert getattr(mem, "_bmt_rawmem", None) is not None
    mem._bmt_rawmem.next_layer = 1
    mem._bmt_rawmem.astack.fill_(float("nan"))           # (the unused layer's rows of the A stack hold anything: RawMemoryFn zeroes what no layer wrote)
    y = att(Qd, mem, mem, m.to(DEV))
    assert mem._bmt_rawmem.used and mem._bmt_rawmem.next_layer == 2, "the reassociated form did not run"
    y.backward(G.to(DEV))
    torch.cuda.synchronize()
    assert_close(y, y64.detach(), atol=2e-3 * float(y64.detach().abs().max()), rtol=0, name="output")
    e = {}
    e["dQ"] = rel_err(Qd.grad.cpu(), Q64.grad.float())
    gx = torch.zeros(B * S, dm)
    gx[rows] = xp.grad.view(-1, dm)[:rows.numel()].cpu()
    e["dX"] = rel_err(gx.view(B, S, dm), X64.grad.float())
    for n, ln, pn in (("Wq", "linear_Q2d", "weight"), ("bq", "linear_Q2d", "bias"), ("Wk", "linear_K2d", "weight"), ("Wv", "linear_V2d", "weight"),
                      ("bv", "linear_V2d", "bias"), ("Wo", "linear_d2Q", "weight"), ("bo", "linear_d2Q", "bias")):
        e[n] = rel_err(getattr(att, ln).__getattr__(pn).grad.cpu(), P64[n].grad.float())
    print("\nreassociated cross-attention vs fp64:", {k: f"{v:.2e}" for k, v in e.items()})
    assert float(att.linear_K2d.bias.grad.abs().max()) == 0.0 and float(P64["bk"].grad.abs().max()) < 1e-12      # exactly zero, both
    assert max(e.values()) < 2e-2, e
    # and the projected form of the same module (what ran before round 5): same bars, and the two agree
    att2 = copy.deepcopy(att)
    for p_ in att2.parameters():
        p_.grad = None
    xp2, _ = _packed(ops, X, m)
    xp2.requires_grad_(True)
    Q2 = Q.to(DEV).requires_grad_(True)
    y2 = att2(Q2, xp2, xp2, m.to(DEV))
    y2.backward(G.to(DEV))
    assert_close(y, y2.detach(), atol=3e-3 * float(y64.detach().abs().max()), rtol=0, name="reassociated vs projected output")
    n = rows.numel()                                    # (the projected form leaves the rows past the count unwritten)
    assert rel_err(Qd.grad, Q2.grad) < 3e-2 and rel_err(xp.grad.view(-1, dm)[:n], xp2.grad.view(-1, dm)[:n]) < 3e-2


@pytest.mark.parametrize("dm,S", [(128, 800), (1024, 256)])
def test_query_projection_inside_the_launch_changes_no_bit(ops, dm, S, monkeypatch):
    """ops.RAW_FUSED_PROJ: q_h = y W_q,h^T + b_q,h inside the fused forward launch adds in the order of the launch it replaces, so the module's output
    is the same tensor, bit for bit, with and without it (a forward that differed in its low bits put the ten-Adam-step trajectory of
    tests/test_gpu_model.py on another path: profiles/r06_z6_raw_fused_ab.txt)"""
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    B, Tq, Dq, D, H, L = 4, 29, 300, 1024, 4, 1
    torch.manual_seed(0)
    att = ops.tag_policy(MultiheadedAttention(Dq, dm, dm, H, 0.0, D), "dec").to(DEV)
    m = _mask(B, S, seed=S + dm)
    X = rnd(B, S, dm, seed=1) * 0.7 + 0.3
    Q = rnd(B, Tq, Dq, seed=2)
    assert lib_ok(ops, dm, ops._pad64(S), D // H)
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(ops, "RAW_FUSED_PROJ", flag)
        xp, _ = _packed(ops, X, m)
        xp.requires_grad_(True)
        mem = ops.raw_memory(xp, L, H, Tq, ops.policy_of(att))
        assert getattr(mem, "_bmt_rawmem", None) is not None
        y = att(Q.to(DEV).requires_grad_(True), mem, mem, m.to(DEV))
        assert mem._bmt_rawmem.used
        torch.cuda.synchronize()
        outs.append(y.detach().clone())
    assert torch.equal(outs[0], outs[1]), f"{amax(outs[0] - outs[1]):.3e}"


def lib_ok(ops, dm, Skp, dk):
    return bool(ops.lib.bmt_raw_attn_fwd_proj_ok(dm, Skp, dk, 320))


@pytest.mark.parametrize("dm,S", [(128, 300), (1024, 256)])
def test_cross_attention_dropout_masks_are_those_of_the_projected_form(ops, dm, S):
    """training mode: the dropout on the attention output (model/multihead_attention.py:22-23) is drawn in the value block product's epilogue and
    re-applied in the out-projection's dX -- the SAME mask (site, element index) the projected form's attention kernel draws, forward and backward:
    output and gradients of the two forms agree as they do without dropout (a wrong index on either side would show at the size of p)"""
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    B, Tq, Dq, D, H, p = 4, 29, 300, 1024, 4, 0.3
    torch.manual_seed(1)
    att = ops.tag_policy(MultiheadedAttention(Dq, dm, dm, H, p, D), "dec").to(DEV).train()
    att2 = copy.deepcopy(att)                            # (same dropout site)
    m = _mask(B, S, seed=3 * S + dm, holes=True)
    X, Q, G = rnd(B, S, dm, seed=1) * 0.7 + 0.3, rnd(B, Tq, Dq, seed=2), rnd(B, Tq, Dq, seed=3) * 0.1
    ops.manual_seed(77)
    out = []
    for raw, mod in ((True, att), (False, att2)):
        xp, rows = _packed(ops, X, m)
        xp.requires_grad_(True)
        Qd = Q.to(DEV).requires_grad_(True)
        mem = ops.raw_memory(xp, 1, H, Tq, ops.policy_of(mod)) if raw else xp
        assert (getattr(mem, "_bmt_rawmem", None) is not None) == raw
        y = mod(Qd, mem, mem, m.to(DEV))
        y.backward(G.to(DEV))
        n = rows.numel()
        out.append((y.detach(), Qd.grad.clone(), xp.grad.view(-1, dm)[:n].clone(), mod.linear_V2d.weight.grad.clone(), mod.linear_Q2d.weight.grad.clone()))
    scale = float(out[1][0].abs().max())
    assert_close(out[0][0], out[1][0], atol=3e-3 * scale, rtol=0, name="output under dropout, raw memory vs projected")
    for a, b_, name in zip(out[0][1:], out[1][1:], ("dQ", "dX", "dW_v", "dW_q")):
        e = rel_err(a, b_)
        assert e < 3e-2, f"{name}: raw memory vs projected under dropout {e:.3e}"


# ------------------------------------------------------------------------------------------ the whole model
@pytest.mark.parametrize("holes", [False, True])
def test_model_with_and_without_projected_keys_and_values(ops, holes):
    """the captioning model (two layers, d_k 128, a ragged batch, optionally with masks that are not suffixes) with the decoder's cross-attentions
    against the raw memories and with keys and values projected (RAW_MEMORY off: the reference's formulation, what BMT_RAW_MEMORY=0 selects): both
    within 1e-3 of the CPU oracle's log-probabilities and inside its gradient bars, and within the operand formats' noise of each other"""
    from oracle import bmt_oracle as orc
    from tests.test_gpu_model import _check_grads
    from tests.test_gpu_packed import _build, _poison_allocator, _run
    cfg = syn.make_cfg(d_model=512, H=4, N=2, d_aud=128, d_vid=256, d_model_caps=64, dout_p=0.0)
    V, B, Tv, Ta, Tc = 60, 4, 90, 210, 11
    model = _build(cfg, V)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=99)
    fs, caps = batch["feature_stacks"], batch["captions"]
    if holes:
        for b, t in ((1, 3), (1, 4), (2, 0), (3, 17)):
            fs["rgb"][b, t, 0] = float(syn.PAD_IDX)
        for b, t in ((0, 100), (2, 1), (3, 0), (3, 1)):
            fs["audio"][b, t, 0] = float(syn.PAD_IDX)
    p = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    oloss, opred, _ = orc.train_cap_loss(p, cfg, fs, caps, syn.PAD_IDX, cfg.smoothing)
    oloss.backward()
    trainable = {k for k, q in model.named_parameters() if q.requires_grad}
    ograds = {k: v.grad for k, v in p.items() if v.grad is not None and k in trainable}
    preds, grads = {}, {}
    calls = [0]
    fwd = ops.RawCrossAttnFn.forward

    def counting(*a, **kw):
        calls[0] += 1
        return fwd(*a, **kw)
    for raw in (True, False):
        model.zero_grad()
        ops.RAW_MEMORY = raw
        ops.RawCrossAttnFn.forward = staticmethod(counting)
        try:
            _poison_allocator()
            pred, loss, masks, packs = _run(model, cfg, fs, caps)
            assert packs is not None
            loss.backward()
        finally:
            ops.RAW_MEMORY = True
            ops.RawCrossAttnFn.forward = staticmethod(fwd)
        assert calls[0] == 4, calls      # two layers x two memories in the raw arm, none added by the projected arm
        err = float((pred.detach().cpu() - opred.detach()).abs().max())
        print(f"\nraw memory {raw} (holes={holes}): max |dlogp| vs the oracle = {err:.3e}")
        assert_close(pred, opred.detach(), atol=1e-3, name=f"log-probs, raw memory {raw}")
        _check_grads([(k, q) for k, q in model.named_parameters() if k in trainable], ograds)
        preds[raw] = pred.detach().float().cpu()
        grads[raw] = {k: q.grad.detach().clone() for k, q in model.named_parameters() if q.grad is not None}
    assert float((preds[True] - preds[False]).abs().max()) < 6e-4
    num = sum(float((grads[True][k].double() - grads[False][k].double()).norm() ** 2) for k in grads[False])
    den = sum(float(grads[False][k].double().norm() ** 2) for k in grads[False])
    assert (num / den) ** 0.5 < 1e-2, f"raw memory vs projected gradients differ by {(num / den) ** 0.5:.3e}"


def test_captured_step_with_the_raw_memory_form(ops):
    """the step as hipGraphs with the reassociated cross-attentions inside: losses of replays over different batches equal the eager step's"""
    from bmt_amd.train import CaptioningTrainStep
    from tests.test_gpu_packed import _build
    cfg = syn.make_cfg(d_model=512, H=4, N=2, d_aud=128, d_vid=256, d_model_caps=64, dout_p=0.0, lr=1e-4)
    V, B, Tv, Ta, Tc = 60, 4, 90, 210, 11
    batches = [syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=s) for s in (5, 6, 7)]
    dev = lambda b: ({k: v.to(DEV) for k, v in b["feature_stacks"].items()}, b["captions"].to(DEV))
    losses = {}
    for mode in ("eager", "graph"):
        model = _build(cfg, V)
        step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, seed=3)
        if mode == "graph":
            step.capture(*dev(batches[0]))
            model.load_state_dict(_build(cfg, V).state_dict())        # (the capture's warm-up steps moved the weights)
        out = []
        for b in batches:
            loss, _ = (step.replay(*dev(b)) if mode == "graph" else step(*dev(b)))
            out.append(float(loss))
        losses[mode] = out
    print("\nraw memory, eager vs graph losses:", losses)
    assert abs(losses["eager"][0] - losses["graph"][0]) < 2e-3, losses     # (first step: identical weights and dropout stream)
    assert all(math.isfinite(x) for x in losses["graph"])
