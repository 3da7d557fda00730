"""-m gpu: the encoder's self-attention over an input narrower than a head (bmt_amd.ops.RankSelfAttnFn, round 6).

model/multihead_attention.py:62-84 projects the 128-wide audio stream to d_model = 1024 for four heads of 256; q_h, k_h, v_h are rank-128 images of
the same input x, and the products reassociate -- S_h = (x W'_h^T + c_h) x^T, O_h = (P_h x) W_v,h^T + b_v,h with W'_h = W_k,h^T W_q,h, c_h = b_q,h W_k,h --
so the attention runs at width 128 against ONE key / value plane (the input itself).  The form must be indistinguishable from the reference's:
every check is against fp64 autograd over the reference's own formulas and against the module's projected form (MHAFn: RANK_ATTN off)."""
import copy
import math

import pytest
import torch

from tests.gpu_util import assert_close, rel_err
from tests.test_gpu_raw_memory import _mask, _packed, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda"

_NAMES = (("Wq", "linear_Q2d", "weight"), ("bq", "linear_Q2d", "bias"), ("Wk", "linear_K2d", "weight"), ("bk", "linear_K2d", "bias"),
          ("Wv", "linear_V2d", "weight"), ("bv", "linear_V2d", "bias"), ("Wo", "linear_d2Q", "weight"), ("bo", "linear_d2Q", "bias"))


@pytest.fixture(scope="module")
def ops():
    from bmt_amd import ops as _ops
    return _ops


def _reference_self_attention(X, m, P, H):
    """model/multihead_attention.py:55-86 in fp64 with Q = K = V = X (dropout off); X padded (B, S, d_in), m (B, 1, S)"""
    q = X @ P["Wq"].t() + P["bq"]
    k = X @ P["Wk"].t() + P["bk"]
    v = X @ P["Wv"].t() + P["bv"]
    B, S, D = q.shape
    dk = D // H
    sp = lambda t: t.view(B, S, H, dk).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dk)
    s = s.masked_fill(m.unsqueeze(1) == 0, -float("inf"))
    o = (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(B, S, D)
    return o @ P["Wo"].t() + P["bo"]


def _module(ops, d_in, H, D, p=0.0, seed=0):
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    torch.manual_seed(seed)
    att = ops.tag_policy(MultiheadedAttention(d_in, d_in, d_in, H, p, D), "enc").to(DEV)
    with torch.no_grad():
        att.linear_K2d.bias.normal_(0, 0.5)              # a key bias that matters if it is (wrongly) kept
        att.linear_Q2d.bias.normal_(0, 0.5)              # ... and a query bias whose chain rule through c = b_q W_k is visible
    return att


def _counted(ops):
    calls = [0]
    fwd = ops.RankSelfAttnFn.forward

    def counting(*a, **kw):
        calls[0] += 1
        return fwd(*a, **kw)
    return calls, fwd, counting


# ------------------------------------------------------------------------------------------ the kernels of the form
@pytest.mark.parametrize("d_b", [128, 384])
def test_rank_prep_and_chain_kernels(ops, d_b):
    """bmt_rank_prep: W'_h = W_k,h^T W_q,h [d_a][d_b] and c_h = b_q,h W_k,h as planes; bmt_rank_chain: the chain rule back through them (fp32 tile products; d_b = d_a: the self-attention; d_b > d_a: queries from a wider stream, dW_k summed over the column chunks with atomics)"""
    H, dk, d_a = 4, 256, 128
    D, Dr = H * dk, H * d_a
    Wq, Wk, bq = rnd(D, d_b, seed=1) * 0.1, rnd(D, d_a, seed=2) * 0.1, rnd(D, seed=3)
    Wq64, Wk64, bq64 = (t.double().requires_grad_(True) for t in (Wq, Wk, bq))
    Wp64 = torch.stack([Wk64[h * dk:(h + 1) * dk].t() @ Wq64[h * dk:(h + 1) * dk] for h in range(H)])       # [H][a][b]
    c64 = torch.stack([bq64[h * dk:(h + 1) * dk] @ Wk64[h * dk:(h + 1) * dk] for h in range(H)])               # [H][a]
    Wqd, Wkd, bqd = Wq.to(DEV), Wk.to(DEV), bq.to(DEV)
    hi = torch.zeros(Dr, d_b, device=DEV, dtype=torch.bfloat16)
    fh, fl = torch.zeros(Dr, d_b, device=DEV, dtype=torch.float16), torch.zeros(Dr, d_b, device=DEV, dtype=torch.float16)
    wp, c = torch.zeros(Dr, d_b, device=DEV), torch.zeros(Dr, device=DEV)
    acc0 = torch.full((Dr, d_b), 5.0, device=DEV)          # (an accumulator of dW' for the launch to zero)
    ops._lib.check(ops.lib.bmt_rank_prep(ops._p(Wqd), d_b, d_b, ops._p(Wkd), d_a, ops._p(bqd), H, dk, d_a, ops._p(hi), ops._p(fh), ops._p(fl), d_b, ops._p(wp),
                                         ops._p(c), ops._p(acc0), ops._st()), "prep")
    assert float(acc0.abs().max()) == 0.0
    want = Wp64.detach().reshape(Dr, d_b)
    assert_close(wp, want, atol=1e-5, rtol=1e-5, name="W' fp32")
    assert_close(fh.float().double() + fl.float().double(), want, atol=2e-6, rtol=2e-6, name="W' fp16 hi + lo")
    assert_close(hi.float(), want, atol=1e-6, rtol=2 ** -8, name="W' bf16")
    assert_close(c, c64.detach().reshape(-1), atol=1e-5, rtol=1e-5, name="c")
    # chain rule: random upstream gradients of W' and c
    gW, gc = rnd(H, d_a, d_b, seed=4), rnd(H, d_a, seed=5)
    (Wp64 * gW.double()).sum().backward(retain_graph=True)
    (c64 * gc.double()).sum().backward()
    dWp = gW.reshape(Dr, d_b).clone().to(DEV)
    dWq, dWk, dbq = torch.ones(D, d_b, device=DEV), torch.ones(D, d_a, device=DEV), torch.ones(D, device=DEV)      # (accumulated into: + 1)
    ops._lib.check(ops.lib.bmt_rank_chain(ops._p(Wqd), d_b, d_b, ops._p(Wkd), d_a, ops._p(bqd), H, dk, d_a, ops._p(dWp), ops._p(gc.reshape(-1).to(DEV)),
                                          ops._p(dWq), d_b, ops._p(dWk), d_a, ops._p(dbq), ops._st()), "chain")
    torch.cuda.synchronize()
    assert_close(dWq - 1.0, Wq64.grad, atol=2e-5, rtol=1e-4, name="dW_q")
    assert_close(dWk - 1.0, Wk64.grad, atol=5e-5, rtol=1e-4, name="dW_k")
    assert_close(dbq - 1.0, bq64.grad, atol=2e-5, rtol=1e-4, name="db_q")


@pytest.mark.parametrize("M,packed", [(300, False), (5000, True)])
def test_block_products(ops, M, packed):
    """bmt_gemm_bf16_args.a_blk_n: out[:, j n : (j + 1) n] = A[:, j k : (j + 1) k] . W_j -- forward against the row blocks of a row-major weight
    (reduction-of-128 kernel, two fp16 planes, bias), backward against the same weight k-major (every block its own reduction rows)"""
    H, dk, d_in = 4, 256, 128
    D = H * dk
    W = rnd(D, d_in, seed=1) * 0.1
    bias = rnd(D, seed=2)
    A = rnd(M, H * d_in, seed=3)
    Ad = A.to(DEV)
    pk, n = None, M
    if packed:
        n = M - 37
        m = (torch.arange(M) < n).view(1, 1, M)
        pk = ops.pack_rows(m.to(DEV))
        Ad[n:] = float("nan")
    Ap = ops.make_planes(Ad, "f16", pack=pk)
    Wp = ops.make_planes(W.to(DEV), "w2")
    o = ops._alloc_planes(M, D, "f16", DEV, ld=D)
    o.hi.fill_(7.0)
    ops.gemm_bf16(Ap, Wp, None, bias=bias.to(DEV), out_planes=o, precision=ops.PREC_F16W2, a_blk=(dk, d_in))
    A16 = A.to(torch.float16).double()
    want = torch.cat([A16[:, h * d_in:(h + 1) * d_in] @ W[h * dk:(h + 1) * dk].double().t() for h in range(H)], 1) + bias.double()
    assert_close(o.fh[:n].float(), want[:n], atol=2e-3, rtol=2e-3, name="forward block product")
    if packed:
        assert bool((o.hi[n:].float() == 7.0).all()), "rows past the pack's count were written"
    # backward: dA[:, j k' : (j + 1) k'] = G[:, j n' : (j + 1) n'] . W_j   (k' = d_in outputs per block, n' = dk reduction rows per block)
    G = rnd(M, D, seed=4)
    Gd = G.to(DEV)
    if packed:
        Gd[n:] = float("nan")
    Gp = ops.make_planes(Gd, "bwd", pack=pk)
    Wb = ops.make_planes(W.to(DEV), "bwd")
    dA = ops.Planes(torch.full((M, H * d_in), 7.0, device=DEV, dtype=torch.bfloat16), None, M, H * d_in)
    ops.gemm_bf16(Gp, ops.Planes(Wb.hi, None, D, d_in), None, out_planes=dA, precision=ops.PREC_BF16, b_km=True, a_blk=(d_in, dk), splitk=1)
    Gb, Wbf = G.to(torch.bfloat16).double(), W.to(torch.bfloat16).double()
    want = torch.cat([Gb[:, h * dk:(h + 1) * dk] @ Wbf[h * dk:(h + 1) * dk] for h in range(H)], 1)
    assert_close(dA.hi[:n].float(), want[:n], atol=2e-2, rtol=2 ** -7, name="backward block product")
    if packed:
        assert bool((dA.hi[n:].float() == 7.0).all())


# ------------------------------------------------------------------------------------------ one attention module, both forms, against fp64
@pytest.mark.parametrize("d_in,S,H,D,packed,holes", [(128, 800, 4, 1024, True, False), (128, 203, 4, 1024, True, True), (128, 150, 4, 1024, False, False),
                                                      (128, 130, 2, 1024, True, True)])
def test_rank_self_attention_against_fp64_and_the_projected_form(ops, d_in, S, H, D, packed, holes):
    B = 4
    att = _module(ops, d_in, H, D)
    m = _mask(B, S, seed=S + d_in, holes=holes)
    X = rnd(B, S, d_in, seed=1) * 0.7 + 0.3
    G = rnd(B, S, d_in, seed=3) * 0.1
    valid = m.view(B, S, 1).float()
    P64 = {n: getattr(getattr(att, ln), pn).detach().double().cpu().requires_grad_(True) for n, ln, pn in _NAMES}
    X64 = X.double().requires_grad_(True)
    y64 = _reference_self_attention(X64, m, P64, H)
    (y64 * valid.double()).backward(G.double())           # (only the valid positions' outputs are consumed downstream)
    rows = torch.nonzero(m.view(-1)).view(-1)

    def run(mod, rank):
        for p_ in mod.parameters():
            p_.grad = None
        if packed:
            xp, _ = _packed(ops, X, m)
            g = torch.zeros(B * S, d_in)
            g[:rows.numel()] = (G * valid).view(-1, d_in)[rows]
            g = g.view(B, S, d_in).to(DEV)
        else:
            xp = X.to(DEV)
            g = (G * valid).to(DEV)
        xp.requires_grad_(True)
        calls, fwd, counting = _counted(ops)
        ops.RANK_ATTN = rank
        ops.RankSelfAttnFn.forward = staticmethod(counting)
        try:
            y = mod(xp, xp, xp, m.to(DEV))
        finally:
            ops.RANK_ATTN = True
            ops.RankSelfAttnFn.forward = staticmethod(fwd)
        assert calls[0] == (1 if rank else 0), "the rank form did not run" if rank else "the projected arm ran the rank form"
        y.backward(g)
        torch.cuda.synchronize()
        if packed:
            yy, gx = torch.zeros(B * S, d_in), torch.zeros(B * S, d_in)
            yy[rows] = y.detach().view(-1, d_in)[:rows.numel()].cpu()
            gx[rows] = xp.grad.view(-1, d_in)[:rows.numel()].cpu()
            return yy.view(B, S, d_in), gx.view(B, S, d_in)
        return y.detach().cpu() * valid, xp.grad.cpu() * valid

    y, gx = run(att, True)
    want = (y64.detach() * valid.double()).float()
    scale = float(want.abs().max())
    assert_close(y, want, atol=2e-3 * scale, rtol=0, name="output")
    e = {"dX": rel_err(gx, (X64.grad * valid.double()).float())}
    for n, ln, pn in _NAMES:
        if n != "bk":
            e[n] = rel_err(getattr(getattr(att, ln), pn).grad.cpu(), P64[n].grad.float())
    print(f"\nrank-form self-attention vs fp64 (d_in {d_in}, S {S}, packed {packed}): out {float((y - want).abs().max()) / scale:.2e}",
          {k: f"{v:.2e}" for k, v in e.items()})
    assert float(att.linear_K2d.bias.grad.abs().max()) == 0.0 and float(P64["bk"].grad.abs().max()) < 1e-9      # zero in both
    assert max(e.values()) < 2e-2, e
    if D // H > 256:          # (the projected form has no attention kernel for heads wider than 256: the rank form is the only one)
        return
    # the projected form of the same module: same bars, and the two agree
    att2 = copy.deepcopy(att)
    y2, gx2 = run(att2, False)
    assert_close(y2, want, atol=2e-3 * scale, rtol=0, name="projected output")
    assert_close(y, y2, atol=3e-3 * scale, rtol=0, name="rank vs projected output")
    e2 = {"dX": rel_err(gx, gx2)}
    for n, ln, pn in _NAMES:
        if n != "bk":
            e2[n] = rel_err(getattr(getattr(att, ln), pn).grad, getattr(getattr(att2, ln), pn).grad)
    print("rank vs projected:", {k: f"{v:.2e}" for k, v in e2.items()})
    assert max(e2.values()) < 3e-2, e2


def test_rank_self_attention_under_dropout_and_a_fused_residual(ops):
    """training mode inside a ResidualConnection: the dropout on the attention output (model/multihead_attention.py:22-23) is drawn in the value
    product's epilogue and re-applied in the out-projection's dX with the element index the projected form's attention kernel uses; the residual's
    dropout and add ride the out-projection's epilogue in both forms -- outputs and gradients of the two forms agree as they do without dropout"""
    from bmt_amd.model.blocks import ResidualConnection
    B, S, d_in, H, D, p = 4, 300, 128, 4, 1024, 0.3
    att = _module(ops, d_in, H, D, p=p, seed=1).train()
    res = ops.tag_policy(ResidualConnection(d_in, p), "enc").to(DEV).train()
    att2, res2 = copy.deepcopy(att), copy.deepcopy(res)
    m = _mask(B, S, seed=5, holes=True)
    X, G = rnd(B, S, d_in, seed=1) * 0.7 + 0.3, rnd(B, S, d_in, seed=3) * 0.1
    rows = torch.nonzero(m.view(-1)).view(-1)
    n = rows.numel()
    ops.manual_seed(77)
    out = []
    for rank, mod, rs in ((True, att, res), (False, att2, res2)):
        xp, _ = _packed(ops, X, m)
        xp.requires_grad_(True)
        g = torch.zeros(B * S, d_in)
        g[:n] = G.view(-1, d_in)[rows]
        ops.RANK_ATTN = rank
        try:
            y = rs(xp, lambda t: mod(t, t, t, m.to(DEV)))
        finally:
            ops.RANK_ATTN = True
        y.backward(g.view(B, S, d_in).to(DEV))
        torch.cuda.synchronize()
        out.append((y.detach().view(-1, d_in)[:n].clone(), xp.grad.view(-1, d_in)[:n].clone(), mod.linear_V2d.weight.grad.clone(),
                    mod.linear_Q2d.weight.grad.clone(), mod.linear_K2d.weight.grad.clone(), mod.linear_d2Q.weight.grad.clone(), rs.norm.weight.grad.clone()))
    scale = float(out[1][0].abs().max())
    assert_close(out[0][0], out[1][0], atol=3e-3 * scale, rtol=0, name="output under dropout, rank vs projected")
    for a, b_, name in zip(out[0][1:], out[1][1:], ("dX", "dW_v", "dW_q", "dW_k", "dW_o", "dgamma")):
        e = rel_err(a, b_)
        print(f"{name}: {e:.2e}")
        assert e < 3e-2, f"{name}: rank vs projected under dropout {e:.3e}"


def test_rank_form_follows_the_optimizer(ops):
    """W' and c are functions of the weights: recomputed when the weights' epoch moves (ops.weights_changed), not before"""
    B, S, d_in, H, D = 2, 96, 128, 4, 1024
    att = _module(ops, d_in, H, D)
    m = _mask(B, S, seed=2)
    X = rnd(B, S, d_in, seed=1)
    xp, _ = _packed(ops, X, m)
    with torch.no_grad():
        y0 = att(xp, xp, xp, m.to(DEV)).clone()
        att.linear_Q2d.weight.mul_(0.5)
        att.linear_Q2d.bias.mul_(0.5)
        ops.weights_changed()
        y1 = att(xp, xp, xp, m.to(DEV)).clone()
    P64 = {n: getattr(getattr(att, ln), pn).detach().double().cpu() for n, ln, pn in _NAMES}
    want = _reference_self_attention(X.double(), m, P64, H)
    rows = torch.nonzero(m.view(-1)).view(-1)
    got = torch.zeros(B * S, d_in)
    got[rows] = y1.view(-1, d_in)[:rows.numel()].cpu()
    valid = m.view(B, S, 1)
    assert_close(got.view(B, S, d_in) * valid, (want * valid).float(), atol=2e-3 * float(want.abs().max()), rtol=0, name="output after the weights moved")
    assert float((y1 - y0).abs().max()) > 1e-3


# ------------------------------------------------------------------------------------------ queries from another stream (video over audio)
def _reference_cross_attention(Y, X, m, P, H):
    """model/multihead_attention.py:55-86 in fp64: queries Y (B, Sq, Dq), keys = values X (B, Sk, d_a) under the key-padding mask m (B, 1, Sk)"""
    q = Y @ P["Wq"].t() + P["bq"]
    k = X @ P["Wk"].t() + P["bk"]
    v = X @ P["Wv"].t() + P["bv"]
    B, Sq, D = q.shape
    dk = D // H
    sp = lambda t: t.view(B, -1, H, dk).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dk)
    s = s.masked_fill(m.unsqueeze(1) == 0, -float("inf"))
    o = (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(B, Sq, D)
    return o @ P["Wo"].t() + P["bo"]


@pytest.mark.parametrize("Sq,Sk,Dq,holes", [(256, 800, 1024, False), (90, 210, 256, True)])
def test_rank_cross_attention_against_fp64_and_the_projected_form(ops, Sq, Sk, Dq, holes):
    """the video stream's attention over the 128-wide audio stream (model/encoders.py:69-79), both streams packed: q' = y W'^T + c with
    W'_h = W_k,h^T W_q,h [128][Dq], the attention at width 128 against the audio stream's own plane"""
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    B, d_a, H, D = 4, 128, 4, 1024
    torch.manual_seed(2)
    att = ops.tag_policy(MultiheadedAttention(Dq, d_a, d_a, H, 0.0, D), "enc").to(DEV)
    with torch.no_grad():
        att.linear_K2d.bias.normal_(0, 0.5)
        att.linear_Q2d.bias.normal_(0, 0.5)
    mq, mk = _mask(B, Sq, seed=Sq, holes=holes), _mask(B, Sk, seed=Sk + 1, holes=holes)
    Y, X = rnd(B, Sq, Dq, seed=1) * 0.5, rnd(B, Sk, d_a, seed=2) * 0.7 + 0.3
    G = rnd(B, Sq, Dq, seed=3) * 0.1
    vq = mq.view(B, Sq, 1).float()
    P64 = {n: getattr(getattr(att, ln), pn).detach().double().cpu().requires_grad_(True) for n, ln, pn in _NAMES}
    Y64, X64 = Y.double().requires_grad_(True), X.double().requires_grad_(True)
    y64 = _reference_cross_attention(Y64, X64, mk, P64, H)
    (y64 * vq.double()).backward(G.double())
    rq, rk = torch.nonzero(mq.view(-1)).view(-1), torch.nonzero(mk.view(-1)).view(-1)

    def run(mod, rank):
        for p_ in mod.parameters():
            p_.grad = None
        yp, _ = _packed(ops, Y, mq)
        xp, _ = _packed(ops, X, mk)
        g = torch.zeros(B * Sq, Dq)
        g[:rq.numel()] = G.view(-1, Dq)[rq]
        yp.requires_grad_(True)
        xp.requires_grad_(True)
        calls = [0]
        fwd = ops.RankCrossAttnFn.forward

        def counting(*a, **kw):
            calls[0] += 1
            return fwd(*a, **kw)
        ops.RANK_CROSS = rank
        ops.RankCrossAttnFn.forward = staticmethod(counting)
        try:
            out = mod(yp, xp, xp, mk.to(DEV))
        finally:
            ops.RANK_CROSS = True
            ops.RankCrossAttnFn.forward = staticmethod(fwd)
        assert calls[0] == (1 if rank else 0)
        out.backward(g.view(B, Sq, Dq).to(DEV))
        torch.cuda.synchronize()
        oo, gy, gx = torch.zeros(B * Sq, Dq), torch.zeros(B * Sq, Dq), torch.zeros(B * Sk, d_a)
        oo[rq] = out.detach().view(-1, Dq)[:rq.numel()].cpu()
        gy[rq] = yp.grad.view(-1, Dq)[:rq.numel()].cpu()
        gx[rk] = xp.grad.view(-1, d_a)[:rk.numel()].cpu()
        return oo.view(B, Sq, Dq), gy.view(B, Sq, Dq), gx.view(B, Sk, d_a)

    y, gy, gx = run(att, True)
    want = (y64.detach() * vq.double()).float()
    scale = float(want.abs().max())
    assert_close(y, want, atol=2e-3 * scale, rtol=0, name="output")
    e = {"dY": rel_err(gy, (Y64.grad * vq.double()).float()), "dX": rel_err(gx, (X64.grad * mk.view(B, Sk, 1).double()).float())}
    for n, ln, pn in _NAMES:
        if n != "bk":
            e[n] = rel_err(getattr(getattr(att, ln), pn).grad.cpu(), P64[n].grad.float())
    print(f"\nrank-form cross-attention vs fp64 (Sq {Sq}, Sk {Sk}, Dq {Dq}): out {float((y - want).abs().max()) / scale:.2e}", {k: f"{v:.2e}" for k, v in e.items()})
    assert float(att.linear_K2d.bias.grad.abs().max()) == 0.0
    assert max(e.values()) < 2e-2, e
    att2 = copy.deepcopy(att)
    y2, gy2, gx2 = run(att2, False)
    assert_close(y2, want, atol=2e-3 * scale, rtol=0, name="projected output")
    assert_close(y, y2, atol=3e-3 * scale, rtol=0, name="rank vs projected output")
    e2 = {"dY": rel_err(gy, gy2), "dX": rel_err(gx, gx2)}
    for n, ln, pn in _NAMES:
        if n != "bk":
            e2[n] = rel_err(getattr(getattr(att, ln), pn).grad, getattr(getattr(att2, ln), pn).grad)
    print("rank vs projected:", {k: f"{v:.2e}" for k, v in e2.items()})
    assert max(e2.values()) < 3e-2, e2
