"""-m gpu: kernel-level parity of libbmt_hip.so (through the C ABI) against the CPU oracle.

Single-pass products (bf16, fp16) are checked against the oracle evaluated on operands ROUNDED to that format in fp64 (so only
the accumulation order differs: tight tolerance that still exposes any tile / lane mapping mistake); the split formats (bf16
hi+lo x3, fp16 activation x fp16 hi+lo weight) against fp64 with only the single-plane operand rounded."""
import math
import os

import numpy as np
import pytest
import torch

from tests.gpu_util import assert_close, bf16_round, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from bmt_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(128, 128, 64), (130, 70, 100), (257, 300, 1024), (64, 10, 24), (1, 1, 1), (300, 1200, 300), (96, 40, 7)]
BF16, X3, F16, F16W2 = 1, 3, 4, 5


def f16_round(t):
    return t.to(torch.float16).to(torch.float32)


def operand_rounding(prec):
    """(rounding of A, rounding of B) the oracle applies to model a product of this precision: the kernel then differs by its fp32
    accumulation order only (and, for the split formats, by the dropped lo.lo term, ~2^-16 / 2^-21 relative)"""
    idt = lambda t: t
    return {BF16: (bf16_round, bf16_round), X3: (idt, idt), F16: (f16_round, f16_round), F16W2: (f16_round, idt)}[prec]


def tol(prec, K):
    return dict(atol=(3e-5 if prec == X3 else 2e-4) * math.sqrt(K), rtol=(2e-5 if prec == X3 else 1e-4))


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("prec", [BF16, X3, F16, F16W2])
def test_gemm_linear_forward(ops, M, N, K, prec):
    x, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = ops.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), precision=prec)
    fa, fb = operand_rounding(prec)
    want = fa(x).double() @ fb(W).double().t() + b.double()
    assert_close(y, want, name=f"linear {ops.prec_name(prec)} {M}x{N}x{K}", **tol(prec, K))


@pytest.mark.parametrize("M,N,K", [(2100, 128, 128), (4099, 1000, 100), (6400, 3072, 128), (2048, 520, 128)])
@pytest.mark.parametrize("prec", [BF16, F16, F16W2])
def test_gemm_reduction_of_128(ops, M, N, K, prec):
    """the weight-chunk-resident kernel (gemm_k128_kernel: Kpad = 128, M >= 2048 -- the audio stream's projections, d_model_audio = 128):
    fp32 output, plane outputs (padded and not), every epilogue it shares with the tile kernels, ragged M / N / K"""
    x, W, b, res = rnd(M, K, seed=31), rnd(N, K, seed=32) * 0.05, rnd(N, seed=33), rnd(M, N, seed=34)
    xd, Wd, bd, rd = x.to(DEV), W.to(DEV), b.to(DEV), res.to(DEV)
    fa, fb = operand_rounding(prec)
    base = fa(x).double() @ fb(W).double().t() + b.double()
    t = tol(prec, K)
    y = ops.linear_fwd(xd, Wd, bd, precision=prec)
    assert_close(y, base, name="bias", **t)
    y = ops.linear_fwd(xd, Wd, bd, relu=True, precision=prec)
    assert_close(y, base.clamp(min=0), name="relu", **t)
    y = ops.linear_fwd(xd, Wd, bd, residual=rd, ldr=N, precision=prec)
    assert_close(y, base + res.double(), name="residual", **t)
    gate = (rnd(M, N, seed=35) > 0).float()
    y = ops.linear_fwd(xd, Wd, None, gate=ops.make_planes(gate.to(DEV), "bwd"), gate_scale=1.25, precision=prec)
    assert_close(y, (base - b.double()) * gate.double() * 1.25, name="gate", **t)
    y0 = ops.linear_fwd(xd, Wd, bd, precision=prec)
    for out_fmt, pad in (("x3", True), ("f16", False), ("f16only", False), ("f16", True)):
        pl = ops.linear_fwd_planes(xd, Wd, bd, precision=prec, out_fmt=out_fmt, pad=pad)
        first = pl.hi if pl.hi is not None else pl.fh
        assert first.shape == (M, (N + 63) // 64 * 64 if pad else N)
        if pl.hi is not None:
            assert torch.equal(pl.hi[:, :N], y0.to(torch.bfloat16)), out_fmt
        if pl.fh is not None:
            assert torch.equal(pl.fh[:, :N], y0.to(torch.float16)), out_fmt
        if pl.lo is not None:
            assert_close(pl.hi[:, :N].float().double() + pl.lo[:, :N].float().double(), y0, atol=1e-6, rtol=2e-5, name="hi+lo planes")
        if pad and N % 64:
            assert float(first[:, N:].float().abs().max()) == 0
    # the fused dropout shares the standalone kernel's mask
    ops.manual_seed(77)
    fused = ops.linear_fwd(xd, Wd, bd, drop_post=True, drop_p=0.25, site=991, precision=prec)
    assert torch.equal(fused, ops.dropout_raw(y0, 0.25, 991))


@pytest.mark.parametrize("M,N,K", [(8192, 1024, 1024), (25600, 384, 128)])
def test_gemm_f16w2_beats_single_pass_on_weight_rounding(ops, M, N, K):
    """the point of PREC_F16W2: the weight enters exactly (hi + lo), so against the UNROUNDED-weight product it is ~2^-11/sqrt-K-accurate
    in the activation only, while one fp16 pass carries the weight's rounding too"""
    x, W = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.03
    xd, Wd = x.to(DEV), W.to(DEV)
    want = f16_round(x).double() @ W.double().t()
    y2 = ops.linear_fwd(xd, Wd, None, precision=F16W2)
    y1 = ops.linear_fwd(xd, Wd, None, precision=F16)
    e2, e1 = rel_err(y2, want), rel_err(y1, want)
    assert e2 < 5e-6 and e1 > 10 * e2, (e1, e2)


@pytest.mark.parametrize("M,N,K", [(130, 70, 100), (256, 128, 512), (33, 300, 1000), (960, 300, 10000)])
def test_gemm_dx_dw_layouts(ops, M, N, K):
    """dX = dY.W and dW = dY^T.X (reduction over rows): every backward operand is a bf16 plane as stored, read k-major."""
    dy, W, x = rnd(M, N, seed=4), rnd(N, K, seed=5), rnd(M, K, seed=6)
    dyd, xd = dy.to(DEV), x.to(DEV)
    dyP, _ = ops.grad_planes(dyd)
    dx = ops.linear_dx(dyP, W.to(DEV))
    want = bf16_round(dy).double() @ bf16_round(W).double()
    assert_close(dx, want, atol=2e-4 * math.sqrt(N), rtol=1e-4, name="dx")
    dW = ops.linear_dw(dyP, ops.bwd_planes(xd))
    want = bf16_round(dy).double().t() @ bf16_round(x).double()
    assert_close(dW, want, atol=2e-4 * math.sqrt(M), rtol=1e-4, name="dw")


@pytest.mark.parametrize("M,N,K", [(6400, 128, 1024), (4100, 100, 512), (25600, 128, 576)])
def test_gemm_one_column_tile_over_many_rows(ops, M, N, K):
    """ONE column tile over 4096 .. 32767 rows (the audio stream's 25600 x 128 outputs; with BMT_GEMM_SHORT=3 the 64-row tiles of
    gemm_prepare): the fp16 forward products with their epilogues, and the k-major dX product"""
    x, W, b, res = rnd(M, K, seed=41), rnd(N, K, seed=42) * 0.05, rnd(N, seed=43), rnd(M, N, seed=44)
    xd, Wd, bd, rd = x.to(DEV), W.to(DEV), b.to(DEV), res.to(DEV)
    for prec in (F16, F16W2):
        fa, fb = operand_rounding(prec)
        base = fa(x).double() @ fb(W).double().t() + b.double()
        t = tol(prec, K)
        assert_close(ops.linear_fwd(xd, Wd, bd, precision=prec), base, name="bias", **t)
        assert_close(ops.linear_fwd(xd, Wd, bd, relu=True, precision=prec), base.clamp(min=0), name="relu", **t)
        assert_close(ops.linear_fwd(xd, Wd, bd, residual=rd, ldr=N, precision=prec), base + res.double(), name="residual", **t)
        y0 = ops.linear_fwd(xd, Wd, bd, precision=prec)
        pl = ops.linear_fwd_planes(xd, Wd, bd, precision=prec, out_fmt="f16", pad=True)
        assert torch.equal(pl.hi[:, :N], y0.to(torch.bfloat16)) and torch.equal(pl.fh[:, :N], y0.to(torch.float16))
        ops.manual_seed(5)
        fused = ops.linear_fwd(xd, Wd, bd, drop_post=True, drop_p=0.25, site=77, precision=prec)
        assert torch.equal(fused, ops.dropout_raw(y0, 0.25, 77))
    # dX = dY . W with W [K][N] as stored (k-major B): dY [M][K], output [M][N] -- one column tile
    dy, Wb = rnd(M, K, seed=45), rnd(K, N, seed=46) * 0.05
    dyP, _ = ops.grad_planes(dy.to(DEV))
    dx = ops.linear_dx(dyP, Wb.to(DEV))
    want = bf16_round(dy).double() @ bf16_round(Wb).double()
    assert_close(dx, want, atol=2e-4 * math.sqrt(K), rtol=1e-4, name="dx")


@pytest.mark.parametrize("M,N,K,splitk", [(300, 200, 1024, 4), (960, 300, 1024, 8), (130, 70, 2048, 16), (257, 129, 640, 3), (960, 300, 1024, 0)])
@pytest.mark.parametrize("prec", [BF16, X3, F16W2])
def test_gemm_splitk_two_pass(ops, M, N, K, splitk, prec):
    """two-pass split-K (partials in the workspace + epilogue kernel): same result as one pass for any epilogue, bit-identical
    between launches (partials are summed in split order); plane outputs (bf16 hi + bf16 lo | fp16) from the epilogue kernel"""
    x, W, b, res = rnd(M, K, seed=7), rnd(N, K, seed=8), rnd(N, seed=9), rnd(M, N, seed=10)
    A = ops.make_planes(x.to(DEV), "all")
    Bw = ops.make_planes(W.to(DEV), "all")
    bd, resd = b.to(DEV), res.to(DEV)
    out_fmt = "f16" if prec == F16W2 else "x3"
    outs = []
    for sk in (1, splitk, splitk):
        out = torch.empty(M, N, device=DEV)
        pl = ops._alloc_planes(M, N, out_fmt, DEV)
        ops.gemm_bf16(A, Bw, out, ldc=N, bias=bd, relu=True, residual=resd, ldr=N, splitk=sk, precision=prec, out_planes=pl)
        outs.append((out, pl))
    fa, fb = operand_rounding(prec)
    want = torch.relu(fa(x).double() @ fb(W).double().t() + b.double()) + res.double()
    for out, pl in outs:
        assert_close(out, want, atol=(3e-4 if prec == X3 else 2e-4) * math.sqrt(K), rtol=1e-4, name=f"splitk out {ops.prec_name(prec)}")
        assert torch.equal(pl.hi[:, :N], out.to(torch.bfloat16)), "plane output of the split-K path != bf16(fp32 output)"
        if out_fmt == "f16":
            assert torch.equal(pl.fh[:, :N], out.to(torch.float16)), "fp16 plane output != fp16(fp32 output)"
    assert torch.equal(outs[1][0], outs[2][0]), "split-K result differs between two launches (reduction order not fixed?)"
    assert_close(outs[1][0], outs[0][0], atol=1e-4 * math.sqrt(K), rtol=1e-5, name="split vs single pass")


@pytest.mark.parametrize("M,N,K", [(130, 70, 100), (256, 128, 512), (33, 300, 1000), (960, 300, 1030), (928, 1024, 300)])
def test_gemm_kmajor_operands(ops, M, N, K):
    """operands given with the reduction index as their row (read through ds_read_b64_tr_b16): dX = dY . W with the weight plane
    as stored, dW = dY^T . X with gradient and activation planes as stored -- no transposed planes anywhere"""
    dy, W, x = rnd(M, N, seed=14), rnd(N, K, seed=15), rnd(M, K, seed=16)
    dyP = ops.make_planes(dy.to(DEV), "bwd")
    WP = ops.make_planes(W.to(DEV), "bwd")
    xP = ops.make_planes(x.to(DEV), "bwd")
    dx = torch.empty(M, K, device=DEV)
    ops.gemm_bf16(dyP, WP, dx, ldc=K, precision=BF16, b_km=True)                      # reduction over N: W [N rows][K cols] is k-major
    want = bf16_round(dy).double() @ bf16_round(W).double()
    assert_close(dx, want, atol=2e-4 * math.sqrt(N), rtol=1e-4, name="dx (k-major W)")
    for sk in (1, 3):
        dW = torch.empty(N, K, device=DEV)
        ops.gemm_bf16(dyP, xP, dW, ldc=K, precision=BF16, a_km=True, b_km=True, splitk=sk)   # reduction over M (rows of both planes)
        want = bf16_round(dy).double().t() @ bf16_round(x).double()
        assert_close(dW, want, atol=2e-4 * math.sqrt(M), rtol=1e-4, name=f"dW (k-major dY and X, splitk={sk})")


@pytest.mark.parametrize("R,C", [(150, 70), (64, 1024), (33, 300)])
def test_planes_formats(ops, R, C):
    """bmt_planes: bf16 hi / lo and fp16 hi / lo of one tensor in one pass, zero padded to the next multiple of 64"""
    x = rnd(R, C, seed=3) * 3
    pl = ops.make_planes(x.to(DEV), "all")
    ld = ops._pad64(C)
    for t in (pl.hi, pl.lo, pl.fh, pl.fl):
        assert t.shape == (R, ld) and float(t[:, C:].float().abs().max() if ld > C else 0.0) == 0
    assert torch.equal(pl.hi[:, :C].float().cpu(), x.to(torch.bfloat16).float())
    assert torch.equal(pl.fh[:, :C].float().cpu(), x.to(torch.float16).float())
    assert_close(pl.hi[:, :C].float() + pl.lo[:, :C].float(), x, atol=0, rtol=2 ** -15, name="bf16 hi+lo")
    assert_close(pl.fh[:, :C].float() + pl.fl[:, :C].float(), x, atol=1e-7, rtol=2 ** -20, name="fp16 hi+lo")
    for fmt, names in (("bwd", ("hi",)), ("x3", ("hi", "lo")), ("f16", ("hi", "fh")), ("w2", ("hi", "fh", "fl"))):
        one = ops.make_planes(x.to(DEV), fmt)
        for n in ("hi", "lo", "fh", "fl"):
            assert (getattr(one, n) is not None) == (n in names)
            if n in names:
                assert torch.equal(getattr(one, n), getattr(pl, n)), (fmt, n)


def test_gemm_epilogues(ops):
    M, N, K = 150, 90, 64
    x, W, b, res = rnd(M, K, seed=7), rnd(N, K, seed=8), rnd(N, seed=9), rnd(M, N, seed=10)
    xd, Wd, bd, rd = x.to(DEV), W.to(DEV), b.to(DEV), res.to(DEV)
    base = x.double() @ W.double().t() + b.double()
    y = ops.linear_fwd(xd, Wd, bd, relu=True, precision=X3)
    assert_close(y, base.clamp(min=0), atol=3e-4, name="relu")
    y = ops.linear_fwd(xd, Wd, bd, residual=rd, ldr=N, precision=X3)
    assert_close(y, base + res.double(), atol=3e-4, name="residual")
    gate = (rnd(M, N, seed=11) > 0).float()
    y = ops.linear_fwd(xd, Wd, None, gate=ops.make_planes(gate.to(DEV), "bwd"), gate_scale=1.25, precision=X3)
    assert_close(y, (x.double() @ W.double().t()) * gate.double() * 1.25, atol=3e-4, name="gate")
    # in-place accumulate through the residual pointer (C aliases residual)
    acc = rd.clone()
    ops.linear_fwd(xd, Wd, None, out=acc, residual=acc, ldr=N, precision=X3)
    assert_close(acc, res.double() + x.double() @ W.double().t(), atol=3e-4, name="residual-alias")
    cs = ops.colsum(rd)
    assert_close(cs, res.double().sum(0), atol=1e-4, name="colsum")


@pytest.mark.parametrize("M,N,K", [(928, 900, 300), (928, 300, 1024), (928, 300, 1200), (928, 1024, 300), (928, 1200, 300), (32, 1024, 300), (29, 10, 1500),
                                   (928, 300, 600), (1, 300, 300)])
def test_gemm_small_tiles_at_the_decoders_shapes(ops, M, N, K):
    """the three-pass product of <= 1.5 M outputs runs on 32 x 32 tiles with the reduction split over a workgroup's waves
    (gemm_small_kernel): every epilogue operation, both output forms and the column sums at the shapes of a decoder layer"""
    x, W, b, res = rnd(M, K, seed=21), rnd(N, K, seed=22) * 0.2, rnd(N, seed=23), rnd(M, N, seed=24)
    xd, Wd, bd, rd = x.to(DEV), W.to(DEV), b.to(DEV), res.to(DEV)
    base = x.double() @ W.double().t() + b.double()
    t = tol(X3, K)
    assert_close(ops.linear_fwd(xd, Wd, bd, precision=X3), base, name="plain", **t)
    assert_close(ops.linear_fwd(xd, Wd, bd, relu=True, residual=rd, ldr=N, alpha=1.0, precision=X3), base.clamp(min=0) + res.double(), name="relu+residual", **t)
    gate = (rnd(M, N, seed=25) > 0).float()
    y = ops.linear_fwd(xd, Wd, None, gate=ops.make_planes(gate.to(DEV), "bwd"), gate_scale=0.5, precision=X3)
    assert_close(y, (x.double() @ W.double().t()) * gate.double() * 0.5, name="gate", **t)
    # planes + column sums without an fp32 output (a dX-style consumer)
    A, Bw = ops.as_planes(xd, ops.act_fmt(X3)), ops.weight_planes(Wd, ops.weight_fmt(X3))
    op = ops._alloc_planes(M, N, "x3", DEV, ld=ops._pad64(N))
    op.hi.fill_(3.0)
    op.lo.fill_(3.0)
    cs = torch.zeros(N, device=DEV)
    ops.gemm_bf16(A, Bw, None, bias=bd, out_planes=op, precision=X3, colsum=cs)
    y = ops.linear_fwd(xd, Wd, bd, precision=X3)
    assert torch.equal(op.hi[:, :N], y.to(torch.bfloat16))
    assert_close(op.hi[:, :N].float().double() + op.lo[:, :N].float().double(), y, atol=1e-6, rtol=2e-5, name="hi+lo planes")
    assert float(op.hi[:, N:].float().abs().max() if op.hi.shape[1] > N else 0.0) == 0.0
    assert_close(cs, y.double().sum(0), atol=2e-3 * math.sqrt(M), rtol=1e-5, name="column sums")
    # accumulate into an existing buffer
    acc = rd.clone()
    ops.gemm_bf16(A, Bw, acc, ldc=N, accum=True, precision=X3)
    assert_close(acc, res.double() + x.double() @ W.double().t(), name="accumulate", **t)
    # dropout: the same mask as the standalone kernel
    ops.manual_seed(77)
    plain = ops.linear_fwd(xd, Wd, None, precision=X3)
    fused = ops.linear_fwd(xd, Wd, None, drop_post=True, drop_p=0.25, site=991, precision=X3)
    assert torch.equal(fused, ops.dropout_raw(plain, 0.25, 991))
    # dX = dY . W of the same layer: one bf16 pass, row-major through the weight's transposed plane (ops.weight_planes_t)
    dy = rnd(M, N, seed=26)
    dx = ops.linear_dx(ops.make_planes(dy.to(DEV), "bwd"), Wd)
    assert_close(dx, bf16_round(dy).double() @ bf16_round(W).double(), atol=2e-4 * math.sqrt(N), rtol=1e-4, name="dx")
    Wd.add_(1.0)                     # the transposed plane follows the weight (refreshed with the others after an optimizer step)
    ops.weights_changed()
    dx2 = ops.linear_dx(ops.make_planes(dy.to(DEV), "bwd"), Wd)
    assert_close(dx2, bf16_round(dy).double() @ bf16_round(W + 1.0).double(), atol=4e-4 * math.sqrt(N), rtol=1e-4, name="dx after the weight moved")


@pytest.mark.parametrize("prec", [X3, F16W2])
def test_gemm_dropout_epilogue_matches_standalone(ops, prec):
    """the fused dropout epilogue and bmt_dropout share one RNG: same site => same mask."""
    M, N, K, p, site = 200, 96, 32, 0.3, 4242
    ops.manual_seed(123)
    x, W = rnd(M, K, seed=12).to(DEV), rnd(N, K, seed=13).to(DEV)
    plain = ops.linear_fwd(x, W, None, precision=prec)
    fused = ops.linear_fwd(x, W, None, drop_post=True, drop_p=p, site=site, precision=prec)
    sep = ops.dropout_raw(plain, p, site)
    assert torch.equal(fused, sep)
    keep = (fused != 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.02, keep
    kept = fused != 0
    assert_close(fused[kept], plain[kept] / (1 - p), atol=1e-5, rtol=1e-5, name="scale")
    # another site / another step -> another mask
    other = ops.dropout_raw(plain, p, site + 1)
    assert not torch.equal(other, sep)
    ops.rng_advance()
    assert not torch.equal(ops.dropout_raw(plain, p, site), sep)


# ------------------------------------------------------------------------------------------ attention
def _oracle_attention(q, k, v, mask, H, rounded):
    """rounded: False / 3 = exact operands, True / 1 = bf16-rounded, 4 = fp16-rounded"""
    from oracle import bmt_oracle as orc
    B, Sq, D = q.shape
    dk = D // H
    f = _round_fn({False: 3, True: 1}.get(rounded, rounded))
    qh = f(q).view(B, Sq, H, dk).transpose(1, 2)
    kh = f(k).view(B, -1, H, dk).transpose(1, 2)
    vh = f(v).view(B, -1, H, dk).transpose(1, 2)
    m = None if mask is None else mask.unsqueeze(1)
    o = orc.attention(qh, kh, vh, m)
    return o.transpose(1, 2).reshape(B, Sq, D)


def _masks(kind, B, Sq, Sk):
    if kind == "none":
        return None
    if kind == "pad":
        m = torch.ones(B, 1, Sk, dtype=torch.bool)
        for b in range(B):
            m[b, 0, max(1, Sk - 3 - 7 * b):] = False
        return m
    if kind == "causal":
        assert Sq == Sk
        m = torch.tril(torch.ones(Sq, Sk, dtype=torch.bool)).unsqueeze(0).repeat(B, 1, 1)
        m[0, :, Sk - 2:] = False
        return m
    raise ValueError(kind)


ATTN_CASES = [  # dk, H, B, Sq, Sk, mask
    (32, 4, 2, 7, 11, "pad"), (32, 4, 2, 7, 7, "causal"), (32, 2, 1, 130, 200, "pad"), (64, 2, 2, 50, 77, "none"),
    (128, 2, 2, 33, 130, "pad"), (256, 2, 1, 40, 70, "pad"), (256, 1, 2, 30, 30, "causal"), (256, 4, 1, 129, 64, "none"),
]


@pytest.mark.parametrize("dk,H,B,Sq,Sk,kind", ATTN_CASES)
@pytest.mark.parametrize("prec", [1, 3])
def test_attention_forward(ops, dk, H, B, Sq, Sk, kind, prec):
    D = dk * H
    q, k, v = rnd(B, Sq, D, seed=20), rnd(B, Sk, D, seed=21), rnd(B, Sk, D, seed=22)
    mask = _masks(kind, B, Sq, Sk)
    o, lse = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), None if mask is None else mask.to(DEV), H, precision=prec)
    want = _oracle_attention(q, k, v, mask, H, rounded=(prec == 1))
    # x1: P is rounded to bf16 inside the kernel as well -> 2^-9 relative on O
    assert_close(o, want, atol=(2e-2 if prec == 1 else 2e-4), rtol=0, name=f"attn o dk={dk} {kind} x{prec}")
    # lse against a direct evaluation
    f = (lambda t: bf16_round(t).double()) if prec == 1 else (lambda t: t.double())
    s = torch.einsum("bqhd,bkhd->bhqk", f(q).view(B, Sq, H, dk), f(k).view(B, Sk, H, dk)) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(~mask.unsqueeze(1), -float("inf"))
    assert_close(lse, torch.logsumexp(s, -1), atol=2e-3 if prec == 1 else 2e-4, name="lse")


@pytest.mark.parametrize("dk,H", [(256, 2), (128, 2), (64, 2)])
@pytest.mark.parametrize("prec", [1, 3, 4])
def test_attention_forward_rescale_branch(ops, dk, H, prec):
    """the online softmax keeps a STALE reference and rescales only when a score exceeds it by e^8: a rare, data-dependent branch
    that bounded random data never takes after the first tiles.  Force it late and repeatedly: a few (query, key) pairs far apart
    along the key axis get scores 10, 25, 45 above everything before them; also rows whose maximum sits in the FIRST tile (the
    branch is never taken again) and a row of large negative scores."""
    B, Sq, Sk = 2, 70, 300
    D = dk * H
    q, k, v = rnd(B, Sq, D, seed=50) * 0.3, rnd(B, Sk, D, seed=51) * 0.3, rnd(B, Sk, D, seed=52)
    scale = math.sqrt(dk)
    for (b, qi, ki, boost) in ((0, 3, 40, 10.0), (0, 3, 150, 25.0), (0, 3, 290, 45.0), (1, 17, 5, 60.0), (1, 33, 299, 30.0), (0, 64, 200, 12.0)):
        for h in range(H):
            sl = slice(h * dk, (h + 1) * dk)
            qv = q[b, qi, sl]
            k[b, ki, sl] = qv / qv.norm() ** 2 * boost * scale      # q . k / sqrt(dk) == boost
    q[1, 50] *= -40.0                                                 # a row of large-magnitude scores of both signs
    mask = torch.ones(B, 1, Sk, dtype=torch.bool)
    mask[1, 0, 280:295] = False
    (qh, ql), (kh, kl), (vh, vl) = _planes(q, prec), _planes(k, prec), _planes(v, prec)
    o, lse = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask.to(DEV), H, precision=prec)
    want = _oracle_attention(q, k, v, mask, H, rounded=prec)
    assert_close(o, want, atol=_ATOL_O[prec] * 1.5, rtol=0, name=f"attn o with forced rescales dk={dk} x{prec}")
    f = _round_fn(prec)
    s_ = torch.einsum("bqhd,bkhd->bhqk", f(q).view(B, Sq, H, dk), f(k).view(B, Sk, H, dk)) / math.sqrt(dk)
    s_ = s_.masked_fill(~mask.unsqueeze(1), -float("inf"))
    assert_close(lse, torch.logsumexp(s_, -1), atol={1: 5e-2, 3: 1e-3, 4: 8e-3}[prec], rtol=1e-5, name="lse with forced rescales")


@pytest.mark.parametrize("dk,H,B,Sq,Sk", [(256, 4, 16, 1000, 333), (128, 8, 8, 1024, 300), (256, 4, 20, 800, 45)])
@pytest.mark.parametrize("prec", [1, 4])
def test_attention_forward_32_query_kernel(ops, dk, H, B, Sq, Sk, prec):
    """problems of >= 2 workgroups per CU (here >= 512 query tiles of 128) with a key-padding mask run attn_fwd32_kernel: 32 queries per
    wave, K / V by LDS-DMA, the mask row in LDS.  Same checks as the 16-query kernels get: ragged valid lengths incl. fully masked
    key tiles and a partly valid last tile, forced late rescales of the stale-maximum softmax, the last query tile past Sq, and the
    dropout mask of the standalone kernel."""
    D = dk * H
    ops.manual_seed(5)
    q, k, v = rnd(B, Sq, D, seed=60) * 0.5, rnd(B, Sk, D, seed=61) * 0.5, rnd(B, Sk, D, seed=62)
    scale = math.sqrt(dk)
    for (b, qi, ki, boost) in ((0, 3, 5, 9.0), (0, 3, Sk // 2, 24.0), (0, 3, Sk - 2, 44.0), (1, 700, 2, 50.0), (2, Sq - 1, Sk // 3, 30.0)):
        for h in range(H):
            sl = slice(h * dk, (h + 1) * dk)
            qv = q[b, qi, sl]
            k[b, ki, sl] = qv / qv.norm() ** 2 * boost * scale      # q . k / sqrt(dk) == boost
    mask = torch.ones(B, 1, Sk, dtype=torch.bool)
    for b in range(1, B):
        mask[b, 0, max(3, Sk - 5 * b - (Sk // 2 if b % 4 == 3 else 0)):] = False      # some rows lose whole 32-key tiles
    (qh, ql), (kh, kl), (vh, vl) = _planes(q, prec), _planes(k, prec), _planes(v, prec)
    o, lse = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask.to(DEV), H, precision=prec)
    want = _oracle_attention(q, k, v, mask, H, rounded=prec)
    assert_close(o, want, atol=_ATOL_O[prec] * 1.5, rtol=0, name=f"attn32 o dk={dk} prec {prec}")
    f = _round_fn(prec)
    s_ = torch.einsum("bqhd,bkhd->bhqk", f(q).view(B, Sq, H, dk), f(k).view(B, Sk, H, dk)) / math.sqrt(dk)
    s_ = s_.masked_fill(~mask.unsqueeze(1), -float("inf"))
    assert_close(lse, torch.logsumexp(s_, -1), atol={1: 5e-2, 4: 8e-3}[prec], rtol=1e-5, name="attn32 lse")
    # dropout on the output: exactly the standalone kernel's mask on this kernel's values
    od, _ = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask.to(DEV), H, drop_p=0.25, site=11, precision=prec)
    assert torch.equal(od, ops.dropout_raw(o, 0.25, 11))


def _planes(t, prec=3):
    """(first plane, second plane) a raw attention-kernel call of this precision takes: bf16 hi + lo, or the fp16 plane"""
    if prec == 4:
        return t.to(torch.float16).to(DEV), None
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.to(DEV), lo.to(DEV)


def _round_fn(prec):
    return {1: lambda t: bf16_round(t).double(), 3: lambda t: t.double(), 4: lambda t: f16_round(t).double()}[prec]


# max |O - oracle on rounded operands|: single-pass kernels also round P (2^-9 bf16, 2^-12 fp16)
_ATOL_O = {1: 2e-2, 3: 2e-4, 4: 3e-3}


@pytest.mark.parametrize("dk,H,B,Sq,Sk,kind", ATTN_CASES + [(256, 2, 2, 200, 333, "pad"), (128, 4, 1, 70, 257, "pad"), (256, 1, 1, 256, 800, "pad")])
@pytest.mark.parametrize("prec", [1, 3, 4])
def test_attention_forward_bf16_planes(ops, dk, H, B, Sq, Sk, kind, prec):
    D = dk * H
    q, k, v = rnd(B, Sq, D, seed=20), rnd(B, Sk, D, seed=21), rnd(B, Sk, D, seed=22)
    mask = _masks(kind, B, Sq, Sk)
    if kind == "pad" and Sk > 100:
        mask[0, 0, Sk // 3:] = False          # whole key tiles fully masked -> exercises the tile-skip path
    (qh, ql), (kh, kl), (vh, vl) = _planes(q, prec), _planes(k, prec), _planes(v, prec)
    o, lse = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, None if mask is None else mask.to(DEV), H, precision=prec)
    want = _oracle_attention(q, k, v, mask, H, rounded=prec)
    assert_close(o, want, atol=_ATOL_O[prec], rtol=0, name=f"attn(planes) o dk={dk} {kind} prec {prec}")
    f = _round_fn(prec)
    s = torch.einsum("bqhd,bkhd->bhqk", f(q).view(B, Sq, H, dk), f(k).view(B, Sk, H, dk)) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(~mask.unsqueeze(1), -float("inf"))
    assert_close(lse, torch.logsumexp(s, -1), atol={1: 2e-3, 3: 2e-4, 4: 5e-4}[prec], name="lse")


@pytest.mark.parametrize("dk,H,B,Sq,Sk,kind", ATTN_CASES + [(256, 2, 2, 200, 333, "pad"), (128, 4, 1, 70, 257, "pad")])
def test_attention_backward_bf16_planes(ops, dk, H, B, Sq, Sk, kind):
    D = dk * H
    q, k, v = rnd(B, Sq, D, seed=30), rnd(B, Sk, D, seed=31), rnd(B, Sk, D, seed=32)
    do = rnd(B, Sq, D, seed=33)
    mask = _masks(kind, B, Sq, Sk)
    if kind == "pad" and Sk > 100:
        mask[0, 0, Sk // 3:] = False
    md = None if mask is None else mask.to(DEV)
    (qh, ql), (kh, kl), (vh, vl) = _planes(q), _planes(k), _planes(v)
    o, lse = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, md, H, precision=3)
    dq, dk_, dv = ops.attn_bwd_bf16(qh, kh, vh, o, do.to(DEV), lse, md, H)
    qr, kr, vr = (t.clone().double().requires_grad_() for t in (q, k, v))
    want = _oracle_attention(qr, kr, vr, mask, H, rounded=False)
    (want * do.double()).sum().backward()
    from tests.gpu_util import report
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk_, kr.grad), ("dv", dv, vr.grad)):
        e = rel_err(got, ref)
        assert e < 2e-2, f"{name} dk={dk} {kind}: relative error {e:.3e}\n" + report(got, ref, name)


@pytest.mark.parametrize("B,H,Sq,Sk,dk", [(2, 4, 200, 200, 256), (2, 4, 12, 200, 256), (2, 8, 70, 130, 128)])
def test_attention_backward_from_fp16_planes(ops, B, H, Sq, Sk, dk):
    """q / k / v saved as fp16 planes only (the fp16 attention policy, d_k >= 128): the backward kernels and the mean-key kernel
    convert them to bf16 while staging.  Same gradients as from the bf16 planes up to the double rounding fp32 -> fp16 -> bf16."""
    D = H * dk
    mk = lambda S, seed: ops.make_planes((rnd(B * S, D, seed=seed) * 0.5).to(DEV), "f16")
    q, k, v = mk(Sq, 81), mk(Sk, 82), mk(Sk, 83)
    mask = torch.ones(B, 1, Sk, dtype=torch.bool)
    mask[0, 0, Sk - 7:] = False
    md = mask.to(DEV)
    o, lse = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, md, H, precision=ops.PREC_F16, out_fmt="f16")
    do = ops.make_planes(rnd(B * Sq, D, seed=84).to(DEV), "bwd")
    do = ops.Planes(do.hi[:, :D].contiguous(), None, B * Sq, D)
    ref = ops.attn_bwd_planes(q.only("hi"), k.only("hi"), v.only("hi"), o, do, lse, B, Sq, Sk, D, md, H, 0.0, (None, None, None))
    f16 = lambda pl: ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh)
    got = ops.attn_bwd_planes(f16(q), f16(k), f16(v), o, do, lse, B, Sq, Sk, D, md, H, 0.0, (None, None, None))
    for name, (a, _), (b, _) in zip(("dq", "dk", "dv"), ref[:3], got[:3]):
        e = rel_err(b.hi.float(), a.hi.float())
        assert e < 1e-2, f"{name}: fp16-plane backward differs from the bf16-plane backward by {e:.3e}"
    ma = ops._mask_args(md, B, Sq, Sk)
    km_b = ops.attn_kmean(k.hi, D, Sk * D, B, Sk, D, ma, f16=False)
    km_h = ops.attn_kmean(k.fh, D, Sk * D, B, Sk, D, ma, f16=True)
    assert_close(km_h, km_b, atol=2e-3, name="mean key from the fp16 plane")


@pytest.mark.parametrize("form", ["recompute", "emit"])
@pytest.mark.parametrize("B,H,Sq,Sk,dk,short", [(2, 4, 800, 800, 256, 300), (2, 4, 256, 800, 256, 97), (2, 4, 800, 256, 256, 128), (3, 2, 130, 45, 256, None),
                                                 (2, 8, 300, 333, 128, 100), (2, 4, 64, 256, 256, 1), (2, 2, 257, 129, 128, 33)])
def test_attention_backward_split_form(ops, B, H, Sq, Sk, dk, short, form, monkeypatch):
    """the SPLIT backward (Sq >= 64, d_k >= 128, fp16 q / k / v planes: the encoder's attentions), both forms.  "emit" (rounds 3-5): the dQ
    kernel leaves P, dS and a scaled bf16 copy of q in workspaces, dK / dV are two plain products over them.  "recompute" (round 6, the
    default): the dQ kernel leaves delta, live-query bits and max |dO| only, the key-side kernel keeps V rows in registers and the K block in
    LDS, streams q / dO and rebuilds S, P, dP and dS (one power-of-two scale per (batch, head) on the fp16 dS).  In both the bias
    gradients are summed from per-tile partials.  Against
    fp64 autograd on the fp16-rounded operands AND against the two-kernel form on the same inputs (BMT_ATTN_BWD_SPLIT=0); ragged prefix
    masks with whole 32- and 128-key tiles masked, lengths that are no multiple of the tiles, dO spanning five decades from row to row
    (the per-query power-of-two scale of the fp16 gradient operands), keys with a common component (the mean-key correction)."""
    D = H * dk
    q = rnd(B * Sq, D, seed=301) * 0.7
    k = rnd(B * Sk, D, seed=302) * 0.7 + 0.4
    v = rnd(B * Sk, D, seed=303)
    g = torch.Generator().manual_seed(304)
    rowscale = 10.0 ** (-1.0 - 5.0 * torch.rand(B * Sq, 1, generator=g))
    do = rnd(B * Sq, D, seed=305) * rowscale
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    if short is not None:
        lens[1] = short
    mask = (torch.arange(Sk)[None, :] < lens[:, None]).view(B, 1, Sk)
    md = mask.to(DEV)
    qp, kp, vp = (ops.make_planes(t.to(DEV), "f16") for t in (q, k, v))
    f16 = lambda pl: ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh)
    o, lse = ops.attn_fwd_planes(qp, kp, vp, B, Sq, Sk, D, md, H, precision=ops.PREC_F16, out_fmt="f16")
    dop = ops.make_planes(do.to(DEV), "bwd")
    dop = ops.Planes(dop.hi[:, :D].contiguous(), None, B * Sq, D)
    biases = tuple(torch.zeros(D, device=DEV, requires_grad=True) for _ in range(3))

    def run(split):
        monkeypatch.setattr(ops, "ATTN_BWD_SPLIT", split is not False)
        monkeypatch.setattr(ops, "ATTN_BWD_RECOMPUTE", split == "recompute")
        r = ops.attn_bwd_planes(f16(qp), f16(kp), f16(vp), o, dop, lse, B, Sq, Sk, D, md, H, 0.0, biases)
        torch.cuda.synchronize()
        return [(pl.hi[:, :D].float().cpu(), db.cpu()) for pl, db in r[:3]]
    new, old = run(form), run(False)
    # fp64 autograd on what the kernels were given: fp16 q / k / v, the bf16 plane of dO
    qr, kr, vr = (pl.fh[:, :D].double().cpu().view(B, S_, D).requires_grad_() for pl, S_ in ((qp, Sq), (kp, Sk), (vp, Sk)))
    want = _oracle_attention(qr, kr, vr, mask, H, rounded=False)
    (want * dop.hi.double().cpu().view(B, Sq, D)).sum().backward()
    from tests.gpu_util import report
    for name, (gn, bn), (go, bo), ref in zip(("dq", "dk", "dv"), new, old, (qr.grad, kr.grad, vr.grad)):
        ref2 = ref.reshape(-1, D)
        assert torch.isfinite(gn).all() and torch.isfinite(bn).all(), f"{name}: non-finite values"
        en, eo = rel_err(gn, ref2), rel_err(go, ref2)
        # (the recompute form's dP = dO . V^T runs on bf16 V like the two-kernel form's: where attention is peaked -- one valid key: P = 1,
        # dP = delta up to rounding -- dS is the difference of two nearly equal numbers and inherits V's 2^-9; the emitting form keeps fp16 V there)
        bar = 6e-3 if form == "emit" else max(6e-3, 1.05 * eo)
        assert en < bar and en < 1.25 * eo + 1e-3, f"{name} (B{B} H{H} {Sq}x{Sk} d_k {dk}): split {en:.3e}, two-kernel {eo:.3e}\n" + report(gn, ref2, name)
        bref = ref2.sum(0)
        if name != "dk":      # (sum_j dS_ij = 0: the bias gradient of the key projection is zero up to rounding, no relative bar)
            assert rel_err(bn, bref) < 8e-3, f"bias gradient of {name}: {rel_err(bn, bref):.3e}"
        else:
            # (recompute form: the peaked-attention residue above also lands in the key bias -- 4x the bar)
            assert float((bn.double() - bref).abs().max()) < (5e-3 if form == "emit" else 2e-2) * float(ref2.abs().max()) * (B * Sk) ** 0.5
    # masked keys get exactly zero gradients
    dead = (~mask.view(B, Sk)).reshape(-1)
    assert float(new[1][0][dead].abs().max() if dead.any() else 0.0) == 0.0 and float(new[2][0][dead].abs().max() if dead.any() else 0.0) == 0.0


@pytest.mark.parametrize("dk,H,B,Sq,Sk", [(32, 4, 2, 12, 200), (64, 2, 2, 29, 300), (128, 2, 1, 29, 333), (256, 2, 2, 29, 800), (256, 1, 1, 200, 200)])
def test_attention_backward_keys_with_a_common_component(ops, dk, H, B, Sq, Sk, monkeypatch):
    """near-uniform attention over keys that share a large common component (the decoder's cross-attention over the encoder
    memory): sum_j dS_ij = 0 exactly, but the bf16-rounded dS does not sum to zero and the residue multiplies the common
    component -- 10-25 % of |dQ| in the model (tests/study_attn_bwd_centering.py).  The dQ kernels subtract (row sum of the rounded
    dS) x mean key: what is left is the rounding of the keys themselves.  Every kernel family (d_k 32 / 64: 32-wide MFMA tiles,
    128 / 256: 16-wide), against the uncorrected kernels on the same inputs."""
    D = dk * H
    q = rnd(B, Sq, D, seed=60) * 0.2
    common = rnd(1, 1, D, seed=61) * 1.5
    k = common + rnd(B, Sk, D, seed=62) * 0.3
    v = rnd(B, Sk, D, seed=63)
    do = rnd(B, Sq, D, seed=64)
    mask = torch.ones(B, 1, Sk, dtype=torch.bool)
    mask[0, 0, Sk - Sk // 5:] = False
    md = mask.to(DEV)
    (qh, ql), (kh, kl), (vh, vl) = _planes(q), _planes(k), _planes(v)
    o, lse = ops.attn_fwd_bf16(qh, ql, kh, kl, vh, vl, md, H, precision=3)
    dq, dk_, dv = ops.attn_bwd_bf16(qh, kh, vh, o, do.to(DEV), lse, md, H)
    monkeypatch.setattr(ops, "ATTN_KMEAN", False)
    dq_plain, _, _ = ops.attn_bwd_bf16(qh, kh, vh, o, do.to(DEV), lse, md, H)
    qr, kr, vr = (t.clone().double().requires_grad_() for t in (q, k, v))
    want = _oracle_attention(qr, kr, vr, mask, H, rounded=False)
    (want * do.double()).sum().backward()
    from tests.gpu_util import report
    e_fix, e_plain = rel_err(dq, qr.grad), rel_err(dq_plain, qr.grad)
    assert e_fix < 1.5e-2 and e_fix < 0.6 * e_plain, f"dq dk={dk}: {e_fix:.3e} corrected vs {e_plain:.3e} plain\n" + report(dq, qr.grad, "dq")
    for name, got, ref, tol in (("dk", dk_, kr.grad, 3e-2), ("dv", dv, vr.grad, 2e-2)):
        e = rel_err(got, ref)
        assert e < tol, f"{name} dk={dk}: relative error {e:.3e}\n" + report(got, ref, name)


@pytest.mark.parametrize("dk,H,B,Sq,Sk,kind", ATTN_CASES + [(256, 2, 2, 200, 336, "pad"), (128, 4, 1, 70, 257, "pad"), (256, 4, 2, 128, 64, "pad")])
@pytest.mark.parametrize("prec,out_fmt", [(3, "x3"), (4, "f16"), (4, "x3")])
def test_attention_plane_outputs_match_fp32_outputs(ops, dk, H, B, Sq, Sk, kind, prec, out_fmt):
    """forward O planes and backward (plane, bias sums) forms against the fp32 outputs of the same kernels:
    hi == bf16(fp32) bit for bit, hi + lo == fp32 to 2^-16 / fp16 plane == fp16(fp32), bias sums == sums of the bf16 values;
    the backward's delta = rowsum(dO * O) from either form of the saved output"""
    D = dk * H
    q, k, v = rnd(B, Sq, D, seed=40), rnd(B, Sk, D, seed=41), rnd(B, Sk, D, seed=42)
    do = rnd(B, Sq, D, seed=43).to(DEV)
    mask = _masks(kind, B, Sq, Sk)
    if kind == "pad" and Sk > 100:
        mask[0, 0, Sk // 3:] = False
    md = None if mask is None else mask.to(DEV)

    def P(t, S):       # every plane of the projection output, as the projection GEMM's epilogue would leave them
        hi = t.to(torch.bfloat16)
        return ops.Planes(hi.to(DEV).view(B * S, D), (t - hi.float()).to(torch.bfloat16).to(DEV).view(B * S, D), B * S, D,
                          fh=t.to(torch.float16).to(DEV).view(B * S, D))
    qp, kp, vp = P(q, Sq), P(k, Sk), P(v, Sk)
    v3 = lambda t, S: None if t is None else t.view(B, S, D)
    raw = (lambda pl, S: (v3(pl.fh, S), None)) if prec == 4 else (lambda pl, S: (v3(pl.hi, S), v3(pl.lo, S)))
    (qa, qb), (ka, kb), (va, vb) = raw(qp, Sq), raw(kp, Sk), raw(vp, Sk)
    o32, lse32 = ops.attn_fwd_bf16(qa, qb, ka, kb, va, vb, md, H, precision=prec)
    o, lse = ops.attn_fwd_planes(qp, kp, vp, B, Sq, Sk, D, md, H, precision=prec, out_fmt=out_fmt)
    assert torch.equal(lse, lse32)
    o2 = o32.view(B * Sq, D)
    finite = torch.isfinite(o2)
    assert torch.equal(o.hi[:, :D][finite], o2.to(torch.bfloat16)[finite]), "O hi plane != bf16(O)"
    if out_fmt == "x3":
        rec = o.hi[:, :D].float() + o.lo[:, :D].float()
        assert_close(rec[finite], o2[finite], atol=1e-6, rtol=2e-5, name="O hi+lo")
        assert o.fh is None
    else:
        assert torch.equal(o.fh[:, :D][finite], o2.to(torch.float16)[finite]), "O fp16 plane != fp16(O)"
        assert o.lo is None
    assert (o.hi[:, D:] == 0).all()

    dq, dk_, dv = ops.attn_bwd_bf16(qp.hi.view(B, Sq, D), kp.hi.view(B, Sk, D), vp.hi.view(B, Sk, D), o32, do, lse32, md, H)
    bq = torch.nn.Parameter(torch.zeros(D, device=DEV))
    res = ops.attn_bwd_planes(qp, kp, vp, o, do, lse, B, Sq, Sk, D, md, H, 0.0, (bq, None, bq))
    for name, (Pl, db), ref, S in (("dq", res[0], dq, Sq), ("dk", res[1], dk_, Sk), ("dv", res[2], dv, Sk)):
        ref2 = ref.view(B * S, D)
        # the plane path rebuilds O from its planes (2^-16 / 2^-12 relative) inside delta: compare to bf16 resolution
        got = Pl.hi[:, :D].float()
        assert_close(got, ref2, atol=2e-3 * float(ref2.abs().max()), rtol=1e-2, name=f"{name} plane")
        if name != "dk":
            assert db is not None
            assert_close(db, got.double().sum(0), atol=1e-4 * float(got.abs().sum(0).max()) + 1e-6, rtol=1e-4, name=f"{name} bias sums")
        else:
            assert db is None


@pytest.mark.parametrize("pad", [False, True])
@pytest.mark.parametrize("prec,out_fmt", [(X3, "x3"), (X3, "f16"), (F16W2, "f16"), (F16W2, "x3")])
def test_gemm_plane_outputs(ops, pad, prec, out_fmt):
    """the epilogue's plane outputs: bf16 hi + (bf16 lo | fp16) of the fp32 result, whatever the product's own precision"""
    M, N, K = 150, 96, 64
    x, W, b = rnd(M, K, seed=7), rnd(N, K, seed=8), rnd(N, seed=9)
    pl = ops.linear_fwd_planes(x.to(DEV), W.to(DEV), b.to(DEV), precision=prec, out_fmt=out_fmt, pad=pad)
    y = ops.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), precision=prec)
    assert pl.hi.dtype == torch.bfloat16 and pl.hi.shape == (M, 128 if pad else N)
    assert torch.equal(pl.hi[:, :N], y.to(torch.bfloat16))
    if out_fmt == "x3":
        assert_close(pl.hi[:, :N].float().double() + pl.lo[:, :N].float().double(), y, atol=1e-6, rtol=2e-5, name="hi+lo planes")
        second = pl.lo
    else:
        assert pl.fh.dtype == torch.float16 and torch.equal(pl.fh[:, :N], y.to(torch.float16))
        second = pl.fh
    if pad:
        assert float(pl.hi[:, N:].float().abs().max()) == 0 and float(second[:, N:].float().abs().max()) == 0


def test_attention_fully_masked_row_is_nan(ops):
    q, k, v = rnd(1, 4, 64, seed=1), rnd(1, 6, 64, seed=2), rnd(1, 6, 64, seed=3)
    mask = torch.ones(1, 4, 6, dtype=torch.bool)
    mask[0, 2, :] = False
    o, _ = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), 2)
    assert torch.isnan(o[0, 2]).all() and not torch.isnan(o[0, [0, 1, 3]]).any()


@pytest.mark.parametrize("dk,H,B,Sq,Sk,kind", ATTN_CASES)
def test_attention_backward(ops, dk, H, B, Sq, Sk, kind):
    D = dk * H
    q, k, v = rnd(B, Sq, D, seed=30), rnd(B, Sk, D, seed=31), rnd(B, Sk, D, seed=32)
    do = rnd(B, Sq, D, seed=33)
    mask = _masks(kind, B, Sq, Sk)
    md = None if mask is None else mask.to(DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.attn_fwd(qd, kd, vd, md, H, precision=3)
    dq, dk_, dv = ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, md, H)
    qr, kr, vr = (t.clone().double().requires_grad_() for t in (q, k, v))
    want = _oracle_attention(qr, kr, vr, mask, H, rounded=False)
    (want * do.double()).sum().backward()
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk_, kr.grad), ("dv", dv, vr.grad)):
        e = rel_err(got, ref)
        assert e < 2e-2, f"{name} dk={dk} {kind}: relative error {e:.3e}\n" + __import__("tests.gpu_util", fromlist=["report"]).report(got, ref, name)


def test_attention_dropout_roundtrip(ops):
    """dropout on the attention OUTPUT: forward mask == the mask bmt_gemm re-applies in backward; delta uses (1-p)."""
    B, H, Sq, Sk, dk, p, site = 2, 2, 40, 50, 64, 0.25, 77
    D = H * dk
    ops.manual_seed(5)
    q, k, v = rnd(B, Sq, D, seed=40).to(DEV), rnd(B, Sk, D, seed=41).to(DEV), rnd(B, Sk, D, seed=42).to(DEV)
    o0, _ = ops.attn_fwd(q, k, v, None, H, precision=3)
    od, lse = ops.attn_fwd(q, k, v, None, H, drop_p=p, site=site, precision=3)
    assert torch.equal(od, ops.dropout_raw(o0, p, site))
    # backward with dropout == backward without dropout fed the masked upstream gradient
    g = rnd(B, Sq, D, seed=43).to(DEV)
    gm = ops.dropout_raw(g, p, site)
    a = ops.attn_bwd(q, k, v, od, gm, lse, None, H, drop_p=p)
    b = ops.attn_bwd(q, k, v, o0, gm, lse, None, H, drop_p=0.0)
    for x, y, n in zip(a, b, ("dq", "dk", "dv")):
        assert rel_err(x, y) < 1e-5, n


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,D", [(37, 128), (9, 300), (5, 600), (130, 1024), (3, 30), (50, 20)])
def test_layernorm(ops, rows, D):
    from oracle import bmt_oracle as orc
    x = rnd(rows, D, seed=50) * 2 + 0.3
    g, b = 1 + 0.1 * rnd(D, seed=51), 0.1 * rnd(D, seed=52)
    xd = x.to(DEV).requires_grad_()
    gd, bd = g.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.LayerNormFn.apply(xd, gd, bd, 1e-5)
    xr, gr, br = (t.clone().double().requires_grad_() for t in (x, g, b))
    want = orc.layer_norm(xr, gr, br)
    assert_close(y, want, atol=2e-5, name="ln fwd")
    w = rnd(rows, D, seed=53)
    (y * w.to(DEV)).sum().backward()
    (want * w.double()).sum().backward()
    assert_close(xd.grad, xr.grad, atol=5e-5, rtol=1e-4, name="ln dx")
    assert_close(gd.grad, gr.grad, atol=2e-4, rtol=1e-4, name="ln dgamma")
    assert_close(bd.grad, br.grad, atol=2e-4, rtol=1e-4, name="ln dbeta")


# ------------------------------------------------------------------------------------------ prep / masks / loss / adam
def test_masks_bit_exact(ops, golden):
    from bmt_amd.model.masking import mask, subsequent_mask
    g = golden("masks_loss.npz")
    rgb, audio, caps = g["mk/rgb"].to(DEV), g["mk/audio"].to(DEV), g["mk/caps"].to(DEV)
    V_mask, C_mask = mask(rgb[:, :, 0], caps[:, :-1], 1)
    A_mask = mask(audio[:, :, 0], None, 1)
    for got, key in ((V_mask, "mk/V_mask"), (A_mask, "mk/A_mask"), (C_mask, "mk/C_mask")):
        assert got.dtype == torch.bool and torch.equal(got.cpu(), g[key]), key
    assert torch.equal(subsequent_mask(5), g["mk/subsequent5"])


def test_prep_and_embed(ops, golden):
    from bmt_amd.model.blocks import PositionalEncoder, VocabularyEmbedder
    g = golden("modules_tiny.npz")
    for d in (20, 128, 300, 1024):
        pe = PositionalEncoder(d, 0.0).eval()
        assert np.array_equal(pe.pos_enc_mat[0, :9].numpy(), g.np(f"pe/{d}/head"))
        y = pe(g[f"pe/{d}/x"].to(DEV))
        assert_close(y, g[f"pe/{d}/y"], atol=1e-6, name=f"pe {d}")
    pe = PositionalEncoder(128, 0.0).eval()
    a, b = rnd(2, 5, 128, seed=60), rnd(2, 5, 128, seed=61)
    from oracle import bmt_oracle as orc
    assert_close(pe(a.to(DEV), fuse_add=b.to(DEV)), orc.positional_encoder(a + b), atol=1e-6, name="fused add")
    ve = VocabularyEmbedder(11, 20)
    with torch.no_grad():
        ve.embedder.weight.copy_(g["vemb/weight"])
    ve = ve.to(DEV)
    assert_close(ve(g["vemb/ids"].to(DEV)), g["vemb/out"], atol=1e-6, name="vocab embed")


@pytest.mark.parametrize("tag,s", [("a", 0.7), ("a", 0.0), ("idx0", 0.7), ("nopad", 0.7), ("big", 0.7)])
def test_label_smoothing_golden(ops, golden, tag, s):
    from bmt_amd.loss.label_smoothing import LabelSmoothing
    g = golden("masks_loss.npz")
    key = f"ls/{tag}/s{s}"
    pred = g[key + "/pred"].to(DEV).requires_grad_()
    loss = LabelSmoothing(s, 1)(pred, g[key + "/target"].to(DEV))
    assert_close(loss, g[key + "/loss"], atol=2e-4, rtol=2e-5, name="ls loss")
    (loss * 0.5).backward()
    assert_close(pred.grad, g[key + "/dpred"] * 0.5, atol=1e-7, rtol=1e-6, name="ls dpred")


def test_generator_log_softmax(ops, golden):
    from bmt_amd.model.generators import Generator
    g = golden("modules_tiny.npz")
    gen = Generator(20, 11)
    gen.load_state_dict(g.sub("gen/sd/"))
    gen = gen.to(DEV)
    x = g["gen/X"].to(DEV).requires_grad_()
    out = gen(x)
    assert_close(out, g["gen/out"], atol=1e-5, name="generator")
    # backward vs the oracle
    from oracle import bmt_oracle as orc
    p = {k: v.clone().double().requires_grad_() for k, v in g.sub("gen/sd/").items()}
    xr = g["gen/X"].clone().double().requires_grad_()
    w = rnd(*out.shape, seed=70)
    (orc.generator(p, "", xr) * w.double()).sum().backward()
    (out * w.to(DEV)).sum().backward()
    assert rel_err(x.grad, xr.grad) < 2e-2
    assert rel_err(gen.linear.weight.grad, p["linear.weight"].grad) < 2e-2
    assert rel_err(gen.linear.bias.grad, p["linear.bias"].grad) < 1e-4


def test_adam_golden(golden):
    from bmt_amd.optim import FusedAdam
    g = golden("adam.npz")
    p = torch.nn.Parameter(g["p0"].to(DEV))
    opt = FusedAdam([p], lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    for step in range(1, 4):
        p.grad = g[f"g{step}"].to(DEV)
        opt.step()
        assert_close(p.data, g[f"p{step}"], atol=1e-7, rtol=1e-6, name=f"adam step {step}")


def test_clip_grad_norm(ops):
    from bmt_amd.optim import clip_grad_norm_
    ps = [torch.nn.Parameter(rnd(300, seed=80).to(DEV)), torch.nn.Parameter(rnd(40, 50, seed=81).to(DEV))]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    for p, r, s in zip(ps, ref, (82, 83)):
        gsrc = rnd(*p.shape, seed=s)
        p.grad, r.grad = gsrc.to(DEV), gsrc.clone()
    n_ref = torch.nn.utils.clip_grad_norm_(ref, 0.5)
    n = clip_grad_norm_(ps, 0.5)
    assert abs(float(n) - float(n_ref)) < 1e-3 * float(n_ref)
    for p, r in zip(ps, ref):
        assert_close(p.grad, r.grad, atol=1e-6, rtol=1e-5, name="clipped grad")


# ---------------------------------------------------------------- fused residual block pieces
@pytest.mark.parametrize("rows,D", [(37, 128), (50, 300), (9, 1024), (5, 77)])
def test_layernorm_forward_writes_its_operand_planes(ops, rows, D):
    """bmt_layernorm_fwd_planes: fp32 output == bmt_layernorm_fwd, planes == bmt_planes of that output (zero padded)"""
    from bmt_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.3).to(DEV)
    gamma, beta = torch.randn(D, generator=g).to(DEV), torch.randn(D, generator=g).to(DEV)
    y0 = torch.empty_like(x)
    m0, r0 = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    _lib.check(lib.bmt_layernorm_fwd(ops._p(x), D, ops._p(gamma), ops._p(beta), ops._p(y0), D, ops._p(m0), ops._p(r0), rows, D, 1e-5,
                                     ops._st()), "ln")
    ld = ops._pad64(D)
    want = ops.make_planes(y0, "all")
    for with_y, f16 in ((True, False), (False, False), (True, True)):
        y = torch.full_like(x, float("nan"))
        hi = torch.full((rows, ld), 7.0, device=DEV, dtype=torch.bfloat16)
        lo = torch.full((rows, ld), 7.0, device=DEV, dtype=torch.float16 if f16 else torch.bfloat16)
        m, r = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        _lib.check(lib.bmt_layernorm_fwd_planes(ops._p(x), D, ops._p(gamma), ops._p(beta), ops._p(y) if with_y else None, D, ops._p(m),
                                                ops._p(r), ops._p(hi), ops._p(lo), int(f16), ld, rows, D, 1e-5, None, ops._st()), "ln planes")
        if with_y:
            assert torch.equal(y, y0)
        assert torch.equal(m, m0) and torch.equal(r, r0)
        assert torch.equal(hi, want.hi) and torch.equal(lo, want.fh if f16 else want.lo)


@pytest.mark.parametrize("rows,D", [(37, 128), (50, 300), (4100, 1024), (5, 77)])
def test_layernorm_backward_adds_the_residual_gradient(ops, rows, D):
    from bmt_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows)
    x, dy, add = [torch.randn(rows, D, generator=g).to(DEV) for _ in range(3)]
    gamma = torch.randn(D, generator=g).to(DEV)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    outs = []
    for a in (None, add):
        dx = torch.empty_like(x)
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        ws = torch.empty(max(1, lib.bmt_layernorm_bwd_blocks(rows)) * 2 * D, device=DEV)
        _lib.check(lib.bmt_layernorm_bwd_add(ops._p(dy), D, ops._p(x), D, ops._p(gamma), ops._p(mean), ops._p(rstd), ops._p(dx), D,
                                             ops._p(a), D, ops._p(dg), ops._p(db), ops._p(ws), rows, D, None, ops._st()), "ln bwd add")
        outs.append((dx, dg, db))
    assert torch.equal(outs[1][0], outs[0][0] + add)              # one fp32 add, same rounding as the separate kernel
    assert_close(outs[1][1], outs[0][1], atol=1e-4, rtol=1e-5, name="dgamma")
    xr = x.double().requires_grad_()
    torch.nn.functional.layer_norm(xr, (D,), gamma.double(), torch.zeros(D, device=DEV, dtype=torch.double), 1e-5).backward(dy.double())
    assert_close(outs[0][0], xr.grad, atol=2e-4, rtol=1e-4, name="dx")


@pytest.mark.parametrize("R,C", [(100, 128), (33, 300), (257, 1024)])
def test_planes_of_a_dropped_gradient(ops, R, C):
    """bmt_planes_dropout == bmt_planes(bmt_dropout(x)) bit for bit, column sums included"""
    ops.manual_seed(3)
    x = torch.randn(R, C, device=DEV)
    p, site = 0.3, 41
    dropped = ops.dropout_raw(x, p, site)
    assert 0.2 < float((dropped == 0).float().mean()) < 0.4
    cs0, cs1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    want = ops.make_planes(dropped, "all", colsum=cs0)
    got = ops.make_planes(x, "all", colsum=cs1, drop=(p, site))
    for n in ("hi", "lo", "fh", "fl"):
        assert torch.equal(getattr(got, n), getattr(want, n)), n
    assert_close(cs1, cs0, atol=1e-4, rtol=1e-5, name="colsum")


@pytest.mark.parametrize("M,N,K", [(8192, 1024, 1024), (8192, 1024, 3072), (8200, 4096, 1024), (25600, 1024, 320)])
def test_dx_of_the_encoder_sized_products(ops, M, N, K):
    """dX = dY . W at the encoder's sizes (M rows, N outputs, reduction K): fp32 output against fp64 on the bf16-rounded operands, and the
    plane-only output with a gate and column sums against the fp32 one -- the shapes that fill the chip with 256-row tiles (the pipelined
    dX tile of BMT_DX_PIPE) as well as the 128-row tiles'"""
    dy, W = rnd(M, K, seed=21).to(DEV), (rnd(K, N, seed=22) * 0.05).to(DEV)
    dyP = ops.make_planes(dy, "bwd")
    dx = ops.linear_dx(dyP, W)
    want = bf16_round(dy.cpu()).double() @ bf16_round(W.cpu()).double()
    assert_close(dx, want, atol=2e-4 * math.sqrt(K), rtol=1e-4, name="dx")
    hid = torch.relu(rnd(M, N, seed=23)).to(DEV)
    hid[:, ::7] = 0
    h = ops.make_planes(hid, "bwd")
    gated = ops.linear_dx(dyP, W, gate=h, gate_scale=1.25)
    op = ops.Planes(torch.full((M, ops._pad64(N)), 3.0, device=DEV, dtype=torch.bfloat16), None, M, N)
    cs = torch.zeros(N, device=DEV)
    ops.linear_dx(dyP, W, out_planes=op, gate=h, gate_scale=1.25, colsum=cs)
    got = op.hi[:, :N].float()
    assert torch.equal(got == 0, gated == 0)
    assert_close(got, gated, atol=1e-6, rtol=2 ** -7, name="gated plane output")
    ref = gated.double().sum(0)
    assert_close(cs, ref, atol=2e-3 * float(ref.abs().max()) + 1e-3, name="column sums")


@pytest.mark.parametrize("M,N,K", [(300, 4096, 1024), (130, 200, 96), (128, 1024, 64)])
def test_gemm_plane_output_with_vector_gate_and_column_sums(ops, M, N, K):
    """dX GEMM of the FFN backward: plane-only output, relu/dropout gate applied per 16-byte segment from the saved hidden plane,
    column sums (the next bias gradient) from the same epilogue -- against the fp32-output GEMM with the per-element gate."""
    dy, W = rnd(M, K, seed=1).to(DEV), (rnd(K, N, seed=2) * 0.1).to(DEV)
    hid = torch.relu(rnd(M, N, seed=3)).to(DEV)
    hid[:, ::5] = 0
    h = ops.make_planes(hid, "bwd")
    dyP = ops.make_planes(dy, "bwd")
    want = ops.linear_dx(dyP, W, gate=h, gate_scale=1.25)                      # fp32 out, per-element gate
    assert bool((want[:, ::5] == 0).all()) and float(want.abs().max()) > 0
    op = ops.Planes(torch.full((M, ops._pad64(N)), 3.0, device=DEV, dtype=torch.bfloat16), None, M, N)
    cs = torch.zeros(N, device=DEV)
    ops.linear_dx(dyP, W, out_planes=op, gate=h, gate_scale=1.25, colsum=cs)
    got = op.hi[:, :N].float()
    assert torch.equal(got == 0, want == 0)                                    # the mask itself: exact
    # values: the fp32-output launch may be split-K (other summation order), so a bf16 rounding can flip: one bf16 ulp
    assert_close(got, want, atol=1e-6, rtol=2 ** -7, name="gated plane output")
    assert bool((op.hi[:, N:] == 0).all())
    ref = want.double().sum(0)
    assert_close(cs, ref, atol=2e-3 * float(ref.abs().max()) + 1e-3, name="column sums (hi + lo of the staged values)")
    cs2 = torch.zeros(N, device=DEV)                                           # no gate: sums of the plain product
    ops.linear_dx(dyP, W, out_planes=op, colsum=cs2)
    full = ops.linear_dx(dyP, W)
    assert_close(cs2, full.double().sum(0), atol=2e-3 * float(full.abs().sum(0).max()) + 1e-3, name="column sums, no gate")


def test_grouped_weight_gradient_launch(ops):
    """bmt_gemm_bf16_grouped: the weight gradients of several layers (different shapes, ragged reduction lengths and widths) in one
    launch, accumulated into live buffers -- against one split-K launch per problem."""
    # (from (8192, 4096, 1024) on: large outputs with ragged reductions -- 5000 rows: the last stage is part zeros --, one and two stages, an odd
    # number of them, extents that are not multiples of the tile; the set the 256 x 256-tile experiment of round 4 was checked on)
    shapes = [(8192, 1024, 1024), (960, 300, 1024), (1000, 128, 512), (257, 130, 70), (64, 10000, 300), (25600, 1024, 128), (5000, 3072, 128),
              (8192, 4096, 1024), (5000, 3072, 1024), (700, 512, 768), (100, 256, 256), (64, 256, 512), (3000, 1000, 2040), (12800, 2048, 1024)]
    items, want = [], []
    for i, (rows, n_out, k_in) in enumerate(shapes):
        dy = ops.make_planes((rnd(rows, n_out, seed=10 + i) * 0.1).to(DEV), "bwd")
        x = ops.make_planes(rnd(rows, k_in, seed=40 + i).to(DEV), "bwd")
        acc = rnd(n_out, k_in, seed=70 + i).to(DEV)
        ref = acc.clone()
        ops.linear_dw(dy, x, into=ref)                     # one launch (split-K workspace + epilogue) per problem
        items.append((dy, x, acc))
        want.append(ref)
    ops.gemm_bf16_grouped(items)
    for (rows, n_out, k_in), (_, _, acc), ref in zip(shapes, items, want):
        assert_close(acc, ref, atol=2e-4 * math.sqrt(rows), rtol=1e-5, name=f"grouped dW {n_out}x{k_in} over {rows} rows")
    # queueing through linear_dw: nothing runs until flush_dw
    ops.context().defer_dw = True
    try:
        accs = [torch.zeros_like(a) for _, _, a in items]
        for (dy, x, _), a in zip(items, accs):
            assert ops.linear_dw(dy, x, into=a) is None
        assert all(float(a.abs().max()) == 0.0 for a in accs)
        ops.flush_dw()
    finally:
        ops.context().defer_dw = False
    for a, (_, _, acc0), ref, (rows, n_out, k_in) in zip(accs, items, want, shapes):
        base = rnd(n_out, k_in, seed=70 + shapes.index((rows, n_out, k_in))).to(DEV)
        assert_close(a, ref - base, atol=3e-4 * math.sqrt(rows), rtol=1e-5, name="deferred dW")


def test_colsum_multi_and_copy_multi(ops):
    """many small reductions / copies in one launch (bmt_colsum_multi, bmt_copy_multi: the item list travels in the kernel arguments):
    more items than one launch carries, ragged shapes, accumulate semantics"""
    import ctypes as C
    from bmt_amd import _lib
    g = torch.Generator().manual_seed(3)
    items, want = [], []
    for i in range(130):                       # > BMT_COLSUM_MAX_ITEMS: two launches
        rows, D = 1 + (i * 7) % 40, 8 * (1 + i % 9)
        ld = D + 8 * (i % 3)
        part = torch.randn(rows, ld, generator=g).to(DEV)
        out = torch.randn(D, generator=g).to(DEV)
        want.append(out.double() + part[:, :D].double().sum(0))
        items.append((part, 0, out, rows, ld, D))
    ops._colsum_launch(items)
    for (part, _, out, rows, ld, D), w in zip(items, want):
        assert_close(out, w, atol=1e-4, rtol=1e-5, name=f"colsum rows {rows} D {D}")
    srcs = [torch.randn(3 + 5 * i, generator=g).to(DEV) for i in range(20)]
    dst = torch.zeros(sum(s.numel() for s in srcs), device=DEV)
    arr, off = (_lib.CopyItem * len(srcs))(), 0
    for a, s in zip(arr, srcs):
        a.src, a.dst, a.n = s.data_ptr(), dst.data_ptr() + 4 * off, s.numel()
        off += s.numel()
    _lib.check(ops.lib.bmt_copy_multi(arr, len(srcs), ops._st()), "bmt_copy_multi")
    assert torch.equal(dst, torch.cat(srcs))
