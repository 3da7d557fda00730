"""-m gpu: the kernels and host paths added in round 4 -- K7 as one forward and one backward kernel, the step-protocol launches
(caption shift + token count, loss finish, arena zero), the live-query shortcut of the split attention backward, the LayerNorm backward
with a second addend behind the bi-modal layers' three-consumer fan-out -- each against the oracle / fp64 autograd / the unfused path."""
import math

import pytest
import torch

from tests.gpu_util import assert_close, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from bmt_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ---------------------------------------------------------------------------------------------- K7
@pytest.mark.parametrize("rows,V", [(7, 10), (50, 1000), (33, 10000), (9, 10172), (5, 17), (3, 16384), (2, 20001)])
def test_log_softmax_with_row_sums(ops, rows, V):
    """bmt_log_softmax_fwd_stats: the register-resident rows (V % 4 == 0, V <= 16384) and the fallback, both == torch.log_softmax and the
    row sums of what they wrote"""
    from bmt_amd import _lib
    x = (rnd(rows, V, seed=V) * 3).to(DEV)
    want = torch.log_softmax(x.double(), dim=-1)
    rs = torch.empty(rows, device=DEV)
    _lib.check(ops.lib.bmt_log_softmax_fwd_stats(ops._p(x), x.stride(0), rows, V, ops._p(rs), ops._st()), "ls")
    assert_close(x, want, atol=3e-6, rtol=1e-6, name="log_softmax")
    assert_close(rs, x.double().sum(-1), atol=0, rtol=2e-6, name="row sums")


def _gen_case(V, B=3, T=11, D=40, pad_rows="some", seed=5):
    from bmt_amd.loss.label_smoothing import LabelSmoothing
    from bmt_amd.model.generators import Generator
    import contextlib, io
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        gen = Generator(D, V).to(DEV)
    x = rnd(B, T, D, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    y = torch.randint(4, V, (B, T), generator=g)
    if pad_rows == "some":
        y[0, T - 3:] = 1
        y[2, 5:] = 1
    elif pad_rows == "idx0":          # the reference's quirk: a lone pad target at flat index 0 is NOT zeroed (label_smoothing.py:26-30)
        y[0, 0] = 1
    return gen, LabelSmoothing(0.7, 1), x, y


@pytest.mark.parametrize("V,pads", [(10, "some"), (1000, "some"), (10000, "some"), (10172, "idx0"), (37, "none"), (16, "idx0")])
def test_generator_and_loss_as_one_node(ops, V, pads, monkeypatch):
    """LabelSmoothing applied to a Generator's output takes the fused autograd node (ops.FusedGenLossFn: the loss from the row sums, the
    backward straight from the saved log-probabilities to the bf16 plane of d loss / d logits).  Loss and gradients against the oracle in
    fp64 and against the two separate nodes (ops.FUSE_GEN_LOSS off) on the same weights."""
    from oracle import bmt_oracle as orc
    gen, crit, x, y = _gen_case(V, pad_rows=pads)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSE_GEN_LOSS", fused)
        for p in gen.parameters():
            p.grad = None
        xd = x.to(DEV).requires_grad_()
        pred = gen(xd)
        loss = crit(pred, y.to(DEV))
        assert (type(loss.grad_fn).__name__ == "FusedGenLossFnBackward") == fused, type(loss.grad_fn).__name__
        (loss * 0.25).backward()
        torch.cuda.synchronize()
        res[fused] = (float(loss), xd.grad.cpu(), gen.linear.weight.grad.cpu().clone(), gen.linear.bias.grad.cpu().clone(), pred.detach().cpu())
    p = {"linear.weight": gen.linear.weight.detach().cpu().double().requires_grad_(), "linear.bias": gen.linear.bias.detach().cpu().double().requires_grad_()}
    xr = x.double().requires_grad_()
    opred = orc.generator(p, "", xr)
    oloss = orc.label_smoothing_kl(opred, y, 0.7, 1)
    (oloss * 0.25).backward()
    for fused in (True, False):
        l, dx, dW, db, pred = res[fused]
        assert_close(pred, opred.detach(), atol=2e-5, name="log-probs")
        assert abs(l - float(oloss)) < 2e-4 * max(1.0, abs(float(oloss))), (fused, l, float(oloss))
        assert rel_err(dx, xr.grad) < 2e-2, report(dx, xr.grad, f"dx fused={fused}")
        assert rel_err(dW, p["linear.weight"].grad) < 2e-2, report(dW, p["linear.weight"].grad, f"dW fused={fused}")
        assert rel_err(db, p["linear.bias"].grad) < 4e-3, report(db, p["linear.bias"].grad, f"db fused={fused}")
    # the two forms see the same bf16-rounded d logits up to the rounding order: gradients agree far inside the bf16 bar
    for a, b, n in zip(res[True][1:4], res[False][1:4], ("dx", "dW", "db")):
        assert rel_err(a, b) < 6e-3, report(a, b, n + " fused vs unfused")


def test_fused_loss_leaves_other_consumers_of_the_log_probs_alone(ops):
    """a second consumer of the Generator's output (here: a plain weighted sum) still gets its gradient through the tensor's own node; the
    two contributions add up in the generator's parameters"""
    gen, crit, x, y = _gen_case(50)
    w = rnd(3, 11, 50, seed=9).to(DEV)
    xd = x.to(DEV).requires_grad_()
    pred = gen(xd)
    (crit(pred, y.to(DEV)) + (pred * w).sum()).backward()
    both = (xd.grad.clone(), gen.linear.weight.grad.clone())
    parts = []
    for which in (0, 1):
        for p in gen.parameters():
            p.grad = None
        xe = x.to(DEV).requires_grad_()
        pr = gen(xe)
        (crit(pr, y.to(DEV)) if which == 0 else (pr * w).sum()).backward()
        parts.append((xe.grad.clone(), gen.linear.weight.grad.clone()))
    assert rel_err(both[0], parts[0][0] + parts[1][0]) < 5e-3 and rel_err(both[1], parts[0][1] + parts[1][1]) < 5e-3
    # a tensor that was written after the generator produced it does not take the fused node
    pr = gen(x.to(DEV).requires_grad_())
    pr2 = pr * 1.0
    assert type(crit(pr2, y.to(DEV)).grad_fn).__name__ != "FusedGenLossFnBackward"


# ---------------------------------------------------------------------------------------------- step protocol
@pytest.mark.parametrize("B,T1", [(2, 8), (32, 31), (5, 2)])
def test_caption_shift_and_token_count(ops, B, T1):
    g = torch.Generator().manual_seed(B)
    caps = torch.randint(0, 9, (B, T1), generator=g)
    big = torch.zeros(B, T1 + 3, dtype=torch.int64)
    big[:, :T1] = caps
    for src in (caps.to(DEV), big.to(DEV)[:, :T1]):          # contiguous and row-strided inputs
        x, y, n = ops.caption_shift(src, 1)
        assert x.is_contiguous() and y.is_contiguous() and n.dim() == 0 and n.dtype == torch.int64
        assert torch.equal(x.cpu(), caps[:, :-1]) and torch.equal(y.cpu(), caps[:, 1:]) and int(n) == int((caps[:, 1:] != 1).sum())


def test_loss_finish_and_zero(ops):
    kl = torch.tensor(123.456, device=DEV)
    n = torch.tensor(607, device=DEV)
    gs = torch.zeros(1, device=DEV)
    loss = ops.loss_finish(kl, n, gs)
    assert loss.dim() == 0 and abs(float(loss) - 123.456 / 607) < 1e-7 and abs(float(gs) - 1.0 / 607) < 1e-10
    for numel in (1, 3, 4, 1000, 1 << 20, (1 << 20) + 3):
        t = torch.full((numel + 8,), 5.0, device=DEV)
        ops.zero_(t[4:4 + numel] if numel % 4 == 0 else t[:numel])
        torch.cuda.synchronize()
        view = t[4:4 + numel] if numel % 4 == 0 else t[:numel]
        assert float(view.abs().sum()) == 0.0 and float(t.sum()) == 5.0 * 8


def test_gradient_arena_is_zeroed_in_one_launch(ops):
    from bmt_amd.parallel import GradientReducer
    ps = [torch.nn.Parameter(rnd(n, 7, seed=n).to(DEV)) for n in (100, 3000, 41, 999)]
    red = GradientReducer(ps, bucket_bytes=16 << 10)
    assert len(red.buckets) >= 2
    try:
        a0 = red.buckets[0]["flat"].untyped_storage().data_ptr()
        assert all(b["flat"].untyped_storage().data_ptr() == a0 for b in red.buckets)          # one allocation
        sum((p * (i + 1)).sum() for i, p in enumerate(ps)).backward()
        assert all(float(p.grad.abs().sum()) > 0 for p in ps)
        red.zero_grad()
        torch.cuda.synchronize()
        assert all(float(p.grad.abs().sum()) == 0.0 for p in ps) and float(red._arena.abs().sum()) == 0.0
    finally:
        red.remove()


# ---------------------------------------------------------------------------------------------- LayerNorm backward, two addends
@pytest.mark.parametrize("rows,D", [(37, 128), (4100, 1024), (50, 300), (5, 77)])
def test_residual_norm_with_a_third_consumer(ops, rows, D):
    """ops.residual_norm(..., kv_alias=True): x -> (x, LN(x), x); the three gradients meet in the LayerNorm backward kernel
    (bmt_layernorm_bwd_partial2; shapes the vector kernel does not take go through one extra add).  Against fp64 autograd."""
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.3)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    w1, w2, w3 = (torch.randn(rows, D, generator=g) for _ in range(3))
    xd, gd, bd = (t.to(DEV).requires_grad_() for t in (x, gamma, beta))
    xid, xn, xkv = ops.residual_norm(xd, gd, bd, 1e-5, ops.PREC_BF16X3, kv_alias=True)
    assert xkv.data_ptr() == xd.data_ptr() and xid.data_ptr() == xd.data_ptr()
    ((xid * w1.to(DEV)).sum() + (xn * w2.to(DEV)).sum() + (xkv * w3.to(DEV)).sum()).backward()
    xr, gr, br = (t.double().requires_grad_() for t in (x, gamma, beta))
    ln = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    ((xr * w1.double()).sum() + (ln * w2.double()).sum() + (xr * w3.double()).sum()).backward()
    assert_close(xn, ln.detach(), atol=5e-6, rtol=1e-5, name="LN")
    assert rel_err(xd.grad, xr.grad) < 1e-5, report(xd.grad, xr.grad, "dx")
    assert rel_err(gd.grad, gr.grad) < 1e-4 and rel_err(bd.grad, br.grad) < 1e-4
    # any subset of the three gradients
    for use in ((1, 0, 1), (0, 1, 1), (0, 0, 1)):
        xd2 = x.to(DEV).requires_grad_()
        outs = ops.residual_norm(xd2, gd.detach(), bd.detach(), 1e-5, ops.PREC_BF16X3, kv_alias=True)
        sum((o * w.to(DEV)).sum() for o, w, u in zip(outs, (w1, w2, w3), use) if u).backward()
        xr2 = x.double().requires_grad_()
        outs_r = (xr2, torch.nn.functional.layer_norm(xr2, (D,), gamma.double(), beta.double(), 1e-5), xr2)
        sum((o * w.double()).sum() for o, w, u in zip(outs_r, (w1, w2, w3), use) if u).backward()
        assert rel_err(xd2.grad, xr2.grad) < 1e-5, use


# ---------------------------------------------------------------------------------------------- split attention backward: live queries
def _attention_fp64(q, k, v, mask, H):
    B, Sq, D = q.shape
    dk = D // H
    qh, kh, vh = (t.view(B, -1, H, dk).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dk)
    s = s.masked_fill(~mask.view(B, 1, 1, -1), float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Sq, D)


@pytest.mark.parametrize("form", ["recompute", "emit"])
@pytest.mark.parametrize("B,H,Sq,Sk,dk,zero", [(2, 4, 800, 800, 256, "suffix"), (2, 4, 256, 800, 256, "suffix"), (2, 2, 300, 200, 128, "hole"),
                                                (2, 4, 800, 256, 256, "all"), (3, 2, 130, 45, 256, "head")])
def test_split_backward_skips_queries_without_gradient(ops, B, H, Sq, Sk, dk, zero, form, monkeypatch):
    """the dQ kernel notes which 32-query groups have a non-zero dO (padded positions of the encoder: a suffix of every sequence); a 128-query
    tile without one skips its key loop and leaves no P / dS, the dK / dV kernel ends its query loop at the last live stage and wipes the
    fragments of a dead stage before it.  suffix: the real pattern (different lengths per batch element); hole: dead tiles and dead 32-row
    groups BEFORE live ones; all: no gradient at all; head: one head's columns of dO zero, the others not (the bits are per head).  Against
    fp64 autograd on the kernels' operands.  Both split forms: "emit" (P and dS through HBM workspaces) and "recompute" (round 6: the key side
    rebuilds them; a dead stage in front of a live one is simply computed there -- dO = 0 makes its dP, delta and dS exactly zero)."""
    monkeypatch.setattr(ops, "ATTN_BWD_RECOMPUTE", form == "recompute")
    D = H * dk
    q = rnd(B * Sq, D, seed=401) * 0.7
    k = rnd(B * Sk, D, seed=402) * 0.7 + 0.4
    v = rnd(B * Sk, D, seed=403)
    do = rnd(B, Sq, D, seed=404)
    g = torch.Generator().manual_seed(405)
    if zero == "suffix":
        for b in range(B):
            do[b, int(torch.randint(Sq // 2, Sq, (1,), generator=g)):] = 0
    elif zero == "hole":
        do[0, 0:128] = 0            # a dead tile in front of live ones
        do[0, 160:192] = 0          # a dead 32-row group inside a live tile
        do[1, 64:290] = 0
    elif zero == "all":
        do.zero_()
    elif zero == "head":
        do[:, :, :dk] = 0
        do[1, 40:, dk:] = 0
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = (torch.arange(Sk)[None, :] < lens[:, None]).view(B, 1, Sk)
    md = mask.to(DEV)
    qp, kp, vp = (ops.make_planes(t.to(DEV), "f16") for t in (q, k, v))
    f16 = lambda pl: ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh)
    o, lse = ops.attn_fwd_planes(qp, kp, vp, B, Sq, Sk, D, md, H, precision=ops.PREC_F16, out_fmt="f16")
    dop = ops.make_planes(do.view(B * Sq, D).to(DEV), "bwd")
    dop = ops.Planes(dop.hi[:, :D].contiguous(), None, B * Sq, D)
    biases = tuple(torch.zeros(D, device=DEV, requires_grad=True) for _ in range(3))
    # poison what the skipped tiles would have written, so that a stage that reads it without being told to shows
    for name in ("attn_P", "attn_dS"):
        ops.stream_scratch(name, 1 << 26, torch.bfloat16, DEV).view(torch.int16).fill_(0x7FC0)       # bf16 NaN
    r = ops.attn_bwd_planes(f16(qp), f16(kp), f16(vp), o, dop, lse, B, Sq, Sk, D, md, H, 0.0, biases)
    torch.cuda.synchronize()
    got = [(pl.hi[:, :D].float().cpu(), db.cpu()) for pl, db in r[:3]]
    qr, kr, vr = (pl.fh[:, :D].double().cpu().view(B, S_, D).requires_grad_() for pl, S_ in ((qp, Sq), (kp, Sk), (vp, Sk)))
    (_attention_fp64(qr, kr, vr, mask, H) * dop.hi.double().cpu().view(B, Sq, D)).sum().backward()
    for name, (gn, bn), ref in zip(("dq", "dk", "dv"), got, (qr.grad, kr.grad, vr.grad)):
        ref2 = ref.reshape(-1, D)
        assert torch.isfinite(gn).all() and torch.isfinite(bn).all(), f"{name}: non-finite values"
        if float(ref2.abs().max()) == 0.0:
            assert float(gn.abs().max()) == 0.0 and float(bn.abs().max()) == 0.0, name
            continue
        e = rel_err(gn, ref2)
        assert e < 6e-3, f"{name} ({zero}): {e:.3e}\n" + report(gn, ref2, name)
    # rows without gradient get exactly zero dQ
    dead = (dop.hi.float().cpu().view(B * Sq, H, dk).abs().amax(-1) == 0)          # (row, head)
    dq = got[0][0].view(B * Sq, H, dk)
    assert float(dq[dead].abs().max() if dead.any() else 0.0) == 0.0


# ---------------------------------------------------------------------------------------------- train_prop under hipGraph
def test_captured_proposal_step_equals_the_eager_step(golden):
    """ProposalTrainStep.capture: the whole train_prop step (zero_grad .. Adam) as one hipGraph over a batch whose targets are padded to a
    fixed number of rows (batch index -1: skipped by the target assignment).  Two models from the same state_dict, dropout off: eager steps
    on one, capture + replays on the other (the capture's warm-up steps are steps too) -- same losses and weights to fp32 atomics noise;
    a replay with FEWER events than the capture held equals the eager step on those events."""
    import copy
    from bmt_amd import synthetic as syn
    from bmt_amd.model.proposal_generator import MultimodalProposalGenerator
    from bmt_amd.train import ProposalTrainStep
    from tests.test_gpu_proposal import _prop_cfg
    g = golden("tiny_prop.npz")
    cfg = _prop_cfg()
    cfg.lr, cfg.grad_clip = 1e-3, None
    anchors = {"audio": [float(a) for a in g.np("anchors_audio")], "video": [float(a) for a in g.np("anchors_video")]}

    def make():
        torch.manual_seed(0)
        m = MultimodalProposalGenerator(cfg, anchors)
        m.load_state_dict(g.sub("sd/"))
        m = m.to(DEV)
        for mod in m.modules():
            if hasattr(mod, "dout_p"):
                mod.dout_p = 0.0
        return m
    fs = {k: g[k].to(DEV) for k in ("rgb", "flow", "audio")}
    tg = g["targets"].to(DEV)
    fewer = tg[: max(1, tg.shape[0] - 2)]
    ma, mb = make(), make()
    sa, sb = ProposalTrainStep(ma, cfg, pad_idx=1), ProposalTrainStep(mb, cfg, pad_idx=1)
    la = [float(sa(fs, tg)[1]) for _ in range(4)] + [float(sa(fs, fewer)[1])]
    sb.capture(fs, tg, max_events=tg.shape[0] + 5, warmup=2)           # = eager steps 1, 2, and the capture pass itself runs nothing
    lb = [float(sb.replay()[1]) for _ in range(2)] + [float(sb.replay(fs, fewer)[1])]
    torch.cuda.synchronize()
    for a, b in zip(la[2:], lb):
        assert abs(a - b) < 2e-3 * max(1.0, abs(a)), (la, lb)
    for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert float((pa - pb).abs().max()) < 5e-4, k
    # padded targets on the eager path too: same assignment as the unpadded ones
    mc = make()
    sc = ProposalTrainStep(mc, cfg, pad_idx=1)
    l_pad = float(sc(fs, ProposalTrainStep.pad_targets(tg, tg.shape[0] + 7))[1])
    assert abs(l_pad - la[0]) < 1e-5 * max(1.0, abs(la[0]))


# ---------------------------------------------------------------------------------------------- proposal heads: backward without fp32 intermediates
@pytest.mark.parametrize("R,C", [(70, 512), (300, 144), (33, 20), (4100, 384)])
def test_planes_through_a_relu_gate(ops, R, C):
    """bmt_planes_gate: planes and column sums of (y != 0) ? dy * scale : 0 == bmt_gate followed by bmt_planes"""
    dy, y = rnd(R, C, seed=R).to(DEV), torch.relu(rnd(R, C, seed=C)).to(DEV)
    cs = torch.zeros(C, device=DEV)
    pl = ops.make_planes(dy, "x3", colsum=cs, gate=(y, 1.25))
    dz = torch.where(y != 0, dy * 1.25, torch.zeros_like(dy))
    cs0 = torch.zeros(C, device=DEV)
    want = ops.make_planes(dz, "x3", colsum=cs0)
    assert torch.equal(pl.hi, want.hi) and torch.equal(pl.lo, want.lo)
    assert_close(cs, cs0, atol=1e-4, rtol=1e-5, name="column sums")


@pytest.mark.parametrize("B,S,C,halo", [(2, 40, 512, 15), (3, 17, 144, 2), (1, 100, 64, 39), (2, 9, 20, 0)])
def test_padded_planes_through_a_relu_gate(ops, B, S, C, halo):
    from bmt_amd import _lib
    dy, y = rnd(B, S, C, seed=S).to(DEV), torch.relu(rnd(B, S, C, seed=C)).to(DEV)
    tail = 64 + 2 * halo + 1
    rows = B * (S + 2 * halo) + tail
    hi = torch.full((rows, ops._pad64(C)), 3.0, device=DEV, dtype=torch.bfloat16)
    db = torch.zeros(C, device=DEV)
    _lib.check(ops.lib.bmt_pad_planes_gate(ops._p(dy), ops._p(y), 2.0, B, S, C, halo, tail, ops._p(hi), hi.stride(0), ops._p(db), ops._st()), "ppg")
    dz = torch.where(y != 0, dy * 2.0, torch.zeros_like(dy))
    want = ops.pad_planes(dz, halo, tail, "bwd")
    assert torch.equal(hi, want.hi)
    assert_close(db, want.hi.float().sum(0)[:C], atol=2e-3, rtol=1e-4, name="bias gradient (sums of the bf16 values)")


@pytest.mark.parametrize("N,C,k,cin", [(16, 24, 5, 64), (512, 128, 211, 128), (40, 1024, 79, 1024), (9, 48, 1, 64)])
def test_conv_weight_relayout(ops, N, C, k, cin):
    """bmt_conv_weight_planes / bmt_conv_weight_grad against the permute they replace"""
    from bmt_amd import _lib
    from bmt_amd.model.proposal_generator import _conv_weight_planes
    W = rnd(N, C, k, seed=k).to(DEV)
    got = _conv_weight_planes(W, cin, "w2")
    Wp = torch.zeros(N, k, cin, device=DEV)
    Wp[:, :, :C] = W.permute(0, 2, 1)
    want = ops.make_planes(Wp.view(N, k * cin), "w2")
    for a, b in ((got.hi, want.hi), (got.fh, want.fh), (got.fl, want.fl)):
        assert torch.equal(a, b)
    dWp = rnd(N, k * cin, seed=7).to(DEV)
    g = torch.ones(N, C, k, device=DEV)
    _lib.check(ops.lib.bmt_conv_weight_grad(ops._p(dWp), k * cin, N, C, k, cin, ops._p(g), ops._st()), "cwg")
    assert torch.equal(g, 1.0 + dWp.view(N, k, cin)[:, :, :C].permute(0, 2, 1))


@pytest.mark.parametrize("rows,D,p", [(300, 128, 0.1), (4100, 1024, 0.1), (70, 1024, 0.0), (50, 300, 0.1), (928, 300, 0.1), (33, 20, 0.0), (9, 260, 0.1)])
def test_layernorm_backward_emits_the_next_consumers_gradient_plane(ops, rows, D, p):
    """ops.request_grad_plane: the producer of a ResidualConnection's input asks for dropout_site(d x) as an operand plane; the LayerNorm
    backward (bmt_layernorm_bwd_emit) writes it with dx and leaves its column partials -- bit-identical to the separate conversion pass over
    dx (same mask: same (seed, step, site, element)), which it replaces.  D = 300 / 20 (the decoder's width; no multiple of 64): the plane's
    pad columns up to the next multiple of 64 come out as zeros."""
    ops.manual_seed(123)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.3).to(DEV).requires_grad_()
    gamma, beta = (torch.randn(D, generator=g).to(DEV).requires_grad_() for _ in range(2))
    w1, w2 = (torch.randn(rows, D, generator=g).to(DEV) for _ in range(2))
    site = ops.new_site()
    xin = x * 1.0                                   # (a non-leaf: stands for the previous sublayer's output)
    ops.request_grad_plane(xin, p, site)
    seen = {}
    xin.register_hook(lambda gr: seen.setdefault("g", gr))
    xid, xn = ops.residual_norm(xin, gamma, beta, 1e-5, ops.PREC_BF16X3)
    ((xid * w1).sum() + (xn * w2).sum()).backward()
    torch.cuda.synchronize()
    dx = seen["g"]
    gp = getattr(dx, "_bmt_gplane", None)
    assert gp is not None, "the LayerNorm backward did not hand the plane back"
    pl, gp_p, gp_site, ws, nblk = gp
    assert (gp_p, gp_site) == (p, site)
    cs = torch.zeros(D, device=DEV)
    want = ops.make_planes(dx.view(rows, D), "bwd", colsum=cs, drop=(p, site) if p > 0 else None)
    assert pl.hi.shape == want.hi.shape and torch.equal(pl.hi, want.hi)          # (pad columns included: zeros in both)
    assert pl.hi.shape[1] % 64 == 0 and not bool(pl.hi[:, D:].any())
    got_cs = ws.view(nblk, 3 * D)[:, 2 * D:].sum(0)
    assert_close(got_cs, cs, atol=2e-3, rtol=1e-4, name="column partials of the masked gradient")
    # and the consumer side: the plane is taken for the matching dropout site, not for another one
    b = torch.zeros(D, device=DEV)
    P1, done = ops.grad_planes_from(dx, dx.view(rows, D), None, (p, site) if p > 0 else None)
    assert P1 is pl and done
    P2, _ = ops.grad_planes_from(dx, dx.view(rows, D), None, (0.25, site + 1))
    assert P2 is not pl
