"""-m gpu: PACKED ROWS (bmt_amd.ops.PACK_ROWS, ABI 7) -- the valid positions of a ragged batch compacted, padded positions never computed.

The reference only ever reads a padded position as a MASKED key (model/multihead_attention.py:17; masks from
epoch_loops/captioning_epoch_loops.py:105-112), so leaving those rows out changes nothing observable: every check here is either
against the padded-dense path on the same inputs or against the CPU oracle.  Covered: the layout kernel (holes inside a sequence, empty
samples), every row-wise kernel family under a device-side row count (rows past the count neither read -- they hold NaN here -- nor
written), the attention kernels that take per-sample row offsets, and the whole model incl. a batch whose masks are NOT suffixes."""
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


def _mask(B, S, seed, holes=True, empty=None):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(S // 3, 1), S + 1, (B,), generator=g)
    lens[0] = S
    m = torch.arange(S)[None, :] < lens[:, None]
    if holes:
        for b in range(1, B, 2):
            t = int(torch.randint(0, max(int(lens[b]) - 1, 1), (1,), generator=g))
            m[b, t] = False
    if empty is not None:
        m[empty] = False
    return m.view(B, 1, S)


def _pack_host(m):
    """(off [B + 1], row_map [n]) of a (B, 1, S) bool mask, by numpy"""
    B, S = m.shape[0], m.shape[-1]
    mm = m.view(B, S).numpy()
    cnt = mm.sum(1)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    rows = np.nonzero(mm.reshape(-1))[0]
    return off, rows


@pytest.mark.parametrize("B,S,empty", [(1, 5, None), (3, 70, None), (32, 800, None), (7, 1000, 3), (4, 256, 0), (300, 33, 299)])
def test_pack_rows_layout(ops, B, S, empty):
    m = _mask(B, S, seed=B * S, empty=empty)
    pk = ops.pack_rows(m.to(DEV))
    off, rows = _pack_host(m)
    got_off = pk.off[:B + 1].cpu().numpy()
    assert np.array_equal(got_off, off), (got_off, off)
    assert np.array_equal(pk.row_map[:off[-1]].cpu().numpy(), rows)
    assert pk.cap == B * S


def _pack_for_rows(ops, M, n):
    """a RowPack of capacity M whose device-side count is n (one 'sample' of M positions, the first n valid)"""
    m = torch.zeros(1, 1, M, dtype=torch.bool)
    m[0, 0, :n] = True
    return ops.pack_rows(m.to(DEV))


GEMM_CASES = [  # (M, N, K, precision, kind) -- one per kernel family of the encoder
    (8192, 3072, 1024, "w2", "fwd"),      # gemm_wide_kernel (256 x 256 tiles)
    (25600, 384, 128, "w2", "fwd"),       # gemm_k128_kernel
    (8192, 1024, 1024, "w2", "fwd"),      # gemm_pipe_kernel, 256 x 128 tiles
    (25600, 128, 1024, "w2", "fwd"),      # gemm_pipe_kernel, 128-row tiles
    (8192, 1024, 4096, "f16", "fwd"),     # one fp16 plane (FFN-2 of a shallow encoder)
    (928, 300, 600, "x3", "fwd"),         # gemm_small_kernel (32 x 32 tiles: the decoder's three-pass products)
    (4000, 512, 600, "x3", "fwd"),        # the register-staged three-pass kernel + split-K epilogue
    (8192, 1024, 1024, "bwd", "dx"),      # dX = dY . W, the weight k-major
    (25600, 128, 1024, "bwd", "dx"),      # ... split over the reduction (few tiles)
    (8192, 1024, 4096, "bwd", "dxcs"),    # ... as a plane with its column sums (FFN-2's dX: the bias gradient of FFN-1)
]


@pytest.mark.parametrize("M,N,K,prec,kind", GEMM_CASES)
@pytest.mark.parametrize("frac", [0.76, 0.0, 1.0])
def test_gemm_families_under_a_device_side_row_count(ops, M, N, K, prec, kind, frac):
    """C[:n] of the packed launch == C[:n] of the full launch bit for bit (the same tiles do the same arithmetic), rows >= n are left
    alone, column sums cover rows < n only -- for every GEMM kernel the encoder's row-wise products run on.  The operand rows >= n hold
    NaN: nothing may read them."""
    n = int(round(M * frac))
    if frac not in (0.0, 1.0):
        n -= 37                               # not a multiple of any tile
    P = {"w2": ops.PREC_F16W2, "f16": ops.PREC_F16, "x3": ops.PREC_BF16X3, "bwd": ops.PREC_BF16}[prec]
    x = (rnd(M, K if kind == "fwd" else N, seed=1) * 0.5).to(DEV)
    W = (rnd(N, K, seed=2) * 0.1).to(DEV)
    bias = rnd(N, seed=3).to(DEV) if kind == "fwd" else None
    pk = _pack_for_rows(ops, M, n)
    afmt = ops.act_fmt(P) if kind == "fwd" else "bwd"
    A_full = ops.make_planes(x, afmt)
    xp = x.clone()
    xp[n:] = float("nan")
    A_pack = ops.make_planes(xp, afmt, pack=pk)
    for t in (A_pack.hi, A_pack.lo, A_pack.fh, A_pack.fl):      # (make_planes left rows >= n unwritten: poison them)
        if t is not None:
            t[n:] = float("nan")
    Wp = ops.weight_planes(W, ops.weight_fmt(P) if kind == "fwd" else "bwd")
    outN = N if kind == "fwd" else K

    def run(A, poison):
        out = torch.full((M, outN), 7.0 if poison else 0.0, device=DEV)
        op = ops._alloc_planes(M, outN, "f16", DEV, ld=ops._pad64(outN))
        op.hi.fill_(3.0)
        op.fh.fill_(3.0)
        cs = torch.zeros(outN, device=DEV) if kind == "dxcs" else None
        if kind == "fwd":
            ops.gemm_bf16(A, Wp, out, ldc=outN, bias=bias, precision=P, relu=True, out_planes=op)
        elif cs is not None:                  # (column sums come with a plane-only output)
            ops.gemm_bf16(A, Wp, None, precision=P, b_km=True, out_planes=ops.Planes(op.hi, None, M, outN), colsum=cs)
            out = op.hi[:, :outN].float()
        else:
            ops.gemm_bf16(A, Wp, out, ldc=outN, precision=P, b_km=True)
        torch.cuda.synchronize()
        return out, op, cs
    full, fop, fcs = run(A_full, False)
    got, gop, gcs = run(A_pack, True)
    assert torch.equal(got[:n], full[:n]), report(got[:n], full[:n], "packed rows")
    if kind == "fwd":
        assert torch.equal(gop.hi[:n], fop.hi[:n]) and torch.equal(gop.fh[:n], fop.fh[:n])
        assert bool((got[n:] == 7.0).all()) and bool((gop.hi[n:] == 3.0).all()), "rows past the count were written"
    elif gcs is not None:
        assert bool((gop.hi[n:] == 3.0).all()), "rows past the count were written"
        want = full[:n].double().sum(0)
        assert_close(gcs, want, atol=2e-2 * max(1.0, float(want.abs().max())), name="column sums over the rows that exist")
    else:
        assert bool((got[n:] == 7.0).all()), "rows past the count were written"


@pytest.mark.parametrize("rows,n", [(8192, 6100), (25600, 19508), (600, 0), (600, 600), (600, 150)])
def test_weight_gradients_reduce_over_the_rows_that_exist(ops, rows, n):
    """dW = dY^T . X through the grouped launch (and the single launch): the reduction stops at the device-side count; rows past it hold NaN"""
    N, K = 384, 256
    dy, x = rnd(rows, N, seed=4).to(DEV), rnd(rows, K, seed=5).to(DEV)
    pk = _pack_for_rows(ops, rows, n)
    dyn, xn = dy.clone(), x.clone()
    dyn[n:] = float("nan")
    xn[n:] = float("nan")
    Pd, Px = ops.make_planes(dyn, "bwd", pack=pk), ops.make_planes(xn, "bwd", pack=pk)
    Pd.hi[n:] = float("nan")
    Px.hi[n:] = float("nan")
    want = dy[:n].to(torch.bfloat16).double().t() @ x[:n].to(torch.bfloat16).double()
    for grouped in (True, False):
        out = torch.zeros(N, K, device=DEV)
        out2 = torch.zeros(64, 64, device=DEV)
        if grouped:
            small = (ops.make_planes(rnd(100, 64, seed=6).to(DEV), "bwd"), ops.make_planes(rnd(100, 64, seed=7).to(DEV), "bwd"), out2)
            ops.gemm_bf16_grouped([(Pd, Px, out), small])
        else:
            ops.gemm_bf16(Pd, Px, out, ldc=K, accum=True, splitk=ops._splitk_for(N, K, rows), precision=ops.PREC_BF16, a_km=True, b_km=True)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert_close(out, want, atol=1e-3 * max(1.0, float(want.abs().max())), name=f"dW grouped={grouped}")


@pytest.mark.parametrize("rows,D,n", [(8192, 1024, 6000), (25600, 128, 19001), (300, 1024, 0), (300, 128, 300)])
def test_layernorm_and_plane_conversion_under_a_row_count(ops, rows, D, n):
    pk = _pack_for_rows(ops, rows, n)
    x = rnd(1, rows, D, seed=8).to(DEV)
    gamma, beta = (rnd(D, seed=9) * 0.3 + 1).to(DEV).requires_grad_(), rnd(D, seed=10).to(DEV).requires_grad_()
    xp = x.clone()
    xp[0, n:] = float("nan")
    xp.requires_grad_()
    xd = x.clone().requires_grad_()
    ops.carry_pack(pk, xp)
    _, yn = ops.residual_norm(xp, gamma, beta, 1e-5, ops.PREC_F16W2)
    _, yd = ops.residual_norm(xd, gamma, beta, 1e-5, ops.PREC_F16W2)
    assert ops.pack_of(yn) is pk and ops.planes_of(yn, "f16").pack is pk
    assert torch.equal(yn[0, :n], yd[0, :n])
    assert torch.equal(ops.planes_of(yn, "f16").fh[:n], ops.planes_of(yd, "f16").fh[:n])
    w = rnd(1, rows, D, seed=11).to(DEV)
    wn = w.clone()
    wn[0, n:] = float("nan")
    gp = torch.autograd.grad((yn * 1.0), (xp, gamma, beta), grad_outputs=wn, retain_graph=False)
    gd = torch.autograd.grad((yd[:, :n] * 1.0), (xd, gamma, beta), grad_outputs=w[:, :n])
    assert torch.equal(gp[0][0, :n], gd[0][0, :n])
    for a, b, name in ((gp[1], gd[1], "dgamma"), (gp[2], gd[2], "dbeta")):
        assert torch.isfinite(a).all(), name
        assert_close(a, b, atol=2e-3 * max(1.0, float(b.abs().max())), name=name)
    # the upstream-gradient conversion (+ bias column sums through a dropout mask)
    cs_p, cs_d = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.manual_seed(3)
    Pp = ops.make_planes(wn[0], "bwd", colsum=cs_p, drop=(0.1, 77), pack=pk)
    Pdn = ops.make_planes(w[0], "bwd", colsum=torch.zeros(D, device=DEV), drop=(0.1, 77))
    assert torch.equal(Pp.hi[:n], Pdn.hi[:n])
    ops.manual_seed(3)
    ref = ops.dropout_raw(w[0].contiguous(), 0.1, 77)[:n].double().sum(0)
    assert_close(cs_p, ref, atol=2e-2, name="column sums of the rows that exist")


def _scatter(packed, pk_off, rows, B, S, D):
    """padded (B, S, D) tensor holding the packed rows at their positions (zeros elsewhere)"""
    out = torch.zeros(B * S, D, dtype=packed.dtype)
    out[torch.from_numpy(rows)] = packed[:len(rows)]
    return out.view(B, S, D)


ATTN_PACKED = [  # (B, H, Sq, Sk, dk, q packed?, holes) -- forward kernel / backward form
    (3, 4, 200, 200, 128, True, True),       # attn_fwd64 + split backward, self-attention layout
    (2, 2, 300, 130, 256, True, True),       # cross-attention: two different layouts
    (24, 4, 800, 800, 256, True, False),     # attn_fwd32 (>= 2 workgroups per CU) + split backward: the audio self-attention
    (20, 8, 256, 800, 128, True, True),      # attn_fwd32 at d_k 128
    (4, 4, 30, 333, 256, False, True),       # the decoder's encoder-decoder attention: dense queries, packed memory; paired backward
    (3, 4, 40, 100, 128, True, True),        # short packed queries (< 64): paired backward with packed queries
    (3, 2, 150, 150, 256, True, "empty"),    # a sample without a single valid position
]


@pytest.mark.parametrize("B,H,Sq,Sk,dk,qpacked,holes", ATTN_PACKED)
def test_attention_over_packed_rows(ops, B, H, Sq, Sk, dk, qpacked, holes):
    """forward + backward over packed q / k / v planes against the SAME kernels on the padded layout with the key-padding mask: outputs
    and gradients of every valid row agree (same tiles per sample up to where a hole shifts them), the bias partials of tiles past a
    sample's length are zero rows, rows past the packed count are never written."""
    D = H * dk
    empty = 1 if holes == "empty" else None
    mk = _mask(B, Sk, seed=31 + Sk, holes=bool(holes), empty=empty)
    mq = mk if (Sq == Sk and qpacked) else (_mask(B, Sq, seed=77 + Sq, holes=bool(holes), empty=empty) if qpacked else torch.ones(B, 1, Sq, dtype=torch.bool))
    offk, rowsk = _pack_host(mk)
    offq, rowsq = _pack_host(mq)
    nq, nk = len(rowsq), len(rowsk)
    qv, kv, vv = rnd(B * Sq, D, seed=1) * 0.7, rnd(B * Sk, D, seed=2) * 0.7 + 0.3, rnd(B * Sk, D, seed=3)
    dov = rnd(B * Sq, D, seed=4)
    pkk = ops.pack_rows(mk.to(DEV))
    pkq = pkk if mq is mk else (ops.pack_rows(mq.to(DEV)) if qpacked else None)

    def planes16(t, n, pack):
        tt = t.clone()
        if pack is not None:
            tt[n:] = float("nan")
        pl = ops.make_planes(tt.to(DEV), "f16")
        return ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh, pack=pack)
    # packed operands: the first n rows are the valid rows in (b, t) order; the dense ones hold the same rows at their padded positions
    qP, kP, vP = planes16(qv, nq, pkq), planes16(kv, nk, pkk), planes16(vv, nk, pkk)
    qD = planes16(_scatter(qv, offq, rowsq, B, Sq, D).view(-1, D) if qpacked else qv, 0, None)
    kD = planes16(_scatter(kv, offk, rowsk, B, Sk, D).view(-1, D), 0, None)
    vD = planes16(_scatter(vv, offk, rowsk, B, Sk, D).view(-1, D), 0, None)
    oP, lseP = ops.attn_fwd_planes(qP, kP, vP, B, Sq, Sk, D, None, H, precision=ops.PREC_F16, out_fmt="f16")
    oD, lseD = ops.attn_fwd_planes(qD, kD, vD, B, Sq, Sk, D, mk.to(DEV), H, precision=ops.PREC_F16, out_fmt="f16")
    torch.cuda.synchronize()
    assert oP.pack is pkq
    qsel = torch.from_numpy(rowsq) if qpacked else torch.arange(B * Sq)
    keys_of_q = torch.from_numpy(np.repeat(mk.view(B, Sk).numpy().any(1), Sq))[qsel]        # queries whose sample has at least one key
    got, want = oP.fh[:nq].float().cpu(), oD.fh.float().cpu()[qsel]
    assert torch.isfinite(got[keys_of_q]).all()
    assert_close(got[keys_of_q], want[keys_of_q], atol=2e-3, name="O over packed rows")
    assert bool(torch.isnan(got[~keys_of_q]).all()), "a query without any key must come out NaN, as the reference's softmax"
    # lse keeps its padded [B, H, Sq] layout, indexed by the position WITHIN the packed sample
    if qpacked:
        for b in range(B):
            nb = int(offq[b + 1] - offq[b])
            tb = rowsq[offq[b]:offq[b + 1]] - b * Sq
            if nb and bool(mk[b].any()):
                assert_close(lseP[b, :, :nb], lseD[b][:, torch.from_numpy(tb)], atol=2e-3, name=f"lse of sample {b}")
    if empty is not None:
        return
    # ---- backward
    def doplane(t, n, pack):
        tt = t.clone()
        if pack is not None:
            tt[n:] = float("nan")
        pl = ops.make_planes(tt.to(DEV), "bwd")
        return ops.Planes(pl.hi[:, :D].contiguous(), None, pl.rows, D, pack=pack)
    doP = doplane(dov, nq, pkq)
    doD = doplane(_scatter(dov, offq, rowsq, B, Sq, D).view(-1, D) if qpacked else dov, 0, None)
    bP = tuple(torch.zeros(D, device=DEV, requires_grad=True) for _ in range(3))
    bD = tuple(torch.zeros(D, device=DEV, requires_grad=True) for _ in range(3))
    rP = ops.attn_bwd_planes(qP, kP, vP, oP, doP, lseP, B, Sq, Sk, D, None, H, 0.0, bP)
    rD = ops.attn_bwd_planes(qD, kD, vD, oD, doD, lseD, B, Sq, Sk, D, mk.to(DEV), H, 0.0, bD)
    torch.cuda.synchronize()
    ksel = torch.from_numpy(rowsk)
    qlive = keys_of_q
    for name, (gp, bp), (gd, bd), sel, n_, ok in zip(("dq", "dk", "dv"), rP[:3], rD[:3], (qsel, ksel, ksel), (nq, nk, nk),
                                                      (qlive, torch.ones(nk, dtype=torch.bool), torch.ones(nk, dtype=torch.bool))):
        a, b_ = gp.hi[:n_, :D].float().cpu()[ok], gd.hi[:, :D].float().cpu()[sel][ok]
        assert torch.isfinite(a).all(), name
        e = rel_err(a, b_)
        assert e < 4e-3, f"{name}: packed vs padded {e:.3e}\n" + report(a, b_, name)
        assert_close(bp, bd, atol=5e-3 * max(1.0, float(bd.abs().max())), name=f"bias gradient of {name}")


def _build(cfg, V):
    from bmt_amd.model.captioning_module import BiModalTransformer
    cfg.device = DEV
    torch.manual_seed(0)
    return BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(DEV)


def _run(model, cfg, fs, caps):
    from bmt_amd.loss.label_smoothing import LabelSmoothing
    from bmt_amd.model.masking import mask
    fs = {k: v.to(DEV) for k, v in fs.items()}
    caps = caps.to(DEV)
    x, y = caps[:, :-1], caps[:, 1:]
    masks = {}
    masks["V_mask"], masks["C_mask"] = mask(fs["rgb"][:, :, 0], x, syn.PAD_IDX)
    masks["A_mask"] = mask(fs["audio"][:, :, 0], None, syn.PAD_IDX)
    model.eval()
    packs = model.row_packs(fs, masks)
    pred = model(fs, x, masks)
    loss = LabelSmoothing(cfg.smoothing, syn.PAD_IDX)(pred, y) / (y != syn.PAD_IDX).sum()
    return pred, loss, masks, packs


def _poison_allocator():
    """the caching allocator hands out blocks full of NaN: a kernel that reads a row nobody wrote shows up in the result"""
    junk = [torch.full((64 << 20,), float("nan"), device=DEV) for _ in range(6)]
    junk += [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(64)]
    del junk


@pytest.mark.parametrize("holes", [False, True])
def test_model_on_packed_rows_against_the_oracle_and_the_padded_path(ops, holes):
    """a two-layer model of d_k 128 on a ragged batch -- with holes: masks that are NOT suffixes (a position inside a sequence whose first
    channel equals the pad value, masking.py:14-21) -- on packed rows: log-probabilities within 1e-3 of the CPU oracle (bar of
    BASELINE.json) and gradients within the bars of tests/test_gpu_model.py; and against the padded path of this library: same
    log-probabilities to 2e-4."""
    from oracle import bmt_oracle as orc
    from tests.test_gpu_model import _check_grads
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
    _poison_allocator()
    pred, loss, masks, packs = _run(model, cfg, fs, caps)
    assert packs is not None, "this model should run on packed rows"
    assert int(packs[0].off[B]) == int(masks["A_mask"].sum()) and int(packs[1].off[B]) == int(masks["V_mask"].sum())
    if holes:
        assert not bool(masks["V_mask"][1, 0, 3]) and bool(masks["V_mask"][1, 0, 5])
    p = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    oloss, opred, _ = orc.train_cap_loss(p, cfg, fs, caps, syn.PAD_IDX, cfg.smoothing)
    err = float((pred.detach().cpu() - opred.detach()).abs().max())
    print(f"\npacked rows (holes={holes}): max |dlogp| vs the oracle = {err:.3e}")
    assert_close(pred, opred.detach(), atol=1e-3, name="log-probs on packed rows")
    loss.backward()
    oloss.backward()
    trainable = {k for k, q in model.named_parameters() if q.requires_grad}
    _check_grads([(k, q) for k, q in model.named_parameters() if k in trainable], {k: v.grad for k, v in p.items() if v.grad is not None and k in trainable})
    gpacked = {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}
    # the padded path of this library on the same batch
    model.zero_grad()
    ops.PACK_ROWS = False
    try:
        pred_d, loss_d, _, packs_d = _run(model, cfg, fs, caps)
        assert packs_d is None
        loss_d.backward()
    finally:
        ops.PACK_ROWS = True
    assert_close(pred, pred_d.detach(), atol=5e-4 if holes else 2e-4, name="packed vs padded log-probs")      # (a hole shifts the key tiles of its sample: fp16 roundings fall differently)
    num = sum(float((gpacked[k].double() - q.grad.double()).norm() ** 2) for k, q in model.named_parameters() if q.grad is not None)
    den = sum(float(q.grad.double().norm() ** 2) for k, q in model.named_parameters() if q.grad is not None)
    assert (num / den) ** 0.5 < 5e-3, f"packed vs padded gradients differ by {(num / den) ** 0.5:.3e}"


def test_captured_step_on_packed_rows_follows_the_batch(ops):
    """the step as hipGraphs replays over batches of DIFFERENT valid lengths: the row counts are read from device memory at replay"""
    from bmt_amd.train import CaptioningTrainStep
    cfg = syn.make_cfg(d_model=512, H=4, N=1, d_aud=128, d_vid=256, d_model_caps=64, dout_p=0.0, lr=1e-4)
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
    print("\nlosses eager / graph:", losses)
    for a, b in zip(losses["eager"][:1], losses["graph"][:1]):        # (first step: identical weights; later ones differ by Adam's state of the warm-up)
        assert abs(a - b) < 2e-3, (losses)
    assert all(math.isfinite(x) for x in losses["graph"])


def test_captured_step_prefetches_the_announced_batch(ops):
    """CaptioningTrainStep.replay(..., next_batch=): the next step's batch is copied into the static input buffers beside this step's optimizer
    graph; the losses are those of replays that copy their own batch, also when the batch that arrives is NOT the one that was announced"""
    from bmt_amd.train import CaptioningTrainStep
    cfg = syn.make_cfg(d_model=512, H=4, N=1, d_aud=128, d_vid=256, d_model_caps=64, dout_p=0.0, lr=1e-4)
    V, B, Tv, Ta, Tc = 60, 4, 90, 210, 11
    batches = [syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=s) for s in (5, 6, 7, 8)]
    devb = [({k: v.to(DEV) for k, v in b["feature_stacks"].items()}, b["captions"].to(DEV)) for b in batches]
    losses = {}
    for mode in ("plain", "prefetch", "wrong announcement"):
        model = _build(cfg, V)
        step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, seed=3)
        step.capture(*devb[0])
        out = []
        for i, b in enumerate(devb):
            nxt = devb[(i + 1) % len(devb)] if mode == "prefetch" else (devb[(i + 2) % len(devb)] if mode != "plain" else None)
            loss, _ = step.replay(*b, next_batch=nxt)
            out.append(float(loss))
        torch.cuda.synchronize()
        losses[mode] = out
        step.uncapture()
    print("\nlosses:", losses)
    # (a batch copied into the wrong step shows as another batch's loss -- they are 0.1 apart; two runs of the SAME mode differ by up to 2e-4 in the
    # fourth step -- 0.88409 / 0.88429 / 0.88434 in six runs of "plain" -- because the grouped weight-gradient launch accumulates with atomics and
    # Adam's first steps turn the rounding of a near-zero gradient into +- lr)
    for mode in ("prefetch", "wrong announcement"):
        for a, b in zip(losses["plain"], losses[mode]):
            assert abs(a - b) < 1e-3 * max(1.0, abs(a)), losses
