"""-m gpu: a train_cap batch differentiated in parts that are in flight together (bmt_amd.train.CaptioningTrainStep(microbatches=M)).

The reference steps a batch in one pass (epoch_loops/captioning_epoch_loops.py:120-143); the parts are a scheduling decision of this
library, so the bar is the library's own plain pass: same loss, same gradient sums up to the order of fp32 additions (weight-gradient
products and column sums run over the rows of a part instead of the batch), same trajectory under a captured replay.  The plain pass is
what tests/test_gpu_model.py holds against the oracle."""
import pytest
import torch

from bmt_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg, V, glove=True):
    from bmt_amd.model.captioning_module import BiModalTransformer
    cfg.device = DEV
    torch.manual_seed(0)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps) if glove else None))
    return m.to(DEV)


def _batch(cfg, B, Tv, Ta, Tc, V, seed=3):
    b = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    return {k: v.to(DEV) for k, v in b["feature_stacks"].items()}, b["captions"].to(DEV)


def _grads_after_pass(M, cfg, V, fs, caps, glove=True):
    from bmt_amd.train import CaptioningTrainStep
    model = _model(cfg, V, glove)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, microbatches=M, seed=11)
    step._forward_backward(fs, caps)       # (a model's first pass runs its parts one after the other: operand planes are being registered)
    kl, n = step._forward_backward(fs, caps)
    torch.cuda.synchronize()
    if M > 1:
        from bmt_amd import ops
        assert step._parts_last == (min(M, caps.shape[0]), "one after the other" if ops.ENC_STREAMS < 2 else "in flight together")
    return float(kl), int(n), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}


def _compare(g1, g2, tol):
    assert g1.keys() == g2.keys()
    nmax = max(float(v.double().norm()) for v in g1.values())
    worst = 0.0
    for k in g1:
        a, b = g1[k].double(), g2[k].double()
        n = float(a.norm())
        e = float((a - b).norm())
        if n < 1e-3 * nmax:          # an analytically zero gradient (a key projection's bias): rounding noise against rounding noise
            assert e <= max(tol, 2e-3) * nmax, k
            continue
        worst = max(worst, e / n)
        assert e <= tol * n, f"{k}: |a - b| = {e:.3e}, |a| = {n:.3e} ({e / n:.2e})"
    return worst


def test_the_parts_dropout_streams():
    """bmt_rng_derive: part 0 draws from the device's stream itself, every other part from a stream with a seed of its own; all at the
    device's step"""
    from bmt_amd import ops
    ops.manual_seed(1234)
    ops.rng_advance()
    base = ops.rng_tensor(torch.device(DEV)).clone()
    outs = [ops.rng_derive(torch.zeros(2, dtype=torch.int64, device=DEV), i).tolist() for i in range(4)]
    assert outs[0] == base.tolist()
    assert len({o[0] for o in outs}) == 4 and all(o[1] == int(base[1]) for o in outs)
    again = ops.rng_derive(torch.zeros(2, dtype=torch.int64, device=DEV), 2).tolist()
    assert again == outs[2]


@pytest.mark.parametrize("M,B", [(2, 4), (3, 5), (2, 3)])
def test_a_batch_in_parts_is_the_same_pass(M, B):
    """dropout off: loss and every gradient of the M-part pass against the plain pass -- to fp32 summation order (1e-4 of a tensor's norm; state
    leaking between the parts' contexts, a missed part or a gradient written instead of accumulated shows up at order one)"""
    V, Tv, Ta, Tc = 500, 48, 150, 12
    cfg = syn.cfg_config1(dout_p=0.0)
    fs, caps = _batch(cfg, B, Tv, Ta, Tc, V)
    kl1, n1, g1 = _grads_after_pass(1, cfg, V, fs, caps)
    klM, nM, gM = _grads_after_pass(M, cfg, V, fs, caps)
    assert n1 == nM
    # (a) against the plain passes over the parts' own samples, added up: the same launches on the same shapes -- fp32 summation order only
    cuts = [(B * i) // M for i in range(M + 1)]
    kls, gs = 0.0, None
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        k, _, g = _grads_after_pass(1, cfg, V, {n_: v[lo:hi].contiguous() for n_, v in fs.items()}, caps[lo:hi].contiguous())
        kls += k
        gs = g if gs is None else {n_: gs[n_] + g[n_] for n_ in gs}
    assert abs(kls - klM) <= 2e-5 * abs(kls), (kls, klM)
    worst_sum = _compare(gs, gM, 2e-4)
    # (b) against the plain pass over the whole batch: kernels are chosen by shape (tile forms, the attention kernels' grid-dependent variants), so
    # operand roundings differ from a B-row launch to a B/M-row launch -- the difference is rounding noise of the 16-bit operands, far below the
    # backward's own error against fp32 (tests/test_gpu_model.py::_check_grads: 0.8 % overall, 4.5 % per tensor)
    assert abs(kl1 - klM) <= 1e-3 * abs(kl1), (kl1, klM)
    worst = _compare(g1, gM, 2e-2)
    print(f"\nM={M} B={B}: sum-KL {kl1:.6f} / {klM:.6f}; worst gradient tensor: {worst_sum:.2e} of its norm against the parts' plain passes added up, "
          f"{worst:.2e} against the plain pass over the batch")


def test_parts_with_a_trainable_embedding():
    """the word embedding's gradient is scattered with atomics straight into its static buffer by every part"""
    V, B = 300, 4
    cfg = syn.cfg_tiny(dout_p=0.0)
    cfg.unfreeze_word_emb = True
    fs, caps = _batch(cfg, B, 20, 36, 9, V)
    _, _, g1 = _grads_after_pass(1, cfg, V, fs, caps, glove=False)
    _, _, g2 = _grads_after_pass(2, cfg, V, fs, caps, glove=False)
    assert any("emb_C" in k for k in g1)
    _compare(g1, g2, 2e-4)


def test_parts_one_after_the_other_on_one_stream(monkeypatch):
    """ops.ENC_STREAMS = 1 (what bench.py's kernel timer and the profiling scripts set): the same parts issued one after the other"""
    from bmt_amd import ops
    V, B = 500, 4
    cfg = syn.cfg_config1(dout_p=0.0)
    fs, caps = _batch(cfg, B, 48, 150, 12, V)
    kl2, _, g2 = _grads_after_pass(2, cfg, V, fs, caps)
    monkeypatch.setattr(ops, "ENC_STREAMS", 1)
    kls, _, gs = _grads_after_pass(2, cfg, V, fs, caps)
    assert abs(kl2 - kls) <= 2e-5 * abs(kl2)
    _compare(g2, gs, 2e-4)


def test_captured_parts_replay_the_eager_trajectory():
    """four optimizer steps: eager M = 2 against a captured M = 2 step's replays (dropout off), and against the plain pass's losses"""
    from bmt_amd.train import CaptioningTrainStep
    V, B = 500, 4
    cfg = syn.cfg_config1(dout_p=0.0)
    fs, caps = _batch(cfg, B, 48, 150, 12, V)
    losses = {}
    for name, M, captured in (("plain", 1, False), ("parts", 2, False), ("parts, captured", 2, True)):
        model = _model(cfg, V)
        step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, microbatches=M, seed=11)
        ls = []
        if captured:
            step.capture(fs, caps, warmup=1)            # (two eager warm-up steps whatever is asked for: a model's first step runs its parts in turn)
            ls += [None, None]
            for _ in range(2):
                loss, _ = step.replay()
                ls.append(float(loss))
        else:
            for _ in range(4):
                loss, _ = step(fs, caps)
                ls.append(float(loss))
        losses[name] = ls
    print("\n", losses)
    for a, b in zip(losses["plain"], losses["parts"]):
        assert abs(a - b) < 3e-3          # (Adam's first steps amplify summation-order noise: tests/study_adam_drift.py)
    for a, b in zip(losses["parts"][2:], losses["parts, captured"][2:]):
        assert abs(a - b) < 3e-3


def test_parts_under_dropout_draw_their_own_masks():
    """dropout on: the step runs, the loss is finite and close to the plain pass's (same expectation), the device's dropout stream advances by
    ONE step per optimizer step, and no part's context keeps its override afterwards"""
    from bmt_amd import ops
    from bmt_amd.train import CaptioningTrainStep
    V, B = 500, 6
    cfg = syn.cfg_config1(dout_p=0.1)
    fs, caps = _batch(cfg, B, 48, 150, 12, V)
    model = _model(cfg, V)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, microbatches=2, seed=5)
    s0 = int(ops.rng_tensor(torch.device(DEV))[1])
    l1, _ = step(fs, caps)
    l2, _ = step(fs, caps)
    assert int(ops.rng_tensor(torch.device(DEV))[1]) == s0 + 2
    assert torch.isfinite(l1) and torch.isfinite(l2) and float(l1) != float(l2)
    assert ops.context().rng is None
    model1 = _model(cfg, V)
    plain = CaptioningTrainStep(model1, cfg, syn.PAD_IDX, static_grads=True, seed=5)
    p1, _ = plain(fs, caps)
    assert abs(float(p1) - float(l1)) < 0.05 * abs(float(p1))
