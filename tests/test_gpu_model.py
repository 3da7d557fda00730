"""-m gpu: module- and model-level parity of bmt_amd.model / bmt_amd.loss (HIP kernels behind the reference's class
surface) against the golden vectors captured from the imported reference, and against the CPU oracle.

Tolerances: masks / indices bit-exact; log-probabilities <= 1e-3 abs (BASELINE.json north_star) -- with the
split-bf16 forward they land around 1e-4; gradients (single-pass bf16 backward) <= 3e-2 relative per tensor."""
import math

import numpy as np
import pytest
import torch

from bmt_amd import synthetic as syn
from tests.gpu_util import assert_close, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOGP_TOL = 1e-3
GRAD_NORM_TOL = 0.02       # |norm - reference norm| / reference norm of a parameter's gradient on the seeded fixtures (measured, round 6: worst 5.7e-3 on
                           # configs[0], 3.6e-3 mid, 3.9e-3 full length; the elementwise bars are _check_grads')
GRAD_REL = 3e-2


def _load(module, sd):
    module.load_state_dict(sd)
    return module.to(DEV)


def _check_grads(named_params, ref_grads, tol=0.045, global_tol=8e-3):
    """Backward runs single-pass bf16 (DESIGN.md "precision"): error over ALL tensors concatenated <= 0.8 %, per tensor <= 4.5 % -- the
    bars follow what is measured (round 4, BMT_GRAD_REPORT=1 over every captioning fixture, gpurun_out/r04_b_grad_report.txt: overall
    0.24-0.66 %, worst tensor 3.55 % = decoder.layers.0.enc_att_A.linear_K2d.weight of the deep fixture, every other tensor <= 2.8 %)
    (with the dQ correction of the attention backward -- ops.ATTN_KMEAN -- the decoder's cross-attention query / key paths, whose
    bf16 cancellation noise was 10-30 %, are within 1 %).  Gradients that are analytically zero (the key-projection bias: softmax
    is invariant to a constant added to every key) come out as cancellation noise in any finite precision -- 1e-9 in the fp32
    reference, bf16-sized here -- and are only required to be small against the largest gradient element around them.
    BMT_GRAD_REPORT=1 prints every tensor's error."""
    import os
    items = [(k, p) for k, p in named_params if k in ref_grads]
    gmax = max(float(ref_grads[k].abs().max()) for k, _ in items)
    nmax = max(float(ref_grads[k].double().norm()) for k, _ in items)
    bad, e2, r2, rows = [], 0.0, 0.0, []
    for k, p in items:
        assert p.grad is not None, f"missing grad for {k}"
        ref = ref_grads[k].double()
        got = p.grad.detach().cpu().double()
        e, n = float((got - ref).norm()), float(ref.norm())
        if n < 1e-4 * nmax:      # analytically-zero gradient
            if float(got.abs().max()) > 0.05 * gmax:
                bad.append(f"{k}: should be ~0, max|got|={float(got.abs().max()):.3e} vs largest gradient element {gmax:.3e}")
            continue
        e2 += e * e
        r2 += n * n
        rows.append((e / n, k, n))
        if e > tol * n:
            bad.append(f"{k}: |err|={e:.3e} |ref|={n:.3e} ({e / n:.1%})\n" + report(p.grad, ref_grads[k], k))
    if os.environ.get("BMT_GRAD_REPORT") == "1":
        print(f"global relative gradient error {(e2 / max(r2, 1e-300)) ** 0.5:.3e}")
        for r, k, n in sorted(rows, reverse=True)[:12]:
            print(f"   {r:8.3%}  |ref| {n:.3e}  {k}")
    if r2 > 0 and (e2 / r2) ** 0.5 > global_tol:
        bad.append(f"global relative gradient error {(e2 / r2) ** 0.5:.3e} > {global_tol}")
    assert not bad, "\n".join(bad)


def test_mha_cross_modal(golden):
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    g = golden("modules_tiny.npz")
    mha = _load(MultiheadedAttention(20, 24, 24, 4, 0.0, 128), g.sub("mha/sd/")).eval()
    Q, K = g["mha/Q"].to(DEV).requires_grad_(), g["mha/K"].to(DEV).requires_grad_()
    out = mha(Q, K, K, g["mha/mask"].to(DEV))
    assert_close(out, g["mha/out"], atol=2e-4, name="mha out")
    (out * g["mha/w"].to(DEV)).sum().backward()
    assert rel_err(Q.grad, g["mha/dQ"]) < GRAD_REL, report(Q.grad, g["mha/dQ"], "dQ")
    assert rel_err(K.grad, g["mha/dK"]) < GRAD_REL, report(K.grad, g["mha/dK"], "dK")
    _check_grads(mha.named_parameters(), g.sub("mha/grad/"))


def test_mha_causal_self_attention(golden):
    from bmt_amd.model.multihead_attention import MultiheadedAttention
    g = golden("modules_tiny.npz")
    sa = _load(MultiheadedAttention(20, 20, 20, 4, 0.0, 128), g.sub("sa/sd/")).eval()
    X = g["sa/X"].to(DEV).requires_grad_()
    out = sa(X, X, X, g["sa/mask"].to(DEV))
    assert_close(out, g["sa/out"], atol=2e-4, name="sa out")
    (out * g["sa/w"].to(DEV)).sum().backward()
    assert rel_err(X.grad, g["sa/dX"]) < GRAD_REL, report(X.grad, g["sa/dX"], "dX")
    _check_grads(sa.named_parameters(), g.sub("sa/grad/"))


def test_attention_function_surface(golden):
    """model.multihead_attention.attention(Q,K,V,mask,dropout) on (B,H,S,d_k) views."""
    from bmt_amd.model.multihead_attention import attention
    from oracle import bmt_oracle as orc
    g = torch.Generator().manual_seed(3)
    Q, K, V = (torch.randn(2, 4, s, 32, generator=g) for s in (9, 13, 13))
    mask = torch.ones(2, 1, 1, 13, dtype=torch.bool)
    mask[1, 0, 0, 9:] = False
    out = attention(Q.to(DEV), K.to(DEV), V.to(DEV), mask.to(DEV))
    assert_close(out, orc.attention(Q.double(), K.double(), V.double(), mask), atol=2e-4, name="attention()")


def test_residual_ffn(golden):
    from bmt_amd.model.blocks import PositionwiseFeedForward, ResidualConnection
    g = golden("modules_tiny.npz")
    sd = g.sub("resffn/sd/")
    res = _load(ResidualConnection(20, 0.0), {k[4:]: v for k, v in sd.items() if k.startswith("res.")}).eval()
    ffn = _load(PositionwiseFeedForward(20, 80, 0.0), {k[4:]: v for k, v in sd.items() if k.startswith("ffn.")}).eval()
    X = g["resffn/X"].to(DEV).requires_grad_()
    out = res(X, ffn)
    assert_close(out, g["resffn/out"], atol=2e-4, name="res+ffn out")
    (out * g["resffn/w"].to(DEV)).sum().backward()
    assert rel_err(X.grad, g["resffn/dX"]) < GRAD_REL
    gr = g.sub("resffn/grad/")
    _check_grads([("res." + k, p) for k, p in res.named_parameters()] + [("ffn." + k, p) for k, p in ffn.named_parameters()], gr)


def test_bridge(golden):
    from bmt_amd.model.blocks import BridgeConnection
    g = golden("modules_tiny.npz")
    br = _load(BridgeConnection(40, 20, 0.0), g.sub("bridge/sd/")).eval()
    X = g["bridge/X"].to(DEV).requires_grad_()
    out = br(X)
    assert_close(out, g["bridge/out"], atol=2e-4, name="bridge out")
    (out * g["bridge/w"].to(DEV)).sum().backward()
    assert rel_err(X.grad, g["bridge/dX"]) < GRAD_REL
    _check_grads(br.named_parameters(), g.sub("bridge/grad/"))


def _build(cfg, V, use_glove, sd=None):
    from bmt_amd.model.captioning_module import BiModalTransformer
    cfg.device = DEV
    torch.manual_seed(0)
    glove = syn.make_glove(V, cfg.d_model_caps) if use_glove else None
    model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, glove))
    if sd is not None:
        model.load_state_dict(sd)
    return model.to(DEV)


def _run_cap(model, cfg, fs, caps, train=False):
    from bmt_amd.loss.label_smoothing import LabelSmoothing
    from bmt_amd.model.masking import mask
    fs = {k: v.to(DEV) for k, v in fs.items()}
    caps = caps.to(DEV)
    x, y = caps[:, :-1], caps[:, 1:]
    masks = {}
    masks["V_mask"], masks["C_mask"] = mask(fs["rgb"][:, :, 0], x, syn.PAD_IDX)
    masks["A_mask"] = mask(fs["audio"][:, :, 0], None, syn.PAD_IDX)
    model.train(train)
    pred = model(fs, x, masks)
    n_tokens = (y != syn.PAD_IDX).sum()
    loss = LabelSmoothing(cfg.smoothing, syn.PAD_IDX)(pred, y) / n_tokens
    return pred, loss, masks


@pytest.mark.parametrize("name", ["tiny_cap.npz", "tiny_cap_trainemb.npz"])
def test_tiny_captioning_model(golden, name):
    g = golden(name)
    cfg = syn.cfg_tiny()
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    model = _build(cfg, V, bool(use_glove), g.sub("sd/"))
    pred, loss, masks = _run_cap(model, cfg, {"rgb": g["rgb"], "flow": g["flow"], "audio": g["audio"]}, g["captions"])
    for k in ("V_mask", "A_mask", "C_mask"):
        assert torch.equal(masks[k].cpu(), g[k]), k
    assert_close(pred, g["pred"], atol=LOGP_TOL, name="log-probs")
    assert_close(loss, g["loss"], atol=1e-3, rtol=1e-4, name="loss")
    loss.backward()
    _check_grads(model.named_parameters(), g.sub("grad/"))


@pytest.mark.parametrize("name,cfgfn", [("cfg0_cap.npz", syn.cfg_config0), ("mid_cap.npz", syn.cfg_config1)])
def test_seeded_captioning_model(golden, name, cfgfn):
    """configs[0] and the mid-scale config[1]-width fixture: weights re-created from the seed (digest pinned by the
    fixture), log-probs within 1e-3 of the reference CPU path."""
    from oracle import bmt_oracle as orc
    g = golden(name)
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = cfgfn()
    model = _build(cfg, V, bool(use_glove))
    assert orc.state_dict_digest({k: v.cpu() for k, v in model.state_dict().items()}) == str(g.np("sd_digest"))
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    pred, loss, masks = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"])
    for k in ("V_mask", "A_mask", "C_mask"):
        assert torch.equal(masks[k].cpu(), g[k]), k
    err = float((pred.detach().cpu() - g["pred"]).abs().max())
    print(f"\n{name}: max |dlogp| = {err:.3e} (bar {LOGP_TOL})")
    assert_close(pred, g["pred"], atol=LOGP_TOL, name="log-probs")
    assert_close(loss, g["loss"], atol=1e-3, rtol=1e-4, name="loss")
    loss.backward()
    names = [str(s) for s in g.np("grad_names")]
    norms = g.np("grad_norms")
    params = dict(model.named_parameters())
    bad, worst = [], 0.0
    nmax = float(max(norms))
    for n, ref_norm in zip(names, norms):
        mine = float(params[n].grad.double().norm())
        if ref_norm < 1e-4 * nmax:          # analytically zero (key-projection biases): bf16 cancellation noise only
            if mine > 2e-2 * nmax:
                bad.append(f"{n}: should be ~0, |grad|={mine:.4e}")
        else:
            worst = max(worst, abs(mine - ref_norm) / ref_norm)
            if abs(mine - ref_norm) > GRAD_NORM_TOL * ref_norm:
                bad.append(f"{n}: |grad|={mine:.4e} reference {ref_norm:.4e}")
    print(f"worst relative error of a gradient norm: {worst:.3e} (bar {GRAD_NORM_TOL})")
    assert not bad, "\n".join(bad)
    _check_grads(model.named_parameters(), g.sub("grad/"))


def test_forward_operand_policies_on_the_mid_fixture(golden):
    """documents the per-site operand policy (bmt_amd.ops.POLICIES, tests/study_precision_policy.py): max |d log-prob| vs the
    reference on the mid-scale fixture when EVERY forward product runs one format, next to the shipped per-site table.
    Single-pass bf16 misses the 1e-3 bar by an order of magnitude, single-pass fp16 everywhere is marginal, the table passes with
    margin at 2/3 (encoder GEMMs) and 1/3 (attention cores) of the split-bf16 passes."""
    from bmt_amd import ops
    g = golden("mid_cap.npz")
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = syn.cfg_config1()
    model = _build(cfg, V, bool(use_glove))
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    errs = {}
    for name, fwd in (("bf16", ops.PREC_BF16), ("fp16", ops.PREC_F16), ("fp16w2", ops.PREC_F16W2), ("bf16x3", ops.PREC_BF16X3), ("policy", None)):
        try:
            ops.set_precision(fwd=fwd)
            with torch.no_grad():
                pred, _, _ = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"])
        finally:
            ops.set_precision()
        errs[name] = float((pred.cpu() - g["pred"]).abs().max())
    print("\nmax |dlogp| by forward operand format:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert 2e-3 < errs["bf16"] < 0.2
    assert errs["bf16x3"] < 1e-4
    assert errs["policy"] < 5e-4 and errs["policy"] < errs["fp16"] and errs["fp16"] < errs["bf16"]


def test_training_mode_dropout(golden):
    from bmt_amd import ops
    g = golden("tiny_cap.npz")
    cfg = syn.cfg_tiny(dout_p=0.1)
    V = int(g.np("meta")[0])
    model = _build(cfg, V, True, g.sub("sd/"))
    fs, caps = {"rgb": g["rgb"], "flow": g["flow"], "audio": g["audio"]}, g["captions"]
    ops.manual_seed(7)
    p1, l1, _ = _run_cap(model, cfg, fs, caps, train=True)
    l1.backward()
    assert torch.isfinite(l1) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert not torch.allclose(p1.detach().cpu(), g["pred"], atol=1e-3)   # dropout really is on
    ops.manual_seed(7)                                                     # same seed & step -> same masks
    model.zero_grad()
    p2, l2, _ = _run_cap(model, cfg, fs, caps, train=True)
    l2.backward()
    assert torch.equal(p1, p2)
    p3, _, _ = _run_cap(model, cfg, fs, caps, train=True)                  # next step -> new masks
    assert not torch.equal(p1, p3)
    # eval() switches every site off
    p4, _, _ = _run_cap(model, cfg, fs, caps, train=False)
    assert_close(p4, g["pred"], atol=LOGP_TOL, name="eval after train")


def test_train_step_matches_oracle(golden):
    """zero_grad -> masks -> forward -> KL/n_tokens -> backward -> clip -> Adam  (training_loop,
    epoch_loops/captioning_epoch_loops.py:128-141) for two steps, against the CPU oracle."""
    from bmt_amd.optim import FusedAdam, clip_grad_norm_
    from oracle import bmt_oracle as orc
    g = golden("tiny_cap_trainemb.npz")
    cfg = syn.cfg_tiny(dout_p=0.0)
    V = int(g.np("meta")[0])
    model = _build(cfg, V, False, g.sub("sd/"))
    opt = FusedAdam(model.parameters(), lr=1e-3)
    fs, caps = {"rgb": g["rgb"], "flow": g["flow"], "audio": g["audio"]}, g["captions"]
    p = {k: v.clone().requires_grad_() for k, v in g.sub("sd/").items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in (1, 2):
        opt.zero_grad()
        _, loss, _ = _run_cap(model, cfg, fs, caps, train=True)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        for t in p.values():
            t.grad = None
        oloss, _, _ = orc.train_cap_loss(p, cfg, fs, caps, 1, cfg.smoothing)
        oloss.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
        with torch.no_grad():
            for k in p:
                orc.adam_step(p[k], p[k].grad, m[k], v2[k], step, 1e-3)
        assert abs(float(loss.detach()) - float(oloss.detach())) < 2e-3, (step, float(loss.detach()), float(oloss.detach()))
    sd = model.state_dict()
    agree, total = 0, 0
    for k in p:
        d = (sd[k].cpu() - p[k].detach()).abs()
        agree += int((d < 2.5e-4).sum())
        total += d.numel()
    assert agree / total > 0.97, agree / total


def test_graph_replay_matches_eager(golden):
    """the step captured into two hipGraphs (fwd+bwd, optimizer) produces the same training trajectory as eager launches"""
    from bmt_amd import ops
    from bmt_amd.train import CaptioningTrainStep
    g = golden("tiny_cap_trainemb.npz")
    V = int(g.np("meta")[0])
    fs = {k: g[k].to(DEV) for k in ("rgb", "flow", "audio")}
    caps = g["captions"].to(DEV)
    results = []
    for use_graph in (False, True):
        ops._site_counter[0] = 10_000       # both arms must draw the same dropout masks: same call-site ids
        cfg = syn.cfg_tiny(dout_p=0.1, lr=1e-3)
        model = _build(cfg, V, False, g.sub("sd/"))
        step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
        ops.manual_seed(11)
        losses = []
        if use_graph:
            # capture() itself runs warm-up steps: rebuild so both arms start from the same weights / rng
            step.capture(fs, caps, warmup=1)
            model.load_state_dict(g.sub("sd/"))
            for st in step.optimizer.state.values():
                st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
            for t in step.optimizer._steps.values():
                t.zero_()
            ops.manual_seed(11)
            for _ in range(3):
                loss, _ = step.replay(fs, caps)
                losses.append(float(loss))
        else:
            for _ in range(3):
                loss, _ = step(fs, caps)
                losses.append(float(loss))
        results.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (l0, sd0), (l1, sd1) = results
    assert all(abs(a - b) < 2e-3 for a, b in zip(l0, l1)), (l0, l1)
    assert l0[0] != l0[1]      # the model does move
    # Adam at lr 1e-3 moves every weight by ~lr per step: a gradient whose sign is decided by atomic-add order may go either
    # way, so compare in bulk rather than by the worst element
    agree = sum(int(((sd0[k] - sd1[k]).abs() < 2e-4).sum()) for k in sd0)
    total = sum(sd0[k].numel() for k in sd0)
    assert agree / total > 0.98, (agree / total, l0, l1)


def test_full_size_properties_config1():
    """BASELINE configs[1] at full size (B=32, N=2, d_model=1024, H=4, T_v=256, T_a=800, T_c=30, V=10000), where the CPU oracle
    is too slow to be the checker: size-independent properties of the path instead.
      * rows of exp(log-probs) sum to 1, nothing is NaN/inf, masks equal the definition;
      * eval forward is deterministic (bit-identical twice) and independent of batch composition (a sample's log-probs do not
        change when the other 31 samples do) -- catches any cross-sample leak in the tiled kernels at full tile counts;
      * three optimizer steps on one batch reduce the loss."""
    from bmt_amd import ops
    from bmt_amd.train import CaptioningTrainStep, make_masks
    V, B, Tv, Ta, Tc = 10000, 32, 256, 800, 30
    cfg = syn.cfg_config1(dout_p=0.1, lr=1e-4)
    cfg.device = DEV
    torch.manual_seed(0)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        from bmt_amd.model.captioning_module import BiModalTransformer
        model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(DEV)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=7)
    fs = {k: v.to(DEV) for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"].to(DEV)
    x = caps[:, :-1]
    masks = make_masks(fs, x, "audio_video", syn.PAD_IDX)
    assert torch.equal(masks["V_mask"], (fs["rgb"][:, :, 0] != syn.PAD_IDX).unsqueeze(1))
    assert torch.equal(masks["A_mask"], (fs["audio"][:, :, 0] != syn.PAD_IDX).unsqueeze(1))
    model.eval()
    with torch.no_grad():
        p1 = model(fs, x, masks)
        p2 = model(fs, x, masks)
    Tx = x.shape[1]
    assert p1.shape == (B, Tx, V) and torch.isfinite(p1).all()
    assert torch.equal(p1, p2), "eval forward is not deterministic"
    assert_close(p1.exp().sum(-1), torch.ones(B, Tx), atol=1e-4, name="sum of probabilities")
    # batch independence: keep sample 5, replace everything else
    other = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=8)
    fs2 = {k: v.to(DEV).clone() for k, v in other["feature_stacks"].items()}
    caps2 = other["captions"].to(DEV).clone()
    for k in fs2:
        fs2[k][5] = fs[k][5]
    caps2[5] = caps[5]
    x2 = caps2[:, :-1]
    with torch.no_grad():
        p3 = model(fs2, x2, make_masks(fs2, x2, "audio_video", syn.PAD_IDX))
    assert_close(p3[5], p1[5], atol=2e-5, name="log-probs of a sample under a different batch")
    # training makes progress on a fixed batch
    ops.manual_seed(3)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
    losses = [float(step(fs, caps)[0]) for _ in range(3)]
    assert all(math.isfinite(l) for l in losses) and losses[2] < losses[0], losses


# ---------------------------------------------------------------- greedy decoding
def _decode_model(cfg, V, wscale):
    model = _build(cfg, V, True)
    with torch.no_grad():
        model.generator.linear.weight.mul_(wscale)
    return model.eval()


@pytest.mark.parametrize("tag,cfgfn", [("tiny", syn.cfg_tiny), ("cfg0", syn.cfg_config0)])
@pytest.mark.parametrize("reuse", [True, False])
def test_greedy_decoder_matches_reference_tokens(golden, tag, cfgfn, reuse):
    """bmt_amd.decode.greedy_decoder (encoder run once, cross-attention K/V planes cached, generator on the last position)
    and the un-cached loop both reproduce, token for token, what the reference's greedy_decoder emitted on the same weights
    and features.  Index work: bit-exact.  (The smallest top-1/top-2 margin in the fixtures is 7.7e-2, the log-prob error of
    the HIP path is < 1e-4.)"""
    from bmt_amd.decode import greedy_decoder
    g = golden("greedy_decode.npz")
    V, B, Tv, Ta, max_len, seed = [int(x) for x in g.np(f"{tag}/meta")]
    cfg = cfgfn()
    model = _decode_model(cfg, V, float(g.np(f"{tag}/wscale")))
    fs = {k: v.to(DEV) for k, v in syn.make_cap_batch(cfg, B, Tv, Ta, 4, V, seed=seed)["feature_stacks"].items()}
    trg = greedy_decoder(model, fs, max_len, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, "audio_video", reuse=reuse)
    assert trg.dtype == torch.long
    assert torch.equal(trg.cpu(), g[f"{tag}/tokens"])
    from bmt_amd import ops
    assert ops.context().kv_cache is None      # the cache does not outlive the call


def test_greedy_decoder_reuse_is_identical_to_full_forward_config1_shapes():
    """At config[1] sizes (d_model 1024, T_a 800, T_v 256, V 10000) the cached decode must pick the same tokens as one full
    forward pass per token, and its last-position log-probs must equal the full pass's (same kernels, same order)."""
    from bmt_amd import ops
    from bmt_amd.decode import greedy_decoder
    from bmt_amd.train import make_masks
    cfg = syn.cfg_config1()
    V, B, Tv, Ta = 10000, 4, 256, 800
    model = _decode_model(cfg, V, 6.0)
    fs = {k: v.to(DEV) for k, v in syn.make_cap_batch(cfg, B, Tv, Ta, 4, V, seed=99)["feature_stacks"].items()}
    a = greedy_decoder(model, fs, 12, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, "audio_video", reuse=True)
    b = greedy_decoder(model, fs, 12, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, "audio_video", reuse=False)
    assert torch.equal(a, b)
    with torch.no_grad():
        masks = make_masks(fs, a, "audio_video", syn.PAD_IDX)
        full = model(fs, a, masks)[:, -1]
        mem = model.encode(fs, masks)
        ops.context().kv_cache = {}
        try:
            C = model.decode(a, mem, masks)
            C2 = model.decode(a, mem, masks)          # second call: keys / values come from the cache
            assert len(ops.context().kv_cache) == 2 * cfg.N
        finally:
            ops.context().kv_cache = None
        last = model.generator(C[:, -1:])[:, 0]
    assert torch.equal(C, C2)
    assert_close(last, full, atol=1e-5, name="cached last-position log-probs")


# ---------------------------------------------------------------- fused residual block (ops.FUSE_RESIDUAL)
def test_fused_residual_block_equals_the_separate_kernels(golden):
    """ResidualConnection fused (LN writes planes, dropout + residual in the last GEMM's epilogue, dropout mask applied while the
    gradient is converted, residual gradient added by the LN backward) against the separate LN / planes / dropout_add / add
    kernels: same dropout masks, same roundings -> log-probs and gradients agree to fp32 round-off, in training mode."""
    from bmt_amd import ops
    g = golden("mid_cap.npz")
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = syn.cfg_config1()
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    res = {}
    first_site = ops._site_counter[0]
    for fused in (True, False):
        ops.FUSE_RESIDUAL = fused
        ops.RAW_MEMORY = False                       # (the separate kernels run on padded rows, where the decoder projects keys and values: the same in both arms)
        try:
            ops._site_counter[0] = first_site        # both models draw the same dropout masks (site ids are per module instance)
            model = _build(cfg, V, bool(use_glove))
            ops.manual_seed(11)
            pred, loss, _ = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"], train=True)
            loss.backward()
            res[fused] = (pred.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            ops.FUSE_RESIDUAL = True
            ops.RAW_MEMORY = True
    assert_close(res[True][0], res[False][0], atol=2e-5, name="log-probs fused vs separate")
    # gradients: the backward products take bf16 operands, so an fp32 1-ulp difference upstream (fma contraction in the fused
    # epilogue) flips bf16 roundings: tensors agree to bf16 noise, not bit for bit.  A wrong mask or a lost residual gradient
    # would show as O(1) errors on every tensor upstream of it.
    # (key-projection biases have an analytically zero gradient -- softmax ignores a constant added to every key -- so what is
    # compared there is cancellation noise: left out)
    errs = sorted(((rel_err(res[True][1][k], gsep), k) for k, gsep in res[False][1].items() if not k.endswith("linear_K2d.bias")),
                  reverse=True)
    print("\nworst fused-vs-separate gradient differences:", [(f"{e:.2e}", k) for e, k in errs[:5]])
    a = torch.cat([res[True][1][k].flatten() for _, k in errs])
    b = torch.cat([res[False][1][k].flatten() for _, k in errs])
    assert rel_err(a, b) < 1e-2, rel_err(a, b)
    assert errs[0][0] < 2e-2, errs[:5]


# ---------------------------------------------------------------- round-2 parity cases (tests/golden/make_golden_r2.py)
def test_full_length_captioning_model(golden):
    """configs[1] at its TRUE lengths (T_v=256, T_a=800, T_c=30, V=10000; 13 key tiles / 7 query tiles in the attention
    kernels), B=2: log-probs within 1e-3 of the reference CPU path, gradient norms within 2 %"""
    from oracle import bmt_oracle as orc
    from tests.test_oracle_golden import check_full_cap_pred
    g = golden("full_cap.npz")
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = syn.cfg_config1()
    model = _build(cfg, V, True)
    assert orc.state_dict_digest({k: v.cpu() for k, v in model.state_dict().items()}) == str(g.np("sd_digest"))
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    pred, loss, masks = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"])
    for k in ("V_mask", "A_mask", "C_mask"):
        assert torch.equal(masks[k].cpu(), g[k]), k
    err = float((pred.detach().cpu()[:, :, ::8] - g["pred_sub"]).abs().max())
    print(f"\nfull_cap: max |dlogp| (every 8th column) = {err:.3e} (bar {LOGP_TOL})")
    check_full_cap_pred(pred.detach().cpu(), g, atol=LOGP_TOL)
    assert_close(loss, g["loss"], atol=1e-3, rtol=1e-4, name="loss")
    loss.backward()
    params = dict(model.named_parameters())
    norms = g.np("grad_norms")
    nmax, bad, worst = float(max(norms)), [], 0.0
    for n, ref_norm in zip([str(s) for s in g.np("grad_names")], norms):
        mine = float(params[n].grad.double().norm())
        if ref_norm < 1e-4 * nmax:
            if mine > 2e-2 * nmax:
                bad.append(f"{n}: should be ~0, |grad|={mine:.4e}")
        else:
            worst = max(worst, abs(mine - ref_norm) / ref_norm)
            if abs(mine - ref_norm) > GRAD_NORM_TOL * ref_norm:
                bad.append(f"{n}: |grad|={mine:.4e} reference {ref_norm:.4e}")
    print(f"worst relative error of a gradient norm: {worst:.3e} (bar {GRAD_NORM_TOL})")
    assert not bad, "\n".join(bad)
    _check_grads(model.named_parameters(), g.sub("grad/"))


def test_deep_config_captioning_model(golden):
    """configs[4]-shaped model (N=6, H=8 -> d_k=128, d_model=1024) at reduced B / T / V against the reference"""
    from oracle import bmt_oracle as orc
    from tests.test_oracle_golden import deep_cfg
    g = golden("deep_cap.npz")
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = deep_cfg()
    model = _build(cfg, V, True)
    assert orc.state_dict_digest({k: v.cpu() for k, v in model.state_dict().items()}) == str(g.np("sd_digest"))
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    pred, loss, masks = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"])
    err = float((pred.detach().cpu() - g["pred"]).abs().max())
    print(f"\ndeep_cap: max |dlogp| = {err:.3e} (bar {LOGP_TOL})")
    assert_close(pred, g["pred"], atol=LOGP_TOL, name="log-probs")
    assert_close(loss, g["loss"], atol=1e-3, rtol=1e-4, name="loss")
    loss.backward()
    _check_grads(model.named_parameters(), g.sub("grad/"))


def test_linear_embedder_captioning_model(golden):
    """use_linear_embedder=True (FeatureEmbedder, model/blocks.py:66-81; model/captioning_module.py:113-120)"""
    g = golden("tiny_cap_linemb.npz")
    cfg = syn.cfg_tiny(use_linear_embedder=True)
    V = int(g.np("meta")[0])
    model = _build(cfg, V, True, g.sub("sd/"))
    pred, loss, masks = _run_cap(model, cfg, {"rgb": g["rgb"], "flow": g["flow"], "audio": g["audio"]}, g["captions"])
    assert_close(pred, g["pred"], atol=LOGP_TOL, name="log-probs")
    assert_close(loss, g["loss"], atol=1e-3, rtol=1e-4, name="loss")
    loss.backward()
    _check_grads(model.named_parameters(), g.sub("grad/"))


def test_ten_adam_steps_against_the_oracle(golden):
    """10 real optimizer steps (bf16 backward, fused Adam) on the mid-scale fixture next to the fp32 CPU path trained the same way.
    Adam's update is lr * m / sqrt(v): early on every weight moves by ~lr whatever the size of its gradient, so the TRAJECTORY is
    chaotic in the gradient's low bits -- tests/study_adam_drift.py: fp32 gradients with 1 % relative noise drift by 6e-2 in the
    log-probs after 10 steps, noise of 0.1 % of a tensor's rms by 0.36 -- and no reduced-precision backward can track the fp32
    run to 1e-3.  What is checked: (a) the loss stays within 3e-3 of the oracle's over the first five steps and within 1.5e-2 at every step, and
    goes down; (b) the FORWARD
    is still within the 1e-3 bar at the trained weights: the oracle evaluated at the weights the HIP path arrived at."""
    from bmt_amd.train import CaptioningTrainStep
    from oracle import bmt_oracle as orc
    g = golden("mid_cap.npz")
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = syn.cfg_config1(dout_p=0.0)
    model = _build(cfg, V, True)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    fs = {k: v.to(DEV) for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"].to(DEV)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
    sd = orc.init_captioning_params(cfg, V, seed=0, glove=syn.make_glove(V, cfg.d_model_caps))
    p = {k: v.clone().requires_grad_(k != "emb_C.embedder.weight") for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    mine, theirs = [], []
    for it in range(1, 11):
        loss, _ = step(fs, caps)
        for t in p.values():
            t.grad = None
        oloss, _, _ = orc.train_cap_loss(p, cfg, batch["feature_stacks"], batch["captions"], syn.PAD_IDX, cfg.smoothing)
        oloss.backward()
        with torch.no_grad():
            for k, t in p.items():
                if t.grad is not None:
                    orc.adam_step(t, t.grad, m[k], v2[k], it, cfg.lr)
        mine.append(float(loss))
        theirs.append(float(oloss.detach()))
    print("\nloss, HIP path :", [f"{v:.4f}" for v in mine], "\nloss, oracle   :", [f"{v:.4f}" for v in theirs])
    # the two trajectories separate as the docstring says: measured on one box, three runs each (weight-gradient atomics make runs differ),
    # the loss gap at steps 1-5 / 6-10 is <= 1.6e-3 / 4.6e-3 ... 6.9e-3 with the decoder's cross-attentions against the raw memories and
    # <= 1.6e-3 / 4.6e-3 ... 6.3e-3 with projected keys and values (round 5; the 5e-3 of round 4 sat inside that spread): early steps tight,
    # late steps bounded by twice what is seen
    gaps = [abs(a - b) for a, b in zip(mine, theirs)]
    assert max(gaps[:5]) < 3e-3 and max(gaps) < 1.5e-2, gaps
    assert mine[-1] < mine[0] - 0.05 and theirs[-1] < theirs[0] - 0.05
    pred, _, _ = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"])
    trained = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        _, opred, _ = orc.train_cap_loss(trained, cfg, batch["feature_stacks"], batch["captions"], syn.PAD_IDX, cfg.smoothing)
    err = float((pred.detach().cpu() - opred).abs().max())
    print(f"after 10 Adam steps, same weights: max |dlogp| = {err:.3e} (bar {LOGP_TOL})")
    assert err < LOGP_TOL


def test_optimizer_state_and_weight_planes_after_graph_replays(golden):
    """(a) the optimizer's state_dict carries the DEVICE step count after graph replays (resuming from it continues
    the eager trajectory); (b) an eager forward after replays sees the replayed optimizer's weights (plane cache refreshed)"""
    from bmt_amd import ops
    from bmt_amd.optim import FusedAdam
    from bmt_amd.train import CaptioningTrainStep
    g = golden("tiny_cap_trainemb.npz")
    V = int(g.np("meta")[0])
    fs = {k: g[k].to(DEV) for k in ("rgb", "flow", "audio")}
    caps = g["captions"].to(DEV)
    cfg = syn.cfg_tiny(dout_p=0.0, lr=1e-3)
    model = _build(cfg, V, False, g.sub("sd/"))
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
    step.capture(fs, caps, warmup=1)                      # 1 eager warm-up step
    for _ in range(4):
        step.replay()
    osd = step.optimizer.state_dict()
    steps = {float(st["step"]) for st in osd["state"].values()}
    assert steps == {5.0}, steps                           # 1 warm-up + 4 replays
    # (b): eval now, one more replay, eval again: the second eval must see the new weights
    with torch.no_grad():
        p1, _, _ = _run_cap(model, cfg, {k: g[k] for k in ("rgb", "flow", "audio")}, g["captions"])
    step.replay()
    with torch.no_grad():
        p2, _, _ = _run_cap(model, cfg, {k: g[k] for k in ("rgb", "flow", "audio")}, g["captions"])
    ops.weights_changed()                                  # force a refresh: must not change anything if the cache was fresh
    with torch.no_grad():
        p3, _, _ = _run_cap(model, cfg, {k: g[k] for k in ("rgb", "flow", "audio")}, g["captions"])
    assert not torch.equal(p1, p2) and torch.equal(p2, p3)
    # (a): resume an eager optimizer from the state_dict: its next step must use bias corrections of step 7
    opt2 = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    opt2.load_state_dict(step.optimizer.state_dict())
    opt2.zero_grad()
    _, loss, _ = _run_cap(model, cfg, {k: g[k] for k in ("rgb", "flow", "audio")}, g["captions"], train=True)
    loss.backward()
    opt2.step()
    assert int(opt2._steps[0].item()) == 7


def test_two_models_step_concurrently_on_two_streams(golden):
    """the per-pass state of the Python layer (ops.StepContext: queued weight-gradient products, the residual offered to a sublayer's
    last GEMM, LayerNorm's operand planes) is keyed by (device, stream), the split-K / descriptor scratch by stream, the weight-plane
    registry is locked: two different models running forward + backward from two threads on two streams give the gradients of the same
    passes run one after the other (up to the order of the fp32 atomics of the bias-gradient column sums: 1e-6 of the tensor's norm --
    state leaking between the passes shows up as errors of order one)."""
    import threading
    from bmt_amd import ops
    jobs = []
    for name, scale in (("tiny_cap.npz", 1.0), ("tiny_cap_trainemb.npz", 0.5)):
        g = golden(name)
        cfg = syn.cfg_tiny()
        V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
        model = _build(cfg, V, bool(use_glove), g.sub("sd/"))
        fs = {"rgb": g["rgb"] * scale, "flow": g["flow"], "audio": g["audio"]}
        jobs.append((model, cfg, fs, g["captions"], torch.cuda.Stream()))

    def run(job, rounds):
        model, cfg, fs, caps, stream = job
        out = []
        with torch.cuda.stream(stream):
            ctx = ops.context()
            for _ in range(rounds):
                model.zero_grad(set_to_none=True)
                ctx.defer_dw = True              # the weight-gradient products of the pass are queued and issued as one grouped launch
                _, loss, _ = _run_cap(model, cfg, fs, caps)
                loss.backward()
                ops.flush_dw()
                ctx.defer_dw = False
                out.append((float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
        stream.synchronize()
        return out

    serial = [run(j, 3) for j in jobs]           # also the warm-up: weight planes, pointer tables, allocator pools
    results = [None, None]
    errors = []

    def worker(i):
        try:
            results[i] = run(jobs[i], 3)
        except Exception as e:          # noqa: BLE001  (reported by the assertion below)
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for (l0, g0), (l1, g1) in zip(serial[i], results[i]):
            assert l0 == l1, (i, l0, l1)
            assert g0.keys() == g1.keys()
            for k in g0:
                d, n = float((g0[k] - g1[k]).double().norm()), float(g0[k].double().norm())
                assert d <= 1e-6 * n + 1e-12, f"model {i}: gradient of {k} differs between the serial and the concurrent pass ({d:.3e} of {n:.3e})"


def test_full_size_properties_config4_slice():
    """BASELINE configs[4]'s per-GPU slice at FULL size: train_cap with N = 6, H = 8 (d_k = 128), d_model 1024, B = 64, T_v = 256, T_a = 800,
    V = 10000 (the N = 6 / H = 8 SHAPE is checked against the reference at reduced B / T by the deep_cap fixture; the CPU oracle is far too
    slow here).  Size-independent properties: log-probabilities normalise, the eval forward is deterministic, a sample's outputs do not
    depend on the rest of the batch nor on what its padded rows hold besides the pad marker, every trainable parameter gets a finite
    gradient, and the loss decreases over optimizer steps on one batch -- eager and through the captured graphs."""
    import contextlib, io
    from bmt_amd import ops
    from bmt_amd.model.captioning_module import BiModalTransformer
    from bmt_amd.train import CaptioningTrainStep, make_masks
    V, B, Tv, Ta, Tc = 10000, 64, 256, 800, 30
    cfg = syn.make_cfg(d_model=1024, H=8, N=6, dout_p=0.1)
    cfg.device = DEV
    cfg.lr = 1e-4
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(DEV)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=77)
    fs = {k: v.to(DEV) for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"].to(DEV)
    x = caps[:, :-1]
    masks = make_masks(fs, x, "audio_video", syn.PAD_IDX)
    model.eval()
    with torch.no_grad():
        p1 = model(fs, x, masks)
        p2 = model(fs, x, masks)
    assert p1.shape == (B, Tc, V) and torch.isfinite(p1).all()
    assert torch.equal(p1, p2), "eval forward is not deterministic"
    assert float(torch.logsumexp(p1.double(), -1).abs().max()) < 1e-4, "rows of log-probabilities do not normalise"
    # sample 5 inside another batch, its padded feature rows filled with different values (channel 0 keeps the pad marker)
    other = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=78)
    fs2 = {k: v.to(DEV).clone() for k, v in other["feature_stacks"].items()}
    caps2 = other["captions"].to(DEV).clone()
    for k in fs2:
        fs2[k][5] = fs[k][5]
    caps2[5] = caps[5]
    pad_v = fs2["rgb"][5, :, 0] == float(syn.PAD_IDX)
    pad_a = fs2["audio"][5, :, 0] == float(syn.PAD_IDX)
    fs2["rgb"][5, pad_v, 1:] = 7.0
    fs2["flow"][5, pad_v, :] = -3.0
    fs2["audio"][5, pad_a, 1:] = 5.0
    x2 = caps2[:, :-1]
    with torch.no_grad():
        p3 = model(fs2, x2, make_masks(fs2, x2, "audio_video", syn.PAD_IDX))
    assert_close(p3[5], p1[5], atol=LOGP_TOL, rtol=0, name="log-probs of a sample under a different batch and different padding")
    ops.manual_seed(9)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
    l0 = float(step(fs, caps)[0])
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)
    step.capture(fs, caps)
    losses = [l0] + [float(step.replay()[0]) for _ in range(3)]
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert torch.cuda.max_memory_allocated() < 40 * 2 ** 30


def test_two_compute_streams_change_nothing_but_the_schedule(golden, monkeypatch):
    """the encoder's audio / video chains and a decoder layer's two memory attentions on two streams (ops.fork_side_stream) against the same
    pass on one stream: same log-probs, same gradients up to the order of the fp32 atomics of the bias column sums"""
    from bmt_amd import ops
    g = golden("mid_cap.npz")
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in g.np("meta")]
    cfg = syn.cfg_config1()
    model = _build(cfg, V, bool(use_glove))
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    out = {}
    for streams in (1, 2, 3, 1):           # (3, the default: + the first decoder layer's self-attention sublayer beside the encoder)
        monkeypatch.setattr(ops, "ENC_STREAMS", streams)
        model.zero_grad(set_to_none=True)
        pred, loss, _ = _run_cap(model, cfg, batch["feature_stacks"], batch["captions"])
        loss.backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
        out.setdefault(streams, []).append((pred.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (p1, g1), (p1b, g1b) = out[1]
    assert torch.equal(p1, p1b), "the one-stream pass is not reproducible: the comparison below would be meaningless"
    for n in (2, 3):
        p2, g2 = out[n][0]
        assert torch.equal(p1, p2), (n, float((p1 - p2).abs().max()))
        for k in g1:
            noise = float((g1[k] - g1b[k]).double().norm())          # run-to-run (atomics) on one stream
            d = float((g1[k] - g2[k]).double().norm())
            assert d <= 10 * noise + 1e-6 * float(g1[k].double().norm()) + 1e-12, (n, k, d, noise)


def test_layernorm_output_as_planes_only_refuses_an_fp32_reader():
    """ResidualConnection(..., fp32_out=False): the normalised tensor's fp32 values are never written; a consumer that cannot use the
    planes it carries must fail loudly instead of reading them"""
    from bmt_amd import ops
    x = torch.randn(4, 8, 128, device=DEV)
    gamma, beta = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    _, xn = ops.residual_norm(x, gamma, beta, 1e-5, ops.PREC_F16W2, fp32_out=False)
    assert ops.planes_of(xn, ops.act_fmt(ops.PREC_F16W2)) is not None
    ref = torch.nn.functional.layer_norm(x, (128,))
    pl = ops.planes_of(xn, "f16")
    assert float((pl.fh[:, :128].float().view(4, 8, 128) - ref).abs().max()) < 2e-3
    with pytest.raises(RuntimeError, match="planes only"):
        ops._need_fp32(xn)
    _, xn32 = ops.residual_norm(x, gamma, beta, 1e-5, ops.PREC_F16W2, fp32_out=True)
    assert float((xn32 - ref).abs().max()) < 1e-5
    ops._need_fp32(xn32)
