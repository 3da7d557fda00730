"""CPU, this container only: the oracle (oracle/bmt_oracle.py) against the REFERENCE itself, imported from /root/reference --
fresh random weights and inputs on every parametrisation, not only the committed golden vectors.  Skipped where the
reference is absent (the GPU box): there the committed fixtures, which these same comparisons produced, carry the pin.

Covers SURVEY.md section 8 rows: (a) captioning model forward + loss + gradients and the proposal generator, (f1) greedy
decoding, (f2) proposal post-processing, (f3) feature loading."""
import importlib
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _refimport  # noqa: E402
from bmt_amd import synthetic as syn  # noqa: E402
from oracle import bmt_oracle as orc  # noqa: E402

pytestmark = pytest.mark.skipif(not _refimport.reference_available(), reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref():
    return _refimport.import_reference()


@pytest.mark.parametrize("seed,use_glove", [(0, True), (3, False), (7, True)])
def test_captioning_forward_loss_and_gradients(ref, seed, use_glove):
    cfg = syn.cfg_tiny()
    cfg.device = "cpu"
    V, B, Tv, Ta, Tc = 13, 3, 8, 17, 6
    glove = syn.make_glove(V, cfg.d_model_caps) if use_glove else None
    torch.manual_seed(seed)
    model = ref.captioning_module.BiModalTransformer(cfg, syn.FakeTrainDataset(V, glove)).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=100 + seed)
    fs, caps = batch["feature_stacks"], batch["captions"]
    x, y = caps[:, :-1], caps[:, 1:]
    masks = ref.cap_loops.make_masks(fs, x, "audio_video", syn.PAD_IDX)
    pred = model(fs, x, masks)
    loss = ref.label_smoothing.LabelSmoothing(cfg.smoothing, syn.PAD_IDX)(pred, y) / (y != syn.PAD_IDX).sum()
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point() and dict(model.named_parameters())[k].requires_grad) for k, v in sd.items()}
    oloss, opred, n_tok = orc.train_cap_loss(p, cfg, fs, caps, syn.PAD_IDX, cfg.smoothing)
    omasks = orc.make_masks(fs, x, syn.PAD_IDX)
    for k in ("V_mask", "A_mask", "C_mask"):
        assert torch.equal(omasks[k], masks[k]), k
    assert int(n_tok) == int((y != syn.PAD_IDX).sum())
    assert torch.allclose(opred, pred, atol=2e-5), float((opred - pred).abs().max())
    assert torch.allclose(oloss, loss, atol=1e-5, rtol=1e-5)
    oloss.backward()
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            assert torch.allclose(p[k].grad, prm.grad, atol=3e-5, rtol=1e-4), (k, float((p[k].grad - prm.grad).abs().max()))


@pytest.mark.parametrize("seed", [1, 5])
def test_greedy_decoder(ref, seed):
    cfg = syn.cfg_tiny()
    cfg.device = "cpu"
    V = 12
    torch.manual_seed(seed)
    model = ref.captioning_module.BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).eval()
    with torch.no_grad():
        model.generator.linear.weight.mul_(6.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    fs = syn.make_cap_batch(cfg, 3, 7, 11, 4, V, seed=200 + seed)["feature_stacks"]
    want = ref.cap_loops.greedy_decoder(model, fs, 9, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, "audio_video")
    got, margins = orc.greedy_decode(sd, cfg, fs, 9, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, return_margins=True)
    if float(margins.min()) > 1e-4:          # a decision this close is not a fair bit-exactness target
        assert torch.equal(got, want)
    else:
        assert got.shape[0] == want.shape[0]


def test_proposal_generator_forward_and_loss(ref):
    cfg = syn.cfg_tiny(procedure="train_prop")
    cfg.device = "cpu"
    cfg.anchors_num_audio, cfg.anchors_num_video = 3, 5
    cfg.conv_layers_audio, cfg.conv_layers_video = [16, 16], [16, 16]
    cfg.kernel_sizes = {"audio": [1, 5], "video": [3, 7]}
    anchors = {"audio": [1.5, 6.0, 20.0], "video": [1.0, 3.0, 8.0, 20.0, 60.0]}
    torch.manual_seed(2)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.proposal_generator.MultimodalProposalGenerator(cfg, anchors).eval()
    batch = syn.make_prop_batch(cfg, 2, 10, 15, seed=9, events_per_video=3)
    fs = batch["feature_stacks"]
    masks = ref.cap_loops.make_masks(fs, None, "audio_video", 1)
    preds, loss, la, lv = model(fs, batch["targets"], masks)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = orc.multimodal_proposal_generator(sd, cfg, anchors, fs, batch["targets"], masks)
    opreds, oloss = out[0], out[1]
    assert torch.allclose(opreds, preds, atol=2e-4, rtol=1e-4), float((opreds - preds).abs().max())
    assert torch.allclose(oloss, loss, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("seed,ties", [(21, False), (22, True)])
def test_proposal_postprocessing(ref, seed, ties):
    from tests.postprocess_util import make_preds
    pu = ref.proposal_utils
    preds, dur = make_preds(2, 3000, seed, ties)
    cfg = type("Cfg", (), {"max_prop_per_vid": 50})()
    want = pu.postprocess_preds(preds.clone(), cfg, {"duration_in_secs": dur})
    got = orc.postprocess_preds(preds, 50, dur)
    if ties:      # the reference's unstable argsort leaves the order inside a tie group open
        assert torch.equal(got[:, :, 2], want[:, :, 2])
    else:
        assert torch.equal(got, want)
    for b in range(2):
        assert torch.equal(orc.non_max_suppression(want[b], 0.5), pu.non_max_suppresion(want[b].clone(), 0.5))
    assert torch.equal(orc.tiou_start_end(want[0, 0], want[0, 1:]),
                       pu.tiou_vectorized(want[0, :1], want[0, 1:], center_length=False).reshape(-1))


def test_feature_loading(ref, tmp_path):
    from tests.ingest_util import items, write_features
    from bmt_amd import ingest
    lf = importlib.import_module("datasets.load_features")
    cfg = write_features(str(tmp_path))
    names = ["i3d_features", "vggish_features"]
    for vid, s, e, dur in items():
        want = lf.load_features_from_npy(cfg, names, vid, s, e, dur, 1, get_full_feat=False)
        got = ingest.load_features_from_npy(cfg, names, vid, s, e, dur, 1, get_full_feat=False)
        for k in ("rgb", "flow", "audio"):
            assert (want[k] is None) == (got[k] is None), (vid, k)
            if want[k] is not None:
                assert torch.equal(want[k], got[k]), (vid, s, e, k)
                x = torch.arange(want[k].shape[0] * 2 + 3).float().view(-1, 1)
                a, b = lf.crop_a_segment(x, s, e, dur), orc.crop_a_segment(x, s, e, dur)
                assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
