"""CPU: checkpoint interchange with the reference -- the dict layout of best_cap_model.pt /
best_prop_model.pt, the ``module.`` prefix in both directions, optimizer state in torch's layout, and the uni-modal
surfaces' state_dict (keys, shapes and bit-identical initial values) against tests/golden/unimodal.npz, which holds what the
REFERENCE wrote (tests/golden/make_golden_unimodal.py)."""
import os

import pytest
import torch

from bmt_amd import checkpoint as ckpt
from bmt_amd import synthetic as syn

V = 11
ANCHORS = {"audio": [1.5, 6.0, 20.0], "video": [1.0, 3.0, 8.0, 20.0, 60.0]}


def unimodal_cfg(**kw):
    """the widths of tests/golden/make_golden_unimodal.py: video d_model 256 (d_k 64), audio 128 (d_k 32)"""
    cfg = syn.make_cfg(d_model=128, H=4, N=1, d_aud=128, d_vid=256, d_model_caps=20, **kw)
    cfg.device = "cpu"
    return cfg


def _prop_cfg(modality, tiny=False):
    cfg = syn.cfg_tiny(procedure="train_prop", modality=modality) if tiny else unimodal_cfg(procedure="train_prop", modality=modality)
    cfg.anchors_num_audio, cfg.anchors_num_video = 3, 5
    cfg.conv_layers_audio, cfg.conv_layers_video = [16, 16], [16, 16]
    cfg.kernel_sizes = {"audio": [1, 5], "video": [3, 7]}
    cfg.device = "cpu"
    return cfg


def _same_state(sd, g, t):
    """keys, shapes and bits (sha256) of the reference's state_dict: same constructor order + same seed -> same weights"""
    from oracle import bmt_oracle as orc
    assert list(sd.keys()) == [str(k) for k in g.np(t + "sd_keys")]
    assert [",".join(str(d) for d in v.shape) for v in sd.values()] == [str(k) for k in g.np(t + "sd_shapes")]
    assert orc.state_dict_digest(sd) == str(g.np(t + "sd_digest"))


def _cap_unimodal(g, modality):
    from bmt_amd.model.captioning_module import Transformer
    cfg = unimodal_cfg(modality=modality)
    glove = syn.make_glove(V, int(g.np(f"cap_{modality}/glove_dim")))
    torch.manual_seed(0)
    return Transformer(syn.FakeTrainDataset(V, glove), cfg), cfg


@pytest.mark.parametrize("modality", ["video", "audio"])
def test_unimodal_transformer_state_dict_is_the_references(golden, modality):
    g = golden("unimodal.npz")
    model, _ = _cap_unimodal(g, modality)
    _same_state(model.state_dict(), g, f"cap_{modality}/")


@pytest.mark.parametrize("modality", ["video", "audio"])
def test_unimodal_proposal_generator_state_dict_is_the_references(golden, modality):
    from bmt_amd.model.proposal_generator import ProposalGenerator
    g = golden("unimodal.npz")
    torch.manual_seed(0)
    model = ProposalGenerator(_prop_cfg(modality), ANCHORS)
    _same_state(model.state_dict(), g, f"prop_{modality}/")


def test_captioning_checkpoint_layout(golden, tmp_path):
    from bmt_amd.model.captioning_module import BiModalTransformer
    g = golden("unimodal.npz")
    cfg = syn.cfg_tiny()
    cfg.device = "cpu"
    cfg.model_checkpoint_path = str(tmp_path / "cap")
    torch.manual_seed(0)
    model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps)))
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5)
    path = ckpt.save_cap_model(cfg, 3, model, opt, 1.5, 2.5, {"m": 1}, {"m": 2}, V)
    assert os.path.basename(path) == str(g.np("cpt_cap/file"))
    cpt = ckpt.load_checkpoint(path)
    # the reference's keys in its order, then ONE extra key its loaders ignore: the dropout stream's {seed, step}
    assert list(cpt.keys()) == [str(k) for k in g.np("cpt_cap/keys")] + ["bmt_dropout_state"]
    assert list(cpt["model_state_dict"].keys()) == [str(k) for k in g.np("cpt_cap/state_keys")]
    assert cpt["epoch"] == 3 and cpt["trg_voc_size"] == V and cpt["val_2_loss"] == 2.5
    # load back into a differently initialised model, through the prefixed and the bare layout
    torch.manual_seed(1)
    other = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps)))
    assert not torch.equal(other.generator.linear.weight, model.generator.linear.weight)
    assert ckpt.load_model_state(other, path)["epoch"] == 3
    for (k, a), (_, b) in zip(other.state_dict().items(), model.state_dict().items()):
        assert torch.equal(a, b), k
    assert ckpt.load_model_state(other, model.state_dict()) is None
    # the proposal generator picks the encoder out of this file, as the reference does (proposal_generator.py:234-246)
    from bmt_amd.model.proposal_generator import MultimodalProposalGenerator
    pcfg = _prop_cfg("audio_video", tiny=True)
    pcfg.pretrained_cap_model_path = path
    prop = MultimodalProposalGenerator(pcfg, ANCHORS)
    for (k, a), (_, b) in zip(prop.encoder.state_dict().items(), model.encoder.state_dict().items()):
        assert torch.equal(a, b), k
    assert all(not p.requires_grad for p in prop.encoder.parameters())


def test_proposal_checkpoint_layout(golden, tmp_path):
    from bmt_amd.model.proposal_generator import ProposalGenerator
    g = golden("unimodal.npz")
    cfg = _prop_cfg("audio")
    cfg.log_path = str(tmp_path / "prop")
    torch.manual_seed(0)
    model = ProposalGenerator(cfg, ANCHORS)
    opt = torch.optim.Adam(model.parameters(), lr=5e-5)
    path = ckpt.save_prop_model(cfg, 2, model, opt, None, {"f1": 0.5}, 0.5)
    assert os.path.basename(path) == str(g.np("cpt_prop/file"))
    cpt = ckpt.load_checkpoint(path)
    assert list(cpt.keys()) == [str(k) for k in g.np("cpt_prop/keys")] + ["bmt_dropout_state"]
    assert list(cpt["model_state_dict"].keys()) == [str(k) for k in g.np("cpt_prop/state_keys")]
    assert cpt["anchors"] == ANCHORS and cpt["scheduler_state_dict"] is None
    # a DataParallel-style state_dict (module.-prefixed) loads into a differently initialised un-wrapped model
    torch.manual_seed(5)
    other = ProposalGenerator(cfg, ANCHORS)
    ckpt.load_model_state(other, {"model_state_dict": cpt["model_state_dict"]})
    _same_state(other.state_dict(), g, "prop_audio/")


def test_prefix_helpers_are_idempotent():
    sd = {"a.b": torch.zeros(1), "module.c": torch.ones(1)}
    assert list(ckpt.with_prefix(sd)) == ["module.a.b", "module.c"]
    assert list(ckpt.without_prefix(ckpt.with_prefix(sd))) == ["a.b", "c"]


def test_fused_adam_accepts_torch_adam_state():
    """optimizer_state_dict of a reference checkpoint (torch.optim.Adam) loads into FusedAdam: same layout, step restored"""
    from bmt_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.randn(5))
    ref = torch.optim.Adam([p], lr=5e-5)
    p.grad = torch.randn(5)
    ref.step(); ref.step()
    q = torch.nn.Parameter(p.detach().clone())
    opt = FusedAdam([q], lr=5e-5)
    opt.load_state_dict(ref.state_dict())
    st = opt.state[q]
    assert float(st["step"]) == 2.0 and torch.equal(st["exp_avg"], ref.state[p]["exp_avg"])
    assert list(opt.state_dict()["state"][0].keys()) == list(ref.state_dict()["state"][0].keys())
