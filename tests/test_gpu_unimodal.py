"""-m gpu: the uni-modal surfaces (Transformer, ProposalGenerator) and checkpoint resume on the HIP path against what the
REFERENCE produced on the same weights and inputs (tests/golden/unimodal.npz)."""
import pytest
import torch

from bmt_amd import synthetic as syn
from tests.gpu_util import assert_close
from tests.test_gpu_model import LOGP_TOL, _check_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"
V = 11
ANCHORS = {"audio": [1.5, 6.0, 20.0], "video": [1.0, 3.0, 8.0, 20.0, 60.0]}


def _cfg(**kw):
    cfg = syn.make_cfg(d_model=128, H=4, N=1, d_aud=128, d_vid=256, d_model_caps=20, **kw)
    cfg.device = DEV
    return cfg


def _pinned(model, g, t):
    """weights re-created from the seed: digest pinned by the fixture (the reference's state_dict)"""
    from oracle import bmt_oracle as orc
    assert orc.state_dict_digest({k: v.cpu() for k, v in model.state_dict().items()}) == str(g.np(t + "sd_digest"))
    return model.to(DEV).eval()


def _grads(model, g, t):
    names, norms = [str(s) for s in g.np(t + "grad_names")], g.np(t + "grad_norms")
    params = dict(model.named_parameters())
    nmax = float(max(norms))
    bad = []
    for n, ref_norm in zip(names, norms):
        mine = float(params[n].grad.double().norm())
        if ref_norm < 1e-4 * nmax:
            if mine > 2e-2 * nmax:
                bad.append(f"{n}: should be ~0, |grad|={mine:.4e}")
        elif abs(mine - ref_norm) > 0.10 * ref_norm:
            bad.append(f"{n}: |grad|={mine:.4e} reference {ref_norm:.4e}")
    assert not bad, "\n".join(bad)
    _check_grads(model.named_parameters(), g.sub(t + "grad/"))


def _cap_model(g, modality):
    from bmt_amd.model.captioning_module import Transformer
    cfg = _cfg(modality=modality)
    glove = syn.make_glove(V, int(g.np(f"cap_{modality}/glove_dim")))
    torch.manual_seed(0)
    model = Transformer(syn.FakeTrainDataset(V, glove), cfg)
    return _pinned(model, g, f"cap_{modality}/"), cfg


@pytest.mark.parametrize("modality", ["video", "audio"])
def test_unimodal_transformer(golden, modality):
    from bmt_amd.decode import greedy_decoder
    from bmt_amd.loss.label_smoothing import LabelSmoothing
    from bmt_amd.train import make_masks
    g = golden("unimodal.npz")
    t = f"cap_{modality}/"
    model, cfg = _cap_model(g, modality)
    fs = {k: g[t + k].to(DEV) for k in ("rgb", "flow", "audio")}
    caps = g[t + "captions"].to(DEV)
    x, y = caps[:, :-1], caps[:, 1:]
    masks = make_masks(fs, x, modality, syn.PAD_IDX)
    for k, m in g.sub(t + "mask/").items():
        assert torch.equal(masks[k].cpu(), m), k
    pred = model(fs, x, masks)
    assert_close(pred, g[t + "pred"], atol=LOGP_TOL, name="log-probs")
    loss = LabelSmoothing(cfg.smoothing, syn.PAD_IDX)(pred, y) / (y != syn.PAD_IDX).sum()
    assert_close(loss, g[t + "loss"], atol=1e-3, rtol=1e-4, name="loss")
    loss.backward()
    _grads(model, g, t)
    # greedy decoding through the uni-modal branch: the reference's tokens on the same (generator x6) weights
    with torch.no_grad():
        model.generator.linear.weight.mul_(6.0)
    from bmt_amd import ops
    ops.weights_changed()
    trg = greedy_decoder(model, fs, 6, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, modality)
    assert torch.equal(trg.cpu(), g[t + "greedy_tokens_wscale6"])


@pytest.mark.parametrize("modality", ["video", "audio"])
def test_unimodal_proposal_generator(golden, modality):
    from bmt_amd.model.proposal_generator import ProposalGenerator
    from bmt_amd.train import make_masks
    g = golden("unimodal.npz")
    t = f"prop_{modality}/"
    cfg = _cfg(procedure="train_prop", modality=modality)
    cfg.anchors_num_audio, cfg.anchors_num_video = 3, 5
    cfg.conv_layers_audio, cfg.conv_layers_video = [16, 16], [16, 16]
    cfg.kernel_sizes = {"audio": [1, 5], "video": [3, 7]}
    torch.manual_seed(0)
    model = _pinned(ProposalGenerator(cfg, ANCHORS), g, t)
    fs = {k: g[t + k].to(DEV) for k in ("rgb", "flow", "audio")}
    masks = make_masks(fs, None, modality, 1)
    preds, loss, ld = model(fs, g[t + "targets"].to(DEV), masks)
    assert_close(preds, g[t + "preds"], atol=2e-3, rtol=1e-3, name="predictions")
    assert_close(loss, g[t + "loss"], atol=2e-3, rtol=1e-3, name="total loss")
    for k, v in g.sub(t + "losses/").items():
        assert_close(ld[k], v, atol=1e-3, rtol=1e-3, name=k)
    loss.backward()
    _grads(model, g, t)


def test_fused_adam_resumes_from_torch_adam_state(golden):
    """steps 1-2 by torch.optim.Adam on the CPU, its state_dict loaded into FusedAdam, step 3 on the device: the parameters
    must equal the reference trajectory's p3 (tests/golden/adam.npz, captured from torch.optim.Adam)."""
    from bmt_amd.optim import FusedAdam
    g = golden("adam.npz")
    p = torch.nn.Parameter(g["p0"].clone())
    ref = torch.optim.Adam([p], lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    for s in (1, 2):
        p.grad = g[f"g{s}"].clone()
        ref.step()
    q = torch.nn.Parameter(p.detach().to(DEV))
    opt = FusedAdam([q], lr=5e-5)
    state = ref.state_dict()
    opt.load_state_dict(state)
    assert opt.state[q]["exp_avg"].device.type == "cuda"
    q.grad = g["g3"].to(DEV)
    opt.step()
    assert_close(q, g["p3"], atol=1e-7, rtol=1e-6, name="resumed Adam step 3")
    assert float(opt.state_dict()["state"][0]["step"]) == 3.0


def test_checkpoint_carries_the_dropout_stream(tmp_path):
    """save_cap_model records {seed, step} of the device-side dropout stream under an extra key; restore_dropout_state puts a
    resumed run back on the same mask sequence (the reference has no resume path: its loaders ignore the key)"""
    import types
    from bmt_amd import checkpoint as ck, ops
    ops.manual_seed(4242, "cuda")
    ops.rng_advance(); ops.rng_advance(); ops.rng_advance()
    torch.cuda.synchronize()
    model = torch.nn.Linear(4, 4).cuda()
    opt = torch.optim.Adam(model.parameters())
    cfg = types.SimpleNamespace(model_checkpoint_path=str(tmp_path))
    path = ck.save_cap_model(cfg, 1, model, opt, 0.0, 0.0, {}, {}, 10)
    cpt = ck.load_checkpoint(path)
    assert cpt["bmt_dropout_state"] == {"seed": 4242, "step": 3}
    ops.manual_seed(7, "cuda")
    assert ck.restore_dropout_state(cpt, "cuda")
    assert [int(x) for x in ops.rng_tensor("cuda").tolist()] == [4242, 3]
    assert not ck.restore_dropout_state({"model_state_dict": {}}, "cuda")
