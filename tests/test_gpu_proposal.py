"""-m gpu: proposal-generator path (Conv1d heads as implicit GEMM, target assignment, decode + YOLO loss) against the
golden vectors captured from the reference and against the CPU oracle."""
import numpy as np
import pytest
import torch

from bmt_amd import synthetic as syn
from tests.gpu_util import assert_close, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["nocollide", "collide"])
def test_make_targets_bit_exact(golden, tag):
    from bmt_amd.model.proposal_generator import make_targets
    g = golden("targets.npz")
    preds = torch.zeros(3, 5, 40, 3, device=DEV)
    obj, noobj, tx, tw, tobj = make_targets(preds, g[f"mt/{tag}/targets"].to(DEV), g[f"mt/{tag}/anchors"].to(DEV),
                                            float(g[f"mt/{tag}/stride"]))
    assert obj.dtype == torch.bool and torch.equal(obj.cpu(), g[f"mt/{tag}/obj"])
    assert torch.equal(noobj.cpu(), g[f"mt/{tag}/noobj"])
    assert torch.equal(tobj.cpu(), g[f"mt/{tag}/tobj"])
    assert torch.equal(tx.cpu(), g[f"mt/{tag}/tx"])                       # gt_x - floor(gt_x): exact fp32 ops
    assert_close(tw, g[f"mt/{tag}/tw"], atol=0, rtol=3e-7, name="target_w (device logf vs host logf)")


@pytest.mark.parametrize("B,S,Din,Dout,k", [(2, 14, 24, 16, 5), (2, 9, 48, 16, 7), (1, 40, 128, 64, 13), (3, 50, 64, 136, 1 + 2 * 15)])
def test_conv1d_heads_vs_torch_conv(B, S, Din, Dout, k):
    from bmt_amd.model.proposal_generator import ConvKFn
    g = torch.Generator().manual_seed(k)
    x = torch.randn(B, S, Din, generator=g)
    W = torch.randn(Dout, Din, k, generator=g) / (Din * k) ** 0.5
    b = torch.randn(Dout, generator=g)
    w = torch.randn(B, S, Dout, generator=g)
    xd, Wd, bd = x.to(DEV).requires_grad_(), W.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ConvKFn.apply(xd, Wd, bd, True, 0.0, 0)
    xr, Wr, br = x.double().requires_grad_(), W.double().requires_grad_(), b.double().requires_grad_()
    want = torch.relu(torch.nn.functional.conv1d(xr.permute(0, 2, 1), Wr, br, padding=k // 2)).permute(0, 2, 1)
    assert_close(y, want, atol=3e-4, name=f"conv k={k}")
    (y * w.to(DEV)).sum().backward()
    (want * w.double()).sum().backward()
    for name, got, ref in (("dx", xd.grad, xr.grad), ("dW", Wd.grad, Wr.grad), ("db", bd.grad, br.grad)):
        assert rel_err(got, ref) < 2e-2, report(got, ref, name)


def _prop_cfg():
    cfg = syn.cfg_tiny(procedure="train_prop")
    cfg.anchors_num_audio, cfg.anchors_num_video = 3, 5
    cfg.conv_layers_audio, cfg.conv_layers_video = [16, 16], [16, 16]
    cfg.kernel_sizes = {"audio": [1, 5], "video": [3, 7]}
    cfg.device = DEV
    return cfg


def test_tiny_proposal_generator(golden):
    from bmt_amd.model.masking import mask
    from bmt_amd.model.proposal_generator import MultimodalProposalGenerator
    g = golden("tiny_prop.npz")
    cfg = _prop_cfg()
    anchors = {"audio": [float(a) for a in g.np("anchors_audio")], "video": [float(a) for a in g.np("anchors_video")]}
    torch.manual_seed(0)
    model = MultimodalProposalGenerator(cfg, anchors)
    sd = g.sub("sd/")
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    fs = {k: g[k].to(DEV) for k in ("rgb", "flow", "audio")}
    masks = {"A_mask": mask(fs["audio"][:, :, 0], None, 1), "V_mask": mask(fs["rgb"][:, :, 0], None, 1)}
    preds, loss, la, lv = model(fs, g["targets"].to(DEV), masks)
    assert_close(preds, g["preds"], atol=2e-3, rtol=1e-3, name="predictions")
    assert_close(loss, g["loss"], atol=2e-3, rtol=1e-3, name="total loss")
    for k, v in g.sub("losses_A/").items():
        assert_close(la[k], v, atol=1e-3, rtol=1e-3, name="A " + k)
    for k, v in g.sub("losses_V/").items():
        assert_close(lv[k], v, atol=1e-3, rtol=1e-3, name="V " + k)
    loss.backward()
    from tests.test_gpu_model import _check_grads
    _check_grads(model.named_parameters(), g.sub("grad/"))
    # inference call: targets None -> loss is the python int 0 and predictions are unchanged
    preds2, loss2, _, _ = model(fs, None, masks)
    assert loss2 == 0
    assert_close(preds2, g["preds_notargets"], atol=2e-3, rtol=1e-3, name="predictions (no targets)")


def test_initial_state_dict_matches_reference_layout(golden):
    """default Sequential indices conv_layers.{0,3,6} with dropout, {0,2,4} without (positional keys)."""
    from bmt_amd.model.proposal_generator import ProposalGenerationHead
    h = ProposalGenerationHead([24, 16, 16, 9], 5, 0.1)
    assert sorted({k.split(".")[1] for k in h.state_dict()}) == ["0", "3", "6"]
    h = ProposalGenerationHead([24, 16, 16, 9], 5, 0.0)
    assert sorted({k.split(".")[1] for k in h.state_dict()}) == ["0", "2", "4"]


def test_proposal_train_step_matches_oracle(golden):
    """train_av_loop (epoch_loops/proposal_epoch_loops.py:35-49): zero_grad -> masks -> forward -> backward -> Adam, two steps
    with dropout off, against the CPU oracle run from the same state_dict."""
    from bmt_amd.model.proposal_generator import MultimodalProposalGenerator
    from bmt_amd.train import ProposalTrainStep
    from oracle import bmt_oracle as orc
    g = golden("tiny_prop.npz")
    cfg = _prop_cfg()        # built with dropout so that the Sequential keys match the fixture; switched off below
    cfg.lr = 1e-3
    cfg.grad_clip = None
    anchors = {"audio": [float(a) for a in g.np("anchors_audio")], "video": [float(a) for a in g.np("anchors_video")]}
    torch.manual_seed(0)
    model = MultimodalProposalGenerator(cfg, anchors)
    sd = g.sub("sd/")
    model.load_state_dict(sd)
    model = model.to(DEV)
    for mod in model.modules():         # dropout off: every module reads its own dout_p at call time
        if hasattr(mod, "dout_p"):
            mod.dout_p = 0.0
    step = ProposalTrainStep(model, cfg, pad_idx=1)
    fs_cpu = {k: g[k] for k in ("rgb", "flow", "audio")}
    fs = {k: v.to(DEV) for k, v in fs_cpu.items()}
    targets = g["targets"]
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    omasks = {"A_mask": orc.mask(fs_cpu["audio"][:, :, 0], None, 1), "V_mask": orc.mask(fs_cpu["rgb"][:, :, 0], None, 1)}
    for it in (1, 2):
        _, loss, _, _ = step(fs, targets.to(DEV))
        for t in p.values():
            t.grad = None
        _, oloss, _, _ = orc.multimodal_proposal_generator(p, cfg, anchors, fs_cpu, targets, omasks)
        oloss.backward()
        with torch.no_grad():
            for k in p:
                if p[k].grad is not None:
                    orc.adam_step(p[k], p[k].grad, m[k], v2[k], it, cfg.lr)
        assert abs(float(loss) - float(oloss.detach())) < 5e-3 * max(1.0, abs(float(oloss.detach()))), (it, float(loss), float(oloss))
    msd = model.state_dict()
    agree = total = 0
    for k in p:
        d = (msd[k].cpu() - p[k].detach()).abs()
        agree += int((d < 2.5e-4).sum())
        total += d.numel()
    assert agree / total > 0.97, agree / total
