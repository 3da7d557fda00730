"""Golden vectors for SURVEY.md 8(f4): the uni-modal surfaces (Transformer :16-98, ProposalGenerator :50-212) and the
checkpoint dict layout (save_model, epoch_loops/captioning_epoch_loops.py:68-88, epoch_loops/proposal_epoch_loops.py:10-25),
captured from the REFERENCE imported from /root/reference in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_unimodal.py

Data only: seeded inputs, the reference's state_dict keys / shapes / sha256 (the weights are re-created bit for bit from the
seed by the same constructor order), outputs / loss / gradient norms / small gradients, and the key list of the checkpoint
dicts the reference's save_model functions write (through nn.DataParallel, i.e. with the ``module.`` prefix).

Widths: video d_model 256 (d_k 64), audio d_model 128 (d_k 32), H = 4 -- head widths the attention kernels are built for."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

import numpy as np
import torch

import _refimport
from bmt_amd import synthetic as syn
from oracle import bmt_oracle as orc

ref = _refimport.import_reference()
import importlib
prop_loops = importlib.import_module("epoch_loops.proposal_epoch_loops")
torch.set_num_threads(8)
out = {}
V, B, Tv, Ta, Tc = 11, 3, 9, 14, 7

SMALL = 4096


def unimodal_cfg(**kw):
    cfg = syn.make_cfg(d_model=128, H=4, N=1, d_aud=128, d_vid=256, d_model_caps=20, **kw)
    cfg.device = "cpu"
    return cfg


def store_model(t, model):
    sd = model.state_dict()
    out[t + "sd_keys"] = np.array(list(sd.keys()))
    out[t + "sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    out[t + "sd_digest"] = np.array(orc.state_dict_digest(sd))
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if p.numel() <= SMALL:
            out[t + "grad/" + k] = p.grad.detach().clone()
    out[t + "grad_names"] = np.array(names)
    out[t + "grad_norms"] = np.array(norms)


# ---------------------------------------------------------------- uni-modal captioning Transformer
for modality, with_glove_dim in (("video", "same"), ("audio", "other")):
    cfg = unimodal_cfg(modality=modality)
    d_model = cfg.d_model_video if modality == "video" else cfg.d_model_audio
    # 'same': GloVe width == d_model (from_pretrained branch); 'other': Embedding -> Linear -> ReLU branch (blocks.py:57-62)
    glove = syn.make_glove(V, d_model if with_glove_dim == "same" else 20)
    torch.manual_seed(0)
    model = ref.captioning_module.Transformer(syn.FakeTrainDataset(V, glove), cfg).eval()
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=77)
    fs, caps = batch["feature_stacks"], batch["captions"]
    x, y = caps[:, :-1], caps[:, 1:]
    masks = ref.cap_loops.make_masks(fs, x, modality, syn.PAD_IDX)
    pred = model(fs, x, masks)
    n_tokens = (y != syn.PAD_IDX).sum()
    loss = ref.label_smoothing.LabelSmoothing(cfg.smoothing, syn.PAD_IDX)(pred, y) / n_tokens
    loss.backward()
    t = f"cap_{modality}/"
    out.update({t + "rgb": fs["rgb"], t + "flow": fs["flow"], t + "audio": fs["audio"], t + "captions": caps, t + "pred": pred,
                t + "loss": loss, t + "glove_dim": np.array(glove.shape[1]), t + "meta": np.array([V, B, Tv, Ta, Tc, 77])})
    for k, m in masks.items():
        out[t + "mask/" + k] = m
    store_model(t, model)
    # greedy decoding through the uni-modal branch of the reference decoder
    with torch.no_grad():
        model.generator.linear.weight.mul_(6.0)
    trg = ref.cap_loops.greedy_decoder(model, fs, 6, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, modality)
    out[t + "greedy_tokens_wscale6"] = trg
    print(modality, "loss", float(loss), "greedy", trg.tolist())

# the checkpoint the reference writes for a (DataParallel-wrapped) captioning model
cfg = syn.cfg_tiny()
cfg.device = "cpu"
torch.manual_seed(0)
cap = ref.captioning_module.BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps)))
class _DP(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.module = m
dp = _DP(cap)
opt = torch.optim.Adam([p for p in dp.parameters() if p.requires_grad], lr=5e-5)
with tempfile.TemporaryDirectory() as d:
    cfg.model_checkpoint_path = d
    ref.cap_loops.save_model(cfg, 3, dp, opt, 1.5, 2.5, {"m": 1}, {"m": 2}, V)
    cpt = torch.load(os.path.join(d, "best_cap_model.pt"), map_location="cpu", weights_only=False)
out["cpt_cap/keys"] = np.array(list(cpt.keys()))
out["cpt_cap/state_keys"] = np.array(list(cpt["model_state_dict"].keys()))
out["cpt_cap/file"] = np.array("best_cap_model.pt")

# ---------------------------------------------------------------- uni-modal proposal generator
anchors = {"audio": [1.5, 6.0, 20.0], "video": [1.0, 3.0, 8.0, 20.0, 60.0]}
for modality in ("video", "audio"):
    cfg = unimodal_cfg(procedure="train_prop", modality=modality)
    cfg.anchors_num_audio, cfg.anchors_num_video = 3, 5
    cfg.conv_layers_audio, cfg.conv_layers_video = [16, 16], [16, 16]
    cfg.kernel_sizes = {"audio": [1, 5], "video": [3, 7]}
    torch.manual_seed(0)
    model = ref.proposal_generator.ProposalGenerator(cfg, anchors).eval()
    batch = syn.make_prop_batch(cfg, 2, Tv, Ta, seed=6, events_per_video=2)
    fs = batch["feature_stacks"]
    masks = ref.cap_loops.make_masks(fs, None, modality, 1)
    preds, loss, ld = model(fs, batch["targets"], masks)
    loss.backward()
    t = f"prop_{modality}/"
    out.update({t + "rgb": fs["rgb"], t + "flow": fs["flow"], t + "audio": fs["audio"], t + "targets": batch["targets"],
                t + "preds": preds, t + "loss": loss})
    for k, v in ld.items():
        out[t + "losses/" + k] = v
    store_model(t, model)
    print(modality, "prop loss", float(loss), tuple(preds.shape))

class _DPp(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.module = m
        self.anchors = m.anchors
dp = _DPp(model)
opt = torch.optim.Adam(dp.parameters(), lr=5e-5)
with tempfile.TemporaryDirectory() as d:
    cfg.log_path = d
    prop_loops.save_model(cfg, 2, dp, opt, None, {"f1": 0.5}, 0.5)
    cpt = torch.load(os.path.join(d, "best_prop_model.pt"), map_location="cpu", weights_only=False)
out["cpt_prop/keys"] = np.array(list(cpt.keys()))
out["cpt_prop/state_keys"] = np.array(list(cpt["model_state_dict"].keys()))
out["cpt_prop/file"] = np.array("best_prop_model.pt")

path = os.path.join(HERE, "unimodal.npz")
np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
print("wrote unimodal.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024, len(out)))
