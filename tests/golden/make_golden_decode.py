"""Golden vectors for greedy decoding: the REFERENCE's greedy_decoder
(epoch_loops/captioning_epoch_loops.py:39-65) run on the reference model, imported from /root/reference in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_decode.py

Stored: the seeded inputs' identifiers, the decoded token matrix and the top-1/top-2 margin of every decision (from the oracle
run on the same weights) -- data only."""
import os
import sys

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
torch.set_num_threads(8)
out = {}
for tag, cfgfn, V, B, Tv, Ta, max_len, wscale in (("tiny", syn.cfg_tiny, 11, 3, 9, 14, 8, 6.0), ("cfg0", syn.cfg_config0, 10, 2, 12, 40, 10, 6.0)):
    cfg = cfgfn()
    cfg.device = "cpu"
    glove = syn.make_glove(V, cfg.d_model_caps)
    torch.manual_seed(0)
    model = ref.captioning_module.BiModalTransformer(cfg, syn.FakeTrainDataset(V, glove))
    # xavier-initialised generators give near-uniform log-probs (arg-max margins ~1e-3): scale the output layer so that the
    # decisions are well separated and the decoded sequences differ between samples
    with torch.no_grad():
        model.generator.linear.weight.mul_(wscale)
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, 4, V, seed=4321)
    fs = batch["feature_stacks"]
    trg = ref.cap_loops.greedy_decoder(model, fs, max_len, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, "audio_video")
    otrg, margins = orc.greedy_decode(sd, cfg, fs, max_len, syn.START_IDX, syn.END_IDX, syn.PAD_IDX, return_margins=True)
    assert torch.equal(trg, otrg), (trg, otrg)
    print(tag, "tokens", trg.tolist(), "min margin %.3e" % float(margins.min()))
    out[f"{tag}/tokens"] = trg.numpy()
    out[f"{tag}/margins"] = margins.numpy()
    out[f"{tag}/meta"] = np.array([V, B, Tv, Ta, max_len, 4321])
    out[f"{tag}/wscale"] = np.array(wscale)
np.savez_compressed(os.path.join(HERE, "greedy_decode.npz"), **out)
print("wrote greedy_decode.npz")
