"""Golden vectors for SURVEY.md 8(f3): the REFERENCE's datasets/load_features.py (imported from /root/reference in this
container) run on the seeded .npy files of tests/ingest_util.py, and the collate convention of
datasets/captioning_dataset.py:257-261 (pad_sequence with pad_idx / 0) applied to its outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ingest.py

Stored: per segment the reference's cropped stacks (shape + values), the padded batch, and the get_full_feat stacks."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

import importlib

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

import _refimport
from ingest_util import D_AUD, D_VID, VIDEOS, items, write_features

_refimport.import_reference()
lf = importlib.import_module("datasets.load_features")
PAD = 1
out = {}
with tempfile.TemporaryDirectory() as d:
    cfg = write_features(d)
    names = ["i3d_features", "vggish_features"]
    rgb, flow, aud = [], [], []
    for i, (vid, s, e, dur) in enumerate(items()):
        st = lf.load_features_from_npy(cfg, names, vid, s, e, dur, PAD, get_full_feat=False)
        for k in ("rgb", "flow", "audio"):
            out[f"seg/{i}/{k}_none"] = np.array(st[k] is None)
            if st[k] is not None:
                out[f"seg/{i}/{k}"] = st[k].numpy()
        # captioning_dataset.py:238-248: missing -> one zero row
        r, f, a = st["rgb"], st["flow"], st["audio"]
        if r is None and f is None:
            r, f = lf.fill_missing_features("zero", D_VID), lf.fill_missing_features("zero", D_VID)
        if a is None:
            a = lf.fill_missing_features("zero", D_AUD)
        rgb.append(r); flow.append(f); aud.append(a)
    out["batch/rgb"] = pad_sequence(rgb, batch_first=True, padding_value=PAD).numpy()
    out["batch/flow"] = pad_sequence(flow, batch_first=True, padding_value=0).numpy()
    out["batch/audio"] = pad_sequence(aud, batch_first=True, padding_value=PAD).numpy()
    for vid in ("v_a", "v_b", "v_c", "v_f"):
        st = lf.load_features_from_npy(cfg, names, vid, None, None, None, PAD, get_full_feat=True)
        for k in ("rgb", "flow", "audio"):
            out[f"full/{vid}/{k}"] = st[k].numpy()
            out[f"full/{vid}/len_{k}"] = np.array(st["orig_feat_length"][k])
    print({k: v.shape for k, v in out.items() if k.startswith("batch/")})
path = os.path.join(HERE, "ingest.npz")
np.savez_compressed(path, **out)
print("wrote ingest.npz %.1f KB" % (os.path.getsize(path) / 1024))
