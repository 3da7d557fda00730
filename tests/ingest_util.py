"""Seeded .npy feature files for the ingest tests (shared by the golden generator and the tests).  Values come from uniform
draws and exact arithmetic only, so every host writes the same bytes."""
import os
from types import SimpleNamespace

import numpy as np
import torch

D_VID, D_AUD = 48, 24
# video_id -> (rows of the i3d stacks, rows of the vggish stack, duration in seconds); None: the file does not exist
VIDEOS = {"v_a": (14, 36, 35.2), "v_b": (1, 3, 2.1), "v_c": (97, 240, 231.9), "v_d": (None, 55, 50.0), "v_e": (30, None, 77.7),
          "v_f": (8, 20, 19.0)}
# (video_id, start, end): ordinary segments, a segment shorter than a feature step, one at the very end, one past the end,
# start == end == duration, zero-length at 0, missing modalities
SEGMENTS = [("v_a", 3.0, 17.5), ("v_a", 34.9, 35.2), ("v_a", 35.2, 35.2), ("v_b", 0.0, 2.1), ("v_c", 100.0, 100.01),
            ("v_c", 0.0, 231.9), ("v_c", 200.0, 260.0), ("v_d", 5.0, 25.0), ("v_e", 10.0, 70.0), ("v_f", 0.0, 0.0),
            ("v_f", 18.99, 19.0), ("v_c", 17.3, 17.4)]


def make_array(video_id, kind, rows, cols):
    g = torch.Generator().manual_seed(sum(ord(c) for c in video_id) * 7 + {"rgb": 1, "flow": 2, "audio": 3}[kind])
    return (torch.rand(rows, cols, generator=g) * 4 - 1).numpy()


def write_features(root, float64_for=("v_f",), version2_for=("v_b",)):
    """writes the seeded files under root/{video,audio}; returns a cfg-like namespace pointing at them"""
    vdir, adir = os.path.join(root, "video"), os.path.join(root, "audio")
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(adir, exist_ok=True)
    for vid, (sv, sa, _) in VIDEOS.items():
        def save(path, arr):
            if vid in float64_for:
                arr = arr.astype(np.float64)
            if vid in version2_for:
                with open(path, "wb") as f:
                    np.lib.format.write_array(f, arr, version=(2, 0))
            else:
                np.save(path, arr)
        if sv is not None:
            save(os.path.join(vdir, f"{vid}_rgb.npy"), make_array(vid, "rgb", sv, D_VID))
            save(os.path.join(vdir, f"{vid}_flow.npy"), make_array(vid, "flow", sv, D_VID))
        if sa is not None:
            save(os.path.join(adir, f"{vid}.npy"), make_array(vid, "audio", sa, D_AUD))
    return SimpleNamespace(video_features_path=vdir, audio_features_path=adir, pad_feats_up_to={"video": 100, "audio": 250},
                           d_vid=D_VID, d_aud=D_AUD)


def items():
    return [(vid, s, e, VIDEOS[vid][2]) for vid, s, e in SEGMENTS]
