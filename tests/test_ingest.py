"""CPU: feature ingest host logic -- the native .npy row reader, the crop arithmetic and
load_features_from_npy against what the REFERENCE's datasets/load_features.py returned on the same seeded files
(tests/golden/ingest.npz), and the oracle's collate against the reference-derived padded batch."""
import os

import numpy as np
import pytest
import torch

from bmt_amd import ingest
from oracle import bmt_oracle as orc
from tests.ingest_util import D_AUD, D_VID, VIDEOS, items, make_array, write_features

PAD = 1
NAMES = ["i3d_features", "vggish_features"]


@pytest.fixture(scope="module")
def feats(tmp_path_factory):
    return write_features(str(tmp_path_factory.mktemp("feats")))


def test_load_features_from_npy_matches_reference(golden, feats):
    g = golden("ingest.npz")
    for i, (vid, s, e, dur) in enumerate(items()):
        st = ingest.load_features_from_npy(feats, NAMES, vid, s, e, dur, PAD, get_full_feat=False)
        for k in ("rgb", "flow", "audio"):
            assert (st[k] is None) == bool(g.np(f"seg/{i}/{k}_none")), (i, k)
            if st[k] is not None:
                assert st[k].dtype == torch.float32 and torch.equal(st[k], g[f"seg/{i}/{k}"]), (i, vid, k)
    for vid in ("v_a", "v_b", "v_c", "v_f"):
        st = ingest.load_features_from_npy(feats, NAMES, vid, None, None, None, PAD, get_full_feat=True)
        for k in ("rgb", "flow", "audio"):
            assert torch.equal(st[k], g[f"full/{vid}/{k}"]), (vid, k)
            assert st["orig_feat_length"][k] == int(g.np(f"full/{vid}/len_{k}"))


def test_oracle_collate_matches_reference_batch(golden):
    g = golden("ingest.npz")
    arrays, segs = [], []
    for vid, s, e, dur in items():
        sv, sa, _ = VIDEOS[vid]
        arrays.append({"rgb": None if sv is None else torch.from_numpy(make_array(vid, "rgb", sv, D_VID)),
                       "flow": None if sv is None else torch.from_numpy(make_array(vid, "flow", sv, D_VID)),
                       "audio": None if sa is None else torch.from_numpy(make_array(vid, "audio", sa, D_AUD))})
        segs.append((s, e, dur))
    got = orc.collate_caption_features(arrays, segs, PAD, D_VID, D_AUD)
    for k in ("rgb", "flow", "audio"):
        assert torch.equal(got[k], g[f"batch/{k}"]), k
    full = orc.collate_proposal_features([arrays[0], arrays[3]], PAD, 100, 250)
    assert torch.equal(full["rgb"][0], g["full/v_a/rgb"]) and torch.equal(full["flow"][1], g["full/v_b/flow"])
    assert torch.equal(full["audio"][1], g["full/v_b/audio"])


def test_crop_rows_is_python_slicing_of_the_reference_indices():
    rng = np.random.default_rng(3)
    for _ in range(2000):
        S = int(rng.integers(1, 400))
        dur = float(rng.uniform(0.5, 300))
        a, b = sorted(rng.uniform(-0.1 * dur, 1.2 * dur, 2))
        if rng.random() < 0.2:
            b = a
        x = torch.arange(S).view(S, 1).float()
        want = orc.crop_a_segment(x, a, b, dur)
        r = ingest.crop_rows(S, a, b, dur)
        if want is None:
            assert r is None
        else:
            assert r is not None and torch.equal(x[r[0]:r[1]], want)


def test_npy_reader_formats_and_errors(tmp_path):
    a = (np.arange(7 * 5, dtype=np.float32).reshape(7, 5) - 3) / 7
    np.save(tmp_path / "f4.npy", a)
    np.save(tmp_path / "f8.npy", a.astype(np.float64) * (1 + 1e-9))
    with open(tmp_path / "v2.npy", "wb") as f:
        np.lib.format.write_array(f, a, version=(2, 0))
    np.save(tmp_path / "vec.npy", a[0])
    np.save(tmp_path / "int.npy", a.astype(np.int32))
    np.save(tmp_path / "fortran.npy", np.asfortranarray(a))
    np.save(tmp_path / "cube.npy", np.zeros((2, 3, 4), np.float32))
    assert ingest.npy_shape(tmp_path / "f4.npy") == (7, 5)
    assert torch.equal(ingest.read_rows(tmp_path / "f4.npy"), torch.from_numpy(a))
    assert torch.equal(ingest.read_rows(tmp_path / "f4.npy", 2, 5), torch.from_numpy(a[2:5]))
    assert torch.equal(ingest.read_rows(tmp_path / "f4.npy", 5, 99), torch.from_numpy(a[5:]))
    assert ingest.read_rows(tmp_path / "f4.npy", 6, 3).shape == (0, 5)
    assert torch.equal(ingest.read_rows(tmp_path / "v2.npy", 1, 4), torch.from_numpy(a[1:4]))
    want64 = torch.from_numpy(np.load(tmp_path / "f8.npy")).float()
    assert torch.equal(ingest.read_rows(tmp_path / "f8.npy"), want64)
    assert ingest.read_rows(tmp_path / "vec.npy").shape == (5, 1)
    for bad in ("int.npy", "fortran.npy", "cube.npy"):
        with pytest.raises(RuntimeError, match="bmt_npy"):
            ingest.read_rows(tmp_path / bad)
    with pytest.raises(FileNotFoundError):
        ingest.read_rows(tmp_path / "nope.npy")
    open(tmp_path / "junk.npy", "wb").write(b"not a numpy file at all")
    with pytest.raises(RuntimeError, match="not a .npy"):
        ingest.npy_shape(tmp_path / "junk.npy")
    raw = open(tmp_path / "f4.npy", "rb").read()
    open(tmp_path / "cut.npy", "wb").write(raw[:-20])
    with pytest.raises(RuntimeError, match="short read"):
        ingest.read_rows(tmp_path / "cut.npy")
    small = torch.empty(4)
    with pytest.raises(RuntimeError, match="destination holds"):
        ingest.read_rows(tmp_path / "f4.npy", 0, 7, out=small)


def test_feature_ingest_needs_a_gpu(feats):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ingest.FeatureIngest(feats, NAMES, PAD, "cpu")
