"""-m gpu: the device side of feature ingest (bmt_pad_batch + the prefetching FeatureIngest) against the batch the REFERENCE's
loader + pad_sequence produced on the same seeded files (tests/golden/ingest.npz).  Byte work: bit-exact."""
import ctypes as C

import pytest
import torch

from tests.ingest_util import VIDEOS, items, write_features

pytestmark = pytest.mark.gpu
DEV = "cuda"
PAD = 1
NAMES = ["i3d_features", "vggish_features"]


@pytest.fixture(scope="module")
def feats(tmp_path_factory):
    return write_features(str(tmp_path_factory.mktemp("feats")))


def test_caption_batches_match_reference_collate(golden, feats):
    from bmt_amd.ingest import FeatureIngest
    from bmt_amd.train import make_masks
    g = golden("ingest.npz")
    ing = FeatureIngest(feats, NAMES, PAD, DEV)
    its = items()
    first, second = ing.submit(its), ing.submit(its[:5])          # two batches in flight: the staging buffers alternate
    third = ing.submit(list(reversed(its)))
    a, b, c = ing.result(first), ing.result(second), ing.result(third)
    for k in ("rgb", "flow", "audio"):
        assert a[k].is_cuda and torch.equal(a[k].cpu(), g[f"batch/{k}"]), k
        want5 = g[f"batch/{k}"][:5]
        T5 = b[k].shape[1]
        assert torch.equal(b[k].cpu(), want5[:, :T5]), k
        pad = 0.0 if k == "flow" else float(PAD)
        assert bool((want5[:, T5:] == pad).all())                # the shorter batch is padded to ITS longest sample
        assert torch.equal(c[k].cpu(), g[f"batch/{k}"].flip(0)), k
    # the masks the training loop derives from the ingested batch are the ones the reference batch gives
    m = make_masks(a, None, "audio_video", PAD)
    ref = make_masks({k: g[f"batch/{k}"].to(DEV) for k in ("rgb", "flow", "audio")}, None, "audio_video", PAD)
    assert torch.equal(m["V_mask"], ref["V_mask"]) and torch.equal(m["A_mask"], ref["A_mask"])
    ing.close()


def test_full_feature_batches_match_reference(golden, feats):
    from bmt_amd.ingest import FeatureIngest
    g = golden("ingest.npz")
    ing = FeatureIngest(feats, NAMES, PAD, DEV, get_full_feat=True)
    vids = ["v_a", "v_b", "v_c", "v_f"]
    out = ing([(v, None, None, None) for v in vids])
    for k in ("rgb", "flow", "audio"):
        want = torch.stack([g[f"full/{v}/{k}"] for v in vids])
        assert torch.equal(out[k].cpu(), want), k
        assert out["orig_feat_length"][k] == [int(g.np(f"full/{v}/len_{k}")) for v in vids]
    with pytest.raises(FileNotFoundError):
        ing([("v_d", None, None, None)])                           # the proposal dataset filters such videos out beforehand
    ing.close()


def test_one_missing_i3d_stack_zeroes_both(tmp_path):
    """datasets/load_features.py:70-93 loads rgb and flow together: if EITHER file is missing both stacks become one zero row
    (planning them independently gave a real rgb crop beside a 1-row zero flow, with different T)"""
    import os
    from bmt_amd.ingest import FeatureIngest
    feats = write_features(str(tmp_path))
    os.remove(os.path.join(feats.video_features_path, "v_f_flow.npy"))
    ing = FeatureIngest(feats, NAMES, PAD, DEV)
    out = ing([("v_f", 0.0, 19.0, VIDEOS["v_f"][2]), ("v_a", 3.0, 17.5, VIDEOS["v_a"][2])])
    assert out["rgb"].shape == out["flow"].shape
    assert bool((out["rgb"][0, 0] == 0).all()) and bool((out["flow"][0, 0] == 0).all())          # the single zero row
    assert bool((out["rgb"][0, 1:] == float(PAD)).all()) and bool((out["flow"][0, 1:] == 0).all())   # then padding
    assert bool((out["rgb"][1, 0] != 0).any()) and bool((out["flow"][1, 0] != 0).any())            # the intact sample is untouched
    ing.close()


@pytest.mark.parametrize("D", [1, 7, 24, 1024])
def test_pad_batch_kernel(D):
    from bmt_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(D)
    lens = [3, 0, 11, 1, 6]
    T = 12
    packed = torch.rand(sum(lens), D, generator=g)
    offs = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int64)
    out = torch.empty(len(lens), T, D, device=DEV)
    pd, od = packed.to(DEV), offs.to(DEV)
    _lib.check(lib.bmt_pad_batch(C.c_void_p(pd.data_ptr()), C.c_void_p(od.data_ptr()), len(lens), T, D, -2.5,
                                 C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "bmt_pad_batch")
    want = torch.full((len(lens), T, D), -2.5)
    for b, n in enumerate(lens):
        want[b, :n] = packed[offs[b]:offs[b] + n]
    assert torch.equal(out.cpu(), want)
