"""Golden vectors for SURVEY.md 8(f2): the REFERENCE's proposal post-processing (utilities/proposal_utils.py:115-212,
imported from /root/reference in this container) on seeded prediction tensors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_postprocess.py

Inputs are re-created from the seed by ``make_preds`` (tests/postprocess_util.py); stored are the reference's outputs.
'distinct' cases have pairwise different confidences (the reference's result is unique: bit-exact target); 'ties' cases
quantise the confidences, where the reference's unstable argsort leaves the order inside a tie group open -- stored for the
property tests (same confidence column, every row drawn from the input)."""
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
from postprocess_util import CASES, make_preds
from oracle import bmt_oracle as orc

ref = _refimport.import_reference()
pu = ref.proposal_utils
out = {}
for tag, (B, S, k, seed, ties) in CASES.items():
    preds, dur = make_preds(B, S, seed, ties)
    # validation flow: postprocess_preds (top-k -> corners -> trim), then NMS per video
    cfg = type("Cfg", (), {"max_prop_per_vid": k})()
    post = pu.postprocess_preds(preds.clone(), cfg, {"duration_in_secs": dur})
    out[f"{tag}/post"] = post
    topk = pu.select_topk_predictions(preds.clone(), k)
    out[f"{tag}/topk"] = topk
    mine = orc.postprocess_preds(preds, k, dur)
    if not ties:
        assert torch.equal(post, mine), tag
        assert torch.equal(topk, orc.select_topk_predictions(preds, k)), tag
    else:
        assert torch.equal(post[:, :, 2], mine[:, :, 2]), tag
    for thr in (0.3, 0.7):
        for b in range(B):
            nms = pu.non_max_suppresion(post[b].clone(), thr)
            out[f"{tag}/nms{thr}/{b}"] = nms
            assert torch.equal(nms, orc.non_max_suppression(post[b], thr)), (tag, thr, b)   # same input rows: always unique
    # single-video flow (generate_proposals): corners -> trim -> remove short -> top-k
    for b in range(B):
        p1 = preds[b:b + 1].clone()
        g = pu.get_corner_coords(p1)
        g = pu.trim_proposals(g, dur[b])
        g = pu.remove_very_short_segments(g, shortest_segment_prior=0.2)
        g = pu.select_topk_predictions(g, k=k)
        out[f"{tag}/gen/{b}"] = g
        m = orc.generate_proposals_post(preds[b:b + 1], dur[b], k)
        if not ties:
            assert torch.equal(g, m), (tag, b)
        else:
            assert torch.equal(g[:, :, 2], m[:, :, 2]), (tag, b)
    # the elementwise pieces on their own
    out[f"{tag}/corners_head"] = pu.get_corner_coords(preds.clone())[:, :64]
    out[f"{tag}/trim_head"] = pu.trim_proposals(pu.get_corner_coords(preds.clone()), dur)[:, :64]
    print(tag, "ok", tuple(post.shape))
path = os.path.join(HERE, "postprocess.npz")
np.savez_compressed(path, **{k: v.numpy() for k, v in out.items()})
print("wrote postprocess.npz %.1f KB" % (os.path.getsize(path) / 1024))
