"""-m gpu: proposal post-processing on the device (csrc/postprocess.hip through bmt_amd.proposals) against the outputs of the
REFERENCE's utilities/proposal_utils.py (tests/golden/postprocess.npz) and against the CPU oracle.  Index / selection work:
bit-exact; the fp32 coordinate arithmetic is the reference's operation order, also compared bit for bit."""
import pytest
import torch

from tests.postprocess_util import CASES, make_preds

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Cfg:
    def __init__(self, k, nms=None):
        self.max_prop_per_vid, self.nms_tiou_thresh = k, nms


@pytest.mark.parametrize("tag", list(CASES))
def test_postprocess_matches_reference(golden, tag):
    from bmt_amd import proposals as pp
    from oracle import bmt_oracle as orc
    g = golden("postprocess.npz")
    B, S, k, seed, ties = CASES[tag]
    preds, dur = make_preds(B, S, seed, ties)
    x = preds.to(DEV)
    post = pp.postprocess_preds(x, _Cfg(k), {"duration_in_secs": dur})
    topk = pp.select_topk_predictions(x, k)
    assert torch.equal(x.cpu(), preds)                                   # neither call writes its input
    assert torch.equal(post.cpu(), orc.postprocess_preds(preds, k, dur))   # the pinned tie order (candidate index)
    assert torch.equal(topk.cpu(), orc.select_topk_predictions(preds, k))
    if not ties:
        assert torch.equal(post.cpu(), g[f"{tag}/post"]) and torch.equal(topk.cpu(), g[f"{tag}/topk"])
    else:
        assert torch.equal(post[:, :, 2].cpu(), g[f"{tag}/post"][:, :, 2])
    # NMS: on the reference's own sorted rows (unique result), standalone and fused
    for thr in (0.3, 0.7):
        for b in range(B):
            got = pp.non_max_suppresion(g[f"{tag}/post"][b].to(DEV), thr)
            assert torch.equal(got.cpu(), g[f"{tag}/nms{thr}/{b}"]), (thr, b)
        out, count = pp.postprocess_preds_nms(x, _Cfg(k, thr), {"duration_in_secs": dur})
        for b in range(B):
            want = orc.non_max_suppression(orc.postprocess_preds(preds, k, dur)[b], thr)
            n = int(count[b])
            assert n == want.shape[0] and torch.equal(out[b, :n].cpu(), want)
            assert not bool(out[b, n:].any())
    # single-video flow of generate_proposals: corners -> trim -> drop short -> top-k
    for b in range(B):
        got = pp.generate_proposals(x[b:b + 1], dur[b], k)
        assert torch.equal(got.cpu(), orc.generate_proposals_post(preds[b:b + 1], dur[b], k))
        if not ties:
            assert torch.equal(got.cpu(), g[f"{tag}/gen/{b}"])
    # the in-place elementwise pieces
    y = x.clone()
    assert pp.get_corner_coords(y) is y
    assert torch.equal(y[:, :64].cpu(), g[f"{tag}/corners_head"])
    pp.trim_proposals(y, dur)
    assert torch.equal(y[:, :64].cpu(), g[f"{tag}/trim_head"])
    assert torch.equal(y.cpu(), orc.trim_proposals(orc.get_corner_coords(preds), dur))


def test_remove_very_short_segments_and_edges():
    from bmt_amd import proposals as pp
    from oracle import bmt_oracle as orc
    preds, dur = make_preds(1, 900, 21)
    seg = orc.trim_proposals(orc.get_corner_coords(preds), dur)
    got = pp.remove_very_short_segments(seg.to(DEV), 0.2)
    assert torch.equal(got.cpu(), orc.remove_very_short_segments(seg, 0.2))
    # nothing survives the filter -> empty result, count 0
    tiny = preds.clone()
    tiny[:, :, 1] = 0.01
    out = pp.generate_proposals(tiny.to(DEV), dur[0], 100)
    assert out.shape == (1, 0, 3)
    # a single candidate, k larger than S
    one = preds[:, :1].to(DEV)
    assert torch.equal(pp.select_topk_predictions(one, 100).cpu(), preds[:, :1])
    # negative and zero confidences, infinities: the order is the IEEE order
    x = preds[:, :64].clone()
    x[0, :, 2] = torch.linspace(-3, 3, 64)
    x[0, 5, 2], x[0, 9, 2], x[0, 11, 2], x[0, 12, 2] = float("inf"), float("-inf"), 0.0, -0.0
    got = pp.select_topk_predictions(x.to(DEV), 64)
    want = orc.select_topk_predictions(x, 64)
    assert torch.equal(got[0, :, 2].cpu(), want[0, :, 2])
    # every confidence equal: pure candidate-index order
    x[0, :, 2] = 0.5
    got, _, idx = pp.select_proposals(x.to(DEV), 10, return_indices=True)
    assert idx[0].tolist() == list(range(10))
    # k beyond the kernel's limit is refused, not truncated
    with pytest.raises(RuntimeError, match="exceeds"):
        pp.select_topk_predictions(torch.zeros(1, 5000, 3, device=DEV), 4096)


def test_full_size_selection_config3():
    """configs[3] candidate count: 10 heads x (48 anchors x 3200 audio + 128 anchors x 1024 video positions) = 2 846 720 per
    video.  Checked against the CPU oracle's stable sort (selection + order bit-exact), with saturated confidences (many
    exact ties at 1.0, as a trained generator produces) in one video."""
    from bmt_amd import proposals as pp
    from oracle import bmt_oracle as orc
    S, B, k = 10 * (48 * 3200 + 128 * 1024), 2, 100
    g = torch.Generator().manual_seed(5)
    conf = torch.sigmoid(torch.randn(B, S, generator=g) * 3)
    conf[1] = torch.sigmoid(torch.randn(S, generator=g) * 12)          # ~8 % of the rows are exactly 1.0
    assert int((conf[1] == 1.0).sum()) > k
    c = torch.rand(B, S, generator=g) * 200
    ln = torch.exp(torch.randn(B, S, generator=g) + 2)
    preds = torch.stack([c, ln, conf], -1).contiguous()
    dur = [180.0, 95.5]
    out, count, idx = pp.select_proposals(preds.to(DEV), k, flags=3, duration_in_secs=dur, return_indices=True)
    want, widx = orc.select_topk_predictions(preds, k, return_indices=True)
    assert torch.equal(idx.cpu(), widx)
    assert torch.equal(out.cpu(), orc.trim_proposals(orc.get_corner_coords(want), dur))
    assert count.tolist() == [k, k]
    # size-independent properties: sorted by confidence, every row drawn from the input, nothing above the k-th was skipped
    assert bool((out[:, 1:, 2] <= out[:, :-1, 2]).all())
    kth = out[:, -1, 2].cpu()
    assert all(int((conf[b] > kth[b]).sum()) < k for b in range(B))
    # the length filter inside the selection == filter first, then select
    gen = pp.generate_proposals(preds[:1].to(DEV), dur[0], k)
    assert torch.equal(gen.cpu(), orc.generate_proposals_post(preds[:1], dur[0], k))
