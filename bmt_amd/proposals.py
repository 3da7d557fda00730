"""Proposal post-processing on the device: drop-in for the functions of the reference's
utilities/proposal_utils.py that sit between the proposal generator and the captioning model.

Same names, arguments and results as the reference (file:line in each docstring); every one of them runs in
libbmt_hip.so (csrc/postprocess.hip: exact radix select instead of a full argsort over up to 2.9 M candidates per video, fused
with the coordinate transforms, the short-segment filter and greedy NMS).  No CPU path: tensors must live on the GPU.

One freedom of the reference is pinned here: equal confidences are ordered by candidate index (the reference's unstable
argsort leaves that order open)."""
import ctypes as C

import torch

from . import _lib
from ._lib import PP_CORNERS, PP_FILTER, PP_TRIM, SelectProposalsArgs
from .ops import _p, _st, lib

_ws = {}


def _workspace(nbytes, device):
    buf = _ws.get(device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[device] = buf
    return buf


def _durations(duration_in_secs, B, device):
    if duration_in_secs is None:
        return None
    d = torch.as_tensor(duration_in_secs, dtype=torch.float32).reshape(-1)
    if d.numel() == 1 and B > 1:
        d = d.expand(B)
    assert d.numel() == B, f"{d.numel()} durations for {B} videos"
    return d.to(device).contiguous()


def select_proposals(model_output, k, flags=0, duration_in_secs=None, min_len=0.0, nms_tiou_thresh=None, return_indices=False):
    """The fused kernel sequence behind every function below.  model_output (B, S, 3) fp32 on the GPU; returns
    (out (B, k', 3), count (B,) int32[, idx (B, k') int64]) with k' = min(k, S); rows >= count[b] are zero."""
    assert model_output.is_cuda and model_output.dtype == torch.float32 and model_output.dim() == 3 and model_output.shape[2] == 3, \
        "predictions must be a (B, S, 3) fp32 tensor on the GPU"
    x = model_output.contiguous()
    B, S, _ = x.shape
    k = min(int(k), S)
    out = torch.empty(B, k, 3, device=x.device, dtype=torch.float32)
    idx = torch.empty(B, k, device=x.device, dtype=torch.int64) if return_indices else None
    count = torch.empty(B, device=x.device, dtype=torch.int32)
    if B == 0 or S == 0 or k == 0:
        return (out, count.zero_(), idx) if return_indices else (out, count.zero_())
    dur = _durations(duration_in_secs, B, x.device)
    ws = _workspace(int(lib.bmt_select_proposals_ws_bytes(B, S, k)), x.device)
    a = SelectProposalsArgs(preds=_p(x), B=B, S=S, k=k, flags=flags, durations=_p(dur), min_len=float(min_len),
                            nms_thresh=-1.0 if nms_tiou_thresh is None else float(nms_tiou_thresh), out=_p(out), out_idx=_p(idx),
                            count=_p(count), ws=_p(ws), ws_bytes=ws.numel())
    _lib.check(lib.bmt_select_proposals(C.byref(a), _st()), "bmt_select_proposals")
    return (out, count, idx) if return_indices else (out, count)


def select_topk_predictions(model_output, k):
    '''model_output (B, S*A, num_feats) -> (B, k, num_feats), rows sorted on confidence (utilities/proposal_utils.py:136-149)'''
    out, _ = select_proposals(model_output, k)
    return out


def get_corner_coords(predictions):
    '''predictions (B, S*A, num_feats): (center, length) -> (start, end), in place (utilities/proposal_utils.py:115-121)'''
    return _transform(predictions, PP_CORNERS, None)


def trim_proposals(model_output, duration_in_secs):
    '''Changes in-place model_output (B, AS, num_feats), starts & ends are in seconds (utilities/proposal_utils.py:152-161)'''
    return _transform(model_output, PP_TRIM, duration_in_secs)


def _transform(x, flags, duration_in_secs):
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 3 and x.is_contiguous(), \
        "predictions must be a contiguous (B, S, 3) fp32 tensor on the GPU"
    B, S, _ = x.shape
    if B and S:
        dur = _durations(duration_in_secs, B, x.device)
        _lib.check(lib.bmt_transform_proposals(_p(x), B, S, flags, _p(dur), _st()), "bmt_transform_proposals")
    return x


def remove_very_short_segments(model_output, shortest_segment_prior):
    '''one video (1, A*S, 3) of (start, end, conf) rows -> the rows longer than the prior, input order kept
    (utilities/proposal_utils.py:163-172).  Variable-size result: the kernel selects with k = S under the length filter and
    the survivors are put back into candidate order.'''
    assert model_output.shape[0] == 1
    S = model_output.shape[1]
    if S > 2048:
        raise ValueError("remove_very_short_segments on more than 2048 rows: use generate_proposals, which filters inside "
                         "the selection instead of materialising the filtered tensor")
    out, count, idx = select_proposals(model_output, S, flags=PP_FILTER, min_len=shortest_segment_prior, return_indices=True)
    n = int(count[0])
    order = idx[0, :n].sort().values
    return model_output[:, order, :]


def non_max_suppresion(video_preds, tIoU_threshold):
    '''video_preds (AS, num_features) sorted by confidence -> the rows greedy NMS keeps (utilities/proposal_utils.py:175-194)'''
    n = video_preds.shape[0]
    if n == 0:
        return video_preds
    # the rows are already sorted: select with k = n on a strictly decreasing surrogate confidence keeps their order
    x = video_preds.detach().clone().view(1, n, 3)
    conf = x[0, :, 2].clone()
    x[0, :, 2] = torch.arange(n, 0, -1, device=x.device, dtype=torch.float32)
    out, count, idx = select_proposals(x, n, nms_tiou_thresh=tIoU_threshold, return_indices=True)
    keep = idx[0, :int(count[0])]
    res = out[0, :int(count[0])].clone()
    res[:, 2] = conf[keep]
    return res


def postprocess_preds(model_output, cfg, batch):
    '''model_output (B, AS, num_features) with center & length: top-[max_prop_per_vid] -> (start, end) -> trimmed to the
    duration (utilities/proposal_utils.py:196-212); one kernel sequence, the transforms touch the k winners only'''
    out, _ = select_proposals(model_output, cfg.max_prop_per_vid, flags=PP_CORNERS | PP_TRIM,
                              duration_in_secs=batch['duration_in_secs'])
    return out


def postprocess_preds_nms(model_output, cfg, batch):
    '''postprocess_preds followed by the per-video NMS of AnetPredictions.add_new_predictions
    (utilities/proposal_utils.py:242-247) when cfg.nms_tiou_thresh is set: (out (B, k, 3), count (B,)); rows >= count are 0'''
    return select_proposals(model_output, cfg.max_prop_per_vid, flags=PP_CORNERS | PP_TRIM,
                            duration_in_secs=batch['duration_in_secs'], nms_tiou_thresh=cfg.nms_tiou_thresh)


def generate_proposals(predictions, duration_in_secs, max_prop_per_vid, shortest_segment_prior=0.2):
    '''the post-processing of generate_proposals (sample/single_video_prediction.py:176-186) on the generator's output
    (1, AS, 3): corners -> trim -> drop segments not longer than the prior -> top-k; returns (1, n <= k, 3)'''
    out, count = select_proposals(predictions, max_prop_per_vid, flags=PP_CORNERS | PP_TRIM | PP_FILTER,
                                  duration_in_secs=duration_in_secs, min_len=shortest_segment_prior)
    if predictions.shape[0] == 1:
        return out[:, :int(count[0])]
    return out, count
