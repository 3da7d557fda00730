"""Feature ingest: .npy features -> the padded device tensors the models take.

Host-side mirror of the reference's datasets/load_features.py (same function names, arguments and results) plus the batch
path that replaces the collate code of datasets/captioning_dataset.py:214-275 and datasets/proposal_dataset.py:69-103:

* the native reader (csrc/ingest.hip, ``bmt_npy_read_rows``) copies only the CROPPED rows of every file straight into one
  pinned staging buffer (parallel, GIL-free) -- no np.load of the whole array, no per-sample tensors, no pad_sequence;
* ONE async H2D copy of the packed ragged rows per batch (padding never crosses PCIe);
* ``bmt_pad_batch`` writes the padded (B, T, D) tensors on the device with the reference's pad convention
  (rgb / audio: pad_idx, flow: 0, captioning_dataset.py:257-261; load_features.py:62,76-77);
* everything runs on a side stream from a worker thread: ``submit`` the next batch, train on the current one, ``result``.

The device path has no CPU fallback (``FeatureIngest`` needs the GPU); ``load_features_from_npy`` is host logic and returns
CPU tensors exactly like the reference's function."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import torch

from . import _lib

SUPPORTED = {'i3d_features', 'vggish_features'}


def _lib_():
    return _lib.load()


# ---------------------------------------------------------------- datasets/load_features.py, function by function
def fill_missing_features(method, feature_size):
    '''datasets/load_features.py:8-12'''
    if method == 'random':
        return torch.rand(1, feature_size)
    elif method == 'zero':
        return torch.zeros(1, feature_size).float()


def crop_rows(S, start, end, duration):
    '''the index arithmetic of crop_a_segment (datasets/load_features.py:14-36): rows [start_idx, end_idx) of an S-row
    feature for the segment [start, end] of a video of ``duration`` seconds, or None when that slice is empty'''
    start_quantile = start / duration
    end_quantile = end / duration
    start_idx = int(S * start_quantile)
    end_idx = int(S * end_quantile)
    if start_idx == end_idx:        # a segment shorter than one feature step: one row, [S:S] -> [S-1:S] at the very end
        if start_idx == S:
            start_idx -= 1
        else:
            end_idx += 1
    # python slice semantics on an S-row array (negative or oversized indices, reversed ranges)
    lo, hi, _ = slice(start_idx, end_idx).indices(S)
    return (lo, hi) if hi > lo else None


def crop_a_segment(feature, start, end, duration):
    '''datasets/load_features.py:14-36'''
    r = crop_rows(feature.shape[0], start, end, duration)
    return None if r is None else feature[r[0]:r[1], :]


def pad_segment(feature, max_feature_len, pad_idx):
    '''datasets/load_features.py:38-44'''
    S, D = feature.shape
    assert S <= max_feature_len
    out = feature.new_full((max_feature_len, D), pad_idx)
    out[:S] = feature
    return out


def npy_shape(path):
    '''(rows, cols) of a .npy file from its header; FileNotFoundError when it cannot be opened'''
    r, c = C.c_int64(), C.c_int64()
    rc = _lib_().bmt_npy_shape(os.fsencode(path), C.byref(r), C.byref(c), None)
    if rc == _lib.ENOENT:
        raise FileNotFoundError(path)
    _lib.check(rc, "bmt_npy_shape")
    return r.value, c.value


def read_rows(path, row0=0, row1=-1, out=None):
    '''rows [row0, row1) of a .npy file as an fp32 CPU tensor (or into ``out``, a contiguous fp32 tensor)'''
    lib = _lib_()
    r, c = C.c_int64(), C.c_int64()
    bpath = os.fsencode(path)
    if out is None:
        rc = lib.bmt_npy_read_rows(bpath, row0, row1, None, 0, C.byref(r), C.byref(c))
        if rc == _lib.ENOENT:
            raise FileNotFoundError(path)
        _lib.check(rc, "bmt_npy_read_rows")
        out = torch.empty(r.value, c.value, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and not out.is_cuda
    rc = lib.bmt_npy_read_rows(bpath, row0, row1, C.c_void_p(out.data_ptr()), out.numel(), C.byref(r), C.byref(c))
    if rc == _lib.ENOENT:
        raise FileNotFoundError(path)
    _lib.check(rc, "bmt_npy_read_rows")
    return out.view(-1)[:r.value * c.value].view(r.value, c.value)


def _paths(cfg, video_id):
    return {'audio': os.path.join(cfg.audio_features_path, f'{video_id}.npy') if getattr(cfg, 'audio_features_path', None) else None,
            'rgb': os.path.join(cfg.video_features_path, f'{video_id}_rgb.npy') if getattr(cfg, 'video_features_path', None) else None,
            'flow': os.path.join(cfg.video_features_path, f'{video_id}_flow.npy') if getattr(cfg, 'video_features_path', None) else None}


def _load_one(path, start, end, duration, get_full_feat):
    '''cropped (or full) rows of one file; None when the crop is empty'''
    if get_full_feat:
        return read_rows(path)
    S, _ = npy_shape(path)
    r = crop_rows(S, start, end, duration)
    return None if r is None else read_rows(path, r[0], r[1])


def load_features_from_npy(cfg, feature_names_list, video_id, start, end, duration, pad_idx, get_full_feat=False):
    '''datasets/load_features.py:47-95, same arguments and the same dict of CPU tensors; only the rows that are kept are read'''
    assert isinstance(feature_names_list, list)
    assert len(feature_names_list) > 0
    assert set(feature_names_list).issubset(SUPPORTED)
    p = _paths(cfg, video_id)
    stacks = {}
    if get_full_feat:
        stacks['orig_feat_length'] = {}

    if 'vggish_features' in feature_names_list:
        try:
            stack_vggish = _load_one(p['audio'], start, end, duration, get_full_feat)
            if get_full_feat:
                stacks['orig_feat_length']['audio'] = stack_vggish.shape[0]
                stack_vggish = pad_segment(stack_vggish, cfg.pad_feats_up_to['audio'], pad_idx)
        except FileNotFoundError:
            stack_vggish = None
        stacks['audio'] = stack_vggish
    if 'i3d_features' in feature_names_list:
        try:
            stack_rgb = _load_one(p['rgb'], start, end, duration, get_full_feat)
            stack_flow = _load_one(p['flow'], start, end, duration, get_full_feat)
            if get_full_feat:
                assert stack_rgb.shape == stack_flow.shape
                stacks['orig_feat_length']['rgb'] = stack_rgb.shape[0]
                stacks['orig_feat_length']['flow'] = stack_flow.shape[0]
                stack_rgb = pad_segment(stack_rgb, cfg.pad_feats_up_to['video'], pad_idx)
                stack_flow = pad_segment(stack_flow, cfg.pad_feats_up_to['video'], 0)
        except FileNotFoundError:
            stack_rgb = None
            stack_flow = None
        stacks['rgb'] = stack_rgb
        stacks['flow'] = stack_flow
    return stacks


# ---------------------------------------------------------------- the batch path
class _Pending:
    __slots__ = ("future",)


class FeatureIngest:
    """Batches of features on the device, one batch ahead of the consumer.

        ing = FeatureIngest(cfg, ['i3d_features', 'vggish_features'], pad_idx, device)
        nxt = ing.submit([(video_id, start, end, duration), ...])      # returns at once
        ... train on the current batch ...
        feature_stacks = ing.result(nxt)                               # {'rgb','flow','audio'} padded, on the device

    Captioning batches (``get_full_feat=False``): every sample is cropped to its segment and the batch is padded to its longest
    sample (pad_sequence); a missing file or an empty crop becomes one zero row (captioning_dataset.py:238-248).  Proposal
    batches (``get_full_feat=True``): whole videos padded to cfg.pad_feats_up_to (proposal_dataset.py:69-86); ``result``
    then also returns 'orig_feat_length'.  ``items`` entries are (video_id, start, end, duration); the last three are ignored
    for full features."""

    KEYS = (('rgb', 'i3d_features', 'video'), ('flow', 'i3d_features', 'video'), ('audio', 'vggish_features', 'audio'))

    def __init__(self, cfg, feature_names_list, pad_idx, device, get_full_feat=False, workers=16):
        assert set(feature_names_list).issubset(SUPPORTED) and len(feature_names_list) > 0
        self.cfg, self.names, self.pad_idx, self.full = cfg, list(feature_names_list), pad_idx, get_full_feat
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("FeatureIngest needs a GPU: the padded batch is produced by a HIP kernel (no CPU path)")
        self.lib = _lib_()
        self.io = ThreadPoolExecutor(max_workers=workers)
        self.driver = ThreadPoolExecutor(max_workers=1)
        self.stream = torch.cuda.Stream(device=self.device)
        self._staging = [None, None]      # two pinned buffers: one in flight, one being filled
        self._turn = 0
        self._inflight = [None, None]     # the event after which a staging buffer may be overwritten

    # -- host side -------------------------------------------------------------------------------------------------
    def _plan(self, item, key):
        """(path or None, row0, row1) of one sample's rows for one feature key"""
        video_id, start, end, duration = item
        path = _paths(self.cfg, video_id)[key]
        try:
            S, D = npy_shape(path)
        except FileNotFoundError:
            if self.full:
                raise
            return None, 0, 1, None
        if self.full:
            return path, 0, S, D
        r = crop_rows(S, start, end, duration)
        return (None, 0, 1, D) if r is None else (path, r[0], r[1], D)

    def _read(self, plan, dst):
        path, r0, r1, _ = plan
        if path is None:
            dst.zero_()                # fill_missing_features('zero', D): one zero row
        else:
            read_rows(path, r0, r1, out=dst)

    def _stage(self, nfloats):
        i = self._turn
        self._turn ^= 1
        if self._inflight[i] is not None:
            self._inflight[i].synchronize()        # the H2D copy that read this buffer two batches ago is done
        buf = self._staging[i]
        if buf is None or buf.numel() < nfloats:
            buf = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, pin_memory=True)
            self._staging[i] = buf
        return i, buf

    def _run(self, items):
        keys = [k for k in self.KEYS if k[1] in self.names]
        plans = {k[0]: list(self.io.map(lambda it, kk=k[0]: self._plan(it, kk), items)) for k in keys}
        if 'rgb' in plans and 'flow' in plans:
            # the reference loads the two i3d stacks TOGETHER (datasets/load_features.py:70-93): if either file is missing or
            # its crop is empty BOTH become the single zero row, and the stacks must have equal shapes
            for i, (pr, pf) in enumerate(zip(plans['rgb'], plans['flow'])):
                if pr[0] is None or pf[0] is None:
                    D = pr[3] if pr[3] is not None else pf[3]
                    plans['rgb'][i] = plans['flow'][i] = (None, 0, 1, D)
                else:
                    assert pr[2] - pr[1] == pf[2] - pf[1], f"rgb / flow stacks of {items[i][0]} differ in length"
        B = len(items)
        layout, total = {}, 0
        for key, _, mod in keys:
            D = next((p[3] for p in plans[key] if p[3] is not None), None)
            if D is None:
                D = {'rgb': self.cfg.d_vid, 'flow': self.cfg.d_vid, 'audio': self.cfg.d_aud}[key]
            lens = [p[2] - p[1] for p in plans[key]]
            offs = [0]
            for n in lens:
                offs.append(offs[-1] + n)
            T = self.cfg.pad_feats_up_to[mod] if self.full else max(lens)
            assert max(lens) <= T, f"{key}: {max(lens)} rows exceed pad_feats_up_to={T}"
            layout[key] = (total, D, offs, T, lens)
            total += offs[-1] * D
        slot, stage = self._stage(total)
        jobs = []
        for key, (base, D, offs, T, lens) in layout.items():
            for b, plan in enumerate(plans[key]):
                dst = stage[base + offs[b] * D: base + offs[b + 1] * D].view(lens[b], D)
                jobs.append(self.io.submit(self._read, plan, dst))
        for j in jobs:
            j.result()
        # -- device side: one copy of the packed rows, one pad kernel per feature
        out = {}
        with torch.cuda.stream(self.stream):
            dev = torch.empty(total, dtype=torch.float32, device=self.device)
            dev.copy_(stage[:total], non_blocking=True)
            for key, (base, D, offs, T, lens) in layout.items():
                o = torch.tensor(offs, dtype=torch.int64).pin_memory().to(self.device, non_blocking=True)
                y = torch.empty(B, T, D, dtype=torch.float32, device=self.device)
                pad = 0.0 if key == 'flow' else float(self.pad_idx)
                _lib.check(self.lib.bmt_pad_batch(C.c_void_p(dev[base:].data_ptr()), C.c_void_p(o.data_ptr()), B, T, D, pad,
                                                  C.c_void_p(y.data_ptr()), C.c_void_p(self.stream.cuda_stream)), "bmt_pad_batch")
                out[key] = y
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._inflight[slot] = ev
        if self.full:
            out['orig_feat_length'] = {key: layout[key][4] for key in layout}
        return out, ev, (dev,)

    # -- API -------------------------------------------------------------------------------------------------------
    def submit(self, items):
        p = _Pending()
        p.future = self.driver.submit(self._run, list(items))
        return p

    def result(self, pending):
        out, ev, keep = pending.future.result()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in list(out.values()) + list(keep):
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
        return out

    def __call__(self, items):
        return self.result(self.submit(items))

    def close(self):
        self.driver.shutdown(wait=True)
        self.io.shutdown(wait=True)
