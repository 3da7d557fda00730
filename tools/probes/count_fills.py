#!/usr/bin/env python
"""which call sites zero-fill device buffers in one eagerly issued train_cap step (configs[1]): torch.zeros / Tensor.zero_ / fill_ are
wrapped and tallied by (caller, shape).  usage: python tools/probes/count_fills.py"""
import collections
import contextlib
import io
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bmt_amd import synthetic as syn  # noqa: E402
from bmt_amd.model.captioning_module import BiModalTransformer  # noqa: E402
from bmt_amd.train import CaptioningTrainStep  # noqa: E402

dev = "cuda"
cfg = syn.cfg_config1(dout_p=0.1)
cfg.device = dev
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = BiModalTransformer(cfg, syn.FakeTrainDataset(10000, syn.make_glove(10000, cfg.d_model_caps))).to(dev)
batch = syn.make_cap_batch(cfg, 32, 256, 800, 30, 10000, seed=1234)
fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}
caps = batch["captions"].to(dev)
step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
for _ in range(2):
    step(fs, caps)
torch.cuda.synchronize()
tally = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "bmt_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


raw_zeros, raw_zero_, raw_cat, raw_clone = torch.zeros, torch.Tensor.zero_, torch.cat, torch.Tensor.clone


def zeros(*a, **k):
    t = raw_zeros(*a, **k)
    if t.is_cuda:
        tally[("zeros", where(), tuple(t.shape))] += 1
    return t


def zero_(self):
    if self.is_cuda:
        tally[("zero_", where(), tuple(self.shape))] += 1
    return raw_zero_(self)


def cat(ts, *a, **k):
    r = raw_cat(ts, *a, **k)
    if r.is_cuda:
        tally[("cat", where(), tuple(r.shape))] += 1
    return r


torch.zeros, torch.Tensor.zero_, torch.cat = zeros, zero_, cat
step(fs, caps)
torch.cuda.synchronize()
torch.zeros, torch.Tensor.zero_, torch.cat = raw_zeros, raw_zero_, raw_cat
for (kind, w, shape), n in sorted(tally.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {kind:6s} {str(shape):22s} {w}")
print("total", sum(tally.values()))

# what the wrappers cannot see (fills issued from C++: autograd materialising undefined gradients, ones_like of the loss ...): the profiler's view
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(fs, caps)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add"):
        st = [f for f in (ev.stack or []) if "bmt_amd" in f]
        by[(ev.name, st[0] if st else "(no python frame: autograd engine)", str(ev.input_shapes[:1]))] += 1
for (name, w, shp), n in sorted(by.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:3d} x {name:12s} {shp:28s} {w}")
