#!/usr/bin/env python
"""How many query rows of the attention backward's dO are exactly zero in a real config[1] step (the padded positions of the encoder), per
launch; and what the live-query shortcut of the split backward buys on the audio self-attention when they are (same launch, dO with and
without a zero suffix).   usage: python tools/probes/zero_rows.py"""
import contextlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bmt_amd import ops, synthetic as syn  # noqa: E402
from bmt_amd.model.captioning_module import BiModalTransformer  # noqa: E402
from bmt_amd.train import CaptioningTrainStep  # noqa: E402

dev = torch.device("cuda", 0)
V, Tv, Ta, Tc, B = 10000, 256, 800, 30, 32
cfg = syn.cfg_config1(dout_p=0.1)
cfg.device = str(dev)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)
batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=1234)
fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}
caps = batch["captions"].to(dev)
step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, seed=1000)
ops.ENC_STREAMS = 1
for _ in range(2):
    step(fs, caps)
raw = ops.attn_bwd_planes
log = []


def spy(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw):
    d = do.hi if isinstance(do, ops.Planes) else do
    z = (d.view(B_ * Sq, -1).float().abs().amax(-1) == 0)
    zb = z.view(B_, Sq)
    # zero rows that form a suffix of their sequence
    suffix = sum(int(zb[b].flip(0).cumprod(0).sum()) for b in range(B_))
    log.append((Sq, Sk, int(z.sum()), B_ * Sq, suffix))
    return raw(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw)


ops.attn_bwd_planes = spy
step(fs, caps)
torch.cuda.synchronize()
ops.attn_bwd_planes = raw
lens = (fs["rgb"][:, :, 0] != 1.0).sum(1)
print("valid video lengths:", lens.tolist())
for Sq, Sk, z, n, suf in log:
    print(f"attention backward Sq {Sq:4d} Sk {Sk:4d}: {z:6d} of {n} dO rows exactly zero ({z / n:.1%}), {suf} of them a suffix of their sequence")

# ---- kernel-level: the audio self-attention's backward with and without a zero suffix in dO
H, D, Sq, Sk = 4, 1024, 800, 800
g = torch.Generator().manual_seed(1)
mk = lambda rows: ops.make_planes((torch.randn(rows, D, generator=g) * 0.5).to(dev), "f16")
q, k, v = mk(B * Sq), mk(B * Sk), mk(B * Sk)
La = torch.round(lens.float() * Ta / Tv).long().clamp(max=Ta)
mask = (torch.arange(Sk, device=dev)[None, :] < La[:, None]).view(B, 1, Sk)
f16 = lambda pl: ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh)
o, lse = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, precision=ops.PREC_F16, out_fmt="f16")
do = torch.randn(B, Sq, D, generator=g).to(dev)
do_z = do.clone()
for b in range(B):
    do_z[b, int(La[b]):] = 0
for name, d in (("dense dO", do), ("zero suffix", do_z)):
    dp = ops.make_planes(d.view(B * Sq, D), "bwd")
    dp = ops.Planes(dp.hi[:, :D].contiguous(), None, B * Sq, D)
    for _ in range(3):
        ops.attn_bwd_planes(f16(q), f16(k), f16(v), o, dp, lse, B, Sq, Sk, D, mask, H, 0.0, (None, None, None))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.attn_bwd_planes(f16(q), f16(k), f16(v), o, dp, lse, B, Sq, Sk, D, mask, H, 0.0, (None, None, None))
    e.record()
    torch.cuda.synchronize()
    print(f"A-self backward, {name}: {s.elapsed_time(e) / 20 * 1e3:.1f} us per launch")
