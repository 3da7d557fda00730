#!/usr/bin/env python
"""BASELINE configs[4] per-GPU slice: train_cap N=6, d_model=1024, H=8 (d_k=128), B=64 per GPU; a few captured steps"""
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmt_amd import ops, synthetic as syn
from bmt_amd.model.captioning_module import BiModalTransformer
from bmt_amd.train import CaptioningTrainStep
dev = torch.device("cuda", 0)
cfg = syn.make_cfg(d_model=1024, H=8, N=6, dout_p=0.1); cfg.device = str(dev)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = BiModalTransformer(cfg, syn.FakeTrainDataset(10000, syn.make_glove(10000, cfg.d_model_caps))).to(dev)
B = 64
batch = syn.make_cap_batch(cfg, B, 256, 800, 30, 10000, seed=1234)
fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}; caps = batch["captions"].to(dev)
tokens = int((caps[:, 1:] != syn.PAD_IDX).sum())
step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True)
step.capture(fs, caps)
for _ in range(2): loss, _ = step.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): loss, _ = step.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"configs[4] slice (N=6 H=8 B={B}): {dt*1e3:.1f} ms/step = {tokens/dt:.0f} caption tokens/s, loss {float(loss):.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
