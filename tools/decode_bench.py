#!/usr/bin/env python
"""Greedy decoding at config[1] sizes (B=32, T_v=256, T_a=800, V=10000, max_len=30): bmt_amd.decode.greedy_decoder with
encoder + K/V reuse against the reference's loop shape (one full forward per token), both on the HIP path.
end_idx = -1 so that every call generates exactly max_len tokens."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import synthetic as syn  # noqa: E402
from bmt_amd.decode import greedy_decoder  # noqa: E402
from bmt_amd.model.captioning_module import BiModalTransformer  # noqa: E402

B, Tv, Ta, V, MAXLEN = 32, 256, 800, 10000, 30
cfg = syn.cfg_config1()
cfg.device = "cuda"
torch.manual_seed(0)
model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to("cuda").eval()
fs = {k: v.cuda() for k, v in syn.make_cap_batch(cfg, B, Tv, Ta, 4, V, seed=7)["feature_stacks"].items()}
res = {}
toks = {}
for reuse in (True, False):
    for _ in range(2):
        toks[reuse] = greedy_decoder(model, fs, MAXLEN, syn.START_IDX, -1, syn.PAD_IDX, "audio_video", reuse=reuse)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        greedy_decoder(model, fs, MAXLEN, syn.START_IDX, -1, syn.PAD_IDX, "audio_video", reuse=reuse)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res["reuse" if reuse else "full_forward_per_token"] = {"ms_per_batch": dt * 1e3, "captions_per_s": B / dt, "tokens_per_s": B * MAXLEN / dt}
res["same_tokens"] = bool(torch.equal(toks[True], toks[False]))
res["speedup"] = res["full_forward_per_token"]["ms_per_batch"] / res["reuse"]["ms_per_batch"]
res["workload"] = f"greedy decode config[1] B={B} T_v={Tv} T_a={Ta} V={V} max_len={MAXLEN}"
print(json.dumps(res))
