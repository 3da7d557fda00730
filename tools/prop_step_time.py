#!/usr/bin/env python
"""step time of train_prop at BASELINE configs[3] (B=16, T_v=1024, T_a=3200, frozen encoder of the configs[1] width); eager launches"""
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmt_amd import ops, synthetic as syn
from bmt_amd.model.proposal_generator import MultimodalProposalGenerator
from bmt_amd.train import ProposalTrainStep
B, Tv, Ta = 16, 1024, 3200
cfg = syn.cfg_config1(procedure="train_prop", dout_p=0.1, lr=1e-4); cfg.device = "cuda"; cfg.grad_clip = None
anchors = {"audio": syn.make_anchors(cfg.anchors_num_audio), "video": syn.make_anchors(cfg.anchors_num_video)}
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = MultimodalProposalGenerator(cfg, anchors).to("cuda")
for frozen in (True, False):
    for p in model.encoder.parameters():
        p.requires_grad = not frozen
    batch = syn.make_prop_batch(cfg, B, Tv, Ta, seed=11)
    fs = {k: v.to("cuda") for k, v in batch["feature_stacks"].items()}
    tg = batch["targets"].to("cuda")
    step = ProposalTrainStep(model, cfg, syn.PAD_IDX)
    for _ in range(3): step(fs, tg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(fs, tg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"train_prop configs[3] B={B} T_v={Tv} T_a={Ta} encoder {'frozen' if frozen else 'fine-tuned'}: {dt*1e3:.1f} ms/step = {B/dt:.0f} videos/s", flush=True)
