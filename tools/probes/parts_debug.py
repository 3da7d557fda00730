"""stage-by-stage run of a train_cap step in parts (faulthandler on): which stage dies, if any.   python tools/probes/parts_debug.py [M] [B] [dropout]"""
import faulthandler, sys, os, contextlib, io
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bmt_amd import ops, synthetic as syn
from bmt_amd.model.captioning_module import BiModalTransformer
from bmt_amd.train import CaptioningTrainStep
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
V, Tv, Ta, Tc = (500, 48, 150, 12) if (len(sys.argv) > 4 and sys.argv[4] == "tiny") else (10000, 256, 800, 30)
dev = torch.device("cuda:0")
cfg = syn.cfg_config1(dout_p=p)
cfg.device = str(dev)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)
batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=1234)
fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}
caps = batch["captions"].to(dev)
step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, static_grads=True, seed=1000, microbatches=M)


def say(*a):
    print(*a, flush=True)


say(f"M={M} B={B} dropout={p}")
for i in range(0 if os.environ.get("PARTS_DEBUG_NO_EAGER") else 3):
    loss, n = step(fs, caps)
    torch.cuda.synchronize()
    say(f"eager step {i}: loss {float(loss):.4f} n {int(n)} mode {getattr(step, '_parts_last', None)}")
step.capture(fs, caps, warmup=2)
torch.cuda.synchronize()
say("captured")
for i in range(3):
    loss, n = step.replay()
    torch.cuda.synchronize()
    say(f"replay {i}: loss {float(loss):.4f}")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for i in range(20):
    step.replay()
ev1.record()
torch.cuda.synchronize()
say(f"replay: {ev0.elapsed_time(ev1) / 20:.3f} ms/step")
