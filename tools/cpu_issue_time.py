import sys, time, torch, io, contextlib
sys.path.insert(0,'.')
from bmt_amd import ops, synthetic as syn
from bmt_amd.model.captioning_module import BiModalTransformer
from bmt_amd.train import CaptioningTrainStep
dev=torch.device('cuda',0)
cfg=syn.cfg_config1(dout_p=0.1); cfg.device=str(dev)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model=BiModalTransformer(cfg, syn.FakeTrainDataset(10000, syn.make_glove(10000,cfg.d_model_caps))).to(dev)
batch=syn.make_cap_batch(cfg,32,256,800,30,10000,seed=1234)
fs={k:v.to(dev) for k,v in batch['feature_stacks'].items()}; caps=batch['captions'].to(dev)
step=CaptioningTrainStep(model,cfg,syn.PAD_IDX,static_grads=True)
for _ in range(3): step(fs,caps)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): step(fs,caps)
t_issue=time.perf_counter()-t0
torch.cuda.synchronize()
t_all=time.perf_counter()-t0
print(f"eager: CPU issue {t_issue/10*1e3:.2f} ms/step, wall {t_all/10*1e3:.2f} ms/step")
