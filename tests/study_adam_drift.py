"""CPU study (not a test; ``python tests/study_adam_drift.py``): how far 10 Adam steps of the fp32 oracle drift when the gradients
carry noise -- relative per element (never flips a sign) and absolute at a fraction of the tensor's rms (what operand rounding
leaves on cancellation-dominated elements).  Adam's update lr * m / sqrt(v) moves every weight by ~lr in the first steps, so
the trajectory is chaotic in the gradient's low bits: the reason tests/test_gpu_model.py::test_ten_adam_steps_against_the_oracle
bounds the loss trajectory and checks the forward at the trained weights instead of asking two trajectories to agree to 1e-3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bmt_amd import synthetic as syn
from oracle import bmt_oracle as orc
torch.set_num_threads(8)
z=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mid_cap.npz')); V,B,Tv,Ta,Tc,seed,_=[int(x) for x in z['meta']]
cfg=syn.cfg_config1(dout_p=0.0)
batch=syn.make_cap_batch(cfg,B,Tv,Ta,Tc,V,seed=seed)
def run(noise, steps=10, gseed=0):
    sd=orc.init_captioning_params(cfg,V,seed=0,glove=syn.make_glove(V,cfg.d_model_caps))
    p={k:v.clone().requires_grad_(k!='emb_C.embedder.weight') for k,v in sd.items()}
    m={k:torch.zeros_like(v) for k,v in p.items()}; v2={k:torch.zeros_like(v) for k,v in p.items()}
    g=torch.Generator().manual_seed(gseed); losses=[]
    for it in range(1,steps+1):
        for t in p.values(): t.grad=None
        loss,_,_=orc.train_cap_loss(p,cfg,batch['feature_stacks'],batch['captions'],1,cfg.smoothing)
        loss.backward(); losses.append(float(loss))
        with torch.no_grad():
            for k,t in p.items():
                if t.grad is None: continue
                gr=t.grad
                if noise>0: gr=gr*(1+noise*torch.randn(gr.shape,generator=g))
                orc.adam_step(t,gr,m[k],v2[k],it,cfg.lr)
    with torch.no_grad():
        _,pred,_=orc.train_cap_loss(p,cfg,batch['feature_stacks'],batch['captions'],1,cfg.smoothing)
    return losses,pred
l0,p0=run(0.0)
for noise in (1e-3,1e-2,3e-2):
    l1,p1=run(noise)
    print('rel noise',noise,'max dlogp',float((p1-p0).abs().max()),'loss diffs',[f'{a-b:.1e}' for a,b in zip(l1,l0)][:10])
# absolute-noise model: noise relative to tensor rms (cancellation-dominated elements)
def run_abs(frac):
    sd=orc.init_captioning_params(cfg,V,seed=0,glove=syn.make_glove(V,cfg.d_model_caps))
    p={k:v.clone().requires_grad_(k!='emb_C.embedder.weight') for k,v in sd.items()}
    m={k:torch.zeros_like(v) for k,v in p.items()}; v2={k:torch.zeros_like(v) for k,v in p.items()}
    g=torch.Generator().manual_seed(1); losses=[]
    for it in range(1,11):
        for t in p.values(): t.grad=None
        loss,_,_=orc.train_cap_loss(p,cfg,batch['feature_stacks'],batch['captions'],1,cfg.smoothing)
        loss.backward(); losses.append(float(loss))
        with torch.no_grad():
            for k,t in p.items():
                if t.grad is None: continue
                gr=t.grad+frac*t.grad.pow(2).mean().sqrt()*torch.randn(t.grad.shape,generator=g)
                orc.adam_step(t,gr,m[k],v2[k],it,cfg.lr)
    with torch.no_grad():
        _,pred,_=orc.train_cap_loss(p,cfg,batch['feature_stacks'],batch['captions'],1,cfg.smoothing)
    return losses,pred
for frac in (1e-3,1e-2,3e-2):
    l1,p1=run_abs(frac)
    print('abs noise (x tensor rms)',frac,'max dlogp',float((p1-p0).abs().max()),'loss diffs',[f'{a-b:.1e}' for a,b in zip(l1,l0)][:10])
