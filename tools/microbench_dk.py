import sys, torch
sys.path.insert(0, '.')
from bmt_amd import ops
DEV='cuda'
def timeit(fn, flops, name, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us=s.elapsed_time(e)*1e3/iters
    print(f"{name:50s} {us:9.1f} us {flops/us/1e6:8.1f} TFLOP/s", flush=True)
B,S,D=32,800,1024
for H in (4,8,16,32):
    dk=D//H
    q=torch.randn(B,S,D,device=DEV);k=torch.randn(B,S,D,device=DEV);v=torch.randn(B,S,D,device=DEV)
    pl=lambda t:(t.to(torch.bfloat16),(t-t.to(torch.bfloat16).float()).to(torch.bfloat16))
    (qh,ql),(kh,kl),(vh,vl)=pl(q),pl(k),pl(v)
    fl=4.0*B*S*S*D
    for prec in (3,1):
        timeit(lambda: ops.attn_fwd_bf16(qh,ql,kh,kl,vh,vl,None,H,precision=prec), fl, f"fwd x{prec} dk={dk} H={H}")
    o,lse=ops.attn_fwd_bf16(qh,ql,kh,kl,vh,vl,None,H,precision=3)
    do=torch.randn_like(o)
    timeit(lambda: ops.attn_bwd_bf16(qh,kh,vh,o,do,lse,None,H), 2.5*fl, f"bwd x1 dk={dk} H={H}")
