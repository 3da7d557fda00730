import sys, torch
sys.path.insert(0, '/root/repo')
from bmt_amd import ops
dev='cuda'
for (B,H,Sq,Sk,dk) in ((2,4,200,200,256),(2,4,12,200,256),(2,8,70,130,128)):
    D=H*dk
    mk=lambda S,s: ops.make_planes(torch.randn(B*S, D, device=dev, generator=torch.Generator(device=dev).manual_seed(s))*0.5, "f16")
    q,k,v=mk(Sq,1),mk(Sk,2),mk(Sk,3)
    mask=torch.ones(B,1,Sk,dtype=torch.bool,device=dev); mask[0,0,Sk-7:]=False
    o,lse=ops.attn_fwd_planes(q,k,v,B,Sq,Sk,D,mask,H,precision=ops.PREC_F16,out_fmt="f16")
    do=ops.make_planes(torch.randn(B*Sq, D, device=dev), "bwd"); do=ops.Planes(do.hi[:, :D].contiguous(), None, B*Sq, D)
    r1=ops.attn_bwd_planes(q.only("hi"),k.only("hi"),v.only("hi"),o,do,lse,B,Sq,Sk,D,mask,H,0.0,(None,None,None))
    f=lambda pl: ops.Planes(None,None,pl.rows,pl.cols,fh=pl.fh)
    r2=ops.attn_bwd_planes(f(q),f(k),f(v),o,do,lse,B,Sq,Sk,D,mask,H,0.0,(None,None,None))
    for n,(a,_),(b,_) in zip("qkv",r1[:3],r2[:3]):
        A,Bt=a.hi.float(),b.hi.float()
        print((B,H,Sq,Sk,dk),'d'+n,'rel diff', float((A-Bt).norm()/A.norm()), 'norms', float(A.norm()), float(Bt.norm()))
B,H,Sk,dk=2,4,200,256; D=H*dk
kp=ops.make_planes(torch.randn(B*Sk, D, device=dev)*0.5+0.3, "f16")
ma=ops._mask_args(None,B,1,Sk)
a=ops.attn_kmean(kp.hi, D, Sk*D, B, Sk, D, ma, f16=False); b=ops.attn_kmean(kp.fh, D, Sk*D, B, Sk, D, ma, f16=True)
print('kmean hi vs fh->bf16:', float((a-b).abs().max()), float(a.abs().max()))
