import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools", "probes"))
from bmt_amd import _lib, ops
from bmt_amd._lib import AttnBwdBf16Args
EXP = C.CDLL("bmt_amd/lib/libbmt_exp.so")
EXP.bmt_exp_attn_bwd_split.restype = C.c_int
EXP.bmt_exp_attn_bwd_split.argtypes = [C.POINTER(AttnBwdBf16Args), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
_p, _st = ops._p, ops._st
dev = "cuda"
B, H, Sq, Sk, dk = 2, 4, 200, 200, 256
D = H * dk
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)
mk = lambda S: ops.make_planes((rnd(B * S, D) * 0.5).to(dev), "f16")
q, k, v = mk(Sq), mk(Sk), mk(Sk)
mask = torch.ones(B, 1, Sk, dtype=torch.bool); mask[0, 0, Sk - 7:] = False
md = mask.to(dev)
o, lse = ops.attn_fwd_planes(q, k, v, B, Sq, Sk, D, md, H, precision=ops.PREC_F16, out_fmt="f16")
do = ops.make_planes(rnd(B * Sq, D).to(dev), "bwd"); do = ops.Planes(do.hi[:, :D].contiguous(), None, B * Sq, D)
f16 = lambda pl: ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh)
res = {}
for split in (False, True):
    ops.ATTN_BWD_SPLIT = split
    r = ops.attn_bwd_planes(f16(q), f16(k), f16(v), o, do, lse, B, Sq, Sk, D, md, H, 0.0, (None, None, None))
    torch.cuda.synchronize()
    res[split] = [pl.hi[:, :D].float().clone() for pl, _ in r[:3]]
for n, a, b in zip(("dq", "dk", "dv"), res[False], res[True]):
    e = (a - b).norm() / a.norm()
    print(n, "product split vs two-kernel:", float(e), "finite", bool(torch.isfinite(b).all()))
    if e > 0.05:
        bad = ((a - b).abs() > 0.05 * a.abs().max()).nonzero()
        print("   bad count", len(bad), "first", bad[:5].tolist(), "last", bad[-5:].tolist())
        rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
        print("   bad rows", rows[:20].tolist(), "... n", len(rows), " bad cols n", len(cols), cols[:10].tolist())
