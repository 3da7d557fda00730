"""CPU study (not a test; ``python tests/study_attn_bwd_centering.py``): does subtracting the per-(batch, head) mean key / value
before the bf16 products of the attention backward remove the 10-30 % errors of the decoder's cross-attention dQ / dK?
Captures (q, k, v, mask, dO) of every attention call of one captioning step of the oracle on the mid fixture's shapes, then
evaluates the backward three ways against the fp32 autograd result:
    bf16        operands rounded to bf16 as the HIP kernels do (P, dS, Q, K, V, dO), fp32 accumulation
    centered    the same with K - mean_j K and V - mean_j V (masked mean over the valid keys), delta' = delta - dO . vbar
prints the relative error of dQ, dK, dV per attention site."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import synthetic as syn          # noqa: E402
from oracle import bmt_oracle as orc          # noqa: E402

REC = []
_att = orc.attention


def attention(Q, K, V, msk, *a, **kw):
    out = _att(Q, K, V, msk, *a, **kw)
    rec = {"q": Q.detach(), "k": K.detach(), "v": V.detach(), "m": None if msk is None else msk.detach()}
    out.register_hook(lambda g, rec=rec: rec.__setitem__("do", g.detach()))
    REC.append(rec)
    return out


orc.attention = attention


def rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def bwd(q, k, v, m, do, mode):
    """returns dq, dk, dv; q,k,v: (B,H,S,d)"""
    d = q.shape[-1]
    scale = 1.0 / np.sqrt(d)
    valid = torch.ones(k.shape[:-1], dtype=torch.bool) if m is None else None
    if mode == "fp32":
        s = (q @ k.transpose(-1, -2)) * scale
        if m is not None:
            s = s.masked_fill(m == 0, -float("inf"))
        p = torch.softmax(s, -1)
        o = p @ v
        dp = do @ v.transpose(-1, -2)
        delta = (do * o).sum(-1, keepdim=True)
        ds = p * (dp - delta) * scale
        return ds @ k, ds.transpose(-1, -2) @ q, p.transpose(-1, -2) @ do
    # forward statistics in fp32 (the kernels keep lse and O from the forward)
    s = (q @ k.transpose(-1, -2)) * scale
    if m is not None:
        s = s.masked_fill(m == 0, -float("inf"))
    lse = torch.logsumexp(s, -1, keepdim=True)
    o = torch.softmax(s, -1) @ v
    delta = (do * o).sum(-1, keepdim=True)
    kk, vv = k, v
    kbar = None
    if mode in ("centered", "k-only", "v-only", "rowsum"):
        if m is not None and m.shape[-2] == 1:          # key-padding mask (B,1,1,Sk): masked mean over valid keys
            wgt = (m != 0).to(torch.float32).transpose(-1, -2)      # (B,1,Sk,1)
            kbar = (k * wgt).sum(-2, keepdim=True) / wgt.sum(-2, keepdim=True)
            vbar = (v * wgt).sum(-2, keepdim=True) / wgt.sum(-2, keepdim=True)
        else:
            kbar, vbar = k.mean(-2, keepdim=True), v.mean(-2, keepdim=True)
        if mode in ("centered", "k-only"):
            kk = k - kbar
            lse = lse - (q @ kbar.transpose(-1, -2)) * scale
        if mode in ("centered", "v-only"):
            vv = v - vbar
            delta = delta - do @ vbar.transpose(-1, -2)
    qb, kb, vb, dob = rb(q), rb(kk), rb(vv), rb(do)
    s = (qb @ kb.transpose(-1, -2)) * scale
    p = torch.exp(s - lse)
    if m is not None:
        p = p.masked_fill(m == 0, 0.0)
    dp = dob @ vb.transpose(-1, -2)
    ds = p * (dp - delta) * scale
    pb, dsb = rb(p), rb(ds)
    dq = dsb @ kb
    if mode == "rowsum":        # uncentered planes, dQ corrected by (row sum of the ROUNDED dS) x kbar in fp32
        dq = dq - dsb.sum(-1, keepdim=True) * kbar
    return dq, dsb.transpose(-1, -2) @ qb, pb.transpose(-1, -2) @ dob


def main():
    torch.set_num_threads(8)
    cfg = syn.cfg_config1(dout_p=0.0)
    z = np.load(os.path.join(ROOT, "tests", "golden", "mid_cap.npz"))
    V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in z["meta"]]
    p = orc.init_captioning_params(cfg, V, seed=0, glove=syn.make_glove(V, cfg.d_model_caps))
    p = {k: v.clone().requires_grad_(k != "emb_C.embedder.weight") for k, v in p.items()}
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    loss, _, _ = orc.train_cap_loss(p, cfg, batch["feature_stacks"], batch["captions"], syn.PAD_IDX, cfg.smoothing)
    loss.backward()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    print(f"{len(REC)} attention calls")
    for i, r in enumerate(REC):
        q, k, v, m, do = r["q"], r["k"], r["v"], r["m"], r["do"]
        ref = bwd(q, k, v, m, do, "fp32")
        a = bwd(q, k, v, m, do, "bf16")
        c = bwd(q, k, v, m, do, "centered")
        extra = "  ".join(f"{md} dq {rel(x[0], ref[0]):.3f} dk {rel(x[1], ref[1]):.3f}" for md in ("k-only", "v-only", "rowsum") for x in [bwd(q, k, v, m, do, md)])
        print(f"site {i:2d} Sq={q.shape[2]:4d} Sk={k.shape[2]:4d}: bf16 dq {rel(a[0], ref[0]):.3f} dk {rel(a[1], ref[1]):.3f} dv {rel(a[2], ref[2]):.3f}   "
              f"centered dq {rel(c[0], ref[0]):.3f} dk {rel(c[1], ref[1]):.3f} dv {rel(c[2], ref[2]):.3f}   {extra}", flush=True)


if __name__ == "__main__":
    main()
