"""CPU study (not a test; run by hand: ``python tests/study_precision_policy.py``): which MFMA operand formats can each
product site of the captioning forward take while max |d log-prob| vs the fp32 reference stays under the 1e-3 bar of
BASELINE.json's north_star.  Uses the oracle's arithmetic with the matmuls replaced by operand-rounded ones:

    f32      exact fp32 (stands for split-bf16 x3, measured 3e-5 on the GPU)
    bf16     one pass, both operands rounded to bf16
    fp16     one pass, both operands rounded to fp16
    fp16a2   two passes: activation (first operand) as fp16 hi + fp16 lo, second operand fp16
    fp16w2   two passes: second operand as fp16 hi + lo, first operand fp16
    bf16x3   hi.hi + hi.lo + lo.hi with bf16 planes (what the GPU forward ran in round 1)

Sites: {enc,dec}.{proj,projkv,qk,pv,oproj,ffn1,ffn2}, dec.bridge, gen.  Prints one line per experiment."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import synthetic as syn          # noqa: E402
from oracle import bmt_oracle as orc          # noqa: E402

POLICY = {}
DEFAULT = ["f32"]
CTX = ["enc."]


def rb(x):
    return x.to(torch.bfloat16).to(x.dtype)


def rh(x):
    return x.to(torch.float16).to(x.dtype)


def mm(a, bt, site):
    """a @ bt with the operand rounding of POLICY[site]"""
    pol = POLICY.get(site, DEFAULT[0])
    if pol == "f32":
        return a @ bt
    if pol == "bf16":
        return rb(a) @ rb(bt)
    if pol == "fp16":
        return rh(a) @ rh(bt)
    if pol == "fp16a2":
        ah = rh(a)
        return (ah + rh(a - ah)) @ rh(bt)
    if pol == "fp16w2":
        bh = rh(bt)
        return rh(a) @ (bh + rh(bt - bh))
    if pol == "bf16x3":
        ah, bh = rb(a), rb(bt)
        al, bl = rb(a - ah), rb(bt - bh)
        return ah @ bh + ah @ bl + al @ bh
    if pol == "bf16a2":
        ah = rb(a)
        return (ah + rb(a - ah)) @ rb(bt)
    raise KeyError(pol)


def _linear(x, w, b, site="linear"):
    y = mm(x, w.transpose(-1, -2), (CTX[0] + site) if site != "gen" else "gen")
    return y if b is None else y + b


def attention(Q, K, V, msk):
    d_k = Q.size(-1)
    s = mm(Q, K.transpose(-1, -2), CTX[0] + "qk") / np.sqrt(d_k)
    if msk is not None:
        s = s.masked_fill(msk == 0, -float("inf"))
    return mm(torch.softmax(s, dim=-1), V, CTX[0] + "pv")


def multiheaded_attention(p, prefix, Q, K, V, msk, H):
    B, Sq, _ = Q.shape
    kvsite = "proj" if K is Q else "projkv"
    q = _linear(Q, p[prefix + "linear_Q2d.weight"], p[prefix + "linear_Q2d.bias"], "proj")
    k = _linear(K, p[prefix + "linear_K2d.weight"], p[prefix + "linear_K2d.bias"], kvsite)
    v = _linear(V, p[prefix + "linear_V2d.weight"], p[prefix + "linear_V2d.bias"], kvsite)
    D = q.shape[-1]
    d_k = D // H
    q = q.view(B, -1, H, d_k).transpose(1, 2)
    k = k.view(B, -1, H, d_k).transpose(1, 2)
    v = v.view(B, -1, H, d_k).transpose(1, 2)
    if msk is not None:
        msk = msk.unsqueeze(1)
    o = attention(q, k, v, msk)
    o = o.transpose(1, 2).contiguous().view(B, Sq, D)
    return _linear(o, p[prefix + "linear_d2Q.weight"], p[prefix + "linear_d2Q.bias"], "oproj")


_enc, _dec = orc.bimodal_encoder, orc.bimodal_decoder


def bimodal_encoder(*a, **k):
    CTX[0] = "enc."
    return _enc(*a, **k)


def bimodal_decoder(*a, **k):
    CTX[0] = "dec."
    return _dec(*a, **k)


orc._linear, orc.attention, orc.multiheaded_attention = _linear, attention, multiheaded_attention
orc.bimodal_encoder, orc.bimodal_decoder = bimodal_encoder, bimodal_decoder

ENC = ["enc." + s for s in ("proj", "projkv", "qk", "pv", "oproj", "ffn1", "ffn2")]
DEC = ["dec." + s for s in ("proj", "projkv", "qk", "pv", "oproj", "ffn1", "ffn2", "bridge")]
ALL = ENC + DEC + ["gen"]


def main():
    torch.set_num_threads(8)
    full = "--full" in sys.argv
    cfg = syn.cfg_config1(dout_p=0.0)
    V = 10000 if full else 1000
    B, Tv, Ta, Tc = (2, 256, 800, 30) if full else (2, 64, 200, 12)
    z = np.load(os.path.join(ROOT, "tests", "golden", "mid_cap.npz"))
    if not full:
        V, B, Tv, Ta, Tc, seed, use_glove = [int(x) for x in z["meta"]]
    else:
        seed = 1234
    glove = syn.make_glove(V, cfg.d_model_caps)
    p = orc.init_captioning_params(cfg, V, seed=0, glove=glove)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=seed)
    fs, caps = batch["feature_stacks"], batch["captions"]
    x = caps[:, :-1]
    masks = orc.make_masks(fs, x, syn.PAD_IDX)

    def run(policy, default="f32"):
        POLICY.clear()
        POLICY.update(policy)
        DEFAULT[0] = default
        with torch.no_grad():
            return orc.bimodal_transformer(p, cfg, fs, x, masks)

    ref = run({})
    if not full:
        print("oracle vs fixture:", float((ref - torch.from_numpy(z["pred"])).abs().max()))
    pad_rows = (x == syn.PAD_IDX)

    def err(pred):
        return float((pred - ref).abs().max())

    def show(name, policy, default="f32"):
        e = err(run(policy, default))
        print(f"{name:58s} max|dlogp| = {e:.3e}  {'OK' if e < 5e-4 else ('marginal' if e < 1e-3 else 'FAIL')}", flush=True)
        return e

    for fmt in ("bf16", "fp16", "fp16a2", "fp16w2", "bf16x3"):
        show(f"everything {fmt}", {}, fmt)
    print("--- one site degraded, everything else exact")
    for fmt in ("bf16", "fp16"):
        for s in ALL:
            show(f"only {s} {fmt}", {s: fmt})
    print("--- groups")
    for fmt in ("bf16", "fp16", "fp16a2", "fp16w2"):
        show(f"encoder all {fmt}", {s: fmt for s in ENC})
        show(f"decoder all {fmt}", {s: fmt for s in DEC})
        show(f"encoder+decoder {fmt}, gen exact", {s: fmt for s in ENC + DEC})
    print("--- candidate policies")
    show("enc fp16, dec fp16, gen fp16a2", {**{s: "fp16" for s in ENC + DEC}, "gen": "fp16a2"})
    show("enc fp16, dec fp16a2, gen exact", {**{s: "fp16" for s in ENC}, **{s: "fp16a2" for s in DEC}})
    show("enc fp16, dec exact, gen exact", {s: "fp16" for s in ENC})
    show("enc fp16 GEMM + fp16 attn; dec projkv fp16, rest exact", {**{s: "fp16" for s in ENC}, "dec.projkv": "fp16"})
    show("enc fp16; dec projkv+qk+pv fp16, rest exact", {**{s: "fp16" for s in ENC}, "dec.projkv": "fp16", "dec.qk": "fp16", "dec.pv": "fp16"})
    show("enc bf16 attn only (qk,pv), rest exact", {"enc.qk": "bf16", "enc.pv": "bf16"})
    show("enc fp16 attn only (qk,pv), rest exact", {"enc.qk": "fp16", "enc.pv": "fp16"})


if __name__ == "__main__":
    main()
