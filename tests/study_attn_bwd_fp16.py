"""CPU study (not a test; `python tests/study_attn_bwd_fp16.py`): numerics of the planned attention backward on fp16 MFMAs with power-of-two
gradient scales (bmt_amd/csrc/exp/attn_bwd32.hip) against today's bf16 backward, both emulated in float64
with the operand roundings of the respective kernels, and against exact float64 autograd.

today (attn_bwd_dq16b / dkv32):  q, k, v = bf16(fp16 planes) -- so S is NOT the S the forward normalised with its lse --, dO bf16, P fp32,
                                  dS rounded to bf16, mean-key correction on dQ.
planned:                          S from the fp16 planes (the forward's S), dO' = fp16(bf16(dO) 2^k(q)), dP' against fp16 v, dS' = fp16
                                  (clamped) -> dQ = (K^T dS' - rs' kmean) 2^-k;  dK from dS'' = fp16(bf16(P) (dP - delta) scale 2^g), g = min_q k(q);
                                  dV from bf16(dO)^T bf16(P).
The upstream gradient spans five decades from row to row, K has a common component, valid lengths are ragged."""
import math

import torch

torch.manual_seed(0)
f64 = torch.float64


def rnd(x, dt):
    return x.to(dt).to(f64)


def study(B=2, H=2, Sq=200, Sk=333, dk=256):
    D = H * dk
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, Sq, D, generator=g).half()
    k = (torch.randn(B, Sk, D, generator=g) + 0.5).half()
    v = torch.randn(B, Sk, D, generator=g).half()
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    lens[0] = Sk
    mask = torch.arange(Sk)[None, :] < lens[:, None]                      # [B, Sk]
    do = torch.randn(B, Sq, D, generator=g) * 10.0 ** (-1.0 - 5.0 * torch.rand(B, Sq, 1, generator=g))
    scale = 1.0 / math.sqrt(dk)
    split = lambda x: x.view(x.shape[0], x.shape[1], H, dk).transpose(1, 2)          # [B, H, S, dk]
    m4 = mask[:, None, None, :]

    # ---- exact reference on the fp16 values
    qd, kd, vd = (x.to(f64).requires_grad_(True) for x in (q, k, v))
    s = (split(qd) @ split(kd).transpose(-1, -2)) * scale
    s = s.masked_fill(~m4, float("-inf"))
    o = torch.softmax(s, -1) @ split(vd)
    (o * split(do.to(f64))).sum().backward()
    ref = {"dq": split(qd.grad), "dk": split(kd.grad), "dv": split(vd.grad)}
    lse = torch.logsumexp(s.detach(), -1)                                   # what the forward left (fp16 operands, fp32 statistics)
    delta = (split(do.to(f64)) * o.detach()).sum(-1)                         # rowsum(dO * O)

    qh, kh, vh = split(q.to(f64)), split(k.to(f64)), split(v.to(f64))
    dob = split(rnd(do, torch.bfloat16))                                     # the bf16 plane of dO both versions read
    valid = mask.to(f64)[:, None, :, None]
    kmean = (kh * valid).sum(2, keepdim=True) / valid.sum(2, keepdim=True)   # [B, H, 1, dk]

    def today():
        qb, kb, vb = rnd(qh, torch.bfloat16), rnd(kh, torch.bfloat16), rnd(vh, torch.bfloat16)
        p = torch.exp((qb @ kb.transpose(-1, -2)) * scale - lse[..., None]).masked_fill(~m4, 0.0)
        dp = dob @ vb.transpose(-1, -2)
        ds = rnd(p * (dp - delta[..., None]) * scale, torch.bfloat16)
        dq = ds @ kb - ds.sum(-1, keepdim=True) * kmean
        dk_ = ds.transpose(-1, -2) @ qb
        dv = rnd(p, torch.bfloat16).transpose(-1, -2) @ dob
        return {"dq": dq, "dk": dk_, "dv": dv}

    def planned():
        p = torch.exp((qh @ kh.transpose(-1, -2)) * scale - lse[..., None]).masked_fill(~m4, 0.0)
        amax = dob.abs().amax(-1, keepdim=True)
        kexp = torch.where(amax > 0, 6 - torch.floor(torch.log2(amax.clamp_min(1e-300))), torch.zeros_like(amax)).clamp(-60, 60)
        up = torch.exp2(kexp)
        dos = rnd(dob * up, torch.float16)                                  # |dO'| < 128
        dps = dos @ vh.transpose(-1, -2)
        dss = rnd((p * (dps - (delta[..., None] * up)) * scale).clamp(-60000, 60000), torch.float16)
        dq = (dss @ kh - dss.sum(-1, keepdim=True) * kmean) / up
        # dK / dV: one scale per (batch, head)
        gup = torch.where(amax > 0, up, torch.full_like(up, float("inf"))).amin(2, keepdim=True)
        gup = torch.where(torch.isfinite(gup), gup, torch.ones_like(gup))
        pb = rnd(p, torch.bfloat16)
        dp = dob @ rnd(vh, torch.bfloat16).transpose(-1, -2)
        ds2 = rnd((pb * (dp - delta[..., None]) * scale * gup).clamp(-60000, 60000), torch.float16)
        dk_ = (ds2.transpose(-1, -2) @ qh) / gup
        dv = pb.transpose(-1, -2) @ dob
        stats = {"max |dO'|": float(dos.abs().max()), "max |dS'|": float(dss.abs().max()), "max |dS''|": float(ds2.abs().max()),
                 "rows of dS' with a subnormal maximum": int((dss.abs().amax(-1) < 6.1e-5).sum())}
        return {"dq": dq, "dk": dk_, "dv": dv}, stats

    a, (b, stats) = today(), planned()
    rel = lambda x, r: float((x - r).norm() / r.norm())
    rowrel = lambda x, r: float(((x - r).norm(dim=-1) / r.norm(dim=-1).clamp_min(1e-300)).max())
    print(f"B{B} H{H} Sq{Sq} Sk{Sk} dk{dk}   (relative error against float64 autograd: overall | worst row)")
    for n in ("dq", "dk", "dv"):
        print(f"  {n}: today {rel(a[n], ref[n]):.2e} | {rowrel(a[n], ref[n]):.2e}    planned {rel(b[n], ref[n]):.2e} | {rowrel(b[n], ref[n]):.2e}")
    print("  " + ", ".join(f"{k_} {v_:.3g}" if isinstance(v_, float) else f"{k_} {v_}" for k_, v_ in stats.items()))
    return a, b, ref


if __name__ == "__main__":
    study()
    study(B=2, H=4, Sq=64, Sk=800, dk=256)
    study(B=2, H=8, Sq=300, Sk=300, dk=128)
