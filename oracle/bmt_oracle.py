"""CPU oracle for the BMT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain, functional (no nn.Module in the arithmetic) CPU restatement
of the reference's train_cap / train_prop hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``bmt_amd/`` does (tests/test_no_oracle_in_product.py
enforces this).

Parity status: PINNED.  Every function below is checked (tests/test_oracle_vs_reference.py,
runs only where /root/reference exists) against the reference modules imported
from /root/reference, and (everywhere) against the golden vectors under
tests/golden/ that tests/golden/make_golden.py captured from that import.

All arithmetic is fp32 (the reference's dtype) unless ``dtype=torch.float64``
is requested by the caller for tolerance studies.  Parameters are passed as a
flat ``dict[str, Tensor]`` whose keys are the reference's state_dict keys
(SURVEY.md Appendix A), so a reference checkpoint can be fed in directly.

Each function cites the reference file:line (relative to /root/reference) it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# optional operand-rounding hook (tolerance studies: emulates bf16 MFMA operand rounding)
# --------------------------------------------------------------------------------------
_QUANT = {"fn": None}


def set_operand_quantizer(fn):
    """fn(tensor, site:str)->tensor applied to every GEMM operand; None disables."""
    _QUANT["fn"] = fn


def _q(x: Tensor, site: str) -> Tensor:
    fn = _QUANT["fn"]
    return x if fn is None else fn(x, site)


# --------------------------------------------------------------------------------------
# optional training-mode dropout (the reference's nn.Dropout sites; off by default = eval())
# --------------------------------------------------------------------------------------
_DROP = {"p": 0.0}


def set_dropout(p: float):
    """p > 0: every nn.Dropout site of the captioning path draws a mask from the torch CPU RNG (train() mode of the
    reference: model/blocks.py:105,135,152,171, model/multihead_attention.py:22-23); 0 = eval()."""
    _DROP["p"] = float(p)


def _drop(x: Tensor) -> Tensor:
    p = _DROP["p"]
    return x if p <= 0.0 else torch.nn.functional.dropout(x, p, training=True)


def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], site: str = "linear") -> Tensor:
    y = _q(x, site) @ _q(w, site).transpose(-1, -2)
    return y if b is None else y + b


# --------------------------------------------------------------------------------------
# masks  (model/masking.py:3-21, epoch_loops/captioning_epoch_loops.py:91-119)
# --------------------------------------------------------------------------------------
def subsequent_mask(size: int) -> Tensor:
    """model/masking.py:3-11 -- lower-triangular (1,S,S) uint8."""
    return torch.tril(torch.ones(1, size, size), 0).to(torch.uint8)


def mask(src: Tensor, trg: Optional[Tensor], pad_idx):
    """model/masking.py:14-21."""
    src_mask = (src != pad_idx).unsqueeze(1)
    if trg is None:
        return src_mask
    trg_mask = (trg != pad_idx).unsqueeze(-2) & subsequent_mask(trg.size(-1)).to(torch.bool)
    return src_mask, trg_mask


def make_masks(feature_stacks: Dict[str, Tensor], captions: Optional[Tensor], pad_idx) -> Dict[str, Tensor]:
    """captioning_epoch_loops.py:105-112 ('audio_video' branch): masks come from
    channel 0 of rgb / audio, compared with pad_idx as a float, before rgb+flow."""
    out = {}
    if captions is None:
        out["A_mask"] = mask(feature_stacks["audio"][:, :, 0], None, pad_idx)
        out["V_mask"] = mask(feature_stacks["rgb"][:, :, 0], None, pad_idx)
    else:
        out["V_mask"], out["C_mask"] = mask(feature_stacks["rgb"][:, :, 0], captions, pad_idx)
        out["A_mask"] = mask(feature_stacks["audio"][:, :, 0], None, pad_idx)
    return out


# --------------------------------------------------------------------------------------
# blocks  (model/blocks.py)
# --------------------------------------------------------------------------------------
def pos_enc_table(seq_len: int, d_model: int) -> np.ndarray:
    """model/blocks.py:89-97 -- float64 table; channel j uses exponent j/d for BOTH
    parities (sin on even j, cos on odd j), i.e. not the textbook (j-1)/d for odd j."""
    pos = np.arange(seq_len, dtype=np.float64)[:, None]
    j = np.arange(d_model, dtype=np.float64)[None, :]
    ang = pos / (10000.0 ** (j / d_model))
    tab = np.where((np.arange(d_model) % 2 == 0)[None, :], np.sin(ang), np.cos(ang))
    return tab


def positional_encoder(x: Tensor) -> Tensor:
    """model/blocks.py:101-107 (dropout after the sum; identity unless set_dropout)."""
    B, S, D = x.shape
    tab = torch.from_numpy(pos_enc_table(S, D)).unsqueeze(0)
    return _drop(x + tab.type_as(x))


def vocabulary_embedder(p: Params, prefix: str, idx: Tensor, emb_dim: int) -> Tensor:
    """model/blocks.py:42-46 -- gather * sqrt(emb_dim)."""
    return p[prefix + "embedder.weight"][idx] * np.sqrt(emb_dim)


def feature_embedder(p: Params, prefix: str, x: Tensor, d_model: int) -> Tensor:
    """model/blocks.py:74-81 -- relu(linear(x) * sqrt(d_model))."""
    y = _linear(x, p[prefix + "embedder.weight"], p[prefix + "embedder.bias"], "emb")
    return torch.relu(y * np.sqrt(d_model))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(size) as used by model/blocks.py:127,143 (biased variance, eps 1e-5)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def residual(p: Params, prefix: str, x: Tensor, sublayer) -> Tensor:
    """model/blocks.py:130-136 -- x + dropout(sublayer(LN(x))); eval: dropout = id."""
    return x + _drop(sublayer(layer_norm(x, p[prefix + "norm.weight"], p[prefix + "norm.bias"])))


def feed_forward(p: Params, prefix: str, x: Tensor) -> Tensor:
    """model/blocks.py:167-174 -- fc2(dropout(relu(fc1(x))))."""
    h = _drop(torch.relu(_linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"], "ffn1")))
    return _linear(h, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"], "ffn2")


def bridge(p: Params, prefix: str, x: Tensor) -> Tensor:
    """model/blocks.py:149-153 -- relu(dropout(linear(LN(x)))), no residual."""
    y = layer_norm(x, p[prefix + "norm.weight"], p[prefix + "norm.bias"])
    return torch.relu(_drop(_linear(y, p[prefix + "linear.weight"], p[prefix + "linear.bias"], "bridge")))


# --------------------------------------------------------------------------------------
# attention  (model/multihead_attention.py)
# --------------------------------------------------------------------------------------
def attention(Q: Tensor, K: Tensor, V: Tensor, msk: Optional[Tensor]) -> Tensor:
    """model/multihead_attention.py:8-26 -- softmax(QK^T/sqrt(d_k) masked with -inf) V.
    Scale is applied BEFORE masking; an all-masked row yields NaN (as the reference)."""
    d_k = Q.size(-1)
    s = (_q(Q, "qk") @ _q(K, "qk").transpose(-1, -2)) / np.sqrt(d_k)
    if msk is not None:
        s = s.masked_fill(msk == 0, -float("inf"))
    return _drop(_q(torch.softmax(s, dim=-1), "pv") @ _q(V, "pv"))


def multiheaded_attention(p: Params, prefix: str, Q: Tensor, K: Tensor, V: Tensor,
                          msk: Optional[Tensor], H: int) -> Tensor:
    """model/multihead_attention.py:55-86."""
    B, Sq, _ = Q.shape
    q = _linear(Q, p[prefix + "linear_Q2d.weight"], p[prefix + "linear_Q2d.bias"], "proj")
    k = _linear(K, p[prefix + "linear_K2d.weight"], p[prefix + "linear_K2d.bias"], "proj")
    v = _linear(V, p[prefix + "linear_V2d.weight"], p[prefix + "linear_V2d.bias"], "proj")
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


# --------------------------------------------------------------------------------------
# encoder / decoder layers  (model/encoders.py, model/decoders.py)
# --------------------------------------------------------------------------------------
def bimodal_encoder_layer(p: Params, pre: str, A: Tensor, V: Tensor, A_mask: Tensor, V_mask: Tensor,
                          H: int) -> Tuple[Tensor, Tensor]:
    """model/encoders.py:49-87.  M1=audio, M2=video (encoders.py:112,126).
    Cross-modal K/V are the OTHER stream's post-self-attention, un-normalised values."""
    A = residual(p, pre + "res_layers_M1.0.", A,
                 lambda x: multiheaded_attention(p, pre + "self_att_M1.", x, x, x, A_mask, H))
    V = residual(p, pre + "res_layers_M2.0.", V,
                 lambda x: multiheaded_attention(p, pre + "self_att_M2.", x, x, x, V_mask, H))
    Av = residual(p, pre + "res_layers_M1.1.", A,
                  lambda x: multiheaded_attention(p, pre + "bi_modal_att_M1.", x, V, V, V_mask, H))
    Va = residual(p, pre + "res_layers_M2.1.", V,
                  lambda x: multiheaded_attention(p, pre + "bi_modal_att_M2.", x, A, A, A_mask, H))
    Av = residual(p, pre + "res_layers_M1.2.", Av, lambda x: feed_forward(p, pre + "feed_forward_M1.", x))
    Va = residual(p, pre + "res_layers_M2.2.", Va, lambda x: feed_forward(p, pre + "feed_forward_M2.", x))
    return Av, Va


def bimodal_encoder(p: Params, pre: str, A: Tensor, V: Tensor, masks: Dict[str, Tensor], H: int, N: int):
    """model/encoders.py:115-128 + LayerStack model/blocks.py:16-19 (no final LayerNorm)."""
    for k in range(N):
        A, V = bimodal_encoder_layer(p, f"{pre}encoder_AV.layers.{k}.", A, V, masks["A_mask"], masks["V_mask"], H)
    return A, V


def bimodal_decoder_layer(p: Params, pre: str, C: Tensor, Av: Tensor, Va: Tensor,
                          masks: Dict[str, Tensor], H: int) -> Tensor:
    """model/decoders.py:55-92."""
    C = residual(p, pre + "res_layer_self_att.", C,
                 lambda x: multiheaded_attention(p, pre + "self_att.", x, x, x, masks["C_mask"], H))
    Ca = residual(p, pre + "res_layer_enc_att_A.", C,
                  lambda x: multiheaded_attention(p, pre + "enc_att_A.", x, Av, Av, masks["A_mask"], H))
    Cv = residual(p, pre + "res_layer_enc_att_V.", C,
                  lambda x: multiheaded_attention(p, pre + "enc_att_V.", x, Va, Va, masks["V_mask"], H))
    C = bridge(p, pre + "bridge.", torch.cat([Ca, Cv], dim=-1))
    return residual(p, pre + "res_layer_ff.", C, lambda x: feed_forward(p, pre + "feed_forward.", x))


def bimodal_decoder(p: Params, pre: str, C: Tensor, Av: Tensor, Va: Tensor, masks, H: int, N: int) -> Tensor:
    """model/decoders.py:123-136."""
    for k in range(N):
        C = bimodal_decoder_layer(p, f"{pre}decoder.layers.{k}.", C, Av, Va, masks, H)
    return C


def generator(p: Params, pre: str, x: Tensor) -> Tensor:
    """model/generators.py:18-19 -- log_softmax(linear(x))."""
    return torch.log_softmax(_linear(x, p[pre + "linear.weight"], p[pre + "linear.bias"], "gen"), dim=-1)


# --------------------------------------------------------------------------------------
# whole captioning model  (model/captioning_module.py:164-187)
# --------------------------------------------------------------------------------------
def bimodal_transformer(p: Params, cfg, src: Dict[str, Tensor], trg: Tensor, masks: Dict[str, Tensor],
                        return_memory: bool = False):
    V = src["rgb"] + src["flow"]
    A = src["audio"]
    if getattr(cfg, "use_linear_embedder", False):
        A = feature_embedder(p, "emb_A.", A, cfg.d_model_audio)
        V = feature_embedder(p, "emb_V.", V, cfg.d_model_video)
    C = vocabulary_embedder(p, "emb_C.", trg, cfg.d_model_caps)
    A, V, C = positional_encoder(A), positional_encoder(V), positional_encoder(C)
    Av, Va = bimodal_encoder(p, "encoder.", A, V, masks, cfg.H, cfg.N)
    C = bimodal_decoder(p, "decoder.", C, Av, Va, masks, cfg.H, cfg.N)
    out = generator(p, "generator.", C)
    return (out, Av, Va) if return_memory else out


# --------------------------------------------------------------------------------------
# loss  (loss/label_smoothing.py:12-32)
# --------------------------------------------------------------------------------------
def label_smoothing_kl(pred: Tensor, target: Tensor, smoothing: float, pad_idx: int) -> Tensor:
    """Dense restatement incl. the index-0 quirk: pad rows are zeroed only if the SUM of
    their flat indices is > 0 (label_smoothing.py:26-30), so a lone pad target at flat
    index 0 keeps its (smoothed) row."""
    B, S, V = pred.shape
    pred = pred.contiguous().view(-1, V)
    target = target.contiguous().view(-1)
    dist = smoothing * torch.ones_like(pred) / (V - 2)
    dist.scatter_(1, target.unsqueeze(-1).long(), 1 - smoothing)
    dist[:, pad_idx] = 0
    pad_rows = torch.nonzero(target == pad_idx)
    if len(pad_rows) > 0 and pad_rows.sum() > 0:
        dist.index_fill_(0, pad_rows.squeeze(-1), 0)
    # F.kl_div(pred, dist, reduction='sum') = sum dist*(log dist - pred), 0*log0 := 0
    pos = dist > 0
    return (dist[pos] * (dist[pos].log() - pred[pos])).sum()


def train_cap_loss(p: Params, cfg, src, caption_idx: Tensor, pad_idx: int, smoothing: float):
    """captioning_epoch_loops.py:130-135 -- shift, masks, forward, KL / n_tokens."""
    x, y = caption_idx[:, :-1], caption_idx[:, 1:]
    masks = make_masks(src, x, pad_idx)
    pred = bimodal_transformer(p, cfg, src, x, masks)
    n_tokens = (y != pad_idx).sum()
    return label_smoothing_kl(pred, y, smoothing, pad_idx) / n_tokens, pred, n_tokens


# --------------------------------------------------------------------------------------
# optimizer  (scripts/train_captioning_module.py:46-48 -> torch.optim.Adam semantics)
# --------------------------------------------------------------------------------------
def adam_step(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0):
    """In-place Adam exactly as torch.optim.Adam (no amsgrad): eps added OUTSIDE the
    bias-corrected sqrt:  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)."""
    if weight_decay != 0.0:
        grad = grad + weight_decay * param
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# proposal generator  (model/proposal_generator.py, utilities/proposal_utils.py:11-57)
# --------------------------------------------------------------------------------------
def tiou_vectorized(seg1: Tensor, seg2: Tensor, without_center_coords: bool = False) -> Tensor:
    """utilities/proposal_utils.py:11-57 (center_length=True path)."""
    if without_center_coords:
        seg1 = torch.cat([torch.zeros_like(seg1), seg1], dim=1)
        seg2 = torch.cat([torch.zeros_like(seg2), seg2], dim=1)
    M, N = seg1.shape[0], seg2.shape[0]
    s1, e1 = (seg1[:, 0] - seg1[:, 1] / 2).view(M, 1), (seg1[:, 0] + seg1[:, 1] / 2).view(M, 1)
    s2, e2 = (seg2[:, 0] - seg2[:, 1] / 2).view(1, N), (seg2[:, 0] + seg2[:, 1] / 2).view(1, N)
    inter = torch.clamp(torch.min(e1, e2) - torch.max(s1, s2), min=0.0)
    union = (e1 - s1) + (e2 - s2) - inter
    union = torch.min(torch.max(e1, e2) - torch.min(s1, s2), union)
    return inter / (union + 1e-8)


def make_targets(B: int, num_anchs: int, G: int, targets: Tensor, anchors: Tensor, stride: float):
    """model/proposal_generator.py:389-448.  ``anchors`` is (A,1) already divided by stride.
    Duplicate (vid,anchor,cell) assignments resolve sequentially, last write wins (the CPU
    index_put order the reference gets)."""
    EPS = 1e-16
    noobj = torch.ones(B, num_anchs, G, dtype=torch.bool)
    obj = torch.zeros(B, num_anchs, G, dtype=torch.bool)
    tx = torch.zeros(B, num_anchs, G)
    tw = torch.zeros(B, num_anchs, G)
    vid = targets[:, 0].long()
    gt_x = targets[:, 1] / stride
    gt_w = targets[:, 2] / stride
    ious = tiou_vectorized(anchors, gt_w.unsqueeze(-1), without_center_coords=True)
    best = ious.max(dim=0)[1]
    cell = gt_x.long().clamp(0, G - 1)
    val_x = gt_x - gt_x.floor()
    val_w = torch.log(gt_w / anchors[best][:, 0] + EPS)
    for i in range(targets.shape[0]):
        obj[vid[i], best[i], cell[i]] = True
        noobj[vid[i], best[i], cell[i]] = False
        tx[vid[i], best[i], cell[i]] = val_x[i]
        tw[vid[i], best[i], cell[i]] = val_w[i]
    return obj, noobj, tx, tw, obj.float()


def proposal_head(p: Params, pre: str, x: Tensor, kernel_size: int) -> Tensor:
    """model/proposal_generator.py:39-47 with the default Sequential (no LN):
    Conv1d(k,pad k//2) -> [Dropout] -> ReLU -> Conv1d(1) -> [Dropout] -> ReLU -> Conv1d(1).
    Key indices are conv_layers.{0,3,6} when dout_p>0 and {0,2,4} when dout_p==0."""
    keys = sorted({int(k[len(pre + "conv_layers."):].split(".")[0]) for k in p if k.startswith(pre + "conv_layers.")})
    assert len(keys) == 3, keys
    w0, b0 = p[f"{pre}conv_layers.{keys[0]}.weight"], p[f"{pre}conv_layers.{keys[0]}.bias"]
    w1, b1 = p[f"{pre}conv_layers.{keys[1]}.weight"], p[f"{pre}conv_layers.{keys[1]}.bias"]
    w2, b2 = p[f"{pre}conv_layers.{keys[2]}.weight"], p[f"{pre}conv_layers.{keys[2]}.bias"]
    B, S, D = x.shape
    pad = kernel_size // 2
    xp = torch.nn.functional.pad(x, (0, 0, pad, pad))                      # (B, S+2p, D)
    cols = xp.unfold(1, kernel_size, 1)                                    # (B, S, D, k)
    h = torch.relu(_linear(cols.reshape(B, S, D * kernel_size), w0.reshape(w0.shape[0], -1), b0, "conv"))
    h = torch.relu(_linear(h, w1[:, :, 0], b1, "conv1"))
    return _linear(h, w2[:, :, 0], b2, "conv1")


def _bce(x: Tensor, t: Tensor) -> Tensor:
    """nn.BCELoss (mean) with log clamped at -100; mean of an empty selection is NaN."""
    return -(t * torch.log(x).clamp(min=-100) + (1 - t) * torch.log(1 - x).clamp(min=-100)).mean()


def forward_modality(p: Params, pre: str, x: Tensor, targets: Optional[Tensor], kernel_size: int,
                     stride: float, anchors_list, obj_coeff: float = 1.0, noobj_coeff: float = 100.0, count_reduce=None):
    """model/proposal_generator.py:272-337.  ``count_reduce`` (data-parallel checks only): a callable that sums a tensor
    over ranks; the MSE / BCE means then divide the LOCAL sums by the GLOBAL obj / noobj cell counts, so that the sum of
    the per-rank losses is the full-batch loss of :316-321."""
    A = len(anchors_list)
    y = proposal_head(p, pre, x, kernel_size)
    B, S, _ = y.shape
    y = y.view(B, S, A, 3).permute(0, 2, 1, 3).contiguous()
    grid = torch.arange(S).view(1, 1, S).float()
    anchors = torch.tensor([[a / stride] for a in anchors_list])
    prior = anchors.view(1, A, 1)
    sc, l, so = torch.sigmoid(y[..., 0]), y[..., 1], torch.sigmoid(y[..., 2])
    preds = y.clone().detach()
    preds[..., 0] = sc.detach() + grid
    preds[..., 1] = prior * torch.exp(l.detach())
    preds[..., 2] = so.detach()
    loss, losses = 0, {}
    if targets is not None:
        obj, noobj, gx, gw, gobj = make_targets(B, A, S, targets, anchors, stride)
        if count_reduce is None:
            lx = ((sc[obj] - gx[obj]) ** 2).mean()
            lw = ((l[obj] - gw[obj]) ** 2).mean()
            lo = _bce(so[obj], gobj[obj])
            ln = _bce(so[noobj], gobj[noobj])
        else:
            n = count_reduce(torch.stack([obj.sum(), noobj.sum()]).float())
            lx = ((sc[obj] - gx[obj]) ** 2).sum() / n[0]
            lw = ((l[obj] - gw[obj]) ** 2).sum() / n[0]
            lo = _bce(so[obj], gobj[obj]) * obj.sum() / n[0] if obj.any() else so.sum() * 0
            ln = _bce(so[noobj], gobj[noobj]) * noobj.sum() / n[1] if noobj.any() else so.sum() * 0
        loss = lx + lw + obj_coeff * lo + noobj_coeff * ln
        losses = {"loss_x": lx, "loss_w": lw, "loss_conf_obj": lo, "loss_conf_noobj": ln}
    preds = preds.view(B, S * A, 3)
    preds[:, :, :2] *= stride
    return preds, loss, losses


def multimodal_proposal_generator(p: Params, cfg, anchors: Dict[str, list], src, targets, masks, count_reduce=None):
    """model/proposal_generator.py:339-387."""
    V = src["rgb"] + src["flow"]
    A = src["audio"]
    if getattr(cfg, "use_linear_embedder", False):
        A = feature_embedder(p, "emb_A.", A, cfg.d_model_audio)
        V = feature_embedder(p, "emb_V.", V, cfg.d_model_video)
    A, V = positional_encoder(A), positional_encoder(V)
    Av, Va = bimodal_encoder(p, "encoder.", A, V, masks, cfg.H, cfg.N)
    preds_A, preds_V, loss_A, loss_V, sum_A, sum_V = [], [], 0, 0, {}, {}
    for i, k in enumerate(cfg.kernel_sizes["audio"]):
        pr, lo, ls = forward_modality(p, f"detection_layers_A.{i}.", Av, targets, k, cfg.strides["audio"],
                                      anchors["audio"], cfg.obj_coeff, cfg.noobj_coeff, count_reduce)
        preds_A.append(pr); loss_A = loss_A + lo
        sum_A = {kk: sum_A.get(kk, 0) + vv for kk, vv in ls.items()}
    for i, k in enumerate(cfg.kernel_sizes["video"]):
        pr, lo, ls = forward_modality(p, f"detection_layers_V.{i}.", Va, targets, k, cfg.strides["video"],
                                      anchors["video"], cfg.obj_coeff, cfg.noobj_coeff, count_reduce)
        preds_V.append(pr); loss_V = loss_V + lo
        sum_V = {kk: sum_V.get(kk, 0) + vv for kk, vv in ls.items()}
    all_preds = torch.cat([torch.cat(preds_A, dim=1), torch.cat(preds_V, dim=1)], dim=1)
    return all_preds, loss_A + loss_V, sum_A, sum_V


# --------------------------------------------------------------------------------------
# parameter construction in the reference's registration order
# (BiModalTransformer.__init__ model/captioning_module.py:111-145)
# --------------------------------------------------------------------------------------
def _mha_shapes(pre, dq, dk, dv, d):
    return [(pre + "linear_Q2d.weight", (d, dq)), (pre + "linear_Q2d.bias", (d,)),
            (pre + "linear_K2d.weight", (d, dk)), (pre + "linear_K2d.bias", (d,)),
            (pre + "linear_V2d.weight", (d, dv)), (pre + "linear_V2d.bias", (d,)),
            (pre + "linear_d2Q.weight", (dq, d)), (pre + "linear_d2Q.bias", (dq,))]


def _ffn_shapes(pre, d, dff):
    return [(pre + "fc1.weight", (dff, d)), (pre + "fc1.bias", (dff,)),
            (pre + "fc2.weight", (d, dff)), (pre + "fc2.bias", (d,))]


def _ln_shapes(pre, d):
    return [(pre + "norm.weight", (d,)), (pre + "norm.bias", (d,))]


def encoder_param_shapes(cfg, pre="encoder."):
    """Registration order of BiModalEncoderLayer.__init__ (model/encoders.py:38-47)."""
    Da, Dv, D = cfg.d_model_audio, cfg.d_model_video, cfg.d_model
    out = []
    for k in range(cfg.N):
        L = f"{pre}encoder_AV.layers.{k}."
        out += _mha_shapes(L + "self_att_M1.", Da, Da, Da, D)
        out += _mha_shapes(L + "self_att_M2.", Dv, Dv, Dv, D)
        out += _mha_shapes(L + "bi_modal_att_M1.", Da, Dv, Dv, D)
        out += _mha_shapes(L + "bi_modal_att_M2.", Dv, Da, Da, D)
        out += _ffn_shapes(L + "feed_forward_M1.", Da, cfg.d_ff_audio)
        out += _ffn_shapes(L + "feed_forward_M2.", Dv, cfg.d_ff_video)
        for i in range(3):
            out += _ln_shapes(f"{L}res_layers_M1.{i}.", Da)
        for i in range(3):
            out += _ln_shapes(f"{L}res_layers_M2.{i}.", Dv)
    return out


def captioning_param_shapes(cfg, voc_size: int):
    """state_dict order of BiModalTransformer."""
    Da, Dv, Dc, D = cfg.d_model_audio, cfg.d_model_video, cfg.d_model_caps, cfg.d_model
    out = []
    if getattr(cfg, "use_linear_embedder", False):
        out += [("emb_A.embedder.weight", (Da, cfg.d_aud)), ("emb_A.embedder.bias", (Da,)),
                ("emb_V.embedder.weight", (Dv, cfg.d_vid)), ("emb_V.embedder.bias", (Dv,))]
    out += [("emb_C.embedder.weight", (voc_size, Dc))]
    out += encoder_param_shapes(cfg)
    for k in range(cfg.N):
        L = f"decoder.decoder.layers.{k}."
        out += _ln_shapes(L + "res_layer_self_att.", Dc)
        out += _mha_shapes(L + "self_att.", Dc, Dc, Dc, D)
        out += _ln_shapes(L + "res_layer_enc_att_A.", Dc)
        out += _ln_shapes(L + "res_layer_enc_att_V.", Dc)
        out += _mha_shapes(L + "enc_att_A.", Dc, Da, Da, D)
        out += _mha_shapes(L + "enc_att_V.", Dc, Dv, Dv, D)
        out += [(L + "bridge.norm.weight", (2 * Dc,)), (L + "bridge.norm.bias", (2 * Dc,)),
                (L + "bridge.linear.weight", (Dc, 2 * Dc)), (L + "bridge.linear.bias", (Dc,))]
        out += _ln_shapes(L + "res_layer_ff.", Dc)
        out += _ffn_shapes(L + "feed_forward.", Dc, cfg.d_ff_caps)
    out += [("generator.linear.weight", (voc_size, Dc)), ("generator.linear.bias", (voc_size,))]
    return out


def init_captioning_params(cfg, voc_size: int, seed: int = 0, glove: Optional[Tensor] = None) -> Params:
    """Re-creates BiModalTransformer's initial weights bit-for-bit WITHOUT the reference:
    under torch.manual_seed(seed) every nn.Module default init consumes the generator in
    construction order, then xavier_uniform_ overwrites every dim>1 parameter in
    ``parameters()`` order (captioning_module.py:139-142), then the embedding is replaced by
    ``glove`` when given (captioning_module.py:145).  Implemented by building a skeleton of
    stock torch.nn layers in the same order -- verified bitwise against the reference in
    tests/test_oracle_vs_reference.py."""
    import torch.nn as nn
    from copy import deepcopy
    torch.manual_seed(seed)

    def mha(dq, dk, dv, d):
        return nn.ModuleDict({"linear_Q2d": nn.Linear(dq, d), "linear_K2d": nn.Linear(dk, d),
                              "linear_V2d": nn.Linear(dv, d), "linear_d2Q": nn.Linear(d, dq)})

    def ffn(d, dff):
        return nn.ModuleDict({"fc1": nn.Linear(d, dff), "fc2": nn.Linear(dff, d)})

    def res(d, n):
        proto = nn.ModuleDict({"norm": nn.LayerNorm(d)})
        return nn.ModuleList([deepcopy(proto) for _ in range(n)])

    Da, Dv, Dc, D = cfg.d_model_audio, cfg.d_model_video, cfg.d_model_caps, cfg.d_model
    root = nn.ModuleDict()
    if getattr(cfg, "use_linear_embedder", False):
        root["emb_A"] = nn.ModuleDict({"embedder": nn.Linear(cfg.d_aud, Da)})
        root["emb_V"] = nn.ModuleDict({"embedder": nn.Linear(cfg.d_vid, Dv)})
    root["emb_C"] = nn.ModuleDict({"embedder": nn.Embedding(voc_size, Dc)})
    enc_layer = nn.ModuleDict()
    enc_layer["self_att_M1"] = mha(Da, Da, Da, D)
    enc_layer["self_att_M2"] = mha(Dv, Dv, Dv, D)
    enc_layer["bi_modal_att_M1"] = mha(Da, Dv, Dv, D)
    enc_layer["bi_modal_att_M2"] = mha(Dv, Da, Da, D)
    enc_layer["feed_forward_M1"] = ffn(Da, cfg.d_ff_audio)
    enc_layer["feed_forward_M2"] = ffn(Dv, cfg.d_ff_video)
    enc_layer["res_layers_M1"] = res(Da, 3)
    enc_layer["res_layers_M2"] = res(Dv, 3)
    root["encoder"] = nn.ModuleDict({"encoder_AV": nn.ModuleDict(
        {"layers": nn.ModuleList([deepcopy(enc_layer) for _ in range(cfg.N)])})})
    dec_layer = nn.ModuleDict()
    dec_layer["res_layer_self_att"] = nn.ModuleDict({"norm": nn.LayerNorm(Dc)})
    dec_layer["self_att"] = mha(Dc, Dc, Dc, D)
    dec_layer["res_layer_enc_att_A"] = nn.ModuleDict({"norm": nn.LayerNorm(Dc)})
    dec_layer["res_layer_enc_att_V"] = nn.ModuleDict({"norm": nn.LayerNorm(Dc)})
    dec_layer["enc_att_A"] = mha(Dc, Da, Da, D)
    dec_layer["enc_att_V"] = mha(Dc, Dv, Dv, D)
    dec_layer["bridge"] = nn.ModuleDict({"norm": nn.LayerNorm(2 * Dc), "linear": nn.Linear(2 * Dc, Dc)})
    dec_layer["res_layer_ff"] = nn.ModuleDict({"norm": nn.LayerNorm(Dc)})
    dec_layer["feed_forward"] = ffn(Dc, cfg.d_ff_caps)
    root["decoder"] = nn.ModuleDict({"decoder": nn.ModuleDict(
        {"layers": nn.ModuleList([deepcopy(dec_layer) for _ in range(cfg.N)])})})
    root["generator"] = nn.ModuleDict({"linear": nn.Linear(Dc, voc_size)})
    for prm in root.parameters():
        if prm.dim() > 1:
            nn.init.xavier_uniform_(prm)
    sd = {k: v.detach().clone() for k, v in root.state_dict().items()}
    if glove is not None:
        sd["emb_C.embedder.weight"] = glove.clone()
    want = [k for k, _ in captioning_param_shapes(cfg, voc_size)]
    assert list(sd.keys()) == want, "parameter order drifted from the reference contract"
    return sd


def state_dict_digest(sd: Params) -> str:
    """sha256 over keys, shapes and raw fp32 bytes -- lets fixtures pin 200 MB of weights in 64 chars."""
    import hashlib
    h = hashlib.sha256()
    for k in sd:
        t = sd[k].detach().cpu().contiguous()
        h.update(k.encode()); h.update(str(tuple(t.shape)).encode()); h.update(t.numpy().tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------------------
# greedy decoding  (epoch_loops/captioning_epoch_loops.py:39-65)
# --------------------------------------------------------------------------------------
def greedy_decode(p: Params, cfg, src: Dict[str, Tensor], max_len: int, start_idx: int, end_idx: int, pad_idx: int,
                  return_margins: bool = False):
    """greedy_decoder, restated: start from <s>, run the WHOLE model on the growing prefix, append the arg-max of the last
    position, stop when every sequence has produced </s> or the prefix is longer than max_len (:58-63).  Sequences keep
    growing after their </s> (the reference does not pad them).  ``return_margins``: also the top-1 minus top-2 log-prob of
    every decision (test infrastructure: a decision with a tiny margin is not a fair bit-exactness target)."""
    B = src["audio"].shape[0]
    done = torch.zeros(B, 1, dtype=torch.bool)
    trg = torch.full((B, 1), start_idx, dtype=torch.long)
    margins = []
    while trg.size(-1) <= max_len and not bool(done.all()):
        masks = make_masks(src, trg, pad_idx)
        preds = bimodal_transformer(p, cfg, src, trg, masks)
        top2 = preds[:, -1].topk(2, dim=-1)
        nxt = top2.indices[:, :1]
        margins.append(top2.values[:, 0] - top2.values[:, 1])
        trg = torch.cat([trg, nxt], dim=-1)
        done = done | (nxt == end_idx)
    return (trg, torch.stack(margins, 1)) if return_margins else trg


# --------------------------------------------------------------------------------------
# proposal post-processing  (utilities/proposal_utils.py:115-212, sample/single_video_prediction.py:176-186)
# --------------------------------------------------------------------------------------
def select_topk_predictions(model_output: Tensor, k: int, return_indices: bool = False):
    """utilities/proposal_utils.py:136-149: sort every video's rows by confidence (column 2), descending, keep k.
    The reference's ``argsort(descending=True)`` is not a stable sort: among EQUAL confidences its order is whatever the
    sort implementation leaves (probed: neither ascending nor descending index).  This restatement -- and the device kernel
    -- fix that freedom to candidate-index order (``stable=True``); on inputs without equal confidences it is the
    reference's result bit for bit, with ties it is one of the orders the reference's contract allows."""
    B, S, F = model_output.shape
    idx = model_output[:, :, 2].argsort(dim=-1, descending=True, stable=True)[:, :k]
    out = model_output.gather(1, idx.view(B, -1, 1).repeat(1, 1, F))
    return (out, idx) if return_indices else out


def get_corner_coords(predictions: Tensor) -> Tensor:
    """:115-121 (center, length) -> (start, end); returns a new tensor (the reference writes in place)"""
    out = predictions.clone()
    out[:, :, 0] = predictions[:, :, 0] - predictions[:, :, 1] / 2
    out[:, :, 1] = predictions[:, :, 0] + predictions[:, :, 1] / 2
    return out


def trim_proposals(model_output: Tensor, duration_in_secs) -> Tensor:
    """:152-161 start clipped to [0, duration], end clipped to <= duration (an end below 0 stays)"""
    dur = torch.as_tensor(duration_in_secs, dtype=torch.float32).view(-1, 1)
    out = model_output.clone()
    out[:, :, 0] = model_output[:, :, 0].max(torch.tensor([0.0])).min(dur)
    out[:, :, 1] = model_output[:, :, 1].min(dur)
    return out


def remove_very_short_segments(model_output: Tensor, shortest_segment_prior: float) -> Tensor:
    """:163-172, batch of one video"""
    assert model_output.shape[0] == 1
    lengths = model_output[0, :, 1] - model_output[0, :, 0]
    return model_output[:, lengths > shortest_segment_prior, :]


def tiou_start_end(one: Tensor, many: Tensor) -> Tensor:
    """tiou_vectorized(center_length=False) of one (start, end) segment against N, :11-57"""
    s1, e1, s2, e2 = one[0], one[1], many[:, 0], many[:, 1]
    inter = torch.clamp(torch.min(e1, e2) - torch.max(s1, s2), min=0.0)
    union = (e1 - s1) + (e2 - s2) - inter
    union = torch.min(torch.max(e1, e2) - torch.min(s1, s2), union)
    return inter / (union + 1e-8)


def non_max_suppression(video_preds: Tensor, tiou_threshold: float) -> Tensor:
    """:175-194 greedy NMS over rows sorted by confidence: keep the head, drop every remaining row whose tIoU with it is not
    below the threshold, repeat"""
    kept = []
    while len(video_preds) > 0:
        kept.append(video_preds[:1])
        if len(video_preds) == 1:
            break
        t = tiou_start_end(video_preds[0], video_preds[1:])
        video_preds = video_preds[1:][t < tiou_threshold]
    return torch.cat(kept) if kept else video_preds


def postprocess_preds(model_output: Tensor, k: int, duration_in_secs) -> Tensor:
    """:196-212 (validation loop): top-k -> corners -> trim"""
    return trim_proposals(get_corner_coords(select_topk_predictions(model_output, k)), duration_in_secs)


def generate_proposals_post(predictions: Tensor, duration_in_secs: float, k: int, shortest_segment_prior: float = 0.2) -> Tensor:
    """sample/single_video_prediction.py:176-186 (one video): corners -> trim -> drop short -> top-k"""
    p = trim_proposals(get_corner_coords(predictions), [duration_in_secs])
    return select_topk_predictions(remove_very_short_segments(p, shortest_segment_prior), k)


# --------------------------------------------------------------------------------------
# feature ingest  (datasets/load_features.py:14-95, datasets/captioning_dataset.py:214-275,
#                  datasets/proposal_dataset.py:69-103)
# --------------------------------------------------------------------------------------
def crop_a_segment(feature: Tensor, start: float, end: float, duration: float):
    """load_features.py:14-36: rows int(S*start/duration) .. int(S*end/duration) (python float arithmetic, truncation); an
    empty range becomes one row ([S:S] -> [S-1:S] at the end of the video); None if the slice is still empty"""
    S = feature.shape[0]
    a, b = int(S * (start / duration)), int(S * (end / duration))
    if a == b:
        if a == S:
            a -= 1
        else:
            b += 1
    out = feature[a:b, :]
    return None if len(out) == 0 else out


def pad_segment(feature: Tensor, max_feature_len: int, pad_idx: float) -> Tensor:
    """load_features.py:38-44"""
    assert feature.shape[0] <= max_feature_len
    return torch.nn.functional.pad(feature, [0, 0, 0, max_feature_len - feature.shape[0]], value=pad_idx)


def collate_caption_features(arrays, items, pad_idx: float, d_vid: int, d_aud: int):
    """captioning_dataset.py:214-261 for in-memory arrays: ``arrays[i]`` = {'rgb','flow','audio'} -> (S, D) fp32 tensor or None
    (missing file); crop every sample to its segment, replace missing / empty stacks by one zero row, pad the batch to its
    longest sample -- rgb and audio with pad_idx, flow with 0."""
    rgb, flow, aud = [], [], []
    for arr, (start, end, duration) in zip(arrays, items):
        r = None if arr.get("rgb") is None else crop_a_segment(arr["rgb"], start, end, duration)
        f = None if arr.get("flow") is None else crop_a_segment(arr["flow"], start, end, duration)
        a = None if arr.get("audio") is None else crop_a_segment(arr["audio"], start, end, duration)
        if r is None or f is None:
            r, f = torch.zeros(1, d_vid), torch.zeros(1, d_vid)
        if a is None:
            a = torch.zeros(1, d_aud)
        rgb.append(r); flow.append(f); aud.append(a)
    ps = torch.nn.utils.rnn.pad_sequence
    return {"rgb": ps(rgb, batch_first=True, padding_value=pad_idx), "flow": ps(flow, batch_first=True, padding_value=0),
            "audio": ps(aud, batch_first=True, padding_value=pad_idx)}


def collate_proposal_features(arrays, pad_idx: float, pad_video: int, pad_audio: int):
    """proposal_dataset.py:69-86 + load_features.py get_full_feat branch: whole videos padded to fixed lengths"""
    return {"rgb": torch.stack([pad_segment(a["rgb"], pad_video, pad_idx) for a in arrays]),
            "flow": torch.stack([pad_segment(a["flow"], pad_video, 0) for a in arrays]),
            "audio": torch.stack([pad_segment(a["audio"], pad_audio, pad_idx) for a in arrays])}
