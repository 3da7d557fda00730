"""Drop-in for the reference's model/multihead_attention.py (attention :8-26, MultiheadedAttention :29-86)."""
import torch
import torch.nn as nn

from .. import ops


def attention(Q, K, V, mask, dropout=None):
    """softmax(QK^T/sqrt(d_k) masked) V on (B,H,S,d_k) views, as model/multihead_attention.py:8-26.
    Runs the flash kernel (the (B,H,Sq,Sk) score tensor is never materialised).  ``dropout`` is an
    nn.Dropout-like module applied to the OUTPUT (reference :22-23)."""
    B, H, Sq, dk = Q.shape
    q = Q.transpose(1, 2).reshape(B, Sq, H * dk)
    k = K.transpose(1, 2).reshape(B, K.shape[2], H * dk)
    v = V.transpose(1, 2).reshape(B, V.shape[2], H * dk)
    out = _CoreFn.apply(q, k, v, mask, H)
    out = out.view(B, Sq, H, dk).transpose(1, 2)
    if dropout is not None:
        out = dropout(out)
    return out


class _CoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, H):
        q, k, v = ops._f32c(q), ops._f32c(k), ops._f32c(v)
        o, lse = ops.attn_fwd(q, k, v, mask, H)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.mask, ctx.H = mask, H
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = ops.attn_bwd(q, k, v, o, ops._f32c(do), lse, ctx.mask, ctx.H)
        return dq, dk, dv, None, None


class MultiheadedAttention(nn.Module):

    def __init__(self, d_model_Q, d_model_K, d_model_V, H, dout_p=0.0, d_model=None):
        super(MultiheadedAttention, self).__init__()
        self.d_model_Q = d_model_Q
        self.d_model_K = d_model_K
        self.d_model_V = d_model_V
        self.H = H
        self.d_model = d_model
        self.dout_p = dout_p

        if self.d_model is None:
            print(f'd_model: is None')
            self.d_model = self.d_model_Q

        self.d_k = self.d_model // H

        self.linear_Q2d = nn.Linear(self.d_model_Q, self.d_model)
        self.linear_K2d = nn.Linear(self.d_model_K, self.d_model)
        self.linear_V2d = nn.Linear(self.d_model_V, self.d_model)
        self.linear_d2Q = nn.Linear(self.d_model, self.d_model_Q)

        self.dropout = nn.Dropout(self.dout_p)   # kept for the module surface; the mask is drawn in-kernel
        self._site = ops.new_site()

        assert self.d_model % H == 0

    def forward(self, Q, K, V, mask):
        """queries (B, T_q, D_q) against keys / values (B, T_k, D_k / D_v) under a key-padding (B, 1, T_k) or per-query (B, T_q, T_k) mask -> (B, T_q, D_q)"""
        p = self.dout_p if self.training else 0.0
        pol = ops.policy_of(self)         # operand formats of this module's sites (set by the enclosing encoder / decoder layer)
        kv_cache = ops.context().kv_cache
        if kv_cache is not None and not torch.is_grad_enabled() and K is V and K is not Q:
            # greedy decoding: the key / value projections of the encoder memory are computed once per decode (bmt_amd.decode)
            return ops.mha_infer(Q, K, mask, self.linear_Q2d.weight, self.linear_Q2d.bias, self.linear_K2d.weight, self.linear_K2d.bias,
                                 self.linear_V2d.weight, self.linear_V2d.bias, self.linear_d2Q.weight, self.linear_d2Q.bias,
                                 self.H, kv_cache, id(self), pol)
        args = (Q, K, V, mask,
                self.linear_Q2d.weight, self.linear_Q2d.bias,
                self.linear_K2d.weight, self.linear_K2d.bias,
                self.linear_V2d.weight, self.linear_V2d.bias,
                self.linear_d2Q.weight, self.linear_d2Q.bias,
                self.H, p, self._site, pol)
        # a decoder layer's attention over an encoder memory that came prepared for the reassociated form (ops.raw_memory): no key / value projections
        if K is V and ops.raw_form_ok(getattr(K, "_bmt_rawmem", None), Q, self, pol):
            fn = ops.RawCrossAttnFn
        elif ops.rank_form_ok(Q, K, V, self, pol):
            # keys = values narrower than a head (the encoder's audio stream, attended by itself or by the video stream): that input itself is
            # the key / value plane of every head
            fn = ops.RankSelfAttnFn if Q is K else ops.RankCrossAttnFn
        else:
            fn = ops.MHAFn
        off = ops.take_residual()        # an enclosing ResidualConnection offers x, p, site: fused into the out-projection
        if off is None:
            return fn.apply(*args, None, 0.0, 0, None)
        off.out = fn.apply(*args, off.x, off.p, off.site, off.planes_fmt)
        return off.out
