"""Drop-in for the reference's model/blocks.py: LayerStack :10, clone :21, Identity :24, VocabularyEmbedder :33,
FeatureEmbedder :66, PositionalEncoder :84, Transpose :110, ResidualConnection :123, BridgeConnection :139,
PositionwiseFeedForward :156.  Same constructor arguments, attribute names and state_dict keys; the arithmetic is
issued by libbmt_hip.so through bmt_amd.ops."""
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from .. import ops


class LayerStack(nn.Module):

    def __init__(self, layer, N):
        super(LayerStack, self).__init__()
        self.layers = clone(layer, N)

    def forward(self, x, masks):
        for layer in self.layers:
            x = layer(x, masks)
        return x


def clone(module, N):
    mods = nn.ModuleList([deepcopy(module) for _ in range(N)])
    # deep copies must not share dropout call sites (each nn.Dropout of the reference draws its own mask)
    for m in mods.modules():
        if hasattr(m, '_site'):
            m._site = ops.new_site()
        if hasattr(m, '_site2'):
            m._site2 = ops.new_site()
    return mods


class Identity(nn.Module):

    def __init__(self):
        super(Identity, self).__init__()

    def forward(self, x):
        return x


class VocabularyEmbedder(nn.Module):

    def __init__(self, voc_size, emb_dim):
        super(VocabularyEmbedder, self).__init__()
        self.voc_size = voc_size
        self.emb_dim = emb_dim
        self.embedder = nn.Embedding(voc_size, emb_dim)

    def forward(self, x):
        if isinstance(self.embedder, nn.Embedding):
            return _embed(self, x, pe=None, p=0.0, site=0)
        # GloVe of another width: Sequential(Embedding, Linear, ReLU)  (blocks.py:57-61)
        e = _embed_table(self.embedder[0], x)
        lin = self.embedder[1]
        e = ops.LinearActFn.apply(e, lin.weight, lin.bias, True, "none", 0.0, 0)
        return e * float(np.sqrt(self.emb_dim))

    def init_word_embeddings(self, weight_matrix, emb_weights_req_grad=True):
        if weight_matrix is None:
            print('Training word embeddings from scratch')
        else:
            pretrained_voc_size, pretrained_emb_dim = weight_matrix.shape
            if self.emb_dim == pretrained_emb_dim:
                self.embedder = self.embedder.from_pretrained(weight_matrix)
                self.embedder.weight.requires_grad = emb_weights_req_grad
                print('Glove emb of the same size as d_model_caps')
            else:
                self.embedder = nn.Sequential(
                    nn.Embedding(self.voc_size, pretrained_emb_dim).from_pretrained(weight_matrix),
                    nn.Linear(pretrained_emb_dim, self.emb_dim),
                    nn.ReLU()
                )
                self.embedder[0].weight.requires_grad = emb_weights_req_grad


_ZERO_PE = {}


def _zero_pe(S, D, device):
    key = (S, D, str(device))
    if key not in _ZERO_PE:
        _ZERO_PE[key] = torch.zeros(S, D, device=device, dtype=torch.float32)
    return _ZERO_PE[key]


def _embed_table(emb, ids):
    S, D = ids.shape[1], emb.weight.shape[1]
    return ops.EmbedFn.apply(ids, emb.weight, _zero_pe(S, D, emb.weight.device), 1.0, 0.0, 0)


def _embed(vocab_embedder, ids, pe, p, site):
    W = vocab_embedder.embedder.weight
    S, D = ids.shape[1], W.shape[1]
    if pe is None:
        pe = _zero_pe(S, D, W.device)
    return ops.EmbedFn.apply(ids, W, pe, float(np.sqrt(vocab_embedder.emb_dim)), p, site)


class FeatureEmbedder(nn.Module):

    def __init__(self, d_feat, d_model):
        super(FeatureEmbedder, self).__init__()
        self.d_model = d_model
        self.embedder = nn.Linear(d_feat, d_model)
        self.activation = nn.ReLU()

    def forward(self, x):
        # relu(linear(x) * sqrt(d_model)) == sqrt(d_model) * relu(linear(x))   (blocks.py:74-81)
        y = ops.LinearActFn.apply(x, self.embedder.weight, self.embedder.bias, True, "none", 0.0, 0)
        return y * float(np.sqrt(self.d_model))


def pos_enc_table(seq_len, d_model):
    """The reference's table (blocks.py:89-97), float64: sin on even channels, cos on odd channels, exponent j/d
    for BOTH parities."""
    pos = np.arange(seq_len, dtype=np.float64)[:, None]
    j = np.arange(d_model, dtype=np.float64)[None, :]
    ang = pos / (10000 ** (j / d_model))
    even = (np.arange(d_model) % 2 == 0)[None, :]
    return np.where(even, np.sin(ang), np.cos(ang))


class PositionalEncoder(nn.Module):

    def __init__(self, d_model, dout_p, seq_len=3660):
        super(PositionalEncoder, self).__init__()
        self.d_model = d_model
        self.dout_p = dout_p
        self.dropout = nn.Dropout(dout_p)
        # not a buffer / parameter: carries no state (as the reference)
        self.pos_enc_mat = torch.from_numpy(pos_enc_table(seq_len, d_model)).unsqueeze(0)
        self._pe_dev = {}
        self._site = ops.new_site()

    def table(self, device):
        key = str(device)
        if key not in self._pe_dev:
            self._pe_dev[key] = self.pos_enc_mat[0].to(device=device, dtype=torch.float32).contiguous()
        return self._pe_dev[key]

    def forward(self, x, fuse_add=None, pack=None):
        """x + PE[:S] then dropout (blocks.py:101-107).  ``fuse_add`` (optional second tensor added to x first)
        lets the caller fold the rgb+flow add of captioning_module.py:165 into the same pass.  ``pack`` (ops.RowPack): the result holds
        the VALID rows only, compacted -- row r is position row_map[r] of the padded batch, with that position's table row."""
        B, S, d_model = x.shape
        p = self.dout_p if self.training else 0.0
        return ops.PrepFeaturesFn.apply(x, fuse_add, self.table(x.device), p, self._site, pack)

    def __deepcopy__(self, memo):
        new = PositionalEncoder.__new__(PositionalEncoder)
        nn.Module.__init__(new)
        new.d_model, new.dout_p = self.d_model, self.dout_p
        new.dropout = nn.Dropout(self.dout_p)
        new.pos_enc_mat = self.pos_enc_mat
        new._pe_dev = {}
        new._site = ops.new_site()
        return new


class Transpose(nn.Module):
    """swaps the last two axes between the channel-last layout of LayerNorm and the channel-first layout of nn.Conv1d (reference blocks.py:110-120;
    the proposal heads here convolve channel-last activations directly, so the module only exists for layer_norm=True heads and state_dict parity)"""

    def __init__(self):
        super(Transpose, self).__init__()

    def forward(self, x):
        return x.permute(0, 2, 1)


def layer_norm(norm: nn.LayerNorm, x, planes_fmt=None):
    """nn.LayerNorm(x); planes_fmt: the kernel also writes the result's operand planes of that format, attached to the result (ops.planes_of)"""
    if planes_fmt is None or not x.is_cuda:
        return ops.LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)
    c = ops.context()
    c.last_ln = None
    y = ops.LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps, planes_fmt)
    if c.last_ln is not None:
        ops.attach_planes(y, c.last_ln)
        c.last_ln = None
    return y


class ResidualConnection(nn.Module):

    def __init__(self, size, dout_p):
        super(ResidualConnection, self).__init__()
        self.norm = nn.LayerNorm(size)
        self.dout_p = dout_p
        self.dropout = nn.Dropout(dout_p)
        self._site = ops.new_site()

    def prenorm(self, x, fp32_out=True):
        """the LayerNorm half of forward() ahead of time, for a caller that hands ``x`` to ANOTHER consumer as well (the bi-modal encoder layer:
        a stream's post-self-attention value is this connection's input and the other modality's key / value input): returns (pre, x_kv) --
        ``pre`` goes to forward(None, sublayer, pre=pre), ``x_kv`` is x for the other consumer, tied to the same autograd node so that the three
        gradients of x meet in the LayerNorm backward kernel (ops.ResidualNormFn).  (None, x) when the fused form is off."""
        if not ops.FUSE_RESIDUAL or not x.is_cuda:
            return None, x
        xid, xn, xkv = ops.residual_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, ops.policy_of(self).gemm,
                                         fp32_out=fp32_out or not ops.LN_PLANES_ONLY, kv_alias=True)
        return (xid, xn), xkv

    def forward(self, x, sublayer, fp32_out=True, out_planes=None, pre=None):
        # x (B, S, D):  x + dropout(sublayer(LN(x)));  fp32_out False (the bi-modal layers' calls): LN(x) is handed to the sublayer as
        # operand planes only (ops.residual_norm); out_planes: a plane format the READER of the result wants -- a sublayer that takes the
        # fused residual writes it from the same epilogue (attached to the result: ops.planes_of); pre: the result of prenorm(x)
        p = self.dout_p if self.training else 0.0
        if not ops.FUSE_RESIDUAL:
            res = sublayer(layer_norm(self.norm, x))
            return ops.DropoutAddFn.apply(x, res, p, self._site)
        # fused form (ops.ResidualNormFn): LN emits its operand planes, the sublayer's last GEMM takes the offered residual and
        # adds dropout + x in its epilogue; a sublayer that does not take the offer gets the separate kernel
        if pre is not None:
            xid, xn = pre
        else:
            xid, xn = ops.residual_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, ops.policy_of(self).gemm,
                                        fp32_out=fp32_out or not ops.LN_PLANES_ONLY)
        off = ops.offer_residual(xid, p, self._site, out_planes)
        res = sublayer(xn)
        ops.take_residual()
        if off.out is None:
            return ops.DropoutAddFn.apply(xid, res, p, self._site)
        if off.out is not res:
            raise RuntimeError("a sublayer took the fused residual but returned a different tensor: wrap extra operations outside "
                               "the ResidualConnection or set ops.FUSE_RESIDUAL = False")
        return res


class BridgeConnection(nn.Module):

    def __init__(self, in_dim, out_dim, dout_p):
        super(BridgeConnection, self).__init__()
        self.norm = nn.LayerNorm(in_dim)
        self.linear = nn.Linear(in_dim, out_dim)
        self.dout_p = dout_p
        self.dropout = nn.Dropout(dout_p)
        self.activation = nn.ReLU()
        self._site = ops.new_site()

    def forward(self, x):
        # relu(dropout(linear(LN(x)))): dropout BEFORE the activation, no residual (blocks.py:149-153)
        x = layer_norm(self.norm, x, planes_fmt=ops.act_fmt(ops.policy_of(None).gemm))      # (the Linear's operand, from the LayerNorm kernel itself)
        p = self.dout_p if self.training else 0.0
        return ops.LinearActFn.apply(x, self.linear.weight, self.linear.bias, True, "pre", p, self._site)


class PositionwiseFeedForward(nn.Module):

    def __init__(self, d_model, d_ff, dout_p):
        super(PositionwiseFeedForward, self).__init__()
        self.d_model = d_model
        self.d_ff = d_ff
        self.dout_p = dout_p
        self.fc1 = nn.Linear(d_model, d_ff)
        self.fc2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dout_p)
        self._site = ops.new_site()

    def forward(self, x):
        """fc2(dropout(relu(fc1(x)))) on (B, T, D); the hidden activation exists as operand planes only"""
        p = self.dout_p if self.training else 0.0
        pol = ops.policy_of(self)
        off = ops.take_residual()
        if off is None:
            return ops.FFNFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, p, self._site, pol, None, 0.0, 0, None)
        off.out = ops.FFNFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, p, self._site, pol, off.x, off.p, off.site,
                                  off.planes_fmt)
        return off.out
