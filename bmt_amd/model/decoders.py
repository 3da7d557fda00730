"""Drop-in for the reference's model/decoders.py: DecoderLayer :9, BiModalDecoderLayer :37, Decoder :95,
BiModelDecoder :114 (sic -- the reference's spelling is what callers import; BiModalDecoder is an alias)."""
import torch
import torch.nn as nn

from .. import ops

from .blocks import (BridgeConnection, LayerStack, PositionwiseFeedForward, ResidualConnection, clone)
from .multihead_attention import MultiheadedAttention


class DecoderLayer(nn.Module):

    def __init__(self, d_model, dout_p, H, d_ff):
        super(DecoderLayer, self).__init__()
        self.res_layers = clone(ResidualConnection(d_model, dout_p), 3)
        self.self_att = MultiheadedAttention(d_model, d_model, d_model, H)
        self.enc_att = MultiheadedAttention(d_model, d_model, d_model, H)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, dout_p=0.0)

    def forward(self, x, memory, src_mask, trg_mask):
        x = self.res_layers[0](x, lambda y: self.self_att(y, y, y, trg_mask))
        x = self.res_layers[1](x, lambda y: self.enc_att(y, memory, memory, src_mask))
        x = self.res_layers[2](x, self.feed_forward)
        return x


class BiModalDecoderLayer(nn.Module):

    def __init__(self, d_model_A, d_model_V, d_model_C, d_model, dout_p, H, d_ff_C):
        super(BiModalDecoderLayer, self).__init__()
        self.res_layer_self_att = ResidualConnection(d_model_C, dout_p)
        self.self_att = MultiheadedAttention(d_model_C, d_model_C, d_model_C, H, dout_p, d_model)
        self.res_layer_enc_att_A = ResidualConnection(d_model_C, dout_p)
        self.res_layer_enc_att_V = ResidualConnection(d_model_C, dout_p)
        self.enc_att_A = MultiheadedAttention(d_model_C, d_model_A, d_model_A, H, dout_p, d_model)
        self.enc_att_V = MultiheadedAttention(d_model_C, d_model_V, d_model_V, H, dout_p, d_model)
        self.bridge = BridgeConnection(2*d_model_C, d_model_C, dout_p)
        self.res_layer_ff = ResidualConnection(d_model_C, dout_p)
        self.feed_forward = PositionwiseFeedForward(d_model_C, d_ff_C, dout_p)
        ops.tag_policy(self, "dec")     # MFMA operand formats of this layer's products (bmt_amd.ops.POLICIES)

    def self_attention_sublayer(self, C, C_mask):
        return self.res_layer_self_att(C, lambda y: self.self_att(y, y, y, C_mask), fp32_out=False)

    def forward(self, x, masks):
        """one bi-modal decoder layer.  ``x`` is the pair LayerStack threads from layer to layer: the caption stream C (B, T_c, d_caps) and the encoder
        memories (audio-attended-by-video, video-attended-by-audio); ``masks`` the dict of make_masks.  Self-attention under C_mask, the two
        encoder-decoder attentions (the video one on the side stream), bridge, feed-forward; the pair comes back with the new C."""
        C, memory = x
        Av, Va = memory.take() if isinstance(memory, _LayerMemories) else memory

        if not getattr(C, "_bmt_self_att_done", False):          # (BiModalTransformer.forward runs the first layer's beside the encoder)
            C = self.self_attention_sublayer(C, masks['C_mask'])
        # the two encoder-decoder attentions read the same C and different memories: the video one (with its projections of the memory)
        # on the side stream (ops.fork_side_stream), joined before the bridge
        s2 = ops.fork_side_stream() if C.is_cuda else None
        C_a, C_v = ops.fanout(C, 2)         # (two consumers: their gradients are added by a library launch, not by autograd's)
        if s2 is None:
            Ca = self.res_layer_enc_att_A(C_a, lambda y: self.enc_att_A(y, Av, Av, masks['A_mask']), fp32_out=False)
            Cv = self.res_layer_enc_att_V(C_v, lambda y: self.enc_att_V(y, Va, Va, masks['V_mask']), fp32_out=False)
        else:
            s1 = torch.cuda.current_stream()
            for t in (C, Va, masks['V_mask']):
                ops.record_stream(t, s2)        # (the tensor, its row pack and its attached operand planes: all read by the side stream's kernels)
            with torch.cuda.stream(s2):
                Cv = self.res_layer_enc_att_V(C_v, lambda y: self.enc_att_V(y, Va, Va, masks['V_mask']), fp32_out=False)
            Ca = self.res_layer_enc_att_A(C_a, lambda y: self.enc_att_A(y, Av, Av, masks['A_mask']), fp32_out=False)
            s1.wait_stream(s2)
            Cv.record_stream(s1)
        # (B, Sc, 2*Dc) -> bridge -> (B, Sc, Dc); no residual across the bridge
        C = self.bridge(ops.cat2(Ca, Cv))
        # (the LAST layer's result is the generator's operand: written as its planes by the feed-forward's last GEMM -- BiModelDecoder sets the format)
        C = self.res_layer_ff(C, self.feed_forward, fp32_out=False, out_planes=getattr(self, "out_planes_fmt", None))

        return C, memory


class _LayerMemories(tuple):
    """the encoder memories (Av, Va) as a decoder stack threads them through its layers, with one alias pair per layer behind it
    (ops.fanout): layer k reads pair k, so that the N layers' gradients w.r.t. a memory are added by library launches in ONE autograd node.
    Still the (Av, Va) tuple the reference's layers pass along (model/decoders.py:55-92)."""

    def __new__(cls, Av, Va, n):
        self = super().__new__(cls, (Av, Va))
        # (a memory prepared for the reassociated cross-attention hands its gradient over in its own node, ops.RawMemoryFn: no aliases)
        al = lambda x: (x,) * n if getattr(x, "_bmt_rawmem", None) is not None else ops.fanout(x, n)
        self._pairs = list(zip(al(Av), al(Va)))
        return self

    def take(self):
        return self._pairs.pop(0) if self._pairs else (self[0], self[1])


class Decoder(nn.Module):

    def __init__(self, d_model, dout_p, H, d_ff, N):
        super(Decoder, self).__init__()
        self.dec_layers = clone(DecoderLayer(d_model, dout_p, H, d_ff), N)

    def forward(self, x, memory, src_mask, trg_mask):
        for layer in self.dec_layers:
            x = layer(x, memory, src_mask, trg_mask)
        return x


class BiModelDecoder(nn.Module):

    def __init__(self, d_model_A, d_model_V, d_model_C, d_model, dout_p, H, d_ff_C, N):
        super(BiModelDecoder, self).__init__()
        layer = BiModalDecoderLayer(d_model_A, d_model_V, d_model_C, d_model, dout_p, H, d_ff_C)
        self.decoder = LayerStack(layer, N)

    def forward(self, x, masks):
        C0, (Av, Va) = x
        layers = self.decoder.layers
        if Av.is_cuda:
            ops.run_deferred_beside()       # (the gradient arena's zero fill of a train step: beside this launch-bound phase)
        if Av.is_cuda and len(layers) > 0 and isinstance(layers[0], BiModalDecoderLayer):
            # the layers' cross-attentions against the raw memories (29 queries per sample: the key / value projections reassociated onto
            # the queries, ops.RawCrossAttnFn) where the memories are packed; otherwise the memories come back as they are
            Av = ops.raw_memory(Av, len(layers), layers[0].enc_att_A.H, C0.shape[1], ops.policy_of(layers[0].enc_att_A))
            Va = ops.raw_memory(Va, len(layers), layers[0].enc_att_V.H, C0.shape[1], ops.policy_of(layers[0].enc_att_V))
            x = (C0, (Av, Va))
        if Av.is_cuda and torch.is_grad_enabled() and len(self.decoder.layers) > 1 and ops.context().kv_cache is None:
            # one alias pair of the memories per layer: the layers' gradients w.r.t. a memory are added by library launches in one node
            x = (C0, _LayerMemories(Av, Va, len(self.decoder.layers)))
        if Av.is_cuda and len(layers) > 0 and isinstance(layers[-1], BiModalDecoderLayer):
            layers[-1].out_planes_fmt = ops.act_fmt(ops.policy_of(None).gemm)      # what Generator's Linear reads (ops.GeneratorFn)
        C, memory = self.decoder(x, masks)
        if C.is_cuda:
            ops.end_of_forward()
        return C


BiModalDecoder = BiModelDecoder
