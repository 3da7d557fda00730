"""Drop-in for the reference's model/encoders.py: EncoderLayer :9, BiModalEncoderLayer :36, Encoder :90,
BiModalEncoder :108.  Same submodule names (state_dict contract) and the same order of
operations -- in particular the cross-modal K/V are the other stream's post-self-attention, un-normalised values
(reference :63-79) and there is no final LayerNorm."""
import torch
import torch.nn as nn

from .. import ops

from .blocks import (BridgeConnection, LayerStack, PositionwiseFeedForward, ResidualConnection, clone)
from .multihead_attention import MultiheadedAttention


class EncoderLayer(nn.Module):

    def __init__(self, d_model, dout_p, H, d_ff):
        super(EncoderLayer, self).__init__()
        self.res_layers = clone(ResidualConnection(d_model, dout_p), 2)
        self.self_att = MultiheadedAttention(d_model, d_model, d_model, H)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, dout_p=0.0)

    def forward(self, x, src_mask):
        """self-attention + feed-forward on one stream (B, T, d_model) under its key-padding mask; shape unchanged"""
        x = self.res_layers[0](x, lambda y: self.self_att(y, y, y, src_mask))
        x = self.res_layers[1](x, self.feed_forward)
        return x


class BiModalEncoderLayer(nn.Module):

    def __init__(self, d_model_M1, d_model_M2, d_model, dout_p, H, d_ff_M1, d_ff_M2):
        super(BiModalEncoderLayer, self).__init__()
        self.self_att_M1 = MultiheadedAttention(d_model_M1, d_model_M1, d_model_M1, H, dout_p, d_model)
        self.self_att_M2 = MultiheadedAttention(d_model_M2, d_model_M2, d_model_M2, H, dout_p, d_model)
        self.bi_modal_att_M1 = MultiheadedAttention(d_model_M1, d_model_M2, d_model_M2, H, dout_p, d_model)
        self.bi_modal_att_M2 = MultiheadedAttention(d_model_M2, d_model_M1, d_model_M1, H, dout_p, d_model)
        self.feed_forward_M1 = PositionwiseFeedForward(d_model_M1, d_ff_M1, dout_p)
        self.feed_forward_M2 = PositionwiseFeedForward(d_model_M2, d_ff_M2, dout_p)
        self.res_layers_M1 = clone(ResidualConnection(d_model_M1, dout_p), 3)
        self.res_layers_M2 = clone(ResidualConnection(d_model_M2, dout_p), 3)
        ops.tag_policy(self, "enc")     # MFMA operand formats of this layer's products (bmt_amd.ops.POLICIES)

    def forward(self, x, masks):
        """one bi-modal encoder layer on the pair of streams (audio, video): each stream's self-attention, then each attends the OTHER stream's
        post-self-attention values (queries normalised, keys / values not), then its feed-forward; ``masks`` is the pair of key-padding
        masks in the same order.  Returns the pair of updated streams, shapes unchanged."""
        M1, M2 = x
        M1_mask, M2_mask = masks
        s2 = getattr(_SESSION, "s2", None)
        if s2 is not None and M1.is_cuda:
            return self._forward_two_streams(M1, M2, M1_mask, M2_mask, s2)

        # 1. self-attention on each stream (its result is the other stream's key / value input: written as those operand planes too)
        kv_fmt = ops.act_fmt(ops.policy_of(self).kv_gemm)
        M1 = self.res_layers_M1[0](M1, lambda y: self.self_att_M1(y, y, y, M1_mask), fp32_out=False, out_planes=kv_fmt)
        M2 = self.res_layers_M2[0](M2, lambda y: self.self_att_M2(y, y, y, M2_mask), fp32_out=False, out_planes=kv_fmt)
        # 2. cross-modal attention: queries are the normalised stream, keys/values the OTHER stream as it is now.  A stream's value has three
        # consumers (residual, LayerNorm, the other stream's key / value projections): all three leave ONE autograd node (prenorm)
        pre1, M1kv = self.res_layers_M1[1].prenorm(M1, fp32_out=False)
        pre2, M2kv = self.res_layers_M2[1].prenorm(M2, fp32_out=False)
        M1m2 = self.res_layers_M1[1](M1, lambda y: self.bi_modal_att_M1(y, M2kv, M2kv, M2_mask), fp32_out=False, pre=pre1)
        M2m1 = self.res_layers_M2[1](M2, lambda y: self.bi_modal_att_M2(y, M1kv, M1kv, M1_mask), fp32_out=False, pre=pre2)
        # 3. feed-forward
        M1m2 = self.res_layers_M1[2](M1m2, self.feed_forward_M1, fp32_out=False, out_planes=getattr(self, "memory_planes_fmt", None))
        M2m1 = self.res_layers_M2[2](M2m1, self.feed_forward_M2, fp32_out=False, out_planes=getattr(self, "memory_planes_fmt", None))

        return M1m2, M2m1


    def _forward_two_streams(self, M1, M2, M1_mask, M2_mask, s2):
        """the same operations with the VIDEO chain issued on the side stream: the chains meet only
        where a cross-modal attention reads the other modality's post-self-attention value (events), and the side chain runs on from
        layer to layer without joining"""
        from types import SimpleNamespace as NS
        s1 = torch.cuda.current_stream()
        a = NS(x=M1, mask=M1_mask, res=self.res_layers_M1, self_att=self.self_att_M1, cross=self.bi_modal_att_M1, ffn=self.feed_forward_M1)
        v = NS(x=M2, mask=M2_mask, res=self.res_layers_M2, self_att=self.self_att_M2, cross=self.bi_modal_att_M2, ffn=self.feed_forward_M2)
        side, main = v, a
        kv_fmt = ops.act_fmt(ops.policy_of(self).kv_gemm)      # a self-attention's result is the other chain's key / value input
        # (a chain's post-self-attention value has three consumers -- residual, LayerNorm, the OTHER chain's key / value projections: prenorm
        # ties them to one autograd node, whose backward kernel adds the three gradients; x1kv is x1 for the other chain)
        with torch.cuda.stream(s2):
            side.x1 = side.res[0](side.x, lambda y: side.self_att(y, y, y, side.mask), fp32_out=False, out_planes=kv_fmt)
            side.pre, side.x1kv = side.res[1].prenorm(side.x1, fp32_out=False)
            e_side = s2.record_event()
        main.x1 = main.res[0](main.x, lambda y: main.self_att(y, y, y, main.mask), fp32_out=False, out_planes=kv_fmt)
        main.pre, main.x1kv = main.res[1].prenorm(main.x1, fp32_out=False)
        e_main = s1.record_event()
        s1.wait_event(e_side)
        ops.record_stream(side.x1kv, s1)
        main.out = main.res[1](main.x1, lambda y: main.cross(y, side.x1kv, side.x1kv, side.mask), fp32_out=False, pre=main.pre)
        main.out = main.res[2](main.out, main.ffn, fp32_out=False, out_planes=getattr(self, "memory_planes_fmt", None))
        with torch.cuda.stream(s2):
            s2.wait_event(e_main)
            ops.record_stream(main.x1kv, s2)
            side.out = side.res[1](side.x1, lambda y: side.cross(y, main.x1kv, main.x1kv, main.mask), fp32_out=False, pre=side.pre)
            side.out = side.res[2](side.out, side.ffn, fp32_out=False, out_planes=getattr(self, "memory_planes_fmt", None))
        return a.out, v.out


_SESSION = __import__("threading").local()        # .s2: the side stream while a BiModalEncoder.forward is running with two streams (per thread)


class Encoder(nn.Module):

    def __init__(self, d_model, dout_p, H, d_ff, N):
        super(Encoder, self).__init__()
        self.enc_layers = clone(EncoderLayer(d_model, dout_p, H, d_ff), N)

    def forward(self, x, src_mask):
        for layer in self.enc_layers:
            x = layer(x, src_mask)
        return x


class BiModalEncoder(nn.Module):

    def __init__(self, d_model_A, d_model_V, d_model, dout_p, H, d_ff_A, d_ff_V, N):
        super(BiModalEncoder, self).__init__()
        layer_AV = BiModalEncoderLayer(d_model_A, d_model_V, d_model, dout_p, H, d_ff_A, d_ff_V)
        self.encoder_AV = LayerStack(layer_AV, N)
        if N <= 2:           # the operand policy depends on the depth (ops.POLICIES: "enc_shallow")
            ops.tag_policy(self.encoder_AV, "enc_shallow")
        # the last layer's results are a decoder's memories: read there as key / value operand planes, written by the last GEMM
        self.encoder_AV.layers[-1].memory_planes_fmt = ops.act_fmt(ops.POLICIES["dec"].kv_gemm)

    def forward(self, x, masks: dict):
        """the stack of bi-modal layers on (audio, video); the masks dict is reordered into the layers' (audio, video) pair.  Returns the two memories"""
        A, V = x
        s2 = ops.fork_side_stream() if A.is_cuda else None
        if s2 is None:
            Av, Va = self.encoder_AV((A, V), (masks['A_mask'], masks['V_mask']))
            return (Av, Va)
        # audio chain on the current stream, video chain on the side stream (ops.fork_side_stream); joined before anything downstream
        _SESSION.s2 = s2
        try:
            for t in (A, V, masks['V_mask'], masks['A_mask']):
                ops.record_stream(t, s2)
            Av, Va = self.encoder_AV((A, V), (masks['A_mask'], masks['V_mask']))
        finally:
            _SESSION.s2 = None
        torch.cuda.current_stream().wait_stream(s2)
        for t in (Av, Va):
            ops.record_stream(t, torch.cuda.current_stream())
        ops.end_of_forward()
        return (Av, Va)
