"""Greedy caption decoding with encoder and key/value reuse.

Drop-in for ``greedy_decoder`` of the reference (epoch_loops/captioning_epoch_loops.py:39-65): same arguments, same result
(B, <= max_len + 1) token matrix that starts with ``start_idx`` and grows until every row has produced ``end_idx`` or
``max_len`` tokens were generated.

The reference calls the whole model once per generated token: the bi-modal encoder (which does not depend on the caption
prefix) is re-run every time and every decoder layer re-projects the encoder memory to keys and values.  Here, for models
that expose ``encode`` / ``decode`` (bmt_amd.model.captioning_module.BiModalTransformer):

* the encoder runs ONCE per call;
* the key / value operand planes of the two cross-attentions of every decoder layer are computed once and kept for the
  whole call (``ops.context().kv_cache``; bmt_amd.ops.mha_infer);
* the generator (d_model -> vocabulary GEMM + log-softmax) runs on the last position only.

The decoder self-attention / bridge / FFN still run over the whole prefix (<= 30 tokens: a launch-bound tail next to the
encoder's 800 audio + 256 video positions), so the arithmetic on the path that produces the next token is the same kernels
in the same order as the full forward pass and the tokens are identical to those of the un-cached loop.

Every FLOP runs in libbmt_hip.so; there is no CPU path.
"""
import torch

from . import ops
from .train import make_masks


def greedy_decoder(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality, reuse=True):
    """reference signature + ``reuse`` (False: the reference's loop verbatim in behaviour, one full forward per token)."""
    assert model.training is False, 'call model.eval first'

    with torch.no_grad():
        if 'audio' in modality:
            B = feature_stacks['audio'].shape[0]
            device = feature_stacks['audio'].device
        elif modality == 'video':
            B = feature_stacks['rgb'].shape[0]
            device = feature_stacks['rgb'].device
        else:
            raise Exception(f'Unknown modality: {modality}')

        reuse = reuse and hasattr(model, 'encode') and hasattr(model, 'decode')
        # 1 where the ending token occurred; stop when it occurred in every sequence
        done = torch.zeros(B, 1, dtype=torch.bool, device=device)
        trg = torch.full((B, 1), start_idx, dtype=torch.long, device=device)

        memory = None
        ctx = ops.context()          # per (device, stream): concurrent decodes on other streams keep their own cache
        prev_cache = ctx.kv_cache
        try:
            if reuse:
                memory = model.encode(feature_stacks, make_masks(feature_stacks, trg, modality, pad_idx))
                ctx.kv_cache = {}
            while trg.size(-1) <= max_len and not bool(done.all()):
                masks = make_masks(feature_stacks, trg, modality, pad_idx)
                if reuse:
                    C = model.decode(trg, memory, masks)
                    last = model.generator(C[:, -1:])[:, 0]
                else:
                    last = model(feature_stacks, trg, masks)[:, -1]
                next_word = last.max(dim=-1)[1].unsqueeze(1)
                trg = torch.cat([trg, next_word], dim=-1)
                done = done | torch.eq(next_word, end_idx)
        finally:
            ctx.kv_cache = prev_cache
        return trg
