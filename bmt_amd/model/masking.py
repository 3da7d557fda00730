"""Padding / causal masks -- drop-in for the reference's model/masking.py:3-21 (bit-exact)."""
import torch

from .. import ops


def subsequent_mask(size):
    """(1, size, size) uint8 lower-triangular mask (model/masking.py:3-11).  Tiny and input-independent:
    built on the host side."""
    return torch.tril(torch.ones(1, size, size), 0).byte()


def mask(src, trg, pad_idx):
    """model/masking.py:14-21.  ``src`` is either a (B,S) channel-0 slice of a feature stack (float) or a token
    matrix; masks are produced by the HIP kernels (bmt_mask_from_features / bmt_mask_from_tokens)."""
    if src.is_floating_point():
        src_mask = ops.mask_from_features(src, pad_idx)
    else:
        src_mask, _ = ops.mask_from_tokens(src, pad_idx, want_src=True, want_trg=False)
    if trg is not None:
        _, trg_mask = ops.mask_from_tokens(trg, pad_idx, want_src=False, want_trg=True)
        return src_mask, trg_mask
    return src_mask
