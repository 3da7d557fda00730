"""The train_cap step, restated from ``training_loop`` (epoch_loops/captioning_epoch_loops.py:122-149):

    zero_grad -> shift captions -> masks -> forward -> LabelSmoothing(pred, y) / n_tokens -> backward
              -> [clip_grad_norm_] -> optimizer.step

with the data-parallel differences of bmt_amd.parallel: n_tokens is the GLOBAL non-pad count and gradients are summed
over ranks while the backward pass is still running.  No host synchronisation inside the step (the reference's
``loss.item()`` becomes a device scalar the caller may read whenever it wants)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .loss.label_smoothing import LabelSmoothing
from .model.masking import mask as make_mask
from .optim import FusedAdam, clip_grad_norm_
from .parallel import GradientReducer, global_sum


def make_masks(feature_stacks: Dict[str, torch.Tensor], captions: Optional[torch.Tensor], modality: str, pad_idx: int):
    """make_masks (epoch_loops/captioning_epoch_loops.py:91-119), 'audio_video' / 'video' / 'audio' branches: masks come
    from channel 0 of rgb / audio compared with pad_idx, before rgb+flow."""
    masks = {}
    if modality == 'video':
        if captions is None:
            masks['V_mask'] = make_mask(feature_stacks['rgb'][:, :, 0], None, pad_idx)
        else:
            masks['V_mask'], masks['C_mask'] = make_mask(feature_stacks['rgb'][:, :, 0], captions, pad_idx)
    elif modality == 'audio':
        if captions is None:
            masks['A_mask'] = make_mask(feature_stacks['audio'][:, :, 0], None, pad_idx)
        else:
            masks['A_mask'], masks['C_mask'] = make_mask(feature_stacks['audio'][:, :, 0], captions, pad_idx)
    elif modality == 'audio_video':
        if captions is None:
            masks['A_mask'] = make_mask(feature_stacks['audio'][:, :, 0], None, pad_idx)
            masks['V_mask'] = make_mask(feature_stacks['rgb'][:, :, 0], None, pad_idx)
        else:
            masks['V_mask'], masks['C_mask'] = make_mask(feature_stacks['rgb'][:, :, 0], captions, pad_idx)
            masks['A_mask'] = make_mask(feature_stacks['audio'][:, :, 0], None, pad_idx)
    else:
        raise ValueError(f'unknown modality {modality}')
    return masks


class CaptioningTrainStep:
    def __init__(self, model, cfg, pad_idx: int, optimizer=None, data_parallel: bool = False, bucket_bytes: int = 32 << 20):
        self.model, self.cfg, self.pad_idx = model, cfg, pad_idx
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        self.optimizer = optimizer or FusedAdam(params, lr=cfg.lr, betas=tuple(cfg.betas), eps=cfg.eps,
                                                weight_decay=cfg.weight_decay)
        self.criterion = LabelSmoothing(cfg.smoothing, pad_idx)
        self.reducer = GradientReducer(params, bucket_bytes=bucket_bytes) if data_parallel else None
        self.modality = getattr(cfg, 'modality', 'audio_video')

    def __call__(self, feature_stacks, caption_idx):
        model = self.model
        model.train()
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad()
        x, y = caption_idx[:, :-1], caption_idx[:, 1:]
        masks = make_masks(feature_stacks, x, self.modality, self.pad_idx)
        pred = model(feature_stacks, x, masks)
        n_tokens = (y != self.pad_idx).sum()
        n_global = global_sum(n_tokens) if self.reducer is not None else n_tokens
        loss = self.criterion(pred, y) / n_global
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        if self.cfg.grad_clip is not None:
            clip_grad_norm_(self.params, self.cfg.grad_clip)
        self.optimizer.step()
        return loss.detach(), n_tokens
