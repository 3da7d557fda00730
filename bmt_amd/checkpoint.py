"""Checkpoint interchange with the reference.

The reference writes ``best_cap_model.pt`` (epoch_loops/captioning_epoch_loops.py:68-88) and ``best_prop_model.pt``
(epoch_loops/proposal_epoch_loops.py:10-25) with ``torch.save`` of a plain dict; its models are always wrapped in
``nn.DataParallel`` so every ``model_state_dict`` key starts with ``module.``.  The functions here write the same files with
the same keys from un-wrapped bmt_amd models (one process per GPU: there is no wrapper module), and read either layout.

``config`` is stored as given: the reference pickles its ``Config`` object, which unpickles only where the reference's
``utilities.config_constructor`` is importable; ``load_checkpoint(..., with_config=False)`` skips nothing but documents
that the weights do not depend on it."""
import os
from collections import OrderedDict

import torch

PREFIX = 'module.'
CAP_FILE = 'best_cap_model.pt'
PROP_FILE = 'best_prop_model.pt'


def with_prefix(state_dict):
    """state_dict keys as nn.DataParallel(model).state_dict() would name them"""
    return OrderedDict((k if k.startswith(PREFIX) else PREFIX + k, v) for k, v in state_dict.items())


def without_prefix(state_dict):
    return OrderedDict((k[len(PREFIX):] if k.startswith(PREFIX) else k, v) for k, v in state_dict.items())


def _module(model):
    return model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model


def dropout_state(device=None):
    """{'seed', 'step'} of the library's counter-based dropout stream on `device` (ops.rng_tensor lives in device memory so that captured
    graphs draw fresh masks), or None on a host without a GPU.  An EXTRA key of the checkpoint dicts ('bmt_dropout_state'): the
    reference's loaders read the keys they know and ignore it; restore_dropout_state() makes a resumed run continue the same mask
    sequence instead of restarting it at step 0."""
    if not torch.cuda.is_available():
        return None
    from . import ops
    seed, step = (int(x) for x in ops.rng_tensor(device).tolist())
    return {'seed': seed, 'step': step}


def restore_dropout_state(checkpoint, device=None):
    """the dropout stream of a checkpoint dict written by save_cap_model / save_prop_model (no-op for a reference checkpoint)"""
    st = checkpoint.get('bmt_dropout_state') if isinstance(checkpoint, dict) else None
    if not st or not torch.cuda.is_available():
        return False
    from . import ops
    ops.manual_seed(int(st['seed']), device)
    ops.rng_tensor(device)[1] = int(st['step'])
    return True


def save_cap_model(cfg, epoch, model, optimizer, val_1_loss_value, val_2_loss_value, val_1_metrics, val_2_metrics,
                   trg_voc_size):
    """save_model of epoch_loops/captioning_epoch_loops.py:68-88: same dict keys, same file name under
    cfg.model_checkpoint_path; returns the path."""
    dict_to_save = {
        'config': cfg,
        'epoch': epoch,
        'model_state_dict': with_prefix(_module(model).state_dict()),
        'optimizer_state_dict': optimizer.state_dict(),
        'val_1_loss': val_1_loss_value,
        'val_2_loss': val_2_loss_value,
        'val_1_metrics': val_1_metrics,
        'val_2_metrics': val_2_metrics,
        'trg_voc_size': trg_voc_size,
        'bmt_dropout_state': dropout_state(next(_module(model).parameters()).device),
    }
    os.makedirs(cfg.model_checkpoint_path, exist_ok=True)
    path_to_save = os.path.join(cfg.model_checkpoint_path, CAP_FILE)
    torch.save(dict_to_save, path_to_save)
    return path_to_save


def save_prop_model(cfg, epoch, model, optimizer, scheduler, anet_metrics, best_metric):
    """save_model of epoch_loops/proposal_epoch_loops.py:10-25: same dict keys, same file name under cfg.log_path."""
    m = _module(model)
    dict_to_save = {
        'config': cfg,
        'epoch': epoch,
        'model_state_dict': with_prefix(m.state_dict()),
        'optimizer_state_dict': optimizer.state_dict(),
        'scheduler_state_dict': None if scheduler is None else scheduler.state_dict(),
        'anchors': m.anchors,
        'val_anet_metrics': anet_metrics,
        'best_metric': best_metric,
        'bmt_dropout_state': dropout_state(next(m.parameters()).device),
    }
    os.makedirs(cfg.log_path, exist_ok=True)
    path_to_save = os.path.join(cfg.log_path, PROP_FILE)
    torch.save(dict_to_save, path_to_save)
    return path_to_save


def load_checkpoint(path, map_location='cpu'):
    """the dict a reference (or bmt_amd) checkpoint holds; pickled Config objects need their class importable"""
    return torch.load(path, map_location=map_location, weights_only=False)


def load_model_state(model, checkpoint, strict=True):
    """weights of a checkpoint dict / path / bare state_dict into an un-wrapped (or wrapped) model; ``module.`` is handled in
    either direction.  Returns the checkpoint dict (or None for a bare state_dict)."""
    cpt = load_checkpoint(checkpoint) if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
    sd = cpt['model_state_dict'] if 'model_state_dict' in cpt else cpt
    _module(model).load_state_dict(without_prefix(sd), strict=strict)
    return cpt if 'model_state_dict' in cpt else None
