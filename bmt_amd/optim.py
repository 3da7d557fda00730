"""Fused multi-tensor Adam and gradient clipping on libbmt_hip.so.

Restates what scripts/train_captioning_module.py:46-48 builds (torch.optim.Adam, lr 5e-5, betas (0.9, 0.999),
eps 1e-8, weight_decay 0) and epoch_loops/captioning_epoch_loops.py:138-139 (clip_grad_norm_) as ONE kernel
launch over every parameter.  Subclasses torch.optim.Optimizer so ``state_dict()`` keeps torch's layout
(``exp_avg`` / ``exp_avg_sq`` / ``step`` per parameter) and checkpoints interchange."""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import _lib
from .ops import _p, _st, lib


class _PtrTable:
    """Device-side table of tensor pointers + sizes, re-uploaded only when a pointer changes (so a captured
    hipGraph, which needs static gradient buffers anyway, replays without host work)."""

    def __init__(self, device):
        self.device = device
        self.key = None
        self.ptrs = None
        self.sizes = None
        self.max_size = 0
        self.n = 0
        self._retired = []   # tables a captured hipGraph may still read: never freed (a few KB each)

    def update(self, columns: List[List[torch.Tensor]]):
        key = tuple(t.data_ptr() for col in columns for t in col)
        if key == self.key:
            return
        n = len(columns[0])
        host = torch.tensor(key, dtype=torch.int64)
        if self.ptrs is not None:
            self._retired.append((self.ptrs, self.sizes))
            if len(self._retired) > 256:      # gradients re-allocated every step (no static buffers, hence no captured graph): bounded
                del self._retired[:128]
        self.ptrs = host.to(self.device)
        sizes = [t.numel() for t in columns[0]]
        self.sizes = torch.tensor(sizes, dtype=torch.int64).to(self.device)
        self.max_size = max(sizes)
        self.n = n
        self.key = key


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self._steps = {}
        self.grad_scale = None   # optional device float multiplied into every gradient inside the kernel

    def load_state_dict(self, state_dict):
        """torch.optim.Adam / FusedAdam state (``optimizer_state_dict`` of a reference checkpoint): the pointer tables and the
        device step counters are rebuilt at the next step from the loaded state"""
        super().load_state_dict(state_dict)
        self._tables, self._steps = {}, {}
        for st in self.state.values():      # torch.optim.Adam keeps ``step`` as a tensor (possibly on the device) or a float
            if "step" in st:
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32)

    def state_dict(self):
        """torch.optim.Adam's layout.  The step count that matters lives on the device (``_steps``: a captured optimizer graph
        advances it without running this Python), so the per-parameter ``step`` entries are refreshed from it first --
        otherwise a checkpoint written after N graph replays would resume Adam's bias correction near step 0 against fully
        warmed moments."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._steps:
                n = float(self._steps[gi].item())
                for p in group["params"]:
                    st = self.state.get(p)
                    if st is not None and "step" in st:
                        st["step"] = torch.as_tensor(n, dtype=torch.float32)
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not p.is_contiguous() or not p.grad.is_contiguous() or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters and gradients")
            if gi not in self._tables:
                self._tables[gi] = _PtrTable(dev)
                # device-side step counter; resumes from a loaded state_dict (torch layout: one ``step`` per parameter)
                self._steps[gi] = torch.full((1,), int(self.state[ps[0]]["step"]), dtype=torch.int64, device=dev)
            tab = self._tables[gi]
            tab.update([ps, [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                        [self.state[p]["exp_avg_sq"] for p in ps]])
            b1, b2 = group["betas"]
            _lib.check(lib.bmt_adam_step(_p(tab.ptrs), _p(tab.sizes), tab.n, tab.max_size, _p(self._steps[gi]),
                                         float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                         float(group["weight_decay"]), _p(self.grad_scale), _st()), "bmt_adam_step")
            from . import ops as _ops
            _ops.WEIGHT_EPOCH[0] += 1      # cached bf16 weight planes are stale now
            for p in ps:   # host-side mirror of the step count (torch's state layout); the kernel uses the device counter
                self.state[p]["step"] += 1
        return loss


_clip_tables = {}      # (device, parameter set) -> (table, sq, coef): one per parameter SET, kept for the life of the process (a captured
                       # optimizer graph has the table's address baked in; _PtrTable.update re-uploads when a gradient pointer moves and
                       # retires the old table, so re-allocated gradients -- zero_grad() sets them to None -- do not add entries)


@torch.no_grad()
def clip_grad_norm_(parameters: Iterable[torch.nn.Parameter], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (L2): returns the total norm (device scalar); gradients are scaled in place by
    min(1, max_norm / (norm + 1e-6)) without a host sync."""
    ps = [p for p in parameters if p.grad is not None]
    if not ps:
        return torch.zeros(())
    dev = ps[0].device
    key = (str(dev), tuple(id(p) for p in ps))
    if key not in _clip_tables:
        _clip_tables[key] = (_PtrTable(dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev))
    tab, sq, coef = _clip_tables[key]
    tab.update([[p.grad for p in ps]])
    _lib.check(lib.bmt_grad_sqnorm(_p(tab.ptrs), _p(tab.sizes), tab.n, tab.max_size, _p(sq), float(max_norm), _p(coef), _st()),
               "bmt_grad_sqnorm")
    _lib.check(lib.bmt_scale_tensors(_p(tab.ptrs), _p(tab.sizes), tab.n, tab.max_size, _p(coef), _st()), "bmt_scale_tensors")
    return sq.sqrt().squeeze(0)
