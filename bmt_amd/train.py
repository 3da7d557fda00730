"""The train_cap step, restated from ``training_loop`` (epoch_loops/captioning_epoch_loops.py:122-149):

    zero_grad -> shift captions -> masks -> forward -> LabelSmoothing(pred, y) / n_tokens -> backward
              -> [clip_grad_norm_] -> optimizer.step

Two things differ in form (not in result):
  * the division by n_tokens is applied to the GRADIENTS inside the fused Adam kernel (``grad_scale``) instead of to the
    loss before backward: d(KL/n)/dw == (dKL/dw)/n.  That makes the normaliser the GLOBAL non-pad count under data
    parallelism (one scalar all-reduce, epoch_loops/captioning_epoch_loops.py:134-135 semantics) and keeps every collective
    outside the captured graphs;
  * the step can be captured into two hipGraphs (forward+backward, optimizer) so that ~800 kernel launches per step are
    replayed by the runtime instead of being issued one at a time from Python; the gradient all-reduce sits between them.
No host synchronisation inside the step (the reference's ``loss.item()`` becomes a device scalar)."""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch

from .loss.label_smoothing import LabelSmoothing
from .model.masking import mask as make_mask
from .optim import FusedAdam, clip_grad_norm_
from .parallel import GradientReducer, global_sum


def _ops_weights_changed():
    from . import ops as _ops
    _ops.weights_changed()


def _seed_dropout(seed: Optional[int], data_parallel: bool, device, keep_if_seeded: bool = False):
    """dropout masks come from the library RNG in device memory (ops.rng_tensor), not from torch's: under data parallelism every
    rank must draw DIFFERENT masks (as nn.DataParallel's replicas do), so the per-rank seed is base + rank.  seed=None keeps
    the current state on a single process and derives base 0x5EED under data parallelism."""
    import torch.distributed as dist
    from . import ops as _ops
    rank = dist.get_rank() if (data_parallel and dist.is_initialized()) else 0
    if seed is None and not data_parallel:
        return
    if not torch.device(device).type == "cuda":
        return
    if keep_if_seeded and seed is None and _ops.rng_is_seeded(device):
        return      # a step built earlier in this process (MixedTrainStep's captioning step) seeded the shared stream: keep it and its counter
    _ops.manual_seed((0x5EED if seed is None else seed) + rank, device)


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
    """One optimizer step of the captioning model.

    ``static_grads`` (implied by ``data_parallel``) binds every ``p.grad`` to a persistent flat bucket
    (bmt_amd.parallel.GradientReducer): that is what the all-reduce runs on and what lets the step be graph-captured.
    ``overlap``: launch each bucket's all-reduce from the backward hooks (eager mode); captured steps reduce after the
    backward graph instead (no collective is ever captured)."""

    def __init__(self, model, cfg, pad_idx: int, optimizer=None, data_parallel: bool = False, bucket_bytes: int = 32 << 20,
                 static_grads: bool = False, overlap: bool = True, seed: Optional[int] = None, collective: str = "allreduce"):
        self.model, self.cfg, self.pad_idx = model, cfg, pad_idx
        self._overlap = overlap
        self._graph_generation = None
        _seed_dropout(seed, data_parallel, next(model.parameters()).device)
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        self.optimizer = optimizer or FusedAdam(params, lr=cfg.lr, betas=tuple(cfg.betas), eps=cfg.eps,
                                                weight_decay=cfg.weight_decay)
        self.criterion = LabelSmoothing(cfg.smoothing, pad_idx)
        self.data_parallel = data_parallel
        from . import ops as _ops
        from .parallel import flush_stages
        self.reducer = GradientReducer(params, bucket_bytes=bucket_bytes, overlap=overlap, groups=_ops.fused_weight_groups(model),
                                       collective=collective, stage_of=flush_stages(model)[0]) if (data_parallel or static_grads) else None
        self.modality = getattr(cfg, 'modality', 'audio_video')
        self.grad_scale = torch.ones(1, device=params[0].device, dtype=torch.float32)
        self._one = torch.ones((), device=params[0].device, dtype=torch.float32)        # the root gradient of every backward pass
        if hasattr(self.optimizer, "grad_scale"):
            self.optimizer.grad_scale = self.grad_scale
        self._fused_scale = hasattr(self.optimizer, "grad_scale")
        self._graphs = None
        self._reduce_events = None      # bench: [(start, end)] HIP events around the exposed part of the gradient reduction
        self._flush_points = 0
        if data_parallel:
            self._install_flush_points()

    def _install_flush_points(self):
        """Overlap of the gradient all-reduce with the backward pass (eager launches, ``reducer.overlap``): the weight-gradient
        products are queued for grouped launches (ops.flush_dw), so a bucket's gradients are final only once its group has been
        issued.  Flush points = the moments the backward pass crosses into an earlier encoder layer (the gradient w.r.t. a layer's
        output arrives: everything after that layer -- later layers, decoder, generator -- has been differentiated): the queue is
        flushed there, the finished buckets' RCCL all-reduces start behind it on the communication stream and run under the
        remaining layers' backward.  ~1000 tiles per flush still fill the chip."""
        from . import ops as _ops
        stack = getattr(getattr(getattr(self.model, "encoder", None), "encoder_AV", None), "layers", None)
        if stack is None:
            return
        step = self

        def attach(mod, inp, out):
            if not (step.reducer is not None and step.reducer.overlap and step.reducer.world > 1 and torch.is_grad_enabled()):
                return
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.requires_grad:
                    t.register_hook(lambda g: (_ops.flush_dw(), g)[1])
        for layer in stack:
            layer.register_forward_hook(attach)
            self._flush_points += 1

    # ---- the three phases -------------------------------------------------------------------------------------
    def _forward_backward(self, feature_stacks, caption_idx):
        """zero_grad -> masks -> forward -> sum-KL -> backward.  Returns (sum-KL, local non-pad token count)."""
        model = self.model
        model.train()
        from . import ops as _ops
        _ops.mark_step_start()          # (the refresh of the weights' operand planes forks from here: ops.EARLY_REFRESH)
        if self.reducer is not None:
            self.reducer.zero_grad(defer=_ops.DEFER_ZERO)      # (issued beside the decoder's forward, joined before backward() below)
        else:
            self.optimizer.zero_grad()
        x, y, n_tokens = _ops.caption_shift(caption_idx, self.pad_idx)
        masks = make_masks(feature_stacks, x, self.modality, self.pad_idx)
        # the encoder's two compute streams: not while gradient buckets are all-reduced from inside the backward pass (a bucket's
        # "final" event is recorded on ONE stream)
        _ops.allow_encoder_streams(self.reducer is None or self.reducer.world == 1 or not self.reducer.overlap)
        pred = model(feature_stacks, x, masks)
        kl = self.criterion(pred, y)
        # the weight-gradient GEMMs of the whole backward pass go out as one grouped launch -- unless gradients are all-reduced
        # bucket by bucket from the backward hooks, which needs them finished in autograd order
        sctx = _ops.context()
        # with bucket-by-bucket overlap the queue is flushed at the layer boundaries (_install_flush_points); without flush points
        # the products run where autograd reaches them, so that a bucket is final when its last hook fires
        sctx.defer_dw = self.reducer is not None and (self.reducer.world == 1 or not self.reducer.overlap or self._flush_points > 0)
        try:
            _ops.join_deferred()
            kl.backward(gradient=self._one if self._one.device == kl.device and kl.dim() == 0 else None)      # (no fill kernel for the root gradient)
            _ops.join_side_stream()
            _ops.flush_dw()
        finally:
            sctx.defer_dw = False
            sctx.pending_dw.clear()
            sctx.pending_cs.clear()
            sctx.pending_post.clear()
            sctx.pending_side.clear()
            sctx.gen_handles.clear()
            _ops.clear_step_start()
        return kl.detach(), n_tokens

    def _reduce(self, kl, n_tokens):
        """gradient sum over ranks (if any) and the global normaliser -> grad_scale = 1 / n_tokens_global."""
        if self.reducer is not None:
            self.reducer.finish()
        n_global = global_sum(n_tokens) if self.data_parallel else n_tokens
        from . import ops as _ops
        return _ops.loss_finish(kl, n_global, self.grad_scale), n_global

    def _optimize(self):
        if not self._fused_scale or self.cfg.grad_clip is not None:
            # generic optimizer, or clipping (which must see the normalised gradients): scale in place first
            for p in self.params:
                if p.grad is not None:
                    p.grad.mul_(self.grad_scale)
            if self._fused_scale:
                self.optimizer.grad_scale = None
        if self.cfg.grad_clip is not None:
            clip_grad_norm_(self.params, self.cfg.grad_clip)
        self.optimizer.step()
        if self._fused_scale:
            self.optimizer.grad_scale = self.grad_scale

    def __call__(self, feature_stacks, caption_idx):
        kl, n_tokens = self._forward_backward(feature_stacks, caption_idx)
        ev = self._reduce_events
        if ev is not None:
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
        loss, _ = self._reduce(kl, n_tokens)
        if ev is not None:
            e_.record()
            ev.append((s_, e_))
        self._optimize()
        return loss, n_tokens

    # ---- hipGraph capture -------------------------------------------------------------------------------------
    def capture(self, feature_stacks, caption_idx, warmup: int = 2, collectives: bool = False):
        """Capture {zero_grad .. backward} and {optimizer} into two graphs over STATIC input buffers (copies of the given
        batch).  ``replay(fs, caps)`` copies a new batch of the same shape into those buffers and launches
        graph 1 -> (eager) gradient all-reduce + normaliser -> graph 2.  Dropout masks still change on every replay
        (seed/step live in device memory, the step counter is advanced by a captured kernel), as do Adam's bias corrections."""
        if self.reducer is None:
            raise RuntimeError("capture() needs static gradient buffers: construct with static_grads=True")
        if collectives:
            return self._capture_with_collectives(feature_stacks, caption_idx, warmup)
        self.reducer.overlap = False            # collectives stay outside the captured region
        self._static_fs = {k: v.clone() for k, v in feature_stacks.items()}
        self._static_caps = caption_idx.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up: lazy inits, pointer tables, weight-plane caches, allocator pools
            for _ in range(warmup):
                self(self._static_fs, self._static_caps)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # capture_error_mode="thread_local": with a process group alive, the RCCL watchdog THREAD polls the events of finished
        # collectives (hipEventQuery); in the default global mode that call is illegal while any stream of the process captures -- it
        # invalidates the capture and the watchdog's exception terminates the process (seen in ~1 of 8 runs of tools/dp_smoke_1gpu.py:
        # "operation not permitted when stream is capturing").  The kernels of autograd's worker threads are still captured: capture
        # is a property of the stream, the mode only says whose unsafe calls are errors.
        from . import ops as _ops
        with _ops.scratch_owner(id(self)), torch.cuda.graph(g1, capture_error_mode="thread_local"):
            self._static_kl, self._static_ntok = self._forward_backward(self._static_fs, self._static_caps)
        self._reduce(self._static_kl, self._static_ntok)
        torch.cuda.synchronize()                # the eager all-reduce has finished before the second capture starts
        with _ops.scratch_owner(id(self)), torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
            self._optimize()
        _ops.finish_capture()                   # (the table images of the captured grouped launches are on the device)
        self._graphs = (g1, g2)
        self._graph_generation = _ops.weights_generation()
        return self._graphs

    def _capture_with_collectives(self, feature_stacks, caption_idx, warmup):
        """ONE graph for the whole step with the RCCL collectives INSIDE it: the bucket all-reduces are launched from the backward hooks
        onto the communication stream exactly as in the eager `overlap` mode (weight-gradient products flushed at the encoder-layer
        boundaries), the capture records them -- and the stream dependencies around them -- as graph nodes, so a replay overlaps each
        bucket's all-reduce with the remaining layers' backward without any host work between kernels (the eager mode pays ~1 ms of
        launch gaps per step for that overlap, the two-graph mode exposes the whole all-reduce).  Opt-in (`bench.py --dp-mode graph-overlap`):
        it relies on RCCL kernels being capturable on the installation, which the 1-rank smoke (tools/dp_smoke_1gpu.py) checks."""
        self.reducer.overlap = True
        self._static_fs = {k: v.clone() for k, v in feature_stacks.items()}
        self._static_caps = caption_idx.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self(self._static_fs, self._static_caps)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from . import ops as _ops
        with _ops.scratch_owner(id(self)), torch.cuda.graph(g, capture_error_mode="thread_local"):
            kl, ntok = self._forward_backward(self._static_fs, self._static_caps)
            self._static_loss, _ = self._reduce(kl, ntok)
            self._static_ntok = ntok
            self._optimize()
        _ops.finish_capture()
        self._graphs = (g,)
        self._graph_generation = _ops.weights_generation()
        return self._graphs

    def uncapture(self):
        """back to eager launches (bucket all-reduces overlapped with the backward pass again)"""
        had = self._graphs is not None
        self._graphs = None
        if had:          # the per-stream scratch the graphs were captured over goes with them
            from . import ops as _ops
            _ops.release_scratch(owner=id(self))
            _ops.release_const_tables(id(self))
        if self.reducer is not None:
            self.reducer.overlap = self._overlap

    def _stage(self, feature_stacks, caption_idx):
        for k, v in feature_stacks.items():
            self._static_fs[k].copy_(v, non_blocking=True)
        self._static_caps.copy_(caption_idx, non_blocking=True)

    def replay(self, feature_stacks=None, caption_idx=None, next_batch=None):
        """one captured step over the given batch (None: the batch the static buffers hold).  ``next_batch`` = (feature_stacks, caption_idx) of
        the step AFTER this one, complete when this call is made: it is copied into the static input buffers as soon as this step's
        forward / backward graph -- their last reader -- has run, on a copy stream, i.e. beside this step's gradient reduction and optimizer
        graph instead of in front of the next replay (~80 MB in four copies, 55 us at configs[1]: profiles/r06_p_replay_dispatches.csv);
        the next call finds its batch staged (same tensor objects) and only waits for that copy.  The prefetch a data loader does."""
        cur = torch.cuda.current_stream()
        staged, self._staged = getattr(self, "_staged", None), None
        if feature_stacks is not None:
            if (staged is not None and staged[1] is caption_idx and len(staged[0]) == len(feature_stacks) and
                    all(staged[0].get(k) is v for k, v in feature_stacks.items())):
                cur.wait_event(staged[2])
            else:
                if staged is not None:          # (another batch than the one announced: its copy must not land on top of this one)
                    cur.wait_event(staged[2])
                self._stage(feature_stacks, caption_idx)
        elif staged is not None:
            cur.wait_event(staged[2])
        from . import ops as _ops
        if _ops.weights_generation() != self._graph_generation:
            raise RuntimeError("the weight-plane registry changed after capture() (a weight's operand planes were re-allocated, or a model was "
                               "garbage-collected): the captured graphs name freed buffers -- call capture() again")
        if len(self._graphs) == 1:          # the whole step incl. its collectives (capture(collectives=True))
            self._graphs[0].replay()
            _ops_weights_changed()
            return self._static_loss, self._static_ntok
        g1, g2 = self._graphs
        g1.replay()
        if next_batch is not None and len(self._graphs) == 2:
            if getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream()
            cs = self._copy_stream
            cs.wait_stream(cur)                 # behind the forward / backward graph (and whatever produced the next batch before this call)
            with torch.cuda.stream(cs):
                self._stage(*next_batch)
                done = torch.cuda.Event()
                done.record(cs)
            for t in list(next_batch[0].values()) + [next_batch[1]]:
                t.record_stream(cs)
            self._staged = (dict(next_batch[0]), next_batch[1], done)
        ev = self._reduce_events
        if ev is not None:
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
        loss, _ = self._reduce(self._static_kl, self._static_ntok)
        if ev is not None:
            e_.record()
            ev.append((s_, e_))
        g2.replay()
        # the captured Adam kernel wrote the parameters behind torch's back (no _version bump, and the host-side epoch
        # increment of FusedAdam.step ran once, at capture time): an eager forward after this replay must refresh the
        # cached weight planes
        _ops_weights_changed()
        return loss, self._static_ntok

    def reduce_timing(self, on: bool):
        """bench: on -> start collecting HIP events around the part of the gradient reduction that is NOT hidden behind the
        backward pass (everything between the end of the backward graph and the optimizer graph); off -> mean ms per step"""
        if on:
            self._reduce_events = []
            return None
        ev, self._reduce_events = self._reduce_events, None
        if not ev:
            return None
        torch.cuda.synchronize()
        return sum(s_.elapsed_time(e_) for s_, e_ in ev) / len(ev)

    def reduce_description(self):
        r = self.reducer
        if r is None:
            return None
        if self._graphs is not None and len(self._graphs) == 1:
            mode = (f"bucket all-reduces (RCCL) captured INSIDE the step's single hipGraph: launched from the backward pass onto the communication "
                    f"stream at {self._flush_points} encoder-layer boundaries, each overlapped with the remaining layers' backward; plus one captured "
                    "scalar all-reduce (global n_tokens)")
        elif self._graphs is not None:
            mode = ("sum all-reduce (RCCL) of the flat fp32 gradient buckets between the backward graph and the optimizer graph (exposed), "
                    "plus one scalar all-reduce (global n_tokens)")
        else:
            mode = (f"bucket all-reduces (RCCL) launched from the backward pass on the communication stream: the queued weight-gradient "
                    f"products are flushed at {self._flush_points} encoder-layer boundaries and each finished bucket reduces under the "
                    "remaining layers' backward; plus one scalar all-reduce (global n_tokens)")
        how = {"allreduce": "sum all-reduce per bucket", "rs_ag": "reduce-scatter + all-gather per bucket"}[r.collective]
        return {"payload_mb": sum(b["flat"].numel() for b in r.buckets) * 4 / 1e6, "buckets": len(r.buckets), "collective": how, "mode": mode}


class ProposalTrainStep:
    """One optimizer step of the proposal generator, restated from ``train_av_loop``
    (epoch_loops/proposal_epoch_loops.py:27-49):

        zero_grad -> masks (no captions) -> model(feature_stacks, targets, masks) -> loss.backward()
                  -> [clip_grad_norm_] -> optimizer.step

    Only parameters with ``requires_grad`` take part: with ``cfg.pretrained_cap_model_path`` the bi-modal encoder is loaded
    from the captioning checkpoint and frozen unless ``cfg.finetune_cap_encoder`` (model/proposal_generator.py:344-353), so
    the step is {frozen encoder forward} + {20 Conv1d heads forward/backward} + Adam over the heads -- configs[3].
    Data parallel: the reference's MSE / BCE are means over the selected cells of the WHOLE batch
    (model/proposal_generator.py:316-321), so every rank divides its LOCAL sums by the GLOBAL obj / noobj cell counts (one
    all-reduce of two scalars per modality, before backward: ``model.count_reduce``) and the gradients are SUMMED over
    ranks -- the result is the full-batch step (tests/test_parallel_gloo.py).  The number of target events
    differs per step, so this step is launched eagerly (no graph capture)."""

    def __init__(self, model, cfg, pad_idx: int, optimizer=None, data_parallel: bool = False, bucket_bytes: int = 32 << 20,
                 overlap: bool = True, seed: Optional[int] = None, keep_seed: bool = False, collective: str = "allreduce",
                 static_grads: bool = False):
        """``static_grads`` (implied by ``data_parallel`` and by capture()): every ``p.grad`` is a view into one flat arena that the
        weight-gradient kernels accumulate into directly -- no per-parameter zero fills, no dW temporaries, no autograd accumulation adds
        (~250 framework launches per eagerly issued configs[3] step otherwise)."""
        import torch.distributed as dist
        self.model, self.cfg, self.pad_idx = model, cfg, pad_idx
        _seed_dropout(seed, data_parallel, next(model.parameters()).device, keep_if_seeded=keep_seed)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.optimizer = optimizer or FusedAdam(self.params, lr=cfg.lr, betas=tuple(getattr(cfg, "betas", (0.9, 0.999))),
                                                eps=getattr(cfg, "eps", 1e-8), weight_decay=getattr(cfg, "weight_decay", 0.0))
        self.data_parallel = data_parallel
        self.reducer = GradientReducer(self.params, bucket_bytes=bucket_bytes, overlap=overlap, collective=collective) if (data_parallel or static_grads) else None
        self.world = dist.get_world_size() if (data_parallel and dist.is_initialized()) else 1
        self.modality = getattr(cfg, 'modality', 'audio_video')
        # data parallel: the loss means use the global cell counts (sum of the per-rank losses == full-batch loss)
        model.count_reduce = global_sum if self.world > 1 else None
        if hasattr(self.optimizer, "grad_scale"):
            self.optimizer.grad_scale = None

    # ---- hipGraph capture: the step is static once the number of target rows is (the reference's batches carry a different number of
    # events each: targets are padded to ``max_events`` rows with batch index -1, which bmt_make_targets skips)
    @staticmethod
    def pad_targets(targets: torch.Tensor, max_events: int) -> torch.Tensor:
        n = targets.shape[0]
        if n > max_events:
            raise ValueError(f"{n} target events, the captured step holds {max_events}")
        out = torch.zeros(max_events, 4, device=targets.device, dtype=torch.float32)
        out[:, 0] = -1.0
        out[:, 2] = 1.0
        out[:n] = targets
        return out

    def capture(self, feature_stacks, targets, max_events: Optional[int] = None, warmup: int = 2):
        """capture {zero_grad .. backward .. Adam} of one step over static copies of the batch into ONE hipGraph; ``replay(fs, targets)``
        copies a batch of the same shape (any number of events <= max_events) in and launches it.  Single process only (under data
        parallelism the obj / noobj counts and the gradients are all-reduced from inside the step: launch that eagerly)."""
        if self.world > 1:
            raise RuntimeError("ProposalTrainStep.capture: single-process only")
        if getattr(self.cfg, "grad_clip", None) is not None:
            raise RuntimeError("ProposalTrainStep.capture: cfg.grad_clip needs the gradient norm on the host (clip_grad_norm_), which a hipGraph "
                               "cannot capture; launch the step eagerly or set cfg.grad_clip = None")
        if self.reducer is None:       # static gradient buffers: what the captured kernels write and the captured Adam reads
            self.reducer = GradientReducer(self.params, overlap=False)
        max_events = max(int(max_events or 0), targets.shape[0])
        self._static_fs = {k: v.clone() for k, v in feature_stacks.items()}
        self._static_tg = self.pad_targets(targets, max_events)
        self._max_events = max_events
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self(self._static_fs, self._static_tg)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from . import ops as _ops
        with _ops.scratch_owner(id(self)), torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._static_out = self(self._static_fs, self._static_tg)
        _ops.finish_capture()
        self._graph = g
        self._graph_generation = _ops.weights_generation()
        return g

    def replay(self, feature_stacks=None, targets=None):
        if (feature_stacks is None) != (targets is None):
            raise ValueError("ProposalTrainStep.replay: give a batch as (feature_stacks, targets), or neither to replay the captured one")
        if feature_stacks is not None:
            for k, v in feature_stacks.items():
                self._static_fs[k].copy_(v, non_blocking=True)
            self._static_tg.copy_(self.pad_targets(targets, self._max_events), non_blocking=True)
        from . import ops as _ops
        if _ops.weights_generation() != self._graph_generation:
            raise RuntimeError("the weight-plane registry changed after capture(): call capture() again")
        self._graph.replay()
        _ops_weights_changed()
        return self._static_out

    def __call__(self, feature_stacks, targets):
        model = self.model
        model.train()
        from . import ops as _ops
        _ops.mark_step_start()          # (the refresh of the weights' operand planes forks from here: ops.EARLY_REFRESH)
        try:
            if self.reducer is not None:
                self.reducer.zero_grad()
            else:
                self.optimizer.zero_grad()
            masks = make_masks(feature_stacks, None, self.modality, self.pad_idx)
            _ops.allow_encoder_streams(self.reducer is None or self.reducer.world == 1 or not self.reducer.overlap)
            predictions, loss, losses_A, losses_V = model(feature_stacks, targets, masks)
            if not hasattr(self, "_one") or self._one.device != loss.device:
                self._one = torch.ones((), device=loss.device, dtype=torch.float32)
            loss.backward(gradient=self._one if loss.dim() == 0 and loss.dtype == torch.float32 else None)      # (no fill kernel for the root gradient)
            _ops.join_side_stream()
        finally:
            _ops.clear_step_start()
        if self.reducer is not None:
            self.reducer.finish()
        if getattr(self.cfg, "grad_clip", None) is not None:
            clip_grad_norm_(self.params, self.cfg.grad_clip)
        self.optimizer.step()
        loss = loss.detach()
        if self.world > 1:
            loss = global_sum(loss)          # the full-batch loss, for logging
        return predictions, loss, losses_A, losses_V


class MixedTrainStep:
    """BASELINE configs[4]: alternating train_cap / train_prop steps of ONE job.  The proposal generator works on top of the
    captioning model's bi-modal encoder -- the reference hands that encoder over through a checkpoint
    (model/proposal_generator.py:344-353, ``pretrained_cap_model_path`` with ``finetune_cap_encoder=False``); here the two models
    share the encoder MODULE, so every train_prop step sees the encoder as the latest train_cap step left it, frozen on the
    proposal side (its parameters belong to the captioning optimizer only).  One call = one train_cap step (captured graphs if
    ``capture`` was called) followed by one train_prop step (eager: the number of target events changes per batch)."""

    def __init__(self, cap_model, prop_model, cfg_cap, cfg_prop, pad_idx: int, data_parallel: bool = False,
                 bucket_bytes: int = 32 << 20, seed: Optional[int] = None):
        prop_model.encoder = cap_model.encoder
        self.cap = CaptioningTrainStep(cap_model, cfg_cap, pad_idx, data_parallel=data_parallel, static_grads=True,
                                       bucket_bytes=bucket_bytes, seed=seed)
        enc = {id(p) for p in cap_model.encoder.parameters()}
        self._enc_flags = [(p, p.requires_grad) for p in cap_model.encoder.parameters()]
        for p in cap_model.encoder.parameters():      # frozen while the proposal step's parameter list is built
            p.requires_grad = False
        try:
            # (the captioning step above seeded the dropout stream both steps draw from: the proposal step must not re-seed it)
            self.prop = ProposalTrainStep(prop_model, cfg_prop, pad_idx, data_parallel=data_parallel, bucket_bytes=bucket_bytes,
                                          keep_seed=True)
        finally:
            for p, f in self._enc_flags:
                p.requires_grad = f
        assert not any(id(p) in enc for p in self.prop.params)

    def capture(self, feature_stacks, caption_idx, warmup: int = 2, collectives: bool = False):
        return self.cap.capture(feature_stacks, caption_idx, warmup=warmup, collectives=collectives)

    def __call__(self, cap_batch, prop_batch):
        """cap_batch = (feature_stacks, caption_idx) or None to replay the captured batch; prop_batch = (feature_stacks, targets).
        Returns (captioning loss, proposal loss)."""
        if self.cap._graphs is not None:
            cap_loss, _ = self.cap.replay(*(cap_batch or (None, None)))
        else:
            cap_loss, _ = self.cap(*cap_batch)
        for p, _ in self._enc_flags:                  # no autograd graph through the shared encoder on the proposal side
            p.requires_grad = False
        try:
            _, prop_loss, _, _ = self.prop(*prop_batch)
        finally:
            for p, f in self._enc_flags:
                p.requires_grad = f
        return cap_loss, prop_loss
